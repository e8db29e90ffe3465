// Adam step of the training loop (reference run.py:253 `optimizer.step()` on the optimiser of run.py:446) as ONE launch over
// every parameter tensor.  HBM-bound: reads p, g, m, v and writes p, m, v once (28 bytes per parameter).
//
// torch's multi-tensor implementation runs ~10 kernels per step, each over 64 Ki-element chunks: the model's 4.6 M parameters
// make 70-odd blocks for 256 CUs, so every one of them is latency-bound (14-25 us each, 0.2 ms per step).  Here a block owns
// 2048 elements; the (tensor, chunk) of a block comes from a table passed by value in the kernel arguments.
//
// Arithmetic follows torch/optim/adam.py::_single_tensor_adam (no amsgrad, no maximize):
//   g' = g + wd * p;  m = m + (g' - m) * (1 - b1);  v = v * b2 + (1 - b2) * g' * g';
//   p = p - step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps),   step_size = lr / (1 - b1^t)
// with the per-tensor scalars computed on the host in double precision, as torch does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>
#include <vector>

#include "icn_launch.h"

namespace icn {

constexpr int ADAM_CHUNK = 2048;   // elements per block: 256 threads x 2 float4

struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    unsigned n[ADAM_MAX_TENSORS];
    unsigned first_block[ADAM_MAX_TENSORS + 1];   // prefix sums of the tensors' chunk counts
    int count;
    float step_size, bc2_sqrt;                    // one launch = tensors with the same step count
    const float* scal;                            // device copy of {step_size, bc2_sqrt} read instead of the two above when non-null:
                                                  // a launch recorded into a HIP graph must not carry the step's scalars in its
                                                  // arguments (icn_adam_step_dev)
};
static_assert(sizeof(AdamTable) <= 4000, "kernel arguments are limited to 4 KB");

// No FMA contraction here: the vector path (16-byte aligned tensors) and the scalar path (e.g. gradients that are views into a
// DistributedDataParallel bucket, which start at any 4-byte offset) must round identically, or the same training run differs in
// the last bit with and without DDP (seen: weights 1 ulp apart after the second step).  With contraction left to the compiler
// the two inlined copies were fused differently.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1c, float b2, float b2c, float eps, float wd,
                                         float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)
    if (wd != 0.f) g = g + wd * p;
    m = m + (g - m) * b1c;
    v = v * b2 + b2c * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam(const AdamTable t, float b1c, float b2, float b2c, float eps, float wd) {
    // tensor of this block: the table is wave-uniform, so this is a scalar binary search over <= 96 entries
    int lo = 0, hi = t.count - 1;
    const unsigned blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.first_block[mid] <= blk) lo = mid;
        else hi = mid - 1;
    }
    const unsigned n = t.n[lo], base = (blk - t.first_block[lo]) * ADAM_CHUNK;
    float* __restrict__ p = t.p[lo];
    const float* __restrict__ g = t.g[lo];
    float* __restrict__ m = t.m[lo];
    float* __restrict__ v = t.v[lo];
    const float ss = t.scal ? t.scal[0] : t.step_size, bc = t.scal ? t.scal[1] : t.bc2_sqrt;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
    for (int r = 0; r < ADAM_CHUNK / 1024; ++r) {
        const unsigned i = base + r * 1024 + threadIdx.x * 4;
        if (vec && i + 4 <= n) {
            float4 pp = *reinterpret_cast<const float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, b1c, b2, b2c, eps, wd, ss, bc);
            adam_one(pp.y, gg.y, mm.y, vv.y, b1c, b2, b2c, eps, wd, ss, bc);
            adam_one(pp.z, gg.z, mm.z, vv.z, b1c, b2, b2c, eps, wd, ss, bc);
            adam_one(pp.w, gg.w, mm.w, vv.w, b1c, b2, b2c, eps, wd, ss, bc);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (unsigned e = i; e < i + 4 && e < n; ++e) {
                float pp = p[e], mm = m[e], vv = v[e];
                adam_one(pp, g[e], mm, vv, b1c, b2, b2c, eps, wd, ss, bc);
                p[e] = pp;
                m[e] = mm;
                v[e] = vv;
            }
        }
    }
}

void launch_adam(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const size_t* numel,
                 const float* step_size, const float* bc2_sqrt, double beta1, double beta2, double eps, double weight_decay, hipStream_t s,
                 const float* scal_dev) {
    // one launch per group of tensors that share (step_size, bc2_sqrt), i.e. the step count -- normally all of them
    std::vector<char> done(count, 0);
    for (int lead = 0; lead < count; ++lead) {
        if (done[lead]) continue;
        AdamTable t{};
        t.step_size = step_size[lead];
        t.bc2_sqrt = bc2_sqrt[lead];
        t.scal = scal_dev;
        unsigned blocks = 0;
        auto flush = [&]() {
            t.first_block[t.count] = blocks;
            if (blocks)
                hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, t, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                                   (float)eps, (float)weight_decay);
            t.count = 0;
            blocks = 0;
        };
        for (int i = lead; i < count; ++i) {
            if (done[i] || step_size[i] != t.step_size || bc2_sqrt[i] != t.bc2_sqrt) continue;
            done[i] = 1;
            const size_t n = numel[i];
            if (n >= ((size_t)1 << 32) - ADAM_CHUNK) throw std::invalid_argument("icn_adam_step: tensor beyond 2^32 elements");
            if (n == 0) continue;
            const int k = t.count++;
            t.p[k] = p[i]; t.g[k] = g[i]; t.m[k] = m[i]; t.v[k] = v[i];
            t.n[k] = (unsigned)n;
            t.first_block[k] = blocks;
            blocks += (unsigned)((n + ADAM_CHUNK - 1) / ADAM_CHUNK);
            if (t.count == ADAM_MAX_TENSORS) flush();
        }
        flush();
    }
}

}  // namespace icn
