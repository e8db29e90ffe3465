// Diagnostic: sustained rate and clock of v_mfma_f32_32x32x2_f32 on this device (what "peak" means in practice).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// variant 0: registers only; variant 1: operands re-read from LDS every 4 MFMAs (ds_read_b128).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int VARIANT>
__global__ __launch_bounds__(256) void k_peak(float* out, unsigned long long* stamps, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * (float)((i * 37) % 101 - 50);
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = seed * (threadIdx.x % 7 - 3), b = seed * (threadIdx.x % 5 - 2);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 1) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x * 4) + it * 16) & 4092]);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x * 4) + it * 48 + 2048) & 4092]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s], vb[(s + j) & 3], acc[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V>
void run(int blocks, int iters, const char* name) {
    float* out; unsigned long long* st;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_peak<V>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1e-3f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        double clk = 0; for (int b = 0; b < blocks; ++b) clk += (double)h[2 * b] / (double)h[2 * b + 1] * 100e6; clk /= blocks;
        const double flops = (double)blocks * 4 * iters * 16 * 4096.0;
        printf("%-28s blocks %4d  %.3f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz  -> peak at that clock %.1f TF\n", name, blocks, ms,
               flops / ms / 1e9, clk / 1e9, 256 * 4 * 64.0 * clk / 1e12);
    }
    hipFree(out); hipFree(st);
}
int main() {
    run<0>(256, 200000, "regs only, 1 wave/SIMD");
    run<0>(512, 100000, "regs only, 2 waves/SIMD");
    run<1>(256, 200000, "LDS operands, 1 wave/SIMD");
    run<1>(512, 100000, "LDS operands, 2 waves/SIMD");
    return 0;
}
