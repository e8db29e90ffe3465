// Fused BatchNorm (+ residual) + ReLU of the residual blocks, training mode, channels-last rows (M = B*P, C).
//
// Replaces the torch builtins the reference strings together inside BasicIcoS2SDownBlock / UpBlock
// (models.py:36-40,58-62):   relu(bn(x))   and   relu(bn_a(a) + bn_b(b)).
// All kernels are HBM-bound streaming passes; sums are reduced in two deterministic stages (row chunks -> finalize).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "icn_launch.h"

// (see icn_kernels.hip: wave priority of the kernels on the backward pass's critical chain)
#ifndef ICN_CHAIN_PRIO
#define ICN_CHAIN_PRIO 0
#endif
#if ICN_CHAIN_PRIO > 0
#define ICN_CHAIN_SETPRIO() __builtin_amdgcn_s_setprio(ICN_CHAIN_PRIO)
#else
#define ICN_CHAIN_SETPRIO() ((void)0)
#endif

namespace icn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 ldv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stv(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }

#ifndef ICN_BN_MAX_CHUNKS
#define ICN_BN_MAX_CHUNKS 512
#endif
constexpr int BN_MAX_CHUNKS = ICN_BN_MAX_CHUNKS;
// (the finalize kernels walk the chunks 16 at a time -- chunk_sums: slice = threadIdx.x / 16 -- and any count works; the workspace
//  is sized by bn_chunks(), which is capped by this constant, so it only has to be positive)
static_assert(BN_MAX_CHUNKS >= 1 && BN_MAX_CHUNKS <= 65536, "ICN_BN_MAX_CHUNKS");

// Sums of the BatchNorm passes are accumulated in DOUBLE from the first add (round 3), as the CPU reference does (torch's
// acc_type<float> on the CPU is double; its device kernels sum in fp32).  A batch statistic enters every element of its channel,
// so its rounding error is a COHERENT perturbation of the tensor, unlike the element-wise rounding of the convolutions, and the
// next cancelling sum (a BatchNorm backward's sum g, sum g * xhat, where |sum g| is 10 - 400 x smaller than sum |g| in this
// network) amplifies exactly that kind of error.  The kernels are HBM-bound: the fp64 adds cost nothing measurable (same
// throughput in the bench).  (What this does NOT remove: gradients of this network are discontinuous at ReLU pre-activations
// within rounding of zero; see tests/test_gpu_training_parity.py for how the parity tests deal with that.)
using f64x4 = __attribute__((ext_vector_type(4))) double;
__device__ __forceinline__ f64x4 to_d(f32x4 v) { return f64x4{(double)v[0], (double)v[1], (double)v[2], (double)v[3]}; }

// pre-activation of the fused op for 4 channels: bn_a(a) [+ bn_b(b)]; scale = invstd * gamma.  Forward and both backward
// passes go through this one function, so the ReLU mask the backward recomputes is bit-identical to the forward's.
template <int DUAL>
__device__ __forceinline__ f32x4 bn_pre4(f32x4 a, f32x4 mean_a, f32x4 scale_a, f32x4 beta_a, f32x4 b, f32x4 mean_b, f32x4 scale_b,
                                         f32x4 beta_b) {
    f32x4 v = (a - mean_a) * scale_a + beta_a;
    if (DUAL) v += (b - mean_b) * scale_b + beta_b;
    return v;
}

// partial[chunk][k][C], k < NS: per-channel sums of NS quantities over the chunk's rows.
//   MODE 0 (forward stats of x):                 k0 = sum (x - x0), k1 = sum (x - x0)^2       x0 = row 0 of x (per channel)
//          The shift keeps var = E[(x-x0)^2] - E[x-x0]^2 free of cancellation when |mean| >> std (torch uses Welford;
//          a plain E[x^2] - mean^2 in fp32 loses the variance of e.g. x = 100 + randn).
//   MODE 3 (forward stats of two tensors a, b in one pass: the two inputs of a residual BatchNorm): k0, k1 as MODE 0 for
//          p0 = a, k2, k3 the same for p2 = b
//   MODE 1 (backward of relu(bn(x))):            k0 = sum g,        k1 = sum g * xhat         g = dy * (v > 0)
//   MODE 2 (backward of relu(bn_a(a)+bn_b(b))):  k0 = sum g,        k1 = sum g * ahat,  k2 = sum g * bhat
//          v = the pre-activation, RECOMPUTED from a (and b) with the forward's own expression (bn_pre4) instead of reading
//          the saved output y: one tensor less to stream in each of the two backward passes.
#ifndef ICN_BN_UN
#define ICN_BN_UN 4
#endif
#ifndef ICN_BN_UN_BWD
#define ICN_BN_UN_BWD 2
#endif
#ifndef ICN_BN_WAVES
#define ICN_BN_WAVES 1
#endif
static_assert(ICN_BN_WAVES >= 1 && ICN_BN_WAVES <= 8, "ICN_BN_WAVES: minimum workgroups per CU of k_bn_partial's launch bounds");
template <int MODE>
__global__ __launch_bounds__(256, ICN_BN_WAVES) void k_bn_partial(const float* __restrict__ p0, const float* __restrict__ p2,
                                                     const float* __restrict__ p3, const float* __restrict__ stat_a,
                                                     const float* __restrict__ stat_b, const float* __restrict__ ga,
                                                     const float* __restrict__ ba, const float* __restrict__ gb,
                                                     const float* __restrict__ bb, double* __restrict__ partial, int M, int C,
                                                     int rows_per_chunk) {
    ICN_CHAIN_SETPRIO();
    constexpr int NS = MODE == 3 ? 4 : (MODE == 2 ? 3 : 2);
    __shared__ f64x4 red[NS][256];
    const int c4n = C / 4, stripes = 256 / c4n;
    const int cq = threadIdx.x % c4n, stripe = threadIdx.x / c4n, c = cq * 4;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    // UN rows per iteration, each with its own accumulators (combined in a fixed order below): the loads of an iteration are
    // independent, so a wave keeps UN x (1..3) 16-byte loads in flight -- with 512 blocks on 256 CUs the kernel is bound by
    // loads in flight, not by HBM (2.9 TB/s with one row per iteration against the 5.9 TB/s of the apply passes).
    // Backward modes: 2 rows per iteration.  They run beside the weight-gradient kernels of the second stream, whose
    // two waves per SIMD leave 176 registers: with 4 rows the backward forms need 128 / 196 and find no slot until a weight-gradient
    // workgroup retires; with 2 rows (78 / 122) one wave per SIMD fits at once.  Measured (gpurun_out/r4_i_*, 4 alternations):
    // 8.236 against 8.288 ms per step overlapped; alone on the chip the 2-row form is the slower one (8.606 against 8.559 with
    // every pass on 2 rows), which is why the forward modes, which never have company, keep 4.
    constexpr int UN = (MODE == 1 || MODE == 2) ? ICN_BN_UN_BWD : ICN_BN_UN;
    static_assert(UN == 1 || UN == 2 || UN == 4, "ICN_BN_UN / ICN_BN_UN_BWD: the accumulator combine below is written for 1, 2 or 4 rows per iteration");
    f64x4 s[UN][NS];
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int k = 0; k < NS; ++k) s[u][k] = f64x4{0.0, 0.0, 0.0, 0.0};
    f32x4 mean_a{}, inv_a{}, mean_b{}, inv_b{}, sc_a{}, sc_b{}, be_a{}, be_b{};
    if (MODE == 1 || MODE == 2) { mean_a = ldv(stat_a + c); inv_a = ldv(stat_a + C + c); sc_a = inv_a * ldv(ga + c); be_a = ldv(ba + c); }
    if (MODE == 2) { mean_b = ldv(stat_b + c); inv_b = ldv(stat_b + C + c); sc_b = inv_b * ldv(gb + c); be_b = ldv(bb + c); }
    f32x4 shift{}, shift_b{};
    if (MODE == 0 || MODE == 3) shift = ldv(p0 + c);
    if (MODE == 3) shift_b = ldv(p2 + c);
    if (stripe < stripes) {
        for (int r = r0 + stripe; r < r1; r += UN * stripes) {
            f32x4 x0[UN], x2[UN], x3[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int ru = r + u * stripes;
                const size_t o = (size_t)min(ru, r1 - 1) * C + c;      // (clamped: rows past the chunk are loaded but not added)
                x0[u] = ldv(p0 + o);
                if (MODE >= 1) x2[u] = ldv(p2 + o);
                if (MODE == 2) x3[u] = ldv(p3 + o);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (r + u * stripes >= r1) continue;
                if (MODE == 0 || MODE == 3) {
                    const f64x4 x = to_d(x0[u] - shift);
                    s[u][0] += x;
                    s[u][1] += x * x;
                    if (MODE == 3) {
                        const f64x4 xb = to_d(x2[u] - shift_b);
                        s[u][NS - 2] += xb;
                        s[u][NS - 1] += xb * xb;
                    }
                } else {
                    // p0 = dy, p2 = x / a, p3 = b
                    const f32x4 dy = x0[u], av = x2[u];
                    f32x4 bv{};
                    if (MODE == 2) bv = x3[u];
                    const f32x4 v = bn_pre4<MODE == 2>(av, mean_a, sc_a, be_a, bv, mean_b, sc_b, be_b);
                    f32x4 g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = v[j] > 0.f ? dy[j] : 0.f;
                    s[u][0] += to_d(g);
                    s[u][1] += to_d(g) * to_d((av - mean_a) * inv_a);
                    if (MODE == 2) s[u][2] += to_d(g) * to_d((bv - mean_b) * inv_b);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (UN == 4) s[0][k] = (s[0][k] + s[1][k]) + (s[2][k] + s[3][k]);
        else if (UN == 2) s[0][k] = s[0][k] + s[1][k];
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) red[k][threadIdx.x] = s[0][k];
    __syncthreads();
    if (stripe == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            f64x4 t = red[k][cq];
            for (int q = 1; q < stripes; ++q) t += red[k][q * c4n + cq];
            *reinterpret_cast<f64x4*>(partial + ((size_t)blockIdx.x * NS + k) * C + c) = t;
        }
    }
}

// sums over the chunks of partial[chunk][k0 + q][C], q < NQ, for 16 channels per block: 16 slices of the chunk range per
// channel, combined through LDS in a fixed order (deterministic).  out[] is valid in the threads with slice == 0.
// A thread's loads of one round (8 chunks x NQ quantities) are all issued before the first add: 512 chunks are 4 rounds of
// memory latency (the first version walked them in 8 rounds of 4 loads, one quantity after the other).
template <int NQ>
__device__ __forceinline__ void chunk_sums(const double* __restrict__ partial, int chunks, int NS, int k0, int C, int c, int slice,
                                           double (*red)[16], double* out) {
    double s[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[q][e] = 0.0;
    if (c < C) {
        const double* p = partial + (size_t)k0 * C + c;
        const size_t st = (size_t)NS * C;
        for (int j = slice; j < chunks; j += 128) {
            double v[NQ][8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) v[q][u] = j + 16 * u < chunks ? p[(size_t)(j + 16 * u) * st + (size_t)q * C] : 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) s[q][u & 3] += v[q][u];
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        red[slice][threadIdx.x % 16] = (s[q][0] + s[q][1]) + (s[q][2] + s[q][3]);
        __syncthreads();
        double t = 0.0;
        if (slice == 0)
            for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x % 16];
        __syncthreads();
        out[q] = t;
    }
}

// forward finalize: mean / invstd of the batch, running statistics (momentum, unbiased variance); 16 channels per block
__global__ __launch_bounds__(256) void k_bn_finalize_fwd(const float* __restrict__ x, const double* __restrict__ partial, int chunks,
                                                          int M, int C, float eps, float momentum,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ stat) {
    __shared__ double red[16][16];
    const int c = blockIdx.x * 16 + threadIdx.x % 16, slice = threadIdx.x / 16;
    double sums[2];
    chunk_sums<2>(partial, chunks, 2, 0, C, c, slice, red, sums);
    const double s = sums[0], s2 = sums[1];
    if (slice != 0 || c >= C) return;
    const double ms = s / M;                              // mean of x - x0 (x0 = first row: the shift of k_bn_partial<0>)
    const double mean = (double)x[c] + ms;
    double var = s2 / M - ms * ms;
    if (var < 0.0) var = 0.0;
    stat[c] = (float)mean;
    stat[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * M / (M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// the same for the two inputs of a residual BatchNorm after a MODE 3 partial pass (4 sums per chunk); blockIdx.y = input
struct BnFin { const float* x; float eps, momentum; float* running_mean; float* running_var; float* stat; };
__global__ __launch_bounds__(256) void k_bn_finalize_fwd2(const BnFin fa, const BnFin fb, const double* __restrict__ partial, int chunks,
                                                           int M, int C) {
    __shared__ double red[16][16];
    const BnFin f = blockIdx.y ? fb : fa;
    const int c = blockIdx.x * 16 + threadIdx.x % 16, slice = threadIdx.x / 16;
    double sums[2];
    chunk_sums<2>(partial, chunks, 4, 2 * blockIdx.y, C, c, slice, red, sums);
    if (slice != 0 || c >= C) return;
    const double ms = sums[0] / M;
    const double mean = (double)f.x[c] + ms;
    double var = sums[1] / M - ms * ms;
    if (var < 0.0) var = 0.0;
    f.stat[c] = (float)mean;
    f.stat[C + c] = (float)(1.0 / sqrt(var + (double)f.eps));
    if (f.running_mean) {
        const double unbiased = M > 1 ? var * M / (M - 1) : var;
        f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
    }
}

// backward finalize: sums[k][C] = sum over chunks (k < NS); blockIdx.y = k.  The same values also go to the parameter
// gradients' own tensors (k = 0: dbeta of bn_a and, for the residual form, of bn_b; 1: dgamma_a; 2: dgamma_b), which may be
// views into a DistributedDataParallel bucket (geniconet_amd/_gradbuf.py) -- `sums` stays the contiguous copy the apply pass reads.
struct BnGradOut { float* dbeta_a; float* dbeta_b; float* dgamma_a; float* dgamma_b; };
__global__ __launch_bounds__(256) void k_bn_finalize_bwd(const double* __restrict__ partial, int chunks, int C, int NS,
                                                          float* __restrict__ sums, const BnGradOut out) {
    __shared__ double red[16][16];
    const int c = blockIdx.x * 16 + threadIdx.x % 16, slice = threadIdx.x / 16, k = blockIdx.y;
    double s;
    chunk_sums<1>(partial, chunks, NS, k, C, c, slice, red, &s);
    if (slice == 0 && c < C) {
        const float v = (float)s;
        sums[k * C + c] = v;
        if (k == 0) {
            if (out.dbeta_a) out.dbeta_a[c] = v;
            if (out.dbeta_b) out.dbeta_b[c] = v;
        } else if (k == 1) {
            if (out.dgamma_a) out.dgamma_a[c] = v;
        } else if (out.dgamma_b) {
            out.dgamma_b[c] = v;
        }
    }
}

// y = relu(bn_a(a) [+ bn_b(b)])
// APPLY_UNR elements per thread and iteration, every load of an iteration issued before the first use: a wave then keeps
// APPLY_UNR x (1..3) 16-byte loads in flight, so the pass stays near HBM speed when only one or two waves per SIMD find room
// beside the weight-gradient kernels of the second stream, not only on an empty chip (7 - 8 waves per SIMD).
#ifndef ICN_BN_APPLY_UNR
#define ICN_BN_APPLY_UNR 1
#endif
constexpr int APPLY_UNR = ICN_BN_APPLY_UNR;
static_assert(APPLY_UNR >= 1 && APPLY_UNR <= 8, "ICN_BN_APPLY_UNR: elements per thread and iteration of the apply passes");

template <int DUAL>
__global__ void k_bn_relu_fwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ stat_a,
                              const float* __restrict__ stat_b, const float* __restrict__ ga, const float* __restrict__ ba,
                              const float* __restrict__ gb, const float* __restrict__ bb, float* __restrict__ y, size_t total4,
                              int C) {
    ICN_CHAIN_SETPRIO();
    const size_t T = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total4; i0 += APPLY_UNR * T) {
        f32x4 av[APPLY_UNR], bv[APPLY_UNR];
#pragma unroll
        for (int u = 0; u < APPLY_UNR; ++u) {
            const size_t i = i0 + u * T < total4 ? i0 + u * T : i0;      // (clamped: loaded, not stored)
            av[u] = ldv(a + i * 4);
            if (DUAL) bv[u] = ldv(b + i * 4);
        }
#pragma unroll
        for (int u = 0; u < APPLY_UNR; ++u) {
            const size_t i = i0 + u * T;
            if (i >= total4) break;
            const int c = (int)((i * 4) % C);
            f32x4 bu{}, mb{}, sb{}, bbv{};
            if (DUAL) { bu = bv[u]; mb = ldv(stat_b + c); sb = ldv(stat_b + C + c) * ldv(gb + c); bbv = ldv(bb + c); }
            const f32x4 v = bn_pre4<DUAL>(av[u], ldv(stat_a + c), ldv(stat_a + C + c) * ldv(ga + c), ldv(ba + c), bu, mb, sb, bbv);
            stv(y + i * 4, relu4(v));
        }
    }
}

// dx = gamma * invstd * (g - sum_g / M - xhat * sum_gx / M),   g = dy * (v > 0), v recomputed (bn_pre4)   (same for b when DUAL)
template <int DUAL>
__global__ void k_bn_relu_bwd(const float* __restrict__ dy, const float* __restrict__ a, const float* __restrict__ b,
                              const float* __restrict__ stat_a, const float* __restrict__ stat_b, const float* __restrict__ ga,
                              const float* __restrict__ ba, const float* __restrict__ gb, const float* __restrict__ bb,
                              const float* __restrict__ sums, float* __restrict__ da, float* __restrict__ db, size_t total4, int C,
                              float inv_m) {
    ICN_CHAIN_SETPRIO();
    const size_t T = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total4; i0 += APPLY_UNR * T) {
        f32x4 dv[APPLY_UNR], avv[APPLY_UNR], bvv[APPLY_UNR];
#pragma unroll
        for (int u = 0; u < APPLY_UNR; ++u) {
            const size_t i = i0 + u * T < total4 ? i0 + u * T : i0;      // (clamped: loaded, not stored)
            dv[u] = ldv(dy + i * 4);
            avv[u] = ldv(a + i * 4);
            if (DUAL) bvv[u] = ldv(b + i * 4);
        }
#pragma unroll
        for (int u = 0; u < APPLY_UNR; ++u) {
            const size_t i = i0 + u * T;
            if (i >= total4) break;
            const int c = (int)((i * 4) % C);
            const f32x4 d = dv[u], av = avv[u];
            const f32x4 mean_a = ldv(stat_a + c), inv_a = ldv(stat_a + C + c), gam_a = ldv(ga + c);
            f32x4 bv{}, mean_b{}, inv_b{}, gam_b{}, be_b{};
            if (DUAL) { bv = bvv[u]; mean_b = ldv(stat_b + c); inv_b = ldv(stat_b + C + c); gam_b = ldv(gb + c); be_b = ldv(bb + c); }
            const f32x4 v = bn_pre4<DUAL>(av, mean_a, inv_a * gam_a, ldv(ba + c), bv, mean_b, inv_b * gam_b, be_b);
            f32x4 g;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = v[j] > 0.f ? d[j] : 0.f;
            const f32x4 sg = ldv(sums + c) * inv_m;
            {
                const f32x4 xh = (av - mean_a) * inv_a;
                stv(da + i * 4, gam_a * inv_a * (g - sg - xh * (ldv(sums + C + c) * inv_m)));
            }
            if (DUAL) {
                const f32x4 xh = (bv - mean_b) * inv_b;
                stv(db + i * 4, gam_b * inv_b * (g - sg - xh * (ldv(sums + 2 * C + c) * inv_m)));
            }
        }
    }
}

bool bn_supported(int C) { return C >= 4 && C <= 1024 && C % 4 == 0 && 256 % (C / 4) == 0; }
int bn_chunks(int M) { return std::min(BN_MAX_CHUNKS, (M + 63) / 64); }

static int stream_blocks(size_t total4) { return (int)std::min((size_t)8192, (total4 + 256 * APPLY_UNR - 1) / (256 * APPLY_UNR)); }

void launch_bn_stats(const float* x, int M, int C, float eps, float momentum, float* running_mean, float* running_var, float* stat,
                     float* ws, hipStream_t s) {
    const int chunks = bn_chunks(M), rows = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(k_bn_partial<0>, dim3(chunks), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, reinterpret_cast<double*>(ws), M, C, rows);
    hipLaunchKernelGGL(k_bn_finalize_fwd, dim3((C + 15) / 16), dim3(256), 0, s, x, reinterpret_cast<const double*>(ws), chunks, M, C, eps, momentum, running_mean,
                       running_var, stat);
}

void launch_bn_stats2(const float* a, const float* b, int M, int C, float eps_a, float mom_a, float* rm_a, float* rv_a, float* stat_a,
                      float eps_b, float mom_b, float* rm_b, float* rv_b, float* stat_b, float* ws, hipStream_t s) {
    const int chunks = bn_chunks(M), rows = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(k_bn_partial<3>, dim3(chunks), dim3(256), 0, s, a, b, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, reinterpret_cast<double*>(ws), M, C, rows);
    hipLaunchKernelGGL(k_bn_finalize_fwd2, dim3((C + 15) / 16, 2), dim3(256), 0, s, BnFin{a, eps_a, mom_a, rm_a, rv_a, stat_a},
                       BnFin{b, eps_b, mom_b, rm_b, rv_b, stat_b}, reinterpret_cast<const double*>(ws), chunks, M, C);
}

void launch_bn_relu_fwd(const float* a, const float* b, const float* stat_a, const float* stat_b, const float* ga, const float* ba,
                        const float* gb, const float* bb, float* y, int M, int C, hipStream_t s) {
    const size_t total4 = (size_t)M * C / 4;
    if (b) hipLaunchKernelGGL(k_bn_relu_fwd<1>, dim3(stream_blocks(total4)), dim3(256), 0, s, a, b, stat_a, stat_b, ga, ba, gb, bb, y, total4, C);
    else hipLaunchKernelGGL(k_bn_relu_fwd<0>, dim3(stream_blocks(total4)), dim3(256), 0, s, a, b, stat_a, stat_b, ga, ba, gb, bb, y, total4, C);
}

void launch_bn_relu_bwd(const float* dy, const float* a, const float* b, const float* stat_a, const float* stat_b, const float* ga,
                        const float* ba, const float* gb, const float* bb, float* da, float* db, float* sums, float* ws, int M, int C,
                        float* dbeta_a, float* dgamma_a, float* dbeta_b, float* dgamma_b, hipStream_t s) {
    const int chunks = bn_chunks(M), rows = (M + chunks - 1) / chunks;
    const int NS = b ? 3 : 2;
    double* wsd = reinterpret_cast<double*>(ws);
    if (b) hipLaunchKernelGGL(k_bn_partial<2>, dim3(chunks), dim3(256), 0, s, dy, a, b, stat_a, stat_b, ga, ba, gb, bb, wsd, M, C, rows);
    else hipLaunchKernelGGL(k_bn_partial<1>, dim3(chunks), dim3(256), 0, s, dy, a, b, stat_a, stat_b, ga, ba, gb, bb, wsd, M, C, rows);
    hipLaunchKernelGGL(k_bn_finalize_bwd, dim3((C + 15) / 16, NS), dim3(256), 0, s, wsd, chunks, C, NS, sums,
                       BnGradOut{dbeta_a, b ? dbeta_b : nullptr, dgamma_a, b ? dgamma_b : nullptr});
    const size_t total4 = (size_t)M * C / 4;
    if (b) hipLaunchKernelGGL(k_bn_relu_bwd<1>, dim3(stream_blocks(total4)), dim3(256), 0, s, dy, a, b, stat_a, stat_b, ga, ba, gb, bb, sums, da, db, total4, C, 1.f / M);
    else hipLaunchKernelGGL(k_bn_relu_bwd<0>, dim3(stream_blocks(total4)), dim3(256), 0, s, dy, a, b, stat_a, stat_b, ga, ba, gb, bb, sums, da, db, total4, C, 1.f / M);
}

}  // namespace icn

// ---------------------------------------------------------------------------------------------------------
// Fused head:  y = tanh(x @ W^T + b),  x (M, Cin) channels-last rows, W (Cout, Cin), Cout <= 4
// (reference models.py:151-154: Conv2d(64, 3, kernel_size=1) + Tanh).  HBM-bound: Cin/4 lanes share a row
// (one float4 each, coalesced), partial dot products are combined with wave shuffles.
// ---------------------------------------------------------------------------------------------------------
namespace icn {

constexpr int HEAD_MAX_OUT = 4;

template <int LPR>   // lanes per row = Cin / 4 (power of two, <= 64)
__global__ __launch_bounds__(256) void k_head_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y, int M, int Cout) {
    constexpr int Cin = LPR * 4, RPB = 256 / LPR;         // rows per block pass
    const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
    f32x4 wv[HEAD_MAX_OUT];
#pragma unroll
    for (int o = 0; o < HEAD_MAX_OUT; ++o) wv[o] = o < Cout ? ldv(w + o * Cin + 4 * sub) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = blockIdx.x * RPB + rloc; r < M; r += gridDim.x * RPB) {
        const f32x4 xv = ldv(x + (size_t)r * Cin + 4 * sub);
        float acc[HEAD_MAX_OUT];
#pragma unroll
        for (int o = 0; o < HEAD_MAX_OUT; ++o) acc[o] = xv[0] * wv[o][0] + xv[1] * wv[o][1] + xv[2] * wv[o][2] + xv[3] * wv[o][3];
#pragma unroll
        for (int d = LPR / 2; d >= 1; d >>= 1)
#pragma unroll
            for (int o = 0; o < HEAD_MAX_OUT; ++o) acc[o] += __shfl_xor(acc[o], d, 64);
        if (sub < Cout) {
            float v = acc[0];
#pragma unroll
            for (int o = 1; o < HEAD_MAX_OUT; ++o) v = sub == o ? acc[o] : v;
            y[(size_t)r * Cout + sub] = tanhf(v + bias[sub]);
        }
    }
}

// g = dy * (1 - y^2);  dx = g @ W;  partial[chunk][o][Cin] = sum_rows g[o] * x,  partial_b[chunk][o] = sum_rows g[o]
template <int LPR>
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                   const float* __restrict__ x, const float* __restrict__ w,
                                                   float* __restrict__ dx, float* __restrict__ partial, int M, int Cout,
                                                   int rows_per_block) {
    constexpr int Cin = LPR * 4, RPB = 256 / LPR;
    __shared__ f32x4 red[HEAD_MAX_OUT][256];
    __shared__ float redb[HEAD_MAX_OUT][256];
    const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
    f32x4 wv[HEAD_MAX_OUT], dwv[HEAD_MAX_OUT];
    float dbv[HEAD_MAX_OUT];
#pragma unroll
    for (int o = 0; o < HEAD_MAX_OUT; ++o) {
        wv[o] = o < Cout ? ldv(w + o * Cin + 4 * sub) : f32x4{0.f, 0.f, 0.f, 0.f};
        dwv[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        dbv[o] = 0.f;
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int r = r0 + rloc; r < r1; r += RPB) {
        const f32x4 xv = ldv(x + (size_t)r * Cin + 4 * sub);
        f32x4 dxa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < HEAD_MAX_OUT; ++o) {
            if (o < Cout) {
                const float yy = y[(size_t)r * Cout + o];
                const float g = dy[(size_t)r * Cout + o] * (1.f - yy * yy);
                dxa += g * wv[o];
                dwv[o] += g * xv;
                dbv[o] += g;
            }
        }
        if (dx) stv(dx + (size_t)r * Cin + 4 * sub, dxa);
    }
#pragma unroll
    for (int o = 0; o < HEAD_MAX_OUT; ++o) { red[o][threadIdx.x] = dwv[o]; redb[o][threadIdx.x] = dbv[o]; }
    __syncthreads();
    if (rloc == 0) {
        for (int o = 0; o < Cout; ++o) {
            f32x4 t = red[o][sub];
            for (int q = 1; q < RPB; ++q) t += red[o][q * LPR + sub];
            stv(partial + ((size_t)blockIdx.x * HEAD_MAX_OUT + o) * (Cin + 4) + 4 * sub, t);
            if (sub == 0) {
                float tb = 0.f;
                for (int q = 0; q < RPB; ++q) tb += redb[o][q * LPR];
                partial[((size_t)blockIdx.x * HEAD_MAX_OUT + o) * (Cin + 4) + Cin] = tb;
            }
        }
    }
}

// dw[o][c] = sum_chunks partial[chunk][o][c];  db[o] = sum_chunks partial[chunk][o][Cin]     (row stride Cin + 4)
// One wave per output element: lanes stride over the chunks (independent loads), then a fixed butterfly (deterministic).
__global__ __launch_bounds__(256) void k_head_reduce(const float* __restrict__ partial, int chunks, int Cin, int Cout,
                                                      float* __restrict__ dw, float* __restrict__ db) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= Cout * (Cin + 1)) return;
    const int o = i / (Cin + 1), c = i % (Cin + 1);
    double s = 0.0;
    for (int k = lane; k < chunks; k += 64) s += partial[((size_t)k * HEAD_MAX_OUT + o) * (Cin + 4) + c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) {
        if (c < Cin) dw[o * Cin + c] = (float)s;
        else db[o] = (float)s;
    }
}

bool head_supported(int Cin, int Cout) {
    return Cout >= 1 && Cout <= HEAD_MAX_OUT && (Cin == 16 || Cin == 32 || Cin == 64 || Cin == 128 || Cin == 256);
}
int head_chunks(int M) { return std::min(1024, (M + 63) / 64); }   // 4 blocks per CU; the reduce walks the chunks

#define ICN_HEAD_DISPATCH(KERNEL, ...)                                                  \
    switch (Cin) {                                                                      \
        case 16: hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__); break;                   \
        case 32: hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__); break;                   \
        case 64: hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__); break;                  \
        case 128: hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__); break;                 \
        default: hipLaunchKernelGGL((KERNEL<64>), __VA_ARGS__); break;                  \
    }

void launch_head_fwd(const float* x, const float* w, const float* bias, float* y, int M, int Cin, int Cout, hipStream_t s) {
    const int rpb = 256 / (Cin / 4);
    const int blocks = std::min(4096, (M + rpb - 1) / rpb);
    ICN_HEAD_DISPATCH(k_head_fwd, dim3(blocks), dim3(256), 0, s, x, w, bias, y, M, Cout)
}

void launch_head_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw, float* db, float* ws,
                     int M, int Cin, int Cout, hipStream_t s) {
    const int chunks = head_chunks(M), rows = (M + chunks - 1) / chunks;
    ICN_HEAD_DISPATCH(k_head_bwd, dim3(chunks), dim3(256), 0, s, dy, y, x, w, dx, ws, M, Cout, rows)
    hipLaunchKernelGGL(k_head_reduce, dim3((Cout * (Cin + 1) + 3) / 4), dim3(256), 0, s, ws, chunks, Cin, Cout, dw, db);
}
#undef ICN_HEAD_DISPATCH

}  // namespace icn
