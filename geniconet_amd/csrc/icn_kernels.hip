// gfx950 (MI355X / CDNA4) kernels of the icosahedral hex-convolution hot path.
//
// Everything here works on channels-last chart tensors (B, P, C) fp32, P = 10*4^r pixels, and on the
// int32 index tables built by icn_geometry.cpp (codes: >=0 pixel, -1 nothing, -2/-3 pole mean).
//
//   k_gather_gemm   implicit GEMM  dst[m, :] = bias + sum_t (sum_e src[idx_t,e(m), :]) @ Wt[t]^T
//                   on v_mfma_f32_32x32x2_f32 (exact fp32).  Used by conv forward (E = 1) and by conv
//                   backward-data (transposed table, E <= 3).  The pad-exchange between charts, the pole
//                   mean and the duplicated tap at five-valent pixels are all folded into the gather that
//                   fills the LDS tile, so no padded copy of the activations ever exists in HBM.
//   k_wgrad         dW_t = X_t^T dY, split over pixel rows, MFMA, deterministic two-pass reduction.
//   k_spmm_ell      HBM-bound sparse row mix (upsample forward and its transpose).
//   k_*_generic     scalar fall-backs for channel counts the MFMA tiles do not cover (e.g. the 3->64 stem).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <stdexcept>
#include <type_traits>
#include <utility>

#include "icn_launch.h"
#include "icn_streamk.h"

// Wave priority of the kernels on the backward pass's critical chain (convolutions, BatchNorm passes): the weight gradients run
// BESIDE them on a second stream at the default priority 0, so on a SIMD that hosts both the chain's waves issue
// first and the weight gradients take the issue slots that are left.  ICN_CHAIN_PRIO = 0 compiles it out (A/B builds).
#ifndef ICN_CHAIN_PRIO
#define ICN_CHAIN_PRIO 0
#endif
#if ICN_CHAIN_PRIO > 0
#define ICN_CHAIN_SETPRIO() __builtin_amdgcn_s_setprio(ICN_CHAIN_PRIO)
#else
#define ICN_CHAIN_SETPRIO() ((void)0)
#endif

// Pricing switches for the persistent conv GEMM (tools/build_exp.sh, tools/price_conv_features.py; DESIGN 4.1 "what production pays
// over the ladder").  A build with -DICN_EXP=<bits> leaves one production-only feature of conv_dma_body out at a time; RESULTS ARE
// THEN WRONG -- only the launch time is read, exactly as profiles/r03_kstep_decomposition.txt did for the K-step.  0 (the default)
// is the product, and compiles to the same instructions as before the switches existed.
//   1 no tile epilogue (accumulators are dropped)          2 no per-tile metadata DMA + convert (every tile reuses the first tile's tables)
//   4 no side-row vote / second resource in the stage issue (also in the weight-gradient kernels)     8 no bias in the epilogue
//   16 epilogue arithmetic and store instructions kept, every offset out of range (no memory side)   32 (bf16x3 kernels) no staggered stage issue
//   64 / 256 epilogue stores with the
//   non-temporal (nt) / system-scope write-through (sc0 sc1) cache policy
#ifndef ICN_EXP
#define ICN_EXP 0
#endif
#ifndef ICN_CONV_WAVES_DEFAULT
#define ICN_CONV_WAVES_DEFAULT 4
#endif
#define ICN_EXP_STORE_POLICY ((ICN_EXP & 64) ? 2 : (ICN_EXP & 256) ? 17 : 0)

namespace icn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ int corner_pixel(int n, int k, int c) {
    return k == 0 ? (c * n) * 2 * n : ((c + 1) * n - 1) * 2 * n + (2 * n - 1);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// mean over the 5 corner pixels of pole k of sample b, 4 channels starting at `ch`
__device__ __forceinline__ f32x4 pole_mean4(const float* src, int b, int Ps, int ns, int k, int K, int ch) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 5; ++c) s += ld4(src + ((size_t)b * Ps + corner_pixel(ns, k, c)) * K + ch);
    return s * 0.2f;
}

// scalar gather of one channel (pixel, pole mean or nothing)
__device__ __forceinline__ float gather1(const float* src, int32_t code, int b, int Ps, int ns, int K, int ch) {
    if (code >= 0) return src[((size_t)b * Ps + code) * K + ch];
    if (code <= -2) {
        float s = 0.f;
        for (int c = 0; c < 5; ++c) s += src[((size_t)b * Ps + corner_pixel(ns, -2 - code, c)) * K + ch];
        return s * 0.2f;
    }
    return 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// weight re-layouts:  w[Cout][Cin][7]  ->  wf[t][Cout][Cin]  (forward B operand, k = ci contiguous)
//                                      ->  wb[t][Cin][Cout]  (backward-data B operand, k = co contiguous)
// ---------------------------------------------------------------------------------------------------------
// Prologue of a conv call, one launch: blocks [0, npack) repack the (Cout, Cin, 7) parameter(s) into the GEMM's B operand
// [7][N][K] (k-contiguous) and concatenate the biases; blocks [npack, npack + nsrc * B * n_slots) fill the side buffer(s)
// of a DmaTable:  side[b][s][:] = sum_e gather(slots[s][e])   (pixels / pole means of the source tensor; icn_geometry.h).
__global__ __launch_bounds__(256) void k_conv_prologue(PrologueArgs a, int npack) {
    if (a.zero && blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.n_zero; i += 256) a.zero[i] = 0;
    if ((int)blockIdx.x < npack) {
        const int Ct = a.Cout + a.Cout2, total = a.packed ? Ct * a.Cin * 7 : 0;
        if (a.packed_b3) {
            // ARITH = 1: the same B operand [T][N][K] as three planes of truncated bf16 pieces (w = w1 + w2 + w3, exact), stored block
            // by block (tap, k-chunk of 32, column tile of b3_bn) exactly as conv_dma_body's LDS stage wants it: [3 planes][b3_bn rows]
            // [64 B], 16-byte chunk q of row r at q ^ ((r >> 2) & 3) -- so the kernel's B DMA is a linear copy.  A thread packs 8
            // consecutive k of one row: one 16-byte store per plane.
            // (b3_flat_n: the dense forward GEMM of icn_upconv_fwd reads [7][Ct][Cin] as ONE tap with N = 7 * Ct columns)
            const int T_ = (a.transpose == 2 || a.b3_flat_n) ? 1 : 7, N_ = a.transpose == 0 ? (a.b3_flat_n ? 7 * Ct : Ct) : a.Cin;
            const int K_ = a.transpose == 0 ? a.Cin : (a.transpose == 1 ? Ct : 7 * Ct);
            const int kg = K_ / 8, nk_ = K_ / 32, bn = a.b3_bn, ntn_ = N_ / bn, groups = T_ * N_ * kg;
            for (int g = blockIdx.x * 256 + threadIdx.x; g < groups; g += npack * 256) {
                const int k8 = g % kg, n = (g / kg) % N_, tt = g / (kg * N_);
                unsigned pl[3][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k8 * 8 + e;
                    int t, co, ci;
                    if (a.transpose == 0) { t = a.b3_flat_n ? n / Ct : tt; co = a.b3_flat_n ? n - t * Ct : n; ci = k; }
                    else if (a.transpose == 1) { t = tt; ci = n; co = k; }
                    else { ci = n; t = k / Ct; co = k - t * Ct; }
                    const float x = co < a.Cout ? a.w[((size_t)co * a.Cin + ci) * 7 + t] : a.w2[((size_t)(co - a.Cout) * a.Cin + ci) * 7 + t];
                    const unsigned u1 = __float_as_uint(x) & 0xffff0000u;
                    const float r = x - __uint_as_float(u1);
                    const unsigned u2 = __float_as_uint(r) & 0xffff0000u;
                    const unsigned u3 = __float_as_uint(r - __uint_as_float(u2));
                    const int sh = 16 * (e & 1);
                    pl[0][e >> 1] |= (u1 >> 16) << sh;
                    pl[1][e >> 1] |= (u2 >> 16) << sh;
                    pl[2][e >> 1] |= (u3 >> 16) << sh;
                }
                const int kc = k8 >> 2, q = k8 & 3, tn = n / bn, nl = n - tn * bn;
                char* blk = reinterpret_cast<char*>(a.packed_b3) + ((size_t)(tt * nk_ + kc) * ntn_ + tn) * ((size_t)bn * 192);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    *reinterpret_cast<u32x4_t*>(blk + (size_t)pp * bn * 64 + nl * 64 + ((q ^ ((nl >> 2) & 3)) * 16)) =
                        u32x4_t{pl[pp][0], pl[pp][1], pl[pp][2], pl[pp][3]};
            }
        }
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += npack * 256) {
            // i enumerates the OUTPUT layout so that stores are coalesced
            int t, co, ci;
            if (a.transpose == 0) { ci = i % a.Cin; co = (i / a.Cin) % Ct; t = i / (a.Cin * Ct); }
            else if (a.transpose == 1) { co = i % Ct; ci = (i / Ct) % a.Cin; t = i / (a.Cin * Ct); }
            else { co = i % Ct; t = (i / Ct) % 7; ci = i / (7 * Ct); }        // [Cin][7][Ct]: one K axis of 7 * Ct (icn_upconv_bwd)
            a.packed[i] = co < a.Cout ? a.w[((size_t)co * a.Cin + ci) * 7 + t] : a.w2[((size_t)(co - a.Cout) * a.Cin + ci) * 7 + t];
        }
        if (a.bias_cat && blockIdx.x == 0)
            for (int c = threadIdx.x; c < Ct; c += 256) a.bias_cat[c] = c < a.Cout ? a.bias[c] : a.bias2[c - a.Cout];
        return;
    }
    if (a.n_slots <= 0 || !a.side) return;                // (a launch that only clears the flag words)
    int j = blockIdx.x - npack;
    const float* src = a.src;
    float* side = a.side;
    if (j >= a.B * a.n_slots) { j -= a.B * a.n_slots; src = a.src2; side = a.side2; }
    const int b = j / a.n_slots, sl = j % a.n_slots;
    for (int ch = 4 * threadIdx.x; ch < a.K; ch += 4 * 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < a.E; ++e) {
            const int32_t c = a.slots[sl * a.E + e];
            if (c >= 0) v += ld4(src + ((size_t)b * a.Ps + c) * a.K + ch);
            else if (c <= -2) v += pole_mean4(src, b, a.Ps, a.ns, -2 - c, a.K, ch);
        }
        *reinterpret_cast<f32x4*>(side + ((size_t)b * a.n_slots + sl) * a.K + ch) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// MFMA gather-GEMM
// ---------------------------------------------------------------------------------------------------------
constexpr int BK = 32;        // k (channel) depth of one LDS stage = one 128-byte row segment
constexpr int EMAX = 3;       // max entries per (tap, row) in a transposed table

// 16-byte chunk swizzle of a [rows][32 floats] LDS tile: conflict-free ds_read_b128 by 32 consecutive rows
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int BM, int BN>
__global__ __launch_bounds__(256) void k_gather_gemm(
    const float* __restrict__ src,      // (B, Ps, K)
    const float* __restrict__ wt,       // [7][N][K]
    const float* __restrict__ bias,     // [N] or null
    float* __restrict__ dst,            // (B, Pd, N)
    const int32_t* __restrict__ idx,    // [7][E][Pd]
    const int32_t* __restrict__ perm,   // [Pd] row -> dst pixel, or null (identity)
    const uint32_t* __restrict__ mask32,// [Pd/32] taps in use per 32 rows, or null (all)
    int M, int Ps, int Pd, int K, int N, int E, int ns, int dbg) {
    constexpr int TM = BM / 64, TN = BN / 64;      // 32x32 MFMA tiles per wave (waves are 2 x 2)
    constexpr int RA = BM / 32, RB = BN / 32;      // 16-byte chunks each thread stages per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                       // [2][BM*32]
    float* Bs = As + 2 * BM * BK;                                     // [2][BN*32]
    int32_t* s_code = reinterpret_cast<int32_t*>(Bs + 2 * BN * BK);   // [7][E][BM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of tiles
    // so that neighbouring M-tiles (shared halo rows, same weight panel) hit the same L2.
    const int ntn = N / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, rem = nblk % 8, x = bid % 8;
        bid = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + bid / 8;
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;

    // ---- per-row gather codes -> LDS; taps in use by this tile
    unsigned tapmask = 0;
    if (mask32) {
#pragma unroll
        for (int g = 0; g < BM / 32; ++g) {
            const int m = m0 + g * 32;
            if (m < M) tapmask |= mask32[(m % Pd) >> 5];
        }
    } else {
        tapmask = 0x7f;
    }
    for (int i = tid; i < 7 * E * BM; i += 256) {
        const int row = i % BM, te = i / BM;
        const int m = m0 + row;
        int32_t code = -1;
        if (m < M) {
            const int b = m / Pd, kk = m % Pd;         // idx is in row order (pre-permuted when perm != null)
            const int32_t v = idx[(size_t)te * Pd + kk];
            code = v >= 0 ? b * Ps + v : v;
        }
        s_code[i] = code;
    }
    __syncthreads();

    const int chunk = tid & 7, srow = tid >> 3;        // staging: row srow + 32*i, 16-byte chunk `chunk`
    // rows (of this thread) x taps that need the slow path: extra entries or a pole mean
    unsigned slow = 0;
#pragma unroll
    for (int i = 0; i < RA; ++i)
        for (int t = 0; t < 7; ++t) {
            bool s = s_code[(t * E) * BM + srow + 32 * i] <= -2;
            for (int e = 1; e < E; ++e) s |= s_code[(t * E + e) * BM + srow + 32 * i] != -1;
            slow |= (unsigned)s << (i * 7 + t);
        }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ntap = __popc(tapmask);
    const int nsteps = ntap * (K / BK);
    f32x4 ra[RA], rb[RB];

    // step -> (tap, k0): taps innermost so that the 7 gathers of one channel chunk reuse the same lines
    unsigned rem_taps = tapmask;
    int k0 = 0;
    auto next_tap = [&]() {
        if (rem_taps == 0) { rem_taps = tapmask; k0 += BK; }
        const int t = __ffs(rem_taps) - 1;
        rem_taps &= rem_taps - 1;
        return t;
    };
    auto stage_load = [&](int t, int kc) {
        if (dbg & 1) return;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int32_t code = s_code[(t * E) * BM + srow + 32 * i];
            ra[i] = code >= 0 ? ld4(src + (size_t)code * K + kc + 4 * chunk) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            rb[i] = ld4(wt + ((size_t)t * N + n0 + srow + 32 * i) * K + kc + 4 * chunk);
    };
    auto stage_write = [&](int buf, int t, int kc) {
        if (dbg & 2) return;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int row = srow + 32 * i;
            f32x4 v = ra[i];
            if ((slow >> (i * 7 + t)) & 1) {
                const int b = (m0 + row) / Pd;
                for (int e = 0; e < E; ++e) {
                    const int32_t code = s_code[(t * E + e) * BM + row];
                    if (e > 0 && code >= 0) v += ld4(src + (size_t)code * K + kc + 4 * chunk);
                    if (code <= -2) v += pole_mean4(src, b, Ps, ns, -2 - code, K, kc + 4 * chunk);
                }
            }
            *reinterpret_cast<f32x4*>(As + buf * BM * BK + row * BK + 4 * (chunk ^ swz(row))) = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = srow + 32 * i;
            *reinterpret_cast<f32x4*>(Bs + buf * BN * BK + row * BK + 4 * (chunk ^ swz(row))) = rb[i];
        }
    };

    if (nsteps > 0) {
        const int t_cur = next_tap();
        stage_load(t_cur, k0);
        stage_write(0, t_cur, k0);
    }
    __syncthreads();

    const int fl = swz(l31);   // rows read by this lane are (32-aligned base) + l31
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        int t_nxt = 0, k_nxt = 0;
        const bool more = step + 1 < nsteps;
        if (more) {
            t_nxt = next_tap();
            k_nxt = k0;
            stage_load(t_nxt, k_nxt);
        }
        const float* a_base = As + buf * BM * BK + (wr * (BM / 2) + l31) * BK;
        const float* b_base = Bs + buf * BN * BK + (wc * (BN / 2) + l31) * BK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = 4 * ((2 * kk + h) ^ fl);
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + off);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + off);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        if (more) stage_write(buf ^ 1, t_nxt, k_nxt);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * (BN / 2) + j * 32 + l31;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = m0 + row;
                if (m < M) {
                    size_t drow = m;
                    if (perm) drow = (size_t)(m / Pd) * Pd + perm[m % Pd];
                    dst[drow * N + col] = acc[i][j][r] + bv;
                }
            }
    }
}

// developer routing flags: initial value from ICN_DEBUG, changed at run time by icn_set_debug_flags (tests)
static std::atomic<int> g_dbg{-1};
int debug_flags() {
    int v = g_dbg.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("ICN_DEBUG");
        v = e ? atoi(e) : 0;
        g_dbg.store(v, std::memory_order_relaxed);
    }
    return v;
}
int set_debug_flags(int flags) {
    const int old = debug_flags();
    g_dbg.store(flags < 0 ? 0 : flags, std::memory_order_relaxed);
    return old;
}
static int dbg_flags() { return debug_flags(); }

// Asynchronous failure word of a device: one int in pinned host memory mapped into the device's address space.  A kernel
// that detects a failure it cannot return (a stream-K finisher whose partner never parked its piece) ORs a bit into it with a
// system-scope atomic; the host reads it without touching the device (icn_device_status).  Allocated once per device.
static std::mutex g_status_mu;
static int* g_status_host[64] = {};
int* device_status_word() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    std::lock_guard<std::mutex> lk(g_status_mu);
    if (g_status_host[dev] == nullptr) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
            throw std::runtime_error("icn: cannot allocate the device status word");
        std::memset(p, 0, 64);
        g_status_host[dev] = static_cast<int*>(p);
    }
    return g_status_host[dev];      // unified addressing: the host pointer of mapped pinned memory is valid on the device
}
int device_status(int clear) {
    int* w = device_status_word();
    const int v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
    if (clear && v) __atomic_fetch_and(w, ~v, __ATOMIC_ACQ_REL);
    return v;
}
// developer: per-block timestamps of the next stream-K launches (icn_debug_trace; the buffer is the caller's device memory)
static unsigned long long* g_trace = nullptr;
static size_t g_trace_cap = 0;
void set_trace_buffer(void* p, size_t n_u64) { g_trace = static_cast<unsigned long long*>(p); g_trace_cap = p ? n_u64 : 0; }
static int current_device_bit() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev & 63;
}

template <int BM, int BN>
static void launch_gather_gemm(const GatherGemmArgs& a, hipStream_t s) {
    const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    const size_t lds = (size_t)2 * (BM + BN) * BK * 4 + (size_t)7 * a.E * BM * 4;
    // > 64 KiB of dynamic LDS needs the opt-in, once per DEVICE (idempotent, so a race is harmless)
    static std::atomic<uint64_t> attr_devices{0};
    if (!((attr_devices.load(std::memory_order_relaxed) >> current_device_bit()) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gather_gemm<BM, BN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devices.fetch_or((uint64_t)1 << current_device_bit(), std::memory_order_relaxed);
    }
    prof_mark_begin(BM == 64 ? (BN == 128 ? PROF_GG_64x128 : PROF_GG_64x64) : (BN == 128 ? PROF_GG_128x128 : PROF_GG_128x64),
                    a.algo_flops, s);
    hipLaunchKernelGGL((k_gather_gemm<BM, BN>), dim3(ntiles), dim3(256), lds, s, a.src, a.wt, a.bias, a.dst, a.idx,
                       a.perm, a.mask32, a.M, a.Ps, a.Pd, a.K, a.N, a.E, a.ns, dbg_flags());
    prof_mark_end(s);
}

bool gather_gemm_supported(int K, int N) { return K % BK == 0 && N % 64 == 0 && K >= BK; }

// ---------------------------------------------------------------------------------------------------------
// k_conv_dma: the production gather-GEMM (same math as k_gather_gemm), persistent and staged by LDS-DMA.
//   * staging: each lane owns BM/32 A rows and BN/32 B rows of a tile; its 16-byte chunk of every row is DMA'd
//     (buffer_load ... lds) straight into the swizzled LDS image -- the swizzle lives in the per-lane SOURCE
//     offset because the LDS side of a DMA is lane-linear -- so the steady state has no VGPR staging and no
//     ds_write.  Rows that contribute nothing (corner_mode 'zeros', rows past M) carry an out-of-range offset
//     and the buffer range check writes zeros for them.
//   * the gather table is a DmaTable (icn_geometry.h): one code per (tap, row).  Everything that is not a single
//     source pixel (pole means, duplicated transposed entries) was summed into the small per-sample `side` buffer
//     by k_conv_prologue; such a row is DMA'd from there (second buffer resource), so the K-step loop contains no
//     ordinary VMEM load at all.  (hipcc answers ordinary loads inside this loop with a `s_waitcnt vmcnt(0)` at the
//     top of every K-step, which drains the DMA ring.)
//   * the 7 x BM source byte offsets of a tile live in an LDS table built one tile ahead; bit 31 marks side-buffer
//     (or empty) rows, which are out of range for the source resource by construction (tensors < 2 GiB).
//   * persistent blocks walk tiles b, b+G, ...: the DMA pointer runs two K-steps ahead of the MFMA pointer and
//     crosses into the next tile, so the MFMA pipe does not drain at tile boundaries and launches are not rounded
//     up to whole dispatch waves.
// ---------------------------------------------------------------------------------------------------------
using lds_ptr_t = __attribute__((address_space(3))) void*;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Row offsets of the DMA kernels: < 2^31 byte offset into the source tensor; SIDE_FLAG | offset (< 2^30) a row of the side
// buffer; NOTHING_OFFSET a row of zeros.  Both flagged forms are out of range for the source resource (tensors < 2 GiB) and
// stay so when chunk / k-chunk offsets (< 2^30) are added, without wrapping; as signed ints: side < NOTHING_OFFSET <= pixel.
constexpr unsigned SIDE_FLAG = 0x80000000u;
constexpr unsigned NOTHING_OFFSET = 0xC0000000u;

// (body of the two kernels below: k_conv_dma<BM, BN, SEG> and its stream-K form k_conv_dma_sk<BM, BN>)
// ARITH (round 6): 0 = exact fp32 (v_mfma_f32_32x32x2_f32); 1 = the three-way bf16 split -- A rows stay fp32 in LDS and are cut into
// three bf16 pieces after the fragment read (a = a1 + a2 + a3 by truncation: exact, 3 x 8 = 24 significand bits), B is the weight
// prologue's packed image of three bf16 planes (pack_b3 below), six v_mfma_f32_32x32x16_bf16 per 32x32x16 block (a1b3 a3b1 a2b2 a1b2
// a2b1 a1b1, small terms first), fp32 accumulate: 2.67x less matrix-pipe time, 1.2x the exact kernel's rounding error
// (profiles/r06_ladder_b3.txt).  Waves are then 4 (rows) x NW/4 (columns), tiles 128 rows high.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
struct Pieces3 { u32x4 p1, p2, p3; };
__device__ __forceinline__ Pieces3 split_bf16x3(const f32x4 lo, const f32x4 hi) {
    unsigned x[8], r[8], r2[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float l_ = lo[e], h_ = hi[e]; x[e] = __float_as_uint(l_); x[4 + e] = __float_as_uint(h_); }   // (__builtin_bit_cast of a vector ELEMENT reads element 0: hipcc 7.2)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float rf = __uint_as_float(x[e]) - __uint_as_float(x[e] & 0xffff0000u);
        r[e] = __float_as_uint(rf);
        r2[e] = __float_as_uint(rf - __uint_as_float(r[e] & 0xffff0000u));
    }
    Pieces3 o;
#pragma unroll
    for (int m = 0; m < 4; ++m) {                 // the high halves of two values: element 2m in the low, 2m + 1 in the high 16 bits
        o.p1[m] = __builtin_amdgcn_perm(x[2 * m + 1], x[2 * m], 0x07060302u);
        o.p2[m] = __builtin_amdgcn_perm(r[2 * m + 1], r[2 * m], 0x07060302u);
        o.p3[m] = __builtin_amdgcn_perm(r2[2 * m + 1], r2[2 * m], 0x07060302u);
    }
    return o;
}

template <int BM, int BN, bool SEG, bool SK, int NW = 4, int NT = 7, int ARITH = 0>   // NW waves per workgroup as 2 x NW/2 (4: 256 threads; 8: 512, round 5); NT: taps of a plain launch (7; 1 = a dense GEMM through the plain code path, round 5); SEG: class-major rows + virtual taps of a composite table (icn_upconv_*)
__device__ __forceinline__ void conv_dma_body(
    const float* __restrict__ src,      // (B, Ps, Ks)   Ks = K, or K / 2 with src2
    const float* __restrict__ src2,     // second half of the K axis (pair bwd-data), or null
    const float* __restrict__ wt,       // [7][N][K]
    const float* __restrict__ bias,     // [N] or null
    float* __restrict__ dst,            // (B, Pd, N0)
    float* __restrict__ dst2,           // (B, Pd, N - N0) columns N0.. (pair forward), or null (N0 = N)
    const int32_t* __restrict__ dcode,  // DmaTable code [7][Pd] (row order: pre-permuted when perm != null)
    const float* __restrict__ side,     // (B, n_slots, Ks) or null
    const float* __restrict__ side2,    // same for src2
    const int32_t* __restrict__ perm,   // [Pd] row -> dst pixel, or null (identity)
    const uint32_t* __restrict__ mask32,// [Pd/32] taps in use per 32 rows, or null (all 7)
    int M, int Ps, int Pd, int K, int N, int N0, int n_slots, unsigned src_bytes, unsigned side_bytes, int ntiles, int T_arg,
    const RowSegs segs,
    int sk_mp,                          // SK: fewest K-steps a piece of a split tile may run (4: the ring's fill and the metadata look-ahead)
    float* __restrict__ sk_part,        // SK: one BM x BN partial-accumulator slot per block (raw register layout)
    int* __restrict__ sk_flag,          // SK: CONV_SK_FLAGS words zeroed by the prologue: [b] = 1: block b's piece is parked
    int* __restrict__ sk_status,        // SK: the device's asynchronous failure word (pinned host memory, icn_device_status)
    int sk_spin_limit,                  // SK: polls before a partner counts as lost; < 0: fault injection (tests): lost at once
    unsigned long long* __restrict__ trace,     // developer: 8 timestamps (100 MHz) per block, or null (icn_debug_trace)
    const int* __restrict__ sk_bnd,     // SK: range boundaries of the launch's two residue-class sizes, [2][G / 8 + 1] (sk_tables)
    const int* __restrict__ tlist) {    // !SK: per-workgroup tile lists, [G + 1] offsets then the tile ids (tile_lists), or null: b, b + G, ...
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource / LDS-DMA builtins only exist in the device pass
    ICN_CHAIN_SETPRIO();
    // T taps in wt / dcode (7 hex taps, or the virtual taps of a composite table); a tile runs at most 7 of them (its tap
    // mask), and the LDS offset table is indexed by a tap's RANK inside that mask.
    // The plain instantiation (SEG = false) is the kernel of every ordinary convolution: 7 taps, rank == tap id, rows
    // m = b * Pd + q; all of the class-major machinery compiles away there.
    constexpr unsigned INVALID_ROW = 0xFFFFFFFFu;      // destination-row table: padding row (nothing is stored)
    const int T = SEG ? T_arg : NT;
    constexpr bool B3 = ARITH == 1;
    constexpr int WR = B3 ? 4 : 2;                     // waves along M
    constexpr int WC = NW / WR, NTHR = 64 * NW;        // waves along N; threads
    constexpr int TM = BM / (32 * WR), TN = BN / (32 * WC);   // 32 x 32 MFMA tiles per wave
    constexpr int BROW = B3 ? 48 : BK;                 // floats of LDS per B row and stage: 128 B of fp32, or 3 bf16 planes x 64 B
    constexpr int RA = BM / (8 * NW), RB = BN * BROW / (256 * NW);   // 1 KiB DMA pieces per wave and stage (A: 8 rows each)
    static_assert(NW == 4 || NW == 8, "2 x 2 / 2 x 4 waves, or 4 x 1 / 4 x 2 with ARITH = 1");
    static_assert(TM >= 1 && TN >= 1 && RA >= 1 && RB >= 1 && RB * 256 * NW == BN * BROW, "tile too small for this wave grid");
    static_assert(!B3 || !SEG, "bf16x3 arithmetic: plain rows only");
    constexpr int NDMA = RA + RB;                      // DMA instructions per wave per stage
    constexpr int RL = BM / 64;                        // tile rows per lane in the metadata pass (row r*64 + lane)
    constexpr int NJ = (SEG ? 7 : NT) * RL;            // code DMA instructions per tile (64 codes each)
    constexpr int JW = (NJ + NW - 1) / NW;             // ... per wave
    // NT = 1 is the dense GEMM with identity rows (k_conv_dense_sk; launcher: conv_dense_plain): GEMM row m reads source row m, so
    // the row offsets are arithmetic -- no gather codes, no offset table, no per-tile metadata DMA + convert (round 5).
    constexpr bool IDENT = !SEG && NT == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);        // [3][BM*32]   3-stage ring
    float* Bs = As + 3 * BM * BK;                      // [3][BN*32]   (ARITH = 1: [3][3 planes][BN][32 bf16])
    unsigned* otab = reinterpret_cast<unsigned*>(Bs + 3 * BN * BROW);   // [2][7][BM] source byte offset of each row
    unsigned* drow_s = otab + 2 * 7 * BM;              // [3][BM] destination row of each tile row (perm != null only)
    float* bias_s = reinterpret_cast<float*>(drow_s + (perm ? 3 * BM : 0));   // [3][BN] (bias != null only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int l31 = lane & 31, h = lane >> 5;
    const int rsub = lane >> 3, pc = lane & 7;         // DMA: row within its 8-row group, physical 16-byte chunk
    const int ntn = N / BN, nk = K / BK;
    const int Ks = src2 ? K / 2 : K, nk0 = Ks / BK;    // channels / k-chunks per source tensor

    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    const auto rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src2 ? src2 : src), 0, src_bytes, 0x00020000);
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, T * N * K * (B3 ? 6 : 4), 0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(side ? side : src), 0, side ? side_bytes : 0u,
                                                          0x00020000);
    const auto rsrc_s2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(side2 ? side2 : src), 0, side2 ? side_bytes : 0u,
                                                           0x00020000);
    const auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(dcode), 0, T * Pd * 4, 0x00020000);
    const auto rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(perm ? perm : dcode), 0, Pd * 4, 0x00020000);
    // destinations as buffers of exactly (rows x row stride) bytes: the epilogue's range check drops what must not be stored
    const unsigned d_rows = SEG ? (unsigned)segs.B * (unsigned)Pd : (unsigned)M;
    const auto rsrc_d = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(d_rows * (unsigned)N0 * 4u), 0x00020000);
    const auto rsrc_d2 = __builtin_amdgcn_make_buffer_rsrc(dst2 ? dst2 : dst, 0, dst2 ? (int)(d_rows * (unsigned)(N - N0) * 4u) : 0, 0x00020000);
    const auto rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias ? bias : src), 0, bias ? N * 4 : 0, 0x00020000);

    // XCD-aware tile order: tiles with equal index mod 8 (one persistent block's residue class, hence one XCD
    // and one L2) form a contiguous run of (m, n) tiles.
    auto tile_origin = [&](int tile, int& m0, int& n0) __attribute__((always_inline)) {
        if constexpr (SEG) {
            // class-major rows: XCD x takes the x-th eighth of EVERY segment (segments are SEG_ALIGN-row aligned, so their
            // tile counts are multiples of 8): all XCDs walk the classes in step and carry equal work.
            const int x = tile % 8;
            int i = tile / 8;
            m0 = 0;
            n0 = 0;
#pragma unroll
            for (int sg = 0; sg < MAX_SEGS; ++sg) {
                if (sg < segs.nseg) {
                    const int end = sg + 1 < segs.nseg ? segs.row0[sg + 1] : M;
                    const int per = ((end - segs.row0[sg]) / BM) * ntn / 8;
                    if (i >= 0 && i < per) {
                        const int sw = x * per + i;
                        m0 = segs.row0[sg] + (sw / ntn) * BM;
                        n0 = (sw % ntn) * BN;
                    }
                    i -= per;
                }
            }
            return;
        }
        const int q = ntiles / 8, rem = ntiles % 8, x = tile % 8;
        const int sw = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + tile / 8;
        m0 = (sw / ntn) * BM;
        n0 = (sw % ntn) * BN;
    };
    // GEMM row m -> sample b (-1: padding row / past M) and position q in the per-sample code / perm tables
    auto decode_row = [&](int m, int& b, int& q) __attribute__((always_inline)) {
        if constexpr (!SEG) {
            b = m < M ? m / Pd : -1;
            q = m - (m / Pd) * Pd;
            return;
        }
        int r0 = segs.row0[0], cn = segs.cnt[0], of = segs.off[0];
#pragma unroll
        for (int sg = 1; sg < MAX_SEGS; ++sg)
            if (sg < segs.nseg && m >= segs.row0[sg]) { r0 = segs.row0[sg]; cn = segs.cnt[sg]; of = segs.off[sg]; }
        const int k = m - r0, bb = k / cn;
        b = bb < segs.B ? bb : -1;
        q = bb < segs.B ? of + k - bb * cn : 0;
    };
    // n-th set bit of a tap mask (wave-uniform arguments)
    auto nth_tap = [&](unsigned mk, int nth) __attribute__((always_inline)) {
        for (int i = 0; i < nth; ++i) mk &= mk - 1u;
        return __ffs(mk) - 1;
    };
    // lane-constant parts of the DMA source offsets (bytes)
    unsigned achunk[RA], bconst[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) achunk[i] = 16u * (pc ^ swz(8 * (wave + NW * i) + rsub));
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = 8 * (wave + NW * i) + rsub;
        bconst[i] = B3 ? (unsigned)((wave + NW * i) * 1024 + lane * 16)      // the bf16 image is stored as the stage wants it: a linear copy
                       : (unsigned)row * (unsigned)K * 4u + 16u * (pc ^ swz(row));
    }
    // gather code of row (sample b) -> DMA byte offset
    auto row_offset = [&](int32_t c, int b) __attribute__((always_inline)) {
        return c >= 0 ? (unsigned)(b * Ps + c) * (unsigned)Ks * 4u
             : c == -1 ? NOTHING_OFFSET
                       : SIDE_FLAG | ((unsigned)(b * n_slots + (-2 - c)) * (unsigned)Ks * 4u);
    };
    // IDENT: byte offset of the lane's i-th row of the tile starting at GEMM row mbase
    auto ident_offset = [&](int mbase, int i) __attribute__((always_inline)) {
        const int m = mbase + 8 * (wave + NW * i) + rsub;
        return m < M ? (unsigned)m * (unsigned)Ks * 4u : NOTHING_OFFSET;
    };
    // Metadata of the block's FIRST tile, built synchronously with ordinary loads: row offsets, destination rows, bias.
    // (Later tiles: ICN_META_ISSUE / ICN_META_CONVERT below, by LDS-DMA, one tile ahead.)
    auto build_first = [&](int m0, int n0, unsigned mk) {
        const int nt = IDENT ? 0 : (SEG ? __popc(mk) : NT);
        for (int e = tid; e < nt * BM; e += NTHR) {
            const int t = SEG ? nth_tap(mk, e / BM) : e / BM, row = e % BM;
            int b, q;
            decode_row(m0 + row, b, q);
            otab[e] = b >= 0 ? row_offset(dcode[(size_t)t * Pd + q], b) : NOTHING_OFFSET;
        }
        if (perm)
            for (int row = tid; row < BM; row += NTHR) {
                int b, q;
                decode_row(m0 + row, b, q);
                drow_s[row] = b >= 0 ? (unsigned)(b * Pd + perm[q]) : INVALID_ROW;
            }
        if (bias)
            for (int c = tid; c < BN; c += NTHR) bias_s[c] = bias[n0 + c];
    };
    // taps in use by a tile (stride-2 dgrad: rows are grouped by lattice parity class, a class uses 1-2 taps);
    // wave-uniform index => scalar loads
    auto tile_taps = [&](int m0) __attribute__((always_inline)) {
        unsigned mk = SEG ? 0x7fu : (1u << NT) - 1u;
        if constexpr (SEG) {                             // the tile lies in one segment
            mk = segs.mask[0];
#pragma unroll
            for (int sg = 1; sg < MAX_SEGS; ++sg)
                if (sg < segs.nseg && m0 >= segs.row0[sg]) mk = segs.mask[sg];
            return (unsigned)__builtin_amdgcn_readfirstlane((int)mk);
        }
        if (mask32) {
            mk = 0;
#pragma unroll
            for (int g2 = 0; g2 < BM / 32; ++g2)
                if (m0 + 32 * g2 < M) mk |= mask32[((m0 + 32 * g2) % Pd) >> 5];
            if (mk == 0) mk = 1;                         // never an empty step list
        }
        return (unsigned)__builtin_amdgcn_readfirstlane((int)mk);
    };
    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    const int fl = swz(l31);
    f32x4 fa[2][TM], fb[2][TN];
    auto frag0 = [&](int ring) {                          // first fragments of a stage (issued before the DMA)
        const float* a_base = As + ring * BM * BK + (wr * (BM / WR) + l31) * BK;
        const float* b_base = Bs + ring * BN * BK + (wc * (BN / WC) + l31) * BK;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + 4 * (h ^ fl));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + 4 * (h ^ fl));
    };
    auto compute = [&](int ring) {
        const float* a_base = As + ring * BM * BK + (wr * (BM / WR) + l31) * BK;
        const float* b_base = Bs + ring * BN * BK + (wc * (BN / WC) + l31) * BK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                const int off = 4 * ((2 * (kk + 1) + h) ^ fl);
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + off);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + off);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the next group's LDS reads ahead of this group's MFMAs
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][sidx], fb[kk & 1][j][sidx], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- ARITH = 1: fragments of one 16-wide k-block (kb = 0, 1 of a stage): per 32-row A tile 8 fp32 of row l31 (k = 16 kb + 8 h ..),
    // per 32-column B tile and plane 8 bf16 of row l31 (one ds_read_b128; 64-byte rows, chunk q at q ^ ((row >> 2) & 3)).
    // The K-step is ROTATED across its barrier: the six-MFMA groups of a stage's second k-block issue after the barrier, beside the
    // reads and the split of the next stage's first k-block, so no wave sits through LDS latency + 44 VALU instructions with an
    // empty matrix pipe; sched_group_barrier fixes the interleave (B3_LEAD MFMAs first, then one MFMA per B3_VPM VALU instructions).
    f32x4 ra3[TM][2];
    u32x4 rb3[2][TN][3];
    Pieces3 pa3[2][TM];
    constexpr int B3_NMF = TM * TN * 6, B3_LEAD = 3, B3_VPM = (44 * TM + (B3_NMF - B3_LEAD) - 1) / (B3_NMF - B3_LEAD);
    auto b3_read = [&](int ring, int kb) __attribute__((always_inline)) {
        const char* a_row = reinterpret_cast<const char*>(As + ring * BM * BK + (wr * (BM / WR) + l31) * BK);
        const char* b_row = reinterpret_cast<const char*>(Bs + ring * BN * BROW) + (wc * (BN / WC) + l31) * 64;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            ra3[i][0] = *reinterpret_cast<const f32x4*>(a_row + i * 32 * 128 + 16 * ((4 * kb + 2 * h) ^ fl));
            ra3[i][1] = *reinterpret_cast<const f32x4*>(a_row + i * 32 * 128 + 16 * ((4 * kb + 2 * h + 1) ^ fl));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                rb3[kb][j][pl] = *reinterpret_cast<const u32x4*>(b_row + pl * BN * 64 + j * 32 * 64 + 16 * ((2 * kb + h) ^ ((l31 >> 2) & 3)));
    };
    auto b3_split = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) pa3[kb][i] = split_bf16x3(ra3[i][0], ra3[i][1]);
    };
    // the pieces are consumed a barrier later: without a use HERE hipcc sinks the whole split below the step's MFMAs and waits
    auto b3_pin = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                asm volatile("" : "+v"(pa3[kb][i].p1[m]), "+v"(pa3[kb][i].p2[m]), "+v"(pa3[kb][i].p3[m]));
    };
#define ICN_MF16(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), c_, 0, 0, 0)
    auto b3_mfmas = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                c = ICN_MF16(pa3[kb][i].p1, rb3[kb][j][2], c);
                c = ICN_MF16(pa3[kb][i].p3, rb3[kb][j][0], c);
                c = ICN_MF16(pa3[kb][i].p2, rb3[kb][j][1], c);
                c = ICN_MF16(pa3[kb][i].p1, rb3[kb][j][1], c);
                c = ICN_MF16(pa3[kb][i].p2, rb3[kb][j][0], c);
                c = ICN_MF16(pa3[kb][i].p1, rb3[kb][j][0], c);
                acc[i][j] = c;
            }
    };
    auto b3_interleave = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_group_barrier(0x8, B3_LEAD, 0);
#pragma unroll
        for (int q = 0; q < B3_NMF - B3_LEAD; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x2, B3_VPM, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        }
    };

    // ---- 3-stage LDS ring: while step s computes from stage s, stages s+1 and s+2 are landed / in flight.
    //   top of step s : issue stage s+2 into ring slot (s+2)%3 -- last read in step s-1, and every wave has passed
    //                   the barrier that ended step s-1.
    //   end of step s : waiting until only the youngest NDMA operations (stage s+2's DMA) are outstanding means
    //                   stage s+1 has landed; a raw barrier publishes it.  (__syncthreads() would drain vmcnt to 0 and
    //                   serialise the ring.)  A stage that had side-buffer rows issues a lane-masked pair of DMAs per
    //                   row, whose instruction count is not fixed (an all-masked one is branched over): such a step
    //                   (p_exact = 0) waits for vmcnt(0) instead.
    // The DMA pointer (i_*) runs two steps ahead of the compute pointer and crosses into the next tile, so the MFMA
    // pipe does not drain at tile boundaries; the first two iterations only fill the ring.
    // The next tile's metadata (gather codes, destination rows, bias) is itself fetched by LDS-DMA at the top of the
    // tile's first step -- ahead of that step's stage, so the step's counted wait retires it --, converted in place
    // to byte offsets after that wait and published by the step's barrier; its first reader is the row-offset
    // prefetch in step S-3 >= 1.  The persistent loop therefore contains no load into a VGPR at all: hipcc answers
    // those with `s_waitcnt vmcnt(0)`, which drains the ring.
    // All loop state is kept in plain ints and passed through readfirstlane: hipcc must see the DMA's LDS base
    // and scalar offset as wave-uniform or it wraps every DMA in a waterfall loop.
    // SK: a block's work is a list of SEGMENTS (tile, K-step range [s0, s1) of the tile's S = taps * nk steps, in the order
    // k-chunk major, tap minor): whole tiles b, b + G, ... first, then its range of the split tiles (icn_streamk.h), walked BACK
    // TO FRONT.  seg_fetch hands out the next one; everything below that says "tile" means the segment's tile, and the K-step
    // loop, the DMA pointer and the epilogue respect the segment's step range.  (Without SK a segment is a whole tile.)
    // Who waits for whom: of a tile cut in K, the block holding its LAST k-chunks finishes it; that piece lies at the front
    // of the block's range, i.e. it is the last thing the block does, while the other pieces lie at the back of LOWER-numbered
    // blocks' ranges, i.e. they are the first thing those blocks do after their whole tiles.  A block therefore only ever
    // waits for lower block ids of its own launch -- dispatched before it, whatever else shares the GPU -- and those never
    // wait for anything before parking their piece.  (A ticket drawn from an atomic counter instead of blockIdx would not
    // even need in-order dispatch; its round trip at the top of every launch cost half of what stream-K gains.)
    unsigned long long tr_t0 = 0, tr_t1 = 0, tr_t2 = 0, tr_split = 0, tr_wait = 0, tr_epi = 0, tr_e0 = 0;
    unsigned tr_nseg = 0;
    if (trace) tr_t0 = __builtin_amdgcn_s_memrealtime();
    SkWalk skw{};                                         // (icn_streamk.h; plain ints: stays in scalar registers)
    const int sk_S = (SEG ? __popc(segs.mask[0]) : NT) * nk;   // SK: K-steps of a tile (every tile of an SK launch runs the same taps)
    if constexpr (SK) skw.init(blockIdx.x, gridDim.x, ntiles, sk_S, sk_mp, sk_bnd);
    auto seg_fetch = [&](int& tile_, int& s0_, int& s1_) __attribute__((always_inline)) { return skw.next(tile_, s0_, s1_); };
    int tile = blockIdx.x, m0, n0;
    int c_s0 = 0, c_s1 = 0, n_s0 = 0, n_s1 = 0;           // K-step range of the compute / next segment
    if constexpr (SK) {
        if (!seg_fetch(tile, c_s0, c_s1)) return;         // more blocks than work (block-uniform, before any barrier)
    }
    // Tile lists (launches whose tiles run different numbers of K-steps: tap masks): the host dealt the tiles of each residue
    // class to its workgroups so that their step totals are even (launch_conv_dma_t).  The ids come through the scalar cache,
    // two tiles ahead of the compute tile.
    const int* const tl_ids = tlist ? tlist + gridDim.x + 1 : nullptr;
    int tl_i = 0, tl_hi = 0, tl_pend = ntiles;
    if constexpr (!SK) {
        if (tlist) {
            tl_i = tlist[blockIdx.x];
            tl_hi = tlist[blockIdx.x + 1];
            if (tl_i >= tl_hi) return;                    // no tile for this workgroup (block-uniform, before any barrier)
            tile = tl_ids[tl_i];
        }
    }
    tile_origin(tile, m0, n0);
    build_first(m0, n0, tile_taps(m0));
    __syncthreads();
    if (trace) tr_t1 = __builtin_amdgcn_s_memrealtime();
    int slot = 0, eslot = 0;                              // offset-table slot / epilogue-table slot of the compute tile
    int next_tile = tile + gridDim.x;
    if (!SK && tlist) {
        next_tile = tl_i + 1 < tl_hi ? tl_ids[tl_i + 1] : ntiles;
        tl_pend = tl_i + 2 < tl_hi ? tl_ids[tl_i + 2] : ntiles;
    }
    int has_next = next_tile < ntiles;
    if constexpr (SK) has_next = seg_fetch(next_tile, n_s0, n_s1);
    int nm0 = m0, nn0 = n0;
    if (has_next) tile_origin(next_tile, nm0, nn0);
    unsigned mask_c = tile_taps(m0), mask_n = has_next ? tile_taps(nm0) : (SEG ? 0x7fu : (1u << NT) - 1u);   // taps of the compute / next tile
    if constexpr (!SK) {                                  // whole tiles: every step of the tile's taps
        c_s1 = __popc(mask_c) * nk;
        n_s1 = __popc(mask_n) * nk;
    }
    // first step of a segment -> (k-chunk, tap) of the DMA pointer
    auto seg_start = [&](unsigned mk_, int s0_, int& kc_, int& t_) __attribute__((always_inline)) {
        const int nt_ = __popc(mk_);
        kc_ = s0_ / nt_;
        t_ = nth_tap(mk_, s0_ - kc_ * nt_);
    };
    int i_t, i_kc, n_t0 = 0, n_kc0 = 0;                   // DMA pointer (tap, k-chunk); where it enters the next segment
    seg_start(mask_c, c_s0, i_kc, i_t);
    if (has_next) seg_start(mask_n, n_s0, n_kc0, n_t0);
    int i_left = c_s1 - c_s0, i_ring = 0, i_own = 1, i_live = 1;   // steps left in the pointer's segment; i_own: inside compute tile
    unsigned pbase[RA];                                   // row offsets of the next stage to be issued (prefetched)
#pragma unroll
    for (int i = 0; i < RA; ++i) pbase[i] = IDENT ? ident_offset(m0, i) : otab[(SEG ? 0 : i_t) * BM + 8 * (wave + NW * i) + rsub];   // SEG: rank 0
    int issued = 0, p_exact = 1;
    int mb[RL];                                           // metadata pass: sample of the lane's rows in the next tile, or -1
    // Issue the stage under the DMA pointer, advance the pointer, prefetch the next stage's row offsets.
    // (macros, not lambdas: hipcc spilled the captured loop state of a lambda to scratch, and a scratch load is a
    // VMEM op whose vmcnt(0) wait drains the DMA ring)
#define ICN_ISSUE_STAGE() do { \
        issued = i_live; \
        p_exact = 1; \
        if (i_live) { \
            const int tn0 = __builtin_amdgcn_readfirstlane(i_own ? n0 : nn0); \
            const int sec_ = __builtin_amdgcn_readfirstlane(i_kc >= nk0);   /* k-chunk of the second source tensor */ \
            const int a_soff = __builtin_amdgcn_readfirstlane((sec_ ? i_kc - nk0 : i_kc) * (BK * 4)); \
            const int b_soff = __builtin_amdgcn_readfirstlane(B3 ? ((i_t * nk + i_kc) * ntn + tn0 / BN) * (BN * 192)   /* image block (tap, k-chunk, column tile) */ \
                                                                  : ((i_t * N + tn0) * K + i_kc * BK) * 4); \
            const auto ra_ = sec_ ? rsrc_a2 : rsrc_a; \
            bool side_row = false; \
_Pragma("unroll") \
            for (int i = 0; i < RA; ++i) side_row |= (int)pbase[i] < (int)NOTHING_OFFSET; \
            if ((ICN_EXP & 4) || side == nullptr || __builtin_amdgcn_ballot_w64(side_row) == 0) {   /* (no side buffer: no row can name one) */ \
_Pragma("unroll") \
                for (int i = 0; i < RA; ++i) { \
                    float* lds_dst = As + __builtin_amdgcn_readfirstlane(i_ring * BM * BK + 8 * (wave + NW * i) * BK); \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_ptr_t)lds_dst, 16, pbase[i] + achunk[i], a_soff, 0, 0); \
                } \
            } else { \
                p_exact = 0; \
                const auto rs_ = sec_ ? rsrc_s2 : rsrc_s; \
_Pragma("unroll") \
                for (int i = 0; i < RA; ++i) { \
                    float* lds_dst = As + __builtin_amdgcn_readfirstlane(i_ring * BM * BK + 8 * (wave + NW * i) * BK); \
                    if ((int)pbase[i] >= (int)NOTHING_OFFSET) \
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_ptr_t)lds_dst, 16, pbase[i] + achunk[i], a_soff, 0, 0); \
                    else \
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)lds_dst, 16, \
                                                                 (pbase[i] & ~SIDE_FLAG) + achunk[i], a_soff, 0, 0); \
                } \
            } \
_Pragma("unroll") \
            for (int i = 0; i < RB; ++i) { \
                float* lds_dst = Bs + __builtin_amdgcn_readfirstlane(i_ring * BN * BROW + 8 * (wave + NW * i) * BK); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)lds_dst, 16, bconst[i], b_soff, 0, 0); \
            } \
            i_ring = i_ring == 2 ? 0 : i_ring + 1; \
            if (--i_left == 0) {                         /* the pointer leaves its segment */ \
                if (i_own && has_next) { i_own = 0; i_left = n_s1 - n_s0; i_kc = n_kc0; i_t = n_t0; } \
                else i_live = 0; \
            } else { \
                const unsigned mk_ = i_own ? mask_c : mask_n; \
                const unsigned higher_ = mk_ & ~((2u << i_t) - 1u); \
                if (higher_) { \
                    i_t = __ffs(higher_) - 1; \
                } else { \
                    ++i_kc; \
                    i_t = __ffs(mk_) - 1; \
                } \
            } \
            if (i_live) { \
                const unsigned mr_ = i_own ? mask_c : mask_n; \
                const int rk_ = SEG ? __popc(mr_ & ((1u << i_t) - 1u)) : i_t;   /* row of the offset table */ \
                const int tb = __builtin_amdgcn_readfirstlane(((i_own ? slot : slot ^ 1) * 7 + rk_) * BM); \
_Pragma("unroll") \
                for (int i = 0; i < RA; ++i) pbase[i] = IDENT ? ident_offset(i_own ? m0 : nm0, i) : otab[tb + 8 * (wave + NW * i) + rsub]; \
            } \
        } \
    } while (0)
    // Metadata of the NEXT tile (rows nm0.., columns nn0..), fetched by LDS-DMA: code DMA j (of NJ, wave j % NW) brings the
    // 64 codes of tap j / RL, rows (j % RL) * 64 + lane, to their final place in the offset table; waves < RL fetch the
    // destination-row permutation, the last BN / 64 waves the bias.
#define ICN_META_ISSUE() do { \
        const int ne_ = __builtin_amdgcn_readfirstlane(eslot == 2 ? 0 : eslot + 1); \
        const int nj_ = IDENT ? 0 : (SEG ? __builtin_amdgcn_readfirstlane(__popc(mask_n) * RL) : NJ);   /* SEG: the next tile's taps only */ \
        int mp_[RL]; \
_Pragma("unroll") \
        for (int r = 0; r < RL; ++r) decode_row(nm0 + r * 64 + lane, mb[r], mp_[r]); \
_Pragma("unroll") \
        for (int jj = 0; jj < JW; ++jj) { \
            const int j = wave + NW * jj; \
            if (j < nj_) { \
                const int r = j % RL; \
                const int b_ = RL == 1 ? mb[0] : (r ? mb[RL - 1] : mb[0]); \
                const int p_ = RL == 1 ? mp_[0] : (r ? mp_[RL - 1] : mp_[0]); \
                const int t_ = SEG ? __builtin_amdgcn_readfirstlane(nth_tap(mask_n, j / RL)) : j / RL; \
                unsigned* dst_ = otab + __builtin_amdgcn_readfirstlane((slot ^ 1) * 7 * BM + j * 64); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lds_ptr_t)dst_, 4, \
                                                         b_ >= 0 ? (unsigned)(t_ * Pd + p_) * 4u : SIDE_FLAG, 0, 0, 0); \
            } \
        } \
        if (perm && wave < RL) { \
            const int b_ = RL == 1 ? mb[0] : (wave ? mb[RL - 1] : mb[0]); \
            const int p_ = RL == 1 ? mp_[0] : (wave ? mp_[RL - 1] : mp_[0]); \
            unsigned* dst_ = drow_s + __builtin_amdgcn_readfirstlane(ne_ * BM + wave * 64); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_p, (lds_ptr_t)dst_, 4, b_ >= 0 ? (unsigned)p_ * 4u : SIDE_FLAG, 0, 0, 0); \
        } \
        if (bias && wave >= NW - BN / 64) { \
            const int c_ = __builtin_amdgcn_readfirstlane((wave - (NW - BN / 64)) * 64); \
            float* dst_ = bias_s + __builtin_amdgcn_readfirstlane(ne_ * BN + c_); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_bias, (lds_ptr_t)dst_, 4, (unsigned)(nn0 + c_ + lane) * 4u, 0, 0, 0); \
        } \
    } while (0)
    // ... and, once landed (after the step's counted wait), turned in place into byte offsets / destination rows by the
    // lane that fetched them; the step's barrier publishes the tables.
#define ICN_META_CONVERT() do { \
        const int ne_ = eslot == 2 ? 0 : eslot + 1; \
        const int nj_ = IDENT ? 0 : (SEG ? __builtin_amdgcn_readfirstlane(__popc(mask_n) * RL) : NJ); \
_Pragma("unroll") \
        for (int jj = 0; jj < JW; ++jj) { \
            const int j = wave + NW * jj; \
            if (j < nj_) { \
                const int r = j % RL; \
                const int b_ = RL == 1 ? mb[0] : (r ? mb[RL - 1] : mb[0]); \
                unsigned* e_ = otab + (slot ^ 1) * 7 * BM + j * 64 + lane; \
                *e_ = b_ >= 0 ? row_offset((int32_t)*e_, b_) : NOTHING_OFFSET; \
            } \
        } \
        if (perm && wave < RL) { \
            const int b_ = RL == 1 ? mb[0] : (wave ? mb[RL - 1] : mb[0]); \
            unsigned* e_ = drow_s + ne_ * BM + wave * 64 + lane; \
            *e_ = b_ >= 0 ? (unsigned)(b_ * Pd) + *e_ : INVALID_ROW; \
        } \
    } while (0)
    // End of a K-step: retire the previous stage (and the metadata fetched ahead of this step's stage), publish.
#define ICN_RETIRE_AND_PUBLISH(META) do { \
        if (issued && p_exact) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        if (META) ICN_META_CONVERT(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier(); \
    } while (0)
    ICN_ISSUE_STAGE();                                    // ring fill: stages 0 and 1 (>= 4 steps per tile)
    ICN_RETIRE_AND_PUBLISH(false);
    ICN_ISSUE_STAGE();
    ICN_RETIRE_AND_PUBLISH(false);
    zero_acc();
    if (trace) tr_t2 = __builtin_amdgcn_s_memrealtime();
    int c_ring = 0;
    for (;;) {
        const int S = c_s1 - c_s0;                        // K-steps of this segment (without SK: of the whole tile)
        if (SK && trace && tr_split == 0 && (c_s0 > 0 || c_s1 < sk_S)) tr_split = __builtin_amdgcn_s_memrealtime();
        for (int step = 0; step < S; ++step) {
            const bool meta = !(ICN_EXP & 2) && step == 0 && has_next;      // wave-uniform
            if constexpr (B3) {
                b3_read(c_ring, 0);
                if (meta) ICN_META_ISSUE();
                // The stage issue is ~100 scalar / address instructions with branches (no MFMA can be scheduled into it), and the two
                // waves of a SIMD (waves w and w + 4 of the workgroup) reach it together after the step's barrier: the matrix pipe then
                // idles for its whole length.  The upper four waves therefore issue their stage AFTER their first MFMA group: while one
                // wave of a SIMD walks the issue code the other one feeds the pipe (ICN_EXP & 32: both first, as before).
                const bool late_issue = NW == 8 && !(ICN_EXP & 32) && wave >= 4;      // wave-uniform
                if (!late_issue) ICN_ISSUE_STAGE();
                __builtin_amdgcn_sched_barrier(0);
                if (step > 0) {                           // the previous stage's second k-block beside this stage's first split
                    b3_mfmas(1);
                    b3_split(0);
                    b3_pin(0);
                    b3_interleave();
                } else {
                    b3_split(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (late_issue) ICN_ISSUE_STAGE();
                b3_read(c_ring, 1);
                __builtin_amdgcn_sched_barrier(0);
                b3_mfmas(0);
                b3_split(1);
                b3_pin(1);
                b3_interleave();
                __builtin_amdgcn_sched_barrier(0);
            } else {
            frag0(c_ring);
            if (meta) ICN_META_ISSUE();                   // next tile's tables: consumed >= 2 K-steps from now
            ICN_ISSUE_STAGE();                            // stage s+2
            compute(c_ring);
            }
            ICN_RETIRE_AND_PUBLISH(meta);                 // stage s+1 landed and visible
            c_ring = c_ring == 2 ? 0 : c_ring + 1;
        }
        if constexpr (B3) b3_mfmas(1);                    // the segment's last k-block
        if (trace) tr_e0 = __builtin_amdgcn_s_memrealtime();          // K-steps of the segment done: epilogue + switch from here
        bool sk_store = true;
        if constexpr (SK) {
            // A tile cut in K: the block holding its LAST k-chunks finishes it (that segment is the last thing the block does,
            // the other pieces are the first thing LOWER-numbered blocks do after their whole tiles, so they are long done).
            // The others park their raw accumulators in their slot and raise their flag; the finisher adds the slots in block
            // order -- a fixed order, so the result does not depend on timing.
            // Slots and flags cross XCDs (one L2 each, not coherent with one another for ordinary accesses): every access to
            // them is a system-scope one (sc0 sc1: written through / fetched past the caches), ordered by hand -- the data
            // stores have completed (vmcnt) on every wave before the flag is raised, the data loads are issued after the
            // flag was seen.  Release / acquire FENCES would do the same by writing back and invalidating the whole L2 of
            // the XCD, once per block: measured, that made every launch 10 - 50 % slower.
            constexpr int SYS = 17;                       // cache policy bits of the buffer builtins: sc0 | sc1
            const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc(sk_part, 0, (int)(gridDim.x * (unsigned)(BM * BN * 4)), 0x00020000);
            const int me = blockIdx.x;                     // this block's slot / flag
            if (c_s1 < sk_S) {                             // a piece that does not reach the tile's end: park it
                if (sk_spin_limit > 0 && (sk_spin_limit & (1 << 28))) {       // tests (debug flag 8192): a SLOW partner -- the finisher
#pragma unroll 1
                    for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(127);   // really waits (~50 us) instead of finding the piece parked
                }
                const unsigned base = (unsigned)me * (unsigned)(BM * BN * 4) + tid * 16u;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const f32x4 v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_k,
                                                                   base + ((i * TN + j) * 4 + r4) * (NTHR * 16u), 0, SYS);
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(sk_flag + me, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                sk_store = false;
            } else if (c_s0 > 0) {                         // the tile's last steps: add the earlier blocks' pieces, nearest first
                const int nku = sk_S, lt = tile / 8 - skw.dp_l;
                int pos = lt * nku + c_s0, nb = skw.bl - 1;         // steps [lt * S, pos) are parked in the slots of blocks nb, nb - 1, ...
                bool sk_lost = false;
                int* const lost_s = reinterpret_cast<int*>(bias_s + (bias ? 3 * BN : 0));   // one LDS word behind the tables
                while (pos > lt * nku && nb >= 0) {
                    const int rs = skw.range_start(nb);             // block nb's range is [rs, range_start(nb + 1))
                    if (rs >= pos) {                                // an empty range: that block parks nothing (degenerate shapes)
                        --nb;
                        continue;
                    }
                    const int blk = nb * 8 + skw.x;
                    // Block-uniform wait: ONE lane polls the partner's flag (system scope) and the verdict goes through LDS, so
                    // every wave takes the same branch (with one counter per thread, waves could disagree on a timeout and leave
                    // a tile half NaN) and 255 threads' worth of polling traffic is gone.
                    if (tid == 0) {
                        const unsigned long long w0_ = trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
                        int lost = sk_spin_limit < 0, spins = 0;
                        while (!lost && __hip_atomic_load(sk_flag + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > sk_spin_limit) lost = 1;   // seconds: a partner that never arrives is a bug, not a reason to hang the GPU
                        }
                        if (lost) __hip_atomic_fetch_or(sk_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (trace) tr_wait += __builtin_amdgcn_s_memrealtime() - w0_;
                        *lost_s = lost;
                    }
                    __syncthreads();
                    if (*lost_s) {
                        sk_lost = true;
                        break;
                    }
                    asm volatile("" ::: "memory");
                    const unsigned base = (unsigned)blk * (unsigned)(BM * BN * 4) + tid * 16u;
                    // all of a piece's loads in flight before the first add (they are system-scope loads: ~2 us each; left to itself
                    // the scheduler of the build without packed fp32 waited for every one of them in turn)
                    f32x4 pv[TM * TN * 4];
#pragma unroll
                    for (int q = 0; q < TM * TN * 4; ++q)
                        pv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, base + q * (NTHR * 16u), 0, SYS));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i][j][4 * r4 + e] += pv[(i * TN + j) * 4 + r4][e];
                    pos = max(lt * nku, rs);                        // block nb's range starts there
                    --nb;
                    __syncthreads();                               // everyone has read lost_s before lane 0 writes it again
                }
                if (sk_lost) {                             // make the failure loud: the tile becomes NaN, and so does the loss
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_nanf("");
                }
            }
        }
        // ---- tile epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        // Buffer stores (round 3): the output tensor is a buffer resource of exactly rows * row-stride bytes, a lane's byte offset
        // is row * stride + column, and everything that must not be stored -- rows past M, padding rows of a permuted launch
        // (INVALID_ROW in the destination-row table) -- is simply out of range, so the epilogue is four instructions per element
        // (accumulator read, bias add, offset add, store) with no exec-mask branch and no 64-bit arithmetic.  The per-element
        // `if (m < M) dst[(size_t)drow * stride + col] = ...` it replaces compiled to ~40 VALU instructions per element and cost
        // 4-5 us per tile (icn_debug_trace: 7-9 % of a launch of 28-step tiles, a third of one of 4-8-step tiles), which was
        // not the stores: leaving them out changed nothing, leaving the loop out took the launch from 215 to 205 us.
        if (sk_store && !(ICN_EXP & 1)) {
            unsigned dr[TM][16];                              // destination rows of this lane's elements (permuted launches)
            if (perm) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dr[i][r] = drow_s[eslot * BM + wr * (BM / WR) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = wc * (BN / WC) + j * 32 + l31, col = n0 + cl;
                const float bv = (bias && !(ICN_EXP & 8)) ? bias_s[eslot * BN + cl] : 0.f;
                // pair forward: 32-column groups at or beyond N0 belong to the second output tensor (wave-uniform)
                const bool second = __builtin_amdgcn_readfirstlane(n0 + wc * (BN / WC) + j * 32 >= N0 ? 1 : 0) != 0;
                const unsigned rs = (unsigned)(second ? N - N0 : N0) * 4u;                  // row stride of the tensor, bytes
                const auto rd = second ? rsrc_d2 : rsrc_d;
                const unsigned cb = (unsigned)(second ? col - N0 : col) * 4u;
                if (!perm) {      // GEMM row m is output row m (also the dense one-tap GEMMs: one identity segment)
                    const unsigned base = (unsigned)(m0 + wr * (BM / WR) + 4 * h) * rs + cb;  // (rows >= d_rows: >= num_records, dropped)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i][j][r] + bv), rd,
                                                                  (ICN_EXP & 16) ? 0xFFFFFF00u : base + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * rs,
                                                                  0, ICN_EXP_STORE_POLICY);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i][j][r] + bv), rd,
                                                                  ((ICN_EXP & 16) || dr[i][r] == INVALID_ROW) ? 0xFFFFFF00u : dr[i][r] * rs + cb,
                                                                  0, ICN_EXP_STORE_POLICY);
                }
            }
        }
        if ((ICN_EXP & 1) && sk_store) {                      // pricing build: the accumulators must stay live without an epilogue
            float keep_ = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) keep_ += acc[i][j][r];
            if (keep_ == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, keep_), rsrc_d, 0, 0, 0);
        }
        if (trace) { tr_epi += __builtin_amdgcn_s_memrealtime() - tr_e0; ++tr_nseg; }
        if (!has_next) break;
        zero_acc();
        tile = next_tile;
        m0 = nm0;
        n0 = nn0;
        mask_c = mask_n;
        if (!(ICN_EXP & 2)) {
            slot ^= 1;
            eslot = eslot == 2 ? 0 : eslot + 1;
        }
        i_own = 1;                                        // the DMA pointer is already inside this tile
        next_tile = tile + gridDim.x;
        if (!SK && tlist) {
            next_tile = tl_pend;
            ++tl_i;
            tl_pend = tl_i + 2 < tl_hi ? tl_ids[tl_i + 2] : ntiles;
        }
        has_next = next_tile < ntiles;
        c_s0 = n_s0;
        c_s1 = n_s1;
        if constexpr (SK) has_next = seg_fetch(next_tile, n_s0, n_s1);
        if (has_next) {
            tile_origin(next_tile, nm0, nn0);
            mask_n = tile_taps(nm0);
            if constexpr (!SK) n_s1 = __popc(mask_n) * nk;
            seg_start(mask_n, n_s0, n_kc0, n_t0);
        }
    }
    if (trace && tid == 0) {
        unsigned long long* o = trace + (size_t)blockIdx.x * 8;
        o[0] = tr_t0; o[1] = tr_t1; o[2] = tr_t2; o[3] = tr_split; o[4] = __builtin_amdgcn_s_memrealtime(); o[5] = tr_wait | ((unsigned long long)tr_nseg << 48) | (tr_epi << 24);   // wait (24 bits) | epilogue time (24 bits) | segments
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // XCC_ID (HW_REG 20, bits 3:0)
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);    // HW_ID
    }
#undef ICN_ISSUE_STAGE
#undef ICN_META_ISSUE
#undef ICN_META_CONVERT
#undef ICN_RETIRE_AND_PUBLISH
#undef ICN_MF16
#endif
}

template <int BM, int BN, bool SEG>
__global__ __launch_bounds__(256) void k_conv_dma(const float* __restrict__ src, const float* __restrict__ src2,
                                                   const float* __restrict__ wt, const float* __restrict__ bias,
                                                   float* __restrict__ dst, float* __restrict__ dst2,
                                                   const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                   const float* __restrict__ side2, const int32_t* __restrict__ perm,
                                                   const uint32_t* __restrict__ mask32, int M, int Ps, int Pd, int K, int N, int N0,
                                                   int n_slots, unsigned src_bytes, unsigned side_bytes, int ntiles, int T_arg,
                                                   const RowSegs segs, unsigned long long* __restrict__ trace,
                                                   const int* __restrict__ tlist) {
    conv_dma_body<BM, BN, SEG, false>(src, src2, wt, bias, dst, dst2, dcode, side, side2, perm, mask32, M, Ps, Pd, K, N, N0, n_slots,
                                      src_bytes, side_bytes, ntiles, T_arg, segs, 4, nullptr, nullptr, nullptr, 0, trace, nullptr, tlist);
}

// Stream-K form: same tiles, same K-step pipeline; the last 1 + frac rounds of tiles are cut into equal k-chunk ranges (sk_plan),
// so that a launch of 2.8 rounds takes 2.8 and not 3 tile times, and a launch with fewer tiles than block slots (the 256 -> 256
// layer at r = 2: 360 tiles for 768 slots) still uses every CU.  For launches whose tiles all run the same taps: the plain
// 7-tap convolutions without row permutation (SEG = false), and the one-tap dense GEMMs of the decoder heads (SEG = true).
template <int BM, int BN, bool SEG>
__global__ __launch_bounds__(256) void k_conv_dma_sk(const float* __restrict__ src, const float* __restrict__ src2,
                                                      const float* __restrict__ wt, const float* __restrict__ bias,
                                                      float* __restrict__ dst, float* __restrict__ dst2,
                                                      const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                      const float* __restrict__ side2, const int32_t* __restrict__ perm, int M, int Ps,
                                                      int Pd, int K, int N, int N0, int n_slots, unsigned src_bytes,
                                                      unsigned side_bytes, int ntiles, int T_arg, const RowSegs segs, int sk_mp,
                                                      float* __restrict__ sk_part, int* __restrict__ sk_flag,
                                                      int* __restrict__ sk_status, int sk_spin_limit,
                                                      unsigned long long* __restrict__ trace, const int* __restrict__ sk_bnd) {
    conv_dma_body<BM, BN, SEG, true>(src, src2, wt, bias, dst, dst2, dcode, side, side2, SEG ? perm : nullptr, nullptr, M, Ps, Pd, K, N, N0, n_slots,
                                     src_bytes, side_bytes, ntiles, T_arg, segs, sk_mp, sk_part, sk_flag, sk_status, sk_spin_limit, trace, sk_bnd, nullptr);
}

// A SINGLE convolution (one source tensor, one destination: src2 = dst2 = side2 = null) through a wrapper that says so at compile
// time (round 5): the pair forms' second buffer resources, the per-step resource selects and the epilogue's second-tensor test
// compile away, as the row permutation and the tap masks already do in k_conv_dma_sk.  Same arguments as k_conv_dma_sk.
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_conv_single_sk(const float* __restrict__ src, const float* __restrict__ src2,
                                                         const float* __restrict__ wt, const float* __restrict__ bias,
                                                         float* __restrict__ dst, float* __restrict__ dst2,
                                                         const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                         const float* __restrict__ side2, const int32_t* __restrict__ perm, int M, int Ps,
                                                         int Pd, int K, int N, int N0, int n_slots, unsigned src_bytes,
                                                         unsigned side_bytes, int ntiles, int T_arg, const RowSegs segs, int sk_mp,
                                                         float* __restrict__ sk_part, int* __restrict__ sk_flag,
                                                         int* __restrict__ sk_status, int sk_spin_limit,
                                                         unsigned long long* __restrict__ trace, const int* __restrict__ sk_bnd) {
    conv_dma_body<BM, BN, false, true>(src, nullptr, wt, bias, dst, nullptr, dcode, side, nullptr, nullptr, nullptr, M, Ps, Pd, K, N, N, n_slots,
                                       src_bytes, side_bytes, ntiles, T_arg, segs, sk_mp, sk_part, sk_flag, sk_status, sk_spin_limit, trace, sk_bnd, nullptr);
}

// A dense GEMM -- one "tap" whose gather is the identity: the decoder heads' z = x W (N = 7 C) and dx = g W (K = 7 C), DESIGN 4.2 --
// through the PLAIN code path (SEG = false, NT = 1) instead of the class-major one (round 5): the plain K-loop is the one hipcc peels
// and carries no segment tables, rank look-ups or per-tile tap counts.  Same arguments as k_conv_dma_sk.
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_conv_dense_sk(const float* __restrict__ src, const float* __restrict__ src2,
                                                        const float* __restrict__ wt, const float* __restrict__ bias,
                                                        float* __restrict__ dst, float* __restrict__ dst2,
                                                        const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                        const float* __restrict__ side2, const int32_t* __restrict__ perm, int M, int Ps,
                                                        int Pd, int K, int N, int N0, int n_slots, unsigned src_bytes,
                                                        unsigned side_bytes, int ntiles, int T_arg, const RowSegs segs, int sk_mp,
                                                        float* __restrict__ sk_part, int* __restrict__ sk_flag,
                                                        int* __restrict__ sk_status, int sk_spin_limit,
                                                        unsigned long long* __restrict__ trace, const int* __restrict__ sk_bnd) {
    conv_dma_body<BM, BN, false, true, 4, 1>(src, nullptr, wt, bias, dst, nullptr, dcode, nullptr, nullptr, nullptr, nullptr, M, Ps, Pd, K, N, N, 0,
                                             src_bytes, side_bytes, ntiles, T_arg, segs, sk_mp, sk_part, sk_flag, sk_status, sk_spin_limit, trace, sk_bnd, nullptr);
}

// ARITH = 1 forms (round 6; conv_dma_body's header): the stream-K kernel of the plain 7-tap convolutions (single or pair) and of the
// dense one-tap GEMMs on the three-way bf16 split.  128-row tiles, waves 4 x NW/4, one workgroup per CU (3 x 40 KB ring at 128 x 128).
template <int BM, int BN, int NW>
__global__ __launch_bounds__(64 * NW) void k_conv_b3_sk(const float* __restrict__ src, const float* __restrict__ src2,
                                                         const float* __restrict__ wt, const float* __restrict__ bias,
                                                         float* __restrict__ dst, float* __restrict__ dst2,
                                                         const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                         const float* __restrict__ side2, const int32_t* __restrict__ perm, int M, int Ps,
                                                         int Pd, int K, int N, int N0, int n_slots, unsigned src_bytes,
                                                         unsigned side_bytes, int ntiles, int T_arg, const RowSegs segs, int sk_mp,
                                                         float* __restrict__ sk_part, int* __restrict__ sk_flag,
                                                         int* __restrict__ sk_status, int sk_spin_limit,
                                                         unsigned long long* __restrict__ trace, const int* __restrict__ sk_bnd) {
    conv_dma_body<BM, BN, false, true, NW, 7, 1>(src, src2, wt, bias, dst, dst2, dcode, side, side2, nullptr, nullptr, M, Ps, Pd, K, N, N0, n_slots,
                                                 src_bytes, side_bytes, ntiles, T_arg, segs, sk_mp, sk_part, sk_flag, sk_status, sk_spin_limit, trace, sk_bnd, nullptr);
}
// ... a SINGLE convolution (no second source / destination tensor) says so at compile time, as k_conv_single_sk does
template <int BM, int BN, int NW>
__global__ __launch_bounds__(64 * NW) void k_conv_b3_single_sk(const float* __restrict__ src, const float* __restrict__ src2,
                                                                const float* __restrict__ wt, const float* __restrict__ bias,
                                                                float* __restrict__ dst, float* __restrict__ dst2,
                                                                const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                                const float* __restrict__ side2, const int32_t* __restrict__ perm, int M, int Ps,
                                                                int Pd, int K, int N, int N0, int n_slots, unsigned src_bytes,
                                                                unsigned side_bytes, int ntiles, int T_arg, const RowSegs segs, int sk_mp,
                                                                float* __restrict__ sk_part, int* __restrict__ sk_flag,
                                                                int* __restrict__ sk_status, int sk_spin_limit,
                                                                unsigned long long* __restrict__ trace, const int* __restrict__ sk_bnd) {
    conv_dma_body<BM, BN, false, true, NW, 7, 1>(src, nullptr, wt, bias, dst, nullptr, dcode, side, nullptr, nullptr, nullptr, M, Ps, Pd, K, N, N, n_slots,
                                                 src_bytes, side_bytes, ntiles, T_arg, segs, sk_mp, sk_part, sk_flag, sk_status, sk_spin_limit, trace, sk_bnd, nullptr);
}
template <int BM, int BN, int NW>
__global__ __launch_bounds__(64 * NW) void k_conv_b3_dense_sk(const float* __restrict__ src, const float* __restrict__ src2,
                                                               const float* __restrict__ wt, const float* __restrict__ bias,
                                                               float* __restrict__ dst, float* __restrict__ dst2,
                                                               const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                               const float* __restrict__ side2, const int32_t* __restrict__ perm, int M, int Ps,
                                                               int Pd, int K, int N, int N0, int n_slots, unsigned src_bytes,
                                                               unsigned side_bytes, int ntiles, int T_arg, const RowSegs segs, int sk_mp,
                                                               float* __restrict__ sk_part, int* __restrict__ sk_flag,
                                                               int* __restrict__ sk_status, int sk_spin_limit,
                                                               unsigned long long* __restrict__ trace, const int* __restrict__ sk_bnd) {
    conv_dma_body<BM, BN, false, true, NW, 1, 1>(src, nullptr, wt, bias, dst, nullptr, dcode, nullptr, nullptr, nullptr, nullptr, M, Ps, Pd, K, N, N, 0,
                                                 src_bytes, side_bytes, ntiles, T_arg, segs, sk_mp, sk_part, sk_flag, sk_status, sk_spin_limit, trace, sk_bnd, nullptr);
}

// ... and the whole-tile form for the masked launches (stride-2 data gradients: rows permuted into tap-set classes, 1 - 2 taps per tile,
// tiles dealt to the workgroups by their step counts)
template <int BM, int BN, int NW>
__global__ __launch_bounds__(64 * NW) void k_conv_b3(const float* __restrict__ src, const float* __restrict__ src2,
                                                      const float* __restrict__ wt, const float* __restrict__ bias,
                                                      float* __restrict__ dst, float* __restrict__ dst2,
                                                      const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                      const float* __restrict__ side2, const int32_t* __restrict__ perm,
                                                      const uint32_t* __restrict__ mask32, int M, int Ps, int Pd, int K, int N, int N0,
                                                      int n_slots, unsigned src_bytes, unsigned side_bytes, int ntiles, int T_arg,
                                                      const RowSegs segs, unsigned long long* __restrict__ trace,
                                                      const int* __restrict__ tlist) {
    conv_dma_body<BM, BN, false, false, NW, 7, 1>(src, src2, wt, bias, dst, dst2, dcode, side, side2, perm, mask32, M, Ps, Pd, K, N, N0, n_slots,
                                                  src_bytes, side_bytes, ntiles, T_arg, segs, 4, nullptr, nullptr, nullptr, 0, trace, nullptr, tlist);
}

// Eight waves per workgroup (2 x 4 waves of 32 x 32; round 5): the same tile, LDS image, ring and tables as the four-wave kernels, the
// stage's DMA rows / metadata entries / epilogue columns dealt over twice the waves -- 3 instead of 6 DMA instructions and 16 instead
// of 32 MFMAs per wave and K-step, four waves per SIMD at two workgroups per CU to cover each other's barriers and stage issues
// (tools/mfma_ladder w / s: +2.6 - 3 % for this tile at the conv's real traffic, on 4 ms and on 0.2 ms launches).
template <int BM, int BN, bool SEG>
__global__ __launch_bounds__(512, 2) void k_conv_dma8(const float* __restrict__ src, const float* __restrict__ src2,
                                                       const float* __restrict__ wt, const float* __restrict__ bias,
                                                       float* __restrict__ dst, float* __restrict__ dst2,
                                                       const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                       const float* __restrict__ side2, const int32_t* __restrict__ perm,
                                                       const uint32_t* __restrict__ mask32, int M, int Ps, int Pd, int K, int N, int N0,
                                                       int n_slots, unsigned src_bytes, unsigned side_bytes, int ntiles, int T_arg,
                                                       const RowSegs segs, unsigned long long* __restrict__ trace,
                                                       const int* __restrict__ tlist) {
    conv_dma_body<BM, BN, SEG, false, 8>(src, src2, wt, bias, dst, dst2, dcode, side, side2, perm, mask32, M, Ps, Pd, K, N, N0, n_slots,
                                         src_bytes, side_bytes, ntiles, T_arg, segs, 4, nullptr, nullptr, nullptr, 0, trace, nullptr, tlist);
}

template <int BM, int BN, bool SEG>
__global__ __launch_bounds__(512, 2) void k_conv_dma_sk8(const float* __restrict__ src, const float* __restrict__ src2,
                                                          const float* __restrict__ wt, const float* __restrict__ bias,
                                                          float* __restrict__ dst, float* __restrict__ dst2,
                                                          const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                          const float* __restrict__ side2, const int32_t* __restrict__ perm, int M, int Ps,
                                                          int Pd, int K, int N, int N0, int n_slots, unsigned src_bytes,
                                                          unsigned side_bytes, int ntiles, int T_arg, const RowSegs segs, int sk_mp,
                                                          float* __restrict__ sk_part, int* __restrict__ sk_flag,
                                                          int* __restrict__ sk_status, int sk_spin_limit,
                                                          unsigned long long* __restrict__ trace, const int* __restrict__ sk_bnd) {
    conv_dma_body<BM, BN, SEG, true, 8>(src, src2, wt, bias, dst, dst2, dcode, side, side2, SEG ? perm : nullptr, nullptr, M, Ps, Pd, K, N, N0, n_slots,
                                        src_bytes, side_bytes, ntiles, T_arg, segs, sk_mp, sk_part, sk_flag, sk_status, sk_spin_limit, trace, sk_bnd, nullptr);
}

// dynamic LDS of k_conv_dma: A/B rings, offset table, destination-row table (row permutation only), bias (bias only)
static size_t conv_dma_lds(int bm, int bn, bool perm, bool bias, bool b3 = false) {
    return (size_t)3 * (bm * 128 + bn * (b3 ? 192 : 128)) + (size_t)2 * 7 * bm * 4 + (perm ? (size_t)3 * bm * 4 : 0) + (bias ? (size_t)3 * bn * 4 : 0) +
           16;   // + the stream-K form's block-uniform "partner lost" word
}

// Tile lists of a masked launch (stride-2 data gradients: a tile runs the taps of its rows' lattice class, 1 or 2 of the 7, so
// tiles differ 2 : 1 in K-steps).  Dealt round-robin (b, b + G, ...) the workgroups' step totals differ by up to 30 %
// (icn_debug_trace on the 128 -> 2x256 block's data gradient: exits from 156 to 278 us, 75 % of the block slots used).  Here
// each residue class's tiles (one XCD, one L2: the same set as before) are dealt longest-first to the workgroup that would
// finish it earliest given the speed factor of its arrival slot (sk_speed_factors), then sorted by id within a workgroup.
// Layout: [G + 1] offsets, then the ids.  Built once per (device, table set, launch shape) and kept.
const int* sk_speed_factors(int occ);
void build_tile_lists(const uint32_t* mask32_host, int M, int Pd, int bm, int ntn, int ntiles, int grid, int occ, std::vector<int>& h) {
    const int GL = grid / 8;
    const int* fac = sk_speed_factors(occ);
    h.assign((size_t)grid + 1 + ntiles, 0);
    std::vector<std::vector<int>> lists(grid);
    const int q = ntiles / 8, rem = ntiles % 8;
    for (int x = 0; x < 8; ++x) {
        // tiles of class x and their step counts (relative: taps in use; the k-chunk count scales all alike)
        std::vector<std::pair<int, int>> tl;              // (taps, tile id)
        for (int tile = x; tile < ntiles; tile += 8) {
            const int sw = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + tile / 8;   // as tile_origin
            const int m0 = (sw / ntn) * bm;
            unsigned mk = 0;
            for (int g2 = 0; g2 < bm / 32; ++g2)
                if (m0 + 32 * g2 < M) mk |= mask32_host[((m0 + 32 * g2) % Pd) >> 5];
            tl.emplace_back(std::max(1, __builtin_popcount(mk)), tile);
        }
        std::stable_sort(tl.begin(), tl.end(), [](const std::pair<int, int>& u, const std::pair<int, int>& v) { return u.first > v.first; });
        std::vector<double> load(GL, 0.0);
        for (const auto& e : tl) {
            int best = 0;
            double best_t = 0;
            for (int bl = 0; bl < GL; ++bl) {
                const double f = fac ? fac[(int)((long)bl * occ / GL)] / 1000.0 : 1.0;
                const double tdone = (load[bl] + e.first) / f;
                if (bl == 0 || tdone < best_t) { best = bl; best_t = tdone; }
            }
            load[best] += e.first;
            lists[best * 8 + x].push_back(e.second);
        }
    }
    int off = 0;
    for (int b = 0; b < grid; ++b) {
        std::sort(lists[b].begin(), lists[b].end());
        h[b] = off;
        for (int id : lists[b]) h[(size_t)grid + 1 + off++] = id;
    }
    h[grid] = off;
}
static const int* tile_lists(const GatherGemmArgs& a, int bm, int bn, int ntiles, int grid, int occ) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int, int, int>, int*> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int ntn = a.N / bn;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(dev, a.mask_key, a.M, bm, ntn, grid, a.Pd);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    std::vector<int> h;
    build_tile_lists(a.mask32_host, a.M, a.Pd, bm, ntn, ntiles, grid, occ, h);
    int* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("icn: cannot upload the tile lists");
    return cache.emplace(key, d).first->second;
}


// The DMA kernels address sources, side buffers and (since the buffer-store epilogue) destinations through 32-bit buffer
// offsets.  launch_gather_gemm_auto only routes here after conv_dma_usable(); a direct caller that skipped it gets an
// exception instead of wrapped offsets (silently dropped or misplaced stores).
static void check_dma_ranges(const GatherGemmArgs& a) {
    const size_t Ks = a.src2 ? a.K / 2 : a.K;
    const size_t nb = a.segs.nseg > 0 ? (size_t)a.segs.B : (size_t)(a.M / a.Pd);
    const size_t dst_bytes = nb * a.Pd * (size_t)std::max(a.dst2 ? a.N0 : a.N, a.dst2 ? a.N - a.N0 : 0) * 4;
    if (nb * a.Ps * Ks * 4 >= ((size_t)1 << 31) || nb * a.n_slots * Ks * 4 >= ((size_t)1 << 30) || dst_bytes >= (size_t)0xFFF00000u)
        throw std::invalid_argument("icn: tensor beyond the LDS-DMA kernels' 32-bit buffer offsets");
}

template <int BM, int BN, bool SEG, int NW = 4>
static void launch_conv_dma_t(const GatherGemmArgs& a, int occ, hipStream_t s) {
    constexpr auto kern = [] {
        if constexpr (NW == 8) return &k_conv_dma8<BM, BN, SEG>;
        else return &k_conv_dma<BM, BN, SEG>;
    }();
    check_dma_ranges(a);
    const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);   // class-major rows: M is a multiple of 8 * BM
    int grid = std::min(ntiles, 256 * occ);              // (more blocks than slots: measured, no difference -- DESIGN 4.2)
    if (grid >= 8) grid -= grid % 8;                     // keep a block's tiles in one residue class mod 8 (one XCD)
    const size_t lds = conv_dma_lds(BM, BN, a.perm != nullptr, a.bias != nullptr);
    static std::atomic<uint64_t> attr_devices{0};        // LDS opt-in, once per device
    if (!((attr_devices.load(std::memory_order_relaxed) >> current_device_bit()) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devices.fetch_or((uint64_t)1 << current_device_bit(), std::memory_order_relaxed);
    }
    const int Ks = a.src2 ? a.K / 2 : a.K;
    const size_t nb = a.segs.nseg > 0 ? (size_t)a.segs.B : (size_t)(a.M / a.Pd);          // samples
    const unsigned src_bytes = (unsigned)(nb * a.Ps * Ks * 4);
    const unsigned side_bytes = (unsigned)(nb * a.n_slots * Ks * 4);
    const int* tlist = nullptr;
    if (!SEG && a.mask32 && a.mask32_host && a.mask_key && grid % 8 == 0 && !(dbg_flags() & 512))
        tlist = tile_lists(a, BM, BN, ntiles, grid, occ);
    prof_mark_begin(NW == 8 ? (SEG ? PROF_DMAS8_64x128 : PROF_DMA8_64x128)
                            : (BM == 64 ? (BN == 128 ? PROF_DMA_64x128 : PROF_DMA_64x64) : (BN == 128 ? PROF_DMA_128x128 : PROF_DMA_128x64)) +
                                  (SEG ? PROF_DMAS_128x128 - PROF_DMA_128x128 : 0),
                    a.algo_flops, s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, a.src, a.src2, a.wt, a.bias, a.dst, a.dst2,
                       a.dcode, a.n_slots > 0 ? a.side : nullptr, (a.n_slots > 0 && a.src2) ? a.side2 : nullptr, a.perm, a.mask32, a.M,
                       a.Ps, a.Pd, a.K, a.N, a.dst2 ? a.N0 : a.N, a.n_slots, src_bytes, side_bytes, ntiles, a.T > 0 ? a.T : 7, a.segs,
                       g_trace_cap >= (size_t)grid * 8 ? g_trace : nullptr, tlist);
    prof_mark_end(s);
}

// ---- stream-K form (k_conv_dma_sk) -------------------------------------------------------------------------------------------
size_t conv_sk_part_bytes() { return (size_t)512 * 64 * 128 * sizeof(float); }   // 2 x 256 slots of 64x128 (>= 3 x 256 of 64x64)

// Would the stream-K kernel split anything for this launch?  (XCD 0 has the most tiles; the kernel plans per XCD.)
static bool conv_sk_splits(long ntiles, int slots, int nk) {
    if (slots % 8 != 0 || ntiles < 8) return false;
    return sk_plan((int)((ntiles + 7) / 8), slots / 8, nk).nsk > 0;
}
// a piece of a split tile must run >= 4 K-steps (the ring's fill and the metadata look-ahead)
constexpr int CONV_SK_MIN_PIECE = 4;
// K-steps of a tile of a stream-K-eligible launch (every tile runs the same taps): taps * k-chunks
static int conv_sk_steps(const GatherGemmArgs& a) { return (a.segs.nseg > 0 ? __builtin_popcount(a.segs.mask[0]) : 7) * (a.K / BK); }
static bool conv_sk_eligible(const GatherGemmArgs& a) {
    if (a.sk_part == nullptr || a.sk_flag == nullptr || a.mask32 != nullptr || (dbg_flags() & 128)) return false;
    if (a.segs.nseg == 0) return a.perm == nullptr && (a.T == 0 || a.T == 7);           // plain 7-tap convolution
    return a.segs.nseg == 1 && a.T == 1 && a.segs.mask[0] == 1u;                          // one-tap dense GEMM
}

const int* sk_speed_factors(int occ) {
    // measured on MI355X with equal shares (tools/trace_conv_blocks.py): see sk_boundaries
    static int fac2[2] = {1060, 940}, fac3[3] = {1150, 1020, 830};
    static const int parsed = [] {
        const char* e = getenv("ICN_SK_FAC");
        if (!e) return 1;
        int v[3] = {0, 0, 0};
        const int n = sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
        if (n == 2) { fac2[0] = v[0]; fac2[1] = v[1]; }
        if (n == 3) { fac3[0] = v[0]; fac3[1] = v[1]; fac3[2] = v[2]; }
        return n >= 2 ? 1 : 0;                            // "0" (or anything else): off
    }();
    if (!parsed) return nullptr;
    return occ == 2 ? fac2 : (occ == 3 ? fac3 : nullptr);
}

// Device copy of a launch shape's stream-K boundary tables (sk_tables), built on first use and kept: immutable per
// (device, tiles, grid, steps per tile), like the gather tables.
static const int* sk_boundary_tables(int ntiles, int grid, int S, int occ) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, int*> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(dev, ntiles, grid, S);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    std::vector<int> h(2 * (grid / 8 + 1));
    sk_tables(ntiles, grid, S, CONV_SK_MIN_PIECE, occ, sk_speed_factors(occ), h.data());
    int* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("icn: cannot upload the stream-K boundary tables");
    return cache.emplace(key, d).first->second;
}

template <int BM, int BN, bool SEG, int NW = 4, bool DENSE = false>
static void launch_conv_dma_sk_t(const GatherGemmArgs& a, int occ, hipStream_t s) {
    constexpr auto kern_general = [] {
        if constexpr (DENSE) return &k_conv_dense_sk<BM, BN>;
        else if constexpr (NW == 8) return &k_conv_dma_sk8<BM, BN, SEG>;
        else return &k_conv_dma_sk<BM, BN, SEG>;
    }();
    // single convolutions (no second source / destination) take the wrapper that knows it (debug flag 65536: the general kernel)
    const bool single = !SEG && !DENSE && NW == 4 && a.src2 == nullptr && a.dst2 == nullptr && !(dbg_flags() & 65536);
    auto kern = kern_general;
    if constexpr (!SEG && !DENSE && NW == 4) {
        if (single) kern = &k_conv_single_sk<BM, BN>;
    }
    check_dma_ranges(a);
    const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);   // SEG: M is the padded row count
    const int grid = 256 * occ;                           // every block slot of the chip: all of them resident at once
    if ((size_t)grid * BM * BN * sizeof(float) > conv_sk_part_bytes() || grid > CONV_SK_ERROR)
        throw std::invalid_argument("icn: stream-K grid beyond its scratch");
    const size_t lds = conv_dma_lds(BM, BN, a.perm != nullptr, a.bias != nullptr);
    static std::atomic<uint64_t> attr_devices[2] = {{0}, {0}};          // LDS opt-in, once per device and kernel (general / single)
    if (!((attr_devices[single].load(std::memory_order_relaxed) >> current_device_bit()) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devices[single].fetch_or((uint64_t)1 << current_device_bit(), std::memory_order_relaxed);
    }
    const int Ks = a.src2 ? a.K / 2 : a.K;
    const size_t nb = SEG ? (size_t)a.segs.B : (size_t)(a.M / a.Pd);
    const unsigned src_bytes = (unsigned)(nb * a.Ps * Ks * 4), side_bytes = (unsigned)(nb * a.n_slots * Ks * 4);
    prof_mark_begin(DENSE ? PROF_DENSEK_64x128 + (BN == 128 ? 0 : 1)
                          : single ? PROF_SINGLEK_64x128 + (BN == 128 ? 0 : 1)
                          : NW == 8 ? (SEG ? PROF_DMAKS8_64x128 : PROF_DMAK8_64x128) : (SEG ? PROF_DMAKS_64x128 : PROF_DMAK_64x128) + (BN == 128 ? 0 : 1),
                    a.algo_flops, s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, a.src, a.src2, a.wt, a.bias, a.dst, a.dst2, a.dcode,
                       a.n_slots > 0 ? a.side : nullptr, (a.n_slots > 0 && a.src2) ? a.side2 : nullptr, a.perm, a.M, a.Ps, a.Pd, a.K,
                       a.N, a.dst2 ? a.N0 : a.N, a.n_slots, src_bytes, side_bytes, ntiles, a.T > 0 ? a.T : 7, a.segs, CONV_SK_MIN_PIECE,
                       a.sk_part, a.sk_flag, device_status_word(), (dbg_flags() & 256) ? -1 : ((1 << 22) | ((dbg_flags() & 8192) ? (1 << 28) : 0)),
                       g_trace_cap >= (size_t)grid * 8 ? g_trace : nullptr,
                       sk_boundary_tables(ntiles, grid, DENSE ? a.K / BK : conv_sk_steps(a), occ));
    prof_mark_end(s);
}

// rows of a class-major launch for tile height bm: every segment padded to a multiple of 8 * bm rows (8 XCDs x one tile)
static long seg_rows(const RowSegs& sg, int bm, int* row0) {
    long row = 0;
    const long al = 8L * bm;
    for (int i = 0; i < sg.nseg; ++i) {
        if (row0) row0[i] = (int)row;
        row += ((long)sg.B * sg.cnt[i] + al - 1) / al * al;
    }
    return row;
}

template <int BM, int BN, int NW = 4>
static void launch_conv_dma(const GatherGemmArgs& a, int occ, hipStream_t s) {
    if (a.segs.nseg == 0) return launch_conv_dma_t<BM, BN, false, NW>(a, occ, s);
    GatherGemmArgs b = a;
    b.M = (int)seg_rows(a.segs, BM, b.segs.row0);
    launch_conv_dma_t<BM, BN, true, NW>(b, occ, s);
}

// dense one-tap GEMM with identity rows (conv_sk_eligible's second form): runs through the plain code path unless debug flag 32768
// (tests, A/B) keeps it on the class-major one
static bool conv_dense_plain(const GatherGemmArgs& a) {
    return a.segs.nseg == 1 && a.T == 1 && a.segs.mask[0] == 1u && a.perm == nullptr && a.segs.off[0] == 0 && a.segs.cnt[0] == a.Pd &&
           a.mask32 == nullptr && a.n_slots == 0 && a.src2 == nullptr && a.M == a.segs.B * a.Pd && a.Ps == a.Pd && !(dbg_flags() & 32768);
}
template <int BM, int BN, int NW = 4>
static void launch_conv_dma_sk(const GatherGemmArgs& a, int occ, hipStream_t s) {
    if (a.segs.nseg == 0) return launch_conv_dma_sk_t<BM, BN, false, NW>(a, occ, s);
    if (NW == 4 && conv_dense_plain(a)) {
        GatherGemmArgs d = a;                             // rows m = b * Pd + q as in a convolution; stores past M are range-checked away
        d.segs = RowSegs{};
        return launch_conv_dma_sk_t<BM, BN, false, 4, true>(d, occ, s);
    }
    GatherGemmArgs b = a;
    b.M = (int)seg_rows(a.segs, BM, b.segs.row0);
    launch_conv_dma_sk_t<BM, BN, true, NW>(b, occ, s);
}

// ---- ARITH = 1 launches (round 6) ---------------------------------------------------------------------------------------------
// Arithmetic of the channel-mixing contraction: 0 = exact fp32 MFMA, 1 = three-way bf16 split (conv_dma_body's header; the default).
// ICN_ARITH=f32|bf16x3 sets the process default; icn_set_arith (include/icn.h) changes it at run time (tests, A/B).
unsigned build_flags() { return ((unsigned)ICN_EXP & 0xffffu) | ((unsigned)ICN_CONV_WAVES_DEFAULT << 16) | ((unsigned)ICN_CHAIN_PRIO << 24); }
static std::atomic<int> g_arith{-1};
int arith_mode() {
    int v = g_arith.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("ICN_ARITH");
        if (e == nullptr || e[0] == 0 || strcmp(e, "bf16x3") == 0) v = 1;    // default since round 6: fp32-grade (tests/test_gpu_arith.py), 1.45x the exact kernels
        else if (strcmp(e, "f32") == 0) v = 0;
        else throw std::invalid_argument("icn: ICN_ARITH must be f32 or bf16x3");
        g_arith.store(v, std::memory_order_relaxed);
    }
    return v;
}
int set_arith_mode(int mode) {
    if (mode != 0 && mode != 1) throw std::invalid_argument("icn: arithmetic mode must be 0 (f32) or 1 (bf16x3)");
    const int old = arith_mode();
    g_arith.store(mode, std::memory_order_relaxed);
    return old;
}
// Column tile of the bf16 image this launch would read (128 or 64), or 0: the launch stays on the fp32 kernels.  Covered: what the
// stream-K kernels cover -- plain 7-tap convolutions (single / pair, forward and stride-1 data gradients) and the dense one-tap
// GEMMs with identity rows.  The caller (icn_api.cpp) asks BEFORE the prologue so that the weights are packed to match.
static bool conv_b3_masked(const GatherGemmArgs& a) {      // stride-2 data gradient: plain rows, permuted, tap masks, tile lists
    return a.segs.nseg == 0 && a.perm != nullptr && a.mask32 != nullptr && a.mask32_host != nullptr && a.mask_key != 0 && (a.T == 0 || a.T == 7) &&
           !(dbg_flags() & (512 | 262144));                 // (debug flag 262144: masked launches stay exact under bf16x3)
}
int conv_b3_bn(const GatherGemmArgs& a) {
    if (arith_mode() != 1 || !conv_dma_usable(a)) return 0;
    if (conv_b3_masked(a)) {
        if (a.N % 64 != 0 || a.K % BK != 0 || (size_t)7 * a.N * a.K * 6 >= ((size_t)1 << 31)) return 0;
        return a.N % 128 == 0 ? 128 : 64;
    }
    if (!conv_sk_eligible(a)) return 0;
    const bool dense = a.segs.nseg > 0;
    if (dense && !conv_dense_plain(a)) return 0;
    const int taps = dense ? 1 : 7;
    if (a.N % 64 != 0 || a.K % BK != 0 || taps * (a.K / BK) < 4) return 0;
    if ((size_t)taps * a.N * a.K * 6 >= ((size_t)1 << 31)) return 0;
    return a.N % 128 == 0 ? 128 : 64;
}

template <int BM, int BN, int NW, bool DENSE>
static void launch_conv_b3_sk_t(const GatherGemmArgs& a, hipStream_t s) {
    constexpr auto kern_general = [] {
        if constexpr (DENSE) return &k_conv_b3_dense_sk<BM, BN, NW>;
        else return &k_conv_b3_sk<BM, BN, NW>;
    }();
    const bool single = !DENSE && a.src2 == nullptr && a.dst2 == nullptr && !(dbg_flags() & 65536);
    auto kern = kern_general;
    if constexpr (!DENSE) {
        if (single) kern = &k_conv_b3_single_sk<BM, BN, NW>;
    }
    constexpr int occ = 1;
    check_dma_ranges(a);
    const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    const int grid = 256 * occ;
    if ((size_t)grid * BM * BN * sizeof(float) > conv_sk_part_bytes() || grid > CONV_SK_ERROR)
        throw std::invalid_argument("icn: stream-K grid beyond its scratch");
    const size_t lds = conv_dma_lds(BM, BN, false, a.bias != nullptr, true);
    static std::atomic<uint64_t> attr_devices[2] = {{0}, {0}};
    if (!((attr_devices[single].load(std::memory_order_relaxed) >> current_device_bit()) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devices[single].fetch_or((uint64_t)1 << current_device_bit(), std::memory_order_relaxed);
    }
    const int Ks = a.src2 ? a.K / 2 : a.K;
    const size_t nb = (size_t)(a.M / a.Pd);
    const unsigned src_bytes = (unsigned)(nb * a.Ps * Ks * 4), side_bytes = (unsigned)(nb * a.n_slots * Ks * 4);
    prof_mark_begin((DENSE ? PROF_B3DENSEK_128x128 : single ? PROF_B3SINGLEK_128x128 : PROF_B3K_128x128) + (BN == 128 ? 0 : 1), a.algo_flops, s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, a.src, a.src2, a.wt, a.bias, a.dst, a.dst2, a.dcode,
                       a.n_slots > 0 ? a.side : nullptr, (a.n_slots > 0 && a.src2) ? a.side2 : nullptr, a.perm, a.M, a.Ps, a.Pd, a.K,
                       a.N, a.dst2 ? a.N0 : a.N, a.n_slots, src_bytes, side_bytes, ntiles, DENSE ? 1 : 7, RowSegs{}, CONV_SK_MIN_PIECE,
                       a.sk_part, a.sk_flag, device_status_word(), (dbg_flags() & 256) ? -1 : ((1 << 22) | ((dbg_flags() & 8192) ? (1 << 28) : 0)),
                       g_trace_cap >= (size_t)grid * 8 ? g_trace : nullptr,
                       sk_boundary_tables(ntiles, grid, DENSE ? a.K / BK : 7 * (a.K / BK), occ));
    prof_mark_end(s);
}
template <int BM, int BN, int NW>
static void launch_conv_b3_masked_t(const GatherGemmArgs& a, hipStream_t s) {
    constexpr auto kern = &k_conv_b3<BM, BN, NW>;
    constexpr int occ = 1;
    check_dma_ranges(a);
    const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    int grid = std::min(ntiles, 256 * occ);
    if (grid >= 8) grid -= grid % 8;
    const size_t lds = conv_dma_lds(BM, BN, true, a.bias != nullptr, true);
    static std::atomic<uint64_t> attr_devices{0};
    if (!((attr_devices.load(std::memory_order_relaxed) >> current_device_bit()) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devices.fetch_or((uint64_t)1 << current_device_bit(), std::memory_order_relaxed);
    }
    const int Ks = a.src2 ? a.K / 2 : a.K;
    const size_t nb = (size_t)(a.M / a.Pd);
    const unsigned src_bytes = (unsigned)(nb * a.Ps * Ks * 4), side_bytes = (unsigned)(nb * a.n_slots * Ks * 4);
    const int* tlist = grid % 8 == 0 ? tile_lists(a, BM, BN, ntiles, grid, occ) : nullptr;
    prof_mark_begin(PROF_B3_MASKED, a.algo_flops, s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, a.src, a.src2, a.wt, a.bias, a.dst, a.dst2,
                       a.dcode, a.n_slots > 0 ? a.side : nullptr, (a.n_slots > 0 && a.src2) ? a.side2 : nullptr, a.perm, a.mask32, a.M,
                       a.Ps, a.Pd, a.K, a.N, a.dst2 ? a.N0 : a.N, a.n_slots, src_bytes, side_bytes, ntiles, 7, RowSegs{},
                       g_trace_cap >= (size_t)grid * 8 ? g_trace : nullptr, tlist);
    prof_mark_end(s);
}
static void launch_conv_b3(const GatherGemmArgs& a, hipStream_t s) {
    const int bn = conv_b3_bn(a);
    if (bn == 0 || bn != a.arith_bn) throw std::invalid_argument("icn: bf16x3 launch whose weights were packed for another tile");
    if (conv_b3_masked(a)) return bn == 128 ? launch_conv_b3_masked_t<128, 128, 8>(a, s) : launch_conv_b3_masked_t<128, 64, 4>(a, s);
    const bool dense = a.segs.nseg > 0;
    if (bn == 128) return dense ? launch_conv_b3_sk_t<128, 128, 8, true>(a, s) : launch_conv_b3_sk_t<128, 128, 8, false>(a, s);
    return dense ? launch_conv_b3_sk_t<128, 64, 4, true>(a, s) : launch_conv_b3_sk_t<128, 64, 4, false>(a, s);
}

// Waves per workgroup of the 64 x 128 tile's launches: 8 (2 x 4 waves of 32 x 32: k_conv_dma8 / k_conv_dma_sk8) or 4 (2 x 2 waves
// of 32 x 64); the other tiles have the four-wave form only.  Default 4: measured in round 5 (profiles/r05_ladder_production_rungs.txt)
// the eight-wave form is +1 % on the dominant class alone, -2 % on the decoder heads' dense GEMMs and -0.9 % in the training step.
// Debug flag 16384 (tests, in-process) or ICN_CONV_WAVES=8 selects it for every 64 x 128 launch; ICN_CONV_WAVES=84 / 80 for the
// plain (SEG = false) launches only / for launches with a bias (the forward convolutions) only (A/B runs).
static int conv_waves_64x128(const GatherGemmArgs& a) {
    static const int mode = [] {
        const char* e = getenv("ICN_CONV_WAVES");
        const int v = e ? atoi(e) : ICN_CONV_WAVES_DEFAULT;
        if (v != 4 && v != 8 && v != 84 && v != 80) throw std::invalid_argument("icn: ICN_CONV_WAVES must be 4, 8, 84 or 80");
        return v;
    }();
    if (dbg_flags() & 16384) return 8;
    if (mode == 84) return a.segs.nseg == 0 ? 8 : 4;
    if (mode == 80) return (a.segs.nseg == 0 && a.bias != nullptr) ? 8 : 4;
    return mode;
}

bool conv_dma_usable(const GatherGemmArgs& a) {
    if (dbg_flags() & 16) return false;
    const int Ks = a.src2 ? a.K / 2 : a.K;
    const size_t nb = a.segs.nseg > 0 ? (size_t)a.segs.B : (size_t)(a.M / a.Pd);
    const size_t src_bytes = nb * a.Ps * Ks * 4, wt_bytes = (size_t)(a.T > 0 ? a.T : 7) * a.N * a.K * 4;
    const size_t side_bytes = nb * a.n_slots * Ks * 4;
    if (a.segs.nseg > 0) {
        // class-major rows: permuted, every tile's tap mask comes from its segment (1..7 taps); the launch pads the segments
        // (no row permutation: only the single-segment identity layout of the dense GEMMs -- GEMM row m IS output row m)
        const bool ident = a.perm == nullptr && a.segs.nseg == 1 && a.segs.off[0] == 0 && a.segs.cnt[0] == a.Pd;
        if ((a.perm == nullptr && !ident) || a.segs.nseg > MAX_SEGS || a.segs.B < 1 || seg_rows(a.segs, 128, nullptr) >= (1L << 31)) return false;
        for (int i = 0; i < a.segs.nseg; ++i) {
            const int pc = __builtin_popcount(a.segs.mask[i]);
            if (pc < 1 || pc > 7 || pc * (a.K / BK) < 4 || a.segs.cnt[i] < 1) return false;
        }
    }
    if (a.dcode == nullptr || (a.n_slots > 0 && (a.side == nullptr || (a.src2 && a.side2 == nullptr)))) return false;
    if (side_bytes >= ((size_t)1 << 30)) return false;
    if (a.src2 && Ks % BK != 0) return false;
    if (a.dst2 && (a.N0 % 64 != 0 || (a.N - a.N0) % 64 != 0)) return false;
    // The cross-tile pipeline needs >= 4 K-steps per tile: the next tile's metadata is fetched in the tile's first
    // step and published by that step's barrier, and the DMA pointer (2 steps ahead) prefetches the next stage's row
    // offsets one step earlier still, i.e. in step S-3 >= 1.  With tap masks a tile may use a single tap.
    if (a.segs.nseg == 0 && (a.mask32 ? 1 : 7) * (a.K / BK) < 4) return false;
    // the epilogue addresses the outputs through 32-bit buffer offsets (rows past M must land beyond the buffer, not wrap)
    const size_t dst_bytes = nb * a.Pd * (size_t)std::max(a.dst2 ? a.N0 : a.N, a.dst2 ? a.N - a.N0 : 0) * 4;
    if (dst_bytes >= (size_t)0xFFF00000u) return false;
    return src_bytes < ((size_t)1 << 31) && wt_bytes < ((size_t)1 << 31);
}

// Tile shape of the persistent kernel.  `occ` blocks share a CU's MFMA pipes and a block runs ceil(tiles / grid) tiles
// back to back, so a launch takes about ceil(tiles / (256 * occ)) * occ * BM * BN / eff.  eff is the measured relative
// throughput of the shapes on I5 / batch 36 layers (tools/bench_layers.py with ICN_TILE=0..3): 64-row tiles at two to
// three blocks per CU hide the per-step barrier best; the 128x128 tile only fits one block per CU (3-stage ring).
struct DmaCfg { int bm, bn, occ; double eff; };
static const DmaCfg kDma[] = {{128, 128, 1, 0.80}, {128, 64, 2, 0.92}, {64, 128, 2, 1.00}, {64, 64, 3, 0.95}};

static void launch_conv_dma_auto(const GatherGemmArgs& a, hipStream_t s) {
    int best = -1, best_occ = 1;
    double best_cost = 0;
    const char* force = getenv("ICN_TILE");          // developer override: 0..3 = index into kDma
    for (int i = 0; i < 4; ++i) {
        const DmaCfg& c = kDma[i];
        if (a.N % c.bn != 0) continue;
        // blocks per CU: the shape's design point, unless the epilogue tables push its LDS over 160 KB / occ
        const int occ = std::max(1, std::min(c.occ, (int)((160 * 1024) / conv_dma_lds(c.bm, c.bn, a.perm != nullptr, a.bias != nullptr))));
        const long rows = a.segs.nseg > 0 ? seg_rows(a.segs, c.bm, nullptr) : a.M;
        const long tiles = ((rows + c.bm - 1) / c.bm) * (a.N / c.bn);
        const long slots = 256L * occ;
        // rounds of tiles a block runs: whole rounds, unless the stream-K form evens out the last one (64-row tiles only)
        const bool sk = c.bm == 64 && conv_sk_eligible(a) && conv_sk_splits(tiles, (int)slots, std::max(1, conv_sk_steps(a) / CONV_SK_MIN_PIECE));
        const double rounds = sk ? (double)tiles / slots : (double)((tiles + slots - 1) / slots);
        double cost = rounds * occ * c.bm * c.bn / (c.eff * occ / c.occ);
        // stride-2 data gradients (tap masks: tiles of 1-2 taps, unequal lengths) balance better on the finer tile: the
        // 128 -> 2x256 block's 258 against 281 us (tools/bench_layers.py --model under ICN_TILE = 3 / 2)
        if (a.mask32 && c.bn == 128) cost *= 1.1;
        if (force && atoi(force) == i) cost = -1;
        if (best < 0 || cost < best_cost) { best = i; best_cost = cost; best_occ = occ; }
    }
    if (best >= 2 && conv_sk_eligible(a)) {
        const DmaCfg& c = kDma[best];
        const long rows = a.segs.nseg > 0 ? seg_rows(a.segs, c.bm, nullptr) : a.M;
        const long tiles = ((rows + c.bm - 1) / c.bm) * (a.N / c.bn);
        if (conv_sk_splits(tiles, 256 * best_occ, std::max(1, conv_sk_steps(a) / CONV_SK_MIN_PIECE)))
            return best == 2 ? (conv_waves_64x128(a) == 8 ? launch_conv_dma_sk<64, 128, 8>(a, best_occ, s) : launch_conv_dma_sk<64, 128>(a, best_occ, s))
                             : launch_conv_dma_sk<64, 64>(a, best_occ, s);
    }
    switch (best) {
        case 0: return launch_conv_dma<128, 128>(a, best_occ, s);
        case 1: return launch_conv_dma<128, 64>(a, best_occ, s);
        case 2: return conv_waves_64x128(a) == 8 ? launch_conv_dma<64, 128, 8>(a, best_occ, s) : launch_conv_dma<64, 128>(a, best_occ, s);
        default: return launch_conv_dma<64, 64>(a, best_occ, s);
    }
}

// tile of the fall-back kernel (k_gather_gemm: row permutation / tap masks, or tensors beyond the DMA's 2 GiB range)
static int pick_tile(int M, int N, int E) {
    static const struct { int bm, bn, occ; double eff; } cfg[] = {{128, 128, 2, 1.00}, {128, 64, 3, 0.96}, {64, 128, 3, 0.96}, {64, 64, 4, 0.88}};
    int best = -1;
    double best_cost = 0;
    for (int i = 0; i < 4; ++i) {
        if (N % cfg[i].bn != 0) continue;
        const int occ = (E > 1 && cfg[i].bm == 128 && cfg[i].bn != 128) ? 2 : cfg[i].occ;   // the code table costs LDS
        const long tiles = (long)((M + cfg[i].bm - 1) / cfg[i].bm) * (N / cfg[i].bn), slots = 256L * occ;
        const double cost = (double)((tiles + slots - 1) / slots) * occ * cfg[i].bm * cfg[i].bn / cfg[i].eff;
        if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    return best;
}

void launch_gather_gemm_auto(const GatherGemmArgs& a, hipStream_t s) {
    if (a.arith_bn != 0) return launch_conv_b3(a, s);      // (set by the caller from conv_b3_bn, with weights packed to match)
    if (conv_dma_usable(a)) return launch_conv_dma_auto(a, s);
    if (a.src2 || a.dst2) throw std::invalid_argument("icn: pair gather-GEMM outside the LDS-DMA kernel's limits");
    if (a.segs.nseg > 0 || a.T > 7) throw std::invalid_argument("icn: composite gather-GEMM outside the LDS-DMA kernel's limits");
    switch (pick_tile(a.M, a.N, a.E)) {
        case 0: return launch_gather_gemm<128, 128>(a, s);
        case 1: return launch_gather_gemm<128, 64>(a, s);
        case 2: return launch_gather_gemm<64, 128>(a, s);
        default: return launch_gather_gemm<64, 64>(a, s);
    }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient:  partial[s][t][ci][co] = sum_{m in split s} x[gather_t(m)][ci] * dy[m][co]
//                   bias_partial[s][co]   = sum_{m in split s} dy[m][co]          (blocks with t == 0, ci0 == 0)
// ---------------------------------------------------------------------------------------------------------
template <int BI, int BJ>
__global__ __launch_bounds__(256) void k_wgrad(
    const float* __restrict__ x,        // (B, Ps, Cin)
    const float* __restrict__ dy,       // (B, Pd, Cout)
    const int32_t* __restrict__ idx,    // forward table [7][Pd]
    float* __restrict__ partial,        // [S][7][Cin][Cout]
    float* __restrict__ bias_partial,   // [S][Cout] or null
    int M, int Ps, int Pd, int Cin, int Cout, int ns, int rows_per_split, int n_splits) {
    constexpr int TI = BI / 64, TJ = BJ / 64;
    constexpr int CI = BI / 4, CJ = BJ / 4;             // 16-byte chunks per row
    constexpr int RI = 32 * CI / 256, RJ = 32 * CJ / 256;   // chunks per thread per stage (32 rows per stage)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Xs = reinterpret_cast<float*>(smem);          // [2][32][BI]
    float* Ys = Xs + 2 * 32 * BI;                        // [2][32][BJ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    // XCD-aware block order: the `group` = (ci tile, co tile, tap) blocks of one row split all read the same x / dy
    // rows.  Blocks b and b + 8 share an XCD (and its L2), so walk each XCD through one split at a time; dispatched
    // round-robin instead, the 7 taps of a split land on 7 different L2s and every row is fetched from HBM 7 times.
    const int ntj = Cout / BJ;
    const int group = (Cin / BI) * ntj * 7;              // blocks per row split
    const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
    const int split = (j / group) * 8 + xcd, g = j % group;
    if (split >= n_splits) return;                       // grid is rounded up to whole XCD rounds
    const int t = g % 7, tile = g / 7;
    const int ci0 = (tile / ntj) * BI, co0 = (tile % ntj) * BJ;
    const int m_begin = split * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const bool do_bias = bias_partial != nullptr && t == 0 && ci0 == 0;   // block-uniform
    float bsum = 0.f;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 rx[RI], ry[RJ];
    auto stage_load = [&](int mb) {
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const int c = tid + 256 * i, row = c / CI, ch = c % CI;
            const int m = mb + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
                const int b = m / Pd, p = m % Pd;
                const int32_t code = idx[(size_t)t * Pd + p];
                if (code >= 0) v = ld4(x + ((size_t)b * Ps + code) * Cin + ci0 + 4 * ch);
                else if (code <= -2) v = pole_mean4(x, b, Ps, ns, -2 - code, Cin, ci0 + 4 * ch);
            }
            rx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RJ; ++i) {
            const int c = tid + 256 * i, row = c / CJ, ch = c % CJ;
            const int m = mb + row;
            ry[i] = m < m_end ? ld4(dy + (size_t)m * Cout + co0 + 4 * ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const int c = tid + 256 * i;
            *reinterpret_cast<f32x4*>(Xs + buf * 32 * BI + 4 * c) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < RJ; ++i) {
            const int c = tid + 256 * i;
            *reinterpret_cast<f32x4*>(Ys + buf * 32 * BJ + 4 * c) = ry[i];
        }
    };

    const int nsteps = (m_end - m_begin + 31) / 32;
    if (nsteps > 0) {
        stage_load(m_begin);
        stage_write(0);
    }
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const bool more = step + 1 < nsteps;
        if (more) stage_load(m_begin + (step + 1) * 32);
        const float* xa = Xs + buf * 32 * BI + wr * (BI / 2) + l31;
        const float* yb = Ys + buf * 32 * BJ + wc * (BJ / 2) + l31;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = 2 * k2 + h;
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = xa[k * BI + i * 32];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = yb[k * BJ + j * 32];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias && tid < BJ) {
#pragma unroll
            for (int k = 0; k < 32; ++k) bsum += Ys[buf * 32 * BJ + k * BJ + tid];
        }
        if (more) stage_write(buf ^ 1);
        __syncthreads();
    }

    float* out = partial + ((size_t)split * 7 + t) * Cin * Cout;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wr * (BI / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int co = co0 + wc * (BJ / 2) + j * 32 + l31;
                out[(size_t)ci * Cout + co] = acc[i][j][r];
            }
    if (do_bias && tid < BJ) bias_partial[(size_t)split * Cout + co0 + tid] = bsum;
}

// ---------------------------------------------------------------------------------------------------------
// k_wgrad_dma: same math as k_wgrad, staged by LDS-DMA through a 3-stage ring of 16-row stages.
//   A stage = 16 consecutive output rows m: X[16][BI] (gathered input rows of tap t, channels ci0..) and Y[16][BJ]
//   (dy rows, channels co0..).  A lane's gather code (forward DmaTable) for stage s+3 is fetched at step s, turned
//   into a byte offset at step s+1 and used by the DMA of stage s+3 issued there; rows whose tap reads a pole are
//   DMA'd from the side buffer of pole means (k_conv_prologue), as in k_conv_dma.  The code fetch is itself an
//   LDS-DMA (one dword per lane into a wave-private LDS slot, read back with ds_read one step later): a load into
//   a VGPR would have to be tracked by the compiler, which then drains vmcnt to 0 every step.  It is issued BEFORE
//   the step's data DMA, so the counted vmcnt at the end of the step retires it together with the previous stage.
// ---------------------------------------------------------------------------------------------------------
constexpr int WG_RS = 16;   // rows per stage

// ARITH = 1 (round 6): a stage's WG_RS = 16 rows are one k-block of v_mfma_f32_32x32x16_bf16; a lane holds rows 8h .. 8h + 7 of its
// channel of x and of dy, both cut into three bf16 pieces in registers (split_bf16x3), six products per 32 x 32 block.
template <int BI, int BJ, bool DENSE, int ARITH = 0>
__device__ __forceinline__ void wgrad_dma_body(
    const float* __restrict__ x,        // (B, Ps, Cin)
    const float* __restrict__ dy,       // (B, Pd, Cout0)
    const float* __restrict__ dy2,      // (B, Pd, Cout - Cout0): output channels Cout0.. (pair sharing x), or null
    const int32_t* __restrict__ dcode,  // forward DmaTable code [7][Pd]
    const float* __restrict__ side,     // (B, n_slots, Cin) pole means of x, or null
    float* __restrict__ partial,        // [S][7][Cin][Cout]
    float* __restrict__ bias_partial,   // [S][Cout] or null
    int M, int Ps, int Pd, int Cin, int Cout, int Cout0, int n_slots, int rows_per_split, int n_splits, unsigned x_bytes,
    unsigned side_bytes, int y_taps, unsigned long long* __restrict__ trace) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long tr_t0 = 0;
    if (trace) tr_t0 = __builtin_amdgcn_s_memrealtime();
    // y_taps = 7: dy is the tap-major aggregate g (M, 7, Cout) of icn_upconv_bwd -- tap t reads channels [t * Cout, (t+1) * Cout)
    // of a row of 7 * Cout floats -- and dcode is ONE table [Pd] shared by the taps (the rows of x are not shifted).
    constexpr int TI = BI / 64, TJ = BJ / 64;
    constexpr int LA = BI / 4, LB = BJ / 4;            // lanes (16-byte chunks) per row
    constexpr int RPA = 64 / LA, RPB = 64 / LB;        // rows per DMA instruction
    constexpr int NA = WG_RS / RPA / 4, NB = WG_RS / RPB / 4;   // DMA instructions per wave per stage (BI,BJ >= 64)
    constexpr int NDMA = NA + NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Xs = reinterpret_cast<float*>(smem);        // [3][WG_RS][BI]
    float* Ys = Xs + 3 * WG_RS * BI;                   // [3][WG_RS][BJ]
    int32_t* Cs = reinterpret_cast<int32_t*>(Ys + 3 * WG_RS * BJ);   // [2][4 waves][NA][64] gather codes (wave-private)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    // XCD-aware block order (see k_wgrad)
    const int ntj = Cout / BJ;
    const int group = (Cin / BI) * ntj * 7;
    const int xcd = blockIdx.x % 8, jb = blockIdx.x / 8;
    const int split = (jb / group) * 8 + xcd, g = jb % group;
    if (split >= n_splits) return;                       // grid is rounded up to whole XCD rounds
    const int t = g % 7, tile = g / 7;
    const int ci0 = (tile / ntj) * BI, co0 = (tile % ntj) * BJ;
    const int m_begin = split * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const int nsteps = m_end > m_begin ? (m_end - m_begin + WG_RS - 1) / WG_RS : 0;   // (0: a split past the last row writes zeros)
    const bool do_bias = bias_partial != nullptr && t == 0 && ci0 == 0;   // block-uniform

    // the block's output-channel tile lies in one of the two dy tensors (block-uniform)
    const bool ysec = co0 >= Cout0;
    const int yC = y_taps ? y_taps * Cout : (ysec ? Cout - Cout0 : Cout0);           // its row stride / first channel there
    const int yc0 = y_taps ? t * Cout + co0 : (ysec ? co0 - Cout0 : co0);
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ysec ? dy2 : dy), 0, (unsigned)M * (unsigned)yC * 4u,
                                                          0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(side ? side : x), 0, side ? side_bytes : 0u,
                                                          0x00020000);
    const auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(dcode + (y_taps ? (size_t)0 : (size_t)t * Pd)), 0, Pd * 4,
                                                          0x00020000);

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum = 0.f;

    // lane-constant pieces
    const int arow = lane / LA, achunk = lane % LA;    // row within its DMA instruction, 16-byte chunk
    const int brow = lane / LB, bchunk = lane % LB;
    // per-lane row state of the CODE pointer (3 stages ahead of compute): row (wave + 4*i)*RPA + arow of that stage
    int c_b[NA], c_p[NA];                              // sample / pixel of the lane's rows at the code pointer
    int c_m0 = m_begin;                                // first row of the stage under the code pointer
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m_begin + (wave + 4 * i) * RPA + arow;
        c_b[i] = m / Pd;
        c_p[i] = m % Pd;
    }
    unsigned aoff[NA];
    int d_ring = 0, d_step = 0;                        // DMA pointer: ring slot and stage index
    int c_slot = 0;                                    // code ring slot the next fetch writes
    int issued = 0, p_exact = 1;

    // fetch the gather codes of the stage under the code pointer into code slot c_slot (c_p is always a valid pixel;
    // rows past m_end are sorted out in ICN_WG_MAKE_OFFSETS)
#define ICN_WG_FETCH_CODES() do { \
        _Pragma("unroll") \
        for (int i = 0; i < (DENSE ? 0 : NA); ++i) { \
            int32_t* dst_ = Cs + __builtin_amdgcn_readfirstlane(((c_slot * 4 + wave) * NA + i) * 64); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lds_ptr_t)dst_, 4, (unsigned)c_p[i] * 4u, 0, 0, 0); \
        } \
        c_slot ^= 1; \
    } while (0)
#define ICN_WG_ADVANCE_CODE_PTR() do { \
        c_m0 += WG_RS; \
        _Pragma("unroll") \
        for (int i = 0; i < (DENSE ? 0 : NA); ++i) { c_p[i] += WG_RS; while (c_p[i] >= Pd) { c_p[i] -= Pd; c_b[i] += 1; } } \
    } while (0)
    // codes (fetched one step ago into slot c_slot ^ 1, landed and retired by that step's counted wait) -> byte offsets;
    // uses the row state BEFORE it is advanced; bit 31: side buffer (pole mean) or nothing
#define ICN_WG_MAKE_OFFSETS() do { \
        _Pragma("unroll") \
        for (int i = 0; i < NA; ++i) { \
            const int m = c_m0 + (wave + 4 * i) * RPA + arow; \
            if constexpr (DENSE) { \
                aoff[i] = m < m_end ? ((unsigned)m * (unsigned)Cin + (unsigned)(ci0 + 4 * achunk)) * 4u : NOTHING_OFFSET; \
                continue; \
            } \
            const int32_t c = m < m_end ? Cs[(((c_slot ^ 1) * 4 + wave) * NA + i) * 64 + lane] : -1; \
            aoff[i] = c >= 0 ? ((unsigned)(c_b[i] * Ps + c) * (unsigned)Cin + (unsigned)(ci0 + 4 * achunk)) * 4u \
                    : c == -1 ? NOTHING_OFFSET \
                              : SIDE_FLAG | (((unsigned)(c_b[i] * n_slots + (-2 - c)) * (unsigned)Cin + (unsigned)(ci0 + 4 * achunk)) * 4u); \
        } \
    } while (0)
    // issue the DMA of stage d_step into ring slot d_ring (offsets prepared by ICN_WG_MAKE_OFFSETS)
#define ICN_WG_ISSUE() do { \
        issued = d_step < nsteps; \
        p_exact = 1; \
        if (issued) { \
            bool side_row = false; \
            _Pragma("unroll") \
            for (int i = 0; i < (DENSE ? 0 : NA); ++i) side_row |= (int)aoff[i] < (int)NOTHING_OFFSET; \
            if (DENSE || (ICN_EXP & 4) || __builtin_amdgcn_ballot_w64(side_row) == 0) { \
                _Pragma("unroll") \
                for (int i = 0; i < NA; ++i) { \
                    float* dst_ = Xs + __builtin_amdgcn_readfirstlane(d_ring * WG_RS * BI + (wave + 4 * i) * RPA * BI); \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst_, 16, aoff[i], 0, 0, 0); \
                } \
            } else { \
                p_exact = 0; \
                _Pragma("unroll") \
                for (int i = 0; i < NA; ++i) { \
                    float* dst_ = Xs + __builtin_amdgcn_readfirstlane(d_ring * WG_RS * BI + (wave + 4 * i) * RPA * BI); \
                    if ((int)aoff[i] >= (int)NOTHING_OFFSET) \
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst_, 16, aoff[i], 0, 0, 0); \
                    else \
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_s, (lds_ptr_t)dst_, 16, aoff[i] & ~SIDE_FLAG, 0, 0, 0); \
                } \
            } \
            const int y_soff = __builtin_amdgcn_readfirstlane((m_begin + d_step * WG_RS) * yC * 4); \
            _Pragma("unroll") \
            for (int i = 0; i < NB; ++i) { \
                float* dst_ = Ys + __builtin_amdgcn_readfirstlane(d_ring * WG_RS * BJ + (wave + 4 * i) * RPB * BJ); \
                const int row_ = (wave + 4 * i) * RPB + brow; \
                const unsigned voff_ = (m_begin + d_step * WG_RS + row_ < m_end) \
                                           ? ((unsigned)row_ * (unsigned)yC + (unsigned)(yc0 + 4 * bchunk)) * 4u : SIDE_FLAG; \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)dst_, 16, voff_, y_soff, 0, 0); \
            } \
            d_ring = d_ring == 2 ? 0 : d_ring + 1; \
            d_step += 1; \
        } \
    } while (0)
    // end of a step: the previous stage has landed (and the codes fetched before this step's DMA have arrived); publish
#define ICN_WG_RETIRE_AND_PUBLISH() do { \
        if (issued && p_exact) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier(); \
    } while (0)

    // ring fill: stages 0 and 1; codes for stage 2 in flight
    ICN_WG_FETCH_CODES();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ICN_WG_MAKE_OFFSETS();
    ICN_WG_ADVANCE_CODE_PTR();
    ICN_WG_FETCH_CODES();                               // codes of stage 1 (before the DMA, retired with it)
    ICN_WG_ISSUE();
    ICN_WG_RETIRE_AND_PUBLISH();
    ICN_WG_MAKE_OFFSETS();
    ICN_WG_ADVANCE_CODE_PTR();
    ICN_WG_FETCH_CODES();                               // codes of stage 2, consumed at the top of step 0
    ICN_WG_ISSUE();
    ICN_WG_RETIRE_AND_PUBLISH();
    int c_ring = 0;
    for (int step = 0; step < nsteps; ++step) {
        // stage step+2: offsets from the codes fetched one step ago, next codes, then the DMA
        ICN_WG_MAKE_OFFSETS();
        ICN_WG_ADVANCE_CODE_PTR();
        ICN_WG_FETCH_CODES();                           // stage step+3 (issued before the DMA)
        ICN_WG_ISSUE();
        const float* xa = Xs + c_ring * WG_RS * BI + wr * (BI / 2) + l31;
        const float* yb = Ys + c_ring * WG_RS * BJ + wc * (BJ / 2) + l31;
        if constexpr (ARITH == 1) {
            static_assert(WG_RS == 16, "one bf16 k-block per stage");
            Pieces3 pa[TI], pb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] = xa[(8 * h + j) * BI + i * 32]; hi[j] = xa[(8 * h + 4 + j) * BI + i * 32]; }
                pa[i] = split_bf16x3(lo, hi);
            }
#pragma unroll
            for (int j2 = 0; j2 < TJ; ++j2) {
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] = yb[(8 * h + j) * BJ + j2 * 32]; hi[j] = yb[(8 * h + 4 + j) * BJ + j2 * 32]; }
                pb[j2] = split_bf16x3(lo, hi);
            }
#define ICN_MF16W(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), c_, 0, 0, 0)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    f32x16 c = acc[i][j];
                    c = ICN_MF16W(pa[i].p1, pb[j].p3, c);
                    c = ICN_MF16W(pa[i].p3, pb[j].p1, c);
                    c = ICN_MF16W(pa[i].p2, pb[j].p2, c);
                    c = ICN_MF16W(pa[i].p1, pb[j].p2, c);
                    c = ICN_MF16W(pa[i].p2, pb[j].p1, c);
                    c = ICN_MF16W(pa[i].p1, pb[j].p1, c);
                    acc[i][j] = c;
                }
#undef ICN_MF16W
        } else
#pragma unroll
        for (int k2 = 0; k2 < WG_RS / 2; ++k2) {
            const int k = 2 * k2 + h;
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = xa[k * BI + i * 32];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = yb[k * BJ + j * 32];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias && tid < BJ) {
#pragma unroll
            for (int k = 0; k < WG_RS; ++k) bsum += Ys[c_ring * WG_RS * BJ + k * BJ + tid];
        }
        ICN_WG_RETIRE_AND_PUBLISH();
        c_ring = c_ring == 2 ? 0 : c_ring + 1;
    }
#undef ICN_WG_FETCH_CODES
#undef ICN_WG_ADVANCE_CODE_PTR
#undef ICN_WG_MAKE_OFFSETS
#undef ICN_WG_ISSUE
#undef ICN_WG_RETIRE_AND_PUBLISH

    float* out = partial + ((size_t)split * 7 + t) * Cin * Cout;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wr * (BI / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int co = co0 + wc * (BJ / 2) + j * 32 + l31;
                out[(size_t)ci * Cout + co] = acc[i][j][r];
            }
    if (do_bias && tid < BJ) bias_partial[(size_t)split * Cout + co0 + tid] = bsum;
    if (trace && tid == 0) {
        unsigned long long* o = trace + (size_t)blockIdx.x * 8;
        o[0] = tr_t0; o[1] = tr_t0; o[2] = tr_t0; o[3] = 0; o[4] = __builtin_amdgcn_s_memrealtime(); o[5] = 0;
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);
    }
#endif
}

template <int BI, int BJ, int ARITH = 0>
__global__ __launch_bounds__(256) void k_wgrad_dma(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ dy2,
                                                   const int32_t* __restrict__ dcode, const float* __restrict__ side,
                                                   float* __restrict__ partial, float* __restrict__ bias_partial, int M, int Ps, int Pd,
                                                   int Cin, int Cout, int Cout0, int n_slots, int rows_per_split, int n_splits,
                                                   unsigned x_bytes, unsigned side_bytes, int y_taps, unsigned long long* __restrict__ trace) {
    wgrad_dma_body<BI, BJ, false, ARITH>(x, dy, dy2, dcode, side, partial, bias_partial, M, Ps, Pd, Cin, Cout, Cout0, n_slots, rows_per_split,
                                         n_splits, x_bytes, side_bytes, y_taps, trace);
}

// k_wgrad_dense: the decoder heads' dW_t = sum_s x[s]^T g_t[s] (y_taps = 7, identity rows, no pole means) through the same body with
// the gather compiled away: row offsets are arithmetic, so the code DMAs, their LDS read-back and the side-row vote are gone.
template <int BI, int BJ, int ARITH = 0>
__global__ __launch_bounds__(256) void k_wgrad_dense(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial,
                                                     float* __restrict__ bias_partial, int M, int P, int Cin, int Cout, int Cout0,
                                                     int rows_per_split, int n_splits, unsigned x_bytes,
                                                     unsigned long long* __restrict__ trace) {
    wgrad_dma_body<BI, BJ, true, ARITH>(x, g, nullptr, nullptr, nullptr, partial, bias_partial, M, P, P, Cin, Cout, Cout0, 0, rows_per_split,
                                        n_splits, x_bytes, 0u, 7, trace);
}

// ---------------------------------------------------------------------------------------------------------
// k_wgrad7 (round 4): ALL SEVEN TAPS of a 64 x 64 (ci, co) tile in one workgroup, the gathered x rows staged ONCE per patch.
//   k_wgrad_dma gives every (tap, ci tile, co tile, row split) its own short workgroup: the seven taps of a split gather seven
//   shifted copies of the same x rows (each row of x and dy travels L2 -> LDS 7 x the tiles' count), ~2300 workgroups of 18 steps
//   each pay their own start-up and 64 KB slab, and 330 slabs per layer go through HBM to a reduction kernel.  Here:
//   * a stage is one PATCH of 16 consecutive output pixels.  Its 7 x 16 gathered rows overlap (the in-row taps of neighbouring
//     pixels are each other's centre rows): the host lists the UNION of the patch's source rows once (icn_geometry.h:
//     build_wgrad7; <= 53 of 112 rows at stride 1) and, per (pixel, tap), the row's position in that list.  The union
//     (U x 256 B), the 16 dy rows (4 KB) and the position table (256 B) are brought in by LDS-DMA, 3-stage ring as before.
//   * the MFMA's A operand of tap t, pixel k is then read from LDS at row pos[k][t] -- an indirect ds_read_b32 per MFMA, its
//     offset unpacked from the 16 bytes of positions a lane fetches per pixel -- and the B operand (dy) once per pixel for all
//     seven taps: 56 MFMAs per wave and step on 7 x 16 accumulator registers.
//   * the launch is ONE round of 512 equal workgroups (two per CU), each owning a (ci, co) tile and a contiguous range of
//     patches: no dispatch tail, ~3 x fewer slabs, x / dy rows travel L2 -> LDS once per (co tile) / (ci tile) instead of 7 x.
// ---------------------------------------------------------------------------------------------------------
constexpr int W7_PX = 16;                    // = WG7_PX (icn_geometry.h)
constexpr int W7_ROWF = 64;                  // floats per staged row (the tile's 64 input channels) = WG7_ROW_BYTES / 4

// U: union rows per patch, a multiple of 4 (a DMA instruction brings 4 rows; ceil(U / 16) instructions per wave and stage);
// NST: ring stages (3: the DMA pointer two steps ahead, counted waits; 2: one step ahead -- the step's 56 MFMAs per wave cover
// the latency -- and a third less LDS)
// ARITH = 1 (round 6): the stage's 16 pixels are ONE k-block of v_mfma_f32_32x32x16_bf16 -- a lane holds pixels 8h .. 8h + 7 of its
// channel: the same 56 indirect ds_read_b32 per stage, every value cut into three bf16 pieces in registers (split_bf16x3; x AND dy are
// activations, so both operands are split here), six products per tap on the shared dy pieces: 42 MFMAs of 32 cycles instead of 56 of
// 64 per wave and stage, beside 352 VALU instructions (the split is then what bounds the stage, at about half the exact form's time).
template <int U, int NST, int ARITH = 0>
__global__ __launch_bounds__(256, 2) void k_wgrad7(
    const float* __restrict__ x,        // (B, Ps, Cin)
    const float* __restrict__ dy,       // (B, Pd, Cout0)
    const float* __restrict__ dy2,      // (B, Pd, Cout - Cout0): output channels Cout0.. (pair sharing x), or null
    const int32_t* __restrict__ urow,   // [Pd / 16][U] union-row codes (Wg7Table)
    const uint16_t* __restrict__ upos,  // [Pd / 16][16][8] positions (byte offsets into the staged union)
    const float* __restrict__ side,     // (B, n_slots, Cin) pole means of x, or null
    float* __restrict__ partial,        // [S][7][Cin][Cout]
    float* __restrict__ bias_partial,   // [S][Cout] or null
    int M, int Ps, int Pd, int Cin, int Cout, int Cout0, int n_slots, int patches_per_split, int n_splits, unsigned x_bytes,
    unsigned side_bytes, unsigned long long* __restrict__ trace) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NI = (U + 15) / 16;
    static_assert(U % 4 == 0 && U <= 128 && (NST == 2 || NST == 3), "at most two code DMAs (64 dwords each) per stage");
    // U % 16 != 0 (U = 56): the last DMA instruction exists in the first (U % 16) / 4 waves only; the counted waits follow
    constexpr int CW = U > 64 ? 128 : 64;                           // code words per stage
    unsigned long long tr_t0 = 0;
    if (trace) tr_t0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_F = U * W7_ROWF + W7_PX * 64 + 64;          // floats per ring slot: X union | Y | positions
    float* ring = reinterpret_cast<float*>(smem);                    // [NST][STAGE_F]
    int32_t* Cs = reinterpret_cast<int32_t*>(ring + NST * STAGE_F);  // [2][CW] union-row codes of the stage under the code pointer

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    // blocks b and b + 8 share an XCD: the (ci, co) tiles of one row split sit side by side on one XCD (they read the same rows)
    const int ntj = Cout / 64, tiles = (Cin / 64) * ntj;
    const int xcd = blockIdx.x % 8, jb = blockIdx.x / 8;
    const int split = (jb / tiles) * 8 + xcd, tile = jb % tiles;
    if (split >= n_splits) return;
    const int ci0 = (tile / ntj) * 64, co0 = (tile % ntj) * 64;
    const int pps = Pd / W7_PX;                                       // patches per sample
    const int p_begin = split * patches_per_split;
    const int p_end = min(M / W7_PX, p_begin + patches_per_split);
    const int nsteps = p_end > p_begin ? p_end - p_begin : 0;          // (0: a split past the last patch writes zeros)
    const bool do_bias = bias_partial != nullptr && ci0 == 0;         // block-uniform

    const bool ysec = co0 >= Cout0;
    const int yC = ysec ? Cout - Cout0 : Cout0;
    const int yc0 = ysec ? co0 - Cout0 : co0;
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ysec ? dy2 : dy), 0, (unsigned)M * (unsigned)yC * 4u, 0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(side ? side : x), 0, side ? side_bytes : 0u, 0x00020000);
    const auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(urow), 0, pps * U * 4, 0x00020000);
    const auto rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(upos), 0, pps * W7_PX * 16, 0x00020000);

    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const int lrow = lane >> 4, lchunk = lane & 15;                   // DMA: row within the instruction's 4 rows, 16-byte chunk
    // code pointer (3 stages ahead of compute) and DMA pointer (2 ahead): patch within its sample, sample
    int c_q = p_begin % pps, c_b = p_begin / pps;
    int d_q = c_q, d_step = 0, d_ring = 0, c_slot = 0;
    unsigned aoff[NI];
    int issued = 0, p_exact = 1;

#define ICN_W7_FETCH_CODES() do { \
        if (wave == 1 || (U > 64 && wave == 2)) { \
            const int w0_ = __builtin_amdgcn_readfirstlane((wave - 1) * 64); \
            int32_t* dst_ = Cs + __builtin_amdgcn_readfirstlane(c_slot * CW + w0_); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lds_ptr_t)dst_, 4, \
                                                     w0_ + lane < U ? (unsigned)(c_q * U + w0_ + lane) * 4u : SIDE_FLAG, 0, 0, 0); \
        } \
        c_slot ^= 1; \
    } while (0)
    // codes of the stage under the code pointer (fetched one step ago into slot c_slot ^ 1, published by that step's barrier)
    // -> byte offsets of this lane's rows; uses the pointer's sample BEFORE it is advanced
#define ICN_W7_MAKE_OFFSETS() do { \
        _Pragma("unroll") \
        for (int i = 0; i < NI; ++i) { \
            const int32_t c = (U % 16 == 0 || 4 * (wave + 4 * i) < U) ? Cs[(c_slot ^ 1) * CW + 4 * (wave + 4 * i) + lrow] : -1; \
            aoff[i] = c >= 0 ? ((unsigned)(c_b * Ps + c) * (unsigned)Cin + (unsigned)(ci0 + 4 * lchunk)) * 4u \
                    : c == -1 ? NOTHING_OFFSET \
                              : SIDE_FLAG | (((unsigned)(c_b * n_slots + (-2 - c)) * (unsigned)Cin + (unsigned)(ci0 + 4 * lchunk)) * 4u); \
        } \
    } while (0)
#define ICN_W7_ADVANCE_CODE_PTR() do { if (++c_q == pps) { c_q = 0; ++c_b; } } while (0)
    // DMA of stage d_step into ring slot d_ring: union rows (offsets from ICN_W7_MAKE_OFFSETS), dy rows, positions (wave 0)
#define ICN_W7_ISSUE() do { \
        issued = d_step < nsteps; \
        p_exact = 1; \
        if (issued) { \
            const int slot_off_ = __builtin_amdgcn_readfirstlane(d_ring * STAGE_F); \
            bool side_row = false; \
            _Pragma("unroll") \
            for (int i = 0; i < NI; ++i) side_row |= (int)aoff[i] < (int)NOTHING_OFFSET; \
            if ((ICN_EXP & 4) || __builtin_amdgcn_ballot_w64(side_row) == 0) { \
                _Pragma("unroll") \
                for (int i = 0; i < NI; ++i) { \
                    if (U % 16 != 0 && 4 * (wave + 4 * i) >= U) continue; \
                    float* dst_ = ring + __builtin_amdgcn_readfirstlane(slot_off_ + 4 * (wave + 4 * i) * W7_ROWF); \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst_, 16, aoff[i], 0, 0, 0); \
                } \
            } else { \
                p_exact = 0; \
                _Pragma("unroll") \
                for (int i = 0; i < NI; ++i) { \
                    if (U % 16 != 0 && 4 * (wave + 4 * i) >= U) continue; \
                    float* dst_ = ring + __builtin_amdgcn_readfirstlane(slot_off_ + 4 * (wave + 4 * i) * W7_ROWF); \
                    if ((int)aoff[i] >= (int)NOTHING_OFFSET) \
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)dst_, 16, aoff[i], 0, 0, 0); \
                    else \
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_s, (lds_ptr_t)dst_, 16, aoff[i] & ~SIDE_FLAG, 0, 0, 0); \
                } \
            } \
            { \
                const int y_soff = __builtin_amdgcn_readfirstlane((p_begin + d_step) * W7_PX * yC * 4); \
                float* dst_ = ring + __builtin_amdgcn_readfirstlane(slot_off_ + U * W7_ROWF + 4 * wave * 64); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_ptr_t)dst_, 16, \
                                                         ((unsigned)(4 * wave + lrow) * (unsigned)yC + (unsigned)(yc0 + 4 * lchunk)) * 4u, y_soff, 0, 0); \
            } \
            if (wave == 0) { \
                float* dst_ = ring + __builtin_amdgcn_readfirstlane(slot_off_ + U * W7_ROWF + W7_PX * 64); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_p, (lds_ptr_t)dst_, 4, (unsigned)lane * 4u, \
                                                         __builtin_amdgcn_readfirstlane(d_q * (W7_PX * 16)), 0, 0); \
            } \
            d_ring = d_ring == NST - 1 ? 0 : d_ring + 1; \
            d_step += 1; \
            if (++d_q == pps) d_q = 0; \
        } \
    } while (0)
    // end of a step: the previous stage has landed (and the codes fetched before this step's DMA); publish.  Wave 0 has one
    // more DMA per stage in flight (the positions) than the others.
#define ICN_W7_RETIRE_AND_PUBLISH() do { \
        if (NST == 3 && issued && p_exact) { \
            constexpr int FULLW = U % 16 == 0 ? 4 : (U % 16) / 4;      /* waves that issue all NI union-row DMAs */ \
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI + 2) : "memory"); \
            else if (wave < FULLW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI + 1) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory"); \
        } else { \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier(); \
    } while (0)
#define ICN_W7_DRAIN_AND_PUBLISH() do { \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier(); \
    } while (0)

    // ring fill: stages 0 .. NST - 2; codes of stage NST - 1 in flight
    ICN_W7_FETCH_CODES();                                 // stage 0
    ICN_W7_DRAIN_AND_PUBLISH();
#pragma unroll
    for (int f = 0; f < NST - 1; ++f) {
        ICN_W7_MAKE_OFFSETS();
        ICN_W7_ADVANCE_CODE_PTR();
        ICN_W7_FETCH_CODES();                             // stage f + 1
        ICN_W7_ISSUE();                                   // stage f
        ICN_W7_DRAIN_AND_PUBLISH();
    }
    int c_ring = 0;
    for (int step = 0; step < nsteps; ++step) {
        ICN_W7_MAKE_OFFSETS();                            // stage step + NST - 1
        ICN_W7_ADVANCE_CODE_PTR();
        ICN_W7_FETCH_CODES();                             // stage step + NST (before the data DMA: retired with the previous stage)
        ICN_W7_ISSUE();
        const float* slot = ring + c_ring * STAGE_F;
        const char* xa = reinterpret_cast<const char*>(slot) + (wr * 32 + l31) * 4;
        const float* yb = slot + U * W7_ROWF + wc * 32 + l31;
        const u32x4* pp = reinterpret_cast<const u32x4*>(slot + U * W7_ROWF + W7_PX * 64);
        if constexpr (ARITH == 1) {
            // this lane's 8 pixels: k = 8 h + j
            u32x4 pk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[j] = pp[8 * h + j];
            f32x4 yl, yh;
#pragma unroll
            for (int j = 0; j < 4; ++j) { yl[j] = yb[(8 * h + j) * 64]; yh[j] = yb[(8 * h + 4 + j) * 64]; }
            f32x4 xl[2], xh[2];
            auto fetch = [&](int buf, int t) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned o0 = (t & 1) ? pk[j][t >> 1] >> 16 : pk[j][t >> 1] & 0xFFFFu;
                    const unsigned o1 = (t & 1) ? pk[4 + j][t >> 1] >> 16 : pk[4 + j][t >> 1] & 0xFFFFu;
                    xl[buf][j] = *reinterpret_cast<const float*>(xa + o0);
                    xh[buf][j] = *reinterpret_cast<const float*>(xa + o1);
                }
            };
            fetch(0, 0);
            const Pieces3 pb = split_bf16x3(yl, yh);
#define ICN_MF16W(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), c_, 0, 0, 0)
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                if (t + 1 < 7) fetch((t + 1) & 1, t + 1);
                const Pieces3 pa = split_bf16x3(xl[t & 1], xh[t & 1]);
                f32x16 c = acc[t];
                c = ICN_MF16W(pa.p1, pb.p3, c);
                c = ICN_MF16W(pa.p3, pb.p1, c);
                c = ICN_MF16W(pa.p2, pb.p2, c);
                c = ICN_MF16W(pa.p1, pb.p2, c);
                c = ICN_MF16W(pa.p2, pb.p1, c);
                c = ICN_MF16W(pa.p1, pb.p1, c);
                acc[t] = c;
            }
#undef ICN_MF16W
        } else {
        // positions of this lane's 8 pixels (k = 2 * k2 + h), then the operands of pixel pair k2 + 1 are fetched ahead of the
        // seven MFMAs of pair k2 (register double buffer, as the conv kernel's fragments)
        u32x4 pk[W7_PX / 2];
#pragma unroll
        for (int k2 = 0; k2 < W7_PX / 2; ++k2) pk[k2] = pp[2 * k2 + h];
        float fa[2][7], fb[2];
#define ICN_W7_OPERANDS(BUF, K2) do { \
            fb[BUF] = yb[(2 * (K2) + h) * 64]; \
            _Pragma("unroll") \
            for (int t = 0; t < 7; ++t) { \
                const unsigned off_ = (t & 1) ? pk[K2][t >> 1] >> 16 : pk[K2][t >> 1] & 0xFFFFu; \
                fa[BUF][t] = *reinterpret_cast<const float*>(xa + off_); \
            } \
        } while (0)
        ICN_W7_OPERANDS(0, 0);
#pragma unroll
        for (int k2 = 0; k2 < W7_PX / 2; ++k2) {
            if (k2 + 1 < W7_PX / 2) ICN_W7_OPERANDS((k2 + 1) & 1, k2 + 1);
            __builtin_amdgcn_sched_barrier(0);            // keep the next pair's LDS reads ahead of this pair's MFMAs
#pragma unroll
            for (int t = 0; t < 7; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k2 & 1][t], fb[k2 & 1], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef ICN_W7_OPERANDS
        }
        if (do_bias && tid < 64) {
#pragma unroll
            for (int k = 0; k < W7_PX; ++k) bsum += slot[U * W7_ROWF + k * 64 + tid];
        }
        ICN_W7_RETIRE_AND_PUBLISH();
        c_ring = c_ring == NST - 1 ? 0 : c_ring + 1;
    }
#undef ICN_W7_FETCH_CODES
#undef ICN_W7_MAKE_OFFSETS
#undef ICN_W7_ADVANCE_CODE_PTR
#undef ICN_W7_ISSUE
#undef ICN_W7_RETIRE_AND_PUBLISH
#undef ICN_W7_DRAIN_AND_PUBLISH

#pragma unroll
    for (int t = 0; t < 7; ++t) {
        float* out = partial + ((size_t)split * 7 + t) * Cin * Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int co = co0 + wc * 32 + l31;
            out[(size_t)ci * Cout + co] = acc[t][r];
        }
    }
    if (do_bias && tid < 64) bias_partial[(size_t)split * Cout + co0 + tid] = bsum;
    if (trace && tid == 0) {
        unsigned long long* o = trace + (size_t)blockIdx.x * 8;
        o[0] = tr_t0; o[1] = tr_t0; o[2] = tr_t0; o[3] = 0; o[4] = __builtin_amdgcn_s_memrealtime(); o[5] = 0;
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);
    }
#endif
}

// dw[co][ci][t] = sum_s partial[s][t][ci][co];   dbias[co] = sum_s bias_partial[s][co]
// A block owns 64 consecutive elements of the [t][ci][co] slab layout (plus, past the weights, of the bias row); its
// four waves each sum a quarter of the S slabs (256-byte coalesced reads, independent loads), the quarters are
// combined in a fixed order through LDS (deterministic), and wave 0 stores into the (Cout, Cin, 7) parameter layout.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw,
                                                       const float* __restrict__ bias_partial, float* __restrict__ dbias,
                                                       float* __restrict__ dw2, float* __restrict__ dbias2, int S, int Cin, int Cout,
                                                       int Cout0) {
    __shared__ float red[4][64];
    const int total = 7 * Cin * Cout;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                  // element of [weights | bias]
    const bool is_w = i < total, is_b = !is_w && bias_partial != nullptr && i - total < Cout;
    const float* src = is_w ? partial + i : bias_partial + (i - total);
    const size_t stride = is_w ? (size_t)total : (size_t)Cout;
    const int k0 = (S * q) / 4, k1 = (S * (q + 1)) / 4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (is_w || is_b) {
        // 8 slabs per round, all loads issued before the first add (a quarter of S is <= 8 for the model's launches: one
        // round of memory latency per block instead of two)
        for (int k = k0; k < k1; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k + u < k1 ? src[(size_t)(k + u) * stride] : 0.f;
            s0 += v[0] + v[4];
            s1 += v[1] + v[5];
            s2 += v[2] + v[6];
            s3 += v[3] + v[7];
        }
    }
    red[q][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0) {
        const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (is_w) {
            const int co = i % Cout, ci = (i / Cout) % Cin, t = i / (Cin * Cout);
            if (co < Cout0) dw[((size_t)co * Cin + ci) * 7 + t] = v;
            else dw2[((size_t)(co - Cout0) * Cin + ci) * 7 + t] = v;
        } else if (is_b) {
            const int co = i - total;
            if (co < Cout0) { if (dbias) dbias[co] = v; }
            else if (dbias2) dbias2[co - Cout0] = v;
        }
    }
}

// partial[chunk][t][ci][co] (+ bias_partial[chunk][co]) over row chunks, any Cin / Cout
__global__ void k_wgrad_generic(const float* __restrict__ x, const float* __restrict__ dy, const int32_t* __restrict__ idx,
                                float* __restrict__ partial, float* __restrict__ bias_partial, int M, int Ps, int Pd, int Cin,
                                int Cout, int ns, int rows_per_split) {
    const int total = 7 * Cin * Cout;
    const int m0 = blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Cout, ci = (i / Cout) % Cin, t = i / (Cin * Cout);
        const bool do_bias = bias_partial != nullptr && t == 0 && ci == 0;
        float s = 0.f, sb = 0.f;
        for (int m = m0; m < m1; ++m) {
            const int b = m / Pd, p = m % Pd;
            const float dyv = dy[(size_t)m * Cout + co];
            if (do_bias) sb += dyv;
            const int32_t code = idx[(size_t)t * Pd + p];
            if (code == -1) continue;
            s += gather1(x, code, b, Ps, ns, Cin, ci) * dyv;
        }
        partial[(size_t)blockIdx.y * total + i] = s;
        if (do_bias) bias_partial[(size_t)blockIdx.y * Cout + co] = sb;
    }
}

// ---- stem (Cin <= 4, e.g. the 3->64 first layer): HBM-bound, VALU ---------------------------------------
// Both stem kernels stage the gathered 7*CIN input values of a group of 64 pixels in LDS first (all 256 threads
// issue independent gathers), then stream the wide side (y or dy, Cout floats per pixel) with coalesced accesses.
constexpr int STEM_PIX = 64;
constexpr int STEM_WG_PIX = 256;                       // pixels whose inputs k_stem_wgrad gathers per phase

template <int CIN, int NPIX = STEM_PIX>
__device__ __forceinline__ void stem_gather(const float* __restrict__ x, const int32_t* __restrict__ idx, float* xs, int m0,
                                            int M, int Ps, int Pd, int ns) {
    for (int i = threadIdx.x; i < NPIX * 7; i += 256) {
        const int row = i / 7, t = i % 7, m = m0 + row;
        float v[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) v[ci] = 0.f;
        if (m < M) {
            const int b = m / Pd, p = m % Pd;
            const int32_t code = idx[(size_t)t * Pd + p];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) v[ci] = gather1(x, code, b, Ps, ns, CIN, ci);
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) xs[row * (7 * CIN) + t * CIN + ci] = v[ci];
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void k_stem_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                   const int32_t* __restrict__ idx, int M, int Ps, int Pd, int Cout, int ns) {
    constexpr int KT = 7 * CIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* w_s = reinterpret_cast<float*>(smem);       // [KT][Cout]
    float* b_s = w_s + KT * Cout;                      // [Cout]
    float* xs = b_s + Cout;                            // [STEM_PIX][KT]
    for (int i = threadIdx.x; i < KT * Cout; i += 256) {
        const int co = i % Cout, k = i / Cout, ci = k % CIN, t = k / CIN;
        w_s[i] = w[((size_t)co * CIN + ci) * 7 + t];
    }
    for (int i = threadIdx.x; i < Cout; i += 256) b_s[i] = bias ? bias[i] : 0.f;
    const int q = Cout / 4;                            // float4 column groups per pixel
    for (int m0 = blockIdx.x * STEM_PIX; m0 < M; m0 += gridDim.x * STEM_PIX) {
        __syncthreads();                               // previous xs consumed (and w_s ready on the first pass)
        stem_gather<CIN>(x, idx, xs, m0, M, Ps, Pd, ns);
        __syncthreads();
        for (int i = threadIdx.x; i < STEM_PIX * q; i += 256) {
            const int row = i / q, c4 = (i % q) * 4, m = m0 + row;
            if (m >= M) continue;
            f32x4 acc = *reinterpret_cast<const f32x4*>(b_s + c4);
#pragma unroll
            for (int k = 0; k < KT; ++k) acc += xs[row * KT + k] * *reinterpret_cast<const f32x4*>(w_s + k * Cout + c4);
            *reinterpret_cast<f32x4*>(y + (size_t)m * Cout + c4) = acc;
        }
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void k_stem_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const int32_t* __restrict__ idx, float* __restrict__ partial,
                                                     float* __restrict__ bias_partial, int M, int Ps, int Pd, int Cout, int ns,
                                                     int rows_per_block) {
    // A thread owns 4 output channels (one 16-byte load of dy per row: a wave streams 1 KiB per instruction instead of 256 B)
    // and every (256 / (Cout / 4))-th row of a 64-pixel group; the gathered inputs come from LDS as broadcasts.
    // Round 5: the inputs of STEM_WG_PIX = 256 pixels (a block's whole row range with the default split) are gathered in ONE phase --
    // 7 independent (code, value) load chains per thread instead of four phases of 2 with a barrier pair each: the gather's latency
    // (~2 us per phase) was most of a block's time -- and the cross-wave reduction buffer aliases the gather buffer.
    constexpr int KT = 7 * CIN, NA = KT + 1;           // 7*CIN weight sums + 1 bias sum
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);        // [STEM_WG_PIX][KT]
    f32x4* red = reinterpret_cast<f32x4*>(smem);       // [G][NA][Cout / 4]  (after the last read of xs)
    const int q = Cout / 4, G = 256 / q, g = threadIdx.x / q, c4 = threadIdx.x % q;
    const int mb = blockIdx.x * rows_per_block, me = min(M, mb + rows_per_block);
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int mg = mb; mg < me; mg += STEM_WG_PIX) {
        __syncthreads();
        stem_gather<CIN, STEM_WG_PIX>(x, idx, xs, mg, me, Ps, Pd, ns);
        __syncthreads();
        for (int m0 = mg; m0 < min(me, mg + STEM_WG_PIX); m0 += STEM_PIX) {
            const float* xg = xs + (m0 - mg) * KT;
            const int nrow = min(STEM_PIX, me - m0);
            // up to 4 rows per thread and group of 64 pixels (G >= 16): all loads issued before the first add, ascending row order
            f32x4 dyv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = g + u * G;
                dyv[u] = (row < nrow && u * G < STEM_PIX) ? ld4(dy + (size_t)(m0 + row) * Cout + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = g + u * G;
                if (row >= nrow || u * G >= STEM_PIX) break;
                acc[KT] += dyv[u];
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] += dyv[u] * xg[row * KT + k];
            }
            for (int row = g + 4 * G; row < nrow; row += G) {       // (G < 16: Cout > 64)
                const f32x4 d = ld4(dy + (size_t)(m0 + row) * Cout + 4 * c4);
                acc[KT] += d;
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] += d * xg[row * KT + k];
            }
        }
    }
    __syncthreads();                                   // every thread is done with xs: `red` may overwrite it
    // the row groups of a wave (64 / q of them, q a power of two) are combined with shuffles, the (at most) 4 waves through LDS:
    // fixed orders, deterministic
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wq = q < 64 ? q : 64;
#pragma unroll
    for (int i = 0; i < NA; ++i)
        for (int off = wq; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] += __shfl_xor(acc[i][e], off, 64);
    const int nwv = q <= 64 ? 4 : 256 / q;              // partial results per (i, c4): one per wave (q <= 64), else per group
    if (lane < wq) {
#pragma unroll
        for (int i = 0; i < NA; ++i) red[((q <= 64 ? wave : g) * NA + i) * q + c4] = acc[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NA * q; i += 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nwv; ++k) s += red[k * NA * q + i];
        const int a = i / q, c = (i % q) * 4;
        if (a < KT) *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * KT + a) * Cout + c) = s;   // [S][t][ci][co]
        else if (bias_partial) *reinterpret_cast<f32x4*>(bias_partial + (size_t)blockIdx.x * Cout + c) = s;
    }
}

bool stem_supported(int Cin, int Cout) { return Cin >= 1 && Cin <= 4 && Cout >= 4 && Cout <= 256 && 256 % Cout == 0; }

void launch_stem_fwd(const float* x, const float* w, const float* bias, float* y, const int32_t* idx, int M, int Ps, int Pd,
                     int Cin, int Cout, int ns, hipStream_t s) {
    const int blocks = std::min(4096, (M + STEM_PIX - 1) / STEM_PIX);
    const size_t lds = ((size_t)(7 * Cin + 1) * Cout + (size_t)STEM_PIX * 7 * Cin) * 4;
#define ICN_SF(C) hipLaunchKernelGGL((k_stem_fwd<C>), dim3(blocks), dim3(256), lds, s, x, w, bias, y, idx, M, Ps, Pd, Cout, ns)
    switch (Cin) { case 1: ICN_SF(1); break; case 2: ICN_SF(2); break; case 3: ICN_SF(3); break; default: ICN_SF(4); }
#undef ICN_SF
}

bool wgrad_supported(int Cin, int Cout) { return Cin % 64 == 0 && Cout % 64 == 0; }

// LDS bytes / co-resident blocks per CU of the MFMA wgrad kernels
static size_t wgrad_lds(int bi, int bj, bool dma) {
    return dma ? (size_t)3 * WG_RS * (bi + bj) * 4 + (size_t)2 * 4 * 2 * 64 * 4 /* code ring, NA <= 2 */ : (size_t)2 * 32 * (bi + bj) * 4;
}
static int wgrad_occ(int bi, int bj) { return std::min(4, (int)((160 * 1024) / wgrad_lds(bi, bj, true))); }

// co tile of the MFMA wgrad kernels; a pair's tiles must not straddle its two output tensors
static int wgrad_bj(int Cout, int Cout0) {
    const int c1 = Cout - Cout0;
    return (Cout0 % 128 == 0 && c1 % 128 == 0) ? 128 : 64;
}

bool wgrad_pair_supported(int M, int Ps, int Pd, int Cin, int Cout0, int Cout1) {
    if (dbg_flags() & 32) return false;
    if (Cin % 64 != 0 || Cout0 % 64 != 0 || Cout1 % 64 != 0) return false;
    const size_t x_bytes = (size_t)(M / Pd) * Ps * Cin * 4, dy_bytes = (size_t)M * std::max(Cout0, Cout1) * 4;
    return x_bytes < ((size_t)1 << 31) && dy_bytes < ((size_t)1 << 31);
}

// number of row splits (= partial slabs) each wgrad flavour uses; shared by the workspace query and the launch
static int wgrad_base_splits(int M, int Cin, int Cout, int Cout0) {     // MFMA kernels
    // One block per (ci tile, co tile, tap, row split).  Measured on I5 / batch 36 (tools/bench_layers.py with
    // ICN_WG_MULT = 0.75 ... 6): about three rounds of blocks over the chip's co-resident block slots is best -- one
    // exactly filled round of long blocks is 15 % slower (blocks drift apart and the tail idles), many more rounds
    // only add partial-slab traffic.
    const int bi = (Cin % 128 == 0) ? 128 : 64, bj = wgrad_bj(Cout, Cout0);
    const long tiles = 7L * (Cin / bi) * (Cout / bj);
    static const double mult = getenv("ICN_WG_MULT") ? atof(getenv("ICN_WG_MULT")) : 3.0;   // developer override
    long s = (long)(mult * 256L * wgrad_occ(bi, bj) / tiles);
    const long max_s = (M + 255) / 256;                    // >= 16 stages of 16 rows per block
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return (int)s;
}
static int wgrad_splits_pertap(int M, int Cin, int Cout, int Cout0) {
    if (Cout0 <= 0 || Cout0 > Cout) Cout0 = Cout;
    if (wgrad_supported(Cin, Cout)) return wgrad_base_splits(M, Cin, Cout, Cout0);
    if (stem_supported(Cin, Cout)) {                    // (ICN_STEM_ROWS: developer A/B of the rows per block; 512: two gather phases per block, half the slabs of 256)
        static const int rows = getenv("ICN_STEM_ROWS") ? std::max(64, atoi(getenv("ICN_STEM_ROWS"))) : 512;
        return std::min(2048, (M + rows - 1) / rows);
    }
    return std::min(512, (M + 127) / 128);
}
int wgrad7_splits(int M, int Pd, int Cin, int Cout, int Cout0);
// slabs of the workspace: enough for whichever kernel the launch picks
int wgrad_splits(int M, int Cin, int Cout, int Cout0) {
    if (Cout0 <= 0 || Cout0 > Cout) Cout0 = Cout;
    return std::max(wgrad_splits_pertap(M, Cin, Cout, Cout0), wgrad7_splits(M, 16, Cin, Cout, Cout0));
}

// ---- k_wgrad7 launch plan: one round of 256 * 2 workgroups, a (ci, co) tile of 64 x 64 and a contiguous range of patches each
struct W7Plan { int tiles, pps, splits; };
static bool wgrad7_plan(int M, int Pd, int Cin, int Cout, int Cout0, W7Plan& p) {
    if (Cin % 64 || Cout % 64 || Cout0 % 64 || Pd % W7_PX || M % W7_PX) return false;
    const long tiles = (long)(Cin / 64) * (Cout / 64), np = M / W7_PX;
    // rounds of workgroups over the chip's 512 slots (developer sweep: ICN_W7_MULT)
    static const double mult = getenv("ICN_W7_MULT") ? atof(getenv("ICN_W7_MULT")) : 1.0;
    long S = std::max(1L, (long)(mult * 512 / tiles));
    long pps = (np + S - 1) / S;
    if (pps < 8) pps = std::min<long>(8, np);              // short splits: the ring's fill and the slab would dominate
    p.tiles = (int)tiles;
    p.pps = (int)pps;
    p.splits = (int)((np + pps - 1) / pps);
    return tiles <= 4096 && pps >= 4;
}
// Which launches take k_wgrad7: class 1 = stride-1 tables (U = 64), class 2 = stride-2 tables (U = 112).  Default: class 1 only.
// Measured at I5 / batch 36 (gpurun_out/r4_f_*, one box, alternating): every launch alone is ~10 % faster on k_wgrad7 (whole calls
// 238 -> 208 us stride 1, 255 -> 231 us stride 2), and on ONE stream the step gains 0.145 ms; but with the weight gradients on the
// second stream beside the backward chain the stride-2 form (67 KB of LDS, 28 union-row DMAs per stage: the taps of
// neighbouring outputs share little) costs the chain more than it saves -- 8.44 ms per step with both classes, 8.37 with class 1
// only, 8.37 with the per-tap kernel everywhere.  Debug flag 4096 (ICN_DEBUG / icn_set_debug_flags) sends class 2 to k_wgrad7 as
// well (tests, A/B); flag 2048 sends everything to the per-tap kernel.
static bool w7_class_enabled(int cls) { return cls == 1 || (dbg_flags() & 4096) != 0; }
static size_t wgrad7_lds(int U, int nst) { return (size_t)nst * (U * W7_ROWF + W7_PX * 64 + 64) * 4 + 2 * (U > 64 ? 128 : 64) * 4; }
int wgrad7_splits(int M, int Pd, int Cin, int Cout, int Cout0) {
    W7Plan p;
    return wgrad7_plan(M, Pd, Cin, Cout, Cout0, p) ? p.splits : 0;
}

void launch_wgrad(const WgradArgs& a, hipStream_t s) {
    const bool pair = a.dy2 != nullptr || (a.y_taps && a.Cout0 > 0 && a.Cout0 < a.Cout);   // two weight tensors to fill
    const int Cout0 = pair ? a.Cout0 : a.Cout;
    int S = wgrad_splits_pertap(a.M, a.Cin, a.Cout, Cout0);
    W7Plan w7{};
    const size_t x_bytes7 = (size_t)(a.M / a.Pd) * a.Ps * a.Cin * 4, dy_bytes7 = (size_t)a.M * std::max(Cout0, a.Cout - Cout0) * 4;
    const bool use7 = a.w7_rows != nullptr && a.w7_pos != nullptr && (a.w7_U == 56 || a.w7_U == 64 || a.w7_U == 112) && !a.y_taps && !(dbg_flags() & (32 | 2048)) &&
                      (a.n_slots == 0 || a.side != nullptr) && x_bytes7 < ((size_t)1 << 31) && dy_bytes7 < ((size_t)1 << 31) &&
                      (size_t)(a.M / a.Pd) * a.n_slots * a.Cin * 4 < ((size_t)1 << 30) &&
                      wgrad7_plan(a.M, a.Pd, a.Cin, a.Cout, Cout0, w7) && w7_class_enabled(a.w7_U > 64 ? 2 : 1);
    if (use7) {
        S = w7.splits;
        // ring depth (developer A/B: ICN_W7_NST=2|3): 3 stages / 2 workgroups per CU, or 2 stages (a third less LDS: 3 per CU)
        static const int nst_env = getenv("ICN_W7_NST") ? atoi(getenv("ICN_W7_NST")) : 0;
        const int nst = a.w7_U > 64 ? 2 : (nst_env == 2 || nst_env == 3 ? nst_env : 3);
        const int occ = std::min(3, (int)((160 * 1024) / wgrad7_lds(a.w7_U, nst)));
        (void)occ;
        const dim3 grid((unsigned)w7.tiles * (unsigned)((S + 7) / 8 * 8));
        const unsigned side_bytes7 = (unsigned)((size_t)(a.M / a.Pd) * a.n_slots * a.Cin * 4);
        const bool b3_ = arith_mode() == 1 && !(dbg_flags() & 131072);   // debug flag 131072: weight gradients stay exact under bf16x3
        prof_mark_begin(b3_ ? PROF_WG7_B3 : PROF_WG7, a.algo_flops, s);
#define ICN_W7(U_, NST_)                                                                                                       \
    do {                                                                                                                       \
        static std::atomic<uint64_t> attr_devices{0};                                                                          \
        if (!((attr_devices.load(std::memory_order_relaxed) >> current_device_bit()) & 1)) {                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad7<U_, NST_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad7<U_, NST_, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                             \
            attr_devices.fetch_or((uint64_t)1 << current_device_bit(), std::memory_order_relaxed);                             \
        }                                                                                                                      \
        if (b3_)                                                                                                               \
            hipLaunchKernelGGL((k_wgrad7<U_, NST_, 1>), grid, dim3(256), wgrad7_lds(U_, NST_), s, a.x, a.dy, a.dy2, a.w7_rows, a.w7_pos, \
                               a.n_slots > 0 ? a.side : nullptr, a.partial, a.bias_partial, a.M, a.Ps, a.Pd, a.Cin, a.Cout, Cout0,   \
                               a.n_slots, w7.pps, S, (unsigned)x_bytes7, side_bytes7, g_trace_cap >= (size_t)grid.x * 8 ? g_trace : nullptr); \
        else                                                                                                                   \
        hipLaunchKernelGGL((k_wgrad7<U_, NST_>), grid, dim3(256), wgrad7_lds(U_, NST_), s, a.x, a.dy, a.dy2, a.w7_rows, a.w7_pos, \
                           a.n_slots > 0 ? a.side : nullptr, a.partial, a.bias_partial, a.M, a.Ps, a.Pd, a.Cin, a.Cout, Cout0,   \
                           a.n_slots, w7.pps, S, (unsigned)x_bytes7, side_bytes7, g_trace_cap >= (size_t)grid.x * 8 ? g_trace : nullptr); \
    } while (0)
        if (a.w7_U == 112) ICN_W7(112, 2);
        else if (a.w7_U == 56 && nst == 2) ICN_W7(56, 2);
        else if (a.w7_U == 56) ICN_W7(56, 3);
        else if (nst == 2) ICN_W7(64, 2);
        else ICN_W7(64, 3);
#undef ICN_W7
        prof_mark_end(s);
    } else if (wgrad_supported(a.Cin, a.Cout)) {
        int rows = (a.M + S - 1) / S;
        rows = (rows + 31) / 32 * 32;
        const bool bi128 = a.Cin % 128 == 0, bj128 = wgrad_bj(a.Cout, Cout0) == 128;
        const int BI = bi128 ? 128 : 64, BJ = bj128 ? 128 : 64;
        dim3 grid((a.Cin / BI) * (a.Cout / BJ) * 7 * ((S + 7) / 8 * 8));   // whole XCD rounds; blocks of splits >= S exit
        const size_t lds = wgrad_lds(BI, BJ, false);
        const size_t x_bytes = (size_t)(a.M / a.Pd) * a.Ps * a.Cin * 4;
        const size_t dy_bytes = a.y_taps ? (size_t)a.M * a.y_taps * a.Cout * 4 : (size_t)a.M * std::max(Cout0, a.Cout - Cout0) * 4;
        const size_t side_bytes = (size_t)(a.M / a.Pd) * a.n_slots * a.Cin * 4;
        const bool dma = !(dbg_flags() & 32) && a.dcode != nullptr && (a.n_slots == 0 || a.side != nullptr) &&
                         x_bytes < ((size_t)1 << 31) && dy_bytes < ((size_t)1 << 31) && side_bytes < ((size_t)1 << 30);
        if ((pair || a.y_taps) && !dma) throw std::invalid_argument("icn: pair weight gradient outside the LDS-DMA kernel's limits");
        // developer A/B (ICN_WG_OCC=2): at most two workgroups per CU, by asking for more than a third of the LDS -- leaves the
        // chain's HBM-bound passes registers to run beside it on the second stream (three workgroups of 168 registers fill a SIMD)
        static const int occ_cap = getenv("ICN_WG_OCC") ? atoi(getenv("ICN_WG_OCC")) : 0;
        const size_t lds_dma = std::max(wgrad_lds(BI, BJ, true), occ_cap == 2 ? (size_t)(160 * 1024 / 3 + 1024) : (size_t)0);
        // the decoder heads' dense dW (identity rows, tap-major g, no pole means): the body with the gather compiled away
        static const bool dense_env = !(getenv("ICN_WG_DENSE") && atoi(getenv("ICN_WG_DENSE")) == 0);   // developer A/B
        const bool dense = dense_env && dma && a.y_taps == 7 && a.identity_rows && a.n_slots == 0 && a.Ps == a.Pd && a.dy2 == nullptr && !(dbg_flags() & 32768);   // identity_rows: k_wgrad_dense never reads dcode
#define ICN_WG(I, J)                                                                                                       \
    do {                                                                                                                   \
        if (dense && b3w)                                                                                                  \
            hipLaunchKernelGGL((k_wgrad_dense<I, J, 1>), grid, dim3(256), lds_dma, s, a.x, a.dy, a.partial, a.bias_partial, a.M,  \
                               a.Pd, a.Cin, a.Cout, a.Cout, rows, S, (unsigned)x_bytes,                                     \
                               g_trace_cap >= (size_t)grid.x * 8 ? g_trace : nullptr);                                      \
        else if (dma && b3w)                                                                                               \
            hipLaunchKernelGGL((k_wgrad_dma<I, J, 1>), grid, dim3(256), lds_dma, s, a.x, a.dy, a.dy2, a.dcode,               \
                               a.n_slots > 0 ? a.side : nullptr, a.partial, a.bias_partial, a.M, a.Ps, a.Pd, a.Cin, a.Cout,   \
                               a.y_taps ? a.Cout : Cout0, a.n_slots, rows, S, (unsigned)x_bytes, (unsigned)side_bytes, a.y_taps, \
                               g_trace_cap >= (size_t)grid.x * 8 ? g_trace : nullptr); \
        else if (dense)                                                                                                    \
            hipLaunchKernelGGL((k_wgrad_dense<I, J>), grid, dim3(256), lds_dma, s, a.x, a.dy, a.partial, a.bias_partial, a.M,  \
                               a.Pd, a.Cin, a.Cout, a.Cout, rows, S, (unsigned)x_bytes,                                     \
                               g_trace_cap >= (size_t)grid.x * 8 ? g_trace : nullptr);                                      \
        else if (dma)                                                                                                      \
            hipLaunchKernelGGL((k_wgrad_dma<I, J>), grid, dim3(256), lds_dma, s, a.x, a.dy, a.dy2, a.dcode,                  \
                               a.n_slots > 0 ? a.side : nullptr, a.partial, a.bias_partial, a.M, a.Ps, a.Pd, a.Cin, a.Cout,   \
                               a.y_taps ? a.Cout : Cout0, a.n_slots, rows, S, (unsigned)x_bytes, (unsigned)side_bytes, a.y_taps, \
                               g_trace_cap >= (size_t)grid.x * 8 ? g_trace : nullptr); \
        else                                                                                                               \
            hipLaunchKernelGGL((k_wgrad<I, J>), grid, dim3(256), lds, s, a.x, a.dy, a.idx, a.partial, a.bias_partial, a.M,   \
                               a.Ps, a.Pd, a.Cin, a.Cout, a.ns, rows, S);                                                  \
    } while (0)
        const bool b3w = dma && arith_mode() == 1 && !(dbg_flags() & 131072);   // (debug flag 131072: weight gradients stay exact)
        prof_mark_begin(b3w ? (dense ? PROF_WGDENSE_B3 : PROF_WGD_B3)
                        : dense ? (bi128 ? (bj128 ? PROF_WGDENSE_128x128 : PROF_WGDENSE_128x64) : (bj128 ? PROF_WGDENSE_64x128 : PROF_WGDENSE_64x64))
                              : (bi128 ? (bj128 ? PROF_WG_128x128 : PROF_WG_128x64) : (bj128 ? PROF_WG_64x128 : PROF_WG_64x64)) -
                                    (dma ? PROF_WG_128x128 - PROF_WGD_128x128 : 0),
                        a.algo_flops, s);
        if (bi128 && bj128) ICN_WG(128, 128);
        else if (bi128) ICN_WG(128, 64);
        else if (bj128) ICN_WG(64, 128);
        else ICN_WG(64, 64);
        prof_mark_end(s);
#undef ICN_WG
    } else if (stem_supported(a.Cin, a.Cout)) {
        const int rows = (a.M + S - 1) / S;
        const size_t lds = std::max((size_t)STEM_WG_PIX * 7 * a.Cin, (size_t)4 * (7 * a.Cin + 1) * a.Cout) * 4;   // xs, then one partial per wave in its place
#define ICN_SW(C)                                                                                                    \
    hipLaunchKernelGGL((k_stem_wgrad<C>), dim3(S), dim3(256), lds, s, a.x, a.dy, a.idx, a.partial, a.bias_partial, a.M, \
                       a.Ps, a.Pd, a.Cout, a.ns, rows)
        switch (a.Cin) { case 1: ICN_SW(1); break; case 2: ICN_SW(2); break; case 3: ICN_SW(3); break; default: ICN_SW(4); }
#undef ICN_SW
    } else {
        const int rows = (a.M + S - 1) / S;
        const int total = 7 * a.Cin * a.Cout;
        dim3 grid(std::min(64, (total + 255) / 256), S);
        hipLaunchKernelGGL(k_wgrad_generic, grid, dim3(256), 0, s, a.x, a.dy, a.idx, a.partial, a.bias_partial, a.M, a.Ps,
                           a.Pd, a.Cin, a.Cout, a.ns, rows);
    }
    const int elems = 7 * a.Cin * a.Cout + a.Cout;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((elems + 63) / 64), dim3(256), 0, s, a.partial, a.dw, a.bias_partial, a.dbias, a.dw2,
                       a.dbias2, S, a.Cin, a.Cout, Cout0);
}

// ---------------------------------------------------------------------------------------------------------
// ELL sparse row mix: out[b, r, :] = sum_e coef[r][e] * in[b, idx[r][e], :]     (HBM-bound)
// ---------------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void k_spmm_ell(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ idx,
                           const float* __restrict__ coef, int B, int Pin, int Pout, int C, int W) {
    const int cv = C / VEC;
    const size_t total = (size_t)B * Pout * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC;
        const size_t br = i / cv;
        const int r = (int)(br % Pout), b = (int)(br / Pout);
        float accv[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) accv[v] = 0.f;
        // Entries in groups of six: first all indices and coefficients, then all row loads, then the sums -- three
        // levels of independent loads instead of a chain of W dependent ones.  Absent entries (rows are left-packed,
        // -1 padded) add +0, in the same order as before, so results are unchanged.
        for (int e0 = 0; e0 < W; e0 += 6) {
            int32_t j[6];
            float w[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const bool in_row = e0 + k < W;
                j[k] = in_row ? idx[(size_t)r * W + e0 + k] : -1;
                w[k] = in_row ? coef[(size_t)r * W + e0 + k] : 0.f;
            }
            if (VEC == 4) {
                f32x4 x[6];
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    x[k] = j[k] >= 0 ? ld4(in + ((size_t)b * Pin + j[k]) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (j[k] >= 0) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) accv[v] += w[k] * x[k][v];
                    }
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (j[k] >= 0) accv[0] += w[k] * in[((size_t)b * Pin + j[k]) * C + c];
            }
        }
        float* o = out + br * C + c;
        if (VEC == 4) *reinterpret_cast<f32x4*>(o) = f32x4{accv[0], accv[1], accv[2], accv[3]};
        else o[0] = accv[0];
    }
}

// Grid-stride walk that keeps an XCD on ONE contiguous eighth of the items: blocks b and b + 8 share an XCD (and its L2), so
// block b walks the (b % 8)-th eighth with the other blocks of its XCD.  For the sparse row mixes below neighbouring output
// rows share source rows; dealt round-robin over the 8 L2s, every shared row is fetched from HBM by several of them.
struct XcdWalk {
    size_t pos, end, step;
    __device__ XcdWalk(size_t total) {
        const unsigned x = blockIdx.x % 8, nb = gridDim.x / 8;            // launches round the grid up to a multiple of 8
        const size_t per = (total + 7) / 8;
        pos = (size_t)x * per + (size_t)(blockIdx.x / 8) * blockDim.x + threadIdx.x;
        end = (x + 1) * per < total ? (x + 1) * per : total;
        step = (size_t)nb * blockDim.x;
    }
};
static unsigned xcd_grid(size_t total, unsigned cap) {
    static const int force = getenv("ICN_XCD_GRID") ? atoi(getenv("ICN_XCD_GRID")) : 0;      // developer sweep
    if (force > 0) cap = (unsigned)force;
    const size_t per = (total + 7) / 8, nb = (per + 255) / 256;
    return 8u * (unsigned)std::max<size_t>(1, std::min<size_t>(nb, cap / 8));
}

// Aggregate of the composite backward (icn_upconv_bwd):  g[b, row, c] (+)= sum_e coef[r][e] * dy[b, idx[r][e], c]  with the
// channel axis c over [dy0 | dy1] (C0 + C1 channels, a pair's two output gradients side by side), row = rows ? rows[r] : r.
// Main pass: every row of the width-8 table; second pass (acc = 1): the few rows with more than 8 entries (next to the
// poles), added on top.  HBM/L2-bound: a thread owns 4 channels of one row, its (up to 8) source rows are independent loads.
__global__ __launch_bounds__(256) void k_upconv_gather(const float* __restrict__ dy0, const float* __restrict__ dy1,
                                                        float* __restrict__ g, const int32_t* __restrict__ idx,
                                                        const float* __restrict__ coef, const int32_t* __restrict__ rows, int B,
                                                        int Pin, int nrows, int rows_total, int C0, int C1, int W, int acc) {
    const int C = C0 + C1, cv = C / 4;
    const size_t total = (size_t)B * nrows * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 4;
        const size_t br = i / cv;
        const int r = (int)(br % nrows), b = (int)(br / nrows);
        const bool second = c >= C0;
        const float* src = (second ? dy1 : dy0) + (size_t)b * Pin * (second ? C1 : C0) + (second ? c - C0 : c);
        const int stride = second ? C1 : C0;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int e0 = 0; e0 < W; e0 += 8) {
            int32_t j[8];
            float w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool in_row = e0 + k < W;
                j[k] = in_row ? idx[(size_t)r * W + e0 + k] : -1;
                w[k] = in_row ? coef[(size_t)r * W + e0 + k] : 0.f;
            }
            f32x4 x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = j[k] >= 0 ? ld4(src + (size_t)j[k] * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += w[k] * x[k];
        }
        float* o = g + ((size_t)b * rows_total + (rows ? rows[r] : r)) * C + c;
        if (acc) sum += ld4(o);
        *reinterpret_cast<f32x4*>(o) = sum;
    }
}

// Mirror image for the forward (icn_upconv_fwd, dense path):  [y0 | y1][b, rows ? rows[r] : r, c] (+)= bias[c] +
// sum_e coef[r][e] * z[b, idx[r][e], c]  with z (B, zrows, C0 + C1) the per-tap products W_t x[s] (row s * 7 + t).
__global__ __launch_bounds__(256) void k_upconv_scatter(const float* __restrict__ z, const float* __restrict__ bias,
                                                         float* __restrict__ y0, float* __restrict__ y1,
                                                         const int32_t* __restrict__ idx, const float* __restrict__ coef,
                                                         const int32_t* __restrict__ rows, int B, int zrows, int nrows, int Pout,
                                                         int C0, int C1, int W, int acc) {
    const int C = C0 + C1, cv = C / 4;
    for (XcdWalk w_((size_t)B * nrows * cv); w_.pos < w_.end; w_.pos += w_.step) {
        const size_t i = w_.pos;
        const int c = (int)(i % cv) * 4;
        const size_t br = i / cv;
        const int r = (int)(br % nrows), b = (int)(br / nrows);
        const float* src = z + (size_t)b * zrows * C + c;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        if (bias && !acc) sum = ld4(bias + c);
        for (int e0 = 0; e0 < W; e0 += 8) {
            int32_t j[8];
            float w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool in_row = e0 + k < W;
                j[k] = in_row ? idx[(size_t)r * W + e0 + k] : -1;
                w[k] = in_row ? coef[(size_t)r * W + e0 + k] : 0.f;
            }
            f32x4 x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = j[k] >= 0 ? ld4(src + (size_t)j[k] * C) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += w[k] * x[k];
        }
        const int row = rows ? rows[r] : r;
        float* o = c < C0 ? y0 + ((size_t)b * Pout + row) * C0 + c : y1 + ((size_t)b * Pout + row) * C1 + (c - C0);
        if (acc) sum += ld4(o);
        *reinterpret_cast<f32x4*>(o) = sum;
    }
}

void launch_upconv_scatter(const float* z, const float* bias, float* y0, float* y1, const int32_t* idx, const float* coef,
                           const int32_t* rows, int B, int zrows, int nrows, int Pout, int C0, int C1, int W, int acc, hipStream_t s) {
    if (nrows <= 0) return;
    const size_t total = (size_t)B * nrows * ((C0 + C1) / 4);
    hipLaunchKernelGGL(k_upconv_scatter, dim3(xcd_grid(total, 16384)), dim3(256), 0, s, z, bias, y0, y1,
                       idx, coef, rows, B, zrows, nrows, Pout, C0, C1, W, acc);
}

// ---------------------------------------------------------------------------------------------------------------------------
// LDS-staged forms of the two sparse passes of the decoder-block head (round 3).  Both are stencils over the pixel grid: an
// output pixel mixes 12-20 source rows, and neighbouring outputs share most of them.  The row-per-thread kernels above fetch
// every source row of every output through the vector-memory path (12.5 resp. 20 16-byte loads per output and 4 channels) and
// are bound by it (TD busy 85 %, 2.7-3.9 TB/s of useful bytes).  Here a workgroup owns a PATCH of the output grid (8 x 16 fine
// pixels, resp. 4 x 8 coarse ones), brings the union of the patch's source rows -- 2.5 resp. 7.5 per output instead of 12.5 /
// 20 -- into LDS by LDS-DMA, 32 channels (one 128-byte line per row) at a time, and every thread then mixes its output from
// LDS.  The patch's row list and each output's positions in it come from a host-built table (icn_api.cpp: build_patches), so
// chart seams, five-valent vertices and poles need no special code; outputs with more entries than the table's width keep their
// overflow pass.  Three workgroups per CU cover each other's load phases.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int PATCH_CH = 32;                              // channels per pass: one 128-byte line per source row
constexpr int PATCH_IT_MAX = 14;                          // row-list capacity: 14 x 32 = 448 rows (56 KB of LDS)

// DMA the patch's rows (sample b, channels [c0, c0 + 32) of `src` with row stride Cs floats) into LDS rows_s[u][32]
__device__ __forceinline__ void patch_stage(const float* __restrict__ src, unsigned src_bytes, const int32_t* __restrict__ prow_l,
                                            int n_it, int b, int rows_per_sample, int Cs, int c0, float* rows_s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
#pragma unroll
    for (int it = 0; it < PATCH_IT_MAX; ++it) {           // (constant trip count: prow_l stays in registers)
        if (it < n_it) {
            const int row = prow_l[it];                   // this lane's source row of instruction `it` (-1: zeros)
            const unsigned off = row >= 0 ? ((unsigned)(b * rows_per_sample + row) * (unsigned)Cs + (unsigned)c0) * 4u + 16u * (lane & 7)
                                          : 0xC0000000u;  // out of range: the buffer range check writes zeros
            float* dst = rows_s + __builtin_amdgcn_readfirstlane((it * 32 + wave * 8) * PATCH_CH);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
    }
#endif
}

constexpr int PS_W = 16;                                  // entries per output pixel of the scatter table
constexpr int PS_PH = 8, PS_PW = 16;                      // patch of fine pixels
__global__ __launch_bounds__(256, 3) void k_upconv_scatter_lds(const float* __restrict__ z, const float* __restrict__ bias,
                                                             float* __restrict__ y0, float* __restrict__ y1,
                                                             const int32_t* __restrict__ prow, const uint16_t* __restrict__ plocal,
                                                             const float* __restrict__ coef, int B, int zrows, int Pout, int C0,
                                                             int C1, int npatch, int umax, int gridW, int bgroup, int cgroup,
                                                             unsigned z_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* rows_s = reinterpret_cast<float*>(smem);       // [umax][32]
    const int C = C0 + C1, nchunk = C / PATCH_CH, n_it = umax / 32;
    const int nbg = (B + bgroup - 1) / bgroup, ncg = (nchunk + cgroup - 1) / cgroup, items = npatch * nbg * ncg;
    // XCD-contiguous item order: (sample group, channel group) outermost, patches of a chart next to each other inside one
    // XCD's eighth
    const int per = (items + 7) / 8, item = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (blockIdx.x / 8 >= per || item >= items) return;
    const int patch = item % npatch, bg = (item / npatch) % nbg, cg = item / (npatch * nbg);
    const int ch_end = min(nchunk, (cg + 1) * cgroup);
    const int ppr = gridW / PS_PW;                        // patches per grid row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = tid >> 1, half = tid & 1;              // output pixel of the patch, 16-channel half of the pass
    const int orow = ((patch / ppr) * PS_PH + px / PS_PW) * gridW + (patch % ppr) * PS_PW + px % PS_PW;
    // this thread's entries: positions in the patch's row list + coefficients
    int li[PS_W];
    float cf[PS_W];
    {
        const uint16_t* lp = plocal + (size_t)orow * PS_W;
        const float* cp = coef + (size_t)orow * PS_W;
#pragma unroll
        for (int e = 0; e < PS_W; ++e) {                  // (absent entries: the zero row behind the list, coefficient 0 -- no branches below)
            const unsigned v = lp[e];
            li[e] = (v == 0xFFFFu ? umax : (int)v) * PATCH_CH + half * 16;
            cf[e] = v == 0xFFFFu ? 0.f : cp[e];
        }
    }
    if (tid < PATCH_CH) rows_s[umax * PATCH_CH + tid] = 0.f;   // the zero row (never written by the DMA)
    int32_t prow_l[PATCH_IT_MAX];                         // rows this lane stages: row (it * 32 + wave * 8 + lane / 8) of the list
#pragma unroll
    for (int it = 0; it < PATCH_IT_MAX; ++it)
        prow_l[it] = it < n_it ? prow[(size_t)patch * umax + it * 32 + wave * 8 + (lane >> 3)] : -1;
    const int b_end = min(B, (bg + 1) * bgroup);
    for (int b = bg * bgroup; b < b_end; ++b)
        for (int ch = cg * cgroup; ch < ch_end; ++ch) {
            const int c0 = ch * PATCH_CH;
            patch_stage(z, z_bytes, prow_l, n_it, b, zrows, C, c0, rows_s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int c = c0 + half * 16;
            f32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = bias ? ld4(bias + c + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < PS_W; ++e) {
                const float* r = rows_s + li[e];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += cf[e] * ld4(r + 4 * q);
            }
            float* o = c < C0 ? y0 + ((size_t)b * Pout + orow) * C0 + c : y1 + ((size_t)b * Pout + orow) * C1 + (c - C0);
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(o + 4 * q) = acc[q];
            __syncthreads();                              // every thread is done with rows_s before the next pass overwrites it
        }
#endif
}

// Passes (sample, 32-channel chunk) a workgroup of the patch kernels runs back to back on its patch: enough to amortise its
// table loads (>= 4), few enough that even the coarsest level makes >= ~4 rounds of workgroups over the chip's slots
static void patch_groups(int npatch, int B, int nchunk, int& bgroup, int& cgroup) {
    const long passes = (long)npatch * B * nchunk;
    long per = passes / (4L * 768);
    per = per < 4 ? 4 : (per > 16 ? 16 : per);
    cgroup = (int)std::min<long>(nchunk, per);
    bgroup = (int)std::max<long>(1, per / cgroup);
}

void launch_upconv_scatter_lds(const float* z, const float* bias, float* y0, float* y1, const PatchTab& t, const float* coef, int B,
                               int zrows, int Pout, int C0, int C1, hipStream_t s) {
    int bgroup, cgroup;
    patch_groups(t.npatch, B, (C0 + C1) / PATCH_CH, bgroup, cgroup);
    const int items = t.npatch * ((B + bgroup - 1) / bgroup) * (((C0 + C1) / PATCH_CH + cgroup - 1) / cgroup);
    const int grid = (items + 7) / 8 * 8;
    const size_t lds = (size_t)(t.umax + 1) * PATCH_CH * 4;      // + the zero row
    static std::atomic<uint64_t> attr_devices{0};
    if (!((attr_devices.load(std::memory_order_relaxed) >> current_device_bit()) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_upconv_scatter_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_devices.fetch_or((uint64_t)1 << current_device_bit(), std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(k_upconv_scatter_lds, dim3(grid), dim3(256), lds, s, z, bias, y0, y1, t.prow, t.plocal, coef, B, zrows, Pout, C0,
                       C1, t.npatch, t.umax, t.gridW, bgroup, cgroup, (unsigned)((size_t)B * zrows * (C0 + C1) * 4));
}
bool upconv_patch_usable(const PatchTab& t, int B, size_t rows_per_sample, int C0, int C1) {
    if (dbg_flags() & 1024) return false;                 // developer: the row-per-thread kernels
    if (t.prow == nullptr || t.umax < 32 || t.umax > 32 * PATCH_IT_MAX || t.umax % 32) return false;
    if (C0 % PATCH_CH || C1 % PATCH_CH) return false;
    return (size_t)B * rows_per_sample * std::max(C0 + C1, 1) * 4 < ((size_t)3 << 30);   // 32-bit buffer offsets below the zero code
}

// The aggregate of the composite backward on patches of 4 x 8 coarse pixels: a thread owns one coarse pixel and 4 channels for
// all 7 taps; its <= 20 fine source rows are positions in the patch's row list (plocal [Pc][20]), the coefficient block comes
// from the pixel's class (as in k_upconv_gather_px below).
constexpr int PG_PH = 4, PG_PW = 8;
__global__ __launch_bounds__(256, 3) void k_upconv_gather_lds(const float* __restrict__ dy0, const float* __restrict__ dy1,
                                                            float* __restrict__ g, const int32_t* __restrict__ prow,
                                                            const uint16_t* __restrict__ plocal, const int32_t* __restrict__ cls,
                                                            const float* __restrict__ cls_coef, int ncls, int B, int Pin, int Pc,
                                                            int C0, int C1, int npatch, int umax, int gridW, int bgroup, int cgroup) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* ctab = reinterpret_cast<f32x4*>(smem);         // [ncls][20][2]
    float* rows_s = reinterpret_cast<float*>(ctab + ncls * 20 * 2);   // [umax][32]  (table sized by the classes in use: 19 of 32 -> 41 KB
                                                                      //  per workgroup, which fits beside two k_wgrad7 workgroups: DESIGN 4.2b)
    for (int i = threadIdx.x; i < ncls * 20 * 2; i += 256) ctab[i] = ld4(cls_coef + 4 * i);
    const int C = C0 + C1, nchunk = C / PATCH_CH, n_it = umax / 32;
    const int nbg = (B + bgroup - 1) / bgroup, ncg = (nchunk + cgroup - 1) / cgroup, items = npatch * nbg * ncg;
    const int per = (items + 7) / 8, item = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (blockIdx.x / 8 >= per || item >= items) return;
    const int patch = item % npatch, bg = (item / npatch) % nbg, cg = item / (npatch * nbg);
    const int ch_end = min(nchunk, (cg + 1) * cgroup);
    const int ppr = gridW / PG_PW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = tid >> 3, q = tid & 7;                 // coarse pixel of the patch, 16-byte chunk of the pass
    const int sp = ((patch / ppr) * PG_PH + px / PG_PW) * gridW + (patch % ppr) * PG_PW + px % PG_PW;
    int li[20];
    {
        const uint16_t* lp = plocal + (size_t)sp * 20;
#pragma unroll
        for (int k = 0; k < 20; ++k) { const unsigned v = lp[k]; li[k] = (v == 0xFFFFu ? umax : (int)v) * PATCH_CH + 4 * q; }
    }
    if (tid < PATCH_CH) rows_s[umax * PATCH_CH + tid] = 0.f;   // the zero row absent sources point at
    const f32x4* cf = ctab + cls[sp] * 40;
    int32_t prow_l[PATCH_IT_MAX];
#pragma unroll
    for (int it = 0; it < PATCH_IT_MAX; ++it)
        prow_l[it] = it < n_it ? prow[(size_t)patch * umax + it * 32 + wave * 8 + (lane >> 3)] : -1;
    const unsigned b0_bytes = (unsigned)((size_t)B * Pin * C0 * 4), b1_bytes = (unsigned)((size_t)B * Pin * C1 * 4);
    const int b_end = min(B, (bg + 1) * bgroup);
    __syncthreads();                                      // the class table is in place
    for (int b = bg * bgroup; b < b_end; ++b)
        for (int ch = cg * cgroup; ch < ch_end; ++ch) {
            const int c0 = ch * PATCH_CH;
            if (c0 < C0) patch_stage(dy0, b0_bytes, prow_l, n_it, b, Pin, C0, c0, rows_s);
            else patch_stage(dy1, b1_bytes, prow_l, n_it, b, Pin, C1, c0 - C0, rows_s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            f32x4 acc[7];
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 20; ++k) {
                const f32x4 x = ld4(rows_s + li[k]);
                const f32x4 ca = cf[k * 2], cb = cf[k * 2 + 1];
                acc[0] += ca[0] * x; acc[1] += ca[1] * x; acc[2] += ca[2] * x; acc[3] += ca[3] * x;
                acc[4] += cb[0] * x; acc[5] += cb[1] * x; acc[6] += cb[2] * x;
            }
            float* o = g + ((size_t)b * Pc + sp) * 7 * C + c0 + 4 * q;
#pragma unroll
            for (int t = 0; t < 7; ++t) *reinterpret_cast<f32x4*>(o + (size_t)t * C) = acc[t];
            __syncthreads();
        }
#endif
}

void launch_upconv_gather_lds(const float* dy0, const float* dy1, float* g, const PatchTab& t, const int32_t* cls,
                              const float* cls_coef, int ncls, int B, int Pin, int Pc, int C0, int C1, hipStream_t s) {
    if (ncls < 1 || ncls > UPCONV_PX_CLASSES) throw std::invalid_argument("icn: coefficient classes of the per-pixel aggregate out of range");
    int bgroup, cgroup;
    patch_groups(t.npatch, B, (C0 + C1) / PATCH_CH, bgroup, cgroup);
    const int items = t.npatch * ((B + bgroup - 1) / bgroup) * (((C0 + C1) / PATCH_CH + cgroup - 1) / cgroup);
    const int grid = (items + 7) / 8 * 8;
    const size_t lds = (size_t)ncls * 20 * 2 * 16 + (size_t)(t.umax + 1) * PATCH_CH * 4;   // + the zero row
    static std::atomic<uint64_t> attr_devices{0};
    if (!((attr_devices.load(std::memory_order_relaxed) >> current_device_bit()) & 1)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_upconv_gather_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_devices.fetch_or((uint64_t)1 << current_device_bit(), std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(k_upconv_gather_lds, dim3(grid), dim3(256), lds, s, dy0, dy1, g, t.prow, t.plocal, cls, cls_coef, ncls, B, Pin, Pc,
                       C0, C1, t.npatch, t.umax, t.gridW, bgroup, cgroup);
}

// Same aggregate, one thread per (sample, coarse pixel, 4 channels) for ALL 7 taps: the 7 rows g_t[s] of a pixel draw on the
// same <= 20 fine rows (the two-ring of its site), each needed by 2-3 of the taps, so they are loaded once (20 loads instead
// of 49) and spread over the taps with a dense (20 x 7, mostly zero) coefficient block.  k_upconv_gather above is bound by
// the vector-memory pipe (2.6 GB of 16-byte loads for the r = 4 -> 5 block), not by HBM.  srcs [Pc][20] (-1 padded; pixels with
// more sources are left to the generic kernel and have all -1 here).  The coefficient blocks come in a handful of classes
// (19 at every level >= 2: the interior pattern covers 87 % of the pixels at r = 4): cls [Pc] names a pixel's class and the
// class table [ncls][20][8] (7 used) sits in LDS -- read per lane from global memory the blocks were two thirds of the
// kernel's vector-memory traffic (40 of 60 16-byte loads per thread, every lane of a pixel fetching the same values).
constexpr int UG_SRC = 20;
constexpr int UG_CLS_MAX = UPCONV_PX_CLASSES;
__global__ __launch_bounds__(256) void k_upconv_gather_px(const float* __restrict__ dy0, const float* __restrict__ dy1,
                                                           float* __restrict__ g, const int32_t* __restrict__ srcs,
                                                           const int32_t* __restrict__ cls, const float* __restrict__ cls_coef,
                                                           int ncls, int B, int Pin, int Pc, int C0, int C1) {
    __shared__ f32x4 ctab[UG_CLS_MAX * UG_SRC * 2];
    for (int i = threadIdx.x; i < ncls * UG_SRC * 2; i += 256) ctab[i] = ld4(cls_coef + 4 * i);
    __syncthreads();
    const int C = C0 + C1, cv = C / 4;
    for (XcdWalk w_((size_t)B * Pc * cv); w_.pos < w_.end; w_.pos += w_.step) {
        const size_t i = w_.pos;
        const int c = (int)(i % cv) * 4;
        const size_t bs = i / cv;
        const int sp = (int)(bs % Pc), b = (int)(bs / Pc);
        const bool second = c >= C0;
        const float* src = (second ? dy1 : dy0) + (size_t)b * Pin * (second ? C1 : C0) + (second ? c - C0 : c);
        const int stride = second ? C1 : C0;
        f32x4 acc[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int32_t* sj = srcs + (size_t)sp * UG_SRC;
        const f32x4* cf = ctab + cls[sp] * (UG_SRC * 2);
#pragma unroll 1                                       // one chunk of 5 sources in flight: 4-5 waves per SIMD
        for (int k0 = 0; k0 < UG_SRC; k0 += 5) {
            int32_t j[5];
            f32x4 x[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) j[k] = sj[k0 + k];
#pragma unroll
            for (int k = 0; k < 5; ++k) x[k] = j[k] >= 0 ? ld4(src + (size_t)j[k] * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const f32x4 ca = cf[(k0 + k) * 2], cb = cf[(k0 + k) * 2 + 1];
                acc[0] += ca[0] * x[k]; acc[1] += ca[1] * x[k]; acc[2] += ca[2] * x[k]; acc[3] += ca[3] * x[k];
                acc[4] += cb[0] * x[k]; acc[5] += cb[1] * x[k]; acc[6] += cb[2] * x[k];
            }
        }
        float* o = g + ((size_t)b * Pc + sp) * 7 * C + c;
#pragma unroll
        for (int t = 0; t < 7; ++t) *reinterpret_cast<f32x4*>(o + (size_t)t * C) = acc[t];
    }
}

void launch_upconv_gather_px(const float* dy0, const float* dy1, float* g, const int32_t* srcs, const int32_t* cls,
                             const float* cls_coef, int ncls, int B, int Pin, int Pc, int C0, int C1, hipStream_t s) {
    if (ncls < 1 || ncls > UG_CLS_MAX) throw std::invalid_argument("icn: coefficient classes of the per-pixel aggregate out of range");
    const size_t total = (size_t)B * Pc * ((C0 + C1) / 4);
    hipLaunchKernelGGL(k_upconv_gather_px, dim3(xcd_grid(total, 16384)), dim3(256), 0, s, dy0, dy1, g, srcs, cls, cls_coef, ncls,
                       B, Pin, Pc, C0, C1);
}

void launch_upconv_gather(const float* dy0, const float* dy1, float* g, const int32_t* idx, const float* coef, const int32_t* rows,
                          int B, int Pin, int nrows, int rows_total, int C0, int C1, int W, int acc, hipStream_t s) {
    if (nrows <= 0) return;
    const size_t total = (size_t)B * nrows * ((C0 + C1) / 4);
    hipLaunchKernelGGL(k_upconv_gather, dim3((unsigned)std::min((size_t)16384, (total + 255) / 256)), dim3(256), 0, s, dy0, dy1, g, idx,
                       coef, rows, B, Pin, nrows, rows_total, C0, C1, W, acc);
}

void launch_spmm_ell(const float* in, float* out, const int32_t* idx, const float* coef, int B, int Pin, int Pout, int C,
                     int W, hipStream_t s) {
    const int vec = (C % 4 == 0) ? 4 : 1;
    const size_t total = (size_t)B * Pout * (C / vec);
    const int blocks = (int)std::min((size_t)8192, (total + 255) / 256);
    if (vec == 4)
        hipLaunchKernelGGL(k_spmm_ell<4>, dim3(blocks), dim3(256), 0, s, in, out, idx, coef, B, Pin, Pout, C, W);
    else
        hipLaunchKernelGGL(k_spmm_ell<1>, dim3(blocks), dim3(256), 0, s, in, out, idx, coef, B, Pin, Pout, C, W);
}

// ---------------------------------------------------------------------------------------------------------
// scalar fall-backs (any Cin / Cout)
// ---------------------------------------------------------------------------------------------------------
// dst[b,p,n] = bias[n] + sum_t sum_e sum_k gather(src, idx[t][e][p])[k] * w(n,k,t)
// w is the parameter tensor [Cout][Cin][7]; transpose = 0: n = co, k = ci (forward); 1: n = ci, k = co (bwd-data)
__global__ void k_conv_generic(const float* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias,
                               float* __restrict__ dst, const int32_t* __restrict__ idx, int B, int Ps, int Pd, int K, int N,
                               int E, int ns, int transpose) {
    const size_t total = (size_t)B * Pd * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const size_t bp = i / N;
        const int p = (int)(bp % Pd), b = (int)(bp / Pd);
        float s = bias ? bias[n] : 0.f;
        for (int t = 0; t < 7; ++t)
            for (int e = 0; e < E; ++e) {
                const int32_t code = idx[((size_t)t * E + e) * Pd + p];
                if (code == -1) continue;
                for (int k = 0; k < K; ++k) {
                    const float wv = transpose ? w[((size_t)k * N + n) * 7 + t] : w[((size_t)n * K + k) * 7 + t];
                    s += gather1(src, code, b, Ps, ns, K, k) * wv;
                }
            }
        dst[i] = s;
    }
}

void launch_conv_generic(const float* src, const float* w, const float* bias, float* dst, const int32_t* idx, int B, int Ps,
                         int Pd, int K, int N, int E, int ns, int transpose, hipStream_t s) {
    const size_t total = (size_t)B * Pd * N;
    const int blocks = (int)std::min((size_t)16384, (total + 255) / 256);
    hipLaunchKernelGGL(k_conv_generic, dim3(blocks), dim3(256), 0, s, src, w, bias, dst, idx, B, Ps, Pd, K, N, E, ns,
                       transpose);
}

// ---------------------------------------------------------------------------------------------------------
// dst[b, q[v], :] += src[b, v, :]: folds the virtual-row GEMM of dgrad (extra transposed entries along chart seams)
// back into dx.  q is sorted, so the thread that owns the first virtual row of a pixel adds all of them in order.
// ---------------------------------------------------------------------------------------------------------
__global__ void k_row_scatter_add(const float* __restrict__ src, float* __restrict__ dst, const int32_t* __restrict__ q, int B,
                                  int nv, int nvp, int P, int C) {
    const int c4n = C / 4;
    const size_t total = (size_t)B * nv * c4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const size_t bv = i / c4n;
        const int v = (int)(bv % nv), b = (int)(bv / nv);
        const int32_t qv = q[v];
        if (v > 0 && q[v - 1] == qv) continue;            // not the first virtual row of this pixel
        float* d = dst + ((size_t)b * P + qv) * C + c;
        f32x4 acc = ld4(d);
        for (int k = v; k < nv && q[k] == qv; ++k) acc += ld4(src + ((size_t)b * nvp + k) * C + c);
        *reinterpret_cast<f32x4*>(d) = acc;
    }
}

void launch_row_scatter_add(const float* src, float* dst, const int32_t* q, int B, int nv, int nvp, int P, int C, hipStream_t s) {
    const size_t total = (size_t)B * nv * (C / 4);
    hipLaunchKernelGGL(k_row_scatter_add, dim3((unsigned)std::min((size_t)4096, (total + 255) / 256)), dim3(256), 0, s, src, dst, q, B,
                       nv, nvp, P, C);
}

// Prologue of a composite upsample + conv call (icn_upconv_*), one launch: blocks [0, npack) build the effective weights
//   packed[v][n][k] = sum_t alpha[v][t] * w(n, k, t)        (B operand [NV][N][K] of the gather-GEMM; n runs over the
// output channels of w, then of w2), and concatenate the biases; the other blocks fill the side buffer of the irregular
// rows,  side[b][s][:] = sum_e coef[s][e] * x[b, idx[s][e], :]  (the upsampled neighbour values those rows gather).
__global__ __launch_bounds__(256) void k_upconv_prologue(UpconvPrologueArgs a, int npack) {
    if ((int)blockIdx.x < npack) {
        const int Ct = a.Cout + a.Cout2, total = a.NV * Ct * a.Cin;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += npack * 256) {
            const int ci = i % a.Cin, co = (i / a.Cin) % Ct, v = i / (a.Cin * Ct);
            const float* wp = co < a.Cout ? a.w + ((size_t)co * a.Cin + ci) * 7 : a.w2 + ((size_t)(co - a.Cout) * a.Cin + ci) * 7;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 7; ++t) acc += a.alpha[v * 7 + t] * wp[t];
            a.packed[i] = acc;
        }
        if (a.bias_cat && blockIdx.x == 0)
            for (int c = threadIdx.x; c < Ct; c += 256) a.bias_cat[c] = c < a.Cout ? a.bias[c] : a.bias2[c - a.Cout];
        return;
    }
    const int j = blockIdx.x - npack;
    const int b = j / a.n_slots, sl = j % a.n_slots;
    for (int ch = 4 * threadIdx.x; ch < a.Cin; ch += 4 * 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < a.E; ++e) {
            const int32_t c = a.slot_idx[sl * a.E + e];
            if (c >= 0) v += a.slot_coef[sl * a.E + e] * ld4(a.src + ((size_t)b * a.Ps + c) * a.Cin + ch);
        }
        *reinterpret_cast<f32x4*>(a.side + ((size_t)b * a.n_slots + sl) * a.Cin + ch) = v;
    }
}

void launch_upconv_prologue(const UpconvPrologueArgs& a, hipStream_t s) {
    const int npack = std::min(4096, (a.NV * (a.Cout + a.Cout2) * a.Cin + 255) / 256);
    const int nside = (a.side && a.n_slots > 0) ? a.B * a.n_slots : 0;
    hipLaunchKernelGGL(k_upconv_prologue, dim3(npack + nside), dim3(256), 0, s, a, npack);
}

void launch_conv_prologue(const PrologueArgs& a, hipStream_t s) {
    if (a.packed_b3 && a.b3_bn != 64 && a.b3_bn != 128) throw std::invalid_argument("icn: bf16 weight image needs a column tile of 64 or 128");
    const int npack = (a.w && (a.packed || a.packed_b3)) ? std::min(2048, ((a.Cout + a.Cout2) * a.Cin * 7 + 255) / 256) : 0;
    const int nside = (a.side && a.n_slots > 0) ? a.B * a.n_slots * (a.src2 ? 2 : 1) : 0;
    if (npack + nside == 0 && !a.zero) return;
    hipLaunchKernelGGL(k_conv_prologue, dim3(std::max(1, npack + nside)), dim3(256), 0, s, a, npack);
}

}  // namespace icn
