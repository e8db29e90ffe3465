// Internal launch interface between the C ABI (icn_api.cpp) and the kernels (icn_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icn {

struct GatherGemmArgs {
    const float* src;       // (B, Ps, K)
    const float* wt;        // [7][N][K]
    const float* bias;      // [N] or null
    float* dst;             // (B, Pd, N)
    const int32_t* idx;     // [7][E][Pd] full table (register-staged fall-back kernel)
    const int32_t* dcode;   // DmaTable code [7][Pd] of the same table (LDS-DMA kernel), or null
    const float* side;      // (B, n_slots, K) side buffer filled by launch_conv_prologue, or null
    int n_slots;
    const int32_t* perm;    // [Pd] or null
    const uint32_t* mask32; // [Pd/32] tap mask per 32 rows (one word each: scalar loads), or null
    int M, Ps, Pd, K, N, E, ns;
    double algo_flops;      // algorithmic FLOPs of this launch (profiling only)
};

struct WgradArgs {
    const float* x;         // (B, Ps, Cin)
    const float* dy;        // (B, Pd, Cout)
    const int32_t* idx;     // forward table [7][Pd]
    const int32_t* dcode;   // its DmaTable code [7][Pd] (LDS-DMA kernel), or null
    const float* side;      // (B, n_slots, Cin) pole means of x filled by launch_conv_prologue, or null
    int n_slots;
    float* partial;         // [S][7][Cin][Cout]
    float* bias_partial;    // [S][Cout] (null when dbias is null)
    float* dw;              // [Cout][Cin][7]
    float* dbias;           // [Cout] or null
    int M, Ps, Pd, Cin, Cout, ns;
    double algo_flops;
};

bool gather_gemm_supported(int K, int N);
void launch_gather_gemm_auto(const GatherGemmArgs& a, hipStream_t s);

bool wgrad_supported(int Cin, int Cout);
int wgrad_splits(int M, int Cin, int Cout);
void launch_wgrad(const WgradArgs& a, hipStream_t s);
bool stem_supported(int Cin, int Cout);
void launch_stem_fwd(const float* x, const float* w, const float* bias, float* y, const int32_t* idx, int M, int Ps, int Pd,
                     int Cin, int Cout, int ns, hipStream_t s);

void launch_spmm_ell(const float* in, float* out, const int32_t* idx, const float* coef, int B, int Pin, int Pout, int C,
                     int W, hipStream_t s);

void launch_conv_generic(const float* src, const float* w, const float* bias, float* dst, const int32_t* idx, int B, int Ps,
                         int Pd, int K, int N, int E, int ns, int transpose, hipStream_t s);

// dst[b, q[v], :] += src[b, v, :], src (B, nvp, C), v < nv   (q sorted; rows of one q are summed in order by one thread
// => deterministic)
void launch_row_scatter_add(const float* src, float* dst, const int32_t* q, int B, int nv, int nvp, int P, int C, hipStream_t s);

// One launch ahead of a conv call: repack the weights (skipped when w is null) and fill the side buffer of a DmaTable
// (skipped when side is null or n_slots == 0) from `src` (B, Ps, K).
void launch_conv_prologue(const float* w, float* packed, int Cout, int Cin, int transpose, const float* src, const int32_t* slots,
                          float* side, int n_slots, int E, int B, int Ps, int K, int ns, hipStream_t s);

// ---- fused BatchNorm (+ residual) + ReLU (icn_bn.hip); stat = [mean | invstd] (2*C), sums = NS*C, ws = chunks*NS*C floats
bool bn_supported(int C);
int bn_chunks(int M);
void launch_bn_stats(const float* x, int M, int C, float eps, float momentum, float* running_mean, float* running_var, float* stat,
                     float* ws, hipStream_t s);
void launch_bn_relu_fwd(const float* a, const float* b, const float* stat_a, const float* stat_b, const float* ga, const float* ba,
                        const float* gb, const float* bb, float* y, int M, int C, hipStream_t s);
void launch_bn_relu_bwd(const float* dy, const float* y, const float* a, const float* b, const float* stat_a, const float* stat_b,
                        const float* ga, const float* gb, float* da, float* db, float* sums, float* ws, int M, int C, hipStream_t s);

// ---- fused 1x1 head + tanh (icn_bn.hip); ws = head_chunks(M) * 4 * (Cin + 4) floats
bool head_supported(int Cin, int Cout);
int head_chunks(int M);
void launch_head_fwd(const float* x, const float* w, const float* bias, float* y, int M, int Cin, int Cout, hipStream_t s);
void launch_head_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw, float* db, float* ws,
                     int M, int Cin, int Cout, hipStream_t s);

// ---- optional per-launch HIP-event timing of the MFMA kernels (bench.py's live roofline measurement) ----------
enum ProfKind { PROF_DMA_128x128 = 0, PROF_DMA_128x64, PROF_DMA_64x128, PROF_DMA_64x64, PROF_GG_128x128, PROF_GG_128x64,
                PROF_GG_64x128, PROF_GG_64x64, PROF_WGD_128x128, PROF_WGD_128x64, PROF_WGD_64x128, PROF_WGD_64x64, PROF_WG_128x128,
                PROF_WG_128x64, PROF_WG_64x128, PROF_WG_64x64, PROF_KINDS };
extern const char* const PROF_NAMES[PROF_KINDS];
void prof_mark_begin(int kind, double flops, hipStream_t s);   // no-ops unless profiling is on
void prof_mark_end(hipStream_t s);

}  // namespace icn
