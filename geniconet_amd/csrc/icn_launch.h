// Internal launch interface between the C ABI (icn_api.cpp) and the kernels (icn_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icn {

// Gather-GEMM  dst[m, :] = sum_t gather_t(src)[m, :] . wt[t]^T (+ bias).
// Pair forms (two convolutions sharing their input, reference models.py:37-39,59-60) are plain concatenations:
//   * src2 != null: the K (channel) axis is [src | src2], K / 2 channels each (bwd-data of a pair: dy0 | dy1);
//   * dst2 != null: the N axis is [dst (N0 columns) | dst2 (N - N0 columns)] (forward of a pair).
// Class-major row map of the composite (upsample + conv) GEMM: rows [row0[s], row0[s] + B * cnt[s]) of segment s are
// (sample b, list position off[s] + j), b-major; every segment starts at a multiple of 8 tiles of rows (8 XCDs: each gets an
// equal share of every segment; the rest is padding that computes nothing), so a tile lies in one segment and runs only that
// segment's taps `mask[s]`.  The caller fills nseg, B, cnt, off, mask; row0 (and the launch's M) follow from the tile height
// the launcher picks.  nseg == 0: plain rows m = b * Pd + q (every other launch).
constexpr int MAX_SEGS = 5;
struct RowSegs {
    int nseg;
    int B;
    int row0[MAX_SEGS], cnt[MAX_SEGS], off[MAX_SEGS];
    unsigned mask[MAX_SEGS];
};

struct GatherGemmArgs {
    const float* src;       // (B, Ps, K)  [or (B, Ps, K/2) with src2]
    const float* src2;      // second half of the K axis, or null
    const float* wt;        // [7][N][K]
    const float* bias;      // [N] or null
    float* dst;             // (B, Pd, N)  [or (B, Pd, N0) with dst2]
    float* dst2;            // (B, Pd, N - N0) or null
    int N0;                 // columns of dst (= N without dst2)
    const int32_t* idx;     // [7][E][Pd] full table (register-staged fall-back kernel)
    const int32_t* dcode;   // DmaTable code [7][Pd] of the same table (LDS-DMA kernel), or null
    const float* side;      // (B, n_slots, K) side buffer filled by launch_conv_prologue, or null
    const float* side2;     // same for src2
    int n_slots;
    const int32_t* perm;    // [Pd] or null
    const uint32_t* mask32; // [Pd/32] tap mask per 32 rows (one word each: scalar loads), or null
    const uint32_t* mask32_host;   // host copy of mask32 (stable pointer) or null; with mask_key != 0 the launcher deals the tiles
    int mask_key;                  // of the launch to the workgroups by their step counts (tile lists, cached under this key)
    int M, Ps, Pd, K, N, E, ns;
    double algo_flops;      // algorithmic FLOPs of this launch (profiling only)
    int T;                  // taps of wt / dcode: 0 = the 7 hex taps; > 7: virtual taps of a composite table (LDS-DMA kernel only)
    RowSegs segs;           // nseg > 0: class-major rows (M = padded row count, perm = list position -> dst pixel)
    float* sk_part;         // stream-K scratch: conv_sk_part_bytes() bytes, or null (no stream-K)
    int* sk_flag;           // CONV_SK_FLAGS ints zeroed before the launch (PrologueArgs::zero), with sk_part
    int arith_bn;           // 0: fp32 arithmetic, wt = [T][N][K] fp32.  64 / 128 (= conv_b3_bn(args) asked BEFORE the prologue): the
                            // three-way bf16 split, wt = the prologue's bf16 image for that column tile (PrologueArgs::packed_b3)
};
// arithmetic of the channel-mixing contraction (icn_kernels.hip): 0 exact fp32, 1 three-way bf16 split; process default ICN_ARITH
int arith_mode();
int set_arith_mode(int mode);
unsigned build_flags();     // ICN_EXP | ICN_CONV_WAVES_DEFAULT << 16 | ICN_CHAIN_PRIO << 24 of this build (icn_build_flags)
int conv_b3_bn(const GatherGemmArgs& a);
constexpr int CONV_SK_ERROR = 1024;                        // flag words: [0, 1024) one per block (the grid's upper bound)
constexpr int CONV_SK_FLAGS = 1032;
// The device's asynchronous failure word (pinned host memory mapped to the device; bit 0: a stream-K partner never arrived)
// and its host-side read (clear != 0: reset the bits returned).  icn_device_status in include/icn.h.
void set_trace_buffer(void* p, size_t n_u64);   // developer: icn_debug_trace
int* device_status_word();
int device_status(int clear);
size_t conv_sk_part_bytes();                               // partial-tile slots of the largest stream-K grid
// per-mille speed factors the stream-K plan assumes for the k-th blocks to arrive on a CU (icn_streamk.h: sk_boundaries), for
// `occ` blocks per CU (1..3); returns the table or null (equal shares).  ICN_SK_FAC="a,b[,c]" overrides the defaults of the
// matching occupancy, ICN_SK_FAC=0 switches the weighting off (developer A/B).
const int* sk_speed_factors(int occ);
}  // namespace icn
#include <vector>
namespace icn {
// Tile lists of a masked launch (host side; icn_table_tile_lists): h = [grid + 1] offsets, then the tile ids of each workgroup
void build_tile_lists(const uint32_t* mask32_host, int M, int Pd, int bm, int ntn, int ntiles, int grid, int occ, std::vector<int>& h);


struct WgradArgs {
    const float* x;         // (B, Ps, Cin)
    const float* dy;        // (B, Pd, Cout)  [or (B, Pd, Cout0) with dy2: Cout = Cout0 + Cout1]
    const float* dy2;       // (B, Pd, Cout - Cout0) or null: weight gradient of a pair sharing x
    int Cout0;              // channels of dy (= Cout without dy2)
    const int32_t* idx;     // forward table [7][Pd]
    const int32_t* dcode;   // its DmaTable code [7][Pd] (LDS-DMA kernel), or null
    const float* side;      // (B, n_slots, Cin) pole means of x filled by launch_conv_prologue, or null
    int n_slots;
    float* partial;         // [S][7][Cin][Cout]
    float* bias_partial;    // [S][Cout] (null when dbias is null)
    float* dw;              // [Cout0][Cin][7]
    float* dbias;           // [Cout0] or null
    float* dw2;             // [Cout - Cout0][Cin][7] (pair only)
    float* dbias2;          // [Cout - Cout0] or null
    int M, Ps, Pd, Cin, Cout, ns;
    double algo_flops;
    int y_taps;             // 7: dy is the tap-major aggregate g (M, 7, Cout) of icn_upconv_bwd and dcode one table [Pd]; else 0
    int identity_rows;      // the caller GUARANTEES dcode is the identity (row m reads x row m): only then may the launcher take k_wgrad_dense,
                            // which compiles the gather away; 0 with y_taps: the general per-tap kernel walks dcode
    const int32_t* w7_rows; // patch form of dcode for the all-taps kernel k_wgrad7 (icn_geometry.h: Wg7Table), or null
    const uint16_t* w7_pos;
    int w7_U;
};

bool gather_gemm_supported(int K, int N);
bool conv_dma_usable(const GatherGemmArgs& a);     // the LDS-DMA kernel can run these arguments (pairs need it)
void launch_gather_gemm_auto(const GatherGemmArgs& a, hipStream_t s);

bool wgrad_supported(int Cin, int Cout);
bool wgrad_pair_supported(int M, int Ps, int Pd, int Cin, int Cout0, int Cout1);
// row splits (= partial slabs); Cout0 < Cout: pair whose co tiles must not straddle the two outputs
int wgrad_splits(int M, int Cin, int Cout, int Cout0 = 0);
// slabs the all-taps kernel k_wgrad7 writes for this shape (0: shape outside its plan); the workspace holds the larger count
int wgrad7_splits(int M, int Pd, int Cin, int Cout, int Cout0);
void launch_wgrad(const WgradArgs& a, hipStream_t s);
bool stem_supported(int Cin, int Cout);
void launch_stem_fwd(const float* x, const float* w, const float* bias, float* y, const int32_t* idx, int M, int Ps, int Pd,
                     int Cin, int Cout, int ns, hipStream_t s);

// g[b, rows ? rows[r] : r, :] (+)= sum_e coef[r][e] * [dy0 | dy1][b, idx[r][e], :]   (aggregate of icn_upconv_bwd; g has
// rows_total rows of C0 + C1 channels per sample; acc: add to what is there)
void launch_upconv_gather(const float* dy0, const float* dy1, float* g, const int32_t* idx, const float* coef, const int32_t* rows,
                          int B, int Pin, int nrows, int rows_total, int C0, int C1, int W, int acc, hipStream_t s);

// LDS-staged forms of the two sparse passes (icn_kernels.hip): a workgroup owns a patch of the output grid; prow [npatch][umax] =
// the union of the patch's source rows (-1 padded, umax a multiple of 32), plocal [outputs][W] = each output's entries as
// positions in its patch's list (0xFFFF: none), gridW = width of the output pixel grid (patches tile it exactly)
struct PatchTab { const int32_t* prow; const uint16_t* plocal; int npatch, umax, gridW; };
bool upconv_patch_usable(const PatchTab& t, int B, size_t rows_per_sample, int C0, int C1);
void launch_upconv_scatter_lds(const float* z, const float* bias, float* y0, float* y1, const PatchTab& t, const float* coef, int B,
                               int zrows, int Pout, int C0, int C1, hipStream_t s);
void launch_upconv_gather_lds(const float* dy0, const float* dy1, float* g, const PatchTab& t, const int32_t* cls,
                              const float* cls_coef, int ncls, int B, int Pin, int Pc, int C0, int C1, hipStream_t s);

// the same aggregate per coarse pixel for all 7 taps at once: srcs [Pc][20] fine rows (-1 padded), coefd [Pc][20][8]
constexpr int UPCONV_PX_CLASSES = 32;   // capacity of k_upconv_gather_px's LDS coefficient-class table
void launch_upconv_gather_px(const float* dy0, const float* dy1, float* g, const int32_t* srcs, const int32_t* cls,
                             const float* cls_coef, int ncls, int B, int Pin, int Pc, int C0, int C1, hipStream_t s);

// [y0 | y1][b, rows ? rows[r] : r, :] (+)= bias + sum_e coef[r][e] * z[b, idx[r][e], :]   (dense forward path of
// icn_upconv_fwd; z has zrows rows of C0 + C1 channels per sample, the outputs Pout rows of C0 / C1 channels)
void launch_upconv_scatter(const float* z, const float* bias, float* y0, float* y1, const int32_t* idx, const float* coef,
                           const int32_t* rows, int B, int zrows, int nrows, int Pout, int C0, int C1, int W, int acc, hipStream_t s);

void launch_spmm_ell(const float* in, float* out, const int32_t* idx, const float* coef, int B, int Pin, int Pout, int C,
                     int W, hipStream_t s);

void launch_conv_generic(const float* src, const float* w, const float* bias, float* dst, const int32_t* idx, int B, int Ps,
                         int Pd, int K, int N, int E, int ns, int transpose, hipStream_t s);

// dst[b, q[v], :] += src[b, v, :], src (B, nvp, C), v < nv   (q sorted; rows of one q are summed in order by one thread
// => deterministic)
void launch_row_scatter_add(const float* src, float* dst, const int32_t* q, int B, int nv, int nvp, int P, int C, hipStream_t s);

// One launch ahead of a conv call: repack the weights (skipped when w is null) and fill the side buffer of a DmaTable
// (skipped when side is null or n_slots == 0) from `src` (B, Ps, K).
// Pairs: w2 (Cout2, Cin, 7) is packed behind w along the output-channel axis, bias / bias2 are concatenated into bias_cat
// (skipped when null), src2 / side2 is a second source tensor of the same shape.
struct PrologueArgs {
    const float* w; const float* w2; float* packed; int Cout, Cout2, Cin, transpose;
    const float* bias; const float* bias2; float* bias_cat;
    const float* src; const float* src2; const int32_t* slots; float* side; float* side2; int n_slots, E, B, Ps, K, ns;
    int* zero; int n_zero;  // words to clear (the stream-K flags of the GEMM this prologue precedes), or null
    int b3_flat_n;                // transpose 0 only: the image is ONE tap of N = 7 * (Cout + Cout2) columns (icn_upconv_fwd's dense GEMM)
    void* packed_b3; int b3_bn;   // bf16 x 3 image of the same B operand for column tile b3_bn (6 bytes per weight), or null; `packed`
                                  // may then be null (no fp32 copy wanted)
};
void launch_conv_prologue(const PrologueArgs& a, hipStream_t s);

// Same for a composite upsample + conv call: effective weights packed[NV][Cout + Cout2][Cin] = sum_t alpha[v][t] W_t,
// concatenated bias, side buffer side[b][slot] = sum_e slot_coef * x[b, slot_idx] (irregular rows; src = x (B, Ps, Cin)).
struct UpconvPrologueArgs {
    const float* w; const float* w2; float* packed; int Cout, Cout2, Cin, NV; const float* alpha;
    const float* bias; const float* bias2; float* bias_cat;
    const float* src; const int32_t* slot_idx; const float* slot_coef; float* side; int n_slots, E, B, Ps;
};
void launch_upconv_prologue(const UpconvPrologueArgs& a, hipStream_t s);

// ---- fused BatchNorm (+ residual) + ReLU (icn_bn.hip); stat = [mean | invstd] (2*C), sums = NS*C, ws = chunks*NS*C DOUBLES (NS <= 4)
bool bn_supported(int C);
int bn_chunks(int M);
void launch_bn_stats(const float* x, int M, int C, float eps, float momentum, float* running_mean, float* running_var, float* stat,
                     float* ws, hipStream_t s);
void launch_bn_stats2(const float* a, const float* b, int M, int C, float eps_a, float mom_a, float* rm_a, float* rv_a, float* stat_a,
                      float eps_b, float mom_b, float* rm_b, float* rv_b, float* stat_b, float* ws, hipStream_t s);   // ws: chunks * 4 * C
void launch_bn_relu_fwd(const float* a, const float* b, const float* stat_a, const float* stat_b, const float* ga, const float* ba,
                        const float* gb, const float* bb, float* y, int M, int C, hipStream_t s);
void launch_bn_relu_bwd(const float* dy, const float* a, const float* b, const float* stat_a, const float* stat_b, const float* ga,
                        const float* ba, const float* gb, const float* bb, float* da, float* db, float* sums, float* ws, int M, int C,
                        float* dbeta_a, float* dgamma_a, float* dbeta_b, float* dgamma_b, hipStream_t s);   // parameter gradients (nullable)

// ---- fused 1x1 head + tanh (icn_bn.hip); ws = head_chunks(M) * 4 * (Cin + 4) floats
bool head_supported(int Cin, int Cout);
int head_chunks(int M);
void launch_head_fwd(const float* x, const float* w, const float* bias, float* y, int M, int Cin, int Cout, hipStream_t s);
void launch_head_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw, float* db, float* ws,
                     int M, int Cin, int Cout, hipStream_t s);

// lap_mode: Laplacian convention, bit 0 = v - mean(ring) instead of mean(ring) - v, bit 1 = times the valence (sum form)
// ---- point-to-point loss (icn_loss.hip): terms[4] = {pos, nor, lap, weighted total}; partial = 3 * p2p_loss_blocks floats
int p2p_loss_blocks(int B, int P);
void launch_p2p_loss_fwd(const float* grid, const float* target, const int32_t* vf, float* partial, float* terms, int B, int P, int n,
                         float f_pos, float f_nor, float f_lap, int lap_mode, hipStream_t s);
void launch_p2p_loss_bwd_pos(const float* grid, const float* target, const float* upstream, float f_pos, float* dgrid, int B, int P,
                             int n, hipStream_t s);
// all three terms (aux = 6 * B * (P + 2) floats; unused, may be null, when f_nor = f_lap = 0)
void launch_p2p_loss_bwd(const float* grid, const float* target, const int32_t* vf, const float* upstream, float f_pos, float f_nor,
                         float f_lap, int lap_mode, float* dgrid, float* aux, int B, int P, int n, hipStream_t s);

// ---- KL term + reparameterisation of the VAE (icn_loss.hip); partial = kld_blocks(n) floats
int kld_blocks(size_t n);
void launch_kld_fwd(const float* mu, const float* logvar, size_t n, float* out, float* partial, hipStream_t s);
void launch_kld_bwd(const float* mu, const float* logvar, const float* upstream, size_t n, float* dmu, float* dlogvar, hipStream_t s);
void launch_reparam_fwd(const float* mu, const float* logvar, const float* eps, size_t n, float* z, hipStream_t s);
void launch_reparam_bwd(const float* dz, const float* logvar, const float* eps, size_t n, float* dmu, float* dlogvar, hipStream_t s);

// point-to-mesh distance (icn_loss.hip): squared distance, closest face and feature kind per point; faces (F, 3) shared
void launch_point_to_mesh(const float* pts, const float* vts, const int32_t* faces, int B, int P, int V, int F, float* dist,
                          int32_t* face, int32_t* kind, hipStream_t s);

// developer routing flags (ICN_DEBUG / icn_set_debug_flags): 16 = convs on k_gather_gemm, 32 = wgrads on k_wgrad,
// 128 = no stream-K, 256 = fault injection: every stream-K finisher reports its partners lost (tests of the failure path),
// 512 = masked launches walk tiles b, b + G, ... instead of the balanced tile lists, 1024 = the sparse passes of the decoder-block
// head on the row-per-thread kernels instead of the LDS-staged ones, 2048 = weight gradients on the per-tap kernel k_wgrad_dma
// instead of the all-taps kernel k_wgrad7, 4096 = stride-2 weight gradients on k_wgrad7 too (default: stride 1 only),
// 8192 = every stream-K workgroup sleeps ~50 us before it parks its piece (tests: the finishers really wait for their partners)
int debug_flags();
int set_debug_flags(int flags);

// ---- optional per-launch HIP-event timing of the MFMA kernels (bench.py's live roofline measurement) ----------
enum ProfKind { PROF_DMA_128x128 = 0, PROF_DMA_128x64, PROF_DMA_64x128, PROF_DMA_64x64, PROF_GG_128x128, PROF_GG_128x64,
                PROF_GG_64x128, PROF_GG_64x64, PROF_WGD_128x128, PROF_WGD_128x64, PROF_WGD_64x128, PROF_WGD_64x64, PROF_WG_128x128,
                PROF_WG_128x64, PROF_WG_64x128, PROF_WG_64x64,
                PROF_DMAS_128x128, PROF_DMAS_128x64, PROF_DMAS_64x128, PROF_DMAS_64x64,   // k_conv_dma<.., true>: class-major rows
                PROF_DMAK_64x128, PROF_DMAK_64x64, PROF_DMAKS_64x128, PROF_DMAKS_64x64,    // k_conv_dma_sk<.., SEG>: stream-K form
                PROF_WG7,                                                                   // k_wgrad7<4>: all-taps weight gradient
                PROF_DMA8_64x128, PROF_DMAS8_64x128, PROF_DMAK8_64x128, PROF_DMAKS8_64x128,  // eight-wave forms of the 64 x 128 tile
                PROF_DENSEK_64x128, PROF_DENSEK_64x64,       // k_conv_dense_sk: one-tap dense GEMMs on the plain path
                PROF_SINGLEK_64x128, PROF_SINGLEK_64x64,     // k_conv_single_sk: single convolutions (no second tensors)
                PROF_WGDENSE_128x128, PROF_WGDENSE_128x64, PROF_WGDENSE_64x128, PROF_WGDENSE_64x64,   // k_wgrad_dense: the heads' dW = x^T g
                PROF_B3K_128x128, PROF_B3K_128x64, PROF_B3DENSEK_128x128, PROF_B3DENSEK_128x64,       // ARITH = 1 forms (round 6)
                PROF_WG7_B3, PROF_WGD_B3, PROF_WGDENSE_B3, PROF_B3_MASKED, PROF_B3SINGLEK_128x128, PROF_B3SINGLEK_128x64,
                PROF_KINDS };
extern const char* const PROF_NAMES[PROF_KINDS];
void prof_mark_begin(int kind, double flops, hipStream_t s);   // no-ops unless profiling is on
void prof_mark_end(hipStream_t s);

// icn_optim.hip: Adam over a list of tensors, one launch per ADAM_MAX_TENSORS of them with equal step counts (table in the kernel arguments)
constexpr int ADAM_MAX_TENSORS = 96;
void launch_adam(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const size_t* numel,
                 const float* step_size, const float* bc2_sqrt, double beta1, double beta2, double eps, double weight_decay, hipStream_t s,
                 const float* scal_dev = nullptr);   // scal_dev: device {step_size, bc2_sqrt} for ALL tensors (graph capture)

}  // namespace icn
