// Host-side icosahedral chart geometry: index tables consumed by the HIP kernels.
//
// Replaces the (absent) icocnn pad/index buffers behind IcoConvS2S / IcoUpsampleS2S
// (reference call sites: models.py:13-14,25-33,45-55).  Convention: SURVEY.md App. A.
// Derivation here is by the three seam affine maps between neighbouring charts (not by the
// pad-slice table the oracle uses), so the two can be cross-checked.
#pragma once
#include <cstdint>
#include <vector>

namespace icn {

enum CornerMode { CORNER_ZEROS = 0, CORNER_AVERAGE = 1 };

// Index codes shared by every table:
//   >= 0        pixel id (row-major over the (5n, 2n) grid of the source level)
//   IDX_ZERO    contributes nothing
//   IDX_POLE-k  mean over the 5 corner pixels of pole k (0 = north, 1 = south) of the source tensor
constexpr int32_t IDX_ZERO = -1;
constexpr int32_t IDX_POLE = -2;   // north = -2, south = -3

constexpr int NTAPS = 7;
// Tap order (centre, then hex ring counter-clockwise in lattice coords (a=row, b=col)).
extern const int TAP_DA[NTAPS];
extern const int TAP_DB[NTAPS];

inline int pixels(int r) { return 10 << (2 * r); }

// Forward gather table of a stride-s conv reading level r_in: out[t * P_out + p].
void build_conv_fwd(int r_in, int stride, int corner_mode, std::vector<int32_t>& out);

// Transposed gather table (for bwd-data): out[(t * E + e) * P_in + q] indexes dy pixels (level
// r_in - log2(stride)); E is the max multiplicity (returned).
int build_conv_bwd(int r_in, int stride, int corner_mode, std::vector<int32_t>& out);

// Split of the transposed table for the fast dgrad path.  `primary` [7][P_in] keeps at most one plain pixel per
// (tap, row).  Every other contribution (second / third entries, pole means) becomes an entry of a "virtual row":
// virtual row v (one per affected input pixel and level) has at most one code per tap, vidx [7][nv], and its GEMM
// result is added to input pixel vq[v].  vq is sorted, so the virtual rows of one pixel are adjacent.
struct VirtualRows {
    int nv = 0;
    std::vector<int32_t> vidx;   // [7][nv]: pixel of dy, IDX_ZERO, or IDX_POLE - k
    std::vector<int32_t> vq;     // [nv] target input pixel
};
void split_conv_bwd(int r_in, int stride, const std::vector<int32_t>& bwd_idx, int E, std::vector<int32_t>& primary,
                    VirtualRows& vr);

// Row order of the virtual-row GEMM.  A virtual row uses ~1 of the 7 taps, so GEMM rows are sorted by the set of taps
// they use and padded to nvp = a multiple of 32 rows; 32-row groups then skip the taps none of their rows use (mask32,
// as for stride-2 bwd-data).  order[k] = virtual row computed by GEMM row k (k >= nv: padding, order[k] = k), i.e. the
// row of the (B, nvp, C) result it is stored to; vidx_o [7][nvp] = the codes in GEMM row order (padding: IDX_ZERO).
void build_virtual_order(const VirtualRows& vr, int& nvp, std::vector<int32_t>& order, std::vector<int32_t>& vidx_o,
                         std::vector<uint8_t>& mask32);

// Gather table in the form the LDS-DMA kernels consume: one code per (tap, row), code [7][P]:
//   >= 0       a single plain pixel (DMA'd straight from the source tensor)
//   IDX_ZERO   nothing
//   -2 - s     slot s of a small per-sample "side" buffer that a pre-pass fills with the sum of the entries
//              slots[s * E + e] (pixels / pole means): everything that is not a single pixel -- pole means,
//              duplicated transposed entries.  Equal entry lists share a slot (a forward table has <= 2 slots: the
//              two pole means).
struct DmaTable {
    int n_slots = 0, E = 1;
    std::vector<int32_t> code;    // [7][P]
    std::vector<int32_t> slots;   // [n_slots][E]
};
void build_dma_table(const std::vector<int32_t>& idx, int E, int P, DmaTable& out);

// Patch form of a forward DmaTable for the all-taps weight-gradient kernel (k_wgrad7): the output pixels of a sample are cut
// into patches of WG7_PX consecutive pixels; a patch's 7 x WG7_PX gathered rows overlap heavily (the in-row taps of
// neighbouring pixels are each other's centre rows), so the kernel stages the UNION of the patch's source rows once:
//   urow [npatch][U]            DmaTable codes of the union rows (pixel / -2 - slot / IDX_ZERO), sorted; row U - 1 is always
//                               IDX_ZERO (a row of zeros: what a tap that reads nothing points at)
//   upos [npatch][WG7_PX][8]    byte offset (row index * row_bytes) of tap t's row of pixel k inside the staged union
//                               ([k][7] is padding); row_bytes = 256 (64 channels per workgroup tile)
// U = max_U (a multiple of 16; shorter unions are padded).  Returns false when P is not a multiple of WG7_PX or a patch's
// union + 1 exceeds max_U (stride 2: ~100 rows) -- the caller then keeps the per-tap kernel.
constexpr int WG7_PX = 16;
constexpr int WG7_ROW_BYTES = 256;
struct Wg7Table {
    int U = 0, npatch = 0;
    std::vector<int32_t> urow;
    std::vector<uint16_t> upos;
};
bool build_wgrad7(const DmaTable& d, int P, int max_U, Wg7Table& out);

// ELL sparse matrices of the r -> r+1 upsample and of its transpose.
struct Ell {
    int rows = 0, width = 0;
    std::vector<int32_t> idx;    // rows * width, IDX_ZERO padded
    std::vector<float> coef;     // rows * width
};
void build_upsample(int r_in, int corner_mode, Ell& fwd, Ell& bwd);
// Raw endpoints of the upsample: out[q], out[Pf + q] = the two level-r vertex ids (pixels, or P / P+1 for the
// N / S pole) whose mean is fine pixel q (equal at coarse sites).
void build_upsample_pairs(int r_in, std::vector<int32_t>& out);

// Composite table of  conv_stride1(upsample(x))  -- the first two operators of every decoder block of the reference
// (models.py:58-60: conv00(upsample00(x)), conv10(upsample10(x))) -- as ONE gather-GEMM over the COARSE tensor.
// The upsample is linear (copy at coarse sites, mean of the two edge endpoints elsewhere), so
//     y[p] = bias + sum_t W_t up[nbr_t(p)] = bias + sum_{s in S(p)} (sum_t alpha_{p,s}[t] W_t) x[s]
// and away from the 12 singular vertices the coefficient vectors alpha only depend on the parity class of the fine pixel p:
// a coarse-site pixel reads 7 coarse pixels, an edge-midpoint pixel only the 4 corners of the two triangles on its edge.
// That is 19 "virtual taps" (7 + 3 x 4) with effective weights W_eff[v] = sum_t alpha[v][t] W_t, and on average
// 4.75 instead of 7 K-blocks per output pixel (0.68 of the multiply-adds of the unfused pair of operators), without the
// 4x larger upsampled tensor ever existing.  Rows are ordered class-major (all samples' site pixels, then the three
// midpoint classes, then the irregular rows) so that a GEMM tile only runs the virtual taps of its class.
// Irregular rows (within two steps of a pole or of a five-valent corner pixel; a few hundred per sample) keep the 7
// original taps, W_eff[19 + t] = W_t, and gather the upsampled neighbour value itself from a small side buffer
// (slot = sum_e coef_e x[pixel_e], filled by the call's prologue).
constexpr int UPCONV_REGULAR = 19;             // virtual taps 0..18
constexpr int UPCONV_TAPS = UPCONV_REGULAR + NTAPS;   // + the 7 original taps (irregular rows)
constexpr int UPCONV_MAX_SEG = 5;
struct UpconvTable {
    int Pc = 0, Pf = 0;                    // coarse / fine pixels per sample
    std::vector<float> alpha;              // [UPCONV_TAPS][NTAPS]
    int nseg = 0;                          // row segments (classes) in order; empty classes are dropped
    int seg_cnt[UPCONV_MAX_SEG] = {};      // fine pixels per sample in the segment
    int seg_off[UPCONV_MAX_SEG] = {};      // first list position of the segment (prefix sum of seg_cnt)
    uint32_t seg_mask[UPCONV_MAX_SEG] = {};// virtual taps the segment's rows use
    std::vector<int32_t> pix;              // [Pf] list position -> fine pixel (class-major)
    std::vector<int32_t> code;             // [UPCONV_TAPS][Pf] by list position: >= 0 coarse pixel, IDX_ZERO, -2 - slot
    int n_slots = 0, E = 1;
    std::vector<int32_t> slot_idx;         // [n_slots][E] coarse pixels (IDX_ZERO padded)
    std::vector<float> slot_coef;          // [n_slots][E]
};
void build_upconv_fwd(int r_in, int corner_mode, UpconvTable& out);

// Backward of the same pair of operators through ONE coarse-level aggregate of dy.  With U the upsample matrix and
// nbr_t the fine conv's gather,
//     g_t[s] = sum_p U[nbr_t(p), s] dy[p]        (7 coarse tensors: dy shifted by tap t, then upsample-transposed)
// gives both gradients as DENSE coarse-level contractions, a quarter of the fine level's multiply-adds:
//     dx[s] = sum_t W_t^T g_t[s]                 dW_t = sum_s x[s]^T g_t[s]
// (and, with corner_mode 'average', dbias = sum_s g_0[s], because every row of U sums to one).  `out` is the ELL matrix of
// dy -> g: row s * 7 + t lists the fine pixels p and coefficients of g_t[s] (7 entries away from the singular vertices).
void build_upconv_bwd(int r_in, int corner_mode, Ell& out);
// The transpose of that matrix serves the FORWARD the same way:  z_t[s] = W_t x[s] for all 7 taps is a dense coarse-level
// GEMM (N = 7 * Cout, a quarter of the multiply-adds), and  y[p] = bias + sum_t sum_s U[nbr_t(p), s] z_t[s]  is an
// HBM-bound sparse combination.  `out`: row p (fine pixel) lists the rows s * 7 + t of z and their coefficients (12-13
// entries away from the singular vertices).
void build_upconv_scatter(int r_in, int corner_mode, Ell& out);

// Row permutation + per-32-row tap masks for stride-2 bwd-data (rows grouped by lattice parity class so
// that all-empty taps can be skipped tile-wise).  perm[k] = input pixel handled by row k.
void build_bwd_row_order(int r_in, int stride, const std::vector<int32_t>& bwd_idx, int E,
                         std::vector<int32_t>& perm, std::vector<uint8_t>& mask32);

// Faces (20*4^r, 3) in the reference vertex order (grid row-major, then N, S).
void build_faces(int r, std::vector<int32_t>& faces);

// Incident faces of every vertex, for the mesh terms of the loss (reference losses.py:52-57): vf[(i * 6 + f) * 2 + {0, 1}] =
// the other two vertices (p, q) of the f-th face at vertex i, in the face's orientation rotated so that i comes first
// (cross(p - i, q - i) is the face normal of generate.py:28-31 whichever corner i is); -1 padded (the 12 five-valent
// vertices have 5 faces).  On the closed mesh the p entries of a vertex are exactly its one-ring, each neighbour once.
void build_vertex_faces(int r, std::vector<int32_t>& vf);

}  // namespace icn
