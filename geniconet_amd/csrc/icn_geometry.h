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
