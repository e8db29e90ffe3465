// Host-side chart geometry (see icn_geometry.h).  Pure C++, no device code.
#include "icn_geometry.h"

#include <algorithm>
#include <cstdlib>
#include <array>
#include <cmath>
#include <map>
#include <stdexcept>

namespace icn {

const int TAP_DA[NTAPS] = {0, 1, 0, -1, -1, 0, 1};
const int TAP_DB[NTAPS] = {0, 0, 1, 1, 0, -1, -1};

namespace {

// Lattice point (a, b) expressed in chart c's hex-axial frame.  Chart c is the closed parallelogram
// a in [0,n], b in [0,2n]; it owns a in [0,n-1], b in [1,2n]; pixel (i,j) <-> (a=i, b=j+1).
struct Pt {
    int c, a, b;
};

// The three seams between chart c and chart c-1, as affine maps of chart-c coordinates into chart c-1
// coordinates (valid in a neighbourhood of the seam):
//   seam 1 (north cap, 60 deg rotation about N):  (0,b)_c == (b,0)_{c-1},      b in [0,n]
//   seam 2 (equatorial band, pure translation):   (0,b)_c == (n,b-n)_{c-1},    b in [n,2n]
//   seam 3 (south cap, 60 deg rotation about S):  (a,2n)_c == (n,a+n)_{c-1},   a in [0,n]
inline Pt prev1(int, Pt p) { return {(p.c + 4) % 5, p.a + p.b, -p.a}; }
inline Pt prev2(int n, Pt p) { return {(p.c + 4) % 5, p.a + n, p.b - n}; }
inline Pt prev3(int n, Pt p) { return {(p.c + 4) % 5, 3 * n - p.b, p.a + p.b - n}; }
// ... and their inverses (chart c -> chart c+1).
inline Pt next1(int, Pt p) { return {(p.c + 1) % 5, -p.b, p.a + p.b}; }
inline Pt next2(int n, Pt p) { return {(p.c + 1) % 5, p.a - n, p.b + n}; }
inline Pt next3(int n, Pt p) { return {(p.c + 1) % 5, p.a + p.b - 2 * n, 3 * n - p.a}; }

constexpr int32_t NORTH = -100, SOUTH = -101;

// Owner of lattice point (a,b) of chart c, a in [-1,n], b in [0,2n+1] (one step outside the owned block).
// Returns a pixel id, NORTH or SOUTH.  At the two five-valent pixels of a chart one of the six lattice
// directions has no distinct neighbour; the seam chosen by the target's b decides which neighbour is
// duplicated.
int32_t resolve(int n, int c, int a, int b) {
    Pt p{c, a, b};
    for (int guard = 0; guard < 8; ++guard) {
        if (p.a == 0 && p.b == 0) return NORTH;
        if (p.a == n && p.b == 2 * n) return SOUTH;
        if (p.a == -1)
            p = (p.b <= n) ? prev1(n, p) : prev2(n, p);
        else if (p.b == 2 * n + 1)
            p = prev3(n, p);
        else if (p.b == 0)
            p = next1(n, p);
        else if (p.a == n)
            p = (p.b <= n) ? next2(n, p) : next3(n, p);
        else {
            if (p.a < 0 || p.a >= n || p.b < 1 || p.b > 2 * n) throw std::logic_error("icn: resolve left the chart");
            return (p.c * n + p.a) * 2 * n + (p.b - 1);
        }
    }
    throw std::logic_error("icn: resolve did not converge");
}

inline int32_t pole_code(int32_t v, int corner_mode) {
    if (v == NORTH) return corner_mode == CORNER_AVERAGE ? IDX_POLE : IDX_ZERO;
    if (v == SOUTH) return corner_mode == CORNER_AVERAGE ? IDX_POLE - 1 : IDX_ZERO;
    return v;
}

// corner pixel c of pole k at a level with n = 2^r (reference: ico_utils.py:13-18)
inline int32_t corner_pixel(int n, int k, int c) {
    return k == 0 ? (c * n) * 2 * n : ((c + 1) * n - 1) * 2 * n + (2 * n - 1);
}

void check_args(int r_in, int stride, int corner_mode) {
    if (r_in < 0 || r_in > 10) throw std::invalid_argument("icn: subdivisions out of range [0,10]");
    if (stride != 1 && stride != 2) throw std::invalid_argument("icn: stride must be 1 or 2");
    if (stride == 2 && r_in < 1) throw std::invalid_argument("icn: stride 2 needs subdivisions >= 1");
    if (corner_mode != CORNER_ZEROS && corner_mode != CORNER_AVERAGE)
        throw std::invalid_argument("icn: corner_mode must be 0 (zeros) or 1 (average)");
}

}  // namespace

void build_conv_fwd(int r_in, int stride, int corner_mode, std::vector<int32_t>& out) {
    check_args(r_in, stride, corner_mode);
    const int n = 1 << r_in, no = n / stride;
    const int Pout = 10 * no * no;
    out.assign((size_t)NTAPS * Pout, IDX_ZERO);
    for (int c = 0; c < 5; ++c)
        for (int i = 0; i < no; ++i)
            for (int j = 0; j < 2 * no; ++j) {
                const int p = (c * no + i) * 2 * no + j;
                // fine site of the output pixel (App. A.4): (i,j) or (2i, 2j+1); lattice (a,b) = (I, J+1)
                const int a = stride == 1 ? i : 2 * i, b = (stride == 1 ? j : 2 * j + 1) + 1;
                for (int t = 0; t < NTAPS; ++t)
                    out[(size_t)t * Pout + p] = pole_code(resolve(n, c, a + TAP_DA[t], b + TAP_DB[t]), corner_mode);
            }
}

int build_conv_bwd(int r_in, int stride, int corner_mode, std::vector<int32_t>& out) {
    std::vector<int32_t> fwd;
    build_conv_fwd(r_in, stride, corner_mode, fwd);
    const int n = 1 << r_in, no = n / stride;
    const int Pin = 10 * n * n, Pout = 10 * no * no;
    std::vector<std::vector<int32_t>> lists((size_t)NTAPS * Pin);
    for (int t = 0; t < NTAPS; ++t) {
        std::array<std::vector<int32_t>, 2> pole_src;
        for (int p = 0; p < Pout; ++p) {
            const int32_t q = fwd[(size_t)t * Pout + p];
            if (q >= 0)
                lists[(size_t)t * Pin + q].push_back(p);
            else if (q <= IDX_POLE)
                pole_src[IDX_POLE - q].push_back(p);
        }
        for (int k = 0; k < 2; ++k) {
            if (pole_src[k].empty()) continue;
            // d(pole_k) = W_t^T sum_{p taps pole} dy[p]; each of the 5 corner inputs receives 1/5 of it.  That is
            // the "pole mean" of dy iff the tapping pixels are exactly dy's corner pixels of pole k.
            std::vector<int32_t> want;
            for (int c = 0; c < 5; ++c) want.push_back(corner_pixel(no, k, c));
            std::sort(want.begin(), want.end());
            std::sort(pole_src[k].begin(), pole_src[k].end());
            if (want != pole_src[k]) throw std::logic_error("icn: pole taps are not the corner pixels");
            for (int c = 0; c < 5; ++c) lists[(size_t)t * Pin + corner_pixel(n, k, c)].push_back(IDX_POLE - k);
        }
    }
    size_t E = 1;
    for (auto& l : lists) E = std::max(E, l.size());
    out.assign((size_t)NTAPS * E * Pin, IDX_ZERO);
    for (int t = 0; t < NTAPS; ++t)
        for (int q = 0; q < Pin; ++q) {
            auto& l = lists[(size_t)t * Pin + q];
            // regular (single-source) entry first so that slot 0 is the fast path
            std::stable_sort(l.begin(), l.end(), [](int32_t x, int32_t y) { return (x >= 0) > (y >= 0); });
            for (size_t e = 0; e < l.size(); ++e) out[((size_t)t * E + e) * Pin + q] = l[e];
        }
    return (int)E;
}

void split_conv_bwd(int r_in, int stride, const std::vector<int32_t>& bwd_idx, int E, std::vector<int32_t>& primary,
                    VirtualRows& vr) {
    const int n = 1 << r_in, Pin = 10 * n * n;
    (void)stride;
    primary.assign((size_t)NTAPS * Pin, IDX_ZERO);
    std::vector<std::array<int32_t, NTAPS>> rows;        // virtual rows in creation order (sorted by q)
    vr = VirtualRows{};
    for (int q = 0; q < Pin; ++q) {
        // leftover[t] = entries of (t, q) that are not the primary plain pixel
        std::array<std::vector<int32_t>, NTAPS> leftover;
        size_t levels = 0;
        for (int t = 0; t < NTAPS; ++t) {
            bool have_primary = false;
            for (int e = 0; e < E; ++e) {
                const int32_t v = bwd_idx[((size_t)t * E + e) * Pin + q];
                if (v == IDX_ZERO) continue;
                if (v >= 0 && !have_primary) {
                    primary[(size_t)t * Pin + q] = v;
                    have_primary = true;
                } else {
                    leftover[t].push_back(v);
                }
            }
            levels = std::max(levels, leftover[t].size());
        }
        for (size_t l = 0; l < levels; ++l) {
            std::array<int32_t, NTAPS> row;
            for (int t = 0; t < NTAPS; ++t) row[t] = l < leftover[t].size() ? leftover[t][l] : IDX_ZERO;
            rows.push_back(row);
            vr.vq.push_back(q);
        }
    }
    vr.nv = (int)rows.size();
    vr.vidx.assign((size_t)NTAPS * std::max(vr.nv, 1), IDX_ZERO);
    for (int v = 0; v < vr.nv; ++v)
        for (int t = 0; t < NTAPS; ++t) vr.vidx[(size_t)t * vr.nv + v] = rows[v][t];
}

void build_virtual_order(const VirtualRows& vr, int& nvp, std::vector<int32_t>& order, std::vector<int32_t>& vidx_o,
                         std::vector<uint8_t>& mask32) {
    const int nv = vr.nv;
    nvp = (nv + 31) / 32 * 32;
    std::vector<unsigned> taps(nv, 0);
    for (int v = 0; v < nv; ++v)
        for (int t = 0; t < NTAPS; ++t)
            if (vr.vidx[(size_t)t * nv + v] != IDX_ZERO) taps[v] |= 1u << t;
    order.resize(nvp);
    for (int k = 0; k < nvp; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.begin() + nv, [&](int x, int y) { return taps[x] < taps[y]; });
    vidx_o.assign((size_t)NTAPS * nvp, IDX_ZERO);
    mask32.assign(nvp / 32, 0);
    for (int k = 0; k < nv; ++k) {
        for (int t = 0; t < NTAPS; ++t) vidx_o[(size_t)t * nvp + k] = vr.vidx[(size_t)t * nv + order[k]];
        mask32[k / 32] |= (uint8_t)taps[order[k]];
    }
}

void build_dma_table(const std::vector<int32_t>& idx, int E, int P, DmaTable& out) {
    out.E = E;
    out.n_slots = 0;
    out.code.assign((size_t)NTAPS * P, IDX_ZERO);
    out.slots.clear();
    std::map<std::vector<int32_t>, int> slot_of;
    std::vector<int32_t> ent;
    for (int t = 0; t < NTAPS; ++t)
        for (int p = 0; p < P; ++p) {
            ent.clear();
            for (int e = 0; e < E; ++e) {
                const int32_t c = idx[((size_t)t * E + e) * P + p];
                if (c != IDX_ZERO) ent.push_back(c);
            }
            int32_t code = IDX_ZERO;
            if (ent.size() == 1 && ent[0] >= 0) {
                code = ent[0];
            } else if (!ent.empty()) {
                ent.resize(E, IDX_ZERO);
                auto it = slot_of.find(ent);
                if (it == slot_of.end()) {
                    it = slot_of.emplace(ent, out.n_slots++).first;
                    out.slots.insert(out.slots.end(), ent.begin(), ent.end());
                }
                code = -2 - it->second;
            }
            out.code[(size_t)t * P + p] = code;
        }
}

bool build_wgrad7(const DmaTable& d, int P, int max_U, Wg7Table& out) {
    out = Wg7Table{};
    if (P <= 0 || P % WG7_PX != 0 || (size_t)NTAPS * P != d.code.size()) return false;
    const int np = P / WG7_PX;
    std::vector<std::vector<int32_t>> lists(np);
    size_t longest = 0;
    for (int q = 0; q < np; ++q) {
        std::vector<int32_t>& l = lists[q];
        for (int k = 0; k < WG7_PX; ++k)
            for (int t = 0; t < NTAPS; ++t) {
                const int32_t c = d.code[(size_t)t * P + q * WG7_PX + k];
                if (c != IDX_ZERO) l.push_back(c);
            }
        // pixels ascending (neighbouring rows of the tensor end up in one DMA instruction), side slots (negative codes) last
        std::sort(l.begin(), l.end(), [](int32_t a, int32_t b) { return (a < 0) != (b < 0) ? b < 0 : (a < 0 ? a > b : a < b); });
        l.erase(std::unique(l.begin(), l.end()), l.end());
        longest = std::max(longest, l.size());
    }
    // (one fixed list length: the kernel is instantiated for max_U rows, shorter unions are padded with rows of zeros)
    const int U = max_U;
    if ((int)longest + 1 > U || U % 4 != 0 || (size_t)(U - 1) * WG7_ROW_BYTES > 0xFFFFu) return false;
    out.U = U;
    out.npatch = np;
    out.urow.assign((size_t)np * U, IDX_ZERO);
    out.upos.assign((size_t)np * WG7_PX * 8, (uint16_t)((U - 1) * WG7_ROW_BYTES));
    for (int q = 0; q < np; ++q) {
        const std::vector<int32_t>& l = lists[q];
        for (size_t u = 0; u < l.size(); ++u) out.urow[(size_t)q * U + u] = l[u];
        for (int k = 0; k < WG7_PX; ++k)
            for (int t = 0; t < NTAPS; ++t) {
                const int32_t c = d.code[(size_t)t * P + q * WG7_PX + k];
                if (c == IDX_ZERO) continue;
                const size_t u = std::find(l.begin(), l.end(), c) - l.begin();
                out.upos[((size_t)q * WG7_PX + k) * 8 + t] = (uint16_t)(u * WG7_ROW_BYTES);
            }
    }
    return true;
}

void build_upsample(int r_in, int corner_mode, Ell& fwd, Ell& bwd) {
    check_args(r_in, 1, corner_mode);
    const int n = 1 << r_in, nf = 2 * n;
    const int Pc = 10 * n * n, Pf = 10 * nf * nf;
    std::vector<std::vector<std::pair<int32_t, float>>> rows(Pf), cols(Pc);
    auto add = [&](int q, int32_t v, float w) {
        if (v >= 0) {
            rows[q].push_back({v, w});
        } else if (corner_mode == CORNER_AVERAGE) {
            const int k = v == NORTH ? 0 : 1;
            for (int c = 0; c < 5; ++c) rows[q].push_back({corner_pixel(n, k, c), w * 0.2f});
        }
    };
    for (int c = 0; c < 5; ++c)
        for (int I = 0; I < nf; ++I)
            for (int J = 0; J < 2 * nf; ++J) {
                const int q = (c * nf + I) * 2 * nf + J;
                const int a = I, b = J + 1;   // fine lattice; coarse lattice points are the (even, even) ones
                if (a % 2 == 0 && b % 2 == 0) {
                    add(q, resolve(n, c, a / 2, b / 2), 1.0f);
                } else {
                    int a0, b0, a1, b1;
                    if (a % 2 != 0 && b % 2 == 0) { a0 = a - 1; b0 = b; a1 = a + 1; b1 = b; }
                    else if (a % 2 == 0) { a0 = a; b0 = b - 1; a1 = a; b1 = b + 1; }
                    else { a0 = a + 1; b0 = b - 1; a1 = a - 1; b1 = b + 1; }
                    add(q, resolve(n, c, a0 / 2, b0 / 2), 0.5f);
                    add(q, resolve(n, c, a1 / 2, b1 / 2), 0.5f);
                }
            }
    for (int q = 0; q < Pf; ++q)
        for (auto& e : rows[q]) cols[e.first].push_back({q, e.second});
    auto pack = [](const std::vector<std::vector<std::pair<int32_t, float>>>& m, Ell& ell) {
        size_t w = 1;
        for (auto& r : m) w = std::max(w, r.size());
        ell.rows = (int)m.size();
        ell.width = (int)w;
        ell.idx.assign(m.size() * w, IDX_ZERO);
        ell.coef.assign(m.size() * w, 0.0f);
        for (size_t r = 0; r < m.size(); ++r)
            for (size_t e = 0; e < m[r].size(); ++e) {
                ell.idx[r * w + e] = m[r][e].first;
                ell.coef[r * w + e] = m[r][e].second;
            }
    };
    pack(rows, fwd);
    pack(cols, bwd);
}

void build_upsample_pairs(int r_in, std::vector<int32_t>& out) {
    check_args(r_in, 1, CORNER_AVERAGE);
    const int n = 1 << r_in, nf = 2 * n;
    const int Pc = 10 * n * n, Pf = 10 * nf * nf;
    auto vid = [&](int c, int a, int b) {
        const int32_t v = resolve(n, c, a, b);
        return v == NORTH ? Pc : v == SOUTH ? Pc + 1 : v;
    };
    out.assign((size_t)2 * Pf, 0);
    for (int c = 0; c < 5; ++c)
        for (int I = 0; I < nf; ++I)
            for (int J = 0; J < 2 * nf; ++J) {
                const int q = (c * nf + I) * 2 * nf + J;
                const int a = I, b = J + 1;
                int a0 = a, b0 = b, a1 = a, b1 = b;
                if (a % 2 != 0 && b % 2 == 0) { a0 = a - 1; a1 = a + 1; }
                else if (a % 2 == 0 && b % 2 != 0) { b0 = b - 1; b1 = b + 1; }
                else if (a % 2 != 0) { a0 = a + 1; b0 = b - 1; a1 = a - 1; b1 = b + 1; }
                out[q] = vid(c, a0 / 2, b0 / 2);
                out[(size_t)Pf + q] = vid(c, a1 / 2, b1 / 2);
            }
}

namespace {

using AlphaKey = std::array<int, NTAPS>;        // coefficient vector in thousandths (exact for 1, 0.5, 0.1, 0.02 ...)

// composite row of fine pixel p: coarse pixel -> coefficient per original tap, and per tap the expansion of its source
struct CompRow {
    std::map<int32_t, std::array<double, NTAPS>> by_pixel;
    std::array<std::map<int32_t, double>, NTAPS> by_tap;
};

void composite_rows(int r_in, int corner_mode, std::vector<CompRow>& rows) {
    const int n = 1 << r_in, nf = 2 * n, Pf = 10 * nf * nf;
    std::vector<int32_t> fwd;
    build_conv_fwd(r_in + 1, 1, corner_mode, fwd);
    Ell uf, ub;
    build_upsample(r_in, corner_mode, uf, ub);
    rows.assign(Pf, CompRow{});
    auto add_fine = [&](CompRow& row, int t, int32_t q, double w) {       // w * up[q] into tap t
        for (int e = 0; e < uf.width; ++e) {
            const int32_t s = uf.idx[(size_t)q * uf.width + e];
            if (s < 0) continue;
            const double c = w * (double)uf.coef[(size_t)q * uf.width + e];
            auto it = row.by_pixel.find(s);
            if (it == row.by_pixel.end()) it = row.by_pixel.emplace(s, std::array<double, NTAPS>{}).first;
            it->second[t] += c;
            row.by_tap[t][s] += c;
        }
    };
    for (int p = 0; p < Pf; ++p)
        for (int t = 0; t < NTAPS; ++t) {
            const int32_t q = fwd[(size_t)t * Pf + p];
            if (q >= 0) {
                add_fine(rows[p], t, q, 1.0);
            } else if (q <= IDX_POLE) {                                      // mean of the 5 FINE corner pixels of the pole
                for (int c = 0; c < 5; ++c) add_fine(rows[p], t, corner_pixel(nf, IDX_POLE - q, c), 0.2);
            }
        }
}

AlphaKey alpha_key(const std::array<double, NTAPS>& a) {
    AlphaKey k;
    for (int t = 0; t < NTAPS; ++t) k[t] = (int)(a[t] * 1000.0 + (a[t] >= 0 ? 0.5 : -0.5));
    return k;
}

inline int parity_class(int nf, int p) { return (((p / (2 * nf)) % nf) & 1) * 2 + (p % (2 * nf) & 1); }

// The regular coefficient vectors per parity class of the fine pixel, taken from interior pixels of a level-3 -> 4 upsample
// (they do not depend on the level): templates[cls] = sorted keys; class 1 (row even, column odd) is the coarse-site class.
const std::array<std::vector<AlphaKey>, 4>& upconv_templates() {
    static const std::array<std::vector<AlphaKey>, 4> tpl = [] {
        std::array<std::vector<AlphaKey>, 4> out;
        std::vector<CompRow> rows;
        composite_rows(3, CORNER_AVERAGE, rows);
        const int nf = 16;
        for (int di = 0; di < 2; ++di)
            for (int dj = 0; dj < 2; ++dj) {
                const int I = 8 + di, J = 14 + dj;                         // interior of chart 0, away from seams and poles
                const int p = I * 2 * nf + J, cls = parity_class(nf, p);
                for (auto& e : rows[p].by_pixel) out[cls].push_back(alpha_key(e.second));
                std::sort(out[cls].begin(), out[cls].end());
            }
        if (out[1].size() != 7 || out[0].size() != 4 || out[2].size() != 4 || out[3].size() != 4)
            throw std::logic_error("icn: unexpected upsample-conv templates");
        return out;
    }();
    return tpl;
}

}  // namespace

void build_upconv_fwd(int r_in, int corner_mode, UpconvTable& out) {
    check_args(r_in, 1, corner_mode);
    if (r_in > 9) throw std::invalid_argument("icn: subdivisions out of range for upsample + conv");
    const int n = 1 << r_in, nf = 2 * n;
    out = UpconvTable{};
    out.Pc = 10 * n * n;
    out.Pf = 10 * nf * nf;
    const auto& tpl = upconv_templates();
    // virtual tap ids: site class first (7), then parity classes 0, 2, 3 (4 each)
    const int cls_order[4] = {1, 0, 2, 3};
    int vt_base[4];
    out.alpha.assign((size_t)UPCONV_TAPS * NTAPS, 0.f);
    for (int k = 0, v = 0; k < 4; ++k) {
        const int cls = cls_order[k];
        vt_base[cls] = v;
        for (auto& key : tpl[cls]) {
            for (int t = 0; t < NTAPS; ++t) out.alpha[(size_t)v * NTAPS + t] = (float)(key[t] / 1000.0);
            ++v;
        }
    }
    for (int t = 0; t < NTAPS; ++t) out.alpha[(size_t)(UPCONV_REGULAR + t) * NTAPS + t] = 1.f;

    std::vector<CompRow> rows;
    composite_rows(r_in, corner_mode, rows);
    // segment of every fine pixel: 0 site, 1..3 midpoint classes, 4 irregular
    std::vector<int> seg(out.Pf, 4);
    for (int p = 0; p < out.Pf; ++p) {
        const int cls = parity_class(nf, p);
        const auto& want = tpl[cls];
        if (rows[p].by_pixel.size() != want.size()) continue;
        std::vector<AlphaKey> have;
        for (auto& e : rows[p].by_pixel) have.push_back(alpha_key(e.second));
        std::sort(have.begin(), have.end());
        if (have == want) seg[p] = cls == 1 ? 0 : cls == 0 ? 1 : cls;      // classes 2, 3 keep their number
    }
    std::vector<int> order(out.Pf);
    for (int p = 0; p < out.Pf; ++p) order[p] = p;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return seg[x] < seg[y]; });
    out.pix.assign(order.begin(), order.end());
    int cnt[5] = {0, 0, 0, 0, 0};
    for (int p = 0; p < out.Pf; ++p) ++cnt[seg[p]];
    const int seg_cls[4] = {1, 0, 2, 3};
    int off = 0;
    for (int sg = 0; sg < 5; ++sg) {
        if (cnt[sg] == 0) continue;
        uint32_t mask = 0;
        if (sg == 4) mask = ((1u << NTAPS) - 1u) << UPCONV_REGULAR;
        else mask = ((1u << tpl[seg_cls[sg]].size()) - 1u) << vt_base[seg_cls[sg]];
        out.seg_cnt[out.nseg] = cnt[sg];
        out.seg_off[out.nseg] = off;
        out.seg_mask[out.nseg] = mask;
        off += cnt[sg];
        ++out.nseg;
    }
    out.code.assign((size_t)UPCONV_TAPS * out.Pf, IDX_ZERO);
    std::map<std::vector<std::pair<int32_t, int>>, int> slot_of;            // entry list (pixel, coef in 1e-6) -> slot
    std::vector<std::vector<std::pair<int32_t, double>>> slot_entries;
    for (int pos = 0; pos < out.Pf; ++pos) {
        const int p = out.pix[pos];
        if (seg[p] != 4) {
            const int cls = parity_class(nf, p);
            for (auto& e : rows[p].by_pixel) {
                const AlphaKey key = alpha_key(e.second);
                const int v = vt_base[cls] + (int)(std::lower_bound(tpl[cls].begin(), tpl[cls].end(), key) - tpl[cls].begin());
                out.code[(size_t)v * out.Pf + pos] = e.first;
            }
            continue;
        }
        for (int t = 0; t < NTAPS; ++t) {
            const auto& ent = rows[p].by_tap[t];
            if (ent.empty()) continue;
            int32_t code;
            if (ent.size() == 1 && std::abs(ent.begin()->second - 1.0) < 1e-9) {
                code = ent.begin()->first;
            } else {
                std::vector<std::pair<int32_t, int>> key;
                for (auto& e : ent) key.push_back({e.first, (int)(e.second * 1e6 + 0.5)});
                auto it = slot_of.find(key);
                if (it == slot_of.end()) {
                    it = slot_of.emplace(key, (int)slot_entries.size()).first;
                    slot_entries.emplace_back(ent.begin(), ent.end());
                }
                code = -2 - it->second;
            }
            out.code[(size_t)(UPCONV_REGULAR + t) * out.Pf + pos] = code;
        }
    }
    out.n_slots = (int)slot_entries.size();
    size_t E = 1;
    for (auto& l : slot_entries) E = std::max(E, l.size());
    out.E = (int)E;
    out.slot_idx.assign((size_t)out.n_slots * E, IDX_ZERO);
    out.slot_coef.assign((size_t)out.n_slots * E, 0.f);
    for (int sl = 0; sl < out.n_slots; ++sl)
        for (size_t e = 0; e < slot_entries[sl].size(); ++e) {
            out.slot_idx[(size_t)sl * E + e] = slot_entries[sl][e].first;
            out.slot_coef[(size_t)sl * E + e] = (float)slot_entries[sl][e].second;
        }
}

void build_upconv_bwd(int r_in, int corner_mode, Ell& out) {
    check_args(r_in, 1, corner_mode);
    if (r_in > 9) throw std::invalid_argument("icn: subdivisions out of range for upsample + conv");
    const int n = 1 << r_in, Pc = 10 * n * n;
    std::vector<CompRow> rows;
    composite_rows(r_in, corner_mode, rows);
    std::vector<std::vector<std::pair<int32_t, float>>> lists((size_t)Pc * NTAPS);
    for (int p = 0; p < (int)rows.size(); ++p)                              // p ascending: entries of a row sorted by p
        for (int t = 0; t < NTAPS; ++t)
            for (auto& e : rows[p].by_tap[t]) lists[(size_t)e.first * NTAPS + t].push_back({p, (float)e.second});
    size_t w = 1;
    for (auto& l : lists) w = std::max(w, l.size());
    out.rows = Pc * NTAPS;
    out.width = (int)w;
    out.idx.assign(lists.size() * w, IDX_ZERO);
    out.coef.assign(lists.size() * w, 0.0f);
    for (size_t r = 0; r < lists.size(); ++r)
        for (size_t e = 0; e < lists[r].size(); ++e) {
            out.idx[r * w + e] = lists[r][e].first;
            out.coef[r * w + e] = lists[r][e].second;
        }
}

void build_upconv_scatter(int r_in, int corner_mode, Ell& out) {
    check_args(r_in, 1, corner_mode);
    if (r_in > 9) throw std::invalid_argument("icn: subdivisions out of range for upsample + conv");
    std::vector<CompRow> rows;
    composite_rows(r_in, corner_mode, rows);
    size_t w = 1;
    for (auto& r : rows) {
        size_t k = 0;
        for (int t = 0; t < NTAPS; ++t) k += r.by_tap[t].size();
        w = std::max(w, k);
    }
    out.rows = (int)rows.size();
    out.width = (int)w;
    out.idx.assign(rows.size() * w, IDX_ZERO);
    out.coef.assign(rows.size() * w, 0.0f);
    for (size_t p = 0; p < rows.size(); ++p) {
        size_t k = 0;
        for (int t = 0; t < NTAPS; ++t)
            for (auto& e : rows[p].by_tap[t]) {
                out.idx[p * w + k] = e.first * NTAPS + t;
                out.coef[p * w + k] = (float)e.second;
                ++k;
            }
    }
}

void build_bwd_row_order(int r_in, int stride, const std::vector<int32_t>& bwd_idx, int E,
                         std::vector<int32_t>& perm, std::vector<uint8_t>& mask32) {
    const int n = 1 << r_in, Pin = 10 * n * n;
    perm.resize(Pin);
    for (int q = 0; q < Pin; ++q) perm[q] = q;
    std::vector<unsigned> taps(Pin, 0);
    for (int q = 0; q < Pin; ++q)
        for (int t = 0; t < NTAPS; ++t)
            for (int e = 0; e < E; ++e)
                if (bwd_idx[((size_t)t * E + e) * Pin + q] != IDX_ZERO) taps[q] |= 1u << t;
    if (stride == 2) {
        // Rows grouped by lattice parity class (a class uses 1-2 of the 7 taps), and inside a class by the exact tap
        // set: pixels on a chart seam reach their neighbours through other taps than interior pixels do, and every
        // second 64-row tile would otherwise contain one of them and run its taps for all 64 rows.
        auto cls = [n](int q) { const int I = (q / (2 * n)) % n, J = q % (2 * n); return (I & 1) * 2 + (J & 1); };
        std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) {
            return cls(x) != cls(y) ? cls(x) < cls(y) : taps[x] < taps[y];
        });
    }
    mask32.assign((Pin + 31) / 32, 0);
    for (int k = 0; k < Pin; ++k) mask32[k / 32] |= (uint8_t)taps[perm[k]];
}

void build_faces(int r, std::vector<int32_t>& faces) {
    check_args(r, 1, CORNER_AVERAGE);
    const int n = 1 << r, P = 10 * n * n;
    auto vid = [&](int c, int a, int b) {
        const int32_t v = resolve(n, c, a, b);
        return v == NORTH ? P : v == SOUTH ? P + 1 : v;
    };
    faces.clear();
    faces.reserve((size_t)60 * n * n);
    for (int c = 0; c < 5; ++c)
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < 2 * n; ++b) {
                faces.insert(faces.end(), {vid(c, a, b), vid(c, a + 1, b), vid(c, a, b + 1)});
                faces.insert(faces.end(), {vid(c, a + 1, b), vid(c, a + 1, b + 1), vid(c, a, b + 1)});
            }
}

void build_vertex_faces(int r, std::vector<int32_t>& vf) {
    std::vector<int32_t> faces;
    build_faces(r, faces);
    const int V = pixels(r) + 2;
    vf.assign((size_t)V * 12, -1);
    std::vector<int> cnt(V, 0);
    for (size_t f = 0; f + 2 < faces.size(); f += 3)
        for (int k = 0; k < 3; ++k) {
            const int i = faces[f + k];
            if (cnt[i] >= 6) throw std::logic_error("build_vertex_faces: vertex with more than 6 faces");
            vf[((size_t)i * 6 + cnt[i]) * 2 + 0] = faces[f + (k + 1) % 3];
            vf[((size_t)i * 6 + cnt[i]) * 2 + 1] = faces[f + (k + 2) % 3];
            ++cnt[i];
        }
}

}  // namespace icn
