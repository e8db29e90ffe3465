// C ABI of libicn (see include/icn.h): argument checking, per-device index-table cache, kernel dispatch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/icn.h"
#include "icn_geometry.h"
#include "icn_launch.h"
#include "icn_streamk.h"

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) {
    g_err = msg;
    return -1;
}

#define ICN_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// icn_host_selfcheck: build every table on the host and skip the device copy (sanitizer runs of the host code need no GPU)
thread_local bool g_dry_run = false;
thread_local size_t g_dry_elems = 0;

template <typename T>
T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    if (v.empty()) return d;
    if (g_dry_run) {
        g_dry_elems += v.size();
        return d;
    }
    ICN_HIP(hipMalloc(reinterpret_cast<void**>(&d), v.size() * sizeof(T)));
    ICN_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

// device copy of an icn::DmaTable (the form the LDS-DMA kernels consume)
struct DevDma {
    int32_t* code = nullptr;   // [7][P]
    int32_t* slots = nullptr;  // [n_slots][E]
    int n_slots = 0, E = 1;
};
DevDma upload_dma(const std::vector<int32_t>& idx, int E, int P) {
    icn::DmaTable h;
    icn::build_dma_table(idx, E, P, h);
    DevDma d;
    d.code = upload(h.code);
    d.slots = upload(h.slots);
    d.n_slots = h.n_slots;
    d.E = E;
    return d;
}
struct ConvTables {
    int Pin = 0, Pout = 0, n_in = 0, n_out = 0, E = 1;
    int32_t* fwd = nullptr;    // [7][Pout]
    int32_t* bwd = nullptr;    // [7][E][Pin]
    int32_t* bwd1 = nullptr;      // [7][Pin] primary (single plain pixel) entries of the transposed table
    int nv = 0, nvp = 0;          // virtual rows: extra transposed entries, run as a second small GEMM of nvp >= nv rows
    int32_t* vidx = nullptr;      // [7][nvp] codes in GEMM row order (rows sorted by the taps they use, padded)
    int32_t* vorder = nullptr;    // [nvp] GEMM row -> row of the (B, nvp, C) result (= virtual row; vq order)
    uint32_t* vmask32 = nullptr;  // [nvp/32] taps in use per 32 GEMM rows
    int32_t* vq = nullptr;        // [nv] target input pixel (sorted)
    int32_t* perm = nullptr;   // [Pin] (stride 2 only)
    int32_t* bwd_perm = nullptr;  // [7][E][Pin] transposed table in permuted row order (stride 2 only)
    uint32_t* mask32 = nullptr; // [Pin/32] (stride 2 only)
    std::vector<uint32_t> mask32_h;   // host copy: the launcher balances the tiles of a masked launch by their step counts
    int key = 0;                  // (r, stride, mode) packed: names this table set in the launcher's tile-list cache
    DevDma d_fwd, d_bwd, d_bwd1, d_virt, d_bwdp;   // DmaTable forms of fwd, bwd, bwd1, vidx, bwd_perm
    int32_t* w7_rows = nullptr;   // patch form of d_fwd for the all-taps weight gradient (icn::Wg7Table), or null (stride 2, r < 2)
    uint16_t* w7_pos = nullptr;
    int w7_U = 0;
};
struct UpTables {
    int Pc = 0, Pf = 0, Wf = 0, Wb = 0;
    int32_t *idx_f = nullptr, *idx_b = nullptr;
    float *coef_f = nullptr, *coef_b = nullptr;
};

// device copy of an icn::UpconvTable (composite upsample + conv over the coarse tensor)
struct UpconvDev {
    int Pc = 0, Pf = 0, n_slots = 0, E = 1, nseg = 0;
    int seg_cnt[icn::UPCONV_MAX_SEG] = {}, seg_off[icn::UPCONV_MAX_SEG] = {};
    uint32_t seg_mask[icn::UPCONV_MAX_SEG] = {};
    int32_t *pix = nullptr, *code = nullptr, *slot_idx = nullptr;
    float *alpha = nullptr, *slot_coef = nullptr;
};

// An ELL matrix on the device, split into a main part of fixed width (every row) and an overflow part over the few rows
// with more entries (rows next to the poles), which a second launch adds on top.
struct EllSplitDev {
    int W = 0, n_ovf = 0, W_ovf = 0;
    int32_t *idx = nullptr, *ovf_rows = nullptr, *ovf_idx = nullptr;
    float *coef = nullptr, *ovf_coef = nullptr;
};
EllSplitDev upload_ell_split(const icn::Ell& e, int W_main) {
    EllSplitDev d;
    const int W = e.width;
    d.W = W_main;
    d.W_ovf = std::max(W - W_main, 0);
    std::vector<int32_t> idx((size_t)e.rows * W_main, icn::IDX_ZERO), ovf_rows, ovf_idx;
    std::vector<float> coef((size_t)e.rows * W_main, 0.f), ovf_coef;
    for (int r = 0; r < e.rows; ++r) {
        for (int k = 0; k < std::min(W, W_main); ++k) {
            idx[(size_t)r * W_main + k] = e.idx[(size_t)r * W + k];
            coef[(size_t)r * W_main + k] = e.coef[(size_t)r * W + k];
        }
        if (W > W_main && e.idx[(size_t)r * W + W_main] != icn::IDX_ZERO) {       // rows are left-packed
            ovf_rows.push_back(r);
            for (int k = W_main; k < W; ++k) {
                ovf_idx.push_back(e.idx[(size_t)r * W + k]);
                ovf_coef.push_back(e.coef[(size_t)r * W + k]);
            }
        }
    }
    d.n_ovf = (int)ovf_rows.size();
    d.idx = upload(idx);
    d.coef = upload(coef);
    d.ovf_rows = upload(ovf_rows);
    d.ovf_idx = upload(ovf_idx);
    d.ovf_coef = upload(ovf_coef);
    return d;
}

// device tables of the aggregated paths of the decoder-block head: dy -> g (icn_upconv_bwd) and z -> y (dense forward of
// icn_upconv_fwd) as split ELL matrices; iota = identity table [Pc] (the coarse-level GEMMs do not gather)
// Patch tables of the LDS-staged sparse passes (icn_launch.h: PatchTab): the outputs (a grid of H x W pixels, row-major ids)
// are tiled by ph x pw patches; prow [npatch][umax] lists the union of the source rows of a patch's outputs, plocal [H * W][width]
// gives each output's entries as positions in that list.  idx [H * W][width_in] (negative: none); only the first `width`
// entries of a row are used (the rest stays with the overflow pass).  umax is rounded up to a multiple of 32.
struct PatchHost { std::vector<int32_t> prow; std::vector<uint16_t> plocal; int npatch = 0, umax = 0, gridW = 0; };
bool build_patches(const std::vector<int32_t>& idx, int width_in, int width, int H, int W, int ph, int pw, PatchHost& out) {
    out = PatchHost{};
    if (H % ph != 0 || W % pw != 0 || (long)H * W * width_in != (long)idx.size()) return false;
    const int pr_n = H / ph, pc_n = W / pw;
    out.npatch = pr_n * pc_n;
    out.gridW = W;
    out.plocal.assign((size_t)H * W * width, 0xFFFFu);
    std::vector<std::vector<int32_t>> lists(out.npatch);
    for (int p = 0; p < out.npatch; ++p) {
        std::vector<int32_t>& u = lists[p];
        for (int i = 0; i < ph; ++i)
            for (int j = 0; j < pw; ++j) {
                const size_t r = (size_t)((p / pc_n) * ph + i) * W + (p % pc_n) * pw + j;
                for (int k = 0; k < width; ++k) {
                    const int32_t v = idx[r * width_in + k];
                    if (v >= 0) u.push_back(v);
                }
            }
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        out.umax = std::max(out.umax, (int)u.size());
    }
    out.umax = (out.umax + 31) / 32 * 32;
    if (out.umax == 0 || out.umax >= 0xFFFF) return false;
    out.prow.assign((size_t)out.npatch * out.umax, -1);
    for (int p = 0; p < out.npatch; ++p) {
        const std::vector<int32_t>& u = lists[p];
        std::copy(u.begin(), u.end(), out.prow.begin() + (size_t)p * out.umax);
        for (int i = 0; i < ph; ++i)
            for (int j = 0; j < pw; ++j) {
                const size_t r = (size_t)((p / pc_n) * ph + i) * W + (p % pc_n) * pw + j;
                for (int k = 0; k < width; ++k) {
                    const int32_t v = idx[r * width_in + k];
                    if (v >= 0) out.plocal[r * width + k] = (uint16_t)(std::lower_bound(u.begin(), u.end(), v) - u.begin());
                }
            }
    }
    return true;
}
struct PatchDev { int32_t* prow = nullptr; uint16_t* plocal = nullptr; int npatch = 0, umax = 0, gridW = 0; };
PatchDev upload_patches(const PatchHost& h) {
    PatchDev d;
    if (getenv("ICN_VERBOSE")) fprintf(stderr, "icn: patch table: %d patches, %d rows per list, grid width %d\n", h.npatch, h.umax, h.gridW);
    d.prow = upload(h.prow);
    d.plocal = upload(h.plocal);
    d.npatch = h.npatch; d.umax = h.umax; d.gridW = h.gridW;
    return d;
}

struct UpconvBwdDev {
    int Pc = 0, Pf = 0;
    PatchDev sc_patch, ga_patch;   // LDS-staged forms of the scatter (fine outputs) and of the per-pixel aggregate (coarse outputs)
    EllSplitDev gather, scatter;   // gather: the rows of the pixels the per-pixel kernel cannot take (main part unused, W = 0)
    int32_t* iota = nullptr;
    int32_t* px_srcs = nullptr;    // [Pc][20] fine rows of a coarse pixel's 7 aggregates (-1 padded; all -1: generic kernel)
    int32_t* px_cls = nullptr;     // [Pc] coefficient class of the pixel
    float* cls_coef = nullptr;     // [n_cls][20][8] coefficient of source k in tap t (7 used); class 0 = all zero
    int n_cls = 0;
};

std::mutex g_mu;
std::map<std::tuple<int, int, int>, UpconvDev> g_upconv;       // (device, r_in, mode)
std::map<std::tuple<int, int, int>, UpconvBwdDev> g_upconv_bwd;
std::map<std::tuple<int, int, int, int>, ConvTables> g_conv;   // (device, r, stride, mode)
std::map<std::tuple<int, int, int>, UpTables> g_up;            // (device, r, mode)

// Union rows per 16-pixel stage of k_wgrad7: 64 at stride 1 (ICN_W7_U=56 selects the 38 KB two-stage form for A/B
// runs), 112 at stride 2.  ONE definition for the table the kernel is launched with (make_conv_tables) and the table the tests and
// the host selfcheck look at (icn_table_wgrad7); a value launch_wgrad has no instantiation for is an error, not a silent fall
// back to the per-tap kernel.
static int wgrad7_union_rows(int stride) {
    if (stride != 1) return 112;
    static const int u1 = [] {
        const char* e = getenv("ICN_W7_U");
        return e ? atoi(e) : 64;
    }();
    if (u1 != 56 && u1 != 64)
        throw std::invalid_argument("icn: ICN_W7_U must be 56 or 64 (the instantiations of k_wgrad7 at stride 1), got " + std::to_string(u1));
    return u1;
}

ConvTables make_conv_tables(int r_in, int stride, int mode) {
    ConvTables t;
    t.n_in = 1 << r_in;
    t.n_out = t.n_in / stride;
    t.Pin = icn::pixels(r_in);
    t.Pout = 10 * t.n_out * t.n_out;
    std::vector<int32_t> fwd, bwd, perm;
    std::vector<uint8_t> mask;
    icn::build_conv_fwd(r_in, stride, mode, fwd);
    t.E = icn::build_conv_bwd(r_in, stride, mode, bwd);
    t.fwd = upload(fwd);
    t.bwd = upload(bwd);
    t.d_fwd = upload_dma(fwd, 1, t.Pout);
    t.d_bwd = upload_dma(bwd, t.E, t.Pin);
    {
        icn::DmaTable h;
        icn::build_dma_table(fwd, 1, t.Pout, h);
        icn::Wg7Table w7;
        if (icn::build_wgrad7(h, t.Pout, wgrad7_union_rows(stride), w7)) {
            t.w7_rows = upload(w7.urow);
            t.w7_pos = upload(w7.upos);
            t.w7_U = w7.U;
        }
    }
    std::vector<int32_t> primary;
    icn::VirtualRows vr;
    icn::split_conv_bwd(r_in, stride, bwd, t.E, primary, vr);
    t.bwd1 = upload(primary);
    t.d_bwd1 = upload_dma(primary, 1, t.Pin);
    t.nv = vr.nv;
    if (vr.nv > 0) {
        std::vector<int32_t> order, vidx_o;
        std::vector<uint8_t> vmask;
        icn::build_virtual_order(vr, t.nvp, order, vidx_o, vmask);
        t.vidx = upload(vidx_o);
        t.vorder = upload(order);
        t.vmask32 = upload(std::vector<uint32_t>(vmask.begin(), vmask.end()));
        t.d_virt = upload_dma(vidx_o, 1, t.nvp);
        t.vq = upload(vr.vq);
    }
    if (stride == 2) {
        icn::build_bwd_row_order(r_in, stride, bwd, t.E, perm, mask);
        t.perm = upload(perm);
        t.mask32_h.assign(mask.begin(), mask.end());
        t.mask32 = upload(t.mask32_h);
        t.key = 1 + ((r_in << 8) | (stride << 4) | mode);
        // the same table in permuted row order (row k of a sample = pixel perm[k]): one load level in the kernels
        std::vector<int32_t> bwd_p(bwd.size());
        for (int te = 0; te < icn::NTAPS * t.E; ++te)
            for (int k = 0; k < t.Pin; ++k) bwd_p[(size_t)te * t.Pin + k] = bwd[(size_t)te * t.Pin + perm[k]];
        t.bwd_perm = upload(bwd_p);
        t.d_bwdp = upload_dma(bwd_p, t.E, t.Pin);
    }
    return t;
}
const ConvTables& conv_tables(int r_in, int stride, int mode) {
    int dev = 0;
    ICN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, r_in, stride, mode);
    auto it = g_conv.find(key);
    if (it != g_conv.end()) return it->second;
    return g_conv.emplace(key, make_conv_tables(r_in, stride, mode)).first->second;
}

std::map<std::pair<int, int>, int32_t*> g_vf;                  // (device, r) -> incident-face table of the loss

const int32_t* vertex_faces(int r) {
    int dev = 0;
    ICN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, r);
    auto it = g_vf.find(key);
    if (it != g_vf.end()) return it->second;
    std::vector<int32_t> vf;
    icn::build_vertex_faces(r, vf);
    return g_vf.emplace(key, upload(vf)).first->second;
}

UpTables make_up_tables(int r_in, int mode) {
    icn::Ell f, b;
    icn::build_upsample(r_in, mode, f, b);
    UpTables t;
    t.Pc = b.rows;
    t.Pf = f.rows;
    t.Wf = f.width;
    t.Wb = b.width;
    t.idx_f = upload(f.idx);
    t.coef_f = upload(f.coef);
    t.idx_b = upload(b.idx);
    t.coef_b = upload(b.coef);
    return t;
}
const UpTables& up_tables(int r_in, int mode) {
    int dev = 0;
    ICN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, r_in, mode);
    auto it = g_up.find(key);
    if (it != g_up.end()) return it->second;
    return g_up.emplace(key, make_up_tables(r_in, mode)).first->second;
}

UpconvDev make_upconv_tables(int r_in, int mode) {
    icn::UpconvTable h;
    icn::build_upconv_fwd(r_in, mode, h);
    UpconvDev d;
    d.Pc = h.Pc; d.Pf = h.Pf; d.n_slots = h.n_slots; d.E = h.E; d.nseg = h.nseg;
    for (int i = 0; i < h.nseg; ++i) { d.seg_cnt[i] = h.seg_cnt[i]; d.seg_off[i] = h.seg_off[i]; d.seg_mask[i] = h.seg_mask[i]; }
    d.pix = upload(h.pix);
    d.code = upload(h.code);
    d.slot_idx = upload(h.slot_idx);
    d.alpha = upload(h.alpha);
    d.slot_coef = upload(h.slot_coef);
    return d;
}
const UpconvDev& upconv_tables(int r_in, int mode) {
    int dev = 0;
    ICN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, r_in, mode);
    auto it = g_upconv.find(key);
    if (it != g_upconv.end()) return it->second;
    return g_upconv.emplace(key, make_upconv_tables(r_in, mode)).first->second;
}

UpconvBwdDev make_upconv_bwd_tables(int r_in, int mode) {
    icn::Ell e, f;
    icn::build_upconv_bwd(r_in, mode, e);
    icn::build_upconv_scatter(r_in, mode, f);
    UpconvBwdDev d;
    d.Pc = icn::pixels(r_in);
    d.Pf = 4 * d.Pc;
    {   // per-pixel form of dy -> g: the 7 rows of a coarse pixel share <= 20 fine source rows (19 away from the singular
        // vertices); a pixel with more keeps the generic row-by-row form (all 7 rows, full width, as "overflow" rows)
        constexpr int NS = 20;
        std::vector<int32_t> srcs((size_t)d.Pc * NS, icn::IDX_ZERO);
        std::vector<float> coefd((size_t)d.Pc * NS * 8, 0.f);
        icn::Ell rest;
        rest.width = e.width;
        std::vector<int32_t> rest_rows;
        for (int sp = 0; sp < d.Pc; ++sp) {
            std::vector<int32_t> uniq;
            for (int t = 0; t < 7; ++t)
                for (int k = 0; k < e.width; ++k) {
                    const int32_t p = e.idx[((size_t)sp * 7 + t) * e.width + k];
                    if (p >= 0 && std::find(uniq.begin(), uniq.end(), p) == uniq.end()) uniq.push_back(p);
                }
            if ((int)uniq.size() <= NS) {
                std::sort(uniq.begin(), uniq.end());
                for (size_t k = 0; k < uniq.size(); ++k) srcs[(size_t)sp * NS + k] = uniq[k];
                for (int t = 0; t < 7; ++t)
                    for (int k = 0; k < e.width; ++k) {
                        const int32_t p = e.idx[((size_t)sp * 7 + t) * e.width + k];
                        if (p < 0) continue;
                        const size_t slot = std::lower_bound(uniq.begin(), uniq.end(), p) - uniq.begin();
                        coefd[((size_t)sp * NS + slot) * 8 + t] += e.coef[((size_t)sp * 7 + t) * e.width + k];
                    }
            } else {
                for (int t = 0; t < 7; ++t) {
                    rest_rows.push_back(sp * 7 + t);
                    rest.idx.insert(rest.idx.end(), e.idx.begin() + ((size_t)sp * 7 + t) * e.width, e.idx.begin() + ((size_t)sp * 7 + t + 1) * e.width);
                    rest.coef.insert(rest.coef.end(), e.coef.begin() + ((size_t)sp * 7 + t) * e.width, e.coef.begin() + ((size_t)sp * 7 + t + 1) * e.width);
                }
            }
        }
        rest.rows = (int)rest_rows.size();
        // the coefficient blocks fall into a handful of classes (the sorted order of a pixel's sources follows the chart
        // layout): a class id per pixel and one table; pixels beyond the kernel's class capacity go the generic way too
        std::vector<int32_t> cls(d.Pc, 0);
        std::vector<float> cls_coef(NS * 8, 0.f);                 // class 0: no sources
        std::map<std::vector<float>, int> seen;
        seen.emplace(std::vector<float>(NS * 8, 0.f), 0);
        for (int sp = 0; sp < d.Pc; ++sp) {
            std::vector<float> blk(coefd.begin() + (size_t)sp * NS * 8, coefd.begin() + (size_t)(sp + 1) * NS * 8);
            auto found = seen.find(blk);
            if (found == seen.end()) {
                if ((int)seen.size() >= icn::UPCONV_PX_CLASSES) {  // does not happen on the icosahedral charts (19 classes)
                    for (int k = 0; k < NS; ++k) srcs[(size_t)sp * NS + k] = icn::IDX_ZERO;
                    for (int t = 0; t < 7; ++t) {
                        rest_rows.push_back(sp * 7 + t);
                        rest.idx.insert(rest.idx.end(), e.idx.begin() + ((size_t)sp * 7 + t) * e.width, e.idx.begin() + ((size_t)sp * 7 + t + 1) * e.width);
                        rest.coef.insert(rest.coef.end(), e.coef.begin() + ((size_t)sp * 7 + t) * e.width, e.coef.begin() + ((size_t)sp * 7 + t + 1) * e.width);
                    }
                    continue;
                }
                found = seen.emplace(blk, (int)seen.size()).first;
                cls_coef.insert(cls_coef.end(), blk.begin(), blk.end());
            }
            cls[sp] = found->second;
        }
        rest.rows = (int)rest_rows.size();
        d.n_cls = (int)seen.size();
        d.px_srcs = upload(srcs);
        {   // coarse pixel grid (5n x 2n) in patches of 4 x 8
            const int n = 1 << r_in;
            PatchHost ph;
            if (build_patches(srcs, NS, NS, 5 * n, 2 * n, 4, 8, ph)) d.ga_patch = upload_patches(ph);
        }
        d.px_cls = upload(cls);
        d.cls_coef = upload(cls_coef);
        d.gather = EllSplitDev{};
        d.gather.n_ovf = rest.rows;
        d.gather.W_ovf = rest.width;
        d.gather.ovf_rows = upload(rest_rows);
        d.gather.ovf_idx = upload(rest.idx);
        d.gather.ovf_coef = upload(rest.coef);
    }
    d.scatter = upload_ell_split(f, 16);           // 12-13 entries per fine pixel
    {   // fine pixel grid (10n x 4n) in patches of 8 x 16; the main part's 16 entries per row
        const int n = 1 << r_in;
        PatchHost ph;
        if (f.width >= 1 && build_patches(f.idx, f.width, std::min(f.width, 16), 10 * n, 4 * n, 8, 16, ph) && std::min(f.width, 16) == 16)
            d.sc_patch = upload_patches(ph);
    }
    std::vector<int32_t> iota(d.Pc);
    for (int i = 0; i < d.Pc; ++i) iota[i] = i;
    d.iota = upload(iota);
    return d;
}
const UpconvBwdDev& upconv_bwd_tables(int r_in, int mode) {
    int dev = 0;
    ICN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, r_in, mode);
    auto it = g_upconv_bwd.find(key);
    if (it != g_upconv_bwd.end()) return it->second;
    return g_upconv_bwd.emplace(key, make_upconv_bwd_tables(r_in, mode)).first->second;
}

// slots of the composite table per level (host only, cached; the larger of the two corner modes)
int upconv_slots(int r_in) {
    static std::mutex mu;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(r_in);
    if (it != cache.end()) return it->second;
    int n = 0;
    for (int mode = 0; mode < 2; ++mode) {
        icn::UpconvTable h;
        icn::build_upconv_fwd(r_in, mode, h);
        n = std::max(n, h.n_slots);
    }
    return cache[r_in] = n;
}

// ---- profiling state (off by default) ------------------------------------------------------------------
struct ProfRec { int kind; double flops; hipEvent_t e0, e1; };
std::vector<ProfRec> g_prof;          // pre-created events, reused by every start / stop
size_t g_prof_used = 0;
bool g_prof_on = false, g_prof_open = false;
int g_prof_only = -1;                 // >= 0: only launches of this kind are timed (icn_profile_select)
std::mutex g_prof_mu;                 // held from prof_mark_begin to prof_mark_end: launches may come from the forward
                                      // thread and from the autograd thread

void check_conv(const void* a, const void* b, const void* c, int B, int Cin, int Cout, int r_in, int stride) {
    if (!a || !b || !c) throw std::invalid_argument("icn: null tensor pointer");
    if (B < 1 || Cin < 1 || Cout < 1) throw std::invalid_argument("icn: B, Cin, Cout must be positive");
    if (r_in < 0 || r_in > 10) throw std::invalid_argument("icn: subdivisions out of range [0,10]");
    if (stride != 1 && stride != 2) throw std::invalid_argument("icn: stride must be 1 or 2");
    if ((size_t)B * icn::pixels(r_in) >= (size_t)1 << 31) throw std::invalid_argument("icn: B * pixels exceeds int32 rows");
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Table sizes the workspace query needs, per (r, stride): virtual rows of the stride-1 transposed gather and side-buffer
// slots of each DmaTable, the larger of the two corner modes.  Host-only, cached.
struct TableCounts { int nv = 0, slots_fwd = 0, slots_bwd = 0; };   // nv: padded (nvp)
TableCounts table_counts(int r_in, int stride) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, TableCounts> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(r_in, stride);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    TableCounts c;
    const int Pin = icn::pixels(r_in), Pout = Pin / (stride * stride);
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<int32_t> fwd, bwd, primary;
        icn::DmaTable d;
        icn::build_conv_fwd(r_in, stride, mode, fwd);
        icn::build_dma_table(fwd, 1, Pout, d);
        c.slots_fwd = std::max(c.slots_fwd, d.n_slots);
        const int E = icn::build_conv_bwd(r_in, stride, mode, bwd);
        icn::build_dma_table(bwd, E, Pin, d);              // (a row permutation does not change the slots)
        c.slots_bwd = std::max(c.slots_bwd, d.n_slots);
        if (stride == 1) {
            icn::VirtualRows vr;
            icn::split_conv_bwd(r_in, 1, bwd, E, primary, vr);
            c.nv = std::max(c.nv, (vr.nv + 31) / 32 * 32);   // rows of the padded virtual-row GEMM
            if (vr.nv > 0) {
                icn::build_dma_table(vr.vidx, 1, vr.nv, d);
                c.slots_bwd = std::max(c.slots_bwd, d.n_slots);
            }
        }
    }
    return cache[key] = c;
}

// workspace layout of bwd-weight: [wgrad partial slabs S x 7 x Cin x Cout][bias partials S x Cout][side buffer]
// (Cout = total output channels, Cout0 of them in the first tensor of a pair)
size_t wgrad_partial_bytes(int M, int Cin, int Cout, int Cout0) {
    return align256((size_t)icn::wgrad_splits(M, Cin, Cout, Cout0) * 7 * Cin * Cout * sizeof(float));
}
size_t wgrad_bias_partial_bytes(int M, int Cin, int Cout, int Cout0) {
    return align256((size_t)icn::wgrad_splits(M, Cin, Cout, Cout0) * Cout * sizeof(float));
}

bool args_ok(int B, int Cin, int Cout, int r_in, int stride) {
    return B >= 1 && Cin >= 1 && Cout >= 1 && r_in >= 0 && r_in <= 10 && (stride == 1 || stride == 2);
}

// Two convolutions of the same input (C0, C1 output channels) can run as one launch per pass when every GEMM of the
// three passes fits the LDS-DMA kernels (icn_launch.h: GatherGemmArgs / WgradArgs pair forms).
bool pair_supported(int B, int Cin, int C0, int C1, int r_in, int stride) {
    if (!args_ok(B, Cin, C0, r_in, stride) || C0 != C1 || Cin % 64 != 0 || C0 % 64 != 0) return false;
    const size_t Pin = icn::pixels(r_in), Pout = Pin / (stride * stride), lim = (size_t)1 << 31;
    if ((size_t)B * Pin >= lim) return false;
    const TableCounts tc = table_counts(r_in, stride);
    const size_t slots = std::max(tc.slots_fwd, tc.slots_bwd);
    if ((size_t)B * Pin * Cin * 4 >= lim || (size_t)B * Pout * C0 * 4 >= lim || (size_t)7 * (C0 + C1) * Cin * 4 >= lim) return false;
    if ((size_t)B * slots * std::max(Cin, C0) * 4 >= lim / 2) return false;
    return icn::wgrad_pair_supported((int)(B * Pout), (int)Pin, (int)Pout, Cin, C0, C1);
}

// Workspace layouts (each piece 256-byte aligned), C = C0 + C1 (C1 = 0: single convolution):
//   fwd        [packed weights 7 x Cin x C][pair: concatenated bias C][side buffer (B, slots_fwd, Cin)][stream-K scratch]
//   bwd-data   [packed weights][stride 1: virtual-row GEMM result (B, nvp, Cin)][side buffer(s) (B, slots_bwd, C0) per dy][stream-K]
//   stream-K scratch = [flag words CONV_SK_FLAGS][partial-tile slots]: the persistent GEMM's last round cut in K (k_conv_dma_sk)
//   bwd-weight [partial slabs S x 7 x Cin x C][bias partials S x C][side buffer (B, slots_fwd, Cin)]
size_t sk_ws_bytes() { return align256((size_t)icn::CONV_SK_FLAGS * sizeof(int)) + align256(icn::conv_sk_part_bytes()); }
// Packed-weight region of a GEMM call: [fp32 B operand, n weights][bf16 x 3 image of the same operand, 6 bytes per weight].  Both are
// always reserved (a workspace size must not depend on the arithmetic mode a later call runs under); a call fills what it reads.
size_t wpack_f32_bytes(size_t n) { return align256(n * sizeof(float)); }
size_t wpack_bytes(size_t n) { return wpack_f32_bytes(n) + align256(n * 6); }
// Decide the arithmetic of one GEMM launch and point the prologue and the launch at the matching weight image (a: complete but for wt)
void choose_arith(icn::GatherGemmArgs& a, icn::PrologueArgs& p, float* wreg, size_t n_weights, bool also_f32) {
    const int bn = icn::conv_b3_bn(a);
    a.arith_bn = bn;
    p.packed = (bn == 0 || also_f32) ? wreg : nullptr;
    p.packed_b3 = bn ? reinterpret_cast<char*>(wreg) + wpack_f32_bytes(n_weights) : nullptr;
    p.b3_bn = bn;
    p.b3_flat_n = (a.T == 1 && p.transpose == 0) ? 1 : 0;
    a.wt = bn ? static_cast<const float*>(p.packed_b3) : wreg;
}
size_t conv_ws_bytes(int op, int B, int Cin, int C0, int C1, int r_in, int stride) {
    const int C = C0 + C1, n_out = (1 << r_in) / stride, M = B * 10 * n_out * n_out;
    const size_t wbytes = wpack_bytes((size_t)7 * Cin * C);
    switch (op) {
        case ICN_OP_CONV_FWD:
            if (!icn::gather_gemm_supported(Cin, C)) return 0;
            return wbytes + (C1 ? align256((size_t)C * sizeof(float)) : 0) +
                   align256((size_t)B * table_counts(r_in, stride).slots_fwd * Cin * sizeof(float)) + sk_ws_bytes();
        case ICN_OP_CONV_BWD_DATA: {
            if (!icn::gather_gemm_supported(C, Cin)) return 0;
            const TableCounts c = table_counts(r_in, stride);
            return wbytes + align256((size_t)B * c.nv * Cin * sizeof(float)) +
                   (C1 ? 2 : 1) * align256((size_t)B * c.slots_bwd * C0 * sizeof(float)) + sk_ws_bytes();
        }
        case ICN_OP_CONV_BWD_WEIGHT:
            return wgrad_partial_bytes(M, Cin, C, C0) + wgrad_bias_partial_bytes(M, Cin, C, C0) +
                   align256((size_t)B * table_counts(r_in, stride).slots_fwd * Cin * sizeof(float));
        default: return 0;
    }
}

char* at(void* ws, size_t off) { return static_cast<char*>(ws) + off; }

// ---- the three passes; w1 == nullptr: one convolution, else a pair sharing x (C1 = its output channels) -------------
void conv_fwd_impl(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, float* y0, float* y1, int B,
                   int Cin, int C0, int C1, int r_in, int stride, const ConvTables& t, void* ws, hipStream_t s) {
    const int C = C0 + C1;
    const size_t wbytes = wpack_bytes((size_t)7 * Cin * C);
    float* wf = static_cast<float*>(ws);
    float* bias_cat = (w1 && b0) ? reinterpret_cast<float*>(at(ws, wbytes)) : nullptr;
    const size_t side_off = wbytes + (w1 ? align256((size_t)C * sizeof(float)) : 0);
    float* side = reinterpret_cast<float*>(at(ws, side_off));
    int* sk_flag = reinterpret_cast<int*>(at(ws, side_off + align256((size_t)B * table_counts(r_in, stride).slots_fwd * Cin * sizeof(float))));
    float* sk_part = reinterpret_cast<float*>(reinterpret_cast<char*>(sk_flag) + align256((size_t)icn::CONV_SK_FLAGS * sizeof(int)));
    icn::PrologueArgs p{};
    p.zero = sk_flag; p.n_zero = icn::CONV_SK_FLAGS;
    p.w = w0; p.w2 = w1; p.packed = wf; p.Cout = C0; p.Cout2 = C1; p.Cin = Cin; p.transpose = 0;
    p.bias = b0; p.bias2 = b1; p.bias_cat = bias_cat;
    p.src = x; p.slots = t.d_fwd.slots; p.side = side; p.n_slots = t.d_fwd.n_slots; p.E = 1; p.B = B; p.Ps = t.Pin; p.K = Cin;
    p.ns = t.n_in;
    icn::GatherGemmArgs a{};
    a.src = x; a.bias = w1 ? bias_cat : b0; a.dst = y0; a.dst2 = w1 ? y1 : nullptr; a.N0 = C0;
    a.idx = t.fwd; a.dcode = t.d_fwd.code; a.side = side; a.n_slots = t.d_fwd.n_slots;
    a.M = B * t.Pout; a.Ps = t.Pin; a.Pd = t.Pout; a.K = Cin; a.N = C; a.E = 1; a.ns = t.n_in;
    a.algo_flops = 2.0 * 7 * Cin * C * (double)B * t.Pout;
    a.sk_part = sk_part; a.sk_flag = sk_flag;
    choose_arith(a, p, wf, (size_t)7 * Cin * C, false);
    icn::launch_conv_prologue(p, s);
    icn::launch_gather_gemm_auto(a, s);
}

void conv_bwd_data_impl(const float* dy0, const float* dy1, const float* w0, const float* w1, float* dx, int B, int Cin, int C0,
                        int C1, int r_in, int stride, const ConvTables& t, void* ws, hipStream_t s) {
    const int C = C0 + C1;
    const size_t wbytes = wpack_bytes((size_t)7 * Cin * C);
    const TableCounts tc = table_counts(r_in, stride);
    const size_t side_bytes = align256((size_t)B * tc.slots_bwd * C0 * sizeof(float));
    float* wb = static_cast<float*>(ws);
    float* vout = reinterpret_cast<float*>(at(ws, wbytes));
    float* side = reinterpret_cast<float*>(at(ws, wbytes + align256((size_t)B * tc.nv * Cin * sizeof(float))));
    float* side2 = dy1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(side) + side_bytes) : nullptr;
    int* sk_flag = reinterpret_cast<int*>(reinterpret_cast<char*>(side) + (dy1 ? 2 : 1) * side_bytes);
    float* sk_part = reinterpret_cast<float*>(reinterpret_cast<char*>(sk_flag) + align256((size_t)icn::CONV_SK_FLAGS * sizeof(int)));
    // source = dy at the output level (pole corners of THAT level), rows = input pixels
    // Stride 1: the transposed gather has extra entries (duplicates / pole means) along the chart seams.
    // Where they are few (fine levels) the main GEMM gathers only the primary entries and a second, small GEMM
    // over "virtual rows" adds the rest; where they are many (coarse levels: 23 % of the rows at r = 3) they go
    // through the main GEMM's side buffer instead of a second launch.
    const bool split = stride == 1 && t.nv > 0 && t.nv * 8 < t.Pin;
    const DevDma& dm = split ? t.d_bwd1 : (stride == 2 ? t.d_bwdp : t.d_bwd);   // main GEMM's table
    const DevDma& ds = split ? t.d_virt : dm;                                    // table the side buffer serves
    icn::PrologueArgs p{};
    p.w = w0; p.w2 = w1; p.packed = wb; p.Cout = C0; p.Cout2 = C1; p.Cin = Cin; p.transpose = 1;
    p.src = dy0; p.src2 = dy1; p.slots = ds.slots; p.side = side; p.side2 = side2; p.n_slots = ds.n_slots; p.E = ds.E; p.B = B;
    p.Ps = t.Pout; p.K = C0; p.ns = t.n_out;
    p.zero = sk_flag; p.n_zero = icn::CONV_SK_FLAGS;
    icn::GatherGemmArgs a{};
    a.src = dy0; a.src2 = dy1; a.dst = dx; a.N0 = Cin;
    a.idx = split ? t.bwd1 : (stride == 2 ? t.bwd_perm : t.bwd); a.dcode = dm.code; a.side = side; a.side2 = side2;
    a.n_slots = dm.n_slots; a.perm = t.perm; a.mask32 = t.mask32;
    a.mask32_host = t.mask32_h.empty() ? nullptr : t.mask32_h.data(); a.mask_key = t.key;
    a.M = B * t.Pin; a.Ps = t.Pout; a.Pd = t.Pin; a.K = C; a.N = Cin; a.E = split ? 1 : t.E; a.ns = t.n_out;
    a.algo_flops = 2.0 * 7 * Cin * C * (double)B * t.Pout;
    a.sk_part = sk_part; a.sk_flag = sk_flag;             // (only the main launch: the flags are cleared once per call)
    choose_arith(a, p, wb, (size_t)7 * Cin * C, split);   // (the virtual-row GEMM below stays on the fp32 operand)
    icn::launch_conv_prologue(p, s);
    icn::launch_gather_gemm_auto(a, s);
    if (split) {
        // second, small GEMM over the virtual rows, then dx[b, vq[v], :] += result[b, v, :]
        icn::GatherGemmArgs v = a;
        v.wt = wb; v.arith_bn = 0;
        v.sk_part = nullptr; v.sk_flag = nullptr;
        v.dst = vout; v.idx = t.vidx; v.dcode = t.d_virt.code; v.n_slots = t.d_virt.n_slots; v.perm = t.vorder; v.mask32 = t.vmask32;
        v.M = B * t.nvp; v.Pd = t.nvp; v.E = 1;
        v.algo_flops = 0.0;   // the main launch above already carries the layer's algorithmic FLOPs; this one adds time only
        icn::launch_gather_gemm_auto(v, s);
        icn::launch_row_scatter_add(vout, dx, t.vq, B, t.nv, t.nvp, t.Pin, Cin, s);
    }
}

void conv_bwd_weight_impl(const float* x, const float* dy0, const float* dy1, float* dw0, float* db0, float* dw1, float* db1, int B,
                          int Cin, int C0, int C1, const ConvTables& t, void* ws, hipStream_t s) {
    const int C = C0 + C1, M = B * t.Pout;
    float* partial = static_cast<float*>(ws);
    float* bpart = (db0 || db1) ? reinterpret_cast<float*>(at(ws, wgrad_partial_bytes(M, Cin, C, C0))) : nullptr;
    float* side = reinterpret_cast<float*>(at(ws, wgrad_partial_bytes(M, Cin, C, C0) + wgrad_bias_partial_bytes(M, Cin, C, C0)));
    const bool mfma = icn::wgrad_supported(Cin, C);
    if (mfma) {
        icn::PrologueArgs p{};
        p.Cin = Cin; p.src = x; p.slots = t.d_fwd.slots; p.side = side; p.n_slots = t.d_fwd.n_slots; p.E = 1; p.B = B; p.Ps = t.Pin;
        p.K = Cin; p.ns = t.n_in;
        icn::launch_conv_prologue(p, s);
    }
    icn::WgradArgs a{};
    a.x = x; a.dy = dy0; a.dy2 = dy1; a.Cout0 = C0; a.idx = t.fwd; a.dcode = mfma ? t.d_fwd.code : nullptr; a.side = side;
    a.n_slots = t.d_fwd.n_slots; a.partial = partial; a.bias_partial = bpart; a.dw = dw0; a.dbias = db0; a.dw2 = dw1; a.dbias2 = db1;
    a.M = M; a.Ps = t.Pin; a.Pd = t.Pout; a.Cin = Cin; a.Cout = C; a.ns = t.n_in;
    a.algo_flops = 2.0 * 7 * Cin * C * (double)M;
    a.w7_rows = t.w7_rows; a.w7_pos = t.w7_pos; a.w7_U = t.w7_U;
    icn::launch_wgrad(a, s);
}

}  // namespace

namespace icn {
// names as rocprofv3 prints them (modulo "void icn::"), so bench.py's live numbers line up with profiles/
const char* const PROF_NAMES[PROF_KINDS] = {"k_conv_dma<128, 128, false>", "k_conv_dma<128, 64, false>", "k_conv_dma<64, 128, false>",
                                            "k_conv_dma<64, 64, false>",
                                            "k_gather_gemm<128, 128>", "k_gather_gemm<128, 64>", "k_gather_gemm<64, 128>",
                                            "k_gather_gemm<64, 64>", "k_wgrad_dma<128, 128>", "k_wgrad_dma<128, 64>",
                                            "k_wgrad_dma<64, 128>", "k_wgrad_dma<64, 64>", "k_wgrad<128, 128>", "k_wgrad<128, 64>",
                                            "k_wgrad<64, 128>", "k_wgrad<64, 64>", "k_conv_dma<128, 128, true>",
                                            "k_conv_dma<128, 64, true>", "k_conv_dma<64, 128, true>", "k_conv_dma<64, 64, true>",
                                            "k_conv_dma_sk<64, 128, false>", "k_conv_dma_sk<64, 64, false>", "k_conv_dma_sk<64, 128, true>",
                                            "k_conv_dma_sk<64, 64, true>", "k_wgrad7",
                                            "k_conv_dma8<64, 128, false>", "k_conv_dma8<64, 128, true>", "k_conv_dma_sk8<64, 128, false>",
                                            "k_conv_dma_sk8<64, 128, true>", "k_conv_dense_sk<64, 128>", "k_conv_dense_sk<64, 64>", "k_conv_single_sk<64, 128>",
                                            "k_conv_single_sk<64, 64>", "k_wgrad_dense<128, 128>", "k_wgrad_dense<128, 64>",
                                            "k_wgrad_dense<64, 128>", "k_wgrad_dense<64, 64>",
                                            "k_conv_b3_sk<128, 128, 8>", "k_conv_b3_sk<128, 64, 4>", "k_conv_b3_dense_sk<128, 128, 8>",
                                            "k_conv_b3_dense_sk<128, 64, 4>", "k_wgrad7<.., 1>", "k_wgrad_dma<.., 1>", "k_wgrad_dense<.., 1>", "k_conv_b3<128, BN, NW>",
                                            "k_conv_b3_single_sk<128, 128, 8>", "k_conv_b3_single_sk<128, 64, 4>"};
void prof_mark_begin(int kind, double flops, hipStream_t s) {
    if (!g_prof_on) return;
    g_prof_mu.lock();
    if (!g_prof_on || g_prof_used >= g_prof.size() || (g_prof_only >= 0 && kind != g_prof_only)) { g_prof_mu.unlock(); return; }
    ProfRec& r = g_prof[g_prof_used];
    r.kind = kind;
    r.flops = flops;
    (void)hipEventRecord(r.e0, s);
    g_prof_open = true;
}
void prof_mark_end(hipStream_t s) {
    if (!g_prof_open) return;         // only true between a successful begin and this call, on the thread holding the mutex
    (void)hipEventRecord(g_prof[g_prof_used].e1, s);
    ++g_prof_used;
    g_prof_open = false;
    g_prof_mu.unlock();
}
}  // namespace icn

extern "C" {

int icn_profile_start(int max_launches) {
    try {
        if (max_launches < 1) throw std::invalid_argument("icn_profile_start: max_launches must be positive");
        std::lock_guard<std::mutex> lk(g_prof_mu);
        while ((int)g_prof.size() < max_launches) {
            ProfRec r{};
            ICN_HIP(hipEventCreate(&r.e0));
            ICN_HIP(hipEventCreate(&r.e1));
            g_prof.push_back(r);
        }
        g_prof_used = 0;
        g_prof_on = true;
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_profile_select(const char* kernel) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_only = -1;
    if (kernel == nullptr || kernel[0] == 0) return 0;
    for (int k = 0; k < icn::PROF_KINDS; ++k)
        if (std::strcmp(kernel, icn::PROF_NAMES[k]) == 0) { g_prof_only = k; return 0; }
    return fail(std::string("icn_profile_select: unknown kernel ") + kernel);
}

int icn_profile_stop(icn_profile_entry* out, int cap) {
    try {
        {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_prof_on = false;
        }
        ICN_HIP(hipDeviceSynchronize());
        icn_profile_entry acc[icn::PROF_KINDS];
        for (int k = 0; k < icn::PROF_KINDS; ++k) acc[k] = icn_profile_entry{icn::PROF_NAMES[k], 0, 0.0, 0.0};
        for (size_t i = 0; i < g_prof_used; ++i) {
            float ms = 0.f;
            ICN_HIP(hipEventElapsedTime(&ms, g_prof[i].e0, g_prof[i].e1));
            acc[g_prof[i].kind].launches += 1;
            acc[g_prof[i].kind].total_ms += ms;
            acc[g_prof[i].kind].total_flops += g_prof[i].flops;
        }
        int n = 0;
        for (int k = 0; k < icn::PROF_KINDS && n < cap; ++k)
            if (acc[k].launches > 0) out[n++] = acc[k];
        return n;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}


int icn_abi_version(void) { return ICN_ABI_VERSION; }
const char* icn_last_error(void) { return g_err.c_str(); }

int icn_prepare_conv(int r_in, int stride, int corner_mode) {
    try {
        (void)conv_tables(r_in, stride, corner_mode);
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_prepare_upsample(int r_in, int corner_mode) {
    try {
        (void)up_tables(r_in, corner_mode);
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

size_t icn_conv_workspace_bytes(int op, int B, int Cin, int Cout, int r_in, int stride) {
    if (!args_ok(B, Cin, Cout, r_in, stride)) return 0;
    try {
        return conv_ws_bytes(op, B, Cin, Cout, 0, r_in, stride);
    } catch (const std::exception&) {
        return 0;
    }
}

int icn_conv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int r_in,
                 int stride, int corner_mode, void* ws, size_t ws_bytes, void* stream) {
    try {
        check_conv(x, w, y, B, Cin, Cout, r_in, stride);
        const ConvTables& t = conv_tables(r_in, stride, corner_mode);
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (icn::gather_gemm_supported(Cin, Cout)) {
            if (!ws || ws_bytes < conv_ws_bytes(ICN_OP_CONV_FWD, B, Cin, Cout, 0, r_in, stride))
                throw std::invalid_argument("icn_conv_fwd: workspace too small");
            conv_fwd_impl(x, w, bias, nullptr, nullptr, y, nullptr, B, Cin, Cout, 0, r_in, stride, t, ws, s);
        } else if (icn::stem_supported(Cin, Cout)) {
            icn::launch_stem_fwd(x, w, bias, y, t.fwd, B * t.Pout, t.Pin, t.Pout, Cin, Cout, t.n_in, s);
        } else {
            icn::launch_conv_generic(x, w, bias, y, t.fwd, B, t.Pin, t.Pout, Cin, Cout, 1, t.n_in, 0, s);
        }
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_conv_bwd_data(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int r_in, int stride,
                      int corner_mode, void* ws, size_t ws_bytes, void* stream) {
    try {
        check_conv(dy, w, dx, B, Cin, Cout, r_in, stride);
        const ConvTables& t = conv_tables(r_in, stride, corner_mode);
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (icn::gather_gemm_supported(Cout, Cin)) {
            if (!ws || ws_bytes < conv_ws_bytes(ICN_OP_CONV_BWD_DATA, B, Cin, Cout, 0, r_in, stride))
                throw std::invalid_argument("icn_conv_bwd_data: workspace too small");
            conv_bwd_data_impl(dy, nullptr, w, nullptr, dx, B, Cin, Cout, 0, r_in, stride, t, ws, s);
        } else {
            icn::launch_conv_generic(dy, w, nullptr, dx, t.bwd, B, t.Pout, t.Pin, Cout, Cin, t.E, t.n_out, 1, s);
        }
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_conv_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int r_in,
                        int stride, int corner_mode, void* ws, size_t ws_bytes, void* stream) {
    try {
        check_conv(x, dy, dw, B, Cin, Cout, r_in, stride);
        const ConvTables& t = conv_tables(r_in, stride, corner_mode);
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (!ws || ws_bytes < conv_ws_bytes(ICN_OP_CONV_BWD_WEIGHT, B, Cin, Cout, 0, r_in, stride))
            throw std::invalid_argument("icn_conv_bwd_weight: workspace too small");
        conv_bwd_weight_impl(x, dy, nullptr, dw, dbias, nullptr, nullptr, B, Cin, Cout, 0, t, ws, s);
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// ---- pair forms: two convolutions of the same input (reference models.py:37-39 conv00 / conv10, 59-60) in one launch per
// pass: forward with the output channels concatenated, bwd-data with the two dy concatenated along the GEMM's K axis
// (one dx, no separate gradient add), bwd-weight with the shared x gathered once.
int icn_conv_pair_supported(int B, int Cin, int Cout0, int Cout1, int r_in, int stride) {
    try {
        return pair_supported(B, Cin, Cout0, Cout1, r_in, stride) ? 1 : 0;
    } catch (const std::exception&) {
        return 0;
    }
}

size_t icn_conv_pair_workspace_bytes(int op, int B, int Cin, int Cout0, int Cout1, int r_in, int stride) {
    try {
        if (!pair_supported(B, Cin, Cout0, Cout1, r_in, stride)) return 0;
        return conv_ws_bytes(op, B, Cin, Cout0, Cout1, r_in, stride);
    } catch (const std::exception&) {
        return 0;
    }
}

int icn_conv_pair_fwd(const float* x, const float* w0, const float* bias0, const float* w1, const float* bias1, float* y0, float* y1,
                      int B, int Cin, int Cout0, int Cout1, int r_in, int stride, int corner_mode, void* ws, size_t ws_bytes,
                      void* stream) {
    try {
        check_conv(x, w0, y0, B, Cin, Cout0, r_in, stride);
        if (!w1 || !y1) throw std::invalid_argument("icn: null tensor pointer");
        if ((bias0 == nullptr) != (bias1 == nullptr)) throw std::invalid_argument("icn_conv_pair_fwd: both biases or none");
        if (!pair_supported(B, Cin, Cout0, Cout1, r_in, stride)) throw std::invalid_argument("icn_conv_pair_fwd: unsupported shape");
        if (!ws || ws_bytes < conv_ws_bytes(ICN_OP_CONV_FWD, B, Cin, Cout0, Cout1, r_in, stride))
            throw std::invalid_argument("icn_conv_pair_fwd: workspace too small");
        const ConvTables& t = conv_tables(r_in, stride, corner_mode);
        conv_fwd_impl(x, w0, bias0, w1, bias1, y0, y1, B, Cin, Cout0, Cout1, r_in, stride, t, ws, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_conv_pair_bwd_data(const float* dy0, const float* dy1, const float* w0, const float* w1, float* dx, int B, int Cin, int Cout0,
                           int Cout1, int r_in, int stride, int corner_mode, void* ws, size_t ws_bytes, void* stream) {
    try {
        check_conv(dy0, w0, dx, B, Cin, Cout0, r_in, stride);
        if (!dy1 || !w1) throw std::invalid_argument("icn: null tensor pointer");
        if (!pair_supported(B, Cin, Cout0, Cout1, r_in, stride)) throw std::invalid_argument("icn_conv_pair_bwd_data: unsupported shape");
        if (!ws || ws_bytes < conv_ws_bytes(ICN_OP_CONV_BWD_DATA, B, Cin, Cout0, Cout1, r_in, stride))
            throw std::invalid_argument("icn_conv_pair_bwd_data: workspace too small");
        const ConvTables& t = conv_tables(r_in, stride, corner_mode);
        conv_bwd_data_impl(dy0, dy1, w0, w1, dx, B, Cin, Cout0, Cout1, r_in, stride, t, ws, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_conv_pair_bwd_weight(const float* x, const float* dy0, const float* dy1, float* dw0, float* dbias0, float* dw1, float* dbias1,
                             int B, int Cin, int Cout0, int Cout1, int r_in, int stride, int corner_mode, void* ws, size_t ws_bytes,
                             void* stream) {
    try {
        check_conv(x, dy0, dw0, B, Cin, Cout0, r_in, stride);
        if (!dy1 || !dw1) throw std::invalid_argument("icn: null tensor pointer");
        if (!pair_supported(B, Cin, Cout0, Cout1, r_in, stride)) throw std::invalid_argument("icn_conv_pair_bwd_weight: unsupported shape");
        if (!ws || ws_bytes < conv_ws_bytes(ICN_OP_CONV_BWD_WEIGHT, B, Cin, Cout0, Cout1, r_in, stride))
            throw std::invalid_argument("icn_conv_pair_bwd_weight: workspace too small");
        const ConvTables& t = conv_tables(r_in, stride, corner_mode);
        conv_bwd_weight_impl(x, dy0, dy1, dw0, dbias0, dw1, dbias1, B, Cin, Cout0, Cout1, t, ws, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// ---- composite upsample + conv (forward): y_k = conv_k(upsample(x)), k = 0 (, 1), from the COARSE tensor ------------------
namespace {
bool upconv_supported(int B, int Cin, int C0, int C1, int r_in) {
    if (B < 1 || r_in < 0 || r_in > 9 || Cin < 32 || Cin % 32 != 0 || C0 < 64 || C0 % 64 != 0 || C1 < 0 || C1 % 64 != 0) return false;
    const size_t Pc = icn::pixels(r_in), Pf = 4 * Pc, lim = (size_t)1 << 31;
    const size_t C = (size_t)C0 + C1;
    if ((size_t)B * Pf + 5 * 8 * 128 >= lim) return false;          // rows incl. the padding of the 5 row segments
    if ((size_t)B * Pc * Cin * 4 >= lim || (size_t)icn::UPCONV_TAPS * C * Cin * 4 >= lim) return false;
    if ((size_t)B * Pf * std::max(C0, C1) * 4 >= ((size_t)1 << 34)) return false;       // dst rows are 32-bit, bytes are size_t
    if ((size_t)B * upconv_slots(r_in) * Cin * 4 >= lim / 2) return false;
    return true;
}
// Two forward methods.  COMPOSITE: one gather-GEMM over the coarse tensor with 19 virtual taps (0.68 of the multiply-adds).
// DENSE: z_t[s] = W_t x[s] for all taps as one dense coarse-level GEMM (N = 7 * C, a quarter of the multiply-adds) followed by
// the HBM-bound combination y[p] = bias + sum U[nbr_t(p), s] z_t[s].  ICN_UPCONV_FWD=composite|dense overrides the choice.
bool upconv_dense_ok(int B, int Cin, int C0, int C1, int r_in) {
    const size_t Pc = icn::pixels(r_in), C = (size_t)C0 + C1;
    return Cin >= 128 && (size_t)B * Pc * 7 * C * 4 < ((size_t)1 << 31) && (size_t)7 * C * Cin * 4 < ((size_t)1 << 31);
}
bool upconv_use_dense(int B, int Cin, int C0, int C1, int r_in) {
    if (!upconv_dense_ok(B, Cin, C0, C1, r_in)) return false;
    static const char* force = getenv("ICN_UPCONV_FWD");
    if (force && force[0] == 'c') return false;
    if (force && force[0] == 'd') return true;
    return true;
}
size_t upconv_ws_bytes(int B, int Cin, int C0, int C1, int r_in) {
    const size_t C = (size_t)C0 + C1;
    const size_t composite = align256((size_t)icn::UPCONV_TAPS * C * Cin * sizeof(float)) + align256(C * sizeof(float)) +
                             align256((size_t)B * upconv_slots(r_in) * Cin * sizeof(float));
    const size_t dense = wpack_bytes((size_t)7 * C * Cin) + align256(C * sizeof(float)) +
                         align256((size_t)B * icn::pixels(r_in) * 7 * C * sizeof(float)) + sk_ws_bytes();
    return upconv_dense_ok(B, Cin, C0, C1, r_in) ? std::max(composite, dense) : composite;
}
}  // namespace

int icn_upconv_supported(int B, int Cin, int Cout0, int Cout1, int r_in) {
    try {
        return upconv_supported(B, Cin, Cout0, Cout1, r_in) ? 1 : 0;
    } catch (const std::exception&) {
        return 0;
    }
}

size_t icn_upconv_workspace_bytes(int B, int Cin, int Cout0, int Cout1, int r_in) {
    try {
        return upconv_supported(B, Cin, Cout0, Cout1, r_in) ? upconv_ws_bytes(B, Cin, Cout0, Cout1, r_in) : 0;
    } catch (const std::exception&) {
        return 0;
    }
}

int icn_upconv_fwd(const float* x, const float* w0, const float* bias0, const float* w1, const float* bias1, float* y0, float* y1, int B,
                   int Cin, int Cout0, int Cout1, int r_in, int corner_mode, void* ws, size_t ws_bytes, void* stream) {
    try {
        if (!x || !w0 || !y0) throw std::invalid_argument("icn_upconv_fwd: null tensor pointer");
        if ((w1 == nullptr) != (Cout1 == 0) || (w1 && !y1)) throw std::invalid_argument("icn_upconv_fwd: second branch needs w1, y1 and Cout1 > 0");
        if (w1 && (bias0 == nullptr) != (bias1 == nullptr)) throw std::invalid_argument("icn_upconv_fwd: both biases or none");
        if (corner_mode != 0 && corner_mode != 1) throw std::invalid_argument("icn: corner_mode must be 0 (zeros) or 1 (average)");
        if (!upconv_supported(B, Cin, Cout0, Cout1, r_in)) throw std::invalid_argument("icn_upconv_fwd: unsupported shape");
        if (!ws || ws_bytes < upconv_ws_bytes(B, Cin, Cout0, Cout1, r_in)) throw std::invalid_argument("icn_upconv_fwd: workspace too small");
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int C = Cout0 + Cout1;
        if (upconv_use_dense(B, Cin, Cout0, Cout1, r_in)) {
            const UpconvBwdDev& d = upconv_bwd_tables(r_in, corner_mode);
            const size_t wb = wpack_bytes((size_t)7 * C * Cin), bb = align256((size_t)C * sizeof(float));
            float* wf = static_cast<float*>(ws);
            float* bias_cat = (w1 && bias0) ? reinterpret_cast<float*>(at(ws, wb)) : nullptr;
            float* z = reinterpret_cast<float*>(at(ws, wb + bb));
            int* sk_flag = reinterpret_cast<int*>(at(ws, wb + bb + align256((size_t)B * d.Pc * 7 * C * sizeof(float))));
            icn::PrologueArgs p{};                     // [7][C][Cin] = the B operand [1][N = 7 * C][K = Cin] of one dense GEMM
            p.zero = sk_flag; p.n_zero = icn::CONV_SK_FLAGS;
            p.w = w0; p.w2 = w1; p.packed = wf; p.Cout = Cout0; p.Cout2 = Cout1; p.Cin = Cin; p.transpose = 0;
            p.bias = bias0; p.bias2 = bias1; p.bias_cat = bias_cat;
            icn::GatherGemmArgs a{};
            a.src = x; a.dst = z; a.N0 = 7 * C; a.dcode = d.iota; a.perm = nullptr;   // identity rows: no destination-row table
            a.Ps = d.Pc; a.Pd = d.Pc; a.K = Cin; a.N = 7 * C; a.E = 1; a.T = 1; a.M = B * d.Pc;
            a.segs.nseg = 1; a.segs.B = B; a.segs.cnt[0] = d.Pc; a.segs.off[0] = 0; a.segs.mask[0] = 1u;
            a.algo_flops = 2.0 * 7 * Cin * C * (double)B * d.Pc;                  // executed: a quarter of the fine-level forward
            a.sk_flag = sk_flag;
            a.sk_part = reinterpret_cast<float*>(reinterpret_cast<char*>(sk_flag) + align256((size_t)icn::CONV_SK_FLAGS * sizeof(int)));
            choose_arith(a, p, wf, (size_t)7 * C * Cin, false);
            icn::launch_conv_prologue(p, s);
            icn::launch_gather_gemm_auto(a, s);
            const EllSplitDev& sc = d.scatter;
            const float* bias = w1 ? bias_cat : bias0;
            const icn::PatchTab pt{d.sc_patch.prow, d.sc_patch.plocal, d.sc_patch.npatch, d.sc_patch.umax, d.sc_patch.gridW};
            if (sc.W == 16 && icn::upconv_patch_usable(pt, B, (size_t)7 * d.Pc, Cout0, Cout1))
                icn::launch_upconv_scatter_lds(z, bias, y0, y1, pt, sc.coef, B, 7 * d.Pc, d.Pf, Cout0, Cout1, s);
            else
                icn::launch_upconv_scatter(z, bias, y0, y1, sc.idx, sc.coef, nullptr, B, 7 * d.Pc, d.Pf, d.Pf, Cout0, Cout1, sc.W, 0, s);
            icn::launch_upconv_scatter(z, nullptr, y0, y1, sc.ovf_idx, sc.ovf_coef, sc.ovf_rows, B, 7 * d.Pc, sc.n_ovf, d.Pf, Cout0, Cout1,
                                       sc.W_ovf, 1, s);
            ICN_HIP(hipGetLastError());
            return 0;
        }
        const UpconvDev& t = upconv_tables(r_in, corner_mode);
        const size_t wbytes = align256((size_t)icn::UPCONV_TAPS * C * Cin * sizeof(float));
        float* weff = static_cast<float*>(ws);
        float* bias_cat = (w1 && bias0) ? reinterpret_cast<float*>(at(ws, wbytes)) : nullptr;
        float* side = reinterpret_cast<float*>(at(ws, wbytes + align256((size_t)C * sizeof(float))));
        icn::UpconvPrologueArgs p{};
        p.w = w0; p.w2 = w1; p.packed = weff; p.Cout = Cout0; p.Cout2 = Cout1; p.Cin = Cin; p.NV = icn::UPCONV_TAPS; p.alpha = t.alpha;
        p.bias = bias0; p.bias2 = bias1; p.bias_cat = bias_cat;
        p.src = x; p.slot_idx = t.slot_idx; p.slot_coef = t.slot_coef; p.side = side; p.n_slots = t.n_slots; p.E = t.E; p.B = B; p.Ps = t.Pc;
        icn::launch_upconv_prologue(p, s);
        icn::GatherGemmArgs a{};
        a.src = x; a.wt = weff; a.bias = w1 ? bias_cat : bias0; a.dst = y0; a.dst2 = w1 ? y1 : nullptr; a.N0 = Cout0;
        a.dcode = t.code; a.side = side; a.n_slots = t.n_slots; a.perm = t.pix;
        a.Ps = t.Pc; a.Pd = t.Pf; a.K = Cin; a.N = C; a.E = 1; a.T = icn::UPCONV_TAPS;
        a.segs.nseg = t.nseg; a.segs.B = B;
        for (int i = 0; i < t.nseg; ++i) { a.segs.cnt[i] = t.seg_cnt[i]; a.segs.off[i] = t.seg_off[i]; a.segs.mask[i] = t.seg_mask[i]; }
        a.M = B * t.Pf;                                          // the launch pads the row segments to its tile height
        double taps = 0;                                         // executed: 7 (coarse-site rows) or 4 (midpoint rows) K-blocks per row,
        for (int i = 0; i < t.nseg; ++i) taps += (double)t.seg_cnt[i] * __builtin_popcount(t.seg_mask[i]);   // ~0.68 of the 7 per row
        a.algo_flops = 2.0 * Cin * C * (double)B * taps;         // of the two operators it replaces (2 * 7 * Cin * C * B * Pf)
        icn::launch_gather_gemm_auto(a, s);
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// ---- backward of the same pair through the coarse-level aggregate g (csrc/icn_geometry.h build_upconv_bwd) ------------------
namespace {
bool upconv_bwd_supported(int B, int Cin, int C0, int C1, int r_in, int corner_mode) {
    if (corner_mode != ICN_CORNER_AVERAGE) return false;                    // dbias = sum g_0 needs upsample rows that sum to one
    if (B < 1 || r_in < 0 || r_in > 9 || Cin < 64 || Cin % 64 != 0 || C0 < 64 || C0 % 64 != 0 || C1 < 0 || C1 % 64 != 0) return false;
    const size_t Pc = icn::pixels(r_in), Pf = 4 * Pc, lim = (size_t)1 << 31, C = (size_t)C0 + C1;
    if ((size_t)B * Pf >= lim || (size_t)B * Pc * 7 * C * 4 >= lim || (size_t)B * Pc * Cin * 4 >= lim) return false;
    if ((size_t)7 * C * Cin * 4 >= lim) return false;
    return icn::gather_gemm_supported((int)(7 * C), Cin) && icn::wgrad_supported(Cin, (int)C);
}
struct UpconvBwdWs { size_t g, wb, partial, bpart, sk, total; };
UpconvBwdWs upconv_bwd_ws(int B, int Cin, int C0, int C1, int r_in) {
    const int C = C0 + C1, M = B * icn::pixels(r_in);
    UpconvBwdWs w{};
    w.g = 0;
    w.wb = align256((size_t)M * 7 * C * sizeof(float));
    w.partial = w.wb + wpack_bytes((size_t)7 * C * Cin);
    w.bpart = w.partial + wgrad_partial_bytes(M, Cin, C, C1 ? C0 : C);
    w.sk = w.bpart + wgrad_bias_partial_bytes(M, Cin, C, C1 ? C0 : C);   // stream-K scratch of the dx GEMM
    w.total = w.sk + sk_ws_bytes();
    return w;
}
}  // namespace

int icn_upconv_bwd_supported(int B, int Cin, int Cout0, int Cout1, int r_in, int corner_mode) {
    try {
        return upconv_bwd_supported(B, Cin, Cout0, Cout1, r_in, corner_mode) ? 1 : 0;
    } catch (const std::exception&) {
        return 0;
    }
}

size_t icn_upconv_bwd_workspace_bytes(int B, int Cin, int Cout0, int Cout1, int r_in) {
    try {
        return upconv_bwd_supported(B, Cin, Cout0, Cout1, r_in, ICN_CORNER_AVERAGE) ? upconv_bwd_ws(B, Cin, Cout0, Cout1, r_in).total : 0;
    } catch (const std::exception&) {
        return 0;
    }
}

int icn_upconv_bwd(const float* x, const float* dy0, const float* dy1, const float* w0, const float* w1, float* dx, float* dw0,
                   float* dbias0, float* dw1, float* dbias1, int B, int Cin, int Cout0, int Cout1, int r_in, int corner_mode, void* ws,
                   size_t ws_bytes, void* stream) {
    return icn_upconv_bwd_streams(x, dy0, dy1, w0, w1, dx, dw0, dbias0, dw1, dbias1, B, Cin, Cout0, Cout1, r_in, corner_mode, ws, ws_bytes,
                                  stream, nullptr);
}

int icn_upconv_bwd_streams(const float* x, const float* dy0, const float* dy1, const float* w0, const float* w1, float* dx, float* dw0,
                           float* dbias0, float* dw1, float* dbias1, int B, int Cin, int Cout0, int Cout1, int r_in, int corner_mode,
                           void* ws, size_t ws_bytes, void* stream, void* weight_stream) {
    try {
        if (!dy0 || (Cout1 > 0) != (dy1 != nullptr)) throw std::invalid_argument("icn_upconv_bwd: dy1 goes with Cout1 > 0");
        if (dx && (!w0 || (Cout1 > 0 && !w1))) throw std::invalid_argument("icn_upconv_bwd: dx needs the weights");
        if (dw0 && (!x || (Cout1 > 0 && !dw1))) throw std::invalid_argument("icn_upconv_bwd: dw needs x (and dw1 for a pair)");
        if (!upconv_bwd_supported(B, Cin, Cout0, Cout1, r_in, corner_mode)) throw std::invalid_argument("icn_upconv_bwd: unsupported shape / corner mode");
        const UpconvBwdWs wo = upconv_bwd_ws(B, Cin, Cout0, Cout1, r_in);
        if (!ws || ws_bytes < wo.total) throw std::invalid_argument("icn_upconv_bwd: workspace too small");
        const UpconvBwdDev& t = upconv_bwd_tables(r_in, corner_mode);
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int C = Cout0 + Cout1, M = B * t.Pc;
        float* g = reinterpret_cast<float*>(at(ws, wo.g));
        // 1. g[b, s, t, :] = sum_p U[nbr_t(p), s] [dy0 | dy1][b, p, :]
        const EllSplitDev& ga = t.gather;
        const icn::PatchTab pt{t.ga_patch.prow, t.ga_patch.plocal, t.ga_patch.npatch, t.ga_patch.umax, t.ga_patch.gridW};
        if (icn::upconv_patch_usable(pt, B, (size_t)t.Pf, Cout0, Cout1))
            icn::launch_upconv_gather_lds(dy0, dy1, g, pt, t.px_cls, t.cls_coef, t.n_cls, B, t.Pf, t.Pc, Cout0, Cout1, s);
        else
            icn::launch_upconv_gather_px(dy0, dy1, g, t.px_srcs, t.px_cls, t.cls_coef, t.n_cls, B, t.Pf, t.Pc, Cout0, Cout1, s);
        icn::launch_upconv_gather(dy0, dy1, g, ga.ovf_idx, ga.ovf_coef, ga.ovf_rows, B, t.Pf, ga.n_ovf, 7 * t.Pc, Cout0, Cout1, ga.W_ovf, 0, s);
        // the weight gradients' stream (icn_upconv_bwd_streams): ordered after the aggregate pass, beside everything that follows
        hipStream_t sw = (dw0 && weight_stream) ? static_cast<hipStream_t>(weight_stream) : s;
        if (sw != s) {
            hipEvent_t ev;
            ICN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            const hipError_t e1 = hipEventRecord(ev, s);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(sw, ev, 0) : e1;
            (void)hipEventDestroy(ev);                                      // (released by the runtime once the wait has consumed it)
            ICN_HIP(e2);
        }
        if (dx) {
            // 2. dx[b, s, :] = g[b, s, (t, c)] . Wb[(t, c), :]: a dense GEMM, K = 7 * C (one "tap" whose gather is the identity)
            float* wb = reinterpret_cast<float*>(at(ws, wo.wb));
            icn::PrologueArgs p{};
            p.w = w0; p.w2 = w1; p.packed = wb; p.Cout = Cout0; p.Cout2 = Cout1; p.Cin = Cin; p.transpose = 2;
            int* sk_flag = reinterpret_cast<int*>(at(ws, wo.sk));
            p.zero = sk_flag; p.n_zero = icn::CONV_SK_FLAGS;
            icn::GatherGemmArgs a{};
            a.src = g; a.dst = dx; a.N0 = Cin; a.dcode = t.iota; a.perm = nullptr;   // identity rows
            a.Ps = t.Pc; a.Pd = t.Pc; a.K = 7 * C; a.N = Cin; a.E = 1; a.T = 1; a.M = M;
            a.segs.nseg = 1; a.segs.B = B; a.segs.cnt[0] = t.Pc; a.segs.off[0] = 0; a.segs.mask[0] = 1u;
            a.algo_flops = 2.0 * 7 * Cin * C * (double)B * t.Pc;            // executed: a quarter of the fine-level bwd-data it replaces
            a.sk_flag = sk_flag;
            a.sk_part = reinterpret_cast<float*>(reinterpret_cast<char*>(sk_flag) + align256((size_t)icn::CONV_SK_FLAGS * sizeof(int)));
            choose_arith(a, p, wb, (size_t)7 * C * Cin, false);
            icn::launch_conv_prologue(p, s);
            icn::launch_gather_gemm_auto(a, s);
        }
        if (dw0) {
            // 3. dW_t = sum_{b, s} x[b, s, :]^T g[b, s, t, :]   (+ dbias = sum g_0)
            icn::WgradArgs a{};
            a.x = x; a.dy = g; a.dy2 = nullptr; a.Cout0 = Cout1 ? Cout0 : C; a.dcode = t.iota; a.identity_rows = 1; a.n_slots = 0; a.y_taps = 7;
            a.partial = reinterpret_cast<float*>(at(ws, wo.partial));
            a.bias_partial = (dbias0 || dbias1) ? reinterpret_cast<float*>(at(ws, wo.bpart)) : nullptr;
            a.dw = dw0; a.dbias = dbias0; a.dw2 = dw1; a.dbias2 = dbias1;
            a.M = M; a.Ps = t.Pc; a.Pd = t.Pc; a.Cin = Cin; a.Cout = C; a.ns = 1 << r_in;
            a.algo_flops = 2.0 * 7 * Cin * C * (double)B * t.Pc;            // executed: a quarter of the fine-level bwd-weight
            icn::launch_wgrad(a, sw);
        }
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_upsample_fwd(const float* x, float* y, int B, int C, int r_in, int corner_mode, void* stream) {
    try {
        if (!x || !y || B < 1 || C < 1) throw std::invalid_argument("icn_upsample_fwd: bad arguments");
        const UpTables& t = up_tables(r_in, corner_mode);
        icn::launch_spmm_ell(x, y, t.idx_f, t.coef_f, B, t.Pc, t.Pf, C, t.Wf, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_upsample_bwd(const float* dy, float* dx, int B, int C, int r_in, int corner_mode, void* stream) {
    try {
        if (!dy || !dx || B < 1 || C < 1) throw std::invalid_argument("icn_upsample_bwd: bad arguments");
        const UpTables& t = up_tables(r_in, corner_mode);
        icn::launch_spmm_ell(dy, dx, t.idx_b, t.coef_b, B, t.Pf, t.Pc, C, t.Wb, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// ---- point-to-point loss -----------------------------------------------------------------------------------------
size_t icn_p2p_loss_workspace_floats(int B, int r) {
    return (B < 1 || r < 0 || r > 10) ? 0 : (size_t)3 * icn::p2p_loss_blocks(B, icn::pixels(r));
}

int icn_p2p_loss_fwd(const float* grid, const float* target, int B, int r, float f_pos, float f_nor, float f_lap, int lap_mode,
                     float* terms, float* ws, void* stream) {
    try {
        if (!grid || !target || !terms || !ws) throw std::invalid_argument("icn_p2p_loss_fwd: null pointer");
        if (B < 1 || r < 0 || r > 10) throw std::invalid_argument("icn_p2p_loss_fwd: bad B / subdivisions");
        if ((size_t)B * (icn::pixels(r) + 2) >= (size_t)1 << 31) throw std::invalid_argument("icn_p2p_loss_fwd: B * vertices exceeds int32");
        if (lap_mode < 0 || lap_mode > 3) throw std::invalid_argument("icn_p2p_loss_fwd: lap_mode must be a combination of ICN_LAP_*");
        icn::launch_p2p_loss_fwd(grid, target, vertex_faces(r), ws, terms, B, icn::pixels(r), 1 << r, f_pos, f_nor, f_lap, lap_mode,
                                 static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

size_t icn_p2p_loss_bwd_workspace_floats(int B, int r) {
    return (B < 1 || r < 0 || r > 10) ? 0 : (size_t)6 * B * (icn::pixels(r) + 2);
}

int icn_p2p_loss_bwd(const float* grid, const float* target, const float* upstream, int B, int r, float f_pos, float f_nor,
                     float f_lap, int lap_mode, float* dgrid, float* ws, void* stream) {
    try {
        if (!grid || !target || !upstream || !dgrid) throw std::invalid_argument("icn_p2p_loss_bwd: null pointer");
        if (B < 1 || r < 0 || r > 10) throw std::invalid_argument("icn_p2p_loss_bwd: bad B / subdivisions");
        if ((size_t)B * (icn::pixels(r) + 2) >= (size_t)1 << 31) throw std::invalid_argument("icn_p2p_loss_bwd: B * vertices exceeds int32");
        if ((f_nor != 0.f || f_lap != 0.f) && !ws) throw std::invalid_argument("icn_p2p_loss_bwd: workspace needed for the mesh terms");
        if (lap_mode < 0 || lap_mode > 3) throw std::invalid_argument("icn_p2p_loss_bwd: lap_mode must be a combination of ICN_LAP_*");
        icn::launch_p2p_loss_bwd(grid, target, vertex_faces(r), upstream, f_pos, f_nor, f_lap, lap_mode, dgrid, ws, B, icn::pixels(r),
                                 1 << r, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// ---- KL term and reparameterisation of the VAE -----------------------------------------------------------------------
size_t icn_kld_workspace_floats(size_t n) { return n < 1 ? 0 : (size_t)icn::kld_blocks(n); }

int icn_kld_fwd(const float* mu, const float* logvar, size_t n, float* out, float* ws, void* stream) {
    try {
        if (!mu || !logvar || !out || !ws || n < 1) throw std::invalid_argument("icn_kld_fwd: bad arguments");
        icn::launch_kld_fwd(mu, logvar, n, out, ws, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_kld_bwd(const float* mu, const float* logvar, const float* upstream, size_t n, float* dmu, float* dlogvar, void* stream) {
    try {
        if (!mu || !logvar || !upstream || !dmu || !dlogvar || n < 1) throw std::invalid_argument("icn_kld_bwd: bad arguments");
        icn::launch_kld_bwd(mu, logvar, upstream, n, dmu, dlogvar, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_reparam_fwd(const float* mu, const float* logvar, const float* eps, size_t n, float* z, void* stream) {
    try {
        if (!mu || !logvar || !eps || !z || n < 1) throw std::invalid_argument("icn_reparam_fwd: bad arguments");
        icn::launch_reparam_fwd(mu, logvar, eps, n, z, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_reparam_bwd(const float* dz, const float* logvar, const float* eps, size_t n, float* dmu, float* dlogvar, void* stream) {
    try {
        if (!dz || !logvar || !eps || !dmu || !dlogvar || n < 1) throw std::invalid_argument("icn_reparam_bwd: bad arguments");
        icn::launch_reparam_bwd(dz, logvar, eps, n, dmu, dlogvar, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_point_to_mesh(const float* points, const float* vertices, const int32_t* faces, int B, int P, int V, int F, float* dist2,
                      int32_t* face, int32_t* kind, void* stream) {
    try {
        if (!points || !vertices || !faces || !dist2 || !face || !kind) throw std::invalid_argument("icn_point_to_mesh: null pointer");
        if (B < 1 || B > 65535 || P < 1 || V < 1 || F < 1) throw std::invalid_argument("icn_point_to_mesh: bad sizes");
        icn::launch_point_to_mesh(points, vertices, faces, B, P, V, F, dist2, face, kind, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const size_t* numel, const float* step_size, const float* bc2_sqrt, double beta1, double beta2, double eps,
                  double weight_decay, void* stream) {
    try {
        if (count < 0 || (count > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !step_size || !bc2_sqrt)))
            throw std::invalid_argument("icn_adam_step: bad arguments");
        for (int i = 0; i < count; ++i)
            if (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]))
                throw std::invalid_argument("icn_adam_step: null tensor pointer");
        if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0))
            throw std::invalid_argument("icn_adam_step: betas must lie in [0, 1), eps must not be negative");
        if (count == 0) return 0;
        icn::launch_adam(count, params, grads, exp_avg, exp_avg_sq, numel, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay,
                         static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_adam_step_dev(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const size_t* numel, const float* scalars_dev, double beta1, double beta2, double eps, double weight_decay,
                      void* stream) {
    try {
        if (count < 0 || (count > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)) || !scalars_dev)
            throw std::invalid_argument("icn_adam_step_dev: bad arguments");
        for (int i = 0; i < count; ++i)
            if (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]))
                throw std::invalid_argument("icn_adam_step_dev: null tensor pointer");
        if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0))
            throw std::invalid_argument("icn_adam_step_dev: betas must lie in [0, 1), eps must not be negative");
        if (count == 0) return 0;
        std::vector<float> same(count, 0.f);              // one step count for all tensors: one group of launches
        icn::launch_adam(count, params, grads, exp_avg, exp_avg_sq, numel, same.data(), same.data(), beta1, beta2, eps, weight_decay,
                         static_cast<hipStream_t>(stream), scalars_dev);
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

long icn_table_stream_k(int ntiles, int grid, int nk, int ku, int32_t* out, size_t cap) {
    try {
        if (ntiles < 1 || grid < 8 || grid % 8 != 0 || nk < 1 || ku < 1)
            throw std::invalid_argument("icn_table_stream_k: grid must be a positive multiple of 8, nk and ku positive");
        size_t n = 0;
        std::vector<int> tables(2 * (grid / 8 + 1));
        const int occ = grid / 256;                       // the launcher's grids are 256 CUs x blocks per CU
        icn::sk_tables(ntiles, grid, nk, ku, occ, grid % 256 == 0 ? icn::sk_speed_factors(occ) : nullptr, tables.data());
        for (int b = 0; b < grid; ++b) {
            icn::SkWalk w{};
            w.init(b, grid, ntiles, nk, ku, tables.data());
            int tile, k0, k1;
            while (w.next(tile, k0, k1)) {
                if (out && n + 4 <= cap) { out[n] = b; out[n + 1] = tile; out[n + 2] = k0; out[n + 3] = k1; }
                n += 4;
            }
        }
        return (long)n;
    } catch (const std::exception& e) {
        fail(e.what());
        return -1;
    }
}

long icn_table_tile_lists(int r_in, int stride, int corner_mode, int B, int bm, int ntn, int grid, int32_t* out, size_t cap) {
    try {
        if (r_in < 1 || r_in > 8 || stride != 2 || B < 1 || (bm != 64 && bm != 128) || ntn < 1 || grid < 8 || grid % 256 != 0)
            throw std::invalid_argument("icn_table_tile_lists: stride-2 tables, bm 64 / 128, grid a multiple of 256");
        std::vector<int32_t> bwd, perm;
        std::vector<uint8_t> mask;
        const int E = icn::build_conv_bwd(r_in, stride, corner_mode, bwd);
        icn::build_bwd_row_order(r_in, stride, bwd, E, perm, mask);
        const std::vector<uint32_t> m32(mask.begin(), mask.end());
        const int Pd = icn::pixels(r_in), M = B * Pd, ntiles = ((M + bm - 1) / bm) * ntn;
        std::vector<int> h;
        icn::build_tile_lists(m32.data(), M, Pd, bm, ntn, ntiles, grid, grid / 256, h);
        if (out)
            for (size_t i = 0; i < h.size() && i < cap; ++i) out[i] = h[i];
        return (long)h.size();
    } catch (const std::exception& e) {
        fail(e.what());
        return -1;
    }
}

int icn_set_debug_flags(int flags) { return icn::set_debug_flags(flags); }

int icn_get_arith(void) {
    try {
        return icn::arith_mode();
    } catch (const std::exception& e) {
        fail(e.what());
        return -1;
    }
}
int icn_set_arith(int mode) {
    try {
        return icn::set_arith_mode(mode);
    } catch (const std::exception& e) {
        fail(e.what());
        return -1;
    }
}
unsigned icn_build_flags(void) { return icn::build_flags(); }

int icn_debug_trace(void* device_buffer, size_t n_u64) {
    icn::set_trace_buffer(device_buffer, n_u64);
    return 0;
}

long icn_host_selfcheck(int r, int corner_mode) {
    // Every host-side table builder and planner the device paths use at level r, with the device copies skipped: lets the
    // whole host side of the library run under ASan / UBSan on a machine without a GPU (tools/asan_host.sh).  Returns the
    // number of table elements that would have been uploaded.
    try {
        if (r < 0 || r > 7 || (corner_mode != 0 && corner_mode != 1)) throw std::invalid_argument("icn_host_selfcheck: r in [0, 7], corner_mode 0 / 1");
        g_dry_run = true;
        g_dry_elems = 0;
        struct Reset { ~Reset() { g_dry_run = false; } } reset;
        (void)make_conv_tables(r, 1, corner_mode);
        if (r >= 1) (void)make_conv_tables(r, 2, corner_mode);
        (void)make_up_tables(r, corner_mode);
        (void)make_upconv_tables(r, corner_mode);
        (void)make_upconv_bwd_tables(r, corner_mode);
        std::vector<int32_t> vf;
        icn::build_vertex_faces(r, vf);
        (void)upload(vf);
        (void)table_counts(r, 1);
        if (r >= 1) (void)table_counts(r, 2);
        (void)upconv_slots(r);
        // launch planning of the GEMMs at this level for the model's channel counts and a few batch sizes
        const int chans[] = {64, 128, 256, 512};
        size_t plan = 0;
        for (int B : {1, 3, 36})
            for (int ci : chans)
                for (int co : chans) {
                    for (int stride = 1; stride <= (r >= 1 ? 2 : 1); ++stride) {
                        for (int op = 0; op < 3; ++op) plan += conv_ws_bytes(op, B, ci, co, 0, r, stride) != 0;
                        if (pair_supported(B, ci, co, co, r, stride))
                            for (int op = 0; op < 3; ++op) plan += conv_ws_bytes(op, B, ci, co, co, r, stride) != 0;
                        const int n_out = (1 << r) / stride, M = B * 10 * n_out * n_out;
                        plan += (size_t)icn::wgrad_splits(M, ci, co, 0);
                    }
                    plan += icn_upconv_workspace_bytes(B, ci, co, co, r) != 0;
                    plan += icn_upconv_bwd_workspace_bytes(B, ci, co, co, r) != 0;
                }
        return (long)(g_dry_elems + plan);
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_device_status(int clear) {
    try {
        return icn::device_status(clear);
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// ---- fused BatchNorm + ReLU ---------------------------------------------------------------------------------
// (the partial sums are doubles: twice the floats)
size_t icn_bn_workspace_floats(int M, int C) { return (M < 1 || C < 1) ? 0 : (size_t)icn::bn_chunks(M) * 4 * C * 2; }

int icn_bn_stats(const float* x, int M, int C, float eps, float momentum, float* running_mean, float* running_var, float* stat,
                 float* ws, void* stream) {
    try {
        if (!x || !stat || !ws || M < 1) throw std::invalid_argument("icn_bn_stats: bad arguments");
        if (!icn::bn_supported(C)) throw std::invalid_argument("icn_bn_stats: unsupported channel count");
        icn::launch_bn_stats(x, M, C, eps, momentum, running_mean, running_var, stat, ws, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_bn_stats2(const float* a, const float* b, int M, int C, float eps_a, float momentum_a, float* running_mean_a,
                  float* running_var_a, float* stat_a, float eps_b, float momentum_b, float* running_mean_b, float* running_var_b,
                  float* stat_b, float* ws, void* stream) {
    try {
        if (!a || !b || !stat_a || !stat_b || !ws || M < 1) throw std::invalid_argument("icn_bn_stats2: bad arguments");
        if ((running_mean_a == nullptr) != (running_var_a == nullptr) || (running_mean_b == nullptr) != (running_var_b == nullptr))
            throw std::invalid_argument("icn_bn_stats2: running mean and variance go together");
        if (!icn::bn_supported(C)) throw std::invalid_argument("icn_bn_stats2: unsupported channel count");
        icn::launch_bn_stats2(a, b, M, C, eps_a, momentum_a, running_mean_a, running_var_a, stat_a, eps_b, momentum_b, running_mean_b,
                              running_var_b, stat_b, ws, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_bn_relu_fwd(const float* a, const float* b, const float* stat_a, const float* stat_b, const float* gamma_a,
                    const float* beta_a, const float* gamma_b, const float* beta_b, float* y, int M, int C, void* stream) {
    try {
        if (!a || !stat_a || !gamma_a || !beta_a || !y || M < 1) throw std::invalid_argument("icn_bn_relu_fwd: bad arguments");
        if (b && (!stat_b || !gamma_b || !beta_b)) throw std::invalid_argument("icn_bn_relu_fwd: second input needs its stats");
        if (!icn::bn_supported(C)) throw std::invalid_argument("icn_bn_relu_fwd: unsupported channel count");
        icn::launch_bn_relu_fwd(a, b, stat_a, stat_b, gamma_a, beta_a, gamma_b, beta_b, y, M, C, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_bn_relu_bwd(const float* dy, const float* a, const float* b, const float* stat_a, const float* stat_b, const float* gamma_a,
                    const float* beta_a, const float* gamma_b, const float* beta_b, float* da, float* db, float* sums, float* ws, int M,
                    int C, float* dbeta_a, float* dgamma_a, float* dbeta_b, float* dgamma_b, void* stream) {
    try {
        if (!dy || !a || !stat_a || !gamma_a || !beta_a || !da || !sums || !ws || M < 1)
            throw std::invalid_argument("icn_bn_relu_bwd: bad arguments");
        if (b && (!stat_b || !gamma_b || !beta_b || !db)) throw std::invalid_argument("icn_bn_relu_bwd: second input needs its buffers");
        if (!icn::bn_supported(C)) throw std::invalid_argument("icn_bn_relu_bwd: unsupported channel count");
        icn::launch_bn_relu_bwd(dy, a, b, stat_a, stat_b, gamma_a, beta_a, gamma_b, beta_b, da, db, sums, ws, M, C, dbeta_a, dgamma_a,
                                dbeta_b, dgamma_b, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// ---- fused 1x1 head + tanh ------------------------------------------------------------------------------------
size_t icn_head_workspace_floats(int M, int Cin) { return (M < 1 || Cin < 1) ? 0 : (size_t)icn::head_chunks(M) * 4 * (Cin + 4); }

int icn_head_fwd(const float* x, const float* w, const float* bias, float* y, int M, int Cin, int Cout, void* stream) {
    try {
        if (!x || !w || !bias || !y || M < 1) throw std::invalid_argument("icn_head_fwd: bad arguments");
        if (!icn::head_supported(Cin, Cout)) throw std::invalid_argument("icn_head_fwd: unsupported channel counts");
        icn::launch_head_fwd(x, w, bias, y, M, Cin, Cout, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int icn_head_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw, float* db, float* ws,
                 int M, int Cin, int Cout, void* stream) {
    try {
        if (!dy || !y || !x || !w || !dw || !db || !ws || M < 1) throw std::invalid_argument("icn_head_bwd: bad arguments");
        if (!icn::head_supported(Cin, Cout)) throw std::invalid_argument("icn_head_bwd: unsupported channel counts");
        icn::launch_head_bwd(dy, y, x, w, dx, dw, db, ws, M, Cin, Cout, static_cast<hipStream_t>(stream));
        ICN_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// ---- host-side introspection ----------------------------------------------------------------------------
long icn_table_conv_fwd(int r_in, int stride, int corner_mode, int32_t* out, size_t cap) {
    try {
        std::vector<int32_t> v;
        icn::build_conv_fwd(r_in, stride, corner_mode, v);
        if (out) std::memcpy(out, v.data(), std::min(cap, v.size()) * sizeof(int32_t));
        return (long)v.size();
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

long icn_table_wgrad7(int r_in, int stride, int corner_mode, int32_t* rows, size_t cap_rows, uint16_t* pos, size_t cap_pos, int* meta) {
    try {
        std::vector<int32_t> fwd;
        icn::build_conv_fwd(r_in, stride, corner_mode, fwd);
        const int P = (int)(fwd.size() / icn::NTAPS);
        icn::DmaTable h;
        icn::build_dma_table(fwd, 1, P, h);
        icn::Wg7Table w7;
        if (!icn::build_wgrad7(h, P, wgrad7_union_rows(stride), w7)) {      // the very table make_conv_tables uploads
            if (meta) meta[0] = meta[1] = 0;
            return 0;
        }
        if (meta) { meta[0] = w7.U; meta[1] = w7.npatch; }
        if (rows) std::memcpy(rows, w7.urow.data(), std::min(cap_rows, w7.urow.size()) * sizeof(int32_t));
        if (pos) std::memcpy(pos, w7.upos.data(), std::min(cap_pos, w7.upos.size()) * sizeof(uint16_t));
        return (long)w7.urow.size();
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

long icn_table_conv_bwd(int r_in, int stride, int corner_mode, int32_t* out, size_t cap, int* width) {
    try {
        std::vector<int32_t> v;
        const int E = icn::build_conv_bwd(r_in, stride, corner_mode, v);
        if (width) *width = E;
        if (out) std::memcpy(out, v.data(), std::min(cap, v.size()) * sizeof(int32_t));
        return (long)v.size();
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

long icn_table_upsample(int r_in, int corner_mode, int transpose, int32_t* idx, float* coef, size_t cap, int* width) {
    try {
        icn::Ell f, b;
        icn::build_upsample(r_in, corner_mode, f, b);
        const icn::Ell& e = transpose ? b : f;
        if (width) *width = e.width;
        if (idx) std::memcpy(idx, e.idx.data(), std::min(cap, e.idx.size()) * sizeof(int32_t));
        if (coef) std::memcpy(coef, e.coef.data(), std::min(cap, e.coef.size()) * sizeof(float));
        return (long)e.idx.size();
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

long icn_table_upsample_pairs(int r_in, int32_t* out, size_t cap) {
    try {
        std::vector<int32_t> v;
        icn::build_upsample_pairs(r_in, v);
        if (out) std::memcpy(out, v.data(), std::min(cap, v.size()) * sizeof(int32_t));
        return (long)v.size();
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

long icn_table_upconv(int r_in, int corner_mode, int32_t* ints, size_t cap_ints, float* floats, size_t cap_floats, int* meta) {
    try {
        icn::UpconvTable t;
        icn::build_upconv_fwd(r_in, corner_mode, t);
        std::vector<int32_t> iv;
        for (int sg = 0; sg < t.nseg; ++sg) {
            iv.push_back(t.seg_cnt[sg]);
            iv.push_back(t.seg_off[sg]);
            iv.push_back((int32_t)t.seg_mask[sg]);
        }
        iv.insert(iv.end(), t.pix.begin(), t.pix.end());
        iv.insert(iv.end(), t.code.begin(), t.code.end());
        iv.insert(iv.end(), t.slot_idx.begin(), t.slot_idx.end());
        std::vector<float> fv(t.alpha);
        fv.insert(fv.end(), t.slot_coef.begin(), t.slot_coef.end());
        if (meta) {
            meta[0] = t.Pf; meta[1] = t.Pc; meta[2] = t.n_slots; meta[3] = t.E; meta[4] = t.nseg; meta[5] = icn::UPCONV_TAPS;
            meta[6] = (int)fv.size();
        }
        if (ints) std::memcpy(ints, iv.data(), std::min(cap_ints, iv.size()) * sizeof(int32_t));
        if (floats) std::memcpy(floats, fv.data(), std::min(cap_floats, fv.size()) * sizeof(float));
        return (long)iv.size();
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

long icn_table_upconv_bwd(int r_in, int corner_mode, int32_t* idx, float* coef, size_t cap, int* width) {
    try {
        icn::Ell e;
        icn::build_upconv_bwd(r_in, corner_mode, e);
        if (width) *width = e.width;
        if (idx) std::memcpy(idx, e.idx.data(), std::min(cap, e.idx.size()) * sizeof(int32_t));
        if (coef) std::memcpy(coef, e.coef.data(), std::min(cap, e.coef.size()) * sizeof(float));
        return (long)e.idx.size();
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

long icn_table_faces(int r, int32_t* out, size_t cap) {
    try {
        std::vector<int32_t> v;
        icn::build_faces(r, v);
        if (out) std::memcpy(out, v.data(), std::min(cap, v.size()) * sizeof(int32_t));
        return (long)v.size();
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

}  // extern "C"
