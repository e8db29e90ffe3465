// Point-to-point loss of the training step on the HIP path (reference losses.py:10-85 Point2Point_Loss; mesh helpers as
// restated in geniconet_amd/losses.py: vertex normals = area-weighted face-normal sums, generate.py:20-43; Laplacian =
// uniform umbrella mean(1-ring) - v by default -- upstream's mesh.utils.compute_laplacian is absent, so its sign and
// normalisation are a parameter: lap_mode bit 0 = opposite sign (v - mean), bit 1 = valence-weighted (sum(ring) - k v)).
// HBM/L2-bound, a few tens of microseconds; replaces ~25 torch launches per step.
//
//   v[b, i]   i < P: network output pixel i (channels-last (B, P, 3));  i = P, P + 1: N / S pole = mean of 5 corner pixels
//   terms[0] = mean_{b,i,c} (v - t_pos)^2           terms[1] = mean_{b,i} (1 - cos(unit vertex normal, t_nor))
//   terms[2] = mean_{b,i,c} (lap(v) - t_lap)^2      terms[3] = f_pos * terms[0] + f_nor * terms[1] + f_lap * terms[2]
// target is (B, 9, V): rows 0:3 positions, 3:6 normals, 6:9 Laplacians (data.py:64-69).
// Sums are two-level with fixed block size and a fixed tree (deterministic).
//
// Backward (d terms[3] / d grid), all in gather form over the incident-face table, so no atomics and no write conflicts
// (derivation and finite-difference check: oracle/loss_ref.py p2p_grad):
//   position : 2 f_pos (v - t_pos) / (3 B V)
//   Laplacian: lap_i = c_i (mean(ring_i) - v_i), c_i = +-1 or +-k_i (lap_mode), e_i = lap_i - t_lap,i, k_i = valence:
//              2 f_lap (sum_{i in ring(j)} c_i e_i / k_i - c_j e_j) / (3 B V)                                    (ring symmetric)
//   normal   : w_i = sum of the face normals at i, u = w / |w|, c = u . t^ (t^ = t_nor / |t_nor|);
//              h_i = d(1 - c_i) / d w_i = -(t^ - (u . t^) u) / |w_i|;  face (j, p, q): d/d v_j = (h_j + h_p + h_q) x (v_q - v_p);
//              times f_nor / (B V).   Where a normalisation clamp is active (|w| <= 1e-10, |t_nor| <= 1e-8) h_i = 0.
//   vertex -> grid: pixels directly; a pole is the mean of its 5 corner pixels, each gets a fifth of the pole's gradient.
// k_p2p_bwd_prep stores c_i e_i / k_i and h_i per vertex (6 floats), k_p2p_bwd gathers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "icn_launch.h"

namespace icn {

namespace {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ int pole_corner(int n, int k, int c) {      // reference ico_utils.py:10-24
    return k == 0 ? (c * n) * 2 * n : ((c + 1) * n - 1) * 2 * n + (2 * n - 1);
}

// vertex j of sample b from the channels-last grid (B, P, 3); poles: sum of the 5 corners / 5 (torch .mean)
__device__ __forceinline__ f3 vertex(const float* __restrict__ g, int b, int j, int P, int n) {
    if (j < P) {
        const float* p = g + ((size_t)b * P + j) * 3;
        return {p[0], p[1], p[2]};
    }
    f3 s = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const float* p = g + ((size_t)b * P + pole_corner(n, j - P, c)) * 3;
        s = s + f3{p[0], p[1], p[2]};
    }
    return {s.x / 5.f, s.y / 5.f, s.z / 5.f};
}

constexpr int LOSS_BLOCK = 256;

// factor c_i of the Laplacian convention: lap_i = c_i (mean(ring_i) - v_i)
__device__ __forceinline__ float lap_factor(int lap_mode, int k) {
    const float w = (lap_mode & 2) ? (float)k : 1.f;
    return (lap_mode & 1) ? -w : w;
}

// fixed-order block sum of three values; result valid in thread 0
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c) {
    __shared__ float red[3][LOSS_BLOCK / 64];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        b += __shfl_xor(b, d, 64);
        c += __shfl_xor(c, d, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = a; red[1][w] = b; red[2][w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        c = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    }
}

__global__ __launch_bounds__(LOSS_BLOCK) void k_p2p_fwd(const float* __restrict__ g, const float* __restrict__ target,
                                                         const int32_t* __restrict__ vf, float* __restrict__ partial, int B, int P,
                                                         int n, int lap_mode) {
    const int V = P + 2;
    const long idx = (long)blockIdx.x * LOSS_BLOCK + threadIdx.x;
    float e_pos = 0.f, e_nor = 0.f, e_lap = 0.f;
    if (idx < (long)B * V) {
        const int b = (int)(idx / V), i = (int)(idx % V);
        const f3 v = vertex(g, b, i, P, n);
        const float* t = target + (size_t)b * 9 * V + i;                 // component c at t[c * V]
        const f3 tp = {t[0], t[(size_t)V], t[(size_t)2 * V]};
        const f3 tn = {t[(size_t)3 * V], t[(size_t)4 * V], t[(size_t)5 * V]};
        const f3 tl = {t[(size_t)6 * V], t[(size_t)7 * V], t[(size_t)8 * V]};
        const f3 d = v - tp;
        e_pos = dot(d, d);
        f3 vn = {0.f, 0.f, 0.f}, ring = {0.f, 0.f, 0.f};
        int k = 0;
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const int p = vf[((size_t)i * 6 + f) * 2], q = vf[((size_t)i * 6 + f) * 2 + 1];
            if (p < 0) continue;
            const f3 vp = vertex(g, b, p, P, n), vq = vertex(g, b, q, P, n);
            vn = vn + cross(vp - v, vq - v);
            ring = ring + vp;
            ++k;
        }
        const float len = fmaxf(sqrtf(dot(vn, vn)), 1e-10f);              // losses.py helper: vn / clamp_min(|vn|, eps)
        const f3 u = {vn.x / len, vn.y / len, vn.z / len};
        // F.cosine_similarity(u, tn, eps = 1e-8): each norm clamped separately
        const float cs = dot(u, tn) / (fmaxf(sqrtf(dot(u, u)), 1e-8f) * fmaxf(sqrtf(dot(tn, tn)), 1e-8f));
        e_nor = 1.f - cs;
        const float inv = 1.f / (float)k, cf = lap_factor(lap_mode, k);
        const f3 um = f3{ring.x * inv, ring.y * inv, ring.z * inv} - v;
        const f3 dl = f3{cf * um.x, cf * um.y, cf * um.z} - tl;
        e_lap = dot(dl, dl);
    }
    block_sum3(e_pos, e_nor, e_lap);
    if (threadIdx.x == 0) {
        partial[(size_t)blockIdx.x * 3 + 0] = e_pos;
        partial[(size_t)blockIdx.x * 3 + 1] = e_nor;
        partial[(size_t)blockIdx.x * 3 + 2] = e_lap;
    }
}

// terms from the per-block partials: one block, lanes stride over the partials in double, fixed tree
__global__ __launch_bounds__(LOSS_BLOCK) void k_p2p_finalize(const float* __restrict__ partial, int nblocks, double count_vec,
                                                              double count_vtx, float f_pos, float f_nor, float f_lap,
                                                              float* __restrict__ terms) {
    __shared__ double red[3][LOSS_BLOCK];
    double s[3] = {0.0, 0.0, 0.0};
    for (int k = threadIdx.x; k < nblocks; k += LOSS_BLOCK)
        for (int c = 0; c < 3; ++c) s[c] += (double)partial[(size_t)k * 3 + c];
    for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int d = LOSS_BLOCK / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d)
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l_pos = (float)(red[0][0] / count_vec), l_nor = (float)(red[1][0] / count_vtx), l_lap = (float)(red[2][0] / count_vec);
        terms[0] = l_pos;
        terms[1] = l_nor;
        terms[2] = l_lap;
        terms[3] = f_pos * l_pos + f_nor * l_nor + f_lap * l_lap;
    }
}

// d(f_pos * terms[0]) / d(grid) * upstream:  2 (v - t_pos) / (B V 3) at every pixel, plus a fifth of the pole's term at each
// of its 5 corner pixels (the pole is their mean).  One thread per (sample, pixel): no write conflicts.
__global__ __launch_bounds__(LOSS_BLOCK) void k_p2p_bwd_pos(const float* __restrict__ g, const float* __restrict__ target,
                                                             const float* __restrict__ upstream, float scale,
                                                             float* __restrict__ dg, int B, int P, int n) {
    const int V = P + 2;
    const long idx = (long)blockIdx.x * LOSS_BLOCK + threadIdx.x;
    if (idx >= (long)B * P) return;
    const int b = (int)(idx / P), j = (int)(idx % P);
    const float w = scale * upstream[0];
    const float* t = target + (size_t)b * 9 * V;
    const float* p = g + ((size_t)b * P + j) * 3;
    f3 d = {p[0] - t[j], p[1] - t[(size_t)V + j], p[2] - t[(size_t)2 * V + j]};
    const int chart = 2 * n * n;                                         // pixels per chart
    int pole = -1;
    if (j % chart == 0) pole = 0;                                        // (row c*n, col 0): corner of the N pole
    else if ((j + 1) % chart == 0) pole = 1;                             // (row (c+1)*n - 1, col 2n - 1): S pole
    if (pole >= 0) {
        const f3 vp = vertex(g, b, P + pole, P, n);
        const f3 dp = {vp.x - t[P + pole], vp.y - t[(size_t)V + P + pole], vp.z - t[(size_t)2 * V + P + pole]};
        d = d + f3{dp.x / 5.f, dp.y / 5.f, dp.z / 5.f};
    }
    float* o = dg + ((size_t)b * P + j) * 3;
    o[0] = w * d.x;
    o[1] = w * d.y;
    o[2] = w * d.z;
}

// per vertex: aux[0:3] = (lap - t_lap) / k,  aux[3:6] = h = d(1 - cos) / d w  (see the header comment)
__global__ __launch_bounds__(LOSS_BLOCK) void k_p2p_bwd_prep(const float* __restrict__ g, const float* __restrict__ target,
                                                              const int32_t* __restrict__ vf, float* __restrict__ aux, int B, int P,
                                                              int n, int lap_mode) {
    const int V = P + 2;
    const long idx = (long)blockIdx.x * LOSS_BLOCK + threadIdx.x;
    if (idx >= (long)B * V) return;
    const int b = (int)(idx / V), i = (int)(idx % V);
    const f3 v = vertex(g, b, i, P, n);
    const float* t = target + (size_t)b * 9 * V + i;
    const f3 tn = {t[(size_t)3 * V], t[(size_t)4 * V], t[(size_t)5 * V]};
    const f3 tl = {t[(size_t)6 * V], t[(size_t)7 * V], t[(size_t)8 * V]};
    f3 w = {0.f, 0.f, 0.f}, ring = {0.f, 0.f, 0.f};
    int k = 0;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const int p = vf[((size_t)i * 6 + f) * 2], q = vf[((size_t)i * 6 + f) * 2 + 1];
        if (p < 0) continue;
        const f3 vp = vertex(g, b, p, P, n), vq = vertex(g, b, q, P, n);
        w = w + cross(vp - v, vq - v);
        ring = ring + vp;
        ++k;
    }
    const float inv = 1.f / (float)k, cf = lap_factor(lap_mode, k);
    const f3 um = f3{ring.x * inv, ring.y * inv, ring.z * inv} - v;
    const f3 e = f3{cf * um.x, cf * um.y, cf * um.z} - tl;
    f3 h = {0.f, 0.f, 0.f};
    const float wl = sqrtf(dot(w, w)), tl2 = sqrtf(dot(tn, tn));
    if (wl > 1e-10f && tl2 > 1e-8f) {
        const f3 u = {w.x / wl, w.y / wl, w.z / wl}, th = {tn.x / tl2, tn.y / tl2, tn.z / tl2};
        const float c = dot(u, th);
        h = {-(th.x - c * u.x) / wl, -(th.y - c * u.y) / wl, -(th.z - c * u.z) / wl};
    }
    float* a = aux + (size_t)idx * 6;
    const float ci = cf * inv;
    a[0] = e.x * ci; a[1] = e.y * ci; a[2] = e.z * ci;
    a[3] = h.x; a[4] = h.y; a[5] = h.z;
}

__device__ __forceinline__ f3 aux3(const float* __restrict__ aux, int b, int x, int V, int off) {
    const float* a = aux + ((size_t)b * V + x) * 6 + off;
    return {a[0], a[1], a[2]};
}

// gradient of the weighted loss with respect to vertex x of sample b (cp, cl, cn: the three terms' constant factors)
__device__ __forceinline__ f3 vertex_grad(const float* __restrict__ g, const float* __restrict__ target,
                                          const int32_t* __restrict__ vf, const float* __restrict__ aux, int b, int x, int P, int n,
                                          float cp, float cl, float cn) {
    const int V = P + 2;
    const f3 v = vertex(g, b, x, P, n);
    const float* t = target + (size_t)b * 9 * V + x;
    f3 d = {cp * (v.x - t[0]), cp * (v.y - t[(size_t)V]), cp * (v.z - t[(size_t)2 * V])};
    const f3 ex = aux3(aux, b, x, V, 0), hx = aux3(aux, b, x, V, 3);
    f3 gl = {0.f, 0.f, 0.f}, gn = {0.f, 0.f, 0.f};
    int k = 0;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const int p = vf[((size_t)x * 6 + f) * 2], q = vf[((size_t)x * 6 + f) * 2 + 1];
        if (p < 0) continue;
        gl = gl + aux3(aux, b, p, V, 0);
        const f3 G = hx + aux3(aux, b, p, V, 3) + aux3(aux, b, q, V, 3);
        gn = gn + cross(G, vertex(g, b, q, P, n) - vertex(g, b, p, P, n));
        ++k;
    }
    const float kf = (float)k;
    d = d + f3{cl * (gl.x - kf * ex.x), cl * (gl.y - kf * ex.y), cl * (gl.z - kf * ex.z)};
    return d + f3{cn * gn.x, cn * gn.y, cn * gn.z};
}

__global__ __launch_bounds__(LOSS_BLOCK) void k_p2p_bwd(const float* __restrict__ g, const float* __restrict__ target,
                                                         const int32_t* __restrict__ vf, const float* __restrict__ aux,
                                                         const float* __restrict__ upstream, float cp, float cl, float cn,
                                                         float* __restrict__ dg, int B, int P, int n) {
    const long idx = (long)blockIdx.x * LOSS_BLOCK + threadIdx.x;
    if (idx >= (long)B * P) return;
    const int b = (int)(idx / P), j = (int)(idx % P);
    f3 d = vertex_grad(g, target, vf, aux, b, j, P, n, cp, cl, cn);
    const int chart = 2 * n * n;
    int pole = -1;
    if (j % chart == 0) pole = 0;
    else if ((j + 1) % chart == 0) pole = 1;
    if (pole >= 0) {
        const f3 dp = vertex_grad(g, target, vf, aux, b, P + pole, P, n, cp, cl, cn);
        d = d + f3{dp.x / 5.f, dp.y / 5.f, dp.z / 5.f};
    }
    const float up = upstream[0];
    float* o = dg + ((size_t)b * P + j) * 3;
    o[0] = up * d.x;
    o[1] = up * d.y;
    o[2] = up * d.z;
}

}  // namespace

int p2p_loss_blocks(int B, int P) { return (int)(((long)B * (P + 2) + LOSS_BLOCK - 1) / LOSS_BLOCK); }

void launch_p2p_loss_fwd(const float* grid, const float* target, const int32_t* vf, float* partial, float* terms, int B, int P, int n,
                         float f_pos, float f_nor, float f_lap, int lap_mode, hipStream_t s) {
    const int nb = p2p_loss_blocks(B, P);
    hipLaunchKernelGGL(k_p2p_fwd, dim3(nb), dim3(LOSS_BLOCK), 0, s, grid, target, vf, partial, B, P, n, lap_mode);
    const double nv = (double)B * (P + 2);
    hipLaunchKernelGGL(k_p2p_finalize, dim3(1), dim3(LOSS_BLOCK), 0, s, partial, nb, nv * 3.0, nv, f_pos, f_nor, f_lap, terms);
}

void launch_p2p_loss_bwd_pos(const float* grid, const float* target, const float* upstream, float f_pos, float* dgrid, int B, int P,
                             int n, hipStream_t s) {
    const long total = (long)B * P;
    const float scale = (float)(2.0 * (double)f_pos / ((double)B * (P + 2) * 3.0));
    hipLaunchKernelGGL(k_p2p_bwd_pos, dim3((unsigned)((total + LOSS_BLOCK - 1) / LOSS_BLOCK)), dim3(LOSS_BLOCK), 0, s, grid, target,
                       upstream, scale, dgrid, B, P, n);
}

void launch_p2p_loss_bwd(const float* grid, const float* target, const int32_t* vf, const float* upstream, float f_pos, float f_nor,
                         float f_lap, int lap_mode, float* dgrid, float* aux, int B, int P, int n, hipStream_t s) {
    if (f_nor == 0.f && f_lap == 0.f) return launch_p2p_loss_bwd_pos(grid, target, upstream, f_pos, dgrid, B, P, n, s);
    const double nv = (double)B * (P + 2);
    const long tv = (long)B * (P + 2), tp = (long)B * P;
    hipLaunchKernelGGL(k_p2p_bwd_prep, dim3((unsigned)((tv + LOSS_BLOCK - 1) / LOSS_BLOCK)), dim3(LOSS_BLOCK), 0, s, grid, target, vf, aux,
                       B, P, n, lap_mode);
    hipLaunchKernelGGL(k_p2p_bwd, dim3((unsigned)((tp + LOSS_BLOCK - 1) / LOSS_BLOCK)), dim3(LOSS_BLOCK), 0, s, grid, target, vf, aux,
                       upstream, (float)(2.0 * f_pos / (3.0 * nv)), (float)(2.0 * f_lap / (3.0 * nv)), (float)(f_nor / nv), dgrid, B, P, n);
}

// ---------------------------------------------------------------------------------------------------------
// KL term of the VAE loss (reference losses.py:105) and the reparameterisation (models.py:89-92): elementwise over the
// n = B * D latent values; the sum is two-level with a fixed tree (deterministic).
// ---------------------------------------------------------------------------------------------------------
constexpr int KLD_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(LOSS_BLOCK) void k_kld_partial(const float* __restrict__ mu, const float* __restrict__ lv, size_t n,
                                                             float* __restrict__ partial) {
    __shared__ float red[LOSS_BLOCK / 64];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * LOSS_BLOCK) {
        const float m = mu[i], l = lv[i];
        s += 1.f + l - m * m - __expf(l);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(LOSS_BLOCK) void k_kld_finalize(const float* __restrict__ partial, int nblocks, double scale,
                                                              float* __restrict__ out) {
    __shared__ double red[LOSS_BLOCK];
    double s = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += LOSS_BLOCK) s += (double)partial[k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = LOSS_BLOCK / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * scale);
}

__global__ __launch_bounds__(LOSS_BLOCK) void k_kld_bwd(const float* __restrict__ mu, const float* __restrict__ lv,
                                                         const float* __restrict__ upstream, float inv_n, size_t n,
                                                         float* __restrict__ dmu, float* __restrict__ dlv) {
    const float w = upstream[0] * inv_n;
    for (size_t i = (size_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * LOSS_BLOCK) {
        dmu[i] = w * mu[i];
        dlv[i] = 0.5f * w * (__expf(lv[i]) - 1.f);
    }
}

__global__ __launch_bounds__(LOSS_BLOCK) void k_reparam_fwd(const float* __restrict__ mu, const float* __restrict__ lv,
                                                             const float* __restrict__ eps, size_t n, float* __restrict__ z) {
    for (size_t i = (size_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * LOSS_BLOCK)
        z[i] = eps[i] * __expf(0.5f * lv[i]) + mu[i];
}

__global__ __launch_bounds__(LOSS_BLOCK) void k_reparam_bwd(const float* __restrict__ dz, const float* __restrict__ lv,
                                                             const float* __restrict__ eps, size_t n, float* __restrict__ dmu,
                                                             float* __restrict__ dlv) {
    for (size_t i = (size_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * LOSS_BLOCK) {
        const float g = dz[i];
        dmu[i] = g;
        dlv[i] = g * eps[i] * 0.5f * __expf(0.5f * lv[i]);
    }
}

static int ew_blocks(size_t n) { return (int)std::min((size_t)KLD_MAX_BLOCKS, (n + LOSS_BLOCK - 1) / LOSS_BLOCK); }

int kld_blocks(size_t n) { return ew_blocks(n); }

void launch_kld_fwd(const float* mu, const float* logvar, size_t n, float* out, float* partial, hipStream_t s) {
    const int nb = ew_blocks(n);
    hipLaunchKernelGGL(k_kld_partial, dim3(nb), dim3(LOSS_BLOCK), 0, s, mu, logvar, n, partial);
    hipLaunchKernelGGL(k_kld_finalize, dim3(1), dim3(LOSS_BLOCK), 0, s, partial, nb, -0.5 / (double)n, out);
}

void launch_kld_bwd(const float* mu, const float* logvar, const float* upstream, size_t n, float* dmu, float* dlogvar, hipStream_t s) {
    hipLaunchKernelGGL(k_kld_bwd, dim3(ew_blocks(n)), dim3(LOSS_BLOCK), 0, s, mu, logvar, upstream, (float)(1.0 / (double)n), n, dmu,
                       dlogvar);
}

void launch_reparam_fwd(const float* mu, const float* logvar, const float* eps, size_t n, float* z, hipStream_t s) {
    hipLaunchKernelGGL(k_reparam_fwd, dim3(ew_blocks(n)), dim3(LOSS_BLOCK), 0, s, mu, logvar, eps, n, z);
}

void launch_reparam_bwd(const float* dz, const float* logvar, const float* eps, size_t n, float* dmu, float* dlogvar, hipStream_t s) {
    hipLaunchKernelGGL(k_reparam_bwd, dim3(ew_blocks(n)), dim3(LOSS_BLOCK), 0, s, dz, logvar, eps, n, dmu, dlogvar);
}

// ---------------------------------------------------------------------------------------------------------
// Test-time metric of the reference (ico_utils.py:26-44, mode 'point2mesh'; upstream: kaolin 0.9.1's CUDA extension):
// squared distance of every point to the closest point of a triangle mesh.  One thread per point; the triangles of the
// point's sample go through LDS in tiles of 256 (each thread stages one triangle's three vertices), so every vertex is
// fetched once per block.  Region classification of Ericson, Real-Time Collision Detection 5.1.5; ties keep the lowest face.
// kind: 0 interior, 1 / 2 / 3 vertex 0 / 1 / 2, 4 / 5 / 6 edge (0,1) / (1,2) / (2,0).
// ---------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float tri_dist2(f3 p, f3 a, f3 b, f3 c, int& kind) {
    const f3 ab = b - a, ac = c - a, ap = p - a;
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    f3 q;
    if (d1 <= 0.f && d2 <= 0.f) { kind = 1; q = a; }
    else {
        const f3 bp = p - b;
        const float d3 = dot(ab, bp), d4 = dot(ac, bp);
        if (d3 >= 0.f && d4 <= d3) { kind = 2; q = b; }
        else {
            const f3 cp = p - c;
            const float d5 = dot(ab, cp), d6 = dot(ac, cp);
            if (d6 >= 0.f && d5 <= d6) { kind = 3; q = c; }
            else {
                const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
                if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
                    const float den = d1 - d3, t = den != 0.f ? d1 / den : 0.f;
                    kind = 4; q = a + f3{ab.x * t, ab.y * t, ab.z * t};
                } else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
                    const float den = d2 - d6, t = den != 0.f ? d2 / den : 0.f;
                    kind = 6; q = a + f3{ac.x * t, ac.y * t, ac.z * t};
                } else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
                    const float den = (d4 - d3) + (d5 - d6), t = den != 0.f ? (d4 - d3) / den : 0.f;
                    const f3 bc = c - b;
                    kind = 5; q = b + f3{bc.x * t, bc.y * t, bc.z * t};
                } else {
                    const float den = va + vb + vc, inv = den != 0.f ? 1.f / den : 0.f;
                    const float v = vb * inv, w = vc * inv;
                    kind = 0; q = a + f3{ab.x * v + ac.x * w, ab.y * v + ac.y * w, ab.z * v + ac.z * w};
                }
            }
        }
    }
    const f3 d = p - q;
    return dot(d, d);
}

__global__ __launch_bounds__(LOSS_BLOCK) void k_point_to_mesh(const float* __restrict__ pts, const float* __restrict__ vts,
                                                               const int32_t* __restrict__ faces, int P, int V, int F,
                                                               float* __restrict__ dist, int32_t* __restrict__ face,
                                                               int32_t* __restrict__ kind) {
    __shared__ float tri[LOSS_BLOCK][9];
    const int b = blockIdx.y, i = blockIdx.x * LOSS_BLOCK + threadIdx.x;
    const bool live = i < P;
    f3 p = {0.f, 0.f, 0.f};
    if (live) { const float* q = pts + ((size_t)b * P + i) * 3; p = {q[0], q[1], q[2]}; }
    float best = 3.4e38f;
    int best_f = 0, best_k = 0;
    for (int f0 = 0; f0 < F; f0 += LOSS_BLOCK) {
        __syncthreads();
        const int f = f0 + threadIdx.x;
        if (f < F) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int v = faces[(size_t)f * 3 + k];
                const float* q = vts + ((size_t)b * V + v) * 3;
                tri[threadIdx.x][3 * k] = q[0]; tri[threadIdx.x][3 * k + 1] = q[1]; tri[threadIdx.x][3 * k + 2] = q[2];
            }
        }
        __syncthreads();
        const int nf = min(LOSS_BLOCK, F - f0);
        if (live)
            for (int j = 0; j < nf; ++j) {
                int k;
                const float d2 = tri_dist2(p, f3{tri[j][0], tri[j][1], tri[j][2]}, f3{tri[j][3], tri[j][4], tri[j][5]},
                                           f3{tri[j][6], tri[j][7], tri[j][8]}, k);
                if (d2 < best) { best = d2; best_f = f0 + j; best_k = k; }
            }
    }
    if (live) {
        dist[(size_t)b * P + i] = best;
        face[(size_t)b * P + i] = best_f;
        kind[(size_t)b * P + i] = best_k;
    }
}

}  // namespace

void launch_point_to_mesh(const float* pts, const float* vts, const int32_t* faces, int B, int P, int V, int F, float* dist,
                          int32_t* face, int32_t* kind, hipStream_t s) {
    hipLaunchKernelGGL(k_point_to_mesh, dim3((P + LOSS_BLOCK - 1) / LOSS_BLOCK, B), dim3(LOSS_BLOCK), 0, s, pts, vts, faces, P, V, F,
                       dist, face, kind);
}

}  // namespace icn
