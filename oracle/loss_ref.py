"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the training losses.

Follows /root/reference/losses.py:47-82 (Point2Point_Loss.forward), :105 (KLD), :137-142 (P2PKLD), with the
absent mesh helpers restated as: vertex normals = /root/reference/generate.py:20-43 (mesh_vertexnormals,
area-weighted), Laplacian = uniform umbrella mean(1-ring) - v (upstream unpinned; `lap_mode` restates the other
conventions the product offers: bit 0 = v - mean, bit 1 = times the valence, i.e. sum(ring) - k v).  float64 throughout.
"""
import numpy as np

from .ico_ref import faces_from_lattice


def grid_to_vertices(x, r):
    """(B, C, 5n, 2n) -> (B, N, C); poles = mean of px[c*n, 0] / px[(c+1)*n-1, 2n-1]  (losses.py:23-31,49-51)."""
    n = 2 ** r
    B, C = x.shape[:2]
    top = x[:, :, np.arange(5) * n, 0].mean(-1)
    bot = x[:, :, np.arange(1, 6) * n - 1, -1].mean(-1)
    v = np.concatenate([x.reshape(B, C, -1), top[:, :, None], bot[:, :, None]], axis=2)
    return v.transpose(0, 2, 1)


def vertex_normals(v, f, eps=1e-10):                      # generate.py:20-43, one mesh
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], axis=1)
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    return vn / np.clip(np.sqrt((vn ** 2).sum(1)), eps, None)[:, None]


def laplacian(v, f, lap_mode=0):
    n = v.shape[0]
    acc, deg = np.zeros_like(v), np.zeros(n)
    seen = set()
    for a, b, c in f:
        for u, w in ((a, b), (b, c), (c, a)):
            if (u, w) not in seen:
                seen.add((u, w)); seen.add((w, u))
                acc[u] += v[w]; acc[w] += v[u]
                deg[u] += 1; deg[w] += 1
    lap = acc / deg[:, None] - v
    if lap_mode & 2:
        lap = lap * deg[:, None]
    return -lap if lap_mode & 1 else lap


def p2p_terms(pred, target, r, lap_mode=0):
    """pred (B,3,5n,2n), target (B,9,N) -> (mse_pos, mean(1-cos), mse_lap)  (losses.py:66-80)."""
    pred, target = np.asarray(pred, np.float64), np.asarray(target, np.float64)
    f = faces_from_lattice(r)
    v = grid_to_vertices(pred, r)
    t = target.transpose(0, 2, 1)
    cos, lap_err = [], []
    for b in range(v.shape[0]):
        nrm, tn = vertex_normals(v[b], f), t[b, :, 3:6]
        den = np.maximum(np.linalg.norm(nrm, axis=1) * np.linalg.norm(tn, axis=1), 1e-8)   # CosineSimilarity eps
        cos.append(1 - (nrm * tn).sum(1) / den)
        lap_err.append((laplacian(v[b], f, lap_mode) - t[b, :, 6:9]) ** 2)
    return ((v - t[:, :, :3]) ** 2).mean(), np.mean(cos), np.mean(lap_err)


def p2p_pos_grad(pred, target, r):
    """d mse_pos / d pred, (B,3,5n,2n): 2 (v - t) / (B N 3) at every pixel; a pole is the mean of its 5 corner pixels
    px[c*n, 0] / px[(c+1)*n-1, 2n-1] (losses.py:23-31,49-51), so each of them also receives a fifth of the pole's term."""
    pred, target = np.asarray(pred, np.float64), np.asarray(target, np.float64)
    n = 2 ** r
    B = pred.shape[0]
    v = grid_to_vertices(pred, r)                                   # (B, N, 3)
    d = 2.0 * (v - target.transpose(0, 2, 1)[:, :, :3]) / v.size    # (B, N, 3)
    g = d[:, :-2].transpose(0, 2, 1).reshape(B, 3, 5 * n, 2 * n).copy()
    for c in range(5):
        g[:, :, c * n, 0] += d[:, -2] / 5.0
        g[:, :, (c + 1) * n - 1, 2 * n - 1] += d[:, -1] / 5.0
    return g


def p2p_grad(pred, target, r, f_pos, f_nor, f_lap, lap_mode=0):
    """d p2p_loss / d pred, (B,3,5n,2n), analytic (float64; checked against finite differences of p2p_loss in
    tests/test_oracle_properties.py).  Generic inputs only: the eps clamps of the normalisations are assumed inactive.
      position : 2 (v - a) / (3 B N)
      Laplacian: lap_i = c_i (mean(ring_i) - v_i) with c_i = +-1 or +-k_i (lap_mode), e_i = lap_i - l_i, k_i the valence:
                 2 (sum_{i in ring(j)} c_i e_i / k_i - c_j e_j) / (3 B N)
      normal   : w_i = sum of the normals n_f of the faces at i, u = w / |w|, c = u . t / |t|;
                 h_i = d(1 - c_i) / d w_i = -(t^ - (u . t^) u) / |w_i|;  a face (a, b, c) with n = (b - a) x (c - a) passes
                 G = h_a + h_b + h_c back as  d/da = G x (c - b),  d/db = G x (a - c),  d/dc = G x (b - a);  all / (B N)
    then vertex -> grid: pixels directly, each pole a fifth to each of its 5 corner pixels (losses.py:23-31,49-51)."""
    pred, target = np.asarray(pred, np.float64), np.asarray(target, np.float64)
    f = faces_from_lattice(r)
    n = 2 ** r
    B = pred.shape[0]
    v = grid_to_vertices(pred, r)
    t = target.transpose(0, 2, 1)
    N = v.shape[1]
    dv = f_pos * 2.0 * (v - t[:, :, :3]) / (3.0 * B * N)
    ring = [set() for _ in range(N)]
    for a, b, c in f:
        ring[a].update((b, c)); ring[b].update((a, c)); ring[c].update((a, b))
    for bi in range(B):
        e = laplacian(v[bi], f, lap_mode) - t[bi, :, 6:9]
        k = np.array([len(s_) for s_ in ring], np.float64)
        cf = (k if lap_mode & 2 else np.ones_like(k)) * (-1.0 if lap_mode & 1 else 1.0)
        ek = e * (cf / k)[:, None]
        gl = -e * cf[:, None]
        for j in range(N):
            for i in ring[j]:
                gl[j] += ek[i]
        dv[bi] += f_lap * 2.0 * gl / (3.0 * B * N)
        fn = np.cross(v[bi][f[:, 1]] - v[bi][f[:, 0]], v[bi][f[:, 2]] - v[bi][f[:, 0]], axis=1)
        w = np.zeros_like(v[bi])
        for kk in range(3):
            np.add.at(w, f[:, kk], fn)
        wl = np.linalg.norm(w, axis=1, keepdims=True)
        u = w / wl
        th = t[bi, :, 3:6] / np.linalg.norm(t[bi, :, 3:6], axis=1, keepdims=True)
        h = -(th - (u * th).sum(1, keepdims=True) * u) / wl
        G = h[f[:, 0]] + h[f[:, 1]] + h[f[:, 2]]
        va, vb, vc = v[bi][f[:, 0]], v[bi][f[:, 1]], v[bi][f[:, 2]]
        gn = np.zeros_like(v[bi])
        np.add.at(gn, f[:, 0], np.cross(G, vc - vb))
        np.add.at(gn, f[:, 1], np.cross(G, va - vc))
        np.add.at(gn, f[:, 2], np.cross(G, vb - va))
        dv[bi] += f_nor * gn / (B * N)
    g = dv[:, :-2].transpose(0, 2, 1).reshape(B, 3, 5 * n, 2 * n).copy()
    for c in range(5):
        g[:, :, c * n, 0] += dv[:, -2] / 5.0
        g[:, :, (c + 1) * n - 1, 2 * n - 1] += dv[:, -1] / 5.0
    return g


def p2p_loss(pred, target, r, f_pos, f_nor, f_lap, lap_mode=0):
    a, b, c = p2p_terms(pred, target, r, lap_mode)
    return f_pos * a + f_nor * b + f_lap * c


def kld(mu, logvar):                                      # losses.py:105
    mu = np.asarray(mu, np.float64).reshape(mu.shape[0], -1)
    lv = np.asarray(logvar, np.float64).reshape(logvar.shape[0], -1)
    return np.mean(-0.5 * np.mean(1 + lv - mu ** 2 - np.exp(lv), axis=1))


def kld_grad(mu, logvar):
    """d kld / d mu = mu / n,  d kld / d logvar = 0.5 (exp(logvar) - 1) / n,  n = number of elements."""
    mu, lv = np.asarray(mu, np.float64), np.asarray(logvar, np.float64)
    return mu / mu.size, 0.5 * (np.exp(lv) - 1.0) / lv.size
