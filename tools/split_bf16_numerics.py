"""Numerics of the bf16 splits of an fp32 contraction, on the CPU (developer tool).

a = a1 + a2 + a3 with every piece a bf16 (8 significand bits); the fp32 product a*b is replaced by bf16 MFMA products
a_i*b_j (exact in fp32: 8 + 8 bits), accumulated in fp32.  Forms:
    3 products   a1b1 + a1b2 + a2b1                      (two-way split; HISTORY.md 4.3, rejected in round 2)
    6 products   + a1b3 + a2b2 + a3b1                    (three-way split, terms down to 2^-16)
    8 products   + a2b3 + a3b2                           (terms down to 2^-24)
Emulated arithmetic: the exact kernel is the fmaf chain of v_mfma_f32_32x32x2_f32 (one rounding per k); the bf16 MFMA
(32x32x16) adds a block of 16 exact products to the accumulator -- 'block' sums the block exactly and rounds once, 'chain'
rounds after every product (pessimistic).  Errors are relative L2 against float64.
"""
import sys

import numpy as np


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_trunc(x):
    u = x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def split3(x, rnd):
    x = x.astype(np.float32)
    p1 = rnd(x)
    r = (x - p1).astype(np.float32)          # exact in fp32
    p2 = rnd(r)
    p3 = rnd((r - p2).astype(np.float32))
    return p1, p2, p3


def f32(x):
    return x.astype(np.float32).astype(np.float64)


def gemm_fmaf_chain(a, b):
    """v_mfma_f32_32x32x2_f32: acc = fl32(acc + a_k*b_k), k ascending."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float64)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    for k in range(a.shape[1]):
        acc = f32(acc + np.outer(a64[:, k], b64[k]))
    return acc


def gemm_split(a, b, pairs, rnd, mode='block', small_first=True, kb=16):
    """bf16 MFMA emulation: per 16-wide k block and per (i, j) piece pair one MFMA into ONE fp32 accumulator."""
    ap = [p.astype(np.float64) for p in split3(a, rnd)]
    bp = [p.astype(np.float64) for p in split3(b, rnd)]
    order = sorted(pairs, key=lambda ij: -(ij[0] + ij[1])) if small_first else list(pairs)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float64)
    for k0 in range(0, a.shape[1], kb):
        for (i, j) in order:
            if mode == 'block':
                acc = f32(acc + ap[i][:, k0:k0 + kb] @ bp[j][k0:k0 + kb])
            else:
                for k in range(k0, min(k0 + kb, a.shape[1])):
                    acc = f32(acc + np.outer(ap[i][:, k], bp[j][k]))
    return acc


P3 = [(0, 0), (0, 1), (1, 0)]
P6 = P3 + [(0, 2), (1, 1), (2, 0)]
P8 = P6 + [(1, 2), (2, 1)]
P9 = P8 + [(2, 2)]


def rel(x, ref):
    return float(np.linalg.norm(x - ref) / np.linalg.norm(ref))


def one_gemm(rng, M=96, N=96, K=7 * 256):
    a = rng.standard_normal((M, K)).astype(np.float32)
    a = np.maximum(a, 0) * 1.3                       # post-ReLU activations
    b = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    rows = [('exact fp32 (fmaf chain, 32x32x2_f32)', gemm_fmaf_chain(a, b))]
    for name, pairs in (('3 products', P3), ('6 products', P6), ('8 products', P8), ('9 products', P9)):
        for rn, rnd in (('rne', bf16_rne), ('trunc', bf16_trunc)):
            for mode in ('block', 'chain'):
                if mode == 'chain' and name in ('8 products', '9 products'):
                    continue
                rows.append(('%s  %-5s  %s' % (name, rn, mode), gemm_split(a, b, pairs, rnd, mode)))
    rows.append(('6 products  rne    block, large terms first', gemm_split(a, b, P6, bf16_rne, 'block', small_first=False)))
    return [(n, rel(v, ref)) for n, v in rows]


def surrogate(rng, layers=13, C=256, M=512, K7=7):
    """13 x (matmul over K = 7*C, batch-norm over rows, ReLU): the network-depth surrogate of HISTORY.md 4.3."""
    x0 = rng.standard_normal((M, C)).astype(np.float32)
    ws = [(rng.standard_normal((K7 * C, C)) / np.sqrt(K7 * C)).astype(np.float32) for _ in range(layers)]
    shifts = [rng.integers(0, M, size=K7) for _ in range(layers)]

    def run(mm, dt):
        x = x0.astype(dt)
        for w, sh in zip(ws, shifts):
            a = np.concatenate([np.roll(x, int(s), axis=0) for s in sh], axis=1)     # 7 shifted row sets = the gather
            y = mm(a, w).astype(dt)
            y = (y - y.mean(0)) / np.sqrt(y.var(0) + dt(1e-5))
            x = np.maximum(y, 0).astype(dt)
        return x.astype(np.float64)

    ref = run(lambda a, w: a.astype(np.float64) @ w.astype(np.float64), np.float64)
    out = [('exact fp32 (fmaf chain)', rel(run(lambda a, w: gemm_fmaf_chain(a, w), np.float32), ref))]
    for name, pairs in (('3 products rne', P3), ('6 products rne', P6), ('8 products rne', P8)):
        out.append((name + ' block', rel(run(lambda a, w: gemm_split(a, w, pairs, bf16_rne, 'block'), np.float32), ref)))
    out.append(('6 products trunc block', rel(run(lambda a, w: gemm_split(a, w, P6, bf16_trunc, 'block'), np.float32), ref)))
    out.append(('6 products rne chain', rel(run(lambda a, w: gemm_split(a, w, P6, bf16_rne, 'chain'), np.float32), ref)))
    return out


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    print('one GEMM, K = 7*256 = 1792, M = N = 96; relative L2 error vs float64')
    for n, e in one_gemm(rng):
        print('  %-48s %.3e' % (n, e))
    if len(sys.argv) > 1 and sys.argv[1] == 'gemm':
        sys.exit(0)
    print('13-layer matmul + batch-norm + ReLU surrogate (C = 256, 512 rows); relative L2 error of the last layer vs float64')
    for n, e in surrogate(rng):
        print('  %-48s %.3e' % (n, e))
