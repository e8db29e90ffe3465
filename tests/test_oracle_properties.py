"""Operator properties of the CPU oracle: the oracle is only trusted after these pass."""
from collections import Counter

import numpy as np
import pytest
import torch

from oracle import ico_ref

torch.manual_seed(0)


def one_ring(r):
    f = ico_ref.faces_from_lattice(r)
    ring = {}
    for a, b, c in f:
        for u, v in ((a, b), (b, c), (c, a)):
            ring.setdefault(int(u), set()).add(int(v))
    return ring


@pytest.mark.parametrize('r', [0, 1, 2, 3])
def test_conv_is_the_true_one_ring_sum(r):
    """With centre weight a and all ring weights b, y[p] = a x[p] + b sum_{q in ring(p)} x[q], plus b x[dup] at a
    five-valent pixel, where the ring comes from the FACES (no padding code) and poles are the 5-corner mean."""
    n = 2 ** r
    P = 10 * n * n
    x = torch.randn(1, 1, 5 * n, 2 * n, dtype=torch.float64)
    a, b = 0.7, -0.3
    w = torch.full((1, 1, 7), b, dtype=torch.float64)
    w[0, 0, 0] = a
    y = ico_ref.ico_conv(x, w, None, r, 1, 'average').reshape(-1).numpy()
    xv = np.concatenate([x.reshape(-1).numpy(), np.zeros(2)])
    xv[P] = np.mean([xv[(c * n) * 2 * n] for c in range(5)])
    xv[P + 1] = np.mean([xv[((c + 1) * n - 1) * 2 * n + 2 * n - 1] for c in range(5)])
    ring = one_ring(r)
    tt = ico_ref.tap_table(r, 1)
    for p in range(P):
        want = a * xv[p] + b * sum(xv[q] for q in ring[p])
        if len(ring[p]) == 5:                                   # the duplicated tap (an upstream-unknown choice)
            dup = [q for q, k in Counter(tt[1:, p].tolist()).items() if k == 2]
            assert len(dup) == 1
            want += b * xv[dup[0]]
        assert abs(y[p] - want) < 1e-12


@pytest.mark.parametrize('r,stride', [(1, 1), (2, 1), (2, 2), (3, 2)])
@pytest.mark.parametrize('mode', ['average', 'zeros'])
def test_chart_shift_equivariance(r, stride, mode):
    """Rotating the sphere by 72 degrees about the pole axis == rolling the 5 charts."""
    n = 2 ** r
    x = torch.randn(2, 3, 5 * n, 2 * n, dtype=torch.float64)
    w, b = torch.randn(4, 3, 7, dtype=torch.float64), torch.randn(4, dtype=torch.float64)
    y = ico_ref.ico_conv(x, w, b, r, stride, mode)
    ys = ico_ref.ico_conv(torch.roll(x, n, dims=2), w, b, r, stride, mode)
    torch.testing.assert_close(ys, torch.roll(y, n // stride, dims=2), atol=1e-12, rtol=0)
    u = ico_ref.ico_upsample(x, r, mode)
    us = ico_ref.ico_upsample(torch.roll(x, n, dims=2), r, mode)
    torch.testing.assert_close(us, torch.roll(u, 2 * n, dims=2), atol=1e-12, rtol=0)


@pytest.mark.parametrize('r', [1, 2, 3])
def test_stride2_is_stride1_sampled(r):
    n = 2 ** r
    x = torch.randn(2, 3, 5 * n, 2 * n, dtype=torch.float64)
    w, b = torch.randn(5, 3, 7, dtype=torch.float64), torch.randn(5, dtype=torch.float64)
    y1 = ico_ref.ico_conv(x, w, b, r, 1, 'average').reshape(2, 5, 5, n, 2 * n)
    y2 = ico_ref.ico_conv(x, w, b, r, 2, 'average').reshape(2, 5, 5, n // 2, n)
    torch.testing.assert_close(y2, y1[..., 0::2, 1::2], atol=0, rtol=0)


@pytest.mark.parametrize('r', [0, 1, 2])
def test_constant_in_constant_out(r):
    """Averaging weights (1/7 each) map a constant field to the same constant everywhere -- also at the 12
    singular vertices, because the duplicated tap keeps 7 contributions there."""
    n = 2 ** r
    x = torch.full((1, 1, 5 * n, 2 * n), 2.5, dtype=torch.float64)
    w = torch.full((1, 1, 7), 1 / 7, dtype=torch.float64)
    torch.testing.assert_close(ico_ref.ico_conv(x, w, None, r, 1, 'average'), x, atol=1e-14, rtol=0)
    torch.testing.assert_close(ico_ref.ico_upsample(x, r, 'average'),
                               torch.full((1, 1, 10 * n, 4 * n), 2.5, dtype=torch.float64), atol=1e-14, rtol=0)


@pytest.mark.parametrize('r', [0, 1, 2])
@pytest.mark.parametrize('mode', ['average', 'zeros'])
def test_upsample_keeps_coarse_sites_and_means_edges(r, mode):
    n = 2 ** r
    P = 10 * n * n
    x = torch.randn(1, 2, 5 * n, 2 * n, dtype=torch.float64)
    u = ico_ref.ico_upsample(x, r, mode).reshape(1, 2, 5, 2 * n, 4 * n)
    torch.testing.assert_close(u[..., 0::2, 1::2], x.reshape(1, 2, 5, n, 2 * n), atol=0, rtol=0)
    xv = torch.cat([x.reshape(2, -1), torch.zeros(2, 2, dtype=torch.float64)], 1)
    if mode == 'average':
        x5 = x.reshape(2, 5, n, 2 * n)
        xv[:, P], xv[:, P + 1] = x5[:, :, 0, 0].mean(1), x5[:, :, -1, -1].mean(1)
    pairs = torch.from_numpy(ico_ref.upsample_table(r))
    want = 0.5 * (xv[:, pairs[0]] + xv[:, pairs[1]])
    torch.testing.assert_close(u.reshape(2, -1), want, atol=1e-14, rtol=0)


@pytest.mark.parametrize('r,stride', [(1, 1), (2, 2)])
@pytest.mark.parametrize('mode', ['average', 'zeros'])
def test_gradcheck(r, stride, mode):
    n = 2 ** r
    x = torch.randn(1, 2, 5 * n, 2 * n, dtype=torch.float64, requires_grad=True)
    w = torch.randn(2, 2, 7, dtype=torch.float64, requires_grad=True)
    b = torch.randn(2, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda *a: ico_ref.ico_conv(*a, r, stride, mode), (x, w, b))
    assert torch.autograd.gradcheck(lambda a: ico_ref.ico_upsample(a, r, mode), (x,))


def test_tap_order_matches_lattice_offsets_in_chart_interior():
    """Interior pixels: tap t reads pixel (i+da, j+db) with the documented offsets (include/icn.h)."""
    r, n = 3, 8
    tt = ico_ref.tap_table(r, 1)
    for c in range(5):
        for i in range(1, n - 1):
            for j in range(1, 2 * n - 1):
                p = (c * n + i) * 2 * n + j
                for t, (da, db) in enumerate(ico_ref.TAPS):
                    assert tt[t, p] == (c * n + i + da) * 2 * n + j + db


@pytest.mark.parametrize('lap_mode', [0, 1, 2, 3])
def test_loss_gradient_oracle_is_pinned_by_finite_differences(lap_mode):
    """oracle/loss_ref.p2p_grad (analytic; the checker of the HIP loss backward) against central differences of
    p2p_loss, VAE factors (reference run.py:694-696), r = 1: 42 vertices, ten of them pole corners; every Laplacian
    convention the product offers."""
    import numpy as np
    from oracle import loss_ref
    rng = np.random.default_rng(4)
    r, B, fac = 1, 1, (0.6, 0.2, 0.2)
    n = 2 ** r
    pred = rng.standard_normal((B, 3, 5 * n, 2 * n))
    tgt = rng.standard_normal((B, 9, 10 * n * n + 2))
    g = loss_ref.p2p_grad(pred, tgt, r, *fac, lap_mode)
    num = np.zeros_like(pred)
    eps = 1e-6
    for idx in np.ndindex(pred.shape):
        p, q = pred.copy(), pred.copy()
        p[idx] += eps
        q[idx] -= eps
        num[idx] = (loss_ref.p2p_loss(p, tgt, r, *fac, lap_mode) - loss_ref.p2p_loss(q, tgt, r, *fac, lap_mode)) / (2 * eps)
    assert np.linalg.norm(g - num) <= 1e-6 * np.linalg.norm(num)
    mu, lv = rng.standard_normal((2, 4, 5, 2)), 0.3 * rng.standard_normal((2, 4, 5, 2))
    gm, gl = loss_ref.kld_grad(mu, lv)
    for arr, grad, which in ((mu, gm, 0), (lv, gl, 1)):
        idx = (1, 2, 3, 1)
        a, b = arr.copy(), arr.copy()
        a[idx] += eps
        b[idx] -= eps
        fd = (loss_ref.kld(*((a, lv) if which == 0 else (mu, a))) - loss_ref.kld(*((b, lv) if which == 0 else (mu, b)))) / (2 * eps)
        assert abs(fd - grad[idx]) <= 1e-6 * abs(fd) + 1e-12
    assert np.allclose(loss_ref.p2p_grad(pred, tgt, r, 1.0, 0.0, 0.0), loss_ref.p2p_pos_grad(pred, tgt, r), rtol=1e-12, atol=0)
