"""Composite table of conv_stride1(upsample(x)) (csrc/icn_geometry.cpp build_upconv_fwd; consumed by icn_upconv_*):
evaluated in numpy (float64) from the table alone and compared with the oracle's two separate operators
(oracle/ico_ref.py ico_upsample -> ico_conv; reference models.py:58-60).  CPU only: this pins the table before any kernel."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_l2
from geniconet_amd import _lib
from oracle import ico_ref


def eval_composite(tab, x, w, bias):
    """x (B, Cin, Pc), w (Cout, Cin, 7), bias (Cout) -> y (B, Cout, Pf) using only the table (vectorised per segment and
    virtual tap, so that the levels production runs -- coarse r = 4, and 5 for the I6 config -- finish in seconds)."""
    B, Cin, _ = x.shape
    weff = np.einsum('vt,oit->voi', tab['alpha'].astype(np.float64), w)                # (NV, Cout, Cin)
    side = np.zeros((B, Cin, max(len(tab['slot_idx']), 1)))
    for s, (idx, coef) in enumerate(zip(tab['slot_idx'], tab['slot_coef'])):
        for i, c in zip(idx, coef):
            if i >= 0:
                side[:, :, s] += float(c) * x[:, :, i]
    y = np.zeros((B, w.shape[0], tab['Pf'])) + bias[None, :, None]
    macs = 0
    for cnt, off, mask in tab['seg']:
        taps = [v for v in range(tab['code'].shape[0]) if (int(mask) >> v) & 1]
        pix = tab['pix'][off:off + cnt]
        for v in taps:
            c = tab['code'][v, off:off + cnt]
            macs += int(cnt)
            src = np.zeros((B, Cin, int(cnt)))
            px, sd = c >= 0, c <= -2
            src[:, :, px] = x[:, :, c[px]]
            src[:, :, sd] = side[:, :, -2 - c[sd]]
            y[:, :, pix] += np.einsum('oi,bip->bop', weff[v], src)
    return y, macs


@pytest.mark.parametrize('mode', ['average', 'zeros'])
@pytest.mark.parametrize('r', [0, 1, 2, 3, 4, 5])
def test_composite_table_equals_upsample_then_conv(r, mode):
    tab = _lib.table_upconv(r, mode)
    n = 2 ** r
    Pc, Pf = 10 * n * n, 40 * n * n
    assert tab['Pc'] == Pc and tab['Pf'] == Pf
    assert sorted(tab['pix'].tolist()) == list(range(Pf))                              # every fine pixel exactly once
    assert int(tab['seg'][:, 0].sum()) == Pf and list(tab['seg'][:, 1]) == list(np.cumsum(np.r_[0, tab['seg'][:-1, 0]]))
    g = torch.Generator().manual_seed(5 + r)
    B, Cin, Cout = 2, 3, 4
    x = torch.randn(B, Cin, 5 * n, 2 * n, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 7, generator=g, dtype=torch.float64)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    want = ico_ref.ico_conv(ico_ref.ico_upsample(x, r, mode), w, b, r + 1, 1, mode)
    got, macs = eval_composite(tab, x.reshape(B, Cin, Pc).numpy(), w.numpy(), b.numpy())
    assert rel_l2(got, want.reshape(B, Cout, Pf).numpy()) < 1e-6      # coefficients are stored as fp32 (0.1, 0.02 are inexact)
    assert macs <= 7 * Pf
    if r >= 2 and mode == 'average':
        irregular = int(tab['seg'][-1, 0]) if int(tab['seg'][-1, 2]) >> 19 else 0
        assert irregular <= 12 * 24, irregular                                           # a patch around each singular vertex
        regular = Pf - irregular
        assert macs == 7 * irregular + sum(int(c) * bin(int(m)).count('1') for c, _, m in tab['seg'] if not int(m) >> 19)
        # three quarters of the regular rows are edge midpoints (4 virtual taps), one quarter coarse sites (7)
        assert abs(macs / (7.0 * Pf) - 4.75 / 7.0) < 0.2 * 4.0 ** (2 - r) + 0.01, (macs, Pf, regular)


def test_virtual_taps_are_the_documented_ones():
    tab = _lib.table_upconv(3, 'average')
    a = tab['alpha']
    assert a.shape == (26, 7)
    assert np.allclose(a[19:], np.eye(7))                                              # irregular rows: the original taps
    assert np.allclose(sorted(a[:7].sum(1)), [0.5] * 6 + [4.0])                         # site class: centre 1 + 6 x 0.5; ring 0.5 each
    assert np.allclose(a[:19].sum(0), 4 * np.ones(7))                                   # every W_t is used with total weight 4 = 4 fine pixels
    masks = [int(m) for m in tab['seg'][:, 2]]
    assert masks == [0x7f, 0xf << 7, 0xf << 11, 0xf << 15, 0x7f << 19]


@pytest.mark.parametrize('mode', ['average', 'zeros'])
@pytest.mark.parametrize('r', [0, 1, 2, 3, 4, 5])
def test_aggregated_backward_table_gives_both_gradients(r, mode):
    """icn_table_upconv_bwd: g_t[s] = sum_p U[nbr_t(p), s] dy[p] turns the backward of conv(upsample(x)) into dense
    coarse-level contractions, dx[s] = sum_t W_t^T g_t[s] and dW_t = sum_s x[s]^T g_t[s] (and dbias = sum_s g_0[s] for
    'average' poles) -- checked against autograd through the oracle's two operators (float64)."""
    idx, coef = _lib.table_upconv_bwd(r, mode)
    n = 2 ** r
    Pc, Pf = 10 * n * n, 40 * n * n
    assert idx.shape[0] == 7 * Pc
    g = torch.Generator().manual_seed(9 + r)
    B, Cin, Cout = 2, 3, 4
    x = torch.randn(B, Cin, 5 * n, 2 * n, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 7, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
    y = ico_ref.ico_conv(ico_ref.ico_upsample(x, r, mode), w, b, r + 1, 1, mode)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dyf = dy.reshape(B, Cout, Pf).numpy()
    used = idx >= 0
    rows = np.repeat(np.arange(7 * Pc), idx.shape[1]).reshape(idx.shape)
    A = sp.coo_matrix((coef[used].astype(np.float64), (rows[used], idx[used])), shape=(7 * Pc, Pf)).tocsr()
    gt = (A @ dyf.reshape(B * Cout, Pf).T).T.reshape(B, Cout, Pc, 7)                   # row s * 7 + t
    wn, xn = w.detach().numpy(), x.detach().reshape(B, Cin, Pc).numpy()
    dx = np.einsum('oit,bost->bis', wn, gt)
    dw = np.einsum('bis,bost->oit', xn, gt)
    assert rel_l2(dx, x.grad.reshape(B, Cin, Pc).numpy()) < 1e-6
    assert rel_l2(dw, w.grad.numpy()) < 1e-6
    if mode == 'average':
        assert rel_l2(gt[:, :, :, 0].sum((0, 2)), b.grad.numpy()) < 1e-6
    if r >= 2:
        used = (idx >= 0).sum(1)
        assert int(np.median(used)) == 7 and idx.shape[1] <= 24
