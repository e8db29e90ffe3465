"""Product's C++ index tables (seam affine maps, icn_geometry.cpp) == oracle's tables (pad-slice copies,
oracle/ico_ref.py): two independent derivations of SURVEY App. A must agree entry for entry."""
import numpy as np
import pytest
import scipy.sparse as sp

from geniconet_amd import _lib
from oracle import ico_ref


def corner_pixel(n, k, c):
    return (c * n) * 2 * n if k == 0 else ((c + 1) * n - 1) * 2 * n + 2 * n - 1


def corner_pixels(n, k):
    return np.array([corner_pixel(n, k, c) for c in range(5)])


def code_matrix(codes, rows, n_src, shape):
    """Sparse matrix of a list of (row, code) pairs: code >= 0 a pixel (weight 1), -2 - k the mean of the 5 corner pixels of
    pole k (5 x 0.2), -1 nothing.  n_src is the chart size n of the level the codes point into.  Duplicates add up."""
    codes, rows = np.asarray(codes).ravel(), np.asarray(rows).ravel()
    px = codes >= 0
    R, C, V = [rows[px]], [codes[px]], [np.ones(int(px.sum()))]
    for k in (0, 1):
        m = codes == -2 - k
        R.append(np.repeat(rows[m], 5))
        C.append(np.tile(corner_pixels(n_src, k), int(m.sum())))
        V.append(np.full(5 * int(m.sum()), 0.2))
    return sp.coo_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(C))), shape=shape).tocsr()


@pytest.mark.parametrize('r', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('stride', [1, 2])
def test_conv_forward_table(r, stride):
    if stride == 2 and r == 0:
        pytest.skip('stride 2 needs r >= 1')
    P = 10 * 4 ** r
    ref = ico_ref.tap_table(r, stride)
    avg = _lib.table_conv_fwd(r, stride, 'average')
    zer = _lib.table_conv_fwd(r, stride, 'zeros')
    assert (avg == np.where(ref >= P, -2 - (ref - P), ref)).all()
    assert (zer == np.where(ref >= P, -1, ref)).all()


@pytest.mark.parametrize('r', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('stride', [1, 2])
@pytest.mark.parametrize('mode', ['average', 'zeros'])
def test_conv_backward_table_is_the_transpose(r, stride, mode):
    """G_t (P_out x P_in, pole mean folded in) built from the forward table, transposed, equals the matrix the backward
    table describes -- as sparse matrices, so that the levels production runs (r = 5, and 6 for the I6 config) are covered."""
    if stride == 2 and r == 0:
        pytest.skip('stride 2 needs r >= 1')
    n = 2 ** r
    no = n // stride
    Pin, Pout = 10 * n * n, 10 * no * no
    fwd, bwd = _lib.table_conv_fwd(r, stride, mode), _lib.table_conv_bwd(r, stride, mode)
    assert bwd.shape[0] == 7 and bwd.shape[2] == Pin
    for t in range(7):
        G = code_matrix(fwd[t], np.arange(Pout), n, (Pout, Pin))
        H = code_matrix(bwd[t], np.tile(np.arange(Pin), bwd.shape[1]), no, (Pin, Pout))
        d = (H - G.T).tocoo()
        assert d.nnz == 0 or np.abs(d.data).max() < 1e-12, (t, np.abs(d.data).max())


@pytest.mark.parametrize('r', [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize('mode', ['average', 'zeros'])
def test_upsample_tables(r, mode):
    """r -> r + 1; r = 5 is the last upsample of the I6 network (5 -> 6)."""
    n = 2 ** r
    Pc, Pf = 10 * n * n, 40 * n * n
    pairs = ico_ref.upsample_table(r)
    codes = np.where(pairs >= Pc, (-2 - (pairs - Pc)) if mode == 'average' else -1, pairs)
    U = 0.5 * code_matrix(codes, np.tile(np.arange(Pf), 2), n, (Pf, Pc))

    def ell(idx, coef, shape):
        used = idx >= 0
        assert (used[:, :-1] >= used[:, 1:]).all(), 'rows must be left-packed'
        rows = np.repeat(np.arange(idx.shape[0]), idx.shape[1]).reshape(idx.shape)
        return sp.coo_matrix((coef[used].astype(np.float64), (rows[used], idx[used])), shape=shape).tocsr()

    def close(a, b):
        d = (a - b).tocoo()
        return d.nnz == 0 or np.abs(d.data).max() < 1e-7
    assert close(ell(*_lib.table_upsample(r, mode, False), (Pf, Pc)), U)
    assert close(ell(*_lib.table_upsample(r, mode, True), (Pc, Pf)), U.T)
    assert (np.sort(_lib.table_upsample_pairs(r), 0) == np.sort(pairs, 0)).all()


@pytest.mark.parametrize('r', [0, 1, 2, 3, 4, 5, 6])
def test_faces_table(r):
    assert (_lib.table_faces(r) == ico_ref.faces_from_lattice(r)).all()


def test_bad_arguments_fail_loudly():
    with pytest.raises(RuntimeError, match='stride'):
        _lib.table_conv_fwd(2, 3, 'average')
    with pytest.raises(RuntimeError, match='subdivisions'):
        _lib.table_conv_fwd(0, 2, 'average')
    with pytest.raises(ValueError, match='corner_mode'):
        _lib.table_conv_fwd(2, 1, 'mirror')
