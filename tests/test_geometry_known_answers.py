"""Implementation-independent known answers of the icosahedral chart geometry.

The reference has no tests; these are the only truths that do not depend on anyone's padding code:
V = 10*4^r + 2, F = 20*4^r, Euler characteristic 2, closed consistently oriented manifold, exactly 12
five-valent vertices, the reference's own pole-neighbour indices (ico_utils.py:13-18, losses.py:24-29).
Checked for the oracle's lattice faces AND the product's C++ faces / tap tables.
"""
from collections import Counter

import numpy as np
import pytest

from geniconet_amd import _lib, geometry
from oracle import ico_ref

LEVELS = [0, 1, 2, 3, 4, 5, 6]      # 5 = the BASELINE configs, 6 = the I6 config (config 5)


def directed_edges(f):
    return Counter((int(u), int(v)) for a, b, c in f for u, v in ((a, b), (b, c), (c, a)))


@pytest.mark.parametrize('r', LEVELS)
@pytest.mark.parametrize('source', ['oracle', 'product'])
def test_closed_oriented_manifold(r, source):
    f = ico_ref.faces_from_lattice(r) if source == 'oracle' else geometry.get_ico_faces(r)
    n = 2 ** r
    P, V = 10 * n * n, 10 * n * n + 2
    assert f.shape == (20 * 4 ** r, 3) and f.dtype == np.int64
    assert f.max() + 1 == V                              # losses.py:38 relies on this
    ed = directed_edges(f)
    assert set(ed.values()) == {1}                       # every directed edge once ...
    assert all((v, u) in ed for (u, v) in ed)            # ... and its reverse once: closed + consistently oriented
    E = len(ed) // 2
    assert V - E + len(f) == 2
    deg = Counter(u for (u, _) in ed)
    assert Counter(deg.values()) == ({5: 12, 6: V - 12} if r > 0 else {5: 12})
    ring = {}
    for (u, v) in ed:
        ring.setdefault(u, set()).add(v)
    # the reference's pole rule: N touches px[c*n, 0], S touches px[(c+1)*n-1, 2n-1]
    assert ring[P] == {(c * n) * 2 * n for c in range(5)}
    assert ring[P + 1] == {((c + 1) * n - 1) * 2 * n + 2 * n - 1 for c in range(5)}
    five = sorted(v for v in range(P) if deg[v] == 5)
    assert five == sorted([(c * n) * 2 * n + n - 1 for c in range(5)] + [(c * n) * 2 * n + 2 * n - 1 for c in range(5)])


@pytest.mark.parametrize('r', LEVELS)
@pytest.mark.parametrize('source', ['oracle', 'product'])
def test_taps_are_the_true_one_ring_in_face_order(r, source):
    """Tap t of pixel p is a true mesh neighbour, the 6 ring taps cover the whole 1-ring, and consecutive taps
    (p, tap_k, tap_k+1) are faces with the mesh's orientation (so tap ORDER is consistent across seams)."""
    n = 2 ** r
    P = 10 * n * n
    if source == 'oracle':
        tt = ico_ref.tap_table(r, 1)
    else:
        tt = _lib.table_conv_fwd(r, 1, 'average').astype(np.int64)
        tt = np.where(tt <= -2, P + (-2 - tt), tt)
    f = ico_ref.faces_from_lattice(r)
    ring = {}
    for (u, v) in directed_edges(f):
        ring.setdefault(u, set()).add(v)
    fs = set()
    for a, b, c in f.tolist():
        fs.update({(a, b, c), (b, c, a), (c, a, b)})
    assert (tt[0] == np.arange(P)).all()
    for p in range(P):
        nb = [int(tt[t, p]) for t in range(1, 7)]
        assert set(nb) == ring[p]
        dup = 0
        for k in range(6):
            a, b = nb[k], nb[(k + 1) % 6]
            if a == b:
                dup += 1
            else:
                assert (p, a, b) in fs, (r, p, k)
        assert dup == (1 if len(ring[p]) == 5 else 0)


@pytest.mark.parametrize('r', [1, 2, 3, 4, 5, 6])
def test_levels_nest(r):
    """Coarse pixel (i,j) of level r-1 is fine pixel (2i, 2j+1); every other fine vertex is the midpoint of exactly
    one coarse edge; the stride-2 table is the stride-1 table sampled at those sites (App. A.4)."""
    nc, nf = 2 ** (r - 1), 2 ** r
    Pc = 10 * nc * nc
    up = ico_ref.upsample_table(r - 1)
    coarse_edges = {frozenset(e) for e in directed_edges(ico_ref.faces_from_lattice(r - 1))}
    seen = set()
    for q in range(up.shape[1]):
        c, I, J = q // (nf * 2 * nf), (q // (2 * nf)) % nf, q % (2 * nf)
        a, b = int(up[0, q]), int(up[1, q])
        if I % 2 == 0 and J % 2 == 1:
            assert a == b == (c * nc + I // 2) * 2 * nc + (J - 1) // 2
        else:
            e = frozenset((a, b))
            assert e in coarse_edges and e not in seen
            seen.add(e)
    assert len(seen) == len(coarse_edges)               # every coarse edge (pole edges included) has its midpoint
    s1, s2 = ico_ref.tap_table(r, 1), ico_ref.tap_table(r, 2)
    sites = np.array([(c * nf + 2 * i) * 2 * nf + 2 * j + 1 for c in range(5) for i in range(nc) for j in range(2 * nc)])
    assert (s2 == s1[:, sites]).all()
    assert Pc == s2.shape[1]


@pytest.mark.parametrize('r', [0, 1, 2, 3, 4, 5, 6])
def test_grid_positions(r):
    v, f = geometry.get_icosahedral_grid(r)
    assert v.shape == (geometry.num_vertices(r), 3)
    np.testing.assert_allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-12)
    nrm = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    assert ((nrm * v[f].mean(1)).sum(1) > 0).all()       # counter-clockwise seen from outside
    e = np.linalg.norm(v[f[:, 0]] - v[f[:, 1]], axis=1)
    assert e.max() / e.min() < 1.25                       # near-uniform icosphere
    np.testing.assert_allclose(v[-2], (0, 0, 1), atol=1e-12)
    np.testing.assert_allclose(v[-1], (0, 0, -1), atol=1e-12)
