"""Icosahedral grid geometry of the plugin surface `icocnn.utils.ico_geometry`.

Replaces (absent upstream) `get_ico_faces` / `get_icosahedral_grid`; reference call sites:
losses.py:34-40 (faces -> int64 index buffer, `max()+1` == vertex count), run.py:144,529,
generate.py:151-152 (`ico_v, ico_f = get_icosahedral_grid(subdivision)`).

Vertex order is the reference's: row-major flatten of the (5*2^r, 2^(r+1)) grid, then the N and S poles
(ico_utils.py:20-23, losses.py:49-51).  Connectivity comes from the library's host-side chart geometry
(icn_table_faces / icn_table_upsample_pairs, include/icn.h), so Python and the HIP kernels cannot disagree.
"""
import functools

import numpy as np

from . import _lib


def num_pixels(subdivisions):
    return 10 * 4 ** subdivisions


def num_vertices(subdivisions):
    return num_pixels(subdivisions) + 2


@functools.lru_cache(maxsize=None)
def _faces(subdivisions):
    f = _lib.table_faces(subdivisions).astype(np.int64)
    f.setflags(write=False)
    return f


def get_ico_faces(subdivisions):
    """(20*4^r, 3) int64 faces, counter-clockwise seen from outside."""
    return _faces(subdivisions).copy()


@functools.lru_cache(maxsize=None)
def _grid(subdivisions):
    # level 0: the icosahedron.  Chart c holds U_c = px(c,0,0) (upper ring, lattice (0,1)) and L_c = px(c,0,1)
    # (lower ring, lattice (0,2)); charts advance clockwise seen from the north pole so that faces are CCW.
    lat = np.arctan(0.5)
    v = np.zeros((12, 3))
    for c in range(5):
        lon_u = -2.0 * np.pi * c / 5.0
        lon_l = lon_u + np.pi / 5.0
        v[2 * c] = (np.cos(lat) * np.cos(lon_u), np.cos(lat) * np.sin(lon_u), np.sin(lat))
        v[2 * c + 1] = (np.cos(lat) * np.cos(lon_l), np.cos(lat) * np.sin(lon_l), -np.sin(lat))
    v[10], v[11] = (0, 0, 1), (0, 0, -1)
    # subdivide: coarse vertices are kept, every new vertex is the normalised midpoint of one coarse edge
    for r in range(subdivisions):
        a, b = _lib.table_upsample_pairs(r)
        fine = 0.5 * (v[a] + v[b])
        fine /= np.linalg.norm(fine, axis=1, keepdims=True)
        v = np.concatenate([fine, v[-2:]], axis=0)
    v.setflags(write=False)
    return v


def get_icosahedral_grid(subdivisions):
    """(v, f): unit-sphere vertex positions (10*4^r + 2, 3) float64 and faces (20*4^r, 3) int64."""
    return _grid(subdivisions).copy(), get_ico_faces(subdivisions)


@functools.lru_cache(maxsize=None)
def vertex_neighbours(subdivisions):
    """(N, 6) int64 one-ring of every vertex (five-valent vertices repeat themselves... never: padded with -1)."""
    f = _faces(subdivisions)
    n = int(f.max()) + 1
    ring = [set() for _ in range(n)]
    for a, b, c in f:
        ring[a].update((b, c)); ring[b].update((a, c)); ring[c].update((a, b))
    out = np.full((n, 6), -1, dtype=np.int64)
    for i, s in enumerate(ring):
        out[i, :len(s)] = sorted(s)
    out.setflags(write=False)
    return out
