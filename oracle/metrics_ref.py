"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference's test-time metric.

/root/reference/ico_utils.py:26-44 (computeDistance, mode 'point2mesh') calls kaolin 0.9.1's
`kaolin.metrics.trianglemesh.point_to_mesh_distance` (absent; a CUDA extension): the squared Euclidean distance of every
point to the closest point of a triangle mesh, of which the reference takes the mean.  Restated here by a method that
shares nothing with the product's Voronoi-region classification: project the point onto the triangle's plane; if the foot
lies inside (barycentric test) that is the closest point, otherwise the closest point lies on one of the three edge
segments (clamped projections).  float64, brute force over all point / triangle pairs -- small meshes only.
"""
import numpy as np


def _segment_d2(p, a, b):
    ab = b - a
    den = (ab * ab).sum(-1)
    t = np.clip(((p - a) * ab).sum(-1) / np.where(den > 0, den, 1.0), 0.0, 1.0)
    q = a + ab * t[..., None]
    return ((p - q) ** 2).sum(-1)


def point_to_mesh_distance(points, vertices, faces):
    """points (P, 3), vertices (V, 3), faces (F, 3) -> (squared distance (P,), index of a closest face (P,))."""
    p = np.asarray(points, np.float64)[:, None, :]
    v = np.asarray(vertices, np.float64)
    a, b, c = (v[np.asarray(faces)[:, k]][None] for k in range(3))
    n = np.cross(b - a, c - a)
    nn = (n * n).sum(-1)
    ok = nn > 0
    n_safe = np.where(ok[..., None], n, 1.0)
    dist_plane = ((p - a) * n_safe).sum(-1) / np.where(ok, nn, 1.0)
    foot = p - dist_plane[..., None] * n_safe
    # barycentric inside test through signed sub-areas
    inside = ok
    for u, w in ((a, b), (b, c), (c, a)):
        inside = inside & ((np.cross(w - u, foot - u) * n_safe).sum(-1) >= 0)
    d_plane = dist_plane ** 2 * nn
    d_edges = np.minimum(np.minimum(_segment_d2(p, a, b), _segment_d2(p, b, c)), _segment_d2(p, c, a))
    d2 = np.where(inside, d_plane, d_edges)
    return d2.min(1), d2.argmin(1)
