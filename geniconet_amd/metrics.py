"""Test-time metric of the reference: point-to-mesh distance (ico_utils.py:26-44 computeDistance, mode 'point2mesh'), which
upstream takes from kaolin's CUDA extension (`kaolin.metrics.trianglemesh.point_to_mesh_distance`, kaolin 0.9.1; absent here).

    dist2, face, kind = point_to_mesh_distance(points (B, P, 3), vertices (B, V, 3), faces (F, 3))
    dist2 (B, P)  squared Euclidean distance of every point to the closest point of the triangle mesh
    face  (B, P)  index of a closest triangle (the lowest index among equally close ones)
    kind  (B, P)  where on that triangle the closest point lies: 0 interior, 1 / 2 / 3 at vertex 0 / 1 / 2,
                  4 / 5 / 6 on edge (0,1) / (1,2) / (2,0)
The reference consumes `torch.mean(dist2)` (ico_utils.py:41).  ROCm fp32 tensors run on a HIP kernel (icn_point_to_mesh in
include/icn.h: one thread per point, triangles staged through LDS); anything else on the chunked torch formulation below,
which is also the kernel's second check in the tests (the first is the numpy oracle, oracle/metrics_ref.py).
"""
import torch

from . import _lib


def _closest_on_triangles(p, a, b, c):
    """Squared distance + region of the closest point on triangles (a, b, c) for points p; all (..., 3), broadcastable.
    Region tests of Ericson, Real-Time Collision Detection 5.1.5 (Voronoi regions of the triangle's features)."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    one = torch.ones_like(d1)
    tiny = torch.finfo(p.dtype).tiny

    def safe(num, den):
        return num / torch.where(den.abs() > tiny, den, one)
    # interior (default), then the edge and vertex regions in increasing priority
    den = va + vb + vc
    v, w = safe(vb, den), safe(vc, den)
    q = a + ab * v[..., None] + ac * w[..., None]
    kind = torch.zeros_like(d1, dtype=torch.int32)

    def put(mask, point, k):
        nonlocal q, kind
        q = torch.where(mask[..., None], point, q)
        kind = torch.where(mask, torch.full_like(kind, k), kind)
    t = safe(d4 - d3, (d4 - d3) + (d5 - d6))
    put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + (c - b) * t[..., None], 5)          # edge (1, 2)
    t = safe(d2, d2 - d6)
    put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + ac * t[..., None], 6)                               # edge (2, 0)
    t = safe(d1, d1 - d3)
    put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + ab * t[..., None], 4)                               # edge (0, 1)
    put((d6 >= 0) & (d5 <= d6), c.expand_as(q), 3)
    put((d3 >= 0) & (d4 <= d3), b.expand_as(q), 2)
    put((d1 <= 0) & (d2 <= 0), a.expand_as(q), 1)
    return ((p - q) ** 2).sum(-1), kind


def _torch_point_to_mesh(points, vertices, faces, chunk=256):
    B, P, _ = points.shape
    f = faces.long()
    tri = vertices[:, f]                                           # (B, F, 3, 3)
    a, b, c = tri[:, None, :, 0], tri[:, None, :, 1], tri[:, None, :, 2]
    best = points.new_full((B, P), float('inf'))
    face = torch.zeros(B, P, dtype=torch.int64, device=points.device)
    kind = torch.zeros(B, P, dtype=torch.int32, device=points.device)
    for lo in range(0, P, chunk):
        d2, k = _closest_on_triangles(points[:, lo:lo + chunk, None, :], a, b, c)      # (B, chunk, F)
        m, j = d2.min(dim=2)
        best[:, lo:lo + chunk], face[:, lo:lo + chunk] = m, j
        kind[:, lo:lo + chunk] = torch.gather(k, 2, j[..., None])[..., 0]
    return best, face, kind


_MAX_GRID_Y = 65535


def point_to_mesh_distance(pointclouds, vertices, faces):
    """See the module docstring.  `faces` may be int32 / int64, (F, 3), shared by the batch (as in kaolin 0.9.1)."""
    if pointclouds.dim() != 3 or vertices.dim() != 3 or pointclouds.shape[-1] != 3 or vertices.shape[-1] != 3:
        raise ValueError('point_to_mesh_distance: expected (B, P, 3) points and (B, V, 3) vertices')
    if pointclouds.shape[0] != vertices.shape[0] or faces.dim() != 2 or faces.shape[1] != 3:
        raise ValueError('point_to_mesh_distance: batch sizes differ or faces is not (F, 3)')
    if faces.shape[0] > 0:
        # the kernel indexes vertices[b, faces[f]] unchecked: a 1-based or mismatched face list (an OFF file read with the
        # wrong vertex set) must fail here like kaolin's / torch's IndexError, not as a device memory fault.  One host sync;
        # this is a test-time metric (ico_utils.py:26-44).
        lo, hi = int(faces.min()), int(faces.max())
        if lo < 0 or hi >= vertices.shape[1]:
            raise IndexError('point_to_mesh_distance: face indices span [%d, %d] but there are %d vertices'
                             % (lo, hi, vertices.shape[1]))
    if pointclouds.is_cuda and pointclouds.dtype == torch.float32 and vertices.dtype == torch.float32 and faces.shape[0] > 0:
        L = _lib.lib()
        B, P, _ = pointclouds.shape
        pts, vts = pointclouds.contiguous(), vertices.contiguous()
        f32 = faces.to(device=pts.device, dtype=torch.int32).contiguous()
        dist = torch.empty(B, P, dtype=torch.float32, device=pts.device)
        face = torch.empty(B, P, dtype=torch.int32, device=pts.device)
        kind = torch.empty(B, P, dtype=torch.int32, device=pts.device)
        with torch.cuda.device(pts.device):
            for b0 in range(0, B, _MAX_GRID_Y):                  # the batch index is gridDim.y (<= 65535 per launch)
                nb = min(_MAX_GRID_Y, B - b0)
                rc = L.icn_point_to_mesh(pts[b0:].data_ptr(), vts[b0:].data_ptr(), f32.data_ptr(), nb, P, vertices.shape[1],
                                         faces.shape[0], dist[b0:].data_ptr(), face[b0:].data_ptr(), kind[b0:].data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
                _lib.check(rc, 'icn_point_to_mesh')
        return dist, face.long(), kind
    return _torch_point_to_mesh(pointclouds, vertices, faces.to(pointclouds.device))


def output2vertices(subdivisions, output):
    """Grid -> vertex list incl. the two pole means (reference ico_utils.py:10-24)."""
    from .losses import grid_to_vertices
    return grid_to_vertices(output, subdivisions)


def compute_distance(outvertices, refvertices, reffaces, mode='point2mesh'):
    """ico_utils.py:26-44 computeDistance for one mesh pair ((N, 3) tensors): mean squared point-to-mesh distance of the
    output's vertices to the reference mesh, as a Python float; None for other modes (as the reference)."""
    if mode != 'point2mesh':
        return None
    dist, _, _ = point_to_mesh_distance(outvertices[None], refvertices[None], reffaces)
    return float(dist.mean())
