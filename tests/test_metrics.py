"""Point-to-mesh distance (reference ico_utils.py:26-44 via kaolin 0.9.1, absent): torch formulation (CPU) and HIP kernel
(GPU) against the numpy oracle (oracle/metrics_ref.py: plane projection + edge segments, a different method)."""
import numpy as np
import pytest
import torch

from geniconet_amd import geometry, metrics
from oracle import metrics_ref


def _case(seed, P=200, r=1):
    g = np.random.default_rng(seed)
    v, f = geometry.get_icosahedral_grid(r)
    v = v * (1 + 0.2 * g.standard_normal((len(v), 1)))                     # a bumpy closed mesh in the ico topology
    pts = g.standard_normal((P, 3)) * 0.8
    pts[:20] = v[:20]                                                        # points ON vertices (distance 0, vertex region)
    pts[20:40] = 0.5 * (v[f[:20, 0]] + v[f[:20, 1]])                        # on edges
    pts[40:60] = (v[f[:20, 0]] + v[f[:20, 1]] + v[f[:20, 2]]) / 3           # inside faces
    return pts.astype(np.float32), v.astype(np.float32), f


def test_torch_formulation_matches_the_oracle():
    pts, v, f = _case(1)
    d2, face, kind = metrics.point_to_mesh_distance(torch.from_numpy(pts)[None], torch.from_numpy(v)[None], torch.from_numpy(f))
    want, _ = metrics_ref.point_to_mesh_distance(pts, v, f)
    np.testing.assert_allclose(d2[0].numpy(), want, rtol=2e-4, atol=1e-9)
    assert float(d2[0, :60].max()) < 1e-10 and set(kind[0, 40:60].tolist()) <= {0, 4, 5, 6}
    assert metrics.compute_distance(torch.from_numpy(pts), torch.from_numpy(v), torch.from_numpy(f)) == pytest.approx(want.mean(), rel=1e-4)
    assert metrics.compute_distance(torch.from_numpy(pts), torch.from_numpy(v), torch.from_numpy(f), mode='chamfer') is None


@pytest.mark.gpu
def test_hip_kernel_matches_the_oracle_and_the_torch_formulation():
    for seed, P, r in ((2, 700, 2), (3, 1000, 1)):
        pts, v, f = _case(seed, P, r)
        tp, tv, tf = torch.from_numpy(pts)[None].repeat(2, 1, 1), torch.from_numpy(v)[None].repeat(2, 1, 1), torch.from_numpy(f)
        tv[1] *= 1.1                                                          # second sample: another mesh
        d2, face, kind = metrics.point_to_mesh_distance(tp.cuda(), tv.cuda(), tf.cuda())
        assert d2.is_cuda and d2.shape == (2, P)
        for b in range(2):
            want, _ = metrics_ref.point_to_mesh_distance(pts, tv[b].numpy(), f)
            np.testing.assert_allclose(d2[b].cpu().numpy(), want, rtol=2e-4, atol=1e-9)
        c2, cface, ckind = metrics._torch_point_to_mesh(tp, tv, tf)
        np.testing.assert_allclose(d2.cpu().numpy(), c2.numpy(), rtol=1e-4, atol=1e-9)
        # the closest FACE is unique only where the closest point is interior to it (an edge / a vertex is shared by several
        # faces at the same distance; 60 of the points were put on such features on purpose)
        interior = (kind.cpu() == 0) & (ckind == 0)
        assert float(interior.float().mean()) > 0.3
        # (... and where two faces are equally near to fp32 rounding either may win: the distances above agree to 1e-4 in every
        #  case; one point in ~500 flipped when the metric kernel lost its packed fp32 instructions in round 6)
        assert bool((face.cpu()[interior] == cface[interior]).float().mean() > 0.995)
        assert bool(((kind.cpu() == 0) == (ckind == 0)).float().mean() > 0.99)


@pytest.mark.gpu
def test_full_size_metric_of_a_mesh_against_itself_and_a_shifted_copy():
    """I5: 10242 points against 20480 triangles, batch 4: a mesh is at distance 0 from itself; moved by eps along x it is at
    most eps away and the mean squared distance is below eps^2."""
    from geniconet_amd import data
    from geniconet_amd.losses import grid_to_vertices
    R = 5
    x, _ = data.synthetic_batch(4, R, seed=2, device='cuda')
    v = grid_to_vertices(x, R)
    f = torch.from_numpy(geometry.get_ico_faces(R)).cuda()
    d2, _, _ = metrics.point_to_mesh_distance(v, v, f)
    assert float(d2.max()) < 1e-10
    eps = 1e-2
    shifted = v.clone()
    shifted[..., 0] += eps
    d2, _, _ = metrics.point_to_mesh_distance(shifted, v, f)
    assert float(d2.max()) <= eps * eps * (1 + 1e-4) and 0 < float(d2.mean()) < eps * eps


def test_face_indices_outside_the_vertex_list_raise_an_index_error():
    """A 1-based or mismatched face list: IndexError on the host, as kaolin / the torch formulation would raise,
    instead of an unchecked device read."""
    import torch
    from geniconet_amd.metrics import point_to_mesh_distance
    pts, vts = torch.rand(1, 5, 3), torch.rand(1, 4, 3)
    with pytest.raises(IndexError):
        point_to_mesh_distance(pts, vts, torch.tensor([[1, 2, 4]]))      # 1-based
    with pytest.raises(IndexError):
        point_to_mesh_distance(pts, vts, torch.tensor([[-1, 2, 3]]))
    d, f, k = point_to_mesh_distance(pts, vts, torch.tensor([[0, 1, 2], [1, 2, 3]]))
    assert d.shape == (1, 5)
