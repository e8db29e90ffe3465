"""Sample format of the hot path and seeded synthetic meshes.

Format (reference data.py:64-69 reader, generate.py:200-203 writer): one `.npz` per mesh with key "data",
float32 (9, 10*4^r + 2): rows 0-2 vertex xyz, 3-5 unit vertex normals, 6-8 Laplacian, columns in the grid
vertex order with the two poles last.  Network input = data[:3, :-2].reshape(3, 5*2^r, 2^(r+1)).

Synthetic meshes (there is no dataset offline): a smoothly perturbed icosphere
v = u * (1 + 0.25 * sum_k a_k sin(w_k . u + p_k)), clipped to (-0.95, 0.95) so it lies inside tanh's range.
"""
import numpy as np
import torch

from . import geometry, losses


def target_to_input(target, subdivisions):
    """(B, 9, N) -> (B, 3, 5n, 2n): positions without the poles (data.py:67-68)."""
    n = 2 ** subdivisions
    return target[:, :3, :-2].reshape(target.shape[0], 3, 5 * n, 2 * n)


def save_sample(path, data):
    """data (9, N) -> .npz with key 'data' (generate.py:203)."""
    np.savez(path, data=np.asarray(data, dtype=np.float32))


def load_sample(path, subdivisions):
    """-> (input (3,5n,2n), target (9,N)) as the reference's loadIcoFile returns them (data.py:64-69)."""
    lbl2 = np.load(path)['data']
    n = 2 ** subdivisions
    assert lbl2.shape == (9, geometry.num_vertices(subdivisions)), lbl2.shape
    return lbl2[:3, :-2].reshape(3, -1, 2 * n), lbl2


def synthetic_batch(batch, subdivisions, seed, device='cpu'):
    """Seeded (input (B,3,5n,2n), target (B,9,N)) fp32 on `device`."""
    g = torch.Generator().manual_seed(seed)
    u = torch.from_numpy(geometry.get_icosahedral_grid(subdivisions)[0]).float()          # (N, 3)
    amp = (torch.rand(batch, 3, generator=g) - 0.5) * 2.0 / 3.0                            # a_k in (-1/3, 1/3)
    freq = torch.randn(batch, 3, 3, generator=g) * 2.0                                     # w_k
    phase = torch.rand(batch, 3, generator=g) * 2 * np.pi
    u, amp, freq, phase = u.to(device), amp.to(device), freq.to(device), phase.to(device)
    wave = torch.sin(torch.einsum('nd,bkd->bkn', u, freq) + phase[:, :, None])            # (B, 3, N)
    radius = 0.75 * (1 + 0.25 * (amp[:, :, None] * wave).sum(1))                           # (B, N)
    v = (u[None] * radius[:, :, None]).clamp(-0.95, 0.95)                                  # (B, N, 3)
    faces = torch.from_numpy(geometry.get_ico_faces(subdivisions)).to(device)
    nbr = torch.from_numpy(geometry.vertex_neighbours(subdivisions).copy()).to(device)
    valid = nbr >= 0
    nbr_w = valid.float() / valid.sum(1, keepdim=True).float()
    normals = losses.compute_vertex_normals(v, faces)
    lap = losses.compute_laplacian_batch(v, nbr.clamp_min(0), nbr_w)
    target = torch.cat((v, normals, lap), dim=2).transpose(1, 2).contiguous()              # (B, 9, N)
    return target_to_input(target, subdivisions).contiguous(), target
