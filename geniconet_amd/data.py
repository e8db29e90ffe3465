"""Sample format of the hot path and seeded synthetic meshes.

Format (reference data.py:64-69 reader, generate.py:200-203 writer): one `.npz` per mesh with key "data",
float32 (9, 10*4^r + 2): rows 0-2 vertex xyz, 3-5 unit vertex normals, 6-8 Laplacian, columns in the grid
vertex order with the two poles last.  Network input = data[:3, :-2].reshape(3, 5*2^r, 2^(r+1)).

Synthetic meshes (there is no dataset offline): a smoothly perturbed icosphere
v = u * (1 + 0.25 * sum_k a_k sin(w_k . u + p_k)), clipped to (-0.95, 0.95) so it lies inside tanh's range.
"""
import os
import re

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


def laplacian_rows(positions, subdivisions, laplacian='mean-v'):
    """(B, 3, N) vertex positions -> (B, 3, N) Laplacian rows under the convention `laplacian` (losses.LAPLACIAN_MODES)."""
    nbr = torch.from_numpy(geometry.vertex_neighbours(subdivisions).copy()).to(positions.device)
    valid = nbr >= 0
    nbr_w = valid.to(positions.dtype) / valid.sum(1, keepdim=True).to(positions.dtype)
    return losses.compute_laplacian_batch(positions.transpose(1, 2), nbr.clamp_min(0), nbr_w, laplacian).transpose(1, 2)


def detect_laplacian_convention(target, subdivisions, tol=1e-3):
    """Which convention of losses.LAPLACIAN_MODES reproduces rows 6:9 of `target` (B, 9, N) from its rows 0:3?  Returns
    (name or None, {name: relative L2 error}).  The reference's datasets carry Laplacians written by the absent
    mesh.utils.compute_laplacian (generate.py:197): this is how its sign / normalisation is recovered from the data."""
    t = torch.as_tensor(target, dtype=torch.float64)
    if t.dim() == 2:
        t = t[None]
    want = t[:, 6:9]
    errs = {}
    for name in losses.LAPLACIAN_MODES:
        got = laplacian_rows(t[:, 0:3], subdivisions, name)
        errs[name] = float((got - want).norm() / want.norm().clamp_min(1e-30))
    best = min(errs, key=errs.get)
    return (best if errs[best] < tol else None), errs


def synthetic_batch(batch, subdivisions, seed, device='cpu', laplacian='mean-v'):
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
    lap = losses.compute_laplacian_batch(v, nbr.clamp_min(0), nbr_w, laplacian)
    target = torch.cat((v, normals, lap), dim=2).transpose(1, 2).contiguous()              # (B, 9, N)
    return target_to_input(target, subdivisions).contiguous(), target


def _natural_key(name):
    """Order of natsort.natsorted, which the reference lists its files with (data.py:18)."""
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', name)]


def list_samples(directory, ext='.npz'):
    """Files of one data directory in the reference's order (data.py:7-21, dataPthLvl 1)."""
    return [os.path.join(directory, f) for f in sorted(os.listdir(directory), key=_natural_key) if f.endswith(ext)]


class IcoDataset:
    """Every sample of a directory resident on the device as ONE (N, 9, V) tensor.

    The reference preloads its `.npz` files into host RAM (data.py:72-80, loadIcoFile :64-69) and feeds batches through a
    DataLoader with 2 x cpu_count workers and a host-to-device copy per step (run.py:52,70-75,241-242,714).  An MI355X holds
    288 GB: ModelNet10 at I5 is ~2 GB, so the samples are uploaded once and a batch is an index_select on the device -- no
    workers, no per-step copies.  Batching follows the DataLoader the reference builds: `batch_size` samples, optional
    shuffle per epoch, the last batch may be short (drop_last=False).

    `laplacian`: the convention the loss will be built with (losses.LAPLACIAN_MODES).  The first sample's rows 6:9 are
    checked against it (detect_laplacian_convention): a dataset written with another sign / normalisation raises, naming
    the convention that does match, instead of silently training the Laplacian term towards the wrong curvature;
    laplacian=None skips the check."""

    def __init__(self, files, subdivisions, device='cpu', laplacian='mean-v'):
        if isinstance(files, str):
            files = list_samples(files)
        if not files:
            raise ValueError('IcoDataset: no samples')
        self.files, self.subdivisions = list(files), subdivisions
        self.targets = torch.from_numpy(np.stack([load_sample(f, subdivisions)[1] for f in self.files])).to(device)
        self._check_laplacian(laplacian)

    @classmethod
    def from_tensors(cls, targets, subdivisions, files=None, laplacian='mean-v'):
        """A dataset over an in-memory (N, 9, V) tensor (synthetic meshes); same checks as the file form."""
        if targets.dim() != 3 or targets.shape[1:] != (9, geometry.num_vertices(subdivisions)):
            raise ValueError('IcoDataset.from_tensors: expected (N, 9, %d), got %s'
                             % (geometry.num_vertices(subdivisions), tuple(targets.shape)))
        self = object.__new__(cls)
        self.subdivisions, self.targets = subdivisions, targets
        self.files = list(files) if files is not None else ['sample%d' % i for i in range(targets.shape[0])]
        self._check_laplacian(laplacian)
        return self

    def _check_laplacian(self, laplacian):
        self.laplacian = laplacian
        if laplacian is None:
            return
        losses.laplacian_code(laplacian)
        found, errs = detect_laplacian_convention(self.targets[:1].detach().cpu(), self.subdivisions)
        if found != laplacian and errs[laplacian] >= 1e-3:
            raise ValueError(
                'IcoDataset: rows 6:9 of %s are not the %r Laplacian of rows 0:3 (relative error %.3g); %s.  Build the '
                'dataset and the loss with the same `laplacian=` (losses.LAPLACIAN_MODES), or pass laplacian=None to skip '
                'this check.' % (self.files[0], laplacian, errs[laplacian],
                                 'they match %r' % found if found else 'no known convention matches: %s' % errs))

    def __len__(self):
        return self.targets.shape[0]

    def subset(self, indices):
        """A view-like dataset of the given sample indices (the reference splits with torch.utils.data.Subset, run.py:69-74)."""
        other = object.__new__(IcoDataset)
        other.subdivisions, other.laplacian = self.subdivisions, self.laplacian
        other.files = [self.files[int(i)] for i in indices]
        other.targets = self.targets[torch.as_tensor(list(indices), device=self.targets.device, dtype=torch.long)]
        return other

    def batches(self, batch_size, shuffle=False, generator=None):
        """Yield (input (B,3,5n,2n), target (B,9,V)) on the dataset's device; `generator` (a CPU torch.Generator) seeds the
        shuffle.  Inputs are channels_last on a ROCm device (the chart layout of the kernels)."""
        n = len(self)
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        order = order.to(self.targets.device)
        for lo in range(0, n, batch_size):
            lbl = self.targets.index_select(0, order[lo:lo + batch_size])
            img = target_to_input(lbl, self.subdivisions)
            img = img.contiguous(memory_format=torch.channels_last) if img.is_cuda else img.contiguous()
            yield img, lbl

