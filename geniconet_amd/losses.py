"""Losses of the ico2ico / ico2ico_vae training step (reference losses.py:10-145).

Same classes, constructor arguments and reporting methods as the reference.  The mesh helpers the reference
imports from the absent `PythonFunctions/mesh/utils.py` (losses.py:7) are restated here:
  * vertex normals: area-weighted face-normal accumulation, normalised -- the formula the reference itself
    ships for its preprocessing, generate.py:20-43;
  * Laplacian: uniform umbrella operator  mean(1-ring) - v  by default.  Upstream's sign / normalisation is unpinned
    (generate.py:197 writes target rows 6:9 with the absent mesh.utils.compute_laplacian), so the convention is a
    constructor option `laplacian` in LAPLACIAN_MODES -- 'mean-v', 'v-mean', 'sum-kv' (sum(ring) - k v), 'kv-sum' -- and
    data.IcoDataset checks the rows 6:9 of a dataset against it (data.detect_laplacian_convention), failing loudly on a
    mismatch instead of silently training towards the wrong curvature.
Loss values are kept as device tensors; `.item()` (a host sync in the reference on every iteration,
losses.py:72,81,129) only happens when get_last_losses() is called.

On ROCm fp32 tensors Point2Point_Loss runs on HIP kernels (libicn's icn_p2p_loss_*), forward and backward, for any
factors; the torch formulation below is what runs for other dtypes and for CPU tensors (as used by the CPU restatement in
bench.py), and it is what the HIP kernels are tested against besides the numpy oracle.  The KLD term (losses.py:105) runs
on icn_kld_fwd / icn_kld_bwd under the same conditions.
"""
import os

import torch

from . import _lib, geometry

_NO_HIP_LOSS = os.environ.get('ICN_NO_HIP_LOSS', '') == '1'
LAPLACIAN_MODES = _lib.LAP_MODES          # name -> ICN_LAP_* code of include/icn.h


def laplacian_code(laplacian):
    try:
        return LAPLACIAN_MODES[laplacian]
    except KeyError:
        raise ValueError('laplacian must be one of %s, got %r' % (sorted(LAPLACIAN_MODES), laplacian))


class _P2PLossFn(torch.autograd.Function):
    """Point2Point_Loss on the HIP path (icn_p2p_loss_* in include/icn.h): one kernel evaluates the three terms; the
    gradient is one kernel for the auto-encoder's factors 1/0/0 (run.py:690-692) and two for factors that weigh the normal
    and Laplacian terms (the VAE's 0.6/0.2/0.2, run.py:694-696)."""

    @staticmethod
    def forward(ctx, inputs, target, r, f_pos, f_nor, f_lap, lap_mode=0):
        L = _lib.lib()
        B = inputs.shape[0]
        grid = inputs.permute(0, 2, 3, 1).contiguous()            # (B, 5n, 2n, 3) = (B, P, 3); free for channels_last
        tgt = target.contiguous()
        terms = torch.empty(4, dtype=torch.float32, device=inputs.device)
        ws = torch.empty(max(L.icn_p2p_loss_workspace_floats(B, r), 1), dtype=torch.float32, device=inputs.device)
        with torch.cuda.device(inputs.device):
            rc = L.icn_p2p_loss_fwd(grid.data_ptr(), tgt.data_ptr(), B, r, f_pos, f_nor, f_lap, lap_mode, terms.data_ptr(),
                                    ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'icn_p2p_loss_fwd')
        ctx.save_for_backward(grid, tgt)
        ctx.cfg = (B, r, f_pos, f_nor, f_lap, lap_mode)
        total = terms[3].clone()
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gtotal, _gterms):
        B, r, f_pos, f_nor, f_lap, lap_mode = ctx.cfg
        L = _lib.lib()
        grid, tgt = ctx.saved_tensors
        dgrid = torch.empty_like(grid)
        up = gtotal.contiguous().to(torch.float32)
        ws = None
        if f_nor != 0 or f_lap != 0:
            ws = torch.empty(L.icn_p2p_loss_bwd_workspace_floats(B, r), dtype=torch.float32, device=grid.device)
        with torch.cuda.device(grid.device):
            rc = L.icn_p2p_loss_bwd(grid.data_ptr(), tgt.data_ptr(), up.data_ptr(), B, r, f_pos, f_nor, f_lap, lap_mode,
                                    dgrid.data_ptr(), ws.data_ptr() if ws is not None else None,
                                    torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'icn_p2p_loss_bwd')
        return dgrid.permute(0, 3, 1, 2), None, None, None, None, None, None


class _KLDFn(torch.autograd.Function):
    """mean_b(-0.5 * mean_d(1 + logvar - mu^2 - exp(logvar))) on the HIP path (icn_kld_fwd / icn_kld_bwd): one
    deterministic two-level sum forward, one elementwise kernel backward -- reference losses.py:105."""

    @staticmethod
    def forward(ctx, mu, logvar):
        L = _lib.lib()
        # elementwise + a full sum: any common element order will do; (B, H, W, C) storage is free for channels_last
        m, lv = (t.permute(0, 2, 3, 1).contiguous() if t.dim() == 4 else t.contiguous() for t in (mu, logvar))
        n = m.numel()
        out = torch.empty(1, dtype=torch.float32, device=m.device)
        ws = torch.empty(max(L.icn_kld_workspace_floats(n), 1), dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            rc = L.icn_kld_fwd(m.data_ptr(), lv.data_ptr(), n, out.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'icn_kld_fwd')
        ctx.save_for_backward(m, lv)
        ctx.dim4 = mu.dim() == 4
        return out[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        L = _lib.lib()
        m, lv = ctx.saved_tensors
        dm, dlv = torch.empty_like(m), torch.empty_like(lv)
        up = g.contiguous().to(torch.float32)
        with torch.cuda.device(m.device):
            rc = L.icn_kld_bwd(m.data_ptr(), lv.data_ptr(), up.data_ptr(), m.numel(), dm.data_ptr(), dlv.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'icn_kld_bwd')
        if ctx.dim4:
            return dm.permute(0, 3, 1, 2), dlv.permute(0, 3, 1, 2)
        return dm, dlv


def kld(mu, logvar):
    """The KL term of losses.py:105 for (B, ...) tensors: HIP kernels for ROCm fp32 tensors of equal shape, torch otherwise."""
    if (not _NO_HIP_LOSS and mu.is_cuda and logvar.is_cuda and mu.dtype == torch.float32 and logvar.dtype == torch.float32
            and mu.shape == logvar.shape and mu.numel() > 0):
        return _KLDFn.apply(mu, logvar)
    mu, logvar = torch.flatten(mu, start_dim=1), torch.flatten(logvar, start_dim=1)
    return torch.mean(-0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp(), dim=1), dim=0)


def grid_to_vertices(x, subdivisions):
    """(B, C, 5n, 2n) -> (B, 10*4^r + 2, C): row-major grid then N, S poles (mean of the 5 corner pixels);
    reference losses.py:47-51, ico_utils.py:10-24."""
    n = 2 ** subdivisions
    B, C = x.shape[:2]
    x5 = x.reshape(B, C, 5, n, 2 * n)
    poles = torch.stack((x5[:, :, :, 0, 0].mean(-1), x5[:, :, :, n - 1, 2 * n - 1].mean(-1)), dim=2)
    return torch.cat((x.reshape(B, C, -1), poles), dim=2).transpose(1, 2).contiguous()


def compute_vertex_normals(v, faces, eps=1e-10):
    """v (B, N, 3), faces (F, 3) int64 -> unit vertex normals (B, N, 3), face-area weighted (generate.py:20-43)."""
    # index_select (backward = index_add) instead of advanced indexing (backward = sort-based scatter)
    v0, v1, v2 = (torch.index_select(v, 1, faces[:, k]) for k in range(3))
    fn = torch.cross(v1 - v0, v2 - v0, dim=2)
    vn = torch.zeros_like(v)
    for k in range(3):
        vn = vn.index_add(1, faces[:, k], fn)
    return vn / vn.norm(dim=2, keepdim=True).clamp_min(eps)


def compute_laplacian_batch(v, nbr_idx, nbr_w, laplacian='mean-v'):
    """Uniform Laplacian: mean of the 1-ring minus the vertex.  nbr_idx (N, 6) int64 (padding entries point at
    vertex 0 with weight 0), nbr_w (N, 6) = 1/valence or 0.  `laplacian` (LAPLACIAN_MODES): 'v-mean' flips the sign,
    'sum-kv' / 'kv-sum' multiply by the valence (sum(ring) - k v and its negative)."""
    mode = laplacian_code(laplacian)
    ring = torch.index_select(v, 1, nbr_idx.reshape(-1)).view(v.shape[0], nbr_idx.shape[0], 6, 3)
    lap = (ring * nbr_w[None, :, :, None]).sum(2) - v
    if mode & 2:
        lap = lap * (nbr_w > 0).sum(1).to(lap.dtype)[None, :, None]
    return -lap if mode & 1 else lap


class Point2Point_Loss(torch.nn.Module):
    """factor_pos * MSE(v) + factor_nor * mean(1 - cos(normals)) + factor_lap * MSE(laplacian)
    against target rows [0:3], [3:6], [6:9]  -- reference losses.py:10-85.  `laplacian` (keyword, not in the reference's
    signature): the convention of the Laplacian rows, see the module docstring."""

    def __init__(self, subdivisions, factor_pos, factor_nor, factor_lap, laplacian='mean-v'):
        super().__init__()
        self.subdivisions = subdivisions
        self.laplacian, self._lap_mode = laplacian, laplacian_code(laplacian)
        self.factor_pos, self.factor_nor, self.factor_lap = factor_pos, factor_nor, factor_lap
        self.register_buffer('ico_faces', torch.from_numpy(geometry.get_ico_faces(subdivisions)))
        nbr = torch.from_numpy(geometry.vertex_neighbours(subdivisions).copy())
        valid = nbr >= 0
        self.register_buffer('nbr_idx', nbr.clamp_min(0))
        self.register_buffer('nbr_w', valid.float() / valid.sum(1, keepdim=True).float())
        self.last_loss_mse = self.last_loss_cos = self.last_loss_lap = self.last_loss_total = 0

    def _hip_path(self, inputs, target):
        """HIP kernels for ROCm fp32 tensors of the expected shapes."""
        if _NO_HIP_LOSS or not (inputs.is_cuda and target.is_cuda) or inputs.dtype != torch.float32 or target.dtype != torch.float32:
            return False
        n = 2 ** self.subdivisions
        if inputs.dim() != 4 or tuple(inputs.shape[1:]) != (3, 5 * n, 2 * n):
            return False
        if tuple(target.shape) != (inputs.shape[0], 9, 10 * n * n + 2):
            return False
        return True

    def forward(self, inputs, target):
        if self._hip_path(inputs, target):
            loss, terms = _P2PLossFn.apply(inputs, target, self.subdivisions, float(self.factor_pos), float(self.factor_nor),
                                           float(self.factor_lap), self._lap_mode)
            self.last_loss_mse, self.last_loss_cos, self.last_loss_lap = terms[0], terms[1], terms[2]
            self.last_loss_total = terms[3]
            return loss
        v = grid_to_vertices(inputs, self.subdivisions)
        tgt = target.transpose(1, 2)
        l_pos = torch.nn.functional.mse_loss(v, tgt[:, :, :3])
        # The normal and Laplacian terms are evaluated (and reported) even when their factor is 0, as the reference
        # does (losses.py:54,57,74-80); a zero-weighted term contributes exactly zero gradient, so it is evaluated
        # without recording a backward graph.
        with torch.set_grad_enabled(torch.is_grad_enabled() and self.factor_nor != 0):
            normals = compute_vertex_normals(v, self.ico_faces)
            l_nor = torch.mean(1 - torch.nn.functional.cosine_similarity(normals, tgt[:, :, 3:6], dim=2))
        with torch.set_grad_enabled(torch.is_grad_enabled() and self.factor_lap != 0):
            lap = compute_laplacian_batch(v, self.nbr_idx, self.nbr_w, self.laplacian)
            l_lap = torch.nn.functional.mse_loss(lap, tgt[:, :, 6:9])
        loss = self.factor_pos * l_pos + self.factor_nor * l_nor + self.factor_lap * l_lap
        self.last_loss_mse, self.last_loss_cos, self.last_loss_lap = l_pos.detach(), l_nor.detach(), l_lap.detach()
        self.last_loss_total = loss.detach()
        return loss

    def get_last_losses(self):
        return float(self.last_loss_mse), self.last_loss_cos, self.last_loss_lap, float(self.last_loss_total)


class KLD_Loss(torch.nn.Module):
    """mean_b( -0.5 * mean_d(1 + logvar - mu^2 - exp(logvar)) )   -- reference losses.py:87-118."""

    def forward(self, output, target):
        _, mu, logvar = output
        if self.factor_kl:
            self.loss = kld(mu, logvar)
        else:
            self.loss = torch.tensor(0.)
        return self.loss

    def get_last_losses(self):
        return 0, 0, 0, 0, -float(self.loss.detach())

    def get_factor(self):
        return self.factor_kl

    def update_factor(self, epoch, factor_step_size, factor_gamma):
        if epoch % factor_step_size == 0:
            self.factor_kl *= factor_gamma


class P2P_Loss(Point2Point_Loss):
    """reference losses.py:121-130."""

    def get_last_losses(self):
        return (float(self.last_loss_mse), float(self.last_loss_cos), float(self.last_loss_lap), 0.,
                float(self.last_loss_total))


class P2PKLD_Loss(P2P_Loss, KLD_Loss):
    """reconstruction + factor_kl * KLD   -- reference losses.py:132-145."""

    def __init__(self, subdivisions, factor_pos, factor_nor, factor_lap, factor_kl, laplacian='mean-v'):
        super().__init__(subdivisions, factor_pos, factor_nor, factor_lap, laplacian=laplacian)
        self.factor_kl = factor_kl

    def forward(self, output, target):
        self.kld_loss = KLD_Loss.forward(self, output, target)
        self.recons_loss = P2P_Loss.forward(self, output[0], target)
        self.loss = self.recons_loss + self.factor_kl * self.kld_loss
        return self.loss

    def get_last_losses(self):
        return float(self.recons_loss.detach()), 0, 0, -float(self.kld_loss.detach()), float(self.loss.detach())
