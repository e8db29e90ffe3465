"""Fused BatchNorm (+ residual) + ReLU of the residual blocks on the HIP path.

The reference strings torch builtins together (models.py:36-40,58-62):
    relu(icobn00(conv00(.)))                      -> bn_relu(a, icobn00)
    relu(icobn01(conv01(.)) + icobn10(conv10(.))) -> bn_add_relu(a, icobn01, b, icobn10)
Here each is one autograd Function over libicn's icn_bn_* kernels (two streaming passes forward, two backward, instead
of 4-7 separate elementwise passes).  The `nn.BatchNorm2d` modules stay in the module tree (state_dict keys, running
statistics, `num_batches_tracked` are updated exactly as torch does in training mode); the fused path is taken in
training mode on ROCm tensors when nobody hooked those modules, and -- as one streaming pass with the running statistics
(bn_relu_eval / bn_add_relu_eval) -- in eval mode when no autograd graph is being recorded (inference); otherwise the
modules are called as usual.
"""
import os
import weakref

import torch

from . import _gradbuf, _lib
from .ico_conv import (_nhwc, _stream, ico_conv_pair, ico_conv_pair_supported, ico_upconv_pair,
                       ico_upconv_pair_supported)

_DISABLED = os.environ.get('ICN_NO_FUSED_BN', '') == '1'


def _bn_shape_ok(bn):
    if not (bn.affine and bn.track_running_stats):
        return False
    if bn._forward_hooks or bn._forward_pre_hooks or bn._backward_hooks:
        return False
    c = bn.num_features
    return not (c % 4 or c > 1024 or 256 % (c // 4))


def can_fuse(x, *bns):
    """Training-mode BatchNorms (batch statistics) on a ROCm fp32 tensor, nobody hooked them.  With or without autograd: under
    torch.no_grad() the same Function runs and nothing is kept for a backward."""
    if _DISABLED or not x.is_cuda or x.dtype != torch.float32:
        return False
    return all(bn.training and bn.momentum is not None and _bn_shape_ok(bn) for bn in bns)


def can_fuse_eval(x, *bns):
    """Eval-mode BatchNorms (running statistics) in an inference forward: no autograd graph is being recorded for the input or
    the BatchNorm parameters (serving, `--process test`).  Anything else in eval mode takes the torch modules."""
    if _DISABLED or not x.is_cuda or x.dtype != torch.float32:
        return False
    if torch.is_grad_enabled() and (x.requires_grad or any(bn.weight.requires_grad or bn.bias.requires_grad for bn in bns)):
        return False
    return all((not bn.training) and _bn_shape_ok(bn) for bn in bns)


_NO_PAIR = os.environ.get('ICN_NO_PAIR', '') == '1'


def conv_pair(x, conv_a, conv_b):
    """(conv_a(x), conv_b(x)) for two IcoConvS2S modules that see the same tensor -- conv00 / conv10 of the reference's
    residual blocks (models.py:37-39,59-60).  One launch per pass (icn_conv_pair_*) when the two agree in stride / level /
    corner mode / bias, the shape is inside the pair path and nobody hooked either module; else two module calls."""
    ok = (not _NO_PAIR and x.is_cuda and x.dtype == torch.float32
          and conv_a.stride == conv_b.stride and conv_a.subdivisions == conv_b.subdivisions
          and conv_a.corner_mode == conv_b.corner_mode and (conv_a.bias is None) == (conv_b.bias is None))
    if ok:
        for m in (conv_a, conv_b):
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
                ok = False
    if ok and ico_conv_pair_supported(x, conv_a.weight, conv_b.weight, conv_a.subdivisions, conv_a.stride):
        return ico_conv_pair(x, conv_a.weight, conv_a.bias, conv_b.weight, conv_b.bias, conv_a.subdivisions,
                             conv_a.stride, conv_a.corner_mode)
    return conv_a(x), conv_b(x)


_NO_UPCONV = os.environ.get('ICN_NO_UPCONV', '') == '1'


def upconv_pair(x, up_a, up_b, conv_a, conv_b):
    """(conv_a(up_a(x)), conv_b(up_b(x))) -- the head of the reference's decoder block (models.py:58-60).  The two
    IcoUpsampleS2S modules are parameter-free and see the same tensor, so the r -> r+1 upsample is shared; when the shape
    allows, upsample and pair convolution run as ONE composite gather-GEMM over the coarse tensor (icn_upconv_fwd), else
    as one upsample + the pair convolution, else module by module (someone hooked a module, unequal branches ...)."""
    mods = (up_a, up_b, conv_a, conv_b)
    plain = not any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in mods)
    same = (up_a.subdivisions == up_b.subdivisions and up_a.corner_mode == up_b.corner_mode == conv_a.corner_mode
            == conv_b.corner_mode and conv_a.stride == conv_b.stride == 1
            and conv_a.subdivisions == conv_b.subdivisions == up_a.subdivisions + 1
            and (conv_a.bias is None) == (conv_b.bias is None))
    if plain and same and not _NO_PAIR and not _NO_UPCONV and x.is_cuda and x.dtype == torch.float32 \
            and ico_upconv_pair_supported(x, conv_a.weight, conv_b.weight, up_a.subdivisions):
        return ico_upconv_pair(x, conv_a.weight, conv_a.bias, conv_b.weight, conv_b.bias, up_a.subdivisions, up_a.corner_mode)
    up = up_a(x)
    hooked = up_b._forward_hooks or up_b._forward_pre_hooks
    up_skip = up_b(x) if hooked else up
    if up_skip is up:
        return conv_pair(up, conv_a, conv_b)
    return conv_a(up), conv_b(up_skip)


class _BnReluFn(torch.autograd.Function):
    """y = relu(bn_a(a) [+ bn_b(b)]) with batch statistics; updates the running statistics in place."""

    @staticmethod
    def forward(ctx, a, ga, ba, rm_a, rv_a, eps_a, mom_a, b, gb, bb, rm_b, rv_b, eps_b, mom_b):
        L = _lib.lib()
        ap = _nhwc(a)
        Bn, H, W, C = ap.shape
        M = Bn * H * W
        dual = b is not None
        bp = _nhwc(b) if dual else None
        dev = a.device
        ws = torch.empty(L.icn_bn_workspace_floats(M, C), dtype=torch.float32, device=dev)
        stat_a = torch.empty(2 * C, dtype=torch.float32, device=dev)
        stat_b = torch.empty(2 * C, dtype=torch.float32, device=dev) if dual else None
        y = torch.empty_like(ap)
        with torch.cuda.device(dev):
            st = _stream()
            if dual:        # both inputs' statistics in one pass
                _lib.check(L.icn_bn_stats2(ap.data_ptr(), bp.data_ptr(), M, C, eps_a, mom_a, rm_a.data_ptr(), rv_a.data_ptr(),
                                           stat_a.data_ptr(), eps_b, mom_b, rm_b.data_ptr(), rv_b.data_ptr(), stat_b.data_ptr(),
                                           ws.data_ptr(), st), 'icn_bn_stats2')
            else:
                _lib.check(L.icn_bn_stats(ap.data_ptr(), M, C, eps_a, mom_a, rm_a.data_ptr(), rv_a.data_ptr(), stat_a.data_ptr(),
                                          ws.data_ptr(), st), 'icn_bn_stats')
            _lib.check(L.icn_bn_relu_fwd(ap.data_ptr(), bp.data_ptr() if dual else None, stat_a.data_ptr(),
                                         stat_b.data_ptr() if dual else None, ga.data_ptr(), ba.data_ptr(),
                                         gb.data_ptr() if dual else None, bb.data_ptr() if dual else None, y.data_ptr(), M, C,
                                         st), 'icn_bn_relu_fwd')
        ctx.save_for_backward(ap, bp, stat_a, stat_b, ga, ba, gb, bb)    # not y: the backward recomputes the ReLU mask
        ctx.dims = (M, C, dual)
        ctx.params = (ga, ba, gb, bb)                    # where the parameter gradients go (_gradbuf.lease)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        L = _lib.lib()
        ap, bp, stat_a, stat_b, ga, ba, gb, bb = ctx.saved_tensors
        M, C, dual = ctx.dims
        gyp = _nhwc(gy)
        dev = gyp.device
        da = torch.empty_like(ap)
        db = torch.empty_like(bp) if dual else None
        sums = torch.empty(3 * C, dtype=torch.float32, device=dev)
        ws = torch.empty(L.icn_bn_workspace_floats(M, C), dtype=torch.float32, device=dev)
        # the parameter gradients in tensors of their own (the kernel writes them beside `sums`): a view into a DDP bucket when
        # the trainer remembered one, else new
        pga, pba, pgb, pbb = ctx.params
        dga, dba = _gradbuf.lease(pga, (C,), dev), _gradbuf.lease(pba, (C,), dev)
        dgb = _gradbuf.lease(pgb, (C,), dev) if dual else None
        dbb = _gradbuf.lease(pbb, (C,), dev) if dual else None
        with torch.cuda.device(dev):
            _lib.check(L.icn_bn_relu_bwd(gyp.data_ptr(), ap.data_ptr(), bp.data_ptr() if dual else None, stat_a.data_ptr(),
                                         stat_b.data_ptr() if dual else None, ga.data_ptr(), ba.data_ptr(),
                                         gb.data_ptr() if dual else None, bb.data_ptr() if dual else None, da.data_ptr(),
                                         db.data_ptr() if dual else None, sums.data_ptr(), ws.data_ptr(), M, C, dba.data_ptr(),
                                         dga.data_ptr(), dbb.data_ptr() if dual else None, dgb.data_ptr() if dual else None,
                                         _stream()), 'icn_bn_relu_bwd')
        out = [da.permute(0, 3, 1, 2), dga, dba, None, None, None, None]
        if dual:
            out += [db.permute(0, 3, 1, 2), dgb, dbb, None, None, None, None]
        else:
            out += [None] * 7
        return tuple(out)


def _args(bn):
    return bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum)


# `num_batches_tracked += 1` of every fused BatchNorm (torch does it inside the module's forward): a model forward has 19 of
# them, each a kernel launch of its own.  Inside `counters_deferred()` (the model's forward) they are collected and bumped
# by ONE multi-tensor launch when the outermost scope closes; outside any scope the bump is immediate, as before.
_pending_counters = []
_defer_depth = 0


class counters_deferred:
    def __enter__(self):
        global _defer_depth
        _defer_depth += 1

    def __exit__(self, *exc):
        global _defer_depth
        _defer_depth -= 1
        if _defer_depth == 0 and _pending_counters:
            todo = list(_pending_counters)
            _pending_counters.clear()
            torch._foreach_add_(todo, 1)
        return False


def _bump(bn):
    # the fused statistics kernels update running_mean / running_var through raw pointers, which does not move the tensors'
    # version counters: the eval-mode cache (below) keyed on them would go stale after a training step -> drop the entry here
    _eval_stats.pop(bn, None)
    if _defer_depth > 0:
        _pending_counters.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


def bn_relu(a, bn):
    """relu(bn(a)) -- training-mode BatchNorm2d `bn` (its running statistics are updated)."""
    _bump(bn)
    return _BnReluFn.apply(a, *_args(bn), None, None, None, None, None, 0.0, 0.0)


def bn_add_relu(a, bn_a, b, bn_b):
    """relu(bn_a(a) + bn_b(b))."""
    _bump(bn_a)
    _bump(bn_b)
    return _BnReluFn.apply(a, *_args(bn_a), b, *_args(bn_b))


# ---- inference: relu(bn(a) [+ bn(b)]) with the RUNNING statistics, one streaming pass (icn_bn_relu_fwd; no statistics pass)
_eval_stats = weakref.WeakKeyDictionary()    # bn module -> (versions / storage of its running statistics, [mean | 1/sqrt(var + eps)])


def _eval_stat(bn):
    """[running_mean | rsqrt(running_var + eps)] of an eval-mode BatchNorm, rebuilt only when the statistics changed (their
    tensors' version counters / storage): a serving forward pays for it once."""
    rm, rv = bn.running_mean, bn.running_var
    key = (rm._version, rv._version, rm.data_ptr(), rv.data_ptr(), float(bn.eps))
    hit = _eval_stats.get(bn)
    if hit is None or hit[0] != key:
        hit = (key, torch.cat([rm.detach().float(), torch.rsqrt(rv.detach().float() + bn.eps)]))
        _eval_stats[bn] = hit
    return hit[1]


def _bn_relu_eval(a, bn_a, b=None, bn_b=None):
    L = _lib.lib()
    ap = _nhwc(a.detach())
    Bn, H, W, C = ap.shape
    M = Bn * H * W
    dual = b is not None
    bp = _nhwc(b.detach()) if dual else None
    stat_a = _eval_stat(bn_a)
    stat_b = _eval_stat(bn_b) if dual else None
    y = torch.empty_like(ap)
    with torch.cuda.device(a.device):
        _lib.check(L.icn_bn_relu_fwd(ap.data_ptr(), bp.data_ptr() if dual else None, stat_a.data_ptr(),
                                     stat_b.data_ptr() if dual else None, bn_a.weight.data_ptr(), bn_a.bias.data_ptr(),
                                     bn_b.weight.data_ptr() if dual else None, bn_b.bias.data_ptr() if dual else None,
                                     y.data_ptr(), M, C, _stream()), 'icn_bn_relu_fwd')
    return y.permute(0, 3, 1, 2)


def bn_relu_eval(a, bn):
    """relu(bn(a)) for an eval-mode BatchNorm2d in an inference forward (can_fuse_eval)."""
    return _bn_relu_eval(a, bn)


def bn_add_relu_eval(a, bn_a, b, bn_b):
    """relu(bn_a(a) + bn_b(b)), eval mode, inference forward."""
    return _bn_relu_eval(a, bn_a, b, bn_b)


class _HeadFn(torch.autograd.Function):
    """y = tanh(conv1x1(x)) : the output head `Conv2d(C, 3, 1) + Tanh` (reference models.py:151-154)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        L = _lib.lib()
        xp = _nhwc(x)
        Bn, H, W, Cin = xp.shape
        Cout, M = weight.shape[0], Bn * H * W
        w2 = weight.reshape(Cout, Cin).contiguous()
        y = torch.empty(Bn, H, W, Cout, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.icn_head_fwd(xp.data_ptr(), w2.data_ptr(), bias.contiguous().data_ptr(), y.data_ptr(), M, Cin, Cout,
                                      _stream()), 'icn_head_fwd')
        ctx.save_for_backward(xp, w2, y)
        ctx.wshape, ctx.wstride = weight.shape, weight.stride()
        ctx.params = (weight, bias)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        L = _lib.lib()
        xp, w2, y = ctx.saved_tensors
        Bn, H, W, Cin = xp.shape
        Cout, M = w2.shape[0], Bn * H * W
        gyp = _nhwc(gy)
        dev = gyp.device
        dx = torch.empty_like(xp) if ctx.needs_input_grad[0] else None
        # (Cout, Cin, 1, 1) in the parameter's own strides -- (Cin, 1, Cin, Cin) for a channels_last weight, i.e. the kernel's
        # (Cout, Cin) row-major order -- so that DistributedDataParallel can alias it (gradient_as_bucket_view)
        dw = _gradbuf.lease(ctx.params[0], ctx.wshape, dev, stride=ctx.wstride)
        db = _gradbuf.lease(ctx.params[1], (Cout,), dev)
        ws = torch.empty(L.icn_head_workspace_floats(M, Cin), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.icn_head_bwd(gyp.data_ptr(), y.data_ptr(), xp.data_ptr(), w2.data_ptr(),
                                      dx.data_ptr() if dx is not None else None, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M,
                                      Cin, Cout, _stream()), 'icn_head_bwd')
        return (dx.permute(0, 3, 1, 2) if dx is not None else None), dw, db


def can_fuse_head(x, seq):
    """seq = nn.Sequential(Conv2d(C, <=4, kernel_size=1), Tanh()) on a ROCm fp32 tensor, nobody hooked it."""
    if _DISABLED or not x.is_cuda or x.dtype != torch.float32 or len(seq) != 2:
        return False
    conv, act = seq[0], seq[1]
    if not isinstance(conv, torch.nn.Conv2d) or not isinstance(act, torch.nn.Tanh):
        return False
    if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.groups != 1 or conv.bias is None:
        return False
    if conv.in_channels not in (16, 32, 64, 128, 256) or conv.out_channels > 4:
        return False
    for m in (seq, conv, act):
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
            return False
    return True


def head(x, seq):
    """seq(x) through the fused kernel when possible."""
    if can_fuse_head(x, seq):
        return _HeadFn.apply(x, seq[0].weight, seq[0].bias)
    return seq(x)


class _ReparamFn(torch.autograd.Function):
    """z = eps * exp(0.5 * logvar) + mu  (reference models.py:89-92) as one kernel per pass (icn_reparam_fwd / _bwd)
    instead of exp, mul, mul, add and their four backward kernels."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        L = _lib.lib()
        m, lv, e = _nhwc(mu), _nhwc(logvar), _nhwc(eps)
        z = torch.empty_like(m)
        with torch.cuda.device(m.device):
            _lib.check(L.icn_reparam_fwd(m.data_ptr(), lv.data_ptr(), e.data_ptr(), m.numel(), z.data_ptr(), _stream()),
                       'icn_reparam_fwd')
        ctx.save_for_backward(lv, e)
        return z.permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gz):
        L = _lib.lib()
        lv, e = ctx.saved_tensors
        g = _nhwc(gz)
        dm, dlv = torch.empty_like(g), torch.empty_like(g)
        with torch.cuda.device(g.device):
            _lib.check(L.icn_reparam_bwd(g.data_ptr(), lv.data_ptr(), e.data_ptr(), g.numel(), dm.data_ptr(), dlv.data_ptr(),
                                         _stream()), 'icn_reparam_bwd')
        return dm.permute(0, 3, 1, 2), dlv.permute(0, 3, 1, 2), None


def reparameterize(mu, logvar):
    """randn_like(std) * std + mu with std = exp(0.5 * logvar) (reference models.py:89-92).  The noise is drawn by
    torch.randn_like exactly as the reference draws it (same generator, same shape and strides as std); the arithmetic is
    one HIP kernel for ROCm fp32 (B, C, H, W) tensors, the reference's torch expression otherwise."""
    if (not _DISABLED and mu.is_cuda and logvar.is_cuda and mu.dtype == torch.float32 and logvar.dtype == torch.float32
            and mu.dim() == 4 and mu.shape == logvar.shape and mu.numel() > 0):
        return _ReparamFn.apply(mu, logvar, torch.randn_like(logvar))
    std = torch.exp(0.5 * logvar)
    return torch.randn_like(std) * std + mu
