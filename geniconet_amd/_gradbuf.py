"""Where a parameter's gradient is written.

Under DistributedDataParallel with `gradient_as_bucket_view=True` a parameter's `.grad` is a view into a flat communication
bucket.  torch's autograd Functions return freshly allocated gradients, and DDP's reducer then moves every one of them into
its bucket view with one small kernel per parameter (`mul_out(bucket_view, grad, 1 / world)`; with a comm hook: `copy_`) -- 78
launches and 0.36 ms per step for this model (measured with rocprofv3, profiles/r03_ddp_vs_plain.txt), 3 % of a step.  The
backward kernels of this package can write a gradient wherever they are told, so the trainer remembers each parameter's bucket
view after a backward (`refresh`) and the Functions ask for it (`lease`) instead of allocating: the reducer finds the gradient
already aliasing its bucket and launches nothing.  A stale or missing view means the reducer copies as before (it checks
`is_alias_of`) -- ON THE CURRENT STREAM, so a gradient that was not served from its view must not be written on the weight
gradients' side stream (ico_conv._readers_can_wait asks `served_from_view`); a parameter is leased at most once per backward,
so a parameter used twice in a graph accumulates correctly through freshly allocated tensors.

Nothing here is specific to DDP: any `.grad` tensor kept from the previous backward is a valid destination.
"""
import torch

_ATTR, _LEASED = '_icn_grad_view', '_icn_grad_leased'
counts = {'view': 0, 'new': 0}      # leases served from a remembered view / by a new tensor (tests, diagnostics)


def refresh(params):
    """After a backward (and the reducer's hooks): remember every parameter's .grad as the place to write the next one."""
    for p in params:
        g = p.grad
        if g is not None and (g.is_contiguous() or g.stride() == p.stride()):
            setattr(p, _ATTR, g)
        setattr(p, _LEASED, False)


def forget(params):
    for p in params:
        if hasattr(p, _ATTR):
            delattr(p, _ATTR)
        setattr(p, _LEASED, False)


def lease(param, shape, device, dtype=torch.float32, stride=None):
    """A tensor of `shape` on `device` to write param's gradient into: the remembered view of its storage when there is one
    that fits and it has not been handed out since the last refresh, else a new tensor.  The result is a fresh tensor object
    (autograd may then take it as the parameter's .grad without cloning)."""
    v = getattr(param, _ATTR, None) if isinstance(param, torch.nn.Parameter) else None
    if (v is not None and not getattr(param, _LEASED, False) and param.grad is None and tuple(v.shape) == tuple(shape)
            and v.device == device and v.dtype == dtype and (stride is None or v.stride() == tuple(stride))
            and (stride is not None or v.is_contiguous())):
        setattr(param, _LEASED, True)
        counts['view'] += 1
        return v.detach()
    counts['new'] += 1
    if stride is not None:
        return torch.empty_strided(tuple(shape), tuple(stride), dtype=dtype, device=device)
    return torch.empty(tuple(shape), dtype=dtype, device=device)


def served_from_view(param, grad):
    """True when `grad` (returned by lease) IS the remembered view of param's gradient storage."""
    v = getattr(param, _ATTR, None) if isinstance(param, torch.nn.Parameter) else None
    return (v is not None and grad is not None and v.data_ptr() == grad.data_ptr() and tuple(v.shape) == tuple(grad.shape)
            and v.stride() == grad.stride())
