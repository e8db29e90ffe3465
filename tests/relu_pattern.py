"""Test helper: compare network GRADIENTS at one and the same ReLU activation pattern.

The gradient of this network is a discontinuous function of its forward pass: a ReLU pre-activation within fp32 rounding of
zero passes or blocks its whole upstream gradient depending on where the rounding lands, and one such flip moves the small
cancelling sums of the BatchNorm backward below it by 1e-3 .. 1e-2 (tools/diag_gradient_error.py).  With ~1e6 pre-activations
per evaluation a few always are that close (density of a standardised pre-activation at zero 0.4 -> 0.8e-6 * N values within
1e-6 sigma), so two correct fp32 implementations do not agree to 2e-3 on every tensor at every batch, and batches without
such values cannot be found by searching seeds (P = exp(-0.8 m N) for a margin m: 3e-5 for m = 1e-5 at N = 1.3e6).

Instead of widening the bound, the comparison removes the discontinuity: the GPU forward's own activation pattern (which
ReLU outputs are > 0) is read off the product network with forward hooks, and the float64 oracle is evaluated with every
ReLU replaced by a multiplication with that fixed 0/1 pattern.  Where the two patterns agree this IS the oracle; where they
differ the oracle's pre-activation is ~0 (asserted: `check_flips`), so its forward changes by ~1e-6 and its gradient becomes
the one for the pattern the GPU actually ran.  Both gradients are then smooth functions of identical structure and are held
to the plain 2e-3 bound on every tensor, no second acceptance path.
"""
import numpy as np
import torch


def capture_relu_outputs(net):
    """Hooks that collect the output of every ReLU of the product network in the order the oracle calls F.relu: the stem's
    (= input of the first residual block), then per block `h = relu(bn00(conv00(.)))` (= input of conv01) and the block's
    output.  None of these hooks sits on a module whose fusion they would switch off (fused.conv_pair looks at conv00 /
    conv10, fused.can_fuse at the BatchNorms, _Encoder at the stem's ReLU).  Returns (list filled at forward time, handles)."""
    from geniconet_amd.models import BasicIcoS2SDownBlock, BasicIcoS2SUpBlock
    outs, handles = [], []
    blocks = [m for m in net.modules() if isinstance(m, (BasicIcoS2SDownBlock, BasicIcoS2SUpBlock))]
    handles.append(blocks[0].register_forward_pre_hook(lambda m, a: outs.append(a[0].detach())))
    for b in blocks:
        handles.append(b.conv01.register_forward_pre_hook(lambda m, a: outs.append(a[0].detach())))
        handles.append(b.register_forward_hook(lambda m, a, o: outs.append(o.detach())))
    return outs, handles


class relu_pattern:
    """Context manager: inside it torch.nn.functional.relu (hence nn.ReLU and the oracle's F.relu calls) multiplies its k-th
    call's input by patterns[k] (0/1, any device / dtype; given as the ReLU OUTPUTS of the other implementation) instead of
    thresholding.  Records, per call, the positions where the pattern differs from the input's own sign and the input's size
    there relative to the tensor's rms (`flips`: list of (call index, count, largest |x| / rms))."""

    def __init__(self, relu_outputs):
        self.masks = [(t > 0).cpu() for t in relu_outputs]
        self.k, self.flips = 0, []

    def __enter__(self):
        self._orig = torch.nn.functional.relu

        def patterned(x, inplace=False):
            m = self.masks[self.k]
            assert m.shape == x.shape, (self.k, tuple(m.shape), tuple(x.shape))
            own = x.detach() > 0
            diff = own != m
            n = int(diff.sum())
            if n:
                rms = float(x.detach().pow(2).mean().sqrt())
                self.flips.append((self.k, n, float(x.detach()[diff].abs().max()) / max(rms, 1e-300)))
            self.k += 1
            return x * m.to(x.dtype)
        torch.nn.functional.relu = patterned
        return self

    def __exit__(self, *exc):
        torch.nn.functional.relu = self._orig
        return False


def check_flips(ctx, n_calls, margin=1e-4, most=200):
    """Every ReLU of the oracle was patterned (n_calls), and wherever the GPU's pattern differs from the oracle's own sign the
    oracle's pre-activation is within `margin` of zero relative to its tensor's rms -- the flip is a rounding event, not a
    forward error -- and there are at most `most` of them.  Returns the total count."""
    assert ctx.k == n_calls, (ctx.k, n_calls)
    total = sum(n for _, n, _ in ctx.flips)
    for k, n, size in ctx.flips:
        assert size < margin, ('ReLU %d: %d elements differ, the largest pre-activation is %.2e of the rms' % (k, n, size))
    assert total <= most, ctx.flips
    return total


def gradient_errors(params_gpu, params_ref, floor_frac=1e-3):
    """rel-L2 of every parameter gradient (GPU against reference), denominators floored at floor_frac of the largest
    reference gradient norm (conv biases in front of a train-mode BatchNorm have a zero true gradient)."""
    floor = floor_frac * max(float(q.grad.norm()) for q in params_ref.values())
    return {k: float((params_gpu[k].grad.detach().cpu().double() - q.grad.double()).norm()) / max(float(q.grad.norm()), floor)
            for k, q in params_ref.items()}


def worst(errs):
    k = max(errs, key=errs.get)
    return errs[k], k


def as_numpy(t):
    return np.asarray(t.detach().cpu())
