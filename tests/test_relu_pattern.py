"""The activation-pattern comparison of network gradients (tests/relu_pattern.py): the method itself on the CPU, and the
whole-network gradients at I5 on the GPU against the float64 oracle at the GPU's own pattern."""
import numpy as np
import pytest
import torch

from oracle import models_ref
from relu_pattern import capture_relu_outputs, check_flips, gradient_errors, relu_pattern, worst


def _oracle_gradients(name, R, state, x, t, eps, dtype, patterns=None, record=None):
    """Gradients of the oracle network + the product's host-side torch loss in `dtype`; `patterns`: ReLU outputs of another
    evaluation to run at (relu_pattern); `record`: list that receives this evaluation's own ReLU outputs."""
    from geniconet_amd import models
    from geniconet_amd.train import build_criterion
    p = models.default_params(name, subdivisions=R)
    m = getattr(models_ref, name)(R=R).train()
    m.load_state_dict(state)
    m = m.to(dtype)
    args = (x.to(dtype),) if name == 'ico2ico' else (x.to(dtype), eps.to(dtype))
    ctx = None
    if patterns is not None:
        with relu_pattern(patterns) as ctx:
            y = m(*args)
    elif record is not None:
        orig = torch.nn.functional.relu

        def rec(v, inplace=False):
            out = orig(v)
            record.append(out.detach())
            return out
        torch.nn.functional.relu = rec
        try:
            y = m(*args)
        finally:
            torch.nn.functional.relu = orig
    else:
        y = m(*args)
    build_criterion(p, 'cpu').to(dtype)(y, t.to(dtype)).backward()
    return dict(m.named_parameters()), ctx


def test_a_relu_flip_is_real_and_the_pattern_comparison_removes_it():
    """CPU only, no GPU involved: the oracle VAE in fp32 against itself in float64 on a batch where ONE ReLU pre-activation
    (of 7.5e5) lies 9e-7 of its tensor's rms from zero and rounds to the other side in fp32.
      (a) the flip is real: exactly that one element differs between the two patterns, |x| < 1e-5 rms in float64;
      (b) it alone puts a third of the network's gradients beyond 2e-3 of float64 (up to 4e-3) although both evaluations are
          correct -- which is why a plain bound on gradients needs an escape hatch, and why this suite does not use one;
      (c) evaluated at the fp32 run's pattern, the float64 oracle agrees with it to 5e-4 on EVERY tensor."""
    from geniconet_amd import data
    name, R, B = 'ico2ico_vae', 3, 3
    torch.manual_seed(5)
    state = getattr(models_ref, name)(R=R).state_dict()
    x, t = data.synthetic_batch(B, R, seed=43)
    eps = torch.randn(B, 512, 5, 2, generator=torch.Generator().manual_seed(70))
    outs32 = []
    g32, _ = _oracle_gradients(name, R, state, x, t, eps, torch.float32, record=outs32)
    g64, _ = _oracle_gradients(name, R, state, x, t, eps, torch.float64)
    g64p, pat = _oracle_gradients(name, R, state, x, t, eps, torch.float64, patterns=outs32)
    assert len(outs32) == 11 and check_flips(pat, 11, margin=1e-5) == 1, pat.flips           # (a)
    plain = gradient_errors(g32, g64)
    over = sorted(k for k, v in plain.items() if v >= 2e-3)
    assert 5 <= len(over) <= 40 and worst(plain)[0] < 1e-2, (len(over), worst(plain))           # (b)
    assert worst(gradient_errors(g32, g64p))[0] < 5e-4                                          # (c)


@pytest.mark.gpu
@pytest.mark.parametrize('name,R,B', [('ico2ico', 5, 2), ('ico2ico_vae', 5, 2), ('ico2ico', 6, 1)], ids=['ico2ico_I5', 'ico2ico_vae_I5', 'ico2ico_I6'])
def test_whole_network_gradients_against_the_oracle_at_the_gpu_pattern(name, R, B, monkeypatch):
    """The backward chain as the bench runs it (split-mode data gradients with virtual rows, stream-K, LDS-staged sparse passes,
    weight gradients on the second stream), end to end against the float64 oracle network evaluated at the GPU forward's ReLU
    pattern: every parameter gradient to 5e-4 (a quarter of the contract's 2e-3; measured 2.1e-4 at I5), and every pattern
    difference a rounding event (< 1e-4 of the tensor's rms).  I5 = BASELINE configs 2-4; I6 (levels 6 -> 3 -> 6, one mesh)
    = config 5, which the reference cannot build (models.py:108-148 hard-wires 5): the oracle is the only check it has, and
    this is the one that covers its whole forward + backward.  (The forward itself: tests/test_golden.py, test_ref_goldens.py.)"""
    from geniconet_amd import data, models
    from geniconet_amd.ico_conv import set_weight_gradient_stream
    from geniconet_amd.train import build_criterion
    torch.manual_seed(11)
    ref = getattr(models_ref, name)(R=R).train()
    p = models.default_params(name, subdivisions=R)
    net = getattr(models, name)(p)
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.cuda().to(memory_format=torch.channels_last).train()
    x, t = data.synthetic_batch(B, R, seed=77)
    eps = torch.randn(B, 512, 5 * 2 ** (R - 3), 2 ** (R - 2), generator=torch.Generator().manual_seed(3))
    monkeypatch.setattr(torch, 'randn_like', lambda s, **kw: eps.to(device=s.device, dtype=s.dtype))
    outs, handles = capture_relu_outputs(net)
    y = net(x.cuda().contiguous(memory_format=torch.channels_last))
    for h in handles:
        h.remove()
    loss = build_criterion(p, 'cuda')(y, t.cuda())
    prev = set_weight_gradient_stream('deferred')                                # as Trainer.step runs it
    try:
        loss.backward()
    finally:
        set_weight_gradient_stream(*prev)
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    n_relu = 1 + 2 * (6 if name == 'ico2ico' else 5)
    g64, pat = _oracle_gradients(name, R, ref.state_dict(), x, t, eps, torch.float64, patterns=outs)
    n_flips = check_flips(pat, n_relu)
    errs = gradient_errors(dict(net.named_parameters()), g64)
    print('%s I%d: %d ReLU flips %s, worst gradient %.2e %s' % ((name, R, n_flips, pat.flips) + worst(errs)))
    assert worst(errs)[0] < 5e-4, sorted(((v, k) for k, v in errs.items()), reverse=True)[:5]
    assert np.isfinite(float(loss.detach()))
    if name == 'ico2ico':                                                        # the forward at this pattern, too
        with torch.no_grad(), relu_pattern(outs):
            m = models_ref.ico2ico(R=R).train()
            m.load_state_dict(ref.state_dict())
            y64 = m.double()(x.double())
        assert float((y.detach().cpu().double() - y64).norm() / y64.norm()) < 1e-4
