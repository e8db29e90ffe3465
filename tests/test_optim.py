"""geniconet_amd.optim.Adam: torch.optim.Adam (reference run.py:446) with the step as one HIP launch (icn_adam_step).

The oracle here is torch's own Adam: on the CPU in float64 for the value check, and on the device for state_dict interchange."""
import copy

import pytest
import torch

from geniconet_amd import optim

SIZES = [(1,), (7,), (2048,), (2049,), (3, 64, 7), (100003,), (256, 256, 7), (5,)]


def _params(device, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g, dtype=torch.float64).to(dtype).to(device).requires_grad_() for s in SIZES]


def _set_grads(ps, step, dtype, device):
    g = torch.Generator().manual_seed(100 + step)
    for p in ps:
        p.grad = (torch.randn(*p.shape, generator=g, dtype=torch.float64) * 10.0 ** (step - 2)).to(dtype).to(device)


def test_cpu_parameters_take_torchs_own_step():
    a, b = _params('cpu'), _params('cpu')
    oa, ob = optim.Adam(a, lr=1e-2), torch.optim.Adam(b, lr=1e-2)
    for step in range(3):
        _set_grads(a, step, torch.float32, 'cpu')
        _set_grads(b, step, torch.float32, 'cpu')
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert oa.state_dict()['state'].keys() == ob.state_dict()['state'].keys()


@pytest.mark.gpu
@pytest.mark.parametrize('wd', [0.0, 0.01])
def test_hip_adam_matches_float64_adam(wd):
    """Five steps with gradients spanning four decades and a learning rate that changes every step (CyclicLR does that):
    parameters and both moments against torch's Adam run on the CPU in float64 from the same fp32 start."""
    ps = _params('cuda')
    ref = [p.detach().cpu().double().requires_grad_() for p in ps]
    o = optim.Adam(ps, lr=1e-3, weight_decay=wd)
    oref = torch.optim.Adam(ref, lr=1e-3, weight_decay=wd)
    for step in range(5):
        for opt in (o, oref):
            opt.param_groups[0]['lr'] = 1e-3 * (1 + step)
        _set_grads(ps, step, torch.float32, 'cuda')
        for p, r in zip(ps, ref):
            r.grad = p.grad.cpu().double()
        o.step()
        oref.step()
        for p, r in zip(ps, ref):
            assert torch.allclose(p.detach().cpu().double(), r.detach(), rtol=2e-6, atol=1e-7), (step, tuple(p.shape))
            for k in ('exp_avg', 'exp_avg_sq'):
                got, want = o.state[p][k].cpu().double(), oref.state[r][k]
                assert torch.allclose(got, want, rtol=2e-6, atol=3e-7 * float(want.abs().max())), (step, k)   # fp32 cancellation
            assert float(o.state[p]['step']) == step + 1 and not o.state[p]['step'].is_cuda


@pytest.mark.gpu
def test_hip_adam_state_dict_interchanges_with_torch():
    """A run that switches optimiser class half-way (through state_dict, the reference's checkpoint format, run.py:330-340)
    ends where torch's Adam alone ends, to fp32 rounding; parameters without a gradient are left alone."""
    a, b = _params('cuda'), _params('cuda')
    oa, ob = optim.Adam(a, lr=3e-3), torch.optim.Adam(b, lr=3e-3)
    for step in range(2):
        _set_grads(a, step, torch.float32, 'cuda')
        _set_grads(b, step, torch.float32, 'cuda')
        a[-1].grad = None
        b[-1].grad = None
        oa.step()
        ob.step()
    assert torch.equal(a[-1], _params('cuda')[-1]) and a[-1] not in oa.state
    sd = copy.deepcopy(oa.state_dict())
    assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    oc = torch.optim.Adam(a, lr=3e-3)          # torch continues from the HIP optimiser's state ...
    oc.load_state_dict(sd)
    od = optim.Adam(b, lr=3e-3)                # ... and the HIP optimiser from torch's
    od.load_state_dict(copy.deepcopy(ob.state_dict()))
    for step in range(2, 4):
        _set_grads(a, step, torch.float32, 'cuda')
        _set_grads(b, step, torch.float32, 'cuda')
        oc.step()
        od.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_hip_adam_unsupported_options_fall_back_to_torch(monkeypatch):
    a, b = _params('cuda'), _params('cuda')
    oa, ob = optim.Adam(a, lr=1e-2, amsgrad=True), torch.optim.Adam(b, lr=1e-2, amsgrad=True)
    _set_grads(a, 1, torch.float32, 'cuda')
    _set_grads(b, 1, torch.float32, 'cuda')
    oa.step()
    ob.step()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert 'max_exp_avg_sq' in oa.state[a[0]]


def test_closure_runs_before_eligibility_is_decided():
    """torch.optim.Adam.step(closure) evaluates the closure FIRST; with zero_grad(set_to_none=True) + backward inside it the
    gradients only exist afterwards, so the first step must already move the parameters (and the closure must run once)."""
    a, b = _params('cpu'), _params('cpu')
    oa, ob = optim.Adam(a, lr=1e-2), torch.optim.Adam(b, lr=1e-2)
    calls = {'a': 0, 'b': 0}

    def make(ps, opt, tag):
        def closure():
            calls[tag] += 1
            opt.zero_grad(set_to_none=True)
            loss = sum((p * p).sum() for p in ps)
            loss.backward()
            return loss
        return closure
    for _ in range(2):
        la, lb = oa.step(make(a, oa, 'a')), ob.step(make(b, ob, 'b'))
        assert float(la) == float(lb)
    assert calls == {'a': 2, 'b': 2}
    for x, y, x0 in zip(a, b, _params('cpu')):
        assert torch.equal(x, y) and not torch.equal(x, x0)
