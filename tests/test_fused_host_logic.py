"""Host-side decisions of geniconet_amd.fused that need no GPU: which BatchNorm configurations take the fused kernels (training /
inference), and the cached [mean | 1/std] vector of the inference path following its module's running statistics."""
import torch

from geniconet_amd import fused


def _bn(c=64, train=True):
    bn = torch.nn.BatchNorm2d(c)
    bn.train(train)
    return bn


def test_cpu_tensors_never_take_the_fused_paths():
    x = torch.zeros(2, 64, 20, 8)
    assert not fused.can_fuse(x, _bn()) and not fused.can_fuse_eval(x, _bn(train=False))


def test_shape_and_module_conditions():
    ok = fused._bn_shape_ok
    assert ok(_bn(64)) and ok(_bn(128)) and ok(_bn(512)) and ok(_bn(1024)) and ok(_bn(4))
    assert not ok(_bn(6)) and not ok(_bn(2048)) and not ok(_bn(12))          # C % 4, C <= 1024, 256 % (C / 4) == 0
    assert not ok(torch.nn.BatchNorm2d(64, affine=False)) and not ok(torch.nn.BatchNorm2d(64, track_running_stats=False))
    hooked = _bn(64)
    hooked.register_forward_hook(lambda m, i, o: None)
    assert not ok(hooked)


def test_the_inference_statistics_vector_follows_the_running_statistics():
    bn = _bn(8, train=False)
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    want = lambda: torch.cat([bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)])
    a = fused._eval_stat(bn)
    assert torch.equal(a, want()) and fused._eval_stat(bn) is a               # cached: the same tensor object
    with torch.no_grad():
        bn.running_mean.add_(1.0)                                              # in-place update (a training step): version bump
    b = fused._eval_stat(bn)
    assert b is not a and torch.equal(b, want())
    bn.load_state_dict({k: v.clone() + (0.25 if 'var' in k else 0) for k, v in bn.state_dict().items()})
    assert torch.equal(fused._eval_stat(bn), want())
    bn.eps = 1e-3                                                              # another epsilon is another vector
    assert torch.equal(fused._eval_stat(bn), want())
    other = _bn(8, train=False)
    assert fused._eval_stat(other) is not fused._eval_stat(bn)
