"""The three-way bf16 split of the channel-mixing contraction (ICN_ARITH=bf16x3; include/icn.h icn_set_arith) is fp32-GRADE arithmetic:
its error against the float64 oracle must stay within 2 x the exact fp32 kernel's on the same inputs -- not merely under the 1e-4 bar
(VERDICT r5 item 1 acceptance).  The reference computes in plain fp32 (models.py / run.py: no autocast), so this is the bar a drop-in has
to hold.  Every case also checks that the split kernels really ran (HIP-event profile of the launch names).
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import ico_ref
from test_gpu_parity import MFMA_CASES, PAIR_CASES

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture
def arith():
    from geniconet_amd import _lib
    prev = _lib.get_arith()
    yield _lib
    _lib.set_arith(prev)


def _kernels_of(fn):
    from geniconet_amd import _lib
    _lib.profile_start(256)
    out = fn()
    torch.cuda.synchronize()
    return out, {e['kernel'] for e in _lib.profile_stop()}


def _conv_case(case, seed, pair, bias=True):
    """Inputs + the float64 oracle's outputs and gradients for a single convolution or a pair sharing its input."""
    r, stride, cin, cout, B, mode = case[:6]
    g = torch.Generator().manual_seed(seed)
    n = 2 ** r
    k = 2 if pair else 1
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g)
    ws = [torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5 for _ in range(k)]
    bs = [torch.randn(cout, generator=g) if bias else None for _ in range(k)]
    xr = x.double().requires_grad_()
    wr = [w.double().requires_grad_() for w in ws]
    yr = [ico_ref.ico_conv(xr, wr[i], bs[i].double() if bias else None, r, stride, mode) for i in range(k)]
    gys = [torch.randn(y.shape, generator=g) for y in yr]
    torch.autograd.backward(yr, [gy.double() for gy in gys])
    ref = {'y%d' % i: yr[i].detach() for i in range(k)}
    ref['dx'] = xr.grad
    for i in range(k):
        ref['dw%d' % i] = wr[i].grad
    return (r, stride, mode, x, ws, bs, gys), ref


def _run(inputs, pair):
    from geniconet_amd.ico_conv import ico_conv, ico_conv_pair
    r, stride, mode, x, ws, bs, gys = inputs
    xg = x.cuda().requires_grad_()
    wg = [w.cuda().requires_grad_() for w in ws]
    bg = [b.cuda() if b is not None else None for b in bs]
    if pair:
        ys = ico_conv_pair(xg, wg[0], bg[0], wg[1], bg[1], r, stride, mode)
    else:
        ys = (ico_conv(xg, wg[0], bg[0], r, stride, mode),)
    torch.autograd.backward(ys, [gy.cuda() for gy in gys])
    out = {'y%d' % i: ys[i].detach() for i in range(len(ys))}
    out['dx'] = xg.grad
    for i in range(len(ys)):
        out['dw%d' % i] = wg[i].grad
    return out


def _errors(got, ref):
    return {k: rel_l2(got[k].cpu().numpy(), ref[k].numpy()) for k in ref}


# stride-1 cases run the split kernels in forward, data gradient and weight gradient (k_wgrad7<.., 1>); stride-2 cases too: their data
# gradients are the masked launches (k_conv_b3<128, BN, NW>), their weight gradients the per-tap kernel (k_wgrad_dma<.., 1>)
SINGLE = [c for c in MFMA_CASES if c[2] % 32 == 0 and c[3] % 64 == 0]
PAIRS = [c for c in PAIR_CASES if c[2] % 32 == 0 and c[3] % 64 == 0]


@pytest.mark.parametrize('case', SINGLE, ids=lambda c: 'r%d_s%d_%dx%d_b%d_%s' % c)
def test_split_arithmetic_is_fp32_grade_single(case, arith):
    inputs, ref = _conv_case(case, 11, False)
    arith.set_arith('f32')
    exact = _errors(_run(inputs, False), ref)
    arith.set_arith('bf16x3')
    got, kernels = _kernels_of(lambda: _run(inputs, False))
    split = _errors(got, ref)
    assert any(k.startswith('k_conv_b3') for k in kernels), kernels
    if case[1] == 1 and case[0] >= 2 and case[2] % 64 == 0 and case[3] % 64 == 0:
        assert 'k_wgrad7<.., 1>' in kernels, kernels
    for k in ref:
        assert split[k] < TOL, (k, split[k])
        assert split[k] <= 2.0 * exact[k] + 1e-9, (k, split[k], exact[k])


@pytest.mark.parametrize('case', PAIRS, ids=lambda c: 'r%d_s%d_%dx2x%d_b%d_%s_bias%d' % c)
def test_split_arithmetic_is_fp32_grade_pair(case, arith):
    inputs, ref = _conv_case(case, 23, True, bias=case[6])
    arith.set_arith('f32')
    exact = _errors(_run(inputs, True), ref)
    arith.set_arith('bf16x3')
    got, kernels = _kernels_of(lambda: _run(inputs, True))
    split = _errors(got, ref)
    assert any(k.startswith('k_conv_b3') for k in kernels), kernels
    if case[1] == 2:
        assert 'k_conv_b3<128, BN, NW>' in kernels and 'k_wgrad_dma<.., 1>' in kernels, kernels
    for k in ref:
        assert split[k] < TOL, (k, split[k])
        assert split[k] <= 2.0 * exact[k] + 1e-9, (k, split[k], exact[k])


@pytest.mark.parametrize('r,cin,cout,B', [(2, 256, 256, 5), (3, 256, 128, 3), (4, 128, 64, 2)])
def test_split_arithmetic_in_the_decoder_heads_dense_gemms(r, cin, cout, B, arith):
    """conv(upsample(x)) pair (models.py:58-60): z = x W and dx = g W are dense one-tap GEMMs -- both on the split kernels, against
    the float64 oracle's upsample-then-convolve."""
    from geniconet_amd import fused
    from geniconet_amd.ico_conv import IcoConvS2S, IcoUpsampleS2S
    torch.manual_seed(5)
    n = 2 ** r
    x = torch.randn(B, cin, 5 * n, 2 * n)
    convs = [IcoConvS2S(cin, cout, subdivisions=r + 1, corner_mode='average') for _ in range(2)]
    xr = x.double().requires_grad_()
    up = ico_ref.ico_upsample(xr, r, 'average')
    yr = [ico_ref.ico_conv(up, c.weight.detach().double(), c.bias.detach().double(), r + 1, 1, 'average') for c in convs]
    gys = [torch.randn(y.shape) for y in yr]
    torch.autograd.backward(yr, [g.double() for g in gys])
    ref = {'y0': yr[0].detach(), 'y1': yr[1].detach(), 'dx': xr.grad}
    ups = [IcoUpsampleS2S(cin, r, 'average').cuda() for _ in range(2)]
    cg = [c.cuda() for c in convs]

    def run():
        xg = x.cuda().requires_grad_()
        ys = fused.upconv_pair(xg, ups[0], ups[1], cg[0], cg[1])
        torch.autograd.backward(ys, [g.cuda() for g in gys])
        return {'y0': ys[0].detach(), 'y1': ys[1].detach(), 'dx': xg.grad}
    arith.set_arith('f32')
    exact = _errors(run(), ref)
    arith.set_arith('bf16x3')
    got, kernels = _kernels_of(run)
    split = _errors(got, ref)
    assert any(k.startswith('k_conv_b3_dense') for k in kernels), kernels
    for k in ref:
        assert split[k] < TOL, (k, split[k])
        assert split[k] <= 2.0 * exact[k] + 1e-9, (k, split[k], exact[k])


def test_split_arithmetic_is_deterministic_and_the_exact_path_is_untouched(arith):
    """Two runs under bf16x3 are bit-identical (stream-K adds its partial tiles in a fixed order there too); switching back to f32
    gives bit-for-bit what f32 gave before the switch."""
    inputs, _ = _conv_case((3, 1, 256, 256, 36, 'average'), 3, False)
    arith.set_arith('f32')
    a0 = _run(inputs, False)
    arith.set_arith('bf16x3')
    b0, b1 = _run(inputs, False), _run(inputs, False)
    arith.set_arith('f32')
    a1 = _run(inputs, False)
    for k in a0:
        assert torch.equal(b0[k], b1[k]), k
        assert torch.equal(a0[k], a1[k]), k
        assert not torch.equal(a0[k], b0[k]), k                       # (the modes really differ)
        assert rel_l2(b0[k].cpu().numpy(), a0[k].cpu().numpy()) < 5e-6, k


def test_special_values_survive_the_split(arith):
    """Zeros stay zero rows, huge and tiny magnitudes keep fp32's range (bf16 has fp32's exponent), a NaN input poisons its outputs."""
    from geniconet_amd.ico_conv import ico_conv
    torch.manual_seed(9)
    r, cin, cout, B = 2, 64, 128, 2
    n = 2 ** r
    x = torch.randn(B, cin, 5 * n, 2 * n, device='cuda')
    w = torch.randn(cout, cin, 7, device='cuda') / (7 * cin) ** 0.5
    arith.set_arith('bf16x3')
    assert torch.count_nonzero(ico_conv(torch.zeros_like(x), w, None, r, 1, 'average')) == 0
    for scale in (1e30, 1e-30):
        y = ico_conv(x * scale, w, None, r, 1, 'average')
        arith.set_arith('f32')
        ye = ico_conv(x * scale, w, None, r, 1, 'average')
        arith.set_arith('bf16x3')
        assert torch.isfinite(y).all()
        assert rel_l2(y.cpu().numpy(), ye.cpu().numpy()) < 5e-6, scale
    xn = x.clone()
    xn[0, 3, 5, 2] = float('nan')
    y = ico_conv(xn, w, None, r, 1, 'average')
    assert torch.isnan(y[0]).any() and not torch.isnan(y[1]).any()


def test_the_mode_is_part_of_the_abi(arith):
    assert arith.set_arith('bf16x3') in ('f32', 'bf16x3')
    assert arith.get_arith() == 'bf16x3'
    with pytest.raises(KeyError):
        arith.set_arith('bf16')
    assert arith.lib().icn_set_arith(7) == -1 and b'arithmetic mode' in arith.lib().icn_last_error()
    assert arith.get_arith() == 'bf16x3'
