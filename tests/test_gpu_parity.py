"""GPU parity proper: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): forward output within 1e-4 rel-L2 of the CPU path, fp32.  Gradients of single
operators are held to the same 1e-4; whole-network gradients (13 convs + 19 batch-norms deep) to 2e-3.
Full-size cases (I5 / batch 36, I6 / batch 8) use size-independent properties instead of the oracle.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import ico_ref, models_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture
def exact_arith():
    """Tests that compare two SCHEDULES of the exact-fp32 kernels bit for bit (stream-K vs whole tiles, wave counts, wrappers) name
    those kernels: they pin the arithmetic to f32 whatever ICN_ARITH says (the split kernels have their own file, test_gpu_arith.py)."""
    from geniconet_amd import _lib
    prev = _lib.set_arith('f32')
    yield
    _lib.set_arith(prev)


def conv_both(r, stride, cin, cout, B, mode, seed, bias=True):
    from geniconet_amd.ico_conv import ico_conv
    g = torch.Generator().manual_seed(seed)
    n = 2 ** r
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g)
    w = torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    yr = ico_ref.ico_conv(xr, wr, br, r, stride, mode)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg, wg = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    bg = b.cuda().requires_grad_() if bias else None
    yg = ico_conv(xg, wg, bg, r, stride, mode)
    yg.backward(gy.cuda())
    out = {'y': (yg, yr), 'dx': (xg.grad, xr.grad), 'dw': (wg.grad, wr.grad)}
    if bias:
        out['db'] = (bg.grad, br.grad)
    return out


# MFMA tiles: 128x128, 128x64, 64x64 forward; bwd-data swaps the roles; wgrad 128/64 in both dims
MFMA_CASES = [
    (2, 1, 64, 64, 2, 'average'), (3, 1, 64, 128, 3, 'average'), (3, 2, 128, 256, 2, 'average'),
    (4, 1, 128, 64, 2, 'average'), (2, 2, 256, 256, 3, 'average'), (3, 1, 256, 128, 2, 'zeros'),
    (4, 2, 64, 128, 1, 'zeros'), (2, 1, 96, 192, 2, 'average'), (1, 1, 64, 64, 1, 'average'),
    (5, 1, 128, 128, 1, 'average'),
    (6, 1, 64, 64, 1, 'average'),          # level 6 (BASELINE config 5; the reference hard-wires 5, models.py:108-148): up3.conv01
]
# odd-but-legal shapes: K = 32 / 64 (short tiles; stride-2 dgrad with Cout = 64 must take the register-staged kernel),
# N = 192 / 320, single sample, coarsest levels, 'zeros' corners with multi-entry transposed taps
EDGE_CASES = [
    (3, 2, 64, 64, 2, 'average'), (4, 2, 128, 64, 1, 'average'), (2, 1, 32, 64, 2, 'average'), (3, 1, 64, 192, 1, 'zeros'),
    (1, 1, 128, 128, 3, 'average'), (0, 1, 64, 64, 5, 'average'), (1, 2, 64, 128, 4, 'zeros'), (3, 2, 96, 160, 2, 'average'),
    (4, 1, 64, 320, 1, 'average'), (5, 2, 64, 128, 2, 'zeros'),
]
SCALAR_CASES = [
    (2, 1, 3, 64, 2, 'average'), (1, 1, 5, 7, 2, 'average'), (2, 2, 3, 8, 2, 'zeros'), (0, 1, 4, 4, 2, 'average'),
    (3, 1, 3, 64, 2, 'zeros'), (2, 1, 64, 3, 2, 'average'), (1, 2, 33, 65, 1, 'average'),
    (6, 1, 3, 64, 1, 'average'),           # the I6 network's stem
]


@pytest.mark.parametrize('case', MFMA_CASES + EDGE_CASES + SCALAR_CASES, ids=lambda c: 'r%d_s%d_%dx%d_b%d_%s' % c)
def test_conv_forward_backward(case):
    for k, (got, want) in conv_both(*case, seed=11).items():
        assert got.shape == want.shape, k
        assert rel_l2(got.detach().cpu().numpy(), want.detach().numpy()) < TOL, k


# Pair path (icn_conv_pair_*): conv00 / conv10 of a residual block share their input (reference models.py:37-39,59-60).
# (r, stride, cin, cout per branch, B, mode, bias): stride-1 bwd-data in split mode (r >= 4) and through the side buffer
# (r <= 3), stride 2 (row permutation + tap masks with the two dy concatenated along K), 64-channel branches (a 128-column
# tile straddles the two outputs), no bias, 'zeros' corners.
PAIR_CASES = [
    (4, 1, 128, 64, 2, 'average', True), (3, 1, 64, 128, 2, 'average', True), (3, 2, 64, 128, 2, 'average', True),
    (4, 2, 128, 256, 1, 'average', True), (2, 2, 256, 256, 3, 'average', False), (5, 1, 64, 64, 1, 'average', True),
    (2, 1, 256, 256, 2, 'zeros', True), (4, 2, 64, 64, 2, 'zeros', False), (1, 1, 128, 128, 2, 'average', True),
    (6, 2, 64, 128, 1, 'average', True),   # the I6 network's first residual block: 64 -> 2 x 128, level 6 -> 5
    (5, 2, 128, 256, 1, 'average', True),  # ... and its second: 128 -> 2 x 256, level 5 -> 4
]


@pytest.mark.parametrize('case', PAIR_CASES, ids=lambda c: 'r%d_s%d_%dx2x%d_b%d_%s_bias%d' % c)
def test_conv_pair_matches_two_oracle_convs(case):
    from geniconet_amd.ico_conv import ico_conv_pair, ico_conv_pair_supported
    r, stride, cin, cout, B, mode, bias = case
    g = torch.Generator().manual_seed(23)
    n = 2 ** r
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g)
    ws = [torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5 for _ in range(2)]
    bs = [torch.randn(cout, generator=g) if bias else None for _ in range(2)]
    xr = x.clone().requires_grad_()
    wr = [w.clone().requires_grad_() for w in ws]
    br = [b.clone().requires_grad_() if bias else None for b in bs]
    yr = [ico_ref.ico_conv(xr, wr[k], br[k], r, stride, mode) for k in range(2)]
    gys = [torch.randn(y.shape, generator=g) for y in yr]
    (yr[0] * gys[0]).sum().add((yr[1] * gys[1]).sum()).backward()          # dx = sum of the two branches' gradients
    xg = x.cuda().requires_grad_()
    wg = [w.cuda().requires_grad_() for w in ws]
    bg = [b.cuda().requires_grad_() if bias else None for b in bs]
    assert ico_conv_pair_supported(xg, wg[0], wg[1], r, stride)
    yg = ico_conv_pair(xg, wg[0], bg[0], wg[1], bg[1], r, stride, mode)
    torch.autograd.backward(yg, [gy.cuda() for gy in gys])
    pairs = {'y0': (yg[0], yr[0]), 'y1': (yg[1], yr[1]), 'dx': (xg.grad, xr.grad), 'dw0': (wg[0].grad, wr[0].grad),
             'dw1': (wg[1].grad, wr[1].grad)}
    if bias:
        pairs.update({'db0': (bg[0].grad, br[0].grad), 'db1': (bg[1].grad, br[1].grad)})
    for k, (got, want) in pairs.items():
        assert got.shape == want.shape, k
        assert rel_l2(got.detach().cpu().numpy(), want.detach().numpy()) < TOL, k


def test_conv_pair_falls_back_outside_its_limits():
    """Unequal branch widths are outside the pair path: the query says so, the op refuses loudly, and the block-level
    dispatch (fused.conv_pair) silently uses two single convolutions -- also when someone hooked one of the modules."""
    from geniconet_amd import fused
    from geniconet_amd.ico_conv import IcoConvS2S, ico_conv_pair, ico_conv_pair_supported
    torch.manual_seed(3)
    x = torch.randn(2, 64, 20, 8, device='cuda')
    a = IcoConvS2S(64, 64, subdivisions=2, corner_mode='average').cuda()
    b = IcoConvS2S(64, 128, subdivisions=2, corner_mode='average').cuda()
    assert not ico_conv_pair_supported(x, a.weight, b.weight, 2, 1)
    with pytest.raises(RuntimeError):
        ico_conv_pair(x, a.weight, a.bias, b.weight, b.bias, 2, 1, 'average')
    ya, yb = fused.conv_pair(x, a, b)
    assert torch.equal(ya, a(x)) and torch.equal(yb, b(x))
    c = IcoConvS2S(64, 64, subdivisions=2, corner_mode='average').cuda()
    paired = fused.conv_pair(x, a, c)
    seen = []
    h = c.register_forward_hook(lambda m, i, o: seen.append(1))
    hooked = fused.conv_pair(x, a, c)
    h.remove()
    assert seen == [1]                                   # the hooked module was really called
    for p_, h_ in zip(paired, hooked):
        assert rel_l2(p_.detach().cpu().numpy(), h_.detach().cpu().numpy()) < 1e-6


W7_CASES = [   # (r, stride, Cin, Cout0, Cout1 (0: single conv), B, corner mode)
    (2, 1, 64, 64, 0, 3, 'average'), (3, 1, 128, 128, 0, 5, 'average'), (4, 1, 64, 128, 0, 2, 'zeros'), (3, 1, 256, 256, 0, 36, 'average'),
    (3, 1, 128, 64, 64, 3, 'average'), (4, 1, 64, 128, 128, 7, 'average'),                       # pairs at stride 1
    (3, 2, 64, 128, 0, 2, 'average'), (4, 2, 128, 256, 256, 3, 'average'), (5, 2, 64, 128, 128, 2, 'zeros'), (3, 2, 256, 256, 256, 36, 'average'),
    (5, 1, 64, 64, 0, 1, 'average'), (2, 1, 192, 320, 0, 2, 'average'), (6, 1, 64, 64, 0, 1, 'zeros'), (3, 1, 256, 512, 512, 5, 'average'),
]


@pytest.mark.parametrize('case', W7_CASES, ids=lambda c: 'r%d_s%d_%dx%d+%d_b%d_%s' % c)
def test_all_taps_weight_gradient_kernel_equals_the_per_tap_kernel(case, exact_arith):
    """k_wgrad7 (a workgroup owns a 64 x 64 tile for all seven taps, the union of a 16-pixel patch's source rows
    staged once, positions from host tables) against k_wgrad_dma (debug flag 2048) on the same inputs: weight and bias
    gradients to 2e-6 (same products, other summation order), for single convolutions and pairs, both strides, both corner
    modes, batches that do not divide into equal splits; the profiling hooks assert which kernel ran.  Both are held to the
    oracle by the conv cases above."""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_conv, ico_conv_pair
    r, stride, cin, c0, c1, B, mode = case
    n = 2 ** r
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(B, cin, 5 * n, 2 * n, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(c, cin, 7, device='cuda', generator=g) / (7 * cin) ** 0.5).requires_grad_() for c in (c0, c1) if c]
    bs = [torch.randn(c, device='cuda', generator=g).requires_grad_() for c in (c0, c1) if c]

    def grads():
        if c1:
            ys = ico_conv_pair(x, ws[0], bs[0], ws[1], bs[1], r, stride, mode)
        else:
            ys = (ico_conv(x, ws[0], bs[0], r, stride, mode),)
        gen = torch.Generator(device='cuda').manual_seed(6)
        gys = [torch.randn(y.shape, device='cuda', generator=gen) for y in ys]
        _lib.profile_start(16)
        out = torch.autograd.grad(ys, ws + bs, gys)
        used = {e['kernel'].split('<')[0] for e in _lib.profile_stop()}
        return out, used

    old_flags = _lib.lib().icn_set_debug_flags(4096)       # stride-2 launches on k_wgrad7 too (off by default: DESIGN 4.2c)
    try:
        new, used_new = grads()
    finally:
        _lib.lib().icn_set_debug_flags(old_flags)
    old_flags = _lib.lib().icn_set_debug_flags(2048)
    try:
        old, used_old = grads()
    finally:
        _lib.lib().icn_set_debug_flags(old_flags)
    assert 'k_wgrad7' in used_new and 'k_wgrad7' not in used_old and 'k_wgrad_dma' in used_old, (used_new, used_old)
    for a, b in zip(new, old):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 2e-6


def test_tensors_beyond_2gib_take_the_fallback_kernels_and_agree_with_half_batches(exact_arith):
    """Maximum sizes: the LDS-DMA kernels address their operands through 32-bit buffer offsets, so a tensor of 2 GiB or
    more is routed to the register-staged kernels (per pass: what counts is the tensor that pass gathers from -- x for the
    forward and the weight gradient, dy for the input gradient).  r = 7, 128 -> 128 channels, batch 36: 3.0 GB each; a
    convolution acts on each sample independently, so the result must equal the two 1.5 GB half-batches run through the DMA kernels (forward,
    input gradient; the weight gradient is the sum of the halves')."""
    from geniconet_amd.ico_conv import ico_conv
    r, B, cin, cout = 7, 36, 128, 128                  # input AND output 3.0 GB: all three passes gather from a >= 2 GiB tensor
    n = 2 ** r
    g = torch.Generator(device='cuda').manual_seed(12)
    x = torch.randn(B, 5 * n, 2 * n, cin, device='cuda', generator=g).permute(0, 3, 1, 2)        # channels_last storage
    assert x.numel() * 4 >= 2 ** 31 and (x.numel() // 2) * 4 < 2 ** 31
    w = (torch.randn(cout, cin, 7, device='cuda', generator=g) / (7 * cin) ** 0.5).requires_grad_()
    b = torch.randn(cout, device='cuda', generator=g).requires_grad_()
    gy = torch.randn(B, 5 * n, 2 * n, cout, device='cuda', generator=g).permute(0, 3, 1, 2)

    def run(xs, gys):
        xs = xs.detach().requires_grad_()
        y = ico_conv(xs, w, b, r, 1, 'average')
        dx, dw, db = torch.autograd.grad(y, (xs, w, b), gys)
        return y.detach(), dx, dw, db

    def close(a, ref):
        return float((a - ref).norm() / ref.norm()) < 1e-5

    from geniconet_amd import _lib

    def kernels(fn, *a):
        _lib.profile_start(64)
        out = fn(*a)
        return out, {e['kernel'].split('<')[0] for e in _lib.profile_stop()}

    (y, dx, dw, db), big = kernels(run, x, gy)
    assert big == {'k_gather_gemm', 'k_wgrad'}, big                     # register-staged fallbacks only
    h = B // 2
    dw_sum, db_sum = torch.zeros_like(dw), torch.zeros_like(db)
    for lo in (0, h):
        (yh, dxh, dwh, dbh), half = kernels(run, x[lo:lo + h], gy[lo:lo + h])
        assert {_family(k) for k in half} == {'k_conv_dma', 'k_wgrad7'}, half      # the production kernels
        assert close(y[lo:lo + h], yh) and close(dx[lo:lo + h], dxh)
        dw_sum += dwh
        db_sum += dbh
        del yh, dxh
    assert close(dw, dw_sum) and close(db, db_sum)


# (the register-staged fallback kernels are exercised in-process through icn_set_debug_flags:
# tests/test_gpu_training_parity.py::test_register_staged_fallback_kernels_stay_correct)


# Stream-K form of the persistent GEMM (k_conv_dma_sk): the last rounds of tiles cut in K across blocks, partial tiles combined in
# a fixed order.  Shapes chosen to hit each plan at batch 36: 2.8 rounds of 64x128 tiles (r = 4, 128 -> 128), fewer tiles than
# block slots (r = 2, 256 -> 256: 360 tiles, every tile shared by 2-3 blocks), a pair (two outputs / two gradient tensors),
# 7.5 rounds of 64x64 tiles with two k-chunks per tile (r = 5, 64 -> 64, batch 12: 2.5 rounds).
def _family(kernel):
    """The LDS-DMA conv GEMM and its compile-time specialisations (round 5) under one name."""
    k = kernel.split('<')[0].rstrip('8')
    return 'k_conv_dma' if k in ('k_conv_dma', 'k_conv_dma_sk', 'k_conv_single_sk', 'k_conv_pairfwd_sk', 'k_conv_dense_sk', 'k_conv_dgrad_masked') else k


def _is_stream_k(kernel):
    """The stream-K forms of the persistent GEMM: the general kernel and its compile-time specialisations (round 5)."""
    return kernel.startswith(('k_conv_dma_sk', 'k_conv_single_sk', 'k_conv_pairfwd_sk', 'k_conv_dense_sk'))


SK_CASES = [(4, 128, 128, 36, False), (2, 256, 256, 36, False), (3, 128, 128, 36, True), (5, 64, 64, 12, False), (3, 256, 256, 7, False)]


@pytest.mark.parametrize('r,cin,cout,B,pair', SK_CASES, ids=lambda v: str(v))
def test_stream_k_equals_whole_tile_schedule(r, cin, cout, B, pair, exact_arith):
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_conv, ico_conv_pair
    g = torch.Generator().manual_seed(r * 100 + cin)
    n = 2 ** r
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda()
    ws = [(torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5).cuda() for _ in range(2)]
    bs = [torch.randn(cout, generator=g).cuda() for _ in range(2)]
    gy = [torch.randn(B, cout, 5 * n, 2 * n, generator=g).cuda() for _ in range(2)]

    def run(flags):
        nonlocal x, gy
        old = _lib.lib().icn_set_debug_flags(flags)
        try:
            xs = x.clone().requires_grad_()
            _lib.profile_start(64)
            if pair:
                ys = ico_conv_pair(xs, ws[0], bs[0], ws[1], bs[1], r, 1, 'average')
            else:
                ys = (ico_conv(xs, ws[0], bs[0], r, 1, 'average'),)
            torch.autograd.backward(ys, gy[:len(ys)])
            torch.cuda.synchronize()
            kernels = {e['kernel'].split('<')[0] for e in _lib.profile_stop()}
            return [y.detach() for y in ys] + [xs.grad], kernels
        finally:
            _lib.lib().icn_set_debug_flags(old)

    for trial in range(3):
        # new data every time: the partial-tile slots are reused from launch to launch, so a block that read a stale copy of
        # a neighbour's slot (the slots cross the XCDs' L2s) would show up here as a mismatch
        x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda()
        gy = [torch.randn(B, cout, 5 * n, 2 * n, generator=g).cuda() for _ in range(2)]
        got, k_sk = run(0)
        want, k_plain = run(128)
        assert any(_is_stream_k(k) for k in k_sk) and not any(_is_stream_k(k) for k in k_plain), (k_sk, k_plain)
        for a, b in zip(got, want):
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 2e-6, trial
        again, _ = run(0)                                    # fixed summation order: bit-identical when repeated
        for a, b in zip(got, again):
            assert torch.equal(a, b), trial
        # slow partners (debug flag 8192: every workgroup sleeps ~50 us before it parks its piece): normally a finisher finds its
        # partners' pieces long parked, here it really spins on their flags -- the hand-off (system-scope slot stores complete
        # before the flag is raised, slot loads issued after the flag was seen) must give the same bits
        slow, k_slow = run(8192)
        assert any(_is_stream_k(k) for k in k_slow)
        for a, b in zip(got, slow):
            assert torch.equal(a, b), trial
    from geniconet_amd import _lib as lib_
    assert lib_.device_status() == 0                         # nobody timed out waiting


@pytest.mark.parametrize('r,cin,cout,B', [(2, 256, 256, 36), (3, 256, 128, 36), (4, 128, 64, 9)])
def test_stream_k_in_the_decoder_heads_dense_gemms(r, cin, cout, B, exact_arith):
    """The one-tap dense GEMMs of icn_upconv_fwd / icn_upconv_bwd in their stream-K form -- since round 5 on the plain code path
    (k_conv_dense_sk: SEG = false with one tap), under debug flag 32768 on the class-major one (k_conv_dma_sk<.., true>) -- against
    the whole-tile schedule (flag 128: k_conv_dma<.., true>), new data each trial."""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_upconv_pair
    g = torch.Generator().manual_seed(r * 7 + cin)
    n = 2 ** r
    ws = [(torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5).cuda().requires_grad_() for _ in range(2)]
    bs = [torch.randn(cout, generator=g).cuda().requires_grad_() for _ in range(2)]

    def run(flags, x, gy):
        old = _lib.lib().icn_set_debug_flags(flags)
        try:
            xs = x.clone().requires_grad_()
            _lib.profile_start(64)
            ys = ico_upconv_pair(xs, ws[0], bs[0], ws[1], bs[1], r, 'average')
            grads = torch.autograd.grad(ys, [xs] + ws + bs, gy)
            torch.cuda.synchronize()
            kernels = {e['kernel'] for e in _lib.profile_stop()}
            return [y.detach() for y in ys] + list(grads), kernels
        finally:
            _lib.lib().icn_set_debug_flags(old)

    seen = set()
    for trial in range(3):
        x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda()
        gy = [torch.randn(B, cout, 10 * n, 4 * n, generator=g).cuda() for _ in range(2)]
        got, k_sk = run(0, x, gy)
        want, k_plain = run(128, x, gy)
        seen |= k_sk
        assert not any(_is_stream_k(k) for k in k_plain), k_plain
        for a, b in zip(got, want):
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 2e-6, trial
        again, _ = run(0, x, gy)
        for a, b in zip(got, again):
            assert torch.equal(a, b), trial
        cm, k_cm = run(32768, x, gy)                              # the class-major stream-K kernel: another split of K, same sums
        assert any(k.startswith('k_conv_dma_sk') and k.endswith('true>') for k in k_cm) and not any('dense' in k for k in k_cm), k_cm
        for a, b in zip(cm, want):
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 2e-6, trial
        # the heads' dense weight gradient: k_wgrad_dense is the per-tap body with the gather compiled away (flag 32768: the
        # general k_wgrad_dma walking an identity table) -- same splits, same sums: dW and dbias bit-identical
        assert any(k.startswith('k_wgrad_dense') for k in k_sk) and not any(k.startswith('k_wgrad_dense') for k in k_cm), (k_sk, k_cm)
        assert any(k.startswith('k_wgrad_dma') for k in k_cm), k_cm
        for a, b in zip(got[3:], cm[3:]):
            assert torch.equal(a, b), trial
    assert any(k.startswith('k_conv_dense_sk') for k in seen), seen


@pytest.mark.parametrize('r,cin,cout,B,tile', [(2, 256, 256, 7, '128, 128'), (3, 128, 64, 5, '128, 64'), (3, 64, 128, 5, '64, 128'),
                                               (3, 64, 64, 7, '64, 64'), (2, 128, 128, 1, '128, 128')])
def test_the_dense_weight_gradient_kernel_equals_the_general_per_tap_kernel(r, cin, cout, B, tile, exact_arith):
    """k_wgrad_dense (round 5) is k_wgrad_dma's body with the identity gather compiled in; debug flag 32768 keeps the general
    kernel walking an identity table.  Same row splits, same MFMA order: dW and dbias of both branches BIT-identical, on every
    tile instantiation (the pair's 2 * cout output channels pick the co tile), ragged row splits and a one-sample batch included."""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_upconv_pair
    g = torch.Generator().manual_seed(r * 11 + cin + cout)
    n = 2 ** r
    ws = [(torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5).cuda().requires_grad_() for _ in range(2)]
    bs = [torch.randn(cout, generator=g).cuda().requires_grad_() for _ in range(2)]
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda()
    gy = [torch.randn(B, cout, 10 * n, 4 * n, generator=g).cuda() for _ in range(2)]

    def run(flags):
        old = _lib.lib().icn_set_debug_flags(flags)
        try:
            _lib.profile_start(64)
            ys = ico_upconv_pair(x, ws[0], bs[0], ws[1], bs[1], r, 'average')
            grads = torch.autograd.grad(ys, ws + bs, gy)
            torch.cuda.synchronize()
            return grads, {e['kernel'] for e in _lib.profile_stop()}
        finally:
            _lib.lib().icn_set_debug_flags(old)

    dense, k_dense = run(0)
    general, k_general = run(32768)
    assert 'k_wgrad_dense<%s>' % tile in k_dense, k_dense
    assert 'k_wgrad_dma<%s>' % tile in k_general and not any(k.startswith('k_wgrad_dense') for k in k_general), k_general
    for a, b in zip(dense, general):
        assert torch.equal(a, b)


@pytest.mark.parametrize('r,cin,cout,B,pair', [(4, 128, 256, 36, True), (5, 64, 128, 12, True), (3, 256, 256, 36, False), (3, 256, 512, 5, True)])
def test_balanced_tile_lists_equal_the_round_robin_walk(r, cin, cout, B, pair, exact_arith):
    """Stride-2 data gradients: the tiles of such a launch run 1 or 2 of the 7 taps, and the launcher deals them to the
    workgroups by their step counts (tile lists, debug flag 512 = the round-robin walk b, b + G, ... of rounds 1-2).  A tile is
    computed whole by one workgroup either way, so the results must be BIT-identical."""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_conv, ico_conv_pair
    g = torch.Generator().manual_seed(r * 10 + cin)
    n = 2 ** r
    ws = [(torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5).cuda() for _ in range(2)]

    def run(flags, x, gy):
        old = _lib.lib().icn_set_debug_flags(flags)
        try:
            xs = x.clone().requires_grad_()
            if pair:
                ys = ico_conv_pair(xs, ws[0], None, ws[1], None, r, 2, 'average')
            else:
                ys = (ico_conv(xs, ws[0], None, r, 2, 'average'),)
            torch.autograd.backward(ys, gy[:len(ys)])
            torch.cuda.synchronize()
            return xs.grad
        finally:
            _lib.lib().icn_set_debug_flags(old)
    for trial in range(2):
        x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda()
        gy = [torch.randn(B, cout, 5 * n // 2, n, generator=g).cuda() for _ in range(2)]
        a, b = run(0, x, gy), run(512, x, gy)
        assert torch.equal(a, b), trial
        assert bool(torch.isfinite(a).all())


@pytest.mark.parametrize('r,cin,cout,B', [(2, 256, 256, 5), (3, 256, 128, 9), (4, 128, 64, 6), (2, 128, 64, 3)])
def test_lds_staged_sparse_passes_equal_the_row_per_thread_kernels(r, cin, cout, B):
    """The two sparse passes of the decoder-block head (z -> y of icn_upconv_fwd, dy -> g of icn_upconv_bwd) staged through LDS
    by patches of the pixel grid (k_upconv_scatter_lds / k_upconv_gather_lds, round 3) against the row-per-thread kernels they
    replace (debug flag 1024): same entries, same order of the sums -> equal to fp32 rounding of the FMA contraction (1e-6), for
    outputs and every gradient.  (Both are checked against the oracle by the test_upconv_* cases.)"""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_upconv_pair
    g = torch.Generator().manual_seed(r * 13 + cin)
    n = 2 ** r
    ws = [(torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5).cuda().requires_grad_() for _ in range(2)]
    bs = [torch.randn(cout, generator=g).cuda().requires_grad_() for _ in range(2)]
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda()
    gy = [torch.randn(B, cout, 10 * n, 4 * n, generator=g).cuda() for _ in range(2)]

    def run(flags):
        old = _lib.lib().icn_set_debug_flags(flags)
        try:
            xs = x.clone().requires_grad_()
            ys = ico_upconv_pair(xs, ws[0], bs[0], ws[1], bs[1], r, 'average')
            grads = torch.autograd.grad(ys, [xs] + ws + bs, gy)
            torch.cuda.synchronize()
            return [y.detach() for y in ys] + list(grads)
        finally:
            _lib.lib().icn_set_debug_flags(old)
    new, ref = run(0), run(1024)
    for a, b in zip(new, ref):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
        assert bool(torch.isfinite(a).all())


def test_the_single_convolution_wrapper_equals_the_general_stream_k_kernel(exact_arith):
    """k_conv_single_sk (round 5: src2 = dst2 = side2 = null known at compile time) against k_conv_dma_sk<.., false> (debug flag
    65536): the same instructions minus the pair forms' selects -- forward and data gradient bit-identical."""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_conv
    g = torch.Generator().manual_seed(5)
    for r, cin, cout, B in ((4, 128, 128, 36), (2, 256, 256, 36), (5, 64, 64, 12)):
        n = 2 ** r
        x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda().requires_grad_()
        w = (torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5).cuda()
        b = torch.randn(cout, generator=g).cuda()
        res = {}
        for flags in (0, 65536):
            old = _lib.lib().icn_set_debug_flags(flags)
            try:
                _lib.profile_start(64)
                y = ico_conv(x, w, b, r, 1, 'average')
                dx, = torch.autograd.grad(y, x, torch.ones_like(y) * 0.25)
                names = {e['kernel'].split('<')[0] for e in _lib.profile_stop()}
            finally:
                _lib.lib().icn_set_debug_flags(old)
            res[flags] = (y.detach().clone(), dx.clone(), names)
        assert 'k_conv_single_sk' in res[0][2] and 'k_conv_single_sk' not in res[65536][2] and 'k_conv_dma_sk' in res[65536][2], (res[0][2], res[65536][2])
        assert torch.equal(res[0][0], res[65536][0]) and torch.equal(res[0][1], res[65536][1])


EIGHT_WAVE_CASES = [(4, 128, 128, 36, False), (3, 256, 256, 7, False), (2, 256, 256, 36, False), (3, 128, 128, 5, True)]


@pytest.mark.parametrize('r,cin,cout,B,pair', EIGHT_WAVE_CASES, ids=lambda v: str(v))
def test_eight_wave_kernels_equal_the_four_wave_kernels(r, cin, cout, B, pair, exact_arith):
    """k_conv_dma8 / k_conv_dma_sk8 (the 64 x 128 tile on 2 x 4 waves of 32 x 32, built, measured and left
    off by default) against the production four-wave kernels: every output element is accumulated by one wave over the same
    K-steps in the same order, and the stream-K plan is the same, so forward outputs and data gradients are BIT-identical --
    plain convolutions (whole tiles + stream-K), a pair, and the decoder head's dense GEMMs (SEG = true) through ico_upconv_pair.
    Debug flag 16384 selects the eight-wave form in-process; the profiling hooks say which kernel ran."""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_conv, ico_conv_pair, ico_upconv_pair
    g = torch.Generator().manual_seed(77)
    n = 2 ** r
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda().requires_grad_()
    ws = [(torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5).cuda().requires_grad_() for _ in range(2)]
    bs = [torch.randn(cout, generator=g).cuda().requires_grad_() for _ in range(2)]

    def run():
        _lib.profile_start(64)
        if pair:
            ys = ico_conv_pair(x, ws[0], bs[0], ws[1], bs[1], r, 1, 'average') + ico_upconv_pair(x, ws[0], bs[0], ws[1], bs[1], r, 'average')
        else:
            ys = (ico_conv(x, ws[0], bs[0], r, 1, 'average'),)
        gys = [torch.ones_like(y) * 0.5 for y in ys]
        grads = torch.autograd.grad(ys, [x] + (ws if pair else ws[:1]), gys)
        names = {e['kernel'].split('<')[0] for e in _lib.profile_stop()}
        return [y.detach().clone() for y in ys] + [q.clone() for q in grads], names
    four, names4 = run()
    old = _lib.lib().icn_set_debug_flags(16384)
    try:
        eight, names8 = run()
    finally:
        _lib.lib().icn_set_debug_flags(old)
    assert not any(k.endswith('8') for k in names4) and any(k in ('k_conv_dma8', 'k_conv_dma_sk8') for k in names8), (names4, names8)
    for a, b in zip(four, eight):
        assert torch.equal(a, b)
    assert _lib.device_status() == 0


def test_a_lost_stream_k_partner_is_loud():
    """The failure path of the stream-K GEMM, by fault injection (debug flag 256: every finisher reports its partners lost):
    the tile becomes NaN AND the device's asynchronous status word is set, which icn_device_status returns (and clears) and
    geniconet_amd._lib.raise_on_device_status / Trainer(check_device_status=True) / bench.py turn into an exception.  Before
    this the only trace of such a failure was NaNs in the output."""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_conv
    r, cin, cout, B = SK_CASES[0][:4]
    n = 2 ** r
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g).cuda()
    w = (torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5).cuda()
    assert _lib.device_status() == 0
    y_ok = ico_conv(x, w, None, r, 1, 'average')
    assert _lib.device_status() == 0 and bool(torch.isfinite(y_ok).all())
    old = _lib.lib().icn_set_debug_flags(256)
    try:
        y_bad = ico_conv(x, w, None, r, 1, 'average')
        torch.cuda.synchronize()
    finally:
        _lib.lib().icn_set_debug_flags(old)
    assert bool(torch.isnan(y_bad).any())
    with pytest.raises(RuntimeError, match='stream-K'):
        _lib.raise_on_device_status()
    assert _lib.device_status() == 0                          # reading cleared it
    y_again = ico_conv(x, w, None, r, 1, 'average')           # and the next launch is healthy
    assert torch.equal(y_again, y_ok) and _lib.device_status() == 0


def test_no_checkpoint_is_written_after_a_kernel_reported_a_failure(tmp_path):
    """save_checkpoint checks the device status BEFORE it writes.  A lost stream-K partner (injected as above, the
    launch still in flight when save_checkpoint is called) raises and leaves NO file that a later --resume could load (run.py:342-372
    picks the newest file of the directory); after the failure has been reported the same call writes normally."""
    import os
    from geniconet_amd import _lib, models
    from geniconet_amd.ico_conv import ico_conv
    from geniconet_amd.train import Trainer, save_checkpoint
    p = models.default_params('ico2ico', subdivisions=3)
    tr = Trainer(p, 'cuda', seed=0)
    r, cin, cout, B = SK_CASES[0][:4]
    n = 2 ** r
    x = torch.randn(B, cin, 5 * n, 2 * n, device='cuda')
    w = torch.randn(cout, cin, 7, device='cuda') / (7 * cin) ** 0.5
    old = _lib.lib().icn_set_debug_flags(256)
    try:
        ico_conv(x, w, None, r, 1, 'average')                 # not synchronised: save_checkpoint has to do that itself
    finally:
        _lib.lib().icn_set_debug_flags(old)
    with pytest.raises(RuntimeError, match='stream-K'):
        save_checkpoint(tr, str(tmp_path), 1, val_loss=0.0)
    saved = os.path.join(str(tmp_path), 'savedModel')
    assert not os.path.isdir(saved) or os.listdir(saved) == []
    path = save_checkpoint(tr, str(tmp_path), 1, val_loss=0.0)
    assert path and os.listdir(saved) == [os.path.basename(path)]


def test_conv_without_bias_and_noncontiguous_input():
    from geniconet_amd.ico_conv import ico_conv
    for k, (got, want) in conv_both(2, 1, 64, 64, 2, 'average', seed=5, bias=False).items():
        assert rel_l2(got.detach().cpu().numpy(), want.detach().numpy()) < TOL, k
    x = torch.randn(2, 64, 20, 8)
    w = torch.randn(64, 64, 7) / 21
    want = ico_ref.ico_conv(x, w, None, 2, 1, 'average')
    for xg in (x.cuda(), x.cuda().to(memory_format=torch.channels_last), x.cuda().transpose(2, 3).contiguous().transpose(2, 3)):
        assert rel_l2(ico_conv(xg, w.cuda(), None, 2, 1, 'average').cpu().numpy(), want.numpy()) < TOL


def test_an_empty_batch_gives_the_oracles_empty_output_and_zero_gradients():
    """B = 0 (a ragged last batch of a loader): torch's conv2d -- hence the oracle and the reference -- returns an empty tensor
    and all-zero gradients; the C ABI rejects B < 1, so the operators answer it themselves, without a launch."""
    from geniconet_amd.ico_conv import ico_conv, ico_conv_pair, ico_upsample, ico_upconv_pair
    x = torch.zeros(0, 64, 20, 8, device='cuda', requires_grad=True)
    w = torch.randn(128, 64, 7, device='cuda', requires_grad=True)
    b = torch.randn(128, device='cuda', requires_grad=True)
    xr = torch.zeros(0, 64, 20, 8)
    for stride in (1, 2):
        y = ico_conv(x, w, b, 2, stride, 'average')
        assert y.shape == ico_ref.ico_conv(xr, w.detach().cpu(), b.detach().cpu(), 2, stride, 'average').shape
        gx, gw, gb = torch.autograd.grad(y.sum(), (x, w, b))
        assert gx.shape == x.shape and not gw.any() and not gb.any()
    y0, y1 = ico_conv_pair(x, w, b, w, b, 2, 2, 'zeros')
    assert y0.shape == y1.shape == (0, 128, 10, 4)
    u = ico_upsample(x, 2, 'average')
    assert u.shape == ico_ref.ico_upsample(xr, 2, 'average').shape and torch.autograd.grad(u.sum(), x)[0].shape == x.shape
    z0, z1 = ico_upconv_pair(x, w, b, w, b, 2, 'average')
    assert z0.shape == z1.shape == (0, 128, 40, 16)
    assert not torch.autograd.grad(z0.sum() + z1.sum(), w)[0].any()


def test_argument_errors_on_gpu():
    from geniconet_amd.ico_conv import ico_conv, ico_upsample
    x = torch.zeros(1, 4, 20, 8, device='cuda')
    with pytest.raises(ValueError, match='expected'):
        ico_conv(x, torch.zeros(4, 4, 7, device='cuda'), None, 3, 1, 'average')
    with pytest.raises(ValueError, match='weight'):
        ico_conv(x, torch.zeros(4, 5, 7, device='cuda'), None, 2, 1, 'average')
    with pytest.raises(TypeError):
        ico_upsample(x.double(), 2, 'average')


@pytest.mark.parametrize('r,C,B,mode', [(0, 4, 2, 'average'), (2, 64, 3, 'average'), (3, 5, 2, 'zeros'), (4, 128, 2, 'average'), (1, 7, 1, 'zeros'),
                                        (5, 64, 1, 'average'), (5, 8, 2, 'zeros')])        # level 5 -> 6 (I6 config)
def test_upsample_forward_backward(r, C, B, mode):
    from geniconet_amd.ico_conv import ico_upsample
    g = torch.Generator().manual_seed(3)
    n = 2 ** r
    x = torch.randn(B, C, 5 * n, 2 * n, generator=g)
    xr = x.clone().requires_grad_()
    yr = ico_ref.ico_upsample(xr, r, mode)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.cuda().requires_grad_()
    yg = ico_upsample(xg, r, mode)
    yg.backward(gy.cuda())
    assert rel_l2(yg.detach().cpu().numpy(), yr.detach().numpy()) < 1e-6
    assert rel_l2(xg.grad.cpu().numpy(), xr.grad.numpy()) < 1e-6


def _load_into_product(ref, name, R):
    from geniconet_amd import models
    net = getattr(models, name)(models.default_params(name, subdivisions=R))
    net.load_state_dict(ref.state_dict(), strict=True)
    return net.cuda()


@pytest.mark.parametrize('R,B', [(3, 3), (4, 2)])
def test_autoencoder_forward_and_gradients(R, B):
    """Whole ico2ico network, train-mode BatchNorm (batch statistics), vs the oracle network with the same weights."""
    torch.manual_seed(R)
    ref = models_ref.ico2ico(R=R).train()
    net = _load_into_product(ref, 'ico2ico', R).train()
    n = 2 ** R
    x = torch.rand(B, 3, 5 * n, 2 * n) * 1.6 - 0.8
    tgt = torch.rand(B, 3, 5 * n, 2 * n) * 1.6 - 0.8
    yr = ref(x)
    ((yr - tgt) ** 2).mean().backward()
    yg = net(x.cuda())
    ((yg - tgt.cuda()) ** 2).mean().backward()
    assert rel_l2(yg.detach().cpu().numpy(), yr.detach().numpy()) < TOL
    # A conv bias in front of a train-mode BatchNorm has an exactly-zero true gradient (BN removes the mean), so both
    # sides hold rounding noise there: errors are measured against max(||g_ref||, 1e-3 * largest gradient norm).
    gr = dict(ref.named_parameters())
    floor = 1e-3 * max(float(q.grad.norm()) for q in gr.values())
    errs = {k: float((p.grad.cpu() - gr[k].grad).norm()) / max(float(gr[k].grad.norm()), floor)
            for k, p in net.named_parameters()}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 2e-3, (worst, errs[worst], float(gr[worst].grad.norm()), floor)
    for k, q in gr.items():          # ... and those biases really are (numerically) zero on both sides
        if k.endswith('.bias') and ('conv0' in k or 'conv1' in k or k == 'encoder.0.bias'):
            assert float(q.grad.norm()) < floor and float(dict(net.named_parameters())[k].grad.norm()) < floor, k
    # BatchNorm running statistics were updated identically
    sr = ref.state_dict()
    for k, v in net.state_dict().items():
        if 'running' in k:
            assert rel_l2(v.cpu().numpy(), sr[k].numpy()) < 1e-4, k


def test_vae_encode_decode():
    R, B = 3, 3
    torch.manual_seed(9)
    ref = models_ref.ico2ico_vae(R=R).train()
    net = _load_into_product(ref, 'ico2ico_vae', R).train()
    n = 2 ** R
    x = torch.rand(B, 3, 5 * n, 2 * n) - 0.5
    mu_r, lv_r = ref.encode(x)
    mu_g, lv_g = net.encode(x.cuda())
    assert rel_l2(mu_g.detach().cpu().numpy(), mu_r.detach().numpy()) < TOL
    assert rel_l2(lv_g.detach().cpu().numpy(), lv_r.detach().numpy()) < TOL
    z = torch.randn_like(mu_r)
    assert rel_l2(net.decode(z.cuda()).detach().cpu().numpy(), ref.decode(z).detach().numpy()) < TOL
    out, mu, lv = net(x.cuda())
    assert out.shape == (B, 3, 5 * n, 2 * n) and mu.shape == (B, 512, 5 * n // 8, 2 * n // 8)


# ------------------------------------------------------------------ full-size, size-independent properties
FULL = [('I5_b36_128to64', 5, 36, 128, 64, 1), ('I5_b36_64to128_s2', 5, 36, 64, 128, 2), ('I6_b8_128to64', 6, 8, 128, 64, 1),
        ('I5_b36_stem_3to64', 5, 36, 3, 64, 1)]


@pytest.mark.parametrize('name,r,B,cin,cout,stride', FULL, ids=[f[0] for f in FULL])
def test_full_size_linearity_and_chart_equivariance(name, r, B, cin, cout, stride):
    """At BASELINE's sizes: conv(a x1 + x2) - conv(0) == a (conv(x1) - conv(0)) + (conv(x2) - conv(0)), and rolling
    the five charts of the input rolls the charts of the output (72-degree rotation about the pole axis)."""
    from geniconet_amd.ico_conv import ico_conv
    g = torch.Generator(device='cuda').manual_seed(1)
    n = 2 ** r
    x1 = torch.randn(B, cin, 5 * n, 2 * n, device='cuda', generator=g)
    x2 = torch.randn(B, cin, 5 * n, 2 * n, device='cuda', generator=g)
    w = torch.randn(cout, cin, 7, device='cuda', generator=g) / (7 * cin) ** 0.5
    b = torch.randn(cout, device='cuda', generator=g)
    f = lambda t: ico_conv(t, w, b, r, stride, 'average')
    y0 = f(torch.zeros_like(x1))
    assert float((y0 - b[None, :, None, None]).abs().max()) == 0.0
    y1, y2, y12 = f(x1), f(x2), f(1.5 * x1 + x2)
    lin = 1.5 * (y1 - y0) + (y2 - y0) + y0
    assert float((y12 - lin).norm() / lin.norm()) < 1e-5
    ys = f(torch.roll(x1, n, dims=2))
    d = (ys - torch.roll(y1, n // stride, dims=2)).abs()
    assert float(d.max()) < 1e-5
    # With every tile computed whole by one workgroup (stream-K off: it cuts some tiles' K range at positions that depend on
    # the tile, i.e. on the chart) it is the same arithmetic on another chart; only the pole mean sums its 5 corners in a
    # rotated order (1-ulp effects).
    from geniconet_amd import _lib
    old = _lib.lib().icn_set_debug_flags(128)
    try:
        d = (f(torch.roll(x1, n, dims=2)) - torch.roll(f(x1), n // stride, dims=2)).abs()
    finally:
        _lib.lib().icn_set_debug_flags(old)
    assert float(d.max()) < 1e-5
    assert int((d > 0).sum()) <= B * cout * 10 * 3


def test_full_size_backward_is_the_adjoint():
    """<conv(x), gy> == <x, conv^T(gy)> at I5 / batch 36 (bias-free): bwd-data is the exact transpose."""
    from geniconet_amd.ico_conv import ico_conv, ico_upsample
    g = torch.Generator(device='cuda').manual_seed(2)
    for stride, cin, cout in ((1, 128, 64), (2, 64, 128)):
        x = torch.randn(36, cin, 160, 64, device='cuda', generator=g, requires_grad=True)
        w = torch.randn(cout, cin, 7, device='cuda', generator=g, requires_grad=True) / 30
        y = ico_conv(x, w, None, 5, stride, 'average')
        gy = torch.randn(y.shape, device='cuda', generator=g)
        dx, dw = torch.autograd.grad(y, (x, w), gy)
        lhs = float((y.detach().double() * gy.double()).sum())
        assert abs(lhs - float((x.detach().double() * dx.double()).sum())) < 1e-6 * abs(lhs) + 1e-3
        assert abs(lhs - float((w.detach().double() * dw.double()).sum())) < 1e-5 * abs(lhs) + 1e-3   # y is linear in w too
    x = torch.randn(36, 128, 80, 32, device='cuda', generator=g, requires_grad=True)
    u = ico_upsample(x, 4, 'average')
    gu = torch.randn(u.shape, device='cuda', generator=g)
    (dx,) = torch.autograd.grad(u, x, gu)
    lhs = float((u.detach().double() * gu.double()).sum())
    assert abs(lhs - float((x.detach().double() * dx.double()).sum())) < 1e-6 * abs(lhs) + 1e-3


@pytest.mark.parametrize('C,dual', [(64, False), (128, True), (256, True), (512, False)])
def test_fused_bn_relu_matches_torch_builtins(C, dual):
    """icn_bn_* vs the chain of torch builtins it replaces (models.py:36-40): outputs, input/affine gradients,
    running statistics and num_batches_tracked."""
    from geniconet_amd import fused
    torch.manual_seed(C)
    mk = lambda: torch.nn.BatchNorm2d(C).cuda().train()
    bn_a, bn_b, rf_a, rf_b = mk(), mk(), mk(), mk()
    for m, r in ((bn_a, rf_a), (bn_b, rf_b)):
        with torch.no_grad():
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
        r.load_state_dict(m.state_dict())
    a = (torch.randn(3, C, 40, 16, device='cuda') * 2 + 0.3).contiguous(memory_format=torch.channels_last)
    b = torch.randn(3, C, 40, 16, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(3, C, 40, 16, device='cuda')
    a1, b1, a2, b2 = (t.clone().requires_grad_() for t in (a, b, a, b))
    assert fused.can_fuse(a1, bn_a, bn_b)
    y1 = fused.bn_add_relu(a1, bn_a, b1, bn_b) if dual else fused.bn_relu(a1, bn_a)
    y2 = torch.relu(rf_a(a2) + rf_b(b2)) if dual else torch.relu(rf_a(a2))
    y1.backward(gy); y2.backward(gy)
    pairs = [(y1, y2), (a1.grad, a2.grad), (bn_a.weight.grad, rf_a.weight.grad), (bn_a.bias.grad, rf_a.bias.grad),
             (bn_a.running_mean, rf_a.running_mean), (bn_a.running_var, rf_a.running_var)]
    if dual:
        pairs += [(b1.grad, b2.grad), (bn_b.weight.grad, rf_b.weight.grad), (bn_b.bias.grad, rf_b.bias.grad),
                  (bn_b.running_var, rf_b.running_var)]
    for got, want in pairs:
        assert rel_l2(got.detach().cpu().numpy(), want.detach().cpu().numpy()) < 2e-5
    assert int(bn_a.num_batches_tracked) == int(rf_a.num_batches_tracked) == 1
    bn_a.eval()
    assert not fused.can_fuse(a1, bn_a)


def test_bn_stats_of_two_tensors_in_one_pass_equal_two_single_passes():
    """icn_bn_stats2 (the two inputs of a residual BatchNorm in one partial + one finalize launch) runs the same arithmetic per
    tensor as icn_bn_stats: statistics and running statistics must be bit-identical."""
    from geniconet_amd import _lib
    L = _lib.lib()
    M, C = 36 * 640 + 13, 128
    g = torch.Generator().manual_seed(4)
    a = (torch.randn(M, C, generator=g) * 2 + 3).cuda()
    b = (torch.randn(M, C, generator=g) * 0.1 - 50).cuda()
    ws = torch.empty(L.icn_bn_workspace_floats(M, C), device='cuda')
    st = torch.cuda.current_stream().cuda_stream

    def fresh():
        return [torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.empty(2 * C, device='cuda')]
    one_a, one_b, two_a, two_b = fresh(), fresh(), fresh(), fresh()
    _lib.check(L.icn_bn_stats(a.data_ptr(), M, C, 1e-5, 0.1, one_a[0].data_ptr(), one_a[1].data_ptr(), one_a[2].data_ptr(),
                              ws.data_ptr(), st), 'icn_bn_stats')
    _lib.check(L.icn_bn_stats(b.data_ptr(), M, C, 1e-3, 0.2, one_b[0].data_ptr(), one_b[1].data_ptr(), one_b[2].data_ptr(),
                              ws.data_ptr(), st), 'icn_bn_stats')
    _lib.check(L.icn_bn_stats2(a.data_ptr(), b.data_ptr(), M, C, 1e-5, 0.1, two_a[0].data_ptr(), two_a[1].data_ptr(),
                               two_a[2].data_ptr(), 1e-3, 0.2, two_b[0].data_ptr(), two_b[1].data_ptr(), two_b[2].data_ptr(),
                               ws.data_ptr(), st), 'icn_bn_stats2')
    for x, y in zip(one_a + one_b, two_a + two_b):
        assert torch.equal(x, y)
    ref = torch.nn.functional.batch_norm(b.double().cpu(), None, None, training=True, eps=1e-3)
    got = (b.double().cpu() - two_b[2][:C].double().cpu()) * two_b[2][C:].double().cpu()
    assert float((got - ref).abs().max()) < 1e-3          # mean -50, std 0.1: the shifted sums keep the variance


def test_fused_head_matches_conv1x1_tanh():
    from geniconet_amd import fused
    torch.manual_seed(4)
    seq = torch.nn.Sequential(torch.nn.Conv2d(64, 3, kernel_size=(1, 1)), torch.nn.Tanh()).cuda()
    ref = torch.nn.Sequential(torch.nn.Conv2d(64, 3, kernel_size=(1, 1)), torch.nn.Tanh()).cuda()
    ref.load_state_dict(seq.state_dict())
    x = torch.randn(5, 64, 40, 16, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(5, 3, 40, 16, device='cuda')
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    assert fused.can_fuse_head(x1, seq)
    y1, y2 = fused.head(x1, seq), ref(x2)
    y1.backward(gy); y2.backward(gy)
    for got, want in ((y1, y2), (x1.grad, x2.grad), (seq[0].weight.grad, ref[0].weight.grad), (seq[0].bias.grad, ref[0].bias.grad)):
        assert got.shape == want.shape
        assert rel_l2(got.detach().cpu().numpy(), want.detach().cpu().numpy()) < 2e-5
    seq.register_forward_hook(lambda *a: None)
    assert not fused.can_fuse_head(x1, seq)


def test_vae_training_step_and_i6_forward():
    """BASELINE configs 4 and 5 in miniature: a VAE step with the P2P+KLD loss, and the AE built at subdivisions=6."""
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer
    p = models.default_params('ico2ico_vae', subdivisions=4)
    tr = Trainer(p, 'cuda', seed=0)
    x, t = data.synthetic_batch(3, 4, seed=2, device='cuda')
    l0 = float(tr.step(x, t))
    assert np.isfinite(l0)
    rec, kld = tr.criterion.get_last_losses()[0], tr.criterion.get_last_losses()[3]
    assert np.isfinite(rec) and np.isfinite(kld)
    net6 = models.ico2ico(models.default_params('ico2ico', subdivisions=6)).cuda()
    x6, _ = data.synthetic_batch(1, 6, seed=3, device='cuda')
    with torch.no_grad():
        y6 = net6(x6)
    assert y6.shape == (1, 3, 320, 128) and bool(torch.isfinite(y6).all())


def test_training_step_runs_and_decreases_loss():
    """A few steps of the mirrored run.py loop on one small batch: loss is finite and goes down."""
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer
    p = models.default_params('ico2ico', subdivisions=4)
    p['ico2ico'].update(lr=1e-3, lr_base=1e-4, lr_max=1e-3)
    tr = Trainer(p, 'cuda', seed=0)
    x, t = data.synthetic_batch(4, 4, seed=1, device='cuda')
    losses_seen = [float(tr.step(x, t)) for _ in range(8)]
    assert all(np.isfinite(losses_seen)) and losses_seen[-1] < losses_seen[0]


# ---- point-to-point loss on the HIP path (icn_p2p_loss_*) vs the numpy oracle (oracle/loss_ref.py) ---------------------
def _loss_case(r, B, seed):
    g = torch.Generator().manual_seed(seed)
    n = 2 ** r
    pred = torch.randn(B, 3, 5 * n, 2 * n, generator=g)
    target = torch.randn(B, 9, 10 * n * n + 2, generator=g)
    return pred, target


@pytest.mark.parametrize('r,B', [(0, 3), (1, 2), (2, 2), (3, 2)])
def test_hip_loss_terms_match_numpy_oracle(r, B):
    """r = 0: all 12 vertices are five-valent and every pixel is a pole corner."""
    from geniconet_amd.losses import P2P_Loss
    from oracle import loss_ref
    pred, target = _loss_case(r, B, 31 + r)
    crit = P2P_Loss(r, 0.7, 0.2, 0.1).cuda()
    with torch.no_grad():
        total = crit(pred.cuda().contiguous(memory_format=torch.channels_last), target.cuda())
    want = loss_ref.p2p_terms(pred.numpy(), target.numpy(), r)
    got = [float(crit.last_loss_mse), float(crit.last_loss_cos), float(crit.last_loss_lap)]
    for k, (a, b) in enumerate(zip(got, want)):
        assert abs(a - b) <= 2e-5 * max(abs(b), 1e-3), (k, a, b)
    assert abs(float(total) - loss_ref.p2p_loss(pred.numpy(), target.numpy(), r, 0.7, 0.2, 0.1)) <= 2e-5 * abs(float(total))


@pytest.mark.parametrize('factors', [(0.7, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), (0.6, 0.2, 0.2)],
                         ids=lambda f: 'f%g_%g_%g' % f)
@pytest.mark.parametrize('r,B', [(0, 2), (1, 3), (2, 2), (3, 2)])
def test_hip_loss_gradient_matches_oracle(r, B, factors):
    """Each term alone and the VAE's 0.6 / 0.2 / 0.2 mixture (reference run.py:694-696); the oracle's analytic gradient is
    itself pinned by finite differences in tests/test_oracle_properties.py."""
    from geniconet_amd.losses import P2P_Loss
    from oracle import loss_ref
    pred, target = _loss_case(r, B, 47 + r)
    crit = P2P_Loss(r, *factors).cuda()
    x = pred.cuda().requires_grad_()
    (3.0 * crit(x, target.cuda())).backward()                      # upstream gradient 3
    want = 3.0 * loss_ref.p2p_grad(pred.numpy(), target.numpy(), r, *factors)
    assert rel_l2(x.grad.cpu().numpy(), want) < 5e-5


def test_hip_loss_agrees_with_the_torch_formulation(monkeypatch):
    """Values and gradients of the HIP kernels against torch autograd through the torch formulation (what CPU tensors run)
    on the same device, VAE factors."""
    from geniconet_amd import losses
    pred, target = _loss_case(3, 2, 5)
    crit = losses.P2P_Loss(3, 0.6, 0.2, 0.2).cuda()
    t = target.cuda()
    x = pred.cuda().requires_grad_()
    crit(x, t).backward()
    hip_terms = [float(crit.last_loss_mse), float(crit.last_loss_cos), float(crit.last_loss_lap), float(crit.last_loss_total)]
    monkeypatch.setattr(losses, '_NO_HIP_LOSS', True)
    y = pred.cuda().requires_grad_()
    assert not crit._hip_path(y, t)
    crit(y, t).backward()
    torch_terms = [float(crit.last_loss_mse), float(crit.last_loss_cos), float(crit.last_loss_lap), float(crit.last_loss_total)]
    for a_, b_ in zip(hip_terms, torch_terms):
        assert abs(a_ - b_) <= 2e-5 * max(abs(b_), 1e-3), (hip_terms, torch_terms)
    assert rel_l2(x.grad.cpu().numpy(), y.grad.cpu().numpy()) < 5e-5


def test_hip_loss_full_size_scaling_property():
    """I5 / batch 36 (the bench shape): scaling positions by s scales the position term by s^2 and, with the Laplacian
    targets scaled alike, the Laplacian term by s^2; the normal term does not move."""
    from geniconet_amd.losses import P2P_Loss
    pred, target = _loss_case(5, 36, 9)
    crit = P2P_Loss(5, 1.0, 0.0, 0.0).cuda()
    x, t = pred.cuda(), target.cuda()
    with torch.no_grad():
        crit(x, t)
        a = [float(crit.last_loss_mse), float(crit.last_loss_cos), float(crit.last_loss_lap)]
        t2 = t.clone()
        t2[:, 0:3] *= 2.0
        t2[:, 6:9] *= 2.0
        crit(2.0 * x, t2)
        b = [float(crit.last_loss_mse), float(crit.last_loss_cos), float(crit.last_loss_lap)]
    assert abs(b[0] - 4 * a[0]) <= 1e-5 * abs(4 * a[0])
    assert abs(b[1] - a[1]) <= 1e-5 * abs(a[1])
    assert abs(b[2] - 4 * a[2]) <= 1e-5 * abs(4 * a[2])


# ---- inference halves (reference models.py:234-252,302-340, run.py:360-367,499-536) ------------------------
@pytest.mark.parametrize('train_mode_bn', [False, True], ids=['eval', 'bn_train_mode'])
def test_inference_halves_compose_to_the_full_model(train_mode_bn):
    """decoder-half(encoder-half(x)) == full model(x), halves restored from the full model by the reference's key filter;
    also with BatchNorm in training mode, as experiment_test runs it (run.py:516)."""
    from geniconet_amd import models

    def restore(half, saved):
        md = half.state_dict()
        half.load_state_dict({k: v for k, v in saved.items() if k in md})
        return half.cuda().train(train_mode_bn)

    R, B = 3, 2
    n = 2 ** R
    x = torch.randn(B, 3, 5 * n, 2 * n, generator=torch.Generator().manual_seed(8)).cuda()
    with torch.no_grad():
        p = models.default_params('ico2ico', subdivisions=R)
        torch.manual_seed(0)
        full = models.ico2ico(p).cuda().train(train_mode_bn)
        saved = full.state_dict()
        enc, dec = restore(models.ico2enc(p), saved), restore(models.enc2ico(p), saved)
        assert rel_l2(dec(enc(x)).cpu().numpy(), full(x).cpu().numpy()) < 1e-6

        p = models.default_params('ico2ico_vae', subdivisions=R)
        torch.manual_seed(0)
        full = models.ico2ico_vae(p).cuda().train(train_mode_bn)
        saved = full.state_dict()
        enc, dec = restore(models.ico2enc_vae(p), saved), restore(models.enc2ico_vae(p), saved)
        mu_f, lv_f = full.encode(x)
        mu_h, lv_h = enc(x)
        assert rel_l2(mu_h.cpu().numpy(), mu_f.cpu().numpy()) < 1e-6 and rel_l2(lv_h.cpu().numpy(), lv_f.cpu().numpy()) < 1e-6
        assert rel_l2(dec(mu_h)[0].cpu().numpy(), full.decode(mu_f).cpu().numpy()) < 1e-6


def test_device_resident_dataset_feeds_the_hip_path(tmp_path):
    """data.IcoDataset on the device + train.train_epoch / validate: the first batch's loss is the one Trainer.step gives on
    the same tensors (twin trainer, same seed), inputs arrive channels_last, validation runs in eval mode."""
    from geniconet_amd import data, models, train
    R = 3
    _, t = data.synthetic_batch(6, R, seed=21)
    for k in range(6):
        data.save_sample(str(tmp_path / ('s%d.npz' % k)), t[k].numpy())
    ds = data.IcoDataset(str(tmp_path), R, device='cuda')
    assert ds.targets.is_cuda and len(ds) == 6
    img, lbl = next(iter(ds.batches(4)))
    assert img.is_contiguous(memory_format=torch.channels_last) and torch.equal(lbl.cpu(), t[:4])
    p = models.default_params('ico2ico', subdivisions=R)
    a, b = train.Trainer(p, 'cuda', seed=7), train.Trainer(p, 'cuda', seed=7)
    want = float(b.step(img, lbl))
    got = train.train_epoch(a, ds, 4, shuffle=False)
    assert got.shape == (2,) and abs(float(got[0]) - want) <= 1e-6 * abs(want)
    v = train.validate(a, ds.subset([4, 5]), 2)
    assert np.isfinite(v) and a.model.training


# ---- composite upsample + conv (icn_upconv_fwd): forward from the coarse tensor, backward of the separate operators -----------
# (r_in, cin, cout per branch, B, mode, bias): every level of the decoder incl. the tiny ones (almost all rows irregular),
# 64-channel branches (a 128-column tile straddles the two outputs), 'zeros' poles, no bias, sample counts that do not fill
# a row segment
UPCONV_CASES = [(0, 64, 64, 2, 'average', True), (1, 64, 128, 3, 'average', True), (2, 256, 256, 2, 'average', True),
                (3, 128, 64, 2, 'average', True), (2, 64, 64, 5, 'zeros', True), (3, 64, 128, 1, 'zeros', False),
                (4, 128, 64, 1, 'average', True), (2, 128, 192, 3, 'average', False),
                (4, 256, 128, 1, 'average', True),      # the I6 network's decoder heads: level 4 -> 5 ...
                (5, 128, 64, 1, 'average', True)]       # ... and the I6 network's last decoder head, level 5 -> 6, all gradients


@pytest.mark.parametrize('case', UPCONV_CASES, ids=lambda c: 'r%d_%dx2x%d_b%d_%s_bias%d' % c)
def test_upconv_pair_matches_oracle_upsample_then_convs(case):
    """(conv0(up(x)), conv1(up(x))) of the reference's decoder block (models.py:58-60): outputs and all gradients against the
    oracle's separate operators, and the routing (the composite kernel really ran: one k_conv_dma launch in the forward)."""
    from geniconet_amd import _lib
    from geniconet_amd.ico_conv import ico_upconv_pair, ico_upconv_pair_supported
    r, cin, cout, B, mode, bias = case
    g = torch.Generator().manual_seed(101 + r)
    n = 2 ** r
    x = torch.randn(B, cin, 5 * n, 2 * n, generator=g)
    ws = [torch.randn(cout, cin, 7, generator=g) / (7 * cin) ** 0.5 for _ in range(2)]
    bs = [torch.randn(cout, generator=g) if bias else None for _ in range(2)]
    xr = x.clone().requires_grad_()
    wr = [w.clone().requires_grad_() for w in ws]
    br = [b.clone().requires_grad_() if bias else None for b in bs]
    up = ico_ref.ico_upsample(xr, r, mode)
    yr = [ico_ref.ico_conv(up, wr[k], br[k], r + 1, 1, mode) for k in range(2)]
    gys = [torch.randn(y.shape, generator=g) for y in yr]
    (yr[0] * gys[0]).sum().add((yr[1] * gys[1]).sum()).backward()
    xg = x.cuda().requires_grad_()
    wg = [w.cuda().requires_grad_() for w in ws]
    bg = [b.cuda().requires_grad_() if bias else None for b in bs]
    assert ico_upconv_pair_supported(xg, wg[0], wg[1], r)
    _lib.profile_start(64)
    yg = ico_upconv_pair(xg, wg[0], bg[0], wg[1], bg[1], r, mode)
    prof = _lib.profile_stop()
    assert sum(e['launches'] for e in prof) == 1 and (prof[0]['kernel'].startswith(('k_conv_dense', 'k_conv_b3_dense')) or
                                                      (prof[0]['kernel'].startswith('k_conv_dma') and 'true' in prof[0]['kernel'])), prof
    torch.autograd.backward(yg, [gy.cuda() for gy in gys])
    pairs = {'y0': (yg[0], yr[0]), 'y1': (yg[1], yr[1]), 'dx': (xg.grad, xr.grad), 'dw0': (wg[0].grad, wr[0].grad),
             'dw1': (wg[1].grad, wr[1].grad)}
    if bias:
        pairs.update({'db0': (bg[0].grad, br[0].grad), 'db1': (bg[1].grad, br[1].grad)})
    for k, (got, want) in pairs.items():
        assert got.shape == want.shape, k
        assert rel_l2(got.detach().cpu().numpy(), want.detach().numpy()) < TOL, k


def test_upconv_full_size_equals_the_separate_operators():
    """BASELINE size (I5 / batch 36: the 128 -> 2 x 64 block from level 4 to 5, and I6 / batch 8): the composite forward
    against upsample -> pair convolution on the same device (both HIP paths; the small cases above pin both to the oracle)."""
    from geniconet_amd.ico_conv import ico_conv_pair, ico_upconv_pair, ico_upsample
    g = torch.Generator(device='cuda').manual_seed(6)
    for r, B, cin, cout in ((4, 36, 128, 64), (2, 36, 256, 256), (5, 8, 128, 64)):
        n = 2 ** r
        x = torch.randn(B, cin, 5 * n, 2 * n, device='cuda', generator=g)
        w0, w1 = (torch.randn(cout, cin, 7, device='cuda', generator=g) / (7 * cin) ** 0.5 for _ in range(2))
        b0, b1 = (torch.randn(cout, device='cuda', generator=g) for _ in range(2))
        with torch.no_grad():
            a0, a1 = ico_upconv_pair(x, w0, b0, w1, b1, r, 'average')
            c0, c1 = ico_conv_pair(ico_upsample(x, r, 'average'), w0, b0, w1, b1, r + 1, 1, 'average')
        assert float((a0 - c0).norm() / c0.norm()) < 1e-5 and float((a1 - c1).norm() / c1.norm()) < 1e-5
