"""Composite upsample + conv forward (icn_upconv_fwd) against upsample -> pair convolution, decoder shapes of the AE
(developer tool; GPU only).   python tools/bench_upconv.py [--batch 36] [--R 5] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd.ico_conv import ico_conv_pair, ico_upconv_pair, ico_upsample  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=36)
ap.add_argument('--R', type=int, default=5)
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


print('forward: upsample -> pair convolution (round 1) vs icn_upconv_fwd (ICN_UPCONV_FWD=composite|dense picks the method);')
print('backward: whole autograd backward of ico_upconv_pair (icn_upconv_bwd, or the separate operators with ICN_NO_UPCONV_BWD=1)')
print('%-24s %8s | %10s %6s | %10s %6s | %6s | %10s' % ('block', 'GFLOP', 'separate us', 'TF/s', 'upconv us', 'TF/s', 'ratio', 'backward us'))
for name, cin, cout, dr in (('up1 256->2x256', 256, 256, -3), ('up2 256->2x128', 256, 128, -2), ('up3 128->2x64', 128, 64, -1)):
    r = a.R + dr
    n = 2 ** r
    x = torch.randn(a.batch, cin, 5 * n, 2 * n, device='cuda').contiguous(memory_format=torch.channels_last)
    w0, w1 = (torch.randn(cout, cin, 7, device='cuda') / (7 * cin) ** 0.5 for _ in range(2))
    b0, b1 = (torch.randn(cout, device='cuda') for _ in range(2))
    gflop = 2 * 2 * 7 * cin * cout * a.batch * 40 * n * n / 1e9
    with torch.no_grad():
        t_sep = timed(lambda: ico_conv_pair(ico_upsample(x, r, 'average'), w0, b0, w1, b1, r + 1, 1, 'average'), a.iters)
        t_cmp = timed(lambda: ico_upconv_pair(x, w0, b0, w1, b1, r, 'average'), a.iters)
    xg = x.clone().requires_grad_()
    wg = [w0.clone().requires_grad_(), w1.clone().requires_grad_()]
    bg = [b0.clone().requires_grad_(), b1.clone().requires_grad_()]
    ys = ico_upconv_pair(xg, wg[0], bg[0], wg[1], bg[1], r, 'average')
    gys = [torch.randn_like(y) for y in ys]
    t_bwd = timed(lambda: torch.autograd.grad(ys, [xg] + wg + bg, gys, retain_graph=True), a.iters)
    print('%-24s %8.2f | %10.1f %6.1f | %10.1f %6.1f | %6.3f | %10.1f' % (name, gflop, t_sep * 1e6, gflop / t_sep / 1e3, t_cmp * 1e6,
                                                                       gflop / t_cmp / 1e3, t_cmp / t_sep, t_bwd * 1e6))
