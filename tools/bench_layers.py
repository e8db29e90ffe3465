"""Per-layer timing of the ico-conv kernels at the BASELINE shapes (developer tool; GPU only).

  python tools/bench_layers.py [--batch 36] [--R 5] [--iters 5]
Times icn_conv_fwd / bwd_data / bwd_weight of every distinct conv of the ico2ico AE with HIP events around each
call (whole C-ABI call incl. weight packing / reductions) and prints algorithmic TFLOP/s.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd.ico_conv import ico_conv  # noqa: E402

# (name, Cin, Cout, level offset from R of the INPUT, stride, count in the AE)
LAYERS = [
    ('stem 3->64', 3, 64, 0, 1, 1),
    ('down1 64->128 s2', 64, 128, 0, 2, 2), ('down1 128->128', 128, 128, -1, 1, 1),
    ('down2 128->256 s2', 128, 256, -1, 2, 2), ('down2 256->256', 256, 256, -2, 1, 1),
    ('down3 256->256 s2', 256, 256, -2, 2, 2), ('down3 256->256', 256, 256, -3, 1, 1),
    ('up1 256->256', 256, 256, -2, 1, 3),
    ('up2 256->128', 256, 128, -1, 1, 2), ('up2 128->128', 128, 128, -1, 1, 1),
    ('up3 128->64', 128, 64, 0, 1, 2), ('up3 64->64', 64, 64, 0, 1, 1),
]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=36)
    ap.add_argument('--R', type=int, default=5)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--mode', default='all', choices=['all', 'fwd'])
    ap.add_argument('--data', default='randn', choices=['randn', 'zeros', 'ones'])
    a = ap.parse_args()
    tot = {'fwd': 0.0, 'bwd_data': 0.0, 'bwd_weight': 0.0}
    print('%-22s %9s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'GFLOP', 'fwd us', 'TF/s', 'dgrad us', 'TF/s', 'wgrad us', 'TF/s'))
    for name, cin, cout, dr, stride, count in LAYERS:
        if a.only and a.only not in name:
            continue
        r = a.R + dr
        n = 2 ** r
        x = torch.randn(a.batch, cin, 5 * n, 2 * n, device='cuda').contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 7, device='cuda') / (7 * cin) ** 0.5)
        b = torch.randn(cout, device='cuda')
        if a.data != 'randn':
            fill = 0.0 if a.data == 'zeros' else 1.0
            x.fill_(fill); w.fill_(fill)
        gflop = 2 * 7 * cin * cout * a.batch * 10 * (n // stride) ** 2 / 1e9
        t_f = timed(lambda: ico_conv(x, w, b, r, stride, 'average'), a.iters)
        if a.mode == 'fwd':
            print('%-22s %9.2f | %8.1f %7.1f' % (name, gflop, t_f * 1e6, gflop / t_f / 1e3))
            continue
        xg = x.clone().requires_grad_()
        y = ico_conv(xg, w, b, r, stride, 'average')
        gy = torch.randn_like(y)
        t_d = timed(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True), a.iters)
        wg, bg = w.clone().requires_grad_(), b.clone().requires_grad_()
        y2 = ico_conv(x, wg, bg, r, stride, 'average')
        t_w = timed(lambda: torch.autograd.grad(y2, (wg, bg), gy, retain_graph=True), a.iters)
        print('%-22s %9.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f   x%d' % (
            name, gflop, t_f * 1e6, gflop / t_f / 1e3, t_d * 1e6, gflop / t_d / 1e3, t_w * 1e6, gflop / t_w / 1e3, count))
        for k, t in (('fwd', t_f), ('bwd_data', t_d), ('bwd_weight', t_w)):
            tot[k] += t * count
    print('per-step conv time (ms): fwd %.2f  bwd_data %.2f  bwd_weight %.2f  total %.2f' % (
        tot['fwd'] * 1e3, tot['bwd_data'] * 1e3, tot['bwd_weight'] * 1e3, sum(tot.values()) * 1e3))


if __name__ == '__main__':
    main()
