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


# The launches the ico2ico AE really makes per step since the two branches of a residual block run as a pair:
# (name, Cin, Cout per conv, level offset of the INPUT, stride, pair?) -- each Down / Up block = one pair + one single conv
MODEL_LAUNCHES = [
    ('down1 pair 64->2x128 s2', 64, 128, 0, 2, True), ('down1 conv01 128->128', 128, 128, -1, 1, False),
    ('down2 pair 128->2x256 s2', 128, 256, -1, 2, True), ('down2 conv01 256->256', 256, 256, -2, 1, False),
    ('down3 pair 256->2x256 s2', 256, 256, -2, 2, True), ('down3 conv01 256->256', 256, 256, -3, 1, False),
    ('up1 pair 256->2x256', 256, 256, -2, 1, True), ('up1 conv01 256->256', 256, 256, -2, 1, False),
    ('up2 pair 256->2x128', 256, 128, -1, 1, True), ('up2 conv01 128->128', 128, 128, -1, 1, False),
    ('up3 pair 128->2x64', 128, 64, 0, 1, True), ('up3 conv01 64->64', 64, 64, 0, 1, False),
]


def model_launches(a):
    """--model: time the step's real MFMA launches (pairs where the model pairs), whole C-ABI calls."""
    from geniconet_amd.ico_conv import ico_conv_pair
    print('%-28s %8s | %8s %6s | %8s %6s | %8s %6s' % ('launch', 'GFLOP', 'fwd us', 'TF/s', 'dgrad us', 'TF/s', 'wgrad us', 'TF/s'))
    tot = [0.0, 0.0, 0.0]
    for name, cin, cout, dr, stride, pair in MODEL_LAUNCHES:
        if a.only and a.only not in name:
            continue
        r = a.R + dr
        n = 2 ** r
        k = 2 if pair else 1
        x = torch.randn(a.batch, cin, 5 * n, 2 * n, device='cuda').contiguous(memory_format=torch.channels_last)
        ws = [torch.randn(cout, cin, 7, device='cuda') / (7 * cin) ** 0.5 for _ in range(k)]
        bs = [torch.randn(cout, device='cuda') for _ in range(k)]
        gflop = k * 2 * 7 * cin * cout * a.batch * 10 * (n // stride) ** 2 / 1e9

        def fwd(xx, wl, bl):
            if pair:
                return ico_conv_pair(xx, wl[0], bl[0], wl[1], bl[1], r, stride, 'average')
            return (ico_conv(xx, wl[0], bl[0], r, stride, 'average'),)
        t_f = timed(lambda: fwd(x, ws, bs), a.iters)
        xg = x.clone().requires_grad_()
        ys = fwd(xg, ws, bs)
        gys = [torch.randn_like(y) for y in ys]
        t_d = timed(lambda: torch.autograd.grad(ys, xg, gys, retain_graph=True), a.iters)
        wg = [w.clone().requires_grad_() for w in ws]
        bg = [b.clone().requires_grad_() for b in bs]
        ys2 = fwd(x, wg, bg)
        t_w = timed(lambda: torch.autograd.grad(ys2, wg + bg, gys, retain_graph=True), a.iters)
        print('%-28s %8.2f | %8.1f %6.1f | %8.1f %6.1f | %8.1f %6.1f' % (name, gflop, t_f * 1e6, gflop / t_f / 1e3, t_d * 1e6,
                                                                     gflop / t_d / 1e3, t_w * 1e6, gflop / t_w / 1e3))
        for i, t in enumerate((t_f, t_d, t_w)):
            tot[i] += t
    print('per-step MFMA calls (ms): fwd %.2f  bwd_data %.2f  bwd_weight %.2f  total %.2f   (the stem and the first block\'s '
          'input gradient are not MFMA launches)' % (tot[0] * 1e3, tot[1] * 1e3, tot[2] * 1e3, sum(tot) * 1e3))


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
    ap.add_argument('--model', action='store_true', help='the launches of the AE step as the model makes them (pairs)')
    a = ap.parse_args()
    if a.model:
        return model_launches(a)
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
