"""Kernel-only timings (HIP events on the launch stream, icn_profile_*) of a few launches of the training step, for A/B builds of the
library (ICN_LIB_PATH; tools/build_exp.sh).  Prints one line per (launch, kernel): average us and executed TF/s.

  python tools/price_conv_features.py [--iters 20] [--tag name]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import _lib  # noqa: E402
from geniconet_amd.ico_conv import ico_conv, ico_conv_pair, ico_upconv_pair  # noqa: E402

B = 36
# (label, kind, Cin, Cout, r_in, stride, pass)
CASES = [
    ('fwd 128->128 r4', 'conv', 128, 128, 4, 1, 'fwd'), ('fwd 256->256 r3', 'conv', 256, 256, 3, 1, 'fwd'),
    ('fwd 64->64 r5', 'conv', 64, 64, 5, 1, 'fwd'), ('dgrad 128->128 r4', 'conv', 128, 128, 4, 1, 'dgrad'),
    ('fwd 256->256 r2', 'conv', 256, 256, 2, 1, 'fwd'),
    ('pair s2 dgrad 64->2x128 r5', 'pair', 64, 128, 5, 2, 'dgrad'), ('pair s2 dgrad 128->2x256 r4', 'pair', 128, 256, 4, 2, 'dgrad'),
    ('pair s2 dgrad 256->2x256 r3', 'pair', 256, 256, 3, 2, 'dgrad'),
    ('head fwd 256->2x128 r3->4', 'up', 256, 128, 3, 1, 'fwd'), ('head fwd 128->2x64 r4->5', 'up', 128, 64, 4, 1, 'fwd'),
    ('head bwd 256->2x128 r3->4', 'up', 256, 128, 3, 1, 'bwd'), ('head bwd 128->2x64 r4->5', 'up', 128, 64, 4, 1, 'bwd'),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--tag', default=os.path.basename(os.environ.get('ICN_LIB_PATH', 'libicn.so')))
    ap.add_argument('--only', default='')
    ap.add_argument('--batch', type=int, default=B)
    a = ap.parse_args()
    globals()['B'] = a.batch
    torch.manual_seed(0)
    for label, kind, cin, cout, r, stride, what in CASES:
        if a.only and a.only not in label:
            continue
        n = 2 ** r
        x = torch.randn(B, cin, 5 * n, 2 * n, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_()
        ws = [(torch.randn(cout, cin, 7, device='cuda') / (7 * cin) ** 0.5).requires_grad_() for _ in range(2)]
        bs = [torch.randn(cout, device='cuda', requires_grad=True) for _ in range(2)]
        if kind == 'conv':
            f = lambda: (ico_conv(x, ws[0], bs[0], r, stride, 'average'),)
        elif kind == 'pair':
            f = lambda: ico_conv_pair(x, ws[0], bs[0], ws[1], bs[1], r, stride, 'average')
        else:
            f = lambda: ico_upconv_pair(x, ws[0], bs[0], ws[1], bs[1], r, 'average')
        ys = f()
        gys = [torch.randn_like(y) for y in ys]
        if what == 'fwd':
            def run():
                with torch.no_grad():
                    f()
        elif what == 'dgrad':
            def run():
                torch.autograd.grad(ys, x, gys, retain_graph=True)
        else:
            def run():
                torch.autograd.grad(ys, [x] + ws, gys, retain_graph=True)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        _lib.profile_start(64 * a.iters)
        for _ in range(a.iters):
            run()
        torch.cuda.synchronize()
        for e in _lib.profile_stop():
            if e['kernel'].startswith('k_wgrad') and what != 'bwd':
                continue
            us = e['total_ms'] / e['launches'] * 1e3
            print('%-22s %-30s %-34s x%-3d %8.1f us %7.1f TF/s' % (a.tag, label, e['kernel'], e['launches'] // a.iters, us,
                                                                 e['total_flops'] / e['total_ms'] / 1e9), flush=True)
        del x, ws, bs, ys, gys


if __name__ == '__main__':
    main()
