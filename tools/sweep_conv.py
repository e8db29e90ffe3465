"""Where does the persistent conv GEMM lose time?  Forward launches over a grid of (Cin, Cout, level, batch), KERNEL time only
(the library's HIP-event hooks around the GEMM launch: no prologue, no Python), so that time = f(K-steps per tile, tiles per
block, N) can be read off (developer tool; GPU only).

  python tools/sweep_conv.py [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import _lib  # noqa: E402
from geniconet_amd.ico_conv import ico_conv  # noqa: E402


def kernel_time(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_start(4 * iters + 8)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ent = _lib.profile_stop()
    assert len(ent) >= 1, ent
    e = max(ent, key=lambda q: q['total_ms'])
    return e['kernel'], e['total_ms'] / e['launches'] * 1e3, e['launches'] // iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    cases = []
    for cin in (64, 128, 256, 512):
        for cout in (64, 128, 256):
            cases.append((4, 36, cin, cout))
    for B in (9, 18, 27, 36, 54, 72, 108):
        cases.append((4, B, 128, 128))
    for B in (9, 18, 36, 72):
        cases.append((4, B, 256, 256))
    for r, B in ((5, 9), (5, 36), (3, 36), (3, 144), (2, 36), (2, 576)):
        cases.append((r, B, 256, 256) if r < 5 else (r, B, 64, 64))
    print('%2s %4s %4s %4s | %-32s %3s | %8s %7s | %6s %6s %7s' % ('r', 'B', 'Cin', 'Cout', 'kernel', 'n', 'us', 'TF/s', 'tiles', 'rounds', 'Ksteps'))
    for r, B, cin, cout in cases:
        n = 2 ** r
        x = torch.randn(B, cin, 5 * n, 2 * n, device='cuda').contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 7, device='cuda') / (7 * cin) ** 0.5
        b = torch.randn(cout, device='cuda')
        with torch.no_grad():
            name, us, per = kernel_time(lambda: ico_conv(x, w, b, r, 1, 'average'), a.iters)
        flop = 2.0 * 7 * cin * cout * B * 10 * n * n
        bm, bn = [int(v) for v in name.split('<')[1].split(',')[:2]]
        occ = {(64, 128): 2, (64, 64): 3, (128, 64): 2, (128, 128): 1}[(bm, bn)]
        tiles = -(-B * 10 * n * n // bm) * (cout // bn)
        print('%2d %4d %4d %4d | %-32s %3d | %8.1f %7.1f | %6d %6.2f %7d' % (r, B, cin, cout, name, per, us, flop / us / 1e6, tiles,
                                                                          tiles / (256.0 * occ), 7 * cin // 32), flush=True)
        del x, w, b


if __name__ == '__main__':
    main()
