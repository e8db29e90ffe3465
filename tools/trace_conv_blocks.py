"""Where a stream-K conv GEMM launch spends its time, per workgroup (developer tool; GPU only): icn_debug_trace makes every
workgroup of k_conv_dma_sk write constant-clock (100 MHz) timestamps -- entry, first-tile tables built, ring filled, first
split-phase segment, exit -- plus its partner-wait time and XCC id.  Prints the launch-relative distribution of each.

  python tools/trace_conv_blocks.py [--r 4 --batch 36 --cin 128 --cout 128] [--mode fwd|dgrad]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import _lib  # noqa: E402
from geniconet_amd.ico_conv import ico_conv  # noqa: E402


def pct(v, name, unit='us'):
    v = np.asarray(v, dtype=np.float64)
    print('  %-34s min %8.2f  p10 %8.2f  median %8.2f  p90 %8.2f  max %8.2f %s' % (
        name, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max(), unit))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--r', type=int, default=4)
    ap.add_argument('--batch', type=int, default=36)
    ap.add_argument('--cin', type=int, default=128)
    ap.add_argument('--cout', type=int, default=128)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--dump', default='')
    ap.add_argument('--mode', default='fwd', choices=['fwd', 'dgrad', 'wgrad', 'upfwd'])
    ap.add_argument('--stride', type=int, default=1)
    ap.add_argument('--pair', action='store_true')
    a = ap.parse_args()
    L = _lib.lib()
    n = 2 ** a.r
    x = torch.randn(a.batch, a.cin, 5 * n, 2 * n, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(a.cout, a.cin, 7, device='cuda') / (7 * a.cin) ** 0.5
    b = torch.randn(a.cout, device='cuda')
    buf = torch.zeros(16384 * 8, dtype=torch.int64, device='cuda')
    from geniconet_amd.ico_conv import ico_conv_pair
    w2 = torch.randn(a.cout, a.cin, 7, device='cuda') / (7 * a.cin) ** 0.5
    xs = x.clone().requires_grad_(a.mode == 'dgrad')
    ws = [w.clone().requires_grad_(a.mode == 'wgrad'), w2.clone().requires_grad_(a.mode == 'wgrad')]
    bs = [b.clone().requires_grad_(a.mode == 'wgrad'), b.clone().requires_grad_(a.mode == 'wgrad')]
    if a.mode == 'upfwd':
        from geniconet_amd.ico_conv import ico_upconv_pair

        def run():
            with torch.no_grad():
                ico_upconv_pair(x, w, b, w2, b, a.r, 'average')      # the dense z-GEMM is the call's only stream-K launch
    elif a.mode != 'fwd':
        if a.pair:
            ys = ico_conv_pair(xs, ws[0], bs[0], ws[1], bs[1], a.r, a.stride, 'average')
        else:
            ys = (ico_conv(xs, ws[0], bs[0], a.r, a.stride, 'average'),)
        gys = [torch.randn_like(y) for y in ys]

        def run():
            torch.autograd.grad(ys, [xs] if a.mode == 'dgrad' else ws[:len(ys)] + bs[:len(ys)], gys, retain_graph=True)
    else:
        def run():
            with torch.no_grad():
                if a.pair:
                    ico_conv_pair(x, w, b, w2, b, a.r, a.stride, 'average')
                else:
                    ico_conv(x, w, b, a.r, a.stride, 'average')
    if True:
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        for rep in range(a.reps):
            buf.zero_()
            torch.cuda.synchronize()
            L.icn_debug_trace(buf.data_ptr(), buf.numel())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            L.icn_debug_trace(None, 0)
            t = buf.cpu().numpy().reshape(-1, 8)
            t = t[t[:, 0] != 0]
            G = len(t)
            if a.mode == 'wgrad':
                t0 = t[:, 0].min()
                ent, ext = (t[:, 0] - t0) / 100.0, (t[:, 4] - t0) / 100.0
                dur = ext - ent
                print('rep %d: %d wgrad workgroups, call %.1f us by events, first entry -> last exit %.1f us' % (
                    rep, G, e0.elapsed_time(e1) * 1e3, ext.max()))
                pct(dur, 'workgroup duration')
                pct(ent, 'entry at')
                pct(ext, 'exit at')
                # resident workgroups over time -> utilisation of the block slots
                ev = sorted([(v, 1) for v in ent] + [(v, -1) for v in ext])
                cur, last, area, peak = 0, 0.0, 0.0, 0
                for tt, d in ev:
                    area += cur * (tt - last)
                    last = tt
                    cur += d
                    peak = max(peak, cur)
                print('  peak resident %d; mean resident / peak over the span = %.1f %%' % (peak, 100.0 * area / (peak * ext.max())))
                for lo in range(0, int(ext.max()) + 1, 20):
                    n = int(((ent <= lo) & (ext > lo)).sum())
                    print('    t = %4d us: %4d resident' % (lo, n))
                continue
            t0 = t[:, 0].min()
            us = lambda col: (t[:, col] - t0) / 100.0
            end = us(4)
            print('rep %d: %d workgroups traced, call (prologue + GEMM) %.1f us by events, GEMM first entry -> last exit %.1f us'
                  % (rep, G, e0.elapsed_time(e1) * 1e3, end.max()))
            pct(us(0), 'entry (dispatch skew)')
            pct((t[:, 1] - t[:, 0]) / 100.0, 'entry -> first tables built')
            pct((t[:, 2] - t[:, 1]) / 100.0, 'tables -> ring filled')
            pct(us(2), 'first MFMA at')
            sp = t[:, 3] != 0
            if sp.any():
                pct((t[sp, 3] - t0) / 100.0, 'split phase starts at')
            pct((t[:, 5] & 0xFFFFFF) / 100.0, 'waiting for partners')
            nseg = (t[:, 5] >> 48) & 0xFFFF
            epi = ((t[:, 5] >> 24) & 0xFFFFFF) / 100.0
            pct(epi, 'epilogues (K-steps done -> stored)')
            pct(epi / np.maximum(nseg, 1), '   per segment')
            pct(nseg, 'segments per workgroup', unit='')
            vm = ((t[:, 6] >> 8) & 0xFFFFFFF) / 100.0
            if vm.any():        # experiment builds only (ICN_EXP & 512): wave 0's time in the per-step vmcnt wait / barrier
                pct(vm, 'in the counted vmcnt waits')
                pct(((t[:, 6] >> 36) & 0xFFFFFFF) / 100.0, 'in the K-step barriers')
            pct(end, 'exit at')
            pct(end.max() - end, 'idle before the launch ends')
            busy = (t[:, 4] - t[:, 2]).sum() / 100.0
            print('  sum over workgroups of (exit - first MFMA) = %.0f us = %.1f %% of %d x launch span' % (
                busy, 100.0 * busy / (G * end.max()), G))
            for xcc in sorted(set(t[:, 6] & 15)):
                m = (t[:, 6] & 15) == xcc
                print('    XCC %d: %3d workgroups, first MFMA median %.2f, exit min %.2f median %.2f max %.2f' % (
                    xcc, m.sum(), np.median(us(2)[m]), end[m].min(), np.median(end[m]), end[m].max()))
            # HW_ID (gfx9): wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13
            cu = ((t[:, 7] >> 8) & 15) | (((t[:, 7] >> 12) & 1) << 4) | (((t[:, 7] >> 13) & 7) << 5) | ((t[:, 6] & 15) << 8)
            pairs = {}
            for c, e in zip(cu, end):
                pairs.setdefault(int(c), []).append(float(e))
            both = [v for v in pairs.values() if len(v) == 2]
            if both:
                d = np.array([abs(v[0] - v[1]) for v in both])
                print('    %d CUs hold two workgroups: |exit difference| within a CU median %.2f max %.2f us; CU means spread %.2f .. %.2f'
                      % (len(both), np.median(d), d.max(), min(np.mean(v) for v in both), max(np.mean(v) for v in both)))
            print('    distinct (XCC, SE, SH, CU) ids: %d' % len(pairs))
            if rep == 0 and a.dump:
                # is the spread tied to the work (same local block index slow in every XCD) or to the place (CU)?
                Gl = G // 8
                byx = np.full((8, Gl), np.nan)
                full = buf.cpu().numpy().reshape(-1, 8)
                for b in range(G):
                    if full[b, 0]:
                        byx[b % 8, b // 8] = (full[b, 4] - t0) / 100.0
                mean_bl = np.nanmean(byx, axis=0)
                resid = byx - mean_bl[None, :]
                print('    exit time by local block index (mean over the 8 XCDs), 8 per line:')
                for i in range(0, Gl, 8):
                    print('      ' + ' '.join('%6.1f' % v for v in mean_bl[i:i + 8]))
                print('    std of per-index means %.2f us; std of residuals (XCD/CU part) %.2f us; total std %.2f us' % (
                    np.nanstd(mean_bl), np.nanstd(resid), np.nanstd(byx)))
                np.save(a.dump, full)


if __name__ == '__main__':
    main()
