"""Timeline of one training step from a rocprofv3 --kernel-trace run (developer tool): which launches of the two streams
overlap, how long each takes beside the other, where the device idles.

  python tools/timeline.py <dir with *_kernel_trace.csv> [step index from the end, default 2]
The last `bench.py` steps are cut at the Adam launch (`k_adam`); prints every kernel of the chosen step with its queue, start
offset, duration, and the fraction of its duration during which the OTHER queue had a kernel running."""
import csv
import glob
import os
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = max(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
adam = [i for i, r in enumerate(rows) if 'k_adam' in r[3]]
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
step = rows[lo:hi]
t0 = step[0][0]
queues = sorted({r[2] for r in step}, key=lambda q: -sum(1 for r in step if r[2] == q))
main = queues[0]


def short(n):
    n = n.replace('void ', '').replace('icn::', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0][:44]


def overlap(a0, a1, others):
    tot = 0
    for b0, b1 in others:
        tot += max(0, min(a1, b1) - max(a0, b0))
    return tot


by_q = {q: [(r[0], r[1]) for r in step if r[2] == q] for q in queues}
print('step of %d kernels, %.3f ms; queues %s' % (len(step), (step[-1][1] - t0) / 1e6, {q: len(v) for q, v in by_q.items()}))
busy_main = sum(e - s for s, e in by_q[main])
print('main queue busy %.3f ms; side queue(s) busy %.3f ms' % (busy_main / 1e6, sum(e - s for q in queues[1:] for s, e in by_q[q]) / 1e6))
for s, e, q, n in step:
    others = [iv for qq in queues if qq != q for iv in by_q[qq]]
    ov = overlap(s, e, others) / max(1, e - s)
    print('%s %9.1f us  +%8.1f us  %5.0f%% beside  %s' % ('M' if q == main else 'S', (s - t0) / 1e3, (e - s) / 1e3, 100 * ov, short(n)))
