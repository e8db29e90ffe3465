"""Summarise rocprofv3 --pmc csv output: per kernel name, mean of each counter per dispatch and mean duration."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else 'icn::'
for d in sorted(glob.glob(root + '/*/')):
    files = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    traces = glob.glob(d + '**/*kernel_trace.csv', recursive=True)
    dur = collections.defaultdict(list)
    for f in traces:
        for r in csv.DictReader(open(f)):
            dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', d)
    for k, cs in agg.items():
        if filt not in k:
            continue
        n = len(next(iter(cs.values())))
        print('  %s  dispatches=%d  avg_us=%.1f' % (k[:60], n, sum(dur[k]) / max(len(dur[k]), 1)))
        for c, v in sorted(cs.items()):
            print('      %-28s %.4g' % (c, sum(v) / len(v)))
