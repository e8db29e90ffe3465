#!/bin/bash
# usage: tools/trace_layer.sh <layer-substring> <outdir>   (GPU box) -- per-kernel durations of one layer's fwd/dgrad/wgrad calls
set -e
L="$1"; OUT="$(realpath -m "$2")"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/bench_layers.py --only "$L" --iters 3 > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'icn::' not in n: continue
    n = n.replace('void icn::', '')[:60]
    d.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in d.items():
    v = sorted(v)
    print('%-60s n=%3d  min %8.1f  med %8.1f us' % (n, len(v), v[0], v[len(v) // 2]))
PY
