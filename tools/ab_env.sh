#!/bin/bash
# On the GPU box: alternate `bench.py` runs under two environment settings inside one call (box-to-box variation is +-5 %).
#   tools/ab_env.sh <out> <reps> "<ENV_A>" "<ENV_B>" [bench args...]      e.g.  tools/ab_env.sh gpurun_out/x.txt 3 "ICN_CONV_WAVES=4" "ICN_CONV_WAVES=8"
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$1; REPS=$2; A=$3; B=$4; shift 4
: > $OUT
for rep in $(seq $REPS); do
  for e in "$A" "$B"; do
    env $e timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
ks=' '.join('%s=%.1f' % (k['kernel'].replace('k_','').replace(' ',''), k['avg_launch_us']) for k in r['all_mfma_kernels'])
print('%-28s %8.1f meshes/s %.3f ms  dom %s %.1f TF/s  | %s' % ('$e', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], ks))" >> $OUT
  done
done
cat $OUT
