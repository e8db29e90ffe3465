#!/bin/bash
# Round evidence for profiles/: run on the GPU box (via gpurun).  usage: tools/profile_round.sh rNN
#   1. rocprofv3 --kernel-trace --stats of the default bench command (no PMC)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, GRBM/TCC) of the same command with fewer steps
# Raw output goes to gpurun_out/prof_<round>/; tools/profile_summary.py condenses it into profiles/.
set -e
ROUND="${1:-r01}"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$ROUND; mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$R'); from geniconet_amd import _lib; print(_lib.source_sha256())" > $OUT/sources_sha256.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $OUT/bench_stats.json 2> $OUT/bench_stats.err
# the same command with every kernel on ONE stream (the weight gradients do not overlap the other launches): the per-kernel
# durations bench.py's `roofline` block is computed from; the counter passes run that way too (a counter belongs to one kernel)
export ICN_WGRAD_STREAM=off
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_one_stream -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $OUT/bench_stats_one_stream.json 2> $OUT/bench_stats_one_stream.err
for c in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-kernel-events > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
done
cd $R && python3 tools/profile_summary.py $ROUND
