#!/bin/bash
# usage: tools/pmc_one_layer.sh <layer-substring> <outdir>   (GPU box; separate --pmc passes, kernel-trace only)
set -e
L="$1"; OUT="$(realpath -m "$2")"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/tools/bench_layers.py --only "$L" --iters 3 $EXTRA > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
run sq2 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES"
run grbm "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
