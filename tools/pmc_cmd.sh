#!/bin/bash
# usage: tools/pmc_cmd.sh <outdir> <program> [args...]   -- separate --pmc passes over one command (GPU box)
set -e
OUT="$1"; shift; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- "${@:3}" > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "$@"
run sq2 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM" "$@"
run grbm "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "$@"
