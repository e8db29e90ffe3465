#!/bin/bash
# usage: tools/pmc_deep.sh <outdir> <program> [args...]   -- deeper separate --pmc passes (issue mix, FIFO stalls, L1 / TLB) over one command
set -e
OUT="$(realpath -m "$1")"; shift; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- "${@:3}" > $OUT/$1.log 2>&1 || echo "pass $1 failed"; }
run a1 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "$@"
run a2 "SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES" "$@"
run b1 "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "$@"
run b2 "SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES" "$@"
run c1 "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_UTCL1_TRANSLATION_MISS" "$@"
run c2 "TCP_UTCL1_TRANSLATION_HIT TA_TA_BUSY TCP_TCP_TA_DATA_STALL_CYCLES TD_TD_BUSY" "$@"
run c3 "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES TCP_CACHE_MISS TCP_TOTAL_READ" "$@"
