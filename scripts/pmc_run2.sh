#!/bin/bash
# Usage (on the GPU box): scripts/pmc_run2.sh <tag> [bench args]
# Second counter set: where a wave's non-MFMA time goes (issue by type, memory-queue back-pressure, TA/TCP stalls,
# L2 round-trip latency, TLB).  One rocprofv3 pass per group, kernel-trace only (no other trace domains with --pmc).
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
  "SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1)); name=g$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu "$@" > $OUT/$name.log 2>&1 || echo "pass $name failed"
done
python3 $ROOT/scripts/pmc_summarize.py $OUT
