#!/bin/bash
# Usage (on the GPU box): scripts/pmc_run.sh <tag> [bench args]
# One rocprofv3 pass per counter group (PMC slots: FETCH_SIZE 3 of 4 TCC, WRITE_SIZE 2), kernel-trace only.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-share "$@" > $OUT/$name.log 2>&1 || echo "pass $name failed"
done
python3 $ROOT/scripts/pmc_summarize.py $OUT
