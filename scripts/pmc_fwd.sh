#!/bin/bash
# Usage (on the GPU box): scripts/pmc_fwd.sh <tag>: SQ-side counters of the c3 forward kernel (one rocprofv3 pass per group)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-share "$@" > $OUT/$name.log 2>&1 || echo "pass $name failed"
done
python3 $ROOT/scripts/pmc_summarize.py $OUT > $OUT/summary.txt
python3 -c "
import json; d=json.load(open('$OUT/summary.json'))
for k in ('forward_wino4','forward_wino','gradw_staged'):
    if k in d: print(k, json.dumps(d[k], indent=1, sort_keys=True))
"
rm -rf $OUT/*/
