#!/bin/bash
# GPU-side: rocprofv3 kernel stats + PMC passes of the streaming-bank kernels at one shape.
#   scripts/prof_stream.sh <tag> B G Cq H W KH KW     -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/pmc_<tag>/summary.json
# One rocprofv3 pass per counter group, kernel-trace only (no other trace domains with --pmc); python3 directly behind `--`.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/scripts/prof_stream.py "$@" 8 > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${TAG}_kernel_stats.csv
i=0
for grp in \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1)); name=g$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$name -- python3 $ROOT/scripts/prof_stream.py "$@" 3 > $OUT/$name.log 2>&1 || echo "pass $name failed"
done
python3 $ROOT/scripts/pmc_summarize.py $OUT > $OUT/summary.txt
rm -rf $OUT/trace $OUT/g?
tail -3 $OUT/trace.log
