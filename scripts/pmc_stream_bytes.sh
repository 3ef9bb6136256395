#!/bin/bash
# GPU-side: FETCH_SIZE / WRITE_SIZE (KiB at the L2's fabric port) of the streaming-bank kernels at one shape, as scripts/pmc_run.sh
# collects them for the other kernels:   scripts/pmc_stream_bytes.sh <tag> B G Cq H W KH KW
# ONE counter per pass: FETCH_SIZE takes 3 of the 4 TCC slots and WRITE_SIZE 2 -- both in one --pmc group abort the profiler with
# signal 6, and it then does not exit (a 7-minute silent run on record, profiles/r05/stream/ablations.txt item 7).  Each pass sits
# under its own timeout for that reason; kernel-trace only, python3 directly behind `--`.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $ROOT/scripts/prof_stream.py "$@" 3 > $OUT/$c.log 2>&1 || echo "pass $c failed or timed out" | tee -a $OUT/failed.txt
  echo "pass $c done" >> $OUT/progress.txt
done
python3 $ROOT/scripts/pmc_summarize.py $OUT > $OUT/summary.txt
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
