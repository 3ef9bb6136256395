#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the round-4 Winograd kernels outside bench.py's workloads -- the tile-pair grad-weight
# (scripts/time_gradw.py big) and the M-split F(4,3) forward (scripts/time_forward.py); output under gpurun_out/<tag>/.
TAG=${1:-newk}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/nk1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nk1 -- python3 $ROOT/scripts/time_gradw.py big > $ROOT/gpurun_out/$TAG/time_gradw_big.txt 2>&1
cp $(find /tmp/nk1 -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/$TAG/gradw_big_kernel_stats.csv
rm -rf /tmp/nk2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nk2 -- python3 $ROOT/scripts/time_forward.py > $ROOT/gpurun_out/$TAG/time_forward.txt 2>&1
cp $(find /tmp/nk2 -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/$TAG/forward_wide_kernel_stats.csv
head -4 $ROOT/gpurun_out/$TAG/gradw_big_kernel_stats.csv | cut -c1-200; head -5 $ROOT/gpurun_out/$TAG/forward_wide_kernel_stats.csv | cut -c1-200
