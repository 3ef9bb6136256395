#!/bin/bash
# GPU helper: the Winograd weight-gradient kernel's own time (rocprofv3 kernel trace) at strips of 16 / 32 columns (FINC_GRADW_WINO_STRIP) and at the
# library's choice, for batches where image x strip units do not divide evenly among the waves.  Usage: scripts/gradw_strip_prof.sh B C H W
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for strip in 16 32 lib; do
  if [ $strip = lib ]; then unset FINC_GRADW_WINO_STRIP; else export FINC_GRADW_WINO_STRIP=$strip; fi
  rm -rf /tmp/gsp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gsp -- python3 $ROOT/scripts/time_backward.py $1 $2 $3 $4 3 > /tmp/gsp.out 2>&1
  f=$(find /tmp/gsp -name "*kernel_stats.csv" | head -1)
  echo "B=$1 C=$2 $3x$4 strip=$strip: $(grep -o 'fwd+bwd [0-9.]* ms' /tmp/gsp.out)  $(grep gradw_wino_kernel $f | awk -F, '{printf "gradw kernel %s calls avg %.1f us", $(NF-6), $(NF-4)/1000}')"
done
