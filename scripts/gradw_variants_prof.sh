#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of scripts/time_gradw.py c3 for each variant library named on the command line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" == "product" ]; then unset FINCFLOW_LIB; else export FINCFLOW_LIB=$ROOT/ablate_build/libfinc_gw_$v.so; fi
  rm -rf /tmp/gwp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gwp -- python3 $ROOT/scripts/time_gradw.py c3 > /tmp/gwp.log 2>&1
  f=$(find /tmp/gwp -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "gradw" $f | awk -F, '{printf "%s calls %s avg %.1f us\n", substr($1,2,60), $(NF-6), $(NF-4)/1000}'
done
