#!/bin/bash
# Usage (on the GPU box): scripts/pmc_quick.sh <tag> "<group 1>" "<group 2>" ... -- [bench args]
# A few rocprofv3 --pmc passes (one per counter group, kernel-trace only) over a short bench run; summary by scripts/pmc_summarize.py.
set -e
TAG=$1; shift
GROUPS_=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
[ "$1" == "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PMC_PY=<script> profiles that script (with the arguments after --) instead of a short bench run
if [ -n "$PMC_PY" ]; then CMD="$ROOT/$PMC_PY"; else CMD="$ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-share --no-clock"; fi
for grp in "${GROUPS_[@]}"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$name -- python3 $CMD "$@" > $OUT/$name.log 2>&1 || echo "pass $name failed"
done
python3 $ROOT/scripts/pmc_summarize.py $OUT > $OUT/summary.txt
