#!/bin/bash
# GPU-side: time one shape on the product library and on the variant libraries named (ablate_build/libfinc_<name>.so), alternating.
#   scripts/ab_libs.sh "B C H W K" passes name...
shape=$1; passes=$2; shift 2
for pass in $(seq $passes); do
  python scripts/time_one.py $shape 2>&1 | tail -1
  for v in "$@"; do FINCFLOW_LIB=ablate_build/libfinc_$v.so python scripts/time_one.py $shape 2>&1 | tail -1 | sed 's/^/   /'; done
done
