#!/bin/bash
# GPU-side A/B on one box: the role-split kernel (band split included) with its loops left at the exact step (the product) against
# the build that rounds the step count to the loop's unroll (ablate_build/libfinc_splitold.so), alternating, two passes
for pass in 1 2; do
  for s in "4 96 64 64 3" "16 96 64 64 3" "32 96 64 64 3" "64 96 64 64 3" "32 96 48 64 3" "64 96 32 32 3"; do
    python scripts/time_one.py $s 2>&1 | tail -1
    FINCFLOW_LIB=ablate_build/libfinc_splitold.so python scripts/time_one.py $s 2>&1 | tail -1 | sed 's/^/   old: /'
  done
done
