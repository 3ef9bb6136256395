#!/bin/bash
# GPU-side: the short-step kernel against the role-split kernel (band split included) on shapes both take
for s in "5 64 128 64 3" "8 48 17 72 3" "4 48 64 64 3" "16 48 64 64 3" "8 64 64 64 3" "6 64 50 64 2" "2 48 128 128 3" "32 48 64 64 3" "64 48 32 32 3" "64 64 32 32 3" "64 16 32 32 3" "64 32 32 32 3" "32 12 16 16 3" "16 24 8 8 3" "8 48 4 4 3" "32 64 32 32 2"; do
  python scripts/time_one.py $s 2>&1 | tail -1
  FINC_NO_CHAIN=1 python scripts/time_one.py $s 2>&1 | tail -1
done
