#!/bin/bash
# GPU-side: correctness on three shapes + timing on two + stamps, for the chain-kernel variant libraries named on the command line
# (ablate_build/libfinc_<name>.so; a name ending in "st" is a stamp build and only prints stamps)
for v in "$@"; do
  export FINCFLOW_LIB=ablate_build/libfinc_$v.so
  if [[ $v == *st ]]; then
    timeout -k 10 100 python scripts/chain_stamps.py 64 48 32 32 3 2>&1 | tail -1
  else
    for s in "64 48 32 32 3" "3 40 20 24 3" "4 48 16 16 3" "2 48 8 8 3"; do timeout -k 5 100 python scripts/check_chain.py $s 2>&1 | grep "^B\|bad\|wrong\|untouched"; done
    timeout -k 5 60 python scripts/time_one.py 64 48 32 32 3 2>&1 | tail -1
    timeout -k 5 60 python scripts/time_one.py 16 48 64 64 3 2>&1 | tail -1
  fi
done
