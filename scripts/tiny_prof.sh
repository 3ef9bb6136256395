#!/bin/bash
# GPU-side: kernel-only durations (rocprofv3 kernel trace) of the inverse on the tiny maps of the CIFAR stack
for s in "128 12 16 16 3" "128 24 8 8 3" "128 48 4 4 3" "32 12 16 16 3" "16 24 8 8 3" "8 48 4 4 3" "128 48 32 32 3"; do
  scripts/prof_one.sh $s tiny 2>&1 | grep -i "finc_\|default" | head -3
done
