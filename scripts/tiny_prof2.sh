#!/bin/bash
# GPU-side: the tiny maps at the CIFAR stack's sampling batch (128 images = 512 problems): wavefront kernel's table (default) against the
# short-step kernel with two workgroups per compute unit (FINC_SPLIT_MAX=512), kernel-only durations
for s in "128 12 16 16 3" "128 24 8 8 3" "128 48 4 4 3" "128 48 32 32 3" "96 48 32 32 3"; do
  scripts/prof_one.sh $s tiny 2>&1 | grep -i "finc_wave\|finc_chain\|finc_split" | head -2
  FINC_SPLIT_MAX=512 scripts/prof_one.sh $s tiny 2>&1 | grep -i "finc_wave\|finc_chain\|finc_split" | head -2
done
