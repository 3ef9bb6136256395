#!/bin/bash
# GPU helper: the inverse at batches that are whole rounds of one-wave problems plus a remainder, with and without the remainder launch
# (FINC_NO_REMAINDER_LAUNCH=1: one launch of the one-wave kernel, as before).  scripts/time_shape.py prints inverse and forward.
run() {
  a=$(FINC_NO_REMAINDER_LAUNCH=1 timeout -k 10 120 python scripts/time_shape.py $1 $2 $3 $4 $5 $6 2>&1 | grep -o "inverse [0-9.]* us")
  b=$(timeout -k 10 120 python scripts/time_shape.py $1 $2 $3 $4 $5 $6 2>&1 | grep -o "inverse [0-9.]* us")
  echo "B$1 C$2 ${3}x$4 k$5: one launch $a   rounds + remainder $b"
}
run 256 96 64 64 3
run 264 96 64 64 3
run 288 96 64 64 3
run 320 96 64 64 3
run 352 96 64 64 3
run 384 96 64 64 3
run 400 96 64 64 3
run 576 96 64 64 3
run 320 48 32 32 3
run 384 48 32 32 3
run 320 80 64 64 3
run 300 64 64 64 2
run 224 112 64 64 3 0.02
run 160 128 64 64 3 0.03
run 192 128 64 64 3 0.03
run 320 128 64 64 3 0.03
run 160 128 64 64 2
run 64 112 64 64 3 0.02
run 32 112 64 64 3 0.02
