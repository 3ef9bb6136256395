run() { # kind B C H W K old
  local var=FINC_WINO_CHUNKS; [ "$1" = strip ] && var=FINC_CONV_CHUNKS
  a=$(env $var=$7 timeout -k 10 120 python scripts/time_shape.py $2 $3 $4 $5 $6 2>&1 | grep -o "forward [0-9.]* us")
  b=$(timeout -k 10 120 python scripts/time_shape.py $2 $3 $4 $5 $6 2>&1 | grep -o "forward [0-9.]* us")
  echo "$1 B$2 C$3 ${4}x$5 k$6: old rule ($7 chunks) $a   new rule ($8 chunks) $b"
}
run f23 12 96 64 64 3 11 10
run f23 20 96 64 64 3 7 6
run f23 24 96 64 64 3 6 5
run f23 28 96 64 64 3 5 8
run f23 96 48 32 32 3 3 5
run f23 128 48 32 32 3 2 4
run f23 160 48 32 32 3 2 3
run f23 192 48 32 32 3 2 4
run f23 320 48 32 32 3 2 4
run f23 48 96 32 32 3 6 5
run f23 96 96 32 32 3 3 5
run strip 16 96 64 64 2 4 8
run strip 24 96 64 64 2 3 5
run strip 32 96 64 64 2 2 4
run strip 40 96 64 64 2 2 3
run strip 160 96 64 64 2 1 2
run strip 64 48 32 32 2 2 4
run strip 96 48 32 32 2 2 4
run strip 160 48 32 32 2 2 4
