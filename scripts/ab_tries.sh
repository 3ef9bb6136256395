for pass in 1 2 3 4 5 6; do
  for s in "32 96 64 64 3" "32 96 128 64 3"; do
    python scripts/time_one.py $s 2>&1 | tail -1
    FINCFLOW_LIB=ablate_build/libfinc_t40.so python scripts/time_one.py $s 2>&1 | tail -1 | sed 's/^/   t40: /'
  done
done
