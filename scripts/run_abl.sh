for v in "$@"; do export FINCFLOW_LIB=ablate_build/libfinc_$v.so; timeout -k 5 60 python scripts/time_one.py 16 48 64 64 3 2>&1 | tail -1; done
