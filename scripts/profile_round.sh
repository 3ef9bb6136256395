#!/bin/bash
# Usage (on the GPU box): scripts/profile_round.sh <tag> [benches|pmc]
# The round's judged evidence: `benches` -- bench lines (c3 default, c2, c5, c3_share8, c4, ref_timing), the other timing scripts and
# rocprofv3 kernel stats of the same bench command for c3, c3_share8 and c5; `pmc` -- the PMC passes for c3 (wavefront kernel),
# c3_share8 (role-split kernel, band split) and c5 (K-split inverse, F(2,5) forward).  Everything lands in gpurun_out/<tag>/; copy
# what is to be judged into profiles/.  (rocprofv3 gets `python3 bench.py` directly after `--`; the profiled runs skip the clock
# probe, a kernel that runs beside the legs for their whole length.)
set -e
TAG=$1
WHAT=${2:-benches}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/${TAG}
if [ "$WHAT" = "benches" ]; then
  python3 bench.py > gpurun_out/${TAG}/bench_c3.json 2> gpurun_out/${TAG}/bench_c3.err
  echo "bench c3 done"
  for w in c2 c5 c3_share8; do
    python3 bench.py --workload $w --steps 50 --no-cpu > gpurun_out/${TAG}/bench_$w.json 2>> gpurun_out/${TAG}/bench.err
    echo "bench $w done"
  done
  python3 bench.py --workload c4 --steps 20 --warmup 3 --no-cpu > gpurun_out/${TAG}/bench_c4.json 2>> gpurun_out/${TAG}/bench.err
  echo "bench c4 done"
  python3 bench.py --workload ref_timing > gpurun_out/${TAG}/bench_ref_timing_block16.json 2>> gpurun_out/${TAG}/bench.err
  python3 bench.py --workload ref_timing --ref-block-size 48 > gpurun_out/${TAG}/bench_ref_timing_block48.json 2>> gpurun_out/${TAG}/bench.err
  echo "ref_timing done"
  python3 scripts/f3_measure.py > gpurun_out/${TAG}/f3.json 2>> gpurun_out/${TAG}/bench.err
  python3 scripts/debug_big.py time > gpurun_out/${TAG}/big_banks.txt 2>> gpurun_out/${TAG}/bench.err
  python3 scripts/time_small.py > gpurun_out/${TAG}/small_batch.txt 2>> gpurun_out/${TAG}/bench.err
  python3 scripts/time_f64.py > gpurun_out/${TAG}/fp64_mfma.txt 2>> gpurun_out/${TAG}/bench.err
  scripts/chain_vs_split.sh > gpurun_out/${TAG}/chain_vs_split.txt 2>> gpurun_out/${TAG}/bench.err
  scripts/tiny_prof2.sh > gpurun_out/${TAG}/tiny_maps_kernel_time.txt 2>> gpurun_out/${TAG}/bench.err
  echo "other timings done"
  cd /tmp && export TMPDIR=/tmp
  for w in c3 c2 c3_share8 c5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof_$w -- python3 $ROOT/bench.py --workload $w --no-cpu --no-share --no-clock > $ROOT/gpurun_out/${TAG}/bench_${w}_under_rocprof.json 2> $ROOT/gpurun_out/${TAG}/prof_$w.err
    cp $(find $ROOT/gpurun_out/${TAG}_prof_$w -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${TAG}/${w}_kernel_stats.csv
    rm -rf $ROOT/gpurun_out/${TAG}_prof_$w
    echo "rocprof $w done"
  done
else
  for w in c3 c2 c3_share8 c5; do
    scripts/pmc_run.sh ${TAG}_$w --workload $w --no-clock > gpurun_out/${TAG}/pmc_$w.log 2>&1
    cp gpurun_out/pmc_${TAG}_$w/summary.json gpurun_out/${TAG}/${w}_pmc_summary.json
    rm -rf gpurun_out/pmc_${TAG}_$w
    echo "pmc $w done"
  done
fi
ls -la $ROOT/gpurun_out/${TAG}
