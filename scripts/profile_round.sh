#!/bin/bash
# Usage (on the GPU box): scripts/profile_round.sh <tag>
# The round's judged evidence in one go: bench lines (c3 default, c2, c5, c3_share8), rocprofv3 kernel stats of the same
# command, and the PMC passes for c3 (wavefront kernel) and c3_share8 (role-split kernel).  Everything lands in
# gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python3 bench.py > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err
echo "bench c3 done"
for w in c2 c5 c3_share8; do
  python3 bench.py --workload $w --steps 50 --no-cpu > gpurun_out/${TAG}_bench_$w.json 2>> gpurun_out/${TAG}_bench.err
  echo "bench $w done"
done
python3 bench.py --workload c4 --steps 20 --warmup 3 --no-cpu > gpurun_out/${TAG}_bench_c4.json 2>> gpurun_out/${TAG}_bench.err
echo "bench c4 done"
python3 scripts/f3_measure.py > gpurun_out/${TAG}_f3.json 2>> gpurun_out/${TAG}_bench.err
echo "f3 done"
python3 scripts/debug_big.py time > gpurun_out/${TAG}_big_banks.txt 2>> gpurun_out/${TAG}_bench.err
echo "big banks done"
python3 scripts/time_small.py > gpurun_out/${TAG}_small_batch.txt 2>> gpurun_out/${TAG}_bench.err
echo "small batch done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof_c3 -- python3 $ROOT/bench.py --no-cpu --no-share > $ROOT/gpurun_out/${TAG}_bench_c3_under_rocprof.json 2> $ROOT/gpurun_out/${TAG}_prof_c3.err
echo "rocprof c3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof_share8 -- python3 $ROOT/bench.py --workload c3_share8 --no-cpu --no-share > $ROOT/gpurun_out/${TAG}_bench_share8_under_rocprof.json 2> $ROOT/gpurun_out/${TAG}_prof_share8.err
echo "rocprof share8 done"
cd $ROOT
scripts/pmc_run.sh ${TAG}_c3 > gpurun_out/${TAG}_pmc_c3.log 2>&1
echo "pmc c3 done"
scripts/pmc_run.sh ${TAG}_share8 --workload c3_share8 > gpurun_out/${TAG}_pmc_share8.log 2>&1
echo "pmc share8 done"
# keep the summaries, drop the raw traces (gpurun merges at most 64 MiB back)
mkdir -p gpurun_out/${TAG}
cp $(find gpurun_out/${TAG}_prof_c3 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}/c3_kernel_stats.csv
cp $(find gpurun_out/${TAG}_prof_share8 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}/c3_share8_kernel_stats.csv
cp gpurun_out/pmc_${TAG}_c3/summary.json gpurun_out/${TAG}/c3_pmc_summary.json
cp gpurun_out/pmc_${TAG}_share8/summary.json gpurun_out/${TAG}/c3_share8_pmc_summary.json
mv gpurun_out/${TAG}_bench_*.json gpurun_out/${TAG}_f3.json gpurun_out/${TAG}_big_banks.txt gpurun_out/${TAG}_small_batch.txt gpurun_out/${TAG}/
rm -rf gpurun_out/${TAG}_prof_c3 gpurun_out/${TAG}_prof_share8 gpurun_out/pmc_${TAG}_c3 gpurun_out/pmc_${TAG}_share8
ls -la gpurun_out/${TAG}
