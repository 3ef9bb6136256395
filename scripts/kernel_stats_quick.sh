set -e
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/gwprof -- python3 $ROOT/bench.py --no-cpu --no-share --steps 40 > $ROOT/gpurun_out/gw_bench.json 2> $ROOT/gpurun_out/gw.err
cd $ROOT
f=$(find gpurun_out/gwprof -name "*kernel_stats.csv" | head -1)
head -5 $f | cut -c1-230
rm -rf gpurun_out/gwprof
python3 -c "
import json; d=json.load(open('gpurun_out/gw_bench.json')); print('inv', d['roofline']['launch_ms'], 'fwd', d['forward']['launch_ms'], 'train', d['training_step']['ms_per_step'])"
