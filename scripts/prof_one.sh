#!/bin/bash
# GPU-side: rocprofv3 kernel stats of scripts/time_one.py for one shape:  scripts/prof_one.sh B C H W K [tag]
ROOT=$PWD
TAG=${6:-prof_one}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG -- python3 $ROOT/scripts/time_one.py $1 $2 $3 $4 $5 > $ROOT/gpurun_out/$TAG.txt 2>&1
cd $ROOT
f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python3 -c "import csv,sys; [print(r[\"Name\"][:90], r[\"Calls\"], r[\"AverageNs\"]) for r in list(csv.DictReader(open(sys.argv[1])))[:6]]" $f
tail -1 gpurun_out/$TAG.txt
rm -rf gpurun_out/$TAG
