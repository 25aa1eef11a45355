#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/exp8
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-knn --train-steps 0 --graph 0 --streams 1 --qsplit 1 --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/exp8/trace -o t -- python3 $R/bench.py $ARGS > $R/gpurun_out/exp8/trace.log 2>&1
cd $R
python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/exp8/bench.json 2> gpurun_out/exp8/bench.err
find gpurun_out/exp8 -name "*kernel_stats.csv" | head -2
