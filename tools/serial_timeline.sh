#!/bin/bash
# rocprofv3 kernel trace of the serial bench pass -> per-launch table (tools/serial_timeline.py)
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--no-cpu-baseline --no-knn --no-netvlad --vox-leg 0 --windows 1 --train-steps 0 --default-prec-leg 0 --graph 0 --streams 1 --qsplit 1 --steps 20 --warmup 2"
mkdir -p $R/gpurun_out/${tag}_stl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_stl -o t -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cd $R && python3 tools/serial_timeline.py gpurun_out/${tag}_stl 8 > gpurun_out/${tag}_serial_timeline.txt 2>&1
find $R/gpurun_out/${tag}_stl -name "*.csv" -size +20M -delete
cat gpurun_out/${tag}_serial_timeline.txt
