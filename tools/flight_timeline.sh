#!/bin/bash
# kernel trace of the default bench step (two steps in flight) -> profiles/<tag>_flight_timeline.txt (tools/flight_timeline.py)
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/${tag}_flighttl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_flighttl -o t -- python3 $R/bench.py --no-cpu-baseline --no-knn --train-steps 0 --steps 24 --warmup 4 > $R/gpurun_out/${tag}_flighttl/line.json 2>/dev/null
cd $R/tools && python3 flight_timeline.py $R/gpurun_out/${tag}_flighttl $R/gpurun_out/${tag}_flight_timeline.txt > /dev/null
find $R/gpurun_out -name "*kernel_trace.csv" -size +40M -delete
