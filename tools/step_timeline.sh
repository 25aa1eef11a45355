#!/bin/bash
# kernel trace of the replayed bench step -> profiles/<tag>_step_timeline.txt (tools/step_timeline.py)
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/${tag}_steptl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_steptl -o t -- python3 $R/bench.py --inflight 1 --no-cpu-baseline --no-knn --train-steps 0 --steps 12 --warmup 3 > $R/gpurun_out/${tag}_steptl/line.json 2>/dev/null
cd $R && python3 tools/step_timeline.py gpurun_out/${tag}_steptl gpurun_out/${tag}_step_timeline.txt > /dev/null
