#!/bin/bash
# kernel trace of the two-stream training step -> gpurun_out/<tag>_train_timeline.txt (tools/train_timeline.py)
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
d=$R/gpurun_out/${tag}_traintl
rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 $R/tools/train_bench.py --batch 16 --streams 2 --steps 4 --warmup 2 > $d/line.json 2>/dev/null
cd $R && python3 tools/train_timeline.py $d > gpurun_out/${tag}_train_timeline.txt; cat gpurun_out/${tag}_train_timeline.txt; cat $d/line.json | cut -c1-300
find $d -name "*kernel_trace.csv" -size +40M -delete
