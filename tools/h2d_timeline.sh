#!/bin/bash
# rocprofv3 kernel + memory-copy traces of the step with resident uint8 inputs and with the pinned-ring upload (bench.py --h2d):
# tools/h2d_timeline.py turns them into profiles/<tag>_h2d_timeline.json (kernel slowdown under the copy vs submission gaps)
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--inflight 1 --no-cpu-baseline --no-knn --train-steps 0 --steps 30 --warmup 3"
mkdir -p $R/gpurun_out/${tag}_tl_res $R/gpurun_out/${tag}_tl_h2d
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/${tag}_tl_res -o t -- python3 $R/bench.py $ARGS --u8 > $R/gpurun_out/${tag}_tl_res/line.json 2>/dev/null
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/${tag}_tl_h2d -o t -- python3 $R/bench.py $ARGS --h2d > $R/gpurun_out/${tag}_tl_h2d/line.json 2>/dev/null
cd $R && python3 tools/h2d_timeline.py $tag
