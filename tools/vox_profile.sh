#!/bin/bash
# kernel trace of the step with the sparse-voxel branch running from coords (single stream, eager) + the bench figure
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r06_voxtrace}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag -o t -- python3 $R/bench.py --vox --no-cpu-baseline --no-knn --train-steps 0 --default-prec-leg 0 --graph 0 --streams 1 --qsplit 1 --steps 6 --warmup 2 > $R/gpurun_out/$tag/stdout.txt 2>&1
cd $R
python3 tools/vox_timeline.py $(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1) | tee gpurun_out/$tag/timeline.txt
find $R/gpurun_out/$tag -name "*kernel_trace.csv" -size +40M -delete
python3 bench.py --vox --no-cpu-baseline --no-knn --train-steps 0 --default-prec-leg 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vox ms_per_step', d['ms_per_step'], 'one in flight', d['config'].get('ms_per_step_one_in_flight'))"
