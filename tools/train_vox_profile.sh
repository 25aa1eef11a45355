#!/bin/bash
# kernel profile of the training step with the voxel branch trained from coords
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r06}
mkdir -p $R/gpurun_out/${tag}_trainvoxdir
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trainvoxdir -o t -- python3 $R/tools/train_bench.py --batch 16 --streams 2 --steps 4 --warmup 2 --vox-points 8000 > $R/gpurun_out/${tag}_trainvoxdir/stdout.txt 2>&1
cd $R
python3 tools/step_profile.py $(find gpurun_out/${tag}_trainvoxdir -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/${tag}_trainvox_step_kernels.txt 2>&1
head -45 gpurun_out/${tag}_trainvox_step_kernels.txt
tail -2 gpurun_out/${tag}_trainvoxdir/stdout.txt
find $R/gpurun_out/${tag}_trainvoxdir -name "*kernel_trace.csv" -size +40M -delete
