#!/bin/bash
# kernel profile of the training step in an opt-in fast mode: bash tools/train_fast_profile.sh <tag> "<train_bench flags>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r06}
flags=${2:---train-precision 16 --dgrad-products 1}
mkdir -p $R/gpurun_out/${tag}_trainfast
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trainfast -o t -- python3 $R/tools/train_bench.py --batch 16 --streams 2 --steps 4 --warmup 2 $flags > $R/gpurun_out/${tag}_trainfast/stdout.txt 2>&1
cd $R
python3 tools/step_profile.py $(find gpurun_out/${tag}_trainfast -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/${tag}_trainfast_step_kernels.txt 2>&1
head -40 gpurun_out/${tag}_trainfast_step_kernels.txt
tail -1 gpurun_out/${tag}_trainfast/stdout.txt | cut -c1-160
find $R/gpurun_out/${tag}_trainfast -name "*kernel_trace.csv" -size +40M -delete
