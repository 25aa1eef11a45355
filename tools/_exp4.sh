#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp4
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d" > gpurun_out/exp4/tests.txt 2>&1
tail -5 gpurun_out/exp4/tests.txt
for v in 0 1; do
  AGP_KXR2=$v timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --only layer1,layer2,layer3,db_l1,db_l3 --reps 30 > gpurun_out/exp4/kxr2_$v.txt 2>&1
  AGP_KXR2=$v timeout 300 python tools/conv_bench.py --prec 4 --batch 32 --res 1 --only layer1,layer2,layer3,db_l1,db_l3 --reps 30 > gpurun_out/exp4/kxr2_${v}_b32.txt 2>&1
done
grep -h "layer\|db_" gpurun_out/exp4/kxr2_*.txt
