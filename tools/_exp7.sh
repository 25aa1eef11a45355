#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp7
for v in 5 6; do
AGP_KXR2_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d" > gpurun_out/exp7/tests_v$v.txt 2>&1
tail -3 gpurun_out/exp7/tests_v$v.txt
done
for v in 0 5 6 0 5; do
  AGP_KXR2_VARIANT=$v timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --only layer1,layer2,layer3,db_l1,db_l3 --reps 30 > gpurun_out/exp7/v${v}.txt 2>&1
  AGP_KXR2_VARIANT=$v timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --group 1 --only layer1,layer2,layer3 --reps 30 > gpurun_out/exp7/v${v}_g.txt 2>&1
  echo "== variant $v"; grep -h "layer\|db_" gpurun_out/exp7/v${v}.txt gpurun_out/exp7/v${v}_g.txt
done
AGP_KXR2_VARIANT=5 AGP_IGEMM_DBG=128 timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --only layer1,layer2,layer3 --reps 30 2>&1 | grep layer
