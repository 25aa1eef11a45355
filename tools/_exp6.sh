#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp6
timeout 600 python -m pytest tests/test_gpu_mining.py tests/test_gpu_models.py -x -q -m gpu -k "mining or train_mode or soft_positives or triplets" > gpurun_out/exp6/tests.txt 2>&1
tail -5 gpurun_out/exp6/tests.txt
for v in 0 1 2 3 4; do
  AGP_KXR2_VARIANT=$v timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --only layer1,layer2,layer3,db_l1,db_l3 --reps 30 > gpurun_out/exp6/v${v}.txt 2>&1
  AGP_KXR2_VARIANT=$v timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --group 1 --only layer1,layer2,layer3 --reps 30 > gpurun_out/exp6/v${v}_g.txt 2>&1
done
for v in 0 1 2 3 4; do echo "== variant $v"; grep -h "layer\|db_" gpurun_out/exp6/v${v}.txt gpurun_out/exp6/v${v}_g.txt; done
