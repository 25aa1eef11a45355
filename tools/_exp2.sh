#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp2
for dbg in 0 1024 2048 3072 128 3200; do
  AGP_IGEMM_DBG=$dbg python tools/conv_bench.py --prec 4 --batch 64 --res 1 --only layer1,layer2,layer3,db_l1,db_l3 --reps 30 > gpurun_out/exp2/dbg_$dbg.txt 2>&1
done
AGP_IGEMM_DBG=0 python tools/conv_bench.py --prec 4 --batch 64 --res 0 --only layer1,layer2,layer3 --reps 30 > gpurun_out/exp2/nores.txt 2>&1
AGP_IGEMM_DBG=0 python tools/conv_bench.py --prec 4 --batch 122 --res 1 --only layer3 --reps 30 > gpurun_out/exp2/l3_b122.txt 2>&1
AGP_IGEMM_DBG=0 python tools/conv_bench.py --prec 4 --batch 16 --res 1 --only layer1,layer2,layer3 --reps 30 > gpurun_out/exp2/b16.txt 2>&1
AGP_KXR_VARIANT=12 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --only layer1,layer2,layer3 --reps 30 > gpurun_out/exp2/var12.txt 2>&1
grep -h layer gpurun_out/exp2/dbg_0.txt
