#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp15
AGP_KXR2_VARIANT=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d" 2>&1 | tail -2
for v in 0 3; do
echo "== variant $v"
AGP_KXR2_VARIANT=$v timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --group 1 --only layer1,layer2,layer3 --reps 30 2>&1 | grep layer
AGP_KXR2_VARIANT=$v timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --only db_l1,db_l3 --reps 30 2>&1 | grep db_
done
AGP_KXR2_VARIANT=3 AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so timeout 200 python tools/census2.py 64 layer1 1 2>&1 | grep -v amdgpu.ids
AGP_KXR2_VARIANT=3 AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so timeout 200 python tools/census2.py 64 layer3 1 2>&1 | grep -v amdgpu.ids
AGP_KXR2_VARIANT=3 timeout 600 python bench.py --no-cpu-baseline --train-steps 0 > gpurun_out/exp15/bench.json 2> gpurun_out/exp15/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/exp15/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'fam', r['conv_family']['frac'], 'eager', r['embed_ms_per_step_eager'])
k=d['knn']; print(k['value'], k['roofline']['avg_launch_ms'], k['roofline']['search_ms'])
PY
timeout 900 python -m pytest tests/test_gpu_knn.py -x -q -m gpu 2>&1 | tail -2
