#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp14
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d" 2>&1 | tail -2
AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so timeout 200 python tools/census2.py 64 layer1 1 2>&1 | grep -v amdgpu.ids
AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so timeout 200 python tools/census2.py 64 layer3 1 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --verbose --no-cpu-baseline --train-steps 0 > gpurun_out/exp14/bench.json 2> gpurun_out/exp14/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/exp14/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'fam', r['conv_family']['frac'], 'eager', r['embed_ms_per_step_eager'])
print(d.get('knn'))
PY
grep "^conv" gpurun_out/exp14/bench.err | head -8
timeout 900 python -m pytest tests/test_gpu_knn.py tests/test_gpu_mining.py -x -q -m gpu 2>&1 | tail -3
