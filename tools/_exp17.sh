#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp17
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d" 2>&1 | tail -2
AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so timeout 200 python tools/census2.py 64 layer1 1 2>&1 | grep -v amdgpu.ids
AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so timeout 200 python tools/census2.py 64 layer1 0 2>&1 | grep -v amdgpu.ids | head -8
timeout 600 python bench.py --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp17/bench.json 2> gpurun_out/exp17/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/exp17/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'fam', r['conv_family']['frac'], 'eager', r['embed_ms_per_step_eager'])
PY
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "c2" 2>&1 | grep -v "^$" | tail -25
for g in 1 0; do timeout 600 python bench.py --h2d --graph $g --no-cpu-baseline --no-knn --train-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h2d graph $g', d['value'], d['ms_per_step'])"; done
