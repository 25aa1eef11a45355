#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp16
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d" 2>&1 | tail -2
AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so timeout 200 python tools/census2.py 64 layer1 1 2>&1 | grep -v amdgpu.ids
AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/libagplace_hip_census.so timeout 200 python tools/census2.py 64 layer3 1 2>&1 | grep -v amdgpu.ids
for v in 0 3; do
AGP_KXR2_VARIANT=$v timeout 600 python bench.py --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp16/bench_$v.json 2> gpurun_out/exp16/bench_$v.err
python - <<PY
import json
d=json.loads(open('gpurun_out/exp16/bench_$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('variant $v', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'fam', r['conv_family']['frac'], 'eager', r['embed_ms_per_step_eager'])
PY
done
timeout 600 python bench.py --h2d --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp16/bench_h2d.json 2> gpurun_out/exp16/bench_h2d.err; tail -c 300 gpurun_out/exp16/bench_h2d.err
python - <<PY
import json
d=json.loads(open('gpurun_out/exp16/bench_h2d.json').read().strip().splitlines()[-1]); print('h2d', d['value'], d['ms_per_step'], d['config']['query_input'])
PY
timeout 600 python bench.py --u8 --no-cpu-baseline --no-knn --train-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('u8 resident', d['value'], d['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "pinned or c2 or f16_against or chunking" 2>&1 | tail -3
