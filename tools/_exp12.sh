#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp12
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d" 2>&1 | tail -3
AGP_KXR2_STAGGER=0 timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --group 1 --only layer1,layer2,layer3 --reps 30 2>&1 | grep layer
AGP_KXR2_STAGGER=0 timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 0 --group 1 --only layer1,layer2,layer3 --reps 30 2>&1 | grep layer
AGP_KXR2_STAGGER=0 timeout 600 python bench.py --verbose --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp12/bench.json 2> gpurun_out/exp12/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/exp12/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'fam', r['conv_family']['frac'], 'eager', r['embed_ms_per_step_eager'])
PY
grep "^conv" gpurun_out/exp12/bench.err
