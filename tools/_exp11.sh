#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp11
for st in 0 -1 300 1000; do
  echo "== stagger $st"
  AGP_KXR2_STAGGER=$st timeout 300 python tools/conv_bench.py --prec 4 --batch 64 --res 1 --group 1 --only layer1,layer2,layer3 --reps 30 2>&1 | grep layer
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d" 2>&1 | tail -2
for st in 0 -1; do
AGP_KXR2_STAGGER=$st timeout 600 python bench.py --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp11/bench_$st.json 2> gpurun_out/exp11/bench_$st.err
python - <<PY
import json
d=json.loads(open('gpurun_out/exp11/bench_$st.json').read().strip().splitlines()[-1]); r=d['roofline']
print('stagger $st', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'fam', r['conv_family']['frac'], 'eager', r['embed_ms_per_step_eager'])
PY
done
