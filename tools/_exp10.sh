#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp10
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/exp10/tests.txt 2>&1
tail -3 gpurun_out/exp10/tests.txt
timeout 600 python bench.py --verbose --no-cpu-baseline --no-knn --train-steps 0 > gpurun_out/exp10/bench.json 2> gpurun_out/exp10/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/exp10/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'fam', r['conv_family']['frac'], 'eager', r['embed_ms_per_step_eager'])
PY
grep "^conv" gpurun_out/exp10/bench.err
