#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -5
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-knn --train-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stem lds ', d['value'], d['ms_per_step'], d['roofline']['frac'])"
AGP_STEM_LDS=0 timeout 600 python bench.py --no-cpu-baseline --no-knn --train-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stem d16 ', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
