timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -5
echo "== fused on"; timeout 300 python bench.py --no-cpu-baseline --no-knn --train-steps 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('ms_per_step_one_in_flight'), d['roofline']['frac'], d['roofline'].get('launches_per_step'), d['roofline'].get('avg_launch_ms'))"
echo "== fused off"; AGP_FUSED_BLOCK=0 timeout 300 python bench.py --no-cpu-baseline --no-knn --train-steps 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('ms_per_step_one_in_flight'), d['roofline']['frac'], d['roofline'].get('launches_per_step'), d['roofline'].get('avg_launch_ms'))"
