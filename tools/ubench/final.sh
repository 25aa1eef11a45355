timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_err.log; tail -c 400 gpurun_out/r04_bench_err.log
python3 - <<'PY'
import json
l=json.loads(open('gpurun_out/r04_bench_line.json').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], l['config']['ms_per_step_one_in_flight'], l['roofline']['frac'], l['roofline']['traffic'], l['knn']['value'], l['knn']['roofline'].get('traffic'), l['knn']['parity']['indices_equal_cpu_port'], l['train']['ms_per_step'], l['train'].get('with_voxel_branch',{}).get('ms_per_step'), l['cpu_baseline']['value'], l['config']['library_default']['ms_per_step'])
PY
