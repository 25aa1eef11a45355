timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_line_a.json 2> gpurun_out/r04_line_a.err; tail -c 600 gpurun_out/r04_line_a.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_line_a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['ms_per_step_one_in_flight'], d['config']['library_default'])
print(d['roofline']['frac'], d['roofline']['launches_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['conv_family']['frac'])
print(d['knn']['value'], d['knn'].get('parity'))
print(d['train'])
print(d['cpu_baseline']['value'])
PY
