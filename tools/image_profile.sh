#!/bin/bash
# serial per-kernel durations of the image path (one eager single-stream pass) + the bench figure
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/imgprof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/imgprof -o t -- python3 $R/bench.py --no-cpu-baseline --no-knn --train-steps 0 --default-prec-leg 0 --graph 0 --streams 1 --qsplit 1 --steps 10 --warmup 2 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/imgprof/**/t_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
rm -rf $R/gpurun_out/imgprof
python3 bench.py --no-cpu-baseline --no-knn --train-steps 0 --default-prec-leg 0 --steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'family', d['roofline']['conv_family']['frac'])"
