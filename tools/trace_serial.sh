#!/bin/bash
# rocprofv3 kernel trace + stats of the serial (one stream, no graph) bench pass -> gpurun_out/<tag>_trace ; prints the per-kernel table
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--no-cpu-baseline --no-knn --no-netvlad --vox-leg 0 --windows 1 --train-steps 0 --graph 0 --streams 1 --qsplit 1 --steps 40 --warmup 2"
mkdir -p $R/gpurun_out/${tag}_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o t -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cd $R
f=$(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# the bench makes 2+2+2 (warm-up) + 7 (steps) + 2x2 + 1 + 1 = 19 embed passes: normalise per pass by the kxr2 launch count (12 per pass)
kx = [r for r in rows if "igemm_kxr2" in r["Name"]]
passes = sum(int(r["Calls"]) for r in kx) / 12.0
tot = 0
for r in rows[:22]:
    us = float(r["TotalDurationNs"]) / 1e3 / passes
    tot += us
    print(f"{us:9.1f} us/pass  {int(r['Calls'])/passes:6.2f} calls/pass  avg {float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:110]}")
print("passes", passes, "listed total us/pass", round(tot, 1))
PY
find $R/gpurun_out -name "*kernel_trace.csv" -size +40M -delete
