#!/bin/bash
# kernel trace of the step with the sparse-voxel branch (bench.py --vox), serial eager passes
tag=${1:-r03vox}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/${tag}_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o t -- python3 $R/bench.py --vox --no-cpu-baseline --graph 0 --streams 1 --qsplit 1 --steps 20 --warmup 2 > /dev/null 2>&1
cd $R
f=$(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
st = [r for r in rows if "stem_walk_kernel" in r["Name"] or "stem_pool_lds_kernel" in r["Name"]]
passes = sum(int(r["Calls"]) for r in st) / 2.0          # one stem launch per trunk and pass (query + database)
for r in rows[:30]:
    us = float(r["TotalDurationNs"]) / 1e3 / passes
    print(f"{us:9.1f} us/pass  {int(r['Calls'])/passes:6.2f} calls/pass  avg {float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:120]}")
print("passes", passes)
PY
find $R/gpurun_out -name "*kernel_trace.csv" -size +40M -delete
