#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/exp26
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/exp26 -o t -- python3 $R/bench.py --no-cpu-baseline --no-knn --train-steps 0 --graph 0 --streams 1 --qsplit 1 --steps 5 --warmup 2 > /dev/null 2>&1
f=$(find $R/gpurun_out/exp26 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "vecprog" in r["Kernel_Name"] or "stem_pool" in r["Kernel_Name"] or "fcode" in r["Kernel_Name"]]
for r in sel[-15:]:
    print(r["Kernel_Name"][:50], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us grid", r.get("Grid_Size_X", r.get("Grid_Size")))
PY
rm -f $f
