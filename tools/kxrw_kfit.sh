#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of igemm_kxrw on ONE map at three K depths -> fixed cost per tile and slope
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/${tag}_kfit
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_kfit -o t -- python3 $R/tools/conv_bench.py --prec 4 --batch 64 --reps 30 --only kw_k576,layer2,kw_k2304,layer3 ${2} > $R/gpurun_out/${tag}_kfit/stdout.txt 2>&1
cd $R
python3 - <<'PY' "$(find gpurun_out/${tag}_kfit -name '*kernel_trace.csv' | head -1)"
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "igemm_kxrw" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# conv_bench runs each shape 3 + reps times in order
groups = collections.OrderedDict()
for r in rows:
    key = (r["Grid_Size_X"], r["LDS_Block_Size"])
    groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
seq = []
prev = None
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not seq or len(seq[-1]) >= 33:
        seq.append([])
    seq[-1].append(d)
for i, s in enumerate(seq):
    s2 = sorted(s[3:])
    print(f"shape {i}: n {len(s)} median {s2[len(s2)//2]:.1f} us min {s2[0]:.1f}")
PY
