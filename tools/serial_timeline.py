#!/usr/bin/env python3
"""Per-launch durations of the LAST serial embed pass in a rocprofv3 kernel trace of
`bench.py --graph 0 --streams 1 --qsplit 1` (one stream, no graph): launch order, grid, duration.
usage: serial_timeline.py <dir with *kernel_trace.csv> [npasses]"""
import csv
import glob
import sys

d = sys.argv[1]
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a pass starts at a stem_walk launch that follows a vecprog launch (or the first stem)
starts = [i for i, r in enumerate(rows) if "stem_walk" in r["Kernel_Name"] and (i == 0 or "stem_walk" not in rows[i - 1]["Kernel_Name"])]
starts = [s for k, s in enumerate(starts) if k == 0 or any("vecprog" in rows[j]["Kernel_Name"] for j in range(starts[k - 1], s))]
starts = starts[-npass - 1:]
import collections
acc = collections.OrderedDict()
for k in range(len(starts) - 1):
    seq = rows[starts[k]:starts[k + 1]]
    for j, r in enumerate(seq):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0][-60:]
        key = (j, name, r["Grid_Size_X"], r["Workgroup_Size_X"])
        acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for (j, name, g, w), v in acc.items():
    if len(v) < npass // 2:
        continue
    v.sort()
    med = v[len(v) // 2]
    tot += med
    print(f"{j:3d} {med:8.1f} us  min {v[0]:7.1f}  wgs {int(g)//int(w):6d}  {name}")
print(f"sum of medians {tot:.1f} us over {len(starts)-1} passes")
