#!/usr/bin/env python3
"""Per-kernel time of the steady-state steps in a rocprofv3 kernel trace (CSV) of tools/train_bench.py.
Steps are delimited by the fused-Adam launches; the first steps (allocation, warm-up) are skipped."""
import collections
import csv
import sys

path = sys.argv[1]
nsteady = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
steps = []
for i in ad:
    if not steps or i - steps[-1][-1] > 50:
        steps.append([i])
    else:
        steps[-1].append(i)
a, b = steps[-nsteady - 1][-1] + 1, steps[-1][-1] + 1
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0][:72]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
wall = int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"steady-state steps: {nsteady}; GPU busy {tot / nsteady / 1e6:.3f} ms/step; wall {wall / nsteady / 1e6:.3f} ms/step")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{n:72s} {c // nsteady:5d} {t / nsteady / 1e6:8.3f} ms {100 * t / tot:5.1f}%")
