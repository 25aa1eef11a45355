#!/usr/bin/env python3
"""One steady-state training step (tools/train_bench.py under rocprofv3 --kernel-trace) as a timeline summary:
python3 tools/train_timeline.py <trace dir>.  Prints the step's wall time, the union of the kernel intervals (GPU busy), the sum of
kernel durations, the idle time, and for each kernel name: launches, summed duration, and the idle time that directly precedes its
launches (bubbles on the dependency chain)."""
import collections
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:60]


def main():
    d = sys.argv[1]
    k = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(k)))
    marks = [i for i, (_, _, n) in enumerate(ker) if "igemm_d16_kernel" in n]
    # the forward stem runs twice per step (panoramas, tiles): steps start at every second marker
    starts = marks[::2]
    a, b = starts[-2], starts[-1]
    sel = ker[a:b]
    t0, t1 = sel[0][0], ker[b][0]
    wall = (t1 - t0) / 1e3
    ev = []
    for s, e, n in sel:
        ev.append((s, 1)); ev.append((min(e, t1), -1))
    busy, depth, last = 0, 0, None
    for t, dlt in sorted(ev):
        if depth > 0:
            busy += t - last
        depth += dlt
        last = t
    tot = sum(e - s for s, e, _ in sel) / 1e3
    print(f"step wall {wall:.0f} us; GPU busy (union) {busy / 1e3:.0f} us; idle {wall - busy / 1e3:.0f} us; sum of kernel durations {tot:.0f} us; "
          f"{len(sel)} launches")
    # idle directly before a launch: nothing running between the latest end so far and this start
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    latest_end = sel[0][0]
    for s, e, n in sel:
        gap = max(0, s - latest_end)
        r = agg[short(n)]
        r[0] += 1; r[1] += (e - s) / 1e3; r[2] += gap / 1e3
        latest_end = max(latest_end, e)
    for n, (c, du, gp) in sorted(agg.items(), key=lambda x: -x[1][1])[:40]:
        print(f"{n:60s} {c:4d} launches {du:8.0f} us   idle in front {gp:6.0f} us")


if __name__ == "__main__":
    main()
