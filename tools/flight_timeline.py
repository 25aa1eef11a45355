#!/usr/bin/env python3
"""Several replayed steps IN FLIGHT (bench.py's default: step i on stream i % 2) as a timeline:
    python3 tools/flight_timeline.py <trace dir> [out.txt]
(trace dir = rocprofv3 --kernel-trace --output-format csv of `bench.py --no-cpu-baseline --no-knn --train-steps 0 --steps 24`).
Takes a steady-state window of EIGHT steps (stem launches count the steps), prints its wall time per step, the union of kernel
intervals (GPU busy), the time with only latency-class kernels running, the mean number of kernels running, and the kernels of
the first 2.2 ms of the window with the queue they ran on (two steps' kernels interleave)."""
import csv
import glob
import os
import sys

from step_timeline import short


def main():
    d = sys.argv[1]
    k = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(k)))
    ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
    stems = [i for i, (_, _, n, _) in enumerate(ker) if "stem_" in n and "kernel" in n]
    per_step = 2                       # whole-batch launches: one stem launch per trunk and step
    nsteps = 8
    starts = stems[::per_step]
    best, bi = None, 0
    for i in range(2, len(starts) - nsteps - 2):
        span = ker[starts[i + nsteps]][0] - ker[starts[i]][0]
        if best is None or span < best:
            best, bi = span, i
    t0, t1 = ker[starts[bi]][0], ker[starts[bi + nsteps]][0]
    sel = [x for x in ker if x[0] < t1 and x[1] > t0]
    light = ("vecprog", "pool_from_conv", "bcast_add", "split_f32")
    ev = []
    for s, e, n, _ in sel:
        ev.append((max(s, t0), 1, n)); ev.append((min(e, t1), -1, n))
    busy = light_only = 0
    cur, last = [], None
    for t, dlt, n in sorted(ev, key=lambda x: (x[0], x[1])):
        if last is not None and cur:
            busy += t - last
            if all(any(x in c for x in light) for c in cur):
                light_only += t - last
        if dlt > 0:
            cur.append(n)
        else:
            cur.remove(n)
        last = t
    wall = (t1 - t0) / 1e3
    dur = sum(min(e, t1) - max(s, t0) for s, e, _, _ in sel) / 1e3
    out = []
    queues = sorted({q for _, _, _, q in sel})
    for s, e, n, q in sel:
        if s - t0 > 2.2e6 or s < t0:
            continue
        conc = sum(1 for s2, e2, _, _ in sel if s2 < e and e2 > s) - 1
        out.append(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  queue {queues.index(q)}  beside {conc}  {short(n)}")
    out.append(f"window of {nsteps} steps: {wall / nsteps:.1f} us per step, GPU busy {busy / 1e3 / nsteps:.1f} us per step, idle "
               f"{(wall - busy / 1e3) / nsteps:.1f} us per step, only latency-class kernels running {light_only / 1e3 / nsteps:.1f} us per step, "
               f"sum of kernel durations {dur / nsteps:.1f} us per step (mean {dur / (busy / 1e3):.2f} kernels running), {len(queues)} queues")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
