#!/usr/bin/env python3
"""One steady-state replayed step of bench.py as a timeline: python3 tools/step_timeline.py <trace dir> [out.txt]
(trace dir = rocprofv3 --kernel-trace --output-format csv of `bench.py --no-cpu-baseline --no-knn --train-steps 0`).
Prints every kernel of the step (start offset, duration, how many other kernels run beside it) and the step's wall time, the
union of kernel intervals (GPU busy) and the time during which ONLY latency-class kernels (vecprog / pool / bcast) run."""
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("void ", "").replace("agp_igemm::", "").replace("agp_fusion::", "").replace("agp_pack::", "").replace("agp_pool::", "")
    return n.split("(")[0][:44]


def main():
    d = sys.argv[1]
    k = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(k)))
    stems = [i for i, (_, _, n) in enumerate(ker) if "stem_" in n and "kernel" in n]
    # stems per step: count in the densest region
    per_step = 4
    starts = stems[::per_step]
    nsteps = 6
    best, bi = None, 0
    for i in range(0, len(starts) - nsteps):
        span = ker[starts[i + nsteps]][0] - ker[starts[i]][0]
        if best is None or span < best:
            best, bi = span, i
    a, b = starts[bi + 2], starts[bi + 3]
    sel = ker[a:b]
    t0 = sel[0][0]
    wall = (ker[b][0] - t0) / 1e3
    out = []
    light = ("vecprog", "pool_from_conv", "bcast_add", "split_f32")
    ev = []
    for s, e, n in sel:
        ev.append((s, 1, n)); ev.append((e, -1, n))
    busy = light_only = 0
    cur = []
    last = None
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
    for s, e, n in sel:
        conc = sum(1 for s2, e2, _ in sel if s2 < e and e2 > s) - 1
        out.append(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  beside {conc}  {short(n)}")
    out.append(f"step wall {wall:.1f} us, GPU busy {busy / 1e3:.1f} us, idle {wall - busy / 1e3:.1f} us, only latency-class kernels running {light_only / 1e3:.1f} us, "
               f"kernels {len(sel)}, sum of durations {sum(e - s for s, e, _ in sel) / 1e3:.1f} us")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
