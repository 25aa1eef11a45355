#!/usr/bin/env python3
"""Workgroup census of the kxr conv kernel: which CU ran each workgroup and when (debug aid)."""
import os, sys, collections
os.environ["AGP_IGEMM_DBG"] = str(0x1000000)   # needs a library built with `make EXTRA=-DAGP_CENSUS=1`
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agplace_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
cin = cout = 64; h, w = 56, 336
xm = ops.SplitMap.alloc(n, h, w, cin, 1, 3, dev); xm.hi[:, 1:-1, 1:-1].normal_()
cw = ops.ConvWeights(torch.randn(cout, cin, 3, 3, device=dev) / 24, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), 1, 1)
out = ops.SplitMap.alloc(n, h, w, cout, 1, 3, dev)
M = n * h * (w + 2); nwg = ((M + 255) // 256 + 7) // 8 * 8
rec = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
fake = ops.SplitMap(rec, rec, n, h, w, cout, 1)     # res_lo pointer carries the record buffer
for _ in range(2):
    rec.zero_(); ops.conv2d(xm, cw, out, residual=fake, relu=True, prec=3)
torch.cuda.synchronize()
full = rec.view(-1, 64).cpu()
r = full[:, :4]
keep = r[:, 3] != 0
full = full[keep]
r = r[keep]
hw, xcc, t0, t1 = r[:, 0], r[:, 1] & 0xf, r[:, 2], r[:, 3]
cu = ((hw >> 8) & 0xf); sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = (xcc * 10000 + se * 1000 + sh * 100 + cu).tolist()
per = collections.defaultdict(list)
for k, a, b in zip(key, t0.tolist(), t1.tolist()):
    per[k].append((a, b))
print("workgroups", len(key), "distinct CUs", len(per), "kernel span (us)", (t1.max() - t0.min()).item() / 100.0)
dur = (t1 - t0).float() / 100.0
print("per-WG duration us: mean %.1f min %.1f max %.1f" % (dur.mean(), dur.min(), dur.max()))
maxc = collections.Counter()
for k, iv in per.items():
    ev = sorted([(a, 1) for a, b in iv] + [(b, -1) for a, b in iv])
    c = m = 0
    for _, d in ev:
        c += d; m = max(m, c)
    maxc[m] += 1
print("max concurrent WGs per CU -> number of CUs:", dict(maxc))
print("WGs per CU histogram:", dict(collections.Counter(len(v) for v in per.values())))

# phase timeline from the in-kernel stamps (shader cycles), median over workgroups
st = full[:, 4:4 + 24].double()
names = ["prologue"] + sum([[f"s{i} issue", f"s{i} landed", f"s{i} compute"] for i in range(6)], []) + ["kloop end", "epilogue end"]
d = (st[:, 1:21] - st[:, 0:20])
med = d.median(0).values
print("phase medians (cycles): total", float((st[:, 20] - st[:, 0]).median()))
for i in range(6):
    print(f"  step{i}: issue {med[3*i]:.0f}  wait {med[3*i+1]:.0f}  compute {med[3*i+2]:.0f}")
print("  kloop->end marker", float(med[18]), " epilogue", float(med[19]))
