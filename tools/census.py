#!/usr/bin/env python3
"""Workgroup census of the kxr conv kernel: which CU ran each workgroup and when (debug aid)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning                                 # the development library, built with `make tuning EXTRA=-DAGP_CENSUS=1`
import torch
from agplace_amd import ops
_tuning.set_switch("IGEMM_DBG", 0x1000000)
dev = torch.device("cuda:0")
# usage: census.py [n] [layer1|layer2|layer3] [prec] [res]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
layer = sys.argv[2] if len(sys.argv) > 2 else "layer1"
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
use_res = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cin, h, w = {"layer1": (64, 56, 336), "layer2": (128, 28, 168), "layer3": (256, 14, 84)}[layer]
cout = cin
xm = ops.SplitMap.alloc(n, h, w, cin, 1, prec, dev); xm.hi[:, 1:-1, 1:-1].normal_()
cw = ops.ConvWeights(torch.randn(cout, cin, 3, 3, device=dev) / 24, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), 1, 1)
out = ops.SplitMap.alloc(n, h, w, cout, 1, prec, dev)
res = ops.SplitMap.alloc(n, h, w, cout, 1, prec, dev) if use_res else None
M = n * h * (w + 2); nwg = (((M + 63) // 64) * ((cout + 63) // 64) + 7) // 8 * 8 + 64    # upper bound over tilings
rec = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
_tuning.set_switch("CENSUS_BUF_LO", (rec.data_ptr() & 0xffffffff) - (1 << 32) if rec.data_ptr() & 0x80000000 else rec.data_ptr() & 0xffffffff)
_tuning.set_switch("CENSUS_BUF_HI", rec.data_ptr() >> 32)
for _ in range(2):
    rec.zero_(); ops.conv2d(xm, cw, out, residual=res, relu=True, prec=prec)
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
ns = 3 * cin // 32
nst = 1 + 3 * ns + 2
if nst <= 56:
    st = full[:, 4:4 + nst].double()
    d = st[:, 1:] - st[:, :-1]
    med = d.median(0).values
    print("phase medians (cycles): total", float((st[:, nst - 1] - st[:, 0]).median()), "steps", ns)
    iss = med[0:3 * ns:3]; wait = med[1:3 * ns:3]; comp = med[2:3 * ns:3]
    print(f"  per macro-step (median of medians): issue {iss.median():.0f}  wait {wait.median():.0f}  compute {comp.median():.0f}")
    print(f"  sum over steps: issue {iss.sum():.0f}  wait {wait.sum():.0f}  compute {comp.sum():.0f}")
    print("  kloop->end marker", float(med[3 * ns]), " epilogue", float(med[3 * ns + 1]))
