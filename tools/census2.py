#!/usr/bin/env python3
"""Per-workgroup timeline of the kxr2 conv kernel (debug aid): needs the development library with the census stamps,
    make -C agplace_amd/csrc tuning EXTRA=-DAGP_CENSUS=1
    python tools/census2.py [n] [layer1|layer2|layer3] [res]
Prints medians (us) of the phases of a workgroup's life and how workgroups follow each other on a CU."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch  # noqa: E402

from agplace_amd import ops  # noqa: E402

_tuning.set_switch("IGEMM_DBG", 0x1000000)

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
layer = sys.argv[2] if len(sys.argv) > 2 else "layer1"
use_res = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cin, h, w = {"layer1": (64, 56, 336), "layer2": (128, 28, 168), "layer3": (256, 14, 84)}[layer]
cout = cin
xm = ops.SplitMap.alloc(n, h, w, cin, 1, 4, dev)
xm.hi[:, 1:-1, 1:-1].normal_()
cw = ops.ConvWeights(torch.randn(cout, cin, 3, 3, device=dev) / 24, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), 1, 1)
out = ops.SplitMap.alloc(n, h, w, cout, 1, 4, dev)
res = ops.SplitMap.alloc(n, h, w, cout, 1, 4, dev) if use_res else None
if res is not None:
    res.hi[:, 1:-1, 1:-1].normal_()
M = n * h * (w + 2)
nwg = (((M + 255) // 256 + 7) // 8 * 8) * ((cout + 63) // 64) + 64
rec = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
_tuning.set_switch("CENSUS_BUF_LO", (rec.data_ptr() & 0xffffffff) - (1 << 32) if rec.data_ptr() & 0x80000000 else rec.data_ptr() & 0xffffffff)
_tuning.set_switch("CENSUS_BUF_HI", rec.data_ptr() >> 32)
for _ in range(3):
    rec.zero_()
    ops.conv2d(xm, cw, out, residual=res, relu=True, prec=4)
torch.cuda.synchronize()
r = rec.view(-1, 64).cpu()
r = r[r[:, 3] != 0]
hw, xcc = r[:, 0], r[:, 1] & 0xf
t0, t1 = r[:, 2], r[:, 3]
s = r[:, 4:8]
us = lambda t: t.double() / 100.0
print(f"{layer} n={n} res={use_res}: workgroups {len(r)}  kernel span {float(us(t1.max() - t0.min())):.1f} us")
parts = {"prologue arithmetic": s[:, 0] - t0, "first stage wait": s[:, 1] - s[:, 0], "main loop": s[:, 2] - s[:, 1],
         "epilogue issue": s[:, 3] - s[:, 2], "store drain": t1 - s[:, 3], "total": t1 - t0}
for k, v in parts.items():
    v = us(v)
    print(f"  {k:22s} median {float(v.median()):7.2f}  mean {float(v.mean()):7.2f}  p90 {float(v.quantile(0.9)):7.2f} us")
cu = ((hw >> 8) & 0xf); sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = (xcc * 10000 + se * 1000 + sh * 100 + cu).tolist()
per = collections.defaultdict(list)
for k, a, b in zip(key, t0.tolist(), t1.tolist()):
    per[k].append((a, b))
print("  distinct CUs", len(per), " workgroups per CU:", dict(collections.Counter(len(v) for v in per.values())))
# concurrency over time on a CU and idle gaps
busy_frac = []
for k, iv in per.items():
    ev = sorted([(a, 1) for a, b in iv] + [(b, -1) for a, b in iv])
    c, last, area = 0, ev[0][0], 0
    for t, d in ev:
        area += c * (t - last)
        last = t
        c += d
    busy_frac.append(area / max(1, (ev[-1][0] - ev[0][0])))
bf = torch.tensor(busy_frac)
print(f"  mean resident workgroups per CU over its active span: {float(bf.mean()):.2f} (min {float(bf.min()):.2f})")
