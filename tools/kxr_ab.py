#!/usr/bin/env python3
"""A/B of the three-product 3x3 stride-1 conv kernels (development build, KXR_VARIANT): 0 = the shipped forms, 7 = the
one-wave-per-SIMD loop (RING 4).  Outputs compared bit for bit, then timed (alternating rounds, minimum)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402,F401
import torch  # noqa: E402

from agplace_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
variants = [int(v) for v in (sys.argv[1:] or ["0", "7"])]
shapes = [("tiles l2", 176, 128, 128, 32, 32), ("tiles l3", 176, 256, 256, 16, 16), ("pano l2", 16, 128, 128, 28, 168),
          ("pano l3", 16, 256, 256, 14, 84), ("ragged", 3, 128, 256, 13, 37)]
for name, n, cin, cout, h, w in shapes:
    g = torch.Generator().manual_seed(n + cin)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
    xm = ops.pack_f32(x, cin, 1, 3)
    cw = ops.ConvWeights(wt, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), 1, 1)
    outs, times = {}, {v: 1e9 for v in variants}
    for v in variants:
        _tuning.set_switch("KXR_VARIANT", v)
        out = ops.SplitMap.alloc(n, h, w, cout, 1, 3, dev)
        ops.conv2d(xm, cw, out, relu=False, prec=3)
        torch.cuda.synchronize()
        outs[v] = (out.hi.clone(), out.lo.clone())
    for rnd in range(3):
        for v in variants:
            _tuning.set_switch("KXR_VARIANT", v)
            out = ops.SplitMap.alloc(n, h, w, cout, 1, 3, dev)
            for _ in range(3):
                ops.conv2d(xm, cw, out, relu=False, prec=3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                ops.conv2d(xm, cw, out, relu=False, prec=3)
            e1.record()
            torch.cuda.synchronize()
            times[v] = min(times[v], e0.elapsed_time(e1) / 20 * 1e3)
    v0 = variants[0]
    same = all(torch.equal(outs[v][0], outs[v0][0]) and torch.equal(outs[v][1], outs[v0][1]) for v in variants)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, 1, 1)
    err = float((ops.SplitMap(outs[variants[-1]][0], outs[variants[-1]][1], n, h, w, cout, 1).to_f32().double() - ref).norm() / ref.norm())
    print(f"{name:9s} {n:4d}x{cin:3d}->{cout:3d} {h:3d}x{w:3d}: " + "  ".join(f"v{v} {times[v]:7.1f} us" for v in variants)
          + f"   identical={same}  rel err vs fp64 {err:.1e}", flush=True)
