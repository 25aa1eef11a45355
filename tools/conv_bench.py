#!/usr/bin/env python3
"""Micro-benchmark of agp_conv2d_fwd on the bench workload's layer shapes (tuning aid).

python tools/conv_bench.py [--prec 3] [--reps 20] [--only layer1] [--batch 32]
Prints per-shape time and algorithmic TFLOP/s; all timings interleaved in one process.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--kxr-variant" in sys.argv:         # the development build's switches: its library has to be chosen before the package loads
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _tuning  # noqa: E402,F401
import torch  # noqa: E402

from agplace_amd import ops  # noqa: E402

SHAPES = {
    # name: (cin, cout, k, stride, pad, hin, win)
    "stem": (3, 64, 7, 2, 3, 224, 1344),
    "layer1": (64, 64, 3, 1, 1, 56, 336),
    "l2ds": (64, 128, 1, 2, 0, 56, 336),
    "l2c1": (64, 128, 3, 2, 1, 56, 336),
    "layer2": (128, 128, 3, 1, 1, 28, 168),
    "l3c1": (128, 256, 3, 2, 1, 28, 168),
    "layer3": (256, 256, 3, 1, 1, 14, 84),
    "l1_k2": (128, 64, 3, 1, 1, 56, 336),      # experiments: layer 1's map with twice the K depth / twice the output channels
    "l1_n2": (64, 128, 3, 1, 1, 56, 336),
    "kw_k576": (64, 128, 3, 1, 1, 28, 168),      # the wide kernel (cout 128) on layer 2's map at K = 576 / 1152 (= layer2) / 2304:
    "kw_k2304": (256, 128, 3, 1, 1, 28, 168),    # time = fixed cost per tile + K * slope (profiles/README.md, round 6)
    "db_l1": (64, 64, 3, 1, 1, 56, 56),
    "db_l3": (256, 256, 3, 1, 1, 14, 14),
    "t_l1": (64, 64, 3, 1, 1, 64, 64),         # the training step's 256 x 256 aerial tiles (--batch 176)
    "t_l2": (128, 128, 3, 1, 1, 32, 32),
    "t_l3": (256, 256, 3, 1, 1, 16, 16),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", type=int, default=2)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--res", type=int, default=0, help="1: add a residual map in the epilogue")
    ap.add_argument("--zeros", type=int, default=0, help="1: all-zero operands (clock-under-load experiment)")
    ap.add_argument("--group", type=int, default=0, help="1: layer1/2/3 as the bench issues them: the query problem and the database "
                                                         "problem (same batch of 224x224 tiles) in ONE grouped launch")
    ap.add_argument("--kxr-variant", type=int, default=0, help="development build (make tuning): the KXR_VARIANT switch of igemm_kxr.hip")
    a = ap.parse_args()
    if "--kxr-variant" in sys.argv:
        _tuning.set_switch("KXR_VARIANT", a.kxr_variant)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, (cin, cout, k, s, p, h, w) in SHAPES.items():
        if a.only and name not in a.only.split(","):
            continue
        n = a.batch
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        if a.zeros:
            wt.zero_()
        stem = name == "stem"
        if stem:
            xm = ops.pack_f32(torch.randn(n, 3, h, w, generator=g).to(dev), 4, 3, a.prec)
        else:
            xm = ops.SplitMap.alloc(n, h, w, cin, 1, a.prec, dev)
            if not a.zeros:
                xm.hi[:, 1:-1, 1:-1].normal_()
            if xm.lo is not None:
                xm.lo[:, 1:-1, 1:-1].normal_(std=2 ** -9)
        cw = ops.ConvWeights(wt.to(dev), torch.ones(cout, device=dev), torch.zeros(cout, device=dev), s, p, stem=stem)
        ho, wo = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
        out = ops.SplitMap.alloc(n, ho, wo, cout, 1, a.prec, dev)
        res = None
        if a.res:
            res = ops.SplitMap.alloc(n, ho, wo, cout, 1, a.prec, dev)
            res.hi[:, 1:-1, 1:-1].normal_()
        jobs = [(xm, cw, out, res, True)]
        fl = 2.0 * n * ho * wo * cout * cw.alg_k
        if a.group and name in ("layer1", "layer2", "layer3"):
            hd, wd = h, w // 6
            xd = ops.SplitMap.alloc(n, hd, wd, cin, 1, a.prec, dev)
            xd.hi[:, 1:-1, 1:-1].normal_()
            wd_t = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
            cwd = ops.ConvWeights(wd_t.to(dev), torch.ones(cout, device=dev), torch.zeros(cout, device=dev), s, p)
            od = ops.SplitMap.alloc(n, hd, wd, cout, 1, a.prec, dev)
            rd = None
            if a.res:
                rd = ops.SplitMap.alloc(n, hd, wd, cout, 1, a.prec, dev)
                rd.hi[:, 1:-1, 1:-1].normal_()
            jobs.append((xd, cwd, od, rd, True))
            fl += 2.0 * n * hd * wd * cout * cw.alg_k
        for _ in range(3):
            ops.conv2d_grouped(jobs, a.prec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            ops.conv2d_grouped(jobs, a.prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        nprod = {2: 2, 3: 3, 4: 1}[a.prec]
        print(f"{name:8s} M={n * ho * wo:7d} N={cout:4d} K={cw.kh * cw.kw * cw.cin:5d}  {ms * 1e3:8.1f} us  "
              f"{fl / ms / 1e9:7.1f} TFLOP/s algorithmic  ({nprod * fl / ms / 1e9:7.1f} MFMA)")


if __name__ == "__main__":
    main()
