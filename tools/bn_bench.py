"""Times the training step's BatchNorm passes alone (agp_bn_bwd = channel sums + apply, agp_map_affine) on the map shapes of a
16-query training step, and prints the HBM rate of each (bytes = the planes the pass must read and write).

    python tools/bn_bench.py [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agplace_amd import _lib  # noqa: E402
from agplace_amd._lib import check, ptr  # noqa: E402

SHAPES = [("stem", 16, 112, 672, 64, 0), ("layer1", 16, 56, 336, 64, 1), ("layer2", 16, 28, 168, 128, 1),
          ("layer3", 16, 14, 84, 256, 1), ("db_l1", 32, 56, 56, 64, 1), ("db_l3", 32, 14, 14, 256, 1)]


def planes(n, h, w, c, pad, dev, g):
    t = torch.randn(n, h + 2 * pad, w + 2 * pad, c, generator=g, device=dev)
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi, lo


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--absmax", type=int, default=0, help="1: agp_bn_bwd also reduces max |gz| per channel")
    ap.add_argument("--h16", type=int, default=0, help="1: agp_map_affine also writes the fp16 operand plane")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.load()
    g = torch.Generator(device=dev).manual_seed(1)
    for name, n, h, w, c, pad in SHAPES:
        z, gy, y = planes(n, h, w, c, pad, dev, g), planes(n, h, w, c, pad, dev, g), planes(n, h, w, c, pad, dev, g)
        gz = (torch.empty_like(z[0]), torch.empty_like(z[1]))
        gr = (torch.empty_like(z[0]), torch.empty_like(z[1]))
        mean, rstd, gamma = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.ones(c, device=dev)
        gg, gb = torch.empty(c, device=dev), torch.empty(c, device=dev)
        ws = torch.empty(L.agp_train_reduce_workspace_floats(n, h, w, c), dtype=torch.float32, device=dev)
        s = _lib.stream()
        el = n * h * w * c
        amax = torch.zeros(c, dtype=torch.int32, device=dev)       # --absmax: max |gz| per channel (one-pass weight gradient)
        h16 = torch.empty_like(z[0]).view(torch.float16)           # --h16: the fp16 operand plane of the output

        def bwd(res):
            check(L.agp_bn_bwd(ptr(z[0]), ptr(z[1]), ptr(gy[0]), ptr(gy[1]), ptr(y[0]), ptr(y[1]), ptr(mean), ptr(rstd), ptr(gamma),
                               n, h, w, c, pad, 1, ptr(gz[0]), ptr(gz[1]), ptr(gr[0]) if res else None, ptr(gr[1]) if res else None,
                               ptr(gg), ptr(gb), ptr(ws), ptr(amax) if a.absmax else None, s), "agp_bn_bwd")

        def aff(res):
            check(L.agp_map_affine(ptr(z[0]), ptr(z[1]), ptr(gamma), ptr(mean), ptr(y[0]) if res else None, ptr(y[1]) if res else None,
                                   n, h, w, c, pad, 1, ptr(gz[0]), ptr(gz[1]), ptr(h16) if a.h16 else None, s), "agp_map_affine")

        for res in (False, True):
            t = timed(lambda: bwd(res), a.iters)
            # sums: z 4 + gy 4 + y.hi 2; apply: z 4 + gy 4 + y.hi 2 + gz 4 (+ gres 4)
            by = el * (10 + 14 + (4 if res else 0))
            print(f"{name:7s} bn_bwd  res={int(res)}  {t:8.1f} us  {by / t / 1e6:7.2f} TB/s  ({by / 1e6:.0f} MB)")
            t = timed(lambda: aff(res), a.iters)
            by = el * (8 + (4 if res else 0))
            print(f"{name:7s} affine  res={int(res)}  {t:8.1f} us  {by / t / 1e6:7.2f} TB/s  ({by / 1e6:.0f} MB)")


if __name__ == "__main__":
    main()
