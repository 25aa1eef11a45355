#!/usr/bin/env python3
"""Layer-1 BasicBlock at the bench size: the fused kernel (csrc/fblock64.hip) against the two grouped conv launches.

python tools/bblock_bench.py [--batch 64] [--reps 20] [--pool 1]
The query problem (56 x 336 maps) and the database problem (56 x 56) as one grouped launch each way, interleaved in one process."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from agplace_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--pool", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    probs = []
    for (h, w) in ((56, 336), (56, 56)):
        xm = ops.SplitMap.alloc(a.batch, h, w, 64, 1, 4, dev)
        xm.hi[:, 1:-1, 1:-1].normal_().relu_()
        cws = [ops.ConvWeights((torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev), (0.5 + torch.rand(64, generator=g)).to(dev),
                               (0.1 * torch.randn(64, generator=g)).to(dev), 1, 1) for _ in range(2)]
        mid, o1, o2 = (ops.SplitMap.alloc(a.batch, h, w, 64, 1, 4, dev) for _ in range(3))
        probs.append((xm, cws, mid, o1, o2))
    fl = sum(2 * 2.0 * a.batch * h * w * 64 * 576 for (h, w) in ((56, 336), (56, 56)))

    def unfused():
        pools = [ops.PoolReq(want_mean=True, want_gem=False) if a.pool else None for _ in probs]
        ops.conv2d_grouped([(xm, cws[0], mid, None, True) for (xm, cws, mid, o1, o2) in probs], 4)
        ops.conv2d_grouped([(mid, cws[1], o1, xm, True, pl) for (xm, cws, mid, o1, o2), pl in zip(probs, pools)], 4)

    def fused():
        pools = [ops.PoolReq(want_mean=True, want_gem=False) if a.pool else None for _ in probs]
        ops.bblock64_grouped([(xm, cws[0], cws[1], o2, pl) for (xm, cws, mid, o1, o2), pl in zip(probs, pools)])

    for f in (unfused, fused):
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    same = all(torch.equal(o1.hi, o2.hi) for (_, _, _, o1, o2) in probs)
    print("fused == unfused bitwise:", same)
    for rnd in range(a.rounds):
        for name, f in (("unfused (2 grouped convs)", unfused), ("fused block", fused)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            print(f"round {rnd} {name:28s} {ms * 1e3:8.1f} us per block  {fl / ms / 1e9:7.1f} TFLOP/s algorithmic")


if __name__ == "__main__":
    main()
