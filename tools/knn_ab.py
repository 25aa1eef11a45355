#!/usr/bin/env python3
"""A/B of the coarse kNN kernels in the development build (`make tuning`): KNN_QW=8 (eight waves, round 4) vs 4 (four waves,
one per SIMD): search results must be identical, coarse pass and search timed."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402,F401  (points AGP_HIP_LIB at the tuning library before the package loads)
import torch  # noqa: E402

from agplace_amd import retrieval  # noqa: E402


def timed(fn, reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for nb, nq in ((100000, 4096), (100000, 16384), (99999, 1000), (5000, 777), (130, 600), (1000003, 4096)):
        g = torch.Generator().manual_seed(nb + nq)
        db = torch.randn(nb, 256, generator=g)
        db = (db / db.norm(dim=1, keepdim=True)).to(dev)
        q = torch.randn(nq, 256, generator=g)
        q = (q / q.norm(dim=1, keepdim=True)).to(dev)
        idx = retrieval.IndexFlatL2(256, device=dev, prec=4)
        idx.add(db)
        res = {8: [None, None, 1e9, 1e9], 4: [None, None, 1e9, 1e9]}
        for rnd in range(3):                             # alternate: the clock state of the first measurement on a box is not the later ones'
            for qw in (8, 4):
                _tuning.set_switch("KNN_QW", qw)
                d, i = idx.search_device(q, 20)
                torch.cuda.synchronize()
                r = res[qw]
                r[0], r[1] = d.clone(), i.clone()
                r[2] = min(r[2], timed(lambda: idx.coarse_pass_device(q), a.reps))
                r[3] = min(r[3], timed(lambda: idx.search_device(q, 20), a.reps))
        same = torch.equal(res[8][0], res[4][0]) and torch.equal(res[8][1], res[4][1])
        print(f"nb={nb} nq={nq}: identical={same}  coarse {res[8][2]*1e3:.0f} -> {res[4][2]*1e3:.0f} us   search {res[8][3]*1e3:.0f} -> "
              f"{res[4][3]*1e3:.0f} us  ({nq / res[4][3] / 1e3:.2f} M q/s)", flush=True)


if __name__ == "__main__":
    main()
