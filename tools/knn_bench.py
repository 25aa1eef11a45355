#!/usr/bin/env python3
"""Micro-benchmark of the exact kNN search (100k x 256 database, k = 20)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from agplace_amd import retrieval  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=100000)
    ap.add_argument("--nq", type=int, default=4096)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--prec", type=int, default=3, help="3 split-bf16, 1 bf16, 4 fp16 coarse pass")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    db = torch.randn(a.nb, 256, generator=g)
    db = (db / db.norm(dim=1, keepdim=True)).to(dev)
    q = torch.randn(a.nq, 256, generator=g)
    q = (q / q.norm(dim=1, keepdim=True)).to(dev)
    idx = retrieval.IndexFlatL2(256, device=dev, prec=a.prec)
    idx.add(db)
    idx.search_device(q, a.k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        idx.search_device(q, a.k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print(f"nb={a.nb} nq={a.nq} k={a.k} prec={a.prec}: {ms:.3f} ms/search  {a.nq / ms * 1e3:.0f} queries/s  "
          f"{a.nq * 2 * a.nb * 256 / ms / 1e9:.1f} TFLOP/s algorithmic")


if __name__ == "__main__":
    main()
