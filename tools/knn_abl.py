#!/usr/bin/env python3
"""Ablation timing of the four-wave coarse kNN kernel in the development build (results are wrong with KNN_ABL != 0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402,F401
import torch  # noqa: E402

from agplace_amd import retrieval  # noqa: E402
from knn_ab import timed  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
db = torch.randn(100000, 256, generator=g)
db = (db / db.norm(dim=1, keepdim=True)).to(dev)
q = torch.randn(4096, 256, generator=g)
q = (q / q.norm(dim=1, keepdim=True)).to(dev)
idx = retrieval.IndexFlatL2(256, device=dev, prec=4)
idx.add(db)
for abl in [int(x) for x in (sys.argv[1:] or ["0", "0", "1", "4", "8", "13", "0"])]:
    _tuning.set_switch("KNN_ABL", abl)
    print(f"KNN_ABL={abl}: coarse {timed(lambda: idx.coarse_pass_device(q), 30) * 1e3:.0f} us", flush=True)
