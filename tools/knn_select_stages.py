#!/usr/bin/env python3
"""Where the selection kernel's time goes: the development build's KNN_DBG switch makes select_rerank_kernel return after
1 = threshold + candidate count, 2 = candidate gather, 3 = exact distances (0 = everything); search time minus coarse pass."""
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
for _ in range(2):
    c = timed(lambda: idx.coarse_pass_device(q), 30)
    print(f"coarse pass {c * 1e3:.0f} us")
    for dbg in (1, 2, 3, 0):
        _tuning.set_switch("KNN_DBG", dbg)
        t = timed(lambda: idx.search_device(q, 20), 30)
        print(f"  KNN_DBG={dbg}: search {t * 1e3:.0f} us, selection ~{(t - c) * 1e3:.0f} us", flush=True)
