#!/bin/bash
# A/B of two library builds on the kNN search (agplace_amd/lib/variants/libA.so, libB.so; AGP_HIP_LIB selects)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in A B; do
AGP_HIP_LIB=$R/agplace_amd/lib/variants/lib$v.so python3 - <<PY
import torch, time
from agplace_amd import retrieval
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
db = torch.randn(100000, 256, generator=g); db = (db / db.norm(dim=1, keepdim=True)).to(dev)
q = torch.randn(4096, 256, generator=g); q = (q / q.norm(dim=1, keepdim=True)).to(dev)
ix = retrieval.IndexFlatL2(256, device=dev, prec=4); ix.add(db)
for _ in range(3): D0, I0 = ix.search_device(q, 20)
blocks = []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): D, I = ix.search_device(q, 20)
    torch.cuda.synchronize(); blocks.append(time.perf_counter() - t0)
dt = sorted(blocks)[3]
print("lib $v", "Mq/s", round(4096 * 20 / dt / 1e6, 2), "ms/search", round(dt / 20 * 1e3, 4), "checksum", int(I.sum()))
PY
done
done
