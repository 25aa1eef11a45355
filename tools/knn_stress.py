#!/usr/bin/env python3
"""Repeatability of the exact kNN search: every call must return the same bits, and the first rows must equal an fp64 brute force
(torch on the GPU).  Also interleaves searches of different sizes on one index (shared workspace)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--xcd0" in sys.argv:                 # development build: round 4's workgroup order
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _tuning  # noqa: E402
    sys.argv.remove("--xcd0")
    _tuning.set_switch("KNN_XCD", 0)
import torch  # noqa: E402

from agplace_amd import retrieval  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
db = torch.randn(100000, 256, generator=g)
db = (db / db.norm(dim=1, keepdim=True)).to(dev)
qall = torch.randn(16384, 256, generator=g)
qall = (qall / qall.norm(dim=1, keepdim=True)).to(dev)
idx = retrieval.IndexFlatL2(256, device=dev, prec=4)
idx.add(db)


def brute(q, k):
    d2 = ((q.double() ** 2).sum(1, keepdim=True) + (db.double() ** 2).sum(1)[None] - 2 * q.double() @ db.double().T)
    return torch.topk(d2, k, dim=1, largest=False, sorted=True)


bad = 0
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    for nq in (4096, 512, 1000, 16384, 512, 300, 777, 200):
        q = qall[:nq]
        d, i = idx.search_device(q, 20)
        torch.cuda.synchronize()
        key = nq
        if rnd == 0 and not hasattr(idx, "_ref_%d" % key):
            setattr(idx, "_ref_%d" % key, (d.clone(), i.clone()))
            bd, bi = brute(q[:64], 20)
            ok = torch.equal(bi, i[:64])
            print(f"nq={nq}: first 64 rows equal the fp64 brute force: {ok}", flush=True)
            bad += (not ok)
        else:
            rd, ri = getattr(idx, "_ref_%d" % key)
            if not (torch.equal(rd, d) and torch.equal(ri, i)):
                rows = (ri != i).any(1).nonzero().flatten().tolist()
                print(f"round {rnd} nq={nq}: DIFFERS from the first call in rows {rows[:10]} ({len(rows)} rows)", flush=True)
                r0 = rows[0]
                bd, bi = brute(q[r0:r0 + 1], 20)
                print(f"     this call {i[r0].tolist()}\n     first     {ri[r0].tolist()}\n     fp64      {bi[0].tolist()}", flush=True)
                print(f"     distances this {[round(x, 6) for x in d[r0].tolist()[:20]]}", flush=True)
                bad += 1
print("mismatches:", bad)
