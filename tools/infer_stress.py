#!/usr/bin/env python3
"""Bit-repeatability of the inference path: MM.forward_q + DBVanilla2D.forward_db on one batch, N times, with other work (a kNN
search, a differently sized forward) interleaved to vary cache state and timing; every output compared with the first run's."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--prec", type=int, default=4)
    ap.add_argument("--vox-points", type=int, default=0)
    a = ap.parse_args()
    from agplace_amd import _lib, retrieval
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    import bench_inputs as onets
    dev = torch.device("cuda:0")
    _lib.load()
    opt = Options(mfma_precision=a.prec)
    torch.manual_seed(0)
    mq = MM(opt=opt).to(dev).eval()
    mdb = DBVanilla2D("db", opt.features_dim, opt=opt).to(dev).eval()

    def inputs(b, seed):
        data = onets.synth_query(b, 224, 1344, opt, seed=seed)
        data = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in data.items()}
        if a.vox_points > 0:
            coords, feats = onets.synth_cloud_lidar(b, a.vox_points, seed=seed + 1)
            data = {k: v for k, v in data.items() if k not in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse")}
            data["coords"], data["features"] = coords.to(dev), feats.to(dev)
        nmap = len(opt.maptype.split("_"))
        db = {"db_map": torch.randn(b, 1, nmap, 3, 224, 224, generator=torch.Generator().manual_seed(seed + 2)).to(dev)}
        return data, db
    data, db = inputs(a.batch, 100)
    data2, db2 = inputs(max(1, a.batch // 4), 200)
    g = torch.Generator().manual_seed(1)
    xb = torch.randn(20000, 256, generator=g).to(dev)
    idx = retrieval.IndexFlatL2(256, device=dev, prec=4)
    idx.add(xb)

    def fwd(d, m):
        with torch.no_grad():
            fq, fd = mq(d, mode="q"), mdb(m, mode="db")
        torch.cuda.synchronize()
        return {"q." + k: v.clone() for k, v in fq.items() if torch.is_tensor(v)} | {"db." + k: v.clone() for k, v in fd.items() if torch.is_tensor(v)}
    ref = fwd(data, db)
    bad = 0
    for r in range(1, a.runs):
        if r % 2:
            fwd(data2, db2)
        if r % 3 == 0:
            idx.search_device(xb[:700], 5)
        out = fwd(data, db)
        diff = [k for k in ref if not torch.equal(ref[k], out[k])]
        if diff:
            bad += 1
            print(f"run {r}: outputs differ: {diff}", flush=True)
    print(f"{a.runs - 1} repeats of the forward ({len(ref)} outputs, batch {a.batch}, prec {a.prec}, vox_points {a.vox_points}): {bad} differ")


if __name__ == "__main__":
    main()
