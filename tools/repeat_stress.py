#!/usr/bin/env python3
"""Bit-repeatability of a training step's gradients: the same parameters and inputs, forward + backward N times (no optimizer
step), every parameter gradient compared bit for bit with the first run's.  --vox-points > 0 includes the sparse-voxel branch."""
import argparse
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ndb", type=int, default=11)
    ap.add_argument("--runs", type=int, default=8)
    ap.add_argument("--vox-points", type=int, default=0)
    ap.add_argument("--train-precision", type=int, default=32, choices=[32, 16])
    ap.add_argument("--dgrad-products", type=int, default=3, choices=[3, 1])
    a = ap.parse_args()
    from agplace_amd import _lib, losses
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    import bench_inputs as onets
    dev = torch.device("cuda:0")
    _lib.load()
    opt = Options(mfma_precision=3, train_precision=a.train_precision, train_dgrad_products=a.dgrad_products)
    torch.manual_seed(0)
    mq = MM(opt=opt).to(dev).train()
    mdb = DBVanilla2D("db", opt.features_dim, opt=opt).to(dev).train()
    b = a.batch
    data = onets.synth_query(b, 224, 1344, opt, seed=100)
    data = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in data.items()}
    if a.vox_points > 0:
        coords, feats = onets.synth_cloud_lidar(b, a.vox_points, seed=700)
        data = {k: v for k, v in data.items() if k not in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse")}
        data["coords"], data["features"] = coords.to(dev), feats.to(dev)
    nmap = len(opt.maptype.split("_"))
    db = {"db_map": torch.randn(b, a.ndb, nmap, 3, 256, 256, generator=torch.Generator().manual_seed(200)).to(dev)}
    gen = torch.Generator().manual_seed(300)
    data["query_eastnorth"] = (torch.rand(b, 2, generator=gen) * 60).to(dev)
    data["db_eastnorth"] = (torch.rand(b, a.ndb, 2, generator=gen) * 60).to(dev)
    per, negs = 1 + a.ndb, a.ndb - 1
    trip = torch.tensor([[per * i, per * i + 1, per * i + 2 + j] for i in range(b) for j in range(negs)]).to(dev)
    largs = types.SimpleNamespace(criterion="triplet", train_batch_size=b, negs_num_per_query=negs, margin=opt.margin)
    named = [("q." + n, p) for n, p in mq.named_parameters()] + [("db." + n, p) for n, p in mdb.named_parameters()]
    side = torch.cuda.Stream(device=dev)

    def grads():
        for _, p in named:
            p.grad = None
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            fd = mdb(db, mode="db")
        fq = mq(data, mode="q")
        cur.wait_stream(side)
        q, d = fq["embedding"], fd["embedding"]
        loss = losses.compute_other_loss(fq, fd, data, opt.train_positives_dist_threshold, opt.val_positive_dist_threshold, opt=opt)
        feats = torch.cat((q.unsqueeze(1), d), dim=1).view(-1, q.shape[-1])
        loss = loss + losses.compute_loss(largs, None, trip, feats) * opt.tripletloss_weight
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.clone() for n, p in named if p.grad is not None}, q.detach().clone(), d.detach().clone()

    l0, g0, q0, d0 = grads()
    bad = 0
    for r in range(1, a.runs):
        l, g, q, d = grads()
        diff = [n for n in g0 if not torch.equal(g0[n], g[n])]
        fwd = torch.equal(q, q0) and torch.equal(d, d0)
        if diff or not fwd or l != l0:
            bad += 1
            worst = max(((g0[n] - g[n]).abs().max() / (g0[n].abs().max() + 1e-30)).item() for n in diff) if diff else 0.0
            print(f"run {r}: forward identical={fwd}, loss identical={l == l0}, {len(diff)} of {len(g0)} gradients differ "
                  f"(largest relative difference {worst:.1e}): {diff[:6]}", flush=True)
    print(f"{a.runs - 1} repeats of forward + backward ({len(g0)} parameter gradients, vox_points={a.vox_points}): {bad} differ from the first run")


if __name__ == "__main__":
    main()
