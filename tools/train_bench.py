#!/usr/bin/env python3
"""Training-step benchmark: forward + backward + Adam of MM (query) and DBVanilla2D (database).

Workload (train.py:303-341 shape): per GPU and step `--batch` queries, each one
6-camera panorama [3,224,1344] plus `--ndb` aerial tiles [3,256,256] (1 positive + 10 negatives in
the reference: ndb=11).  Loss: mean squared distance between the query embedding and its tiles'
embeddings by default; `--loss ref` = the reference's step loss (train.py:319-331): compute_other_loss +
TripletMarginLoss over the 10 negatives of every query, both on the fused HIP loss kernels.
Under torch.distributed (RCCL) gradients are all-reduced in flat buckets after backward.

Prints one JSON line: ms/step, queries/s, images/s, and the convention-based pairs-equivalents/s
(1 query + ndb tiles = ndb pairs-equivalents).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ndb", type=int, default=11)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prec", type=int, default=3)
    ap.add_argument("--tile", type=int, default=256)
    ap.add_argument("--loss", type=str, default="ref", choices=["ref", "mse"])
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: the database network's forward (and hence its backward) runs on a second stream")
    ap.add_argument("--graph", type=int, default=0, help="1: capture the whole step (fwd+bwd+Adam) in a hipGraph")
    ap.add_argument("--vox-points", type=int, default=0,
                    help="> 0: train the sparse-voxel branch from coords / features (that many requested voxels per query) instead of "
                         "feeding its outputs as fixed tensors")
    ap.add_argument("--wgrad-f16", type=str, default="all", choices=["all", "s1", "s1+gather", "none"],
                    help="A/B: which weight gradients run as one fp16 product (train_graph.WGRAD_F16*)")
    ap.add_argument("--train-precision", type=int, default=32, choices=[32, 16], help="Options.train_precision (16 = one-product forward convs)")
    ap.add_argument("--dgrad-products", type=int, default=3, choices=[3, 1], help="Options.train_dgrad_products (1 = one bf16 product)")
    ap.add_argument("--fuse-bn-stats", type=int, default=1, choices=[0, 1], help="A/B: BatchNorm statistics from the conv epilogues (train_graph.FUSE_BN_STATS)")
    ap.add_argument("--y16-only", type=int, default=1, choices=[0, 1], help="A/B: conv1 outputs of a block as ONE fp16 plane in the fast mode")
    ap.add_argument("--fwd16-entries", type=int, default=1, choices=[0, 1], help="A/B: the fast mode's one-product forward also on the stage entries")
    args = ap.parse_args()
    import types
    from agplace_amd import train_graph
    train_graph.FUSE_BN_STATS = bool(args.fuse_bn_stats)
    train_graph.FWD_F16_ENTRIES = bool(args.fwd16_entries)
    train_graph.Y16_ONLY = bool(args.y16_only)
    train_graph.WGRAD_F16 = args.wgrad_f16 != "none"
    train_graph.WGRAD_F16_GATHER = args.wgrad_f16 in ("all", "s1+gather")
    train_graph.WGRAD_F16_STEM = args.wgrad_f16 == "all"
    from agplace_amd import _lib, losses, parallel
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    import bench_inputs as onets

    rank, world, local = parallel.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _lib.load()
    opt = Options(mfma_precision=args.prec, train_precision=args.train_precision, train_dgrad_products=args.dgrad_products)
    torch.manual_seed(0)
    mq = MM(opt=opt).to(dev).train()
    mdb = DBVanilla2D("db", opt.features_dim, opt=opt).to(dev).train()
    b = args.batch
    data = onets.synth_query(b, 224, 1344, opt, seed=100 + rank)
    data = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in data.items()}
    if args.vox_points > 0:
        coords, feats = onets.synth_cloud_lidar(b, args.vox_points, seed=700 + rank)
        data = {k: v for k, v in data.items() if k not in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse")}
        data["coords"], data["features"] = coords.to(dev), feats.to(dev)
    nmap = len(opt.maptype.split("_"))
    db = {"db_map": torch.randn(b, args.ndb, nmap, 3, args.tile, args.tile,
                                generator=torch.Generator().manual_seed(200 + rank)).to(dev)}
    gen = torch.Generator().manual_seed(300 + rank)
    data["query_eastnorth"] = (torch.rand(b, 2, generator=gen) * 60).to(dev)
    data["db_eastnorth"] = (torch.rand(b, args.ndb, 2, generator=gen) * 60).to(dev)
    per = 1 + args.ndb
    negs = args.ndb - 1
    trip = torch.tensor([[per * i, per * i + 1, per * i + 2 + j] for i in range(b) for j in range(negs)]).to(dev)
    largs = types.SimpleNamespace(criterion="triplet", train_batch_size=b, negs_num_per_query=negs, margin=opt.margin)
    params = [p for p in list(mq.parameters()) + list(mdb.parameters()) if p.requires_grad]
    from agplace_amd.train_fns import reference_optimizers
    optim_db, optim_q = reference_optimizers(mdb, mq, fused=True, capturable=bool(args.graph))      # train.py:165-190, 213-214

    side = torch.cuda.Stream(device=dev) if args.streams == 2 else None
    # N > 1: every gradient is a view into one flat buffer, all-reduced bucket by bucket while backward runs
    gb = parallel.GradBuckets(params) if world > 1 else None

    def step():
        if gb is not None:
            gb.zero_grad()
        else:
            optim_db.zero_grad(set_to_none=True)
            optim_q.zero_grad(set_to_none=True)
        if side is not None:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                fd = mdb(db, mode="db")
            fq = mq(data, mode="q")
            cur.wait_stream(side)
        else:
            fq = mq(data, mode="q")
            fd = mdb(db, mode="db")
        q, d = fq["embedding"], fd["embedding"]
        if args.loss == "mse":
            loss = ((q[:, None, :] - d) ** 2).sum(-1).mean()
        else:
            loss = losses.compute_other_loss(fq, fd, data, opt.train_positives_dist_threshold,
                                             opt.val_positive_dist_threshold, opt=opt)
            feats = torch.cat((q.unsqueeze(1), d), dim=1).view(-1, q.shape[-1])
            loss = loss + losses.compute_loss(largs, None, trip, feats) * opt.tripletloss_weight
        loss.backward()
        if gb is not None:
            gb.finish()
        optim_db.step()
        optim_q.step()
        return loss

    for _ in range(args.warmup):
        step()
    if args.graph:
        torch.cuda.synchronize()
        eager_step = step
        g = torch.cuda.CUDAGraph()
        optim_db.zero_grad(set_to_none=False)
        optim_q.zero_grad(set_to_none=False)
        with torch.cuda.graph(g):
            static_loss = eager_step()

        def step():
            g.replay()
            return static_loss
        step()
    parallel.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    parallel.barrier()
    ms = e0.elapsed_time(e1) / args.steps
    if world > 1:
        t = torch.tensor([ms], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ms = float(t)
    if rank == 0:
        nq = b * world
        print(json.dumps({
            "metric": "training step (fwd+bwd+Adam), MM query + DBVanilla2D tiles",
            "ms_per_step": round(ms, 3), "queries_per_s": round(nq / ms * 1e3, 2),
            "images_per_s": round(nq * (1 + args.ndb) / ms * 1e3, 1),
            "pairs_equiv_per_s": round(nq * args.ndb / ms * 1e3, 1),
            "n_gpus": world, "batch_per_gpu": b, "ndb": args.ndb, "tile": args.tile, "prec": args.prec,
            "loss": float(loss.detach()), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
