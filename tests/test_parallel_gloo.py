"""world_size-2 gloo tests of the data-parallel plumbing (the N>1 path of bench.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from agplace_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # ragged all-gather of descriptor rows: rank r owns its shard of 7 rows
    full = torch.arange(7 * 4, dtype=torch.float32).view(7, 4)
    lo, hi = parallel.shard_range(7, rank, world)
    got = parallel.all_gather_rows(full[lo:hi].clone())
    ok_gather = torch.equal(got, full)
    # equal-size fast path (one collective, no size exchange): the per-step exchange of bench.py
    blk = torch.full((3, 4), float(rank))
    eq = parallel.all_gather_rows(blk, equal=True)
    ok_gather = ok_gather and eq.shape == (3 * world, 4) and all(
        torch.equal(eq[3 * r_:3 * r_ + 3], torch.full((3, 4), float(r_))) for r_ in range(world))
    # flat-bucket gradient all-reduce (average)
    p1 = torch.nn.Parameter(torch.zeros(3))
    p2 = torch.nn.Parameter(torch.zeros(2, 2))
    p3 = torch.nn.Parameter(torch.zeros(1))          # no grad: must be skipped
    p1.grad = torch.full((3,), float(rank + 1))
    p2.grad = torch.full((2, 2), 10.0 * (rank + 1))
    parallel.allreduce_grads([p1, p2, p3], average=True)
    ok_red = torch.allclose(p1.grad, torch.full((3,), 1.5)) and torch.allclose(p2.grad, torch.full((2, 2), 15.0)) \
        and p3.grad is None
    parallel.barrier()
    q.put((rank, ok_gather, ok_red))
    dist.destroy_process_group()


def test_gloo_world2_allgather_and_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(g and r for _, g, r in res), res


# ---------------------------------------------------------------------------------------------------------------
# bucketed, overlapped gradient exchange + sharded retrieval / mining: index bookkeeping on gloo

class _CpuIndex:
    """Stand-in for the HIP IndexFlatL2 (no GPU in this test): exact fp64 brute force with the same interface."""

    def __init__(self, d, device="cpu", prec=None):
        self.d, self.xb = d, None

    def add(self, xb):
        xb = torch.as_tensor(xb, dtype=torch.float32)
        self.xb = xb if self.xb is None else torch.cat([self.xb, xb], 0)

    def search_device(self, xq, k):
        from oracle import knn
        D, I, _ = knn.knn_l2_fp64(torch.as_tensor(xq).numpy(), self.xb.numpy(), k)
        return torch.from_numpy(D.astype("float32")), torch.from_numpy(I.astype("int64"))


def _worker2(rank, world, port, q):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    ok = {}
    # ---- GradBuckets: several small buckets, gradients through autograd hooks AND through the side-effect path,
    # one parameter unused on rank 1 only, one unused everywhere; ready order differs from bucket order
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(4, 4)),
          torch.nn.Parameter(torch.randn(2)), torch.nn.Parameter(torch.randn(3))]
    gb = parallel.GradBuckets(ps, bucket_mb=64 / (1 << 20))         # 16 floats per bucket
    ok["nbuckets"] = len(gb.buckets) >= 2
    for step in range(2):
        gb.zero_grad()
        x = torch.full((3,), float(rank + 1 + step))
        loss = (ps[0] @ x).sum() * (rank + 1) + (ps[1] * (2.0 + rank)).sum()
        if rank == 0:
            loss = loss + (ps[2] * 3.0).sum()                       # ps[2] has a gradient on rank 0 only
        loss.backward()
        # side-effect gradient (the map backward's way): written in place, then notified
        from agplace_amd import train_graph
        # ... by TWO nodes that share the parameter (share_dbfe / stg2nlayers > 1): final only after the second
        train_graph.expect_grads([ps[3]])
        train_graph.expect_grads([ps[3]])
        train_graph._acc_grad(ps[3], torch.full((2,), 4.0 * (rank + 1)))
        train_graph.notify_grads_ready([ps[3]])
        ok[f"early_{step}"] = id(ps[3]) not in gb.seen
        train_graph._acc_grad(ps[3], torch.full((2,), 6.0 * (rank + 1)))
        train_graph.notify_grads_ready([ps[3]])
        ok[f"late_{step}"] = id(ps[3]) in gb.seen
        gb.finish()
        e0 = torch.stack([torch.full((3,), float(r + 1 + step)) * (r + 1) for r in range(world)]).mean(0).expand(5, 3)
        ok[f"g0_{step}"] = torch.allclose(ps[0].grad, e0)
        ok[f"g1_{step}"] = torch.allclose(ps[1].grad, torch.full((7,), sum(2.0 + r for r in range(world)) / world))
        ok[f"g2_{step}"] = torch.allclose(ps[2].grad, torch.full((4, 4), 3.0 / world))
        ok[f"g3_{step}"] = torch.allclose(ps[3].grad, torch.full((2,), sum(10.0 * (r + 1) for r in range(world)) / world))
        ok[f"g4_{step}"] = float(ps[4].grad.abs().max()) == 0.0     # unused everywhere: zeros, same layout on every rank
        ok[f"view_{step}"] = all(p.grad.data_ptr() == gb.flat.data_ptr() + 4 * gb.slice_of[id(p)][0] for p in ps)
    # after the first step the layout follows the observed ready order: ps[4] (silent everywhere) left the exchange, ps[2] (silent
    # on rank 1) sits in the last bucket, and a silent parameter no longer forces the buckets behind it
    ok["relearned"] = (not gb.reorder_pending) and gb.excluded == [4] and gb.bucket_of[id(ps[2])] == len(gb.buckets) - 1
    ok["stats"] = gb.stats["forced_last"] <= 1 and gb.stats["excluded_bytes"] == 12 and gb.stats["launched_before_finish"] >= len(gb.buckets) - 1
    # a gradient for a parameter outside the exchange must not vanish silently
    gb.zero_grad()
    try:
        ps[4].sum().backward()
        ok["excluded_raises"] = False
    except RuntimeError as e:
        ok["excluded_raises"] = "rebuild()" in str(e)
    gb.rebuild()
    # a second backward after a bucket's collective was launched must not be dropped silently
    gb.zero_grad()
    (ps[0].sum() + ps[1].sum() + ps[2].sum() + ps[3].sum() + ps[4].sum()).backward()       # every bucket launches
    try:
        (ps[0].sum() * 2.0).backward()
        ok["late_grad_raises"] = False
    except RuntimeError as e:
        ok["late_grad_raises"] = "accumulate=True" in str(e)
    gb.finish()
    gb.close()
    for p_ in ps:
        p_.grad = None
    # accumulate=True: two backward passes, ONE exchange at finish()
    ga = parallel.GradBuckets(ps, bucket_mb=64 / (1 << 20), accumulate=True)
    ga.zero_grad()
    (ps[0].sum() * float(rank + 1)).backward()
    (ps[0].sum() * 10.0 + ps[1].sum()).backward()
    ok["acc_nothing_launched"] = ga.next_launch == 0 and not ga.handles
    ga.finish()
    ok["acc0"] = torch.allclose(ps[0].grad, torch.full((5, 3), sum(r + 1 for r in range(world)) / world + 10.0))
    ok["acc1"] = torch.allclose(ps[1].grad, torch.ones(7))
    ga.close()
    # ---- allreduce_grads: rank-invariant layout although ps[2] has no gradient on rank 1
    for p_ in ps:
        p_.grad = None
    ps[0].grad = torch.full((5, 3), float(rank))
    if rank == 0:
        ps[2].grad = torch.full((4, 4), 8.0)
    parallel.allreduce_grads(ps, average=True)
    ok["ar0"] = torch.allclose(ps[0].grad, torch.full((5, 3), 0.5))
    ok["ar2"] = ps[2].grad is not None and torch.allclose(ps[2].grad, torch.full((4, 4), 4.0))
    ok["ar4"] = ps[4].grad is None
    # ---- BatchNorm buffers averaged before checkpointing
    bn = torch.nn.BatchNorm2d(3)
    bn.running_mean.fill_(float(rank)); bn.running_var.fill_(1.0 + rank); bn.num_batches_tracked.fill_(3 + rank)
    parallel.sync_bn_buffers([bn])
    ok["bn"] = torch.allclose(bn.running_mean, torch.full((3,), 0.5)) and torch.allclose(bn.running_var, torch.full((3,), 1.5)) \
        and int(bn.num_batches_tracked) == 4
    # ---- sharded retrieval: every rank passes ITS rows, gets the full result in dataset order
    from agplace_amd import retrieval, mining
    from oracle import knn, mining as omining
    rng = np.random.default_rng(3)
    db = rng.standard_normal((53, 32)).astype(np.float32)
    qs = rng.standard_normal((11, 32)).astype(np.float32)
    dlo, dhi = parallel.shard_range(53, rank, world)
    qlo, qhi = parallel.shard_range(11, rank, world)
    retrieval.IndexFlatL2 = _CpuIndex
    D, I = retrieval.distributed_search(qs[qlo:qhi], db[dlo:dhi], 5, device="cpu")
    Dr, Ir, _ = knn.knn_l2_fp64(qs, db, 5)
    ok["search"] = np.array_equal(I.numpy(), Ir) and np.allclose(D.numpy(), Dr, rtol=1e-5)

    class _DS:
        queries_num = 11
        def get_positives(self):
            return [np.array([int(Ir[i, 0]) if i % 2 == 0 else 52 - int(Ir[i, 0])]) for i in range(11)]
    import types
    a = types.SimpleNamespace(features_dim=32, recall_values=[1, 5])
    rec, _ = retrieval.distributed_compute_recall(a, qs[qlo:qhi], db[dlo:dhi], _DS(), device="cpu")
    retrieval_ref = retrieval.recall_from_predictions(a, Ir, _DS())[0]
    ok["recall"] = np.allclose(rec, retrieval_ref)
    # ---- sharded mining: the per-rank tables concatenate to the single-rank table
    hard = [rng.choice(53, size=3, replace=False) for _ in range(40)]
    soft = [np.unique(np.concatenate([h, rng.choice(53, size=4, replace=False)])) for h in hard]
    sq = rng.choice(40, size=9, replace=False)
    qf = rng.standard_normal((9, 32)).astype(np.float32)
    sdb = rng.choice(53, size=30, replace=False)
    mining.compute_triplets_partial = lambda qf_, db_, sq_, h_, s_, sd_, n_, dev_: torch.from_numpy(
        omining.compute_triplets_partial(qf_, db_, sq_, h_, s_, sd_, n_))
    got = mining.compute_triplets_partial_sharded(qf, db, sq, hard, soft, sdb, 4, device="cpu")
    ok["mining"] = np.array_equal(got.numpy(), omining.compute_triplets_partial(qf, db, sq, hard, soft, sdb, 4))
    # a failure on ONE rank's shard (too few negatives) raises on EVERY rank instead of hanging the gather
    def _one_rank_fails(qf_, db_, sq_, h_, s_, sd_, n_, dev_):
        if rank == 1:
            raise ValueError("fewer candidate negatives than negs_num_per_query for some query")
        return torch.zeros((len(sq_), 2 + n_), dtype=torch.int64)
    mining.compute_triplets_partial = _one_rank_fails
    try:
        mining.compute_triplets_partial_sharded(qf, db, sq, hard, soft, sdb, 4, device="cpu")
        ok["mining_error_everywhere"] = False
    except ValueError as e:
        ok["mining_error_everywhere"] = "rank 1" in str(e)
    parallel.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gloo_world2_buckets_sharded_search_and_mining():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker2, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok in res:
        assert all(ok.values()), (rank, {k: v for k, v in ok.items() if not v})


# ---------------------------------------------------------------------------------------------------------------
# the bench's own parameter list with the voxel side silent, as in its stand-in training step

def _bench_params():
    from agplace_amd.options import Options
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    opt = Options()
    torch.manual_seed(1)
    mq, mdb = MM(opt=opt), DBVanilla2D("db", opt.features_dim, opt=opt)
    named = [("db." + n, p) for n, p in mdb.named_parameters()] + [("q." + n, p) for n, p in mq.named_parameters()]
    named = [(n, p) for n, p in named if p.requires_grad]        # bench.py:train_measurement's list
    vox_side = ("q.vox_fe.", "q.vox_pool.", "q.stg2fuseblock.ffnsvox.", "q.stg2fuseblock.projsvoxfuse.", "q.stg2fuseblock.projsfusevox.",
                "q.stg2fuseblock.poolvox.")
    silent = [any(n.startswith(v) for v in vox_side) for n, _ in named]
    return named, silent


def _worker3(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    named, silent = _bench_params()
    params = [p for _, p in named]
    ok = {"no_fc": not any(".fc." in n and ".fe." in n for n, _ in named),            # torchvision's unused fc is frozen
          "has_silent": sum(silent) > 20}
    gb = parallel.GradBuckets(params, bucket_mb=16.0)
    steps = []
    for step in range(3):
        gb.zero_grad()
        # the backward of the stand-in step: gradients become final in reverse registration order, the voxel side stays silent
        for i in reversed(range(len(params))):
            if not silent[i]:
                params[i].grad.add_(float(rank + 1))
                gb.mark_ready([params[i]])
        launched_in_backward = gb.next_launch
        gb.finish()
        steps.append((launched_in_backward, dict(gb.stats)))
    first, later = steps[0][1], steps[1:]
    ok["first_step_held_back"] = first["forced_last"] >= 1                                   # the reverse-order layout: silent parameters in early buckets
    ok["forced_last_0"] = all(st["forced_last"] == 0 for _, st in later)
    # (30 of the 60 MB left the exchange with the silent voxel side: two 16 MB buckets remain, both launched during backward)
    ok["overlap"] = all(st["launched_before_finish"] == st["buckets"] and st["buckets"] >= 2 for _, st in later)
    ok["all_in_backward"] = all(n == st["buckets"] for n, st in later)
    ok["no_zeros_shipped"] = all(st["zero_bytes"] == 0 for _, st in later)
    silent_bytes = sum(p.numel() for p, s_ in zip(params, silent) if s_) * 4
    ok["excluded"] = later[0][1]["excluded_bytes"] == silent_bytes and later[0][1]["bytes"] == sum(p.numel() for p in params) * 4 - silent_bytes
    ok["averaged"] = all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p, s_ in zip(params, silent) if not s_) and \
        all(float(p.grad.abs().max()) == 0.0 for p, s_ in zip(params, silent) if s_)
    gb.close()
    parallel.barrier()
    q.put((rank, ok, steps[1][1]))
    dist.destroy_process_group()


def test_gloo_world2_bench_parameter_list_overlaps_with_silent_voxel_side():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker3, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, st in res:
        assert all(ok.values()), (rank, {k: v for k, v in ok.items() if not v}, st)


def _worker4(rank, world, port, q):
    """bench.py's N > 1 kNN leg (knn_measurement -> knn_distributed_leg -> retrieval.distributed_search) under gloo on CPU, with
    the exact fp64 stand-in for the HIP index: every rank owns db_rows / world database rows and its own queries; the all-gather
    is timed and reported; the gathered database equals the unsharded one; the rank's results equal a single-rank search."""
    import types
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    import bench
    from agplace_amd import retrieval
    from agplace_amd.options import Options
    retrieval.IndexFlatL2 = _CpuIndex
    args = types.SimpleNamespace(no_cpu_baseline=True, cpu_knn_queries=8)
    res = bench.knn_measurement(args, Options(), torch.device("cpu"), rank, world, parallel, retrieval, db_rows=1001, nq_rank=24, reps=2)
    d = res["distributed"]
    ok = (res["nq"] == 24 * world and d["gathered_database_equals_unsharded"] and d["allgather_ms"] > 0
          and d["allgather_GBps"] is not None and d["database_rows_per_rank"] in (500, 501)
          and d["allgather_bytes_received_per_rank"] == (1001 - d["database_rows_per_rank"]) * 256 * 4
          and d["strong_scaled_queries_per_rank"] == 12 and d["strong_scaled_queries_per_s"] > 0 and res["value"] > 0
          and res["scaling"].startswith("weak"))
    # the leg's own search equals the unsharded one on this rank's queries
    g = torch.Generator().manual_seed(1)
    db = torch.randn(1001, 256, generator=g)
    db = db / db.norm(dim=1, keepdim=True)
    gq = torch.Generator().manual_seed(1000 + rank)
    qq = torch.randn(24, 256, generator=gq)
    qq = qq / qq.norm(dim=1, keepdim=True)
    lo, hi = parallel.shard_range(1001, rank, world)
    D, I = retrieval.distributed_search(qq, db[lo:hi], 20, device="cpu")
    ref = _CpuIndex(256, device="cpu")
    ref.add(db)
    _, Iref = ref.search_device(qq, 20)
    ok = ok and torch.equal(I[24 * rank:24 * (rank + 1)], Iref)
    parallel.barrier()
    q.put((rank, bool(ok), {k: v for k, v in d.items()}))
    dist.destroy_process_group()


def test_gloo_world2_bench_knn_leg_shards_the_database_and_times_the_allgather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
