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
