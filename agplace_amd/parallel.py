"""Data-parallel plumbing for the hot path (one process per GPU, torch.distributed over RCCL).

The reference is single-GPU (tools/options.py:295); DP is new capability:
  * embedding extraction shards samples across ranks, no data-path collective;
  * the eval descriptor database is all-gathered (xGMI) so every rank holds the full
    [N,256] matrix, then queries are sharded for the kNN -- zero further communication;
  * gradients (when a training path exists) are all-reduced in one flat bucket.
`backend='nccl'` is RCCL on ROCm; CPU tests use gloo.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n, rank, world):
    """Contiguous [lo, hi) shard of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(x, n_total=None, equal=False):
    """All-gather row blocks [n_r, d] (possibly ragged) into the full [sum n_r, d] on every rank.
    equal=True: every rank holds the same number of rows -> ONE collective into a preallocated tensor,
    no size exchange and no host synchronisation (the per-step exchange of bench.py)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    world = dist.get_world_size()
    if equal:
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        try:
            dist.all_gather_into_tensor(out, x)
        except (RuntimeError, NotImplementedError):      # backend without the flat variant
            bufs = list(out.chunk(world, 0))
            dist.all_gather(bufs, x)
        return out
    counts = torch.tensor([x.shape[0]], device=x.device, dtype=torch.int64)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    sizes = [int(c.item()) for c in all_counts]
    mx = max(sizes)
    pad = x
    if x.shape[0] < mx:
        pad = torch.cat([x, x.new_zeros((mx - x.shape[0],) + tuple(x.shape[1:]))], 0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


def allreduce_grads(params, average=True):
    """One flat-bucket all-reduce of every existing .grad (sum, then /world if average)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
