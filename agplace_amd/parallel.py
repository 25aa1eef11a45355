"""Data-parallel plumbing for the hot path (one process per GPU, torch.distributed over RCCL).

The reference is single-GPU (tools/options.py:295); DP is new capability:
  * embedding extraction shards samples across ranks, no data-path collective;
  * the eval descriptor database is all-gathered (xGMI) so every rank holds the full
    [N,256] matrix, then queries are sharded for the kNN -- zero further communication;
  * gradients are all-reduced in buckets overlapped with backward (GradBuckets: one flat buffer the `.grad`
    tensors view, no copies) or, simplest, in one flat collective after backward (allreduce_grads).
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
            # AGP_DIST_BACKEND=gloo: smoke-test the multi-rank control flow where RCCL cannot run (several ranks on ONE GPU)
            backend = os.environ.get("AGP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


def all_gather_object(obj):
    """A small python object from every rank, in rank order (host-side bookkeeping only, never the data path)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, obj)
    return got


def allreduce_grads(params, average=True):
    """One flat all-reduce of the gradients AFTER backward (sum, then /world if average).  The flat layout covers EVERY
    parameter that requires grad -- a parameter without a gradient on this rank (an unused branch, e.g. `drop`, or
    torchvision's unused `fc`) contributes zeros -- so the collective has the same size on every rank whatever each
    rank's graph looked like.  Parameters that received no gradient on ANY rank keep `.grad = None`.
    For the overlapped form (buckets reduced while backward still runs, no copies) use GradBuckets."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    dev = params[0].device
    sizes = [p.numel() for p in params]
    flat = torch.zeros(sum(sizes) + len(params), dtype=torch.float32, device=dev)
    off = 0
    for i, (p, n) in enumerate(zip(params, sizes)):
        if p.grad is not None:
            flat[off:off + n].copy_(p.grad.reshape(-1))
            flat[-len(params) + i] = 1.0                    # "some rank has a gradient for this parameter"
        off += n
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat[:-len(params)] /= dist.get_world_size()
    has = flat[-len(params):].tolist()
    off = 0
    for p, n, h in zip(params, sizes, has):
        if h > 0:
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
        off += n


class GradBuckets:
    """Gradient storage and exchange of one data-parallel rank ("RCCL all-reduce of gradients, bucketed
    and overlapped with backward").

    * ONE flat fp32 buffer holds every gradient; each `param.grad` is a VIEW into it, so the collectives run on the
      gradients where they are: no concatenation before and no copy-back after the exchange.
    * The buffer is cut into buckets of about `bucket_mb`.  A bucket is all-reduced asynchronously (`async_op=True`: RCCL
      on its own stream, ordered behind the compute stream's work at the time of the call) as soon as every gradient in it
      is final, while backward continues with the earlier layers.  Buckets are launched strictly in index order, so every
      rank issues the same sequence of collectives whatever order its gradients became ready in.
    * Bucket ORDER.  The first step runs on the reverse parameter order (backward produces the last layers' gradients
      first).  At the end of that step the layout is rebuilt from the OBSERVED ready order (`reorder=True`): every rank
      reports at which position each parameter's gradient became final (or that none came), the ranks agree on the latest
      position per parameter (one small all_gather_object), parameters are laid out first-ready-first, parameters that were
      silent on some rank go into the last bucket(s) -- one silent parameter no longer holds back every later bucket -- and
      parameters that were silent on EVERY rank leave the exchange altogether (`drop_unused=True`: torchvision's unused
      `fc`, the voxel side of MM when a step feeds the voxel branch's outputs instead of `coords`): no collective ships
      their zeros.  A gradient that later arrives for such a parameter raises (call `rebuild()` when the graph changes).
    * The layout is rank-invariant by construction (it is computed from the all-gathered positions); a parameter that gets
      no gradient on one rank contributes zeros there, so the exchange has the same shape on every rank (no hang when one
      rank's graph skipped a branch).
    * Gradients arrive two ways: through autograd (leaf parameters of the vector path: a post-accumulate hook), and
      as side effects of the hand-orchestrated map backward (train_fns.TrunkFn etc. write conv / BatchNorm gradients
      with train_graph._acc_grad and then call train_graph.notify_grads_ready, which `mark_ready` is subscribed to).
    Use: `gb = GradBuckets(params)`; per step `gb.zero_grad()` (instead of optimizer.zero_grad), `loss.backward()`,
    `gb.finish()` (launches what is left, waits, averages), `optimizer.step()`.  `gb.stats` describes the last exchange."""

    def __init__(self, params, bucket_mb=16.0, average=True, accumulate=False, collective_on_single_rank=False, reorder=True,
                 drop_unused=True, names=None):
        """accumulate=True: several backward passes feed one exchange (gradient accumulation, losses backpropagated
        separately): nothing is launched before finish().  With accumulate=False a gradient that arrives for a bucket
        whose collective is already in flight raises (it would never be reduced: the ranks would diverge silently)."""
        self.accumulate = accumulate
        # issue the all-reduces even in a one-rank group (tests: the collective library's stream ordering on one GPU)
        self.single_rank_collective = collective_on_single_rank and dist.is_initialized()
        self.forced_last = 0
        self._warned = False
        params = list(params)
        self.params = [p for p in params if p.requires_grad]
        # optional names (error messages): parallel to `params`
        self.names = [n for n, p in zip(names, params) if p.requires_grad] if names is not None else None
        if not self.params:
            raise ValueError("GradBuckets: no parameter requires grad")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("GradBuckets: parameters must be contiguous fp32")
        self.average = average
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.bucket_mb = bucket_mb
        self.index_of = {id(p): i for i, p in enumerate(self.params)}
        self.reorder = bool(reorder)
        self.reorder_pending = bool(reorder and not accumulate)
        self.drop_unused = drop_unused
        self.stats = None
        self.flat = None
        self._layout(list(reversed(range(len(self.params)))), [])
        self._hooks = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        from . import train_graph
        train_graph.GRAD_READY_CALLBACKS.append(self._mark_ready_side)
        self._reset()

    # ------------------------------------------------------------------ layout
    def _layout(self, order, excluded):
        """Lay the parameters out in `order` (launch order: bucket 0 first) followed by `excluded` (outside every bucket);
        gradients already present are carried over."""
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        old = self.flat
        old_slices = getattr(self, "slice_of", None)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slice_of, off = {}, 0
        for i in list(order) + list(excluded):
            p = self.params[i]
            n = p.numel()
            self.slice_of[id(p)] = (off, n)
            if old is not None:
                o0, _ = old_slices[id(p)]
                flat[off:off + n].copy_(old[o0:o0 + n])
            off += n
        self.flat = flat
        for p in self.params:
            o, n = self.slice_of[id(p)]
            p.grad = self.flat[o:o + n].view_as(p)
        self.order, self.excluded = list(order), list(excluded)
        self.excluded_ids = {id(self.params[i]) for i in excluded}
        limit = max(1, int(self.bucket_mb * (1 << 20) / 4))
        self.buckets, self.bucket_of = [], {}
        cur, lo = [], 0
        for i in order:
            p = self.params[i]
            o, n = self.slice_of[id(p)]
            if not cur:
                lo = o
            cur.append(p)
            if o + n - lo >= limit:
                self.buckets.append((lo, o + n, cur))
                cur = []
        if cur:
            o, n = self.slice_of[id(cur[-1])]
            self.buckets.append((lo, o + n, cur))
        for b, (_, _, ps) in enumerate(self.buckets):
            for p in ps:
                self.bucket_of[id(p)] = b

    def rebuild(self):
        """Forget the learned order (the step's graph changed: other parameters are used now); the next step runs on the
        reverse parameter order again and the layout is re-learned at its end."""
        self._layout(list(reversed(range(len(self.params)))), [])
        self.reorder_pending = self.reorder and not self.accumulate
        self._reset()

    def _relearn(self):
        """End of the first step: parameters first-ready-first, silent ones last / out (see the class docstring)."""
        INF = 1 << 40
        pos = [INF] * len(self.params)
        for k, i in enumerate(self._ready_seq):
            pos[i] = k
        # A parameter nobody REPORTED may still have received a gradient (a producer that wrote into the view in place without
        # notifying): a silent parameter whose slice is non-zero on this rank counts as "ready last", never as unused -- moved out
        # of the exchange it would be folded into the excluded region on every later step, never reduced and never raise, and
        # the ranks would diverge silently.  One host read; this runs once, next to an all_gather_object.
        silent = [i for i in range(len(pos)) if pos[i] >= INF]
        if silent:
            nz = torch.stack([self.flat[self.slice_of[id(self.params[i])][0]:sum(self.slice_of[id(self.params[i])])].any()
                              for i in silent]).tolist()
            for i, has in zip(silent, nz):
                if has:
                    pos[i] = INF - 1
        if self.world > 1:
            everyone = [None] * self.world
            dist.all_gather_object(everyone, pos)
        else:
            everyone = [pos]
        latest = [max(r[i] for r in everyone) for i in range(len(pos))]
        earliest = [min(r[i] for r in everyone) for i in range(len(pos))]
        seen_all = sorted((i for i in range(len(pos)) if latest[i] < INF), key=lambda i: (latest[i], i))
        some = [i for i in range(len(pos)) if latest[i] >= INF and earliest[i] < INF]      # silent on some rank only
        none = [i for i in range(len(pos)) if earliest[i] >= INF]                            # silent everywhere
        if self.drop_unused:
            self._layout(seen_all + some, none)
        else:
            self._layout(seen_all + some + none, [])
        self.reorder_pending = False

    def _reset(self):
        self.pending = [len(ps) for _, _, ps in self.buckets]
        self.seen = set()
        self.side_seen = set()
        self.side_hooked = set()
        self._ready_seq = []
        self.next_launch = 0
        self.handles = []
        # streams on which a bucket's gradients were produced (backward runs a node on its forward's stream: the
        # database network's on a side stream in bench.py): the collective must be ordered behind ALL of them
        self.bstreams = [[] for _ in self.buckets]

    def close(self):
        from . import train_graph
        for h in self._hooks:
            h.remove()
        if self._mark_ready_side in train_graph.GRAD_READY_CALLBACKS:
            train_graph.GRAD_READY_CALLBACKS.remove(self._mark_ready_side)

    def zero_grad(self):
        """Zero every gradient in one fill and re-attach the views (an optimizer's set_to_none would drop them)."""
        self.flat.zero_()
        for p in self.params:
            o, n = self.slice_of[id(p)]
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                p.grad = self.flat[o:o + n].view_as(p)
        from . import train_graph
        train_graph.reset_expected()
        self._reset()

    def _hook(self, p):
        # A parameter of a hand-orchestrated node can ALSO be that node's autograd anchor (train_fns.anchor_of: an input of the
        # autograd.Function whose backward returns None for it): the engine still runs its AccumulateGrad node -- and this hook --
        # after the node's backward has reported the finished gradient through notify_grads_ready.  That second call carries nothing.
        # Only that ONE call is skipped: a later autograd gradient for the same parameter (shared between a hand-orchestrated node
        # and the autograd vector path) goes through mark_ready, whose "bucket already launched" check then raises instead of
        # letting it accumulate into a reduced bucket.
        if id(p) in self.side_seen and id(p) not in self.side_hooked:
            self.side_hooked.add(id(p))
            return
        self.mark_ready([p])

    def _mark_ready_side(self, params):
        self.side_seen.update(id(p) for p in params)
        self.mark_ready(params)

    def mark_ready(self, params):
        for p in params:
            k = id(p)
            if k in self.excluded_ids:
                raise RuntimeError("GradBuckets: a gradient arrived for a parameter that had none on any rank when the bucket layout "
                                   "was learned (it is outside the exchange): call GradBuckets.rebuild() when the step's graph changes, "
                                   "or build GradBuckets(..., drop_unused=False)")
            b = self.bucket_of.get(k)
            if b is None:
                continue
            if k in self.seen:
                if b < self.next_launch:
                    who = self.names[self.index_of[k]] if self.names else f"parameter {self.index_of[k]} of shape {tuple(p.shape)}"
                    raise RuntimeError(f"GradBuckets: a gradient arrived for {who}, whose bucket's all-reduce was already launched (a second "
                                       "backward before finish()?): build GradBuckets(..., accumulate=True) for gradient accumulation")
                continue
            o, n = self.slice_of[k]
            if p.grad is not None and p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                # a producer replaced the view (train_graph._acc_grad adopts tensors when .grad is None): fold it back
                self.flat[o:o + n].add_(p.grad.reshape(-1))
                p.grad = self.flat[o:o + n].view_as(p)
            self.seen.add(k)
            self._ready_seq.append(self.index_of[k])
            self.pending[b] -= 1
            if self.flat.is_cuda:
                st = torch.cuda.current_stream(self.flat.device)
                if all(st != s_ for s_ in self.bstreams[b]):
                    self.bstreams[b].append(st)
        self._launch_ready()

    def _launch_ready(self, force=False):
        if self.accumulate and not force:
            return
        while self.next_launch < len(self.buckets) and (force or self.pending[self.next_launch] == 0):
            lo, hi, _ = self.buckets[self.next_launch]
            if self.flat.is_cuda:
                cur = torch.cuda.current_stream(self.flat.device)
                for st in self.bstreams[self.next_launch]:
                    if st != cur:
                        cur.wait_stream(st)        # the gradient kernels enqueued there so far (they were, before mark_ready)
            if self.world > 1 or self.single_rank_collective:
                self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            self.next_launch += 1

    def finish(self):
        """After backward: reduce the buckets whose gradients never all arrived (unused parameters: zeros), wait for
        every collective, average; after the first step, re-learn the bucket order.  Fills `self.stats`."""
        for p in self.params:                   # gradients written by a path that did not notify
            k = id(p)
            o, n = self.slice_of[k]
            if k not in self.seen and p.grad is not None and p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                if k in self.excluded_ids:
                    raise RuntimeError("GradBuckets: a gradient was written for a parameter that is outside the exchange (it had none on "
                                       "any rank when the bucket layout was learned): call GradBuckets.rebuild() when the step's graph changes")
                self.flat[o:o + n].add_(p.grad.reshape(-1))
                p.grad = self.flat[o:o + n].view_as(p)
                if k in self.bucket_of:          # it HAS a gradient: ready (last) as far as the learned order is concerned
                    self.seen.add(k)
                    self._ready_seq.append(self.index_of[k])
        # buckets still waiting for a gradient that never came (an unused parameter): they, and every bucket behind
        # them in launch order, lost their overlap with backward
        launched_before = 0 if self.accumulate else self.next_launch
        self.forced_last = 0 if self.accumulate else sum(1 for b in range(self.next_launch, len(self.buckets)) if self.pending[b] > 0)
        held = len(self.buckets) - self.next_launch
        exch = sum(hi - lo for lo, hi, _ in self.buckets) * 4
        zero = sum(p.numel() for _, _, ps in self.buckets for p in ps if id(p) not in self.seen) * 4
        # bytes whose all-reduce was enqueued while backward was still producing gradients (buckets launch in index order)
        early = sum(hi - lo for lo, hi, _ in self.buckets[:launched_before]) * 4
        self.stats = {"buckets": len(self.buckets), "launched_before_finish": launched_before, "forced_last": self.forced_last,
                      "bytes": exch, "overlap_frac": round(early / exch, 4) if exch else 0.0, "zero_bytes": zero, "excluded_bytes": sum(self.params[i].numel() for i in self.excluded) * 4,
                      "order": "learned from the first step's ready order" if not self.reorder_pending and not self.accumulate else
                               ("reverse parameter order" if not self.accumulate else "one exchange at finish() (accumulate)")}
        if self.forced_last and not self._warned and not self.reorder_pending:
            self._warned = True
            import warnings
            warnings.warn(f"GradBuckets.finish(): {self.forced_last} bucket(s) hold parameters that reported no gradient this step; "
                          f"{held} of {len(self.buckets)} all-reduces could not overlap with backward")
        self._launch_ready(force=True)
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.average and self.world > 1:
            self.flat /= self.world
        if self.reorder_pending:
            self._relearn()


def sync_bn_buffers(modules, average=True):
    """BatchNorm running statistics are per-rank under data parallelism (the batch statistics of a rank's own samples,
    like the reference's single-GPU default without SyncBN, train.py:253-256): before evaluating or checkpointing,
    average running_mean / running_var over the ranks (num_batches_tracked: maximum) so every rank holds one model."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    bufs = []
    for m in modules:
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) and mod.running_mean is not None:
                bufs += [mod.running_mean, mod.running_var]
                if mod.num_batches_tracked is not None:
                    dist.all_reduce(mod.num_batches_tracked, op=dist.ReduceOp.MAX)
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off:off + n].view_as(b))
        off += n


_SYNC_BN_GROUP = None


def enable_sync_batchnorm(group=True):
    """Optional synchronised BatchNorm for data-parallel training (SURVEY.md 8e "optional"; reference
    model/sync_batchnorm/batchnorm.py:121-166, converted but never activated at train.py:253-256): every train-mode BatchNorm of
    the hand-written training graph -- ResNet trunks, stage-2 blocks, MinkowskiBatchNorm -- all-reduces its fp64
    (sum, sum of squares, count) vector in forward and its (sum g, sum g * zhat) vector in backward, so the statistics are those
    of the GLOBAL batch, as on one GPU; the affine parameters' gradients stay per rank and are averaged with all the others
    (GradBuckets / allreduce_grads).  One small collective per layer and direction.
    group=True: a process group of ALL ranks created for BatchNorm alone (every rank must make this call: dist.new_group is
    collective) -- GradBuckets launches its bucket all-reduces during the same backward, at rank-dependent moments when a rank has
    unused parameters, and two kinds of collectives may not interleave differently across ranks on ONE communicator; a
    ProcessGroup object: that group.  `enable_sync_batchnorm(None)` returns to per-rank statistics (the default, like the
    reference's run)."""
    global _SYNC_BN_GROUP
    from . import train_graph
    if group is True and dist.is_initialized() and dist.get_world_size() > 1:
        if _SYNC_BN_GROUP is None:
            _SYNC_BN_GROUP = dist.new_group()
        group = _SYNC_BN_GROUP
    train_graph.SYNC_BN = group


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
