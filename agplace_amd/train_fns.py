"""Autograd boundary between the feature-map graph and the vector graph in train mode.

The reference trains by autograd through cuDNN (train.py:337-341).  Here feature maps never become
torch tensors: they live as halo-padded split-bf16 planes (ops.SplitMap) in module workspaces, and
their forward/backward is orchestrated by hand (resnet.ResNet.forward_maps_train/backward_maps,
train_graph.ConvBNUnit).  Two torch.autograd.Functions splice that into the ordinary autograd graph
of the small [b,256] vectors:

    TrunkFn      image -> (mean(l1), mean(l2), mean(l3), GeM(l3))      (ImageFE + GeM + level pools)
    Stage2ImgFn  (l3, Linear(fusevec)) -> (mean(o), GeM(o)),  o = BasicBlock(l3 + vec[:, :, None, None])

Stage2ImgFn consumes the l3 MAP of TrunkFn.  Its gradient w.r.t. that map is handed over through a
`MapSink` side channel; autograd's topological order guarantees Stage2ImgFn.backward runs before
TrunkFn.backward because Stage2ImgFn takes one of TrunkFn's outputs as a (zero-gradient) token.
Parameter gradients of convs / BatchNorms / GeM exponents are accumulated straight into `.grad`.
"""
import torch

from . import ops, train_graph


class MapSink:
    """Carries the stage maps forward and map gradients backward between the two Functions."""

    def __init__(self):
        self.maps = None
        self.extra = {}      # map key -> SplitMap gradient contributed by a downstream consumer
        self.layers = {}     # ("s2", i) -> output map of stage-2 layer i (the input of layer i + 1 when opt.stg2nlayers > 1)

    def get(self, key):
        return self.layers[key] if isinstance(key, tuple) else self.maps[key]


def anchor_of(module, *extra):
    """The tensor that ties a hand-orchestrated node into autograd: ANY parameter of the node that requires grad (its
    outputs then require grad and backward runs).  The stem weight alone would silently cut the gradients of a trunk whose
    stem is frozen while deeper layers train."""
    first = None
    for p in list(module.parameters()) + [e for e in extra if e is not None]:
        first = p if first is None else first
        if p.requires_grad:
            return p
    return first


def _c(t):
    return None if t is None else t.contiguous().float()


class TrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, trunk, gem, sink, prec, want_means, slot=0):
        # `anchor` is any trunk parameter: it only makes the outputs require grad.  `slot`: see forward_maps_train.
        ctx.set_materialize_grads(False)
        maps = trunk.forward_maps_train(x, prec=prec, slot=slot)
        if ctx.needs_input_grad[0]:
            train_graph.expect_grads(list(trunk.parameters()) + [gem.p])
        p = gem.p.detach().float()
        means = []
        for m in maps[:-1]:
            if want_means:
                means.append(ops.pool_map(m, None, want_mean=True, want_gem=False)[0])
        mean_last, gemvec = ops.pool_map(maps[-1], p, want_mean=want_means, want_gem=True, eps=gem.eps)
        if want_means:
            means.append(mean_last)
        sink.maps = maps
        ctx.trunk, ctx.gem, ctx.sink, ctx.prec, ctx.want_means = trunk, gem, sink, prec, want_means
        ctx.maps, ctx.p, ctx.slot, ctx.gen = maps, p, slot, trunk.tape_generation(slot)
        ctx.save_for_backward(gemvec)
        return (*means, gemvec)

    @staticmethod
    def backward(ctx, *gs):
        trunk, gem, sink, maps = ctx.trunk, ctx.gem, ctx.sink, ctx.maps
        if trunk.tape_generation(ctx.slot) != ctx.gen:
            raise RuntimeError("agplace_amd: the trunk ran another training forward before this backward; "
                               "its workspace (activations) has been overwritten")
        (gemvec,) = ctx.saved_tensors
        S = len(maps)
        gmeans = list(gs[:S]) if ctx.want_means else [None] * S
        ggem = gs[-1]
        dev = gemvec.device
        gp = ops.new_gp(dev) if (ggem is not None and gem.p.requires_grad) else None
        grads = []
        for i, m in enumerate(maps):
            last = i == S - 1
            gm, gg, base = _c(gmeans[i]), _c(ggem) if last else None, sink.extra.pop(i, None)
            if gm is None and gg is None:
                grads.append(base)
                continue
            out = trunk._ws.map(f"t{ctx.slot or ''}.gpool{i}", m.n, m.h, m.w, m.c, 1, ctx.prec, dev)
            train_graph.pool_bwd(m, out, gmean=gm, ggem=gg, gem_y=gemvec if gg is not None else None,
                                 p=ctx.p if gg is not None else None, eps=gem.eps, base=base,
                                 gp=gp if gg is not None else None)
            grads.append(out)
        trunk.backward_maps(grads, slot=ctx.slot)
        if gp is not None:
            train_graph._acc_grad(gem.p, gp[:1])
        train_graph.notify_grads_ready(list(trunk.parameters()) + [gem.p])
        return (None,) * 8


class Stage2ImgFn(torch.autograd.Function):
    """One image-side layer of Stage2FuseBlockAdd (reference stage2fuse_blockadd.py:212-216): map `stage` of the sink
    (a trunk stage index, or ("s2", i-1): the previous layer's output) + vec broadcast -> BasicBlock -> (mean, GeM).
    The output map is left in the sink under `out_key`; `token` is any tensor produced by the node that owns the input
    map (it orders that node's backward after this one's)."""

    @staticmethod
    def forward(ctx, token, vec, block, gem, sink, stage, prec, want_mean, out_key=None):
        ctx.set_materialize_grads(False)
        l3 = sink.get(stage)
        dev = l3.hi.device
        y0 = block._ws.map("t.add", l3.n, l3.h, l3.w, l3.c, 1, prec, dev)
        ops.bcast_add(l3, vec.contiguous().float(), y0)
        o = block.forward_map_train(y0, prec=prec)
        if out_key is not None:
            sink.layers[out_key] = o
        if any(ctx.needs_input_grad):
            train_graph.expect_grads(list(block.parameters()) + [gem.p])
        p = gem.p.detach().float()
        mean, gemvec = ops.pool_map(o, p, want_mean=want_mean, want_gem=True, eps=gem.eps)
        ctx.block, ctx.gem, ctx.sink, ctx.stage, ctx.prec, ctx.want_mean = block, gem, sink, stage, prec, want_mean
        ctx.o, ctx.p, ctx.out_key = o, p, out_key
        ctx.save_for_backward(gemvec)
        return (mean, gemvec) if want_mean else gemvec

    @staticmethod
    def backward(ctx, *gs):
        block, gem, o = ctx.block, ctx.gem, ctx.o
        (gemvec,) = ctx.saved_tensors
        gmean, ggem = (_c(gs[0]), _c(gs[1])) if ctx.want_mean else (None, _c(gs[0]))
        base = ctx.sink.extra.pop(ctx.out_key, None) if ctx.out_key is not None else None    # from the next layer
        if gmean is None and ggem is None and base is None:
            train_graph.notify_grads_ready(list(block.parameters()) + [gem.p])
            return (None,) * 9
        dev = gemvec.device
        gp = ops.new_gp(dev) if (ggem is not None and gem.p.requires_grad) else None
        if gmean is None and ggem is None:
            go = base
        else:
            go = block._ws.map("t.go", o.n, o.h, o.w, o.c, 1, ctx.prec, dev)
            train_graph.pool_bwd(o, go, gmean=gmean, ggem=ggem, gem_y=gemvec if ggem is not None else None,
                                 p=ctx.p if ggem is not None else None, eps=gem.eps, base=base, gp=gp)
        gy0 = block.backward_map(go)
        if gp is not None:
            train_graph._acc_grad(gem.p, gp[:1])
        train_graph.notify_grads_ready(list(block.parameters()) + [gem.p])
        if ctx.stage in ctx.sink.extra:
            raise RuntimeError("agplace_amd: two consumers of one stage map are not supported")
        ctx.sink.extra[ctx.stage] = gy0
        gvec = None
        if ctx.needs_input_grad[1]:
            gvec = ops.pool_map(gy0, None, want_mean=True, want_gem=False)[0] * float(gy0.h * gy0.w)
        return None, gvec, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------ sparse-voxel branch
class VoxSink:
    """Carries the top voxel map forward and its stage-2 gradient backward (as MapSink does for l3)."""

    def __init__(self):
        self.top = None
        self.extra = None
        self.layers = {}         # i -> output of stage-2 voxel layer i (input of layer i + 1 when opt.stg2nlayers > 1)
        self.layer_extra = {}    # i -> gradient w.r.t. that output, left by layer i + 1


class VoxTrunkFn(torch.autograd.Function):
    """coords/features -> (avg(level 1), ..., avg(top), MinkGeM(top)); MinkFPN in train mode."""

    @staticmethod
    def forward(ctx, anchor, sp, vox_fe, vox_pool, sink):
        from .sparse import train as st
        from .sparse.modules import global_avg_pool
        ctx.set_materialize_grads(False)
        tr = getattr(vox_fe, "_train_obj", None)
        if tr is None:
            tr = vox_fe._train_obj = st.MinkFPNTrain(vox_fe)
        top, maps = tr.forward(sp)
        means = [global_avg_pool(m) for m in maps]
        gemv = vox_pool(top)
        sink.top = top
        ctx.tr, ctx.maps, ctx.sink, ctx.pool = tr, maps, sink, vox_pool
        ctx.top_idx = len(maps) - 1 - vox_fe.num_top_down        # `top` is the last tensor of the top-down pass (minkfpn.py:116-118)
        ctx.p = vox_pool.p.detach().float()
        ctx.save_for_backward(gemv)
        return (*means, gemv)

    @staticmethod
    def backward(ctx, *gs):
        from .sparse import train as st
        (gemv,) = ctx.saved_tensors
        maps, pool = ctx.maps, ctx.pool
        ggem = _c(gs[-1])
        gp = ops.new_gp(gemv.device) if (ggem is not None and pool.p.requires_grad) else None
        gmaps = []
        for i, m in enumerate(maps):
            last = i == ctx.top_idx
            gm, gg = _c(gs[i]), ggem if last else None
            base = ctx.sink.extra if last else None
            if gm is None and gg is None:
                gmaps.append(base)
                continue
            gmaps.append(st.seg_pool_bwd(m, gmean=gm, ggem=gg, gem_y=gemv if gg is not None else None,
                                         p=ctx.p if gg is not None else None, eps=pool.eps, base=base,
                                         gp=gp if gg is not None else None))
        ctx.sink.extra = None
        ctx.tr.backward(gmaps)
        if gp is not None:
            train_graph._acc_grad(pool.p, gp[:1])
        train_graph.notify_grads_ready(list(ctx.tr.net.parameters()) + [pool.p])
        return (None,) * 5


class Stage2VoxFn(torch.autograd.Function):
    """(top voxel map, Linear(fusevec)) -> (avg(proj(o)), MinkGeM(o)),  o = ECABasicBlock(top + vec[batch])
    (reference stage2fuse_blockadd.py:194-211, sparse side)."""

    @staticmethod
    def forward(ctx, token, vec, block, gem, proj, sink, layer=0):
        from .sparse import train as st
        from .sparse.modules import global_avg_pool, seg_affine
        ctx.set_materialize_grads(False)
        top = sink.top if layer == 0 else sink.layers[layer - 1]
        y0 = seg_affine(top, add=vec.contiguous().float())
        bt = getattr(block, "_train_obj", None)
        if bt is None:
            bt = block._train_obj = st.ECABlockTrain(block)
        o = bt.forward(y0)
        sink.layers[layer] = o
        gemv = gem(o)
        unit = None
        if proj is not None:
            unit = st.SparseConvUnit(proj, None)
            pf = unit.forward(o)
        else:
            pf = o
        mean = global_avg_pool(pf)
        if any(ctx.needs_input_grad):
            train_graph.expect_grads(list(bt.blk.parameters()) + [gem.p] + (list(proj.parameters()) if proj is not None else []))
        ctx.bt, ctx.unit, ctx.o, ctx.pf, ctx.gem, ctx.sink, ctx.layer = bt, unit, o, pf, gem, sink, layer
        ctx.p = gem.p.detach().float()
        ctx.save_for_backward(gemv)
        return mean, gemv

    @staticmethod
    def backward(ctx, gmean, ggem):
        from .sparse import train as st
        (gemv,) = ctx.saved_tensors
        gmean, ggem = _c(gmean), _c(ggem)
        params = list(ctx.bt.blk.parameters()) + [ctx.gem.p] + (list(ctx.unit.conv.parameters()) if ctx.unit is not None else [])
        base = ctx.sink.layer_extra.pop(ctx.layer, None)             # from the next layer
        if gmean is None and ggem is None and base is None:
            train_graph.notify_grads_ready(params)
            return (None,) * 7
        gem, o = ctx.gem, ctx.o
        gp = ops.new_gp(gemv.device) if (ggem is not None and gem.p.requires_grad) else None
        if ctx.unit is not None and gmean is not None:
            b2 = ctx.unit.backward(st.seg_pool_bwd(ctx.pf, gmean=gmean))
            base = b2 if base is None else st._add(base, b2)
            gmean = None
        if gmean is None and ggem is None:
            go = base
        else:
            go = st.seg_pool_bwd(o, gmean=gmean, ggem=ggem, gem_y=gemv if ggem is not None else None,
                                 p=ctx.p if ggem is not None else None, eps=gem.eps, base=base, gp=gp)
        gy0 = ctx.bt.backward(go)
        if gp is not None:
            train_graph._acc_grad(gem.p, gp[:1])
        train_graph.notify_grads_ready(params)
        if ctx.layer == 0:
            ctx.sink.extra = gy0
        else:
            ctx.sink.layer_extra[ctx.layer - 1] = gy0
        gvec = st.seg_sum(gy0) if ctx.needs_input_grad[1] else None
        return None, gvec, None, None, None, None, None


def reference_optimizers(model_db, model_q, lr=1e-5, lrpc=1e-4, lrdb=1e-5, **adam_kw):
    """The reference's optimiser layout (train.py:165-190, 213-214; learning-rate defaults tools/options.py:56-58): Adam over ONE
    parameter group for the database model at `lrdb`, and a second Adam over the query model's SIXTEEN groups -- image_fe,
    image_pool, vox_fe, vox_pool, fuseblocktoshallow, stg2fuseblock, stg2fusefc and the nine mixing weights -- the voxel side
    (vox_fe, vox_pool, vox_weight) at `lrpc`, everything else at `lr`.  Returns (optimizer, optimizerq), stepped and zeroed in
    that order like train.py:337-341.  adam_kw: e.g. fused=True, capturable=True."""
    params_db = [{'params': list(model_db.parameters()), 'lr': lrdb}]
    q = model_q
    params_q = [
        {'params': list(q.image_fe.parameters()), 'lr': lr},
        {'params': list(q.image_pool.parameters()), 'lr': lr},
        {'params': list(q.vox_fe.parameters()), 'lr': lrpc},
        {'params': list(q.vox_pool.parameters()), 'lr': lrpc},
        {'params': list(q.fuseblocktoshallow.parameters()), 'lr': lr},
        {'params': list(q.stg2fuseblock.parameters()), 'lr': lr},
        {'params': list(q.stg2fusefc.parameters()), 'lr': lr},
        {'params': [q.image_weight], 'lr': lr},
        {'params': [q.vox_weight], 'lr': lrpc},
        {'params': [q.shallow_weight], 'lr': lr},
        {'params': [q.imageorg_weight], 'lr': lr},
        {'params': [q.voxorg_weight], 'lr': lr},
        {'params': [q.shalloworg_weight], 'lr': lr},
        {'params': [q.stg2image_weight], 'lr': lr},
        {'params': [q.stg2vox_weight], 'lr': lr},
        {'params': [q.stg2fuse_weight], 'lr': lr},
    ]
    return torch.optim.Adam(params_db, **adam_kw), torch.optim.Adam(params_q, **adam_kw)
