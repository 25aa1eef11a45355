"""ResNet18/34/50 trunk with torchvision's parameter names, executed by the HIP conv kernels.

The reference builds `torchvision.models.resnet{18,34,50}(pretrained=True)` and drives
conv1/bn1/relu/maxpool/layer1..3 by hand (network_mm/image_fe.py:19-30,97-113;
network/image_fe.py:47-59).  torchvision is not part of this build: the modules below are
parameter containers whose state_dict keys equal torchvision's (conv1.weight, bn1.*,
layer{L}.{i}.conv{j}.weight, layer{L}.{i}.downsample.{0,1}.*, fc.*), so reference checkpoints
load unchanged.  `pretrained=True` needs a download that is unavailable here: parameters are
randomly initialised with torchvision's scheme (Kaiming-normal fan_out convs, BN 1/0).

Forward = agp_conv2d_fwd per conv with BatchNorm(eval) folded into the epilogue scale/shift,
residual add + ReLU fused, activations as halo-padded NHWC split-bf16 planes (ops.SplitMap).
"""
import math

import os

import torch
import torch.nn as nn

from . import ops, train_graph

ARCH = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
}
PLANES = [64, 128, 256, 512]


def _conv(cin, cout, k, stride, pad):
    m = nn.Conv2d(cin, cout, k, stride, pad, bias=False)
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return m


class _Block(nn.Module):
    def __init__(self, kind, inplanes, planes, stride):
        super().__init__()
        self.kind, self.stride = kind, stride
        exp = 1 if kind == "basic" else 4
        if kind == "basic":
            self.conv1 = _conv(inplanes, planes, 3, stride, 1)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = _conv(planes, planes, 3, 1, 1)
            self.bn2 = nn.BatchNorm2d(planes)
        else:
            self.conv1 = _conv(inplanes, planes, 1, 1, 0)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = _conv(planes, planes, 3, stride, 1)
            self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = _conv(planes, planes * exp, 1, 1, 0)
            self.bn3 = nn.BatchNorm2d(planes * exp)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or inplanes != planes * exp:
            self.downsample = nn.Sequential(_conv(inplanes, planes * exp, 1, stride, 0),
                                            nn.BatchNorm2d(planes * exp))
        self.out_planes = planes * exp

    def convs(self):
        """[(conv, bn)] in execution order, then the downsample pair (or None)."""
        seq = [(self.conv1, self.bn1), (self.conv2, self.bn2)]
        if self.kind == "bottleneck":
            seq.append((self.conv3, self.bn3))
        ds = (self.downsample[0], self.downsample[1]) if self.downsample is not None else None
        return seq, ds


FUSE_STEM_POOL = True      # inference, fp16 maps: agp_stem_pool_fwd instead of conv + max-pool


class ResNet(nn.Module):
    """Parameter-compatible stand-in for torchvision.models.resnetXX, `nstages` stages kept."""

    def __init__(self, fe_type, nstages=4):
        super().__init__()
        kind, layers = ARCH[fe_type]
        self.fe_type, self.kind, self.nstages = fe_type, kind, nstages
        self.conv1 = _conv(3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        exp = 1 if kind == "basic" else 4
        for li in range(4):
            if li < nstages:
                blocks = []
                for bi in range(layers[li]):
                    stride = 2 if (li > 0 and bi == 0) else 1
                    blk = _Block(kind, inplanes, PLANES[li], stride)
                    inplanes = blk.out_planes
                    blocks.append(blk)
                setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
            else:
                setattr(self, f"layer{li + 1}", nn.Identity())   # image_fe.py:23-26
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * exp, 1000)    # registered-but-unused in the reference too
        # (never part of a forward: it cannot receive a gradient.  Frozen, so that a data-parallel gradient exchange does not carry
        # 513 k zeros per trunk and optimizers built from .parameters() skip it; the state_dict keys stay)
        self.fc.requires_grad_(False)
        self._prep = None
        self._prep_key = None
        self._ws = ops.Workspace()

    # ------------------------------------------------------------------ weights
    def _version_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + \
            tuple((b.data_ptr(), b._version) for b in self.buffers())

    def _prepared(self):
        key = self._version_key()
        if self._prep is not None and key == self._prep_key:
            return self._prep
        prep = {}

        def fold(conv, bn, stem=False):
            s, t = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
            return ops.ConvWeights(conv.weight, s, t, conv.stride[0], conv.padding[0], stem=stem)

        prep["stem"] = fold(self.conv1, self.bn1, stem=True)
        for li in range(self.nstages):
            for bi, blk in enumerate(getattr(self, f"layer{li + 1}")):
                seq, ds = blk.convs()
                prep[(li, bi)] = ([fold(c, b) for c, b in seq], fold(*ds) if ds else None)
        self._prep, self._prep_key = prep, key
        return prep

    @staticmethod
    def _input_geometry(x):
        """(n, h, w, device) of a stem input: fp32 [n,3,h,w], uint8 camera tiles [n,ncam,h,w,3] or a packed SplitMap."""
        if isinstance(x, ops.SplitMap):
            return x.n, x.h, x.w, x.hi.device
        if x.dtype == torch.uint8:
            n, ncam, h, w, _ = x.shape
            return n, h, ncam * w, x.device
        n, _, h, w = x.shape
        return n, h, w, x.device

    def _stem_input(self, x, tag, prec, lo=0, hi=None, h16=False):
        """Images [lo, hi) of a stem input -- fp32 [n,3,h,w] image batch, uint8 [n,ncam,h,w,3] camera tiles (device-side
        input pipeline, ops.pack_cameras_u8) or an already packed NHWC4 halo-3 SplitMap -- as the stem's input map (a
        slice of the full-batch workspace map)."""
        n, h, w, dev = self._input_geometry(x)
        hi = n if hi is None else hi
        if isinstance(x, ops.SplitMap):
            if x.c != 4 or x.pad != 3 or x.prec != (3 if prec == 3 else 2):
                raise ValueError("stem input map must be NHWC4 with halo 3 in the conv's storage format")
            return ops.slice_map(x, lo, hi)
        # h16 (training, fp32 images): the fp16 operand plane of the stem's one-pass weight gradient, written by the same pass
        want16 = h16 and prec == 3 and x.dtype != torch.uint8
        xin = ops.slice_map(self._ws.map(tag, n, h, w, 4, 3, prec, dev, h16=want16), lo, hi)
        if not want16 and xin.h16 is not None:
            xin = ops.SplitMap(xin.hi, xin.lo, xin.n, xin.h, xin.w, xin.c, xin.pad)
        if x.dtype == torch.uint8:
            ops.pack_cameras_u8(x[lo:hi], prec, out=xin)
        else:
            xin = ops.pack_f32(x[lo:hi], 4, 3, prec, out=xin)      # (a plane-less view of xin when the fp16 pack was refused)
        return xin

    # ------------------------------------------------------------------ forward
    def forward_maps(self, x, prec=3, level_means=None, final_pool=None):
        """x fp32 [n,3,h,w] on the GPU -> list of SplitMap stage outputs [l1, l2, l3(, l4)].

        Eval-mode BatchNorm (running statistics).  The returned maps alias this module's
        workspace and are overwritten by its next forward.

        level_means: optional list; the channel means of every stage output but the last are appended to it
        (pooled in the epilogue of the conv that produces the stage output: ops.PoolReq).
        final_pool: optional ops.PoolReq for the LAST stage output (GeM and / or mean), filled the same way."""
        return forward_maps_multi([self], [x], prec=prec, level_means=[level_means], final_pools=[final_pool])[0]

    # ------------------------------------------------------- training forward / backward
    def _unit(self, name, conv, bn, stem=False, pre="t."):
        u = self._units.get(pre + name)
        if u is None:
            u = train_graph.ConvBNUnit(conv, bn, pre + name, self._ws, stem=stem)
            self._units[pre + name] = u
        return u

    def forward_maps_train(self, x, prec=3, slot=0):
        """Train-mode forward (batch-statistics BatchNorm, running stats updated); records what
        `backward_maps` needs.  Returns the stage outputs like forward_maps.

        slot: one trunk applied to several inputs inside one step (DBVanilla2D under opt.share_dbfe, reference
        models_baseline/dbvanilla2d.py:69-72) keeps one set of activations and one tape per slot; the parameter
        gradients of all slots accumulate."""
        if not hasattr(self, "_units"):
            self._units, self._tapes, self._tape_gens = {}, {}, {}
        self._tape_gens[slot] = self._tape_gens.get(slot, 0) + 1
        pre = "t." if slot == 0 else f"t{slot}."
        n, h, w, dev = self._input_geometry(x)
        ws = self._ws
        stem = self._unit("stem", self.conv1, self.bn1, stem=True, pre=pre)
        xin = self._stem_input(x, pre + "in", prec, h16=stem.wgrad_f16_ok(prec))
        c1 = self.conv1
        hs = ops.conv_out_size(xin.h, c1.kernel_size[0], c1.stride[0], c1.padding[0])
        wss = ops.conv_out_size(xin.w, c1.kernel_size[0], c1.stride[0], c1.padding[0])
        h2, w2 = ops.conv_out_size(hs, 3, 2, 1), ops.conv_out_size(wss, 3, 2, 1)
        pooled = ws.map(pre + "pool", n, h2, w2, 64, 1, prec, dev)
        argmax = ws.tensor(pre + "pool.argmax", (n, h2, w2, 64), torch.uint8, dev)
        # the blocks in order, so that every unit knows its consumer: a map read by a conv whose weight gradient runs as one fp16
        # product (train_graph.ConvBNUnit.wgrad_f16_ok) also keeps an fp16 operand plane, written by the pass that writes the map
        blocks = []
        for li in range(self.nstages):
            for bi, blk in enumerate(getattr(self, f"layer{li + 1}")):
                seq, ds = blk.convs()
                units = [self._unit(f"l{li}.{bi}.c{ci}", c, b, pre=pre) for ci, (c, b) in enumerate(seq)]
                ud = self._unit(f"l{li}.{bi}.ds", ds[0], ds[1], pre=pre) if ds else None
                blocks.append((li, units, ud))
        # BatchNorm apply + ReLU + max-pool in one pass over the stem conv's output; the full-size activation is not stored
        stem.forward(xin, relu=True, prec=prec, pool=(pooled, argmax), out_h16=blocks[0][1][0].wgrad_f16_ok(prec))
        s = pooled
        cur, outs, tape = pooled, [], []
        for k, (li, units, ud) in enumerate(blocks):
            idt = ud.forward(cur, relu=False, prec=prec) if ud else cur
            t = cur
            for ci, u in enumerate(units):
                last = ci == len(units) - 1
                # the consumer of this unit's output: the block's next conv, or the next block's first (the trunk's last
                # output feeds whatever follows the trunk: `last_out_h16`)
                if not last:
                    want = units[ci + 1].wgrad_f16_ok(prec)
                elif k + 1 < len(blocks):
                    want = blocks[k + 1][1][0].wgrad_f16_ok(prec)
                else:
                    want = bool(getattr(self, "last_out_h16", False))
                # (inside a block the next conv is the map's only reader)
                only16 = (not last) and units[ci + 1].reads_f16_plane_only(prec)
                t = u.forward(t, residual=idt if last else None, relu=True, prec=prec, out_h16=want, out_f16_only=only16)
            tape.append((units, ud))
            cur = t
            if k + 1 == len(blocks) or blocks[k + 1][0] != li:
                outs.append(cur)
                tape.append(("stage_end", li))
        self._tapes[slot] = (tape, s, argmax, stem, prec, pre)
        return outs

    def tape_generation(self, slot=0):
        return getattr(self, "_tape_gens", {}).get(slot, 0)

    def backward_maps(self, stage_grads, slot=0):
        """stage_grads[i]: SplitMap gradient w.r.t. stage output i (or None) of the slot's last train-mode forward.
        Accumulates `.grad` of every conv / BatchNorm parameter of the trunk."""
        tape, s, argmax, stem, prec, pre = self._tapes[slot]
        ws, dev = self._ws, s.hi.device
        g, part = None, None              # part: the channel sums of the NEXT unit's BatchNorm backward, reduced by the conv that made g
        rev = list(reversed(tape))
        for pos, item in enumerate(rev):
            if item[0] == "stage_end":
                sg = stage_grads[item[1]]
                if sg is not None:
                    if g is None:
                        g = sg
                    else:
                        acc = ws.map(f"{pre}gstage{item[1]}", g.n, g.h, g.w, g.c, 1, prec, dev)
                        g = train_graph.map_add(g, sg, acc)
                continue
            units, ud = item
            if g is None:
                continue                     # no gradient reaches this block
            # The unit whose output is this block's input, if this block's input gradient is that unit's WHOLE output gradient
            # (no stage gradient is added in between): its BatchNorm-backward sums ride in this block's last data-gradient conv.
            prev_last = None
            if pos + 1 < len(rev):
                nxt = rev[pos + 1]
                if nxt[0] == "stage_end":
                    if stage_grads[nxt[1]] is None and pos + 2 < len(rev) and rev[pos + 2][0] != "stage_end":
                        prev_last = rev[pos + 2][0][-1]
                else:
                    prev_last = nxt[0][-1]
            gh, gres, (_, pnext) = units[-1].backward(g, partial=part, stats_for=units[-2] if len(units) > 1 else None)
            gx2 = None
            added = False
            for ci in range(len(units) - 2, -1, -1):
                if ci == 0:
                    gx2 = ud.backward(gres)[0] if ud is not None else gres
                    gh, _, (added, pnext) = units[0].backward(gh, partial=pnext, add=gx2, stats_for=prev_last)
                else:
                    gh, _, (_, pnext) = units[ci].backward(gh, partial=pnext, stats_for=units[ci - 1])
            if len(units) == 1:
                gx2 = ud.backward(gres)[0] if ud is not None else gres
                pnext = None
            if added:
                g, part = gh, pnext
            else:
                acc = ws.map(units[0].tag + ".gsum", gh.n, gh.h, gh.w, gh.c, 1, prec, dev)
                g, part = train_graph.map_add(gh, gx2, acc), None
        if g is not None:
            stem.backward(g, need_gx=False, pool_argmax=argmax, pooled=s)      # max-pool backward inside the stem's BatchNorm backward


# The stem reading the network's input itself (agp_stem_pool_raw_fwd) instead of a packed NHWC4 copy of it: bit-identical, no
# packing pass (99 us and 318 MB of HBM traffic per 64 panoramas).  Default ("auto"): wherever the walking stem kernel can
# fetch the input by LDS-DMA (ops.stem_walk_reads: aligned fp32 images; 182 us against 100 + 154 us per 64 panoramas).
# STEM_READS_INPUT = "1" also sends uint8 tiles and unaligned images to the per-block raw kernel (round 2: slower than packing);
# "0": always pack.
STEM_READS_INPUT = "auto"
# fp16 maps (precision modes 2 / 4) saturate at +-65504.  The first inference forward after a weight (re)load counts the
# saturated elements of the stage outputs (one small reduction + one host read, never inside a stream capture) and warns:
# such a checkpoint needs Options.mfma_precision = 3 (split-bf16 maps, fp32 range).  SATURATION_CHECK = False turns it off.
SATURATION_CHECK = True
STAGE1_CHUNK = 1 << 30   # images of the first (largest) trunk per pass over stem + stage 1 (off: see forward_maps_multi)


def forward_maps_multi(nets, xs, prec=3, level_means=None, final_pools=None):
    """Several ResNet trunks of ONE architecture (e.g. the query network's and the database network's, reference
    network_mm/image_fe.py:97-113 and network/image_fe.py:112-128) advanced in lock-step: nets[i] on xs[i]
    (different batch sizes, image sizes and weights).  Every 3x3 stride-1 conv of a layer is issued for all trunks as
    ONE grouped launch (ops.conv2d_grouped -> agp_conv2d_fwd_grouped), so the small trunk's convs ride in the big
    one's grids instead of being launches of their own; results are bit-identical to separate forwards.

    Stem + stage 1 CAN run in chunks of at most STAGE1_CHUNK images of nets[0] (the other trunks are cut into the same
    number of chunks), which keeps a conv's input + residual + output (154 MB each for 64 panoramas) inside the 256 MB
    Infinity Cache.  Measured on the bench step (round 2): 8 x 75 us against 4 x 150 us for the whole batch -- no gain,
    the stage-1 convs are not bound by where their maps live -- so it is off by default; the mechanism (bit-identical
    for every chunking, tests/test_gpu_models.py) stays for batches whose maps do not fit the device at once.
    Returns [maps of nets[0], maps of nets[1], ...]; level_means / final_pools: per net None or a list / an ops.PoolReq
    (see ResNet.forward_maps): the pooling of a stage output rides in the launch of the conv that produces it."""
    R = len(nets)
    level_means = level_means or [None] * R
    final_pools = final_pools or [None] * R
    stage_pool = [{} for _ in range(R)]          # per net: stage index -> PoolReq of that stage's output
    a = nets[0]
    for b in nets[1:]:
        if (b.fe_type, b.nstages) != (a.fe_type, a.nstages):
            raise ValueError("forward_maps_multi: the trunks must share one architecture")
    preps = [net._prepared() for net in nets]
    geo = [net._input_geometry(x) for net, x in zip(nets, xs)]
    devs = [g[3] for g in geo]
    nchunks = max(1, -(-geo[0][0] // STAGE1_CHUNK))
    for r in range(R):
        for li in range(a.nstages):
            last_stage = li == a.nstages - 1
            if last_stage and final_pools[r] is not None:
                stage_pool[r][li] = final_pools[r]
            elif not last_stage and level_means[r] is not None:
                stage_pool[r][li] = ops.PoolReq(want_mean=True, want_gem=False)

    def run_stage(li, cur, lo, hi):
        """Stage li on images [lo[r], hi[r]) of every trunk r that has images in this pass; `cur` holds the input
        views.  Maps come from the full-batch workspace buffers and are sliced, so passes write disjoint images."""
        act = [r for r in range(R) if hi[r] > lo[r]]
        nblocks = len(getattr(a, f"layer{li + 1}"))
        for bi in range(nblocks):
            blks = {r: getattr(nets[r], f"layer{li + 1}")[bi] for r in act}
            cws = {r: preps[r][(li, bi)] for r in act}
            idt = dict(cur)

            def view(r, tag, h_, w_, c_):
                return ops.slice_map(nets[r]._ws.map(tag, geo[r][0], h_, w_, c_, 1, prec, devs[r]), lo[r], hi[r])
            ds_jobs = []
            if cws[act[0]][1] is not None:
                for r in act:
                    dsw = cws[r][1]
                    ho = ops.conv_out_size(cur[r].h, 3, blks[r].stride, 1)
                    wo = ops.conv_out_size(cur[r].w, 3, blks[r].stride, 1)
                    idt[r] = view(r, f"ds{li}.{bi}", ho, wo, dsw.cout)
                    ds_jobs.append((cur[r], dsw, idt[r], None, False))
            t = dict(cur)
            nconv = len(cws[act[0]][0])
            if a.kind == "basic" and cws[act[0]][1] is None and len(act) <= 4 and \
                    all(ops.bblock64_ok(cur[r], cws[r][0][0], cws[r][0][1], prec) for r in act):
                # a whole 64-channel BasicBlock as ONE kernel (csrc/fblock64.hip): the intermediate map stays in LDS, the residual
                # comes from the staged input rows.  The default 16x16x32 MFMA form differs from the two conv launches below by the
                # fp32 rounding of another accumulation order (same products; the parity tests hold both to the oracle);
                # ops.bblock64_grouped(jobs, exact=True) is the 32x32x16 form, bit-identical to them
                jobs, pools = [], {}
                for r in act:
                    pool = stage_pool[r].get(li) if (bi == nblocks - 1 and (li > 0 or nchunks == 1)) else None
                    pools[r] = pool
                    out_ = view(r, f"c{li}.{bi}.1", cur[r].h, cur[r].w, 64)
                    jobs.append((cur[r], cws[r][0][0], cws[r][0][1], out_, pool if (pool is not None and not pool.want_gem) else None))
                outs_ = ops.bblock64_grouped(jobs)
                for r, o in zip(act, outs_):
                    if pools[r] is not None and pools[r].want_gem:       # (GeM of a 64-channel stage output: pooled from the stored map)
                        pools[r].fused = False
                        pools[r].finish(o)
                cur = {r: o for r, o in zip(act, outs_)}
                continue
            for ci in range(nconv):
                last = ci == nconv - 1
                jobs = []
                for r in act:
                    cw = cws[r][0][ci]
                    oh = ops.conv_out_size(t[r].h, cw.kh, cw.stride, cw.pad)
                    ow = ops.conv_out_size(t[r].w, cw.kw, cw.stride, cw.pad)
                    # the conv that writes a stage output also pools it (whole-batch passes only: a chunk's partial sums
                    # would cover a slice of the images)
                    pool = stage_pool[r].get(li) if (last and bi == nblocks - 1 and (li > 0 or nchunks == 1)) else None
                    jobs.append((t[r], cw, view(r, f"c{li}.{bi}.{ci}", oh, ow, cw.cout), idt[r] if last else None, True, pool))
                if ci == 0 and ds_jobs:
                    # the downsample reads the block's input like conv1 and is independent of it: one grouped launch
                    # (agp_conv2d_fwd_grouped: the latency-bound 1x1 hides between the tiles of the stride-2 3x3)
                    if len(jobs) + len(ds_jobs) <= 4 and not (cws[act[0]][0][0].kh == 3 and cws[act[0]][0][0].stride == 1):
                        outs_ = ops.conv2d_grouped(jobs + ds_jobs, prec)[:len(jobs)]
                    else:
                        ops.conv2d_grouped(ds_jobs, prec)
                        outs_ = ops.conv2d_grouped(jobs, prec)
                else:
                    outs_ = ops.conv2d_grouped(jobs, prec)
                t = {r: o for r, o in zip(act, outs_)}
            cur = t
        return cur

    # ---- stem + stage 1, chunked
    stage1_tags = None
    for k in range(nchunks):
        lo = [(g[0] * k) // nchunks for g in geo]
        hi = [(g[0] * (k + 1)) // nchunks for g in geo]
        cur = {}
        for r, (net, x, prep) in enumerate(zip(nets, xs, preps)):
            if hi[r] <= lo[r]:
                continue
            n, h, w, dev = geo[r]
            ws = net._ws
            h1, w1 = ops.conv_out_size(h, 7, 2, 3), ops.conv_out_size(w, 7, 2, 3)
            h2, w2 = ops.conv_out_size(h1, 3, 2, 1), ops.conv_out_size(w1, 3, 2, 1)
            c = ops.slice_map(ws.map("pool", n, h2, w2, 64, 1, prec, dev), lo[r], hi[r])
            if prec == 4 and FUSE_STEM_POOL and STEM_READS_INPUT != "0" and not isinstance(x, ops.SplitMap) and (
                    STEM_READS_INPUT == "1" or ops.stem_walk_reads(x[lo[r]:hi[r]])):
                # the stem kernel converts the raw input (fp32 image or uint8 tiles) on its way into LDS: no packed copy
                ops.stem_pool_raw(x[lo[r]:hi[r]], prep["stem"], c)
                cur[r] = c
                continue
            xin = net._stem_input(x, "in", prec, lo[r], hi[r])
            if prec != 3 and FUSE_STEM_POOL:
                # one kernel: the full-resolution stem map (4x the pooled one) is never written or read
                ops.stem_pool(xin, prep["stem"], c, prec=prec)
            else:
                s_ = ops.slice_map(ws.map("stem", n, h1, w1, 64, 1, prec, dev), lo[r], hi[r])
                ops.conv2d(xin, prep["stem"], s_, relu=True, prec=prec)
                ops.maxpool3x3s2(s_, c)
            cur[r] = c
        last_views = run_stage(0, cur, lo, hi)
        if stage1_tags is None:
            stage1_tags = {r: (v.h, v.w, v.c) for r, v in last_views.items()}
        else:
            stage1_tags.update({r: (v.h, v.w, v.c) for r, v in last_views.items()})
    # the stage-1 outputs as full-batch maps (the chunks wrote disjoint image ranges of them)
    nb0 = len(a.layer1)
    nc0 = len(preps[0][(0, nb0 - 1)][0])
    cur = {r: nets[r]._ws.map(f"c0.{nb0 - 1}.{nc0 - 1}", geo[r][0], *stage1_tags[r], 1, prec, devs[r]) for r in range(R)}
    outs = [[] for _ in nets]
    zero, full = [0] * R, [g[0] for g in geo]
    for li in range(a.nstages):
        if li > 0:
            cur = run_stage(li, cur, zero, full)
        for r in range(R):
            outs[r].append(cur[r])
            req = stage_pool[r].get(li)
            if req is not None:
                if req.mean is None and req.gem is None:      # not pooled with its conv (chunked stage 1): pool the stored map
                    req.fused = False
                    req.finish(cur[r])
                if li < a.nstages - 1:
                    level_means[r].append(req.mean)
    if SATURATION_CHECK and prec != 3:
        for r, net in enumerate(nets):
            if net.__dict__.get("_sat_checked") != net._prep_key and not torch.cuda.is_current_stream_capturing():
                net.__dict__["_sat_checked"] = net._prep_key
                nsat = sum(ops.count_saturated(m) for m in outs[r])
                net.__dict__["_sat_count"] = nsat
                if nsat:
                    import warnings
                    warnings.warn(f"agplace_amd: {nsat} feature-map elements of a {net.fe_type} trunk sit at the fp16 limit (+-65504): "
                                  f"these weights / inputs leave fp16's range in precision mode {prec}; run this model with "
                                  "Options.mfma_precision = 3 (split-bf16 maps)")
    return outs
