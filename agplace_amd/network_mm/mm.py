"""MM (query network), drop-in for reference network_mm/mm.py:31-172.

`MM(drop=None)`; `modelq(data_dict, mode='q') -> dict` with keys imagevec_org, voxvec_org,
shallowvec_org, stg2fusevec, stg2imagevec, stg2voxvec, embedding (mm.py:150-158).  Attribute
names match what train.py:175-190 reads for its optimizer groups.

Scope: the image backbone, GeM, stage-1 Neural-ODE fusion, stage-2 fusion
and the scalar-weight glue run on hand-written gfx950 kernels.  The sparse-voxel branch
(MinkFPN / MinkGeM / ECABasicBlock, agplace_amd/sparse) runs from `coords` [N,4] / `features` [N,1]
exactly like mm.py:86-89, in inference and in .train() mode (agplace_amd/sparse/train.py); when
`coords` is absent data_dict carries the voxel branch's dense outputs instead:
    vox_levels  [ [b,64], [b,128], [b,256] ]   globally pooled v1..v3  (fuse_block_toshallow.py:83)
    voxfeatvec  [b,256]                          MinkGeM(voxfeatmap)      (mm.py:89)
    stg2voxvec  [b,256], voxvec_fuse [b,256]     stage-2 voxel outputs    (stage2fuse_blockadd.py:201,207)
Execution modes:
  * .eval() under torch.no_grad(): inference, BatchNorm folded into the conv epilogues.
  * .train() with gradients enabled: end-to-end training.  Batch-statistics BatchNorm and the conv
    backward run on HIP kernels (resnet.ResNet.forward_maps_train / backward_maps, train_graph.py);
    feature maps never become autograd tensors (train_fns.py), the vector path (up-dims, Neural-ODE
    blocks, projections, Basic MLP, stg2fusefc, normalisations) uses autograd_ops.py.
  * .eval() with gradients enabled and trainable parameters: the same training graph on FROZEN BatchNorm
    statistics (train_graph.bn_frozen: running statistics read, not updated, constants of the backward).
  * .eval() + freeze_backbone(): train the fusion path only, on frozen image features.
"""
import torch
import torch.nn as nn

from .. import autograd_ops, ops, sparse, train_fns
from ..options import get_options
from .ffns import _PreparedLinear
from .fuse_block_toshallow import FuseBlockToShallow
from ..vecprog import VecProgram, VecProgramUnfit
from .image_fe import ImageFE
from .image_pooling import GeM
from .stage2fuse_blockadd import Stage2FuseBlockAdd


class MM(nn.Module):
    def __init__(self, drop=None, opt=None):
        super().__init__()
        self.opt = opt = opt or get_options()
        self.drop = drop
        self.image_fe = ImageFE(fe_type=opt.mm_imgfe, layers=opt.mm_imgfe_layers)
        self.image_pool = GeM()
        planes = [int(x) for x in opt.mm_voxfe_planes.split('_')]
        layers = [int(x) for x in opt.mm_voxfe_layers.split('_')]
        self.vox_fe = sparse.MinkFPN(in_channels=1, out_channels=planes[-1], planes=planes, layers=layers,
                                     num_top_down=opt.mm_voxfe_ntd, conv0_kernel_size=5, block=sparse.ECABasicBlock)
        self.vox_pool = sparse.MinkGeM()
        self.fuseblocktoshallow = FuseBlockToShallow(
            dims=[opt.mm_stg2fuse_dim for _ in range(len(planes))],
            img_dims=[int(e) for e in opt.mm_imgfe_planes.split('_')],
            vox_dims=[int(e) for e in opt.mm_voxfe_planes.split('_')],
            bev_dims=[int(e) for e in opt.mm_bevfe_planes.split('_')], opt=opt)
        self.stg2fuseblock = Stage2FuseBlockAdd(fusedim=opt.mm_stg2fuse_dim, imgdim=opt.mm_imgfe_dim,
                                                bevdim=opt.mm_bevfe_dim, voxdim=opt.mm_voxfe_dim, opt=opt)
        self.stg2fusefc = nn.Linear(opt.mm_stg2fuse_dim, opt.mm_stg2fuse_dim)
        self._prep_fc = _PreparedLinear(self.stg2fusefc)

        def w(v, learn):
            return nn.Parameter(torch.tensor(v, dtype=torch.float32), requires_grad=learn)
        self.image_weight = w(opt.image_weight, opt.image_learnweight)
        self.vox_weight = w(opt.vox_weight, opt.vox_learnweight)
        self.shallow_weight = w(opt.shallow_weight, opt.shallow_learnweight)
        self.imageorg_weight = w(opt.imagevoxorg_weight, opt.imagevoxorg_learnweight)
        self.voxorg_weight = w(opt.imagevoxorg_weight, opt.imagevoxorg_learnweight)
        self.shalloworg_weight = w(opt.shalloworg_weight, opt.shalloworg_learnweight)
        self.stg2image_weight = w(opt.stg2imagevox_weight, opt.stg2imagevox_learnweight)
        self.stg2vox_weight = w(opt.stg2imagevox_weight, opt.stg2imagevox_learnweight)
        self.stg2fuse_weight = w(opt.stg2fuse_weight, opt.stg2fuse_learnweight)

    def freeze_backbone(self):
        """Fine-tune the fusion path on FROZEN image features (eval-mode BatchNorm, folded convs):
        `image_fe` and the stage-2 `BasicBlock` get requires_grad=False and the model may then be run
        in .eval() with gradients enabled.  Gradient does not flow through feature maps in this mode:
        the stage-2 image block is a constant function of its input (the path fusevec -> projsfuseimg
        -> conv block is cut).  Every vector-path parameter (up-dims, Neural-ODE blocks, projsimgfuse,
        Basic MLP, stg2fusefc) receives its exact gradient w.r.t. the remaining graph.  For end-to-end
        training use .train() instead."""
        for m in [self.image_fe] + list(self.stg2fuseblock.ffnsimg) + list(self.stg2fuseblock.projsfuseimg):
            for p in m.parameters():
                p.requires_grad_(False)
        self.image_pool.p.requires_grad_(False)
        self.stg2fuseblock.poolimage.p.requires_grad_(False)
        self._frozen_backbone = True
        return self


    # ---- the voxel-range flag.  agp_sparse_build zeroes a device word at the start of every build and sets it when it had to clamp
    # a voxel coordinate into the +-32511 range of the 16-bit key fields, met a batch index outside [0, batch size) or a sample of
    # more than 65536 points: the voxel branch's outputs of that batch are then wrong.  Every inference forward from `coords` ORs
    # that word into a STICKY device word and copies the sticky word to pinned host memory, both enqueued behind the build on the
    # calling stream -- captured into a hipGraph they are nodes of the graph, so every REPLAY publishes its own flag and the host
    # can look at it without touching the stream (poll_voxel_range).
    def _vox_slot(self, dev):
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
        slots = self.__dict__.setdefault('_vox_flag_slots', {})
        if key not in slots:
            slots[key] = {'sticky': torch.zeros(1, dtype=torch.int32, device=dev),
                          'host': torch.zeros(1, dtype=torch.int32).pin_memory(), 'event': None, 'calls': 0}
        return slots[key]

    def _raise_voxel_range(self, which):
        for sl in self.__dict__.get('_vox_flag_slots', {}).values():       # the error is reported once: start again from zero
            sl['sticky'].mul_(0)       # (an elementwise kernel, not zero_(): no eager memset beside replayed graphs, csrc/coords.hip)
            sl['host'].zero_()
            sl['event'] = None
        raise ValueError(f"MM.forward_q: in {which} batch a voxel coordinate lies outside the supported range (|c| <= 32511 after "
                         "flooring), a batch index outside [0, batch size), or one sample holds more than 65536 points: the voxel "
                         "branch's outputs of that batch are wrong")

    def poll_voxel_range(self):
        """NON-BLOCKING: raises ValueError if any inference forward from `coords` whose flag has reached the host so far -- eager
        or REPLAYED from a hipGraph -- saw an out-of-range cloud (reads words of pinned host memory; no stream is synchronised,
        so the most recent forwards may not be covered yet: finish a loop with voxel_coords_in_range()).  Call it every few
        replays of a captured forward (agplace_amd.pair.CapturedPair.replay does)."""
        for sl in self.__dict__.get('_vox_flag_slots', {}).values():
            if int(sl['host'][0]) != 0:
                self._raise_voxel_range("an earlier")

    def voxel_coords_in_range(self):
        """False if ANY inference forward from `coords` since the last report (eager or replayed) had to clamp a voxel coordinate,
        met a bad batch index or an oversized sample.  Synchronises the device: call it after the loop, not inside it."""
        slots = self.__dict__.get('_vox_flag_slots', {})
        if not slots:
            return True
        torch.cuda.synchronize()
        return all(int(sl['sticky'].item()) == 0 for sl in slots.values())

    def _publish_voxel_flag(self, flag):
        """Behind the build, on the calling stream (which has joined the voxel side stream).  Eager calls also check: the first call
        of a stream its own flag at once, every later call the flag of the call BEFORE it on that stream (an event behind that
        call's copy: the wait is for work the host enqueued a whole forward ago) -- a bad batch is reported one call late, the
        LAST batch of a loop by voxel_coords_in_range()."""
        dev = flag.device
        sl = self._vox_slot(dev)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            if sl['event'] is not None:
                sl['event'].synchronize()
                if int(sl['host'][0]) != 0:
                    self._raise_voxel_range("the previous")
        sl['sticky'].bitwise_or_(flag)
        sl['host'].copy_(sl['sticky'], non_blocking=True)
        if capturing:
            return
        sl['event'] = torch.cuda.Event()
        sl['event'].record(torch.cuda.current_stream(dev))
        sl['calls'] += 1
        if sl['calls'] == 1:
            sl['event'].synchronize()
            if int(sl['host'][0]) != 0:
                self._raise_voxel_range("this")

    def load_reference_state_dict(self, sd):
        """Load a reference checkpoint's `modelq_state_dict` (the voxel branch uses MinkowskiEngine's
        parameter names, so every key has a home)."""
        return self.load_state_dict(sd, strict=True)

    # ==== query
    def query_image(self, data_dict):
        """The image tensor the trunk sees (mm.py:70-75: drop='image' zeroes it)."""
        image = data_dict['query_image']
        if self.drop == 'image':
            if image.dtype == torch.uint8:
                raise NotImplementedError("drop='image' with uint8 camera tiles")
            image = image * 0
        return image

    def final_pool_request(self):
        """The pooling of the last stage output that forward_q consumes: its GeM (image descriptor, mm.py:85) and its mean
        (fusion level 3, fuse_block_toshallow.py:82) -- an ops.PoolReq the trunk's last conv fills in its own launch."""
        return ops.PoolReq(self.image_pool.p, eps=self.image_pool.eps, want_mean=True, want_gem=True)

    OUT_KEYS = ('imagevec_org', 'voxvec_org', 'shallowvec_org', 'stg2fusevec', 'stg2imagevec', 'stg2voxvec', 'embedding')

    def forward_q(self, data_dict, image_maps=None, out_rows=None, rider=None):
        """image_maps: optional (stage maps, level means, final PoolReq) of `query_image(data_dict)` computed by the
        caller -- agplace_amd.pair runs this trunk in lock-step with the database network's (grouped conv launches).
        out_rows: optional {output key: preallocated fp32 [b, 256] tensor}: the inference path writes those outputs there
        (a sub-batch's row slice of the whole batch's output: the sub-batches then need no concatenation).
        rider: optional list holding ONE deferred vector program of another network (DBVanilla2D.forward_db(defer_head=...)): the
        fused inference path launches it inside its first program's launch and empties the list; a caller that still finds it
        there (per-op path taken) runs it itself."""
        opt = self.opt
        # .train() under torch.no_grad() is a live reference configuration (`with torch.set_grad_enabled(args.train_modelq)`
        # around a model in train mode, train.py:307): batch-statistics BatchNorm with running-stat updates, no tape.
        # The train-mode kernels run; the autograd Functions record nothing when no input requires grad.
        # .eval() with gradients enabled and trainable parameters (fine-tuning on frozen BatchNorm statistics) also takes the
        # training graph: its BatchNorm units read the running statistics and hold them constant in the backward
        # (train_graph.bn_frozen), exactly F.batch_norm(training=False) under autograd.
        train = self.training or (torch.is_grad_enabled() and not getattr(self, "_frozen_backbone", False)
                                  and any(p.requires_grad for p in self.parameters()))
        prec = 3 if train else opt.mfma_precision       # training runs on split-bf16 maps (range + precision of gradients)
        if train:
            from .. import train_graph
            train_graph.FWD_F16 = opt.train_precision == 16      # the opt-in fast mode: one-product forward convs (train_graph.py)
            train_graph.DGRAD_HI_ONLY = opt.train_dgrad_products == 1      # ... and one-product data gradients
        if (not train and torch.is_grad_enabled() and getattr(self, "_frozen_backbone", False) and prec == 4
                and any(p.requires_grad for p in self.parameters())):
            # fine-tuning the fusion path on FROZEN features (freeze_backbone): gradients of near-cancelling sums (a mixing weight's
            # gradient is <G, descriptor>) amplify the one-product mode's 2.4e-4 .. 3.8e-4 descriptor error; the frozen trunk runs in
            # the tight two-product mode (3e-5 .. 1.6e-4) whatever the inference default is
            prec = 2
        image = self.query_image(data_dict)
        if self.drop == 'pc':
            if 'coords' not in data_dict:
                raise NotImplementedError("drop='pc' acts on the sparse voxel branch: pass coords / features")
            data_dict = dict(data_dict)
            c = data_dict['coords'].clone()
            c[:, 1:] = c[:, 1:] * 0
            data_dict['coords'] = c
        if not ('image' in opt.output_type and 'vox' in opt.output_type and 'shallow' in opt.output_type):
            raise NotImplementedError   # other output_type values crash in the reference (mm.py:115-118)
        if True:
            output = []
            # ---- voxel branch (mm.py:86-89).  Training: autograd nodes over its pooled vectors, levels from the device-side
            # coordinate manager with one read-back.  Inference: a chain of ~90 small launches (no host synchronisation), so it
            # runs on its own stream AFTER the image branch's long kernels have been enqueued and is joined where its vectors
            # are needed.
            voxmap, vox_train_ctx = None, None
            vox_side = None
            if 'coords' in data_dict:
                data_dict = dict(data_dict)
                if train:
                    sp = sparse.SparseTensor.from_coords_levels(data_dict['features'], data_dict['coords'], image.shape[0], len(self.vox_fe.convs))
                    vsink = train_fns.VoxSink()
                    *vmeans, vgem = train_fns.VoxTrunkFn.apply(train_fns.anchor_of(self.vox_fe, self.vox_pool.p), sp, self.vox_fe, self.vox_pool, vsink)
                    voxmap = vsink.top
                    data_dict['voxfeatvec'], data_dict['vox_levels'] = vgem, list(vmeans)
                    vox_train_ctx = (vsink, vmeans[-1])
                else:
                    dev = image.device
                    # side stream and capacity workspace per CALLING stream: two forwards in flight on two streams (two captured
                    # graphs replayed side by side, bench.py --inflight 2) must share neither
                    vkey = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
                    vpool = self.__dict__.setdefault('_vox_side', {})
                    if vkey not in vpool:
                        vpool[vkey] = (torch.cuda.Stream(device=dev), ops.Workspace())
                    vox_side, vox_ws = vpool[vkey]
                    vox_side.wait_stream(torch.cuda.current_stream(dev))     # inputs are ready; nothing of the image branch yet
            # ---- image branch
            train_ctx = None
            if train:
                # feature maps stay inside the HIP graph; autograd sees the pooled vectors (train_fns.py)
                sink = train_fns.MapSink()
                *means, imagefeatvec = train_fns.TrunkFn.apply(
                    train_fns.anchor_of(self.image_fe.fe, self.image_pool.p), image, self.image_fe.fe, self.image_pool, sink, prec, True)
                levels = [_Pooled(m) for m in means]
                imagefeatmap = sink.maps[-1]
                train_ctx = (sink, means[-1], len(means) - 1)
            else:
                # the level means, l3's GeM (image descriptor) and l3's mean (fusion level 3) come out of the epilogues of
                # the convs that write those maps (ops.PoolReq): no pass re-reads a stage output
                if image_maps is not None:
                    maps, lvl_means, fpool = image_maps
                else:
                    lvl_means, fpool = [], self.final_pool_request()
                    maps = self.image_fe.forward_maps(image, prec=prec, level_means=lvl_means, final_pool=fpool)
                imagefeatmap = maps[-1]
                mean3, imagefeatvec = fpool.mean, fpool.gem
                levels = [_Pooled(m) for m in lvl_means] + [_Pooled(mean3)]
            if vox_side is not None:
                cur = torch.cuda.current_stream(image.device)
                with torch.cuda.stream(vox_side):
                    # capacity-mode levels: sort / unique / segment offsets on the device, no host synchronisation, every buffer
                    # from the module's workspace -- the branch is hipGraph-capturable (agplace_amd/sparse/coords.py)
                    sp = sparse.SparseTensor.from_coords_capacity(data_dict['features'], data_dict['coords'], image.shape[0], vox_ws)
                    vox_flag = sp.range_flag
                    voxmap, voxmaplist = self.vox_fe(sp, prec=prec)
                    data_dict['voxfeatvec'] = self.vox_pool(voxmap)
                    data_dict['vox_levels'] = [sparse.modules.global_avg_pool(e) for e in voxmaplist]
                cur.wait_stream(vox_side)
                for t in [voxmap.hi, voxmap.lo, data_dict['voxfeatvec']] + data_dict['vox_levels']:
                    if t is not None:
                        t.record_stream(cur)
                self._publish_voxel_flag(vox_flag)
            # ---- inference: the whole vector path as two launches (vecprog.hip) around the stage-2 conv block
            if not train and not torch.is_grad_enabled() and opt.fused_vector_path:
                try:
                    return self._vector_path_fused(data_dict, imagefeatmap, levels, imagefeatvec, voxmap, prec, out_rows or {}, rider)
                except VecProgramUnfit:
                    pass                      # an option set the program cannot express: the per-op path below
            if opt.output_l2 is True:
                imagefeatvec = autograd_ops.l2normalize(imagefeatvec)
            imagefeatvec_org = imagefeatvec
            output.append(autograd_ops.wsum([imagefeatvec], [self.image_weight]))
            # ---- voxel branch outputs (computed above) or the dense stand-ins
            voxfeatvec = data_dict['voxfeatvec'].float()
            if opt.output_l2 is True:
                voxfeatvec = autograd_ops.l2normalize(voxfeatvec)
            voxfeatvec_org = voxfeatvec
            output.append(autograd_ops.wsum([voxfeatvec], [self.vox_weight]))
            # ---- stage-1 fusion
            shallowfeatvec = self.fuseblocktoshallow(levels, None, data_dict['vox_levels'], type='vox')
            shallowfeatvecorg = shallowfeatvec
            if opt.output_l2 is True:
                shallowfeatvec = autograd_ops.l2normalize(shallowfeatvec)
            output.append(autograd_ops.wsum([shallowfeatvec], [self.shallow_weight]))
            # ---- stage-2 fusion
            stg2fusevec, stg2imagevec, _, stg2voxvec = self.stg2fuseblock(
                imagefeatmap, None,
                voxmap if voxmap is not None else (data_dict['stg2voxvec'].float(), data_dict['voxvec_fuse'].float()),
                output[-1], type='vox', prec=prec, train_ctx=train_ctx, vox_train_ctx=vox_train_ctx)
            stg2fusevec = autograd_ops.linear(stg2fusevec, self.stg2fusefc, self._prep_fc)
            # ---- final output
            terms, weights = [], []
            for name, vec, wt in (('imageorg', imagefeatvec_org, self.imageorg_weight),
                                  ('voxorg', voxfeatvec_org, self.voxorg_weight),
                                  ('shalloworg', shallowfeatvec, self.shalloworg_weight),
                                  ('stg2image', stg2imagevec, self.stg2image_weight),
                                  ('stg2vox', stg2voxvec, self.stg2vox_weight),
                                  ('stg2fuse', stg2fusevec, self.stg2fuse_weight)):
                if name in opt.final_type:
                    terms.append(vec)
                    weights.append(wt)
            if opt.final_fusetype == 'add':
                x = autograd_ops.wsum(terms, weights)
            elif opt.final_fusetype == 'cat':
                x = torch.cat([autograd_ops.wsum([t], [w]) for t, w in zip(terms, weights)], dim=-1)
            elif opt.final_fusetype == 'catadd':
                x = torch.cat([autograd_ops.wsum([t], [w]) for t, w in zip(terms[:-1], weights[:-1])], dim=-1)
                x = autograd_ops.wsum([x, terms[-1]], [None, weights[-1]])
            else:
                raise NotImplementedError
            if opt.final_l2 is True:
                x = autograd_ops.l2normalize(x)
        return {
            'imagevec_org': imagefeatvec_org,
            'voxvec_org': voxfeatvec_org,
            'shallowvec_org': shallowfeatvecorg,
            'stg2fusevec': stg2fusevec,
            'stg2imagevec': stg2imagevec,
            'stg2voxvec': stg2voxvec,
            'embedding': x,
        }

    def _vector_path_fused(self, data_dict, imagefeatmap, levels, gem3, voxmap, prec, out_rows, rider=None):
        """Everything of forward_q after the backbones (mm.py:91-129) for inference: program 1 = descriptors' F.normalize,
        FuseBlockToShallow, the stage-2 projections of the fusion vector; the stage-2 conv block (and sparse block) on
        their own kernels; program 2 = fusion update, FFNFuse, stg2fusefc, the final weighted sum.  Same arithmetic as
        the per-op path (autograd_ops), 2 launches instead of ~27."""
        opt, s2 = self.opt, self.stg2fuseblock
        if (opt.stg2nlayers != 1 or opt.stg2_type != 'full' or opt.stg2fuse_type is None or opt.final_fusetype != 'add'
                or opt.mm_stg2fuse_dim != 256 or gem3.shape[1] != 256):
            raise VecProgramUnfit("options")
        b, dev = gem3.shape[0], gem3.device
        sparse_vox = voxmap is not None
        vp = VecProgram(b, dev)
        vp.load(0, gem3)
        if opt.output_l2 is True:
            vp.l2norm(0, 0)
        imagevec_org = vp.store(0, out_rows.get('imagevec_org'))
        vp.load(1, data_dict['voxfeatvec'].float())
        if opt.output_l2 is True:
            vp.l2norm(1, 1)
        voxvec_org = vp.store(1, out_rows.get('voxvec_org'))
        r = self.fuseblocktoshallow.emit(vp, levels, data_dict['vox_levels'])
        shallow_org = vp.store(r, out_rows.get('shallowvec_org'))
        shallow_n = shallow_org
        if opt.output_l2 is True:
            vp.l2norm(r, r)
            shallow_n = vp.store(r)
        vp.wsum(1, [r], [self.shallow_weight])
        fusevec = vp.store(1)
        fv_img = fv_vox = fusevec
        if s2._prep_fuseimg[0] is not None:
            vp.linear(2, s2._prep_fuseimg[0].get(), 1)
            fv_img = vp.store(2)
        if sparse_vox and s2._prep_fusevox[0] is not None:
            vp.linear(3, s2._prep_fusevox[0].get(), 1)
            fv_vox = vp.store(3)
        if rider and vp.fits_beside(rider[0]):
            vp.run(rider=rider.pop())       # the other network's head on workgroups of its own, inside this launch
        else:
            vp.run()
        # ---- stage-2 blocks (stage2fuse_blockadd.py:194-216)
        m = s2._ws.map("add0", imagefeatmap.n, imagefeatmap.h, imagefeatmap.w, imagefeatmap.c, 1, prec, dev)
        ops.bcast_add(imagefeatmap, fv_img, m)
        s2pool = ops.PoolReq(s2.poolimage.p, eps=s2.poolimage.eps, want_mean=True, want_gem=True, gem_out=out_rows.get('stg2imagevec'))
        imap = s2.ffnsimg[0].forward_map(m, prec, pool=s2pool)       # GeM + mean of the block's output ride in its last conv
        mean, stg2imagevec = s2pool.mean, s2pool.gem
        if sparse_vox:
            vm = sparse.modules.seg_affine(voxmap, add=fv_vox)
            vm = s2.ffnsvox[0](vm, prec=prec)
            stg2voxvec = s2.poolvox(vm)
            vf = s2.projsvoxfuse[0][0](vm, prec=prec) if opt.stg2_useproj is True else vm
            voxvec_fuse = sparse.modules.global_avg_pool(vf)
        else:
            stg2voxvec, voxvec_fuse = data_dict['stg2voxvec'].float(), data_dict['voxvec_fuse'].float()
        # ---- program 2
        vt = VecProgram(b, dev)
        if s2._prep_imgfuse[0] is not None:
            vt.linear(0, s2._prep_imgfuse[0].get(), mean)
        else:
            vt.load(0, mean)
        vt.load(1, fusevec)
        vt.load(2, voxvec_fuse.float())
        vt.wsum(1, [1, 0, 2])
        r = s2.ffnsfuse[0].emit(vt, 1, [0, 2, 3, 4, 5])
        vt.linear(0, self._prep_fc.get(), r)
        stg2fusevec = vt.store(0, out_rows.get('stg2fusevec'))
        regs, weights, nxt = [], [], 1
        for name, vec, wt in (('imageorg', imagevec_org, self.imageorg_weight), ('voxorg', voxvec_org, self.voxorg_weight),
                              ('shalloworg', shallow_n, self.shalloworg_weight), ('stg2image', stg2imagevec, self.stg2image_weight),
                              ('stg2vox', stg2voxvec, self.stg2vox_weight), ('stg2fuse', None, self.stg2fuse_weight)):
            if name not in opt.final_type:
                continue
            if vec is None:
                regs.append(0)
            else:
                regs.append(vt.load(nxt, vec.float()))
                nxt += 1
            weights.append(wt)
        if not regs:
            raise VecProgramUnfit("empty final_type")
        vt.wsum(nxt if nxt < 6 else 0, regs, weights)
        out = nxt if nxt < 6 else 0
        if opt.final_l2 is True:
            vt.l2norm(out, out)
        x = vt.store(out, out_rows.get('embedding'))
        vt.run()
        return {
            'imagevec_org': imagevec_org,
            'voxvec_org': voxvec_org,
            'shallowvec_org': shallow_org,
            'stg2fusevec': stg2fusevec,
            'stg2imagevec': stg2imagevec,
            'stg2voxvec': stg2voxvec,
            'embedding': x,
        }

    def forward(self, data_dict, mode):
        if mode == 'q':
            k = self.opt.query_substreams
            img = data_dict.get('query_image')
            if (k > 1 and not self.training and not torch.is_grad_enabled() and 'coords' not in data_dict
                    and torch.is_tensor(img) and img.shape[0] % k == 0 and img.shape[0] >= 2 * k):
                return self._forward_q_substreams(data_dict, k)
            return self.forward_q(data_dict)
        raise NotImplementedError

    def _forward_q_substreams(self, data_dict, k):
        """Inference only: the batch as k equal sub-batches on k HIP streams (the caller's stream + k-1
        side streams owned by the module).  Same arithmetic per sample; one sub-batch's kernel tails
        overlap the others' kernels (a launch of the 3x3 conv kernel has only a few workgroups per CU at
        these sizes).  Module workspaces are keyed by the launching stream, so the passes do not collide."""
        dev = data_dict['query_image'].device
        cur = torch.cuda.current_stream(dev)
        # side streams per CALLING stream: two forwards in flight on two streams (two captured graphs replayed
        # back to back, bench.py --inflight 2) must not share side streams, whose ids key the workspaces
        key = (str(dev), k, cur.cuda_stream)
        pool = self.__dict__.setdefault('_substream_pool', {})
        if key not in pool:
            pool[key] = [torch.cuda.Stream(device=dev) for _ in range(k - 1)]
        substreams = pool[key]
        b = data_dict['query_image'].shape[0]
        hb = b // k

        def part(i):
            out = {}
            for name, v in data_dict.items():
                if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == b:
                    out[name] = v[i * hb:(i + 1) * hb]
                elif isinstance(v, (list, tuple)) and all(torch.is_tensor(t) and t.shape[0] == b for t in v):
                    out[name] = [t[i * hb:(i + 1) * hb] for t in v]
                else:
                    out[name] = v
            return out
        outs = [None] * k
        # whole-batch outputs; every sub-batch writes its row slice (the fused vector path: no concatenation afterwards)
        full = {name: torch.empty((b, self.opt.mm_stg2fuse_dim), dtype=torch.float32, device=dev) for name in self.OUT_KEYS}

        def rows(i):
            return {name: t[i * hb:(i + 1) * hb] for name, t in full.items()}
        for i, st in enumerate(substreams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs[i + 1] = self.forward_q(part(i + 1), out_rows=rows(i + 1))
        outs[0] = self.forward_q(part(0), out_rows=rows(0))
        for i, st in enumerate(substreams):
            cur.wait_stream(st)
            for t in outs[i + 1].values():
                t.record_stream(cur)
        for t in full.values():
            for st in substreams:
                t.record_stream(st)
        return {name: ops.join_rows([o[name] for o in outs]) for name in outs[0]}


class _Pooled:
    """An already average-pooled level (saves re-reading l3, which GeM just streamed)."""

    def __init__(self, vec):
        self.vec = vec
