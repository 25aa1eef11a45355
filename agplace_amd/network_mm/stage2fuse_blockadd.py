"""Stage2FuseBlockAdd (+ BasicBlock, Basic, FFNFuse, GeM), drop-ins for reference
network_mm/stage2fuse_blockadd.py:61-135,139-219,286-293.

Per layer (opt.stg2nlayers, default 1), image side:
    imgmap  = imgmap + Linear(fusevec)[:, :, None, None]          (:193-195)
    imgmap  = BasicBlock(imgmap)   conv3x3(+bias)-BN-ReLU-conv3x3(+bias)-BN, +id, ReLU   (:61-79)
    imgout  = GeM(imgmap)                                           (:200)
    imgfuse = avgpool(conv1x1(imgmap))                              (:205-206)
    fusevec = FFNFuse(fusevec + imgfuse + voxfuse)                  (:209-213)
MI355X build: broadcast-add = agp_bcast_add_fwd, the two 3x3 convs = agp_conv2d_fwd with bias and
BN(eval) folded into the epilogue, GeM and the average pool share ONE pass (agp_pool_fwd), and
avgpool(conv1x1(x)) is evaluated as conv1x1(avgpool(x)) (both are linear; exact up to fp32
rounding) with agp_linear_fwd.  The sparse voxel side (ECABasicBlock, MinkGeM, ME 1x1 conv) is
out of scope: `voxmap` carries its two outputs as dense stand-ins
(stg2voxvec [b,C], voxvec_fuse [b,D]).
state_dict keys: projsfuseimg.{i}.0.*, projsfusevox.{i}.0.*, projsimgfuse.{i}.0.* (Conv2d 1x1),
ffnsimg.{i}.{conv1,bn1,conv2,bn2}.*, ffnsfuse.{i}.ffns.{j}.{fc1,ln1,fc2,ln2}.*, poolimage.p,
poolfuse.p, and the sparse side under MinkowskiEngine's names: ffnsvox.{i}.* (ECABasicBlock),
projsvoxfuse.{i}.0.kernel, poolvox.p.  `voxmap` is either an agplace_amd.sparse.SparseTensor (the real
voxel branch, inference) or the pair of dense stand-ins (stg2voxvec, voxvec_fuse).
"""
import torch
import torch.nn as nn

from .. import autograd_ops, ops, sparse, train_fns, train_graph
from ..options import get_options
from .ffns import _PreparedLinear
from .image_pooling import GeM  # noqa: F401  (same class the reference defines locally)


class _Conv1x1AsLinear:
    """View of a 1x1 Conv2d's parameters as an nn.Linear-like object (weight [out,in], bias)."""

    def __init__(self, conv):
        self.conv = conv

    @property
    def weight(self):
        return self.conv.weight.view(self.conv.out_channels, self.conv.in_channels)

    @property
    def bias(self):
        return self.conv.bias


class _PreparedConv1x1(_PreparedLinear):
    def __init__(self, conv):
        self.as_linear = _Conv1x1AsLinear(conv)
        super().__init__(self.as_linear)

    def get(self, with_transpose=False):
        w, b = self.as_linear.conv.weight, self.as_linear.conv.bias
        wt = self.with_transpose or with_transpose
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if key != self._key or (wt and self._lw.wt_hi is None):
            self._lw = ops.LinearWeights(self.as_linear.weight, b, with_transpose=wt)
            self._key = key
        return self._lw


class BasicBlock(nn.Module):
    """conv3x3(bias)-BN-ReLU-conv3x3(bias)-BN, +identity, ReLU on [b,dim,h,w]."""

    def __init__(self, dim):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(dim)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(dim)
        self._key, self._cw = None, None
        self._ws = ops.Workspace()
        self._units = None

    def _prepared(self):
        key = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        if key != self._key:
            cws = []
            for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2)):
                s, t = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                   conv_bias=conv.bias)
                cws.append(ops.ConvWeights(conv.weight, s, t, 1, 1))
            self._cw, self._key = cws, key
        return self._cw

    def forward_map(self, x: ops.SplitMap, prec=3, pool=None):
        """pool: optional ops.PoolReq filled with the global pooling of the block's output (in the last conv's launch)."""
        c1, c2 = self._prepared()
        dev = x.hi.device
        t = self._ws.map("t", x.n, x.h, x.w, x.c, 1, prec, dev)
        o = self._ws.map("o", x.n, x.h, x.w, x.c, 1, prec, dev)
        ops.conv2d(x, c1, t, relu=True, prec=prec)
        ops.conv2d(t, c2, o, residual=x, relu=True, prec=prec, pool=pool)
        return o

    def forward(self, x, prec=3):
        if isinstance(x, ops.SplitMap):
            return self.forward_map(x, prec)
        xm = ops.pack_f32(x, x.shape[1], 1, prec)
        return self.forward_map(xm, prec).to_f32()

    # ---- train mode: batch-statistics BatchNorm, hand-written backward (train_graph.ConvBNUnit)
    def forward_map_train(self, x: ops.SplitMap, prec=3):
        if self._units is None:
            self._units = (train_graph.ConvBNUnit(self.conv1, self.bn1, "t.c1", self._ws),
                           train_graph.ConvBNUnit(self.conv2, self.bn2, "t.c2", self._ws))
        u1, u2 = self._units
        t = u1.forward(x, relu=True, prec=prec, out_h16=u2.wgrad_f16_ok(prec),     # (conv2's one-pass weight gradient reads it)
                       out_f16_only=u2.reads_f16_plane_only(prec))
        return u2.forward(t, residual=x, relu=True, prec=prec)

    def backward_map(self, go: ops.SplitMap):
        """go = dL/d(output map) -> dL/d(input map); accumulates conv / BN parameter gradients."""
        u1, u2 = self._units
        # u2's data-gradient conv reduces u1's BatchNorm-backward sums; u1's adds the residual branch's gradient in its epilogue
        gh, gres, (_, part) = u2.backward(go, stats_for=u1)
        gx, _, (added, _) = u1.backward(gh, partial=part, add=gres)
        if added:
            return gx
        out = self._ws.map("t.gin", gx.n, gx.h, gx.w, gx.c, 1, 3 if gx.lo is not None else 1, gx.hi.device)
        return train_graph.map_add(gx, gres, out)


class Basic(nn.Module):
    """fc-LN-ReLU-fc-LN, +identity, ReLU on [b,dim]."""

    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, dim)
        self.ln1 = nn.LayerNorm(dim)
        self.relu = nn.ReLU(inplace=True)
        self.fc2 = nn.Linear(dim, dim)
        self.ln2 = nn.LayerNorm(dim)
        self._p1, self._p2 = _PreparedLinear(self.fc1), _PreparedLinear(self.fc2)

    def forward(self, x):
        x = x.contiguous()
        out = autograd_ops.linear(x, self.fc1, self._p1)
        out = autograd_ops.layernorm(out, self.ln1, relu=True)
        out = autograd_ops.linear(out, self.fc2, self._p2)
        return autograd_ops.layernorm(out, self.ln2, relu=True, residual=x)


class FFNFuse(nn.Module):
    def __init__(self, dim, stg2fuse_type):
        super().__init__()
        self.stg2fuse_type = stg2fuse_type.split('_')
        self.ffns = nn.ModuleList()
        for e in self.stg2fuse_type:
            if e == 'basic':
                self.ffns.append(Basic(dim))
            else:
                raise NotImplementedError

    def forward(self, x):
        outlist = [ffn(x) for ffn in self.ffns]
        return outlist[0] if len(outlist) == 1 else autograd_ops.wsum(outlist)

    def emit(self, vp, src, free):
        """forward as ops of a vecprog.VecProgram: input register `src` (kept), scratch registers `free` (>= 2 + number
        of Basic blocks).  Returns the result register."""
        from ..vecprog import VecProgramUnfit
        if len(free) < 2 + len(self.ffns):
            raise VecProgramUnfit("registers")
        a, b_ = free[0], free[1]
        outs = []
        for j, ffn in enumerate(self.ffns):
            o = free[2 + j]
            vp.linear(a, ffn._p1.get(), src)
            vp.layernorm(a, ffn.ln1, a, relu=True)
            vp.linear(b_, ffn._p2.get(), a)
            vp.layernorm(o, ffn.ln2, b_, relu=True, residual=src)
            outs.append(o)
        if len(outs) == 1:
            return outs[0]
        return vp.wsum(a, outs)


class Stage2FuseBlockAdd(nn.Module):
    def __init__(self, fusedim, imgdim, bevdim, voxdim, opt=None):
        super().__init__()
        self.opt = opt or get_options()
        opt = self.opt
        self.projsfusebev = nn.ModuleList()
        self.projsfuseimg = nn.ModuleList()
        self.projsfusevox = nn.ModuleList()
        self.ffnsbev = nn.ModuleList()
        self.ffnsimg = nn.ModuleList()
        self.ffnsvox = nn.ModuleList()
        self.projsbevfuse = nn.ModuleList()
        self.projsimgfuse = nn.ModuleList()
        self.projsvoxfuse = nn.ModuleList()
        self.ffnsfuse = nn.ModuleList()
        for i in range(opt.stg2nlayers):
            if opt.stg2_useproj is True:
                self.projsfuseimg.append(nn.Sequential(nn.Linear(fusedim, imgdim)))
                self.projsfusevox.append(nn.Sequential(nn.Linear(fusedim, voxdim)))
                self.projsimgfuse.append(nn.Sequential(nn.Conv2d(imgdim, fusedim, kernel_size=1)))
                self.projsvoxfuse.append(nn.Sequential(sparse.MinkowskiConvolution(voxdim, fusedim, kernel_size=1, dimension=3)))
            else:
                self.projsfuseimg.append(nn.Identity())
                self.projsfusevox.append(nn.Identity())
                self.projsimgfuse.append(nn.Identity())
                self.projsvoxfuse.append(nn.Identity())
            self.ffnsimg.append(BasicBlock(imgdim))
            self.ffnsvox.append(sparse.ECABasicBlock(voxdim, voxdim))
            if opt.stg2fuse_type is not None:
                self.ffnsfuse.append(FFNFuse(dim=fusedim, stg2fuse_type=opt.stg2fuse_type))
        self.poolimage = GeM()
        self.poolvox = sparse.MinkGeM()
        self.poolfuse = GeM()
        self._prep_fuseimg = [_PreparedLinear(m[0]) if isinstance(m, nn.Sequential) else None
                              for m in self.projsfuseimg]
        self._prep_imgfuse = [_PreparedConv1x1(m[0]) if isinstance(m, nn.Sequential) else None
                              for m in self.projsimgfuse]
        self._prep_fusevox = [_PreparedLinear(m[0]) if isinstance(m, nn.Sequential) else None
                              for m in self.projsfusevox]
        self._ws = ops.Workspace()

    def forward_imgvox(self, imgmap, bevmap, voxmap, fusevec, prec=3, train_ctx=None, vox_train_ctx=None):
        # imgmap: ops.SplitMap or fp32 [b,c,h,w]; voxmap: (stg2voxvec [b,C], voxvec_fuse [b,D])
        # train_ctx = (MapSink, token tensor, stage index): train-mode path through train_fns.Stage2ImgFn
        opt = self.opt
        if opt.stg2_type != 'full':
            raise NotImplementedError
        if not isinstance(imgmap, ops.SplitMap):
            imgmap = ops.pack_f32(imgmap, imgmap.shape[1], 1, prec)
        sparse_vox = isinstance(voxmap, sparse.SparseTensor)
        if not sparse_vox:
            voxoutvec, voxvec_fuse = voxmap
        fusevec = fusevec.contiguous()
        imgoutvec = None
        for i in range(opt.stg2nlayers):
            if sparse_vox:
                # sparse side (reference :194-211): broadcast-add the projected fusion vector, ECABasicBlock,
                # MinkGeM; 1x1 ME convolution + global average pool for the fusion update
                if self._prep_fusevox[i] is not None:
                    fusevec_vox = autograd_ops.linear(fusevec, self.projsfusevox[i][0], self._prep_fusevox[i])
                else:
                    fusevec_vox = fusevec
            if sparse_vox and vox_train_ctx is not None:
                # train mode: the sparse side as one autograd node (train_fns.Stage2VoxFn)
                # (layer i > 0 reads layer i-1's output from the sink; its token is an output of that layer's node)
                vsink, vtoken = vox_train_ctx
                voxvec_fuse, voxoutvec = train_fns.Stage2VoxFn.apply(
                    vtoken if i == 0 else voxoutvec, fusevec_vox, self.ffnsvox[i], self.poolvox,
                    self.projsvoxfuse[i][0] if opt.stg2_useproj is True else None, vsink, i)
            elif sparse_vox:
                voxmap = sparse.modules.seg_affine(voxmap, add=fusevec_vox.detach().contiguous().float())
                voxmap = self.ffnsvox[i](voxmap, prec=prec)
                voxoutvec = self.poolvox(voxmap)
                vf = self.projsvoxfuse[i][0](voxmap, prec=prec) if opt.stg2_useproj is True else voxmap
                voxvec_fuse = sparse.modules.global_avg_pool(vf)
            if self._prep_fuseimg[i] is not None:
                fusevec_img = autograd_ops.linear(fusevec, self.projsfuseimg[i][0], self._prep_fuseimg[i])
            else:
                fusevec_img = fusevec
            want_fuse = opt.stg2fuse_type is not None
            if train_ctx is not None:
                sink, token, stage = train_ctx
                res = train_fns.Stage2ImgFn.apply(token if i == 0 else imgoutvec, fusevec_img, self.ffnsimg[i], self.poolimage,
                                                  sink, stage if i == 0 else ("s2", i - 1), prec, want_fuse, ("s2", i))
                mean, imgoutvec = res if want_fuse else (None, res)
            else:
                m = self._ws.map(f"add{i}", imgmap.n, imgmap.h, imgmap.w, imgmap.c, 1, prec, imgmap.hi.device)
                ops.bcast_add(imgmap, fusevec_img, m)
                req = ops.PoolReq(self.poolimage.p, eps=self.poolimage.eps, want_mean=want_fuse, want_gem=True)
                imgmap = self.ffnsimg[i].forward_map(m, prec, pool=req)
                mean, imgoutvec = req.mean, req.gem
            if want_fuse:
                if opt.stg2_useproj is True:
                    imgvec_fuse = autograd_ops.linear(mean, self._prep_imgfuse[i].as_linear, self._prep_imgfuse[i])
                else:
                    imgvec_fuse = mean
                fusevec = autograd_ops.wsum([fusevec, imgvec_fuse, voxvec_fuse.float()])
                fusevec = self.ffnsfuse[i](fusevec)
        return fusevec, imgoutvec, None, voxoutvec

    def forward(self, imagemap, bevmap, voxmap, fusevec, type, prec=3, train_ctx=None, vox_train_ctx=None):
        if type == 'vox':
            return self.forward_imgvox(imagemap, bevmap, voxmap, fusevec, prec=prec, train_ctx=train_ctx,
                                       vox_train_ctx=vox_train_ctx)
        raise NotImplementedError   # 'bev': ffnsbev / poolbev are never built in the reference
