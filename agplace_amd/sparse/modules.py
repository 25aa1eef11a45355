"""MinkowskiEngine-shaped modules of the voxel branch on the HIP kernels (inference forward).

Drop-ins (same constructor arguments, parameter names and state_dict keys) for
    ME.MinkowskiConvolution / ME.MinkowskiBatchNorm          (kernel [K, Cin, Cout] or [Cin, Cout]; `.bn`)
    layers/eca_block.py:14-79      ECALayer, ECABasicBlock
    models/minkfpn.py:19-123       MinkFPN (bottom-up, lateral and top-down paths; reference default num_top_down = 0)
    layers/pooling.py:70-87        MinkGeM
BatchNorm runs in eval mode (folded into the conv epilogue); training of this branch is not built.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import check, ptr
from .coords import SparseTensor


# False: centred convolutions in the natural row order, every tap (measurements)
GROUP_ROWS = True


def _L():
    return _lib.load()


def _alloc_feats(n, c, prec, dev, ws=None, tag=None):
    """[n + 1, c] feature matrix with a zero last row, in the storage format of `prec`.  With a workspace (capacity-mode
    sparse tensors: inference) the buffer is cached under `tag` and zero-filled ONCE at allocation: kernels only ever write
    rows < n, so the zero row stays zero and a steady-state forward allocates and fills nothing."""
    paired = prec == _lib.PREC_BF16X3
    if ws is not None:
        hi = ws.tensor(tag + ".hi", (n + 1, c), torch.bfloat16 if paired else torch.float16, dev, zero=True)
        lo = ws.tensor(tag + ".lo", (n + 1, c), torch.bfloat16, dev, zero=True) if paired else None
        return hi, lo
    if paired:
        # both planes in one allocation, their zero rows in ONE fill (a training step makes ~80 of these matrices: the second
        # 4-microsecond fill per matrix was 0.3 ms of it)
        buf = torch.empty((2, n + 1, c), dtype=torch.bfloat16, device=dev)
        buf[:, n].zero_()
        return buf[0], buf[1]
    hi = torch.empty((n + 1, c), dtype=torch.float16, device=dev)
    hi[n].zero_()
    return hi, None


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

    def fold(self):
        """eval-mode scale / shift, cached per parameter / buffer version (a dozen tiny launches otherwise)"""
        bn = self.bn
        key = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        if getattr(self, "_fold_key", None) != key:
            self._fold, self._fold_key = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps), key
        return self._fold


class MinkowskiConvolution(nn.Module):
    """kernel_size 1/3/5 with stride 1, or kernel_size 2 with stride 2; no bias (ME default)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        if bias or dilation != 1 or dimension != 3 or (kernel_size, stride) not in ((1, 1), (3, 1), (5, 1), (2, 2)):
            raise NotImplementedError("MinkowskiConvolution: kernel/stride (1,1), (3,1), (5,1), (2,2), no bias, D=3")
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, kernel_size, stride
        vol = kernel_size ** 3
        shape = (in_channels, out_channels) if vol == 1 else (vol, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        # ME.utils.kaiming_normal_(mode='fan_out', nonlinearity='relu') (models/resnet.py:77)
        nn.init.normal_(self.kernel, 0.0, math.sqrt(2.0 / (out_channels * vol)))
        self._key, self._planes = None, {}

    def _weights(self, prec):
        key = (self.kernel.data_ptr(), self.kernel._version)
        if key != self._key:
            self._planes, self._key = {}, key
        pl = self._planes.get(prec)
        if pl is None:
            k = self.kernel.detach().float()
            k = k.view(-1, self.in_channels, self.out_channels)
            if self.in_channels == 1:
                pl = (k[:, 0, :].contiguous(),)                                  # fp32 [ntaps][cout]
            else:
                w = k.permute(2, 0, 1).contiguous()                              # [cout][ntaps][cin]
                if prec == _lib.PREC_BF16X3:
                    pl = ops.split_weight(w, _lib.FMT_BF16)
                elif prec == _lib.PREC_F16W2:
                    pl = ops.split_weight(w, _lib.FMT_F16)
                else:
                    pl = ops.split_weight(w, _lib.FMT_F16, want_lo=False)
            self._planes[prec] = pl
        return pl

    def forward(self, x: SparseTensor, bn: MinkowskiBatchNorm = None, relu=False, residual: SparseTensor = None, prec=2, tag=None):
        """tag: name of the output buffer in the tensor's workspace (capacity mode: one buffer per layer, reused every call)."""
        dev = x.keys.device
        scale, shift = bn.fold() if bn is not None else (None, None)
        L = _L()
        tag = tag or f"sp.o{id(self)}"
        if self.in_channels == 1 and self.stride == 1 and x.n_dev is not None and self.out_channels in (32, 64) and self.kernel_size > 1:
            # inference, first layer: no materialised kernel map (125 taps x every voxel), the kernel searches the sorted keys itself
            if x.f32 is None or x.f32.shape[1] != 1:
                raise ValueError("a 1-channel convolution takes the fp32 input features")
            (w,) = self._weights(prec)
            hi, lo = _alloc_feats(x.n, self.out_channels, prec, dev, x._ws, tag)
            check(L.agp_sparse_conv0_fwd(ptr(x.keys), x.n, ptr(x.n_dev), ptr(x.f32), self.kernel_size, x.stride, ptr(w),
                                         self.out_channels, ptr(scale), ptr(shift), 1 if relu else 0, ptr(hi), ptr(lo),
                                         ptr(x.segments()[0]), prec, _lib.stream()),
                  "agp_sparse_conv0_fwd")
            return x.with_feats(hi, lo)
        if self.stride == 2:
            out_sp, nbr = x.strided()
        else:
            out_sp, nbr = x, x.kernel_map(self.kernel_size)
        n_out, ntaps = out_sp.n, nbr.shape[0]
        hi, lo = _alloc_feats(n_out, self.out_channels, prec, dev, x._ws, tag)
        if self.in_channels == 1:
            if x.f32 is None:
                raise ValueError("a 1-channel convolution takes the fp32 input features")
            (w,) = self._weights(prec)
            check(L.agp_sparse_conv_cin1_fwd(ptr(x.f32), x.n, ptr(nbr), n_out, ntaps, ptr(w), self.out_channels, ptr(scale),
                                             ptr(shift), 1 if relu else 0, ptr(hi), ptr(lo), ptr(out_sp.n_dev), _lib.stream()),
                  "agp_sparse_conv_cin1_fwd")
        else:
            w_hi, w_lo = self._weights(prec)
            # a centred 27-tap convolution works in the level's z-plane row order and skips, per tile, the taps nobody has
            grouped = GROUP_ROWS and self.stride == 1 and 1 < ntaps <= 32
            check(L.agp_sparse_conv_fwd(ptr(x.hi), ptr(x.lo), x.n + 1, ptr(nbr), n_out, self.in_channels, self.out_channels,
                                        ntaps, ptr(w_hi), ptr(w_lo), ptr(scale), ptr(shift),
                                        ptr(residual.hi) if residual is not None else None,
                                        ptr(residual.lo) if residual is not None else None, 1 if relu else 0, ptr(hi), ptr(lo),
                                        prec, ptr(out_sp.n_dev), ptr(x.zperm()) if grouped else None,
                                        ptr(x.tile_taps(self.kernel_size)) if grouped else None, _lib.stream()), "agp_sparse_conv_fwd")
        return out_sp.with_feats(hi, lo)


class MinkowskiConvolutionTranspose(nn.Module):
    """ME.MinkowskiConvolutionTranspose(kernel_size=2, stride=2), no bias (models/minkfpn.py:62-63), onto the EXISTING coordinates
    of the next finer level: each fine row has one parent and one active tap (SparseTensor.up_map), so it is the gather-GEMM of
    MinkowskiConvolution on a one-hot table.  kernel [8, Cin, Cout] like ME's."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        if bias or dilation != 1 or dimension != 3 or (kernel_size, stride) != (2, 2):
            raise NotImplementedError("MinkowskiConvolutionTranspose: kernel 2 / stride 2, no bias, D=3")
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, 2, 2
        self.kernel = nn.Parameter(torch.empty((8, in_channels, out_channels)))
        nn.init.normal_(self.kernel, 0.0, math.sqrt(2.0 / (out_channels * 8)))
        self._key, self._planes = None, {}

    _weights = MinkowskiConvolution._weights

    def forward(self, x: SparseTensor, fine: SparseTensor, residual: SparseTensor = None, prec=2, tag=None):
        """x on the coarse level, `fine` any tensor of the next finer level (its coordinates are the output's);
        residual: optional tensor on `fine`'s rows added in the epilogue (the lateral connection, minkfpn.py:117)."""
        dev = x.keys.device
        nbr = fine.up_map(x)
        hi, lo = _alloc_feats(fine.n, self.out_channels, prec, dev, x._ws, tag or f"sp.t{id(self)}")
        w_hi, w_lo = self._weights(prec)
        check(_L().agp_sparse_conv_fwd(ptr(x.hi), ptr(x.lo), x.n + 1, ptr(nbr), fine.n, self.in_channels, self.out_channels, 8,
                                       ptr(w_hi), ptr(w_lo), None, None, ptr(residual.hi) if residual is not None else None,
                                       ptr(residual.lo) if residual is not None else None, 0, ptr(hi), ptr(lo), prec,
                                       ptr(fine.n_dev), None, None, _lib.stream()), "agp_sparse_conv_fwd")
        return fine.with_feats(hi, lo)


def global_avg_pool(x: SparseTensor):
    """ME.MinkowskiGlobalPooling / MinkowskiGlobalAvgPooling -> fp32 [B, C]."""
    seg_off, _ = x.segments()
    c = x.hi.shape[1]
    out = torch.empty((x.nbatch, c), dtype=torch.float32, device=x.hi.device)
    check(_L().agp_seg_pool_fwd(ptr(x.hi), ptr(x.lo), ptr(seg_off), x.nbatch, c, None, 1e-6, ptr(out), None, _lib.stream()),
          "agp_seg_pool_fwd")
    return out


def seg_affine(y: SparseTensor, scale=None, add=None, residual: SparseTensor = None, relu=False, tag=None):
    """relu?(y * scale[b] + add[b] + residual), per-sample vectors broadcast over the sample's rows."""
    _, bidx = y.segments()
    c = y.hi.shape[1]
    for name, v in (("scale", scale), ("add", add)):
        if v is not None and (v.dim() != 2 or v.shape[1] != c or v.device != y.hi.device):
            raise RuntimeError(f"seg_affine: {name} of shape {tuple(v.shape)} on {v.device}, expected [batch, {c}] on {y.hi.device}")
    scale = None if scale is None else scale.contiguous().float()       # read through raw pointers
    add = None if add is None else add.contiguous().float()
    if y._ws is not None:
        hi, lo = _alloc_feats(y.n, c, _lib.PREC_BF16X3 if y.lo is not None else _lib.PREC_F16, y.hi.device, y._ws, tag or f"sp.aff{c}.{y.stride}")
    else:
        hi, lo = torch.empty_like(y.hi), (torch.empty_like(y.lo) if y.lo is not None else None)
        hi[y.n].zero_()
        if lo is not None:
            lo[y.n].zero_()
    check(_L().agp_seg_affine_fwd(ptr(y.hi), ptr(y.lo), ptr(bidx), ptr(scale), ptr(add),
                                  ptr(residual.hi) if residual is not None else None,
                                  ptr(residual.lo) if residual is not None else None, y.n, c, 1 if relu else 0, ptr(hi), ptr(lo),
                                  ptr(y.n_dev), _lib.stream()), "agp_seg_affine_fwd")
    return y.with_feats(hi, lo)


class ECALayer(nn.Module):
    """layers/eca_block.py:14-43"""

    def __init__(self, channels, gamma=2, b=1):
        super().__init__()
        t = int(abs((np.log2(channels) + b) / gamma))
        k_size = t if t % 2 else t + 1
        self.conv = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)
        self.k_size = k_size

    def scale(self, x: SparseTensor):
        mean = global_avg_pool(x)
        out = torch.empty_like(mean)
        w = self.conv.weight.detach().view(-1)            # fp32 [k], contiguous
        check(_L().agp_eca_scale_fwd(ptr(mean), mean.shape[0], mean.shape[1], ptr(w), self.k_size, ptr(out), _lib.stream()),
              "agp_eca_scale_fwd")
        return out


class ECABasicBlock(nn.Module):
    """layers/eca_block.py:46-79 on ME's BasicBlock (conv3-norm-relu-conv3-norm, +residual, relu)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, dimension=3):
        super().__init__()
        if stride != 1:
            raise NotImplementedError
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes)
        self.downsample = downsample
        self.eca = ECALayer(planes, gamma=2, b=1)

    def forward(self, x: SparseTensor, prec=2):
        t = f"sp.blk{id(self)}"
        out = self.conv1(x, self.norm1, relu=True, prec=prec, tag=t + ".c1")
        out = self.conv2(out, self.norm2, relu=False, prec=prec, tag=t + ".c2")
        s = self.eca.scale(out)
        residual = x
        if self.downsample is not None:
            residual = self.downsample[0](x, self.downsample[1], relu=False, prec=prec, tag=t + ".ds")
        return seg_affine(out, scale=s, residual=residual, relu=True, tag=t + ".out")


class MinkGeM(nn.Module):
    """layers/pooling.py:70-87 -> fp32 [B, C]"""

    def __init__(self, input_dim=None, p=3, eps=1e-6):
        super().__init__()
        self.p = nn.Parameter(torch.ones(1) * p)
        self.eps = eps

    def forward(self, x: SparseTensor):
        seg_off, _ = x.segments()
        c = x.hi.shape[1]
        out = torch.empty((x.nbatch, c), dtype=torch.float32, device=x.hi.device)
        check(_L().agp_seg_pool_fwd(ptr(x.hi), ptr(x.lo), ptr(seg_off), x.nbatch, c, ptr(self.p.detach().float()), self.eps,
                                    None, ptr(out), _lib.stream()), "agp_seg_pool_fwd")
        return out


class MinkFPN(nn.Module):
    """models/minkfpn.py:19-123: bottom-up path, lateral 1x1 convolutions and the top-down path of transposed convolutions
    (num_top_down < number of levels: the reference's own forward indexes out_maps out of range when they are equal)."""

    def __init__(self, in_channels, out_channels, num_top_down=0, conv0_kernel_size=5, block=ECABasicBlock,
                 layers=(1, 1, 1), planes=(32, 64, 64)):
        super().__init__()
        assert len(layers) == len(planes) and len(layers) >= 1
        assert 0 <= num_top_down <= len(layers)
        self.num_bottom_up, self.num_top_down = len(layers), num_top_down
        self.planes, self.layers, self.lateral_dim = list(planes), list(layers), out_channels
        self.inplanes = planes[0]
        self.conv0 = MinkowskiConvolution(in_channels, self.inplanes, kernel_size=conv0_kernel_size, dimension=3)
        self.bn0 = MinkowskiBatchNorm(self.inplanes)
        self.convs, self.bns, self.blocks = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.tconvs, self.conv1x1s = nn.ModuleList(), nn.ModuleList()
        for plane, layer in zip(planes, layers):
            self.convs.append(MinkowskiConvolution(self.inplanes, self.inplanes, kernel_size=2, stride=2, dimension=3))
            self.bns.append(MinkowskiBatchNorm(self.inplanes))
            self.blocks.append(self._make_layer(block, plane, layer))
        # lateral connections (minkfpn.py:58-73): one more 1x1 than transposed convolutions
        for i in range(num_top_down):
            self.conv1x1s.append(MinkowskiConvolution(planes[-1 - i], self.lateral_dim, kernel_size=1, stride=1, dimension=3))
            self.tconvs.append(MinkowskiConvolutionTranspose(self.lateral_dim, self.lateral_dim, kernel_size=2, stride=2, dimension=3))
        last = planes[-1 - num_top_down] if num_top_down < self.num_bottom_up else planes[0]
        self.conv1x1s.append(MinkowskiConvolution(last, self.lateral_dim, kernel_size=1, stride=1, dimension=3))

    def _make_layer(self, block, planes, blocks):
        downsample = None
        if self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(MinkowskiConvolution(self.inplanes, planes * block.expansion, kernel_size=1, stride=1,
                                                            dimension=3), MinkowskiBatchNorm(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride=1, downsample=downsample, dimension=3)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, stride=1, dimension=3))
        return nn.Sequential(*layers)

    def forward(self, x: SparseTensor, prec=2):
        if self.num_top_down >= self.num_bottom_up:
            raise IndexError("MinkFPN: num_top_down == number of levels indexes out_maps out of range (models/minkfpn.py:118)")
        out_maps, feature_maps = [], []
        nbu, ntd = self.num_bottom_up, self.num_top_down
        x = self.conv0(x, self.bn0, relu=True, prec=prec)
        for ndx, (conv, bn, blocks) in enumerate(zip(self.convs, self.bns, self.blocks)):
            x = conv(x, bn, relu=True, prec=prec)
            for blk in blocks:
                x = blk(x, prec=prec)
            if nbu - 1 - ntd <= ndx < nbu - 1:
                feature_maps.append(x)
            out_maps.append(x)
        x = self.conv1x1s[0](x, None, relu=False, prec=prec)
        out_maps[-1] = x
        # top-down pass (minkfpn.py:114-118): transposed convolution onto the finer level + its lateral 1x1, one launch each
        for ndx, tconv in enumerate(self.tconvs):
            fm = feature_maps[-ndx - 1]
            lat = self.conv1x1s[ndx + 1](fm, None, relu=False, prec=prec)
            x = tconv(x, fm, residual=lat, prec=prec)
            out_maps[-2 - ndx] = x
        return x, out_maps
