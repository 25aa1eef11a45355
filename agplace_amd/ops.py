"""Thin host wrappers over the C ABI: tensors in, kernels enqueued on the current stream.

Every function here launches hand-written gfx950 kernels from libagplace_hip.so.  PyTorch
only provides device memory and the stream.  Nothing in this file computes on the CPU.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import ptr, check

GEM_EPS = 1e-6
# bench.py sets this to a list to time every conv launch with events on the launch stream:
# entries are (start_event, end_event, algorithmic_MACs).
CONV_PROFILE = None
# Module-level switches (plain attributes: tests and the A/B tools set them; nothing here reads the process environment).
# F16W2 3x3 convs: run the weight-residual (`lo`) product on the block-scaled fp8 MFMA (agp_conv_desc.w_q8)
LO_FP8 = True
# training: the split kernel writes the weight planes of 3x3 convs chunk-major (agp_conv_desc.w_cm) where the 3x3 stride-1 kernel
# reads them
CHUNK_MAJOR_TRAIN = True
# hand the chunk-major weight planes (agp_conv_desc.w_cm / w_cm_lo) to the kernels that stage weights chunk-wise; False: they read
# w_hi / w_lo (bit-identical, tests/test_gpu_kernels.py)
USE_W_CM = True


def _L():
    return _lib.load()


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor must live on the GPU (agplace_amd has no CPU path)")


# --------------------------------------------------------------------------- maps
class SplitMap:
    """Halo-padded NHWC feature map [n, h+2*pad, w+2*pad, c]; the halo is zero and is never
    written by a kernel.  Two storage formats (include/agplace_hip.h):
      prec 3 (BF16X3): split-bf16 planes, value = float(hi) + float(lo)
      prec 2 / 4 (F16W2 / F16): ONE fp16 plane in `hi`, lo is None
    (`hi` is allocated as a 16-bit torch tensor either way; only the kernels interpret it.)
    """
    __slots__ = ("hi", "lo", "n", "h", "w", "c", "pad", "h16")

    def __init__(self, hi, lo, n, h, w, c, pad, h16=None):
        self.hi, self.lo, self.n, self.h, self.w, self.c, self.pad = hi, lo, n, h, w, c, pad
        # training (split-bf16 maps): the same values once more as ONE fp16 plane (zero halo), written by the pass that
        # produces the map (agp_map_affine / agp_affine_maxpool3x3s2_fwd: o_h16) for the one-pass weight gradient of the conv
        # that consumes it (agp_conv_desc.in_h16); None = not kept
        self.h16 = h16

    def with_h16(self):
        """Allocate the fp16 operand plane (once; zero halo -- the kernels write the interior only)."""
        if self.h16 is None:
            self.h16 = torch.zeros(self.hi.shape, dtype=torch.float16, device=self.hi.device)
        return self

    @staticmethod
    def alloc(n, h, w, c, pad, prec, device):
        shape = (n, h + 2 * pad, w + 2 * pad, c)
        if prec not in (_lib.PREC_F16W2, _lib.PREC_BF16X3, _lib.PREC_F16):
            raise ValueError(f"feature-map precision must be 2 (F16W2), 3 (BF16X3) or 4 (F16), got {prec}")
        paired = prec == _lib.PREC_BF16X3
        # ONLY THE HALO IS ZEROED: the interior is uninitialised memory until the kernel that produces the map has written ALL of
        # it (every in-tree producer does: pack, upsample2_zero, the conv epilogues).  A caller that fills part of a map itself
        # must zero the rest.
        hi = torch.empty(shape, dtype=torch.bfloat16 if paired else torch.float16, device=device)
        lo = torch.empty(shape, dtype=torch.bfloat16, device=device) if paired else None
        if pad and hi.numel():
            check(_L().agp_map_zero_halo(ptr(hi), ptr(lo), n, h, w, c, pad, _lib.stream()), "agp_map_zero_halo")
        return SplitMap(hi, lo, n, h, w, c, pad)

    @property
    def prec(self):
        return _lib.PREC_BF16X3 if self.lo is not None else _lib.PREC_F16W2

    def to_f32(self):
        """Dense fp32 tensor of logical shape [n,c,h,w] in channels_last memory format."""
        out = torch.empty((self.n, self.h, self.w, self.c), dtype=torch.float32, device=self.hi.device)
        check(_L().agp_unpack_nhwc_to_f32(ptr(self.hi), ptr(self.lo), self.n, self.h, self.w, self.c,
                                          self.pad, ptr(out), _lib.stream()), "agp_unpack_nhwc_to_f32")
        return out.permute(0, 3, 1, 2)


def join_rows(parts):
    """Row blocks [n_i, ...] of the sub-batches of one batch -> the whole batch.  When the blocks are consecutive row ranges
    of ONE buffer (sub-batch outputs written straight into their slice of a preallocated tensor, or slices of an input that
    were passed through) the result is a view of it -- no copy, no launch; otherwise torch.cat."""
    p0 = parts[0]
    if all(t.is_contiguous() and t.dtype == p0.dtype and t.device == p0.device and t.shape[1:] == p0.shape[1:] for t in parts):
        row = p0[0].numel() if p0.shape[0] else 0
        off, ok, base = p0.storage_offset(), row > 0, p0.untyped_storage().data_ptr()
        for t in parts:
            ok = ok and t.untyped_storage().data_ptr() == base and t.storage_offset() == off
            off += t.shape[0] * row
        if ok:
            n = sum(t.shape[0] for t in parts)
            return p0.new_empty(0).set_(p0.untyped_storage(), p0.storage_offset(), (n,) + tuple(p0.shape[1:]), p0.stride())
    return torch.cat(parts, 0)


def slice_map(m: SplitMap, lo, hi):
    """Images [lo, hi) of a map (a view: the planes are contiguous in the image index)."""
    if lo == 0 and hi == m.n:
        return m
    return SplitMap(m.hi[lo:hi], None if m.lo is None else m.lo[lo:hi], hi - lo, m.h, m.w, m.c, m.pad,
                    None if m.h16 is None else m.h16[lo:hi])


def count_saturated(m: SplitMap):
    """DIAGNOSTIC (not on the product path; uses torch): how many elements of an fp16 map sit at +-65504, the value the
    kernels' fp16 stores saturate at.  A non-zero count on a real checkpoint means its activations leave fp16's range:
    run that model with Options.mfma_precision = 3 (split-bf16 maps, fp32 range)."""
    if m.lo is not None:
        return 0
    return int((m.hi.view(torch.float16).abs() >= 65504).sum().item())


class Workspace:
    """Caches zero-haloed buffers by (tag, geometry, stream) so steady-state steps allocate nothing (a map's INTERIOR is
    uninitialised until its producer kernel has run: SplitMap.alloc).
    The launching stream is part of the key: two forwards of one module issued on two HIP streams
    (bench.py splits a batch that way so that one half's kernel tails overlap the other half's kernels)
    get disjoint activation buffers."""

    def __init__(self):
        self.bufs = {}

    def map(self, tag, n, h, w, c, pad, prec, device, h16=False):
        key = (tag, n, h, w, c, pad, prec, str(device), torch.cuda.current_stream(device).cuda_stream)
        m = self.bufs.get(key)
        if m is None:
            m = SplitMap.alloc(n, h, w, c, pad, prec, device)
            self.bufs[key] = m
        if h16:
            m.with_h16()
        return m

    def tensor(self, tag, shape, dtype, device, zero=False):
        """Cached buffer; `zero=True` zero-fills it ONCE at allocation (callers then only ever write
        the same positions, so padding/margins stay zero)."""
        key = (tag, tuple(shape), dtype, str(device), torch.cuda.current_stream(device).cuda_stream)
        t = self.bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
            self.bufs[key] = t
        return t


def pack_f32(x, cpad, pad, prec, out=None):
    """fp32 [n,c,h,w] (any strides) -> SplitMap with `cpad` channels and halo `pad`."""
    _need_cuda(x, "pack_f32")
    if x.dtype != torch.float32:
        x = x.float()
    n, c, h, w = x.shape
    if out is None:
        out = SplitMap.alloc(n, h, w, cpad, pad, prec, x.device)
    sn, sc, sh, sw = x.stride()
    if out.h16 is not None:
        # the training stem's input: the fp16 operand plane of its one-pass weight gradient from the same pass
        rc = _L().agp_pack_f32_to_nhwc4_h16(ptr(x), sn, sc, sh, sw, n, c, h, w, pad, ptr(out.hi), ptr(out.lo), ptr(out.h16),
                                            _lib.stream()) if cpad == 4 else _lib.E_UNSUPPORTED
        if rc == 0:
            return out
        if rc != _lib.E_UNSUPPORTED:
            check(rc, "agp_pack_f32_to_nhwc4_h16")
        # strided / unaligned input: no plane, three-product weight gradient.  A plane-less VIEW is returned: `out` may be a
        # module's cached workspace map, whose plane later steps (and captured graphs) keep using
        out = SplitMap(out.hi, out.lo, out.n, out.h, out.w, out.c, out.pad)
    check(_L().agp_pack_f32_to_nhwc(ptr(x), sn, sc, sh, sw, n, c, h, w, cpad, pad, ptr(out.hi),
                                    ptr(out.lo), _lib.stream()), "agp_pack_f32_to_nhwc")
    return out


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def pack_cameras_u8(img_u8, prec, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None):
    """uint8 [n, ncam, h, w, 3] camera tiles (HWC, on the GPU) -> the stem's input map: ToTensor +
    Normalize + width-concat + NHWC4 packing in ONE pass (reference datasets_ws_nuscenes.py:608-634
    does the first three on CPU workers, image_fe.py:98 feeds the fp32 result to conv1)."""
    _need_cuda(img_u8, "pack_cameras_u8")
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 5 or img_u8.shape[-1] != 3:
        raise ValueError("pack_cameras_u8 expects uint8 [n, ncam, h, w, 3]")
    img_u8 = img_u8.contiguous()
    n, ncam, h, w, _ = img_u8.shape
    if out is None:
        out = SplitMap.alloc(n, h, ncam * w, 4, 3, prec, img_u8.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    check(_L().agp_pack_u8_cams_to_nhwc(ptr(img_u8), n, ncam, h, w, m, s, 3, ptr(out.hi), ptr(out.lo), _lib.stream()),
          "agp_pack_u8_cams_to_nhwc")
    return out


def split_weight(w, fmt=_lib.FMT_BF16, want_lo=True):
    """fp32 tensor -> (hi, lo) 16-bit planes (bf16 or fp16) on the same device (kernel: agp_split_f32)."""
    _need_cuda(w, "split_weight")
    w = w.detach().contiguous().float()
    dt = torch.bfloat16 if fmt == _lib.FMT_BF16 else torch.float16
    hi = torch.empty(w.shape, dtype=dt, device=w.device)
    lo = torch.empty(w.shape, dtype=dt, device=w.device) if want_lo else None
    check(_L().agp_split_f32(ptr(w), ptr(hi), ptr(lo), w.numel(), fmt, _lib.stream()), "agp_split_f32")
    return hi, lo


# ---------------------------------------------------------------------------- conv
class ConvWeights:
    """Device-side prepared conv: [cout][kh][kw][cin] weight planes (split lazily per MFMA
    precision: bf16 pair / fp16 pair / fp16 single) + folded scale/shift."""
    __slots__ = ("w", "_planes", "scale", "shift", "cout", "cin", "kh", "kw", "stride", "pad",
                 "in_w_step_stem", "alg_k", "_train_cm")

    def __init__(self, weight, scale, shift, stride, pad, stem=False):
        cout, cin, kh, kw = weight.shape
        self.alg_k = cin * kh * kw      # algorithmic reduction length (147 for the stem, not 224)
        w = weight.detach().float()
        if stem:
            # 7x7x3 stem -> taps (ky) of 8 pixels x 4 channels = 32 contiguous elements of the
            # packed NHWC4 image; kx = 7 and channel 3 carry zero weights.
            wp = torch.zeros((cout, kh, 8, 4), dtype=torch.float32, device=w.device)
            wp[:, :, :kw, :cin] = w.permute(0, 2, 3, 1)
            self.cin, self.kh, self.kw = 32, kh, 1
            w = wp.reshape(cout, kh, 1, 32)
        else:
            w = w.permute(0, 2, 3, 1).contiguous()
            self.cin, self.kh, self.kw = cin, kh, kw
        self.w, self._planes = w, {}
        self.scale = None if scale is None else scale.detach().float().contiguous()
        self.shift = None if shift is None else shift.detach().float().contiguous()
        self.cout, self.stride, self.pad = cout, stride, pad
        self.in_w_step_stem = 4 if stem else 0


    @classmethod
    def for_training(cls, weight, shift, stride, pad, dgrad=False, fwd_stride=None, plane_pixels=None):
        """Split-bf16 planes straight from an nn.Conv2d weight [cout][cin][kh][kw] in ONE launch (agp_split_conv_weight):
        the forward conv's layout, or (dgrad) the flipped / transposed weights of its data-gradient conv.  Training
        rebuilds these every step (the optimizer moves the parameter), so the permute / flip / contiguous / split
        chain of the generic constructor (up to six small launches) matters there."""
        cout, cin, kh, kw = weight.shape
        _need_cuda(weight, "ConvWeights.for_training")
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        self = cls.__new__(cls)
        n, c = (cin, cout) if dgrad else (cout, cin)
        if cin % 8 == 0 and cout % 8 == 0:
            # both plane pairs (forward + data gradient) in ONE launch per weight VERSION, kept on the parameter: a step asks for
            # the forward planes in its forward and the data-gradient planes in its backward (and a shared trunk several times)
            # chunk-major planes (agp_conv_desc.w_cm) where the 3x3 stride-1 kernel is the one that reads them: the forward
            # planes of a 3x3 / stride-1 / pad-1 conv, the data-gradient planes of any 3x3 conv (its dgrad is such a conv)
            fs = stride if (not dgrad and fwd_stride is None) else fwd_stride
            cm_ok = kh == 3 and kw == 3 and CHUNK_MAJOR_TRAIN
            # (exactly the convs agp_conv2d_fwd hands to that kernel -- igemm.hip conv_kxr_ok: cin % 32 == 0 and cout % 64 == 0 of the
            # conv that reads the pair, and its input plane below 2 GiB: `plane_pixels` = padded pixels n (h+2) (w+2) of the forward
            # conv's input map, which bounds the data-gradient conv's input too; a plane the 3x3 kernel cannot address goes to the
            # generic kernel, which reads row-major planes)
            big_f = plane_pixels is not None and plane_pixels * cin * 2 >= (1 << 31)
            big_d = plane_pixels is not None and plane_pixels * cout * 2 >= (1 << 31)
            cmbits = (1 if (cm_ok and fs == 1 and cin % 32 == 0 and cout % 64 == 0 and (dgrad or pad == 1) and not big_f) else 0) \
                | (2 if (cm_ok and cout % 32 == 0 and cin % 64 == 0 and not big_d) else 0)
            key = (weight._version, w.data_ptr(), torch.cuda.current_stream(w.device).cuda_stream, cmbits)
            cache = getattr(weight, "_agp_train_planes", None)
            capturing = torch.cuda.is_current_stream_capturing()       # a captured step must contain its own split launches
            # a forward request always rebuilds (in-place edits through `.data` do not move `_version`): the cache only carries the
            # data-gradient pair from a step's forward to its backward
            if cache is None or cache[0] != key or capturing or not dgrad:
                hi = torch.empty((cout, kh, kw, cin), dtype=torch.bfloat16, device=w.device)
                lo = torch.empty_like(hi)
                hi_d = torch.empty((cin, kh, kw, cout), dtype=torch.bfloat16, device=w.device)
                lo_d = torch.empty_like(hi_d)
                check(_L().agp_split_conv_weight_both(ptr(w), cout, cin, kh, kw, ptr(hi), ptr(lo), ptr(hi_d), ptr(lo_d), cmbits,
                                                      _lib.stream()), "agp_split_conv_weight_both")
                cache = (key, (hi, lo), (hi_d, lo_d))
                weight._agp_train_planes = None if capturing else cache
            hi, lo = cache[2] if dgrad else cache[1]
            self._train_cm = bool(cmbits & (2 if dgrad else 1))
        else:
            self._train_cm = False
            hi = torch.empty((n, kh, kw, c), dtype=torch.bfloat16, device=w.device)
            lo = torch.empty_like(hi)
            check(_L().agp_split_conv_weight(ptr(w), cout, cin, kh, kw, 1 if dgrad else 0, ptr(hi), ptr(lo), _lib.stream()),
                  "agp_split_conv_weight")
        self.w, self._planes = None, {_lib.PREC_BF16X3: (hi, lo)}
        self.scale = None
        self.shift = None if shift is None else shift.detach().float().contiguous()
        self.cout, self.cin, self.kh, self.kw = n, c, kh, kw
        self.stride, self.pad, self.in_w_step_stem = stride, pad, 0
        self.alg_k = c * kh * kw
        return self

    def planes(self, prec):
        pl = self._planes.get(prec)
        if pl is None and self.w is None:
            raise ValueError("ConvWeights.for_training holds split-bf16 planes only")
        if pl is None:
            if prec == _lib.PREC_BF16X3:
                pl = split_weight(self.w, _lib.FMT_BF16)
            elif prec == _lib.PREC_F16W2:
                pl = split_weight(self.w, _lib.FMT_F16)
            elif prec == _lib.PREC_F16:
                pl = split_weight(self.w, _lib.FMT_F16, want_lo=False)
            else:
                raise ValueError(f"conv precision must be 2 (F16W2), 3 (BF16X3) or 4 (F16), got {prec}")
            self._planes[prec] = pl
        return pl

    def cm2(self, prec):
        """(hi, lo) planes of a two-plane mode (F16W2 / BF16X3) in chunk-major order (agp_conv_desc.w_cm / w_cm_lo) for the 3x3
        stride-1 kernel, or None (other convs; weights that exist as training planes only)."""
        key = ("cm2", prec)
        if key not in self._planes:
            pl = None
            if self.w is not None and self.kh == 3 and self.kw == 3 and self.stride == 1 and self.pad == 1 \
                    and not self.in_w_step_stem and self.cin % 32 == 0:
                hi, lo = self.planes(prec)
                if lo is not None:
                    pl = tuple(t.view(self.cout, -1, 32).permute(1, 0, 2).contiguous() for t in (hi, lo))
            self._planes[key] = pl
        return self._planes[key]

    def cm(self):
        """The fp16 weights in chunk-major order [kh*kw*cin / 32][cout][32] (agp_conv_desc.w_cm) for the kernels that
        take them (3x3 pad-1 convs of stride 1 or 2, the 1x1 stride-2 downsample beside the latter), or None."""
        if "cm" not in self._planes:
            pl = None
            k3 = self.kh == 3 and self.kw == 3 and self.pad == 1 and self.stride in (1, 2)
            k1 = self.kh == 1 and self.kw == 1 and self.pad == 0 and self.stride == 2
            if (k3 or k1) and not self.in_w_step_stem and self.cin % 32 == 0:
                hi = self.planes(_lib.PREC_F16)[0]
                pl = hi.view(self.cout, -1, 32).permute(1, 0, 2).contiguous()
            self._planes["cm"] = pl
        return self._planes["cm"]

    def q8(self):
        """(plane, exp) of the optional e4m3 lo plane of the F16W2 mode (agp_conv_desc.w_q8), or None when the conv
        is not a 3x3 stride-1 conv with cin % 64 == 0.  plane[n][pair][lh][tap][ks][e] = e4m3((w - fp16(w)) * 2^exp)
        of channel 32*cc + 16*ks + 8*lh + e at the tap of phase 2*pair + tap, phases in the kernel's execution order
        (ky, cc, kx)."""
        if "q8" not in self._planes:
            pl = None
            if self.kh == 3 and self.kw == 3 and self.stride == 1 and self.pad == 1 and self.cin % 64 == 0:
                _need_cuda(self.w, "ConvWeights.q8")
                plane = torch.empty((self.cout, 9 * self.cin), dtype=torch.uint8, device=self.w.device)
                e = C.c_int32(0)
                check(_L().agp_conv_w_q8_prepare(ptr(self.w), self.cout, self.cin, ptr(plane), C.byref(e), _lib.stream()),
                      "agp_conv_w_q8_prepare")
                pl = (plane, int(e.value))
            self._planes["q8"] = pl
        return self._planes["q8"]

    @property
    def w_hi(self):
        return self.planes(_lib.PREC_BF16X3)[0]

    @property
    def w_lo(self):
        return self.planes(_lib.PREC_BF16X3)[1]


def fold_bn(bn_weight, bn_bias, mean, var, eps, conv_bias=None):
    """Eval-mode BatchNorm (and an optional conv bias) as per-channel scale/shift (fp64 math)."""
    s = bn_weight.double() / torch.sqrt(var.double() + eps)
    t = bn_bias.double() - mean.double() * s
    if conv_bias is not None:
        t = t + conv_bias.double() * s
    return s.float(), t.float()


def conv_desc(x: SplitMap, cw: ConvWeights, out: SplitMap, prec):
    """Geometry-only agp_conv_desc (no pointers): for the size queries of the C ABI."""
    d = _lib.ConvDesc()
    d.n, d.hin, d.win, d.pin = x.n, x.h, x.w, x.pad
    d.cin = cw.cin
    d.in_w_step = cw.in_w_step_stem or cw.cin
    d.hout, d.wout, d.cout, d.pout = out.h, out.w, cw.cout, out.pad
    d.kh, d.kw, d.stride, d.pad = cw.kh, cw.kw, cw.stride, cw.pad
    d.prec = prec
    return d


def conv_stat_tiles(x: SplitMap, cw: ConvWeights, out: SplitMap, prec, hi_only=False):
    """Row tiles of the per-tile channel statistics the conv kernel can emit for this conv (0 = it cannot); hi_only: of the
    one-product form (agp_conv_desc.hi_only: other tiles)."""
    d = conv_desc(x, cw, out, prec)
    if hi_only:
        d.hi_only = 1
    return int(_L().agp_conv2d_stat_tiles(C.byref(d)))


class PoolReq:
    """Global pooling of a conv's output map requested WITH the conv (reference: GeM / adaptive_avg_pool2d of a stage output,
    network_mm/image_pooling.py:16, fuse_block_toshallow.py:82, stage2fuse_blockadd.py:201-206): the 3x3 kernel of the fp16
    path reduces the values in its epilogue (agp_conv_desc::pool_partial) and agp_pool_from_conv finishes the sums, so no
    pass re-reads the map; convs that kernel does not run are pooled by agp_pool_fwd after the launch.  After
    ops.conv2d / ops.conv2d_grouped: `.mean` [n,c] and / or `.gem` [n,c]."""
    __slots__ = ("p", "eps", "want_mean", "want_gem", "mean", "gem", "fused", "_partial", "gem_out")

    def __init__(self, p=None, eps=GEM_EPS, want_mean=True, want_gem=False, gem_out=None):
        """gem_out: optional preallocated fp32 [n, c] target of the GeM vector (a row slice of a whole-batch buffer)."""
        if want_gem and p is None:
            raise ValueError("PoolReq: GeM needs its exponent tensor p")
        self.p = None if p is None else p.detach()
        self.eps, self.want_mean, self.want_gem = eps, want_mean, want_gem
        self.mean = self.gem = self._partial = None
        self.gem_out = gem_out
        self.fused = False

    def attach(self, d, x, cw, out, prec):
        """Before the launch: point the descriptor at a partial buffer if the kernel can pool."""
        self.fused = False
        if x.lo is None and out.lo is None:
            blocks = int(_L().agp_conv2d_pool_blocks(C.byref(d)))
            if blocks > 0:
                self._partial = torch.empty(blocks * 2 * cw.cout, dtype=torch.float32, device=out.hi.device)
                d.pool_partial = ptr(self._partial)
                d.pool_p = ptr(self.p) if self.want_gem else None
                d.pool_eps = self.eps
                self.fused = True

    def finish(self, out):
        """After the launch (same stream)."""
        go = self.gem_out
        if go is not None and (tuple(go.shape) != (out.n, out.c) or go.dtype != torch.float32 or not go.is_contiguous()):
            go = None
        if not self.fused:
            self.mean, self.gem = pool_map(out, self.p, want_mean=self.want_mean, want_gem=self.want_gem, eps=self.eps, gem_out=go)
            return
        dev = out.hi.device
        self.mean = torch.empty((out.n, out.c), dtype=torch.float32, device=dev) if self.want_mean else None
        self.gem = (go if go is not None else torch.empty((out.n, out.c), dtype=torch.float32, device=dev)) if self.want_gem else None
        check(_L().agp_pool_from_conv(ptr(self._partial), out.n, out.h, out.w, out.c, ptr(self.p) if self.want_gem else None,
                                      ptr(self.mean), ptr(self.gem), _lib.stream()), "agp_pool_from_conv")


class SqStatReq:
    """Per-channel sum and sum of squares of a conv's output requested WITH the conv (agp_conv_desc::pool_stat = 1): the first
    stage of a train-mode BatchNorm's statistics, from the epilogue of the fp16 3x3 stride-1 kernels (the training graph's
    one-product forward convs).  After ops.conv2d: `.fused` says whether the kernel did it; then `.partial` is
    [blocks][2][cout] fp32 and `.blocks` the number of written 64-row blocks (agp_bn_stats_from_partial finishes them)."""
    __slots__ = ("fused", "partial", "blocks")

    def __init__(self):
        self.fused, self.partial, self.blocks = False, None, 0

    def attach(self, d, x, cw, out, prec):
        self.fused = False
        if x.lo is None and out.lo is None:
            blocks = int(_L().agp_conv2d_pool_blocks(C.byref(d)))
            if blocks > 0:
                self.partial = torch.empty(blocks * 2 * cw.cout, dtype=torch.float32, device=out.hi.device)
                d.pool_partial = ptr(self.partial)
                d.pool_stat = 1
                self.blocks = out.n * ((out.h * (out.w + 2) + 63) // 64)
                assert self.blocks <= blocks
                self.fused = True

    def finish(self, out):
        pass


def _fill_conv_desc(d, x, cw, out, residual, relu, prec, stat_partial=None, bstat=None):
    d.in_hi, d.in_lo = ptr(x.hi), ptr(x.lo)
    w_hi, w_lo = cw.planes(prec)
    d.w_hi, d.w_lo = ptr(w_hi), ptr(w_lo)
    d.out_hi, d.out_lo = ptr(out.hi), ptr(out.lo)
    d.res_hi = ptr(residual.hi) if residual is not None else None
    d.res_lo = ptr(residual.lo) if residual is not None else None
    d.scale, d.shift = ptr(cw.scale), ptr(cw.shift)
    d.n, d.hin, d.win, d.pin = x.n, x.h, x.w, x.pad
    d.cin = cw.cin
    d.in_w_step = cw.in_w_step_stem or cw.cin
    d.hout, d.wout, d.cout, d.pout = out.h, out.w, cw.cout, out.pad
    d.kh, d.kw, d.stride, d.pad = cw.kh, cw.kw, cw.stride, cw.pad
    d.relu = 1 if relu else 0
    d.prec = prec
    if stat_partial is not None:
        d.stat_partial = ptr(stat_partial)
        if bstat is not None:          # backward-statistics mode (agp_conv_desc.bstat_*): (z, y or None, mean, rstd) of the consumer unit
            bz, by, bmean, brstd = bstat
            d.bstat_z_hi, d.bstat_z_lo = ptr(bz.hi), ptr(bz.lo)
            d.bstat_y_hi = ptr(by.hi) if by is not None else None
            d.bstat_mean, d.bstat_rstd = ptr(bmean), ptr(brstd)
    if not USE_W_CM and not getattr(cw, "_train_cm", False):
        pass
    elif prec == _lib.PREC_F16:
        d.w_cm = ptr(cw.cm())
    elif prec in (_lib.PREC_F16W2, _lib.PREC_BF16X3):
        if getattr(cw, "_train_cm", False):       # training planes written chunk-major by the split kernel itself
            d.w_cm, d.w_cm_lo = d.w_hi, d.w_lo
        else:
            c2 = cw.cm2(prec)
            if c2 is not None:
                d.w_cm, d.w_cm_lo = ptr(c2[0]), ptr(c2[1])
    if LO_FP8 and prec == _lib.PREC_F16W2:
        q = cw.q8()
        if q is not None:
            d.w_q8, d.w_q8_exp = ptr(q[0]), q[1]
    return d


def conv2d_grouped(jobs, prec):
    """jobs: [(x, cw, out, residual, relu[, pool]), ...] -- convolutions of one layer shape issued as ONE launch
    (agp_conv2d_fwd_grouped: AGP_PREC_F16 3x3 stride-1 convs sharing cin / cout; anything else runs as separate
    launches inside the library); `pool`: an optional PoolReq filled with the global pooling of that job's output.
    Returns the output maps."""
    jobs = [tuple(j) + (None,) * (6 - len(j)) for j in jobs]
    if len(jobs) == 1:
        x, cw, out, residual, relu, pool = jobs[0]
        return [conv2d(x, cw, out, residual=residual, relu=relu, prec=prec, pool=pool)]
    arr = (_lib.ConvDesc * len(jobs))()
    keep = []
    for d, (x, cw, out, residual, relu, pool) in zip(arr, jobs):
        keep.append(cw.planes(prec))
        _fill_conv_desc(d, x, cw, out, residual, relu, prec)
        if pool is not None:
            pool.attach(d, x, cw, out, prec)
    e0 = e1 = None
    if CONV_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().agp_conv2d_fwd_grouped(arr, len(jobs), _lib.stream()), "agp_conv2d_fwd_grouped")
    if CONV_PROFILE is not None:
        e1.record()
        x, cw, out = jobs[0][:3]
        macs = sum(j[0].n * j[2].h * j[2].w * j[1].cout * j[1].alg_k for j in jobs)
        CONV_PROFILE.append((e0, e1, macs, (sum(j[0].n for j in jobs), out.h, out.w, cw.cin, cw.cout, cw.kh, cw.kw, cw.stride)))
    for j in jobs:
        if j[5] is not None:
            j[5].finish(j[2])
    return [j[2] for j in jobs]


def conv2d(x: SplitMap, cw: ConvWeights, out: SplitMap, residual: SplitMap = None, relu=False, prec=3, stat_partial=None, pool=None,
           bstat=None, hi_only=False):
    """hi_only (prec 3, 3x3 stride-1 convs on 1-pixel-halo maps): ONE bf16 product on the hi planes (agp_conv_desc.hi_only)."""
    d = _fill_conv_desc(_lib.ConvDesc(), x, cw, out, residual, relu, prec, stat_partial, bstat)
    if hi_only:
        d.hi_only = 1
    if pool is not None:
        pool.attach(d, x, cw, out, prec)
    if CONV_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_L().agp_conv2d_fwd(C.byref(d), _lib.stream()), "agp_conv2d_fwd")
        e1.record()
        CONV_PROFILE.append((e0, e1, x.n * out.h * out.w * cw.cout * cw.alg_k,
                             (x.n, out.h, out.w, cw.cin, cw.cout, cw.kh, cw.kw, cw.stride)))
    else:
        check(_L().agp_conv2d_fwd(C.byref(d), _lib.stream()), "agp_conv2d_fwd")
    if pool is not None:
        pool.finish(out)
    return out


FUSED_BLOCK64 = True


def bblock64_ok(x: SplitMap, cw1: ConvWeights, cw2: ConvWeights, prec):
    """Can csrc/fblock64.hip run this BasicBlock?  fp16 single-product arithmetic, 64 -> 64 -> 64 channels, 3x3 / stride 1 /
    pad 1 twice, 1-pixel-halo maps of even height below 2 GiB."""
    if not FUSED_BLOCK64 or prec != _lib.PREC_F16 or x.lo is not None or x.pad != 1 or x.c != 64 or x.h % 2:
        return False
    for cw in (cw1, cw2):
        if (cw.cin, cw.cout, cw.kh, cw.kw, cw.stride, cw.pad) != (64, 64, 3, 3, 1, 1) or cw.in_w_step_stem or cw.w is None:
            return False
    return x.n * (x.h + 2) * (x.w + 2) * 128 < (1 << 31)


def bblock64_grouped(jobs, exact=False):
    """jobs: [(x, cw1, cw2, out[, pool]), ...] -- BasicBlocks on 64-channel fp16 maps (out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x),
    BatchNorm folded into the ConvWeights) of up to four trunks as ONE launch of the fused kernel (agp_bblock64_fwd_grouped: the
    intermediate map stays in LDS).  pool: optional PoolReq (mean only) filled with the channel means of `out`.
    exact=True: the 32x32x16 MFMA form, bit-identical to conv2d(x, cw1, relu) followed by conv2d(., cw2, residual=x, relu) at
    AGP_PREC_F16; the default 16x16x32 form differs from it by the fp32 rounding of the accumulation order."""
    jobs = [tuple(j) + (None,) * (5 - len(j)) for j in jobs]
    arr = (_lib.BBlock64Desc * len(jobs))()
    keep = []
    for d, (x, cw1, cw2, out, pool) in zip(arr, jobs):
        if (out.n, out.h, out.w, out.c, out.pad) != (x.n, x.h, x.w, 64, 1) or out.lo is not None:
            raise ValueError("bblock64: the output map must have the input's geometry (fp16, halo 1, 64 channels)")
        w1, w2 = cw1.planes(_lib.PREC_F16)[0], cw2.planes(_lib.PREC_F16)[0]
        keep.append((w1, w2))
        d.inp, d.out, d.w1, d.w2 = ptr(x.hi), ptr(out.hi), ptr(w1), ptr(w2)
        d.scale1, d.shift1, d.scale2, d.shift2 = ptr(cw1.scale), ptr(cw1.shift), ptr(cw2.scale), ptr(cw2.shift)
        d.n, d.h, d.w, d.form = x.n, x.h, x.w, 1 if exact else 0
        if pool is not None:
            if pool.want_gem:
                raise ValueError("bblock64: the fused block pools the mean only")
            nfl = int(_L().agp_bblock64_pool_floats(C.byref(d)))
            pool._partial = torch.empty(nfl, dtype=torch.float32, device=out.hi.device)
            d.pool_partial = ptr(pool._partial)
    e0 = e1 = None
    if CONV_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().agp_bblock64_fwd_grouped(arr, len(jobs), _lib.stream()), "agp_bblock64_fwd_grouped")
    if CONV_PROFILE is not None:
        e1.record()
        macs = sum(2 * j[0].n * j[0].h * j[0].w * 64 * 576 for j in jobs)
        x = jobs[0][0]
        CONV_PROFILE.append((e0, e1, macs, (sum(j[0].n for j in jobs), x.h, x.w, 64, 64, 3, 3, 1, "bblock64")))
    for x, cw1, cw2, out, pool in jobs:
        if pool is not None:
            pool.fused = True
            pool.mean = torch.empty((out.n, 64), dtype=torch.float32, device=out.hi.device)
            pool.gem = None
            check(_L().agp_bblock64_pool_finish(ptr(pool._partial), out.n, out.h, out.w, ptr(pool.mean), _lib.stream()),
                  "agp_bblock64_pool_finish")
    return [j[3] for j in jobs]


def stem_pool(x: SplitMap, cw: ConvWeights, out: SplitMap, prec=2):
    """The ResNet stem in one launch: packed 7x7/2 conv + folded BN + ReLU + MaxPool2d(3, 2, 1); `out` is the POOLED
    map.  fp16 maps only (prec 2 / 4)."""
    d = _lib.ConvDesc()
    d.in_hi, d.in_lo = ptr(x.hi), ptr(x.lo)
    w_hi, w_lo = cw.planes(prec)
    d.w_hi, d.w_lo = ptr(w_hi), ptr(w_lo)
    d.out_hi, d.out_lo = ptr(out.hi), ptr(out.lo)
    d.res_hi = d.res_lo = None
    d.scale, d.shift = ptr(cw.scale), ptr(cw.shift)
    d.n, d.hin, d.win, d.pin = x.n, x.h, x.w, x.pad
    d.cin, d.in_w_step = cw.cin, cw.in_w_step_stem
    d.hout, d.wout, d.cout, d.pout = out.h, out.w, cw.cout, out.pad
    d.kh, d.kw, d.stride, d.pad = cw.kh, cw.kw, cw.stride, cw.pad
    d.relu, d.prec = 1, prec
    check(_L().agp_stem_pool_fwd(C.byref(d), _lib.stream()), "agp_stem_pool_fwd")
    return out


def stem_walk_reads(x):
    """Can the walking stem kernel (csrc/stem_walk.hip, stem_walk_kernel<1> / <2>) fetch this input itself?  An fp32 [n, 3, h, w]
    image with unit column stride and 16-byte aligned base / strides / width, or contiguous uint8 camera tiles whose width is a
    multiple of 32, addressable by 31-bit byte offsets (the C side's agp_internal_stem_walk_reads / _reads_u8; other inputs take
    the packing pass or the per-block raw kernel)."""
    if torch.is_tensor(x) and x.dtype == torch.uint8 and x.dim() == 5 and x.shape[-1] == 3:
        # uint8 camera tiles [n, ncam, h, wcam, 3] (stem_walk_kernel<2>): contiguous, tile width a multiple of 32, 16-byte aligned
        n, ncam, h, wcam, _ = x.shape
        return x.is_contiguous() and wcam % 32 == 0 and x.data_ptr() % 16 == 0 and n * ncam * h * wcam * 3 < (1 << 31)
    if not torch.is_tensor(x) or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
        return False
    sn, sc, sh, sw = x.stride()
    n, _, h, w = x.shape
    if sw != 1 or sn % 4 or sc % 4 or sh % 4 or w % 4 or x.data_ptr() % 16 or min(sn, sc, sh) < 0:
        return False
    return ((n - 1) * sn + 2 * sc + (h - 1) * sh + w) * 4 < (1 << 31)


def stem_pool_raw(x, cw: ConvWeights, out: SplitMap, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """stem_pool reading the network's input itself (agp_stem_pool_raw_fwd, AGP_PREC_F16 maps): x is the fp32 image batch
    [n, 3, h, w] (any strides) or the uint8 camera tiles [n, ncam, h, wcam, 3]; no packed NHWC4 copy is made."""
    _need_cuda(x, "stem_pool_raw")
    d = _lib.ConvDesc()
    w_hi, _ = cw.planes(4)
    d.w_hi, d.w_lo = ptr(w_hi), None
    d.out_hi, d.out_lo = ptr(out.hi), None
    d.res_hi = d.res_lo = None
    d.in_lo = None
    d.scale, d.shift = ptr(cw.scale), ptr(cw.shift)
    m = (C.c_float * 3)(*mean)
    sd = (C.c_float * 3)(*std)
    if x.dtype == torch.uint8:
        if x.dim() != 5 or x.shape[-1] != 3:
            raise ValueError("stem_pool_raw expects uint8 [n, ncam, h, wcam, 3]")
        x = x.contiguous()
        n, ncam, h, wcam, _ = x.shape
        kind, w, st = 2, ncam * wcam, (0, 0, 0, 0)
    else:
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("stem_pool_raw expects fp32 [n, 3, h, w]")
        n, _, h, w = x.shape
        kind, ncam, st = 1, 1, tuple(x.stride())
    d.in_hi = ptr(x)
    d.n, d.hin, d.win, d.pin = n, h, w, 3
    d.cin, d.in_w_step = cw.cin, cw.in_w_step_stem
    d.hout, d.wout, d.cout, d.pout = out.h, out.w, cw.cout, out.pad
    d.kh, d.kw, d.stride, d.pad = cw.kh, cw.kw, cw.stride, cw.pad
    d.relu, d.prec = 1, 4
    check(_L().agp_stem_pool_raw_fwd(C.byref(d), kind, st[0], st[1], st[2], st[3], ncam, m, sd, _lib.stream()),
          "agp_stem_pool_raw_fwd")
    return out


def conv_out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def maxpool3x3s2(x: SplitMap, out: SplitMap, argmax=None):
    """argmax: optional uint8 [n, ho, wo, c] tensor receiving the first-maximum window positions (training)."""
    check(_L().agp_maxpool3x3s2_fwd(ptr(x.hi), ptr(x.lo), x.n, x.h, x.w, x.c, x.pad, ptr(out.hi),
                                    ptr(out.lo), out.h, out.w, out.pad, ptr(argmax), _lib.stream()),
          "agp_maxpool3x3s2_fwd")
    return out


def bcast_add(x: SplitMap, vec, out: SplitMap):
    """out = x + vec[n, :, None, None]; vec [n, c] (any strides: ops.linear hands out column slices of a padded buffer)."""
    if tuple(vec.shape) != (x.n, x.c) or vec.device != x.hi.device:
        raise RuntimeError(f"bcast_add: vector of shape {tuple(vec.shape)} on {vec.device} for a map [{x.n}, {x.c}, ...] on {x.hi.device}")
    vec = vec.contiguous().float()
    check(_L().agp_bcast_add_fwd(ptr(x.hi), ptr(x.lo), ptr(vec), x.n, x.h, x.w, x.c, x.pad,
                                 ptr(out.hi), ptr(out.lo), out.pad, _lib.stream()), "agp_bcast_add_fwd")
    return out


# ------------------------------------------------------------------------- pooling
def pool_map(x: SplitMap, p=None, want_mean=True, want_gem=True, eps=GEM_EPS, gem_out=None):
    """(mean [n,c] or None, gem [n,c] or None) of a SplitMap in one pass."""
    dev = x.hi.device
    nfl = _L().agp_pool_workspace_floats(x.n, x.c, x.h, x.w)
    partial = torch.empty(nfl, dtype=torch.float32, device=dev)
    mean = torch.empty((x.n, x.c), dtype=torch.float32, device=dev) if want_mean else None
    gem = (gem_out if gem_out is not None else torch.empty((x.n, x.c), dtype=torch.float32, device=dev)) if want_gem else None
    check(_L().agp_pool_fwd(ptr(x.hi), ptr(x.lo), x.n, x.h, x.w, x.c, x.pad, ptr(p) if want_gem else None,
                            eps, ptr(mean), ptr(gem), ptr(partial), _lib.stream()), "agp_pool_fwd")
    return mean, gem


def pool_f32(x, p=None, want_mean=False, want_gem=True, eps=GEM_EPS):
    """Same reductions on a dense fp32 [n,c,h,w] tensor (any strides)."""
    _need_cuda(x, "pool_f32")
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    mean = torch.empty((n, c), dtype=torch.float32, device=x.device) if want_mean else None
    gem = torch.empty((n, c), dtype=torch.float32, device=x.device) if want_gem else None
    check(_L().agp_pool_f32_fwd(ptr(x), sn, sc, sh, sw, n, c, h, w, ptr(p) if want_gem else None, eps,
                                ptr(mean), ptr(gem), None, _lib.stream()), "agp_pool_f32_fwd")
    return mean, gem


def new_gp(device):
    """The dL/dp buffer of the GeM backward entries (agp_gem_f32_bwd, agp_pool_bwd, agp_seg_pool_bwd): AGP_GP_FLOATS zeroed
    floats -- element 0 receives the gradient, the rest is the scratch of its fixed-order sum over the grid."""
    return torch.zeros(_lib.GP_FLOATS, dtype=torch.float32, device=device)


def gem_f32_bwd(x, p, y, gy, need_gx=True, eps=GEM_EPS):
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    gx = torch.empty_strided(x.shape, x.stride(), dtype=torch.float32, device=x.device) if need_gx else None
    gp = new_gp(x.device)
    check(_L().agp_gem_f32_bwd(ptr(x), sn, sc, sh, sw, n, c, h, w, ptr(p), eps, ptr(y), ptr(gy),
                               ptr(gx), ptr(gp), _lib.stream()), "agp_gem_f32_bwd")
    return gx, gp[:1]


# ------------------------------------------------------------------ MLP / ODE ops
class LinearWeights:
    """[n][k] split planes (+ fp32 bias); n padded up to a multiple of 256 with zero rows."""
    __slots__ = ("w_hi", "w_lo", "wt_hi", "wt_lo", "bias", "n", "k", "npad", "_frag")

    def __init__(self, weight, bias, with_transpose=False):
        w = weight.detach().float()
        n, k = w.shape
        self.n, self.k = n, k
        self.npad = (n + 255) // 256 * 256
        if self.npad != n:
            w = torch.cat([w, torch.zeros(self.npad - n, k, device=w.device)], 0)
        self.w_hi, self.w_lo = split_weight(w)
        self.wt_hi = self.wt_lo = None
        self._frag = None
        if with_transpose:
            # W^T as [k padded to 256][n padded to 32] for the backward products gz W
            kp, np32 = (k + 255) // 256 * 256, (n + 31) // 32 * 32
            wt = torch.zeros((kp, np32), dtype=torch.float32, device=w.device)
            wt[:k, :n] = weight.detach().float().t()
            self.wt_hi, self.wt_lo = split_weight(wt)
        if bias is None:
            self.bias = None
        else:
            b = bias.detach().float()
            if self.npad != n:
                b = torch.cat([b, torch.zeros(self.npad - n, device=b.device)], 0)
            self.bias = b.contiguous()


def linear_fragment_planes(lw: LinearWeights):
    """The planes of a [256, k] Linear in the FRAGMENT-MAJOR order the vector programs read (agp_vecprog_run): for wave w
    (16 output features), K-step ks (32 inputs) the 64 lanes' 16-byte MFMA A-fragments are contiguous --
    [w][ks][q = lane >> 4][row = lane & 15][8] = W[16 w + row][32 ks + 8 q .. + 7] -- so one wave instruction reads 1 KB of
    consecutive bytes (8 whole lines) instead of 16 half-lines 512 bytes apart.  Built once per weight version."""
    if lw._frag is None:
        if lw.npad != 256 or lw.k % 32:
            raise ValueError("fragment-major planes: a [256, k % 32 == 0] Linear")
        nks = lw.k // 32

        def frag(t):
            return t.view(16, 16, nks, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
        lw._frag = (frag(lw.w_hi), frag(lw.w_lo))
    return lw._frag


def _vec_operand(t, like, what):
    """An optional [b,k] operand the kernels read through a raw pointer: same shape and device as `like`, dense fp32."""
    if t is None:
        return None
    if t.shape != like.shape or t.device != like.device:
        raise RuntimeError(f"{what}: operand of shape {tuple(t.shape)} on {t.device}, expected {tuple(like.shape)} on {like.device}")
    return t.contiguous().float()


def linear(x, lw: LinearWeights, act=None, add1=None, add2=None):
    """act((x + add1 + add2) W^T + b) on [b,k] fp32."""
    _need_cuda(x, "linear")
    x = x.contiguous().float()
    b, k = x.shape
    if k != lw.k:
        raise RuntimeError(f"linear: expected k={lw.k}, got {k}")
    add1, add2 = _vec_operand(add1, x, "linear add1"), _vec_operand(add2, x, "linear add2")
    y = torch.empty((b, lw.npad), dtype=torch.float32, device=x.device)
    check(_L().agp_linear_fwd(ptr(x), ptr(add1), ptr(add2), ptr(lw.w_hi), ptr(lw.w_lo), ptr(lw.bias),
                              b, k, lw.npad, _lib.ACT[act], ptr(y), _lib.stream()), "agp_linear_fwd")
    return y if lw.npad == lw.n else y[:, :lw.n]


def ode_grid_dts(step_size):
    """torchdiffeq's fixed grid for t=[0,1] in fp32 (see agplace_amd/network_mm/ffns.py)."""
    t0 = torch.tensor(0.0, dtype=torch.float32)
    t1 = torch.tensor(1.0, dtype=torch.float32)
    niters = int(torch.ceil((t1 - t0) / step_size + 1).item())
    grid = torch.arange(0, niters, dtype=torch.float32) * step_size + t0
    grid[-1] = t1
    return (grid[1:] - grid[:-1]).tolist()


def fcode(x, lw: LinearWeights, act, method, dts, add1=None, add2=None, want_traj=False):
    _need_cuda(x, "fcode")
    x = x.contiguous().float()
    b, d = x.shape
    add1, add2 = _vec_operand(add1, x, "fcode add1"), _vec_operand(add2, x, "fcode add2")
    if d != 256 or lw.k != 256 or lw.n != 256:
        raise NotImplementedError("FCODE kernel is built for dim=256 (reference mm_stg2fuse_dim default)")
    if method not in _lib.ODE:
        raise NotImplementedError(method)
    if act not in _lib.ACT:
        raise NotImplementedError(act)
    n = len(dts)
    arr = (C.c_float * n)(*dts)
    y = torch.empty_like(x)
    traj = None
    if want_traj:
        traj = torch.empty(_L().agp_fcode_traj_floats(b, _lib.ODE[method], n), dtype=torch.float32, device=x.device)
    check(_L().agp_fcode_fwd(ptr(x), ptr(add1), ptr(add2), ptr(lw.w_hi), ptr(lw.w_lo), ptr(lw.bias), b,
                             _lib.ACT[act], _lib.ODE[method], arr, n, ptr(y), ptr(traj), _lib.stream()),
          "agp_fcode_fwd")
    return (y, traj) if want_traj else y


def fcode_bwd(traj, gy, lw: LinearWeights, act, method, dts):
    """(gx, gw, gb) of FCODE given the trajectory recorded by fcode(..., want_traj=True)."""
    gy = gy.contiguous().float()
    b, d = gy.shape
    n = len(dts)
    arr = (C.c_float * n)(*dts)
    gx = torch.empty_like(gy)
    gw = torch.empty((d, d), dtype=torch.float32, device=gy.device)
    gb = torch.empty(d, dtype=torch.float32, device=gy.device)
    nbytes = _L().agp_fcode_bwd_workspace_bytes(b, _lib.ODE[method], n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gy.device)
    check(_L().agp_fcode_bwd(ptr(traj), ptr(gy), ptr(lw.wt_hi), ptr(lw.wt_lo), b, _lib.ACT[act], _lib.ODE[method],
                             arr, n, ptr(gx), ptr(gw), ptr(gb), ptr(ws), nbytes, _lib.stream()), "agp_fcode_bwd")
    return gx, gw, gb


def linear_bwd(x, y, gy, lw: LinearWeights, act=None, need_gx=True, need_gw=True, need_gb=True):
    """Backward of y = act(x W^T + b): (gx [b,k], gw [n,k], gb [n]); lw must hold W^T planes."""
    gy = gy.contiguous().float()
    y = None if y is None else y.contiguous()
    x = None if x is None else x.contiguous()
    b, n = gy.shape
    k = lw.k
    kp = (k + 255) // 256 * 256
    gx = torch.empty((b, kp), dtype=torch.float32, device=gy.device) if need_gx else None
    gw = torch.empty((n, k), dtype=torch.float32, device=gy.device) if need_gw else None
    gb = torch.empty(n, dtype=torch.float32, device=gy.device) if need_gb else None
    nbytes = _L().agp_linear_bwd_workspace_bytes(b, k, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gy.device)
    check(_L().agp_linear_bwd(ptr(x), ptr(y), ptr(gy), ptr(lw.wt_hi), ptr(lw.wt_lo), b, k, n, _lib.ACT[act],
                              ptr(gx), ptr(gw), ptr(gb), ptr(ws), nbytes, _lib.stream()), "agp_linear_bwd")
    if need_gx and kp != k:
        gx = gx[:, :k]
    return gx, gw, gb


def layernorm_bwd(x, gamma, y, gy, eps=1e-5, relu=False, need_res=False):
    gy = gy.contiguous().float()
    b, d = gy.shape
    gx = torch.empty_like(gy)
    gres = torch.empty_like(gy) if need_res else None
    gg = torch.zeros(d, dtype=torch.float32, device=gy.device)
    gbeta = torch.zeros(d, dtype=torch.float32, device=gy.device)
    check(_L().agp_layernorm_bwd(ptr(x), ptr(gamma), ptr(y), ptr(gy), b, d, eps, 1 if relu else 0, ptr(gx),
                                 ptr(gres), ptr(gg), ptr(gbeta), _lib.stream()), "agp_layernorm_bwd")
    return gx, gres, gg, gbeta


def l2normalize_bwd(x, gy):
    gy = gy.contiguous().float()
    b, d = gy.shape
    gx = torch.empty_like(gy)
    check(_L().agp_l2normalize_bwd(ptr(x), ptr(gy), b, d, ptr(gx), _lib.stream()), "agp_l2normalize_bwd")
    return gx


def layernorm(x, gamma, beta, eps=1e-5, relu=False, residual=None):
    _need_cuda(x, "layernorm")
    x = x.contiguous().float()
    b, d = x.shape
    residual = _vec_operand(residual, x, "layernorm residual")
    y = torch.empty_like(x)
    check(_L().agp_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(residual), b, d, eps,
                                 1 if relu else 0, ptr(y), _lib.stream()), "agp_layernorm_fwd")
    return y


def l2normalize(x):
    _need_cuda(x, "l2normalize")
    x = x.contiguous()
    b, d = x.shape
    y = torch.empty_like(x)
    check(_L().agp_l2normalize_fwd(ptr(x), b, d, ptr(y), _lib.stream()), "agp_l2normalize_fwd")
    return y


def wsum(xs, ws=None):
    """sum_t ws[t] * xs[t] for up to 6 same-shape fp32 tensors; ws[t] are 1-element device tensors
    (None = 1.0)."""
    xs = [x.contiguous().float() for x in xs]
    _need_cuda(xs[0], "wsum")
    for x in xs[1:]:
        if x.shape != xs[0].shape or x.device != xs[0].device:
            raise RuntimeError("wsum: terms must share shape and device")
    if len(xs) > 6:
        head = wsum(xs[:5], None if ws is None else ws[:5])
        return wsum([head] + xs[5:], None if ws is None else [None] + list(ws[5:]))
    ws = [None] * len(xs) if ws is None else list(ws)
    for i, w in enumerate(ws):
        if w is not None:
            if w.numel() != 1 or w.device != xs[0].device:
                raise RuntimeError("wsum: a weight must be a 1-element tensor on the terms' device")
            ws[i] = w.detach().reshape(1).float().contiguous()
    y = torch.empty_like(xs[0])
    px = [ptr(x) for x in xs] + [None] * (6 - len(xs))
    pw = [ptr(w) for w in ws] + [None] * (6 - len(ws))
    check(_L().agp_wsum_fwd(*px, *pw, xs[0].numel(), ptr(y), _lib.stream()), "agp_wsum_fwd")
    return y


def dot(a, b):
    """1-element fp32 tensor sum(a * b) (kernel: agp_dot_f32)."""
    _need_cuda(a, "dot")
    a, b = a.contiguous().float(), b.contiguous().float()
    if a.numel() != b.numel():
        raise RuntimeError("dot: operands differ in size")
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(_L().agp_dot_f32(ptr(a), ptr(b), a.numel(), ptr(out), _lib.stream()), "agp_dot_f32")
    return out


NETVLAD_MFMA = True     # False: the fp32 VALU forward for every shape (one workgroup per image; what d not in {128, 256} takes anyway)


def _netvlad_fwd(x, conv_w, centroids, normalize_input):
    n, d, h, w = x.shape
    k = centroids.shape[0]
    out = torch.empty((n, k * d), dtype=torch.float32, device=x.device)
    nbytes = _L().agp_netvlad_workspace_bytes(n, d, h * w, k) if NETVLAD_MFMA and k <= 64 else 0
    if nbytes > 0:
        # the matrix-pipe forward (csrc/netvlad.hip, round 6): exact fp32 MFMAs, several workgroups per image
        wsp = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(_L().agp_netvlad_fwd_mfma(ptr(x), ptr(conv_w), ptr(centroids), n, d, h * w, k, 1 if normalize_input else 0, ptr(out),
                                        ptr(wsp), nbytes, _lib.stream()), "agp_netvlad_fwd_mfma")
        return out
    check(_L().agp_netvlad_fwd(ptr(x), ptr(conv_w), ptr(centroids), n, d, h * w, k,
                               1 if normalize_input else 0, ptr(out), _lib.stream()), "agp_netvlad_fwd")
    return out


class _NetVLADFn(torch.autograd.Function):
    """NetVLAD.forward with a hand-written backward (agp_netvlad_bwd: the image's statistics recomputed, then the chain of
    reference model/aggregation.py:126-146 walked backwards: dx, d conv.weight, d centroids)."""

    @staticmethod
    def forward(ctx, x, conv_w, centroids, normalize_input):
        ctx.save_for_backward(x, conv_w, centroids)
        ctx.normalize_input = normalize_input
        return _netvlad_fwd(x, conv_w, centroids, normalize_input)

    @staticmethod
    def backward(ctx, gout):
        x, conv_w, centroids = ctx.saved_tensors
        n, d, h, w = x.shape
        k = centroids.shape[0]
        dx, dw, dc = torch.empty_like(x), torch.empty_like(conv_w), torch.empty_like(centroids)
        wsp = torch.empty((3, n, k, d), dtype=torch.float32, device=x.device)
        check(_L().agp_netvlad_bwd(ptr(x), ptr(conv_w), ptr(centroids), ptr(gout.contiguous().float()), n, d, h * w, k,
                                   1 if ctx.normalize_input else 0, ptr(dx), ptr(dw), ptr(dc), ptr(wsp), _lib.stream()),
              "agp_netvlad_bwd")
        return dx, dw, dc, None


def netvlad(x, conv_w, centroids, normalize_input=True):
    """x [n, d, h, w] fp32, conv_w [k, d(, 1, 1)], centroids [k, d] -> [n, k * d]; differentiable in all three (d in 64 / 128 / 256
    for the backward)."""
    _need_cuda(x, "netvlad")
    x = x.contiguous().float()
    k, d = centroids.shape[0], x.shape[1]
    w2 = conv_w.reshape(k, d).contiguous().float()
    c2 = centroids.contiguous().float()
    if torch.is_grad_enabled() and (x.requires_grad or conv_w.requires_grad or centroids.requires_grad):
        return _NetVLADFn.apply(x, w2, c2, bool(normalize_input))
    return _netvlad_fwd(x, w2, c2, normalize_input)
