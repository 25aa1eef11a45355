"""Training-mode execution of conv -> BatchNorm -> (+residual) -> ReLU units on the HIP kernels.

The reference trains by plain autograd through cuDNN convolutions and train-mode BatchNorm
(train.py:337-341).  Here every unit records what its backward needs and the backward pass is run by
hand on halo-padded split-bf16 maps (ops.SplitMap):

    forward : z = conv(x)(+bias)            agp_conv2d_fwd   (raw, no BN folding in train mode)
              mean, rstd, scale, shift      agp_bn_stats     (batch statistics, running-stat update)
              y = relu?(z*scale+shift+res)  agp_map_affine
    backward: gz, gres, dgamma, dbeta       agp_bn_bwd
              dW                            agp_conv2d_wgrad (NHWC strips + LDS transpose reads, split-K)
              dx                            agp_conv2d_fwd   with flipped/transposed weights
                                            (stride 2: zero-upsampled gz, agp_upsample2_zero)

Parameter gradients are accumulated into `.grad` of the nn.Conv2d / nn.BatchNorm2d containers.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import ptr, check
from .ops import SplitMap

# train-mode conv + BatchNorm: take the batch statistics from the conv kernel's epilogue (agp_conv_desc.stat_partial)
# where that kernel can produce them, instead of a reduction pass over the conv output
FUSE_BN_STATS = True
# the BatchNorm backward's channel sums from the epilogue of the data-gradient conv that produces the gradient (and the residual
# branch's gradient added there): ConvBNUnit.backward(partial=, add=, stats_for=)
FUSE_BN_BWD = True
# the stem in training: BatchNorm apply + ReLU + max-pool as one pass, the full-size activation not stored (ConvBNUnit.forward(pool=))
FUSE_STEM_POOL = True
# the weight gradient of a 3x3 (stride 1 or 2) or 1x1 stride-2 conv as ONE fp16 MFMA product (agp_conv_desc.in_h16 / out_absmax; csrc/wgrad_tr.hip:
# wgrad_f16_kernel) instead of three bf16 ones: the conv's input keeps an fp16 operand plane (written by the pass that produces
# the map), the BatchNorm backward folds max |gz| per channel into the words the kernel takes its operand scale from.  Emulated
# per conv role in the fp64 oracle (tools/grad_prec_emul.py): 7e-4 on a weight gradient, inside the 1e-3 bar; the forward and
# the data gradient cannot drop a product (3e-3 / 1e-3 with thin margin) and stay at three.
WGRAD_F16 = True
# ... of the gather shapes (stride-2 3x3, 1x1 stride-2) and of the packed stem (A/B switches of tools/train_bench.py)
WGRAD_F16_GATHER = True
WGRAD_F16_STEM = True
# Options.train_precision = 16, the opt-in FAST training mode (set by MM.forward_q / DBVanilla2D.forward_db around their training
# forwards): the FORWARD of every 3x3 conv (stride 1: igemm_kxrw / igemm_kxr2; stride 2 and the 1x1 stride-2 downsample: the
# generic kernel) whose input keeps an fp16 operand plane runs as ONE fp16 x fp16 MFMA product on the inference kernels (fp16
# activations x fp16 weights, fp32 accumulate, the weights' fp32 masters untouched); its z is then ONE fp16 plane, the BatchNorm
# statistics the stride-1 kernels' epilogue sums (agp_conv_desc.pool_stat) or a pass of their own (fp64 finalisation as always).  The data
# gradient keeps three products (a gradient map has no fp16 range without a scale per tensor), the weight gradient its one.
# tools/grad_prec_emul.py prices this plan at 4.0-4.4 x the tight mode's 1e-3 gradient bar on a randomly initialised trunk
# (train-mode BatchNorm amplifies the forward's rounding layer by layer): tests/test_gpu_train.py measures what it is.
FWD_F16 = False
# Options.train_dgrad_products = 1 (opt-in, on top of either mode; set where FWD_F16 is): the DATA gradient of every 3x3 conv
# (stride 1, and the stride-2 entries' conv over the zero-upsampled gradient) as ONE bf16 product of the hi planes
# (agp_conv_desc.hi_only: gz.hi x W.hi, fp32 accumulate, the residual / statistics epilogue and the stored pair unchanged) -- bf16
# keeps fp32's range, so no scale; operands to 2^-9, i.e. about 4 x the forward's fp16 rounding per conv, accumulating down the trunk.
# Measured against the fp64 oracle by tests/test_gpu_train.py (FASTGRAD lines); timing: bench.py train.fast_mode.
DGRAD_HI_ONLY = False
# FWD_F16: a unit's output that ONLY a one-product conv consumes (conv1 -> conv2 of a BasicBlock) is stored as ONE fp16 plane -- the
# consumer's forward and weight gradient read that plane anyway and the BatchNorm backward takes its ReLU mask from it: 4 of the 6
# bytes per element the BatchNorm apply pass writes are never read.  Accuracy-neutral.  A/B switch of tools/train_bench.py
Y16_ONLY = True
# FWD_F16 also for the stage entries (3x3 stride 2, 1x1 stride-2 downsample): A/B switch of tools/train_bench.py
FWD_F16_ENTRIES = True


def _L():
    return _lib.load()


# Synchronised BatchNorm under data parallelism (optional; parallel.sync_batchnorm): when set, a torch.distributed process
# group (or True = the default group) over which every train-mode BatchNorm of the graph -- 2-D and sparse -- all-reduces its
# fp64 (sum, sum of squares, count) vector in forward and its (sum g, sum g*zhat) vector in backward: the statistics of the
# GLOBAL batch, as one GPU would compute them (reference model/sync_batchnorm/batchnorm.py:121-166; converted, but never
# activated, at train.py:253-256).  One small collective per layer and direction: latency, not bytes.
SYNC_BN = None


def _sync_group():
    """the process group of SYNC_BN, or None when BatchNorm statistics are per rank"""
    if SYNC_BN is None or SYNC_BN is False:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        return None
    return dist.group.WORLD if SYNC_BN is True else SYNC_BN


def _allreduce_sums(sums, group):
    import torch.distributed as dist
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("synchronised BatchNorm issues collectives: not capturable in a hipGraph")
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


def _stats_from_sums(sums, z, bn, update_running):
    dev = z.hi.device
    mean = torch.empty(z.c, dtype=torch.float32, device=dev)
    rstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    mom = _momentum(bn, update_running)
    check(_L().agp_bn_stats_from_sums(ptr(sums), z.c, bn.eps, mom, ptr(mean), ptr(rstd),
                                      ptr(bn.running_mean) if update_running else None,
                                      ptr(bn.running_var) if update_running else None,
                                      ptr(bn.weight), ptr(bn.bias), ptr(scale), ptr(shift), _lib.stream()),
          "agp_bn_stats_from_sums")
    if update_running and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    bn._agp_sync_count = sums[2 * z.c:]            # the global count, on the device: the backward's 1 / N
    return mean, rstd, scale, shift


# Subscribers (parallel.GradBuckets.mark_ready) told when a hand-orchestrated backward has written the FINAL gradients
# of a set of parameters as a side effect (they never pass through autograd's accumulation hooks).
GRAD_READY_CALLBACKS = []


# A parameter that several hand-orchestrated nodes write (one trunk applied to several inputs under opt.share_dbfe, the GeM
# shared by the stage-2 layers when opt.stg2nlayers > 1) is final only after the LAST of them: every node announces its
# parameters in forward (expect_grads) and reports them in backward; subscribers hear of a parameter when its count is 0.
_EXPECTED = {}


def expect_grads(params):
    for p in params:
        if p.requires_grad:
            _EXPECTED[id(p)] = _EXPECTED.get(id(p), 0) + 1


def reset_expected():
    """Start of a step (GradBuckets.zero_grad): forget forwards whose backward never ran."""
    _EXPECTED.clear()


def notify_grads_ready(params):
    done = []
    for p in params:
        if not p.requires_grad:
            continue
        c = _EXPECTED.get(id(p), 0)
        if c > 1:
            _EXPECTED[id(p)] = c - 1
            continue
        _EXPECTED.pop(id(p), None)
        done.append(p)
    if done:
        for cb in list(GRAD_READY_CALLBACKS):
            cb(done)


def _acc_grad(param, g):
    if not param.requires_grad:
        return
    if g.shape != param.shape:
        g = g.reshape(param.shape)
    g = g.to(param.dtype)
    if param.grad is None:
        # parameter layout (fused optimizers require it).  Every caller hands over a freshly allocated tensor it does
        # not touch again, so a gradient that already HAS the parameter's layout is adopted as is: no copy kernel
        # (one small launch per parameter on the backward's serial chain otherwise).
        if g.stride() == param.stride() and g.data_ptr() != param.data_ptr() and g._base is None:
            param.grad = g
        else:
            param.grad = torch.empty_like(param).copy_(g)
    else:
        param.grad += g


def _reduce_ws(m: SplitMap):
    nfl = _L().agp_train_reduce_workspace_floats(m.n, m.h, m.w, m.c)
    return torch.empty(nfl, dtype=torch.float32, device=m.hi.device)


def _momentum(bn, update_running):
    """nn.BatchNorm's update factor: `momentum`, or (momentum=None) the cumulative average 1 / num_batches_tracked.  The
    count lives on the device; a host mirror avoids a synchronisation per layer and step."""
    if bn.momentum is not None:
        return bn.momentum
    if not update_running:
        return 0.0
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("BatchNorm(momentum=None) changes its update factor every step: not capturable in a hipGraph")
    c = getattr(bn, "_agp_nbt", None)
    if c is None:
        c = int(bn.num_batches_tracked.item()) if bn.num_batches_tracked is not None else 0
    bn._agp_nbt = c + 1
    return 1.0 / (c + 1)


def bn_frozen(bn):
    """Eval-mode BatchNorm inside a gradient graph (fine-tuning on frozen statistics): (mean, rstd, scale, shift) from the
    running statistics, which are left untouched.  The backward then holds them constant (bn_bwd(frozen=True))."""
    if bn.running_mean is None or bn.running_var is None:
        raise NotImplementedError("eval-mode BatchNorm without running statistics (track_running_stats=False)")
    c, dev = bn.running_mean.numel(), bn.running_mean.device
    mean = torch.empty(c, dtype=torch.float32, device=dev)
    rstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    check(_L().agp_bn_frozen_coeffs(ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.weight), ptr(bn.bias), c, bn.eps,
                                    ptr(mean), ptr(rstd), ptr(scale), ptr(shift), _lib.stream()), "agp_bn_frozen_coeffs")
    return mean, rstd, scale, shift


def bn_stats(z: SplitMap, bn, update_running=True):
    if not bn.training:
        return bn_frozen(bn)
    dev = z.hi.device
    group = _sync_group()
    if group is not None:
        sums = torch.empty(2 * z.c + 1, dtype=torch.float64, device=dev)
        check(_L().agp_bn_sums(ptr(z.hi), ptr(z.lo), z.n, z.h, z.w, z.c, z.pad, ptr(sums), ptr(_reduce_ws(z)), _lib.stream()),
              "agp_bn_sums")
        return _stats_from_sums(_allreduce_sums(sums, group), z, bn, update_running)
    mean = torch.empty(z.c, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    scale = torch.empty_like(mean)
    shift = torch.empty_like(mean)
    mom = _momentum(bn, update_running)
    check(_L().agp_bn_stats(ptr(z.hi), ptr(z.lo), z.n, z.h, z.w, z.c, z.pad, bn.eps, mom, ptr(mean), ptr(rstd),
                            ptr(bn.running_mean) if update_running else None,
                            ptr(bn.running_var) if update_running else None,
                            ptr(bn.weight), ptr(bn.bias), ptr(scale), ptr(shift), ptr(_reduce_ws(z)), _lib.stream()),
          "agp_bn_stats")
    if update_running and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return mean, rstd, scale, shift


def bn_stats_from_partial(partial, tiles, z: SplitMap, bn, update_running=True):
    """bn_stats from the per-tile sums a conv wrote next to z (agp_conv_desc.stat_partial)."""
    if not bn.training:
        return bn_frozen(bn)
    dev = z.hi.device
    group = _sync_group()
    if group is not None:
        sums = torch.empty(2 * z.c + 1, dtype=torch.float64, device=dev)
        check(_L().agp_bn_sums_from_partial(ptr(partial), tiles, z.c, z.n * z.h * z.w, ptr(sums), _lib.stream()),
              "agp_bn_sums_from_partial")
        return _stats_from_sums(_allreduce_sums(sums, group), z, bn, update_running)
    mean = torch.empty(z.c, dtype=torch.float32, device=dev)
    rstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    mom = _momentum(bn, update_running)
    check(_L().agp_bn_stats_from_partial(ptr(partial), tiles, z.c, z.n * z.h * z.w, bn.eps, mom, ptr(mean), ptr(rstd),
                                         ptr(bn.running_mean) if update_running else None,
                                         ptr(bn.running_var) if update_running else None,
                                         ptr(bn.weight), ptr(bn.bias), ptr(scale), ptr(shift), _lib.stream()),
          "agp_bn_stats_from_partial")
    if update_running and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return mean, rstd, scale, shift


def map_affine(a: SplitMap, scale, shift, out: SplitMap, residual: SplitMap = None, relu=False):
    check(_L().agp_map_affine(ptr(a.hi), ptr(a.lo), ptr(scale), ptr(shift),
                              ptr(residual.hi) if residual is not None else None,
                              ptr(residual.lo) if residual is not None else None,
                              a.n, a.h, a.w, a.c, a.pad, 1 if relu else 0, ptr(out.hi), ptr(out.lo), ptr(out.h16), _lib.stream()),
          "agp_map_affine")
    return out


def bn_bwd(z, gy, y, mean, rstd, gamma, relu, gz, gres=None, frozen=False, sync_count=None, partial=None, absmax=None):
    """absmax: int32[c] that receives max |gz| per channel as fp32 bit patterns (integer atomic max; the one-pass weight gradient
    reads and re-zeroes it) -- not produced by the synchronised path (its caller then runs the three-product weight gradient).
    partial: (tensor [tiles][2][c], tiles) -- the channel sums already reduced per tile by the conv that produced gy
    (ConvBNUnit._dgrad with `stats_for`): no reduction pass.
    frozen: the forward used the running statistics (bn_frozen): they are constants of the backward.
    sync_count: the forward ran synchronised (bn._agp_sync_count, the global count on the device): the backward's sums are
    all-reduced too; ggamma / gbeta stay this rank's (averaged over ranks with every other gradient)."""
    dev = z.hi.device
    gg = torch.empty(z.c, dtype=torch.float32, device=dev)
    gb = torch.empty_like(gg)
    group = _sync_group() if (sync_count is not None and not frozen) else None
    if group is not None:
        yh, yl = (ptr(y.hi), ptr(y.lo)) if y is not None else (None, None)
        sums = torch.empty(2 * z.c, dtype=torch.float64, device=dev)
        wsr = _reduce_ws(z)
        check(_L().agp_bn_bwd_sums(ptr(z.hi), ptr(z.lo), ptr(gy.hi), ptr(gy.lo), yh, yl, ptr(mean), ptr(rstd), z.n, z.h, z.w, z.c,
                                   z.pad, 1 if relu else 0, ptr(sums), ptr(gg), ptr(gb), ptr(wsr), _lib.stream()), "agp_bn_bwd_sums")
        _allreduce_sums(sums, group)
        check(_L().agp_bn_bwd_apply(ptr(z.hi), ptr(z.lo), ptr(gy.hi), ptr(gy.lo), yh, yl, ptr(mean), ptr(rstd), ptr(gamma),
                                    ptr(sums), ptr(sync_count), z.n, z.h, z.w, z.c, z.pad, 1 if relu else 0, ptr(gz.hi), ptr(gz.lo),
                                    ptr(gres.hi) if gres is not None else None, ptr(gres.lo) if gres is not None else None,
                                    ptr(wsr), _lib.stream()), "agp_bn_bwd_apply")
        return gg, gb
    if partial is not None:
        check(_L().agp_bn_bwd_from_partial(ptr(partial[0]), partial[1], ptr(z.hi), ptr(z.lo), ptr(gy.hi), ptr(gy.lo),
                                           ptr(y.hi) if y is not None else None, ptr(y.lo) if y is not None else None,
                                           ptr(mean), ptr(rstd), ptr(gamma), z.n, z.h, z.w, z.c, z.pad, 1 if relu else 0,
                                           1 if frozen else 0, ptr(gz.hi), ptr(gz.lo),
                                           ptr(gres.hi) if gres is not None else None, ptr(gres.lo) if gres is not None else None,
                                           ptr(gg), ptr(gb), ptr(absmax), _lib.stream()), "agp_bn_bwd_from_partial")
        return gg, gb
    fn = _L().agp_bn_bwd_frozen if frozen else _L().agp_bn_bwd
    check(fn(ptr(z.hi), ptr(z.lo), ptr(gy.hi), ptr(gy.lo), ptr(y.hi) if y is not None else None,
                          ptr(y.lo) if y is not None else None, ptr(mean), ptr(rstd), ptr(gamma), z.n, z.h, z.w, z.c,
                          z.pad, 1 if relu else 0, ptr(gz.hi), ptr(gz.lo),
                          ptr(gres.hi) if gres is not None else None, ptr(gres.lo) if gres is not None else None,
                          ptr(gg), ptr(gb), ptr(_reduce_ws(z)), ptr(absmax), _lib.stream()), "agp_bn_bwd")
    return gg, gb


def chan_sum(a: SplitMap):
    out = torch.empty(a.c, dtype=torch.float32, device=a.hi.device)
    check(_L().agp_map_chan_sum(ptr(a.hi), ptr(a.lo), a.n, a.h, a.w, a.c, a.pad, ptr(out), ptr(_reduce_ws(a)),
                                _lib.stream()), "agp_map_chan_sum")
    return out


def map_add(a: SplitMap, b: SplitMap, out: SplitMap, mask: SplitMap = None):
    check(_L().agp_map_add(ptr(a.hi), ptr(a.lo), ptr(b.hi) if b is not None else None,
                           ptr(b.lo) if b is not None else None, ptr(mask.hi) if mask is not None else None,
                           ptr(mask.lo) if mask is not None else None, a.n, a.h, a.w, a.c, a.pad, ptr(out.hi),
                           ptr(out.lo), _lib.stream()), "agp_map_add")
    return out


def upsample2_zero(g: SplitMap, out: SplitMap):
    check(_L().agp_upsample2_zero(ptr(g.hi), ptr(g.lo), g.n, g.h, g.w, g.c, g.pad, ptr(out.hi), ptr(out.lo), out.h,
                                  out.w, out.pad, _lib.stream()), "agp_upsample2_zero")
    return out


def maxpool_bwd(argmax, gy: SplitMap, gx: SplitMap):
    """gx (the pool's input geometry) from the argmax recorded by ops.maxpool3x3s2(..., argmax=...)."""
    check(_L().agp_maxpool3x3s2_bwd(ptr(argmax), ptr(gy.hi), ptr(gy.lo), gx.n, gx.h, gx.w, gx.c, gx.pad, gy.h, gy.w, gy.pad,
                                    ptr(gx.hi), ptr(gx.lo), _lib.stream()), "agp_maxpool3x3s2_bwd")
    return gx


def maxpool_bn_bwd(argmax, gp: SplitMap, z, y, mean, rstd, gamma, relu, gz, frozen=False, scale=None, shift=None, pooled=None,
                   beta=None, absmax=None):
    """agp_maxpool_bn_bwd: BatchNorm backward of the unit UNDER a 3x3/2 max-pool straight from the pooled gradient `gp` (the
    gradient at the unit's output is not materialised).  Returns (ggamma, gbeta), or None when the library cannot (channel
    count): the caller then runs maxpool_bwd + bn_bwd."""
    gg = torch.empty(z.c, dtype=torch.float32, device=z.hi.device)
    gb = torch.empty_like(gg)
    rc = _L().agp_maxpool_bn_bwd(ptr(argmax), ptr(gp.hi), ptr(gp.lo), gp.h, gp.w, gp.pad, ptr(z.hi), ptr(z.lo),
                                 ptr(y.hi) if y is not None else None, ptr(y.lo) if y is not None else None, ptr(mean), ptr(rstd),
                                 ptr(gamma), ptr(scale), ptr(shift), ptr(pooled.hi) if pooled is not None else None,
                                 ptr(pooled.lo) if pooled is not None else None, ptr(beta), z.n, z.h, z.w, z.c, z.pad,
                                 1 if relu else 0, 1 if frozen else 0,
                                 ptr(gz.hi), ptr(gz.lo), ptr(gg), ptr(gb), ptr(_reduce_ws(z)), ptr(absmax), _lib.stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "agp_maxpool_bn_bwd")
    return gg, gb


def affine_maxpool(z: SplitMap, scale, shift, out: SplitMap, argmax):
    """out = MaxPool2d(3, 2, 1)(relu(z * scale + shift)) + argmax, one pass over z (agp_affine_maxpool3x3s2_fwd)."""
    check(_L().agp_affine_maxpool3x3s2_fwd(ptr(z.hi), ptr(z.lo), ptr(scale), ptr(shift), z.n, z.h, z.w, z.c, z.pad, ptr(out.hi),
                                           ptr(out.lo), out.h, out.w, out.pad, ptr(argmax), ptr(out.h16), _lib.stream()),
          "agp_affine_maxpool3x3s2_fwd")
    return out


def pool_bwd(x: SplitMap, out: SplitMap, gmean=None, ggem=None, gem_y=None, p=None, eps=1e-6, base: SplitMap = None,
             gp=None):
    """out = base? + gmean/HW + ggem * dGeM/dx  (gradient of agp_pool_fwd w.r.t. the map).
    gp: ops.new_gp(device) -- element 0 receives dL/dp of the GeM exponent."""
    check(_L().agp_pool_bwd(ptr(x.hi), ptr(x.lo), ptr(gmean), ptr(ggem), ptr(gem_y), ptr(p), eps,
                            ptr(base.hi) if base is not None else None, ptr(base.lo) if base is not None else None,
                            x.n, x.h, x.w, x.c, x.pad, ptr(out.hi), ptr(out.lo), ptr(gp), _lib.stream()), "agp_pool_bwd")
    return out


class ConvBNUnit:
    """One conv (+bias) -> BatchNorm2d(train) -> (+residual) -> (ReLU) with a hand-written backward."""

    def __init__(self, conv, bn, tag, ws: ops.Workspace, stem=False):
        self.conv, self.bn, self.tag, self.ws, self.stem = conv, bn, tag, ws, stem
        self.saved = None

    # ------------------------------------------------------------------ forward
    def can_skip_output(self, prec):
        """Whether forward(pool=...) may leave the full-size output unstored: the backward must be able to go through the pool
        itself (agp_maxpool_bn_bwd)."""
        synced = "_agp_sync_count" in self.bn.__dict__ and self.bn.training and _sync_group() is not None
        c = self.conv.out_channels
        return FUSE_BN_BWD and FUSE_STEM_POOL and prec == 3 and not synced and c % 8 == 0 and c // 8 <= 256 and 256 % (c // 8) == 0

    def output_map(self):
        """The unit's output y of its last forward; recomputed from z when forward(pool=...) did not store it."""
        z, y, prec = self.saved[1], self.saved[2], self.saved[7]
        if y is None:          # (only a ReLU unit's output is ever left unstored)
            y = map_affine(z, self._pool_coeffs[0], self._pool_coeffs[1], ops.SplitMap.alloc(z.n, z.h, z.w, z.c, 1, prec, z.hi.device),
                           relu=True)
        return y

    def wgrad_f16_ok(self, prec=3):
        """Whether this unit's weight gradient can run as one fp16 product, given an input map with an fp16 operand plane."""
        conv = self.conv
        if self.stem:                                    # the packed 7x7 / 2 stem (its input map: NHWC4 with an fp16 plane)
            return (WGRAD_F16 and WGRAD_F16_STEM and prec == 3 and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
                    and conv.out_channels % 64 == 0)
        # wgrad_f16_kernel (3x3 stride 1) takes cin % 64 == 0 (conv2d_wgrad_impl); the gather form (stride-2 entries, 1x1
        # downsamples) cin % 32 == 0.  A conv this says yes to MUST take a one-product path in the library: its producer then writes
        # the fp16 plane and the BatchNorm backward folds gradient maxima that only those kernels consume and re-zero (ADVICE r5:
        # cin = 32 / 96 stride-1 convs used to say yes here and fall through to the three-product kernel, leaving a stale maximum)
        s1 = conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.in_channels % 64 == 0
        gather = (WGRAD_F16_GATHER and conv.in_channels % 32 == 0
                  and ((conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1))
                       or (conv.kernel_size == (1, 1) and conv.stride == (2, 2) and conv.padding == (0, 0))))
        return WGRAD_F16 and prec == 3 and (s1 or gather) and conv.out_channels % 64 == 0

    def reads_f16_plane_only(self, prec=3):
        """Whether this unit, as the ONLY consumer of a map, needs nothing but the map's fp16 plane: its forward runs as one fp16
        product (the fast mode, 3x3 stride 1) and its weight gradient as the one-pass fp16 kernel; the data gradient reads no
        activation.  (Not under synchronised BatchNorm, whose backward runs the three-product weight gradient on the pair.)"""
        conv = self.conv
        return bool(FWD_F16 and Y16_ONLY and prec == 3 and not self.stem and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                    and conv.padding == (1, 1) and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0
                    and self.wgrad_f16_ok(prec) and conv.weight.requires_grad and _sync_group() is None)

    def forward(self, x: SplitMap, residual: SplitMap = None, relu=True, prec=3, out_hw=None, pool=None, out_h16=False,
                out_f16_only=False):
        """out_h16: the output also keeps an fp16 operand plane (SplitMap.h16) -- a consumer's weight gradient wants it
        (wgrad_f16_ok); written by the pass that writes the output.
        out_f16_only: the output's only consumer is a unit with reads_f16_plane_only(): y is stored as ONE fp16 plane.
        pool: (pooled map, argmax tensor) -- the unit is followed by MaxPool2d(3, 2, 1) (the stem) and only the pooled map is
        wanted: BatchNorm apply, ReLU and the pool run as one pass over z, y is not stored (returns the pooled map)."""
        conv, dev = self.conv, x.hi.device
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        if self.stem or prec != 3 or conv.in_channels % 8:
            cw = ops.ConvWeights(conv.weight, None, conv.bias, s, p, stem=self.stem)
        else:
            cw = ops.ConvWeights.for_training(conv.weight, conv.bias, s, p, plane_pixels=x.n * (x.h + 2 * x.pad) * (x.w + 2 * x.pad))
        hin, win = out_hw if out_hw is not None else (x.h, x.w)      # stem: logical image size
        ho, wo = ops.conv_out_size(hin, k, s, p), ops.conv_out_size(win, k, s, p)
        frozen = not self.bn.training          # eval-mode BatchNorm under autograd: running statistics, held constant
        # (3x3 stride 1: the inference kernels; the stage entries -- 3x3 stride 2 and the 1x1 stride-2 downsample -- the generic kernel)
        entry = FWD_F16_ENTRIES and ((k == 3 and s == 2 and p == 1) or (k == 1 and s == 2 and p == 0))
        fwd16_shape = (k == 3 and s == 1 and p == 1) or entry
        fwd16 = (FWD_F16 and prec == 3 and not self.stem and x.h16 is not None and fwd16_shape
                 and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0)
        # (not the stem: as one fp16 product it takes 0.1 ms off the step and adds a quarter to every gradient's error -- the first
        # layer's rounding passes through every BatchNorm behind it: median 3.1 -> 3.9e-3, worst 4.4 -> 5.5e-3)
        if fwd16:
            # the fast mode: x's fp16 operand plane x fp16 weights, one product, on the inference kernel; z = one fp16 plane
            x16 = SplitMap(x.h16, None, x.n, x.h, x.w, x.c, x.pad)
            cw16 = ops.ConvWeights(conv.weight, None, conv.bias, s, p)
            z = self.ws.map(self.tag + ".z16", x.n, ho, wo, cw.cout, 1, 4, dev)
            # ... whose epilogue also reduces z's channel sums and sums of squares (agp_conv_desc.pool_stat): no statistics pass
            sq = ops.SqStatReq() if (FUSE_BN_STATS and not frozen) else None
            ops.conv2d(x16, cw16, z, relu=False, prec=4, pool=sq)
            if sq is not None and sq.fused:
                mean, rstd, scale, shift = bn_stats_from_partial(sq.partial, sq.blocks, z, self.bn)
            else:
                mean, rstd, scale, shift = bn_stats(z, self.bn)
            tiles = -1
        else:
            z = self.ws.map(self.tag + ".z", x.n, ho, wo, cw.cout, 1, prec, dev)
            tiles = 0 if (frozen or not FUSE_BN_STATS) else ops.conv_stat_tiles(x, cw, z, prec)
        if tiles < 0:
            pass
        elif tiles > 0:
            # the conv's epilogue also writes the per-tile channel sums: BatchNorm's statistics cost no pass over z
            part = self.ws.tensor(self.tag + ".stat", (tiles, 2, cw.cout), torch.float32, dev)
            ops.conv2d(x, cw, z, relu=False, prec=prec, stat_partial=part)
            mean, rstd, scale, shift = bn_stats_from_partial(part, tiles, z, self.bn)
        else:
            ops.conv2d(x, cw, z, relu=False, prec=prec)
            mean, rstd, scale, shift = bn_stats(z, self.bn)
        if pool is not None and out_h16 and prec == 3:
            pool[0].with_h16()
        if pool is not None and relu and residual is None and self.can_skip_output(prec):
            affine_maxpool(z, scale, shift, pool[0], pool[1])
            y, out = None, pool[0]
            self._pool_coeffs = (scale, shift)
        elif out_f16_only and prec == 3 and relu and residual is None and pool is None:
            y = self.ws.map(self.tag + ".y16", x.n, ho, wo, cw.cout, 1, 4, dev)
            map_affine(z, scale, shift, y, relu=True)
            y.h16 = y.hi                       # (the plane IS the fp16 operand plane)
            out = y
        else:
            y = self.ws.map(self.tag + ".y", x.n, ho, wo, cw.cout, 1, prec, dev, h16=out_h16 and prec == 3 and pool is None)
            map_affine(z, scale, shift, y, residual=residual, relu=relu)
            out = y
            if pool is not None:
                ops.maxpool3x3s2(y, pool[0], argmax=pool[1])
                out = pool[0]
                if out.h16 is not None:          # (rare path: the fused pass was not available) no producer wrote the plane
                    out.h16 = None
        sync_count = self.bn.__dict__.pop("_agp_sync_count", None)
        self._dgrad_hi_only = bool(DGRAD_HI_ONLY)      # (the switch as it stood at THIS forward: the backward runs after other models' forwards)
        self.saved = (x, z, y, mean, rstd, relu, residual is not None, prec, (hin, win), frozen, sync_count)
        return out

    # ----------------------------------------------------------------- backward
    def stats_request(self):
        """What the conv that produces this unit's output gradient needs to reduce the BatchNorm backward's channel sums in its
        own epilogue (agp_conv_desc.bstat_*), or None when this unit's backward all-reduces its sums (synchronised BatchNorm)."""
        _, z, y, mean, rstd, relu, _, prec, _, frozen, sync_count = self.saved
        if not FUSE_BN_BWD or prec != 3 or (sync_count is not None and not frozen and _sync_group() is not None):
            return None
        if relu and y is None:           # forward(pool=...) did not store the output: no mask plane to hand out
            return None
        return (z, y if relu else None, mean, rstd)

    def backward(self, gy: SplitMap, need_gx=True, partial=None, add=None, stats_for=None, pool_argmax=None, pooled=None):
        """gy: gradient at this unit's output.  Returns (gx, gres, fused) with fused = (add_done, partial_next):
        pool_argmax: gy is the gradient at the 3x3/2 max-pool of this unit's output (the stem) and this is the pool's argmax:
        the BatchNorm backward gathers the gradient through the pool itself (agp_maxpool_bn_bwd), no full-size gradient map;
        pooled: the forward's pooled map (still intact): the channel sums are then taken over the pooled elements alone;
        partial: this unit's BatchNorm-backward channel sums, already reduced by the conv that produced gy (see stats_for);
        add: a map to add to gx (the other branch's gradient at this unit's input) -- added in the data-gradient conv's epilogue
        when that kernel takes a residual (add_done), otherwise left to the caller;
        stats_for: the ConvBNUnit whose output is this unit's input: gx (+ add) is ITS output gradient, and the data-gradient
        conv reduces its BatchNorm-backward sums (partial_next = (tensor, tiles), or None when the kernel cannot)."""
        x, z, y, mean, rstd, relu, has_res, prec, (hin, win), frozen, sync_count = self.saved
        conv, bn, dev, ws, tag = self.conv, self.bn, z.hi.device, self.ws, self.tag
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        cout = conv.out_channels
        gz = ws.map(tag + ".gz", z.n, z.h, z.w, z.c, 1, prec, dev)
        gres = ws.map(tag + ".gres", z.n, z.h, z.w, z.c, 1, prec, dev) if has_res else None
        # one-pass weight gradient: the BatchNorm backward below also folds max |gz| per channel into `absmax` (not the
        # synchronised path)
        synced_bwd = sync_count is not None and not frozen and _sync_group() is not None
        absmax = None
        if x.h16 is not None and self.wgrad_f16_ok(prec) and not synced_bwd and conv.weight.requires_grad:
            absmax = ws.tensor(tag + ".gabsmax", (cout,), torch.int32, dev, zero=True)
        done = None
        if pool_argmax is not None:
            assert not has_res and partial is None
            synced = sync_count is not None and not frozen and _sync_group() is not None
            if y is None and relu:          # forward(pool=...) did not store the output: the mask is recomputed from z
                done = maxpool_bn_bwd(pool_argmax, gy, z, None, mean, rstd, bn.weight, relu, gz, frozen=frozen,
                                      scale=self._pool_coeffs[0], shift=self._pool_coeffs[1], pooled=pooled, beta=bn.bias, absmax=absmax)
                if done is None:
                    raise RuntimeError("agp_maxpool_bn_bwd refused a unit whose output was not stored")
            elif FUSE_BN_BWD and prec == 3 and not synced:
                done = maxpool_bn_bwd(pool_argmax, gy, z, y if relu else None, mean, rstd, bn.weight, relu, gz, frozen=frozen,
                                      pooled=pooled if relu else None, beta=bn.bias, absmax=absmax)
            if done is None:
                gy = maxpool_bwd(pool_argmax, gy, ws.map(tag + ".gpool", z.n, z.h, z.w, z.c, 1, prec, dev))
        gg, gb = done if done is not None else bn_bwd(z, gy, y if relu else None, mean, rstd, bn.weight, relu, gz, gres, frozen=frozen,
                                                      sync_count=sync_count, partial=partial, absmax=absmax)
        _acc_grad(bn.weight, gg)
        _acc_grad(bn.bias, gb)
        if conv.bias is not None:
            _acc_grad(conv.bias, chan_sum(gz))
        self._wgrad(x, gz, prec, hin, win, absmax)
        gx, fused = None, (False, None)
        if need_gx and not self.stem:
            gx, fused = self._dgrad(x, gz, prec, add, stats_for)
        return gx, gres, fused

    def _wgrad(self, x, gz, prec, hin, win, absmax=None):
        """dW by agp_conv2d_wgrad (NHWC maps + LDS transpose reads); absmax (with x.h16): one fp16 product."""
        conv, dev = self.conv, gz.hi.device
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        cin, cout = conv.in_channels, conv.out_channels
        d = _lib.ConvDesc()
        d.in_hi, d.in_lo = ptr(x.hi), ptr(x.lo)
        d.out_hi, d.out_lo = ptr(gz.hi), ptr(gz.lo)
        d.n, d.hin, d.win, d.pin = x.n, hin, win, x.pad
        d.cin, d.in_w_step = (32, 4) if self.stem else (cin, cin)
        d.hout, d.wout, d.cout, d.pout = gz.h, gz.w, cout, gz.pad
        d.kh, d.kw = (k, 1) if self.stem else (k, k)
        d.stride, d.pad, d.prec = s, p, prec
        if absmax is not None and x.h16 is not None:
            d.in_h16, d.out_absmax = ptr(x.h16), ptr(absmax)
        L = _L()
        nbytes = L.agp_conv2d_wgrad_workspace_bytes(C.byref(d))
        if nbytes < 0 or prec != 3:
            raise NotImplementedError(f"conv weight gradient for kernel {k}x{k} stride {s} cin {cin} cout {cout} prec {prec}")
        wsb = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        if self.stem:
            gw = torch.empty((k, 8, 4, cout), dtype=torch.float32, device=dev)
            check(L.agp_conv2d_wgrad(C.byref(d), ptr(gw), ptr(wsb), nbytes, _lib.stream()), "agp_conv2d_wgrad")
            _acc_grad(conv.weight, gw[:, :k, :cin].permute(3, 2, 0, 1))
        elif conv.weight.requires_grad:
            # straight into weight.grad, in the parameter's layout (accumulated when a gradient exists: shared trunks, several
            # backward passes per step): no transposing copy and no add per conv
            w = conv.weight
            acc = w.grad is not None
            if not acc:
                w.grad = torch.empty_like(w, memory_format=torch.contiguous_format)
            if w.grad.dtype != torch.float32 or not w.grad.is_contiguous():
                gw = torch.empty((k, k, cin, cout), dtype=torch.float32, device=dev)
                check(L.agp_conv2d_wgrad(C.byref(d), ptr(gw), ptr(wsb), nbytes, _lib.stream()), "agp_conv2d_wgrad")
                if acc:
                    w.grad += gw.permute(3, 2, 0, 1).to(w.grad.dtype)
                else:
                    w.grad.copy_(gw.permute(3, 2, 0, 1))
            else:
                check(L.agp_conv2d_wgrad_param(C.byref(d), ptr(w.grad), 1 if acc else 0, ptr(wsb), nbytes, _lib.stream()),
                      "agp_conv2d_wgrad_param")

    def _dgrad(self, x, gz, prec, add=None, stats_for=None):
        conv, dev, ws, tag = self.conv, gz.hi.device, self.ws, self.tag
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        cin = conv.in_channels
        if prec == 3 and conv.out_channels % 8 == 0:
            cwt = ops.ConvWeights.for_training(conv.weight, None, 1, (k - 1) // 2, dgrad=True, fwd_stride=s,
                                               plane_pixels=x.n * (x.h + 2 * x.pad) * (x.w + 2 * x.pad))
        else:
            wflip = conv.weight.detach().flip(2, 3).transpose(0, 1).contiguous()      # [cin][cout][k][k]
            cwt = ops.ConvWeights(wflip, None, None, 1, (k - 1) // 2)
        gx = ws.map(tag + ".gx", x.n, x.h, x.w, cin, 1, prec, dev)

        def last_conv(src):
            # the conv that writes gx: with the other branch's gradient as its residual and the consumer unit's BatchNorm-backward
            # sums in its epilogue, where the kernel that runs it can (the 3x3 stride-1 kernel on bf16-pair maps)
            req = stats_for.stats_request() if stats_for is not None else None
            # the opt-in one-product form: the 3x3 stride-1 kernel's shapes (conv_stat_tiles > 0 for a 3x3 conv <=> that kernel runs it)
            hi = bool(getattr(self, "_dgrad_hi_only", False) and prec == 3 and k == 3 and ops.conv_stat_tiles(src, cwt, gx, prec) > 0)
            # (backward sums: the 3x3 stride-1 kernel alone; other kernels' tiles are forward statistics)
            tiles = ops.conv_stat_tiles(src, cwt, gx, prec, hi_only=hi) if (FUSE_BN_BWD and k == 3 and (add is not None or req is not None)) else 0
            if tiles <= 0:
                ops.conv2d(src, cwt, gx, relu=False, prec=prec, hi_only=hi)
                return False, None
            if req is None:
                ops.conv2d(src, cwt, gx, residual=add, relu=False, prec=prec, hi_only=hi)
                return add is not None, None
            part = ws.tensor(tag + ".bstat", (tiles, 2, cin), torch.float32, dev)
            ops.conv2d(src, cwt, gx, residual=add, relu=False, prec=prec, stat_partial=part, bstat=req, hi_only=hi)
            return add is not None, (part, tiles)
        if s == 1:
            fused = last_conv(gz)
        elif k == 1:
            t = ws.map(tag + ".gxs", gz.n, gz.h, gz.w, cin, 1, prec, dev)
            ops.conv2d(gz, cwt, t, relu=False, prec=prec)
            upsample2_zero(t, gx)
            fused = (False, None)
        else:
            u = ws.map(tag + ".gu", gz.n, x.h, x.w, gz.c, 1, prec, dev)
            upsample2_zero(gz, u)
            fused = last_conv(u)
        return gx, fused
