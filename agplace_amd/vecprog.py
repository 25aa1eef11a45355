"""Host side of agp_vecprog_run (agplace_amd/csrc/vecprog.hip): the vector path of an inference forward -- everything
between the backbones' pooled vectors and the descriptors (reference network_mm/mm.py:91-129, fuse_block_toshallow.py:79-121,
stage2fuse_blockadd.py:216-218, models_baseline/dbvanilla2d.py:81-92) -- written as a short program of row-wise operations and
issued as ONE launch instead of one launch per Linear / FCODE / LayerNorm / normalize / weighted sum.

    vp = VecProgram(b, device)
    r = vp.load(mean3)                       # register <- [b, k] fp32 tensor
    r = vp.fcode(fcode_module, r, add1=v)    # registers are plain ints (0 .. VECPROG_NREG-1), managed by the caller
    vp.store(r, out)                         # [b, 256] fp32 tensor <- register
    vp.run()

Inference only (no autograd): callers use it under torch.no_grad(); the per-op autograd Functions (autograd_ops.py) remain
the training path.  A program that does not fit (more than VECPROG_MAXOPS ops, a Linear that is not [256, k<=256]) raises
VecProgramUnfit at build time and the caller falls back to the per-op path.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr


class VecProgramUnfit(Exception):
    pass


class VecProgram:
    def __init__(self, b, device):
        self.b, self.device = int(b), device
        self.ops, self.keep = [], []
        self.ode = None                      # (method, dts) shared by every FCODE op

    # ------------------------------------------------------------------ helpers
    def _op(self, op, dst=-1, r=(), k=0, act=0, aux=0, n=0, f0=0.0, p=()):
        if len(self.ops) >= _lib.VECPROG_MAXOPS:
            raise VecProgramUnfit("too many ops")
        o = _lib.VecProgOp()
        o.op, o.dst, o.k, o.act, o.aux, o.n, o.f0 = op, dst, k, act, aux, n, f0
        for i in range(6):
            o.r[i] = r[i] if i < len(r) else -1
            o.p[i] = p[i] if i < len(p) else None
        self.ops.append(o)
        return dst

    def _vec(self, t, k=None):
        """A [b, k] fp32 tensor the program reads through a raw pointer."""
        if t.dim() != 2 or t.shape[0] != self.b or t.device != self.device or (k is not None and t.shape[1] != k):
            raise VecProgramUnfit(f"operand of shape {tuple(t.shape)} on {t.device}")
        t = t.detach().contiguous().float()
        self.keep.append(t)
        return t

    def _weights(self, lw):
        if lw.npad != 256 or lw.n != 256 or lw.k > 256 or lw.k % 32:
            raise VecProgramUnfit(f"Linear [{lw.n}, {lw.k}]")
        self.keep.append(lw)
        from .ops import linear_fragment_planes
        fh, fl = linear_fragment_planes(lw)          # the programs read the weights fragment-major (one 1 KB run per wave instruction)
        return ptr(fh), ptr(fl), ptr(lw.bias)

    def _scalar(self, w):
        if w is None:
            return None
        w = w.detach()
        if w.numel() != 1 or w.device != self.device or w.dtype != torch.float32:
            raise VecProgramUnfit("weight must be one fp32 element on the device")
        self.keep.append(w)
        return ptr(w)

    # ------------------------------------------------------------------ ops
    def load(self, dst, t, scale=None):
        k = t.shape[1]
        if k > 256 or k % 4:
            raise VecProgramUnfit(f"load of width {k}")
        return self._op(_lib.VP_LOAD, dst, k=k, p=(ptr(self._vec(t)), self._scalar(scale)))

    def store(self, r, out=None):
        if out is None:
            out = torch.empty((self.b, 256), dtype=torch.float32, device=self.device)
        if tuple(out.shape) != (self.b, 256) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise VecProgramUnfit("store target")
        self.keep.append(out)
        self._op(_lib.VP_STORE, r=(r,), p=(ptr(out),))
        return out

    def linear(self, dst, lw, src, add1=-1, add2=-1, act=None):
        """src: a register, or a [b, lw.k] tensor read straight from memory."""
        wh, wl, bias = self._weights(lw)
        if torch.is_tensor(src):
            if src.shape[1] != lw.k:
                raise VecProgramUnfit("operand width")
            src = self.load(dst, src)                 # through the destination register (the product reads before it writes)
        elif lw.k != 256:
            raise VecProgramUnfit("a register operand is 256 wide")
        return self._op(_lib.VP_LINEAR, dst, r=(src, add1, add2), k=lw.k, act=_lib.ACT[act], p=(wh, wl, bias))

    def fcode(self, dst, mod, src, add1=-1, add2=-1):
        """mod: network_mm.ffns.FCODE."""
        lw = mod._prep.get()
        if lw.k != 256:
            raise VecProgramUnfit("FCODE width")
        ode = (_lib.ODE[mod.method], tuple(mod.dts))
        if self.ode is None:
            self.ode = ode
        elif self.ode != ode:
            raise VecProgramUnfit("FCODE blocks with different solvers")
        if len(mod.dts) > 48:
            raise VecProgramUnfit("ODE grid longer than 48 steps")
        wh, wl, bias = self._weights(lw)
        return self._op(_lib.VP_FCODE, dst, r=(src, add1, add2), k=256, act=_lib.ACT[mod.act_name], p=(wh, wl, bias))

    def l2norm(self, dst, src):
        return self._op(_lib.VP_L2NORM, dst, r=(src,))

    def layernorm(self, dst, ln, src, relu=False, residual=-1):
        if tuple(ln.normalized_shape) != (256,):
            raise VecProgramUnfit("LayerNorm width")
        g = None if ln.weight is None else ln.weight.detach().float().contiguous()
        bta = None if ln.bias is None else ln.bias.detach().float().contiguous()
        self.keep += [g, bta]
        return self._op(_lib.VP_LAYERNORM, dst, r=(src, residual), act=1 if relu else 0, f0=float(ln.eps), p=(ptr(g), ptr(bta)))

    def wsum(self, dst, regs, weights=None):
        weights = [None] * len(regs) if weights is None else list(weights)
        if not 1 <= len(regs) <= 6:
            raise VecProgramUnfit("weighted sum of more than 6 terms")
        return self._op(_lib.VP_WSUM, dst, r=tuple(regs), n=len(regs), p=tuple(self._scalar(w) for w in weights))

    # ------------------------------------------------------------------ launch
    def fits_beside(self, other):
        """Can `other` ride in this program's launch (VecProgram.run(rider=other))?"""
        return (other is not None and other.device == self.device and len(self.ops) + len(other.ops) <= _lib.VECPROG_MAXOPS
                and (other.ode is None or other.ode == self.ode))

    def run(self, rider=None):
        """rider: a second, independent VecProgram (its own rows, inputs and outputs) executed by the SAME launch on workgroups
        of its own (agp_vecprog_run2): two latency chains side by side instead of one behind the other.  The caller checks
        fits_beside(rider) first."""
        n = len(self.ops)
        arr = (_lib.VecProgOp * n)(*self.ops)
        method, dts = self.ode if self.ode is not None else (0, ())
        dt = (C.c_float * max(1, len(dts)))(*dts)
        if rider is None:
            check(_lib.load().agp_vecprog_run(arr, n, self.b, method, dt, len(dts), _lib.stream()), "agp_vecprog_run")
            return
        if not self.fits_beside(rider):
            raise VecProgramUnfit("rider program does not fit beside this one")
        m = len(rider.ops)
        arr_b = (_lib.VecProgOp * m)(*rider.ops)
        self.keep.append(rider)
        check(_lib.load().agp_vecprog_run2(arr, n, self.b, arr_b, m, rider.b, method, dt, len(dts), _lib.stream()), "agp_vecprog_run2")
