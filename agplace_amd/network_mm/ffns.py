"""FC / ODEFunc / FCODE, drop-ins for reference network_mm/ffns.py:14-21,51-76,78-87.

FCODE.forward(x[b,D]) = odeint(f, x, t=[0,1], method=opt.odeint_method,
options={'step_size': opt.odeint_size})[-1] with f(t, y) = act(Linear(y)).  torchdiffeq is not
used: the whole fixed-grid solve (all steps, all stages) is ONE persistent HIP kernel
(agp_fcode_fwd) that keeps W's MFMA fragments in registers.  The time grid follows
torchdiffeq's constructor exactly (niters = ceil(1/step + 1), arange * step, last point forced
to 1) and is kept in fp32 (the reference builds t with .float().type_as(x)).
'rk4' is torchdiffeq's 3/8-rule variant.  rtol/atol (opt.tol) are ignored by fixed-grid solvers.
state_dict keys: func.func.fc.{weight,bias} (via ODEFunc -> FC -> nn.Linear).
"""
import torch
import torch.nn as nn

from .. import autograd_ops, ops
from ..options import get_options

_ACTS = (None, 'id', 'relu', 'tanh', 'sigmoid')


def select_act(act):
    """Same contract as the reference: returns the activation NAME's module (for introspection);
    unknown names raise NotImplementedError (ffns.py:62-63)."""
    if act is None or act == 'id':
        return nn.Identity()
    if act == 'relu':
        return nn.ReLU()
    if act == 'tanh':
        return nn.Tanh()
    if act == 'sigmoid':
        return nn.Sigmoid()
    raise NotImplementedError


class _PreparedLinear:
    """Caches the split-bf16 planes of an nn.Linear until its parameters change."""

    def __init__(self, linear, with_transpose=False):
        self.linear, self.with_transpose = linear, with_transpose
        self._key, self._lw = None, None

    def get(self, with_transpose=False):
        """Split planes of W (and of W^T when a backward pass needs gz W)."""
        w, b = self.linear.weight, self.linear.bias
        wt = self.with_transpose or with_transpose
        key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
        if key != self._key or (wt and self._lw.wt_hi is None):
            self._lw = ops.LinearWeights(w, b, with_transpose=wt)
            self._key = key
        return self._lw


class FC(nn.Module):
    def __init__(self, indim, outdim, act=None):
        super().__init__()
        if act not in _ACTS:
            raise NotImplementedError
        self.fc = nn.Linear(indim, outdim)
        self.act = select_act(act)
        self.act_name = act
        self._prep = _PreparedLinear(self.fc)

    def forward(self, x):
        return autograd_ops.linear(x, self.fc, self._prep, act=self.act_name)


class ODEFunc(nn.Module):
    def __init__(self, func):
        super().__init__()
        self.func = func

    def forward(self, t, x):
        return self.func(x)


class FCODE(nn.Module):
    def __init__(self, dim, act=None, opt=None):
        super().__init__()
        opt = opt or get_options()
        self.func = ODEFunc(FC(dim, dim, act))
        self.act_name = act
        self.method = opt.odeint_method
        self.step_size = opt.odeint_size
        if self.method not in ('euler', 'midpoint', 'rk4'):
            raise NotImplementedError(self.method)
        self.dts = ops.ode_grid_dts(self.step_size)
        self._prep = _PreparedLinear(self.func.func.fc)

    def forward(self, x, add1=None, add2=None):
        fc = self.func.func.fc
        if fc.in_features == 256:
            return autograd_ops.FCODEFn.apply(x, fc.weight, fc.bias, self, add1, add2)
        return self._forward_any_width(x, add1, add2)

    def _forward_any_width(self, x, add1, add2):
        """FCODE(dim) for dim != 256 (the reference's class takes any width, ffns.py:78-87; `--mm_stg2fuse_dim`,
        tools/options.py:113): the persistent one-launch solver keeps W as 64 VGPRs of MFMA fragments per wave, which is a
        256 x 256 matrix; other widths run the same fixed-grid steps as a sequence of launches -- act(Linear(y)) on
        agp_linear_fwd and the stage combinations on agp_wsum_fwd (both differentiable through autograd_ops) -- with the
        time grid of torchdiffeq's constructor.  Latency-bound like the fused kernel, ~3 launches per stage instead of one
        per solve."""
        fc, dev = self.func.func.fc, x.device
        if fc.in_features % 32:
            raise NotImplementedError("FCODE: dim must be a multiple of 32")
        key = str(dev)
        if getattr(self, "_consts_dev", None) != key:
            def c(v):
                return torch.tensor([v], dtype=torch.float32, device=dev)
            self._consts = [{"dt": c(dt), "hdt": c(0.5 * dt), "dt3": c(dt / 3.0), "mdt3": c(-dt / 3.0), "mdt": c(-dt),
                             "dt8": c(dt * 0.125), "3dt8": c(dt * 0.375)} for dt in self.dts]
            self._consts_dev = key

        def f(y):
            return autograd_ops.linear(y, fc, self._prep, act=self.act_name)
        ws = autograd_ops.wsum
        y = x if add1 is None and add2 is None else ws([t for t in (x, add1, add2) if t is not None])
        for k in self._consts:
            if self.method == 'euler':
                y = ws([y, f(y)], [None, k["dt"]])
            elif self.method == 'midpoint':
                k1 = f(y)
                y = ws([y, f(ws([y, k1], [None, k["hdt"]]))], [None, k["dt"]])
            else:       # torchdiffeq's 'rk4' = the 3/8 rule (rk4_alt_step_func)
                k1 = f(y)
                k2 = f(ws([y, k1], [None, k["dt3"]]))
                k3 = f(ws([y, k2, k1], [None, k["dt"], k["mdt3"]]))
                k4 = f(ws([y, k1, k2, k3], [None, k["dt"], k["mdt"], k["dt"]]))
                y = ws([y, k1, k2, k3, k4], [None, k["dt8"], k["3dt8"], k["3dt8"], k["dt8"]])
        return y
