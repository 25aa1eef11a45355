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
        return autograd_ops.FCODEFn.apply(x, fc.weight, fc.bias, self, add1, add2)
