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

from .. import ops
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

    def get(self):
        w, b = self.linear.weight, self.linear.bias
        key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
        if key != self._key:
            self._lw = ops.LinearWeights(w, b, with_transpose=self.with_transpose)
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
        return ops.linear(x, self._prep.get(), act=self.act_name)


class ODEFunc(nn.Module):
    def __init__(self, func):
        super().__init__()
        self.func = func

    def forward(self, t, x):
        return self.func(x)


class _FCODEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mod, add1, add2):
        lw = mod._prep.get()
        need_grad = any(ctx.needs_input_grad[:3])
        out = ops.fcode(x, lw, mod.act_name, mod.method, mod.dts, add1=add1, add2=add2,
                        want_traj=need_grad)
        if need_grad:
            y, traj = out
            ctx.save_for_backward(traj)
            ctx.mod = mod
            return y
        return out

    @staticmethod
    def backward(ctx, gy):
        raise NotImplementedError(
            "FCODE backward (agp_fcode_bwd) is not implemented yet; run the query model under "
            "torch.no_grad() / requires_grad_(False). See DESIGN.md, 'what comes next'.")


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
        self._prep = _PreparedLinear(self.func.func.fc, with_transpose=True)

    def forward(self, x, add1=None, add2=None):
        fc = self.func.func.fc
        return _FCODEFn.apply(x, fc.weight, fc.bias, self, add1, add2)
