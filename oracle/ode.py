"""Oracle: fixed-grid ODE integration as the reference uses it (TEST INFRASTRUCTURE).

Reference call site: network_mm/ffns.py:78-87

    t   = torch.tensor([0, 1]).float().type_as(x)
    out = odeint(self.func, x, t, method=opt.odeint_method,
                 options={'step_size': opt.odeint_size}, rtol=opt.tol, atol=opt.tol)[-1]

`odeint` is torchdiffeq's (third-party, NOT in /root/reference, version unpinned
in README.md:42) -> PARITY UNPINNED for the solver arithmetic.  What follows
restates torchdiffeq's published fixed-grid algorithm:

  grid constructor (FixedGridODESolver._grid_constructor_from_step_size):
      niters = ceil((t1 - t0) / step + 1)
      grid   = arange(niters) * step + t0 ;  grid[-1] = t1          (dtype of t)
  integrate loop:  for (ta, tb) in zip(grid[:-1], grid[1:]):
      dt = tb - ta ;  y <- y + step_func(f, ta, dt, tb, y)
  step functions:
      euler    : dt * f(ta, y)
      midpoint : dt * f(ta + dt/2, y + f(ta, y) * dt/2)
      rk4      : the 3/8-rule variant (rk4_alt_step_func)
                 k1 = f(ta, y)
                 k2 = f(ta + dt/3,   y + dt*k1/3)
                 k3 = f(ta + 2dt/3,  y + dt*(k2 - k1/3))
                 k4 = f(tb,          y + dt*(k1 - k2 + k3))
                 dy = (k1 + 3*(k2 + k3) + k4) * dt / 8
  rtol/atol are ignored by fixed-grid solvers.  t=[0,1] so the last grid point is
  exactly t[-1] and the returned state is the last y (no interpolation).

The dynamics are autonomous (ODEFunc.forward ignores t, ffns.py:19-21).
"""
import math
import torch

METHODS = ("euler", "midpoint", "rk4")
ACTS = ("id", "relu", "tanh", "sigmoid")


def fixed_grid(step_size, t0=0.0, t1=1.0, dtype=torch.float32):
    """Time grid exactly as torchdiffeq builds it, in `dtype` (reference: fp32)."""
    t0_t = torch.tensor(t0, dtype=dtype)
    t1_t = torch.tensor(t1, dtype=dtype)
    niters = int(torch.ceil((t1_t - t0_t) / step_size + 1).item())
    grid = torch.arange(0, niters, dtype=dtype) * step_size + t0_t
    grid[-1] = t1_t
    return grid


def grid_dts(step_size, dtype=torch.float32):
    """Per-step dt_n = grid[n+1] - grid[n] (computed in `dtype`, like the solver)."""
    g = fixed_grid(step_size, dtype=dtype)
    return g[1:] - g[:-1]


def act_fn(name):
    """reference network_mm/ffns.py:51-64 (select_act): None/'id' -> identity."""
    if name is None or name == "id":
        return lambda v: v
    if name == "relu":
        return torch.relu
    if name == "tanh":
        return torch.tanh
    if name == "sigmoid":
        return torch.sigmoid
    raise NotImplementedError(name)


def fc(x, weight, bias, act):
    """reference network_mm/ffns.py:68-76 (FC): act(Linear(x))."""
    return act_fn(act)(torch.nn.functional.linear(x, weight, bias))


def odeint_fixed(f, y0, method, step_size, dt_dtype=torch.float32):
    """Last state of torchdiffeq.odeint(f, y0, t=[0,1], method, step_size)."""
    if method not in METHODS:
        raise NotImplementedError(method)
    dts = grid_dts(step_size, dtype=dt_dtype).to(y0.dtype)
    y = y0
    third = 1.0 / 3.0
    for dt in dts:
        if method == "euler":
            dy = dt * f(y)
        elif method == "midpoint":
            half = 0.5 * dt
            dy = dt * f(y + f(y) * half)
        else:
            k1 = f(y)
            k2 = f(y + dt * k1 * third)
            k3 = f(y + dt * (k2 - k1 * third))
            k4 = f(y + dt * (k1 - k2 + k3))
            dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        y = y + dy
    return y


def fcode(x, weight, bias, act, method, step_size):
    """reference network_mm/ffns.py:78-87 (FCODE.forward)."""
    return odeint_fixed(lambda y: fc(y, weight, bias, act), x, method, step_size)


def parse_diff_type(diff_type):
    """reference network_mm/diff_block.py:26-34: 'fcode@relu_fcode@tanh' -> [('fcode','relu'),...]."""
    out = []
    for e in diff_type.split("_"):
        kind, act = e.split("@")
        if kind != "fcode":
            raise NotImplementedError(kind)
        out.append((kind, act))
    return out


def diff_block(x, params, prefix, diff_type, method, step_size):
    """reference network_mm/diff_block.py:36-49: sum over blocks of FCODE(x).

    params keys: f'{prefix}blocks.{j}.func.func.fc.{weight,bias}'.
    """
    outs = []
    for j, (_, act) in enumerate(parse_diff_type(diff_type)):
        w = params[f"{prefix}blocks.{j}.func.func.fc.weight"]
        b = params[f"{prefix}blocks.{j}.func.func.fc.bias"]
        outs.append(fcode(x, w, b, act, method, step_size))
    return sum(outs)


def n_feval(method, step_size):
    n = len(grid_dts(step_size))
    return n * {"euler": 1, "midpoint": 2, "rk4": 4}[method]


def euler_linear_closed_form(y0, W, b, h, n):
    """KAT: Euler with act=id: y_N = (I+hW)^N y0 + sum_k (I+hW)^k h b (fp64)."""
    d = W.shape[0]
    A = torch.eye(d, dtype=W.dtype) + h * W
    y = y0
    for _ in range(n):
        y = y @ A.T + h * b
    return y


def rk4_linear_closed_form(y0, A, h, n):
    """KAT: any 4-stage order-4 RK on y' = A y gives y1 = sum_{k<=4} (hA)^k/k! y0."""
    d = A.shape[0]
    hA = h * A
    P = torch.eye(d, dtype=A.dtype)
    term = torch.eye(d, dtype=A.dtype)
    for k in range(1, 5):
        term = term @ hA / k
        P = P + term
    y = y0
    for _ in range(n):
        y = y @ P.T
    return y
