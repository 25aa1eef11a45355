"""torch.autograd.Function drop-ins over the HIP kernels for the vector (fusion / MLP) path.

Forward and backward are both libagplace_hip.so kernels (agp_linear_fwd/_bwd, agp_fcode_fwd/_bwd,
agp_layernorm_fwd/_bwd, agp_l2normalize_fwd/_bwd, agp_wsum_fwd); no ATen arithmetic.  Under
torch.no_grad() they reduce to the plain forward launches.
"""
import torch

from . import ops


class LinearFn(torch.autograd.Function):
    """y = act((x + add1 + add2) W^T + b).  `prep` caches the split-bf16 planes of (W, W^T)."""

    @staticmethod
    def forward(ctx, x, weight, bias, prep, act, add1, add2):
        lw = prep.get()
        y = ops.linear(x, lw, act=act, add1=add1, add2=add2)
        if any(ctx.needs_input_grad):
            xin = x.contiguous() if add1 is None and add2 is None else \
                ops.wsum([t for t in (x, add1, add2) if t is not None])
            ctx.save_for_backward(xin, y if act not in (None, "id") else None)
            ctx.prep, ctx.act = prep, act
            ctx.has = (add1 is not None, add2 is not None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xin, y = ctx.saved_tensors
        lw = ctx.prep.get(with_transpose=True)
        need_gx = ctx.needs_input_grad[0] or ctx.needs_input_grad[5] or ctx.needs_input_grad[6]
        gx, gw, gb = ops.linear_bwd(xin, y, gy[:, :lw.n] if gy.shape[1] != lw.n else gy, lw, act=ctx.act,
                                    need_gx=need_gx, need_gw=ctx.needs_input_grad[1],
                                    need_gb=ctx.needs_input_grad[2] and ctx.has[2])
        return (gx if ctx.needs_input_grad[0] else None, gw, gb, None, None,
                gx if ctx.has[0] and ctx.needs_input_grad[5] else None,
                gx if ctx.has[1] and ctx.needs_input_grad[6] else None)


def linear(x, module, prep, act=None, add1=None, add2=None):
    """nn.Linear `module` applied through the HIP kernels with autograd support."""
    return LinearFn.apply(x, module.weight, module.bias, prep, act, add1, add2)


class FCODEFn(torch.autograd.Function):
    """Fixed-grid Neural-ODE block; backward is discretise-then-optimise (agp_fcode_bwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod, add1, add2):
        need = any(ctx.needs_input_grad)
        lw = mod._prep.get(with_transpose=need)
        out = ops.fcode(x, lw, mod.act_name, mod.method, mod.dts, add1=add1, add2=add2, want_traj=need)
        if not need:
            return out
        y, traj = out
        ctx.save_for_backward(traj)
        ctx.mod = mod
        ctx.has = (add1 is not None, add2 is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        (traj,) = ctx.saved_tensors
        mod = ctx.mod
        lw = mod._prep.get(with_transpose=True)
        gx, gw, gb = ops.fcode_bwd(traj, gy, lw, mod.act_name, mod.method, mod.dts)
        return (gx if ctx.needs_input_grad[0] else None, gw if ctx.needs_input_grad[1] else None,
                gb if ctx.needs_input_grad[2] else None, None,
                gx if ctx.has[0] and ctx.needs_input_grad[4] else None,
                gx if ctx.has[1] and ctx.needs_input_grad[5] else None)


class LayerNormFn(torch.autograd.Function):
    """y = relu?(LayerNorm(x) * gamma + beta + residual)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, residual):
        x = x.contiguous()
        y = ops.layernorm(x, gamma, beta, eps, relu=relu, residual=residual)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, gamma, y)
            ctx.eps, ctx.relu, ctx.has_res = eps, relu, residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, y = ctx.saved_tensors
        gx, gres, gg, gbeta = ops.layernorm_bwd(x, gamma, y, gy, ctx.eps, ctx.relu, need_res=ctx.has_res)
        return (gx if ctx.needs_input_grad[0] else None, gg if ctx.needs_input_grad[1] else None,
                gbeta if ctx.needs_input_grad[2] else None, None, None,
                gres if ctx.has_res and ctx.needs_input_grad[5] else None)


def layernorm(x, ln, relu=False, residual=None):
    return LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, relu, residual)


class L2NormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x)
        return ops.l2normalize(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.l2normalize_bwd(x, gy)


def l2normalize(x):
    return L2NormalizeFn.apply(x)


class WsumFn(torch.autograd.Function):
    """sum_t w_t * x_t with 1-element device weights; gradients flow to the vectors and, when a weight is a learnable
    parameter (reference tools/options.py:139-146, xxx_learnweight=True), to the weight: dL/dw_t = <dL/dy, x_t>."""

    @staticmethod
    def forward(ctx, nterms, *args):
        xs, ws = list(args[:nterms]), list(args[nterms:])
        ctx.ws, ctx.n = [None if w is None else w.detach() for w in ws], nterms
        learn = [w is not None and w.requires_grad for w in ws]
        ctx.learn = learn
        ctx.save_for_backward(*[x if lw else None for x, lw in zip(xs, learn)])
        return ops.wsum(xs, ws)

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        outs = [None]
        for t in range(ctx.n):
            if ctx.needs_input_grad[1 + t]:
                outs.append(gy if ctx.ws[t] is None else ops.wsum([gy], [ctx.ws[t]]))
            else:
                outs.append(None)
        xs = ctx.saved_tensors
        for t in range(ctx.n):
            if ctx.learn[t] and ctx.needs_input_grad[1 + ctx.n + t]:
                outs.append(ops.dot(gy, xs[t]).reshape(ctx.ws[t].shape))
            else:
                outs.append(None)
        return tuple(outs)


def wsum(xs, ws=None):
    ws = [None] * len(xs) if ws is None else list(ws)
    if len(xs) > 6:
        head = wsum(xs[:5], ws[:5])
        return wsum([head] + list(xs[5:]), [None] + ws[5:])
    return WsumFn.apply(len(xs), *xs, *ws)
