"""GeM pooling, drop-in for reference network_mm/image_pooling.py:8-16.

forward(x[b,c,h,w]) -> [b,c,1,1] = avg_pool2d(x.clamp(min=eps).pow(p), (h,w)).pow(1/p) with a
learnable 1-element p (state_dict key `p`).  One HIP kernel pass forward (agp_pool_f32_fwd) and
one backward (agp_gem_f32_bwd, dL/dx and dL/dp); no ATen arithmetic.
"""
import torch
import torch.nn as nn

from .. import ops


class _GeMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, eps):
        x = x.float()
        _, y = ops.pool_f32(x, p.detach().float(), want_mean=False, want_gem=True, eps=eps)
        ctx.save_for_backward(x, p.detach().float(), y)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, gy):
        x, p, y = ctx.saved_tensors
        gx, gp = ops.gem_f32_bwd(x, p, y, gy.contiguous().float(), need_gx=ctx.needs_input_grad[0],
                                 eps=ctx.eps)
        return gx, (gp if ctx.needs_input_grad[1] else None), None


def gem_op(x, p, eps=1e-6):
    """[b,c,h,w] -> [b,c] GeM descriptor (autograd-capable)."""
    return _GeMFn.apply(x, p, eps)


class GeM(nn.Module):
    def __init__(self, p=3, eps=1e-6):
        super().__init__()
        self.p = nn.Parameter(torch.ones(1) * p)
        self.eps = eps

    def forward(self, x):
        # x: [b, c, h, w] -> [b, c, 1, 1]
        return gem_op(x, self.p, self.eps).view(x.size(0), x.size(1), 1, 1)

    def pool_map(self, m):
        """ops.SplitMap -> [b,c] (inference fast path, shares the pass with the level avg-pool)."""
        return ops.pool_map(m, self.p.detach(), want_mean=False, want_gem=True, eps=self.eps)[1]
