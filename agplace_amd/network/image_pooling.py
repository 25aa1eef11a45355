"""GeM pooling, drop-in for reference network/image_pooling.py:8-18 (flattens to [b,c])."""
import torch
import torch.nn as nn

from .. import ops
from ..network_mm.image_pooling import gem_op


class GeM(nn.Module):
    def __init__(self, p=3, eps=1e-6):
        super().__init__()
        self.p = nn.Parameter(torch.ones(1) * p)
        self.eps = eps

    def forward(self, x):
        # x: [b, c, h, w] -> [b, c]
        return gem_op(x, self.p, self.eps)

    def pool_map(self, m):
        return ops.pool_map(m, self.p.detach(), want_mean=False, want_gem=True, eps=self.eps)[1]
