"""DiffBlock, drop-in for reference network_mm/diff_block.py:18-49.

Parses opt.diff_type ('fcode@relu', '_'-separated list, '@act') into a ModuleList `blocks`
of FCODE and returns the SUM of the block outputs.  state_dict keys: blocks.{j}.func.func.fc.*
"""
import torch.nn as nn

from .. import autograd_ops
from ..options import get_options
from .ffns import FCODE


class DiffBlock(nn.Module):
    def __init__(self, dim, ode_dim, opt=None):
        super().__init__()
        opt = opt or get_options()
        self.blocks = nn.ModuleList()
        for e in opt.diff_type.split('_'):
            e, act = e.split('@')
            if e == 'fcode':
                self.blocks.append(FCODE(dim, act, opt=opt))
            else:
                raise NotImplementedError

    def forward(self, x, z0=None, add1=None, add2=None):
        # x: [b, c]; add1/add2 are folded into the solver kernel's initial state (x+add1+add2)
        if z0 is not None:
            raise NotImplementedError("CDE blocks are not part of the reference's live path")
        outlist = [block(x, add1, add2) for block in self.blocks]
        return outlist[0] if len(outlist) == 1 else autograd_ops.wsum(outlist)
