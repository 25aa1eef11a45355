"""FuseBlockToShallow, drop-in for reference network_mm/fuse_block_toshallow.py:11-30,79-134.

Stage-1 fusion: per pyramid level, global-average-pool the image map and the voxel map,
Linear up-dim to dims[-1] (Identity on the last level), then run deep->shallow
(opt.diff_direction='backward') fusevec = DiffBlock_i(fusevec + img_i + vox_i).

MI355X build: the level average pools are one HBM pass per map (agp_pool_fwd), the up-dims
are agp_linear_fwd, the add is folded into the ODE kernel's initial state.  The sparse voxel
branch (MinkowskiEngine) is out of scope: `voxfeatmaplist` holds the already
globally pooled voxel vectors [b, C_i] (= ME.MinkowskiGlobalPooling()(v_i).F, :83).
state_dict keys: blocks.{i}.blocks.{j}.func.func.fc.*, updimsimg.{0,1}.*, updimsvox.{0,1}.*
"""
import torch
import torch.nn as nn

from .. import autograd_ops, ops
from ..options import get_options
from .diff_block import DiffBlock
from .ffns import _PreparedLinear


def _avgpool(m):
    """adaptive_avg_pool2d(e, 1).flatten(1) for an ops.SplitMap or an fp32 [b,c,h,w] tensor."""
    if hasattr(m, "vec"):          # pre-pooled level handed over by MM.forward_q
        return m.vec
    if isinstance(m, ops.SplitMap):
        return ops.pool_map(m, None, want_mean=True, want_gem=False)[0]
    return ops.pool_f32(m.float(), None, want_mean=True, want_gem=False)[0]


class FuseBlockToShallow(nn.Module):
    def __init__(self, dims=[256, 256, 256], img_dims=[64, 128, 256], vox_dims=[64, 128, 256],
                 bev_dims=[64, 128, 256], opt=None):
        super().__init__()
        self.opt = opt or get_options()
        self.dims, self.img_dims, self.vox_dims, self.bev_dims = dims, img_dims, vox_dims, bev_dims
        self.blocks = nn.ModuleList()
        self.updimsbev = nn.ModuleList()
        self.updimsimg = nn.ModuleList()
        self.updimsvox = nn.ModuleList()
        for i in range(len(dims)):
            self.blocks.append(DiffBlock(dim=dims[-1], ode_dim=dims[-1], opt=self.opt))
            if i < len(dims) - 1:
                self.updimsimg.append(nn.Linear(self.img_dims[i], dims[-1]))
                self.updimsvox.append(nn.Linear(self.vox_dims[i], dims[-1]))
            else:
                self.updimsimg.append(nn.Identity())
                self.updimsvox.append(nn.Identity())
        self._prep_img = [_PreparedLinear(m) if isinstance(m, nn.Linear) else None for m in self.updimsimg]
        self._prep_vox = [_PreparedLinear(m) if isinstance(m, nn.Linear) else None for m in self.updimsvox]

    def forward_imgvox(self, imagemaplist, bevmaplist=None, voxmaplist=None):
        assert len(imagemaplist) == len(self.dims)
        if 'cde' in self.opt.diff_type:
            raise NotImplementedError
        imageveclist = [_avgpool(e) for e in imagemaplist]
        voxveclist = list(voxmaplist)
        fusevec = None
        n = len(self.dims)
        for it in range(n):
            if self.opt.diff_direction == 'forward':
                i = it
            elif self.opt.diff_direction == 'backward':
                i = n - 1 - it
            else:
                raise NotImplementedError
            imagevec, voxvec = imageveclist[i], voxveclist[i].float()
            if self._prep_img[i] is not None:
                imagevec = autograd_ops.linear(imagevec, self.updimsimg[i], self._prep_img[i])
                voxvec = autograd_ops.linear(voxvec, self.updimsvox[i], self._prep_vox[i])
            if fusevec is None:        # fusevec = 0 + imagevec + voxvec
                fusevec = self.blocks[i](imagevec, add1=voxvec)
            else:
                fusevec = self.blocks[i](fusevec, add1=imagevec, add2=voxvec)
        return fusevec

    def emit(self, vp, imagemaplist, voxmaplist):
        """forward_imgvox as ops of a vecprog.VecProgram (inference, one launch with the rest of the vector path):
        registers 0 = fusevec, 1 / 2 = the level's image / voxel vector, 3.. = block outputs.  Returns the result register."""
        from ..vecprog import VecProgramUnfit
        if 'cde' in self.opt.diff_type or self.opt.diff_direction not in ('forward', 'backward') or self.dims[-1] != 256:
            raise VecProgramUnfit("diff block options")
        imageveclist = [_avgpool(e) for e in imagemaplist]
        n = len(self.dims)
        first = True
        for it in range(n):
            i = it if self.opt.diff_direction == 'forward' else n - 1 - it
            imagevec, voxvec = imageveclist[i], voxmaplist[i].float()
            if self._prep_img[i] is not None:
                vp.linear(1, self._prep_img[i].get(), imagevec)
                vp.linear(2, self._prep_vox[i].get(), voxvec)
            else:
                vp.load(1, imagevec)
                vp.load(2, voxvec)
            blocks = list(self.blocks[i].blocks)
            src = (1, 2, -1) if first else (0, 1, 2)          # fusevec = 0 + imagevec + voxvec on the first level
            if len(blocks) == 1:
                vp.fcode(0, blocks[0], *src)
            else:
                if len(blocks) > 3:
                    raise VecProgramUnfit("more than 3 ODE blocks per level")
                outs = [vp.fcode(3 + j, blk, *src) for j, blk in enumerate(blocks)]
                vp.wsum(0, outs)
            first = False
        return 0

    def forward(self, imagefeatmaplist, bevfeatmaplist, voxfeatmaplist, type=None):
        if type == 'vox':
            return self.forward_imgvox(imagefeatmaplist, bevfeatmaplist, voxfeatmaplist)
        raise NotImplementedError   # 'bev' is dead in the reference too (updimsbev is empty)
