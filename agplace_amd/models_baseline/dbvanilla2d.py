"""DBVanilla2D (database / aerial network), drop-in for reference models_baseline/dbvanilla2d.py:17-114.

`DBVanilla2D(mode='db', dim=args.features_dim)`; `model(data_dict, mode='db') -> {'embedding'}`.
Per map type: ImageFE -> GeM -> MLP(Linear, LayerNorm, ReLU, Linear); stack over map types,
F.normalize, mean over map types; db_map is [b,nmap,3,h,w] (cache/test) or [b,ndb,nmap,3,h,w]
(train).  state_dict keys: dbimage_fes.{i}.fe.*, dbimage_pools.{i}.p, dbimage_mlps.{i}.seq.{0,1,3}.*
All arithmetic runs in libagplace_hip.so; inference only in this round.
"""
from typing import List

import torch
import torch.nn as nn

from .. import ops
from ..network.image_fe import ImageFE
from ..network.image_pooling import GeM
from ..network_mm.ffns import _PreparedLinear
from ..options import get_options


class MLP(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.seq = nn.Sequential(
            nn.Linear(input_dim, output_dim),
            nn.LayerNorm(output_dim),
            nn.ReLU(),
            nn.Linear(output_dim, output_dim),
        )
        self._p0, self._p3 = _PreparedLinear(self.seq[0]), _PreparedLinear(self.seq[3])

    def forward(self, x):
        out = ops.linear(x, self._p0.get())
        out = ops.layernorm(out, self.seq[1].weight, self.seq[1].bias, self.seq[1].eps, relu=True)
        return ops.linear(out, self._p3.get())


class DBVanilla2D(nn.Module):
    def __init__(self, mode: List[str], dim, opt=None):
        super().__init__()
        self.opt = opt = opt or get_options()
        if mode == 'db':
            maptype = opt.maptype.split('_')
            fes = [ImageFE(fe_type=opt.dbimage_fe, layers=opt.dbimage_fe_layers) for _ in maptype]
            self.dbimage_fes = nn.ModuleList(fes)
            self.dbimage_pools = nn.ModuleList([GeM() for _ in maptype])
            self.dbimage_mlps = nn.ModuleList([MLP(e.last_dim, dim) for e in fes])

    def forward_db(self, data_dict):
        opt = self.opt
        if self.training:
            raise NotImplementedError("agplace_amd.DBVanilla2D: training-mode forward is not built yet; "
                                      "call .eval().")
        db_map = data_dict['db_map']
        if db_map.dim() == 5:      # [b,nmap,3,h,w]  caching / testing
            mode = 'cachetest'
            b, nmap, c, h, w = db_map.shape
            db_map = db_map.unsqueeze(1)
            ndb = 1
        elif db_map.dim() == 6:    # [b,ndb,nmap,3,h,w]  training layout
            mode = 'train'
            b, ndb, nmap, c, h, w = db_map.shape
        else:
            raise NotImplementedError
        assert c == 3
        prec = opt.mfma_precision
        with torch.no_grad():
            vecs = []
            for i in range(nmap):
                j = 0 if opt.share_dbfe is True else i
                x = db_map[:, :, i].reshape(b * ndb, c, h, w)       # view when possible; strides are honoured
                maps = self.dbimage_fes[j].forward_maps(x, prec=prec)
                v = self.dbimage_pools[j].pool_map(maps[-1])
                v = self.dbimage_mlps[j](v)
                if opt.output_l2 is True:
                    v = ops.l2normalize(v)
                vecs.append(v)
            out = vecs[0] if nmap == 1 else ops.wsum(
                vecs, [torch.full((1,), 1.0 / nmap, device=vecs[0].device)] * nmap)
            out = out.view(b, ndb, -1)
            if mode == 'cachetest':
                out = out.view(b, -1)
            if opt.final_l2 is True:
                shp = out.shape
                out = ops.l2normalize(out.reshape(-1, shp[-1])).view(shp)
        return {'embedding': out}

    def forward(self, data_dict, mode: List[str]):
        if mode == 'db':
            return self.forward_db(data_dict)
        raise NotImplementedError
