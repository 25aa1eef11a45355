"""DBVanilla2D (database / aerial network), drop-in for reference models_baseline/dbvanilla2d.py:17-114.

`DBVanilla2D(mode='db', dim=args.features_dim)`; `model(data_dict, mode='db') -> {'embedding'}`.
Per map type: ImageFE -> GeM -> MLP(Linear, LayerNorm, ReLU, Linear); stack over map types,
F.normalize, mean over map types; db_map is [b,nmap,3,h,w] (cache/test) or [b,ndb,nmap,3,h,w]
(train).  state_dict keys: dbimage_fes.{i}.fe.*, dbimage_pools.{i}.p, dbimage_mlps.{i}.seq.{0,1,3}.*
All arithmetic runs in libagplace_hip.so.  .eval()+no_grad = inference; .train() = end-to-end training
(batch-statistics BatchNorm + conv backward on HIP kernels, train_fns.TrunkFn).
"""
from typing import List

import torch
import torch.nn as nn

from .. import autograd_ops, ops, train_fns
from ..network.image_fe import ImageFE
from ..network.image_pooling import GeM
from ..network_mm.ffns import _PreparedLinear
from ..options import get_options
from ..vecprog import VecProgram


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
        out = autograd_ops.linear(x, self.seq[0], self._p0)
        out = autograd_ops.layernorm(out, self.seq[1], relu=True)
        return autograd_ops.linear(out, self.seq[3], self._p3)


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

    def freeze_backbone(self):
        """requires_grad=False on the ResNets and GeM exponents (the conv kernels have no backward yet);
        the per-map MLP heads stay trainable through the HIP backward kernels."""
        for m in list(self.dbimage_fes) + list(self.dbimage_pools):
            for p in m.parameters():
                p.requires_grad_(False)
        self._frozen_backbone = True
        return self

    def final_pool_request(self, i):
        """The GeM of map type i's last stage output (dbvanilla2d.py:74) as an ops.PoolReq the trunk's last conv fills."""
        j = 0 if self.opt.share_dbfe is True else i
        return ops.PoolReq(self.dbimage_pools[j].p, eps=self.dbimage_pools[j].eps, want_mean=False, want_gem=True)

    def forward_db(self, data_dict, trunk_maps=None, out_rows=None, defer_head=None):
        """trunk_maps: optional {map index: (stage maps, filled final PoolReq)} of the tiles computed by the caller
        (agplace_amd.pair runs a trunk in lock-step with the query network's).
        out_rows: optional preallocated fp32 [b * ndb, 256] tensor the fused inference head writes the embedding rows to.
        defer_head: optional list; the fused inference head (one vector program) is then NOT launched but appended to it: the
        caller lets it ride in another program's launch (VecProgram.run(rider=...), agplace_amd.pair) -- the returned embedding
        is written when that launch runs."""
        opt = self.opt
        # .train() under torch.no_grad() (train.py:315 with --train_modeldb False): batch-statistics BatchNorm with
        # running-stat updates and no tape -- the train-mode kernels run, the autograd Functions record nothing.
        # .eval() with gradients enabled and trainable parameters: the training graph on frozen BatchNorm statistics
        # (train_graph.bn_frozen) -- F.batch_norm(training=False) under autograd.
        train = self.training or (torch.is_grad_enabled() and not getattr(self, "_frozen_backbone", False)
                                  and any(p.requires_grad for p in self.parameters()))
        db_map = data_dict['db_map']
        u8 = db_map.dtype == torch.uint8
        if u8:
            # decoded uint8 tiles [b,nmap,h,w,3] or [b,ndb,nmap,h,w,3] (HWC, as the image decoder leaves them): ToTensor +
            # Normalize happen on the device inside the stem's input packing (ops.pack_cameras_u8, 4x fewer bytes over PCIe)
            if db_map.shape[-1] != 3 or db_map.dim() not in (5, 6):
                raise NotImplementedError("uint8 db_map must be [b,nmap,h,w,3] or [b,ndb,nmap,h,w,3]")
            db_map = db_map.permute(*range(db_map.dim() - 3), -1, -3, -2)      # logical [...,3,h,w] view, no copy
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
        prec = 3 if train else opt.mfma_precision
        if train:
            from .. import train_graph
            train_graph.FWD_F16 = opt.train_precision == 16      # the opt-in fast mode: one-product forward convs (train_graph.py)
            train_graph.DGRAD_HI_ONLY = opt.train_dgrad_products == 1      # ... and one-product data gradients
        if (not train and torch.is_grad_enabled() and getattr(self, "_frozen_backbone", False) and prec == 4
                and any(p.requires_grad for p in self.parameters())):
            prec = 2          # heads trained on frozen features: the tight mode, as in MM.forward_q
        # inference: the MLP heads, F.normalize and the mean over map types as ONE program launch (vecprog.hip) when the
        # heads fit it (Linear(<=256, 256): a ResNet18/34 trunk; a ResNet50 head, Linear(1024, dim), runs per op)
        fused = None
        if not train and not torch.is_grad_enabled() and opt.fused_vector_path and nmap <= 4 and all(
                m.seq[0].in_features <= 256 and m.seq[0].in_features % 32 == 0 and m.seq[0].out_features == 256
                and m.seq[3].out_features == 256 for m in self.dbimage_mlps):
            fused = VecProgram(b * ndb, db_map.device)
        if True:
            vecs = []
            for i in range(nmap):
                j = 0 if opt.share_dbfe is True else i
                x = db_map[:, :, i].reshape(b * ndb, c, h, w)       # view when possible; strides are honoured
                if u8:
                    x = x.permute(0, 2, 3, 1).unsqueeze(1)          # uint8 [n,1,h,w,3]: one "camera" per tile (contiguous again)
                if train:
                    # share_dbfe: ONE trunk over every map type (reference :69-72); each application keeps its own
                    # activations and tape (slot i), the parameter gradients of all of them accumulate
                    fe = self.dbimage_fes[j].fe
                    v = train_fns.TrunkFn.apply(train_fns.anchor_of(fe, self.dbimage_pools[j].p), x, fe, self.dbimage_pools[j], train_fns.MapSink(),
                                                prec, False, i if opt.share_dbfe is True else 0)[0]
                else:
                    if trunk_maps is not None and i in trunk_maps:
                        fpool = trunk_maps[i][1]
                    else:
                        fpool = self.final_pool_request(i)
                        self.dbimage_fes[j].forward_maps(x, prec=prec, final_pool=fpool)
                    v = fpool.gem
                if fused is not None:
                    # MLP + F.normalize of this map type; register 2 + i holds its vector
                    mlp = self.dbimage_mlps[j]
                    fused.linear(0, mlp._p0.get(), v)
                    fused.layernorm(0, mlp.seq[1], 0, relu=True)
                    fused.linear(2 + i, mlp._p3.get(), 0)
                    if opt.output_l2 is True:
                        fused.l2norm(2 + i, 2 + i)
                    vecs.append(v)
                    continue
                v = self.dbimage_mlps[j](v)
                if opt.output_l2 is True:
                    v = autograd_ops.l2normalize(v)
                vecs.append(v)
            if fused is not None:
                if nmap > 1:
                    wmean = torch.full((1,), 1.0 / nmap, device=db_map.device)
                    fused.wsum(2, [2 + i for i in range(nmap)], [wmean] * nmap)
                if opt.final_l2 is True:
                    fused.l2norm(2, 2)
                out = fused.store(2, out_rows)
                if defer_head is not None:
                    defer_head.append(fused)
                else:
                    fused.run()
                out = out.view(b, ndb, -1)
                if mode == 'cachetest':
                    out = out.view(b, -1)
                return {'embedding': out}
        if True:
            out = vecs[0] if nmap == 1 else autograd_ops.wsum(
                vecs, [torch.full((1,), 1.0 / nmap, device=vecs[0].device)] * nmap)
            out = out.view(b, ndb, -1)
            if mode == 'cachetest':
                out = out.view(b, -1)
            if opt.final_l2 is True:
                shp = out.shape
                out = autograd_ops.l2normalize(out.reshape(-1, shp[-1])).view(shp)
        return {'embedding': out}

    def forward(self, data_dict, mode: List[str]):
        if mode == 'db':
            return self.forward_db(data_dict)
        raise NotImplementedError
