"""Training losses on the HIP kernels, drop-ins for

    train.py:51-79            compute_loss(args, criterion_triplet, triplets_local_indexes, features)
    compute_other_loss.py:56  compute_other_loss(feats_ground, feats_aerial, data_dict, positive_thd, negative_thd)

Both are torch.autograd.Functions over `agp_triplet_loss` / `agp_pairdist_loss`: forward and
backward each run a fixed, small number of launches with fixed-order reductions (the reference
issues ~60 ATen kernels per step for the same arithmetic).  All three criteria of train.py:226-231 are built:
"triplet" (the reference default, tools/options.py:189) on agp_triplet_loss, "sare_ind" / "sare_joint"
(model/functional.py:5-27) on agp_sare_loss.
"""
import torch

from . import _lib
from ._lib import check, ptr
from .options import get_options

_TYPES = {"bce": 0, "mse": 1, "l1": 2}


class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, triplets, margin):
        L = _lib.load()
        f = feats.contiguous().float()
        t = triplets.to(device=f.device, dtype=torch.int64).contiguous()
        nt = t.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        need = ctx.needs_input_grad[0]
        g = torch.empty_like(f) if need else None
        ws = torch.empty(L.agp_triplet_loss_workspace_floats(nt), dtype=torch.float32, device=f.device)
        check(L.agp_triplet_loss(ptr(f), f.shape[0], f.shape[1], ptr(t), nt, float(margin), ptr(loss), ptr(g), ptr(ws),
                                 _lib.stream()), "agp_triplet_loss")
        if need:
            ctx.save_for_backward(g)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return g * gout, None, None


class _SareFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, triplets, group):
        L = _lib.load()
        f = feats.contiguous().float()
        t = triplets.to(device=f.device, dtype=torch.int64).contiguous()
        nt = t.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        need = ctx.needs_input_grad[0]
        g = torch.empty_like(f) if need else None
        ws = torch.empty(L.agp_triplet_loss_workspace_floats(nt), dtype=torch.float32, device=f.device)
        check(L.agp_sare_loss(ptr(f), f.shape[0], f.shape[1], ptr(t), nt, int(group), ptr(loss), ptr(g), ptr(ws),
                              _lib.stream()), "agp_sare_loss")
        if need:
            ctx.save_for_backward(g)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return g * gout, None, None


def sare_ind(query, positive, negative):
    """model/functional.py:5-15 on the HIP kernel: query, positive [1,d]; negative [n,d] (n = 1 for sare_ind proper)."""
    feats = torch.cat([query, positive, negative], dim=0)
    n = negative.shape[0]
    trip = torch.stack([torch.zeros(n, dtype=torch.int64), torch.ones(n, dtype=torch.int64),
                        torch.arange(2, 2 + n, dtype=torch.int64)], dim=1)
    return _SareFn.apply(feats, trip, n)


def sare_joint(query, positive, negatives):
    """model/functional.py:17-27 (the same arithmetic as sare_ind with all the negatives at once)."""
    return sare_ind(query, positive, negatives)


def compute_loss(args, criterion_triplet, triplets_local_indexes, features):
    """train.py:51-79.  `criterion_triplet` is accepted for signature compatibility (the reference
    passes nn.TripletMarginLoss(margin=args.margin, p=2, reduction="sum"), sare_ind or sare_joint); a margin on it
    is honoured.  The SARE branches run the whole table in one agp_sare_loss call: groups of 10 rows for sare_joint
    (train.py:64, the reference hard-codes 10 negatives per query there), single rows for sare_ind."""
    if args.criterion in ("sare_ind", "sare_joint"):
        t = triplets_local_indexes.view(-1, 3)
        group = 10 if args.criterion == "sare_joint" else 1
        if group == 10 and t.shape[0] != args.train_batch_size * 10:
            raise RuntimeError(f"sare_joint expects train_batch_size * 10 triplets (train.py:64), got {t.shape[0]}")
        return _SareFn.apply(features, t, group) / (args.train_batch_size * args.negs_num_per_query)
    if args.criterion != "triplet":
        raise ValueError(f"criterion {args.criterion!r}: triplet | sare_ind | sare_joint (reference tools/options.py:189)")
    margin = getattr(criterion_triplet, "margin", None)
    if margin is None:
        margin = args.margin
    t = triplets_local_indexes.view(-1, 3)
    loss = _TripletFn.apply(features, t, margin)
    return loss / (args.train_batch_size * args.negs_num_per_query)


class _PairFn(torch.autograd.Function):
    """loss_sum / count of one compute_other_loss term; x [n,d], y [m,d] (may share rows upstream)."""

    @staticmethod
    def forward(ctx, x, y, ex, ey, pos_thd, neg_thd, typ):
        L = _lib.load()
        x, y = x.contiguous().float(), y.contiguous().float()
        ex, ey = ex.contiguous().float(), ey.contiguous().float()
        n, m, d = x.shape[0], y.shape[0], x.shape[1]
        dev = x.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        count = torch.empty(1, dtype=torch.float32, device=dev)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        ws = torch.empty(L.agp_pairdist_loss_workspace_floats(n, m), dtype=torch.float32, device=dev)
        check(L.agp_pairdist_loss(ptr(x), ptr(y), n, m, d, ptr(ex), ptr(ey), float(pos_thd), float(neg_thd), typ, ptr(loss),
                                  ptr(count), ptr(gx), ptr(gy), ptr(ws), _lib.stream()), "agp_pairdist_loss")
        ctx.save_for_backward(gx if gx is not None else loss, gy if gy is not None else loss, count)
        ctx.has = (gx is not None, gy is not None)
        return (loss / count).view(())

    @staticmethod
    def backward(ctx, gout):
        gx, gy, count = ctx.saved_tensors
        s = gout / count
        return (gx * s if ctx.has[0] else None, gy * s if ctx.has[1] else None, None, None, None, None, None)


def compute_other_loss(feats_ground, feats_aerial, data_dict, positive_thd=10, negative_thd=25, opt=None):
    """compute_other_loss.py:56-113: four distance-vs-geography terms (aerial-aerial and the ground
    embedding / image / voxel descriptors against [aerial; ground]), each a masked mean, summed with
    opt.otherloss_weight."""
    opt = opt or get_options()
    typ = _TYPES.get(opt.otherloss_type)
    if typ is None:
        raise NotImplementedError(opt.otherloss_type)
    g_embed, g_img, g_vox = feats_ground['embedding'], feats_ground['imagevec_org'], feats_ground['voxvec_org']
    a_embed = feats_aerial['embedding']
    b, ndb, c = a_embed.shape
    a_embed = a_embed.reshape(-1, c)
    en_g = data_dict['query_eastnorth']
    en_a = data_dict['db_eastnorth'].reshape(-1, 2)
    en_ag = torch.cat([en_a, en_g], dim=0)
    w = opt.otherloss_weight
    loss = _PairFn.apply(a_embed, a_embed, en_a, en_a, positive_thd, negative_thd, typ) * w
    for g in (g_embed, g_img, g_vox):
        loss = loss + _PairFn.apply(g, torch.cat([a_embed, g], dim=0), en_g, en_ag, positive_thd, negative_thd, typ) * w
    return loss
