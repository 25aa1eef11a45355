"""-m gpu parity of the fused loss kernels (agplace_amd/losses.py) against the golden vectors
generated from the reference's compute_other_loss / nn.TripletMarginLoss and against the oracle."""
import types

import numpy as np
import pytest
import torch

from oracle import losses as olosses
from gpu_util import rel_l2

pytestmark = pytest.mark.gpu


def _inputs(golden, dev):
    lx = golden("losses")
    t = {k: torch.from_numpy(lx[k]).float().to(dev) for k in ("g_embed", "g_img", "g_vox", "a_embed", "q_en", "db_en")}
    for k in ("g_embed", "g_img", "g_vox", "a_embed"):
        t[k].requires_grad_(True)
    return lx, t


@pytest.mark.parametrize("typ", ["bce", "mse", "l1"])
def test_compute_other_loss_golden(dev, golden, typ):
    from agplace_amd import losses
    from agplace_amd.options import Options
    lx, t = _inputs(golden, dev)
    opt = Options(otherloss_type=typ, otherloss_weight=0.01)
    loss = losses.compute_other_loss({"embedding": t["g_embed"], "imagevec_org": t["g_img"], "voxvec_org": t["g_vox"]},
                                     {"embedding": t["a_embed"]}, {"query_eastnorth": t["q_en"], "db_eastnorth": t["db_en"]},
                                     positive_thd=10, negative_thd=25, opt=opt)
    loss.backward()
    ref = float(lx[f"other_{typ}"])
    assert abs(float(loss.detach()) - ref) < 1e-4 * abs(ref)
    for k in ("g_embed", "g_img", "g_vox", "a_embed"):
        # the reference's fp32 cdist uses the |x|^2+|y|^2-2xy form (1e-3-level rounding in its gradients)
        assert rel_l2(t[k].grad, torch.from_numpy(lx[f"other_{typ}_grad_{k}"])) < 2e-3, k


def test_compute_other_loss_vs_fp64_oracle(dev, golden):
    """Against the oracle in fp64 the kernels are exact to fp32 rounding (tighter than the fp32 reference)."""
    from agplace_amd import losses
    from agplace_amd.options import Options
    lx, t = _inputs(golden, dev)
    loss = losses.compute_other_loss({"embedding": t["g_embed"], "imagevec_org": t["g_img"], "voxvec_org": t["g_vox"]},
                                     {"embedding": t["a_embed"]}, {"query_eastnorth": t["q_en"], "db_eastnorth": t["db_en"]},
                                     opt=Options())
    loss.backward()
    o = {k: v.detach().cpu().double().requires_grad_(k in ("g_embed", "g_img", "g_vox", "a_embed")) for k, v in t.items()}
    ol = olosses.compute_other_loss({"embedding": o["g_embed"], "imagevec_org": o["g_img"], "voxvec_org": o["g_vox"]},
                                    {"embedding": o["a_embed"]}, {"query_eastnorth": o["q_en"], "db_eastnorth": o["db_en"]})
    ol.backward()
    assert abs(float(loss.detach()) - float(ol.detach())) < 1e-6 * abs(float(ol.detach()))
    for k in ("g_embed", "g_img", "g_vox", "a_embed"):
        assert rel_l2(t[k].grad, o[k].grad) < 1e-5, k


def test_triplet_loss_golden_and_error_paths(dev, golden):
    from agplace_amd import losses
    lx, t = _inputs(golden, dev)
    feats = torch.cat([t["g_embed"].detach().unsqueeze(1), t["a_embed"].detach()], 1).view(-1, 256).clone().requires_grad_(True)
    args = types.SimpleNamespace(criterion="triplet", train_batch_size=4, negs_num_per_query=10, margin=0.1)
    crit = torch.nn.TripletMarginLoss(margin=0.1, p=2, reduction="sum")
    loss = losses.compute_loss(args, crit, torch.from_numpy(lx["triplets"]).to(dev), feats)
    loss.backward()
    assert abs(float(loss.detach()) - float(lx["triplet_loss"])) < 1e-5 * abs(float(lx["triplet_loss"]))
    assert rel_l2(feats.grad, torch.from_numpy(lx["triplet_grad"])) < 1e-5
    args.criterion = "contrastive"
    with pytest.raises(ValueError):
        losses.compute_loss(args, crit, torch.from_numpy(lx["triplets"]).to(dev), feats)


@pytest.mark.parametrize("crit", ["sare_joint", "sare_ind"])
def test_sare_losses_golden_and_oracle(dev, golden, crit):
    """train.py:62-77 with model/functional.py:5-27 on agp_sare_loss: the reference's own loss and gradient (fixture) and,
    on a table with shared rows and widely spread distances, the fp64 oracle."""
    from agplace_amd import losses
    sx = golden("losses_sare")
    key = crit.split("_")[1]
    feats = torch.from_numpy(sx["feats"]).to(dev).requires_grad_(True)
    args = types.SimpleNamespace(criterion=crit, train_batch_size=3, negs_num_per_query=10, margin=0.1)
    loss = losses.compute_loss(args, getattr(losses, crit), torch.from_numpy(sx["triplets"]).to(dev), feats)
    loss.backward()
    assert abs(float(loss.detach()) - float(sx[key + "_loss"])) < 1e-5 * abs(float(sx[key + "_loss"]))
    assert rel_l2(feats.grad, torch.from_numpy(sx[key + "_grad"])) < 1e-5
    g = torch.Generator().manual_seed(5)
    f = torch.randn(40, 96, generator=g) * 0.8                       # squared distances up to ~300: exp() must not overflow
    trip = torch.randint(0, 40, (60, 3), generator=g)
    fd = f.to(dev).requires_grad_(True)
    args = types.SimpleNamespace(criterion=crit, train_batch_size=6, negs_num_per_query=10, margin=0.1)
    loss = losses.compute_loss(args, None, trip.to(dev), fd)
    loss.backward()
    fo = f.double().requires_grad_(True)
    ol = olosses.compute_loss_sare(trip, fo, 6, 10, crit)
    ol.backward()
    assert abs(float(loss.detach()) - float(ol.detach())) < 1e-5 * abs(float(ol.detach()))
    assert rel_l2(fd.grad, fo.grad) < 1e-5
    # the criterion functions themselves (what train.py:228-231 binds to criterion_triplet)
    q, p, n = f[0:1].to(dev), f[1:2].to(dev), f[2:12].to(dev)
    one = losses.sare_joint(q, p, n)
    ref = olosses.compute_loss_sare(torch.tensor([[0, 1, 2 + j] for j in range(10)]), f.double(), 1, 1, "sare_joint")
    assert abs(float(one) - float(ref)) < 1e-5 * abs(float(ref))
    if crit == "sare_joint":
        with pytest.raises(RuntimeError):
            losses.compute_loss(args, None, trip[:50].to(dev), fd)


def test_triplet_loss_inactive_and_shared_rows(dev):
    """Random triplets with repeated rows and a margin that leaves about half of them inactive."""
    from agplace_amd import losses
    g = torch.Generator().manual_seed(3)
    f = torch.nn.functional.normalize(torch.randn(48, 128, generator=g), dim=-1)
    trip = torch.randint(0, 48, (80, 3), generator=g)
    trip[:, 2] = (trip[:, 0] + 1 + torch.randint(0, 46, (80,), generator=g)) % 48      # negative != query
    fd = f.to(dev).requires_grad_(True)
    args = types.SimpleNamespace(criterion="triplet", train_batch_size=8, negs_num_per_query=10, margin=0.02)
    loss = losses.compute_loss(args, None, trip.to(dev), fd)
    loss.backward()
    fo = f.double().requires_grad_(True)
    ol = olosses.compute_loss(trip, fo, 8, 10, 0.02)
    ol.backward()
    assert abs(float(loss.detach()) - float(ol.detach())) < 1e-5 * abs(float(ol.detach()))
    assert rel_l2(fd.grad, fo.grad) < 1e-5


@pytest.mark.parametrize("crit", ["triplet", "sare_joint", "sare_ind"])
def test_compute_loss_against_the_references_train_compute_loss(dev, golden, crit):
    """agplace_amd.losses.compute_loss (the drop-in for train.py:51-79) against the reference's own function executed by
    make_golden.py section 11: loss and gradient for all three criteria."""
    from agplace_amd import losses
    g = golden("train_compute_loss")
    feats = torch.from_numpy(g["feats"]).to(dev).requires_grad_(True)
    args = types.SimpleNamespace(criterion=crit, train_batch_size=4, negs_num_per_query=10, margin=float(g["margin"]))
    critfn = torch.nn.TripletMarginLoss(margin=0.1, p=2, reduction="sum") if crit == "triplet" else getattr(losses, crit)
    loss = losses.compute_loss(args, critfn, torch.from_numpy(g["triplets"]).to(dev), feats)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[crit + "_loss"])) < 1e-5 * abs(float(g[crit + "_loss"]))
    assert rel_l2(feats.grad, torch.from_numpy(g[crit + "_grad"])) < 1e-5
