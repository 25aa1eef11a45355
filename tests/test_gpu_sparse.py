"""-m gpu parity of the sparse-voxel branch (agplace_amd/sparse) against the dictionary-based CPU
oracle (oracle/sparse.py; MinkowskiEngine itself is not installed: parity unpinned, see there)."""
import pytest
import torch

from oracle import sparse as osp
from gpu_util import rel_l2

pytestmark = pytest.mark.gpu


def _load(module, params, prefix):
    sd = {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)
    return module


def _feats(sp):
    f = sp.hi[:sp.n].float()
    if sp.lo is not None:
        f = f + sp.lo[:sp.n].float()
    return f.cpu()


@pytest.mark.parametrize("prec,tol", [(3, 2e-5), (2, 1e-3)])
def test_minkfpn_matches_oracle(dev, prec, tol):
    from agplace_amd.sparse import ECABasicBlock, MinkFPN, MinkGeM, SparseTensor
    from agplace_amd.sparse.modules import global_avg_pool
    params = osp.init_vox_params(seed=3)
    net = _load(MinkFPN(1, 256, 0, 5, ECABasicBlock, [1, 1, 1], [64, 128, 256]), params, "vox_fe.").to(dev).eval()
    coords, feats = osp.synth_cloud(3, 150, extent=20, seed=2)
    coords[::7, 1:] += 0.4                                      # float coordinates are floored
    coords = torch.cat([coords, coords[:20]], 0)                # duplicates are merged
    feats = torch.ones((coords.shape[0], 1))
    x = SparseTensor.from_coords(feats.to(dev), coords.to(dev))
    top, maps = net(x, prec=prec)
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    otop, omaps = osp.minkfpn(osp.from_coords(feats.double(), coords), p64, "vox_fe.")
    assert [m.n for m in maps] == [len(m.coords) for m in omaps]
    for m, om in zip(maps, omaps):
        assert m.coords.cpu().tolist() == [list(c) for c in om.coords]
        assert rel_l2(_feats(m), om.feats) < tol
    gem = MinkGeM().to(dev)
    assert rel_l2(gem(top), osp.mink_gem(otop, torch.tensor(3.0, dtype=torch.float64))) < tol
    for m, om in zip(maps, omaps):
        assert rel_l2(global_avg_pool(m), osp.global_avg(om)) < tol


def test_state_dict_keys_match_reference_names(dev):
    from agplace_amd.sparse import ECABasicBlock, MinkFPN
    net = MinkFPN(1, 256, 0, 5, ECABasicBlock, [1, 1, 1], [64, 128, 256])
    assert set(net.state_dict().keys()) == set(k[len("vox_fe."):] for k in osp.init_vox_params())
    assert net.state_dict()["conv0.kernel"].shape == (125, 1, 64)
    assert net.state_dict()["blocks.1.0.downsample.0.kernel"].shape == (64, 128)
    assert net.state_dict()["blocks.2.0.eca.conv.weight"].shape == (1, 1, 5)


def test_empty_sample_and_negative_coordinates(dev):
    """A batch index without points pools to zero; negative coordinates floor-divide towards -inf."""
    from agplace_amd.sparse import ECABasicBlock, MinkFPN, SparseTensor
    from agplace_amd.sparse.modules import global_avg_pool
    params = osp.init_vox_params(seed=4)
    net = _load(MinkFPN(1, 256, 0, 5, ECABasicBlock, [1, 1, 1], [64, 128, 256]), params, "vox_fe.").to(dev).eval()
    coords, feats = osp.synth_cloud(1, 80, extent=10, seed=5)
    coords[:, 1:3] -= 5.0
    coords[:, 0] = 2                                            # samples 0 and 1 are empty
    x = SparseTensor.from_coords(feats.to(dev), coords.to(dev), nbatch=3)
    top, maps = net(x, prec=3)
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    otop, omaps = osp.minkfpn(osp.from_coords(feats.double(), coords, nbatch=3), p64, "vox_fe.")
    assert maps[0].coords.cpu().tolist() == [list(c) for c in omaps[0].coords]
    pooled = global_avg_pool(top).cpu()
    assert float(pooled[:2].abs().max()) == 0
    assert rel_l2(pooled[2], osp.global_avg(otop)[2]) < 2e-5


# ---------------------------------------------------------------------------- training of the branch
def _mask(sp_feats):
    return (sp_feats > 0).float().cpu()


def test_minkfpn_training_forward_and_gradients(dev):
    """Train-mode MinkFPN (batch-statistics BatchNorm) + backward of sum_i <G_i, avg(out_i)> + <Gg, GeM(top)>:
    every kernel / BatchNorm / ECA parameter gradient against fp64 autograd through the oracle, with the
    product's ReLU pattern imposed and conditioning-scaled tolerances (see tests/test_gpu_train.py)."""
    from agplace_amd.sparse import ECABasicBlock, MinkFPN, SparseTensor
    from agplace_amd.sparse import train as st
    from agplace_amd.sparse.modules import global_avg_pool
    params = osp.init_vox_params(seed=8)
    net = _load(MinkFPN(1, 256, 0, 5, ECABasicBlock, [1, 1, 1], [64, 128, 256]), params, "vox_fe.").to(dev).train()
    coords, feats = osp.synth_cloud(3, 400, extent=28, seed=9)
    x = SparseTensor.from_coords(feats.to(dev), coords.to(dev))
    tr = st.MinkFPNTrain(net)
    top, maps = tr.forward(x)
    p3 = torch.tensor([3.0], device=dev)
    g = torch.Generator().manual_seed(1)
    Gm = [torch.randn(3, m.hi.shape[1], generator=g) for m in maps]
    Gg = torch.randn(3, 256, generator=g)
    from agplace_amd.sparse import MinkGeM
    gem = MinkGeM().to(dev)
    gem_y = gem(top)
    gmaps = []
    for i, m in enumerate(maps):
        last = i == len(maps) - 1
        gmaps.append(st.seg_pool_bwd(m, gmean=Gm[i].to(dev), ggem=Gg.to(dev) if last else None,
                                     gem_y=gem_y if last else None, p=p3 if last else None))
    tr.backward(gmaps)
    # ---- oracle
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    for k, v in p64.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    pattern = {"vox_fe.relu0": _mask(_feats(tr.u0.saved[3]))}
    for i in range(3):
        pattern[f"vox_fe.relus.{i}"] = _mask(_feats(tr.down[i].saved[3]))
        b = tr.blocks[i][0]
        pattern[f"vox_fe.blocks.{i}.0.relu1"] = _mask(_feats(b.u1.saved[3]))
        pattern[f"vox_fe.blocks.{i}.0.relu2"] = _mask(_feats(b.saved[5]))

    def run(f):
        for v in p64.values():
            v.grad = None
        otop, omaps = osp.minkfpn(osp.from_coords(f, coords), p64, "vox_fe.", training=True, pattern=pattern)
        loss = sum((osp.global_avg(m) * Gm[i].double()).sum() for i, m in enumerate(omaps))
        loss = loss + (osp.mink_gem(otop, torch.tensor(3.0, dtype=torch.float64)) * Gg.double()).sum()
        loss.backward()
        return omaps, {k: v.grad.clone() for k, v in p64.items() if v.grad is not None}

    free_top, free_maps = osp.minkfpn(osp.from_coords(feats.double(), coords), p64, "vox_fe.", training=True)
    for m, om in zip(maps, free_maps):
        assert rel_l2(_feats(m), om.feats.detach()) < 3e-4
    _, ref = run(feats.double())
    gp = torch.Generator().manual_seed(2)
    _, refp = run(feats.double() * (1 + 1e-5 * torch.randn(feats.shape, generator=gp, dtype=torch.float64)))
    bad, checked = [], 0
    for name, prm in net.named_parameters():
        key = "vox_fe." + name
        if key not in ref or name.startswith("conv1x1s.1."):
            continue
        assert prm.grad is not None, name
        r = ref[key].reshape(prm.grad.shape)
        err = rel_l2(prm.grad, r)
        tol = max(1e-3, 3 * rel_l2(refp[key].reshape(r.shape), r))
        if not err < tol:
            bad.append((name, err, tol))
        checked += 1
    print("GRADERR " + " ".join(f"{n}:{e:.1e}/{t:.1e}" for n, e, t in bad))
    assert not bad, bad[:6]
    assert checked >= 40, checked
    assert int(net.bn0.bn.num_batches_tracked) == 1
