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
