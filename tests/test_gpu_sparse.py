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


@pytest.mark.parametrize("prec,tol,ntd", [(3, 2e-5, 0), (2, 1e-3, 0), (3, 2e-5, 2), (2, 1e-3, 1)])
def test_minkfpn_matches_oracle(dev, prec, tol, ntd):
    """ntd > 0: the top-down path (transposed convolutions onto the finer levels + lateral 1x1, models/minkfpn.py:114-118)."""
    from agplace_amd.sparse import ECABasicBlock, MinkFPN, MinkGeM, SparseTensor
    from agplace_amd.sparse.modules import global_avg_pool
    params = osp.init_vox_params(seed=3, num_top_down=ntd)
    net = _load(MinkFPN(1, 256, ntd, 5, ECABasicBlock, [1, 1, 1], [64, 128, 256]), params, "vox_fe.").to(dev).eval()
    coords, feats = osp.synth_cloud(3, 150, extent=20, seed=2)
    coords[::7, 1:] += 0.4                                      # float coordinates are floored
    coords = torch.cat([coords, coords[:20]], 0)                # duplicates are merged
    feats = torch.ones((coords.shape[0], 1))
    x = SparseTensor.from_coords(feats.to(dev), coords.to(dev))
    top, maps = net(x, prec=prec)
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    otop, omaps = osp.minkfpn(osp.from_coords(feats.double(), coords), p64, "vox_fe.", num_top_down=ntd)
    assert [m.n for m in maps] == [len(m.coords) for m in omaps]
    assert [m.hi.shape[1] for m in maps] == ([64, 128, 256] if ntd == 0 else [64, 256, 256] if ntd == 1 else [256, 256, 256])
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


@pytest.mark.parametrize("ntd", [0, 2])
def test_minkfpn_training_forward_and_gradients(dev, ntd):
    """ntd = 2: with the top-down path (transposed convolutions, laterals).
    Train-mode MinkFPN (batch-statistics BatchNorm) + backward of sum_i <G_i, avg(out_i)> + <Gg, GeM(top)>:
    every kernel / BatchNorm / ECA parameter gradient against fp64 autograd through the oracle, with the
    product's ReLU pattern imposed and conditioning-scaled tolerances (see tests/test_gpu_train.py)."""
    from agplace_amd.sparse import ECABasicBlock, MinkFPN, SparseTensor
    from agplace_amd.sparse import train as st
    from agplace_amd.sparse.modules import global_avg_pool
    params = osp.init_vox_params(seed=8, num_top_down=ntd)
    net = _load(MinkFPN(1, 256, ntd, 5, ECABasicBlock, [1, 1, 1], [64, 128, 256]), params, "vox_fe.").to(dev).train()
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
        last = i == len(maps) - 1 - ntd                         # `top` is the last tensor of the top-down pass
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
        otop, omaps = osp.minkfpn(osp.from_coords(f, coords), p64, "vox_fe.", training=True, pattern=pattern, num_top_down=ntd)
        loss = sum((osp.global_avg(m) * Gm[i].double()).sum() for i, m in enumerate(omaps))
        loss = loss + (osp.mink_gem(otop, torch.tensor(3.0, dtype=torch.float64)) * Gg.double()).sum()
        loss.backward()
        return omaps, {k: v.grad.clone() for k, v in p64.items() if v.grad is not None}

    free_top, free_maps = osp.minkfpn(osp.from_coords(feats.double(), coords), p64, "vox_fe.", training=True, num_top_down=ntd)
    for m, om in zip(maps, free_maps):
        assert rel_l2(_feats(m), om.feats.detach()) < 3e-4
    _, ref = run(feats.double())
    gp = torch.Generator().manual_seed(2)
    _, refp = run(feats.double() * (1 + 1e-5 * torch.randn(feats.shape, generator=gp, dtype=torch.float64)))
    bad, checked = [], 0
    for name, prm in net.named_parameters():
        key = "vox_fe." + name
        if key not in ref:
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
    assert checked >= (40 if ntd == 0 else 44), checked
    assert int(net.bn0.bn.num_batches_tracked) == 1


# ---------------------------------------------------------------------------------------------------------------
# capacity mode (inference): the coordinate manager on the device, no host synchronisation

def _valid(sp):
    return int(sp.n_dev.item())


@pytest.mark.parametrize("kind", ["float32", "int64", "float64"])
def test_capacity_mode_levels_equal_the_oracle_and_the_exact_mode(dev, kind):
    """agp_sparse_build / agp_sparse_coarsen: keys, row counts, segment offsets, batch indices and kernel maps of every level are
    bit-equal to oracle/sparse.py (and to the exact-size host path) on a cloud with float coordinates, duplicates, negative
    coordinates, an empty sample and rows in random order."""
    from agplace_amd import ops
    from agplace_amd.sparse import SparseTensor
    coords, _ = osp.synth_cloud(4, 300, extent=28, seed=5)
    coords[:, 1:] -= 9.0
    coords = coords[coords[:, 0] != 2]                           # sample 2 has no points
    coords = torch.cat([coords, coords[:40]], 0)                 # duplicates
    if kind != "int64":
        coords[::5, 1:] += 0.37
    coords = coords[torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(1))]
    feats = torch.rand((coords.shape[0], 1), generator=torch.Generator().manual_seed(2))
    c_in = coords.to({"float32": torch.float32, "int64": torch.int64, "float64": torch.float64}[kind])
    if kind == "int64":
        coords = c_in.double()
    ws = ops.Workspace()
    cap = SparseTensor.from_coords_capacity(feats.to(dev), c_in.to(dev), 4, ws)
    ex = SparseTensor.from_coords(feats.to(dev), c_in.to(dev), nbatch=4)
    o = osp.from_coords(feats.double(), coords, nbatch=4)
    levels = []
    for lvl in range(4):
        n = _valid(cap)
        assert n == ex.n == len(o.coords)
        assert cap.n == coords.shape[0]                          # the capacity never changes
        assert torch.equal(cap.keys[:n], ex.keys) and bool((cap.keys[n:] == 0x7fffffffffffffff).all())
        assert cap.coords[:n].cpu().tolist() == [list(c) for c in o.coords]
        so_c, bi_c = cap.segments()
        so_e, bi_e = ex.segments()
        assert torch.equal(so_c, so_e) and torch.equal(bi_c[:n], bi_e)
        if lvl == 0:
            assert torch.equal(cap.f32[:n], ex.f32)              # duplicate rows averaged in input order
            assert rel_l2(cap.f32[:n], o.feats) < 1e-6
        for ks in (3,) if lvl else (3, 5):
            mc, me = cap.kernel_map(ks), ex.kernel_map(ks)
            got = mc[:, :n].clone()
            got[got == cap.n] = n                                # "absent" is the zero row: index cap here, n there
            assert torch.equal(got, me)
        levels.append(n)
        if lvl < 3:
            (cap2, mapc), (ex2, mape) = cap.strided(), ex.strided()
            n2 = _valid(cap2)
            got = mapc[:, :n2].clone()
            got[got == cap.n] = n
            assert torch.equal(got, mape)
            cap, ex = cap2, ex2
            o = osp.SpT(sorted({tuple([c[0]] + [(v // (2 * o.stride)) * (2 * o.stride) for v in c[1:]]) for c in o.coords}),
                        None, 2 * o.stride, 4) if hasattr(osp, "SpT") else o
    assert levels[0] > levels[1] > levels[2] >= levels[3] > 0
    assert int(cap.range_flag.item()) == 0


def test_capacity_mode_large_samples_and_duplicate_runs(dev):
    """csrc/coords.hip sorts one batch sample per workgroup: <= 16384 rows in LDS, more in global memory (same network); a voxel's
    duplicate points are averaged in input-row order whatever order the bucket scatter's atomics produced; a sample of more than
    65536 points cannot be numbered by the 16-bit slot field and is flagged (bit 1), not silently mis-sorted."""
    from agplace_amd import ops
    from agplace_amd.sparse import SparseTensor
    g = torch.Generator().manual_seed(11)
    big = torch.cat([torch.zeros(20000, 1), torch.randint(-90, 90, (20000, 3), generator=g).float()], 1)      # global-memory sort
    mid = torch.cat([torch.ones(9000, 1), torch.randint(-40, 40, (9000, 3), generator=g).float()], 1)         # LDS sort, P = 16384
    dup = torch.cat([torch.full((600, 1), 2.0), torch.randint(0, 3, (600, 3), generator=g).float()], 1)       # 27 voxels, ~22 points each
    coords = torch.cat([big, mid, dup], 0)
    coords = coords[torch.randperm(coords.shape[0], generator=g)]
    feats = torch.rand((coords.shape[0], 2), generator=g)
    ws = ops.Workspace()
    cap = SparseTensor.from_coords_capacity(feats.to(dev), coords.to(dev), 3, ws)
    ex = SparseTensor.from_coords(feats.to(dev), coords.to(dev), nbatch=3)
    for lvl in range(3):
        n = _valid(cap)
        assert n == ex.n
        assert torch.equal(cap.keys[:n], ex.keys) and bool((cap.keys[n:] == 0x7fffffffffffffff).all())
        assert torch.equal(cap.segments()[0], ex.segments()[0]) and torch.equal(cap.segments()[1][:n], ex.segments()[1])
        if lvl == 0:
            # fp32 sums in input-row order, then one division: the sequential definition, bit for bit
            keys = ((coords[:, 0].long() << 48) | ((coords[:, 1].long() + 32768) << 32) | ((coords[:, 2].long() + 32768) << 16)
                    | (coords[:, 3].long() + 32768))
            order = {int(k): i for i, k in enumerate(ex.keys.cpu().tolist())}
            acc = torch.zeros((n, 2), dtype=torch.float32)
            cnt = torch.zeros(n, dtype=torch.float32)
            for i, k in enumerate(keys.tolist()):
                acc[order[k]] += feats[i]
                cnt[order[k]] += 1
            assert torch.equal(cap.f32[:n].cpu(), acc / cnt[:, None])
            again = SparseTensor.from_coords_capacity(feats.to(dev), coords.to(dev), 3, ops.Workspace())
            assert torch.equal(again.f32[:n], cap.f32[:n])
        if lvl < 2:
            cap, ex = cap.strided()[0], ex.strided()[0]
    assert int(cap.range_flag.item()) == 0
    # tiny inputs: one point; one voxel hit five times; two samples of which the first is empty
    for pts, nb in ((torch.tensor([[0., 3., -2., 1.]]), 1), (torch.tensor([[0., 1., 1., 1.]] * 5), 1),
                    (torch.tensor([[1., 0., 0., 0.], [1., 5., 5., 5.], [1., 0., 0., 0.]]), 2)):
        ft = torch.arange(1, pts.shape[0] + 1, dtype=torch.float32).view(-1, 1)
        a = SparseTensor.from_coords_capacity(ft.to(dev), pts.to(dev), nb, ops.Workspace())
        e = SparseTensor.from_coords(ft.to(dev), pts.to(dev), nbatch=nb)
        n = _valid(a)
        assert n == e.n and torch.equal(a.keys[:n], e.keys) and torch.equal(a.segments()[0], e.segments()[0])
        assert torch.equal(a.f32[:n], e.f32)
        a2, e2 = a.strided()[0], e.strided()[0]
        assert _valid(a2) == e2.n and torch.equal(a2.keys[:e2.n], e2.keys)
    huge = torch.cat([torch.zeros(70000, 1), torch.randint(-200, 200, (70000, 3), generator=g).float()], 0 + 1)
    both = torch.cat([huge, mid], 0)
    sp = SparseTensor.from_coords_capacity(torch.ones((both.shape[0], 1), device=dev), both.to(dev), 2, ops.Workspace())
    assert int(sp.range_flag.item()) == 2
    so = sp.segments()[0].cpu().tolist()
    assert so[0] == so[1] == 0 and so[2] == _valid(sp) > 0          # the oversized sample is empty, the other one intact


def test_zplane_row_order_groups_planes_and_changes_no_bit(dev, monkeypatch):
    """SparseTensor.zperm (agp_sparse_zplane_perm): a permutation of every sample's rows that is sorted by z-plane and keeps the
    (x, y) order inside a plane, identity past the valid rows; a 27-tap convolution computed in that row order (tiles of one
    plane skip the taps towards a plane that does not exist) equals the natural order bit for bit, in both level modes."""
    from agplace_amd import ops
    from agplace_amd.sparse import SparseTensor
    from agplace_amd.sparse.modules import MinkowskiConvolution
    g = torch.Generator().manual_seed(3)
    rows = []
    for b, npts in enumerate((5000, 0, 3000)):                       # three z-planes, an empty sample, plenty of rows per plane
        if npts:
            rows.append(torch.cat([torch.full((npts, 1), float(b)), torch.randint(-40, 40, (npts, 2), generator=g).float(),
                                   torch.randint(0, 3, (npts, 1), generator=g).float()], 1))
    coords = torch.cat(rows, 0)
    feats = torch.ones((coords.shape[0], 1))
    conv = MinkowskiConvolution(64, 64, kernel_size=3).to(dev)
    for mode in ("capacity", "exact"):
        sp = (SparseTensor.from_coords_capacity(feats.to(dev), coords.to(dev), 3, ops.Workspace()) if mode == "capacity"
              else SparseTensor.from_coords(feats.to(dev), coords.to(dev), nbatch=3))
        n = _valid(sp) if mode == "capacity" else sp.n
        perm = sp.zperm().cpu().long()
        assert torch.equal(torch.sort(perm[:n])[0], torch.arange(n)) and torch.equal(perm[n:], torch.arange(n, sp.n))
        k = sp.keys.cpu()[perm[:n]]
        bz = (k >> 48) * 65536 + (k & 0xffff)                        # (sample, z): non-decreasing along the order ...
        assert bool((bz[1:] >= bz[:-1]).all())
        same = bz[1:] == bz[:-1]
        assert bool((perm[1:n][same] > perm[:n - 1][same]).all())    # ... and stable inside a plane
        x = sp.with_feats(torch.randn((sp.n + 1, 64), generator=g).half().to(dev))
        x.hi[sp.n].zero_()
        with torch.no_grad():
            a = conv(x, None, relu=False, prec=4, tag="t.a").hi[:n].clone()
            taps_z = x.tile_taps(3).clone()
            monkeypatch.setattr(SparseTensor, "zperm", lambda self: torch.arange(self.n, dtype=torch.int32, device=self.keys.device))
            x._maps.pop(("taps", 3))                                 # the tap sets belong to the row order: rebuilt for the natural one
            b_ = conv(x, None, relu=False, prec=4, tag="t.b").hi[:n].clone()
            taps_n = x.tile_taps(3).clone()
            x._maps.pop(("taps", 3))
            monkeypatch.undo()
        full = (n // 128)
        pc = lambda t: sum(bin(v & 0xffffffff).count("1") for v in t[:full].cpu().tolist())
        assert pc(taps_z) < 0.9 * pc(taps_n) and pc(taps_n) == 27 * full      # three planes: the outer two drop 9 taps each
        assert torch.equal(a, b_) and float(a.float().abs().max()) > 0


@pytest.mark.parametrize("ntd", [0, 2])
def test_capacity_mode_minkfpn_equals_exact_mode_and_flags_out_of_range(dev, ntd):
    """The whole voxel trunk in capacity mode against the oracle and against the exact-size path (the first layer sums its taps
    in another order there -- it searches its neighbours instead of reading a kernel map -- so equality is to rounding); a
    coordinate outside the 16-bit key range is flagged, not silently wrapped."""
    from agplace_amd import ops
    from agplace_amd.sparse import ECABasicBlock, MinkFPN, MinkGeM, SparseTensor
    from agplace_amd.sparse.modules import global_avg_pool
    params = osp.init_vox_params(seed=4, num_top_down=ntd)
    net = _load(MinkFPN(1, 256, ntd, 5, ECABasicBlock, [1, 1, 1], [64, 128, 256]), params, "vox_fe.").to(dev).eval()
    coords, feats = osp.synth_cloud(3, 400, extent=24, seed=8)
    coords = torch.cat([coords, coords[:30]], 0)
    feats = torch.ones((coords.shape[0], 1))
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    otop, omaps = osp.minkfpn(osp.from_coords(feats.double(), coords), p64, "vox_fe.", num_top_down=ntd)
    with torch.no_grad():
        for prec, tol in ((4, 1e-3), (3, 2e-5)):
            ws = ops.Workspace()
            tc, mc = net(SparseTensor.from_coords_capacity(feats.to(dev), coords.to(dev), 3, ws), prec=prec)
            te, me = net(SparseTensor.from_coords(feats.to(dev), coords.to(dev), nbatch=3), prec=prec)

            def f(sp, n):
                v = sp.hi[:n].float()
                return (v + sp.lo[:n].float()) if sp.lo is not None else v
            for a, b, om in zip(mc, me, omaps):
                n = _valid(a)
                assert n == b.n == len(om.coords)
                assert rel_l2(f(a, n), om.feats) < tol and rel_l2(f(a, n), f(b, n)) < tol
                assert float(a.hi[a.n].float().abs().max()) == 0          # the zero row sits at index capacity
                assert rel_l2(global_avg_pool(a), osp.global_avg(om)) < tol
            gem = MinkGeM().to(dev)
            assert rel_l2(gem(tc), osp.mink_gem(otop, torch.tensor(3.0, dtype=torch.float64))) < tol
            assert rel_l2(gem(tc), gem(te)) < tol
    bad = coords.clone()
    bad[5, 2] = 40000.0
    ws_ = ops.Workspace()
    sp = SparseTensor.from_coords_capacity(feats.to(dev), bad.to(dev), 3, ws_)
    assert int(sp.range_flag.item()) == 1
    # the flag describes the LAST build: a clean cloud through the same workspace clears it ...
    sp = SparseTensor.from_coords_capacity(feats.to(dev), coords.to(dev), 3, ws_)
    assert int(sp.range_flag.item()) == 0
    # ... and a batch index outside [0, nbatch) is flagged (it would index the per-sample tables out of bounds) and clamped
    bad = coords.clone()
    bad[7, 0] = 3.0
    sp = SparseTensor.from_coords_capacity(feats.to(dev), bad.to(dev), 3, ws_)
    seg_off_, bidx_ = sp._seg
    assert int(sp.range_flag.item()) == 1 and int(bidx_[:int(seg_off_[3].item())].max().item()) <= 2


def test_mm_forward_from_coords_is_hipgraph_capturable_and_matches_eager(dev):
    """MM.forward_q from query_image + coords / features (reference mm.py:86-93) captured in ONE hipGraph: the voxel branch makes
    no host synchronisation and no data-dependent allocation; the replayed outputs equal the eager ones bit for bit, also after the
    cloud changed in place (same number of points, different coordinates: the graph bakes in no row count)."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    from oracle import nets
    opt = Options(mfma_precision=4)
    torch.manual_seed(3)
    model = MM(opt=opt).to(dev).eval()
    data = nets.synth_query(2, 64, 128, opt, seed=3)
    for k in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse"):
        data.pop(k)
    c1, f1 = osp.synth_cloud(2, 500, extent=30, seed=1)
    c2, _ = osp.synth_cloud(2, 500, extent=14, seed=2)          # fewer distinct voxels: other row counts on every level
    d = {k: v.to(dev) for k, v in data.items()}
    d["coords"], d["features"] = c1.clone().to(dev), f1.to(dev)
    with torch.no_grad():
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2):
                model(d, mode="q")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            out = model(d, mode="q")
        for cloud in (c1, c2, c1):
            d["coords"].copy_(cloud.to(dev))
            g.replay()
            torch.cuda.synchronize()
            rep = {k: v.clone() for k, v in out.items()}
            with torch.cuda.stream(st):
                eager = model(d, mode="q")
            torch.cuda.synchronize()
            for k in rep:
                assert torch.equal(rep[k], eager[k]), k
            ref = nets.mm_forward_q({**{k: v.cpu() for k, v in data.items()}, "coords": cloud, "features": f1},
                                    {k: v.cpu() for k, v in model.state_dict().items()}, opt)
            assert rel_l2(rep["embedding"], ref["embedding"]) < 1e-3
        assert model.voxel_coords_in_range()
        # an eager forward checks the flag itself (the first call at once, every later call the call before it) and raises like the
        # exact-size path
        model2 = MM(opt=opt).to(dev).eval()
        dbad = dict(d)
        cbad = d["coords"].clone()
        cbad[3, 1] = 1e6
        dbad["coords"] = cbad
        with pytest.raises(ValueError, match="voxel coordinate"):
            model2(dbad, mode="q")
        model3 = MM(opt=opt).to(dev).eval()
        model3(d, mode="q")                      # a good batch,
        model3(dbad, mode="q")                   # a bad one in the middle of a run: found at the next call, whatever that call holds
        with pytest.raises(ValueError, match="the previous batch"):
            model3(d, mode="q")
        model3(d, mode="q")                      # (and the run goes on)
        assert model3.voxel_coords_in_range()
        # REPLAYED forwards publish their flag too (the sticky word and its pinned mirror are nodes of the graph): a bad cloud
        # copied into the static input is found by the non-blocking poll a few replays later, and by finish() at the latest.
        # Both kinds of violation: a batch index outside [0, batch size), a coordinate outside the key range.
        from agplace_amd import pair
        from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
        mdb = DBVanilla2D("db", opt.features_dim, opt=opt).to(dev).eval()
        tiles = {"db_map": torch.randn(d["query_image"].shape[0], 1, 3, 64, 64, device=dev)}
        dg = dict(d)
        dg["coords"] = d["coords"].clone()
        cbatch = d["coords"].clone()
        cbatch[5, 0] = 7.0
        cp = pair.CapturedPair(model3, mdb, dg, tiles, poll_every=1)
        for _ in range(3):
            cp.replay()
        cp.finish()
        good = {k: v.clone() for k, v in cp.out_q.items()}
        for bad_cloud in (cbatch, cbad):
            dg["coords"].copy_(bad_cloud)
            cp.replay()
            with pytest.raises(ValueError, match="voxel coordinate"):
                for _ in range(4):
                    cp.replay()
                    torch.cuda.synchronize()     # (the test makes the mirror land; a live loop just sees it a replay or two later)
            dg["coords"].copy_(d["coords"])
            cp.replay()
            cp.finish()                          # reported once; the run goes on and the good cloud embeds as before
            for k in good:
                assert torch.equal(good[k], cp.out_q[k]), k
        dg["coords"].copy_(cbatch)
        cp.replay()
        with pytest.raises(ValueError, match="voxel coordinate"):
            cp.finish()                          # the LAST replay of a loop: finish() finds it
        # a LEGAL cloud with one voxel a thousand cells away from the rest (other row counts, an isolated row on every level)
        # between good ones: no flag, and the good cloud embeds as before afterwards.  (Round 6: sequences like this one left
        # 0x01010101 in the sticky word or faulted -- a captured hipMemsetAsync node of agp_sparse_build beside eager memsets;
        # the library's capturable entry points issue no memset any more, csrc/coords.hip.)
        cfar = d["coords"].clone()
        cfar[3, 1] = 1000.0
        for _ in range(3):
            dg["coords"].copy_(cfar)
            cp.replay()
            cp.replay()
            cp.finish()
            far = {k: v.clone() for k, v in cp.out_q.items()}
            with torch.cuda.stream(cp.stream):
                eager_q, _ = pair.embed_pair(model3, mdb, dg, tiles)
            torch.cuda.synchronize()
            for k in far:
                assert torch.equal(far[k], eager_q[k]), k
            dg["coords"].copy_(d["coords"])
            cp.replay()
            cp.finish()
            for k in good:
                assert torch.equal(good[k], cp.out_q[k]), k
