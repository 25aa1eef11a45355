"""-m gpu parity tests, kernel level: every call goes product host code -> C ABI -> HIP kernel and
is compared with the CPU oracle / the reference-generated golden vectors.

Tolerances: split-bf16 (prec 3) results are fp32-class -> 1e-4 or tighter; the north_star bar for
model outputs is 1e-3 relative (tests/test_gpu_models.py); the fp16-activation modes (prec 2 / 4)
carry 2^-12 per stored element and are bounded at a few 1e-4 per op."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets, ode
from gpu_util import rel_l2, rel_max

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_pack_unpack_roundtrip_and_halo(dev):
    from agplace_amd import ops
    torch.manual_seed(0)
    for cl in (False, True):
        x = torch.randn(2, 64, 5, 7, device=dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        m = ops.pack_f32(x, 64, 1, 3)
        y = m.to_f32()
        assert y.shape == x.shape
        assert rel_max(y, x) < 2 ** -16
        assert float(m.hi[:, 0].abs().max()) == 0 and float(m.hi[:, :, -1].abs().max()) == 0
        assert float(m.lo[:, -1].abs().max()) == 0 and float(m.lo[:, :, 0].abs().max()) == 0
    # hi plane alone is the bf16 rounding of x
    assert torch.equal(m.hi[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).float(), x.to(torch.bfloat16).float())


CONV_CASES = [
    # cin, cout, k, stride, pad, h, w, n
    (64, 64, 3, 1, 1, 12, 20, 2),
    (64, 128, 3, 2, 1, 12, 20, 2),
    (64, 128, 1, 2, 0, 12, 20, 2),
    (128, 128, 3, 1, 1, 9, 7, 3),
    (256, 256, 3, 1, 1, 14, 10, 1),
    (256, 64, 1, 1, 0, 8, 8, 2),
    (64, 256, 1, 1, 0, 17, 19, 1),
    (128, 256, 3, 2, 1, 28, 30, 2),
    (512, 128, 1, 1, 0, 6, 6, 3),
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("prec", [3, 2, 4])
def test_conv2d_matches_oracle(dev, case, prec):
    from agplace_amd import ops
    cin, cout, k, stride, pad, h, w, n = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = 0.5 + torch.rand(cout, generator=g)
    shift = torch.randn(cout, generator=g) * 0.3
    ho, wo = ops.conv_out_size(h, k, stride, pad), ops.conv_out_size(w, k, stride, pad)
    res = torch.randn(n, cout, ho, wo, generator=g)
    ref = F.conv2d(x.double(), wt.double(), None, stride, pad) * scale.double().view(1, -1, 1, 1) \
        + shift.double().view(1, -1, 1, 1)
    xm = ops.pack_f32(x.to(dev), cin, 1, prec)
    cw = ops.ConvWeights(wt.to(dev), scale.to(dev), shift.to(dev), stride, pad)
    # 3: split-bf16 everywhere; 2: fp16 activations (2^-12 per element in, residual and out) with exact
    # weights; 4: fp16 weights as well
    tol = {3: 2e-5, 2: 4e-4, 4: 6e-4}[prec]
    # (a) plain conv + scale/shift
    out = ops.SplitMap.alloc(n, ho, wo, cout, 1, prec, dev)
    ops.conv2d(xm, cw, out, relu=False, prec=prec)
    assert rel_l2(out.to_f32(), ref) < tol
    # (b) + residual + ReLU, halo must stay zero
    rm = ops.pack_f32(res.to(dev), cout, 1, prec)
    out2 = ops.SplitMap.alloc(n, ho, wo, cout, 1, prec, dev)
    ops.conv2d(xm, cw, out2, residual=rm, relu=True, prec=prec)
    assert rel_l2(out2.to_f32(), torch.relu(ref + res.double())) < tol
    assert float(out2.hi[:, 0].abs().max()) == 0 and float(out2.hi[:, :, 0].abs().max()) == 0


@pytest.mark.parametrize("cin,cout,h,w,n", [(64, 64, 12, 20, 2), (128, 128, 9, 7, 3), (256, 256, 14, 10, 2), (64, 128, 33, 31, 2), (128, 64, 8, 40, 1)])
def test_conv2d_hi_only_is_one_bf16_product_of_the_hi_planes(dev, cin, cout, h, w, n):
    """agp_conv_desc.hi_only (the opt-in one-product data gradient of the training graph): a split-bf16 3x3 stride-1 conv whose MFMA
    operands are the hi planes alone -- equal, to fp32 accumulation and the stored pair's 2^-17, to an fp64 conv of exactly those
    bf16 values (+ the full two-plane residual); about 2^-9 per operand away from the three-product result; shapes the 3x3 stride-1
    kernel does not run are refused."""
    from agplace_amd import ops, _lib
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    res = torch.randn(n, cout, h, w, generator=g)
    xm = ops.pack_f32(x.to(dev), cin, 1, 3)
    cw = ops.ConvWeights(wt.to(dev), None, None, 1, 1)
    xh = xm.hi[:, 1:-1, 1:-1, :].float().permute(0, 3, 1, 2).double().cpu()
    wh = cw.planes(3)[0].float().double().cpu().permute(0, 3, 1, 2)              # planes: [cout][kh][kw][cin]
    assert float((xh - x.double()).abs().max()) > 0                 # (the hi plane alone is not x)
    ref1 = F.conv2d(xh, wh, None, 1, 1)
    ref3 = F.conv2d(x.double(), wt.double(), None, 1, 1)
    out = ops.SplitMap.alloc(n, h, w, cout, 1, 3, dev)
    ops.conv2d(xm, cw, out, relu=False, prec=3, hi_only=True)
    assert rel_l2(out.to_f32(), ref1) < 2e-5
    assert 2e-4 < rel_l2(out.to_f32(), ref3) < 6e-3
    rm = ops.pack_f32(res.to(dev), cout, 1, 3)
    out2 = ops.SplitMap.alloc(n, h, w, cout, 1, 3, dev)
    tiles = ops.conv_stat_tiles(xm, cw, out2, 3, hi_only=True)
    assert tiles > 0
    part = torch.empty((tiles, 2, cout), dtype=torch.float32, device=dev)
    ops.conv2d(xm, cw, out2, residual=rm, relu=False, prec=3, stat_partial=part, hi_only=True)
    want = ref1 + res.double()
    assert rel_l2(out2.to_f32(), want) < 2e-5
    assert float(out2.hi[:, 0].abs().max()) == 0 and float(out2.hi[:, :, 0].abs().max()) == 0
    sums = part.double().sum(0).cpu()                               # the statistics epilogue rides along: sum and sum of squares
    assert rel_l2(sums[0], want.sum((0, 2, 3))) < 1e-4 and rel_l2(sums[1], (want * want).sum((0, 2, 3))) < 1e-4
    # refused: a 1x1 conv, a stride-2 conv, an fp16 map
    for k, s_, pd in ((1, 1, 0), (3, 2, 1)):
        cwb = ops.ConvWeights(torch.randn(cout, cin, k, k).to(dev), None, None, s_, pd)
        ob = ops.SplitMap.alloc(n, ops.conv_out_size(h, k, s_, pd), ops.conv_out_size(w, k, s_, pd), cout, 1, 3, dev)
        with pytest.raises(RuntimeError, match="agp_conv2d_fwd"):
            ops.conv2d(xm, cwb, ob, prec=3, hi_only=True)
    x4 = ops.pack_f32(x.to(dev), cin, 1, 4)
    with pytest.raises(RuntimeError, match="agp_conv2d_fwd"):
        ops.conv2d(x4, cw, ops.SplitMap.alloc(n, h, w, cout, 1, 4, dev), prec=4, hi_only=True)


@pytest.mark.parametrize("prec", [3, 2])
@pytest.mark.parametrize("hw", [(32, 48), (33, 47), (64, 20)])
def test_stem_conv7x7(dev, hw, prec):
    from agplace_amd import ops
    h, w = hw
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    ref = torch.relu(F.conv2d(x.double(), wt.double(), None, 2, 3))
    xm = ops.pack_f32(x.to(dev), 4, 3, prec)
    cw = ops.ConvWeights(wt.to(dev), None, None, 2, 3, stem=True)
    ho, wo = ops.conv_out_size(h, 7, 2, 3), ops.conv_out_size(w, 7, 2, 3)
    out = ops.SplitMap.alloc(2, ho, wo, 64, 1, prec, dev)
    ops.conv2d(xm, cw, out, relu=True, prec=prec)
    assert rel_l2(out.to_f32(), ref) < (2e-5 if prec == 3 else 4e-4)


def test_maxpool_and_bcast_add(dev):
    from agplace_amd import ops
    g = torch.Generator().manual_seed(6)
    for h, w in ((12, 16), (13, 15)):
        x = torch.relu(torch.randn(2, 64, h, w, generator=g))
        xm = ops.pack_f32(x.to(dev), 64, 1, 3)
        ho, wo = ops.conv_out_size(h, 3, 2, 1), ops.conv_out_size(w, 3, 2, 1)
        out = ops.SplitMap.alloc(2, ho, wo, 64, 1, 3, dev)
        ops.maxpool3x3s2(xm, out)
        assert rel_max(out.to_f32(), F.max_pool2d(x, 3, 2, 1)) < 2 ** -16
        vec = torch.randn(2, 64, generator=g)
        o2 = ops.SplitMap.alloc(2, h, w, 64, 1, 3, dev)
        ops.bcast_add(xm, vec.to(dev), o2)
        assert rel_max(o2.to_f32(), x + vec[:, :, None, None]) < 2 ** -15


@pytest.mark.parametrize("c,h,w", [(64, 56, 84), (128, 9, 7), (256, 14, 84), (1024, 5, 6)])
@pytest.mark.parametrize("p", [3.0, 2.5])
def test_pool_map_mean_and_gem(dev, c, h, w, p):
    from agplace_amd import ops
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(3, c, h, w, generator=g) * 0.8
    xm = ops.pack_f32(x.to(dev), c, 1, 3)
    pt = torch.tensor([p], device=dev)
    mean, gem = ops.pool_map(xm, pt)
    assert rel_l2(mean, x.double().mean((2, 3))) < 1e-5
    assert rel_l2(gem, nets.gem(x.double(), torch.tensor([p], dtype=torch.float64)).flatten(1)) < 1e-5
    # fp32 dense entry point, both memory formats
    for xx in (x.to(dev), x.to(dev).contiguous(memory_format=torch.channels_last)):
        m2, g2 = ops.pool_f32(xx, pt, want_mean=True, want_gem=True)
        assert rel_l2(m2, x.double().mean((2, 3))) < 1e-5 and rel_l2(g2, gem) < 1e-5


def test_gem_modules_forward_backward_golden(dev, golden):
    from agplace_amd.network_mm.image_pooling import GeM as GeMmm
    from agplace_amd.network.image_pooling import GeM as GeMnet
    from agplace_amd.model.aggregation import GeM as GeMagg
    g = golden("gem")
    x = T(g["x"]).to(dev)
    for name, cls in (("mm", GeMmm), ("net", GeMnet), ("stg2", GeMmm), ("mm", GeMagg)):
        for p in (3.0, 2.5):
            tag = f"{name}_p{p}"
            m = cls(p=p).to(dev)
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            assert y.shape == tuple(g[tag + "_y"].shape)
            assert rel_l2(y, T(g[tag + "_y"])) < 1e-5
            (y * T(g[tag + "_gy"]).to(dev)).sum().backward()
            assert rel_l2(xi.grad, T(g[tag + "_gx"])) < 1e-4
            assert rel_l2(m.p.grad, T(g[tag + "_gp"])) < 1e-4


@pytest.mark.parametrize("b,k,n", [(5, 64, 256), (16, 128, 256), (33, 256, 256), (7, 1024, 256), (4, 256, 128), (20, 256, 512)])
def test_linear_matches_oracle(dev, b, k, n):
    from agplace_amd import ops
    g = torch.Generator().manual_seed(b + k)
    x, a1, a2 = (torch.randn(b, k, generator=g) for _ in range(3))
    w = torch.randn(n, k, generator=g) / k ** 0.5
    bias = torch.randn(n, generator=g)
    lw = ops.LinearWeights(w.to(dev), bias.to(dev))
    for act in (None, "relu", "tanh", "sigmoid"):
        y = ops.linear(x.to(dev), lw, act=act, add1=a1.to(dev), add2=a2.to(dev))
        ref = ode.fc((x + a1 + a2).double(), w.double(), bias.double(), act)
        assert y.shape == (b, n) and rel_l2(y, ref) < 2e-5


@pytest.mark.parametrize("method,step", [("euler", 0.1), ("euler", 0.3), ("midpoint", 0.25), ("rk4", 0.25), ("rk4", 0.1), ("rk4", 1.0)])
@pytest.mark.parametrize("act", ["relu", "tanh", "sigmoid", "id"])
def test_fcode_matches_oracle(dev, method, step, act):
    from agplace_amd import ops
    g = torch.Generator().manual_seed(11)
    for b in (3, 16, 37):
        x, a1 = torch.randn(b, 256, generator=g), torch.randn(b, 256, generator=g) * 0.5
        w = torch.randn(256, 256, generator=g) / 16
        bias = torch.randn(256, generator=g) * 0.1
        lw = ops.LinearWeights(w.to(dev), bias.to(dev))
        dts = ops.ode_grid_dts(step)
        y = ops.fcode(x.to(dev), lw, act, method, dts, add1=a1.to(dev))
        ref = ode.fcode((x + a1).double(), w.double(), bias.double(), act, method, step)
        assert rel_l2(y, ref) < 1e-4, (method, step, act, b)


def test_ffns_modules_against_reference_golden(dev, golden):
    from agplace_amd.network_mm.ffns import FC, FCODE
    from agplace_amd.network_mm.diff_block import DiffBlock
    from agplace_amd.options import Options
    g = golden("ffns")
    for act in ("id", "relu", "tanh", "sigmoid"):
        m = FC(64, 64, act).to(dev)
        m.load_state_dict({"fc.weight": T(g[f"fc_{act}_w"]), "fc.bias": T(g[f"fc_{act}_b"])})
        assert rel_l2(m(T(g["x64"]).to(dev)), T(g[f"fc_{act}_y"])) < 2e-5
    with torch.no_grad():
        for method, step in (("euler", 0.1), ("rk4", 0.25), ("midpoint", 0.3)):
            m = FCODE(256, "relu", opt=Options(odeint_method=method, odeint_size=step)).to(dev)
            m.load_state_dict({"func.func.fc.weight": T(g[f"fcode_{method}_w"]),
                               "func.func.fc.bias": T(g[f"fcode_{method}_b"])})
            assert rel_l2(m(T(g["x"]).to(dev)), T(g[f"fcode_{method}_y"])) < 1e-4
        gd = golden("diffblock")
        db = DiffBlock(256, 256, opt=Options(diff_type="fcode@relu_fcode@tanh")).to(dev)
        db.load_state_dict({k[len("diff_"):]: T(v) for k, v in gd.items() if k.startswith("diff_blocks")})
        assert rel_l2(db(T(gd["x"]).to(dev)), T(gd["diff_y"])) < 1e-4


def test_stage2_blocks_against_reference_golden(dev, golden):
    from agplace_amd.network_mm.stage2fuse_blockadd import Basic, FFNFuse, BasicBlock
    from agplace_amd.models_baseline.dbvanilla2d import MLP
    g = golden("stage2_blocks")
    with torch.no_grad():
        m = Basic(128).to(dev)
        m.load_state_dict({k[len("basic_"):]: T(v) for k, v in g.items()
                           if k.startswith("basic_") and k not in ("basic_x", "basic_y")})
        assert rel_l2(m(T(g["basic_x"]).to(dev)), T(g["basic_y"])) < 1e-4
        f = FFNFuse(64, "basic_basic").to(dev)
        f.load_state_dict({k[len("ffnfuse_"):]: T(v) for k, v in g.items() if k.startswith("ffnfuse_ffns")})
        assert rel_l2(f(T(g["ffnfuse_x"]).to(dev)), T(g["ffnfuse_y"])) < 1e-4
        blk = BasicBlock(64).to(dev).eval()
        blk.load_state_dict({k[len("block_"):]: T(v) for k, v in g.items()
                             if k.startswith("block_") and "_y_" not in k and k != "block_x"})
        y = blk(T(g["block_x"]).to(dev))
        assert y.shape == tuple(g["block_y_eval"].shape) and rel_l2(y, T(g["block_y_eval"])) < 1e-4
        gm = golden("db_mlp")
        mlp = MLP(256, 128).to(dev)
        mlp.load_state_dict({k[len("mlp_"):]: T(v) for k, v in gm.items() if k.startswith("mlp_")})
        assert rel_l2(mlp(T(gm["x"]).to(dev)), T(gm["y"])) < 1e-4


def test_rowwise_ops(dev):
    from agplace_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(9, 256, generator=g)
    gam, bet, res = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g), torch.randn(9, 256, generator=g)
    y = ops.layernorm(x.to(dev), gam.to(dev), bet.to(dev), 1e-5, relu=True, residual=res.to(dev))
    ref = torch.relu(F.layer_norm(x.double(), (256,), gam.double(), bet.double(), 1e-5) + res.double())
    assert rel_l2(y, ref) < 1e-5
    assert rel_l2(ops.l2normalize(x.to(dev)), F.normalize(x.double(), dim=-1)) < 1e-6
    z = torch.zeros(2, 8, device=dev)
    assert torch.equal(ops.l2normalize(z), z)          # x / max(|x|, 1e-12) -> 0, no NaN
    ws = [torch.tensor(v, device=dev) for v in (0.0, 1.0, 0.1)]
    xs = [torch.randn(9, 256, generator=g) for _ in range(3)]
    out = ops.wsum([t.to(dev) for t in xs], ws)
    assert rel_l2(out, xs[1].double() + 0.1 * xs[2].double()) < 1e-6
    seven = [torch.randn(4, 16, generator=g) for _ in range(7)]
    assert rel_l2(ops.wsum([t.to(dev) for t in seven]), sum(t.double() for t in seven)) < 1e-6


def test_netvlad_against_reference_golden(dev, golden):
    from agplace_amd.model.aggregation import NetVLAD
    g = golden("netvlad")
    for tag, K, D in (("k16_d64", 16, 64), ("k64_d256", 64, 256)):
        m = NetVLAD(clusters_num=K, dim=D).to(dev)
        m.load_state_dict({"conv.weight": T(g[tag + "_conv_w"]), "centroids": T(g[tag + "_centroids"])})
        y = m(T(g[tag + "_x"]).to(dev))
        assert y.shape == tuple(g[tag + "_y"].shape)
        assert rel_l2(y, T(g[tag + "_y"])) < 1e-4


def test_netvlad_matrix_pipe_forward_equals_the_valu_forward_and_the_oracle(dev):
    """agp_netvlad_fwd_mfma (exact fp32 MFMAs, several workgroups per image) against fp64 through oracle/nets.netvlad and against the
    one-workgroup-per-image VALU kernel: d in {128, 256}, cluster counts below 64, pixel counts that are not multiples of 64 or of 4
    (scalar staging loads), one image and many, with and without the input normalisation; other widths keep the VALU kernel."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(23)
    for n, K, D, h, w, norm in ((64, 64, 256, 14, 84, True), (3, 64, 128, 14, 6, True), (2, 37, 256, 3, 11, True), (1, 64, 256, 5, 13, False),
                                (5, 8, 128, 9, 9, True), (2, 16, 64, 5, 7, True)):
        x = torch.randn(n, D, h, w, generator=g) * (1.0 + 3.0 * torch.rand(n, 1, h, w, generator=g))
        cw = torch.randn(K, D, generator=g) * 2.0
        cc = torch.randn(K, D, generator=g)
        ref = nets.netvlad(x.double(), cw.double().view(K, D, 1, 1), cc.double(), normalize_input=norm)
        xd, cwd, ccd = x.to(dev), cw.to(dev), cc.to(dev)
        taken = ops._L().agp_netvlad_workspace_bytes(n, D, h * w, K) > 0
        assert taken == (D in (128, 256))
        y = ops.netvlad(xd, cwd, ccd, norm)
        ops.NETVLAD_MFMA = False
        try:
            yv = ops.netvlad(xd, cwd, ccd, norm)
        finally:
            ops.NETVLAD_MFMA = True
        assert y.shape == (n, K * D)
        assert rel_l2(y, ref) < 1e-5, (n, K, D, h, w, rel_l2(y, ref))
        assert rel_l2(y, yv) < 2e-6, (n, K, D, h, w, rel_l2(y, yv))
        if taken:
            assert torch.equal(y, ops.netvlad(xd, cwd, ccd, norm))          # fixed summation order: the same bits every call


def test_netvlad_backward_matches_autograd_through_the_oracle(dev):
    """agp_netvlad_bwd: dx, d conv.weight, d centroids of NetVLAD.forward against fp64 autograd through oracle/nets.netvlad (the
    reference's lines 126-146 restated), with and without the input normalisation, several pixel counts (full and ragged chunks)."""
    from agplace_amd.model.aggregation import NetVLAD
    g = torch.Generator().manual_seed(17)
    for K, D, h, w, norm in ((16, 64, 5, 7, True), (64, 128, 14, 6, True), (37, 256, 3, 11, True), (8, 64, 4, 8, False)):
        with torch.random.fork_rng(devices=[]):                   # (the module's own random init must not shift later tests' streams)
            m = NetVLAD(clusters_num=K, dim=D, normalize_input=norm).to(dev)
        cw = torch.randn(K, D, 1, 1, generator=g) * 2.0
        cc = torch.randn(K, D, generator=g)
        m.load_state_dict({"conv.weight": cw, "centroids": cc})
        x = torch.randn(3, D, h, w, generator=g)
        go = torch.randn(3, K * D, generator=g)
        xd = x.to(dev).requires_grad_(True)
        y = m(xd)
        (y * go.to(dev)).sum().backward()
        x64, w64, c64 = x.double().requires_grad_(True), cw.double().requires_grad_(True), cc.double().requires_grad_(True)
        y64 = nets.netvlad(x64, w64, c64, normalize_input=norm)
        (y64 * go.double()).sum().backward()
        assert rel_l2(y, y64) < 1e-5
        for name, got, ref in (("dx", xd.grad, x64.grad), ("dw", m.conv.weight.grad, w64.grad), ("dc", m.centroids.grad, c64.grad)):
            assert rel_l2(got, ref) < 2e-4, (K, D, name, rel_l2(got, ref))


# ----------------------------------------------------------------------- backward kernels
@pytest.mark.parametrize("method,step", [("euler", 0.1), ("midpoint", 0.3), ("rk4", 0.25), ("rk4", 0.1)])
@pytest.mark.parametrize("act", ["relu", "tanh", "sigmoid", "id"])
def test_fcode_backward_matches_autograd_oracle(dev, method, step, act):
    """Discretise-then-optimise gradients (reference: plain odeint + autograd, ffns.py:84)."""
    from agplace_amd.network_mm.ffns import FCODE
    from agplace_amd.options import Options
    g = torch.Generator().manual_seed(21)
    # the module's weights come from a fixed stream of their own (whatever ran before this test): the kernels agree with fp64
    # autograd to ~2e-6 unless a pre-activation within ~1e-5 of zero lands on the other side of the ReLU kink in the two
    # arithmetics, which moves the weight gradient by up to 3e-3 (seeds 0 / 5 / 8 of tools/ubench/fcode_seedscan.py do, this one not)
    torch.manual_seed(1)
    for b in (5, 16, 35):
        m = FCODE(256, act, opt=Options(odeint_method=method, odeint_size=step)).to(dev)
        x = torch.randn(b, 256, generator=g)
        a1 = torch.randn(b, 256, generator=g) * 0.3
        G = torch.randn(b, 256, generator=g)
        xd, a1d = x.to(dev).requires_grad_(True), a1.to(dev).requires_grad_(True)
        y = m(xd, add1=a1d)
        (y * G.to(dev)).sum().backward()
        W = m.func.func.fc.weight.detach().cpu().double().requires_grad_(True)
        B = m.func.func.fc.bias.detach().cpu().double().requires_grad_(True)
        xr = x.double().requires_grad_(True)
        yr = ode.fcode(xr + a1.double(), W, B, act, method, step)
        (yr * G.double()).sum().backward()
        # bar: 1e-3 (north star).  ReLU kinks make the gradient discontinuous in the state, so a
        # 1e-5 forward difference can flip act' of a few elements; typical error is ~1e-4.
        assert rel_l2(y, yr) < 1e-4
        assert rel_l2(xd.grad, xr.grad) < 1e-3 and rel_l2(a1d.grad, xr.grad) < 1e-3
        assert rel_l2(m.func.func.fc.weight.grad, W.grad) < 1e-3
        assert rel_l2(m.func.func.fc.bias.grad, B.grad) < 1e-3


@pytest.mark.parametrize("b,k,n", [(5, 64, 256), (16, 128, 256), (33, 256, 128), (7, 1024, 256), (20, 256, 256)])
@pytest.mark.parametrize("act", [None, "relu", "tanh", "sigmoid"])
def test_linear_backward_matches_autograd_oracle(dev, b, k, n, act):
    from agplace_amd.network_mm.ffns import FC
    g = torch.Generator().manual_seed(b + n)
    m = FC(k, n, act).to(dev)
    x, G = torch.randn(b, k, generator=g), torch.randn(b, n, generator=g)
    xd = x.to(dev).requires_grad_(True)
    y = m(xd)
    (y * G.to(dev)).sum().backward()
    W = m.fc.weight.detach().cpu().double().requires_grad_(True)
    B = m.fc.bias.detach().cpu().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    yr = ode.fc(xr, W, B, act)
    (yr * G.double()).sum().backward()
    assert rel_l2(xd.grad, xr.grad) < 1e-3
    assert rel_l2(m.fc.weight.grad, W.grad) < 1e-3 and rel_l2(m.fc.bias.grad, B.grad) < 1e-3


def test_basic_mlp_and_normalize_backward(dev):
    from agplace_amd.network_mm.stage2fuse_blockadd import Basic
    from agplace_amd import autograd_ops
    torch.manual_seed(5)
    m = Basic(256).to(dev)
    for p in m.parameters():
        p.data += 0.05 * torch.randn_like(p)
    x, G = torch.randn(9, 256), torch.randn(9, 256)
    xd = x.to(dev).requires_grad_(True)
    y = autograd_ops.l2normalize(m(xd))
    (y * G.to(dev)).sum().backward()
    params = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.double().requires_grad_(True)
    yr = F.normalize(nets.basic_mlp(xr, params, ""), dim=-1)
    (yr * G.double()).sum().backward()
    assert rel_l2(y, yr) < 1e-4 and rel_l2(xd.grad, xr.grad) < 1e-3
    for name, p in m.named_parameters():
        assert rel_l2(p.grad, params[name].grad) < 1e-3, name


@pytest.mark.parametrize("prec", [2, 3])
def test_device_input_pipeline_u8_cameras(dev, prec):
    """ToTensor + Normalize + width-concat + NHWC4 packing of uint8 camera tiles in one kernel
    (reference datasets_ws_nuscenes.py:608-634 on CPU workers, then image_fe.py:98)."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(9)
    img = torch.randint(0, 256, (2, 3, 10, 12, 3), generator=g, dtype=torch.uint8)
    m = ops.pack_cameras_u8(img.to(dev), prec)
    assert (m.n, m.h, m.w, m.c, m.pad) == (2, 10, 36, 4, 3)
    mean = torch.tensor(ops.IMAGENET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(ops.IMAGENET_STD).view(1, 1, 3, 1, 1)
    ref = (img.permute(0, 1, 4, 2, 3).float() / 255 - mean) / std            # [n,cam,3,h,w]
    ref = torch.cat([ref[:, c] for c in range(3)], dim=-1)                    # width concat -> [n,3,h,3w]
    full = m.hi.float() + (m.lo.float() if m.lo is not None else 0)           # [n, h+6, w+6, 4]
    got = full[:, 3:-3, 3:-3].permute(0, 3, 1, 2).cpu()
    tol = 1e-5 if prec == 3 else 6e-4
    assert rel_l2(got[:, :3], ref) < tol
    assert float(got[:, 3].abs().max()) == 0
    assert float(full[:, :3].abs().max()) == 0 and float(full[:, :, :3].abs().max()) == 0    # halo untouched


@pytest.mark.parametrize("prec", [2, 4])
@pytest.mark.parametrize("n,hw", [(2, (64, 96)), (1, (50, 70)), (3, (224, 224)), (1, (224, 1344)), (2, (37, 45)), (1, (100, 263)),
                                  (5, (30, 520)), (1, (16, 24)), (2, (14, 300)), (70, (40, 72))])
def test_fused_stem_pool_equals_conv_then_maxpool(dev, n, hw, prec):
    """agp_stem_pool_fwd (7x7/2 conv + BN + ReLU + MaxPool2d(3,2,1) in one kernel, 16x16 conv blocks with
    recomputed seams) against the two separate kernels: bit-identical pooled maps, zero halo."""
    from agplace_amd import ops
    h, w = hw
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    scale, shift = 0.5 + torch.rand(64, generator=g), 0.3 * torch.randn(64, generator=g)
    scale[::3] *= -1                      # negative BatchNorm scales (gamma < 0): the walking stem kernel moves the sign into W
    scale[5] = 0.0
    xm = ops.pack_f32(x.to(dev), 4, 3, prec)
    cw = ops.ConvWeights(wt.to(dev), scale.to(dev), shift.to(dev), 2, 3, stem=True)
    h1, w1 = ops.conv_out_size(h, 7, 2, 3), ops.conv_out_size(w, 7, 2, 3)
    h2, w2 = ops.conv_out_size(h1, 3, 2, 1), ops.conv_out_size(w1, 3, 2, 1)
    s = ops.SplitMap.alloc(n, h1, w1, 64, 1, prec, dev)
    ops.conv2d(xm, cw, s, relu=True, prec=prec)
    ref = ops.SplitMap.alloc(n, h2, w2, 64, 1, prec, dev)
    ops.maxpool3x3s2(s, ref)
    got = ops.SplitMap.alloc(n, h2, w2, 64, 1, prec, dev)
    ops.stem_pool(xm, cw, got, prec=prec)
    assert torch.equal(got.hi, ref.hi)
    oracle = F.max_pool2d(torch.relu(F.conv2d(x.double(), wt.double(), None, 2, 3) * scale.double().view(1, -1, 1, 1)
                                     + shift.double().view(1, -1, 1, 1)), 3, 2, 1)
    assert rel_l2(got.to_f32(), oracle) < 6e-4


# ------------------------------------------------------------- F16W2: e4m3 lo plane (agp_conv_desc.w_q8)
def _q8_plane_host(w, hi):
    """torch restatement of agp_conv_w_q8_prepare: w [cout][3][3][cin] fp32, hi = fp16(w)."""
    import math
    lo = w - hi.float()
    m = float(lo.abs().max())
    exp = 0 if m == 0.0 else int(math.floor(math.log2(448.0 / m)))
    n, cc = w.shape[0], w.shape[3] // 32
    ph = (lo * 2.0 ** exp).view(n, 3, 3, cc, 32).permute(0, 1, 3, 2, 4).reshape(n, 9 * cc, 2, 2, 8)   # [n][phase][ks][lh][e]
    pr = ph.reshape(n, 9 * cc // 2, 2, 2, 2, 8).permute(0, 1, 4, 2, 3, 5).contiguous()               # [n][pair][lh][tap][ks][e]
    return pr.to(torch.float8_e4m3fn).view(torch.uint8).reshape(n, -1), exp


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 64), (256, 128)])
def test_conv_w_q8_plane_matches_host_construction(dev, cin, cout):
    from agplace_amd import ops, _lib
    g = torch.Generator().manual_seed(cin + cout)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * torch.rand(cout, 1, 1, 1, generator=g) / (cin * 9) ** 0.5).to(dev)
    cw = ops.ConvWeights(wt, None, None, 1, 1)
    plane, exp = cw.q8()
    ref, rexp = _q8_plane_host(cw.w, cw.planes(_lib.PREC_F16W2)[0])
    assert exp == rexp
    assert torch.equal(plane.cpu(), ref.cpu())
    # not a 3x3 stride-1 conv / cin not a multiple of 64: no plane, the fp16 lo product runs
    assert ops.ConvWeights(wt, None, None, 2, 1).q8() is None
    assert ops.ConvWeights(wt[:, :32].contiguous(), None, None, 1, 1).q8() is None


@pytest.mark.parametrize("case", [(64, 64, 12, 20, 2), (128, 128, 9, 7, 3), (256, 256, 14, 10, 1), (192, 64, 9, 7, 3), (64, 128, 33, 31, 2)])
def test_conv2d_lo_fp8_equals_fp16_lo_product(dev, case, monkeypatch):
    """The e4m3 lo product (block-scaled fp8 MFMA, activations converted in registers) against the fp16 lo product
    and the fp64 oracle: same error as F16W2, clearly better than dropping the lo product (F16)."""
    from agplace_amd import ops
    cin, cout, h, w, n = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    res = torch.randn(n, cout, h, w, generator=g)
    ref = torch.relu(F.conv2d(x.double(), wt.double(), None, 1, 1) + res.double())
    xm = ops.pack_f32(x.to(dev), cin, 1, 2)
    rm = ops.pack_f32(res.to(dev), cout, 1, 2)
    cw = ops.ConvWeights(wt.to(dev), None, None, 1, 1)
    outs = {}
    for name, lo8, prec in (("fp16lo", False, 2), ("fp8lo", True, 2), ("nolo", False, 4)):
        monkeypatch.setattr(ops, "LO_FP8", lo8)
        xin, rin = (xm, rm) if prec == 2 else (ops.pack_f32(x.to(dev), cin, 1, 4), ops.pack_f32(res.to(dev), cout, 1, 4))
        out = ops.SplitMap.alloc(n, h, w, cout, 1, prec, dev)
        ops.conv2d(xin, cw, out, residual=rin, relu=True, prec=prec)
        outs[name] = out.to_f32()
        assert float(out.hi[:, 0].abs().max()) == 0 and float(out.hi[:, :, -1].abs().max()) == 0
    e16, e8, e0 = (rel_l2(outs[k], ref) for k in ("fp16lo", "fp8lo", "nolo"))
    assert e8 < 1.02 * e16 + 1e-6 and e8 < 4e-4
    assert e0 > 1.05 * e8                                       # the lo product matters and the fp8 form delivers it
    assert rel_l2(outs["fp8lo"], outs["fp16lo"].double()) < 1.5e-4   # both round to fp16 maps: differences are 1-ulp flips


@pytest.mark.parametrize("prec", [2, 3])
@pytest.mark.parametrize("shape", [(3, 3, 10, 16), (2, 3, 7, 12), (2, 3, 6, 10), (1, 2, 5, 8)])
def test_pack_stem_input_fast_and_generic_paths(dev, prec, shape):
    """fp32 NCHW -> NHWC4 (+3-pixel halo): the four-pixels-per-thread kernel (unit pixel stride, w % 4 == 0, aligned)
    and the generic kernel (here: w % 4 != 0, or a strided view) give the planes torch's own casts give."""
    from agplace_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    for strided in (False, True):
        x = torch.randn(n, c, h, 2 * w if strided else w, generator=g).to(dev)
        xv = x[..., ::2] if strided else x
        m = ops.pack_f32(xv, 4, 3, prec)
        want = xv.permute(0, 2, 3, 1)
        if prec == 2:
            assert m.lo is None
            assert torch.equal(m.hi[:, 3:-3, 3:-3, :c], want.half())
        else:
            hi = want.to(torch.bfloat16)
            assert torch.equal(m.hi[:, 3:-3, 3:-3, :c], hi)
            assert torch.equal(m.lo[:, 3:-3, 3:-3, :c], (want - hi.float()).to(torch.bfloat16))
        assert float(m.hi[:, 3:-3, 3:-3, c:].abs().max()) == 0
        assert float(m.hi[:, :3].abs().max()) == 0 and float(m.hi[:, :, -3:].abs().max()) == 0


@pytest.mark.parametrize("chan", [(64, 64), (128, 128), (256, 256), (64, 128)])
def test_conv2d_grouped_equals_separate_launches(dev, chan):
    """agp_conv2d_fwd_grouped: 2-4 F16 3x3 stride-1 convs of one channel shape (different images, map sizes, weights,
    scale/shift, residual, relu) as ONE launch -- bit-identical to separate launches and correct against fp64."""
    from agplace_amd import ops
    cin, cout = chan
    g = torch.Generator().manual_seed(cin + 3 * cout)
    probs = [(3, 14, 40, True, True), (5, 9, 9, False, False), (1, 30, 17, True, False), (2, 7, 130, False, True)]
    for nprob in (2, 3, 4):
        jobs, refs, sep = [], [], []
        for (n, h, w, use_res, relu) in probs[:nprob]:
            x = torch.randn(n, cin, h, w, generator=g)
            wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
            scale, shift = 0.5 + torch.rand(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
            res = torch.randn(n, cout, h, w, generator=g)
            ref = F.conv2d(x.double(), wt.double(), None, 1, 1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
            if use_res:
                ref = ref + res.double()
            refs.append(torch.relu(ref) if relu else ref)
            xm = ops.pack_f32(x.to(dev), cin, 1, 4)
            rm = ops.pack_f32(res.to(dev), cout, 1, 4) if use_res else None
            cw = ops.ConvWeights(wt.to(dev), scale.to(dev), shift.to(dev), 1, 1)
            jobs.append((xm, cw, ops.SplitMap.alloc(n, h, w, cout, 1, 4, dev), rm, relu))
            o = ops.SplitMap.alloc(n, h, w, cout, 1, 4, dev)
            ops.conv2d(xm, cw, o, residual=rm, relu=relu, prec=4)
            sep.append(o)
        outs = ops.conv2d_grouped(jobs, 4)
        torch.cuda.synchronize()
        for o, s, r in zip(outs, sep, refs):
            assert torch.equal(o.hi, s.hi)
            assert rel_l2(o.to_f32(), r) < 6e-4
            assert float(o.hi[:, 0].abs().max()) == 0 and float(o.hi[:, :, -1].abs().max()) == 0


def test_conv2d_grouped_falls_back_for_other_groups(dev):
    """Groups the one-launch kernel cannot take (other precision, mixed channel shapes, a strided conv) run as separate
    launches inside the library with the same results."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(5)
    for prec, shapes in ((2, [(64, 64, 1), (64, 64, 1)]), (4, [(64, 64, 1), (64, 128, 1)]), (4, [(64, 128, 2), (64, 128, 2)])):
        jobs, sep = [], []
        for cin, cout, stride in shapes:
            x = torch.randn(2, cin, 12, 16, generator=g)
            wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
            xm = ops.pack_f32(x.to(dev), cin, 1, prec)
            cw = ops.ConvWeights(wt.to(dev), None, None, stride, 1)
            ho, wo = ops.conv_out_size(12, 3, stride, 1), ops.conv_out_size(16, 3, stride, 1)
            jobs.append((xm, cw, ops.SplitMap.alloc(2, ho, wo, cout, 1, prec, dev), None, True))
            o = ops.SplitMap.alloc(2, ho, wo, cout, 1, prec, dev)
            ops.conv2d(xm, cw, o, relu=True, prec=prec)
            sep.append(o)
        for o, s in zip(ops.conv2d_grouped(jobs, prec), sep):
            assert torch.equal(o.hi, s.hi)


@pytest.mark.parametrize("case", [(64, 64, 56, 338, 3), (128, 128, 28, 170, 5), (256, 256, 14, 86, 9)])
def test_conv2d_f16_large_maps_match_oracle(dev, case):
    """The inference hot kernel (igemm_kxr2.hip) at the bench workload's layer shapes (several row tiles per image,
    every K depth): against fp64 on a sample of output positions."""
    from agplace_amd import ops
    cin, cout, h, w, n = case
    g = torch.Generator().manual_seed(h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    res = torch.randn(n, cout, h, w, generator=g)
    xm, rm = ops.pack_f32(x.to(dev), cin, 1, 4), ops.pack_f32(res.to(dev), cout, 1, 4)
    cw = ops.ConvWeights(wt.to(dev), None, None, 1, 1)
    out = ops.SplitMap.alloc(n, h, w, cout, 1, 4, dev)
    ops.conv2d(xm, cw, out, residual=rm, relu=True, prec=4)
    got = out.to_f32().cpu()
    # first, middle and last image in full
    for i in sorted({0, n // 2, n - 1}):
        ref = torch.relu(F.conv2d(x[i:i + 1].double(), wt.double(), None, 1, 1) + res[i:i + 1].double())
        assert rel_l2(got[i:i + 1], ref) < 6e-4, i
    assert float(out.hi[:, 0].abs().max()) == 0 and float(out.hi[:, -1].abs().max()) == 0
    assert float(out.hi[:, :, 0].abs().max()) == 0 and float(out.hi[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize("amp", [300.0, 3000.0])
@pytest.mark.parametrize("prec", [2, 4])
def test_conv2d_large_activations(dev, amp, prec):
    """Activations far beyond e4m3's +-448 (an unnormalised ResNet50 reaches 10^3 - 10^4): the F16W2 lo product converts
    them to fp8 in registers and must SATURATE there (MODE.FP16_OVFL), not produce NaN -- round 1 returned garbage."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(int(amp))
    x = torch.randn(2, 256, 14, 14, generator=g) * amp
    wt = torch.randn(256, 256, 3, 3, generator=g) / (256 * 9) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    xm = ops.pack_f32(x.to(dev), 256, 1, prec)
    cw = ops.ConvWeights(wt.to(dev), None, None, 1, 1)
    out = ops.SplitMap.alloc(2, 14, 14, 256, 1, prec, dev)
    ops.conv2d(xm, cw, out, relu=False, prec=prec)
    assert rel_l2(out.to_f32(), ref) < 6e-4


def test_fp16_maps_saturate_finite_and_are_counted(dev):
    """fp16 feature maps saturate at +-65504 instead of overflowing (include/agplace_hip.h); ops.count_saturated is the
    opt-in diagnostic for checkpoints whose activations could get there."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 64, 8, 8, generator=g) * 1e5
    wt = torch.randn(64, 64, 3, 3, generator=g) / 24
    for prec in (2, 4):
        xm = ops.pack_f32(x.to(dev), 64, 1, prec)
        assert ops.count_saturated(xm) > 0
        out = ops.SplitMap.alloc(1, 8, 8, 64, 1, prec, dev)
        ops.conv2d(xm, ops.ConvWeights(wt.to(dev), None, None, 1, 1), out, relu=False, prec=prec)
        o = out.to_f32()
        assert torch.isfinite(o).all() and float(o.abs().max()) <= 65504.0
        assert ops.count_saturated(out) > 0
    small = ops.pack_f32(torch.randn(1, 64, 8, 8, generator=g).to(dev), 64, 1, 4)
    assert ops.count_saturated(small) == 0


def test_bcast_add_takes_a_strided_vector(dev):
    """ops.linear returns column slices of a padded buffer when the output width is not a multiple of its tile (a 64-wide
    stage-2 block): bcast_add must honour the strides (found by the reference-generated fusion fixture)."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 64, 5, 7, generator=g).to(dev)
    buf = torch.randn(3, 256, generator=g).to(dev)
    vec = buf[:, :64]
    assert not vec.is_contiguous()
    xm = ops.pack_f32(x, 64, 1, 3)
    out = ops.SplitMap.alloc(3, 5, 7, 64, 1, 3, dev)
    ops.bcast_add(xm, vec, out)
    ref = x + vec[:, :, None, None]
    assert float((out.to_f32() - ref).abs().max()) < 1e-4
    with pytest.raises(RuntimeError):
        ops.bcast_add(xm, buf, out)


@pytest.mark.parametrize("chan", [(64, 128), (128, 256), (256, 512)])
def test_conv2d_grouped_stride2_entry_equals_separate_launches(dev, chan):
    """agp_conv2d_fwd_grouped, second kind of group: the stride-2 entry of a ResNet stage -- the 3x3/s2 conv and the 1x1/s2
    downsample of up to two trunks (different images and map sizes) as ONE launch of the generic kernel; bit-identical to
    separate launches and correct against fp64.  A 1x1 stride-1 pair (the bottleneck's entry) groups the same way."""
    from agplace_amd import ops
    cin, cout = chan
    g = torch.Generator().manual_seed(cin + cout)
    for trunks, k1, s in (([(3, 20, 36), (2, 9, 14)], 3, 2), ([(2, 11, 30)], 3, 2), ([(2, 10, 12), (1, 7, 9)], 1, 1)):
        jobs, sep, refs = [], [], []
        for (n, h, w) in trunks:
            x = torch.randn(n, cin, h, w, generator=g)
            xm = ops.pack_f32(x.to(dev), cin, 1, 4)
            for (k, relu) in ((k1, True), (1, False)):
                wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
                scale, shift = 0.5 + torch.rand(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
                ref = F.conv2d(x.double(), wt.double(), None, s, k // 2) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
                refs.append(torch.relu(ref) if relu else ref)
                cw = ops.ConvWeights(wt.to(dev), scale.to(dev), shift.to(dev), s, k // 2)
                ho, wo = ops.conv_out_size(h, k, s, k // 2), ops.conv_out_size(w, k, s, k // 2)
                jobs.append((xm, cw, ops.SplitMap.alloc(n, ho, wo, cout, 1, 4, dev), None, relu))
                o = ops.SplitMap.alloc(n, ho, wo, cout, 1, 4, dev)
                ops.conv2d(xm, cw, o, relu=relu, prec=4)
                sep.append(o)
        outs = ops.conv2d_grouped(jobs, 4)
        torch.cuda.synchronize()
        # (one trunk's [3x3/s2, 1x1/s2] pair is the stage-entry kernel's own pattern, igemm_s2.hip: another summation order than
        # the generic kernel's separate launches -- checked against fp64 here and in test_stage_entry_kernel_*)
        fused_entry = len(trunks) == 1 and s == 2
        for o, s_, r in zip(outs, sep, refs):
            assert fused_entry or torch.equal(o.hi, s_.hi)
            assert rel_l2(o.to_f32(), r) < 6e-4
            assert float(o.hi[:, 0].abs().max()) == 0 and float(o.hi[:, :, -1].abs().max()) == 0


def test_vecprog_ops_against_fp64(dev):
    """agp_vecprog_run op by op (LOAD/STORE, LINEAR from a register and from memory with K = 64/128/256, FCODE for every
    solver, L2NORM, LAYERNORM with residual, WSUM with device weights) on 37 rows against fp64 torch."""
    from agplace_amd import ops, vecprog
    from agplace_amd.network_mm.ffns import FCODE
    from agplace_amd.options import Options
    g = torch.Generator().manual_seed(2)
    b = 37
    x = torch.randn(b, 256, generator=g)
    x64 = torch.randn(b, 64, generator=g)
    y = torch.randn(b, 256, generator=g)
    lin = torch.nn.Linear(256, 256)
    lin64 = torch.nn.Linear(64, 256)
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.3)
    w1, w2 = torch.tensor([0.3]), torch.tensor([-1.7])
    for method, size in (("euler", 0.1), ("midpoint", 0.3), ("rk4", 0.25)):
        fc = FCODE(256, "tanh", opt=Options(odeint_method=method, odeint_size=size)).to(dev)
        vp = vecprog.VecProgram(b, dev)
        vp.load(0, x.to(dev))
        vp.load(1, y.to(dev), scale=w1.to(dev))
        vp.linear(2, ops.LinearWeights(lin.weight.to(dev), lin.bias.to(dev)), 0, add1=1, act="relu")
        o_lin = vp.store(2)
        vp.linear(3, ops.LinearWeights(lin64.weight.to(dev), lin64.bias.to(dev)), x64.to(dev))
        o_lin64 = vp.store(3)
        vp.fcode(4, fc, 0, 1)
        o_fc = vp.store(4)
        vp.l2norm(5, 2)
        o_l2 = vp.store(5)
        vp.layernorm(5, ln.to(dev), 3, relu=True, residual=0)
        o_ln = vp.store(5)
        vp.wsum(5, [0, 1, 3], [w1.to(dev), None, w2.to(dev)])
        o_ws = vp.store(5)
        vp.run()
        X, Y = x.double(), y.double() * 0.3
        W, B = lin.weight.double(), lin.bias.double()
        r_lin = torch.relu((X + Y) @ W.t() + B)
        r_lin64 = x64.double() @ lin64.weight.double().t() + lin64.bias.double()
        fw, fb = fc.func.func.fc.weight.detach().cpu().double(), fc.func.func.fc.bias.detach().cpu().double()
        r_fc = ode.odeint_fixed(lambda z: torch.tanh(z @ fw.t() + fb), X + Y, method, size)
        r_l2 = F.normalize(r_lin, dim=-1)
        r_ln = torch.relu(F.layer_norm(r_lin64, (256,), ln.weight.detach().cpu().double(), ln.bias.detach().cpu().double(), ln.eps) + X)
        r_ws = 0.3 * X + Y - 1.7 * r_lin64
        for name, got, ref in (("linear", o_lin, r_lin), ("linear64", o_lin64, r_lin64), ("fcode", o_fc, r_fc), ("l2", o_l2, r_l2),
                               ("ln", o_ln, r_ln), ("wsum", o_ws, r_ws)):
            assert rel_l2(got, ref) < 1e-5, (method, name, rel_l2(got, ref))      # split-bf16 x3 products: ~4e-6
    with pytest.raises(vecprog.VecProgramUnfit):
        vecprog.VecProgram(b, dev).linear(0, ops.LinearWeights(torch.randn(128, 256).to(dev), None), 0)


@pytest.mark.parametrize("n,h,w", [(3, 64, 96), (2, 224, 448), (1, 50, 70), (1, 224, 1344), (2, 37, 46)])
def test_stem_reading_the_raw_input_equals_pack_then_stem(dev, n, h, w):
    """agp_stem_pool_raw_fwd (the stem converts the fp32 image / the uint8 camera tiles on their way into LDS) is bit-identical
    to packing the input to an NHWC4 map first (agp_pack_f32_to_nhwc / agp_pack_u8_cams_to_nhwc + agp_stem_pool_fwd),
    for contiguous and strided fp32 images and for 1- and 2-camera uint8 tiles."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(n + h)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    scale, shift = 0.5 + torch.rand(64, generator=g), 0.3 * torch.randn(64, generator=g)
    scale[1::4] *= -1
    cw = ops.ConvWeights(wt.to(dev), scale.to(dev), shift.to(dev), 2, 3, stem=True)
    h1, w1 = ops.conv_out_size(h, 7, 2, 3), ops.conv_out_size(w, 7, 2, 3)
    h2, w2 = ops.conv_out_size(h1, 3, 2, 1), ops.conv_out_size(w1, 3, 2, 1)
    big = torch.randn(n, 3, h + 2, w + 5, generator=g).to(dev)
    for x in (big[:, :, 1:-1, 2:-3].contiguous(), big[:, :, 1:-1, 2:-3], big[:, :, 1:-1, 2:-3].contiguous(memory_format=torch.channels_last)):
        assert x.shape == (n, 3, h, w)
        ref = ops.SplitMap.alloc(n, h2, w2, 64, 1, 4, dev)
        ops.stem_pool(ops.pack_f32(x, 4, 3, 4), cw, ref, prec=4)
        got = ops.SplitMap.alloc(n, h2, w2, 64, 1, 4, dev)
        ops.stem_pool_raw(x, cw, got)
        assert torch.equal(got.hi, ref.hi)
    # (tile widths that are multiples of 32 take the walking kernel's uint8 path -- 3 cameras of 32 columns: every step's fringes
    # come from the neighbouring tiles --, the others the per-block kernel)
    for ncam in (1, 2, 3, 6):
        if w % ncam:
            continue
        u8 = torch.randint(0, 256, (n, ncam, h, w // ncam, 3), dtype=torch.uint8, generator=g).to(dev)
        ref = ops.SplitMap.alloc(n, h2, w2, 64, 1, 4, dev)
        ops.stem_pool(ops.pack_cameras_u8(u8, 4), cw, ref, prec=4)
        got = ops.SplitMap.alloc(n, h2, w2, 64, 1, 4, dev)
        ops.stem_pool_raw(u8, cw, got)
        assert torch.equal(got.hi, ref.hi)


@pytest.mark.parametrize("n,h,w,c,p", [(3, 14, 20, 128, 3.0), (5, 8, 6, 64, 2.5), (2, 28, 40, 64, 3.0), (7, 5, 11, 256, 3.0),
                                       (9, 14, 14, 256, 3.0)])
def test_conv_epilogue_pooling_matches_pool_pass(dev, n, h, w, c, p):
    """ops.PoolReq: the 3x3 kernel of the fp16 path pools the map it stores (agp_conv_desc::pool_partial + agp_pool_from_conv).
    Same values as a pooling pass over the stored map (different, fixed summation order), the map itself bit-identical to a conv
    without the request, bit-reproducible, and independent of the image's position in the batch."""
    from agplace_amd import ops
    torch.manual_seed(n * h + w)
    x = torch.randn(n, c, h, w, device=dev)
    wt = torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5
    sc, sh = 0.5 + torch.rand(c, device=dev), 0.1 * torch.randn(c, device=dev)
    pt = torch.tensor([p], device=dev)
    cw = ops.ConvWeights(wt, sc, sh, 1, 1)
    xm = ops.pack_f32(x, c, 1, 4)
    res = ops.pack_f32(torch.randn(n, c, h, w, device=dev), c, 1, 4)
    o0 = ops.SplitMap.alloc(n, h, w, c, 1, 4, dev)
    ops.conv2d(xm, cw, o0, residual=res, relu=True, prec=4)
    outs = []
    for _ in range(2):
        req = ops.PoolReq(pt, want_mean=True, want_gem=True)
        o1 = ops.SplitMap.alloc(n, h, w, c, 1, 4, dev)
        ops.conv2d(xm, cw, o1, residual=res, relu=True, prec=4, pool=req)
        assert req.fused
        assert torch.equal(o1.hi, o0.hi)
        outs.append((req.mean.clone(), req.gem.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # an image's pooled values do not depend on where it sits in the batch (the summation order is image-relative)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).to(dev)
    reqp = ops.PoolReq(pt, want_mean=True, want_gem=True)
    ops.conv2d(ops.pack_f32(x[perm], c, 1, 4), cw, ops.SplitMap.alloc(n, h, w, c, 1, 4, dev),
               residual=ops.pack_f32(res.to_f32()[perm], c, 1, 4), relu=True, prec=4, pool=reqp)
    assert torch.equal(reqp.mean, outs[0][0][perm]) and torch.equal(reqp.gem, outs[0][1][perm])
    mean_ref, gem_ref = ops.pool_map(o0, pt)
    dense = o0.to_f32().double()
    assert rel_l2(outs[0][0], dense.mean((2, 3))) < 1e-6 and rel_l2(outs[0][1], dense.clamp(min=1e-6).pow(p).mean((2, 3)).pow(1 / p)) < 1e-6
    assert rel_max(outs[0][0], mean_ref) < 1e-5 and rel_max(outs[0][1], gem_ref) < 1e-5
    # mean only, and a grouped launch where one problem pools and the other does not
    req_m = ops.PoolReq(want_mean=True, want_gem=False)
    x2 = ops.pack_f32(torch.randn(2, c, 9, 9, device=dev), c, 1, 4)
    oa, ob = ops.SplitMap.alloc(n, h, w, c, 1, 4, dev), ops.SplitMap.alloc(2, 9, 9, c, 1, 4, dev)
    ops.conv2d_grouped([(xm, cw, oa, res, True, req_m), (x2, cw, ob, None, True)], 4)
    assert torch.equal(oa.hi, o0.hi) and req_m.gem is None
    assert torch.equal(req_m.mean, outs[0][0])
    ob_ref = ops.SplitMap.alloc(2, 9, 9, c, 1, 4, dev)
    ops.conv2d(x2, cw, ob_ref, relu=True, prec=4)
    assert torch.equal(ob.hi, ob_ref.hi)


@pytest.mark.parametrize("n,h,w,cin,c", [(3, 14, 20, 128, 128), (5, 8, 6, 64, 64), (2, 28, 40, 64, 64), (7, 5, 11, 256, 256), (9, 14, 14, 128, 256),
                                         (12, 28, 28, 128, 128), (4, 56, 56, 64, 64)])
def test_conv_epilogue_sum_of_squares_gives_the_batchnorm_statistics(dev, n, h, w, cin, c):
    """ops.SqStatReq (agp_conv_desc::pool_stat = 1): the fp16 3x3 kernels reduce the map they store to per-channel sums and sums
    of squares; finished by agp_bn_stats_from_partial they are the train-mode BatchNorm statistics of that map -- equal to the
    statistics pass over z (train_graph.bn_stats) and to fp64 on the stored values; the map is bit-identical to a plain conv."""
    from agplace_amd import ops, train_graph
    torch.manual_seed(n * h + w + c)
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(c, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    cw = ops.ConvWeights(wt, None, 0.3 * torch.randn(c, device=dev), 1, 1)
    xm = ops.pack_f32(x, cin, 1, 4)
    z0 = ops.SplitMap.alloc(n, h, w, c, 1, 4, dev)
    ops.conv2d(xm, cw, z0, relu=False, prec=4)
    req = ops.SqStatReq()
    z1 = ops.SplitMap.alloc(n, h, w, c, 1, 4, dev)
    ops.conv2d(xm, cw, z1, relu=False, prec=4, pool=req)
    assert req.fused and torch.equal(z1.hi, z0.hi)
    part = req.partial.view(-1, 2, c)[:req.blocks].double()
    dense = z0.to_f32().double()
    assert rel_l2(part[:, 0].sum(0), dense.sum((0, 2, 3))) < 1e-6
    assert rel_l2(part[:, 1].sum(0), (dense * dense).sum((0, 2, 3))) < 1e-6
    bn_a, bn_b = torch.nn.BatchNorm2d(c).to(dev).train(), torch.nn.BatchNorm2d(c).to(dev).train()
    a = train_graph.bn_stats_from_partial(req.partial, req.blocks, z1, bn_a)
    b = train_graph.bn_stats(z0, bn_b)
    for u, v in zip(a, b):
        assert rel_l2(u, v) < 1e-5
    assert rel_l2(bn_a.running_var, bn_b.running_var) < 1e-6 and rel_l2(a[0], dense.mean((0, 2, 3))) < 1e-6
    assert rel_l2(a[1], (dense.var((0, 2, 3), unbiased=False) + bn_a.eps).rsqrt()) < 1e-5
    # a split-bf16 conv has no such epilogue (the request stays unfused: its caller runs the statistics pass), and an unknown
    # statistic selector is refused
    r3 = ops.SqStatReq()
    ops.conv2d(ops.pack_f32(x, cin, 1, 3), cw, ops.SplitMap.alloc(n, h, w, c, 1, 3, dev), relu=False, prec=3, pool=r3)
    assert not r3.fused
    from agplace_amd import _lib
    import ctypes
    d = ops._fill_conv_desc(_lib.ConvDesc(), xm, cw, z1, None, False, 4)
    d.pool_partial, d.pool_stat = req.partial.data_ptr(), 1
    assert _lib.load().agp_conv2d_fwd(ctypes.byref(d), _lib.stream()) == 0          # (the full descriptor is a valid call ...)
    d.pool_stat = 2
    assert _lib.load().agp_conv2d_fwd(ctypes.byref(d), _lib.stream()) != 0          # (... and only the selector makes it invalid)


@pytest.mark.parametrize("dim", [128, 512, 96])
@pytest.mark.parametrize("method,step,act", [("euler", 0.1, "relu"), ("midpoint", 0.3, "tanh"), ("rk4", 0.25, "sigmoid"), ("euler", 0.25, "id")])
def test_fcode_any_width_matches_oracle_forward_and_backward(dev, dim, method, step, act):
    """FCODE(dim) for dim != 256 (reference network_mm/ffns.py:78-87 takes any width; `--mm_stg2fuse_dim`, tools/options.py:113):
    forward and every gradient against fp64 autograd through the oracle's fixed-grid solver."""
    from agplace_amd.network_mm.ffns import FCODE
    from agplace_amd.options import Options
    torch.manual_seed(dim + len(method))
    m = FCODE(dim, act, opt=Options(odeint_method=method, odeint_size=step)).to(dev)
    fc = m.func.func.fc
    with torch.no_grad():
        fc.weight.mul_(0.5)
    for b in (5, 33):
        x = torch.randn(b, dim, device=dev, requires_grad=True)
        a1 = (0.5 * torch.randn(b, dim, device=dev)).requires_grad_(True)
        y = m(x, add1=a1)
        xr, ar = x.detach().double().cpu().requires_grad_(True), a1.detach().double().cpu().requires_grad_(True)
        wr, br = fc.weight.detach().double().cpu().requires_grad_(True), fc.bias.detach().double().cpu().requires_grad_(True)
        ref = ode.fcode(xr + ar, wr, br, act, method, step)
        assert y.shape == (b, dim) and rel_l2(y, ref) < 1e-4
        gy = torch.randn(b, dim, generator=torch.Generator().manual_seed(b))
        for p_ in (fc.weight, fc.bias):
            p_.grad = None
        y.backward(gy.to(dev))
        ref.backward(gy.double())
        assert rel_l2(x.grad, xr.grad) < 2e-4 and rel_l2(a1.grad, ar.grad) < 2e-4
        assert rel_l2(fc.weight.grad, wr.grad) < 2e-4 and rel_l2(fc.bias.grad, br.grad) < 2e-4


@pytest.mark.parametrize("chan", [(64, 128), (128, 256), (256, 512)])
def test_stage_entry_kernel_conv3x3s2_plus_downsample(dev, chan):
    """igemm_s2.hip: the stride-2 entry of a ResNet stage -- [3x3/s2 conv of every trunk ..., 1x1/s2 downsample of every trunk ...]
    (the order resnet.forward_maps_multi issues) as ONE launch in which the downsample rides on the 3x3's staged centre tap.
    Against fp64 for one and two trunks, odd and even map sizes, a batch that spans several row tiles; a trunk's outputs do not
    depend on what else is in the launch (bit-identical alone and grouped); halo rows / columns stay zero."""
    from agplace_amd import ops
    cin, cout = chan
    g = torch.Generator().manual_seed(7 * cin + cout)
    trunks = [(3, 20, 36), (2, 9, 15), (5, 28, 44), (1, 2, 2)]
    made = []
    for (n, h, w) in trunks:
        x = torch.randn(n, cin, h, w, generator=g)
        xm = ops.pack_f32(x.to(dev), cin, 1, 4)
        w3 = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        w1 = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
        s3, t3 = 0.5 + torch.rand(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
        s1, t1 = 0.5 + torch.rand(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
        r3 = torch.relu(F.conv2d(x.double(), w3.double(), None, 2, 1) * s3.double().view(1, -1, 1, 1) + t3.double().view(1, -1, 1, 1))
        r1 = F.conv2d(x.double(), w1.double(), None, 2, 0) * s1.double().view(1, -1, 1, 1) + t1.double().view(1, -1, 1, 1)
        c3 = ops.ConvWeights(w3.to(dev), s3.to(dev), t3.to(dev), 2, 1)
        c1 = ops.ConvWeights(w1.to(dev), s1.to(dev), t1.to(dev), 2, 0)
        ho, wo = ops.conv_out_size(h, 3, 2, 1), ops.conv_out_size(w, 3, 2, 1)
        assert (ho, wo) == (ops.conv_out_size(h, 1, 2, 0), ops.conv_out_size(w, 1, 2, 0))
        made.append((xm, c3, c1, n, ho, wo, r3, r1))

    def run(sel):
        convs = [(made[i][0], made[i][1], ops.SplitMap.alloc(made[i][3], made[i][4], made[i][5], cout, 1, 4, dev), None, True) for i in sel]
        dss = [(made[i][0], made[i][2], ops.SplitMap.alloc(made[i][3], made[i][4], made[i][5], cout, 1, 4, dev), None, False) for i in sel]
        outs = ops.conv2d_grouped(convs + dss, 4)
        torch.cuda.synchronize()
        return outs[:len(sel)], outs[len(sel):]
    alone = {}
    for i in range(len(trunks)):
        (o3,), (o1,) = run([i])
        alone[i] = (o3.hi.clone(), o1.hi.clone())
        assert rel_l2(o3.to_f32(), made[i][6]) < 6e-4 and rel_l2(o1.to_f32(), made[i][7]) < 6e-4, i
        for o in (o3, o1):
            assert float(o.hi[:, 0].abs().max()) == 0 and float(o.hi[:, -1].abs().max()) == 0
            assert float(o.hi[:, :, 0].abs().max()) == 0 and float(o.hi[:, :, -1].abs().max()) == 0
    for pair in ((0, 1), (2, 3), (1, 2)):
        o3s, o1s = run(list(pair))
        for i, o3, o1 in zip(pair, o3s, o1s):
            assert torch.equal(o3.hi, alone[i][0]) and torch.equal(o1.hi, alone[i][1]), (pair, i)


@pytest.mark.parametrize("case", [(64, 64, 1, 40, 70), (128, 128, 1, 28, 60), (256, 256, 1, 14, 30), (64, 128, 2, 40, 70), (128, 256, 2, 27, 45)])
def test_chunk_major_weight_plane_is_bitwise_neutral(dev, case, monkeypatch):
    """agp_conv_desc::w_cm: the fp16 weights in [K/32][cout][32] order for the kernels that stage them chunk-wise (igemm_kxr2,
    igemm_kxrw, igemm_s2 + its 1x1 downsample).  The same convs with the plane withheld (ops.USE_W_CM = False) read w_hi:
    outputs must be bit-identical, and the plane is a pure permutation of w_hi."""
    from agplace_amd import _lib, ops
    cin, cout, stride, h, w = case
    g = torch.Generator().manual_seed(cin + 3 * cout + stride)
    x = torch.randn(3, cin, h, w, generator=g)
    xm = ops.pack_f32(x.to(dev), cin, 1, 4)
    w3 = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    c3 = ops.ConvWeights(w3.to(dev), (0.5 + torch.rand(cout, generator=g)).to(dev), torch.randn(cout, generator=g).to(dev), stride, 1)
    hi, cm = c3.planes(_lib.PREC_F16)[0], c3.cm()
    assert cm is not None and cm.shape == (9 * cin // 32, cout, 32)
    assert torch.equal(cm.permute(1, 0, 2).reshape(cout, -1), hi.reshape(cout, -1))
    ho, wo = ops.conv_out_size(h, 3, stride, 1), ops.conv_out_size(w, 3, stride, 1)
    jobs = [(xm, c3, None, None, True)]
    if stride == 2:
        w1 = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
        c1 = ops.ConvWeights(w1.to(dev), None, None, 2, 0)
        assert c1.cm() is not None
        jobs.append((xm, c1, None, None, False))

    def run():
        js = [(j[0], j[1], ops.SplitMap.alloc(3, ho, wo, cout, 1, 4, dev), None, j[4]) for j in jobs]
        outs = ops.conv2d_grouped(js, 4) if len(js) > 1 else [ops.conv2d(js[0][0], js[0][1], js[0][2], relu=True, prec=4)]
        torch.cuda.synchronize()
        return [o.hi.clone() for o in outs]
    with_cm = run()
    monkeypatch.setattr(ops, "USE_W_CM", False)
    without = run()
    for a, b in zip(with_cm, without):
        assert torch.equal(a, b)
    ref = torch.relu(F.conv2d(x.double(), w3.double(), None, stride, 1) * c3.scale.cpu().double().view(1, -1, 1, 1)
                     + c3.shift.cpu().double().view(1, -1, 1, 1))
    got = ops.SplitMap.alloc(3, ho, wo, cout, 1, 4, dev)
    got.hi.copy_(with_cm[0])
    assert rel_l2(got.to_f32().cpu(), ref) < 6e-4


@pytest.mark.parametrize("prec", [2, 3])
@pytest.mark.parametrize("case", [(64, 64, 40, 70), (128, 256, 14, 30), (96, 64, 9, 11)])
def test_chunk_major_planes_of_the_two_plane_modes_are_bitwise_neutral(dev, case, prec, monkeypatch):
    """agp_conv_desc::w_cm + w_cm_lo for F16W2 / BF16X3 3x3 stride-1 convs (igemm_kxr): the same bits as with w_hi / w_lo, for
    both output planes; the planes are permutations of the row-major ones."""
    from agplace_amd import ops
    cin, cout, h, w = case
    g = torch.Generator().manual_seed(cin + cout + prec)
    x = torch.randn(3, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    cw = ops.ConvWeights(wt.to(dev), (0.5 + torch.rand(cout, generator=g)).to(dev), torch.randn(cout, generator=g).to(dev), 1, 1)
    c2 = cw.cm2(prec)
    assert c2 is not None
    for cmp_, pl in zip(c2, cw.planes(prec)):
        assert torch.equal(cmp_.permute(1, 0, 2).reshape(cout, -1), pl.reshape(cout, -1))
    xm = ops.pack_f32(x.to(dev), cin, 1, prec)

    def run():
        o = ops.conv2d(xm, cw, ops.SplitMap.alloc(3, h, w, cout, 1, prec, dev), relu=True, prec=prec)
        torch.cuda.synchronize()
        return o
    a = run()
    monkeypatch.setattr(ops, "USE_W_CM", False)
    b = run()
    assert torch.equal(a.hi, b.hi) and (a.lo is None or torch.equal(a.lo, b.lo))
    ref = torch.relu(F.conv2d(x.double(), wt.double(), None, 1, 1) * cw.scale.cpu().double().view(1, -1, 1, 1)
                     + cw.shift.cpu().double().view(1, -1, 1, 1))
    assert rel_l2(a.to_f32().cpu(), ref) < (3e-4 if prec == 2 else 3e-5)


def _bblock_problem(dev, g, n, h, w):
    from agplace_amd import ops
    x = torch.relu(torch.randn(n, 64, h, w, generator=g))
    ws = [torch.randn(64, 64, 3, 3, generator=g) / (64 * 9) ** 0.5 for _ in range(2)]
    sc = [0.5 + torch.rand(64, generator=g) for _ in range(2)]
    sh = [0.3 * torch.randn(64, generator=g) for _ in range(2)]
    xm = ops.pack_f32(x.to(dev), 64, 1, 4)
    cws = [ops.ConvWeights(ws[i].to(dev), sc[i].to(dev), sh[i].to(dev), 1, 1) for i in range(2)]
    x64 = xm.to_f32().double().cpu()                      # the fp16-rounded input is what both paths see
    t = F.conv2d(x64, ws[0].double(), None, 1, 1) * sc[0].double().view(1, -1, 1, 1) + sh[0].double().view(1, -1, 1, 1)
    ref = F.conv2d(torch.relu(t), ws[1].double(), None, 1, 1) * sc[1].double().view(1, -1, 1, 1) + sh[1].double().view(1, -1, 1, 1)
    return xm, cws, torch.relu(ref + x64)


def _bblock_unfused(dev, xm, cws, pool=None):
    from agplace_amd import ops
    mid = ops.SplitMap.alloc(xm.n, xm.h, xm.w, 64, 1, 4, dev)
    out = ops.SplitMap.alloc(xm.n, xm.h, xm.w, 64, 1, 4, dev)
    ops.conv2d(xm, cws[0], mid, relu=True, prec=4)
    ops.conv2d(mid, cws[1], out, residual=xm, relu=True, prec=4, pool=pool)
    return out


@pytest.mark.parametrize("shapes", [[(2, 8, 30)], [(3, 56, 56)], [(1, 14, 100)], [(5, 6, 28), (2, 20, 57)], [(4, 56, 336), (4, 56, 56)],
                                    [(1, 2, 3), (2, 4, 29), (1, 10, 10), (3, 12, 84)]])
def test_fused_basicblock64_is_bit_identical_to_two_convs(dev, shapes):
    """csrc/fblock64.hip (agp_bblock64_fwd_grouped): a ResNet layer-1 BasicBlock as one kernel, the intermediate map in LDS --
    bit-identical to conv2d + conv2d at AGP_PREC_F16 (same MFMA sequence, same fp16 rounding of the intermediate), the halo of
    the output untouched, and inside the fp16 bound of the fp64 block (torchvision BasicBlock, network_mm/image_fe.py:102)."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(len(shapes) * 100 + shapes[0][2])
    jobs, want, refs = [], [], []
    for (n, h, w) in shapes:
        xm, cws, ref = _bblock_problem(dev, g, n, h, w)
        assert ops.bblock64_ok(xm, cws[0], cws[1], 4)
        want.append(_bblock_unfused(dev, xm, cws))
        out = ops.SplitMap.alloc(n, h, w, 64, 1, 4, dev)
        out.hi[:, 1:-1, 1:-1].fill_(7.0)                   # every interior element must be overwritten
        jobs.append((xm, cws[0], cws[1], out))
        refs.append(ref)
    outs = ops.bblock64_grouped(jobs, exact=True)          # the 32x32x16 form: the default conv kernels' MFMA sequence
    torch.cuda.synchronize()
    for o, s, r in zip(outs, want, refs):
        assert torch.equal(o.hi, s.hi)
        assert rel_l2(o.to_f32(), r) < 8e-4
    # the production form (16x16x32 tiles): the same block up to the fp32 rounding of another accumulation order
    jobs16 = [(xm, c1, c2, ops.SplitMap.alloc(xm.n, xm.h, xm.w, 64, 1, 4, dev)) for (xm, c1, c2, _) in jobs]
    outs16 = ops.bblock64_grouped(jobs16)
    torch.cuda.synchronize()
    for o, s, r in zip(outs16, want, refs):
        assert rel_l2(o.to_f32(), r) < 8e-4
        assert rel_l2(o.to_f32(), s.to_f32()) < 3e-4
        # fp16 values one rounding apart at most, except where an accumulation-order difference crossed a rounding boundary twice
        assert float((o.hi.float() - s.hi.float()).abs().max()) <= 4 * float(s.hi.float().abs().max()) * 2 ** -10
        assert float(o.hi[:, 0].abs().max()) == 0 and float(o.hi[:, -1].abs().max()) == 0
        assert float(o.hi[:, :, 0].abs().max()) == 0 and float(o.hi[:, :, -1].abs().max()) == 0


def test_fused_basicblock64_pooling_is_position_independent(dev):
    """The level mean reduced in the fused block's epilogue (fuse_block_toshallow.py:82): equal to pooling the stored map,
    bit-identical when the images are permuted or the batch is cut (image-relative summation units)."""
    from agplace_amd import ops
    g = torch.Generator().manual_seed(11)
    n, h, w = 6, 56, 84
    xm, cws, _ = _bblock_problem(dev, g, n, h, w)

    def run(xmap):
        req = ops.PoolReq(want_mean=True, want_gem=False)
        out = ops.SplitMap.alloc(xmap.n, h, w, 64, 1, 4, dev)
        ops.bblock64_grouped([(xmap, cws[0], cws[1], out, req)])
        return out, req.mean
    out, mean = run(xm)
    ref_mean, _ = ops.pool_map(out, None, want_mean=True, want_gem=False)
    assert rel_l2(mean, ref_mean) < 1e-6
    assert rel_l2(mean, out.to_f32().double().mean((2, 3))) < 1e-6
    perm = torch.tensor([4, 0, 5, 2, 1, 3], device=dev)
    xp = ops.SplitMap(xm.hi[perm].contiguous(), None, n, h, w, 64, 1)
    out_p, mean_p = run(xp)
    assert torch.equal(out_p.hi, out.hi[perm]) and torch.equal(mean_p, mean[perm])
    xs = ops.SplitMap(xm.hi[2:5].contiguous(), None, 3, h, w, 64, 1)
    out_s, mean_s = run(xs)
    assert torch.equal(out_s.hi, out.hi[2:5]) and torch.equal(mean_s, mean[2:5])


