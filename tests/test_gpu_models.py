"""-m gpu parity tests, model level: ImageFE / MM.forward_q / DBVanilla2D through the product
modules (C ABI underneath) against the CPU oracle on identical parameters and inputs.

Bar (BASELINE.json north_star): outputs within 1e-3 relative of the fp32 reference forward."""
import pytest
import torch

from oracle import nets, resnet
from gpu_util import frac_within, rel_l2, rel_max, elem_rel, randomize_bn, cpu_state, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_mode(request):
    """Like the reference's test.py / mining code, inference runs under torch.no_grad(); only the
    gradient test enables autograd (the models refuse grad mode unless freeze_backbone() was called)."""
    if "gradients" in request.node.name:
        yield
    else:
        with torch.no_grad():
            yield


TOL = 1e-3          # north_star tolerance, relative to the fp32/fp64 reference forward


# map-level bounds per MFMA precision mode: (rel_l2, rel_max) against the fp64 oracle.  Mode 3 (split-bf16) is near fp32; modes 2
# and 4 store fp16 maps (2^-11 per element) and mode 4 also rounds the weights to fp16.  The 1e-3 bar of BASELINE.json holds for
# every map the op-level drop-in ImageFE.forward EXPORTS: its default (prec=None) is the tight mode 2 when the process default is
# the one-product mode 4 (network_mm/image_fe.py: export_precision) -- ResNet18 / 34 / 50 below, values printed.
# Mode 4 is the INTERNAL mode of MM / DBVanilla2D, whose contract is on the network outputs (descriptors <= 5e-4, tests below):
# its maps meet 1e-3 on ResNet18 (<= 8.5e-4 measured) and are held to F16_DEEP_INTERNAL on the deeper trunks (ResNet34 layer 3,
# 14 residual blocks deep: 1.01e-3 / 1.6e-3 measured in round 3) as a regression guard, not as an exported tolerance.
FE_BOUNDS = {3: (1e-4, 1e-3), 2: (1e-3, 4e-3), 4: (1e-3, 4e-3)}
F16_DEEP_INTERNAL = (1.5e-3, 4e-3)


def _fe_bound(prec, fe_type):
    return F16_DEEP_INTERNAL if (prec == 4 and fe_type != "resnet18") else FE_BOUNDS[prec]


@pytest.mark.parametrize("prec", [3, 2, 4, None])
@pytest.mark.parametrize("fe_type,layers,hw", [("resnet18", "2_2_2", (64, 96)), ("resnet34", "2_2_2", (64, 64)),
                                                ("resnet18", "2_2_2_2", (64, 64))])
def test_image_fe_query_side(dev, fe_type, layers, hw, prec):
    """prec None: ImageFE.forward's default = the export precision (the tight mode 2 under the library default 4): every exported
    map inside the 1e-3 bar.  prec 4: the internal mode of the fused models (the bench's kernels are reached)."""
    from agplace_amd.network_mm.image_fe import ImageFE
    torch.manual_seed(0)
    fe = randomize_bn(ImageFE(fe_type, layers)).to(dev).eval()
    x = torch.randn(2, 3, *hw)
    last, maps = fe(x.to(dev)) if prec is None else fe(x.to(dev), prec=prec)
    ref = resnet.forward_resnet(x.double(), {k: v.double() if v.is_floating_point() else v
                                             for k, v in cpu_state(fe.fe).items()}, fe_type, len(layers.split("_")))
    assert len(maps) == len(ref) and last.shape == ref[-1].shape
    if prec is None:
        assert fe.export_precision() == 2
    b2, bm = (TOL, 4e-3) if prec is None else _fe_bound(prec, fe_type)
    for m, r in zip(maps, ref):
        assert m.shape == r.shape
        print(f"FEMAP {fe_type} prec {prec} rel_l2 {rel_l2(m, r):.2e} rel_max {rel_max(m, r):.2e}")
        assert rel_l2(m, r) < b2 and rel_max(m, r) < bm
    if prec is None:
        assert fe.fe._sat_count == 0          # the first forward's fp16 saturation check ran and found nothing


@pytest.mark.parametrize("prec", [3, 2, 4, None])
def test_image_fe_resnet50_db_side(dev, prec):
    from agplace_amd.network.image_fe import ImageFE
    torch.manual_seed(1)
    fe = randomize_bn(ImageFE("resnet50", "3_4_6")).to(dev).eval()
    assert fe.last_dim == 1024
    x = torch.randn(2, 3, 64, 64)
    last, maps = fe(x.to(dev)) if prec is None else fe(x.to(dev), prec=prec)
    ref = resnet.forward_resnet(x.double(), {k: v.double() if v.is_floating_point() else v
                                             for k, v in cpu_state(fe.fe).items()}, "resnet50", 3)
    assert last.shape == (2, 1024, 4, 4)
    b2, bm = (TOL, 4e-3) if prec is None else _fe_bound(prec, "resnet50")
    for m, r in zip(maps, ref):
        print(f"FEMAP resnet50 prec {prec} rel_l2 {rel_l2(m, r):.2e} rel_max {rel_max(m, r):.2e}")
        assert rel_l2(m, r) < b2 and rel_max(m, r) < bm


def test_fp16_saturation_is_reported(dev):
    """fp16 maps saturate silently at 65504 on weights / inputs outside fp16's range; the first forward after a
    weight load must say so (and split-bf16 maps, mode 3, must not care)."""
    import warnings
    from agplace_amd.network_mm.image_fe import ImageFE
    torch.manual_seed(2)
    fe = ImageFE("resnet18", "2_2_2").to(dev).eval()
    x = (torch.randn(1, 3, 64, 64) * 3.0e4).to(dev)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        fe.forward_maps(x, prec=4)
    assert fe.fe._sat_count > 0 and any("fp16 limit" in str(w.message) for w in rec)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        fe.forward_maps(x, prec=4)                 # same weights: checked once
    assert not rec
    fe3 = ImageFE("resnet18", "2_2_2").to(dev).eval()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        maps = fe3.forward_maps(x, prec=3)
    assert not rec and torch.isfinite(maps[-1].to_f32()).all()


MM_VARIANTS = [
    dict(),
    dict(mfma_precision=4),
    dict(mfma_precision=3),
    dict(odeint_method="rk4", odeint_size=0.25),
    dict(odeint_method="midpoint", odeint_size=0.3, diff_type="fcode@tanh_fcode@relu", diff_direction="forward"),
    dict(final_fusetype="cat", final_l2=True, final_type=["imageorg", "shalloworg", "stg2image", "stg2fuse"],
         imagevoxorg_weight=0.5, stg2fuse_weight=0.25),
    dict(mm_imgfe="resnet34", diff_type="fcode@sigmoid"),
]


@pytest.mark.parametrize("variant", MM_VARIANTS)
def test_mm_forward_q_matches_oracle(dev, variant):
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(**variant)
    torch.manual_seed(3)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    data = nets.synth_query(3, 64, 192, opt, seed=5)
    out = model(to_dev(data, dev), mode="q")
    params = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu_state(model).items()}
    d64 = {k: ([t.double() for t in v] if isinstance(v, list) else v.double()) for k, v in data.items()}
    ref = nets.mm_forward_q(d64, params, opt)
    assert set(out.keys()) == set(ref.keys())
    for k in ref:
        assert out[k].shape == ref[k].shape, k
        assert rel_l2(out[k], ref[k]) < TOL and rel_max(out[k], ref[k]) < TOL, (k, rel_l2(out[k], ref[k]))
        # 99.9 % of the ELEMENTS, small ones included (on these 768-element vectors: the worst element).  The bound is on
        # r = |a-b| / (|b| + 1e-3 max|b|).  For an absolute error eps * sigma on elements ~ N(0, sigma^2) the 0.999-quantile
        # of r is eps / (1.25e-3 + 3e-3) = 235 eps, and the measured values follow that (round 2, this test's inputs):
        #   split-bf16 (3)        rel_l2 <= 7e-6    elem_rel <= 1.9e-3     bound 5e-3
        #   fp16 x fp16+e4m3 (2)  rel_l2 <= 2e-4    elem_rel <= 2.4e-2     bound 5e-2   (the library default)
        #   fp16 x fp16 (4)       rel_l2 <= 3.4e-4  elem_rel <= 7.9e-2     bound 0.15   (the bench's precision)
        etol = {3: 5 * TOL, 2: 5e-2, 4: 0.15}[opt.mfma_precision]
        assert elem_rel(out[k], ref[k]) < etol, (k, elem_rel(out[k], ref[k]))
        # ... and a bound that can FAIL: >= 99 % of a descriptor's elements within rtol * (|b| + 1e-2 max|b|).  An fp16
        # descriptor (eps = 3.4e-4 sigma per element) has ~99.9 % inside 2e-2; at twice that error ~98.8 % -- the test fails.
        rtol = {3: 1e-3, 2: 1e-2, 4: 2e-2}[opt.mfma_precision]
        assert frac_within(out[k], ref[k], rtol) >= 0.99, (k, frac_within(out[k], ref[k], rtol))


@pytest.mark.parametrize("prec,tol", [(3, 5e-5), (2, 2e-4), (4, 1e-3)])
def test_mm_precision_modes(dev, prec, tol):
    """Descriptor error per MFMA precision mode (include/agplace_hip.h): split-bf16 ~1e-5, the default
    F16W2 ~3e-5 (exact weights, fp16 activation noise averages out in the pools), F16 ~4e-4."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=prec)
    torch.manual_seed(3)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    data = nets.synth_query(2, 64, 128, opt, seed=6)
    out = model(to_dev(data, dev), mode="q")
    ref = nets.mm_forward_q(data, cpu_state(model), opt)
    err = rel_l2(out["embedding"], ref["embedding"])
    print(f"PRECMODE {prec} embedding rel_l2 {err:.2e}")
    assert err < tol


def test_mm_drop_image_and_error_paths(dev):
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options()
    model = MM(drop="image", opt=opt).to(dev).eval()
    data = nets.synth_query(1, 64, 64, opt, seed=7)
    out = model(to_dev(data, dev), mode="q")
    d0 = dict(data)
    d0["query_image"] = data["query_image"] * 0
    ref = nets.mm_forward_q(d0, cpu_state(model), opt)
    assert rel_l2(out["embedding"], ref["embedding"]) < TOL
    with pytest.raises(NotImplementedError):
        model(to_dev(data, dev), mode="db")


def test_train_mode_under_no_grad_matches_oracle(dev):
    """`.train()` under torch.no_grad() (reference train.py:307,315: `torch.set_grad_enabled(args.train_modelq)` around
    models in train mode, i.e. a frozen tower): batch-statistics BatchNorm, running statistics updated, plain tensors."""
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options()
    torch.manual_seed(31)
    model = randomize_bn(MM(opt=opt)).to(dev).train()
    data = nets.synth_query(4, 64, 128, opt, seed=32)
    params = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu_state(model).items()}
    rm0 = model.image_fe.fe.bn1.running_mean.clone()
    out = model(to_dev(data, dev), mode="q")
    assert not out["embedding"].requires_grad
    d64 = {k: ([t.double() for t in v] if isinstance(v, list) else v.double()) for k, v in data.items()}
    ref = nets.mm_forward_q(d64, params, opt, training=True)
    for k in ref:
        assert rel_l2(out[k], ref[k]) < TOL, (k, rel_l2(out[k], ref[k]))
    assert not torch.equal(model.image_fe.fe.bn1.running_mean, rm0)           # running statistics moved
    assert int(model.image_fe.fe.bn1.num_batches_tracked) == 1
    db = randomize_bn(DBVanilla2D("db", 256, opt=opt), seed=2).to(dev).train()
    x = torch.randn(3, 2, 1, 3, 64, 64, generator=torch.Generator().manual_seed(33))
    pd = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu_state(db).items()}
    e = db({"db_map": x.to(dev)}, mode="db")["embedding"]
    r = nets.dbvanilla2d_forward_db({"db_map": x.double()}, pd, opt, training=True)["embedding"]
    assert not e.requires_grad and rel_l2(e, r) < TOL


@pytest.mark.parametrize("shape", [(4, 1, 3, 224, 224), (2, 3, 1, 3, 64, 64), (2, 2, 3, 64, 96)])
def test_dbvanilla2d_matches_oracle(dev, shape):
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.options import Options
    nmap = shape[-4]
    opt = Options(maptype="_".join(["satellite", "roadmap", "terrain"][:nmap]))
    torch.manual_seed(4)
    model = randomize_bn(DBVanilla2D("db", 256, opt=opt)).to(dev).eval()
    x = torch.randn(*shape)
    out = model({"db_map": x.to(dev)}, mode="db")["embedding"]
    ref = nets.dbvanilla2d_forward_db({"db_map": x}, cpu_state(model), opt)["embedding"]
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < TOL and rel_max(out, ref) < TOL


def test_full_size_sample_independence(dev):
    """C3-sized input (6-cam panorama 224x1344): size-independent property -- every sample is
    embedded independently, so permuting the batch permutes the outputs bit-for-bit."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=4)        # the bench's precision mode
    torch.manual_seed(8)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    data = to_dev(nets.synth_query(5, 224, 1344, opt, seed=9), dev)
    e1 = model(data, mode="q")["embedding"].clone()
    perm = torch.tensor([3, 0, 4, 1, 2], device=dev)
    d2 = {k: ([t[perm] for t in v] if isinstance(v, list) else v[perm]) for k, v in data.items()}
    e2 = model(d2, mode="q")["embedding"]
    assert torch.isfinite(e1).all()
    assert torch.equal(e1[perm], e2)


@pytest.mark.parametrize("prec,tol", [(2, 3e-4), (3, 2e-5)])
def test_full_size_descriptors_against_oracle(dev, prec, tol):
    """The bench workload's shape (6-camera panorama 224x1344, default options): every output descriptor
    against the fp32 oracle.  In the default F16W2 mode the fp16 activation rounding averages out over the
    1176 pooled positions: measured 3e-5 on the image descriptor, 1.2e-4 on the embedding (stage-2 path),
    against the 1e-3 bar."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=prec)
    torch.manual_seed(18)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    data = nets.synth_query(2, 224, 1344, opt, seed=19)
    out = model(to_dev(data, dev), mode="q")
    ref = nets.mm_forward_q(data, cpu_state(model), opt)
    errs = {k: rel_l2(out[k], ref[k]) for k in ref}
    print("FULLSIZE", prec, " ".join(f"{k}:{v:.1e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v < tol, (k, v)


def _image_like_tiles(b, ncam, h, w, seed):
    """uint8 camera-like tiles [b, ncam, h, w, 3]: smooth shading + blocks of constant colour + sensor noise (the value statistics
    of a photograph: strong low frequencies, edges, saturated regions), not N(0, 1)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(b * ncam, 3, h // 16, w // 16, generator=g)
    img = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)
    for i in range(b * ncam):
        for _ in range(6):
            y0, x0 = int(torch.randint(0, h - 40, (1,), generator=g)), int(torch.randint(0, w - 40, (1,), generator=g))
            hh, ww = int(torch.randint(16, 40, (1,), generator=g)), int(torch.randint(16, 40, (1,), generator=g))
            img[i, :, y0:y0 + hh, x0:x0 + ww] = torch.rand(3, 1, 1, generator=g)
    img = (img + 0.03 * torch.randn(img.shape, generator=g)).clamp(0, 1)
    img[:, :, : h // 8] = img[:, :, : h // 8].clamp(min=0.92)               # an over-exposed sky
    return (img * 255).round().to(torch.uint8).view(b, ncam, 3, h, w).permute(0, 1, 3, 4, 2).contiguous()


def test_full_size_f16_with_checkpoint_like_statistics(dev):
    """the bench's precision (one fp16 product, fp16 maps) on a network with the statistics of a TRAINED
    checkpoint rather than seeded random BatchNorm buffers and N(0, 1) inputs: Kaiming fan-out convolutions, BatchNorm gamma = 1 /
    beta = 0 with running statistics CALIBRATED on image-like inputs (so every layer's activations are normalised, as after
    training), inputs = normalised uint8 camera tiles with saturated regions, full size (6 x 224 x 224 panorama).  No map element
    may sit at the fp16 limit and every descriptor must stay within 5e-4 of the fp32 oracle."""
    from agplace_amd import ops
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=4)
    torch.manual_seed(77)
    model = MM(opt=opt).to(dev)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    # calibration: train-mode forwards under no_grad (batch statistics; momentum 1 -> the running buffers become the statistics
    # of the last calibration batch), on the product's own split-bf16 training kernels (fp32-class)
    for m in bns:
        m.momentum = 1.0
    model.train()
    calib = nets.synth_query(4, 224, 1344, opt, seed=5)
    tiles_c = _image_like_tiles(4, 6, 224, 224, seed=6)
    dc = to_dev(calib, dev)
    dc["query_image"] = tiles_c.to(dev)
    model(dc, mode="q")
    model.eval()
    for m in bns:
        assert float(m.running_var.min()) > 0 and bool(torch.isfinite(m.running_mean).all())
    # evaluation batch (other images), the fp32 oracle on the normalised width-concatenated panorama
    data = nets.synth_query(2, 224, 1344, opt, seed=7)
    tiles = _image_like_tiles(2, 6, 224, 224, seed=8)
    mean = torch.tensor(ops.IMAGENET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(ops.IMAGENET_STD).view(1, 1, 3, 1, 1)
    img = (tiles.permute(0, 1, 4, 2, 3).float() / 255 - mean) / std
    data["query_image"] = torch.cat([img[:, c] for c in range(6)], dim=-1)
    d2 = to_dev(data, dev)
    d2["query_image"] = tiles.to(dev)
    out = model(d2, mode="q")
    assert model.image_fe.fe._sat_count == 0                                  # the first-forward saturation check found nothing
    maps = model.image_fe.forward_maps(tiles.to(dev), prec=4)
    assert sum(ops.count_saturated(m) for m in maps) == 0
    peak = max(float(m.hi.float().abs().max()) for m in maps)
    ref = nets.mm_forward_q(data, cpu_state(model), opt)
    errs = {k: rel_l2(out[k], ref[k]) for k in ref}
    print("CKPTLIKE peak map value", peak, " ".join(f"{k}:{v:.1e}" for k, v in errs.items()))
    assert peak < 6.0e4
    for k, v in errs.items():
        assert v < 5e-4, (k, v)


def test_mm_fusion_path_gradients_match_oracle(dev):
    """Gradients of a scalar loss on the embedding w.r.t. every fusion-path parameter (up-dims,
    Neural-ODE blocks, projections, Basic MLP) against autograd through the fp64 oracle, with the conv
    backbone frozen (MM.freeze_backbone: a constant feature extractor on both sides)."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(odeint_method="rk4", odeint_size=0.25, final_type=["imageorg", "shalloworg", "stg2fuse"],
                  stg2fuse_weight=0.5, mfma_precision=3)       # frozen trunk features on split-bf16 maps: the 1e-3 bar below is about
                                                               # the fusion path's backward, not about fp16 features (3e-4) through its kinks
    torch.manual_seed(11)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    model.freeze_backbone()      # (not frozen: .eval() + grads = end-to-end on frozen BN statistics, tests/test_gpu_train.py)
    data = nets.synth_query(4, 64, 128, opt, seed=12)
    G = torch.randn(4, 256)
    out = model(to_dev(data, dev), mode="q")
    (out["embedding"] * G.to(dev)).sum().backward()
    params = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu_state(model).items()}
    watch = [k for k in params if k.startswith(("fuseblocktoshallow.", "stg2fuseblock.projsimgfuse",
                                                "stg2fuseblock.ffnsfuse", "stg2fusefc."))]
    for k in watch:
        params[k].requires_grad_(True)
    d64 = {k: ([t.double() for t in v] if isinstance(v, list) else v.double()) for k, v in data.items()}
    # the oracle cuts the same path the product cuts (freeze_backbone): the stage-2 conv block sees a
    # detached fusion vector
    import oracle.nets as onets
    orig = onets.basic_block_conv
    onets.basic_block_conv = lambda x, p_, pre, training=False, pattern=None: orig(x.detach(), p_, pre, training, pattern)
    try:
        ref = nets.mm_forward_q(d64, params, opt)
    finally:
        onets.basic_block_conv = orig
    (ref["embedding"] * G.double()).sum().backward()
    got = dict(model.named_parameters())
    checked = 0
    for k in watch:
        if params[k].grad is None or float(params[k].grad.abs().max()) == 0:
            continue
        assert got[k].grad is not None, k
        gr = params[k].grad.reshape(got[k].grad.shape)
        assert rel_l2(got[k].grad, gr) < 1e-3, (k, rel_l2(got[k].grad, gr))
        checked += 1
    assert checked >= 18
    # note: projsfuseimg feeds the stage-2 conv block, whose backward is not built -> its gradient
    # only flows through ... nothing; it must therefore be absent, not silently wrong


def test_mm_accepts_uint8_camera_tiles(dev):
    """data_dict['query_image'] as uint8 [b,ncam,h,w,3]: the device-side input pipeline feeds the stem
    directly; same descriptors as the fp32 path on the normalised, width-concatenated image."""
    from agplace_amd import ops
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=4)        # the bench's precision mode
    torch.manual_seed(5)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    data = nets.synth_query(2, 64, 128, opt, seed=8)
    g = torch.Generator().manual_seed(4)
    tiles = torch.randint(0, 256, (2, 2, 64, 64, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor(ops.IMAGENET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(ops.IMAGENET_STD).view(1, 1, 3, 1, 1)
    img = (tiles.permute(0, 1, 4, 2, 3).float() / 255 - mean) / std
    data["query_image"] = torch.cat([img[:, 0], img[:, 1]], dim=-1)
    ref = nets.mm_forward_q(data, cpu_state(model), opt)
    d2 = to_dev(data, dev)
    d2["query_image"] = tiles.to(dev)
    out = model(d2, mode="q")
    assert rel_l2(out["embedding"], ref["embedding"]) < TOL
    assert rel_l2(out["imagevec_org"], ref["imagevec_org"]) < TOL


@pytest.mark.parametrize("prec,ntd", [(2, 0), (3, 0), (3, 1)])
def test_mm_end_to_end_with_sparse_voxel_branch(dev, prec, ntd):
    """MM.forward_q from query_image + coords/features exactly like the reference (mm.py:76-160): image
    branch, MinkFPN voxel branch, stage-1 and stage-2 fusion; all seven outputs against the oracle.
    ntd = 1: --mm_voxfe_ntd 1 (MinkFPN's top-down path; needs equal voxel planes, as in the reference, whose
    FuseBlockToShallow takes the level widths from --mm_voxfe_planes)."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    from oracle import sparse as osp
    opt = Options(mfma_precision=prec) if ntd == 0 else Options(mfma_precision=prec, mm_voxfe_ntd=ntd, mm_voxfe_planes="256_256_256")
    torch.manual_seed(9)
    model = MM(opt=opt)
    params = nets.init_mm_params(opt, seed=12)
    model.load_reference_state_dict(params)
    model = model.to(dev).eval()
    data = nets.synth_query(2, 64, 128, opt, seed=9)
    for k in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse"):
        data.pop(k)
    coords, feats = osp.synth_cloud(2, 200, extent=24, seed=6)
    coords[::5, 1:] += 0.3
    data["coords"], data["features"] = coords, feats
    out = model(to_dev(data, dev), mode="q")
    ref = nets.mm_forward_q(data, params, opt)
    for k in ref:
        assert rel_l2(out[k], ref[k]) < TOL, (k, rel_l2(out[k], ref[k]))
    # drop='pc' zeroes the voxel coordinates (mm.py:73-74)
    model.drop = 'pc'
    out2 = model(to_dev(data, dev), mode="q")
    d0 = dict(data)
    d0["coords"] = torch.cat([coords[:, :1], coords[:, 1:] * 0], 1)
    ref2 = nets.mm_forward_q(d0, params, opt)
    assert rel_l2(out2["embedding"], ref2["embedding"]) < TOL


def test_mm_sub_batches_on_streams_give_identical_outputs(dev):
    """Options.query_substreams = 2: MM.forward embeds the batch as two halves on two HIP streams
    (workspaces keyed by stream); every output is identical to the single-stream pass."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=4)        # the bench's precision mode
    torch.manual_seed(13)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    data = to_dev(nets.synth_query(6, 64, 128, opt, seed=14), dev)
    ref = model(data, mode="q")
    opt.query_substreams = 2
    out = model(data, mode="q")
    out2 = model(data, mode="q")                    # steady state: reuses the per-stream workspaces
    torch.cuda.synchronize()
    for k in ref:
        assert torch.equal(out[k], ref[k]) and torch.equal(out2[k], ref[k]), k
    opt.query_substreams = 4                        # 6 is not divisible by 4 -> single pass
    out3 = model(data, mode="q")
    assert torch.equal(out3["embedding"], ref["embedding"])


@pytest.mark.parametrize("prec,qsub", [(4, 1), (4, 2), (2, 1)])
def test_embed_pair_equals_separate_forwards(dev, prec, qsub):
    """agplace_amd.pair.embed_pair: the query and database trunks in lock-step (grouped conv launches, F16) give the
    same bits as `modelq(data, 'q')` + `model(data, 'db')` (reference train.py:308,316); other precisions fall back to
    per-network launches inside the same runner."""
    from agplace_amd import pair
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=prec)
    torch.manual_seed(21)
    mq = randomize_bn(MM(opt=opt)).to(dev).eval()
    md = randomize_bn(DBVanilla2D("db", 256, opt=opt), seed=1).to(dev).eval()
    data = to_dev(nets.synth_query(4, 64, 192, opt, seed=22), dev)
    tiles = {"db_map": torch.randn(4, 1, 3, 64, 64, generator=torch.Generator().manual_seed(23)).to(dev)}
    ref_q, ref_d = mq(data, mode="q"), md(tiles, mode="db")
    ref_q = {k: v.clone() for k, v in ref_q.items()}
    ref_d = ref_d["embedding"].clone()
    opt.query_substreams = qsub
    for _ in range(2):
        out_q, out_d = pair.embed_pair(mq, md, data, tiles)
        torch.cuda.synchronize()
        for k in ref_q:
            assert torch.equal(out_q[k], ref_q[k]), k
        assert torch.equal(out_d["embedding"], ref_d)
    # the training layout of the database batch [b, ndb, nmap, 3, h, w]
    t6 = {"db_map": torch.randn(2, 3, 1, 3, 64, 64, generator=torch.Generator().manual_seed(24)).to(dev)}
    d2 = {k: ([t[:2] for t in v] if isinstance(v, list) else v[:2]) for k, v in data.items()}
    opt.query_substreams = 1
    r6 = md(t6, mode="db")["embedding"].clone()
    _, o6 = pair.embed_pair(mq, md, d2, t6)
    assert o6["embedding"].shape == (2, 3, 256) and torch.equal(o6["embedding"], r6)
    # oracle parity of the paired path itself
    ref = nets.mm_forward_q({k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in data.items()},
                            cpu_state(mq), opt)
    assert rel_l2(out_q["embedding"], ref["embedding"]) < TOL


def test_stage1_chunking_is_bitwise_neutral(dev, monkeypatch):
    """resnet.forward_maps_multi runs stem + stage 1 in chunks of STAGE1_CHUNK images (cache residency of the big
    maps); every chunking gives the same bits, for one trunk and for two trunks in lock-step with uneven chunk sizes."""
    from agplace_amd import resnet
    from agplace_amd.network_mm.image_fe import ImageFE
    torch.manual_seed(41)
    fa = randomize_bn(ImageFE("resnet18", "2_2_2")).to(dev).eval()
    fb = randomize_bn(ImageFE("resnet18", "2_2_2"), seed=3).to(dev).eval()
    xa = torch.randn(11, 3, 64, 96, generator=torch.Generator().manual_seed(1)).to(dev)
    xb = torch.randn(5, 3, 32, 32, generator=torch.Generator().manual_seed(2)).to(dev)
    monkeypatch.setattr(resnet, "STAGE1_CHUNK", 1000)
    ref = [[m.hi.clone() for m in maps] for maps in resnet.forward_maps_multi([fa.fe, fb.fe], [xa, xb], prec=4)]
    for chunk in (4, 3, 1):
        monkeypatch.setattr(resnet, "STAGE1_CHUNK", chunk)
        got = resnet.forward_maps_multi([fa.fe, fb.fe], [xa, xb], prec=4)
        for gm, rm in zip(got, ref):
            for g, r in zip(gm, rm):
                assert torch.equal(g.hi, r)
        one = fa.fe.forward_maps(xa, prec=4)
        for g, r in zip(one, ref[0]):
            assert torch.equal(g.hi, r)


def test_c2_kitti_workload_against_oracle(dev):
    """BASELINE.json config C2 as a workload: KITTI-360-AG cam00 at 224 x 224 -- the database network
    with a ResNet50 trunk (reference network/image_fe.py:47-59, layers '3_4_6', 1024-channel maps, models_baseline/
    dbvanilla2d.py) and a single-camera MM whose Neural-ODE blocks use torchdiffeq's 'rk4' (3/8 rule) with step 0.25 =
    4 steps (network_mm/ffns.py:78-87) -- every output against the fp64 oracle at full size."""
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(dbimage_fe="resnet50", dbimage_fe_layers="3_4_6", odeint_method="rk4", odeint_size=0.25)
    torch.manual_seed(51)
    mq = randomize_bn(MM(opt=opt)).to(dev).eval()
    md = randomize_bn(DBVanilla2D("db", 256, opt=opt), seed=4).to(dev).eval()
    assert md.dbimage_fes[0].last_dim == 1024
    data = nets.synth_query(2, 224, 224, opt, seed=52)
    out = mq(to_dev(data, dev), mode="q")
    pq = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu_state(mq).items()}
    d64 = {k: ([t.double() for t in v] if isinstance(v, list) else v.double()) for k, v in data.items()}
    ref = nets.mm_forward_q(d64, pq, opt)
    errs = {k: rel_l2(out[k], ref[k]) for k in ref}
    print("C2 query", " ".join(f"{k}:{v:.1e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v < TOL, (k, v)
    tiles = torch.randn(2, 1, 3, 224, 224, generator=torch.Generator().manual_seed(53))
    e = md({"db_map": tiles.to(dev)}, mode="db")["embedding"]
    pd = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu_state(md).items()}
    r = nets.dbvanilla2d_forward_db({"db_map": tiles.double()}, pd, opt)["embedding"]
    print(f"C2 database (ResNet50) {rel_l2(e, r):.1e}")
    assert e.shape == (2, 256) and rel_l2(e, r) < TOL and rel_max(e, r) < TOL


@pytest.mark.parametrize("prec,tol", [(4, 5e-4)])
def test_full_size_descriptors_f16_against_oracle(dev, prec, tol):
    """The bench's precision (AGP_PREC_F16: fp16 activations x fp16 weights, one MFMA product) at the bench workload's
    shape: every output descriptor within 5e-4 of the fp32 oracle (bar 1e-3); CPU emulation of the operand roundings
    (tools/prec_plan_emul.py) predicts 2.4e-4 .. 3.8e-4, every conv contributing ~1e-4 in quadrature."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=prec)
    torch.manual_seed(18)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    data = nets.synth_query(2, 224, 1344, opt, seed=19)
    out = model(to_dev(data, dev), mode="q")
    ref = nets.mm_forward_q(data, cpu_state(model), opt)
    errs = {k: rel_l2(out[k], ref[k]) for k in ref}
    print("FULLSIZE", prec, " ".join(f"{k}:{v:.1e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v < tol, (k, v)


def test_pinned_ring_feeds_uint8_camera_and_aerial_tiles(dev):
    """Input pipeline: decoded uint8 tiles staged in pinned host memory, uploaded on a copy stream
    (input_pipeline.PinnedRing), normalised + packed on the device; the paired forward on them equals the forward on the
    host-normalised fp32 tensors the reference's loaders produce (datasets_ws_nuscenes.py:286-304,604-634)."""
    from agplace_amd import ops, pair
    from agplace_amd.input_pipeline import PinnedRing
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(mfma_precision=4)
    torch.manual_seed(61)
    mq = randomize_bn(MM(opt=opt)).to(dev).eval()
    md = randomize_bn(DBVanilla2D("db", 256, opt=opt), seed=5).to(dev).eval()
    b = 3
    ring = PinnedRing({"q": ((b, 2, 64, 64, 3), torch.uint8), "t": ((b, 1, 64, 64, 3), torch.uint8)}, depth=2, device=dev)
    g = torch.Generator().manual_seed(62)
    mean = torch.tensor(ops.IMAGENET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(ops.IMAGENET_STD).view(1, 1, 3, 1, 1)
    base = nets.synth_query(b, 64, 128, opt, seed=63)
    for step in range(3):                           # slot 0 is reused on the third step: release / upload ordering
        s = step % 2
        q8 = torch.randint(0, 256, (b, 2, 64, 64, 3), generator=g, dtype=torch.uint8)
        t8 = torch.randint(0, 256, (b, 1, 64, 64, 3), generator=g, dtype=torch.uint8)
        ring.host(s)["q"].copy_(q8)
        ring.host(s)["t"].copy_(t8)
        ring.upload(s)
        d = ring.acquire(s)
        data = to_dev(base, dev)
        data["query_image"] = d["q"]
        oq, od = pair.embed_pair(mq, md, data, {"db_map": d["t"]})
        ring.release(s)
        qimg = (q8.permute(0, 1, 4, 2, 3).float() / 255 - mean) / std
        ref_in = dict(base)
        ref_in["query_image"] = torch.cat([qimg[:, 0], qimg[:, 1]], dim=-1)
        timg = ((t8.permute(0, 1, 4, 2, 3).float() / 255 - mean) / std)          # [b,1,3,h,w]
        rq = nets.mm_forward_q(ref_in, cpu_state(mq), opt)
        rd = nets.dbvanilla2d_forward_db({"db_map": timg}, cpu_state(md), opt)["embedding"]
        assert rel_l2(oq["embedding"], rq["embedding"]) < TOL
        assert rel_l2(od["embedding"], rd) < TOL


def test_learnable_fusion_weights_gradients_match_oracle(dev):
    """xxx_learnweight=True (reference tools/options.py:139-146): MM's scalar mixing weights are trained parameters; their
    gradients (agp_dot_f32 in WsumFn.backward) against autograd through the fp64 oracle."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    opt = Options(image_learnweight=True, vox_learnweight=True, shallow_learnweight=True, imagevoxorg_learnweight=True,
                  shalloworg_learnweight=True, stg2imagevox_learnweight=True, stg2fuse_learnweight=True,
                  imagevoxorg_weight=0.3, stg2fuse_weight=0.2,
                  final_type=["imageorg", "voxorg", "shalloworg", "stg2image", "stg2vox", "stg2fuse"])
    # The LIBRARY DEFAULT precision (this test used to be pinned to mfma_precision=2).  A mixing weight's gradient is the
    # dot product of G with a descriptor -- near-cancelling on this 64 x 128 input -- so the one-product mode's descriptor error
    # would show up amplified: freeze_backbone() + gradients makes the model run its frozen trunk in the tight two-product mode
    # whatever the inference default is (MM.forward_q), and that is what this test now exercises.
    assert opt.mfma_precision == Options().mfma_precision == 4
    torch.manual_seed(71)
    model = randomize_bn(MM(opt=opt)).to(dev).eval()
    model.freeze_backbone()
    names = ["shallow_weight", "imageorg_weight", "voxorg_weight", "shalloworg_weight", "stg2image_weight",
             "stg2vox_weight", "stg2fuse_weight"]
    assert all(getattr(model, n).requires_grad for n in names)
    data = nets.synth_query(3, 64, 128, opt, seed=72)
    G = torch.randn(3, 256, generator=torch.Generator().manual_seed(73))
    out = model(to_dev(data, dev), mode="q")
    (out["embedding"] * G.to(dev)).sum().backward()
    params = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu_state(model).items()}
    for n in names:
        params[n].requires_grad_(True)
    d64 = {k: ([t.double() for t in v] if isinstance(v, list) else v.double()) for k, v in data.items()}
    import oracle.nets as onets
    orig = onets.basic_block_conv
    onets.basic_block_conv = lambda x, p_, pre, training=False, pattern=None: orig(x.detach(), p_, pre, training, pattern)
    try:
        ref = nets.mm_forward_q(d64, params, opt)
    finally:
        onets.basic_block_conv = orig
    (ref["embedding"] * G.double()).sum().backward()
    checked = 0
    for n in names:
        if params[n].grad is None:
            continue
        g = getattr(model, n).grad
        assert g is not None, n
        assert abs(float(g) - float(params[n].grad)) <= 1e-3 * max(abs(float(params[n].grad)), 1e-3), (n, float(g), float(params[n].grad))
        checked += 1
    assert checked >= 5


def test_fusion_blocks_against_reference_generated_fixture(dev, golden):
    """a5 / a6 pinned to the reference itself: FuseBlockToShallow and Stage2FuseBlockAdd of the product,
    loaded with the reference modules' parameters, against the outputs of the reference's own forward_imgvox
    (tests/golden/make_golden.py section 8)."""
    from agplace_amd.network_mm.fuse_block_toshallow import FuseBlockToShallow
    from agplace_amd.network_mm.stage2fuse_blockadd import Stage2FuseBlockAdd
    from agplace_amd.options import Options
    g = golden("fusion_wiring")
    T = torch.from_numpy
    for direction in ("backward", "forward"):
        opt = Options(diff_direction=direction)
        blk = FuseBlockToShallow(opt=opt)
        blk.load_state_dict({k[len("fbts_p_"):]: T(v) for k, v in g.items() if k.startswith("fbts_p_")}, strict=True)
        blk = blk.to(dev).eval()
        maps = [T(g[f"fbts_map{i}"]).to(dev) for i in range(3)]
        voxs = [T(g[f"fbts_vox{i}"]).to(dev) for i in range(3)]
        y = blk(maps, None, voxs, type="vox")
        assert rel_l2(y, T(g[f"fbts_y_{direction}"])) < 2e-4 and elem_rel(y, T(g[f"fbts_y_{direction}"])) < 2e-3
    for variant, ftype in (("basic", "basic"), ("basic2", "basic_basic")):
        tag = f"stg2_{variant}_"
        opt = Options(stg2fuse_type=ftype)
        st = Stage2FuseBlockAdd(64, 64, 64, 64, opt=opt)
        sd = {k[len(tag + "p_"):]: T(v) for k, v in g.items() if k.startswith(tag + "p_")}
        missing = st.load_state_dict(sd, strict=False)
        assert all(k.startswith(("ffnsvox", "projsvoxfuse", "poolvox")) for k in missing.missing_keys) and not missing.unexpected_keys
        st = st.to(dev).eval()
        for prec, tol in ((3, 5e-5), (4, 1e-3)):
            fo, io, _, vo = st(T(g[tag + "imgmap"]).to(dev), None, (T(g[tag + "voxgem"]).to(dev), T(g[tag + "voxfuse"]).to(dev)),
                               T(g[tag + "fusevec"]).to(dev), type="vox", prec=prec)
            assert rel_l2(fo, T(g[tag + "fuse_out"])) < tol, (variant, prec, rel_l2(fo, T(g[tag + "fuse_out"])))
            assert rel_l2(io, T(g[tag + "img_out"])) < tol
            assert torch.equal(vo.cpu(), T(g[tag + "vox_out"]))


@pytest.mark.parametrize("variant", MM_VARIANTS + [dict(diff_type="fcode@tanh_fcode@relu_fcode@sigmoid", stg2fuse_type="basic_basic"),
                                                   dict(final_type=["imageorg", "stg2fuse"], output_l2=False, final_l2=True)])
def test_fused_vector_path_equals_per_op_path(dev, variant):
    """The two-launch vector path (agp_vecprog_run) against the per-op kernels (agp_linear_fwd, agp_fcode_fwd, ...): the same
    arithmetic up to the order of the MFMA accumulation chains (three chains per product in the program, one in agp_linear_fwd),
    so every output agrees to a few 1e-6 (the split-bf16 product's own error is ~4e-6); option sets the program cannot express (final_fusetype 'cat') fall
    back to the per-op path and agree exactly."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.options import Options
    outs = {}
    for fused in (True, False):
        # split-bf16 maps unless the variant says otherwise: an fp16 stage-2 block turns a 3e-6 difference of its input
        # vector into rounding flips of the map (1.5e-5 on stg2imagevec), which is the maps' precision, not this path's
        opt = Options(**{"mfma_precision": 3, **variant}, fused_vector_path=fused)
        torch.manual_seed(3)
        model = randomize_bn(MM(opt=opt)).to(dev).eval()
        data = nets.synth_query(19, 64, 128, opt, seed=5)          # 19 rows: one full and one ragged 16-row workgroup
        with torch.no_grad():
            outs[fused] = model(to_dev(data, dev), mode="q")
    for k in outs[True]:
        assert outs[True][k].shape == outs[False][k].shape
        assert rel_l2(outs[True][k], outs[False][k]) < 1e-5, (k, rel_l2(outs[True][k], outs[False][k]))
    for mt in ("satellite", "satellite_roadmap"):
        o = {}
        for fused in (True, False):
            opt = Options(maptype=mt, fused_vector_path=fused)
            torch.manual_seed(4)
            m = randomize_bn(DBVanilla2D(mode="db", dim=256, opt=opt)).to(dev).eval()
            x = torch.randn(5, len(mt.split("_")), 3, 64, 64, generator=torch.Generator().manual_seed(1)).to(dev)
            with torch.no_grad():
                o[fused] = m({"db_map": x}, mode="db")["embedding"]
        assert rel_l2(o[True], o[False]) < 1e-5


@pytest.mark.parametrize("prec,tol", [(3, 5e-5), (4, 1e-3)])
def test_mm_full_size_panorama_with_realistic_cloud(dev, prec, tol):
    """the query network at the bench's image size (6-camera panorama 224 x 1344) WITH the sparse-voxel
    branch on clouds of ~8000 voxels per sample (the parity runs above use 200 points on 64 x 128 images): every output
    key against the fp64 oracle."""
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    from oracle import sparse as osp
    opt = Options(mfma_precision=prec)
    torch.manual_seed(19)
    model = MM(opt=opt)
    params = nets.init_mm_params(opt, seed=21)
    model.load_reference_state_dict(params)
    model = model.to(dev).eval()
    data = nets.synth_query(2, 224, 1344, opt, seed=19)
    for k in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse"):
        data.pop(k)
    coords, feats = osp.synth_cloud(2, 8000, extent=120, seed=16)
    data["coords"], data["features"] = coords, feats
    assert all(int((coords[:, 0] == b).sum()) > 5000 for b in range(2))
    with torch.no_grad():
        out = model(to_dev(data, dev), mode="q")
    ref = nets.mm_forward_q(data, params, opt)
    errs = {k: rel_l2(out[k], ref[k]) for k in ref}
    print("FULLCLOUD", prec, " ".join(f"{k}:{v:.1e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v < tol, (k, v)


@pytest.mark.parametrize("vox", [False, True])
def test_two_batches_in_flight_on_two_streams_match_serial_forwards(dev, vox):
    """Workspaces, side streams and the voxel branch's capacity buffers are per CALLING stream: batch A on stream 1 and batch B on
    stream 2, enqueued back to back with nothing between them (bench.py --inflight 2 replays two such graphs), give the bits of the
    two serial forwards -- repeated, so that the second round overwrites the first round's buffers while the other stream runs."""
    from agplace_amd import pair
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    from oracle import sparse as osp
    opt = Options(mfma_precision=4)
    torch.manual_seed(31)
    mq = randomize_bn(MM(opt=opt)).to(dev).eval()
    md = randomize_bn(DBVanilla2D("db", 256, opt=opt), seed=2).to(dev).eval()
    batches = []
    for i in range(2):
        d = nets.synth_query(4, 64, 192, opt, seed=40 + i)
        if vox:
            for k in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse"):
                d.pop(k)
            coords, feats = osp.synth_cloud(4, 300 + 50 * i, extent=24, seed=50 + i)
            d["coords"], d["features"] = coords, feats
        t = {"db_map": torch.randn(4, 1, 3, 64, 64, generator=torch.Generator().manual_seed(60 + i)).to(dev)}
        batches.append((to_dev(d, dev), t))
    ref = []
    for d, t in batches:
        oq, od = pair.embed_pair(mq, md, d, t)
        torch.cuda.synchronize()
        ref.append((oq["embedding"].clone(), od["embedding"].clone()))
    assert not torch.equal(ref[0][0], ref[1][0])
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for _round in range(3):
        outs = []
        for (d, t), s in zip(batches, streams):
            with torch.cuda.stream(s):
                oq, od = pair.embed_pair(mq, md, d, t)
                outs.append((oq["embedding"], od["embedding"]))
        torch.cuda.synchronize()
        for (a, b), (ra, rb) in zip(outs, ref):
            assert torch.equal(a, ra) and torch.equal(b, rb), _round


# ---- the glue rows a1 / a7 / a8 against the reference's OWN forward_resnet / MM.forward_q / DBVanilla2D.forward_db
# (tests/golden/glue.npz, make_golden.py section 13; the oracle side: tests/test_oracle_golden.py::test_glue_*)
GLUE_TOL = {3: 1e-4, 2: 1e-3, 4: 1e-3}


def _glue_load(module, params):
    sd = module.state_dict()
    module.load_state_dict({k: v for k, v in params.items() if k in sd}, strict=False)
    missing = [k for k in sd if k not in params]
    assert all("fc." in k for k in missing), missing[:5]
    return module


@pytest.mark.parametrize("prec", [3, None, 4])
def test_glue_image_fe_matches_the_references_forward_resnet(dev, golden, prec):
    """prec None = what the op-level drop-in exports by default (export_precision: the tight mode): the reference's own
    forward_resnet outputs to 1e-3 on every trunk; prec 4 = the fused models' internal mode (1e-3 on ResNet18, the deep-trunk
    guard F16_DEEP_INTERNAL on ResNet34 / 50)."""
    from agplace_amd.network_mm.image_fe import ImageFE as FEmm
    from agplace_amd.network.image_fe import ImageFE as FEnet
    g = golden("glue")
    x = torch.from_numpy(g["fe_x"]).to(dev)
    for tag, cls, fe_type, layers in (("mm_r18", FEmm, "resnet18", "2_2_2"), ("mm_r34", FEmm, "resnet34", "3_4_6"),
                                      ("net_r18", FEnet, "resnet18", "2_2_2"), ("net_r50", FEnet, "resnet50", "3_4_6")):
        fe = cls(fe_type, layers)
        prm = {"fe." + k: v for k, v in resnet.init_params(fe_type, 3, seed=int(g[f"fe_{tag}_seed"])).items()}
        _glue_load(fe, prm).to(dev).eval()
        assert sorted(k for k in fe.state_dict() if not k.startswith("fe.fc.")) == [str(k) for k in g[f"fe_{tag}_statekeys"]]
        if prec is None:
            maps = fe(x)[1]                          # ImageFE.forward itself: fp32 [b,C,h,w] tensors, default precision
            tol = TOL
        else:
            maps = [m.to_f32() for m in fe.fe.forward_maps(x, prec=prec)]
            tol = 1e-4 if prec == 3 else _fe_bound(prec, fe_type)[0]
        assert len(maps) == 3
        for i, m in enumerate(maps):
            ref = torch.from_numpy(g[f"fe_{tag}_l{i + 1}"])
            print(f"GLUEFE {tag} prec {prec} l{i + 1} rel_l2 {rel_l2(m, ref):.2e}")
            assert rel_l2(m, ref) < tol, (tag, i, rel_l2(m, ref))


@pytest.mark.parametrize("prec", [3, 2, 4])
def test_glue_mm_forward_q_matches_the_references_own_forward(dev, golden, prec):
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options
    from test_oracle_golden import GLUE_MM_VARIANTS, glue_mm_inputs
    g = golden("glue")
    data = to_dev(glue_mm_inputs(g), dev)
    for tag, var in GLUE_MM_VARIANTS:
        opt = Options(mfma_precision=prec, **var)
        model = MM(opt=opt)
        prm = nets.init_mm_params(opt, seed=int(g["mm_seed"]))
        model.load_state_dict(prm, strict=True)
        vox_side = ("vox_fe.", "vox_pool.", "stg2fuseblock.ffnsvox.", "stg2fuseblock.projsvoxfuse.", "stg2fuseblock.poolvox.")
        assert sorted(k for k in model.state_dict() if not k.startswith(vox_side) and "fe.fc." not in k) == [str(k) for k in g["mm_statekeys"]]
        out = model.to(dev).eval()(dict(data), mode="q")
        assert sorted(out) == sorted(["imagevec_org", "voxvec_org", "shallowvec_org", "stg2fusevec", "stg2imagevec", "stg2voxvec", "embedding"])
        for k, v in out.items():
            ref = torch.from_numpy(g[f"mm_{tag}_{k}"])
            assert tuple(v.shape) == tuple(ref.shape), (tag, k)
            assert rel_l2(v, ref) < GLUE_TOL[prec], (tag, k, rel_l2(v, ref))


@pytest.mark.parametrize("prec", [3, 4])
def test_glue_dbvanilla2d_forward_db_matches_the_references_own_forward(dev, golden, prec):
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.options import Options
    g = golden("glue")
    for tag in ("5d", "6d"):
        opt = Options(maptype=str(g[f"db_{tag}_maptype"]), mfma_precision=prec)
        model = DBVanilla2D("db", 256, opt=opt)
        prm = nets.init_db_params(opt, seed=int(g["db_seed"]))
        _glue_load(model, prm).to(dev).eval()
        assert sorted(k for k in model.state_dict() if "fe.fc." not in k) == [str(k) for k in g[f"db_{tag}_statekeys"]]
        e = model({"db_map": torch.from_numpy(g[f"db_{tag}_x"]).to(dev)}, mode="db")["embedding"]
        ref = torch.from_numpy(g[f"db_{tag}_embedding"])
        assert tuple(e.shape) == tuple(ref.shape)
        assert rel_l2(e, ref) < GLUE_TOL[prec], (tag, rel_l2(e, ref))
