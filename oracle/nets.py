"""Oracle: GeM, fusion blocks, MM.forward_q, DBVanilla2D, NetVLAD (TEST INFRASTRUCTURE).

Plain functional PyTorch on flat parameter dicts that use the reference's
state_dict key names.  `opt` is any namespace carrying the
reference's hot-path flags (tools/options.py:101-155); agplace_amd.options.Options
has the same field names and defaults.

Sparse-voxel inputs (MinkowskiEngine, out of scope -- SURVEY.md 8f) enter as
precomputed dense tensors in data_dict:
    vox_levels   : [ [b,64], [b,128], [b,256] ]  = ME.MinkowskiGlobalPooling()(v_i).F
                   (fuse_block_toshallow.py:83)
    voxfeatvec   : [b,256] = MinkGeM(voxfeatmap)            (mm.py:89, before normalize)
    stg2voxvec   : [b,256] = poolvox(ffnvox(voxmap+proj))   (stage2fuse_blockadd.py:197-201)
    voxvec_fuse  : [b,256] = GlobalAvgPool(projvoxfuse(voxmap)).F  (stage2fuse_blockadd.py:207-211)
"""
import torch
import torch.nn.functional as F

from . import ode
from . import resnet

LN_EPS = 1e-5


# --------------------------------------------------------------------------- GeM
def gem(x, p, eps=1e-6):
    """reference network_mm/image_pooling.py:16 -> [b,c,1,1].

    avg_pool2d(x.clamp(min=eps).pow(p), (H,W)).pow(1/p); p is a 1-element tensor.
    """
    return F.avg_pool2d(x.clamp(min=eps).pow(p), (x.size(-2), x.size(-1))).pow(1.0 / p)


def gem_flat(x, p, eps=1e-6):
    """reference network/image_pooling.py:15-17 -> [b,c]."""
    return gem(x, p, eps).view(x.size(0), -1)


def avgpool_vec(x):
    """reference fuse_block_toshallow.py:82: adaptive_avg_pool2d(e,1).flatten(1)."""
    return F.adaptive_avg_pool2d(x, output_size=1).flatten(1)


# ------------------------------------------------------------------ stage-1 fusion
def fuse_block_toshallow(imagemaplist, voxveclist, params, prefix, opt):
    """reference network_mm/fuse_block_toshallow.py:79-121 (forward_imgvox, non-cde).

    voxveclist are the already globally-pooled vox vectors (see module docstring).
    """
    n = len(imagemaplist)
    imageveclist = [avgpool_vec(e) for e in imagemaplist]
    fusevec = 0
    for it in range(n):
        i = it if opt.diff_direction == "forward" else n - 1 - it
        imagevec, voxvec = imageveclist[i], voxveclist[i]
        if i < n - 1:  # Linear up-dims; last level is Identity (:24-29)
            imagevec = F.linear(imagevec, params[f"{prefix}updimsimg.{i}.weight"],
                                params[f"{prefix}updimsimg.{i}.bias"])
            voxvec = F.linear(voxvec, params[f"{prefix}updimsvox.{i}.weight"],
                              params[f"{prefix}updimsvox.{i}.bias"])
        fusevec = fusevec + imagevec + voxvec
        fusevec = ode.diff_block(fusevec, params, f"{prefix}blocks.{i}.", opt.diff_type,
                                 opt.odeint_method, opt.odeint_size)
    return fusevec


# ------------------------------------------------------------------ stage-2 fusion
def basic_block_conv(x, params, prefix, training=False, pattern=None):
    """reference stage2fuse_blockadd.py:61-79 (BasicBlock: convs WITH bias).
    `pattern` (test infrastructure): imposed ReLU masks under keys prefix+"relu1"/"relu2"."""
    def bn(t, name):
        return F.batch_norm(t, params[name + ".running_mean"], params[name + ".running_var"],
                            params[name + ".weight"], params[name + ".bias"],
                            training=training, momentum=0.0, eps=1e-5)
    out = F.conv2d(x, params[prefix + "conv1.weight"], params[prefix + "conv1.bias"], 1, 1)
    out = resnet._relu(bn(out, prefix + "bn1"), pattern, prefix + "relu1")
    out = F.conv2d(out, params[prefix + "conv2.weight"], params[prefix + "conv2.bias"], 1, 1)
    out = bn(out, prefix + "bn2")
    return resnet._relu(out + x, pattern, prefix + "relu2")


def basic_mlp(x, params, prefix):
    """reference stage2fuse_blockadd.py:82-100 (Basic: fc-LN-ReLU-fc-LN, +id, ReLU)."""
    d = x.shape[-1]
    out = F.linear(x, params[prefix + "fc1.weight"], params[prefix + "fc1.bias"])
    out = F.layer_norm(out, (d,), params[prefix + "ln1.weight"], params[prefix + "ln1.bias"], LN_EPS)
    out = F.relu(out)
    out = F.linear(out, params[prefix + "fc2.weight"], params[prefix + "fc2.bias"])
    out = F.layer_norm(out, (d,), params[prefix + "ln2.weight"], params[prefix + "ln2.bias"], LN_EPS)
    return F.relu(out + x)


def ffn_fuse(x, params, prefix, stg2fuse_type):
    """reference stage2fuse_blockadd.py:117-135 (FFNFuse: sum of Basic blocks)."""
    outs = []
    for j, e in enumerate(stg2fuse_type.split("_")):
        if e != "basic":
            raise NotImplementedError(e)
        outs.append(basic_mlp(x, params, f"{prefix}ffns.{j}."))
    return sum(outs)


def stage2_fuse_block_add(imgmap, fusevec, stg2voxvec, voxvec_fuse, params, prefix, opt,
                          training=False, pattern=None, voxmap=None):
    """reference stage2fuse_blockadd.py:180-219 (forward_imgvox), image side dense.

    Returns (fusevec, imgoutvec, None, voxoutvec).  stg2voxvec / voxvec_fuse stand in
    for the sparse branch outputs of the (single, stg2nlayers=1) layer.
    """
    if opt.stg2_type != "full":
        raise NotImplementedError(opt.stg2_type)
    imgoutvec = None
    for i in range(opt.stg2nlayers):
        if opt.stg2_useproj:
            fusevec_img = F.linear(fusevec, params[f"{prefix}projsfuseimg.{i}.0.weight"],
                                   params[f"{prefix}projsfuseimg.{i}.0.bias"])
        else:
            fusevec_img = fusevec
        imgmap = imgmap + fusevec_img.unsqueeze(-1).unsqueeze(-1)
        if voxmap is not None:
            # sparse side, stage2fuse_blockadd.py:194-211 (oracle/sparse.py)
            from . import sparse
            fusevec_vox = F.linear(fusevec, params[f"{prefix}projsfusevox.{i}.0.weight"],
                                   params[f"{prefix}projsfusevox.{i}.0.bias"]) if opt.stg2_useproj else fusevec
            voxmap = sparse.broadcast_add(voxmap, fusevec_vox)
            voxmap = sparse.eca_basic_block(voxmap, params, f"{prefix}ffnsvox.{i}.", training, pattern)
            stg2voxvec = sparse.mink_gem(voxmap, params[f"{prefix}poolvox.p"])
            vf = sparse.conv(voxmap, params[f"{prefix}projsvoxfuse.{i}.0.kernel"], 1) if opt.stg2_useproj else voxmap
            voxvec_fuse = sparse.global_avg(vf)
        imgmap = basic_block_conv(imgmap, params, f"{prefix}ffnsimg.{i}.", training, pattern)
        imgoutvec = gem(imgmap, params[f"{prefix}poolimage.p"]).flatten(1)
        if opt.stg2fuse_type is not None:
            if opt.stg2_useproj:
                imgmap_fuse = F.conv2d(imgmap, params[f"{prefix}projsimgfuse.{i}.0.weight"],
                                       params[f"{prefix}projsimgfuse.{i}.0.bias"])
            else:
                imgmap_fuse = imgmap
            imgvec_fuse = F.adaptive_avg_pool2d(imgmap_fuse, [1, 1]).squeeze(-1).squeeze(-1)
            fusevec = fusevec + imgvec_fuse + voxvec_fuse
            fusevec = ffn_fuse(fusevec, params, f"{prefix}ffnsfuse.{i}.", opt.stg2fuse_type)
    return fusevec, imgoutvec, None, stg2voxvec


# ------------------------------------------------------------------------ MM (query)
def mm_forward_q(data_dict, params, opt, training=False, pattern=None):
    """reference network_mm/mm.py:70-160 (MM.forward_q), vox branch as inputs.
    `pattern`: imposed activation pattern of the conv parts (keys as resnet.forward_resnet, the
    stage-2 block under "stg2fuseblock.ffnsimg.0.relu1/2"); test infrastructure only."""
    image = data_dict["query_image"]
    output = []
    nst = len(opt.mm_imgfe_layers.split("_"))
    maps = resnet.forward_resnet(image, params, opt.mm_imgfe, nst, prefix="image_fe.fe.",
                                 training=training, pattern=pattern)
    imagefeatmap = maps[-1]
    imagefeatvec = gem(imagefeatmap, params["image_pool.p"]).flatten(1)
    if opt.output_l2:
        imagefeatvec = F.normalize(imagefeatvec, dim=-1)
    imagefeatvec_org = imagefeatvec
    output.append(imagefeatvec * params["image_weight"])

    voxmap = None
    if "coords" in data_dict:
        # the voxel branch itself (mm.py:86-89): MinkFPN + MinkGeM on the sparse tensor
        from . import sparse
        sp = sparse.from_coords(data_dict["features"].to(image.dtype), data_dict["coords"], nbatch=image.shape[0])
        voxmap, voxmaplist = sparse.minkfpn(sp, params, "vox_fe.", nlevels=len(opt.mm_voxfe_planes.split("_")),
                                            training=training, pattern=pattern, num_top_down=getattr(opt, "mm_voxfe_ntd", 0))
        data_dict = dict(data_dict)
        data_dict["voxfeatvec"] = sparse.mink_gem(voxmap, params["vox_pool.p"])
        data_dict["vox_levels"] = [sparse.global_avg(e) for e in voxmaplist]
        data_dict["stg2voxvec"] = data_dict["voxvec_fuse"] = None
    voxfeatvec = data_dict["voxfeatvec"]
    if opt.output_l2:
        voxfeatvec = F.normalize(voxfeatvec, dim=-1)
    voxfeatvec_org = voxfeatvec
    output.append(voxfeatvec * params["vox_weight"])

    shallowfeatvec = fuse_block_toshallow(maps, data_dict["vox_levels"], params,
                                          "fuseblocktoshallow.", opt)
    shallowfeatvecorg = shallowfeatvec
    if opt.output_l2:
        shallowfeatvec = F.normalize(shallowfeatvec, dim=-1)
    output.append(shallowfeatvec * params["shallow_weight"])

    stg2fusevec, stg2imagevec, _, stg2voxvec = stage2_fuse_block_add(
        imagefeatmap, output[-1], data_dict["stg2voxvec"], data_dict["voxvec_fuse"],
        params, "stg2fuseblock.", opt, training, pattern, voxmap=voxmap)
    stg2fusevec = F.linear(stg2fusevec, params["stg2fusefc.weight"], params["stg2fusefc.bias"])

    final = []
    ft = opt.final_type if isinstance(opt.final_type, (list, tuple)) else opt.final_type.split("_")
    if "imageorg" in ft:
        final.append(imagefeatvec_org * params["imageorg_weight"])
    if "voxorg" in ft:
        final.append(voxfeatvec_org * params["voxorg_weight"])
    if "shalloworg" in ft:
        final.append(shallowfeatvec * params["shalloworg_weight"])
    if "stg2image" in ft:
        final.append(stg2imagevec * params["stg2image_weight"])
    if "stg2vox" in ft:
        final.append(stg2voxvec * params["stg2vox_weight"])
    if "stg2fuse" in ft:
        final.append(stg2fusevec * params["stg2fuse_weight"])
    if opt.final_fusetype == "add":
        x = sum(final)
    elif opt.final_fusetype == "cat":
        x = torch.cat(final, dim=-1)
    elif opt.final_fusetype == "catadd":
        x = torch.cat(final[:-1], dim=-1) + final[-1]
    else:
        raise NotImplementedError(opt.final_fusetype)
    if opt.final_l2:
        x = F.normalize(x, dim=-1)
    return {
        "imagevec_org": imagefeatvec_org, "voxvec_org": voxfeatvec_org,
        "shallowvec_org": shallowfeatvecorg, "stg2fusevec": stg2fusevec,
        "stg2imagevec": stg2imagevec, "stg2voxvec": stg2voxvec, "embedding": x,
    }


# ------------------------------------------------------------------- DBVanilla2D
def db_mlp(x, params, prefix):
    """reference models_baseline/dbvanilla2d.py:17-28 (MLP: Linear-LN-ReLU-Linear)."""
    d = params[prefix + "seq.0.weight"].shape[0]
    out = F.linear(x, params[prefix + "seq.0.weight"], params[prefix + "seq.0.bias"])
    out = F.layer_norm(out, (d,), params[prefix + "seq.1.weight"], params[prefix + "seq.1.bias"], LN_EPS)
    out = F.relu(out)
    return F.linear(out, params[prefix + "seq.3.weight"], params[prefix + "seq.3.bias"])


def dbvanilla2d_forward_db(data_dict, params, opt, training=False, patterns=None):
    """reference models_baseline/dbvanilla2d.py:50-101 (forward_db).
    `patterns[i]`: imposed activation pattern of map type i's trunk (test infrastructure)."""
    db_map = data_dict["db_map"]
    if db_map.dim() == 5:
        mode = "cachetest"
        b, nmap, c, h, w = db_map.shape
        db_map = db_map.unsqueeze(1)
        ndb = 1
    elif db_map.dim() == 6:
        mode = "train"
        b, ndb, nmap, c, h, w = db_map.shape
    else:
        raise NotImplementedError
    assert c == 3
    db_map = db_map.permute(2, 0, 1, 3, 4, 5).contiguous()
    nst = len(opt.dbimage_fe_layers.split("_"))
    vecs = []
    for i in range(nmap):
        j = 0 if getattr(opt, "share_dbfe", False) else i
        x = db_map[i].view(-1, c, h, w)
        m = resnet.forward_resnet(x, params, opt.dbimage_fe, nst, prefix=f"dbimage_fes.{j}.fe.",
                                  training=training, pattern=patterns[i] if patterns else None)[-1]
        v = gem_flat(m, params[f"dbimage_pools.{j}.p"])
        vecs.append(db_mlp(v, params, f"dbimage_mlps.{j}."))
    out = torch.stack(vecs, dim=1)
    if opt.output_l2:
        out = F.normalize(out, p=2, dim=-1)
    out = out.mean(dim=1).view(b, ndb, -1)
    if mode == "cachetest":
        out = out.view(b, -1)
    if opt.final_l2:
        out = F.normalize(out, p=2, dim=-1)
    return {"embedding": out}


# ------------------------------------------------------------------------- NetVLAD
def netvlad(x, conv_weight, centroids, normalize_input=True):
    """reference model/aggregation.py:126-146 (NetVLAD.forward, work_with_tokens=False).

    conv_weight [K,D,1,1] (bias=False), centroids [K,D] -> [N, K*D].
    Vectorised restatement of the per-cluster python loop:
        vlad[n,k,:] = sum_hw a[n,k,hw] * (x[n,:,hw] - c[k,:])
    """
    N, D = x.shape[:2]
    K = centroids.shape[0]
    if normalize_input:
        x = F.normalize(x, p=2, dim=1)
    xf = x.reshape(N, D, -1)
    a = F.softmax(F.conv2d(x, conv_weight).reshape(N, K, -1), dim=1)
    vlad = torch.einsum("nkp,ndp->nkd", a, xf) - a.sum(-1).unsqueeze(-1) * centroids.unsqueeze(0)
    vlad = F.normalize(vlad, p=2, dim=2)
    return F.normalize(vlad.reshape(N, -1), p=2, dim=1)


# ------------------------------------------------------------------ synthetic params
def init_mm_params(opt, seed=0, dtype=torch.float32):
    """Seeded MM parameter dict with the reference's state_dict keys (incl. the MinkowskiEngine-named
    voxel-branch keys: vox_fe.*, vox_pool.p, stg2fuseblock.{ffnsvox,projsvoxfuse,poolvox}.*)."""
    g = torch.Generator().manual_seed(seed + 1000)
    nst = len(opt.mm_imgfe_layers.split("_"))
    p = {"image_fe.fe." + k: v for k, v in
         resnet.init_params(opt.mm_imgfe, nst, seed=seed, dtype=dtype).items()}
    D = opt.mm_stg2fuse_dim

    def lin(name, o, i):
        bound = 1.0 / (i ** 0.5)
        p[name + ".weight"] = ((torch.rand(o, i, generator=g) * 2 - 1) * bound).to(dtype)
        p[name + ".bias"] = ((torch.rand(o, generator=g) * 2 - 1) * bound).to(dtype)

    def ln(name, d):
        p[name + ".weight"] = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
        p[name + ".bias"] = (0.1 * torch.randn(d, generator=g)).to(dtype)

    p["image_pool.p"] = torch.ones(1, dtype=dtype) * 3
    img_dims = [int(e) for e in opt.mm_imgfe_planes.split("_")]
    vox_dims = [int(e) for e in opt.mm_voxfe_planes.split("_")]
    n = len(vox_dims)
    for i in range(n):
        for j, _ in enumerate(ode.parse_diff_type(opt.diff_type)):
            lin(f"fuseblocktoshallow.blocks.{i}.blocks.{j}.func.func.fc", D, D)
        if i < n - 1:
            lin(f"fuseblocktoshallow.updimsimg.{i}", D, img_dims[i])
            lin(f"fuseblocktoshallow.updimsvox.{i}", D, vox_dims[i])
    C = opt.mm_imgfe_dim
    for i in range(opt.stg2nlayers):
        lin(f"stg2fuseblock.projsfuseimg.{i}.0", C, D)
        lin(f"stg2fuseblock.projsfusevox.{i}.0", opt.mm_voxfe_dim, D)
        lin(f"stg2fuseblock.projsimgfuse.{i}.0", D, C)
        p[f"stg2fuseblock.projsimgfuse.{i}.0.weight"] = \
            p[f"stg2fuseblock.projsimgfuse.{i}.0.weight"].view(D, C, 1, 1)
        for cn, bnn in (("conv1", "bn1"), ("conv2", "bn2")):
            pre = f"stg2fuseblock.ffnsimg.{i}."
            bound = 1.0 / ((C * 9) ** 0.5)
            p[pre + cn + ".weight"] = ((torch.rand(C, C, 3, 3, generator=g) * 2 - 1) * bound).to(dtype)
            p[pre + cn + ".bias"] = ((torch.rand(C, generator=g) * 2 - 1) * bound).to(dtype)
            p[pre + bnn + ".weight"] = (0.5 + torch.rand(C, generator=g)).to(dtype)
            p[pre + bnn + ".bias"] = (0.2 * torch.randn(C, generator=g)).to(dtype)
            p[pre + bnn + ".running_mean"] = (0.3 * torch.randn(C, generator=g)).to(dtype)
            p[pre + bnn + ".running_var"] = (0.5 + 1.5 * torch.rand(C, generator=g)).to(dtype)
            p[pre + bnn + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        for j, _ in enumerate(opt.stg2fuse_type.split("_")):
            pre = f"stg2fuseblock.ffnsfuse.{i}.ffns.{j}."
            lin(pre + "fc1", D, D)
            ln(pre + "ln1", D)
            lin(pre + "fc2", D, D)
            ln(pre + "ln2", D)
    p["stg2fuseblock.poolimage.p"] = torch.ones(1, dtype=dtype) * 3
    p["stg2fuseblock.poolfuse.p"] = torch.ones(1, dtype=dtype) * 3
    # voxel branch (MinkFPN + stage-2 ECABasicBlock / 1x1 projection), MinkowskiEngine key names
    from . import sparse
    V = opt.mm_voxfe_dim
    p.update(sparse.init_vox_params(tuple(vox_dims), seed=seed + 7, dtype=dtype, prefix="vox_fe.", num_top_down=getattr(opt, "mm_voxfe_ntd", 0),
                                    extra_blocks=[(f"stg2fuseblock.ffnsvox.{i}.", V) for i in range(opt.stg2nlayers)]))
    p["vox_pool.p"] = torch.ones(1, dtype=dtype) * 3
    p["stg2fuseblock.poolvox.p"] = torch.ones(1, dtype=dtype) * 3
    for i in range(opt.stg2nlayers):
        p[f"stg2fuseblock.projsvoxfuse.{i}.0.kernel"] = (torch.randn(V, D, generator=g) * (2.0 / D) ** 0.5).to(dtype)
    lin("stg2fusefc", D, D)
    for name, val in (("image_weight", opt.image_weight), ("vox_weight", opt.vox_weight),
                      ("shallow_weight", opt.shallow_weight),
                      ("imageorg_weight", opt.imagevoxorg_weight),
                      ("voxorg_weight", opt.imagevoxorg_weight),
                      ("shalloworg_weight", opt.shalloworg_weight),
                      ("stg2image_weight", opt.stg2imagevox_weight),
                      ("stg2vox_weight", opt.stg2imagevox_weight),
                      ("stg2fuse_weight", opt.stg2fuse_weight)):
        p[name] = torch.tensor(val, dtype=dtype)
    return p


def init_db_params(opt, seed=0, dtype=torch.float32):
    """Seeded DBVanilla2D parameter dict with the reference's state_dict keys."""
    g = torch.Generator().manual_seed(seed + 2000)
    nst = len(opt.dbimage_fe_layers.split("_"))
    p = {}
    for i, _ in enumerate(opt.maptype.split("_")):
        p.update({f"dbimage_fes.{i}.fe." + k: v for k, v in
                  resnet.init_params(opt.dbimage_fe, nst, seed=seed + 10 + i, dtype=dtype).items()})
        p[f"dbimage_pools.{i}.p"] = torch.ones(1, dtype=dtype) * 3
        cin = resnet.stage_dims(opt.dbimage_fe, nst)[-1]
        d = opt.features_dim
        for name, o, ii in ((f"dbimage_mlps.{i}.seq.0", d, cin), (f"dbimage_mlps.{i}.seq.3", d, d)):
            bound = 1.0 / (ii ** 0.5)
            p[name + ".weight"] = ((torch.rand(o, ii, generator=g) * 2 - 1) * bound).to(dtype)
            p[name + ".bias"] = ((torch.rand(o, generator=g) * 2 - 1) * bound).to(dtype)
        p[f"dbimage_mlps.{i}.seq.1.weight"] = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
        p[f"dbimage_mlps.{i}.seq.1.bias"] = (0.1 * torch.randn(d, generator=g)).to(dtype)
    return p


from bench_inputs import synth_query  # noqa: E402,F401  (shared synthetic-input generator)
