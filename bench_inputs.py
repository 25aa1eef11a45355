"""Synthetic inputs of the benchmarks and tests: seeded, shape-faithful stand-ins for what the
reference's dataloaders produce (datasets/datasets_ws_nuscenes.py:106-167,551-646).  No arithmetic of the hot path
lives here; bench.py, tools/ and oracle/ all import these generators."""
import torch


def synth_query(b, h, w, opt, seed=0, dtype=torch.float32):
    """Synthetic query data_dict: N(0,1) image (real inputs are mean/std-normalised, datasets_ws_nuscenes.py:610),
    U(0,1) stand-ins for the voxel branch's dense outputs."""
    g = torch.Generator().manual_seed(seed + 3000)
    vox_dims = [int(e) for e in opt.mm_voxfe_planes.split("_")]
    D = opt.mm_stg2fuse_dim
    return {
        "query_image": torch.randn(b, 3, h, w, generator=g).to(dtype),
        "vox_levels": [torch.rand(b, c, generator=g).to(dtype) for c in vox_dims],
        "voxfeatvec": torch.rand(b, vox_dims[-1], generator=g).to(dtype),
        "stg2voxvec": torch.rand(b, opt.mm_voxfe_dim, generator=g).to(dtype),
        "voxvec_fuse": torch.rand(b, D, generator=g).to(dtype),
    }


def synth_cloud(nbatch, npts, extent=24, seed=0):
    """Random occupied voxels on a few planes / lines (LiDAR-like sparsity): coords float [N,4] (batch id + xyz, as
    ME.utils.sparse_quantize + the collate function give them), features ones [N,1] (mm.py:87)."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for b in range(nbatch):
        xy = torch.randint(0, extent, (npts, 2), generator=g)
        z = torch.randint(0, 4, (npts, 1), generator=g)
        rows.append(torch.cat([torch.full((npts, 1), b), xy, z], 1))
    c = torch.cat(rows, 0).float()
    return c, torch.ones((c.shape[0], 1))


def synth_cloud_lidar(nbatch, npts, seed=0):
    """`npts` random occupied voxels per sample on a 128 x 128 x 8 grid around the origin (x, y in [-64, 64), z in [-3, 5)):
    the size of a quantised nuScenes LiDAR sweep (~8000 voxels per sample after ME.utils.sparse_quantize); a few per cent of
    the draws coincide and are merged by the sparse tensor, as duplicates in real data are."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for b in range(nbatch):
        xy = torch.randint(-64, 64, (npts, 2), generator=g)
        z = torch.randint(-3, 5, (npts, 1), generator=g)
        rows.append(torch.cat([torch.full((npts, 1), b), xy, z], 1))
    c = torch.cat(rows, 0).float()
    return c, torch.ones((c.shape[0], 1))


def resnet_gmacs(fe_type, nstages, h, w):
    """Algorithmic multiply-accumulates of a truncated torchvision ResNet (stem + `nstages` stages) on one h x w image."""
    arch = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet34": ("basic", [3, 4, 6, 3]), "resnet50": ("bottleneck", [3, 4, 6, 3])}
    kind, layers = arch[fe_type]
    exp = 1 if kind == "basic" else 4
    planes_all = [64, 128, 256, 512]
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    macs = ho * wo * 64 * 147
    ho, wo = (ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1
    inplanes = 64
    for li in range(nstages):
        planes = planes_all[li]
        for bi in range(layers[li]):
            stride = 2 if (li > 0 and bi == 0) else 1
            hi, wi = ho, wo
            ho, wo = (hi + 2 - 3) // stride + 1, (wi + 2 - 3) // stride + 1
            if kind == "basic":
                macs += ho * wo * planes * inplanes * 9 + ho * wo * planes * planes * 9
            else:
                macs += hi * wi * planes * inplanes + ho * wo * planes * planes * 9 + ho * wo * planes * exp * planes
            if stride != 1 or inplanes != planes * exp:
                macs += ho * wo * planes * exp * inplanes
            inplanes = planes * exp
    return macs


def kernel_source_sha16(root):
    """Fingerprint of the device code (agplace_amd/csrc + the C header): profiles/*pmc*.json carry the value they were
    measured at, and bench.py quotes their HBM-traffic figures only while it still matches (a stale counter summary must
    not survive a kernel change silently; .git does not travel to the GPU box, the sources do)."""
    import glob
    import hashlib
    import os
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "agplace_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(root, "agplace_amd", "csrc", "*.hpp")) +
                   [os.path.join(root, "include", "agplace_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
