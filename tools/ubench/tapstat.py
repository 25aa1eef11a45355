import torch, sys
sys.path.insert(0, '.')
import bench_inputs
from agplace_amd import ops
from agplace_amd.sparse import SparseTensor
dev = torch.device('cuda')
coords, feats = bench_inputs.synth_cloud_lidar(64, 8000, seed=400)
sp = SparseTensor.from_coords_capacity(feats.to(dev), coords.to(dev), 64, ops.Workspace())
for lvl in range(1, 4):
    sp = sp.strided()[0]
    n = int(sp.n_dev.item())
    nbr = sp.kernel_map(3)[:, :n]
    perm = sp.zperm()[:n].long()
    present = (nbr != sp.n)
    for name, order in (("natural", torch.arange(n, device=dev)), ("zplane", perm)):
        for BM in (128, 256):
            pr = present[:, order]
            nt = n // BM
            t = pr[:, :nt * BM].view(27, nt, BM).any(dim=2)       # [27][tiles]
            print(f"level {lvl} rows {n} order {name} BM {BM}: active taps per tile {t.sum().item() / nt:.2f} of 27")
    z = (sp.keys[:n] & 0xffff) - 32768
    print("  z planes:", torch.unique(z, return_counts=True))
