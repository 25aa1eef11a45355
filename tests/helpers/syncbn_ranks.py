"""Child program of tests/test_gpu_rccl.py::test_sync_batchnorm_two_ranks_equal_one_rank_whole_batch: two ranks on the ONE
GPU of the test box over gloo (RCCL cannot put two ranks on one device), each training on HALF of a batch with
parallel.enable_sync_batchnorm(); rank 0 then runs the whole batch alone with per-rank (= whole-batch) statistics and compares
embeddings, averaged gradients and running statistics.  A sparse MinkFPN (different row counts per rank) is checked the same
way.  Rank 0 prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from agplace_amd import parallel  # noqa: E402
from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D  # noqa: E402
from agplace_amd.options import Options  # noqa: E402
from gpu_util import randomize_bn, rel_l2  # noqa: E402

rank, world, _ = parallel.init_from_env(backend="gloo")
assert world == 2
dev = torch.device("cuda:0")
torch.manual_seed(8)
model = randomize_bn(DBVanilla2D(mode="db", dim=256, opt=Options())).to(dev).train()
state = {k: v.clone() for k, v in model.state_dict().items()}
x = torch.randn(8, 1, 1, 3, 64, 64, generator=torch.Generator().manual_seed(4)).to(dev)
G = torch.randn(8, 1, 256, generator=torch.Generator().manual_seed(5)).to(dev)
params = [p for p in model.parameters()]

# ---- two ranks, synchronised statistics
parallel.enable_sync_batchnorm()
lo, hi = 4 * rank, 4 * rank + 4
o = model({"db_map": x[lo:hi]}, mode="db")["embedding"]
(o * G[lo:hi]).sum().backward()
parallel.allreduce_grads(params, average=True)
outs = parallel.all_gather_rows(o.detach().reshape(4, -1), equal=True)
g2 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
rm2 = model.dbimage_fes[0].fe.bn1.running_mean.clone()
rv2 = model.dbimage_fes[0].fe.layer2[0].bn2.running_var.clone()

# ---- sparse branch: rank r holds cloud r
from agplace_amd.sparse import ECABasicBlock, MinkFPN, SparseTensor  # noqa: E402
from agplace_amd.sparse import train as st  # noqa: E402
from agplace_amd.sparse.modules import global_avg_pool  # noqa: E402
from oracle import sparse as osp  # noqa: E402
vp = osp.init_vox_params(seed=8)
net = MinkFPN(1, 256, 0, 5, ECABasicBlock, [1, 1, 1], [64, 128, 256])
net.load_state_dict({k[len("vox_fe."):]: v for k, v in vp.items()}, strict=True)
net = net.to(dev).train()
vstate = {k: v.clone() for k, v in net.state_dict().items()}
coords, feats = osp.synth_cloud(2, 300 + 0, extent=24, seed=9)
mine = coords[:, 0] == rank
c_r = coords[mine].clone()
c_r[:, 0] = 0
top_r, _ = st.MinkFPNTrain(net).forward(SparseTensor.from_coords(feats[mine].to(dev), c_r.to(dev), nbatch=1))
v2 = parallel.all_gather_rows(global_avg_pool(top_r), equal=True)
vrm2 = net.bns[1].bn.running_mean.clone()

res = None
if rank == 0:
    # ---- one rank, the whole batch, per-rank (= whole-batch) statistics
    parallel.enable_sync_batchnorm(None)
    model.load_state_dict(state)
    for p in params:
        p.grad = None
    o1 = model({"db_map": x}, mode="db")["embedding"]
    (o1 * G).sum().backward()
    o1 = o1.detach().clone()
    g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    # conditioning: a randomly initialised train-mode-BN trunk turns a 1e-5 relative input perturbation into ReLU flips that
    # move gradients by ~1e-2 (tests/test_gpu_train.py); the two runs differ by rounding of that size (split-bf16 storage), so
    # each parameter's bound is 3x the one-rank run's own response to such a perturbation
    model.load_state_dict(state)
    for p in params:
        p.grad = None
    xp = x * (1 + 1e-5 * torch.randn(x.shape, generator=torch.Generator().manual_seed(6)).to(dev))
    (model({"db_map": xp}, mode="db")["embedding"] * G).sum().backward()
    g1p = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    dg = {n: rel_l2(2.0 * g2[n], g1[n]) / max(1e-3, 3 * rel_l2(g1p[n], g1[n])) for n in g1 if float(g1[n].abs().max()) > 0}
    model.load_state_dict(state)
    model({"db_map": x}, mode="db")                      # (running statistics of the unperturbed whole batch, compared below)
    net.load_state_dict(vstate)
    top1, _ = st.MinkFPNTrain(net).forward(SparseTensor.from_coords(feats.to(dev), coords.to(dev), nbatch=2))
    res = {
        "world": world,
        "outputs": rel_l2(outs, o1.detach().reshape(8, -1)),
        "grads_worst_over_bound": max(dg.values()), "grads_checked": len(dg), "grads_median_over_bound": sorted(dg.values())[len(dg) // 2],
        "running_mean": rel_l2(rm2, model.dbimage_fes[0].fe.bn1.running_mean),
        "running_var": rel_l2(rv2, model.dbimage_fes[0].fe.layer2[0].bn2.running_var),
        "sparse_outputs": rel_l2(v2, global_avg_pool(top1)),
        "sparse_running_mean": rel_l2(vrm2, net.bns[1].bn.running_mean),
    }
dist.barrier()
if rank == 0:
    print(json.dumps(res))
dist.destroy_process_group()
