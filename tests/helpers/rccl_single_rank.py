"""Child program of tests/test_gpu_rccl.py: a ONE-rank RCCL ("nccl") process group on cuda:0 -- the collective library
loads, creates its communicator and runs all-reduce / all-gather on its own stream; GradBuckets' stream ordering
(gradient kernels on two compute streams -> async all-reduce on RCCL's stream -> finish) is exercised against it.
Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from agplace_amd import parallel  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
t = torch.arange(1 << 20, device=dev, dtype=torch.float32)
ref = t.clone()
dist.all_reduce(t)
res["allreduce_identity"] = bool(torch.equal(t, ref))
out = torch.empty(4096, 512, device=dev)
x = torch.randn(4096, 512, device=dev)
dist.all_gather_into_tensor(out, x)
res["allgather_identity"] = bool(torch.equal(out, x))

# GradBuckets: gradients produced by long-running kernels on TWO streams, the collectives forced although world == 1
torch.manual_seed(0)
ps = [torch.nn.Parameter(torch.randn(1024, 1024, device=dev)) for _ in range(6)]
gb = parallel.GradBuckets(ps, bucket_mb=8.0, collective_on_single_rank=True)
res["nbuckets"] = len(gb.buckets)
side = torch.cuda.Stream(device=dev)
bigs = [torch.randn(4096, 4096, device=dev) * 1e-2, torch.randn(4096, 4096, device=dev) * 1e-2]    # one per compute stream
torch.cuda.synchronize()
ok = True
for step in range(3):
    gb.zero_grad()
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    for i, p in enumerate(reversed(ps)):
        st = side if i % 2 else cur
        with torch.cuda.stream(st):
            for _ in range(4):
                bigs[i % 2] = bigs[i % 2] @ bigs[i % 2] * 1e-2      # keep the stream busy: the gradient write below lands late
            p.grad.add_(float(i + 1 + step))
            gb.mark_ready([p])
    gb.finish()
    torch.cuda.synchronize()
    for i, p in enumerate(reversed(ps)):
        ok = ok and bool(torch.all(p.grad == float(i + 1 + step)))
res["gradbuckets_values_after_async_allreduce"] = ok
res["handles_waited"] = len(gb.handles) == 0
gb.close()

# the bench's own training step (MM + DBVanilla2D, the voxel branch's outputs as fixed tensors) with its gradient exchange on
# this one-rank RCCL group: after the first step every bucket's all-reduce is launched while backward still runs, nothing is
# held back by the silent voxel-side parameters, and their zeros are not exchanged
import types  # noqa: E402
import bench  # noqa: E402
from agplace_amd.options import Options  # noqa: E402
args = types.SimpleNamespace(train_steps=2, sync_bn=False, force_buckets=True)
tr = bench.train_measurement(args, Options(), dev, 0, 1, parallel)
res["train_grad_exchange"] = tr["grad_exchange"]
res["train_ms_per_step"] = tr["ms_per_step"]
dist.destroy_process_group()
print(json.dumps(res))
