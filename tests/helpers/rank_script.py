"""A stand-in rank program for tests/test_launcher.py: joins the gloo group the launcher's environment describes,
all-reduces ones, rank 0 prints one JSON line; `--fail-rank R --rc C` makes rank R exit with C after the collective."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from agplace_amd import parallel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--fail-rank", type=int, default=-1)
ap.add_argument("--rc", type=int, default=0)
args = ap.parse_args()
rank, world, local = parallel.init_from_env(backend="gloo")
assert world == args.gpus, (world, args.gpus)
t = torch.ones(1)
if world > 1:
    dist.all_reduce(t)
print(f"noise from rank {rank}", file=sys.stderr)
if rank == 0:
    print(json.dumps({"world": world, "allreduce_ones": float(t), "local": local}))
if world > 1:
    dist.destroy_process_group()
sys.exit(args.rc if rank == args.fail_rank else 0)
