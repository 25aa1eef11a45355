#!/usr/bin/env python3
"""Cost of the sparse-voxel branch: MM.forward_q on [b,3,224,1344] panoramas with the dense voxel stand-ins
vs. with coords/features (`--points` occupied voxels per sample) through MinkFPN + the stage-2 sparse side."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from agplace_amd.network_mm.mm import MM  # noqa: E402
from agplace_amd.options import Options  # noqa: E402
import bench_inputs as onets  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    opt = Options()
    torch.manual_seed(0)
    model = MM(opt=opt).to(dev).eval()
    data = onets.synth_query(a.batch, 224, 1344, opt, seed=1)
    data = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in data.items()}
    g = torch.Generator().manual_seed(2)
    rows = []
    for b in range(a.batch):
        xy = torch.randint(-64, 64, (a.points, 2), generator=g)
        z = torch.randint(-3, 5, (a.points, 1), generator=g)
        rows.append(torch.cat([torch.full((a.points, 1), b), xy, z], 1))
    coords = torch.cat(rows, 0).float().to(dev)
    sp = dict(data)
    sp["coords"], sp["features"] = coords, torch.ones((coords.shape[0], 1), device=dev)

    def run(d):
        with torch.no_grad():
            for _ in range(2):
                model(d, mode="q")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                model(d, mode="q")
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e3
    dense_ms, sparse_ms = run(data), run(sp)
    print(json.dumps({"batch": a.batch, "voxels_per_sample_requested": a.points, "ms_standins": round(dense_ms, 3),
                      "ms_with_voxel_branch": round(sparse_ms, 3), "voxel_branch_ms": round(sparse_ms - dense_ms, 3)}))


if __name__ == "__main__":
    main()
