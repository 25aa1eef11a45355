#!/usr/bin/env python3
"""Which torch (ATen) ops still launch work inside one eager 64-pair embedding pass, and from which line of the host code:
the step should consist of this library's kernels only (16 copyBuffer + fills + ~20 elementwise launches)."""
import os
import sys
import collections
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench_inputs  # noqa: E402
from agplace_amd import pair  # noqa: E402
from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D  # noqa: E402
from agplace_amd.network_mm.mm import MM  # noqa: E402
from agplace_amd.options import Options  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        base = name.split(".")[1] if name.startswith("aten.") else name
        if base not in ("view", "slice", "select", "detach", "alias", "as_strided", "empty", "empty_like", "empty_strided", "_unsafe_view",
                        "expand", "permute", "unsqueeze", "squeeze", "reshape", "t", "transpose", "_local_scalar_dense", "record_stream",
                        "sym_size", "sym_stride", "stride", "size", "lift_fresh", "unbind", "split", "chunk", "narrow", "_reshape_alias"):
            st = [f for f in traceback.extract_stack() if "/agplace_amd/" in f.filename or f.filename.endswith("bench.py")]
            site = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:])
            self.sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    qs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    opt = Options(mfma_precision=4)
    opt.query_substreams = qs
    mq = MM(opt=opt).to(dev).eval()
    md = DBVanilla2D("db", 256, opt=opt).to(dev).eval()
    data = bench_inputs.synth_query(64, 224, 1344, opt, seed=100)
    data = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in data.items()}
    tiles = torch.randn(64, 1, 3, 224, 224).to(dev)
    for _ in range(2):
        pair.embed_pair(mq, md, data, {"db_map": tiles})
    torch.cuda.synchronize()
    with Spy() as spy:
        pair.embed_pair(mq, md, data, {"db_map": tiles})
    torch.cuda.synchronize()
    for (name, site), n in sorted(spy.sites.items(), key=lambda kv: -kv[1]):
        print(f"{n:4d}  {name:40s} {site}")
    print("total", sum(spy.sites.values()))


if __name__ == "__main__":
    main()
