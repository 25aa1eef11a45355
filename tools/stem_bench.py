"""Time the stem kernels on the bench's shapes: python tools/stem_bench.py [n] [walk]  (walk = 0: the one-workgroup-per-block
kernels of the development library instead of the walking kernel; needs `make -C agplace_amd/csrc tuning`).
Prints us per launch of pack + stem (packed input) and of the raw-input stem, for the panorama and the aerial tile."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WALK = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if not WALK:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _tuning  # noqa: E402
    _tuning.set_switch("STEM_WALK", 0)
from agplace_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    scale, shift = 0.5 + torch.rand(64, generator=g), 0.3 * torch.randn(64, generator=g)
    cw = ops.ConvWeights(wt.to(dev), scale.to(dev), shift.to(dev), 2, 3, stem=True)
    for (h, w) in ((224, 1344), (224, 224)):
        x = torch.randn(n, 3, h, w, generator=g).to(dev)
        h1, w1 = ops.conv_out_size(h, 7, 2, 3), ops.conv_out_size(w, 7, 2, 3)
        h2, w2 = ops.conv_out_size(h1, 3, 2, 1), ops.conv_out_size(w1, 3, 2, 1)
        out = ops.SplitMap.alloc(n, h2, w2, 64, 1, 4, dev)
        xm = ops.pack_f32(x, 4, 3, 4)
        t_pack = timeit(lambda: ops.pack_f32(x, 4, 3, 4))
        t_stem = timeit(lambda: ops.stem_pool(xm, cw, out, prec=4))
        ref = out.hi.clone()
        t_raw = timeit(lambda: ops.stem_pool_raw(x, cw, out))
        same = torch.equal(ref, out.hi)
        print(f"STEM walk={WALK} n={n} {h}x{w}: pack {t_pack:.1f} us, stem {t_stem:.1f} us, raw stem {t_raw:.1f} us, raw==packed {same}")


if __name__ == "__main__":
    main()
