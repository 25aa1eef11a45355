#!/usr/bin/env python3
"""Times the conv weight gradient alone (agp_conv2d_wgrad_param) on the shapes of the training step (16 panoramas 224 x 1344 +
176 tiles of 256^2: the 3x3 stride-1 convs of layer 1 / 2 / 3, the stride-2 stage entries and their 1x1 downsamples),
three-product bf16 kernel against the one-pass fp16 kernels (agp_conv_desc.in_h16 / out_absmax), and checks the two against
each other.

    python tools/wgrad_bench.py [--iters 20]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from agplace_amd import _lib, ops  # noqa: E402
from agplace_amd._lib import ptr, check  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.load()
    # (name, [(n, h, w)], channels): the query panoramas and the database tiles are separate launches in the step
    # (name, n, h, w, cin, cout, k, stride)
    shapes = [("l1 tiles", 176, 64, 64, 64, 64, 3, 1), ("l1 pano", 16, 56, 336, 64, 64, 3, 1), ("l2 tiles", 176, 32, 32, 128, 128, 3, 1),
              ("l2 pano", 16, 28, 168, 128, 128, 3, 1), ("l3 tiles", 176, 16, 16, 256, 256, 3, 1), ("l3 pano", 16, 14, 84, 256, 256, 3, 1),
              ("l2 entry t", 176, 64, 64, 64, 128, 3, 2), ("l2 entry p", 16, 56, 336, 64, 128, 3, 2), ("l3 entry t", 176, 32, 32, 128, 256, 3, 2),
              ("l3 entry p", 16, 28, 168, 128, 256, 3, 2), ("l2 ds t", 176, 64, 64, 64, 128, 1, 2), ("l2 ds p", 16, 56, 336, 64, 128, 1, 2),
              ("l3 ds t", 176, 32, 32, 128, 256, 1, 2), ("l3 ds p", 16, 28, 168, 128, 256, 1, 2)]
    for name, n, h, w, c, co, k, st in shapes:
        pd = (k - 1) // 2
        ho, wo = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n, c, h, w, generator=g).relu_().to(dev)
        gz = (torch.randn(n, co, ho, wo, generator=g) * 1e-3).to(dev)
        xm = ops.pack_f32(x, c, 1, 3).with_h16()
        xm.h16[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1).half()
        gm = ops.pack_f32(gz, co, 1, 3)
        amax = gz.abs().amax(dim=(0, 2, 3)).float().contiguous()
        amax_bits = amax.view(torch.int32).clone()
        d = _lib.ConvDesc()
        d.in_hi, d.in_lo, d.out_hi, d.out_lo = ptr(xm.hi), ptr(xm.lo), ptr(gm.hi), ptr(gm.lo)
        d.n, d.hin, d.win, d.pin, d.cin, d.in_w_step = n, h, w, 1, c, c
        d.hout, d.wout, d.cout, d.pout = ho, wo, co, 1
        d.kh, d.kw, d.stride, d.pad, d.prec = k, k, st, pd, 3
        nbytes = L.agp_conv2d_wgrad_workspace_bytes(C.byref(d))
        d.in_h16, d.out_absmax = ptr(xm.h16), ptr(amax_bits)
        nbytes = max(nbytes, L.agp_conv2d_wgrad_workspace_bytes(C.byref(d)))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        gw3 = torch.empty(co, c, k, k, device=dev)
        gw1 = torch.empty(co, c, k, k, device=dev)
        s = _lib.stream()

        def run(one_pass, out):
            if one_pass:
                bits = amax_bits.clone()
                d.in_h16, d.out_absmax = ptr(xm.h16), ptr(bits)
            else:
                d.in_h16, d.out_absmax = None, None
            check(L.agp_conv2d_wgrad_param(C.byref(d), ptr(out), 0, ptr(ws), nbytes, s), "agp_conv2d_wgrad_param")

        def timed(one_pass, out):
            for _ in range(3):
                run(one_pass, out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                run(one_pass, out)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters * 1e3
        t3, t1 = timed(False, gw3), timed(True, gw1)
        gf = 2 * n * ho * wo * c * co * k * k / 1e9
        err = float((gw1 - gw3).norm() / gw3.norm())
        print(f"{name:10s} {n:4d}x{h:3d}x{w:4d}x{c:3d}->{co:3d} k{k} s{st}  3-pass {t3:7.1f} us ({gf / t3:5.2f} PF)   1-pass {t1:7.1f} us ({gf / t1:5.2f} PF)"
              f"   (incl. the reduce launch)   rel diff {err:.1e}")


if __name__ == "__main__":
    main()
