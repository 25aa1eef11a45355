#!/usr/bin/env python3
"""CPU emulation of per-ROLE operand rounding in the training convolutions (tuning aid, build container only).

Every conv of the oracle's train-mode-BN ResNet trunk runs in fp64 through an autograd Function whose three products
round their operands the way a candidate MFMA plan would:

    forward : z  = conv(rx(x), rw(w))          role "f"
    dgrad   : gx = conv^T(rg(gz), rw(w))       role "d"
    wgrad   : gw = corr(rx(x), rg(gz))         role "w"

and (option "store") the conv INPUT is rounded once where it is stored, so forward and wgrad see the same rounded x.
Roundings: "x" exact (what the hi+lo bf16 pair products are, 2^-17), "b" one bf16 plane, "h" one fp16 plane,
"hs" fp16 after a per-tensor power-of-two scale (fp16 mantissa, no range problem: what a scaled gradient plane is).

The gradient error of every parameter is measured the way tests/test_gpu_train.py::test_resnet_trunk_training_gradients
measures the product: relative L2 against the unrounded fp64 gradients under the SAME activation pattern, next to that
test's tolerance max(1e-3, 3 x the oracle's response to a 1e-5 relative input perturbation).

    python tools/grad_prec_emul.py [--fe resnet18] [--b 3] [--h 64] [--w 96]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import nets, resnet  # noqa: E402

_real_conv2d = F.conv2d


def r_exact(t):
    return t


def r_bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def r_f16(t):
    return t.to(torch.float32).to(torch.float16).to(t.dtype)


def r_f16s(t):
    """fp16 after a power-of-two scale that puts max|t| near 2^14 (the scale is exact, only the mantissa is rounded and
    elements below 2^-24 * scale flush)."""
    m = float(t.abs().max())
    if m == 0.0:
        return t
    import math
    s = 2.0 ** (14 - math.ceil(math.log2(m)))
    return (t * s).to(torch.float32).to(torch.float16).to(t.dtype) / s


ROUND = {"x": r_exact, "b": r_bf16, "h": r_f16, "hs": r_f16s}


class EmulConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, plan):
        ctx.stride, ctx.padding, ctx.plan = stride, padding, plan
        xs = ROUND[plan["store"]](x)
        ctx.save_for_backward(xs, w)
        return _real_conv2d(ROUND[plan["f"][0]](xs), ROUND[plan["f"][1]](w), None, stride, padding)

    @staticmethod
    def backward(ctx, g):
        xs, w = ctx.saved_tensors
        plan = ctx.plan
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.nn.grad.conv2d_input(xs.shape, ROUND[plan["d"][1]](w), ROUND[plan["d"][0]](g), ctx.stride, ctx.padding)
        if ctx.needs_input_grad[1]:
            gw = torch.nn.grad.conv2d_weight(ROUND[plan["w"][0]](xs), w.shape, ROUND[plan["w"][1]](g), ctx.stride, ctx.padding)
        return gx, gw, None, None, None


STATE = {"plan": None}


def conv2d_emul(x, w, b=None, stride=1, padding=0, *a, **k):
    if STATE["plan"] is None:
        return _real_conv2d(x, w, b, stride, padding, *a, **k)
    assert b is None
    return EmulConv.apply(x, w, stride, padding, STATE["plan"])


def P(store="x", f="xx", d="xx", w="xx"):
    def two(s):
        out, i = [], 0
        while i < len(s):
            if s[i:i + 2] == "hs":
                out.append("hs"); i += 2
            else:
                out.append(s[i]); i += 1
        assert len(out) == 2, s
        return tuple(out)
    return {"store": store, "f": two(f), "d": two(d), "w": two(w)}


# (name, plan, MFMA passes forward / dgrad / wgrad)
PLANS = [
    ("x3 everywhere (today)",            P(), "3/3/3"),
    ("wgrad bf16 x bf16",                P(w="bb"), "3/3/1"),
    ("wgrad x exact, g bf16",            P(w="xb"), "3/3/2"),
    ("wgrad x bf16, g exact",            P(w="bx"), "3/3/2"),
    ("wgrad f16 x f16s",                 P(w="hhs"), "3/3/1"),
    ("dgrad g bf16, w exact",            P(d="bx"), "3/2/3"),
    ("dgrad g exact, w bf16",            P(d="xb"), "3/2/3"),
    ("dgrad g bf16, w bf16",             P(d="bb"), "3/1/3"),
    ("dgrad g f16s, w exact",            P(d="hsx"), "3/2/3"),
    ("dgrad g f16s, w f16",              P(d="hsh"), "3/1/3"),
    ("fwd x bf16, w exact",              P(f="bx"), "2/3/3"),
    ("fwd x f16, w exact (F16W2)",       P(f="hx"), "2/3/3"),
    ("fwd x f16, w f16",                 P(f="hh"), "1/3/3"),
    ("store f16 (x f16 in fwd+wgrad)",   P(store="h"), "2/3/2"),
    ("store f16, w f16 fwd",             P(store="h", f="xh"), "1/3/2"),
    ("store f16; dgrad f16s,x; wgrad x,f16s", P(store="h", d="hsx", w="xhs"), "2/2/1"),
    ("all one-pass f16 (w f16 too)",     P(store="h", f="xh", d="hsh", w="xhs"), "1/1/1"),
    ("all one-pass bf16",                P(store="b", f="xb", d="bb", w="xb"), "1/1/1"),
    ("wgrad x exact, g f16s",            P(w="xhs"), "3/3/2"),
    ("wgrad x f16, g exact",             P(w="hx"), "3/3/2"),
    ("dgrad g exact, w f16",             P(d="xh"), "3/2/3"),
    ("W1+D2: wgrad h,hs; dgrad hs,x",    P(w="hhs", d="hsx"), "3/2/1"),
    ("W1+D1: wgrad h,hs; dgrad hs,h",    P(w="hhs", d="hsh"), "3/1/1"),
    ("W2+D2: wgrad x,hs; dgrad hs,x",    P(w="xhs", d="hsx"), "3/2/2"),
    ("W2'+D2: wgrad h,x; dgrad hs,x",    P(w="hx", d="hsx"), "3/2/2"),
    ("W1+D2': wgrad h,hs; dgrad x,h",    P(w="hhs", d="xh"), "3/2/1"),
]


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fe", default="resnet18")
    ap.add_argument("--b", type=int, default=3)
    ap.add_argument("--h", type=int, default=64)
    ap.add_argument("--w", type=int, default=96)
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    torch.set_num_threads(8)
    params = resnet.init_params(a.fe, 3, seed=a.seed, dtype=torch.float64)
    for k, v in params.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(a.b, 3, a.h, a.w, generator=g, dtype=torch.float64)
    dims = resnet.stage_dims(a.fe, 3)
    Gm = [torch.randn(a.b, c, generator=g, dtype=torch.float64) for c in dims]
    Gg = torch.randn(a.b, dims[-1], generator=g, dtype=torch.float64)

    # the activation pattern of the exact forward, imposed on every run (kink flips are not what is measured here)
    pattern = {}
    with torch.no_grad():
        p = params
        s = F.relu(resnet._bn(resnet._conv(x, p, "conv1", 2, 3), p, "bn1", True))
        pattern["relu"] = s > 0
        pattern["maxpool_idx"] = F.max_pool2d(s, 3, 2, 1, return_indices=True)[1]
        kind, layers = resnet.ARCH[a.fe]
        h = F.max_pool2d(s, 3, 2, 1)
        for li in range(3):
            for bi in range(layers[li]):
                pre = f"layer{li + 1}.{bi}."
                stride = 2 if (li > 0 and bi == 0) else 1
                rec = {}

                def relu_rec(z, pat, key, rec=rec):
                    rec[key] = z > 0
                    return F.relu(z)
                orig = resnet._relu
                resnet._relu = relu_rec
                try:
                    h = resnet._block(h, p, pre, kind, stride, True, None)
                finally:
                    resnet._relu = orig
                pattern.update(rec)

    def run(xin, plan):
        for v in params.values():
            v.grad = None
        STATE["plan"] = plan
        F.conv2d = conv2d_emul
        try:
            outs = resnet.forward_resnet(xin, params, a.fe, 3, training=True, pattern=pattern)
            loss = sum((o.mean((2, 3)) * Gm[i]).sum() for i, o in enumerate(outs))
            loss = loss + (nets.gem(outs[-1], torch.tensor([3.0], dtype=torch.float64)).flatten(1) * Gg).sum()
            loss.backward()
        finally:
            F.conv2d = _real_conv2d
            STATE["plan"] = None
        return [o.detach() for o in outs], {k: v.grad.clone() for k, v in params.items() if v.grad is not None}

    outs0, g0 = run(x, None)
    gp = torch.Generator().manual_seed(11)
    outs_p, g_p = run(x * (1 + 1e-5 * torch.randn(x.shape, generator=gp, dtype=torch.float64)), None)
    names = [k for k in g0 if not k.startswith("fc.")]
    tol = {k: max(1e-3, 3 * rel(g_p[k], g0[k])) for k in names}
    print(f"{a.fe} {a.b}x3x{a.h}x{a.w}; conditioning (response to 1e-5): maps "
          + " ".join(f"{rel(op, o):.1e}" for o, op in zip(outs0, outs_p))
          + f"; gradients median {sorted(rel(g_p[k], g0[k]) for k in names)[len(names) // 2]:.1e}"
          + f" max {max(rel(g_p[k], g0[k]) for k in names):.1e}")
    print("plan".ljust(44) + "passes  maps(l1 l2 l3)              grad: median   max      worst err/tol  (parameter)   #>tol  #>1e-3")
    for name, plan, passes in PLANS:
        outs, gr = run(x, plan)
        errs = {k: rel(gr[k], g0[k]) for k in names}
        ratio = {k: errs[k] / tol[k] for k in names}
        worst = max(ratio, key=ratio.get)
        srt = sorted(errs.values())
        print(name.ljust(44) + passes.ljust(8) + " ".join(f"{rel(o, o0):.1e}" for o, o0 in zip(outs, outs0)).ljust(28)
              + f"{srt[len(srt) // 2]:9.1e} {srt[-1]:9.1e}   {ratio[worst]:6.2f} ({worst})".ljust(60)
              + f"{sum(r > 1 for r in ratio.values()):5d} {sum(e > 1e-3 for e in errs.values()):6d}")


if __name__ == "__main__":
    main()
