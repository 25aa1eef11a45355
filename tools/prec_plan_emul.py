#!/usr/bin/env python3
"""CPU emulation of per-layer MFMA precision plans (tuning aid, build container only).

Runs the oracle's MM.forward_q / DBVanilla2D forward in fp64 with the operand roundings a plan would
apply -- fp16 activations into every conv; conv weights either exact (hi+lo product) or rounded to
fp16 (hi product only) -- and prints the relative L2 error of every output descriptor against the
unrounded fp64 forward.  Used to choose which convs keep the weight-residual (`lo`) product
("precision plan").

    python tools/prec_plan_emul.py [--b 2] [--h 224] [--w 1344] [--plans all,none,deep,...]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import nets  # noqa: E402
from agplace_amd.options import Options  # noqa: E402

_real_conv2d = F.conv2d
STATE = {"plan": None, "idx": 0, "log": []}


def f16(t):
    return t.to(torch.float16).to(t.dtype)


def conv2d_emul(x, w, b=None, stride=1, padding=0, *a, **k):
    plan = STATE["plan"]
    if plan is None:
        return _real_conv2d(x, w, b, stride, padding, *a, **k)
    i = STATE["idx"]
    STATE["idx"] += 1
    mode = plan(i, tuple(w.shape), stride)
    STATE["log"].append((i, tuple(w.shape), stride, mode))
    x = f16(x)
    if mode == "hi":
        w = f16(w)
    return _real_conv2d(x, w, b, stride, padding, *a, **k)


def run(data, params, opt, plan):
    STATE["plan"], STATE["idx"], STATE["log"] = plan, 0, []
    F.conv2d = conv2d_emul
    try:
        with torch.no_grad():
            return nets.mm_forward_q(data, params, opt)
    finally:
        F.conv2d = _real_conv2d
        STATE["plan"] = None


def rel(a, b):
    return float((a - b).norm() / b.norm())


# conv index -> name for the ResNet18 query network (call order of the oracle)
NAMES = ["stem", "l1.0.c1", "l1.0.c2", "l1.1.c1", "l1.1.c2",
         "l2.0.c1", "l2.0.c2", "l2.0.ds", "l2.1.c1", "l2.1.c2",
         "l3.0.c1", "l3.0.c2", "l3.0.ds", "l3.1.c1", "l3.1.c2",
         "s2.c1", "s2.c2", "s2.proj"]


def make_plan(exact_names):
    ex = set(exact_names)

    def plan(i, shape, stride):
        n = NAMES[i] if i < len(NAMES) else f"conv{i}"
        return "exact" if n in ex else "hi"
    return plan


PLANS = {
    "all": NAMES,
    "none": [],
    "stem": ["stem"],
    "l1": [n for n in NAMES if n.startswith("l1")],
    "l2": [n for n in NAMES if n.startswith("l2")],
    "l3": [n for n in NAMES if n.startswith("l3")],
    "s2": [n for n in NAMES if n.startswith("s2")],
    "l3+s2": [n for n in NAMES if n.startswith(("l3", "s2"))],
    "l2+l3+s2": [n for n in NAMES if n.startswith(("l2", "l3", "s2"))],
    "not3x3s1": ["stem", "l2.0.c1", "l2.0.ds", "l3.0.c1", "l3.0.ds", "s2.proj"],
    "not3x3s1+s2": ["stem", "l2.0.c1", "l2.0.ds", "l3.0.c1", "l3.0.ds", "s2.proj", "s2.c1", "s2.c2"],
    "not3x3s1+l3+s2": ["stem", "l2.0.c1", "l2.0.ds", "l3.0.c1", "l3.0.ds", "s2.proj", "s2.c1", "s2.c2",
                       "l3.0.c2", "l3.1.c1", "l3.1.c2"],
    "c2only": [n for n in NAMES if n.endswith(("c2", "ds", "proj")) or n == "stem"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=2)
    ap.add_argument("--h", type=int, default=224)
    ap.add_argument("--w", type=int, default=1344)
    ap.add_argument("--seed", type=int, default=18)
    ap.add_argument("--plans", type=str, default=",".join(PLANS))
    ap.add_argument("--single", action="store_true", help="also: every conv alone in hi mode (its own contribution)")
    a = ap.parse_args()
    torch.set_num_threads(8)
    opt = Options()
    params = nets.init_mm_params(opt, seed=a.seed, dtype=torch.float64)
    data = nets.synth_query(a.b, a.h, a.w, opt, seed=a.seed + 1, dtype=torch.float64)
    ref = run(data, params, opt, None)
    keys = list(ref.keys())
    print("plan".ljust(18) + " ".join(k[:12].rjust(12) for k in keys))
    for name in a.plans.split(","):
        out = run(data, params, opt, make_plan(PLANS[name]))
        print(name.ljust(18) + " ".join(f"{rel(out[k], ref[k]):12.2e}" for k in keys))
    if a.single:
        for n in NAMES:
            out = run(data, params, opt, make_plan([m for m in NAMES if m != n]))
            print(("only-hi:" + n).ljust(18) + " ".join(f"{rel(out[k], ref[k]):12.2e}" for k in keys))


if __name__ == "__main__":
    main()
