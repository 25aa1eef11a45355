"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rel_max(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_rel(a, b, q=0.999):
    """Element-wise relative error with a floor: the q-quantile of |a - b| / (|b| + 1e-3 max|b|).  rel_l2 and rel_max are
    norm-wise; this one bounds (nearly) every element, small ones included."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    r = (a - b).abs() / (b.abs() + 1e-3 * b.abs().max().clamp_min(1e-30))
    if r.numel() > 4_000_000:
        r = r[torch.randperm(r.numel(), generator=torch.Generator().manual_seed(0))[:4_000_000]]
    return float(torch.quantile(r, q))


def frac_within(a, b, rtol, floor=1e-2):
    """Fraction of the elements with |a - b| <= rtol * (|b| + floor * max|b|): an element-wise bound that CAN fail (a descriptor
    whose error doubles drops below 0.99 at the tolerances the model tests use), unlike a quantile bound set far above the data."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float(((a - b).abs() <= rtol * (b.abs() + floor * b.abs().max().clamp_min(1e-30))).double().mean())


def randomize_bn(module, seed=0):
    """Non-trivial BatchNorm affine + running stats so that BN folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = (0.5 + torch.rand(m.num_features, generator=g)).to(m.weight.device)
            m.bias.data = (0.2 * torch.randn(m.num_features, generator=g)).to(m.weight.device)
            m.running_mean.data = (0.3 * torch.randn(m.num_features, generator=g)).to(m.weight.device)
            m.running_var.data = (0.5 + 1.5 * torch.rand(m.num_features, generator=g)).to(m.weight.device)
    return module


def cpu_state(module):
    return {k: v.detach().cpu() for k, v in module.state_dict().items()}


def to_dev(d, dev):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            out[k] = v.to(dev)
        elif isinstance(v, (list, tuple)):
            out[k] = [t.to(dev) for t in v]
        else:
            out[k] = v
    return out
