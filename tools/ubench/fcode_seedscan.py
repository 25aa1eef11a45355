import torch, sys
sys.path.insert(0, '.')
from oracle import ode
from agplace_amd.network_mm.ffns import FCODE
from agplace_amd.options import Options
dev = torch.device('cuda')
def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm())
for seed in range(12):
    worst = 0.0
    for act in ("relu", "tanh", "sigmoid", "id"):
        for method, step in (("euler", 0.1), ("midpoint", 0.3), ("rk4", 0.25), ("rk4", 0.1)):
            g = torch.Generator().manual_seed(21)
            torch.manual_seed(seed)
            for b in (5, 16, 35):
                m = FCODE(256, act, opt=Options(odeint_method=method, odeint_size=step)).to(dev)
                x = torch.randn(b, 256, generator=g); a1 = torch.randn(b, 256, generator=g) * 0.3; G = torch.randn(b, 256, generator=g)
                xd, a1d = x.to(dev).requires_grad_(True), a1.to(dev).requires_grad_(True)
                y = m(xd, add1=a1d); (y * G.to(dev)).sum().backward()
                W = m.func.func.fc.weight.detach().cpu().double().requires_grad_(True)
                B = m.func.func.fc.bias.detach().cpu().double().requires_grad_(True)
                xr = x.double().requires_grad_(True)
                yr = ode.fcode(xr + a1.double(), W, B, act, method, step); (yr * G.double()).sum().backward()
                worst = max(worst, rel_l2(xd.grad, xr.grad), rel_l2(m.func.func.fc.weight.grad, W.grad), rel_l2(m.func.func.fc.bias.grad, B.grad))
    print("seed", seed, "worst", f"{worst:.2e}")
