import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
from gpu_util import rel_l2
from agplace_amd import ops
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
g = torch.Generator().manual_seed(0)
for (cin, cout, k, s, p, h, w, n, amp) in [(1024, 256, 1, 1, 0, 14, 14, 2, 1.0), (256, 1024, 1, 1, 0, 14, 14, 2, 1.0), (256, 256, 3, 1, 1, 14, 14, 2, 1.0),
                                           (512, 256, 1, 1, 0, 28, 28, 2, 1.0), (256, 256, 3, 2, 1, 28, 28, 2, 1.0), (512, 1024, 1, 2, 0, 28, 28, 2, 1.0),
                                           (1024, 256, 1, 1, 0, 14, 14, 2, 300.0), (256, 256, 3, 1, 1, 14, 14, 2, 300.0), (256, 256, 3, 1, 1, 14, 14, 2, 3000.0)]:
    x = torch.randn(n, cin, h, w, generator=g) * amp
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, s, p)
    ho, wo = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
    out = []
    for prec in (2, 4):
        xm = ops.pack_f32(x.to(dev), cin, 1, prec)
        cw = ops.ConvWeights(wt.to(dev), None, None, s, p)
        o = ops.SplitMap.alloc(n, ho, wo, cout, 1, prec, dev)
        ops.conv2d(xm, cw, o, relu=False, prec=prec)
        out.append(rel_l2(o.to_f32(), ref))
    print((cin, cout, k, s, h, amp), "prec2 %.1e prec4 %.1e" % tuple(out), "max|ref| %.0f" % float(ref.abs().max()))
