"""Oracle: truncated torchvision-style ResNet forward (TEST INFRASTRUCTURE).

Reference call sites: network_mm/image_fe.py:97-113 and network/image_fe.py:112-128
(`forward_resnet`): conv1 -> bn1 -> relu -> maxpool -> layer1 -> layer2 -> layer3
(-> layer4 when 4 entries in `layers`); returns the list of stage outputs.

The ResNet definition itself lives in torchvision==0.15.1 (README.md:19), which is
not in /root/reference and not installed here -> PARITY UNPINNED for the network
definition; restated from the published architecture (He et al. 2015; torchvision
"v1.5": the stride sits on the 3x3 conv of a Bottleneck):

  stem       : Conv2d(3,64,7,stride 2,pad 3,bias=False) BN ReLU MaxPool(3,2,1)
  BasicBlock : conv3x3(stride) BN ReLU conv3x3 BN (+ downsample: conv1x1(stride) BN) add ReLU
  Bottleneck : conv1x1 BN ReLU conv3x3(stride) BN ReLU conv1x1(x4) BN (+ downsample) add ReLU
  resnet18 [2,2,2,2] basic, resnet34 [3,4,6,3] basic, resnet50 [3,4,6,3] bottleneck
  planes 64/128/256/512, stage strides 1/2/2/2.

Parameters are a flat dict with torchvision's state_dict key names
('conv1.weight', 'bn1.running_mean', 'layer2.0.downsample.0.weight', ...), which is
also the checkpoint-compatibility surface of the product modules.
"""
import math
import torch
import torch.nn.functional as F

ARCH = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
}
PLANES = [64, 128, 256, 512]
BN_EPS = 1e-5


def expansion(kind):
    return 1 if kind == "basic" else 4


def stage_dims(fe_type, nstages):
    kind, _ = ARCH[fe_type]
    return [p * expansion(kind) for p in PLANES[:nstages]]


def init_params(fe_type, nstages=3, seed=0, dtype=torch.float32, randomize_bn=True):
    """Seeded random parameters with torchvision key names.

    Conv: Kaiming-normal fan_out (torchvision's init).  BN affine/running stats are
    randomised (not the identity defaults) so that BN folding is actually exercised.
    Includes the unused `fc` (image_fe.py leaves it registered).
    """
    g = torch.Generator().manual_seed(seed)
    kind, layers = ARCH[fe_type]
    p = {}

    def conv(name, cout, cin, k):
        std = math.sqrt(2.0 / (cout * k * k))
        p[name + ".weight"] = (torch.randn(cout, cin, k, k, generator=g) * std).to(dtype)

    def bn(name, c):
        if randomize_bn:
            p[name + ".weight"] = (0.5 + torch.rand(c, generator=g)).to(dtype)
            p[name + ".bias"] = (0.2 * torch.randn(c, generator=g)).to(dtype)
            p[name + ".running_mean"] = (0.3 * torch.randn(c, generator=g)).to(dtype)
            p[name + ".running_var"] = (0.5 + 1.5 * torch.rand(c, generator=g)).to(dtype)
        else:
            p[name + ".weight"] = torch.ones(c, dtype=dtype)
            p[name + ".bias"] = torch.zeros(c, dtype=dtype)
            p[name + ".running_mean"] = torch.zeros(c, dtype=dtype)
            p[name + ".running_var"] = torch.ones(c, dtype=dtype)
        p[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    inplanes = 64
    exp = expansion(kind)
    for li in range(nstages):
        planes = PLANES[li]
        for bi in range(layers[li]):
            stride = 2 if (li > 0 and bi == 0) else 1
            pre = f"layer{li + 1}.{bi}."
            if kind == "basic":
                conv(pre + "conv1", planes, inplanes, 3)
                bn(pre + "bn1", planes)
                conv(pre + "conv2", planes, planes, 3)
                bn(pre + "bn2", planes)
            else:
                conv(pre + "conv1", planes, inplanes, 1)
                bn(pre + "bn1", planes)
                conv(pre + "conv2", planes, planes, 3)
                bn(pre + "bn2", planes)
                conv(pre + "conv3", planes * exp, planes, 1)
                bn(pre + "bn3", planes * exp)
            if stride != 1 or inplanes != planes * exp:
                conv(pre + "downsample.0", planes * exp, inplanes, 1)
                bn(pre + "downsample.1", planes * exp)
            inplanes = planes * exp
    fc_in = 512 * exp
    p["fc.weight"] = (torch.randn(1000, fc_in, generator=g) / math.sqrt(fc_in)).to(dtype)
    p["fc.bias"] = torch.zeros(1000, dtype=dtype)
    return p


def _bn(x, p, name, training=False):
    return F.batch_norm(
        x, p[name + ".running_mean"].to(x.dtype), p[name + ".running_var"].to(x.dtype),
        p[name + ".weight"].to(x.dtype), p[name + ".bias"].to(x.dtype),
        training=training, momentum=0.0 if training else 0.1, eps=BN_EPS)


def _conv(x, p, name, stride, pad):
    return F.conv2d(x, p[name + ".weight"].to(x.dtype), None, stride, pad)


def _relu(z, pattern, key):
    """ReLU; with `pattern[key]` (a 0/1 mask) the activation pattern is imposed instead of
    recomputed, which makes gradient comparisons immune to sign flips of near-zero
    pre-activations between two arithmetic implementations (test infrastructure only)."""
    if pattern is None or key not in pattern:
        return F.relu(z)
    return z * pattern[key].to(z.dtype)


def _block(x, p, pre, kind, stride, training, pattern=None):
    idt = x
    if kind == "basic":
        o = _relu(_bn(_conv(x, p, pre + "conv1", stride, 1), p, pre + "bn1", training), pattern, pre + "relu1")
        o = _bn(_conv(o, p, pre + "conv2", 1, 1), p, pre + "bn2", training)
        last = "relu2"
    else:
        o = _relu(_bn(_conv(x, p, pre + "conv1", 1, 0), p, pre + "bn1", training), pattern, pre + "relu1")
        o = _relu(_bn(_conv(o, p, pre + "conv2", stride, 1), p, pre + "bn2", training), pattern, pre + "relu2")
        o = _bn(_conv(o, p, pre + "conv3", 1, 0), p, pre + "bn3", training)
        last = "relu3"
    if (pre + "downsample.0.weight") in p:
        idt = _bn(_conv(x, p, pre + "downsample.0", stride, 0), p, pre + "downsample.1", training)
    return _relu(o + idt, pattern, pre + last)


def forward_resnet(x, p, fe_type, nstages=3, prefix="", training=False, pattern=None):
    """Returns [l1, l2, l3(, l4)]  (reference forward_resnet contract).

    `training=True` uses batch statistics (train-mode BN) without touching the
    running stats in `p` (the oracle is functional).  `pattern` (optional) imposes ReLU
    masks ("relu", "layerL.B.reluK") and max-pool argmax indices ("maxpool_idx", as
    F.max_pool2d(return_indices=True) gives them) instead of recomputing them.
    """
    if prefix:
        p = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}
    kind, layers = ARCH[fe_type]
    x = _relu(_bn(_conv(x, p, "conv1", 2, 3), p, "bn1", training), pattern, "relu")
    if pattern is not None and "maxpool_idx" in pattern:
        idx = pattern["maxpool_idx"]
        x = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    else:
        x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li in range(nstages):
        for bi in range(layers[li]):
            stride = 2 if (li > 0 and bi == 0) else 1
            x = _block(x, p, f"layer{li + 1}.{bi}.", kind, stride, training, pattern)
        outs.append(x)
    return outs


def image_fe(x, p, fe_type, nstages=3, prefix=""):
    """reference ImageFE.forward (image_fe.py:153-174): (last_map, [maps])."""
    outs = forward_resnet(x, p, fe_type, nstages, prefix)
    return outs[-1], outs


def gmacs(fe_type, nstages, h, w):
    """Algorithmic multiply-accumulates of stem + stages for one h x w image."""
    kind, layers = ARCH[fe_type]
    exp = expansion(kind)
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    macs = ho * wo * 64 * 147
    ho, wo = (ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1
    inplanes = 64
    for li in range(nstages):
        planes = PLANES[li]
        for bi in range(layers[li]):
            stride = 2 if (li > 0 and bi == 0) else 1
            hi, wi = ho, wo
            ho, wo = (hi + 2 - 3) // stride + 1, (wi + 2 - 3) // stride + 1
            if kind == "basic":
                macs += ho * wo * planes * inplanes * 9 + ho * wo * planes * planes * 9
            else:
                macs += hi * wi * planes * inplanes
                macs += ho * wo * planes * planes * 9
                macs += ho * wo * planes * exp * planes
            if stride != 1 or inplanes != planes * exp:
                macs += ho * wo * planes * exp * inplanes
            inplanes = planes * exp
    return macs
