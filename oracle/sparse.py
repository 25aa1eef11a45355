"""CPU restatement of the sparse-voxel branch (TEST INFRASTRUCTURE ONLY).

Reference: models/minkfpn.py:19-123 (MinkFPN, bottom-up and top-down passes), layers/eca_block.py:14-79
(ECALayer, ECABasicBlock on MinkowskiEngine's BasicBlock), layers/pooling.py:70-87 (MinkGeM),
network_mm/mm.py:86-92, fuse_block_toshallow.py:83, stage2fuse_blockadd.py:26-32,196-207.

MinkowskiEngine (git HEAD, README.md:29) is a third-party dependency that is not installed here ->
PARITY UNPINNED.  Its generalized sparse convolution is restated from its documentation:
out[u] = sum_{i in N(u)} W_i x[u + i] over the offsets i of the kernel for which u + i is an
occupied input site; coordinates stay in original units (a stride-s tensor lives on multiples of s);
stride-2 output sites = unique(floor(c / 2s) * 2s); odd kernels are centred, even kernels span
{0, s}; kernel index -> offset with the first spatial axis fastest.  Implemented with python
dictionaries (small clouds only); tests/test_oracle_kat.py cross-checks it against dense F.conv3d.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


class SpT:
    """coords: list of (b,x,y,z) tuples sorted lexicographically; feats [n, C]; stride."""

    def __init__(self, coords, feats, stride=1, nbatch=None):
        self.coords, self.feats, self.stride = coords, feats, stride
        self.nbatch = nbatch if nbatch is not None else (max(c[0] for c in coords) + 1 if coords else 0)
        self.index = {c: i for i, c in enumerate(coords)}


def from_coords(features, coordinates, nbatch=None):
    """ME.SparseTensor(features, coordinates): floor float coordinates, merge duplicates (average)."""
    c = torch.floor(coordinates.double()).long() if coordinates.is_floating_point() else coordinates.long()
    groups = {}
    for i, row in enumerate(c.tolist()):
        groups.setdefault(tuple(row), []).append(i)
    coords = sorted(groups)
    feats = torch.stack([features[groups[k]].mean(0) for k in coords], 0)
    return SpT(coords, feats, 1, nbatch)


def _offsets(ksize, stride):
    if ksize % 2:
        r = ksize // 2
        rng = [(-r + i) * stride for i in range(ksize)]
    else:
        rng = [i * stride for i in range(ksize)]
    return [(dx, dy, dz) for dz in rng for dy in rng for dx in rng]          # x fastest


def conv(x, kernel, ksize, stride=1):
    """kernel [K, Cin, Cout] (or [Cin, Cout] for ksize 1)."""
    kernel = kernel.reshape(-1, kernel.shape[-2], kernel.shape[-1])
    if stride == 1:
        out_coords, out_stride = x.coords, x.stride
    else:
        s2 = x.stride * stride
        out_coords = sorted({(b, (cx // s2) * s2, (cy // s2) * s2, (cz // s2) * s2) for b, cx, cy, cz in x.coords})
        out_stride = s2
    offs = _offsets(ksize, x.stride)
    # gather formulation (differentiable): per tap, the input row of every output row (or a zero row)
    n_in = len(x.coords)
    fz = torch.cat([x.feats, torch.zeros((1, x.feats.shape[1]), dtype=x.feats.dtype)], 0)
    out = torch.zeros((len(out_coords), kernel.shape[-1]), dtype=x.feats.dtype)
    for k, (dx, dy, dz) in enumerate(offs):
        idx = torch.tensor([x.index.get((b, cx + dx, cy + dy, cz + dz), n_in) for b, cx, cy, cz in out_coords],
                           dtype=torch.long)
        if bool((idx < n_in).any()):
            out = out + fz[idx] @ kernel[k]
    return SpT(out_coords, out, out_stride, x.nbatch)


def conv_transpose(x, kernel, fine):
    """ME.MinkowskiConvolutionTranspose(kernel_size=2, stride=2) (models/minkfpn.py:62-63) onto the EXISTING coordinates of the
    next finer level `fine` (MinkowskiEngine's coordinate manager hands a transposed convolution the coordinate map its target
    tensor stride already has -- the one the bottom-up pass created -- which is what lets minkfpn.py:117 add the lateral
    feature map to it).  out[v] = W_i x[u] for the one coarse site u = floor(v / 2s) * 2s and offset i = v - u in {0, s}^3
    (s = fine.stride), kernel index i_x + 2 i_y + 4 i_z as for the strided convolution.  kernel [8, Cin, Cout]."""
    assert x.stride == 2 * fine.stride
    s2, st = x.stride, fine.stride
    n_in = len(x.coords)
    fz = torch.cat([x.feats, torch.zeros((1, x.feats.shape[1]), dtype=x.feats.dtype)], 0)
    out = torch.zeros((len(fine.coords), kernel.shape[-1]), dtype=x.feats.dtype)
    par, tap = [], []
    for b, cx, cy, cz in fine.coords:
        u = (b, (cx // s2) * s2, (cy // s2) * s2, (cz // s2) * s2)
        par.append(x.index.get(u, n_in))
        tap.append((cx - u[1]) // st + 2 * ((cy - u[2]) // st) + 4 * ((cz - u[3]) // st))
    par, tap = torch.tensor(par, dtype=torch.long), torch.tensor(tap, dtype=torch.long)
    for k in range(8):
        idx = torch.where(tap == k, par, torch.full_like(par, n_in))
        if bool((idx < n_in).any()):
            out = out + fz[idx] @ kernel[k]
    return SpT(fine.coords, out, st, x.nbatch)


def bn(x, p, name, training=False):
    """MinkowskiBatchNorm = BatchNorm1d over the rows of the feature matrix; training=True uses the
    batch statistics (biased variance), running stats untouched (functional oracle)."""
    if training:
        mean = x.feats.mean(0)
        var = x.feats.var(0, unbiased=False)
    else:
        mean, var = p[name + ".bn.running_mean"], p[name + ".bn.running_var"]
    f = (x.feats - mean) / torch.sqrt(var + BN_EPS) * p[name + ".bn.weight"] + p[name + ".bn.bias"]
    return SpT(x.coords, f, x.stride, x.nbatch)


def relu(x, pattern=None, key=None):
    """ReLU; `pattern[key]` (0/1 mask, test infrastructure) imposes the activation pattern instead."""
    if pattern is not None and key in pattern:
        return SpT(x.coords, x.feats * pattern[key].to(x.feats.dtype), x.stride, x.nbatch)
    return SpT(x.coords, torch.relu(x.feats), x.stride, x.nbatch)


def _batch_ids(x):
    return torch.tensor([c[0] for c in x.coords], dtype=torch.long)


def global_avg(x):
    b = _batch_ids(x)
    out = torch.zeros((x.nbatch, x.feats.shape[1]), dtype=x.feats.dtype)
    for i in range(x.nbatch):
        if (b == i).any():
            out[i] = x.feats[b == i].mean(0)
    return out


def mink_gem(x, p, eps=1e-6):
    t = SpT(x.coords, x.feats.clamp(min=eps).pow(p), x.stride, x.nbatch)
    return global_avg(t).pow(1.0 / p)


def eca(x, w):
    """eca_block.py:25-43: Conv1d over the channel axis of the per-sample mean, sigmoid, broadcast mul."""
    y = global_avg(x)
    k = w.numel()
    y = F.conv1d(y.unsqueeze(1), w.view(1, 1, k), padding=(k - 1) // 2).squeeze(1)
    s = torch.sigmoid(y)
    return SpT(x.coords, x.feats * s[_batch_ids(x)], x.stride, x.nbatch)


def eca_basic_block(x, p, pre, training=False, pattern=None):
    """eca_block.py:62-79"""
    out = relu(bn(conv(x, p[pre + "conv1.kernel"], 3), p, pre + "norm1", training), pattern, pre + "relu1")
    out = bn(conv(out, p[pre + "conv2.kernel"], 3), p, pre + "norm2", training)
    out = eca(out, p[pre + "eca.conv.weight"])
    residual = x
    if (pre + "downsample.0.kernel") in p:
        residual = bn(conv(x, p[pre + "downsample.0.kernel"], 1), p, pre + "downsample.1", training)
    return relu(SpT(out.coords, out.feats + residual.feats, out.stride, out.nbatch), pattern, pre + "relu2")


def minkfpn(x, p, pre, nlevels=3, training=False, pattern=None, num_top_down=0):
    """minkfpn.py:88-123 -> (x, out_maps).  num_top_down < nlevels (the reference's own forward indexes out_maps[-2 - ndx]
    out of range when the two are equal)."""
    out_maps, feature_maps = [], []
    x = relu(bn(conv(x, p[pre + "conv0.kernel"], 5), p, pre + "bn0", training), pattern, pre + "relu0")
    for i in range(nlevels):
        x = relu(bn(conv(x, p[f"{pre}convs.{i}.kernel"], 2, stride=2), p, f"{pre}bns.{i}", training), pattern,
                 f"{pre}relus.{i}")
        x = eca_basic_block(x, p, f"{pre}blocks.{i}.0.", training, pattern)
        if nlevels - 1 - num_top_down <= i < nlevels - 1:
            feature_maps.append(x)
        out_maps.append(x)
    x = conv(x, p[pre + "conv1x1s.0.kernel"], 1)
    out_maps[-1] = x
    for ndx in range(num_top_down):
        fm = feature_maps[-ndx - 1]
        lat = conv(fm, p[f"{pre}conv1x1s.{ndx + 1}.kernel"], 1)
        up = conv_transpose(x, p[f"{pre}tconvs.{ndx}.kernel"], fm)
        x = SpT(up.coords, up.feats + lat.feats, up.stride, up.nbatch)
        out_maps[-2 - ndx] = x
    return x, out_maps


def broadcast_add(x, vec):
    return SpT(x.coords, x.feats + vec[_batch_ids(x)], x.stride, x.nbatch)


# ------------------------------------------------------------------ parameters
def init_vox_params(planes=(64, 128, 256), seed=0, dtype=torch.float32, prefix="vox_fe.", extra_blocks=(), num_top_down=0):
    """MinkFPN parameters under `prefix` (+ one ECABasicBlock(c, c) per (prefix, c) in extra_blocks); the lateral 1x1 and
    transposed convolutions as minkfpn.py:58-73 registers them (num_top_down + 1 laterals, num_top_down transposed)."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def kern(name, vol, cin, cout):
        shape = (cin, cout) if vol == 1 else (vol, cin, cout)
        p[name] = (torch.randn(shape, generator=g) * (2.0 / (cout * vol)) ** 0.5).to(dtype)

    def bnp(name, c):
        p[name + ".bn.weight"] = (0.5 + torch.rand(c, generator=g)).to(dtype)
        p[name + ".bn.bias"] = (0.2 * torch.randn(c, generator=g)).to(dtype)
        p[name + ".bn.running_mean"] = (0.3 * torch.randn(c, generator=g)).to(dtype)
        p[name + ".bn.running_var"] = (0.5 + 1.5 * torch.rand(c, generator=g)).to(dtype)
        p[name + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def block(pre, cin, c):
        kern(pre + "conv1.kernel", 27, cin, c); bnp(pre + "norm1", c)
        kern(pre + "conv2.kernel", 27, c, c); bnp(pre + "norm2", c)
        t = int(abs((np.log2(c) + 1) / 2))
        k = t if t % 2 else t + 1
        p[pre + "eca.conv.weight"] = (torch.randn(1, 1, k, generator=g) * 0.5).to(dtype)
        if cin != c:
            kern(pre + "downsample.0.kernel", 1, cin, c); bnp(pre + "downsample.1", c)

    kern(prefix + "conv0.kernel", 125, 1, planes[0]); bnp(prefix + "bn0", planes[0])
    inpl = planes[0]
    for i, pl in enumerate(planes):
        kern(f"{prefix}convs.{i}.kernel", 8, inpl, inpl); bnp(f"{prefix}bns.{i}", inpl)
        block(f"{prefix}blocks.{i}.0.", inpl, pl)
        inpl = pl
    lateral = planes[-1]
    for i in range(num_top_down):
        kern(f"{prefix}conv1x1s.{i}.kernel", 1, planes[-1 - i], lateral)
        kern(f"{prefix}tconvs.{i}.kernel", 8, lateral, lateral)
    last = planes[-1 - num_top_down] if num_top_down < len(planes) else planes[0]
    kern(f"{prefix}conv1x1s.{num_top_down}.kernel", 1, last, lateral)
    for pre, c in extra_blocks:
        block(pre, c, c)
    return p


from bench_inputs import synth_cloud  # noqa: E402,F401  (shared synthetic-input generator)
