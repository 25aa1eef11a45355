"""Training-mode execution of the sparse-voxel branch (batch-statistics MinkowskiBatchNorm, sparse
convolution data / weight gradients, ECA and pooling backward) on split-bf16 feature matrices.

The reference trains this branch by autograd through MinkowskiEngine (train.py:337-341).  Here:
    train-mode BatchNorm   = agp_bn_stats / agp_map_affine / agp_bn_bwd on the matrix seen as a 1 x n map
    data gradient          = agp_sparse_conv_fwd on the transposed kernel map with flipped weights
    weight gradient        = agp_sparse_conv_wgrad (LDS transpose reads) / agp_sparse_conv_cin1_wgrad
    ECA / pooling backward = agp_seg_dot_fwd, agp_eca_scale_bwd, agp_seg_affine_fwd, agp_seg_pool_bwd
Parameter gradients are accumulated into `.grad` (train_graph._acc_grad).
"""
import torch

from .. import _lib, ops, train_graph
from .._lib import check, ptr
from .coords import SparseTensor
from . import modules
from .modules import _alloc_feats, global_avg_pool, seg_affine

PREC = 3


def _L():
    return _lib.load()


def as_map(sp: SparseTensor):
    """the [n, C] rows of a feature matrix as a 1 x n halo-free map (for the BatchNorm / mask kernels)"""
    c = sp.hi.shape[1]
    return ops.SplitMap(sp.hi[:sp.n], sp.lo[:sp.n], 1, 1, sp.n, c, 0)


def _new_like(sp: SparseTensor, c=None):
    hi, lo = _alloc_feats(sp.n, c or sp.hi.shape[1], PREC, sp.hi.device)
    return sp.with_feats(hi, lo)


class SparseConvUnit:
    """MinkowskiConvolution (-> MinkowskiBatchNorm(train) -> ReLU?) with a hand-written backward."""

    def __init__(self, conv, bn=None):
        self.conv, self.bn = conv, bn
        self.saved = None

    def forward(self, x: SparseTensor, relu=False):
        conv = self.conv
        z = conv(x, None, relu=False, prec=PREC)                    # raw convolution
        if conv.stride == 2:
            _, nbr = x.strided()
        else:
            nbr = x.kernel_map(conv.kernel_size)
        if self.bn is None:
            self.saved = (x, nbr, z, None, None, None, False, False, None)
            return z
        zm = as_map(z)
        mean, rstd, scale, shift = train_graph.bn_stats(zm, self.bn.bn)
        y = _new_like(z)
        train_graph.map_affine(zm, scale, shift, as_map(y), relu=relu)
        self.saved = (x, nbr, z, y, mean, rstd, relu, not self.bn.bn.training, self.bn.bn.__dict__.pop("_agp_sync_count", None))
        return y

    def backward(self, gy: SparseTensor, need_gx=True):
        x, nbr, z, y, mean, rstd, relu, frozen, sync_count = self.saved
        conv = self.conv
        dev = z.hi.device
        if self.bn is not None:
            gz = _new_like(z)
            gg, gb = train_graph.bn_bwd(as_map(z), as_map(gy), as_map(y) if relu else None, mean, rstd, self.bn.bn.weight, relu,
                                        as_map(gz), frozen=frozen, sync_count=sync_count)
            train_graph._acc_grad(self.bn.bn.weight, gg)
            train_graph._acc_grad(self.bn.bn.bias, gb)
        else:
            gz = gy
        L = _L()
        cin, cout, ntaps = conv.in_channels, conv.out_channels, nbr.shape[0]
        # ---- weight gradient
        if cin == 1:
            gw = torch.empty((ntaps, cout), dtype=torch.float32, device=dev)
            nsl = max(1, min(64, z.n // 2048))            # row slices: ~2000 blocks for conv0's 125 taps, partials added in slice order
            part = torch.empty((nsl, ntaps, cout), dtype=torch.float32, device=dev)
            check(L.agp_sparse_conv_cin1_wgrad(ptr(x.f32), x.n, ptr(nbr), z.n, ntaps, ptr(gz.hi), ptr(gz.lo), cout, ptr(gw),
                                               ptr(part), nsl, _lib.stream()), "agp_sparse_conv_cin1_wgrad")
        else:
            gw = torch.empty((ntaps, cin, cout), dtype=torch.float32, device=dev)
            nbytes = L.agp_sparse_conv_wgrad_workspace_bytes(z.n, cin, cout, ntaps)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            check(L.agp_sparse_conv_wgrad(ptr(x.hi), ptr(x.lo), x.n + 1, ptr(nbr), z.n, cin, cout, ntaps, ptr(gz.hi), ptr(gz.lo),
                                          ptr(gw), ptr(ws), nbytes, _lib.stream()), "agp_sparse_conv_wgrad")
        train_graph._acc_grad(conv.kernel, gw.view(conv.kernel.shape))
        if not need_gx or cin == 1:
            return None
        # ---- data gradient: the same gather-GEMM on the transposed kernel map
        k = conv.kernel.detach().float().view(ntaps, cin, cout)
        if conv.stride == 2:
            # gx[j] = gz[parent(j)] W[childpos(j)]^T : one valid tap per input row -- the table is the finer level's up map (row of
            # j's parent at j's child position, z.n elsewhere): one launch, where inverting `nbr` tap by tap with boolean masks
            # cost 8 host synchronisations per strided convolution
            tab, wd = x.up_map(z), k.permute(1, 0, 2).contiguous()                 # [cin][tap][cout]
        else:
            # centred odd kernel: the row that sees j through tap t is j's neighbour through the mirrored tap
            tab, wd = nbr, k.flip(0).permute(1, 0, 2).contiguous()
        w_hi, w_lo = ops.split_weight(wd, _lib.FMT_BF16)
        gx = _new_like(x, cin)
        grouped = modules.GROUP_ROWS and conv.stride == 1 and 1 < ntaps <= 32     # (the forward's row order and tap lists: the same kernel map)
        check(L.agp_sparse_conv_fwd(ptr(gz.hi), ptr(gz.lo), z.n + 1, ptr(tab), x.n, cout, cin, ntaps, ptr(w_hi), ptr(w_lo),
                                    None, None, None, None, 0, ptr(gx.hi), ptr(gx.lo), PREC, None,
                                    ptr(x.zperm()) if grouped else None, ptr(x.tile_taps(conv.kernel_size)) if grouped else None,
                                    _lib.stream()),
              "agp_sparse_conv_fwd")
        return gx


class SparseTConvUnit:
    """MinkowskiConvolutionTranspose (kernel 2 / stride 2 onto the finer level, + the lateral tensor) with a hand-written
    backward: the weight gradient is the sparse wgrad on the one-hot up map, the data gradient the gather-GEMM on the
    strided convolution's own table (a coarse row collects its up to 8 children)."""

    def __init__(self, tconv):
        self.conv = tconv
        self.saved = None

    def forward(self, x: SparseTensor, fine: SparseTensor, residual: SparseTensor = None):
        y = self.conv(x, fine, residual=residual, prec=PREC)
        self.saved = (x, fine)
        return y

    def backward(self, g: SparseTensor):
        """g: gradient w.r.t. the output (fine rows; also the residual's gradient) -> gradient w.r.t. x (coarse rows)"""
        x, fine = self.saved
        conv, dev, L = self.conv, g.hi.device, _L()
        cin, cout = conv.in_channels, conv.out_channels
        up = fine.up_map(x)
        gw = torch.empty((8, cin, cout), dtype=torch.float32, device=dev)
        nbytes = L.agp_sparse_conv_wgrad_workspace_bytes(fine.n, cin, cout, 8)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        check(L.agp_sparse_conv_wgrad(ptr(x.hi), ptr(x.lo), x.n + 1, ptr(up), fine.n, cin, cout, 8, ptr(g.hi), ptr(g.lo), ptr(gw),
                                      ptr(ws), nbytes, _lib.stream()), "agp_sparse_conv_wgrad")
        train_graph._acc_grad(conv.kernel, gw)
        _, down = fine.strided()                                                  # [8][n_coarse]: child rows (fine.n = none)
        wd = conv.kernel.detach().float().permute(1, 0, 2).contiguous()           # gx[u] = sum_t g[child_t(u)] W[t]^T: [cin][tap][cout]
        w_hi, w_lo = ops.split_weight(wd, _lib.FMT_BF16)
        gx = _new_like(x, cin)
        check(L.agp_sparse_conv_fwd(ptr(g.hi), ptr(g.lo), fine.n + 1, ptr(down), x.n, cout, cin, 8, ptr(w_hi), ptr(w_lo),
                                    None, None, None, None, 0, ptr(gx.hi), ptr(gx.lo), PREC, None, None, None, _lib.stream()),
              "agp_sparse_conv_fwd")
        return gx


def _masked(g: SparseTensor, y: SparseTensor):
    """g * [y > 0]"""
    out = _new_like(g)
    train_graph.map_add(as_map(g), None, as_map(out), mask=as_map(y))
    return out


def _add(a: SparseTensor, b: SparseTensor):
    out = _new_like(a)
    train_graph.map_add(as_map(a), as_map(b), as_map(out))
    return out


class ECABlockTrain:
    """ECABasicBlock (layers/eca_block.py:46-79) forward/backward in train mode."""

    def __init__(self, blk):
        self.blk = blk
        self.u1 = SparseConvUnit(blk.conv1, blk.norm1)
        self.u2 = SparseConvUnit(blk.conv2, blk.norm2)
        self.ud = SparseConvUnit(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
        self.saved = None

    def forward(self, x: SparseTensor):
        y1 = self.u1.forward(x, relu=True)
        y2 = self.u2.forward(y1, relu=False)
        mean = global_avg_pool(y2)
        eca = self.blk.eca
        s = torch.empty_like(mean)
        w = eca.conv.weight.detach().float().contiguous().view(-1)
        check(_L().agp_eca_scale_fwd(ptr(mean), mean.shape[0], mean.shape[1], ptr(w), eca.k_size, ptr(s), _lib.stream()),
              "agp_eca_scale_fwd")
        res = self.ud.forward(x, relu=False) if self.ud is not None else x
        out = seg_affine(y2, scale=s, residual=res, relu=True)
        self.saved = (x, y2, mean, s, w, out)
        return out

    def backward(self, gout: SparseTensor):
        x, y2, mean, s, w, out = self.saved
        eca = self.blk.eca
        L = _L()
        g = _masked(gout, out)                                   # through the final ReLU; also the residual's gradient
        seg_off, _ = y2.segments()
        c = mean.shape[1]
        gs = torch.empty_like(mean)                              # dL/dscale[b][c] = sum_i g[i][c] * y2[i][c]
        check(L.agp_seg_dot_fwd(ptr(g.hi), ptr(g.lo), ptr(y2.hi), ptr(y2.lo), ptr(seg_off), y2.nbatch, c, ptr(gs), _lib.stream()),
              "agp_seg_dot_fwd")
        add = torch.empty_like(mean)
        gw = torch.empty(eca.k_size, dtype=torch.float32, device=mean.device)
        check(L.agp_eca_scale_bwd(ptr(mean), ptr(s), ptr(gs), ptr(seg_off), y2.nbatch, c, ptr(w), eca.k_size, ptr(add), ptr(gw),
                                  _lib.stream()), "agp_eca_scale_bwd")
        train_graph._acc_grad(eca.conv.weight, gw)
        gy2 = seg_affine(g, scale=s, add=add)                    # g * s[b] + dL/dmean[b] / n_b
        gy1 = self.u2.backward(gy2)
        gx = self.u1.backward(gy1)
        if self.ud is not None:
            gx2 = self.ud.backward(g)
            return _add(gx, gx2)
        return _add(gx, g)


class MinkFPNTrain:
    """MinkFPN (models/minkfpn.py:88-123) forward/backward in train mode: bottom-up, lateral and top-down passes."""

    def __init__(self, net):
        self.net = net
        self.u0 = SparseConvUnit(net.conv0, net.bn0)
        self.down = [SparseConvUnit(c, b) for c, b in zip(net.convs, net.bns)]
        self.blocks = [[ECABlockTrain(b) for b in seq] for seq in net.blocks]
        self.lats = [SparseConvUnit(c, None) for c in net.conv1x1s]
        self.lat = self.lats[0]
        self.tconvs = [SparseTConvUnit(t) for t in net.tconvs]

    def forward(self, x: SparseTensor):
        net = self.net
        nbu, ntd = net.num_bottom_up, net.num_top_down
        if ntd >= nbu:
            raise IndexError("MinkFPN: num_top_down == number of levels indexes out_maps out of range (models/minkfpn.py:118)")
        out_maps, fms = [], []
        x = self.u0.forward(x, relu=True)
        for i, (d, blks) in enumerate(zip(self.down, self.blocks)):
            x = d.forward(x, relu=True)
            for b in blks:
                x = b.forward(x)
            if nbu - 1 - ntd <= i < nbu - 1:
                fms.append(x)
            out_maps.append(x)
        top = self.lat.forward(x)
        out_maps[-1] = top
        x = top
        for k, t in enumerate(self.tconvs):
            fm = fms[-k - 1]
            x = t.forward(x, fm, residual=self.lats[k + 1].forward(fm))
            out_maps[-2 - k] = x
        return x, out_maps

    def backward(self, gmaps):
        """gmaps[i]: gradient w.r.t. out_maps[i] (SparseTensor or None).  out_maps[n-1-k] (k = 0 .. num_top_down) are the
        top-down tensors x_k, the others the block outputs."""
        n, T = len(self.down), len(self.tconvs)

        def acc(a, b):
            return b if a is None else (a if b is None else _add(a, b))
        # ---- top-down pass backwards: x_{k+1} = tconv_k(x_k) + lat_{k+1}(block output n-2-k)
        gx = [gmaps[n - 1 - k] for k in range(T + 1)]
        gfm = {}                                                  # level -> gradient w.r.t. its block output from the lateral
        for k in range(T - 1, -1, -1):
            g = gx[k + 1]
            if g is None:
                continue
            gfm[n - 2 - k] = self.lats[k + 1].backward(g)
            gx[k] = acc(gx[k], self.tconvs[k].backward(g))
        g = self.lat.backward(gx[0]) if gx[0] is not None else None
        # ---- bottom-up pass backwards
        for i in range(n - 1, -1, -1):
            if i < n - 1:
                g = acc(g, gfm.get(i) if i >= n - 1 - T else gmaps[i])
            if g is None:
                continue
            for b in reversed(self.blocks[i]):
                g = b.backward(g)
            g = self.down[i].backward(g)
        if g is not None:
            self.u0.backward(g, need_gx=False)


def seg_pool_bwd(x: SparseTensor, gmean=None, ggem=None, gem_y=None, p=None, eps=1e-6, base: SparseTensor = None, gp=None):
    """gradient of the per-sample mean / GeM w.r.t. the rows (+ base)"""
    seg_off, bidx = x.segments()
    out = _new_like(x)
    c = x.hi.shape[1]
    check(_L().agp_seg_pool_bwd(ptr(x.hi), ptr(x.lo), ptr(bidx), ptr(seg_off), ptr(gmean), ptr(ggem), ptr(gem_y), ptr(p), eps,
                                ptr(base.hi) if base is not None else None, ptr(base.lo) if base is not None else None, x.n, c,
                                ptr(out.hi), ptr(out.lo), ptr(gp), _lib.stream()), "agp_seg_pool_bwd")
    return out


def seg_sum(g: SparseTensor):
    """[B, C] fp32: sum of the rows of every sample (gradient of a broadcast addition)"""
    seg_off, _ = g.segments()
    c = g.hi.shape[1]
    out = torch.empty((g.nbatch, c), dtype=torch.float32, device=g.hi.device)
    check(_L().agp_seg_dot_fwd(ptr(g.hi), ptr(g.lo), None, None, ptr(seg_off), g.nbatch, c, ptr(out), _lib.stream()),
          "agp_seg_dot_fwd")
    return out
