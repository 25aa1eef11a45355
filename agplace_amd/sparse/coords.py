"""Sparse tensor and coordinate maps (the part of MinkowskiEngine's CoordinateManager the path uses).

Reference: `ME.SparseTensor(features=..., coordinates=...)` at network_mm/mm.py:87 with coords
[N,4] = (batch, x, y, z) from `ME.utils.batched_coordinates` (datasets_ws_nuscenes.py:139-143; float
after the random rotation -> floored, as ME does for floating coordinates), strided convolutions
`kernel_size=2, stride=2` (models/minkfpn.py:53) and stride-1 convolutions of kernel 1/3/5.

Semantics restated from MinkowskiEngine's documentation (MinkowskiEngine is not installed here:
PARITY UNPINNED):
  * coordinates stay in ORIGINAL units; a tensor of stride s has coordinates that are multiples of s;
  * duplicate input coordinates are merged (features averaged);
  * stride-2 convolution: output coordinates = unique(floor(c / (2s)) * 2s); kernel offsets of an EVEN
    kernel are {0, s}, of an ODD kernel {-(k//2) .. k//2} * s per axis;
  * kernel index -> offset with the FIRST spatial axis fastest: kidx = ix + k*iy + k*k*iz.
Coordinates are linearised into sorted int64 keys; every kernel map is ONE launch of agp_sparse_kernel_map (binary
search of key + offset).  The arithmetic of the layers runs in csrc/igemm.hip (agp_sparse_conv_fwd) and csrc/sparse.hip.

Three ways to build the levels:
  * `SparseTensor.from_coords_capacity` (inference): every level has `cap` = number-of-input-points rows of which the first
    n are valid, n stays on the DEVICE (seg_off[nbatch]); sort / unique / compaction / segment offsets are kernels of
    csrc/coords.hip (agp_sparse_build, agp_sparse_coarsen: one workgroup sorts one batch sample's keys in LDS) -- no host
    synchronisation, no data-dependent allocation: the whole voxel branch is hipGraph-capturable;
  * `SparseTensor.from_coords_levels` (training, what MM uses): the same device-side kernels for ALL levels back to back, then ONE
    read-back of the row counts -> exact-size tensors (the counts size the activation tapes of the backward pass);
  * `SparseTensor.from_coords` (+ `strided()` on its result): exact-size levels from torch.unique on the keys, one host
    synchronisation per level -- the plain restatement the other two are tested against.
"""
import torch

from .. import _lib
from .._lib import check, ptr

_OFF = 1 << 15        # coordinates in [-32768, 32767] per axis
_BITS = 16
_DKEYS = {}            # (device, offsets) -> int64 key offsets on the device (built once: no per-call H2D copy)


def _dkeys(offsets, dev):
    key = (str(dev), tuple(offsets))
    t = _DKEYS.get(key)
    if t is None:
        t = torch.tensor([(dx << (2 * _BITS)) + (dy << _BITS) + dz for dx, dy, dz in offsets], dtype=torch.int64, device=dev)
        _DKEYS[key] = t
    return t


def _keys(coords):
    """int64 [n,4] (b,x,y,z) -> int64 keys, monotone in (b,x,y,z) lexicographic order."""
    c = coords.to(torch.int64)
    k = c[:, 0]
    for a in (1, 2, 3):
        k = (k << _BITS) | (c[:, a] + _OFF)
    return k


class SparseTensor:
    """Rows sorted by (batch, x, y, z).  `feats` is either the fp32 input features [n, C] (only for
    the first layer) or a 16-bit feature matrix [n + 1, C] in map storage format (hi, lo or None)
    whose last row is zero."""

    def __init__(self, coords, keys, nbatch, stride=1, f32=None, hi=None, lo=None, maps=None, n_dev=None, ws=None):
        self._coords, self.keys, self.nbatch, self.stride = coords, keys, nbatch, stride
        self.f32, self.hi, self.lo = f32, hi, lo
        self.n = keys.shape[0]                  # rows (capacity mode: the capacity; the valid count is *n_dev)
        self._maps = maps if maps is not None else {}
        self._seg = None
        self.n_dev = n_dev                      # capacity mode: int64 [1] device view of seg_off[nbatch] (valid rows), else None
        self._ws = ws                           # capacity mode: the workspace (ops.Workspace) buffers come from
        self.range_flag = None

    # ------------------------------------------------------------------ capacity mode (inference, no host sync)
    @staticmethod
    def from_coords_capacity(features, coordinates, nbatch, ws):
        """features [N, C] float, coordinates [N, 4] (batch, x, y, z), int64 / float32 / float64 (floored); nbatch must be
        given (reading it off the data would synchronise).  Every buffer comes from `ws` (ops.Workspace): steady-state calls
        with the same N allocate nothing."""
        dev = features.device
        c = coordinates.to(dev)
        if c.dtype not in (torch.int64, torch.float32, torch.float64):
            c = c.to(torch.int64 if not c.is_floating_point() else torch.float32)
        c = c.contiguous()
        kind = {torch.int64: 0, torch.float32: 1, torch.float64: 2}[c.dtype]
        n, cf = c.shape[0], features.shape[1]
        f = features.float().contiguous()
        L = _lib.load()
        keys = ws.tensor("sp.keys0", (n,), torch.int64, dev)
        f_out = ws.tensor("sp.f0", (n, cf), torch.float32, dev)
        seg_off = ws.tensor("sp.seg0", (nbatch + 1,), torch.int64, dev)
        bidx = ws.tensor("sp.bidx0", (n,), torch.int32, dev)
        flag = ws.tensor("sp.flag", (1,), torch.int32, dev, zero=True)
        nbytes = L.agp_sparse_coords_workspace_bytes(n, nbatch, cf)
        tmp = ws.tensor("sp.tmp0", (nbytes,), torch.uint8, dev)
        check(L.agp_sparse_build(ptr(c), kind, n, ptr(f), cf, nbatch, ptr(keys), ptr(f_out), ptr(seg_off), ptr(bidx), ptr(flag),
                                 ptr(tmp), nbytes, _lib.stream()), "agp_sparse_build")
        t = SparseTensor(None, keys, nbatch, 1, f32=f_out, n_dev=seg_off[nbatch:], ws=ws)
        t._seg = (seg_off, bidx)
        t.range_flag = flag
        return t

    # ------------------------------------------------------------------ training: device-built levels, ONE host synchronisation
    @staticmethod
    def from_coords_levels(features, coordinates, nbatch, nlevels):
        """The exact-size tensors the training path works on (their row counts size the activation tapes of the backward pass),
        built by the DEVICE-side coordinate manager: agp_sparse_build + `nlevels` x agp_sparse_coarsen run back to back, then
        ONE read-back fetches every level's row count and the range flag -- where `from_coords` + `strided()` synchronise per
        level (torch.unique) and once more for the range check.  Level l + 1 hangs off level l: `strided()` returns it.
        Raises like `from_coords` when a coordinate lies outside the key range."""
        dev = features.device
        c = coordinates.to(dev)
        if c.dtype not in (torch.int64, torch.float32, torch.float64):
            c = c.to(torch.int64 if not c.is_floating_point() else torch.float32)
        c = c.contiguous()
        kind = {torch.int64: 0, torch.float32: 1, torch.float64: 2}[c.dtype]
        n, cf = c.shape[0], features.shape[1]
        f = features.float().contiguous()
        L = _lib.load()
        nbytes = L.agp_sparse_coords_workspace_bytes(n, nbatch, cf)
        tmp = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)

        def level_buffers():
            return (torch.empty((n,), dtype=torch.int64, device=dev), torch.empty((nbatch + 1,), dtype=torch.int64, device=dev),
                    torch.empty((n,), dtype=torch.int32, device=dev))
        keys, seg, bidx = level_buffers()
        f_out = torch.empty((n, cf), dtype=torch.float32, device=dev)
        check(L.agp_sparse_build(ptr(c), kind, n, ptr(f), cf, nbatch, ptr(keys), ptr(f_out), ptr(seg), ptr(bidx), ptr(flag),
                                 ptr(tmp), nbytes, _lib.stream()), "agp_sparse_build")
        levels = [(keys, seg, bidx)]
        for l in range(nlevels):
            k2, s2, b2 = level_buffers()
            check(L.agp_sparse_coarsen(ptr(levels[-1][0]), ptr(levels[-1][1]), n, 1 << l, nbatch, ptr(k2), ptr(s2), ptr(b2), ptr(tmp),
                                       nbytes, _lib.stream()), "agp_sparse_coarsen")
            levels.append((k2, s2, b2))
        counts = torch.cat([s_[nbatch:] for _, s_, _ in levels] + [flag.to(torch.int64)]).tolist()      # the one synchronisation
        if counts[-1]:
            raise ValueError("voxel coordinates out of the +-32511 range (16-bit key fields, kernel offsets need headroom), a batch "
                             "index outside [0, nbatch) or more than 65536 points in one sample")
        tensors = []
        for l, (k_, s_, b_) in enumerate(levels):
            nl = int(counts[l])
            t = SparseTensor(None, k_[:nl], nbatch, 1 << l, f32=f_out[:nl] if l == 0 else None)
            t._seg = (s_, b_[:nl])
            tensors.append(t)
        for a, b in zip(tensors[:-1], tensors[1:]):
            a._maps[("coarser",)] = b
        return tensors[0]

    # ------------------------------------------------------------------ construction
    @staticmethod
    def from_coords(features, coordinates, nbatch=None):
        """features [N, C] float, coordinates [N, 4] (batch, x, y, z), int or float (floored)."""
        dev = features.device
        c = torch.floor(coordinates.to(dev).double()).to(torch.int64) if coordinates.is_floating_point() \
            else coordinates.to(dev, torch.int64)
        if c.numel() and (int(c[:, 1:].abs().max()) >= _OFF - 256):
            raise ValueError("voxel coordinates out of the +-32511 range (16-bit key fields, kernel offsets need headroom)")
        keys = _keys(c)
        ukeys, inv = torch.unique(keys, sorted=True, return_inverse=True)
        n = ukeys.shape[0]
        # merge duplicates: average their features (all ones in the reference's data)
        f = torch.zeros((n, features.shape[1]), dtype=torch.float32, device=dev)
        f.index_add_(0, inv, features.float())
        cnt = torch.zeros(n, dtype=torch.float32, device=dev).index_add_(0, inv, torch.ones_like(inv, dtype=torch.float32))
        f = f / cnt.view(-1, 1)
        first = torch.full((n,), keys.shape[0], dtype=torch.int64, device=dev)
        first = first.scatter_reduce(0, inv, torch.arange(keys.shape[0], device=dev), reduce="amin")
        uc = c[first]
        nb = int(nbatch) if nbatch is not None else (int(uc[:, 0].max()) + 1 if n else 0)
        return SparseTensor(uc, ukeys, nb, 1, f32=f.contiguous())

    @property
    def coords(self):
        """int64 [n,4] (batch, x, y, z), decoded from the keys on demand"""
        if self._coords is None:
            k = self.keys
            f = (1 << _BITS) - 1
            self._coords = torch.stack([k >> (3 * _BITS), ((k >> (2 * _BITS)) & f) - _OFF, ((k >> _BITS) & f) - _OFF,
                                        (k & f) - _OFF], 1)
        return self._coords

    def with_feats(self, hi, lo=None):
        """Same coordinates (and cached maps), new feature matrix."""
        t = SparseTensor(self._coords, self.keys, self.nbatch, self.stride, hi=hi, lo=lo, maps=self._maps, n_dev=self.n_dev,
                         ws=self._ws)
        t._seg = self._seg
        t.range_flag = self.range_flag
        return t

    # ------------------------------------------------------------------ segments
    def segments(self):
        """(seg_off int64 [B+1], bidx int32 [n])"""
        if self._seg is None:
            b = (self.keys >> (3 * _BITS)).contiguous()
            bounds = torch.arange(self.nbatch + 1, device=b.device, dtype=torch.int64)
            seg_off = torch.searchsorted(b, bounds).to(torch.int64).contiguous()
            self._seg = (seg_off, b.to(torch.int32).contiguous())
        return self._seg

    # ------------------------------------------------------------------ kernel maps
    def _map(self, out_keys, offsets, tag="o"):
        """int32 [len(offsets), n_out]: row of (out coordinate + offset) in this tensor, n when absent -- arbitrary offset lists
        (the layers' regular grids go through _map_grid)"""
        dev = self.keys.device
        dk = _dkeys(offsets, dev)
        n_out = out_keys.shape[0]
        if self._ws is not None:
            nbr = self._ws.tensor(f"sp.map.{self.stride}.{len(offsets)}.{tag}", (len(offsets), n_out), torch.int32, dev)
        else:
            nbr = torch.empty((len(offsets), n_out), dtype=torch.int32, device=dev)
        if n_out:
            check(_lib.load().agp_sparse_kernel_map(ptr(self.keys), self.n, ptr(out_keys), n_out, ptr(dk), len(offsets),
                                                    ptr(nbr), None, _lib.stream()), "agp_sparse_kernel_map")
        return nbr

    def _map_grid(self, out_keys, ksize, centered, st, out_n_dev):
        """The table of a regular offset grid (agp_sparse_kernel_map_grid: one search per (dx, dy) column)."""
        dev = self.keys.device
        n_out = out_keys.shape[0]
        shape = (ksize ** 3, n_out)
        if self._ws is not None:
            nbr = self._ws.tensor(f"sp.map.{self.stride}.{ksize}.{centered}", shape, torch.int32, dev)
        else:
            nbr = torch.empty(shape, dtype=torch.int32, device=dev)
        if n_out:
            check(_lib.load().agp_sparse_kernel_map_grid(ptr(self.keys), self.n, ptr(out_keys), n_out, ksize, centered, st, ptr(nbr),
                                                         ptr(out_n_dev), ptr(self.n_dev), ptr(self.segments()[0]), _lib.stream()),
                  "agp_sparse_kernel_map_grid")
        return nbr

    def zperm(self):
        """int32 [n]: the row order the gather-GEMM of a centred convolution on THIS level works in -- every batch sample's rows
        grouped by z-plane (agp_sparse_zplane_perm), so that a tile's rows share their absent dz taps and the kernel skips them.
        Cached with the level's maps; one launch."""
        key = ("zperm",)
        m = self._maps.get(key)
        if m is None:
            dev = self.keys.device
            if self._ws is not None:
                m = self._ws.tensor(f"sp.zperm.{self.stride}", (self.n,), torch.int32, dev)
            else:
                m = torch.empty((self.n,), dtype=torch.int32, device=dev)
            if self.n:
                check(_lib.load().agp_sparse_zplane_perm(ptr(self.keys), ptr(self.segments()[0]), self.nbatch, self.n, ptr(m),
                                                         _lib.stream()), "agp_sparse_zplane_perm")
            self._maps[key] = m
        return m

    def tile_taps(self, ksize):
        """uint32 [ceil(n / 256) * 2]: per 128 GEMM rows of a centred `ksize` convolution on this level (rows in zperm() order), the
        taps that occur among them (agp_sparse_tile_taps on kernel_map(ksize)); None for more than 32 taps.  Cached with the maps."""
        if ksize ** 3 > 32 or ksize == 1:
            return None
        key = ("taps", ksize)
        m = self._maps.get(key)
        if m is None:
            dev = self.keys.device
            ngran = (self.n + 255) // 256 * 2
            if self._ws is not None:
                m = self._ws.tensor(f"sp.taps.{self.stride}.{ksize}", (ngran,), torch.int32, dev)
            else:
                m = torch.empty((ngran,), dtype=torch.int32, device=dev)
            if self.n:
                check(_lib.load().agp_sparse_tile_taps(ptr(self.kernel_map(ksize)), self.n, ksize ** 3, self.n, ptr(self.zperm()),
                                                       ptr(self.n_dev), ptr(m), ngran, _lib.stream()), "agp_sparse_tile_taps")
            self._maps[key] = m
        return m

    def kernel_map(self, ksize):
        """stride-1 convolution of odd kernel `ksize`: int32 [ksize^3, n] neighbour rows."""
        key = ("s1", ksize)
        m = self._maps.get(key)
        if m is None:
            if ksize == 1:
                m = torch.arange(self.n, dtype=torch.int32, device=self.keys.device).view(1, -1)
            else:
                # offsets ((ix - r) * st, (iy - r) * st, (iz - r) * st), kidx = ix + k*iy + k*k*iz
                m = self._map_grid(self.keys, ksize, 1, self.stride, self.n_dev)
            self._maps[key] = m
        return m

    def up_map(self, coarse):
        """Transposed kernel 2 / stride 2 from `coarse` (the coordinates of self.strided()[0]) onto this tensor's rows
        (ME.MinkowskiConvolutionTranspose onto the existing finer coordinate map, models/minkfpn.py:62,116): int32 [8, n],
        entry [t][j] = row of j's parent in `coarse` if t is j's child position (kidx = ix + 2*iy + 4*iz), else coarse.n.
        One launch of agp_sparse_kernel_map: the key of row j minus the offset of tap t is a key of `coarse` exactly when t is
        j's child position (coarse coordinates are multiples of 2 * stride)."""
        key = ("up",)
        m = self._maps.get(key)
        if m is None:
            if coarse.stride != 2 * self.stride:
                raise ValueError("up_map: `coarse` must be the next coarser level")
            st = self.stride
            offsets = [(-ix * st, -iy * st, -iz * st) for iz in (0, 1) for iy in (0, 1) for ix in (0, 1)]
            m = coarse._map(self.keys, offsets, tag="up")
            self._maps[key] = m
        return m

    def strided(self):
        """kernel 2 / stride 2: (coarser SparseTensor without features, int32 [8, n_out] table)."""
        key = ("s2",)
        got = self._maps.get(key)
        if got is None:
            s2, st = self.stride * 2, self.stride
            if self.n_dev is not None:
                # capacity mode: the coarser level on the device (csrc/coords.hip), same capacity, no host synchronisation
                ws, dev, L = self._ws, self.keys.device, _lib.load()
                okeys = ws.tensor(f"sp.keys{s2}", (self.n,), torch.int64, dev)
                seg_off = ws.tensor(f"sp.seg{s2}", (self.nbatch + 1,), torch.int64, dev)
                bidx = ws.tensor(f"sp.bidx{s2}", (self.n,), torch.int32, dev)
                nbytes = L.agp_sparse_coords_workspace_bytes(self.n, self.nbatch, 0)
                tmp = ws.tensor("sp.tmp", (nbytes,), torch.uint8, dev)
                check(L.agp_sparse_coarsen(ptr(self.keys), ptr(self.segments()[0]), self.n, st, self.nbatch, ptr(okeys), ptr(seg_off),
                                           ptr(bidx), ptr(tmp), nbytes, _lib.stream()), "agp_sparse_coarsen")
                out = SparseTensor(None, okeys, self.nbatch, s2, n_dev=seg_off[self.nbatch:], ws=ws)
                out._seg = (seg_off, bidx)
                out.range_flag = self.range_flag
            else:
                # floor(c / s2) * s2 per axis = clearing the low bits of every (2^15-biased) 16-bit field
                low = s2 - 1
                mask = ~((low << (2 * _BITS)) | (low << _BITS) | low)
                out = self._maps.get(("coarser",))             # from_coords_levels built it on the device
                if out is None:
                    out = SparseTensor(None, torch.unique(self.keys & mask, sorted=True), self.nbatch, s2)
                okeys = out.keys
            # children of an output coordinate: offsets (ix * st, iy * st, iz * st), kidx = ix + 2*iy + 4*iz
            got = (out, self._map_grid(okeys, 2, 0, st, out.n_dev))
            self._maps[key] = got
        return got
