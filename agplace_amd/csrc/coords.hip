// Coordinate manager of the sparse-voxel branch on the device, WITHOUT host synchronisation (the part of
// MinkowskiEngine's CoordinateManager the path uses: reference network_mm/mm.py:86-87 `ME.SparseTensor(features,
// coordinates)`, models/minkfpn.py:53 stride-2 convolutions; semantics restated in agplace_amd/sparse/coords.py).
//
// Capacity mode: a level of `cap` rows (cap = the number of INPUT points, an upper bound for every level) holds
//   keys[cap]      sorted unique int64 keys (batch, x, y, z: 16-bit biased fields), rows >= n padded with SENT
//   seg_off[B+1]   first row of every batch sample; seg_off[B] = n, the number of valid rows -- the device-side count every
//                  row-wise kernel of the branch reads instead of a host integer (no .item(), no torch.unique)
//   bidx[cap]      batch index of a row
//
// Round 4: the batch index is the top key field, so a level is B independent sorts of a few thousand keys.  ONE workgroup
// sorts one sample's keys in LDS (bitonic network over <= 16384 eight-byte keys = 128 KB of the CU's 160 KB), removes the
// duplicates and counts; a second kernel places the samples behind each other.  A coarser level = 2 launches, the input level
// = 5 (+ 2 memsets) -- rounds 2-3 ran rocPRIM's device-wide radix sort per level: ~27 launches of ~5 us each, 130 launches
// per forward, most of them 1024-thread grids that queued behind the image network's convolutions when two steps were in
// flight.  No library primitive is left in this file.
//   input level  : keys + per-sample histogram (wave-aggregated atomics) -> exclusive scan -> scatter into per-sample buckets
//                  (the order inside a bucket is whatever the atomics gave: the sort below makes it irrelevant) -> sort
//                  (x, y, z | bucket slot) per sample, duplicates of a voxel merged in INPUT-ROW order (mean of their
//                  features: deterministic whatever the bucket order was) -> placement
//   coarser level: sort (key & mask) per sample of the finer level's rows -> placement
// A sample of more than 16384 rows is sorted by the same network in global memory (slow, correct); a sample of more than
// 65536 input points does not fit the 16-bit slot field: its build is FLAGGED (bit 1 of the range flag) like an out-of-range
// coordinate.  All launches go to the caller's stream, temporary storage is the caller's workspace: hipGraph-capturable.
#include <cstring>

#include "common.hpp"

namespace agp_coords {

constexpr int64_t SENT = 0x7fffffffffffffffLL;          // padding key: sorts after every real key
constexpr uint64_t USENT = ~0ull;                        // padding inside the sort (unsigned compare)
constexpr int BITS = 16, OFF = 1 << 15;
constexpr int ST = 1024;                                 // threads of a segment workgroup
constexpr int LDS_KEYS = 16384;                          // keys a workgroup sorts in LDS
constexpr int MAX_SLOT = 65536;                          // input points of one sample (16-bit bucket slot beside the 48-bit voxel)

// coords [n][4] (batch, x, y, z) as int64 (kind 0), float32 (kind 1) or float64 (kind 2; floored like ME does for floating
// coordinates) -> key (input order) + the per-sample histogram.  Out-of-range coordinates (|c| >= 32512: kernel offsets need
// headroom) send their point to the sample's origin voxel, a batch index outside [0, nbatch) is clamped; both are flagged; the flag word describes THIS build (agp_sparse_build
// zeroes it first).  A workgroup owns PB consecutive points; clouds arrive sample after sample, so nearly every workgroup sees
// ONE batch index and adds its whole count with one atomic (same-address atomics serialise in L2: one per wave cost 48 us
// for 64 x 8000 points, one per point would cost milliseconds).
constexpr int PB = 1024;
__global__ void __launch_bounds__(256) keys_hist_kernel(const void* __restrict__ coords, int kind, int64_t n, int nbatch,
                                                        int64_t* __restrict__ keys, int32_t* __restrict__ hist, int32_t* __restrict__ flag) {
    const int64_t base = (int64_t)blockIdx.x * PB;
    const int b_first = [&] {
        int64_t c0;
        if (kind == 0) c0 = ((const int64_t*)coords)[base * 4];
        else if (kind == 1) c0 = (int64_t)floorf(((const float*)coords)[base * 4]);
        else c0 = (int64_t)floor(((const double*)coords)[base * 4]);
        return (int)(c0 < 0 ? 0 : (c0 >= nbatch ? nbatch - 1 : c0));
    }();
    int bs[PB / 256];
    bool uniform = true;
#pragma unroll
    for (int u = 0; u < PB / 256; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        bs[u] = -1;
        if (i >= n) continue;
        int64_t c[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (kind == 0) c[a] = ((const int64_t*)coords)[i * 4 + a];
            else if (kind == 1) c[a] = (int64_t)floorf(((const float*)coords)[i * 4 + a]);
            else c[a] = (int64_t)floor(((const double*)coords)[i * 4 + a]);
        }
        // (a batch index >= nbatch would index the per-sample tables of the segment kernels out of bounds: clamped and flagged too)
        bool bad = c[0] < 0 || c[0] >= nbatch;
        bool far = false;
#pragma unroll
        for (int a = 1; a < 4; ++a) far = far || c[a] > OFF - 257 || c[a] < -(OFF - 257);
        if (far) {
            // a point outside the key range joins the ORIGIN voxel of its sample (the sensor position, inside every LiDAR sweep) --
            // not a voxel at the edge of the range: the build is flagged either way, and an isolated voxel tens of thousands of
            // cells from the cloud is exactly the input the flag exists to keep out of the kernels behind it
            c[1] = c[2] = c[3] = 0;
            bad = true;
        }
        if (c[0] < 0) c[0] = 0;
        if (c[0] >= nbatch) c[0] = nbatch - 1;
        if (bad) atomicOr(flag, 1);
        int64_t k = c[0];
#pragma unroll
        for (int a = 1; a < 4; ++a) k = (k << BITS) | (c[a] + OFF);
        keys[i] = k;
        bs[u] = (int)c[0];
        uniform = uniform && bs[u] == b_first;
    }
    if (__syncthreads_and(uniform ? 1 : 0)) {
        if (threadIdx.x == 0) atomicAdd(&hist[b_first], (int)(n - base < PB ? n - base : PB));
    } else {
        // a block that straddles samples: one atomic per distinct batch index of a wave (a same-address atomic per POINT cost 0.3 us each)
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int u = 0; u < PB / 256; ++u) {
            uint64_t todo = __builtin_amdgcn_ballot_w64(bs[u] >= 0);
            while (todo) {
                const int leader = __ffsll((unsigned long long)todo) - 1;
                const int b0 = __shfl(bs[u], leader, 64);
                const uint64_t same = __builtin_amdgcn_ballot_w64(bs[u] == b0);
                if (lane == leader) atomicAdd(&hist[b0], __popcll(same));
                todo &= ~same;
            }
        }
    }
}

// block-wide exclusive scan of one int per thread (ST threads); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
    __shared__ int wsum[ST / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ST / 64; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// one workgroup: pseg = exclusive scan of the histogram (pseg[B] = n), cursor = its copy (the scatter's running positions)
__global__ void __launch_bounds__(ST) scan_hist_kernel(const int32_t* __restrict__ hist, int nbatch, int64_t* __restrict__ pseg,
                                                       int32_t* __restrict__ cursor) {
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nbatch; b0 += ST) {
        const int b = b0 + threadIdx.x;
        const int v = b < nbatch ? hist[b] : 0;
        int tot;
        const int ex = block_excl_scan(v, &tot);
        const int carry = carry_s;
        if (b < nbatch) { pseg[b] = carry + ex; cursor[b] = carry + ex; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) pseg[nbatch] = carry_s;
}

// points -> their sample's bucket: bkeys / brow [pseg[b], pseg[b+1]).  The same workgroups of PB points: one atomic claims the
// whole block's range when it holds one sample (its points keep their input order and the stores coalesce).
__global__ void __launch_bounds__(256) scatter_kernel(const int64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ cursor,
                                                      int64_t* __restrict__ bkeys, int32_t* __restrict__ brow) {
    __shared__ int s_base;
    const int64_t base = (int64_t)blockIdx.x * PB;
    const int b_first = (int)(keys[base] >> (3 * BITS));
    int64_t ks[PB / 256];
    bool uniform = true;
#pragma unroll
    for (int u = 0; u < PB / 256; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        ks[u] = i < n ? keys[i] : -1;
        uniform = uniform && (i >= n || (int)(ks[u] >> (3 * BITS)) == b_first);
    }
    if (__syncthreads_and(uniform ? 1 : 0)) {
        if (threadIdx.x == 0) s_base = atomicAdd(&cursor[b_first], (int)(n - base < PB ? n - base : PB));
        __syncthreads();
        const int pb = s_base;
#pragma unroll
        for (int u = 0; u < PB / 256; ++u) {
            const int64_t i = base + u * 256 + threadIdx.x;
            if (i < n) { bkeys[pb + (i - base)] = ks[u]; brow[pb + (i - base)] = (int32_t)i; }
        }
    } else {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int u = 0; u < PB / 256; ++u) {
            const int64_t i = base + u * 256 + threadIdx.x;
            const int b = i < n ? (int)(ks[u] >> (3 * BITS)) : -1;
            uint64_t todo = __builtin_amdgcn_ballot_w64(b >= 0);
            while (todo) {
                const int leader = __ffsll((unsigned long long)todo) - 1;
                const int b0 = __shfl(b, leader, 64);
                const uint64_t same = __builtin_amdgcn_ballot_w64(b == b0);
                int pb = 0;
                if (lane == leader) pb = atomicAdd(&cursor[b0], __popcll(same));
                pb = __shfl(pb, leader, 64);
                if (b == b0) {
                    const int pos = pb + __popcll(same & ((1ull << lane) - 1ull));
                    bkeys[pos] = ks[u];
                    brow[pos] = (int32_t)i;
                }
                todo &= ~same;
            }
        }
    }
}

// the key array of a sort: in LDS one key of padding follows every 32 (a thread's chunk of C consecutive keys starts C * 8 bytes
// after its neighbour's: 16 lanes on one bank pair without it), in global memory plain
template <bool PAD>
struct KeyArr {
    uint64_t* p;
    __device__ __forceinline__ uint64_t& operator[](int i) const { return p[PAD ? i + (i >> 5) : i]; }
};

// ascending bitonic network over s[0, P) (P a power of two), all ST threads of the workgroup.  A pass moves every key through
// LDS once (8192 keys x 91 passes = 12 MB: the kernel is bound by that, not by its barriers), so the passes whose partner
// distance fits a thread's own C = P / ST consecutive keys run in registers: one load + store per size k instead of log2(C).
template <int C, class PTR>
__device__ __forceinline__ void bitonic_sort_c(PTR s, int P) {
    const int tid = threadIdx.x;
    uint64_t e[C];
    static_assert(C <= 32, "a thread's chunk lies inside one padding group");
    auto local = [&](int k_lo, int k_hi) {          // every pass with j < C of the sizes k_lo .. k_hi (k_hi <= C: whole sizes)
#pragma unroll
        for (int r = 0; r < C; ++r) e[r] = s[tid * C + r];
        for (int k = k_lo; k <= k_hi; k <<= 1) {
#pragma unroll
            for (int j = C >> 1; j > 0; j >>= 1) {
                if (j >= k) continue;
#pragma unroll
                for (int r = 0; r < C; ++r) {
                    if (r & j) continue;
                    const bool up = ((tid * C + r) & k) == 0;
                    const uint64_t a = e[r], b = e[r | j];
                    if ((a > b) == up) { e[r] = b; e[r | j] = a; }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < C; ++r) s[tid * C + r] = e[r];
        __syncthreads();
    };
    if (C > 1) local(2, C);
    for (int k = 2 * C; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) {
            uint64_t a[(C + 1) / 2], b[(C + 1) / 2];
#pragma unroll
            for (int u = 0; u < (C + 1) / 2; ++u) {
                const int t = tid + u * ST;
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                a[u] = s[i]; b[u] = s[i | j];
            }
#pragma unroll
            for (int u = 0; u < (C + 1) / 2; ++u) {
                const int t = tid + u * ST;
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const bool up = (i & k) == 0;
                if ((a[u] > b[u]) == up) { s[i] = b[u]; s[i | j] = a[u]; }
            }
            __syncthreads();
        }
        if (C > 1) local(k, k);
    }
}

template <class PTR>
__device__ __forceinline__ void bitonic_sort(PTR s, int P) {
    const int tid = threadIdx.x;
    if (P == 16 * ST) return bitonic_sort_c<16>(s, P);
    if (P == 8 * ST) return bitonic_sort_c<8>(s, P);
    if (P == 4 * ST) return bitonic_sort_c<4>(s, P);
    if (P == 2 * ST) return bitonic_sort_c<2>(s, P);
    for (int k = 2; k <= P; k <<= 1)                  // a small sample (P <= ST) or a huge one in global memory
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += ST) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const uint64_t a = s[i], b = s[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
}

// One workgroup per batch sample: sort the sample's keys, drop the duplicates, count.
//   L0 = false: in = the finer level's keys, rows [seg[b], seg[b+1]); the sorted value is key & mask
//   L0 = true : in = bkeys (bucket order), the sorted value is (x, y, z) << 16 | bucket slot; the duplicates of a voxel are merged
//               in input-row order: f_tmp = mean of their feature rows
// unique keys -> uniq_tmp[seg[b] ...), their count -> cnt[b]
template <bool L0, class PTR>
__device__ __forceinline__ void seg_sort_body(PTR s, const int64_t* __restrict__ in, int64_t r0, int n, int P, int64_t mask, int b,
                                              const int32_t* __restrict__ brow, const float* __restrict__ f, int cf,
                                              int64_t* __restrict__ uniq_tmp, float* __restrict__ f_tmp, int32_t* __restrict__ cnt) {
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += ST) {
        uint64_t v = USENT;
        if (i < n) {
            const uint64_t k = (uint64_t)in[r0 + i];
            v = L0 ? (((k & 0xffffffffffffull) << 16) | (uint64_t)i) : (k & (uint64_t)mask);
        }
        s[i] = v;
    }
    __syncthreads();
    bitonic_sort(s, P);
    const int per = (P + ST - 1) / ST, i0 = tid * per;
    auto vox = [&](int i) -> uint64_t { const uint64_t v = s[i]; return L0 ? (v >> 16) : v; };
    int c = 0;
    for (int u = 0; u < per; ++u) {
        const int i = i0 + u;
        if (i < n && (i == 0 || vox(i) != vox(i - 1))) ++c;
    }
    int tot;
    int pos = block_excl_scan(c, &tot);
    for (int u = 0; u < per; ++u) {
        const int i = i0 + u;
        if (!(i < n && (i == 0 || vox(i) != vox(i - 1)))) continue;
        const uint64_t v = vox(i);
        uniq_tmp[r0 + pos] = L0 ? (int64_t)(((uint64_t)b << (3 * BITS)) | v) : (int64_t)v;
        if (L0 && f_tmp) {
            int e = i + 1;
            while (e < n && vox(e) == v) ++e;
            if (e == i + 1) {
                const int row = brow[r0 + (int)(s[i] & 0xffffull)];
                for (int ch = 0; ch < cf; ++ch) f_tmp[(size_t)(r0 + pos) * cf + ch] = f[(size_t)row * cf + ch];
            } else {
                // the voxel's points in ascending input row (a handful: selection by repeated minimum), summed in that order
                const float dup = (float)(e - i);
                for (int ch = 0; ch < cf; ++ch) {
                    float sum = 0.f;
                    int last = -1;
                    for (int t = i; t < e; ++t) {
                        int m = 0x7fffffff;
                        for (int jj = i; jj < e; ++jj) {
                            const int row = brow[r0 + (int)(s[jj] & 0xffffull)];
                            if (row > last && row < m) m = row;
                        }
                        sum += f[(size_t)m * cf + ch];
                        last = m;
                    }
                    f_tmp[(size_t)(r0 + pos) * cf + ch] = sum / dup;
                }
            }
        }
        ++pos;
    }
    if (tid == 0) cnt[b] = tot;
}

template <bool L0>
__global__ void __launch_bounds__(ST) seg_sort_kernel(const int64_t* __restrict__ in, const int64_t* __restrict__ seg, int64_t mask,
                                                      const int32_t* __restrict__ brow, const float* __restrict__ f, int cf,
                                                      int64_t* __restrict__ uniq_tmp, float* __restrict__ f_tmp, int32_t* __restrict__ cnt,
                                                      uint64_t* __restrict__ gscr, int32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
    const int b = blockIdx.x;
    const int64_t r0 = seg[b];
    const int64_t nn = seg[b + 1] - r0;
    if (nn <= 0) { if (threadIdx.x == 0) cnt[b] = 0; return; }
    if (L0 && nn > MAX_SLOT) {          // (uniform) the slot field cannot number this sample's points: empty sample, flagged
        if (threadIdx.x == 0) { cnt[b] = 0; atomicOr(flag, 2); }
        return;
    }
    const int n = (int)nn;
    int P = 2;
    while (P < n) P <<= 1;
    if (n <= LDS_KEYS) seg_sort_body<L0>(KeyArr<true>{sk}, in, r0, n, P, mask, b, brow, f, cf, uniq_tmp, f_tmp, cnt);
    else seg_sort_body<L0>(KeyArr<false>{gscr + 2 * r0}, in, r0, n, P, mask, b, brow, f, cf, uniq_tmp, f_tmp, cnt);     // P < 2 n: inside [2 r0, 2 r1)
}

// placement: sample b's unique keys go to rows [off_b, off_b + cnt[b]) with off = exclusive scan of cnt; batch indices, segment
// offsets (seg_off[B] = n), padding of rows >= n; optional feature rows
__global__ void __launch_bounds__(256) place_kernel(const int64_t* __restrict__ uniq_tmp, const float* __restrict__ f_tmp, int cf,
                                                    const int64_t* __restrict__ seg_in, const int32_t* __restrict__ cnt, int nbatch,
                                                    int64_t cap, int64_t* __restrict__ keys_out, float* __restrict__ f_out,
                                                    int64_t* __restrict__ seg_off, int32_t* __restrict__ bidx) {
    __shared__ int64_t red[2][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    int64_t pre = 0, all = 0;
    for (int a = tid; a < nbatch; a += 256) {
        const int c = cnt[a];
        all += c;
        if (a < b) pre += c;
    }
    red[0][tid] = pre; red[1][tid] = all;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    const int64_t off = red[0][0], tot = red[1][0];
    const int64_t r0 = seg_in[b];
    const int c = cnt[b];
    for (int j = tid; j < c; j += 256) {
        keys_out[off + j] = uniq_tmp[r0 + j];
        bidx[off + j] = b;
        if (f_out)
            for (int ch = 0; ch < cf; ++ch) f_out[(size_t)(off + j) * cf + ch] = f_tmp[(size_t)(r0 + j) * cf + ch];
    }
    if (tid == 0) {
        seg_off[b] = off;
        if (b == nbatch - 1) seg_off[nbatch] = tot;
    }
    const int64_t lo = cap * b / nbatch, hi = cap * (b + 1) / nbatch;
    for (int64_t i = (lo > tot ? lo : tot) + tid; i < hi; i += 256) { keys_out[i] = SENT; bidx[i] = 0; }
}

// Row order for the gather-GEMM: inside every batch sample, rows grouped by z-plane (stable: (x, y) order inside a plane).  Rows
// sort (x, y, z) with z FASTEST, so a 128-row tile of the natural order holds every z-plane of its columns and therefore needs
// every tap; a tile of ONE plane at the bottom (top) of the sample has no dz = -1 (+1) neighbour at all, and the gather-GEMM's
// per-tile tap list drops those 9 of 27 taps.  A coarse LiDAR level is 1-3 voxels thick: a third of its MFMAs.
// One workgroup per sample: z values present -> LDS bitmap -> ascending list (<= 64 planes, else the identity order) -> per plane
// one block scan over the threads' row chunks (samples of more than 16384 rows keep the natural order).  perm[m] = the output row
// GEMM row m computes; identity past the valid rows.
__global__ void __launch_bounds__(ST) zplane_perm_kernel(const int64_t* __restrict__ keys, const int64_t* __restrict__ seg_off, int nbatch,
                                                         int64_t cap, int32_t* __restrict__ perm) {
    __shared__ uint32_t bm[2048];
    __shared__ int zs[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = seg_off[b], n_total = seg_off[nbatch];
    const int n = (int)(seg_off[b + 1] - r0);
    {
        const int64_t lo = cap * b / nbatch, hi = cap * (b + 1) / nbatch;
        for (int64_t i = (lo > n_total ? lo : n_total) + tid; i < hi; i += ST) perm[i] = (int32_t)i;
    }
    if (n <= 0) return;
    bm[tid] = 0u; bm[tid + ST] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += ST) {
        const uint32_t z = (uint32_t)(keys[r0 + i] & 0xffff);
        atomicOr(&bm[z >> 5], 1u << (z & 31));
    }
    __syncthreads();
    const uint32_t w0 = bm[2 * tid], w1 = bm[2 * tid + 1];
    int tot;
    int pos = block_excl_scan(__popc(w0) + __popc(w1), &tot);
    if (tot > 64) {                                         // (uniform) too many planes to be worth grouping
        for (int i = tid; i < n; i += ST) perm[r0 + i] = (int32_t)(r0 + i);
        return;
    }
    for (uint32_t m = w0; m; m &= m - 1) zs[pos++] = 64 * tid + __builtin_ctz(m);
    for (uint32_t m = w1; m; m &= m - 1) zs[pos++] = 64 * tid + 32 + __builtin_ctz(m);
    __syncthreads();
    // a thread's rows' z values stay in registers over the plane loop (<= 16 rows per thread; a larger sample keeps its natural order)
    constexpr int ZR = LDS_KEYS / ST;
    const int per = (n + ST - 1) / ST, i0 = tid * per;
    if (per > ZR) {
        for (int i = tid; i < n; i += ST) perm[r0 + i] = (int32_t)(r0 + i);
        return;
    }
    int zv[ZR];
#pragma unroll
    for (int u = 0; u < ZR; ++u) zv[u] = (u < per && i0 + u < n) ? (int)(keys[r0 + i0 + u] & 0xffff) : -1;
    int base = 0;
    for (int pl = 0; pl < tot; ++pl) {
        const int zp = zs[pl];
        int c = 0;
#pragma unroll
        for (int u = 0; u < ZR; ++u) c += zv[u] == zp ? 1 : 0;
        int total;
        int at = base + block_excl_scan(c, &total);
#pragma unroll
        for (int u = 0; u < ZR; ++u)
            if (zv[u] == zp) perm[r0 + at++] = (int32_t)(r0 + i0 + u);
        base += total;
    }
}

inline int64_t al(int64_t v) { return (v + 255) / 256 * 256; }
struct Ws { int64_t keys, bkeys, brow, uniq, ftmp, gscr, hist, pseg, cursor, cnt, total; };
inline Ws layout(int64_t cap, int nbatch, int cf) {
    Ws w;
    w.keys = 0;
    w.bkeys = al(w.keys + cap * 8);
    w.brow = al(w.bkeys + cap * 8);
    w.uniq = al(w.brow + cap * 4);
    w.ftmp = al(w.uniq + cap * 8);
    w.gscr = al(w.ftmp + cap * 4 * (int64_t)(cf > 0 ? cf : 0));
    w.hist = al(w.gscr + 2 * cap * 8);
    w.pseg = al(w.hist + (int64_t)nbatch * 4);
    w.cursor = al(w.pseg + (int64_t)(nbatch + 1) * 8);
    w.cnt = al(w.cursor + (int64_t)nbatch * 4);
    w.total = al(w.cnt + (int64_t)nbatch * 4);
    return w;
}
inline dim3 grid_rows(int64_t n) { return dim3((unsigned)((n + PB - 1) / PB)); }

template <bool L0>
static int launch_seg_sort(const int64_t* in, const int64_t* seg, int64_t mask, const int32_t* brow, const float* f, int cf,
                           int64_t* uniq, float* ftmp, int32_t* cnt, uint64_t* gscr, int32_t* flag, int nbatch, hipStream_t s) {
    constexpr int lds = (LDS_KEYS + LDS_KEYS / 32) * 8;
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)seg_sort_kernel<L0>, lds, attr_done)) return AGP_E_LAUNCH;
    AGP_LAUNCH(seg_sort_kernel<L0>, dim3(nbatch), dim3(ST), lds, s, in, seg, mask, brow, f, cf, uniq, ftmp, cnt, gscr, flag);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

__global__ void zero_words_kernel(int32_t* flag, int32_t* hist, int nbatch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *flag = 0;
    if (i < nbatch) hist[i] = 0;
}

}  // namespace agp_coords
using namespace agp_coords;

extern "C" int64_t agp_sparse_coords_workspace_bytes(int64_t cap, int nbatch, int cfeat) {
    if (cap < 1) cap = 1;
    if (nbatch < 1) nbatch = 1;
    return layout(cap, nbatch, cfeat).total;
}

extern "C" int agp_sparse_build(const void* coords, int kind, int64_t n, const float* feats, int cfeat, int nbatch, int64_t* keys,
                                float* feats_out, int64_t* seg_off, int32_t* bidx, int32_t* range_flag, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    if (!coords || !keys || !seg_off || !bidx || !range_flag || !workspace || n <= 0 || n >= (1ll << 30) || nbatch <= 0 ||
        nbatch >= 0x7fff || kind < 0 || kind > 2 || (feats && (!feats_out || cfeat <= 0)))
        return AGP_E_BADARG;
    const int cf = feats ? cfeat : 0;
    const Ws w = layout(n, nbatch, cf);
    if (workspace_bytes < w.total) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int32_t* hist = (int32_t*)(ws + w.hist);
    // the flag of THIS build, not of an earlier one; the histogram.  A KERNEL, not hipMemsetAsync: this entry point is captured into
    // hipGraphs (MM.forward_q from coords), and a captured memset node of ROCm 7.2 is not safe beside eager memsets issued between
    // replays -- found in round 6: after an eager 4-byte memset of another buffer the replayed graph's memset node left 0x01010101
    // in a word it had no business with (or faulted on a wild address) in every run; the same graph with this kernel: clean
    // (profiles/README.md, round 6).  No library entry point that may be captured issues a memset any more.
    AGP_LAUNCH(zero_words_kernel, dim3((nbatch + 255) / 256), dim3(256), 0, s, range_flag, hist, nbatch);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(keys_hist_kernel, grid_rows(n), dim3(256), 0, s, coords, kind, n, nbatch, (int64_t*)(ws + w.keys), hist, range_flag);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(scan_hist_kernel, dim3(1), dim3(ST), 0, s, hist, nbatch, (int64_t*)(ws + w.pseg), (int32_t*)(ws + w.cursor));
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(scatter_kernel, grid_rows(n), dim3(256), 0, s, (const int64_t*)(ws + w.keys), n, (int32_t*)(ws + w.cursor),
               (int64_t*)(ws + w.bkeys), (int32_t*)(ws + w.brow));
    AGP_CHECK_LAUNCH();
    const int rc = launch_seg_sort<true>((const int64_t*)(ws + w.bkeys), (const int64_t*)(ws + w.pseg), 0, (const int32_t*)(ws + w.brow),
                                         feats, cf, (int64_t*)(ws + w.uniq), cf ? (float*)(ws + w.ftmp) : nullptr,
                                         (int32_t*)(ws + w.cnt), (uint64_t*)(ws + w.gscr), range_flag, nbatch, s);
    if (rc != AGP_OK) return rc;
    AGP_LAUNCH(place_kernel, dim3(nbatch), dim3(256), 0, s, (const int64_t*)(ws + w.uniq), cf ? (const float*)(ws + w.ftmp) : nullptr,
               cf, (const int64_t*)(ws + w.pseg), (const int32_t*)(ws + w.cnt), nbatch, n, keys, cf ? feats_out : nullptr, seg_off, bidx);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_sparse_coarsen(const int64_t* keys, const int64_t* seg_off_in, int64_t cap, int stride, int nbatch,
                                  int64_t* keys_out, int64_t* seg_off, int32_t* bidx, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
    if (!keys || !seg_off_in || !keys_out || !seg_off || !bidx || !workspace || cap <= 0 || cap >= (1ll << 30) || nbatch <= 0 ||
        nbatch >= 0x7fff || stride < 1 || stride > 4096 || (stride & (stride - 1)))
        return AGP_E_BADARG;
    const Ws w = layout(cap, nbatch, 0);
    if (workspace_bytes < w.total) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int64_t low = 2 * stride - 1;
    const int64_t mask = ~((low << (2 * BITS)) | (low << BITS) | low);     // floor(c / 2s) * 2s per axis on 2^15-biased fields
    const int rc = launch_seg_sort<false>(keys, seg_off_in, mask, nullptr, nullptr, 0, (int64_t*)(ws + w.uniq), nullptr,
                                          (int32_t*)(ws + w.cnt), (uint64_t*)(ws + w.gscr), nullptr, nbatch, s);
    if (rc != AGP_OK) return rc;
    AGP_LAUNCH(place_kernel, dim3(nbatch), dim3(256), 0, s, (const int64_t*)(ws + w.uniq), (const float*)nullptr, 0, seg_off_in,
               (const int32_t*)(ws + w.cnt), nbatch, cap, keys_out, (float*)nullptr, seg_off, bidx);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_sparse_zplane_perm(const int64_t* keys, const int64_t* seg_off, int nbatch, int64_t cap, int32_t* perm, void* stream) {
    if (!keys || !seg_off || !perm || nbatch <= 0 || cap <= 0 || cap >= (1ll << 31)) return AGP_E_BADARG;
    AGP_LAUNCH(zplane_perm_kernel, dim3(nbatch), dim3(ST), 0, (hipStream_t)stream, keys, seg_off, nbatch, cap, perm);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
