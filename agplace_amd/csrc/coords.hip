// Coordinate manager of the sparse-voxel branch on the device, WITHOUT host synchronisation (the part of
// MinkowskiEngine's CoordinateManager the path uses: reference network_mm/mm.py:86-87 `ME.SparseTensor(features,
// coordinates)`, models/minkfpn.py:53 stride-2 convolutions; semantics restated in agplace_amd/sparse/coords.py).
//
// Capacity mode: a level of `cap` rows (cap = the number of INPUT points, an upper bound for every level) holds
//   keys[cap]      sorted unique int64 keys (batch, x, y, z: 16-bit biased fields), rows >= n padded with SENT
//   seg_off[B+1]   first row of every batch sample; seg_off[B] = n, the number of valid rows -- the device-side count every
//                  row-wise kernel of the branch reads instead of a host integer (no .item(), no torch.unique)
//   bidx[cap]      batch index of a row
// Building a level = keys -> radix sort -> head flags -> exclusive scan -> compaction (-> mean of duplicate rows' features).
// The sort is rocPRIM's device radix sort (a plain library primitive, like a library GEMM); everything else is written here.
// All launches go to the caller's stream, temporary storage is the caller's workspace: hipGraph-capturable.
#include <cstring>

#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace agp_coords {

constexpr int64_t SENT = 0x7fffffffffffffffLL;          // padding key: sorts after every real key
constexpr int64_t SENT_MIN = (int64_t)0x7fff << 48;      // keys >= this are padding (batch field 0x7fff)
constexpr int BITS = 16, OFF = 1 << 15;
constexpr int SCAN_T = 256, SCAN_I = 8, SCAN_B = SCAN_T * SCAN_I;

// coords [n][4] (batch, x, y, z) as int64 (kind 0), float32 (kind 1) or float64 (kind 2; floored like ME does for floating
// coordinates) -> key, payload = input row.  Out-of-range coordinates (|c| >= 32512: kernel offsets need headroom; batch index
// outside [0, nbatch)) are clamped and flagged; the flag word describes THIS build (agp_sparse_build zeroes it first).
__global__ void keys_kernel(const void* __restrict__ coords, int kind, int64_t n, int64_t cap, int nbatch, int64_t* __restrict__ keys,
                            int32_t* __restrict__ idx, int32_t* __restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = SENT;
        if (i < n) {
            int64_t c[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (kind == 0) c[a] = ((const int64_t*)coords)[i * 4 + a];
                else if (kind == 1) c[a] = (int64_t)floorf(((const float*)coords)[i * 4 + a]);
                else c[a] = (int64_t)floor(((const double*)coords)[i * 4 + a]);
            }
            // (a batch index >= nbatch would index the per-sample tables of the segment kernels out of bounds: clamped and flagged too)
            bool bad = c[0] < 0 || c[0] >= nbatch;
#pragma unroll
            for (int a = 1; a < 4; ++a) {
                if (c[a] > OFF - 257) { c[a] = OFF - 257; bad = true; }
                if (c[a] < -(OFF - 257)) { c[a] = -(OFF - 257); bad = true; }
            }
            if (c[0] < 0) c[0] = 0;
            if (c[0] >= nbatch) c[0] = nbatch - 1;
            if (bad) atomicOr(flag, 1);
            k = c[0];
#pragma unroll
            for (int a = 1; a < 4; ++a) k = (k << BITS) | (c[a] + OFF);
        }
        keys[i] = k;
        idx[i] = (int32_t)i;
    }
}

// floor(c / s2) * s2 per axis = clearing the low bits of every (2^15-biased) 16-bit field; padding stays padding
__global__ void mask_kernel(const int64_t* __restrict__ in, int64_t cap, int64_t mask, int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = in[i];
        out[i] = k >= SENT_MIN ? SENT : (k & mask);
    }
}

__device__ __forceinline__ bool is_head(const int64_t* ks, int64_t i, int64_t cap) {
    if (i >= cap) return false;
    const int64_t k = ks[i];
    return k < SENT_MIN && (i == 0 || ks[i - 1] != k);
}

// block-wide exclusive scan of one int per thread (256 threads); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
    __shared__ int wsum[SCAN_T / 64];
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
    for (int w = 0; w < SCAN_T / 64; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ void __launch_bounds__(SCAN_T) count_heads_kernel(const int64_t* __restrict__ ks, int64_t cap, int32_t* __restrict__ bsum) {
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
    int c = 0;
#pragma unroll
    for (int u = 0; u < SCAN_I; ++u) c += is_head(ks, i0 + u, cap) ? 1 : 0;
    int tot;
    block_excl_scan(c, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the block sums in place; total -> *n_out
__global__ void __launch_bounds__(SCAN_T) scan_bsum_kernel(int32_t* __restrict__ bsum, int nblk, int64_t* __restrict__ n_out) {
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += SCAN_T) {
        const int b = b0 + threadIdx.x;
        const int v = b < nblk ? bsum[b] : 0;
        int tot;
        const int ex = block_excl_scan(v, &tot);
        const int carry = carry_s;
        if (b < nblk) bsum[b] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = carry_s;
}

// compaction: the j-th head goes to row j; optional features: mean over the run of equal keys of f[idx[.]][0..cf)
__global__ void __launch_bounds__(SCAN_T) compact_kernel(const int64_t* __restrict__ ks, const int32_t* __restrict__ idx, int64_t cap,
                                                         const int32_t* __restrict__ bsum, int64_t* __restrict__ keys_out,
                                                         const float* __restrict__ f, int cf, float* __restrict__ f_out) {
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
    bool h[SCAN_I];
    int c = 0;
#pragma unroll
    for (int u = 0; u < SCAN_I; ++u) { h[u] = is_head(ks, i0 + u, cap); c += h[u] ? 1 : 0; }
    int tot;
    int pos = bsum[blockIdx.x] + block_excl_scan(c, &tot);
#pragma unroll
    for (int u = 0; u < SCAN_I; ++u) {
        if (!h[u]) continue;
        const int64_t i = i0 + u, k = ks[i];
        keys_out[pos] = k;
        if (f_out) {
            int64_t e = i + 1;
            while (e < cap && ks[e] == k) ++e;
            const float cnt = (float)(e - i);
            for (int ch = 0; ch < cf; ++ch) {
                float s = 0.f;
                for (int64_t j = i; j < e; ++j) s += f[(size_t)idx[j] * cf + ch];
                f_out[(size_t)pos * cf + ch] = s / cnt;
            }
        }
        ++pos;
    }
}

// padding + segments: keys_out[i] = SENT for i >= n; bidx[i]; seg_off[b] = first row of sample b (seg_off[B] = n)
__global__ void finish_kernel(int64_t* __restrict__ keys_out, int64_t cap, const int64_t* __restrict__ n_ptr, int nbatch,
                              int64_t* __restrict__ seg_off, int32_t* __restrict__ bidx) {
    const int64_t n = *n_ptr;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= n) { keys_out[i] = SENT; bidx[i] = 0; }
        else bidx[i] = (int32_t)(keys_out[i] >> (3 * BITS));
    }
    if (t0 < nbatch) {          // (seg_off[nbatch] = n is already in place: the scan wrote it, and this kernel reads it)
        const int64_t q = (int64_t)t0 << (3 * BITS);
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys_out[mid] < q) lo = mid + 1; else hi = mid;     // (rows < n hold their final keys: written by compact_kernel)
        }
        seg_off[t0] = lo;
    }
}

inline int64_t al(int64_t v) { return (v + 255) / 256 * 256; }
struct Ws { int64_t k0, k1, i0, i1, bsum, tmp, tmp_bytes, total; int nblk; };
inline Ws layout(int64_t cap) {
    Ws w;
    w.nblk = (int)((cap + SCAN_B - 1) / SCAN_B);
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                    (unsigned int)cap, 0, 63, (hipStream_t)0);
    w.tmp_bytes = (int64_t)tb;
    w.k0 = 0; w.k1 = al(w.k0 + cap * 8); w.i0 = al(w.k1 + cap * 8); w.i1 = al(w.i0 + cap * 4);
    w.bsum = al(w.i1 + cap * 4); w.tmp = al(w.bsum + (int64_t)w.nblk * 4); w.total = al(w.tmp + w.tmp_bytes);
    return w;
}
inline dim3 grid_rows(int64_t n) { int64_t g = (n + 255) / 256; return dim3((unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g))); }

// sorted unique compaction of ws.k0 (keys, payload ws.i0) into keys_out / seg_off / bidx / f_out
static int sort_unique(char* ws, const Ws& w, int64_t cap, int nbatch, bool pairs, const float* f, int cf, int64_t* keys_out,
                       float* f_out, int64_t* seg_off, int32_t* bidx, hipStream_t s) {
    size_t tb = (size_t)w.tmp_bytes;
    hipError_t e;
    if (pairs)
        e = rocprim::radix_sort_pairs(ws + w.tmp, tb, (const uint64_t*)(ws + w.k0), (uint64_t*)(ws + w.k1), (const int32_t*)(ws + w.i0),
                                      (int32_t*)(ws + w.i1), (unsigned int)cap, 0, 63, s);
    else
        e = rocprim::radix_sort_keys(ws + w.tmp, tb, (const uint64_t*)(ws + w.k0), (uint64_t*)(ws + w.k1), (unsigned int)cap, 0, 63, s);
    if (e != hipSuccess) return AGP_E_LAUNCH;
    const int64_t* ks = (const int64_t*)(ws + w.k1);
    AGP_LAUNCH(count_heads_kernel, dim3(w.nblk), dim3(SCAN_T), 0, s, ks, cap, (int32_t*)(ws + w.bsum));
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(scan_bsum_kernel, dim3(1), dim3(SCAN_T), 0, s, (int32_t*)(ws + w.bsum), w.nblk, seg_off + nbatch);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(compact_kernel, dim3(w.nblk), dim3(SCAN_T), 0, s, ks, (const int32_t*)(ws + w.i1), cap, (const int32_t*)(ws + w.bsum),
               keys_out, pairs ? f : nullptr, cf, pairs ? f_out : nullptr);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(finish_kernel, grid_rows(cap > nbatch ? cap : nbatch), dim3(256), 0, s, keys_out, cap, seg_off + nbatch, nbatch,
               seg_off, bidx);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

}  // namespace agp_coords
using namespace agp_coords;

extern "C" int64_t agp_sparse_coords_workspace_bytes(int64_t cap) {
    if (cap < 1) cap = 1;
    return layout(cap).total;
}

extern "C" int agp_sparse_build(const void* coords, int kind, int64_t n, const float* feats, int cfeat, int nbatch, int64_t* keys,
                                float* feats_out, int64_t* seg_off, int32_t* bidx, int32_t* range_flag, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    if (!coords || !keys || !seg_off || !bidx || !range_flag || !workspace || n <= 0 || n >= (1ll << 31) || nbatch <= 0 ||
        nbatch >= 0x7fff || kind < 0 || kind > 2 || (feats && (!feats_out || cfeat <= 0)))
        return AGP_E_BADARG;
    const Ws w = layout(n);
    if (workspace_bytes < w.total) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    if (hipMemsetAsync(range_flag, 0, sizeof(int32_t), s) != hipSuccess) return AGP_E_LAUNCH;     // the flag of this build, not of an earlier one
    AGP_LAUNCH(keys_kernel, grid_rows(n), dim3(256), 0, s, coords, kind, n, n, nbatch, (int64_t*)(ws + w.k0), (int32_t*)(ws + w.i0), range_flag);
    AGP_CHECK_LAUNCH();
    return sort_unique(ws, w, n, nbatch, true, feats, cfeat, keys, feats_out, seg_off, bidx, s);
}

extern "C" int agp_sparse_coarsen(const int64_t* keys, int64_t cap, int stride, int nbatch, int64_t* keys_out, int64_t* seg_off,
                                  int32_t* bidx, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!keys || !keys_out || !seg_off || !bidx || !workspace || cap <= 0 || cap >= (1ll << 31) || nbatch <= 0 || stride < 1 ||
        stride > 4096 || (stride & (stride - 1)))
        return AGP_E_BADARG;
    const Ws w = layout(cap);
    if (workspace_bytes < w.total) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int64_t low = 2 * stride - 1;
    const int64_t mask = ~((low << (2 * BITS)) | (low << BITS) | low);
    AGP_LAUNCH(mask_kernel, grid_rows(cap), dim3(256), 0, s, keys, cap, mask, (int64_t*)(ws + w.k0));
    AGP_CHECK_LAUNCH();
    return sort_unique(ws, w, cap, nbatch, false, nullptr, 0, keys_out, nullptr, seg_off, bidx, s);
}
