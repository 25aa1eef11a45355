// Sparse-voxel branch: the kernels beside the gather-GEMM (agp_sparse_conv_fwd
// in igemm.hip).  A sparse tensor is a feature matrix [n + 1][C] in map storage format (the last row is
// zero = "missing neighbour"), rows sorted by (batch, x, y, z), so every batch sample is one
// contiguous segment [seg_off[b], seg_off[b+1]).
#include "common.hpp"

namespace agp_sparse {

// first layer (MinkFPN.conv0, kernel 5, Cin = 1): direct gather, fp32 weights
//   out[i][co] = act( scale[co] * sum_k f[nbr[k][i]] * w[k][co] + shift[co] )
__global__ void conv_cin1_kernel(const float* __restrict__ f, const int32_t* __restrict__ nbr, int64_t n_in, int64_t n_out,
                                 int ntaps, const float* __restrict__ w, int cout, const float* __restrict__ scale,
                                 const float* __restrict__ shift, int relu, bf16_t* __restrict__ o_hi, bf16_t* __restrict__ o_lo,
                                 const int64_t* __restrict__ n_dev) {
    const int groups = cout / 8;
    const int64_t total = (n_dev ? min(n_out, *n_dev) : n_out) * groups;      // capacity mode: the valid rows only
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        const int64_t i = t / groups;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int k = 0; k < ntaps; ++k) {
            const int32_t j = nbr[(size_t)k * n_out + i];
            if (j < 0 || j >= n_in) continue;
            const float v = f[j];
            const float* wk = w + (size_t)k * cout + g * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v * wk[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] = acc[e] * (scale ? scale[g * 8 + e] : 1.f) + (shift ? shift[g * 8 + e] : 0.f);
            if (relu) acc[e] = fmaxf(acc[e], 0.f);
        }
        map_store8(o_hi, o_lo, (size_t)i * cout + g * 8, acc);
    }
}

// per-segment mean and GeM of a feature matrix: one block of 1024 threads per (segment, 64-channel chunk): 8 channel groups x
// 128 row lanes (a segment is thousands of rows and there are only batch x C / 64 blocks: with 32 row lanes the kernel was a
// chain of ~200 dependent memory trips per block, 100-120 us for 64 samples of 6000 rows)
constexpr int SEGP_T = 1024;
__global__ void __launch_bounds__(SEGP_T) seg_pool_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo,
                                                       const int64_t* __restrict__ seg_off, int c, const float* __restrict__ pptr,
                                                       float eps, float* __restrict__ mean_out, float* __restrict__ gem_out) {
    __shared__ float red[SEGP_T][17];
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int g = threadIdx.x & 7, pl = threadIdx.x >> 3;           // 8 channel groups x 128 point lanes
    const int64_t r0 = seg_off[b], r1 = seg_off[b + 1];
    const float p = gem_out ? pptr[0] : 1.f;
    float sm[8], sg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sg[e] = 0.f; }
    const int ch0 = chunk * 64 + g * 8;
    if (ch0 < c) {
        // four rows' loads in flight per thread (the same accumulation order as one at a time: only 64 x C / 64 workgroups
        // exist, memory-level parallelism has to come from inside them)
        constexpr int U = 4, RS = SEGP_T / 8;
        for (int64_t r = r0 + pl; r < r1; r += U * RS) {
            float v[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t ru = r + u * RS;
                map_load8(hi, lo, (size_t)(ru < r1 ? ru : r) * c + ch0, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u * RS >= r1) break;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sm[e] += v[u][e];
                    if (gem_out) sg[e] += __builtin_exp2f(p * __builtin_log2f(fmaxf(v[u][e], eps)));
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = sm[e]; red[threadIdx.x][8 + e] = sg[e]; }
    __syncthreads();
    // 128 row lanes -> one value per channel in two fixed stages (8 x 16, then 8): one thread per channel walking all 128 was a
    // chain of 256 dependent LDS reads, ~9 us of a 33 us launch
    __shared__ double part[8][64][2];
    if (threadIdx.x < 512) {
        const int ch_l = threadIdx.x & 63, pt = threadIdx.x >> 6, gg = ch_l >> 3, e = ch_l & 7;
        double a = 0, q = 0;
        for (int k = pt * 16; k < pt * 16 + 16; ++k) { a += red[k * 8 + gg][e]; q += red[k * 8 + gg][8 + e]; }
        part[pt][ch_l][0] = a; part[pt][ch_l][1] = q;
    }
    __syncthreads();
    if (threadIdx.x < 64 && chunk * 64 + threadIdx.x < c) {
        double a = 0, q = 0;
        for (int pt = 0; pt < 8; ++pt) { a += part[pt][threadIdx.x][0]; q += part[pt][threadIdx.x][1]; }
        const double cnt = (double)(r1 - r0);
        const int ch = chunk * 64 + threadIdx.x;
        if (mean_out) mean_out[(size_t)b * c + ch] = cnt > 0 ? (float)(a / cnt) : 0.f;
        if (gem_out) gem_out[(size_t)b * c + ch] = cnt > 0 ? __builtin_exp2f(__builtin_log2f((float)(q / cnt)) / p) : 0.f;
    }
}

// ECALayer: scale[b][c] = sigmoid( sum_j w[j] * mean[b][c + j - k/2] )   (Conv1d over the channel axis, zero padding)
__global__ void eca_kernel(const float* __restrict__ mean, int nb, int c, const float* __restrict__ w, int k, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * c) return;
    const int b = i / c, ch = i - b * c;
    float s = 0.f;
    for (int j = 0; j < k; ++j) {
        const int cc = ch + j - k / 2;
        if (cc >= 0 && cc < c) s += w[j] * mean[(size_t)b * c + cc];
    }
    out[i] = 1.f / (1.f + __expf(-s));
}

// out[i] = relu?( y[i] * scale[b(i)]? + add[b(i)]? + res[i]? )
__global__ void seg_affine_kernel(const bf16_t* __restrict__ y_hi, const bf16_t* __restrict__ y_lo, const int32_t* __restrict__ bidx,
                                  const float* __restrict__ scale, const float* __restrict__ add, const bf16_t* __restrict__ r_hi,
                                  const bf16_t* __restrict__ r_lo, int64_t n, int c, int relu, bf16_t* __restrict__ o_hi,
                                  bf16_t* __restrict__ o_lo, const int64_t* __restrict__ n_dev) {
    const int groups = c / 8;
    const int64_t total = (n_dev ? min(n, *n_dev) : n) * groups;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        const int64_t i = t / groups;
        const size_t off = (size_t)i * c + g * 8;
        const int b = bidx[i];
        float v[8];
        map_load8(y_hi, y_lo, off, v);
        if (scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= scale[(size_t)b * c + g * 8 + e];
        }
        if (add) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += add[(size_t)b * c + g * 8 + e];
        }
        if (r_hi) {
            float r[8];
            map_load8(r_hi, r_lo, off, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        map_store8(o_hi, o_lo, off, v);
    }
}

// kernel map: nbr[k][i] = row of (out_keys[i] + dkey[k]) in the SORTED in_keys, n_in when absent.
// Keys linearise (batch, x, y, z) with 16-bit biased fields, so a coordinate offset is a key offset.
// Capacity mode (n_dev != nullptr): the key arrays hold `n_in` / `n_out` rows of which only *n_in_dev / *n_dev are valid (the
// rest is padding that sorts last); table entries of padding rows up to the next multiple of 256 rows (what a conv tile
// may touch) point at the zero row, the others are not written.
__global__ void kernel_map_kernel(const int64_t* __restrict__ in_keys, int64_t n_in, const int64_t* __restrict__ out_keys,
                                  int64_t n_out, const int64_t* __restrict__ dkey, int ntaps, int32_t* __restrict__ nbr,
                                  const int64_t* __restrict__ n_dev) {
    const int64_t n_valid = n_dev ? min(n_out, *n_dev) : n_out;
    const int64_t n_rows = n_dev ? min(n_out, (n_valid + 255) / 256 * 256) : n_out;
    const int64_t total = n_rows * ntaps;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(t / n_rows);
        const int64_t i = t - (int64_t)k * n_rows;
        if (i >= n_valid) { nbr[(size_t)k * n_out + i] = (int32_t)n_in; continue; }
        const int64_t q = out_keys[i] + dkey[k];
        int64_t lo = 0, hi = n_in;                       // first position with in_keys[pos] >= q
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (in_keys[mid] < q) lo = mid + 1; else hi = mid;
        }
        nbr[(size_t)k * n_out + i] = (lo < n_in && in_keys[lo] == q) ? (int32_t)lo : (int32_t)n_in;
    }
}

// The same table for the regular offset grids of the path -- an odd kernel centred on the output (offsets (i - k/2) * stride per
// axis) or the 2 x 2 x 2 children of a stride-2 output (offsets i * stride).  (x, y, z) is the sort order, so the k^2 candidate
// cells of one x-plane lie in ONE short key range: a thread owns (row, dx), does one binary search for the first key
// >= (x + dx, y_lo, z_lo) and scans to (x + dx, y_hi, z_hi), dropping every key inside the (dy, dz) window into its tap's table
// (k searches per row; round 3 searched once per (dx, dy) column: k^2).
// kidx = ix + k * iy + k * k * iz (first spatial axis fastest), as in agp_sparse_kernel_map's callers.
__global__ void kernel_map_grid_kernel(const int64_t* __restrict__ in_keys, int64_t n_in, const int64_t* __restrict__ out_keys,
                                       int64_t n_out, int ksize, int centered, int stride, int32_t* __restrict__ nbr,
                                       const int64_t* __restrict__ n_dev, const int64_t* __restrict__ n_in_dev,
                                       const int64_t* __restrict__ in_seg_off) {
    const int64_t n_valid = n_dev ? min(n_out, *n_dev) : n_out;
    const int64_t n_rows = n_dev ? min(n_out, (n_valid + 255) / 256 * 256) : n_out;
    const int64_t n_search = n_in_dev ? min(n_in, *n_in_dev) : n_in;      // valid input rows (padding keys sort last anyway)
    const int k2 = ksize * ksize, r = centered ? ksize / 2 : 0;
    const int span = (ksize - 1) * stride;
    const int64_t total = n_rows * ksize;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(t / n_rows);
        const int64_t i = t - (int64_t)ix * n_rows;
        for (int c = 0; c < k2; ++c) nbr[(size_t)(ix + ksize * c) * n_out + i] = (int32_t)n_in;      // (iy, iz) = (c % k, c / k)
        if (i >= n_valid) continue;
        const int64_t key = out_keys[i];
        const int64_t q0 = key + ((int64_t)((ix - r) * stride) << 32) - ((int64_t)(r * stride) << 16) - (int64_t)r * stride;
        const int64_t q1 = q0 + ((int64_t)span << 16) + (int64_t)span;
        // a neighbour lies in the same batch sample: search that sample's rows only (13 instead of 19 steps at 8000 of 512 k rows)
        int64_t lo = 0, hi = n_search;
        if (in_seg_off) { const int64_t b = key >> 48; lo = in_seg_off[b]; hi = in_seg_off[b + 1]; }
        const int64_t end = hi;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (in_keys[mid] < q0) lo = mid + 1; else hi = mid;
        }
        const int y0 = (int)((q0 >> 16) & 0xffff), z0 = (int)(q0 & 0xffff);
        for (; lo < end; ++lo) {
            const int64_t kk = in_keys[lo];
            if (kk > q1) break;
            const int dy = (int)((kk >> 16) & 0xffff) - y0, dz = (int)(kk & 0xffff) - z0;
            if (dz < 0 || dz > span || dy % stride || dz % stride) continue;
            nbr[(size_t)(ix + ksize * (dy / stride) + k2 * (dz / stride)) * n_out + i] = (int32_t)lo;
        }
    }
}

// MinkFPN.conv0 (odd kernel, Cin = 1) WITHOUT a materialised kernel map: a thread owns one output row and all CO output
// channels and finds its neighbours itself.  z is the lowest key field, so the `ksize` z-neighbours of one (dx, dy) column
// are adjacent in the sorted key array: one binary search per column, then a short forward scan -- ksize^2 searches per
// row instead of ksize^3, and no [ksize^3][n] table (256 MB written and read back for 64 x 8000 voxels at kernel 5).
template <int CO>
__global__ void __launch_bounds__(256) conv0_search_kernel(const int64_t* __restrict__ keys, int64_t cap, const int64_t* __restrict__ n_dev,
                                                           const float* __restrict__ f, int ksize, int stride, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                           bf16_t* __restrict__ o_hi, bf16_t* __restrict__ o_lo,
                                                           const int64_t* __restrict__ seg_off) {
    const int64_t n = n_dev ? min(cap, *n_dev) : cap;
    const int r = ksize / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t key = keys[i];
        float acc[CO];
#pragma unroll
        for (int e = 0; e < CO; ++e) acc[e] = 0.f;
        for (int iy = 0; iy < ksize; ++iy)
            for (int ix = 0; ix < ksize; ++ix) {
                // kidx = ix + k * iy + k * k * iz (first spatial axis fastest); key offset of (dx, dy, dz) = dx << 32 | dy << 16 | dz
                const int64_t q0 = key + ((int64_t)((ix - r) * stride) << 32) + ((int64_t)((iy - r) * stride) << 16) - (int64_t)r * stride;
                int64_t lo = 0, hi = n;                  // (the sample's own rows when the segment offsets are given)
                if (seg_off) { const int64_t b = key >> 48; lo = seg_off[b]; hi = seg_off[b + 1]; }
                const int64_t end = hi;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (keys[mid] < q0) lo = mid + 1; else hi = mid;
                }
                for (int iz = 0; iz < ksize; ++iz) {
                    const int64_t q = q0 + (int64_t)iz * stride;
                    while (lo < end && keys[lo] < q) ++lo;
                    if (lo < end && keys[lo] == q) {
                        const float v = f[lo];
                        const float* wk = w + (size_t)(ix + ksize * iy + ksize * ksize * iz) * CO;
#pragma unroll
                        for (int e = 0; e < CO; ++e) acc[e] += v * wk[e];
                    }
                }
            }
#pragma unroll
        for (int g = 0; g < CO / 8; ++g) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = acc[g * 8 + e] * (scale ? scale[g * 8 + e] : 1.f) + (shift ? shift[g * 8 + e] : 0.f);
                if (relu) o[e] = fmaxf(o[e], 0.f);
            }
            map_store8(o_hi, o_lo, (size_t)i * CO + g * 8, o);
        }
    }
}

// The same layer at AGP_PREC_F16 (one fp16 x fp16 product, fp32 accumulate: the precision the other layers of the branch run at)
// on the MATRIX pipe.  conv0_search_kernel issues ksize^3 * CO multiply-adds per row whether a neighbour exists or not (a wave
// executes a tap's CO FMAs as soon as ONE of its 64 rows has that neighbour: 8000 vector instructions per wave at kernel 5,
// 120 of its 280 us for 64 x 8000 voxels) behind ksize^2 binary searches per row.  Here a workgroup owns 128 rows:
//   1. search: one task per (row, dx): ONE binary search for the first key >= (x + dx, y - r, z - r), then a forward scan to
//      (x + dx, y + r, z + r) -- (x, y, z) is the sort order, so the ksize^2 candidate cells of an x-plane lie in one short key
//      range; every key inside the (dy, dz) window drops its feature as fp16 into the row's line of a zeroed [128][KP] LDS tile;
//   2. out[128][CO] = tile [128][KP] x W [KP][CO] as 32 x 32 x 16 MFMAs (KP = ksize^3 rounded up to 32), BatchNorm + ReLU + fp16
//      in the accumulator layout (W rows permuted so that a lane holds 8 consecutive channels: 16-byte stores).
constexpr int C0_ROWS = 128;
constexpr int C0_WK = 1536;                 // keys (+ features) of a tile's search window in LDS
template <int CO>
__global__ void __launch_bounds__(256) conv0_mfma_kernel(const int64_t* __restrict__ keys, int64_t cap, const int64_t* __restrict__ n_dev,
                                                         const float* __restrict__ f, int ksize, int stride, const float* __restrict__ w,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                         bf16_t* __restrict__ o_hi, const int64_t* __restrict__ seg_off, int KP) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int64_t s_win[2];
    const int ROWB = KP * 2 + 16;                              // bytes of a tile / weight line (padding: 16-lane groups on distinct banks)
    char* const xs = smem;                                     // [C0_ROWS][ROWB]
    char* const wsm = smem + C0_ROWS * ROWB;                   // [CO][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = n_dev ? min(cap, *n_dev) : cap;
    const int r = ksize / 2, k2 = ksize * ksize, ntaps = k2 * ksize;
    // weights fp32 [ntaps][CO] -> fp16 [CO][KP] (zero beyond ntaps), once per workgroup
    for (int t = tid; t < CO * KP; t += 256) {
        const int co = t % CO, k = t / CO;
        *(bf16_t*)(wsm + co * ROWB + k * 2) = k < ntaps ? f2h(w[(size_t)k * CO + co]) : (bf16_t)0;
    }
    const int l31 = lane & 31, lh = lane >> 5;
    const int wrow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const float relu_lo = relu ? 0.f : -65504.f;
    const int ntiles = (int)((n + C0_ROWS - 1) / C0_ROWS);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m0 = (int64_t)tile * C0_ROWS;
        __syncthreads();                                       // the previous tile's fragment reads are done (and W is in place)
        for (int t = tid; t < C0_ROWS * ROWB / 16; t += 256) *(u32x4*)(xs + t * 16) = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
        // ---- 1. search.  Everything the tile's 128 sorted rows can see lies between (x_first - r, ..) and (x_last + r, ..): a few
        // hundred consecutive keys.  Two waves find that window with 64-ary searches (4 dependent round trips instead of 13), it is
        // copied to LDS with its features, and the tasks' binary searches + scans run there; a window that does not fit (a very
        // dense cloud) is searched in global memory as before.  A thread's (up to 3) tasks walk their searches in lockstep.
        int64_t* const kw = (int64_t*)(smem + (C0_ROWS + CO) * ROWB);
        float* const fw = (float*)(kw + C0_WK);
        const int64_t span = ((int64_t)(r * stride) << 32) + ((int64_t)(r * stride) << 16) + (int64_t)(r * stride);
        if (wave < 2) {
            const int64_t last = min(m0 + C0_ROWS - 1, n - 1);
            const int64_t q = wave == 0 ? keys[m0] - span : keys[last] + span;
            int64_t lo_ = 0, hi_ = n;
            if (seg_off) {                                   // (inside the samples the tile touches)
                lo_ = seg_off[keys[m0] >> 48];
                hi_ = seg_off[(keys[last] >> 48) + 1];
            }
            while (hi_ - lo_ > 64) {
                const int64_t step = (hi_ - lo_ + 63) >> 6, pp = lo_ + lane * step;
                const bool before = pp < hi_ && (wave == 0 ? keys[pp] < q : keys[pp] <= q);
                const int c = __popcll(__builtin_amdgcn_ballot_w64(before));
                if (c == 0) { hi_ = lo_; break; }
                const int64_t nlo = lo_ + (c - 1) * step + 1, nhi = lo_ + c * step;
                lo_ = nlo; hi_ = nhi < hi_ ? nhi : hi_;
            }
            if (hi_ > lo_) {
                const int64_t pp = lo_ + lane;
                const bool before = pp < hi_ && (wave == 0 ? keys[pp] < q : keys[pp] <= q);
                lo_ += __popcll(__builtin_amdgcn_ballot_w64(before));
            }
            if (lane == 0) s_win[wave] = lo_;
        }
        __syncthreads();
        const int64_t wlo = s_win[0], wn = s_win[1] - s_win[0];
        const bool in_lds = wn <= C0_WK;
        if (in_lds) {
            for (int i = tid; i < (int)wn; i += 256) { kw[i] = keys[wlo + i]; fw[i] = f[wlo + i]; }
            __syncthreads();
        }
        constexpr int NT = 3;
        static_assert(NT * 256 >= C0_ROWS * 5, "tasks of a tile at kernel 5");
        auto tasks = [&](auto key_at, auto feat_at, bool windowed) {
            int64_t lo[NT], hi[NT], q0[NT], end[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int t = tid + u * 256;
                lo[u] = 0; hi[u] = 0; q0[u] = 0;
                if (t < C0_ROWS * ksize) {
                    const int row = t % C0_ROWS, ix = t / C0_ROWS;
                    const int64_t i = m0 + row;
                    if (i < n) {
                        const int64_t key = keys[i];
                        q0[u] = key + ((int64_t)((ix - r) * stride) << 32) - ((int64_t)(r * stride) << 16) - (int64_t)r * stride;
                        if (windowed) hi[u] = wn;
                        else {
                            hi[u] = n;
                            if (seg_off) { const int64_t b = key >> 48; lo[u] = seg_off[b]; hi[u] = seg_off[b + 1]; }
                        }
                    }
                }
                end[u] = hi[u];
            }
            while (true) {
                bool more = false;
                int64_t km[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    if (lo[u] < hi[u]) km[u] = key_at((lo[u] + hi[u]) >> 1);
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    if (lo[u] < hi[u]) {
                        const int64_t mid = (lo[u] + hi[u]) >> 1;
                        if (km[u] < q0[u]) lo[u] = mid + 1; else hi[u] = mid;
                        more = more || lo[u] < hi[u];
                    }
                if (!more) break;
            }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int t = tid + u * 256;
                if (end[u] == 0) continue;
                const int row = t % C0_ROWS, ix = t / C0_ROWS;
                const int64_t q1 = q0[u] + ((int64_t)(2 * r * stride) << 16) + (int64_t)(2 * r * stride);
                const int y0 = (int)((q0[u] >> 16) & 0xffff), z0 = (int)(q0[u] & 0xffff);
                for (int64_t p_ = lo[u]; p_ < end[u]; ++p_) {
                    const int64_t k = key_at(p_);
                    if (k > q1) break;
                    const int dy = (int)((k >> 16) & 0xffff) - y0, dz = (int)(k & 0xffff) - z0;
                    if (dz < 0 || dz > 2 * r * stride) continue;                       // same x-plane, y in range, z outside the window
                    const int iy = dy / stride, iz = dz / stride;                      // (coordinates of a level are multiples of its stride)
                    *(bf16_t*)(xs + row * ROWB + (ix + ksize * iy + k2 * iz) * 2) = f2h(feat_at(p_));
                }
            }
        };
        if (in_lds) tasks([&](int64_t i) { return kw[i]; }, [&](int64_t i) { return fw[i]; }, true);
        else tasks([&](int64_t i) { return keys[i]; }, [&](int64_t i) { return f[i]; }, false);
        __syncthreads();
        // ---- 2. out = tile x W on the matrix pipe: a wave = 32 rows x CO channels
        f32x16 acc[CO / 32];
#pragma unroll
        for (int a = 0; a < CO / 32; ++a)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
        const char* xr = xs + (wave * 32 + l31) * ROWB + lh * 16;
        const char* wr = wsm + wrow * ROWB + lh * 16;
        for (int ks = 0; ks < KP / 16; ++ks) {
            const bf16x8 xf = *(const bf16x8*)(xr + ks * 32);
#pragma unroll
            for (int a = 0; a < CO / 32; ++a) {
                const bf16x8 wf = *(const bf16x8*)(wr + a * 32 * ROWB + ks * 32);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, xf), acc[a], 0, 0, 0);
            }
        }
        const int64_t i = m0 + wave * 32 + l31;
        if (i < n) {
#pragma unroll
            for (int jj = 0; jj < CO / 16; ++jj) {                                 // channels 16 jj + 8 lh .. + 7
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = 16 * jj + 8 * lh + e;
                    v[e] = acc[jj >> 1][8 * (jj & 1) + e] * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
                }
                *(u32x4*)(o_hi + (size_t)i * CO + 16 * jj + 8 * lh) = pack8_h_lo(v, relu_lo);
            }
        }
    }
#endif
}

// Per 128 GEMM rows (in the gather-GEMM's row order: `perm`, or the natural one), the set of taps for which at least one of them
// has a neighbour: mask[g] bit t.  One wave per granule; the level's convolutions share the result (agp_sparse_conv_fwd skips a
// tile's absent taps).  Rows past the valid count have no neighbours; granules past them get 0.
__global__ void __launch_bounds__(256) tile_taps_kernel(const int32_t* __restrict__ nbr, int64_t n_out, int ntaps, int zero_row,
                                                        const int32_t* __restrict__ perm, const int64_t* __restrict__ n_dev,
                                                        uint32_t* __restrict__ mask, int ngran) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= ngran) return;
    const int64_t n_valid = n_dev ? min(n_out, *n_dev) : n_out;
    const int64_t m0 = (int64_t)g * 128 + lane, m1 = m0 + 64;
    const bool v0 = m0 < n_valid, v1 = m1 < n_valid;
    const int64_t r0 = v0 ? (perm ? perm[m0] : m0) : 0, r1 = v1 ? (perm ? perm[m1] : m1) : 0;
    uint32_t mk = 0u;
    for (int t0 = 0; t0 < ntaps; t0 += 9) {               // nine taps' words in flight, then their ballots
        int a[9], b[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int32_t* tab = nbr + (size_t)(t0 + u < ntaps ? t0 + u : ntaps - 1) * n_out;
            a[u] = tab[r0]; b[u] = tab[r1];
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const bool any = (v0 && a[u] != zero_row) || (v1 && b[u] != zero_row);
            if (__builtin_amdgcn_ballot_w64(any) && t0 + u < ntaps) mk |= 1u << (t0 + u);
        }
    }
    if (lane == 0) mask[g] = mk;
}

// ---- training-path helpers ---------------------------------------------------------------------
// out[b][c] = sum over the rows of segment b of a[i][c] * b[i][c]   (ECA scale gradient); b == nullptr: plain sum
__global__ void __launch_bounds__(256) seg_dot_kernel(const bf16_t* __restrict__ a_hi, const bf16_t* __restrict__ a_lo,
                                                      const bf16_t* __restrict__ b_hi, const bf16_t* __restrict__ b_lo,
                                                      const int64_t* __restrict__ seg_off, int c, float* __restrict__ out) {
    __shared__ float red[256][9];
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int g = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int64_t r0 = seg_off[b], r1 = seg_off[b + 1];
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const int ch0 = chunk * 64 + g * 8;
    if (ch0 < c) {
        for (int64_t r = r0 + pl; r < r1; r += 32) {
            float x[8], y[8];
            map_load8(a_hi, a_lo, (size_t)r * c + ch0, x);
            if (b_hi) {
                map_load8(b_hi, b_lo, (size_t)r * c + ch0, y);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += x[e] * y[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += x[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64 && chunk * 64 + threadIdx.x < c) {
        const int gg = threadIdx.x >> 3, e = threadIdx.x & 7;
        double a = 0;
        for (int k = 0; k < 32; ++k) a += red[k * 8 + gg][e];
        out[(size_t)b * c + chunk * 64 + threadIdx.x] = (float)a;
    }
}

// ECALayer backward on the [B][C] vectors (one block, fixed order):
//   s = sigmoid(conv1d(mean));  g_pre = gs * s * (1 - s)
//   add[b][c] = (1/n_b) * sum_j w[j] * g_pre[b][c - (j - k/2)]       (gradient w.r.t. the mean, spread over the rows)
//   gw[j]     = sum_{b,c} g_pre[b][c] * mean[b][c + j - k/2]
__global__ void eca_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ sc, const float* __restrict__ gs,
                               const int64_t* __restrict__ seg_off, int nb, int c, const float* __restrict__ w, int k,
                               float* __restrict__ add, float* __restrict__ gw) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    for (int i = tid; i < nb * c; i += 256) {
        const int b = i / c, ch = i - b * c;
        const double cnt = (double)(seg_off[b + 1] - seg_off[b]);
        float a = 0.f;
        for (int j = 0; j < k; ++j) {
            const int cc = ch - (j - k / 2);
            if (cc >= 0 && cc < c) {
                const float s = sc[(size_t)b * c + cc];
                a += w[j] * gs[(size_t)b * c + cc] * s * (1.f - s);
            }
        }
        add[i] = cnt > 0 ? (float)(a / cnt) : 0.f;
    }
    for (int j = 0; j < k; ++j) {
        double a = 0;
        for (int i = tid; i < nb * c; i += 256) {
            const int b = i / c, ch = i - b * c;
            const int cc = ch + j - k / 2;
            if (cc >= 0 && cc < c) {
                const float s = sc[i];
                a += (double)(gs[i] * s * (1.f - s)) * mean[(size_t)b * c + cc];
            }
        }
        red[tid] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) gw[j] = (float)red[0];
        __syncthreads();
    }
}

// gradient of the per-sample mean / GeM pooling w.r.t. the feature rows (segment version of pool_bwd):
//   out[i] = base[i]? + gmean[b]/n_b + ggem[b] * y[b]^(1-p) * max(x,eps)^(p-1) * [x >= eps] / n_b ;  gp += dL/dp
__global__ void seg_pool_bwd_kernel(const bf16_t* __restrict__ x_hi, const bf16_t* __restrict__ x_lo, const int32_t* __restrict__ bidx,
                                    const int64_t* __restrict__ seg_off, const float* __restrict__ gmean,
                                    const float* __restrict__ ggem, const float* __restrict__ gem_y, const float* __restrict__ pptr,
                                    float eps, const bf16_t* __restrict__ b_hi, const bf16_t* __restrict__ b_lo, int64_t n, int c,
                                    bf16_t* __restrict__ o_hi, bf16_t* __restrict__ o_lo, float* __restrict__ gp) {
    const int groups = c / 8;
    const int64_t total = n * groups;
    const float p = ggem ? pptr[0] : 1.f;
    float dp = 0.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        const int64_t i = t / groups;
        const int b = bidx[i];
        const size_t off = (size_t)i * c + g * 8;
        const float inv = 1.f / (float)(seg_off[b + 1] - seg_off[b]);
        const bool first = i == seg_off[b];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (b_hi) map_load8(b_hi, b_lo, off, v);
        if (gmean) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += gmean[(size_t)b * c + g * 8 + e] * inv;
        }
        if (ggem) {
            float x[8];
            map_load8(x_hi, x_lo, off, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const size_t vi = (size_t)b * c + g * 8 + e;
                const float yy = gem_y[vi], l2y = __builtin_log2f(yy);
                const float xc = fmaxf(x[e], eps), l2x = __builtin_log2f(xc);
                const float r = ggem[vi] * inv * __builtin_exp2f((1.f - p) * l2y + (p - 1.f) * l2x);
                if (x[e] >= eps) v[e] += r;
                if (gp) {
                    dp += r * xc * (l2x * 0.6931471805599453f) / p;
                    if (first) dp -= ggem[vi] * yy * (l2y * 0.6931471805599453f) / p;
                }
            }
        }
        map_store8(o_hi, o_lo, off, v);
    }
    if (gp) agp_grid_sum_ordered(agp_block_sum_ordered(dp), gp);      // (uniform: every thread of every block)
}

// first-layer weight gradient (Cin = 1): gw[k][co] = sum_i f[nbr[k][i]] * g[i][co].  One block per (tap, 64-channel chunk, row
// SLICE): with one block per tap the 125 blocks of MinkFPN.conv0 walked 128 k rows each in 4000 dependent trips -- 7.3 ms,
// 18 % of the training step with the voxel branch.  Slice partials [nslices][ntaps][cout] are added in slice order by
// conv_cin1_wgrad_reduce_kernel: the same bits every run.
__global__ void __launch_bounds__(256) conv_cin1_wgrad_kernel(const float* __restrict__ f, const int32_t* __restrict__ nbr, int64_t n_in,
                                                              int64_t n_out, const bf16_t* __restrict__ g_hi,
                                                              const bf16_t* __restrict__ g_lo, int cout, int ntaps,
                                                              float* __restrict__ partial) {
    __shared__ float red[256][9];
    const int k = blockIdx.x, chunk = blockIdx.y, slice = blockIdx.z;
    const int g = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int64_t per = ((n_out + gridDim.z - 1) / gridDim.z + 31) / 32 * 32;
    const int64_t r0 = slice * per, r1 = r0 + per < n_out ? r0 + per : n_out;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const int ch0 = chunk * 64 + g * 8;
    if (ch0 < cout) {
        for (int64_t i = r0 + pl; i < r1; i += 32) {
            const int32_t j = nbr[(size_t)k * n_out + i];
            if (j < 0 || j >= n_in) continue;
            float gv[8];
            map_load8(g_hi, g_lo, (size_t)i * cout + ch0, gv);
            const float v = f[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v * gv[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64 && chunk * 64 + threadIdx.x < cout) {
        const int gg = threadIdx.x >> 3, e = threadIdx.x & 7;
        double a = 0;
        for (int q = 0; q < 32; ++q) a += red[q * 8 + gg][e];
        partial[((size_t)slice * ntaps + k) * cout + chunk * 64 + threadIdx.x] = (float)a;
    }
}

__global__ void __launch_bounds__(256) conv_cin1_wgrad_reduce_kernel(const float* __restrict__ partial, int nslices, int total,
                                                                     float* __restrict__ gw) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    double a = 0;
    for (int sl = 0; sl < nslices; ++sl) a += partial[(size_t)sl * total + t];
    gw[t] = (float)a;
}

inline int grid_for(int64_t threads) {
    int64_t g = (threads + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace agp_sparse
using namespace agp_sparse;

#define BF(p) ((bf16_t*)(p))
#define CBF(p) ((const bf16_t*)(p))

extern "C" int agp_sparse_conv_cin1_fwd(const float* f, int64_t n_in, const int32_t* nbr, int64_t n_out, int ntaps, const float* w,
                                        int cout, const float* scale, const float* shift, int relu, void* out_hi, void* out_lo,
                                        const int64_t* n_dev, void* stream) {
    if (!f || !nbr || !w || !out_hi || n_out <= 0 || cout % 8 || ntaps <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(conv_cin1_kernel, dim3(grid_for(n_out * (cout / 8))), dim3(256), 0, (hipStream_t)stream, f, nbr, n_in, n_out, ntaps, w,
               cout, scale, shift, relu, BF(out_hi), BF(out_lo), n_dev);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_sparse_conv0_fwd(const int64_t* keys, int64_t cap, const int64_t* n_dev, const float* f, int ksize, int stride,
                                    const float* w, int cout, const float* scale, const float* shift, int relu, void* out_hi,
                                    void* out_lo, const int64_t* seg_off, int prec, void* stream) {
    if (!keys || !f || !w || !out_hi || cap <= 0 || ksize < 1 || !(ksize & 1) || ksize > 7 || stride < 1) return AGP_E_BADARG;
    if (cout != 32 && cout != 64) return AGP_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (prec == AGP_PREC_F16 && !out_lo && ksize <= 5 && stride <= 64) {
        // one fp16 product: the matrix-pipe form (every other precision keeps the fp32 vector form below)
        const int KP = (ksize * ksize * ksize + 31) / 32 * 32;
        const int lds = (C0_ROWS + cout) * (KP * 2 + 16) + C0_WK * 12;
        const int tiles = (int)((cap + C0_ROWS - 1) / C0_ROWS);
        const int grid = tiles < 1024 ? tiles : 1024;
        if (cout == 32) {
            AGP_LAUNCH(conv0_mfma_kernel<32>, dim3(grid), dim3(256), lds, s, keys, cap, n_dev, f, ksize, stride, w, scale, shift, relu,
                       BF(out_hi), seg_off, KP);
        } else {
            static std::atomic<uint64_t> attr_done{0};
            if (!agp_lds_attr((const void*)conv0_mfma_kernel<64>, 96 * 1024, attr_done)) return AGP_E_LAUNCH;
            AGP_LAUNCH(conv0_mfma_kernel<64>, dim3(grid), dim3(256), lds, s, keys, cap, n_dev, f, ksize, stride, w, scale, shift, relu,
                       BF(out_hi), seg_off, KP);
        }
        AGP_CHECK_LAUNCH();
        return AGP_OK;
    }
    if (cout == 32) {
        AGP_LAUNCH(conv0_search_kernel<32>, dim3(grid_for(cap)), dim3(256), 0, s, keys, cap, n_dev, f, ksize, stride, w, scale, shift, relu,
                   BF(out_hi), BF(out_lo), seg_off);
    } else {
        AGP_LAUNCH(conv0_search_kernel<64>, dim3(grid_for(cap)), dim3(256), 0, s, keys, cap, n_dev, f, ksize, stride, w, scale, shift, relu,
                   BF(out_hi), BF(out_lo), seg_off);
    }
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_sparse_kernel_map(const int64_t* in_keys, int64_t n_in, const int64_t* out_keys, int64_t n_out,
                                     const int64_t* dkey, int ntaps, int32_t* nbr, const int64_t* n_dev, void* stream) {
    if (!in_keys || !out_keys || !dkey || !nbr || n_in < 0 || n_out <= 0 || ntaps <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(kernel_map_kernel, dim3(grid_for(n_out * ntaps)), dim3(256), 0, (hipStream_t)stream, in_keys, n_in, out_keys, n_out,
               dkey, ntaps, nbr, n_dev);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_sparse_kernel_map_grid(const int64_t* in_keys, int64_t n_in, const int64_t* out_keys, int64_t n_out, int ksize,
                                          int centered, int stride, int32_t* nbr, const int64_t* n_dev, const int64_t* n_in_dev,
                                          const int64_t* in_seg_off, void* stream) {
    if (!in_keys || !out_keys || !nbr || n_in < 0 || n_out <= 0 || ksize < 1 || ksize > 7 || stride < 1) return AGP_E_BADARG;
    if (centered && !(ksize & 1)) return AGP_E_BADARG;
    AGP_LAUNCH(kernel_map_grid_kernel, dim3(grid_for(n_out * ksize)), dim3(256), 0, (hipStream_t)stream, in_keys, n_in, out_keys,
               n_out, ksize, centered, stride, nbr, n_dev, n_in_dev, in_seg_off);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_seg_pool_fwd(const void* hi, const void* lo, const int64_t* seg_off, int nseg, int c, const float* p, float eps,
                                float* mean_out, float* gem_out, void* stream) {
    if (!hi || !seg_off || nseg <= 0 || c % 8 || (!mean_out && !gem_out) || (gem_out && !p)) return AGP_E_BADARG;
    AGP_LAUNCH(seg_pool_kernel, dim3(nseg, (c + 63) / 64), dim3(SEGP_T), 0, (hipStream_t)stream, CBF(hi), CBF(lo), seg_off, c, p, eps,
               mean_out, gem_out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_eca_scale_fwd(const float* mean, int nb, int c, const float* w, int k, float* out, void* stream) {
    if (!mean || !w || !out || nb <= 0 || c <= 0 || k <= 0 || !(k & 1)) return AGP_E_BADARG;
    AGP_LAUNCH(eca_kernel, dim3((nb * c + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, nb, c, w, k, out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_seg_affine_fwd(const void* y_hi, const void* y_lo, const int32_t* bidx, const float* scale, const float* add,
                                  const void* r_hi, const void* r_lo, int64_t n, int c, int relu, void* o_hi, void* o_lo,
                                  const int64_t* n_dev, void* stream) {
    if (!y_hi || !bidx || !o_hi || n <= 0 || c % 8) return AGP_E_BADARG;
    AGP_LAUNCH(seg_affine_kernel, dim3(grid_for(n * (c / 8))), dim3(256), 0, (hipStream_t)stream, CBF(y_hi), CBF(y_lo), bidx, scale,
               add, CBF(r_hi), CBF(r_lo), n, c, relu, BF(o_hi), BF(o_lo), n_dev);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_seg_dot_fwd(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, const int64_t* seg_off,
                               int nseg, int c, float* out, void* stream) {
    if (!a_hi || !seg_off || !out || nseg <= 0 || c % 8) return AGP_E_BADARG;
    AGP_LAUNCH(seg_dot_kernel, dim3(nseg, (c + 63) / 64), dim3(256), 0, (hipStream_t)stream, CBF(a_hi), CBF(a_lo), CBF(b_hi),
               CBF(b_lo), seg_off, c, out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_eca_scale_bwd(const float* mean, const float* scale, const float* gscale, const int64_t* seg_off, int nb, int c,
                                 const float* w, int k, float* add, float* gw, void* stream) {
    if (!mean || !scale || !gscale || !seg_off || !w || !add || !gw || nb <= 0 || c <= 0 || k <= 0 || !(k & 1)) return AGP_E_BADARG;
    AGP_LAUNCH(eca_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean, scale, gscale, seg_off, nb, c, w, k, add, gw);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_seg_pool_bwd(const void* x_hi, const void* x_lo, const int32_t* bidx, const int64_t* seg_off, const float* gmean,
                                const float* ggem, const float* gem_y, const float* p, float eps, const void* b_hi, const void* b_lo,
                                int64_t n, int c, void* o_hi, void* o_lo, float* gp, void* stream) {
    if (!bidx || !seg_off || !o_hi || n <= 0 || c % 8 || (ggem && (!x_hi || !gem_y || !p)) || (gp && !ggem)) return AGP_E_BADARG;
    AGP_LAUNCH(seg_pool_bwd_kernel, dim3(grid_for(n * (c / 8))), dim3(256), 0, (hipStream_t)stream, CBF(x_hi), CBF(x_lo), bidx,
               seg_off, gmean, ggem, gem_y, p, eps, CBF(b_hi), CBF(b_lo), n, c, BF(o_hi), BF(o_lo), gp);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_sparse_conv_cin1_wgrad(const float* f, int64_t n_in, const int32_t* nbr, int64_t n_out, int ntaps,
                                          const void* g_hi, const void* g_lo, int cout, float* gw, float* partial, int nslices,
                                          void* stream) {
    if (!f || !nbr || !g_hi || !gw || !partial || nslices < 1 || nslices > 4096 || n_out <= 0 || ntaps <= 0 || cout % 8)
        return AGP_E_BADARG;
    AGP_LAUNCH(conv_cin1_wgrad_kernel, dim3(ntaps, (cout + 63) / 64, nslices), dim3(256), 0, (hipStream_t)stream, f, nbr, n_in, n_out,
               CBF(g_hi), CBF(g_lo), cout, ntaps, partial);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(conv_cin1_wgrad_reduce_kernel, dim3((ntaps * cout + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, nslices,
               ntaps * cout, gw);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_sparse_tile_taps(const int32_t* nbr, int64_t n_out, int ntaps, int64_t zero_row, const int32_t* perm,
                                    const int64_t* n_dev, uint32_t* mask, int64_t ngran, void* stream) {
    if (!nbr || !mask || n_out <= 0 || ntaps < 1 || ntaps > 32 || ngran < (n_out + 127) / 128 || ngran >= (1ll << 30)) return AGP_E_BADARG;
    AGP_LAUNCH(tile_taps_kernel, dim3((unsigned)((ngran + 3) / 4)), dim3(256), 0, (hipStream_t)stream, nbr, n_out, ntaps, (int)zero_row, perm,
               n_dev, mask, (int)ngran);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
