// Global pooling over H*W: mean and GeM in ONE pass over the map (HBM-bound).
//   mean[n][c] = avg(x)                         fuse_block_toshallow.py:82
//   gem[n][c]  = (avg(max(x,eps)^p))^(1/p)       network_mm/image_pooling.py:16
// Stage 1: grid (splits, n); lanes run over 8-channel chunks (16 B per plane per lane),
// per-thread fp32 partial sums, a cross-thread LDS reduction per block, one partial row
// per block.  Stage 2 sums the partials in a fixed order (bit-reproducible; no atomics).
#include "common.hpp"

namespace agp_pool {

constexpr int POOL_TPB = 256;

__host__ __device__ inline int pool_splits(int n, int c, int h, int w) {
    // enough blocks to fill 256 CUs a few times over, at least ~64 pixels per block (a block's threads then make
    // only a few dependent trips to memory: these launches sit on the latency-bound tail of a forward)
    const int64_t pix = (int64_t)h * w;
    int s = (int)((pix + 63) / 64);
    const int want = (256 * 8 + n - 1) / n;
    if (s > want) s = want;
    if (s < 1) s = 1;
    return s;
}

__device__ __forceinline__ float powp(float v, float p, bool cube) {
    return cube ? v * v * v : __builtin_exp2f(p * __builtin_log2f(v));
}

__global__ __launch_bounds__(POOL_TPB) void pool_partial_kernel(
    const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, int h, int w, int c, int pad,
    const float* __restrict__ pptr, float eps, int splits, FastDiv dw, float* __restrict__ partial,
    int want_gem) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [POOL_TPB][16]
    const int groups = c / 8;
    const int ppb = POOL_TPB / groups > 0 ? POOL_TPB / groups : 1;   // pixels per block-iteration
    const int tid = threadIdx.x;
    const int g = tid % groups, pl = tid / groups;
    const int im = blockIdx.y, sp = blockIdx.x;
    const int hw = h * w;
    const int per = (hw + splits - 1) / splits;
    const int q0 = sp * per, q1 = min(hw, q0 + per);
    const int hp = h + 2 * pad, wp = w + 2 * pad;
    const float p = want_gem ? pptr[0] : 1.f;
    const bool cube = (p == 3.f);
    float sm[8], sg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sg[e] = 0.f; }
    if (pl < ppb && tid < ppb * groups) {
        // four pixels per trip: the loads of a trip are independent and issued together
        for (int qb = q0 + pl; qb < q1; qb += 4 * ppb) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = qb + u * ppb;
                const int qc = q < q1 ? q : q1 - 1;
                const uint32_t y = fdiv((uint32_t)qc, dw);
                const uint32_t x = (uint32_t)qc - y * dw.d;
                const size_t off = (((size_t)im * hp + y + pad) * wp + x + pad) * c + g * 8;
                map_load8(hi, lo, off, v[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (qb + u * ppb < q1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        sm[e] += v[u][e];
                        if (want_gem) sg[e] += powp(fmaxf(v[u][e], eps), p, cube);
                    }
                }
            }
        }
    }
    // reduce over the ppb pixel-lanes that share a channel group
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = sm[e]; red[tid * 16 + 8 + e] = sg[e]; }
    __syncthreads();
    if (tid < groups) {
        float am[8], ag[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { am[e] = 0.f; ag[e] = 0.f; }
        for (int k = 0; k < ppb; ++k) {
            const float* r = red + (k * groups + tid) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { am[e] += r[e]; ag[e] += r[8 + e]; }
        }
        float* o = partial + (((size_t)im * splits + sp) * 2) * c + tid * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = am[e]; o[c + e] = ag[e]; }
    }
}

__global__ void pool_final_kernel(const float* __restrict__ partial, int n, int c, int splits,
                                  float inv_hw, const float* __restrict__ pptr,
                                  float* __restrict__ mean_out, float* __restrict__ gem_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    const int im = t / c, ch = t % c;
    float sm = 0.f, sg = 0.f;
    for (int s0 = 0; s0 < splits; s0 += 8) {       // 16 independent loads per trip, summed in split order
        float a[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + u < splits ? s0 + u : splits - 1;
            const float* r = partial + (((size_t)im * splits + s) * 2) * c + ch;
            a[u] = r[0];
            g[u] = r[c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < splits) { sm += a[u]; sg += g[u]; }
    }
    if (mean_out) mean_out[t] = sm * inv_hw;
    if (gem_out) {
        const float p = pptr[0];
        gem_out[t] = __builtin_exp2f(__builtin_log2f(sg * inv_hw) / p);
    }
}

// Second stage of the conv-epilogue pooling (agp_conv_desc::pool_partial, igemm_kxr2.hip): partial[block][stat][c] over
// 64-row blocks of a raster in which image i owns blocks [i * bpi, (i + 1) * bpi).  One workgroup per (image, 64
// channels): 4 block-strided partial sums per channel (8 loads in flight each), combined in a fixed order that depends
// on image-relative positions only.
__global__ __launch_bounds__(256) void pool_from_conv_kernel(const float* __restrict__ partial, int n, int c, int bpi, float inv_hw,
                                                             const float* __restrict__ pptr, float* __restrict__ mean_out,
                                                             float* __restrict__ gem_out) {
    __shared__ float red[2][4][64];
    const int im = blockIdx.y, ch = blockIdx.x * 64 + (threadIdx.x & 63), k = threadIdx.x >> 6;
    const bool want_gem = gem_out != nullptr;
    float sm = 0.f, sg = 0.f;
    if (ch < c) {
        const float* base = partial + ((size_t)im * bpi * 2) * c + ch;
        for (int bb = k * 8; bb < bpi; bb += 32) {
            float a[8], g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = bb + u < bpi ? bb + u : bpi - 1;
                const float* r = base + (size_t)b * 2 * c;
                a[u] = r[0];
                g[u] = want_gem ? r[c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (bb + u < bpi) { sm += a[u]; sg += g[u]; }
        }
    }
    red[0][k][threadIdx.x & 63] = sm;
    red[1][k][threadIdx.x & 63] = sg;
    __syncthreads();
    if (k == 0 && ch < c) {
        const int l = threadIdx.x & 63;
        sm = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        sg = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
        if (mean_out) mean_out[(size_t)im * c + ch] = sm * inv_hw;
        if (want_gem) gem_out[(size_t)im * c + ch] = __builtin_exp2f(__builtin_log2f(sg * inv_hw) / pptr[0]);
    }
}

// Dense fp32 tensor with arbitrary strides: one wave per (n, c).
__global__ __launch_bounds__(256) void pool_f32_kernel(const float* __restrict__ x, int64_t sn,
                                                       int64_t sc, int64_t sh, int64_t sw, int n,
                                                       int c, int h, int w, const float* __restrict__ pptr,
                                                       float eps, float* __restrict__ mean_out,
                                                       float* __restrict__ gem_out) {
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wv >= n * c) return;
    const int im = wv / c, ch = wv % c;
    const float p = gem_out ? pptr[0] : 1.f;
    const bool cube = (p == 3.f);
    const float* base = x + im * sn + ch * sc;
    float sm = 0.f, sg = 0.f;
    const int hw = h * w;
    for (int q = lane; q < hw; q += 64) {
        const int y = q / w, xx = q - y * w;
        const float v = base[y * sh + xx * sw];
        sm += v;
        if (gem_out) sg += powp(fmaxf(v, eps), p, cube);
    }
    sm = wave_sum(sm);
    sg = wave_sum(sg);
    if (lane == 0) {
        if (mean_out) mean_out[wv] = sm / hw;
        if (gem_out) gem_out[wv] = __builtin_exp2f(__builtin_log2f(sg / hw) / p);
    }
}

// GeM backward on a dense fp32 tensor: one wave per (n, c).
//   y = S^(1/p), S = mean(c_i^p), c_i = max(x_i, eps)
//   dy/dx_i = y^(1-p) c_i^(p-1) [x_i >= eps] / HW
//   dy/dp   = y * ( -ln(S)/p^2 + mean(c_i^p ln c_i) / (p S) )
__global__ __launch_bounds__(256) void gem_f32_bwd_kernel(const float* __restrict__ x, int64_t sn,
                                                          int64_t sc, int64_t sh, int64_t sw, int n,
                                                          int c, int h, int w, const float* __restrict__ pptr,
                                                          float eps, const float* __restrict__ y,
                                                          const float* __restrict__ gy, float* __restrict__ gx,
                                                          float* __restrict__ gp) {
    const int wv0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const bool livew = wv0 < n * c;                      // (no early return: every thread takes part in the ordered sum of dL/dp)
    const int wv = livew ? wv0 : n * c - 1;
    const int im = wv / c, ch = wv % c;
    const float p = pptr[0];
    const float* base = x + im * sn + ch * sc;
    float* gbase = gx ? gx + im * sn + ch * sc : nullptr;
    const int hw = h * w;
    const float yy = y[wv], g = gy[wv];
    const float coef = g * __builtin_exp2f((1.f - p) * __builtin_log2f(yy)) / hw;
    float t = 0.f;
    for (int q = lane; q < hw; q += 64) {
        const int yq = q / w, xx = q - yq * w;
        const float v = base[yq * sh + xx * sw];
        const float cv = fmaxf(v, eps);
        const float lg = __builtin_log2f(cv);
        const float cp1 = __builtin_exp2f((p - 1.f) * lg);
        if (gbase && livew) gbase[yq * sh + xx * sw] = (v >= eps) ? coef * cp1 : 0.f;
        t += cp1 * cv * lg * 0.6931471805599453f;
    }
    if (gp) {
        t = wave_sum(t);
        const float S = __builtin_exp2f(p * __builtin_log2f(yy));
        const float dydp = yy * (-__logf(S) / (p * p) + (t / hw) / (p * S));
        // the wave's plane contributes g * dydp: one lane carries it into the block's ordered sum
        agp_grid_sum_ordered(agp_block_sum_ordered((lane == 0 && livew) ? g * dydp : 0.f), gp);
    }
}

}  // namespace agp_pool
using namespace agp_pool;

extern "C" int64_t agp_pool_workspace_floats(int n, int c, int h, int w) {
    return (int64_t)n * pool_splits(n, c, h, w) * 2 * c;
}

extern "C" int agp_pool_fwd(const void* hi, const void* lo, int n, int h, int w, int c, int pad,
                            const float* p, float eps, float* mean_out, float* gem_out,
                            float* partial, void* stream) {
    if (!hi || !partial || c % 8 || c / 8 > POOL_TPB || n <= 0) return AGP_E_BADARG;
    if (gem_out && !p) return AGP_E_BADARG;
    const int splits = pool_splits(n, c, h, w);
    AGP_LAUNCH(pool_partial_kernel, dim3(splits, n), dim3(POOL_TPB), POOL_TPB * 16 * 4,
                       (hipStream_t)stream, (const bf16_t*)hi, (const bf16_t*)lo, h, w, c, pad, p, eps,
                       splits, make_fastdiv((uint32_t)w), partial, gem_out ? 1 : 0);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(pool_final_kernel, dim3((n * c + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       partial, n, c, splits, 1.f / (float)(h * w), p, mean_out, gem_out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_pool_from_conv(const float* partial, int n, int h, int w, int c, const float* p, float* mean_out,
                                  float* gem_out, void* stream) {
    if (!partial || n <= 0 || c <= 0 || h <= 0 || w <= 0 || (gem_out && !p) || (!mean_out && !gem_out)) return AGP_E_BADARG;
    AGP_LAUNCH(pool_from_conv_kernel, dim3((c + 63) / 64, n), dim3(256), 0, (hipStream_t)stream, partial, n, c,
               (h * (w + 2) + 63) / 64, 1.f / (float)(h * w), p, mean_out, gem_out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_pool_f32_fwd(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n,
                                int c, int h, int w, const float* p, float eps, float* mean_out,
                                float* gem_out, float* partial, void* stream) {
    (void)partial;
    if (!x || n <= 0 || (gem_out && !p)) return AGP_E_BADARG;
    const int64_t waves = (int64_t)n * c;
    AGP_LAUNCH(pool_f32_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, x, sn, sc, sh, sw, n, c, h, w, p, eps, mean_out, gem_out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_gem_f32_bwd(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n,
                               int c, int h, int w, const float* p, float eps, const float* y,
                               const float* gy, float* gx, float* gp, void* stream) {
    if (!x || !p || !y || !gy || n <= 0) return AGP_E_BADARG;
    const int64_t waves = (int64_t)n * c;
    AGP_LAUNCH(gem_f32_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, x, sn, sc, sh, sw, n, c, h, w, p, eps, y, gy, gx, gp);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
