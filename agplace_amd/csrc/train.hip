// Training-path kernels on halo-padded NHWC split-bf16 maps: train-mode BatchNorm (statistics, apply,
// backward), ReLU/residual/max-pool/global-pool backward and zero-upsampling for stride-2 data
// gradients (the weight gradient reads the NHWC maps directly: wgrad_tr.hip).
// All HBM-bound: one thread moves 8 channels (16 B per plane) of one pixel.
#include <algorithm>

#include "common.hpp"

namespace agp_train {

// geometry + fast 32-bit division by the channel groups, the width and the height: every kernel here turns a
// linear (image, y, x, 8-channel group) index into an address per element, and 64-bit divisions (five per element)
// made these "HBM-bound" kernels ALU-bound at ~2.3 TB/s
struct MapGeo { int n, h, w, c, pad; FastDiv dg, dw, dh; };
static inline MapGeo geo_of(int n, int h, int w, int c, int pad) {
    MapGeo g;
    g.n = n; g.h = h; g.w = w; g.c = c; g.pad = pad;
    g.dg = make_fastdiv((uint32_t)(c / 8 > 0 ? c / 8 : 1));
    g.dw = make_fastdiv((uint32_t)(w > 0 ? w : 1));
    g.dh = make_fastdiv((uint32_t)(h > 0 ? h : 1));
    return g;
}
static inline bool geo_fits(int n, int h, int w, int c) { return (int64_t)n * h * w * (c / 8 > 0 ? c / 8 : 1) < (1ll << 31); }

__device__ __forceinline__ void load8(const bf16_t* hi, const bf16_t* lo, size_t off, float* v) {
    map_load8(hi, lo, off, v);
}
__device__ __forceinline__ void store8(bf16_t* hi, bf16_t* lo, size_t off, const float* v) {
    map_store8(hi, lo, off, v);
}
// interior (pixel, 8-channel group) iteration
#define AGP_FOR_MAP(geo)                                                                                  \
    const int groups_ = (geo).c / 8;                                                                      \
    const uint32_t total_ = (uint32_t)(geo).n * (geo).h * (geo).w * groups_;       /* < 2^31: geo_fits */ \
    const int hp_ = (geo).h + 2 * (geo).pad, wp_ = (geo).w + 2 * (geo).pad;                               \
    for (uint32_t t_ = blockIdx.x * blockDim.x + threadIdx.x; t_ < total_; t_ += gridDim.x * blockDim.x)

#define AGP_MAP_INDEX(geo)                                                                                \
    const uint32_t r1_ = fdiv(t_, (geo).dg);                                                              \
    const int g = (int)(t_ - r1_ * (uint32_t)groups_);                                                    \
    const uint32_t r2_ = fdiv(r1_, (geo).dw);                                                             \
    const int px = (int)(r1_ - r2_ * (uint32_t)(geo).w);                                                  \
    const int im = (int)fdiv(r2_, (geo).dh);                                                              \
    const int py = (int)(r2_ - (uint32_t)im * (uint32_t)(geo).h);                                         \
    const size_t off = (((size_t)im * hp_ + py + (geo).pad) * wp_ + px + (geo).pad) * (geo).c + g * 8;    \
    (void)im; (void)px; (void)py;

// ---- per-channel reductions: partial[block][2][C] then a finalize kernel (fixed order, fp64)
//   mode 0: (sum z, sum z^2)                      -> BN statistics
//   mode 1: (sum g, sum g*zhat), g = gy*[y>0]?    -> BN backward  (zhat from mean/rstd)
__global__ __launch_bounds__(256) void chan_reduce_kernel(MapGeo geo, const bf16_t* a_hi, const bf16_t* a_lo,
                                                          const bf16_t* b_hi, const bf16_t* b_lo,
                                                          const bf16_t* y_hi, const bf16_t* y_lo,
                                                          const float* mean, const float* rstd, int mode, int relu,
                                                          float* partial) {
    extern __shared__ __attribute__((aligned(16))) float red[];    // [256][16]
    const int groups = geo.c / 8;
    const int tid = threadIdx.x;
    const int g = tid % groups, pl = tid / groups;
    const int ppb = 256 / groups;
    const uint32_t npix = (uint32_t)geo.n * geo.h * geo.w;
    const uint32_t per = (npix + gridDim.x - 1) / gridDim.x;
    const uint32_t q0 = min(npix, blockIdx.x * per), q1 = min(npix, q0 + per);
    const int hp = geo.h + 2 * geo.pad, wp = geo.w + 2 * geo.pad;
    float s1[8], s2[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s1[e] = 0.f; s2[e] = 0.f;
        mu[e] = (mode == 1) ? mean[g * 8 + e] : 0.f;
        rs[e] = (mode == 1) ? rstd[g * 8 + e] : 0.f;
    }
    if (pl < ppb) {
#pragma unroll 4
        for (uint32_t q = q0 + pl; q < q1; q += ppb) {
            const uint32_t r = fdiv(q, geo.dw);
            const int px = (int)(q - r * (uint32_t)geo.w);
            const int im = (int)fdiv(r, geo.dh);
            const int py = (int)(r - (uint32_t)im * (uint32_t)geo.h);
            const size_t off = (((size_t)im * hp + py + geo.pad) * wp + px + geo.pad) * geo.c + g * 8;
            float a[8];
            load8(a_hi, a_lo, off, a);
            if (mode == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += a[e]; s2[e] += a[e] * a[e]; }
            } else {
                float gg[8];
                load8(b_hi, b_lo, off, gg);
                if (relu) {
                    const unsigned pm = pos_mask8(y_hi, off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (!((pm >> e) & 1u)) gg[e] = 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += gg[e]; s2[e] += gg[e] * (a[e] - mu[e]) * rs[e]; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    if (tid < groups) {
        float t1[8], t2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
        for (int k = 0; k < ppb; ++k) {
            const float* r = red + (k * groups + tid) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { t1[e] += r[e]; t2[e] += r[8 + e]; }
        }
        float* o = partial + (size_t)blockIdx.x * 2 * geo.c + tid * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = t1[e]; o[geo.c + e] = t2[e]; }
    }
}

// Sum the per-block partials of one channel: one WORKGROUP (256 threads) per channel, threads stride over the blocks, fp64
// accumulation in a fixed order (deterministic), butterfly reduction per wave, the four wave totals added in wave order.
// (One wave per channel made a finalize of 2 900 tiles eleven dependent round trips: 12 us, 66 of them per training step.)
// Every thread returns the totals.
__device__ __forceinline__ void partial_sums(const float* partial, int nblocks, int c, int ch, double& s1, double& s2) {
    __shared__ double wsum[2][4];
    const int t = threadIdx.x;
    double a = 0, b = 0;
    for (int k0 = t; k0 < nblocks; k0 += 1024) {      // eight independent loads per trip, summed in block order
        float va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 256 * u < nblocks ? k0 + 256 * u : nblocks - 1;
            va[u] = partial[(size_t)k * 2 * c + ch];
            vb[u] = partial[(size_t)k * 2 * c + c + ch];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + 256 * u < nblocks) { a += va[u]; b += vb[u]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if ((t & 63) == 0) { wsum[0][t >> 6] = a; wsum[1][t >> 6] = b; }
    __syncthreads();
    s1 = ((wsum[0][0] + wsum[0][1]) + wsum[0][2]) + wsum[0][3];
    s2 = ((wsum[1][0] + wsum[1][1]) + wsum[1][2]) + wsum[1][3];
}

// BN statistics finalize: mean, rstd (biased variance) + running-stat update (unbiased variance)
__global__ void bn_stats_final_kernel(const float* partial, int nblocks, int c, double count, float eps, float momentum,
                                      float* mean, float* rstd, float* running_mean, float* running_var,
                                      const float* gamma, const float* beta, float* scale, float* shift) {
    const int ch = blockIdx.x;                      // one workgroup of 256 threads per channel
    double s1, s2;
    partial_sums(partial, nblocks, c, ch, s1, s2);
    if (threadIdx.x) return;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0) var = 0;
    mean[ch] = (float)m;
    const double rs = 1.0 / sqrt(var + (double)eps);
    rstd[ch] = (float)rs;
    if (scale) {
        const double sc = (gamma ? (double)gamma[ch] : 1.0) * rs;
        scale[ch] = (float)sc;
        shift[ch] = (float)((beta ? (double)beta[ch] : 0.0) - m * sc);
    }
    if (running_mean) {
        const double unb = count > 1 ? var * count / (count - 1) : var;
        running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * m);
        running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
    }
}

__global__ void sum2_final_kernel(const float* partial, int nblocks, int c, float* out1, float* out2) {
    const int ch = blockIdx.x;                      // one workgroup of 256 threads per channel
    double s1, s2;
    partial_sums(partial, nblocks, c, ch, s1, s2);
    if (threadIdx.x) return;
    if (out1) out1[ch] = (float)s1;
    if (out2) out2[ch] = (float)s2;
}

// ---- synchronised BatchNorm (statistics over every rank's samples; reference model/sync_batchnorm/batchnorm.py:121-166): the
// local sums leave as fp64 [2c + 1] = (sum, sum of squares, count) for ONE all-reduce per layer, the statistics come from
// the reduced vector; the backward exchanges (sum g, sum g * zhat) the same way.
__global__ void sums_f64_kernel(const float* partial, int nblocks, int c, double count, double* sums, float* out1, float* out2) {
    const int ch = blockIdx.x;                      // one workgroup of 256 threads per channel
    double s1, s2;
    partial_sums(partial, nblocks, c, ch, s1, s2);
    if (threadIdx.x) return;
    sums[ch] = s1;
    sums[c + ch] = s2;
    if (ch == 0 && count >= 0) sums[2 * c] = count;
    if (out1) out1[ch] = (float)s1;
    if (out2) out2[ch] = (float)s2;
}

__global__ void bn_stats_from_sums_kernel(const double* sums, int c, float eps, float momentum, float* mean, float* rstd,
                                          float* running_mean, float* running_var, const float* gamma, const float* beta,
                                          float* scale, float* shift) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double count = sums[2 * c];
    const double m = sums[ch] / count;
    double var = sums[c + ch] / count - m * m;
    if (var < 0) var = 0;
    mean[ch] = (float)m;
    const double rs = 1.0 / sqrt(var + (double)eps);
    rstd[ch] = (float)rs;
    if (scale) {
        const double sc = (gamma ? (double)gamma[ch] : 1.0) * rs;
        scale[ch] = (float)sc;
        shift[ch] = (float)((beta ? (double)beta[ch] : 0.0) - m * sc);
    }
    if (running_mean) {
        const double unb = count > 1 ? var * count / (count - 1) : var;
        running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * m);
        running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
    }
}

// the reduced backward sums as the fp32 operands of bn_bwd_apply_kernel (+ 1 / global count)
__global__ void bn_bwd_coeffs_kernel(const double* sums, const double* count, int c, float* sum_g, float* sum_gz, float* inv_count) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    sum_g[ch] = (float)sums[ch];
    sum_gz[ch] = (float)sums[c + ch];
    if (ch == 0) inv_count[0] = (float)(1.0 / count[0]);
}

// eval-mode BatchNorm (running statistics): the per-channel constants of the forward affine and of the backward
__global__ void bn_frozen_coeffs_kernel(const float* running_mean, const float* running_var, const float* gamma,
                                        const float* beta, int c, float eps, float* mean, float* rstd, float* scale,
                                        float* shift) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double m = running_mean[ch];
    const double rs = 1.0 / sqrt((double)running_var[ch] + (double)eps);
    const double sc = (gamma ? (double)gamma[ch] : 1.0) * rs;
    mean[ch] = (float)m;
    rstd[ch] = (float)rs;
    scale[ch] = (float)sc;
    shift[ch] = (float)((beta ? (double)beta[ch] : 0.0) - m * sc);
}

// 16-byte NON-TEMPORAL load of a plane that the kernel reads exactly once (the BatchNorm apply / backward passes stream 150-2000 MB
// per launch through a 32 MB L2): the training step 15.62 -> 15.51 ms in an A/B of two libraries; back-to-back replays of ONE launch
// (tools/bn_bench.py) are slower with it -- they were living off the previous replay's lines, which the step never does.
__device__ __forceinline__ u32x4 ld_once(const bf16_t* p) { return __builtin_nontemporal_load((const u32x4*)p); }
// out = relu?(a*scale[c] + shift[c] + r)
// o_h16 (optional): the same values once more as ONE fp16 plane -- the operand plane of the one-pass weight gradient
// (wgrad_tr.hip: wgrad_f16_kernel), which the conv that consumes this map as its input reads instead of the bf16 pair.
__global__ void affine_kernel(MapGeo geo, const bf16_t* a_hi, const bf16_t* a_lo, const float* scale, const float* shift,
                              const bf16_t* r_hi, const bf16_t* r_lo, int relu, bf16_t* o_hi, bf16_t* o_lo, bf16_t* o_h16) {
    // The channel group of a thread is the same in every iteration when the grid stride is a multiple of the groups
    // (256 threads, groups a power of two <= 32): its 16 coefficients are loaded once, not per element.
    const int groups0 = geo.c / 8;
    const bool fixed_g = ((gridDim.x * blockDim.x) % groups0) == 0;
    const int g0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) % groups0);
    float sc0[8], sh0[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc0[e] = scale ? scale[g0 * 8 + e] : 1.f; sh0[e] = shift ? shift[g0 * 8 + e] : 0.f; }
    if (fixed_g && (!r_hi || r_lo)) {      // (o_lo == nullptr: the output is ONE fp16 plane, store8 writes it)
        // the common case (bf16-pair planes; `a` may be ONE fp16 plane -- the z of a unit whose forward conv ran as one fp16
        // product, Options.train_precision = 16): two items per trip, their loads issued before the first store (see bn_bwd_apply_kernel)
        const int groups_ = geo.c / 8;
        const uint32_t total_ = (uint32_t)geo.n * geo.h * geo.w * groups_;
        const int hp_ = geo.h + 2 * geo.pad, wp_ = geo.w + 2 * geo.pad;
        const uint32_t S = gridDim.x * blockDim.x;
        auto index = [&](uint32_t t_) -> size_t {
            const uint32_t r1_ = fdiv(t_, geo.dg);
            const int g = (int)(t_ - r1_ * (uint32_t)groups_);
            const uint32_t r2_ = fdiv(r1_, geo.dw);
            const int px = (int)(r1_ - r2_ * (uint32_t)geo.w);
            const int im = (int)fdiv(r2_, geo.dh);
            const int py = (int)(r2_ - (uint32_t)im * (uint32_t)geo.h);
            return (((size_t)im * hp_ + py + geo.pad) * wp_ + px + geo.pad) * geo.c + g * 8;
        };
        for (uint32_t t_ = blockIdx.x * blockDim.x + threadIdx.x; t_ < total_; t_ += 2 * S) {
            const bool two = t_ + S < total_;
            const size_t off[2] = {index(t_), index(two ? t_ + S : t_)};
            u32x4 ah[2], al[2], rh[2], rl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ah[u] = ld_once(a_hi + off[u]);
                if (a_lo) al[u] = ld_once(a_lo + off[u]);
                if (r_hi) { rh[u] = ld_once(r_hi + off[u]); rl[u] = ld_once(r_lo + off[u]); }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                float v[8], l[8];
                if (a_lo) {
                    unpack8(ah[u], v); unpack8(al[u], l);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += l[e];
                } else {
                    unpack8_h(ah[u], v);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc0[e] + sh0[e];
                if (r_hi) {
                    float r[8], r2[8];
                    unpack8(rh[u], r); unpack8(rl[u], r2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e] + r2[e];
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                store8(o_hi, o_lo, off[u], v);
                if (o_h16) *(u32x4*)(o_h16 + off[u]) = pack8_h(v);
            }
        }
        return;
    }
    AGP_FOR_MAP(geo) {
        AGP_MAP_INDEX(geo)
        float v[8];
        load8(a_hi, a_lo, off, v);
        if (fixed_g) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc0[e] + sh0[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * (scale ? scale[g * 8 + e] : 1.f) + (shift ? shift[g * 8 + e] : 0.f);
        }
        if (r_hi) {
            float r[8];
            load8(r_hi, r_lo, off, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        store8(o_hi, o_lo, off, v);
        if (o_h16) *(u32x4*)(o_h16 + off) = pack8_h(v);
    }
}

// Per-channel maximum of |value| over a kernel's elements, as fp32 bit patterns (order-preserving for non-negative floats, so an
// integer atomicMax is an exact and order-independent maximum): every thread folds its own eight channels' running maxima into
// an LDS table, the table goes to global memory with one atomic per channel and workgroup.
__device__ __forceinline__ void absmax_flush(uint32_t* gmax, int c, int g, const float* mx) {
    __shared__ uint32_t lm[2048];
    if (c <= 2048) {
        for (int i = threadIdx.x; i < c; i += blockDim.x) lm[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicMax(&lm[g * 8 + e], __builtin_bit_cast(uint32_t, mx[e]));
        __syncthreads();
        // the maximum only grows: a relaxed read first, the atomic only for a value that would raise it (a stale read costs one
        // redundant atomic, never a lost maximum).  Without the test a launch sent c atomics PER WORKGROUP at the same c words:
        // up to 8192 x 64 of them serialised in L2 (pool_bn_bwd_apply 237 -> 358 us with them, profiles/README.md round 5)
        for (int i = threadIdx.x; i < c; i += blockDim.x)
            if (lm[i] > __atomic_load_n(&gmax[i], __ATOMIC_RELAXED)) atomicMax(&gmax[i], lm[i]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t b = __builtin_bit_cast(uint32_t, mx[e]);
            if (b > __atomic_load_n(&gmax[g * 8 + e], __ATOMIC_RELAXED)) atomicMax(&gmax[g * 8 + e], b);
        }
    }
}

// BN backward apply: g = gy*[y>0]? ; gz = coefa[c]*(g - sb[c] - zhat*sg[c]) with zhat = (z-mean)*rstd,
// coefa = gamma*rstd, sb = sum(g)/N, sg = sum(g*zhat)/N.  Optionally writes g itself (residual branch).
__global__ void bn_bwd_apply_kernel(MapGeo geo, const bf16_t* z_hi, const bf16_t* z_lo, const bf16_t* gy_hi,
                                    const bf16_t* gy_lo, const bf16_t* y_hi, const bf16_t* y_lo, const float* mean,
                                    const float* rstd, const float* gamma, const float* sum_g, const float* sum_gz,
                                    float inv_count_arg, const float* inv_count_dev, int relu, bf16_t* gz_hi, bf16_t* gz_lo,
                                    bf16_t* gr_hi, bf16_t* gr_lo, uint32_t* gz_absmax) {
    const float inv_count = inv_count_dev ? inv_count_dev[0] : inv_count_arg;       // synchronised BatchNorm: 1 / global count
    // gz = A*g + B*z + C per channel; a thread's channel group is loop-invariant (see affine_kernel): 24 coefficients
    // once instead of 40 scalar loads per element
    const int groups0 = geo.c / 8;
    const bool fixed_g = ((gridDim.x * blockDim.x) % groups0) == 0;
    const int g0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) % groups0);
    float cA[8], cB[8], cC[8], mx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = g0 * 8 + e;
        const float gr = (gamma ? gamma[ch] : 1.f) * rstd[ch];
        mx[e] = 0.f;
        cA[e] = gr;
        cB[e] = -gr * rstd[ch] * sum_gz[ch] * inv_count;
        cC[e] = -gr * sum_g[ch] * inv_count - cB[e] * mean[ch];
    }
    if (fixed_g && gy_lo && gz_lo && (!gr_hi || gr_lo)) {       // (z: a bf16 pair, or ONE fp16 plane -- z_lo == nullptr)
        // the common case (bf16-pair planes, a thread's channel group fixed): TWO items per trip, all ten 16-byte loads of the
        // pair issued before the first store (one item per trip left a thread with five loads in flight: 3.8 TB/s)
        const int groups_ = geo.c / 8;
        const uint32_t total_ = (uint32_t)geo.n * geo.h * geo.w * groups_;
        const int hp_ = geo.h + 2 * geo.pad, wp_ = geo.w + 2 * geo.pad;
        const uint32_t S = gridDim.x * blockDim.x;
        auto index = [&](uint32_t t_) -> size_t {
            const uint32_t r1_ = fdiv(t_, geo.dg);
            const int g = (int)(t_ - r1_ * (uint32_t)groups_);
            const uint32_t r2_ = fdiv(r1_, geo.dw);
            const int px = (int)(r1_ - r2_ * (uint32_t)geo.w);
            const int im = (int)fdiv(r2_, geo.dh);
            const int py = (int)(r2_ - (uint32_t)im * (uint32_t)geo.h);
            return (((size_t)im * hp_ + py + geo.pad) * wp_ + px + geo.pad) * geo.c + g * 8;
        };
        for (uint32_t t_ = blockIdx.x * blockDim.x + threadIdx.x; t_ < total_; t_ += 2 * S) {
            const bool two = t_ + S < total_;
            const size_t off[2] = {index(t_), index(two ? t_ + S : t_)};
            u32x4 zh[2], zl[2], gh[2], gl[2], yh[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                zh[u] = ld_once(z_hi + off[u]);
                if (z_lo) zl[u] = ld_once(z_lo + off[u]);
                gh[u] = ld_once(gy_hi + off[u]); gl[u] = ld_once(gy_lo + off[u]);
                if (relu) yh[u] = ld_once(y_hi + off[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                float z[8], zl_[8], gg[8], gl_[8], o[8];
                if (z_lo) {
                    unpack8(zh[u], z); unpack8(zl[u], zl_);
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] += zl_[e];
                } else {
                    unpack8_h(zh[u], z);
                }
                unpack8(gh[u], gg); unpack8(gl[u], gl_);
#pragma unroll
                for (int e = 0; e < 8; ++e) gg[e] += gl_[e];
                if (relu) {
                    const unsigned pm = pos_mask8_raw(yh[u]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (!((pm >> e) & 1u)) gg[e] = 0.f;
                }
                if (gr_hi) store8(gr_hi, gr_lo, off[u], gg);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = cA[e] * gg[e] + (cB[e] * z[e] + cC[e]);
                    mx[e] = fmaxf(mx[e], fabsf(o[e]));
                }
                store8(gz_hi, gz_lo, off[u], o);
            }
        }
        if (gz_absmax) absmax_flush(gz_absmax, geo.c, g0, mx);
        return;
    }
    AGP_FOR_MAP(geo) {
        AGP_MAP_INDEX(geo)
        float z[8], gg[8];
        load8(z_hi, z_lo, off, z);
        load8(gy_hi, gy_lo, off, gg);
        if (relu) {
            const unsigned pm = pos_mask8(y_hi, off);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (!((pm >> e) & 1u)) gg[e] = 0.f;
        }
        if (gr_hi) store8(gr_hi, gr_lo, off, gg);
        float o[8];
        if (fixed_g) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = cA[e] * gg[e] + (cB[e] * z[e] + cC[e]);
                mx[e] = fmaxf(mx[e], fabsf(o[e]));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = g * 8 + e;
                const float zh = (z[e] - mean[ch]) * rstd[ch];
                o[e] = (gamma ? gamma[ch] : 1.f) * rstd[ch] * (gg[e] - sum_g[ch] * inv_count - zh * sum_gz[ch] * inv_count);
                if (gz_absmax) atomicMax(&gz_absmax[ch], __builtin_bit_cast(uint32_t, fabsf(o[e])));      // (rare grid shapes only)
            }
        }
        store8(gz_hi, gz_lo, off, o);
    }
    if (gz_absmax && fixed_g) absmax_flush(gz_absmax, geo.c, g0, mx);       // (uniform branch: every thread of the block)
}

// out = a (*[y>0]) + b
__global__ void add_kernel(MapGeo geo, const bf16_t* a_hi, const bf16_t* a_lo, const bf16_t* b_hi, const bf16_t* b_lo,
                           const bf16_t* y_hi, const bf16_t* y_lo, bf16_t* o_hi, bf16_t* o_lo) {
    AGP_FOR_MAP(geo) {
        AGP_MAP_INDEX(geo)
        float a[8];
        load8(a_hi, a_lo, off, a);
        if (y_hi) {
            float yy[8];
            load8(y_hi, y_lo, off, yy);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (!(yy[e] > 0.f)) a[e] = 0.f;
        }
        if (b_hi) {
            float b[8];
            load8(b_hi, b_lo, off, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
        }
        store8(o_hi, o_lo, off, a);
    }
}

// U[n][2*oy][2*ox] = g[n][oy][ox], zero elsewhere (U interior hu x wu must be pre-zeroed by the caller? no:
// every interior pixel of U is written here).
__global__ void upsample2_kernel(MapGeo gu, const bf16_t* g_hi, const bf16_t* g_lo, int ho, int wo, int gpad,
                                 bf16_t* u_hi, bf16_t* u_lo) {
    AGP_FOR_MAP(gu) {
        AGP_MAP_INDEX(gu)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (!(py & 1) && !(px & 1) && (py >> 1) < ho && (px >> 1) < wo) {
            const size_t so = (((size_t)im * (ho + 2 * gpad) + (py >> 1) + gpad) * (wo + 2 * gpad) + (px >> 1) + gpad) * gu.c + g * 8;
            load8(g_hi, g_lo, so, v);
        }
        store8(u_hi, u_lo, off, v);
    }
}

// max-pool 3x3/2 pad 1 backward from the forward's argmax: input pixel i receives gy[o] from every window o
// (at most four) whose recorded first-maximum position is i.
__global__ void maxpool_bwd_kernel(MapGeo gin, const uint8_t* __restrict__ idx, const bf16_t* gy_hi, const bf16_t* gy_lo, int ho,
                                   int wo, int opad, bf16_t* gx_hi, bf16_t* gx_lo) {
    AGP_FOR_MAP(gin) {
        AGP_MAP_INDEX(gin)
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        // windows: o with 2o-1 <= i <= 2o+1  ->  o in {floor(i/2), floor((i+1)/2)}
        const int oy0 = py >> 1, oy1 = (py + 1) >> 1, ox0 = px >> 1, ox1 = (px + 1) >> 1;
        for (int oy = oy0; oy <= oy1; ++oy) {
            if (oy >= ho) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                if (ox >= wo) continue;
                const int wpos = 3 * (py - (2 * oy - 1)) + (px - (2 * ox - 1));
                const u32x2 pk = *(const u32x2*)(idx + ((((size_t)im * ho + oy) * wo + ox) * gin.c + g * 8));
                const size_t oo = (((size_t)im * (ho + 2 * opad) + oy + opad) * (wo + 2 * opad) + ox + opad) * gin.c + g * 8;
                float gv[8];
                load8(gy_hi, gy_lo, oo, gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int id = (int)((pk[e >> 2] >> (8 * (e & 3))) & 0xffu);
                    if (id == wpos) acc[e] += gv[e];
                }
            }
        }
        store8(gx_hi, gx_lo, off, acc);
    }
}

// ---- BatchNorm apply + ReLU + max-pool 3x3/2 pad 1 in one pass over the conv output z (the ResNet stem in training): the
// full-size activation y = relu(z*scale + shift) is never stored -- the pooled map and the argmax are all the forward needs,
// and the backward recomputes the ReLU mask from z with the SAME fp32 fma (agp_maxpool_bn_bwd with scale / shift).
__device__ __forceinline__ float stem_act(float z, float sc, float sh) { return fmaxf(__builtin_fmaf(z, sc, sh), 0.f); }

__global__ void affine_maxpool_kernel(MapGeo gin, const bf16_t* __restrict__ z_hi, const bf16_t* __restrict__ z_lo,
                                      const float* __restrict__ scale, const float* __restrict__ shift, bf16_t* __restrict__ o_hi,
                                      bf16_t* __restrict__ o_lo, int ho, int wo, int opad, uint8_t* __restrict__ idx,
                                      bf16_t* __restrict__ o_h16) {
    const int groups = gin.c / 8;
    const int64_t total = (int64_t)gin.n * ho * wo * groups;
    const int hip_ = gin.h + 2 * gin.pad, wip = gin.w + 2 * gin.pad;
    const int hop = ho + 2 * opad, wop = wo + 2 * opad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t;
        const int g = r % groups; r /= groups;
        const int ox = r % wo; r /= wo;
        const int oy = r % ho;
        const int im = r / ho;
        float sc[8], sh[8], best[8];
        uint8_t bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = scale[g * 8 + e]; sh[e] = shift[g * 8 + e]; best[e] = 0.f; bi[e] = 4; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;           // pixel of the unpadded map; outside: nothing
                if (iy < 0 || ix < 0 || iy >= gin.h || ix >= gin.w) continue;
                const size_t off = (((size_t)im * hip_ + iy + gin.pad) * wip + ix + gin.pad) * gin.c + g * 8;
                float v[8];
                load8(z_hi, z_lo, off, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = stem_act(v[e], sc[e], sh[e]);
                    if (a > best[e]) { best[e] = a; bi[e] = (uint8_t)(3 * ky + kx); }
                }
            }
        const size_t ooff = (((size_t)im * hop + oy + opad) * wop + ox + opad) * gin.c + g * 8;
        store8(o_hi, o_lo, ooff, best);
        if (o_h16) *(u32x4*)(o_h16 + ooff) = pack8_h(best);
        u32x2 pk = {(uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24),
                    (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24)};
        *(u32x2*)(idx + ((((size_t)im * ho + oy) * wo + ox) * gin.c + g * 8)) = pk;
    }
}

// ---- max-pool backward FUSED with the BatchNorm backward of the unit below it (the ResNet stem: conv -> BN -> ReLU -> max-pool):
// the gradient at the BN unit's output is never stored -- each element gathers it from the (at most four) pooled windows that
// recorded it as their maximum, once for the channel sums and once for gz.  29 -> 18.5 bytes per element of the stem map.
__device__ __forceinline__ void pool_gather8(const MapGeo& gin, const uint8_t* __restrict__ idx, const bf16_t* gy_hi,
                                             const bf16_t* gy_lo, int ho, int wo, int opad, int im, int py, int px, int g,
                                             float* acc) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int oy0 = py >> 1, oy1 = (py + 1) >> 1, ox0 = px >> 1, ox1 = (px + 1) >> 1;
    for (int oy = oy0; oy <= oy1; ++oy) {
        if (oy >= ho) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            if (ox >= wo) continue;
            const int wpos = 3 * (py - (2 * oy - 1)) + (px - (2 * ox - 1));
            const u32x2 pk = *(const u32x2*)(idx + ((((size_t)im * ho + oy) * wo + ox) * gin.c + g * 8));
            const size_t oo = (((size_t)im * (ho + 2 * opad) + oy + opad) * (wo + 2 * opad) + ox + opad) * gin.c + g * 8;
            float gv[8];
            load8(gy_hi, gy_lo, oo, gv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int id = (int)((pk[e >> 2] >> (8 * (e & 3))) & 0xffu);
                if (id == wpos) acc[e] += gv[e];
            }
        }
    }
}

// per-block (sum g, sum g*(z - mean)*rstd), g = gathered pooled gradient * [y>0]; requires 256 % (c/8) == 0 (a thread's channel
// group is loop-invariant)
__global__ __launch_bounds__(256) void pool_bn_bwd_sums_kernel(MapGeo gin, const uint8_t* __restrict__ idx, const bf16_t* gy_hi,
                                                               const bf16_t* gy_lo, int ho, int wo, int opad, const bf16_t* z_hi,
                                                               const bf16_t* z_lo, const bf16_t* y_hi, const float* mean,
                                                               const float* rstd, int relu, const float* fsc, const float* fsh,
                                                               float* partial) {
    __shared__ __attribute__((aligned(16))) float red[256 * 16];
    const int tid = threadIdx.x;
    const int groups0 = gin.c / 8, g0 = tid % groups0, ppb = 256 / groups0;
    float s1[8], s2[8], mu[8], msc[8], msh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s1[e] = 0.f; s2[e] = 0.f; mu[e] = mean[g0 * 8 + e];
        msc[e] = fsc ? fsc[g0 * 8 + e] : 0.f; msh[e] = fsc ? fsh[g0 * 8 + e] : 0.f;
    }
    AGP_FOR_MAP(gin) {
        AGP_MAP_INDEX(gin)
        float acc[8], z[8];
        pool_gather8(gin, idx, gy_hi, gy_lo, ho, wo, opad, im, py, px, g, acc);
        load8(z_hi, z_lo, off, z);
        unsigned pm = 0xffu;
        if (relu) {
            if (y_hi) pm = pos_mask8(y_hi, off);
            else {          // the output was never stored: the forward's own decision, recomputed
                pm = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) pm |= (stem_act(z[e], msc[e], msh[e]) > 0.f) ? (1u << e) : 0u;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gm = ((pm >> e) & 1u) ? acc[e] : 0.f;
            s1[e] += gm;
            s2[e] += gm * (z[e] - mu[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    if (tid < groups0) {
        float t1[8], t2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
        for (int k = 0; k < ppb; ++k) {
            const float* r = red + (k * groups0 + tid) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { t1[e] += r[e]; t2[e] += r[8 + e]; }
        }
        float* o = partial + (size_t)blockIdx.x * 2 * gin.c + tid * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = t1[e]; o[gin.c + e] = t2[e] * rstd[tid * 8 + e]; }
    }
}

__device__ __forceinline__ float bf16_bits_to_f32(bf16_t b) { return __uint_as_float((uint32_t)b << 16); }

// The same channel sums taken in the POOLED domain (a quarter of the elements, no gather): a window's gradient reaches the
// BatchNorm only if its maximum is positive (pooled value > 0 <=> ReLU mask at the argmax), and there y = gamma*zhat + beta, so
// zhat = (pooled - beta) / gamma -- no access to z.  Channels whose gamma is too small for that division (|gamma| < 1e-4 or
// |beta / gamma| > 16: the pooled map's 2^-17 storage rounding would be amplified) read z at the recorded argmax instead.
__global__ __launch_bounds__(256) void pool_bn_bwd_sums_pooled_kernel(MapGeo gin, MapGeo gp, const uint8_t* __restrict__ idx,
                                                                      const bf16_t* gy_hi, const bf16_t* gy_lo, const bf16_t* pv_hi,
                                                                      const bf16_t* pv_lo, const bf16_t* z_hi, const bf16_t* z_lo,
                                                                      const float* mean, const float* rstd, const float* gamma,
                                                                      const float* beta, float* partial) {
    __shared__ __attribute__((aligned(16))) float red[256 * 16];
    const int tid = threadIdx.x;
    const int groups0 = gp.c / 8, g0 = tid % groups0, ppb = 256 / groups0;
    float s1[8], s2[8], be[8], ig[8], mu[8], rs[8];
    unsigned slow = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = g0 * 8 + e;
        const float ga = gamma ? gamma[ch] : 1.f;
        be[e] = beta ? beta[ch] : 0.f;
        s1[e] = 0.f; s2[e] = 0.f; mu[e] = mean[ch]; rs[e] = rstd[ch];
        const bool bad = !(fabsf(ga) >= 1e-4f) || fabsf(be[e]) > 16.f * fabsf(ga);
        slow |= bad ? (1u << e) : 0u;
        ig[e] = bad ? 0.f : 1.f / ga;
    }
    const int hip_ = gin.h + 2 * gin.pad, wip = gin.w + 2 * gin.pad;
    AGP_FOR_MAP(gp) {
        AGP_MAP_INDEX(gp)             // (im, py, px) = pooled pixel, off = its offset in the pooled maps
        float pv[8], gv[8];
        load8(pv_hi, pv_lo, off, pv);
        load8(gy_hi, gy_lo, off, gv);
        float zh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) zh[e] = (pv[e] - be[e]) * ig[e];
        if (slow) {
            const u32x2 pk = *(const u32x2*)(idx + ((((size_t)im * gp.h + py) * gp.w + px) * gp.c + g * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (!((slow >> e) & 1u) || !(pv[e] > 0.f)) continue;
                const int id = (int)((pk[e >> 2] >> (8 * (e & 3))) & 0xffu);
                const int iy = 2 * py - 1 + id / 3, ix = 2 * px - 1 + id % 3;
                const size_t zo = (((size_t)im * hip_ + iy + gin.pad) * wip + ix + gin.pad) * gin.c + g * 8 + e;
                const float zv = z_lo ? bf16_bits_to_f32(z_hi[zo]) + bf16_bits_to_f32(z_lo[zo]) : h2f(z_hi[zo]);      // (a bf16 pair, or ONE fp16 plane)
                zh[e] = (zv - mu[e]) * rs[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (pv[e] > 0.f) { s1[e] += gv[e]; s2[e] += gv[e] * zh[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    if (tid < groups0) {
        float t1[8], t2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
        for (int k = 0; k < ppb; ++k) {
            const float* r = red + (k * groups0 + tid) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { t1[e] += r[e]; t2[e] += r[8 + e]; }
        }
        float* o = partial + (size_t)blockIdx.x * 2 * gp.c + tid * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = t1[e]; o[gp.c + e] = t2[e]; }
    }
}

// gz of the unit under the pool.  A thread owns the 2x2 block of full-size pixels (2oy + {0,1}, 2ox + {0,1}) of one pooled cell:
// the four windows (oy + {0,1}, ox + {0,1}) that can have their maximum inside it are loaded ONCE (one window per even, two per
// odd coordinate: nine window visits per block when every pixel gathers for itself, 2.25 per element instead of 1).
__global__ void __launch_bounds__(256) pool_bn_bwd_apply_kernel(MapGeo gin, MapGeo gp, const uint8_t* __restrict__ idx, const bf16_t* gy_hi,
                                         const bf16_t* gy_lo, const bf16_t* z_hi, const bf16_t* z_lo, const bf16_t* y_hi,
                                         const float* mean, const float* rstd, const float* gamma, const float* sum_g,
                                         const float* sum_gz, float inv_count, int relu, const float* fsc, const float* fsh,
                                         bf16_t* gz_hi, bf16_t* gz_lo, uint32_t* gz_absmax) {
    const int groups0 = gin.c / 8;
    const int g0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) % groups0);      // loop-invariant: 256 % groups0 == 0
    float cA[8], cB[8], cC[8], msc[8], msh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = g0 * 8 + e;
        msc[e] = fsc ? fsc[ch] : 0.f; msh[e] = fsc ? fsh[ch] : 0.f;
        const float gr = (gamma ? gamma[ch] : 1.f) * rstd[ch];
        cA[e] = gr;
        cB[e] = -gr * rstd[ch] * sum_gz[ch] * inv_count;
        cC[e] = -gr * sum_g[ch] * inv_count - cB[e] * mean[ch];
    }
    const int hip_ = gin.h + 2 * gin.pad, wip = gin.w + 2 * gin.pad;
    const int hop = gp.h + 2 * gp.pad, wop = gp.w + 2 * gp.pad;
    float mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // max |gz| of this thread's eight channels (gz_absmax)
    AGP_FOR_MAP(gp) {
        AGP_MAP_INDEX(gp)             // (im, py, px) = pooled cell (oy, ox)
        (void)off;
        u32x2 pk[2][2];
        float gv[2][2][8];
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = py + wy, ox = px + wx;
                const bool ok = oy < gp.h && ox < gp.w;
                pk[wy][wx] = u32x2{0xffffffffu, 0xffffffffu};          // matches no window position
                if (ok) {
                    pk[wy][wx] = *(const u32x2*)(idx + ((((size_t)im * gp.h + oy) * gp.w + ox) * gp.c + g * 8));
                    load8(gy_hi, gy_lo, (((size_t)im * hop + oy + gp.pad) * wop + ox + gp.pad) * gp.c + g * 8, gv[wy][wx]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[wy][wx][e] = 0.f;
                }
            }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int iy = 2 * py + dy, ix = 2 * px + dx;
                if (iy >= gin.h || ix >= gin.w) continue;
                float acc[8], z[8], o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
                for (int wy = 0; wy <= dy; ++wy)
#pragma unroll
                    for (int wx = 0; wx <= dx; ++wx) {
                        const unsigned wpos = 3u * (unsigned)(dy - 2 * wy + 1) + (unsigned)(dx - 2 * wx + 1);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned id = (pk[wy][wx][e >> 2] >> (8 * (e & 3))) & 0xffu;
                            if (id == wpos) acc[e] += gv[wy][wx][e];
                        }
                    }
                const size_t zo = (((size_t)im * hip_ + iy + gin.pad) * wip + ix + gin.pad) * gin.c + g * 8;
                load8(z_hi, z_lo, zo, z);
                unsigned pm = 0xffu;
                if (relu) {
                    if (y_hi) pm = pos_mask8(y_hi, zo);
                    else {
                        pm = 0;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pm |= (stem_act(z[e], msc[e], msh[e]) > 0.f) ? (1u << e) : 0u;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gm = ((pm >> e) & 1u) ? acc[e] : 0.f;
                    o[e] = cA[e] * gm + (cB[e] * z[e] + cC[e]);
                    mx[e] = fmaxf(mx[e], fabsf(o[e]));
                }
                store8(gz_hi, gz_lo, zo, o);
            }
    }
    if (gz_absmax) absmax_flush(gz_absmax, gin.c, g0, mx);      // (uniform branch: every thread of the block)
}

// Global pooling backward into a map gradient:
//   g = (b?) + gmean[n][c]/HW + ggem[n][c] * y^(1-p) * max(x,eps)^(p-1) * [x>=eps] / HW
// and, when gp != nullptr, dL/dp of the GeM exponent accumulated into gp[0] (one atomic per wave):
//   dy/dp = y * ( -ln(y)/p + mean(xc^p ln xc) / (p y^p) ),  xc = max(x, eps)
__global__ void pool_bwd_kernel(MapGeo geo, const bf16_t* x_hi, const bf16_t* x_lo, const float* gmean, const float* ggem,
                                const float* gem_y, const float* pptr, float eps, const bf16_t* b_hi, const bf16_t* b_lo,
                                bf16_t* o_hi, bf16_t* o_lo, float* gp) {
    const float inv_hw = 1.f / (float)(geo.h * geo.w);
    const float p = ggem ? pptr[0] : 1.f;
    float dp = 0.f;
    AGP_FOR_MAP(geo) {
        AGP_MAP_INDEX(geo)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (b_hi) load8(b_hi, b_lo, off, v);
        if (gmean) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += gmean[(size_t)im * geo.c + g * 8 + e] * inv_hw;
        }
        if (ggem) {
            float x[8];
            load8(x_hi, x_lo, off, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const size_t vi = (size_t)im * geo.c + g * 8 + e;
                const float yy = gem_y[vi], l2y = __builtin_log2f(yy);
                const float xc = fmaxf(x[e], eps), l2x = __builtin_log2f(xc);
                const float r = ggem[vi] * inv_hw * __builtin_exp2f((1.f - p) * l2y + (p - 1.f) * l2x);   // g * y^(1-p) xc^(p-1) / HW
                if (x[e] >= eps) v[e] += r;
                if (gp) {
                    dp += r * xc * (l2x * 0.6931471805599453f) / p;
                    if (px == 0 && py == 0) dp -= ggem[vi] * yy * (l2y * 0.6931471805599453f) / p;
                }
            }
        }
        store8(o_hi, o_lo, off, v);
    }
    if (gp) agp_grid_sum_ordered(agp_block_sum_ordered(dp), gp);      // (uniform: every thread of every block)
}

// Grid of the element-wise passes: four (pixel, channel group) items per thread -- a thread's per-channel coefficients (up to
// 40 scalar loads) are loop-invariant, and with one item per thread they cost more than the item (tools/bn_bench.py: the
// BatchNorm backward of a 16 x 14 x 84 x 256 map 97 -> 48 us, 28 x 168 x 128: 125 -> 73 us).
inline int grid_for(int64_t threads) {
    int64_t g = (threads + 1023) / 1024;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
// Blocks of the channel reductions: eight items per thread, but no fewer than 512 blocks while there are two items per thread
// (a 14 x 14 map of 32 images used to be reduced by 25 workgroups on a 256-CU device).
inline int reduce_blocks(const MapGeo& g) {
    const int64_t items = (int64_t)g.n * g.h * g.w * (g.c / 8 > 0 ? g.c / 8 : 1);
    int64_t b = (items + 2047) / 2048;
    if (b < 512) b = std::min<int64_t>(512, (items + 511) / 512);
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace agp_train
using namespace agp_train;

#define BF(p) ((bf16_t*)(p))
#define CBF(p) ((const bf16_t*)(p))

extern "C" int64_t agp_train_reduce_workspace_floats(int n, int h, int w, int c) {
    const MapGeo g = geo_of(n, h, w, c, 1);
    return (int64_t)reduce_blocks(g) * 2 * c + 8;       // + 8: agp_bn_bwd_apply's 2c + 1 coefficients when there is one block
}

extern "C" int agp_bn_stats(const void* z_hi, const void* z_lo, int n, int h, int w, int c, int pad, float eps,
                            float momentum, float* mean, float* rstd, float* running_mean, float* running_var,
                            const float* gamma, const float* beta, float* scale, float* shift, float* workspace,
                            void* stream) {
    if (!z_hi || !mean || !rstd || !workspace || c % 8 || c / 8 > 256 || n <= 0) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    const int nb = reduce_blocks(g);
    hipStream_t s = (hipStream_t)stream;
    AGP_LAUNCH(chan_reduce_kernel, dim3(nb), dim3(256), 256 * 16 * 4, s, g, CBF(z_hi), CBF(z_lo), nullptr, nullptr, nullptr,
               nullptr, nullptr, nullptr, 0, 0, workspace);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(bn_stats_final_kernel, dim3(c), dim3(256), 0, s, workspace, nb, c, (double)n * h * w, eps,
               momentum, mean, rstd, running_mean, running_var, gamma, beta, scale, shift);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bn_stats_from_partial(const float* partial, int tiles, int c, int64_t count, float eps, float momentum,
                                         float* mean, float* rstd, float* running_mean, float* running_var,
                                         const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
    if (!partial || !mean || !rstd || tiles <= 0 || c <= 0 || count <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(bn_stats_final_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, partial, tiles, c, (double)count, eps,
               momentum, mean, rstd, running_mean, running_var, gamma, beta, scale, shift);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_map_affine(const void* a_hi, const void* a_lo, const float* scale, const float* shift, const void* r_hi,
                              const void* r_lo, int n, int h, int w, int c, int pad, int relu, void* o_hi, void* o_lo,
                              void* o_h16, void* stream) {
    if (!a_hi || !o_hi || c % 8 || n <= 0) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    AGP_LAUNCH(affine_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8))), dim3(256), 0, (hipStream_t)stream, g, CBF(a_hi),
               CBF(a_lo), scale, shift, CBF(r_hi), CBF(r_lo), relu, BF(o_hi), BF(o_lo), BF(o_h16));
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

// frozen: the statistics are constants (eval-mode BatchNorm): gz = gamma*rstd*g, no mean / projection terms
static int bn_bwd_impl(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo, const void* y_hi,
                       const void* y_lo, const float* mean, const float* rstd, const float* gamma, int n, int h, int w,
                       int c, int pad, int relu, void* gz_hi, void* gz_lo, void* gres_hi, void* gres_lo, float* ggamma,
                       float* gbeta, float* workspace, uint32_t* gz_absmax, void* stream, bool frozen) {
    if (!z_hi || !gy_hi || !mean || !rstd || !gz_hi || !ggamma || !gbeta || !workspace || c % 8 || c / 8 > 256 || n <= 0)
        return AGP_E_BADARG;
    if (relu && !y_hi) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    const int nb = reduce_blocks(g);
    hipStream_t s = (hipStream_t)stream;
    AGP_LAUNCH(chan_reduce_kernel, dim3(nb), dim3(256), 256 * 16 * 4, s, g, CBF(z_hi), CBF(z_lo), CBF(gy_hi), CBF(gy_lo),
               CBF(y_hi), CBF(y_lo), mean, rstd, 1, relu, workspace);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sum2_final_kernel, dim3(c), dim3(256), 0, s, workspace, nb, c, gbeta, ggamma);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(bn_bwd_apply_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8))), dim3(256), 0, s, g, CBF(z_hi), CBF(z_lo),
               CBF(gy_hi), CBF(gy_lo), CBF(y_hi), CBF(y_lo), mean, rstd, gamma, gbeta, ggamma,
               frozen ? 0.f : 1.f / (float)((double)n * h * w), (const float*)nullptr, relu, BF(gz_hi), BF(gz_lo), BF(gres_hi),
               BF(gres_lo), gz_absmax);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bn_bwd_from_partial(const float* partial, int tiles, const void* z_hi, const void* z_lo, const void* gy_hi,
                                       const void* gy_lo, const void* y_hi, const void* y_lo, const float* mean,
                                       const float* rstd, const float* gamma, int n, int h, int w, int c, int pad, int relu,
                                       int frozen, void* gz_hi, void* gz_lo, void* gres_hi, void* gres_lo, float* ggamma,
                                       float* gbeta, uint32_t* gz_absmax, void* stream) {
    if (!partial || tiles <= 0 || !z_hi || !gy_hi || !mean || !rstd || !gz_hi || !ggamma || !gbeta || c % 8 || c / 8 > 256 || n <= 0)
        return AGP_E_BADARG;
    if (relu && !y_hi) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    AGP_LAUNCH(sum2_final_kernel, dim3(c), dim3(256), 0, s, partial, tiles, c, gbeta, ggamma);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(bn_bwd_apply_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8))), dim3(256), 0, s, g, CBF(z_hi), CBF(z_lo),
               CBF(gy_hi), CBF(gy_lo), CBF(y_hi), CBF(y_lo), mean, rstd, gamma, gbeta, ggamma,
               frozen ? 0.f : 1.f / (float)((double)n * h * w), (const float*)nullptr, relu, BF(gz_hi), BF(gz_lo), BF(gres_hi),
               BF(gres_lo), gz_absmax);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bn_bwd(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo, const void* y_hi,
                          const void* y_lo, const float* mean, const float* rstd, const float* gamma, int n, int h, int w,
                          int c, int pad, int relu, void* gz_hi, void* gz_lo, void* gres_hi, void* gres_lo, float* ggamma,
                          float* gbeta, float* workspace, uint32_t* gz_absmax, void* stream) {
    return bn_bwd_impl(z_hi, z_lo, gy_hi, gy_lo, y_hi, y_lo, mean, rstd, gamma, n, h, w, c, pad, relu, gz_hi, gz_lo, gres_hi,
                       gres_lo, ggamma, gbeta, workspace, gz_absmax, stream, false);
}

extern "C" int agp_bn_bwd_frozen(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo, const void* y_hi,
                                 const void* y_lo, const float* mean, const float* rstd, const float* gamma, int n, int h,
                                 int w, int c, int pad, int relu, void* gz_hi, void* gz_lo, void* gres_hi, void* gres_lo,
                                 float* ggamma, float* gbeta, float* workspace, uint32_t* gz_absmax, void* stream) {
    return bn_bwd_impl(z_hi, z_lo, gy_hi, gy_lo, y_hi, y_lo, mean, rstd, gamma, n, h, w, c, pad, relu, gz_hi, gz_lo, gres_hi,
                       gres_lo, ggamma, gbeta, workspace, gz_absmax, stream, true);
}

extern "C" int agp_bn_sums(const void* z_hi, const void* z_lo, int n, int h, int w, int c, int pad, double* sums,
                           float* workspace, void* stream) {
    if (!z_hi || !sums || !workspace || c % 8 || c / 8 > 256 || n <= 0) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    const int nb = reduce_blocks(g);
    hipStream_t s = (hipStream_t)stream;
    AGP_LAUNCH(chan_reduce_kernel, dim3(nb), dim3(256), 256 * 16 * 4, s, g, CBF(z_hi), CBF(z_lo), nullptr, nullptr, nullptr,
               nullptr, nullptr, nullptr, 0, 0, workspace);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sums_f64_kernel, dim3(c), dim3(256), 0, s, workspace, nb, c, (double)n * h * w, sums, (float*)nullptr,
               (float*)nullptr);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bn_sums_from_partial(const float* partial, int tiles, int c, int64_t count, double* sums, void* stream) {
    if (!partial || !sums || tiles <= 0 || c <= 0 || count <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(sums_f64_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, partial, tiles, c, (double)count, sums,
               (float*)nullptr, (float*)nullptr);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bn_stats_from_sums(const double* sums, int c, float eps, float momentum, float* mean, float* rstd,
                                      float* running_mean, float* running_var, const float* gamma, const float* beta,
                                      float* scale, float* shift, void* stream) {
    if (!sums || !mean || !rstd || c <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(bn_stats_from_sums_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, c, eps, momentum, mean,
               rstd, running_mean, running_var, gamma, beta, scale, shift);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bn_bwd_sums(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo, const void* y_hi,
                               const void* y_lo, const float* mean, const float* rstd, int n, int h, int w, int c, int pad,
                               int relu, double* sums, float* ggamma, float* gbeta, float* workspace, void* stream) {
    if (!z_hi || !gy_hi || !mean || !rstd || !sums || !ggamma || !gbeta || !workspace || c % 8 || c / 8 > 256 || n <= 0)
        return AGP_E_BADARG;
    if (relu && !y_hi) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    const int nb = reduce_blocks(g);
    hipStream_t s = (hipStream_t)stream;
    AGP_LAUNCH(chan_reduce_kernel, dim3(nb), dim3(256), 256 * 16 * 4, s, g, CBF(z_hi), CBF(z_lo), CBF(gy_hi), CBF(gy_lo),
               CBF(y_hi), CBF(y_lo), mean, rstd, 1, relu, workspace);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sums_f64_kernel, dim3(c), dim3(256), 0, s, workspace, nb, c, -1.0, sums, gbeta, ggamma);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bn_bwd_apply(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo, const void* y_hi,
                                const void* y_lo, const float* mean, const float* rstd, const float* gamma, const double* sums,
                                const double* count, int n, int h, int w, int c, int pad, int relu, void* gz_hi, void* gz_lo,
                                void* gres_hi, void* gres_lo, float* workspace, void* stream) {
    if (!z_hi || !gy_hi || !mean || !rstd || !sums || !count || !gz_hi || !workspace || c % 8 || c / 8 > 256 || n <= 0)
        return AGP_E_BADARG;
    if (relu && !y_hi) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float* sg = workspace, *sgz = workspace + c, *ic = workspace + 2 * c;       // (the reduction workspace holds >= 2c + 1 floats)
    AGP_LAUNCH(bn_bwd_coeffs_kernel, dim3((c + 255) / 256), dim3(256), 0, s, sums, count, c, sg, sgz, ic);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(bn_bwd_apply_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8))), dim3(256), 0, s, g, CBF(z_hi), CBF(z_lo),
               CBF(gy_hi), CBF(gy_lo), CBF(y_hi), CBF(y_lo), mean, rstd, gamma, (const float*)sg, (const float*)sgz, 0.f,
               (const float*)ic, relu, BF(gz_hi), BF(gz_lo), BF(gres_hi), BF(gres_lo), (uint32_t*)nullptr);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bn_frozen_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                                    int c, float eps, float* mean, float* rstd, float* scale, float* shift, void* stream) {
    if (!running_mean || !running_var || !mean || !rstd || !scale || !shift || c <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(bn_frozen_coeffs_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, running_mean, running_var,
               gamma, beta, c, eps, mean, rstd, scale, shift);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_map_chan_sum(const void* a_hi, const void* a_lo, int n, int h, int w, int c, int pad, float* out,
                                float* workspace, void* stream) {
    if (!a_hi || !out || !workspace || c % 8 || c / 8 > 256 || n <= 0) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    const int nb = reduce_blocks(g);
    hipStream_t s = (hipStream_t)stream;
    AGP_LAUNCH(chan_reduce_kernel, dim3(nb), dim3(256), 256 * 16 * 4, s, g, CBF(a_hi), CBF(a_lo), nullptr, nullptr, nullptr,
               nullptr, nullptr, nullptr, 0, 0, workspace);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sum2_final_kernel, dim3(c), dim3(256), 0, s, workspace, nb, c, out, nullptr);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_map_add(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, const void* mask_hi,
                           const void* mask_lo, int n, int h, int w, int c, int pad, void* o_hi, void* o_lo, void* stream) {
    if (!a_hi || !o_hi || c % 8 || n <= 0) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    AGP_LAUNCH(add_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8))), dim3(256), 0, (hipStream_t)stream, g, CBF(a_hi),
               CBF(a_lo), CBF(b_hi), CBF(b_lo), CBF(mask_hi), CBF(mask_lo), BF(o_hi), BF(o_lo));
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_upsample2_zero(const void* g_hi, const void* g_lo, int n, int ho, int wo, int c, int gpad, void* u_hi,
                                  void* u_lo, int hu, int wu, int upad, void* stream) {
    if (!g_hi || !u_hi || c % 8 || n <= 0) return AGP_E_BADARG;
    const MapGeo gu = geo_of(n, hu, wu, c, upad);
    if (!geo_fits(n, hu, wu, c)) return AGP_E_BADARG;
    AGP_LAUNCH(upsample2_kernel, dim3(grid_for((int64_t)n * hu * wu * (c / 8))), dim3(256), 0, (hipStream_t)stream, gu,
               CBF(g_hi), CBF(g_lo), ho, wo, gpad, BF(u_hi), BF(u_lo));
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_maxpool3x3s2_bwd(const uint8_t* argmax, const void* gy_hi, const void* gy_lo, int n, int hin, int win, int c,
                                    int pin, int hout, int wout, int pout, void* gx_hi, void* gx_lo, void* stream) {
    if (!argmax || !gy_hi || !gx_hi || c % 8 || n <= 0) return AGP_E_BADARG;
    const MapGeo gin = geo_of(n, hin, win, c, pin);
    if (!geo_fits(n, hin, win, c)) return AGP_E_BADARG;
    AGP_LAUNCH(maxpool_bwd_kernel, dim3(grid_for((int64_t)n * hin * win * (c / 8))), dim3(256), 0, (hipStream_t)stream, gin, argmax,
               CBF(gy_hi), CBF(gy_lo), hout, wout, pout, BF(gx_hi), BF(gx_lo));
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_maxpool_bn_bwd(const uint8_t* argmax, const void* gp_hi, const void* gp_lo, int hout, int wout, int pout,
                                  const void* z_hi, const void* z_lo, const void* y_hi, const void* y_lo, const float* mean,
                                  const float* rstd, const float* gamma, const float* scale, const float* shift,
                                  const void* pv_hi, const void* pv_lo, const float* beta, int n, int h,
                                  int w, int c, int pad, int relu, int frozen, void* gz_hi, void* gz_lo, float* ggamma,
                                  float* gbeta, float* workspace, uint32_t* gz_absmax, void* stream) {
    if (!argmax || !gp_hi || !z_hi || !mean || !rstd || !gz_hi || !ggamma || !gbeta || !workspace || c % 8 || n <= 0) return AGP_E_BADARG;
    if (relu && !y_hi && !(scale && shift)) return AGP_E_BADARG;
    const float* fsc = y_hi ? nullptr : scale;
    const float* fsh = y_hi ? nullptr : shift;
    if (c / 8 > 256 || 256 % (c / 8)) return AGP_E_UNSUPPORTED;
    // the pooled geometry is checked BEFORE any launch: the sums kernels index argmax / gp / pv with it
    if (hout != (h + 2 - 3) / 2 + 1 || wout != (w + 2 - 3) / 2 + 1 || pout < 0 || pad < 0) return AGP_E_BADARG;      // every full-size pixel lies in a 2x2 block
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    int nb = reduce_blocks(g);
    hipStream_t s = (hipStream_t)stream;
    if (pv_hi && relu) {
        // the sums in the pooled domain (a quarter of the elements, no gather); fewer blocks than the workspace holds
        const MapGeo gq = geo_of(n, hout, wout, c, pout);
        nb = std::min(nb, reduce_blocks(gq));
        AGP_LAUNCH(pool_bn_bwd_sums_pooled_kernel, dim3(nb), dim3(256), 0, s, g, gq, argmax, CBF(gp_hi), CBF(gp_lo), CBF(pv_hi),
                   CBF(pv_lo), CBF(z_hi), CBF(z_lo), mean, rstd, gamma, beta, workspace);
    } else {
        AGP_LAUNCH(pool_bn_bwd_sums_kernel, dim3(nb), dim3(256), 0, s, g, argmax, CBF(gp_hi), CBF(gp_lo), hout, wout, pout, CBF(z_hi),
                   CBF(z_lo), CBF(y_hi), mean, rstd, relu, fsc, fsh, workspace);
    }
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sum2_final_kernel, dim3(c), dim3(256), 0, s, workspace, nb, c, gbeta, ggamma);
    AGP_CHECK_LAUNCH();
    const MapGeo gpool = geo_of(n, hout, wout, c, pout);
    AGP_LAUNCH(pool_bn_bwd_apply_kernel, dim3(grid_for((int64_t)n * hout * wout * (c / 8))), dim3(256), 0, s, g, gpool, argmax,
               CBF(gp_hi), CBF(gp_lo), CBF(z_hi), CBF(z_lo), CBF(y_hi), mean, rstd, gamma, gbeta, ggamma,
               frozen ? 0.f : 1.f / (float)((double)n * h * w), relu, fsc, fsh, BF(gz_hi), BF(gz_lo), gz_absmax);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_affine_maxpool3x3s2_fwd(const void* z_hi, const void* z_lo, const float* scale, const float* shift, int n, int h,
                                           int w, int c, int pad, void* out_hi, void* out_lo, int hout, int wout, int pout,
                                           uint8_t* argmax, void* out_h16, void* stream) {
    if (!z_hi || !scale || !shift || !out_hi || !argmax || c % 8 || n <= 0) return AGP_E_BADARG;
    if (hout != (h + 2 - 3) / 2 + 1 || wout != (w + 2 - 3) / 2 + 1) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    AGP_LAUNCH(affine_maxpool_kernel, dim3(grid_for((int64_t)n * hout * wout * (c / 8))), dim3(256), 0, (hipStream_t)stream, g,
               CBF(z_hi), CBF(z_lo), scale, shift, BF(out_hi), BF(out_lo), hout, wout, pout, argmax, BF(out_h16));
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_pool_bwd(const void* x_hi, const void* x_lo, const float* gmean, const float* ggem, const float* gem_y,
                            const float* p, float eps, const void* b_hi, const void* b_lo, int n, int h, int w, int c, int pad,
                            void* o_hi, void* o_lo, float* gp, void* stream) {
    if (!o_hi || c % 8 || n <= 0 || (ggem && (!x_hi || !gem_y || !p)) || (gp && !ggem)) return AGP_E_BADARG;
    const MapGeo g = geo_of(n, h, w, c, pad);
    if (!geo_fits(n, h, w, c)) return AGP_E_BADARG;
    AGP_LAUNCH(pool_bwd_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8))), dim3(256), 0, (hipStream_t)stream, g, CBF(x_hi),
               CBF(x_lo), gmean, ggem, gem_y, p, eps, CBF(b_hi), CBF(b_lo), BF(o_hi), BF(o_lo), gp);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
