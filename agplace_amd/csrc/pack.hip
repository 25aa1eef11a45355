// HBM-bound layout / elementwise kernels on halo-padded NHWC map planes (split-bf16 pair or one
// fp16 plane, see common.hpp map_load8 / map_store8).
// All of them move 16 bytes per lane per plane (8 bf16 channels) with lanes running
// over channels first, so every wave touches whole 128-byte lines.
#include <math.h>
#include <string.h>
#include "common.hpp"

namespace agp_pack {

// fmt: AGP_FMT_BF16 (hi = rn_bf16(x), lo = rn_bf16(x - hi)) or AGP_FMT_F16 (same in fp16)
__global__ void split_f32_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi,
                                 bf16_t* __restrict__ lo, int64_t n, int fmt) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (; i < n; i += stride) {
        if (i + 8 <= n) {
            const f32x4 a = *(const f32x4*)(x + i), b = *(const f32x4*)(x + i + 4);
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            u32x4 h, l;
            if (fmt == AGP_FMT_F16) {
                h = pack8_h(v);
                float r[8];
                unpack8_h(h, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = v[e] - r[e];
                l = pack8_h(r);
            } else {
                split8(v, h, l);
            }
            *(u32x4*)(hi + i) = h;
            if (lo) *(u32x4*)(lo + i) = l;
        } else {
            for (int64_t j = i; j < n; ++j) {
                bf16_t h, l;
                if (fmt == AGP_FMT_F16) split_f16(x[j], h, l);
                else split_bf16(x[j], h, l);
                hi[j] = h;
                if (lo) lo[j] = l;
            }
        }
    }
}

// one thread per (pixel, group of CG channels); CG = 4 (stem, cpad == 4) or 8
template <int CG>
__global__ void pack_kernel(const float* __restrict__ x, int64_t sn, int64_t sc, int64_t sh,
                            int64_t sw, int n, int c, int h, int w, int cpad, int pad,
                            bf16_t* __restrict__ hi, bf16_t* __restrict__ lo) {
    const int groups = cpad / CG;
    const int64_t total = (int64_t)n * h * w * groups;
    const int hp = h + 2 * pad, wp = w + 2 * pad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        // groups fastest only when the source is channels-last; otherwise pixels fastest
        int g, px, py, im;
        int64_t r = t;
        if (sc == 1) { g = r % groups; r /= groups; px = r % w; r /= w; py = r % h; im = r / h; }
        else { px = r % w; r /= w; py = r % h; r /= h; g = r % groups; im = r / groups; }
        const float* src = x + im * sn + py * sh + px * sw;
        bf16_t hh[CG], ll[CG];
#pragma unroll
        for (int e = 0; e < CG; ++e) {
            const int ch = g * CG + e;
            const float v = ch < c ? src[ch * sc] : 0.f;
            map_split1(v, lo != nullptr, hh[e], ll[e]);
        }
        const size_t off = (((size_t)im * hp + py + pad) * wp + px + pad) * cpad + g * CG;
        if (CG == 4) {
            u32x2 a = {pack2(hh[0], hh[1]), pack2(hh[2], hh[3])};
            *(u32x2*)(hi + off) = a;
            if (lo) { u32x2 b = {pack2(ll[0], ll[1]), pack2(ll[2], ll[3])}; *(u32x2*)(lo + off) = b; }
        } else {
            u32x4 a = {pack2(hh[0], hh[1]), pack2(hh[2], hh[3]), pack2(hh[4 % CG], hh[5 % CG]), pack2(hh[6 % CG], hh[7 % CG])};
            *(u32x4*)(hi + off) = a;
            if (lo) {
                u32x4 b = {pack2(ll[0], ll[1]), pack2(ll[2], ll[3]), pack2(ll[4 % CG], ll[5 % CG]), pack2(ll[6 % CG], ll[7 % CG])};
                *(u32x4*)(lo + off) = b;
            }
        }
    }
}

// Fast path of the stem input: fp32 NCHW planes with unit pixel stride, c <= 4 -> NHWC4.  One thread per FOUR pixels of a
// row: one 16-byte load per channel plane, 32 contiguous output bytes, 32-bit index arithmetic with fast division
// (the generic kernel does five 64-bit divisions per pixel and moves 12 + 8 bytes per thread).
__global__ __launch_bounds__(256) void pack_nchw4_kernel(const float* __restrict__ x, int64_t sn, int64_t sc, int64_t sh,
                                                         int n, int c, int h, int w, int pad, FastDiv dw4, FastDiv dh,
                                                         bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                         bf16_t* __restrict__ h16 = nullptr) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t w4 = dw4.d;
    if (t >= (uint32_t)n * h * w4) return;
    const uint32_t q = fdiv(t, dw4), px = (t - q * w4) * 4;
    const uint32_t im = fdiv(q, dh), py = q - im * dh.d;
    const float* src = x + im * sn + py * sh + px;
    f32x4 v[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) v[ch] = ch < c ? *(const f32x4*)(src + ch * sc) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int hp = h + 2 * pad, wp = w + 2 * pad;
    const size_t off = (((size_t)im * hp + py + pad) * wp + px + pad) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bf16_t hh[4], ll[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) map_split1(v[ch][k], lo != nullptr, hh[ch], ll[ch]);
        u32x2 a = {pack2(hh[0], hh[1]), pack2(hh[2], hh[3])};
        *(u32x2*)(hi + off + 4 * k) = a;
        if (lo) { u32x2 b = {pack2(ll[0], ll[1]), pack2(ll[2], ll[3])}; *(u32x2*)(lo + off + 4 * k) = b; }
        if (h16) {                                       // the fp16 operand plane of the stem's one-pass weight gradient
            u32x2 e = {pack2(f2h(v[0][k]), f2h(v[1][k])), pack2(f2h(v[2][k]), f2h(v[3][k]))};
            *(u32x2*)(h16 + off + 4 * k) = e;
        }
    }
}

__global__ void unpack_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, int n,
                              int h, int w, int c, int pad, float* __restrict__ out) {
    const int groups = c / 8;
    const int64_t total = (int64_t)n * h * w * groups;
    const int hp = h + 2 * pad, wp = w + 2 * pad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t;
        const int g = r % groups; r /= groups;
        const int px = r % w; r /= w;
        const int py = r % h;
        const int im = r / h;
        const size_t off = (((size_t)im * hp + py + pad) * wp + px + pad) * c + g * 8;
        float v[8];
        map_load8(hi, lo, off, v);
        float* o = out + t * 8;
        *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// idx (optional, training): uint8 [n][hout][wout][c], the window position 3*ky + kx of the FIRST maximum
// in row-major window order (torch keeps the first element that is strictly greater)
template <bool IDX>
__global__ void maxpool_kernel(const bf16_t* __restrict__ ihi, const bf16_t* __restrict__ ilo, int n,
                               int hin, int win, int c, int pin, bf16_t* __restrict__ ohi,
                               bf16_t* __restrict__ olo, int hout, int wout, int pout, uint8_t* __restrict__ idx) {
    const int groups = c / 8;
    const int64_t total = (int64_t)n * hout * wout * groups;
    const int hip_ = hin + 2 * pin, wip = win + 2 * pin;
    const int hop = hout + 2 * pout, wop = wout + 2 * pout;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t;
        const int g = r % groups; r /= groups;
        const int ox = r % wout; r /= wout;
        const int oy = r % hout;
        const int im = r / hout;
        float best[8];
        uint8_t bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = 0.f; bi[e] = 4; }   // inputs are post-ReLU (>= 0); all-zero window -> centre
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // input pixel (2*oy + ky - 1, 2*ox + kx - 1); halo (pin >= 1) holds zeros
                const int iy = 2 * oy + ky - 1 + pin, ix = 2 * ox + kx - 1 + pin;
                if (iy >= hip_ || ix >= wip) continue;
                const size_t off = (((size_t)im * hip_ + iy) * wip + ix) * c + g * 8;
                float v[8];
                map_load8(ihi, ilo, off, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (IDX) { if (v[e] > best[e]) { best[e] = v[e]; bi[e] = (uint8_t)(3 * ky + kx); } }
                    else best[e] = fmaxf(best[e], v[e]);
                }
            }
        const size_t off = (((size_t)im * hop + oy + pout) * wop + ox + pout) * c + g * 8;
        map_store8(ohi, olo, off, best);
        if (IDX) {
            u32x2 pk = {(uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24),
                        (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24)};
            *(u32x2*)(idx + ((((size_t)im * hout + oy) * wout + ox) * c + g * 8)) = pk;
        }
    }
}

__global__ void bcast_add_kernel(const bf16_t* __restrict__ ihi, const bf16_t* __restrict__ ilo,
                                 const float* __restrict__ vec, int n, int h, int w, int c, int pin,
                                 bf16_t* __restrict__ ohi, bf16_t* __restrict__ olo, int pout) {
    const int groups = c / 8;
    const int64_t total = (int64_t)n * h * w * groups;
    const int hip_ = h + 2 * pin, wip = w + 2 * pin, hop = h + 2 * pout, wop = w + 2 * pout;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t;
        const int g = r % groups; r /= groups;
        const int px = r % w; r /= w;
        const int py = r % h;
        const int im = r / h;
        const size_t ioff = (((size_t)im * hip_ + py + pin) * wip + px + pin) * c + g * 8;
        const size_t ooff = (((size_t)im * hop + py + pout) * wop + px + pout) * c + g * 8;
        float v[8];
        map_load8(ihi, ilo, ioff, v);
        const float* ve = vec + (size_t)im * c + g * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += ve[e];
        map_store8(ohi, olo, ooff, v);
    }
}

// uint8 HWC camera tiles -> normalised NHWC4 stem input, tiles concatenated along W:
//   v = (u8 / 255 - mean[c]) / std[c]   (torchvision ToTensor + Normalize, fp32), 4th channel zero
__global__ void pack_u8_cams_kernel(const uint8_t* __restrict__ img, int n, int ncam, int h, int w, float m0, float m1,
                                    float m2, float s0, float s1, float s2, int pad, bf16_t* __restrict__ hi,
                                    bf16_t* __restrict__ lo) {
    const int wt = ncam * w;
    const int64_t total = (int64_t)n * h * wt;
    const int hp = h + 2 * pad, wp = wt + 2 * pad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t;
        const int px = (int)(r % wt); r /= wt;
        const int py = (int)(r % h);
        const int im = (int)(r / h);
        const int cam = px / w, x = px - cam * w;
        const uint8_t* src = img + ((((int64_t)im * ncam + cam) * h + py) * w + x) * 3;
        const float v0 = ((float)src[0] / 255.f - m0) / s0;
        const float v1 = ((float)src[1] / 255.f - m1) / s1;
        const float v2 = ((float)src[2] / 255.f - m2) / s2;
        bf16_t hh[4], ll[4];
        map_split1(v0, lo != nullptr, hh[0], ll[0]);
        map_split1(v1, lo != nullptr, hh[1], ll[1]);
        map_split1(v2, lo != nullptr, hh[2], ll[2]);
        hh[3] = 0; ll[3] = 0;
        const size_t off = (((size_t)im * hp + py + pad) * wp + px + pad) * 4;
        u32x2 a = {pack2(hh[0], hh[1]), pack2(hh[2], hh[3])};
        *(u32x2*)(hi + off) = a;
        if (lo) { u32x2 b = {pack2(ll[0], ll[1]), pack2(ll[2], ll[3])}; *(u32x2*)(lo + off) = b; }
    }
}

inline int grid_for(int64_t threads, int tpb) {
    int64_t g = (threads + tpb - 1) / tpb;
    const int64_t cap = 256 * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// Conv weights straight from the parameter layout [cout][cin][kh][kw] (fp32) to the kernels' split bf16 planes, one launch:
//   dgrad == 0: out[n = cout][ky][kx][c = cin]  = w[n][c][ky][kx]                     (forward)
//   dgrad == 1: out[n = cin][ky][kx][c = cout]  = w[c][n][kh-1-ky][kw-1-kx]          (data gradient: flipped, transposed)
// One thread per 8 consecutive output channels c (the source is strided by kh*kw, or cin*kh*kw: small tensors).
// dgrad == 2: BOTH in one launch -- threads [0, total_fwd) write (hi, lo), the rest the data-gradient planes (hi_d, lo_d).
// cm bit 0 / bit 1: the forward / the data-gradient planes in CHUNK-MAJOR order [K/32][n][32] (agp_conv_desc::w_cm) instead of [n][K].
__global__ void split_conv_weight_kernel(const float* __restrict__ w, int cout, int cin, int kh, int kw, int dgrad,
                                         bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, bf16_t* __restrict__ hi_d = nullptr,
                                         bf16_t* __restrict__ lo_d = nullptr, int cm = 0) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (dgrad == 2) {
        const int64_t total_f = (int64_t)cout * kh * kw * (cin / 8);
        dgrad = t >= total_f ? 1 : 0;
        if (dgrad) { t -= total_f; hi = hi_d; lo = lo_d; }
    }
    const bool chunk_major = (cm >> dgrad) & 1;
    const int nn = dgrad ? cin : cout, cc = dgrad ? cout : cin, taps = kh * kw;
    const int64_t total = (int64_t)nn * taps * (cc / 8);
    if (t >= total) return;
    const int cg = (int)(t % (cc / 8));
    const int tap = (int)((t / (cc / 8)) % taps);
    const int n = (int)(t / ((int64_t)(cc / 8) * taps));
    const int ky = tap / kw, kx = tap % kw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cg * 8 + e;
        v[e] = dgrad ? w[(((int64_t)c * cin + n) * kh + (kh - 1 - ky)) * kw + (kw - 1 - kx)]
                     : w[(((int64_t)n * cin + c) * kh + ky) * kw + kx];
    }
    u32x4 h, l;
    split8(v, h, l);
    size_t off = ((size_t)n * taps + tap) * cc + cg * 8;
    if (chunk_major) {
        const int k = tap * cc + cg * 8;                     // K index of the group's first element (cc % 32 == 0)
        off = ((size_t)(k >> 5) * nn + n) * 32 + (k & 31);
    }
    *(u32x4*)(hi + off) = h;
    *(u32x4*)(lo + off) = l;
}

// ---- e4m3 lo plane of the F16W2 mode (agp_conv_desc::w_q8), 3x3 convs, w = [cout][3][3][cin] fp32
__global__ void q8_absmax_kernel(const float* __restrict__ w, int64_t n, unsigned int* __restrict__ mx) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i] - h2f(f2h(w[i]))));
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(mx, __float_as_uint(m));      // non-negative floats order as their bit patterns
}

// one thread per dword of the plane: plane[n][pair][lh][tap][ks][e 0..7], phase 2*pair + tap in the kernel's order (ky, cc, kx)
__global__ void q8_write_kernel(const float* __restrict__ w, int cout, int cin, float mul, unsigned int* __restrict__ q8) {
    const int cchunks = cin / 32, npair = 9 * cchunks / 2;
    const int64_t total = (int64_t)cout * npair * 16;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int d = (int)(t & 1), ks = (int)((t >> 1) & 1), tap = (int)((t >> 2) & 1), lh = (int)((t >> 3) & 1);
    const int64_t np = t >> 4;
    const int pair = (int)(np % npair), n = (int)(np / npair);
    const int phase = 2 * pair + tap, st = phase / 3, kx = phase % 3, ky = st / cchunks, cc = st % cchunks;
    const float* src = w + (((int64_t)n * 3 + ky) * 3 + kx) * cin + 32 * cc + 16 * ks + 8 * lh + 4 * d;
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (src[e] - h2f(f2h(src[e]))) * mul;
    int o = 0;
    o = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], o, false);
    o = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], o, true);
    q8[t] = (unsigned int)o;
}

// zero the halo of a freshly allocated map [n][h + 2 pad][w + 2 pad][c]: thread = (image, border pixel, 8-channel group).
// A map's interior is written by the kernel that produces it; filling the WHOLE map first cost the training step 162 fills
// of 4-100 MB (0.7 ms of 17.6).
template <class T>
__global__ void zero_halo_kernel(bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int n, int h, int w, int c, int pad) {
    const int hp = h + 2 * pad, wp = w + 2 * pad, g8 = c * 2 / (int)sizeof(T);       // T-sized pieces of a pixel
    const int64_t border = (int64_t)2 * pad * wp + (int64_t)2 * pad * h;             // pixels: top + bottom rows, then the side columns
    const int64_t total = (int64_t)n * border * g8;
    T z;
    __builtin_memset(&z, 0, sizeof(T));
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % g8);
        const int64_t q = t / g8;
        const int img = (int)(q / border);
        int64_t b = q - (int64_t)img * border;
        int y, x;
        if (b < (int64_t)2 * pad * wp) {
            const int r = (int)(b / wp);
            x = (int)(b - (int64_t)r * wp);
            y = r < pad ? r : h + r;                                                  // rows 0 .. pad-1 and h + pad .. h + 2 pad - 1
        } else {
            b -= (int64_t)2 * pad * wp;
            const int r = (int)(b / (2 * pad)), cidx = (int)(b % (2 * pad));
            y = pad + r;
            x = cidx < pad ? cidx : w + cidx;
        }
        const size_t off = (((size_t)img * hp + y) * wp + x) * c;
        ((T*)(hi + off))[g] = z;
        if (lo) ((T*)(lo + off))[g] = z;
    }
}

}  // namespace agp_pack
using namespace agp_pack;

extern "C" int agp_conv_w_q8_prepare(const float* w, int cout, int cin, void* q8, int32_t* exp_host, void* stream) {
    if (!w || !q8 || !exp_host || cout <= 0 || cin <= 0 || cin % 64) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    unsigned int* mx = nullptr;
    if (hipMalloc(&mx, 4) != hipSuccess) return AGP_E_LAUNCH;
    unsigned int h = 0;
    int rc = AGP_OK;
    const int64_t n = (int64_t)cout * 9 * cin;
    if (hipMemsetAsync(mx, 0, 4, s) != hipSuccess) rc = AGP_E_LAUNCH;
    if (rc == AGP_OK) {
        hipLaunchKernelGGL(q8_absmax_kernel, dim3(grid_for(n, 256) > 1024 ? 1024 : grid_for(n, 256)), dim3(256), 0, s, w, n, mx);
        if (hipMemcpyAsync(&h, mx, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = AGP_E_LAUNCH;
    }
    if (rc == AGP_OK) {
        float m;
        memcpy(&m, &h, 4);
        int e = 0;
        if (m > 0.f) {
            e = (int)floorf(log2f(448.f / m));
            while (ldexpf(m, e) > 448.f) --e;         // guard the rounding of log2f
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
        }
        *exp_host = e;
        const int64_t dwords = n / 4;
        hipLaunchKernelGGL(q8_write_kernel, dim3(grid_for(dwords, 256)), dim3(256), 0, s, w, cout, cin, ldexpf(1.f, e), (unsigned int*)q8);
        if (hipGetLastError() != hipSuccess) rc = AGP_E_LAUNCH;
    }
    (void)hipFree(mx);
    return rc;
}

extern "C" int agp_split_f32(const float* x, void* hi, void* lo, int64_t n, int fmt, void* stream) {
    if (!x || !hi || n < 0 || (fmt != AGP_FMT_BF16 && fmt != AGP_FMT_F16)) return AGP_E_BADARG;
    if (n == 0) return AGP_OK;
    AGP_LAUNCH(split_f32_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, (bf16_t*)hi, (bf16_t*)lo, n, fmt);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_split_conv_weight(const float* w, int cout, int cin, int kh, int kw, int dgrad, void* hi, void* lo,
                                     void* stream) {
    if (!w || !hi || !lo || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || (dgrad ? cout : cin) % 8) return AGP_E_BADARG;
    const int64_t total = (int64_t)(dgrad ? cin : cout) * kh * kw * ((dgrad ? cout : cin) / 8);
    hipLaunchKernelGGL(split_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout,
                       cin, kh, kw, dgrad, (bf16_t*)hi, (bf16_t*)lo, (bf16_t*)nullptr, (bf16_t*)nullptr, 0);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_split_conv_weight_both(const float* w, int cout, int cin, int kh, int kw, void* hi, void* lo, void* hi_d, void* lo_d,
                                          int chunk_major, void* stream) {
    if (!w || !hi || !lo || !hi_d || !lo_d || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || cin % 8 || cout % 8) return AGP_E_BADARG;
    if (((chunk_major & 1) && cin % 32) || ((chunk_major & 2) && cout % 32) || (chunk_major & ~3)) return AGP_E_BADARG;
    const int64_t total = (int64_t)cout * kh * kw * (cin / 8) + (int64_t)cin * kh * kw * (cout / 8);
    hipLaunchKernelGGL(split_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout,
                       cin, kh, kw, 2, (bf16_t*)hi, (bf16_t*)lo, (bf16_t*)hi_d, (bf16_t*)lo_d, chunk_major);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_pack_f32_to_nhwc4_h16(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n, int c, int h, int w,
                                        int pad, void* hi, void* lo, void* h16, void* stream) {
    if (!x || !hi || !h16 || c > 4 || n <= 0) return AGP_E_BADARG;
    if (!(sw == 1 && w % 4 == 0 && ((uintptr_t)x % 16) == 0 && sn % 4 == 0 && sc % 4 == 0 && sh % 4 == 0 &&
          (int64_t)n * h * (w / 4) < (1ll << 31)))
        return AGP_E_UNSUPPORTED;
    const int64_t threads = (int64_t)n * h * (w / 4);
    AGP_LAUNCH(pack_nchw4_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, sn, sc, sh, n, c, h,
               w, pad, make_fastdiv((uint32_t)(w / 4)), make_fastdiv((uint32_t)h), (bf16_t*)hi, (bf16_t*)lo, (bf16_t*)h16);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_pack_f32_to_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                    int n, int c, int h, int w, int cpad, int pad, void* hi,
                                    void* lo, void* stream) {
    if (!x || !hi || cpad < c || n <= 0) return AGP_E_BADARG;
    if (cpad == 4 && sw == 1 && c <= 4 && w % 4 == 0 && ((uintptr_t)x % 16) == 0 && sn % 4 == 0 && sc % 4 == 0 && sh % 4 == 0 &&
        (int64_t)n * h * (w / 4) < (1ll << 31)) {
        const int64_t threads = (int64_t)n * h * (w / 4);
        AGP_LAUNCH(pack_nchw4_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, sn, sc, sh, n, c,
                   h, w, pad, make_fastdiv((uint32_t)(w / 4)), make_fastdiv((uint32_t)h), (bf16_t*)hi, (bf16_t*)lo);
    } else if (cpad == 4) {
        AGP_LAUNCH(pack_kernel<4>, dim3(grid_for((int64_t)n * h * w, 256)), dim3(256), 0,
                           (hipStream_t)stream, x, sn, sc, sh, sw, n, c, h, w, cpad, pad,
                           (bf16_t*)hi, (bf16_t*)lo);
    } else if (cpad % 8 == 0) {
        AGP_LAUNCH(pack_kernel<8>, dim3(grid_for((int64_t)n * h * w * (cpad / 8), 256)),
                           dim3(256), 0, (hipStream_t)stream, x, sn, sc, sh, sw, n, c, h, w, cpad,
                           pad, (bf16_t*)hi, (bf16_t*)lo);
    } else {
        return AGP_E_BADARG;
    }
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_pack_u8_cams_to_nhwc(const uint8_t* img, int n, int ncam, int h, int w, const float* mean3,
                                        const float* std3, int pad, void* hi, void* lo, void* stream) {
    if (!img || !hi || !mean3 || !std3 || n <= 0 || ncam <= 0 || h <= 0 || w <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(pack_u8_cams_kernel, dim3(grid_for((int64_t)n * h * ncam * w, 256)), dim3(256), 0, (hipStream_t)stream, img, n,
               ncam, h, w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], pad, (bf16_t*)hi, (bf16_t*)lo);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_unpack_nhwc_to_f32(const void* hi, const void* lo, int n, int h, int w, int c,
                                      int pad, float* out, void* stream) {
    if (!hi || !out || c % 8 || n <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(unpack_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8), 256)), dim3(256),
                       0, (hipStream_t)stream, (const bf16_t*)hi, (const bf16_t*)lo, n, h, w, c, pad,
                       out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_map_zero_halo(void* hi, void* lo, int n, int h, int w, int c, int pad, void* stream) {
    if (!hi || c <= 0 || n <= 0 || h <= 0 || w <= 0 || pad < 0) return AGP_E_BADARG;
    if (pad == 0) return AGP_OK;
    const int64_t pixels = (int64_t)n * ((int64_t)2 * pad * (w + 2 * pad) + (int64_t)2 * pad * h);
    hipStream_t s = (hipStream_t)stream;
    if (c % 8 == 0) {
        AGP_LAUNCH(zero_halo_kernel<u32x4>, dim3(grid_for(pixels * (c / 8), 256)), dim3(256), 0, s, (bf16_t*)hi, (bf16_t*)lo, n, h, w, c, pad);
    } else if (c % 4 == 0) {
        AGP_LAUNCH(zero_halo_kernel<u32x2>, dim3(grid_for(pixels * (c / 4), 256)), dim3(256), 0, s, (bf16_t*)hi, (bf16_t*)lo, n, h, w, c, pad);
    } else {
        AGP_LAUNCH(zero_halo_kernel<bf16_t>, dim3(grid_for(pixels * c, 256)), dim3(256), 0, s, (bf16_t*)hi, (bf16_t*)lo, n, h, w, c, pad);
    }
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_maxpool3x3s2_fwd(const void* in_hi, const void* in_lo, int n, int hin, int win,
                                    int c, int pin, void* out_hi, void* out_lo, int hout, int wout,
                                    int pout, uint8_t* argmax, void* stream) {
    if (!in_hi || !out_hi || c % 8 || pin < 1 || n <= 0) return AGP_E_BADARG;
    if (hout != (hin + 2 - 3) / 2 + 1 || wout != (win + 2 - 3) / 2 + 1) return AGP_E_BADARG;
    if (argmax) {
        AGP_LAUNCH((maxpool_kernel<true>), dim3(grid_for((int64_t)n * hout * wout * (c / 8), 256)),
                           dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_hi, (const bf16_t*)in_lo,
                           n, hin, win, c, pin, (bf16_t*)out_hi, (bf16_t*)out_lo, hout, wout, pout, argmax);
    } else {
        AGP_LAUNCH((maxpool_kernel<false>), dim3(grid_for((int64_t)n * hout * wout * (c / 8), 256)),
                           dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_hi, (const bf16_t*)in_lo,
                           n, hin, win, c, pin, (bf16_t*)out_hi, (bf16_t*)out_lo, hout, wout, pout, argmax);
    }
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_bcast_add_fwd(const void* in_hi, const void* in_lo, const float* vec, int n, int h,
                                 int w, int c, int pin, void* out_hi, void* out_lo, int pout,
                                 void* stream) {
    if (!in_hi || !out_hi || !vec || c % 8 || n <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(bcast_add_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8), 256)), dim3(256),
                       0, (hipStream_t)stream, (const bf16_t*)in_hi, (const bf16_t*)in_lo, vec, n, h,
                       w, c, pin, (bf16_t*)out_hi, (bf16_t*)out_lo, pout);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
