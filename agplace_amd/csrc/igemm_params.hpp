// Parameters shared by the implicit-GEMM kernels (igemm.hip, igemm_d16.hip).
#pragma once
#include "common.hpp"

namespace agp_igemm {

enum { EPI_CONV = 0, EPI_GMIN = 1 };

struct IgemmParams {
    const void* x_hi; const void* x_lo; uint32_t x_bytes;
    const void* w_hi; const void* w_lo; uint32_t w_bytes;
    int M, N, Ktot;            // GEMM sizes (Ktot = KH*KW*CK elements per W row)
    int KW, CK, ntaps;         // taps and channels per tap
    FastDiv d_howo, d_wo;      // m -> (img, oy, ox)
    int x_sn, x_sh, x_sw, x_base, sy, sx;   // input strides (elements)
    void* o_hi; void* o_lo;
    int o_sn, o_sh, o_sw, o_base;           // output strides (elements), channel stride 1
    const void* r_hi; const void* r_lo;
    const float* scale; const float* shift;
    int relu;
    float* gmin; const float* wnorm; int gq_stride;   // GMIN epilogue
    const int* xrow_tab;       // sparse conv: per-tap gather table (see tap_stride)
    int tap_stride, tab_mul;   // sparse conv: xrow_tab is [ntaps][tap_stride] ROW indices, element offset = index * tab_mul
    const void* w2_hi; const float* scale2; const float* shift2; void* o2_hi;   // igemm_s2: the fused 1x1 / stride-2 downsample (or NULL)
    const int64_t* m_dev;      // sparse conv, capacity mode: device-side count of valid rows (tiles beyond it exit at once), or NULL
    const int* row_perm;       // sparse conv: GEMM row m computes output row row_perm[m] (a permutation of the valid rows that puts rows
                               // with the same absent taps into the same tiles: agp_sparse_zplane_perm), or NULL
    const uint32_t* tile_taps; // sparse conv (<= 32 taps): per 128 GEMM rows, the taps any of them has a neighbour for (agp_sparse_tile_taps), or NULL
    // fused conv + max-pool 3x3/2 (stem): conv map h1 x w1, pooled map h2 x w2, 7x7 pooled outputs per
    // workgroup from a 16x16 conv tile (rows/cols 14*t - 1 ...); o_* strides then address the POOLED map
    int pool_h1, pool_w1, pool_h2, pool_w2, pool_ty, pool_tx;
    float* stat_partial;       // optional [row tiles][2][N] per-tile channel sums / sums of squares (kxr kernel, bf16-pair maps)
    // backward-statistics mode of stat_partial (agp_conv_desc::bstat_*): the conv is a data-gradient conv whose output g is the
    // gradient at the OUTPUT y = relu?(BN(z) + r) of an earlier unit; the tile sums are (sum g*[y>0], sum g*[y>0]*zhat)
    const void* bs_z_hi; const void* bs_z_lo; const void* bs_y_hi; const float* bs_mean; const float* bs_rstd;
    float* pool_partial; const float* pool_p; float pool_eps;   // optional conv-epilogue pooling (agp_conv_desc::pool_partial), igemm_kxr2 only
    int pool_sq;               // ... stat 1 = the sum of squares instead of the GeM sum (agp_conv_desc::pool_stat = 1)
    int img_rows;              // kxr kernels: real raster rows per image (= d_howo.d unless the raster is padded for pooling)
    const void* w_cm;                 // optional chunk-major fp16 weights [Ktot/32][N][32] (agp_conv_desc::w_cm): kxr2, kxrw, s2 kernels
    const void* w2_cm;                // the same of the s2 kernel's 1x1 downsample weights (w2_hi)
    const void* w_cm_lo;              // the lo plane of the two-plane modes in the same order (igemm_kxr)
    const void* w_q8; int w_q8_exp;   // optional e4m3 lo plane of the F16W2 mode (agp_conv_desc::w_q8), kxr kernel only
    int dbg;                   // timing-only experiments (AGP_IGEMM_DBG), 0 in production
    int MT, NT, mt_chunk;      // tiles; mt_chunk = ceil(MT/8) row tiles per XCD
};

// The stem kernels' view of the network's INPUT when they read it themselves (no packed NHWC4 copy): kind 1 = an fp32
// [n][3][h][w] image with arbitrary element strides, kind 2 = uint8 camera tiles [n][ncam][h][wcam][3] (ToTensor + Normalize
// on the fly, the arithmetic of pack_u8_cams_kernel).
struct StemRaw {
    const void* x;
    long long sn, sc, sh, sw;       // fp32 image: element strides
    int h, w, ncam, wcam;           // image size; uint8 tiles: cameras per image, tile width (w = ncam * wcam)
    float m[3], s[3];
    uint32_t bytes;                 // stem_walk_kernel<1>: bytes spanned by the image batch (buffer descriptor)
};

// MFMA operand precision of a kernel instantiation (NPREC = AGP_PREC_*):
//   1 BF16   : x bf16,       w bf16            1 product
//   2 F16W2  : x fp16,       w fp16 hi+lo      2 products
//   3 BF16X3 : x bf16 hi+lo, w bf16 hi+lo      3 products
//   4 F16    : x fp16,       w fp16            1 product
template <int NPREC> struct PrecT {
    static constexpr int XPL = (NPREC == 3) ? 2 : 1;                  // X planes staged / loaded
    static constexpr int WPL = (NPREC == 3 || NPREC == 2) ? 2 : 1;    // W planes
    static constexpr int NPROD = (NPREC == 3) ? 3 : (NPREC == 2 ? 2 : 1);
    static constexpr bool F16 = (NPREC == 2 || NPREC == 4);
};

template <int NPREC>
__device__ __forceinline__ void mfma32(f32x16& acc, const bf16x8& wh, const bf16x8& wl, const bf16x8& xh,
                                       const bf16x8& xl) {
    if (NPREC == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc, 0, 0, 0);
    } else if (NPREC == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc, 0, 0, 0);
    } else {
        const f16x8 x = __builtin_bit_cast(f16x8, xh);
        if (NPREC == 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wl), x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh), x, acc, 0, 0, 0);
    }
}
template <int NPREC>
__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& wh, const bf16x8& wl, const bf16x8& xh,
                                       const bf16x8& xl) {
    if (NPREC == 3) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0);
    } else if (NPREC == 1) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0);
    } else {
        const f16x8 x = __builtin_bit_cast(f16x8, xh);
        if (NPREC == 2)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), x, acc, 0, 0, 0);
    }
}

constexpr int EPI_ROWB = 64 * 4 + 16;  // 64 fp32 channels + 16 B pad per pixel row

}  // namespace agp_igemm
