/*
 * agplace_hip.h -- C ABI of libagplace_hip.so (gfx950 / MI355X).
 *
 * The reference (sijieaaa/AGPlace) has no FFI layer: its hot path is eager PyTorch
 * (cuDNN / cuBLAS / ATen), torchdiffeq and faiss-cpu.  This header is the boundary
 * the MI355X build inserts BENEATH the reference's Python call signatures
 * (SURVEY.md section 8b).  Each entry point names the reference interface it
 * replaces (file:line under the reference tree).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching
 *    allocator in practice); the library allocates nothing and keeps no state;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*),
 *    never synchronises, and is therefore hipGraph-capturable;
 *  - return value: 0 = AGP_OK, otherwise an AGP_E_* code (the Python host raises
 *    RuntimeError, mirroring the reference's assert / NotImplementedError style,
 *    e.g. network_mm/ffns.py:62-63, network_mm/mm.py:170);
 *  - "split-bf16 plane pair": a tensor stored as two bf16 arrays hi, lo with
 *    value = float(hi) + float(lo), hi = rn_bf16(v), lo = rn_bf16(v - hi).
 *  - FEATURE MAPS (every `*_hi, *_lo` map argument) are NHWC with a zero halo of `pad`
 *    pixels on H and W, in one of two storage formats chosen by the pointers:
 *        lo != NULL : split-bf16 plane pair (~2^-17 relative), AGP_PREC_BF16X3
 *        lo == NULL : ONE fp16 plane (2^-12 relative, saturating at +-65504),
 *                     AGP_PREC_F16W2 / AGP_PREC_F16
 *  - conv WEIGHTS follow the conv's precision: bf16 hi/lo pair (BF16X3), fp16 hi/lo pair
 *    (F16W2) or one fp16 plane (F16); see agp_split_f32's `fmt`.
 */
#ifndef AGPLACE_HIP_H
#define AGPLACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGP_OK 0
#define AGP_E_BADARG 1      /* unsupported shape / enum / null pointer */
#define AGP_E_LAUNCH 2      /* hipLaunchKernel failed */
#define AGP_E_UNSUPPORTED 3 /* valid in the reference, not implemented here */

/* MFMA operand precision.
 *   BF16X3 (3): split-bf16 activations AND weights, hi*hi + hi*lo + lo*hi = three MFMA products,
 *               ~2^-17 relative operand error.  The training path runs in this mode.
 *   F16W2  (2): ONE fp16 activation plane, fp16 hi/lo weight pair, x*w_hi + x*w_lo = two products.
 *               Weights are exact to ~2^-22; activation rounding (2^-12) is independent per pixel
 *               and averages out in the pooled descriptors.  The host library's inference default
 *               (Options.mfma_precision): descriptors 3e-5 .. 1.6e-4, feature maps <= 6e-4 relative (bar: 1e-3).
 *   F16    (4): one fp16 plane each, ONE product: descriptors <= 4e-4, feature maps <= 8.5e-4.  What bench.py
 *               runs (the C3 configuration's 16-bit MFMA arithmetic) on its own kernel (igemm_kxr2.hip).
 *   fp16 maps (modes 2 and 4) saturate at +-65504; the host library counts saturated elements on the first
 *   inference forward after a weight load and warns (mode 3 has fp32 range).
 *   BF16   (1): plain bf16, one product; kNN coarse pass only (fails the 1e-3 bar for convs). */
#define AGP_PREC_BF16 1
#define AGP_PREC_F16W2 2
#define AGP_PREC_BF16X3 3
#define AGP_PREC_F16 4

/* 16-bit storage formats of agp_split_f32 */
#define AGP_FMT_BF16 0
#define AGP_FMT_F16 1

/* activations, reference network_mm/ffns.py:51-64 (select_act) */
#define AGP_ACT_ID 0
#define AGP_ACT_RELU 1
#define AGP_ACT_TANH 2
#define AGP_ACT_SIGMOID 3

/* fixed-grid solvers, torchdiffeq names used at network_mm/ffns.py:84 */
#define AGP_ODE_EULER 0
#define AGP_ODE_MIDPOINT 1
#define AGP_ODE_RK4 2 /* torchdiffeq 'rk4' = 3/8 rule */

const char* agp_version(void);
/* Returns the gfx arch string the device code was built for ("gfx950"). */
const char* agp_arch(void);

/* ---------------------------------------------------------------- layout */

/* Split an fp32 tensor into 16-bit hi/lo planes (elementwise), n elements:
 * hi = rn(x), lo = rn(x - hi) in bf16 (AGP_FMT_BF16) or saturating fp16 (AGP_FMT_F16).
 * lo may be NULL (hi only). */
int agp_split_f32(const float* x, void* hi, void* lo, int64_t n, int fmt, void* stream);
/* Training: conv weights straight from the parameter layout w[cout][cin][kh][kw] (fp32) to split-bf16 planes in the
 * kernels' layout, one launch.  dgrad = 0: [cout][kh][kw][cin] (agp_conv2d_fwd); dgrad = 1: the flipped, transposed
 * weights [cin][kh][kw][cout] of the data-gradient conv.  Channels of the output's innermost dimension % 8 == 0. */
int agp_split_conv_weight(const float* w, int cout, int cin, int kh, int kw, int dgrad, void* hi, void* lo, void* stream);
/* Both plane pairs of agp_split_conv_weight -- the forward conv's (hi, lo) and its data-gradient conv's (hi_d, lo_d) -- in ONE
 * launch (cin % 8 == 0 and cout % 8 == 0): a training step needs both for every conv, once per weight version. */
int agp_split_conv_weight_both(const float* w, int cout, int cin, int kh, int kw, void* hi, void* lo, void* hi_d, void* lo_d,
                               int chunk_major, void* stream);
/* chunk_major bit 0 / bit 1: the forward / the data-gradient pair is written in agp_conv_desc::w_cm's order [K/32][n][32]
 * (cin % 32 == 0 / cout % 32 == 0) -- for convs the 3x3 stride-1 kernel runs, which then take it as w_cm / w_cm_lo. */

/* fp32 image batch, arbitrary strides (elements) -> halo-padded NHWC split planes.
 * dst layout [n][h+2*pad][w+2*pad][cpad], channels >= c zero, halo untouched
 * (caller zeroes the buffer once).  Used for the stem input (cpad=4, pad=3;
 * reference network_mm/image_fe.py:98 feeds NCHW fp32 to conv1) and to import
 * fp32 feature maps at op-level drop-in boundaries. */
int agp_pack_f32_to_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                         int n, int c, int h, int w, int cpad, int pad,
                         void* hi, void* lo, void* stream);
/* The stem input of a TRAINING step: agp_pack_f32_to_nhwc with cpad = 4, and the same values once more as ONE fp16 plane `h16`
 * (same layout) -- agp_conv_desc::in_h16 of the stem's one-pass weight gradient (agp_conv2d_wgrad).  Unit pixel stride, w % 4 == 0
 * and 16-byte aligned rows only (AGP_E_UNSUPPORTED otherwise: pack without the plane, the weight gradient then runs three products). */
int agp_pack_f32_to_nhwc4_h16(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n, int c, int h, int w, int pad,
                              void* hi, void* lo, void* h16, void* stream);

/* Device-side input pipeline (reference datasets/datasets_ws_nuscenes.py:604-634):
 * uint8 HWC camera tiles [n][ncam][h][w][3] (decoded and resized on the host) -> ToTensor (/255),
 * Normalize(mean3, std3: HOST pointers to 3 floats), tiles concatenated along W, packed as the
 * stem's NHWC4 map [n][h+2*pad][ncam*w+2*pad][4] (halo untouched, 4th channel zero) in the map's
 * storage format (lo == NULL: one fp16 plane). */
int agp_pack_u8_cams_to_nhwc(const uint8_t* img, int n, int ncam, int h, int w, const float* mean3,
                             const float* std3, int pad, void* hi, void* lo, void* stream);

/* halo-padded NHWC split planes -> dense fp32 NHWC [n][h][w][c] (a torch
 * channels_last tensor of logical shape [n,c,h,w]). */
int agp_unpack_nhwc_to_f32(const void* hi, const void* lo, int n, int h, int w, int c,
                           int pad, float* out, void* stream);

/* Zero the halo (the `pad` outermost rows / columns of every image) of a map [n][h + 2 pad][w + 2 pad][c]: what a freshly
 * allocated map needs before the kernel that writes its interior (the halo is a convolution's zero padding; no kernel writes it). */
int agp_map_zero_halo(void* hi, void* lo, int n, int h, int w, int c, int pad, void* stream);

/* ------------------------------------------------------------ convolution */

/* One fused implicit-GEMM convolution on the MFMA pipes:
 *   out = relu?( conv(in, w) * scale[c] + shift[c] + residual )
 * Replaces one (Conv2d -> BatchNorm2d(eval) -> [+identity] -> [ReLU]) group of
 * torchvision's ResNet BasicBlock/Bottleneck as driven by the reference at
 * network_mm/image_fe.py:97-113 and of stage2fuse_blockadd.py:61-79 (BasicBlock,
 * conv bias folded into shift).
 * Geometry: input [n][hin+2*pin][win+2*pin][cin], weights [cout][kh][kw][cin]
 * (bf16 split planes, K-contiguous), output [n][hout+2*pout][wout+2*pout][cout].
 * `in_w_step` is the element distance between horizontally adjacent input pixels
 * (= cin normally; 4 for the packed stem, where cin=32 spans 8 pixels x 4 ch).
 * Requirements: cin % 32 == 0, cout % 64 == 0. */
typedef struct agp_conv_desc {
    const void* in_hi;  const void* in_lo;
    const void* w_hi;   const void* w_lo;
    void* out_hi;       void* out_lo;
    const void* res_hi; const void* res_lo;  /* residual (same geometry as out) or NULL */
    const float* scale; const float* shift;  /* per-cout fp32, NULL = 1 / 0 */
    int32_t n, hin, win, cin, pin, in_w_step;
    int32_t hout, wout, cout, pout;
    int32_t kh, kw, stride, pad;
    int32_t relu;
    int32_t prec;  /* AGP_PREC_* */
    /* Optional, F16W2 only (NULL = off): an e4m3 (OCP fp8) copy of the weight residual w - fp16(w), scaled by
     * 2^w_q8_exp, in the order agp_conv_w_q8_prepare writes it.  The 3x3 stride-1 kernel then runs the `lo`
     * product on the block-scaled fp8 MFMA (K = 64 per instruction, activations converted to e4m3 in registers):
     * the weights stay exact to 2^-16 instead of 2^-22, at 3/4 of the MFMA work.  Other convs ignore it. */
    const void* w_q8;
    int32_t w_q8_exp;
    /* Optional (BF16X3: every conv with agp_conv2d_stat_tiles(d) > 0; NULL = off): the kernel also writes, per row tile, the channel sums
     * and sums of squares of the values it stores -- [agp_conv2d_stat_tiles(d)][2][cout] floats, the layout of
     * agp_bn_stats' own first stage -- so that train-mode BatchNorm needs no extra pass over the conv output
     * (agp_bn_stats_from_partial). */
    float* stat_partial;
    /* Optional (AGP_PREC_F16 3x3 stride-1 convs, igemm_kxr2; NULL = off): the kernel also reduces the map it stores for the
     * global pooling that follows it in the network -- GeM / adaptive_avg_pool2d of a stage output (reference
     * network_mm/image_pooling.py:16, fuse_block_toshallow.py:82, stage2fuse_blockadd.py:201-206) -- so that no pass
     * re-reads the map: the kernel then gives every image a multiple of 64 rows of its raster (the extra rows are
     * computed and dropped) and writes, per 64-row block b, [b][stat 2][cout] floats -- stat 0 = sum of the stored
     * (fp16-rounded) values, stat 1 = sum of max(x, pool_eps)^p (written only when pool_p != NULL, p = *pool_p on the
     * device); image i owns blocks [i * bpi, (i + 1) * bpi), bpi = ceil(hout * (wout + 2) / 64).  Fixed, image-relative
     * summation order: bit-reproducible and independent of an image's position in the batch.  agp_pool_from_conv
     * finishes it.  Size: agp_conv2d_pool_blocks(d) * 2 * cout floats. */
    float* pool_partial;
    const float* pool_p;
    float pool_eps;
    /* AGP_PREC_BF16X3 3x3 stride-1 pad-1 convs on 1-pixel-halo maps (igemm_kxr; anything else: AGP_E_BADARG).  1 = multiply the HI
     * planes alone: one bf16 MFMA product (operands to ~2^-9 relative) instead of three; in_lo / w_lo / w_cm_lo are not read;
     * the accumulators (fp32), the residual, the statistics epilogues and the stored pair out_hi / out_lo are as in the
     * three-product form.  The training graph's opt-in fast data gradients (agplace_amd/train_graph.py, DGRAD_HI_ONLY). */
    int32_t hi_only;
    /* Optional (AGP_PREC_F16; 3x3 pad-1 convs of stride 1 or 2 and the 1x1 stride-2 downsample; NULL = off): the same fp16 weights as w_hi in CHUNK-MAJOR order
     * [kh*kw*cin / 32][cout][32] -- 32-channel chunk c of output channel n (K index 32*c .. 32*c+31 of w_hi's row n) at
     * element (c * cout + n) * 32.  The kernels stage a weight tile as 16 rows x 64 B per wave instruction; in w_hi's
     * [cout][K] order those are 16 half lines 2*K bytes apart, chunk-major they are 1 KB contiguous (measured: -4..5 % per
     * launch, profiles/README.md).  Kernels that do not take it read w_hi, which must always be set. */
    const void* w_cm;
    /* Optional, together with stat_partial (BF16X3 3x3 stride-1 convs; bstat_z_hi NULL = off): BACKWARD-statistics mode.  The
     * conv is a data-gradient conv whose output g (= conv + residual) is the gradient at the output y = relu?(BN(z) + r) of an
     * earlier unit, whose maps z / y have this conv's output geometry: stat_partial then receives, per row tile,
     * [2][cout] = (sum g*[y>0], sum g*[y>0]*(z - mean)*rstd) -- the first stage of that unit's BatchNorm backward
     * (agp_bn_bwd_from_partial), so the gradient is not read again to be summed.  bstat_y_hi NULL = no ReLU mask; bstat_z_lo
     * NULL = z is one fp16 plane (the library's map convention: lo == NULL means fp16), the z of a unit whose forward conv ran
     * as ONE fp16 product (Options.train_precision = 16).
     * Reference: the autograd of nn.BatchNorm2d in train.py:303-341. */
    const void* bstat_z_hi; const void* bstat_z_lo;
    const void* bstat_y_hi;
    const float* bstat_mean; const float* bstat_rstd;
    /* Optional: the lo plane in w_cm's chunk-major order.  With it, w_cm also serves the two-plane modes (AGP_PREC_F16W2,
     * AGP_PREC_BF16X3) of 3x3 stride-1 convs (igemm_kxr); without it those modes read w_hi / w_lo. */
    const void* w_cm_lo;
    /* Optional, WEIGHT GRADIENT only (agp_conv2d_wgrad*; 3x3 stride-1 pad-1 convs on 1-pixel-halo maps; both or neither):
     * in_h16 = the input map once more as ONE fp16 plane (agp_map_affine's o_h16; zero halo), out_absmax = max |g| per
     * output channel of the gradient map out_hi / out_lo as fp32 bit patterns (agp_bn_bwd's gz_absmax).  The weight
     * gradient then runs as ONE fp16 MFMA product instead of three bf16 ones: x from the fp16 plane, g converted in
     * registers to fp16 after a per-channel power-of-two scale that puts its maximum at 2^13..2^14 (fp16's mantissa, no
     * range problem; the scale is divided out of the result).  Emulated in the fp64 oracle (tools/grad_prec_emul.py):
     * 7e-4 of a weight gradient against 1e-5 for the three-product form, inside the 1e-3 bar; the forward and the data
     * gradient stay at three products.  The call zeroes out_absmax for the next step.  Reference: the autograd of
     * nn.Conv2d in train.py:337-341. */
    const void* in_h16;
    uint32_t* out_absmax;
    /* With pool_partial: 0 = stat 1 is the GeM sum described there (written when pool_p != NULL); 1 = stat 1 is the SUM OF SQUARES
     * of the stored (fp16-rounded) values (pool_p, pool_eps ignored) -- with stat 0, the first stage of a train-mode BatchNorm's
     * statistics over the conv's output (blocks [0, n * ceil(hout * (wout + 2) / 64)) are written; agp_bn_stats_from_partial
     * finishes them): the fast training mode's one-product forward convs (agplace_amd/train_graph.py, FWD_F16) need no pass
     * over z for them. */
    int32_t pool_stat;
    int32_t reserved0;
} agp_conv_desc;
int agp_conv2d_fwd(const agp_conv_desc* d, void* stream);

/* `n` convolutions as ONE launch where the kernel allows it: 2..4 AGP_PREC_F16 3x3 / stride-1 / pad-1 convs on
 * 1-pixel-halo fp16 maps that share (cin, cout) -- e.g. the query network's (network_mm/image_fe.py:97-113) and the
 * database network's (network/image_fe.py:112-128) conv of the same ResNet layer, which the reference issues as two
 * cuDNN calls from two modules (train.py:308,316; test.py:128,161).  Images, map sizes, weights, scale/shift,
 * residual and relu are per problem; the tiles of all problems form one grid.  Any other group runs as `n` calls of
 * agp_conv2d_fwd in order.  Results are bit-identical to separate launches. */
int agp_conv2d_fwd_grouped(const agp_conv_desc* descs, int n, void* stream);

/* Row tiles of agp_conv_desc::stat_partial for `d`, or 0 when the kernel that runs `d` cannot produce it. */
int agp_conv2d_stat_tiles(const agp_conv_desc* d);

/* 64-row blocks of agp_conv_desc::pool_partial for `d` (the buffer holds blocks * 2 * cout floats), or 0 when the kernel
 * that runs `d` cannot pool in its epilogue (the caller then pools the stored map with agp_pool_fwd). */
int agp_conv2d_pool_blocks(const agp_conv_desc* d);
/* Second stage of the conv-epilogue pooling: mean[n][c] = sum / (h*w), gem[n][c] = (sum_p / (h*w))^(1/p) from the
 * block partials of a conv whose OUTPUT map is n x h x w x c (agp_conv_desc::pool_partial).  Either output may be NULL
 * (gem needs p, the device pointer the conv was given). */
int agp_pool_from_conv(const float* partial, int n, int h, int w, int c, const float* p, float* mean_out,
                       float* gem_out, void* stream);

/* A whole BasicBlock on 64-channel fp16 maps as ONE kernel (AGP_PREC_F16 arithmetic; csrc/fblock64.hip):
 *   out = relu( conv3x3(relu(conv3x3(in, w1) * scale1 + shift1), w2) * scale2 + shift2 + in )
 * = torchvision's BasicBlock(64, 64) of ResNet18/34 layer1 with eval-mode BatchNorm folded, as the reference runs it at
 * network_mm/image_fe.py:102 (query trunk) and network/image_fe.py:117 (database trunk).  The intermediate map stays in
 * LDS and the residual is taken from the staged input rows: the block reads `in` once and writes `out` once.
 * form 0 (default): v_mfma_f32_16x16x32_f16 tiles (the MFMA-bound loop holds a higher clock on this shape), bit-identical to two
 * agp_conv2d_fwd launches of the 3x3 kernel's own 16x16x32 variant; form 1: v_mfma_f32_32x32x16_f16 tiles, bit-identical to two
 * default agp_conv2d_fwd launches (prec AGP_PREC_F16) with an fp16 map in between.  The two forms differ by the order in which
 * the hardware adds the products of a K-step (fp32 rounding of the accumulation).
 * in / out: [n][h+2][w+2][64] fp16 planes with a zero 1-pixel halo (out's halo is not written), h even;
 * w1 / w2: fp16 [64][3][3][64] (the w_hi plane of agp_conv_desc); scale / shift: 64 floats each.
 * pool_partial (optional, else NULL): agp_bblock64_pool_floats(d) floats; the kernel also writes the channel sums of the
 * stored output per pair of map rows and column strip -- [n (h+2)/2][ceil(w/28)][64], image-relative units in a fixed
 * order: bit-reproducible and independent of an image's position in the batch -- for the level mean that follows the
 * stage (fuse_block_toshallow.py:82); agp_bblock64_pool_finish adds them up.  One launch runs ONE form: descs[0].form. */
typedef struct agp_bblock64_desc {
    const void* in; void* out;
    const void* w1; const void* w2;
    const float* scale1; const float* shift1;
    const float* scale2; const float* shift2;
    float* pool_partial;
    int32_t n, h, w, form;
} agp_bblock64_desc;
/* 1..4 blocks (e.g. the query and the database trunk's block of one layer) as one launch. */
int agp_bblock64_fwd_grouped(const agp_bblock64_desc* descs, int n, void* stream);
int64_t agp_bblock64_pool_floats(const agp_bblock64_desc* d);
/* mean_out[n][64] = sum of an image's pair sums / (h w) */
int agp_bblock64_pool_finish(const float* partial, int n, int h, int w, float* mean_out, void* stream);

/* Builds agp_conv_desc::w_q8 for a 3x3 conv (cin % 64 == 0) from the fp32 weights w[cout][3][3][cin]:
 * q8 = cout*9*cin bytes, plane[n][pair][lh][tap][ks][e] = e4m3((w - fp16(w)) * 2^exp) of channel
 * 32*cc + 16*ks + 8*lh + e at the tap of phase 2*pair + tap, phases in the kernel's order (ky, cc, kx);
 * exp (written to HOST memory) = the largest power of two that keeps the plane inside e4m3's range.
 * Prepare-time helper: synchronises `stream`. */
int agp_conv_w_q8_prepare(const float* w, int cout, int cin, void* q8, int32_t* exp_host, void* stream);

/* The ResNet stem in one kernel (inference, fp16 maps): packed 7x7/2 conv + folded BatchNorm + ReLU +
 * MaxPool2d(3, 2, 1) (reference network_mm/image_fe.py:98-101).  `d` as for agp_conv2d_fwd's stem
 * (cin = 32, in_w_step = 4, kh = 7, kw = 1, stride 2, pad 3, pin 3, cout = 64, relu = 1, no residual,
 * prec F16W2 or F16) but out_* / hout / wout describe the POOLED map: the full-resolution stem map is
 * never written. */
int agp_stem_pool_fwd(const agp_conv_desc* d, void* stream);
/* The same kernel reading the network's INPUT itself, so that no packed NHWC4 copy of the image is written and re-read
 * (agp_pack_f32_to_nhwc / agp_pack_u8_cams_to_nhwc + agp_stem_pool_fwd in one launch; prec F16 only).  `d` as above
 * with in_hi = the raw input, hin / win = the image size (win = ncam * tile width for kind 2), pin ignored.
 *   kind 1: fp32 image [n][3][h][w], element strides sn, sc, sh, sw (reference network_mm/image_fe.py:98 feeds exactly this)
 *   kind 2: uint8 camera tiles [n][ncam][h][win/ncam][3] (HWC, contiguous): ToTensor + Normalize(mean3, std3: HOST
 *           pointers) + width-concat on the fly (reference datasets/datasets_ws_nuscenes.py:608-634) */
int agp_stem_pool_raw_fwd(const agp_conv_desc* d, int kind, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int ncam,
                          const float* mean3, const float* std3, void* stream);

/* MaxPool2d(kernel 3, stride 2, padding 1) on post-ReLU (>= 0) maps, so the zero
 * halo is the padding value.  argmax (optional, training): uint8 [n][hout][wout][c] = window position
 * 3*ky + kx of the first maximum (torch's tie rule), consumed by agp_maxpool3x3s2_bwd.
 * Reference: torchvision ResNet.maxpool via
 * network_mm/image_fe.py:101. */
int agp_maxpool3x3s2_fwd(const void* in_hi, const void* in_lo, int n, int hin, int win, int c,
                         int pin, void* out_hi, void* out_lo, int hout, int wout, int pout,
                         uint8_t* argmax, void* stream);

/* out = in + vec[n][c] broadcast over H,W (interior only).  Reference:
 * stage2fuse_blockadd.py:195 (imgmap + fusevec_img.unsqueeze(-1).unsqueeze(-1)). */
int agp_bcast_add_fwd(const void* in_hi, const void* in_lo, const float* vec, int n, int h,
                      int w, int c, int pin, void* out_hi, void* out_lo, int pout, void* stream);

/* ----------------------------------------------------------------- pooling */

/* Per (image, channel) reductions over H*W of a halo-padded NHWC split map in one
 * HBM pass: mean[n][c] = avg(x)            (fuse_block_toshallow.py:82,
 *                                            stage2fuse_blockadd.py:206)
 *           gem[n][c]  = (avg(max(x,eps)^p))^(1/p)  (network_mm/image_pooling.py:16)
 * `p` is a device pointer to the 1-element GeM exponent.  Either output may be NULL.
 * `partial` is caller workspace of agp_pool_workspace_floats(n,c,h,w) floats. */
int64_t agp_pool_workspace_floats(int n, int c, int h, int w);
int agp_pool_fwd(const void* hi, const void* lo, int n, int h, int w, int c, int pad,
                 const float* p, float eps, float* mean_out, float* gem_out, float* partial,
                 void* stream);
/* Same reductions on a dense fp32 tensor with arbitrary strides (op-level GeM
 * drop-in on torch tensors). */
int agp_pool_f32_fwd(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n,
                     int c, int h, int w, const float* p, float eps, float* mean_out,
                     float* gem_out, float* partial, void* stream);
/* GeM backward: dL/dx (dense fp32, same strides as x) and dL/dp.  y = gem output [n][c], gy = dL/dy [n][c].
 * gp (here and in agp_pool_bwd / agp_seg_pool_bwd): a buffer of AGP_GP_FLOATS floats the caller zeroes ONCE; [0] receives
 * dL/dp (+=), the rest is scratch of a fixed-order sum over the grid (the same bits every run) that every call leaves ready
 * for the next one. */
#define AGP_GP_FLOATS (1 + 16384 + 1)
int agp_gem_f32_bwd(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n,
                    int c, int h, int w, const float* p, float eps, const float* y,
                    const float* gy, float* gx, float* gp, void* stream);

/* ------------------------------------------------- fusion MLPs and Neural ODE */

/* y = act(x W^T + b) for a [b,k] fp32 matrix, W [n][k] as split planes; n % 256 == 0,
 * k % 32 == 0.  Replaces nn.Linear (+ select_act) call sites:
 * fuse_block_toshallow.py:24-25, stage2fuse_blockadd.py:152-155, mm.py:58,
 * dbvanilla2d.py:21,24.  Optional fused pre-add: x := x + add1 + add2 (NULL ok). */
int agp_linear_fwd(const float* x, const float* add1, const float* add2, const void* w_hi,
                   const void* w_lo, const float* bias, int b, int k, int n, int act,
                   float* y, void* stream);

/* FCODE forward: y(1) of dy/dt = act(y W^T + b), y(0) = x (+ add1 + add2), by the
 * fixed-grid `method` over `nsteps` steps with step sizes dt[0..nsteps) (host array,
 * copied into the launch).  D = 256 only.  One persistent workgroup per 16 batch rows
 * keeps W's MFMA fragments in registers for all steps.  Replaces
 * network_mm/ffns.py:78-87 (FCODE.forward -> torchdiffeq.odeint(...)[-1]).
 * If `traj` != NULL the solver state and every stage derivative are recorded for the backward
 * pass: traj[s][0][b][256] = y before step s, traj[s][1+i][b][256] = k_i of step s
 * (i < stages: euler 1, midpoint 2, rk4 4); agp_fcode_traj_floats(b, method, nsteps) floats. */
int64_t agp_fcode_traj_floats(int b, int method, int nsteps);
int agp_fcode_fwd(const float* x, const float* add1, const float* add2, const void* w_hi,
                  const void* w_lo, const float* bias, int b, int act, int method,
                  const float* dt, int nsteps, float* y, float* traj, void* stream);

/* FCODE backward (discretise-then-optimise, like autograd through odeint at ffns.py:84): given
 * gy = dL/dy(1) and the recorded trajectory, produce gx = dL/dx (also the gradient of add1/add2)
 * and OVERWRITE gw[256][256] = dL/dW, gb[256] = dL/db.  wt_hi/wt_lo are the split planes of W^T.
 * Stage inputs are rebuilt from the recorded y and k_i; the adjoint products g_z W run on the MFMA
 * pipes with W^T fragments resident in registers; dW is one [256 x R] x [R x 256] GEMM over all
 * R = nsteps*stages*b recorded (g_z, stage-input) pairs.  `workspace`:
 * agp_fcode_bwd_workspace_bytes(b, method, nsteps) bytes. */
int64_t agp_fcode_bwd_workspace_bytes(int b, int method, int nsteps);
int agp_fcode_bwd(const float* traj, const float* gy, const void* wt_hi, const void* wt_lo, int b,
                  int act, int method, const float* dt, int nsteps, float* gx, float* gw, float* gb,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* Linear backward for y = act(x W^T + b): gz = gy * act'(y) (y = forward output, NULL for act id);
 * gx[b][k] = gz W (wt planes = split W^T, k % 256 == 0 after padding by the caller), and
 * gw[n][k] = gz^T x, gb[n] = sum_b gz (overwritten).  workspace: agp_linear_bwd_workspace_bytes. */
int64_t agp_linear_bwd_workspace_bytes(int b, int k, int n);
int agp_linear_bwd(const float* x, const float* y, const float* gy, const void* wt_hi,
                   const void* wt_lo, int b, int k, int n, int act, float* gx, float* gw, float* gb,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* LayerNorm / L2-normalise backward (row-wise, fp32).  LayerNorm: forward was
 * y = relu?(LN(x)*g + beta + res); gy is dL/dy; outputs gx, gres (= masked gy, may alias NULL),
 * and ggamma / gbeta (assigned; summed over the rows in a fixed order by a second launch: the same bits every run). */
int agp_layernorm_bwd(const float* x, const float* gamma, const float* y, const float* gy, int b, int d,
                      float eps, int relu, float* gx, float* gres, float* ggamma, float* gbeta,
                      void* stream);
int agp_l2normalize_bwd(const float* x, const float* gy, int b, int d, float* gx, void* stream);

/* Row-wise ops on [b][d] fp32 (d <= 4096):
 * LayerNorm(eps) with affine, optional ReLU, optional residual add before the ReLU:
 *   y = relu?( LN(x) * g + beta + res )      stage2fuse_blockadd.py:91-100, dbvanilla2d.py:22-23 */
int agp_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* res,
                      int b, int d, float eps, int relu, float* y, void* stream);
/* y = x / max(||x||_2, 1e-12)      F.normalize at mm.py:83,91,103; dbvanilla2d.py:82 */
int agp_l2normalize_fwd(const float* x, int b, int d, float* y, void* stream);

/* y[i] = sum_t w_t * x_t[i] over up to 6 fp32 vectors of n elements.  x_t == NULL ends the
 * list; w_t is a DEVICE pointer to a 1-element weight (NULL = 1.0).  Replaces the scalar-weight
 * glue of MM.forward_q (mm.py:84,92,104,123-138: vec * self.xxx_weight, sum(finaloutput)) and
 * the vector adds of the fusion blocks (fuse_block_toshallow.py:115, stage2fuse_blockadd.py:212). */
int agp_wsum_fwd(const float* x0, const float* x1, const float* x2, const float* x3,
                 const float* x4, const float* x5, const float* w0, const float* w1,
                 const float* w2, const float* w3, const float* w4, const float* w5, int64_t n,
                 float* y, void* stream);
/* out[0] = sum_i a[i] * b[i] (fp32, fixed-order reduction): the gradient of one of MM's scalar mixing weights when the
 * reference's xxx_learnweight flags make it a trained parameter (tools/options.py:139-146; forward at
 * network_mm/mm.py:84,92,104,123-138): dL/dw_t = <dL/dy, x_t>. */
int agp_dot_f32(const float* a, const float* b, int64_t n, float* out, void* stream);

/* The whole vector path of a forward as ONE launch (agplace_amd/csrc/vecprog.hip): a program of at most
 * AGP_VECPROG_MAXOPS row-wise operations on [b, 256] fp32 vectors held in AGP_VECPROG_NREG on-chip registers, replacing the
 * per-op launches above for inference (reference network_mm/mm.py:91-129 after the backbones -- F.normalize,
 * FuseBlockToShallow's up-dim Linears and FCODE blocks, the stage-2 projections, Basic's fc-LayerNorm-ReLU-fc-LayerNorm,
 * stg2fusefc, the weighted final sum; models_baseline/dbvanilla2d.py:17-28,81-92).  Same arithmetic as agp_linear_fwd /
 * agp_fcode_fwd / agp_layernorm_fwd / agp_l2normalize_fwd / agp_wsum_fwd (split-bf16 x3 products, fp32 state).
 *   AGP_VP_LOAD       dst <- p[0] (fp32 [b][k], k <= 256, k % 4 == 0; features >= k are zero) (* p[1][0] when p[1] != NULL)
 *   AGP_VP_STORE      p[0] (fp32 [b][256]) <- r[0]
 *   AGP_VP_LINEAR     dst <- act(W x + bias), x = r[0] + r[1] + r[2] (r[1], r[2] optional: -1; only the first k features of x
 *                     enter); W = bf16 planes p[0] (hi), p[1] (lo) of [256][k] (agp_split_f32) in FRAGMENT-MAJOR order:
 *                     element [w][ks][q][row][e] = W[16 w + row][32 ks + 8 q + e] for w < 16, ks < k / 32, q < 4, row < 16,
 *                     e < 8 (a wave instruction of the kernel then reads 1 KB of consecutive bytes); bias p[2] or NULL, k % 32 == 0
 *   AGP_VP_FCODE      dst <- odeint(y' = act(W y + bias), y0 = r[0] + r[1] + r[2]) on the program's fixed grid; W [256][256],
 *                     fragment-major like AGP_VP_LINEAR's
 *   AGP_VP_L2NORM     dst <- r[0] / max(|r[0]|_2, 1e-12)
 *   AGP_VP_LAYERNORM  dst <- relu?(LayerNorm(r[0]) * p[0] + p[1] + r[1]), eps = f0, relu = act != 0, r[1] optional
 *   AGP_VP_WSUM       dst <- sum_{t < n} p[t][0] * r[t]   (weight 1 when p[t] == NULL), n <= 6, summed in order */
#define AGP_VECPROG_MAXOPS 36
#define AGP_VECPROG_NREG 6
#define AGP_VP_LOAD 1
#define AGP_VP_STORE 2
#define AGP_VP_LINEAR 3
#define AGP_VP_FCODE 4
#define AGP_VP_L2NORM 5
#define AGP_VP_LAYERNORM 6
#define AGP_VP_WSUM 7
typedef struct agp_vecprog_op {
    int op, dst, r[6], k, act, aux, n;
    float f0;
    int pad;
    const void* p[6];
} agp_vecprog_op;
/* ode_method / ode_dt (HOST pointer to ode_nsteps <= 48 step sizes): the grid of every AGP_VP_FCODE op of the program. */
int agp_vecprog_run(const agp_vecprog_op* ops, int nops, int b, int ode_method, const float* ode_dt, int ode_nsteps,
                    void* stream);
/* Two independent programs in ONE launch, each on workgroups of its own (nops_a + nops_b <= AGP_VECPROG_MAXOPS; program B has
 * b_b rows and no AGP_VP_FCODE unless it shares A's grid): the database network's head (models_baseline/dbvanilla2d.py:81-92)
 * beside the query network's (network_mm/mm.py:91-129) -- both are latency chains on a few CUs. */
int agp_vecprog_run2(const agp_vecprog_op* ops_a, int nops_a, int b_a, const agp_vecprog_op* ops_b, int nops_b, int b_b,
                     int ode_method, const float* ode_dt, int ode_nsteps, void* stream);

/* ------------------------------------------------------------- training path */
/* (the reference trains with plain autograd through cuDNN conv / BatchNorm, train.py:337-341) */

/* u[2*oy][2*ox] = g[oy][ox], zeros elsewhere: turns the data gradient of a stride-2 conv into a
 * stride-1 conv with flipped weights. */
int agp_upsample2_zero(const void* g_hi, const void* g_lo, int n, int ho, int wo, int c, int gpad,
                       void* u_hi, void* u_lo, int hu, int wu, int upad, void* stream);

/* Train-mode BatchNorm2d over a map: batch mean / rstd (biased variance) + running-stat update with
 * `momentum` (unbiased variance), as torch.nn.BatchNorm2d does.  workspace:
 * agp_train_reduce_workspace_floats(n,h,w,c) floats. */
int64_t agp_train_reduce_workspace_floats(int n, int h, int w, int c);
int agp_bn_stats(const void* z_hi, const void* z_lo, int n, int h, int w, int c, int pad, float eps,
                 float momentum, float* mean, float* rstd, float* running_mean, float* running_var,
                 const float* gamma, const float* beta, float* scale, float* shift, float* workspace,
                 void* stream);   /* scale = gamma*rstd, shift = beta - mean*scale (for agp_map_affine) */
/* The second stage of agp_bn_stats alone, on per-tile partials [tiles][2][c] written by a conv
 * (agp_conv_desc::stat_partial); `count` = n*h*w pixels per channel. */
int agp_bn_stats_from_partial(const float* partial, int tiles, int c, int64_t count, float eps, float momentum,
                              float* mean, float* rstd, float* running_mean, float* running_var,
                              const float* gamma, const float* beta, float* scale, float* shift, void* stream);
/* out = relu?(a * scale[c] + shift[c] + r)   (BatchNorm apply with optional residual).
 * o_h16 (optional, else NULL): the output once more as ONE fp16 plane of the same geometry (halo not written) -- the
 * input operand of the one-pass weight gradient (agp_conv_desc::in_h16) of the conv that consumes this map. */
int agp_map_affine(const void* a_hi, const void* a_lo, const float* scale, const float* shift,
                   const void* r_hi, const void* r_lo, int n, int h, int w, int c, int pad, int relu,
                   void* o_hi, void* o_lo, void* o_h16, void* stream);
/* BatchNorm backward through y = relu?(BN(z) + res): gz (pre-BN gradient), optional gres (= masked
 * gy, the residual branch), ggamma, gbeta (overwritten).
 * gz_absmax (optional, else NULL): c words that receive max |gz| per channel as fp32 BIT PATTERNS, folded in with an
 * integer atomic max (exact, order-independent) -- the caller zeroes them before the step; the one-pass weight gradient
 * (agp_conv_desc::out_absmax) derives its per-channel power-of-two operand scale from them and zeroes them again. */
int agp_bn_bwd(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo, const void* y_hi,
               const void* y_lo, const float* mean, const float* rstd, const float* gamma, int n, int h,
               int w, int c, int pad, int relu, void* gz_hi, void* gz_lo, void* gres_hi, void* gres_lo,
               float* ggamma, float* gbeta, float* workspace, uint32_t* gz_absmax, void* stream);
/* agp_bn_bwd with its channel sums already reduced per tile by the conv that produced gy (agp_conv_desc::bstat_*):
 * `partial` = [tiles][2][c].  frozen != 0: eval-mode statistics (agp_bn_bwd_frozen). */
int agp_bn_bwd_from_partial(const float* partial, int tiles, const void* z_hi, const void* z_lo, const void* gy_hi,
                            const void* gy_lo, const void* y_hi, const void* y_lo, const float* mean, const float* rstd,
                            const float* gamma, int n, int h, int w, int c, int pad, int relu, int frozen, void* gz_hi,
                            void* gz_lo, void* gres_hi, void* gres_lo, float* ggamma, float* gbeta, uint32_t* gz_absmax,
                            void* stream);
/* Synchronised BatchNorm under data parallelism (statistics over every rank's samples; reference
 * model/sync_batchnorm/batchnorm.py:121-166, train.py:253-256).  The library never communicates: it
 * hands out the LOCAL sums as fp64 [2c + 1] = (sum, sum of squares, count), the host all-reduces that
 * vector (RCCL), and the statistics / the backward's coefficients come from the reduced vector.
 *   forward : agp_bn_sums (reduction pass) or agp_bn_sums_from_partial (a conv epilogue's per-tile
 *             sums) -> all-reduce -> agp_bn_stats_from_sums (= agp_bn_stats' second stage)
 *   backward: agp_bn_bwd_sums: sums[2c] = (sum g, sum g*zhat) in fp64, and the LOCAL ggamma / gbeta
 *             (the parameter gradients of this rank's samples, averaged over ranks like every other
 *             gradient) -> all-reduce of sums -> agp_bn_bwd_apply with the reduced sums and the
 *             forward's reduced count (a device pointer: sums_fwd + 2c). */
int agp_bn_sums(const void* z_hi, const void* z_lo, int n, int h, int w, int c, int pad, double* sums,
                float* workspace, void* stream);
int agp_bn_sums_from_partial(const float* partial, int tiles, int c, int64_t count, double* sums,
                             void* stream);
int agp_bn_stats_from_sums(const double* sums, int c, float eps, float momentum, float* mean, float* rstd,
                           float* running_mean, float* running_var, const float* gamma, const float* beta,
                           float* scale, float* shift, void* stream);
int agp_bn_bwd_sums(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo,
                    const void* y_hi, const void* y_lo, const float* mean, const float* rstd, int n, int h,
                    int w, int c, int pad, int relu, double* sums, float* ggamma, float* gbeta,
                    float* workspace, void* stream);
int agp_bn_bwd_apply(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo,
                     const void* y_hi, const void* y_lo, const float* mean, const float* rstd,
                     const float* gamma, const double* sums, const double* count, int n, int h, int w,
                     int c, int pad, int relu, void* gz_hi, void* gz_lo, void* gres_hi, void* gres_lo,
                     float* workspace, void* stream);
/* Eval-mode BatchNorm inside a gradient graph (fine-tuning on frozen statistics; torch's F.batch_norm with
 * training=False under autograd): mean = running_mean, rstd = 1/sqrt(running_var + eps), scale = gamma*rstd,
 * shift = beta - mean*scale; the running statistics are not written. */
int agp_bn_frozen_coeffs(const float* running_mean, const float* running_var, const float* gamma,
                         const float* beta, int c, float eps, float* mean, float* rstd, float* scale,
                         float* shift, void* stream);
/* agp_bn_bwd with the statistics held constant: gz = gamma*rstd*g (g = gy masked by y>0 when relu),
 * ggamma = sum g*(z-mean)*rstd, gbeta = sum g. */
int agp_bn_bwd_frozen(const void* z_hi, const void* z_lo, const void* gy_hi, const void* gy_lo,
                      const void* y_hi, const void* y_lo, const float* mean, const float* rstd,
                      const float* gamma, int n, int h, int w, int c, int pad, int relu, void* gz_hi,
                      void* gz_lo, void* gres_hi, void* gres_lo, float* ggamma, float* gbeta,
                      float* workspace, uint32_t* gz_absmax, void* stream);
/* out[c] = sum over pixels of a map (conv-bias gradient). */
int agp_map_chan_sum(const void* a_hi, const void* a_lo, int n, int h, int w, int c, int pad, float* out,
                     float* workspace, void* stream);
/* out = a * [mask > 0]? + b?   (gradient accumulation / ReLU masking on maps) */
int agp_map_add(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo,
                const void* mask_hi, const void* mask_lo, int n, int h, int w, int c, int pad, void* o_hi,
                void* o_lo, void* stream);
int agp_maxpool3x3s2_bwd(const uint8_t* argmax, const void* gy_hi, const void* gy_lo, int n, int hin,
                         int win, int c, int pin, int hout, int wout, int pout, void* gx_hi, void* gx_lo,
                         void* stream);
/* agp_maxpool3x3s2_bwd fused with the BatchNorm backward (agp_bn_bwd / agp_bn_bwd_frozen) of the conv -> BN -> ReLU unit under
 * the pool -- the ResNet stem (reference network_mm/image_fe.py:97-103 under autograd): gp is the gradient at the POOLED map
 * [n][hout+2*pout][wout+2*pout][c]; z / y / gz are the unit's maps [n][h+2*pad][w+2*pad][c].  The gradient at the unit's
 * output is never materialised (each element gathers it from the windows that recorded it as their maximum).  The ReLU mask
 * comes from y, or -- y_hi NULL, the forward was agp_affine_maxpool3x3s2_fwd and y never existed -- from
 * fma(z, scale, shift) > 0, the forward's own arithmetic.  pv (optional, with ReLU): the POOLED map of the forward and the
 * BatchNorm's beta -- the channel sums are then taken over the pooled elements alone (pooled > 0 <=> the mask at the argmax, and
 * zhat = (pooled - beta) / gamma there; channels with a gamma too small for that division read z at the argmax).  Needs
 * 256 % (c / 8) == 0 (AGP_E_UNSUPPORTED otherwise: use the separate calls); workspace as agp_bn_bwd. */
int agp_maxpool_bn_bwd(const uint8_t* argmax, const void* gp_hi, const void* gp_lo, int hout, int wout, int pout,
                       const void* z_hi, const void* z_lo, const void* y_hi, const void* y_lo, const float* mean,
                       const float* rstd, const float* gamma, const float* scale, const float* shift, const void* pv_hi,
                       const void* pv_lo, const float* beta, int n, int h, int w, int c, int pad, int relu, int frozen,
                       void* gz_hi, void* gz_lo, float* ggamma, float* gbeta, float* workspace, uint32_t* gz_absmax, void* stream);
/* (gz_absmax, optional: as agp_bn_bwd's -- max |gz| per channel for the stem's one-pass weight gradient, agp_conv_desc::out_absmax.) */
/* pooled = MaxPool2d(3, 2, 1)(relu(z * scale[c] + shift[c])) with the argmax of agp_maxpool3x3s2_fwd, in ONE pass over the conv
 * output z: BatchNorm apply + ReLU + max-pool of the ResNet stem in training (reference network_mm/image_fe.py:97-103) without
 * storing the full-size activation (agp_map_affine + agp_maxpool3x3s2_fwd: 13 bytes per element of the step's largest map,
 * here 5).  out_h16 (optional, else NULL): the pooled map once more as one fp16 plane (agp_map_affine's o_h16). */
int agp_affine_maxpool3x3s2_fwd(const void* z_hi, const void* z_lo, const float* scale, const float* shift, int n, int h,
                                int w, int c, int pad, void* out_hi, void* out_lo, int hout, int wout, int pout,
                                uint8_t* argmax, void* out_h16, void* stream);
/* Backward of agp_pool_fwd into a map gradient: o = b? + gmean/HW + ggem * dGeM/dx.
 * gp (optional, AGP_GP_FLOATS floats zeroed once by the caller: see agp_gem_f32_bwd): [0] += dL/dp of the GeM exponent, summed
 * over the grid in a fixed order (reference: autograd through GeM.forward, network_mm/image_pooling.py:14-16). */
int agp_pool_bwd(const void* x_hi, const void* x_lo, const float* gmean, const float* ggem,
                 const float* gem_y, const float* p, float eps, const void* b_hi, const void* b_lo, int n,
                 int h, int w, int c, int pad, void* o_hi, void* o_lo, float* gp, void* stream);

/* Convolution weight gradient straight from the NHWC maps (no transposed copies):
 *   gw[kh][kw][cin][cout] (fp32) = sum over pixels of  in[...]^T * gout[...]
 * `d` describes the forward conv (reference: autograd of nn.Conv2d, train.py:337-341): in_* = its
 * input map, out_* = the GRADIENT w.r.t. its output (halo 1, zero), w_* / res_* / scale / shift
 * unused; prec must be AGP_PREC_BF16X3.  Supported: 3x3 (stride 1 or 2), 1x1 (stride 1 or 2) and the
 * packed 7x7 stem (cin = 32, in_w_step = 4, kw = 1: gw is [7][1][32][cout], element 4*kx + c).
 * Split-K partials live in the caller's workspace (agp_conv2d_wgrad_workspace_bytes; -1 = shape
 * not supported). */
int64_t agp_conv2d_wgrad_workspace_bytes(const agp_conv_desc* d);
int agp_conv2d_wgrad(const agp_conv_desc* d, float* gw, void* workspace, int64_t workspace_bytes,
                     void* stream);
/* The same with gw in the nn.Conv2d PARAMETER layout [cout][cin][kh][kw] (what `weight.grad` is), written or -- accumulate != 0 --
 * added to: no transposing copy and no add per parameter after the kernel.  Not for the packed stem (AGP_E_UNSUPPORTED). */
int agp_conv2d_wgrad_param(const agp_conv_desc* d, float* gw, int accumulate, void* workspace, int64_t workspace_bytes,
                           void* stream);

/* ------------------------------------------------------------------ NetVLAD */

/* NetVLAD.forward, reference model/aggregation.py:126-146.  x: dense fp32 [n][d][hw]
 * (NCHW), conv_w [k][d], centroids [k][d] -> out [n][k*d].  k <= 64, d <= 512. */
int agp_netvlad_fwd(const float* x, const float* conv_w, const float* centroids, int n, int d,
                    int hw, int k, int normalize_input, float* out, void* stream);
/* The same forward on the matrix pipe in exact fp32 (v_mfma_f32_32x32x2_f32): soft-assignment logits [k] x [d] x [hw] and
 * V = a x^T as MFMA GEMMs, several workgroups per image, partial V / S through `workspace`
 * (agp_netvlad_workspace_bytes(n, d, hw, k) bytes; 0 = this shape is not taken: d in {128, 256} only), a finishing launch
 * subtracts diag(sum a) c and normalises.  Same result as agp_netvlad_fwd to fp32 rounding.  k <= 64. */
int64_t agp_netvlad_workspace_bytes(int n, int d, int hw, int k);
int agp_netvlad_fwd_mfma(const float* x, const float* conv_w, const float* centroids, int n, int d, int hw, int k,
                         int normalize_input, float* out, void* workspace, int64_t workspace_bytes, void* stream);
/* Its backward: gout [n][k*d] -> dx [n][d][hw], dw [k][d] (conv.weight), dc [k][d] (centroids); workspace 3 n k d floats.
 * d in {64, 128, 256}.  (Reference: autograd through the same lines.) */
int agp_netvlad_bwd(const float* x, const float* conv_w, const float* centroids, const float* gout, int n, int d, int hw,
                    int k, int normalize_input, float* dx, float* dw, float* dc, float* workspace, void* stream);

/* ---------------------------------------------------------------------- kNN */

/* Exact squared-L2 k-nearest-neighbour search; replaces faiss.IndexFlatL2
 * add/search at test.py:27-32 and datasets/datasets_ws_nuscenes.py:1241-1258.
 *
 * agp_knn_prepare_db : xb fp32 [nb][d] -> split planes padded to nb_pad rows
 *                      (nb_pad = agp_knn_pad_rows(nb)) + squared norms [nb_pad].
 *                      prec: AGP_PREC_BF16X3 (bf16 hi/lo planes), AGP_PREC_BF16 (hi only) or
 *                      AGP_PREC_F16 (one fp16 plane in db_hi; fastest coarse pass, the exact pass
 *                      keeps the result identical -- out-of-range magnitudes only cost speed).
 * agp_knn_search     : coarse pass  = split-bf16 MFMA GEMM with a fused min over
 *                                     16-row database groups (never materialises
 *                                     the [nq][nb] distance matrix);
 *                      exact pass   = fp64 re-evaluation of every group whose
 *                                     coarse minimum is within the proven error
 *                                     bound of the k-th best, then a sorted top-k.
 * Outputs follow faiss: dist fp32 [nq][k] ascending squared L2, idx int64 [nq][k],
 * (FLT_MAX, -1) beyond nb; ties ordered by ascending index.  d % 32 == 0, k <= 128.
 * Workspace sizes are queried with agp_knn_workspace_bytes. */
int64_t agp_knn_pad_rows(int64_t nb);
int agp_knn_prepare_db(const float* xb, int64_t nb, int d, int prec, void* db_hi, void* db_lo,
                       float* db_norm, void* stream);
int64_t agp_knn_workspace_bytes(int64_t nq, int64_t nb, int d, int k);
int agp_knn_search(const float* xq, int64_t nq, const float* xb, const void* db_hi,
                   const void* db_lo, const float* db_norm, int64_t nb, int d, int k, int prec,
                   float* dist, int64_t* idx, void* workspace, int64_t workspace_bytes,
                   void* stream);
/* The first stage of agp_knn_search alone -- query preparation + the coarse pass into `workspace` (sized as for the search) --
 * with the search's arguments and no outputs.  The coarse pass is the search's dominant kernel (faiss's sgemm, test.py:27-32);
 * this entry lets a caller time it on its own with events on the launch stream (bench.py's kNN roofline). */
int agp_knn_coarse_pass(const float* xq, int64_t nq, const void* db_hi, const void* db_lo, const float* db_norm,
                        int64_t nb, int d, int prec, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------- mining */

/* Best positive of every query (reference datasets/datasets_ws_nuscenes.py:1241-1248,
 * get_best_positive_index: a faiss.IndexFlatL2 over the query's hard positives, search k=1).
 * Candidate database rows come as CSR lists: query q owns pos_idx[pos_off[q] .. pos_off[q+1]).
 * out_best[q] = the candidate with the smallest exact (fp64) squared L2 distance, the first one in
 * list order on ties, -1 for an empty list; out_dist (optional) = that distance. */
int agp_mine_best_positive(const float* xq, int64_t nq, const float* xb, int64_t nb, int d,
                           const int64_t* pos_off, const int64_t* pos_idx, int64_t* out_best,
                           float* out_dist, void* stream);

/* ------------------------------------------------------------------- losses */

/* nn.TripletMarginLoss(margin, p=2, eps=1e-6, reduction="sum") summed over a table of triplets
 * (reference train.py:51-61 evaluates it over 10 index views of triplets_local_indexes and adds them
 * up; the caller divides by train_batch_size * negs_num_per_query).  feats fp32 [nrows][d],
 * triplets int64 [nt][3] = (query row, positive row, negative row).  loss_sum: 1 float.
 * grad_feats (optional) fp32 [nrows][d] = d loss_sum / d feats.  Fixed-order reductions.
 * workspace: agp_triplet_loss_workspace_floats(nt) floats. */
int64_t agp_triplet_loss_workspace_floats(int nt);
int agp_triplet_loss(const float* feats, int nrows, int d, const int64_t* triplets, int nt, float margin,
                     float* loss_sum, float* grad_feats, float* workspace, void* stream);

/* The SARE criteria (reference model/functional.py:5-27 sare_ind / sare_joint, called from train.py:62-74): the table of
 * triplets is cut into groups of `group` consecutive rows (1 = sare_ind, 10 = sare_joint); query and positive come from
 * the group's first row, the negatives from every row.  loss_sum = sum over groups of
 * -log_softmax([-|q-p|^2, -|q-n_1|^2, ...])[0]; grad_feats (optional) = d loss_sum / d feats.  Same buffers, workspace
 * (agp_triplet_loss_workspace_floats(nt)) and fixed-order reductions as agp_triplet_loss. */
int agp_sare_loss(const float* feats, int nrows, int d, const int64_t* triplets, int nt, int group,
                  float* loss_sum, float* grad_feats, float* workspace, void* stream);

/* One term of compute_other_loss (reference compute_other_loss.py:21-53,72-101): dist = cdist(x, y)
 * on fp32 [n][d] x [m][d]; target_ij = 0 if |e_i - e_j| < pos_thd, 1 if > neg_thd, ignored otherwise
 * (ex [n][2], ey [m][2] east/north coordinates); elementwise loss on the kept pairs, type 0 'bce'
 * (BCEWithLogits), 1 'mse', 2 'l1' (both after a sigmoid).  Outputs: loss_sum and count (1 float each;
 * the term is loss_sum / count), gx [n][d] / gy [m][d] (optional) = d loss_sum / d x, d y.
 * workspace: agp_pairdist_loss_workspace_floats(n, m) floats. */
int64_t agp_pairdist_loss_workspace_floats(int n, int m);
int agp_pairdist_loss(const float* x, const float* y, int n, int m, int d, const float* ex, const float* ey,
                      float pos_thd, float neg_thd, int type, float* loss_sum, float* count, float* gx,
                      float* gy, float* workspace, void* stream);

/* ------------------------------------------------------- sparse-voxel branch */

/* A sparse tensor (reference: ME.SparseTensor, network_mm/mm.py:87) is a feature matrix
 * [n + 1][C] in map storage format -- the extra last row is zero and stands for a missing
 * neighbour -- with rows sorted by (batch, x, y, z): sample b owns rows [seg_off[b], seg_off[b+1]).
 * Coordinate bookkeeping: the inference path builds every level on the device without host
 * synchronisation (agp_sparse_build / agp_sparse_coarsen below, "capacity mode"); the training path
 * keeps exact-size levels built on the host side of the ABI (agplace_amd/sparse/coords.py).
 *
 * Capacity mode: a level has `cap` rows (the number of input points: an upper bound), of which the first
 * n are valid; n lives on the DEVICE (seg_off[nbatch]) and every row-wise entry point below takes it as
 * the optional `n_dev` pointer (NULL = all rows valid).  Padding keys sort last, padding rows of feature
 * matrices are never referenced by a valid row's neighbour table, the zero row sits at index cap.
 *
 * agp_sparse_conv_fwd: MinkowskiConvolution (+ folded MinkowskiBatchNorm, residual, ReLU) as a
 * gather-GEMM on the MFMA implicit-GEMM kernel.  nbr int32 [ntaps][n_out]: row of the input
 * matrix that tap k of output row i reads (n_in = the zero row when the neighbour is absent).
 * Weights [cout][ntaps][cin] in the precision's format.  cin % 32 == 0, cout % 64 == 0.
 * (reference models/minkfpn.py:52-58, layers/eca_block.py:62-79, ME kernel [ntaps][cin][cout].)
 * With `tile_taps` (agp_sparse_tile_taps) the kernel walks, per tile, only the taps that occur in it.  row_perm (optional): GEMM row m computes output row
 * row_perm[m] -- agp_sparse_zplane_perm's order puts the rows of one z-plane of a sample into the same tiles, whose taps towards
 * an absent plane are then skipped; results do not depend on it. */
int agp_sparse_conv_fwd(const void* f_hi, const void* f_lo, int64_t n_in_rows, const int32_t* nbr,
                        int64_t n_out, int cin, int cout, int ntaps, const void* w_hi, const void* w_lo,
                        const float* scale, const float* shift, const void* res_hi, const void* res_lo,
                        int relu, void* out_hi, void* out_lo, int prec, const int64_t* n_dev, const int32_t* row_perm,
                        const uint32_t* tile_taps, void* stream);
/* mask [ngran] uint32 (ngran >= ceil(n_out / 128), ntaps <= 32): bit t of mask[g] = some row among GEMM rows 128 g .. 128 g + 127
 * (row order `perm`, or natural when NULL) has a neighbour through tap t.  agp_sparse_conv_fwd's optional `tile_taps`. */
int agp_sparse_tile_taps(const int32_t* nbr, int64_t n_out, int ntaps, int64_t zero_row, const int32_t* perm,
                         const int64_t* n_dev, uint32_t* mask, int64_t ngran, void* stream);
/* perm [cap] int32: the valid rows of every batch sample grouped by z-plane (stable), identity past seg_off[nbatch]. */
int agp_sparse_zplane_perm(const int64_t* keys, const int64_t* seg_off, int nbatch, int64_t cap, int32_t* perm, void* stream);
/* Kernel map of a sparse convolution (the part of ME's CoordinateManager the path needs): keys are
 * (batch, x, y, z) linearised with 16-bit biased fields (batch << 48 | x+2^15 << 32 | y+2^15 << 16 |
 * z+2^15), so a coordinate offset is a key offset dkey[k].  nbr[k][i] = row of out_keys[i] + dkey[k]
 * in the SORTED in_keys, n_in (= the zero feature row) when that site is unoccupied. */
int agp_sparse_kernel_map(const int64_t* in_keys, int64_t n_in, const int64_t* out_keys, int64_t n_out,
                          const int64_t* dkey, int ntaps, int32_t* nbr, const int64_t* n_dev, void* stream);
/* The same table for the path's regular offset grids, one binary search per (dx, dy) column (z-neighbours are adjacent in
 * the sorted keys): centered = 1: odd kernel, offsets (i - ksize/2) * stride per axis (stride-1 convolutions of a tensor of
 * that stride); centered = 0: offsets i * stride, i in [0, ksize) (ksize = 2: the children of a stride-2 output).
 * nbr [ksize^3][n_out], kidx = ix + ksize * iy + ksize^2 * iz.  n_in_dev: optional device-side count of valid input rows;
 * in_seg_off: optional segment offsets of the INPUT keys (a neighbour lies in the same sample: shorter searches). */
int agp_sparse_kernel_map_grid(const int64_t* in_keys, int64_t n_in, const int64_t* out_keys, int64_t n_out, int ksize,
                               int centered, int stride, int32_t* nbr, const int64_t* n_dev, const int64_t* n_in_dev,
                               const int64_t* in_seg_off, void* stream);
/* First layer (MinkFPN.conv0: kernel 5, one input channel; models/minkfpn.py:48-50): direct gather
 * with fp32 input features f [n_in] and fp32 weights w [ntaps][cout]; nbr entries outside
 * [0, n_in) are skipped. */
int agp_sparse_conv_cin1_fwd(const float* f, int64_t n_in, const int32_t* nbr, int64_t n_out, int ntaps,
                             const float* w, int cout, const float* scale, const float* shift, int relu,
                             void* out_hi, void* out_lo, const int64_t* n_dev, void* stream);
/* The same layer without a materialised kernel map (inference): every output row finds its ksize^3 neighbours in the
 * sorted keys itself.  cout 32 or 64.  prec = AGP_PREC_F16 (out_lo NULL): one binary search per (row, dx) + a scan over
 * the x-plane's (dy, dz) window fills a [128 rows][taps] fp16 tile in LDS, the taps x cout product runs on the matrix pipe;
 * any other precision: fp32 vector arithmetic, one binary search per (dx, dy) column (z-neighbours are adjacent). */
int agp_sparse_conv0_fwd(const int64_t* keys, int64_t cap, const int64_t* n_dev, const float* f, int ksize, int stride,
                         const float* w, int cout, const float* scale, const float* shift, int relu, void* out_hi,
                         void* out_lo, const int64_t* seg_off, int prec, void* stream);
/* Level 0 of a sparse tensor from the network's inputs (reference network_mm/mm.py:87 `ME.SparseTensor(features,
 * coordinates)`): coords [n][4] (batch, x, y, z) as int64 (kind 0), float32 (kind 1) or float64 (kind 2; floored),
 * features [n][cfeat] fp32 (or NULL).  Output, all of capacity n: sorted unique keys (padded), the mean feature row of
 * every unique coordinate, seg_off [nbatch + 1] (seg_off[nbatch] = number of valid rows), bidx [n]; *range_flag |= 1
 * if a point lay outside the 16-bit key fields (|c| > 32511: the point joins the ORIGIN voxel of its sample) or its batch index
 * outside [0, nbatch) (clamped), |= 2 if one sample holds
 * more than 65536 input points (that sample comes out empty).  No host synchronisation, no library primitive: one workgroup
 * sorts one batch sample's keys in LDS (csrc/coords.hip); workspace from agp_sparse_coords_workspace_bytes(n, nbatch, cfeat). */
int64_t agp_sparse_coords_workspace_bytes(int64_t cap, int nbatch, int cfeat);
int agp_sparse_build(const void* coords, int kind, int64_t n, const float* feats, int cfeat, int nbatch, int64_t* keys,
                     float* feats_out, int64_t* seg_off, int32_t* bidx, int32_t* range_flag, void* workspace,
                     int64_t workspace_bytes, void* stream);
/* The next coarser level (kernel 2 / stride 2 convolution, models/minkfpn.py:53): keys_out = sorted unique of
 * floor(c / 2 stride) * 2 stride, same capacity, with its seg_off / bidx.  `keys` / `seg_off_in` = the finer level
 * (two launches: per-sample sort + placement). */
int agp_sparse_coarsen(const int64_t* keys, const int64_t* seg_off_in, int64_t cap, int stride, int nbatch, int64_t* keys_out,
                       int64_t* seg_off, int32_t* bidx, void* workspace, int64_t workspace_bytes, void* stream);
/* Per-sample mean (ME.MinkowskiGlobalPooling / GlobalAvgPooling) and GeM (layers/pooling.py:70-87)
 * of a feature matrix: mean_out / gem_out fp32 [nseg][c] (either may be NULL). */
int agp_seg_pool_fwd(const void* hi, const void* lo, const int64_t* seg_off, int nseg, int c, const float* p,
                     float eps, float* mean_out, float* gem_out, void* stream);
/* ECALayer (layers/eca_block.py:14-43): out[b][c] = sigmoid(Conv1d_k over channels of mean[b][:]). */
int agp_eca_scale_fwd(const float* mean, int nb, int c, const float* w, int k, float* out, void* stream);
/* out[i] = relu?( y[i] * scale[b(i)]? + add[b(i)]? + res[i]? ), b(i) = bidx[i]: ECA broadcast
 * multiplication + residual + ReLU (eca_block.py:70-79) and ME_broadcast_add
 * (network_mm/stage2fuse_blockadd.py:26-32). */
int agp_seg_affine_fwd(const void* y_hi, const void* y_lo, const int32_t* bidx, const float* scale,
                       const float* add, const void* r_hi, const void* r_lo, int64_t n, int c, int relu,
                       void* o_hi, void* o_lo, const int64_t* n_dev, void* stream);

/* ---- training of the sparse branch (split-bf16 feature matrices).  Train-mode MinkowskiBatchNorm
 * reuses agp_bn_stats / agp_map_affine / agp_bn_bwd on the feature matrix seen as a 1 x n map (pad 0);
 * the data gradient of a sparse convolution is agp_sparse_conv_fwd on the transposed kernel map. */
/* gw[tap][cin][cout] (ME's kernel layout) = sum_i x[nbr[tap][i]]^T g[i]; LDS transpose reads, split-K. */
int64_t agp_sparse_conv_wgrad_workspace_bytes(int64_t n_out, int cin, int cout, int ntaps);
int agp_sparse_conv_wgrad(const void* x_hi, const void* x_lo, int64_t n_in_rows, const int32_t* nbr,
                          int64_t n_out, int cin, int cout, int ntaps, const void* g_hi, const void* g_lo,
                          float* gw, void* workspace, int64_t workspace_bytes, void* stream);
/* first layer (Cin = 1): gw[tap][cout] = sum_i f[nbr[tap][i]] * g[i][cout]; the rows are cut into `nslices` slices (one block
 * per tap, 64-channel chunk and slice), `partial` [nslices][ntaps][cout] fp32 holds their sums, added in slice order. */
int agp_sparse_conv_cin1_wgrad(const float* f, int64_t n_in, const int32_t* nbr, int64_t n_out, int ntaps,
                               const void* g_hi, const void* g_lo, int cout, float* gw, float* partial, int nslices,
                               void* stream);
/* out[b][c] = sum over sample b's rows of a[i][c] * b[i][c]  (b == NULL: plain sum) */
int agp_seg_dot_fwd(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo,
                    const int64_t* seg_off, int nseg, int c, float* out, void* stream);
/* ECALayer backward: given mean, scale = sigmoid(conv1d(mean)) and gscale = dL/dscale ([nb][c] each):
 * add[b][c] = dL/dmean[b][c] / n_b (to be broadcast-added to the row gradients), gw[k] = dL/dw. */
int agp_eca_scale_bwd(const float* mean, const float* scale, const float* gscale, const int64_t* seg_off,
                      int nb, int c, const float* w, int k, float* add, float* gw, void* stream);
/* Backward of agp_seg_pool_fwd into the rows: o = base? + gmean[b]/n_b + ggem[b] * dGeM/dx; gp: dL/dp (AGP_GP_FLOATS floats, as
 * agp_gem_f32_bwd's). */
int agp_seg_pool_bwd(const void* x_hi, const void* x_lo, const int32_t* bidx, const int64_t* seg_off,
                     const float* gmean, const float* ggem, const float* gem_y, const float* p, float eps,
                     const void* b_hi, const void* b_lo, int64_t n, int c, void* o_hi, void* o_lo, float* gp,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AGPLACE_HIP_H */
