// Implicit-GEMM convolution, "direct-X" variant (the default conv kernel).
//
// Measured on MI355X, the LDS-staged kernel of igemm.hip is bound by LDS traffic, not by the
// matrix cores: every K-step writes the 256-pixel X tile into LDS by LDS-DMA (~64 B/clk/CU) and
// reads it back exactly once -- with one output-channel tile per workgroup the X rows are not
// shared between waves at all.  This kernel therefore keeps X OUT of LDS:
//
//   * MFMA 16x16x32 bf16, A = W (rows = output channels), B = X (cols = pixels).  The B fragment
//     of a lane (pixel l&15, channels 8*(l>>4)..+7) is 16 contiguous bytes of the NHWC plane, so a
//     wave-wide buffer_load_dwordx4 fetches 16 pixels x 64 B straight into MFMA operand registers
//     (per-lane pixel base in voffset, the (ky,kx,c) tap in the scalar soffset; the zero halo
//     supplies the padding).  X fragments are double-buffered in registers one K-step ahead.
//   * Only W (shared by the 4 waves) goes through LDS: 8-16 KB per K-step by LDS-DMA into a
//     2-stage ring, XOR-swizzled on the source side for conflict-free ds_read_b128.
//   * Each wave owns 64 pixels x all BN (64 or 128) channels of the workgroup: X is fetched once
//     per tap per workgroup; accumulators 64/128 VGPRs.
//   * Split-bf16 (NPREC 3): hi*hi + hi*lo + lo*hi into one fp32 accumulator.
//   * Epilogue as in igemm.hip: LDS transpose -> scale/shift, residual, ReLU, hi/lo split,
//     whole-line stores.
#include "igemm_params.hpp"

namespace agp_igemm {

// chunk swizzle of a 64-byte LDS row for the 16x16x32 A-fragment read pattern
__device__ __forceinline__ int swz16(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int NTW, int NPREC>
constexpr int d16_lds_bytes() {
    constexpr int ring = 2 * NTW * 16 * 64 * PrecT<NPREC>::WPL;
    constexpr int epi = 4 * 32 * (NTW * 16 * 4 + 16) + 4 * NTW * 16 * 2 * 4;      // + the waves' channel sums (stat_partial)
    return ring > epi ? ring : epi;
}

// POOL = 1 (the stem, fp16 maps): the workgroup's 256 pixels are a 16x16 block of the conv map starting at
// (14*ty - 1, 14*tx - 1); BatchNorm + ReLU are applied in registers, the block is laid down in LDS as fp16 and
// the 7x7 outputs of MaxPool2d(3, 2, 1) it fully contains are written.  The full-resolution stem map (4x the
// pooled one, otherwise written once and read once) never exists; 27 % of the conv is recomputed at the
// block seams.
template <int NTW, int NPREC, int POOL = 0>
__global__ void __launch_bounds__(256, (NTW == 4 ? 3 : 2)) igemm_d16_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BN = NTW * 16;
    constexpr int NPL = PrecT<NPREC>::XPL, WPL = PrecT<NPREC>::WPL;   // NPL: X planes held in registers
    constexpr int W_PLANE = BN * 64;            // one K-step (32 ch) of BN rows
    constexpr int WSTAGE = W_PLANE * WPL;
    constexpr int WI = BN / 64;                 // W LDS-DMA instructions per wave per plane
    constexpr int EROWB = BN * 4 + 16;          // epilogue row: BN fp32 + pad
    constexpr int LPP = BN / 8;                 // lanes per pixel in the epilogue read-back

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;

    const int bid = blockIdx.x;
    int m0 = 0, n0 = 0;
    int pimg = 0, pty = 0, ptx = 0;             // POOL: image and block coordinates
    unsigned pvalid = 0;                        // POOL: bit mt = this lane's pixel of m-tile mt lies inside the conv map
    int xoff[4];
    if constexpr (POOL) {
        const int per_img = p.pool_ty * p.pool_tx;
        pimg = bid / per_img;
        const int r = bid - pimg * per_img;
        pty = r / p.pool_tx; ptx = r - pty * p.pool_tx;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int cy = 14 * pty - 1 + 4 * wave + mt, cx = 14 * ptx - 1 + l15;
            const bool ok = cy >= 0 && cy < p.pool_h1 && cx >= 0 && cx < p.pool_w1;
            pvalid |= (ok ? 1u : 0u) << mt;
            const int yy = min(max(cy, 0), p.pool_h1 - 1), xx = min(max(cx, 0), p.pool_w1 - 1);
            const int el = pimg * p.x_sn + yy * p.sy * p.x_sh + xx * p.sx * p.x_sw + p.x_base;
            xoff[mt] = el * 2 + lq * 16;
        }
    } else {
        const int xcd = bid & 7, j = bid >> 3;
        const int nt0 = j % p.NT;
        const int mt0 = xcd * p.mt_chunk + j / p.NT;
        if (mt0 >= p.MT) return;
        m0 = mt0 * BM + wave * 64; n0 = nt0 * BN;
        // ---- per-lane X offsets (bytes): 4 m-tiles of 16 pixels
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int m = m0 + mt * 16 + l15;
            m = m < p.M ? m : p.M - 1;
            const uint32_t img = fdiv((uint32_t)m, p.d_howo);
            const uint32_t rem = (uint32_t)m - img * p.d_howo.d;
            const uint32_t oy = fdiv(rem, p.d_wo);
            const uint32_t ox = rem - oy * p.d_wo.d;
            const int el = (int)img * p.x_sn + (int)oy * p.sy * p.x_sh + (int)ox * p.sx * p.x_sw + p.x_base;
            xoff[mt] = el * 2 + lq * 16;
        }
    }
    // ---- per-lane W offsets for the LDS-DMA (16 rows of 64 B per instruction)
    int woff[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = (wave + 4 * i) * 16 + (lane >> 2);
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        woff[i] = n * p.Ktot * 2 + (((lane & 3) ^ swz16(row)) << 4);
    }
    const int wfo = l15 * 64 + ((lq ^ swz16(l15)) << 4);   // A-fragment read offset inside a 16-row tile

    const __amdgpu_buffer_rsrc_t rx_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(NPL == 2 ? p.x_lo : p.x_hi), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(WPL == 2 ? p.w_lo : p.w_hi), 0, p.w_bytes, 0x00020000);

    const int cchunks = p.CK / 32;
    const int nk = p.ntaps * cchunks;
    int kx = 0, ky = 0, cc = 0;

    auto load_w = [&](int buf, int kt) {
        const int ws = __builtin_amdgcn_readfirstlane(kt * 64);
        char* base = smem + buf * WSTAGE;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int ldsoff = (wave + 4 * i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_hi, LDS_PTR(base + ldsoff), 16, woff[i], ws, 0, 0);
            if (WPL == 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_lo, LDS_PTR(base + W_PLANE + ldsoff), 16, woff[i], ws, 0, 0);
        }
    };
    auto load_x = [&](u32x4 (&dst)[4][NPL]) {
        const int xs = __builtin_amdgcn_readfirstlane((ky * p.x_sh + kx * p.x_sw + cc * 32) * 2);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            dst[mt][0] = __builtin_amdgcn_raw_buffer_load_b128(rx_hi, xoff[mt], xs, 0);
            if (NPL == 2) dst[mt][NPL - 1] = __builtin_amdgcn_raw_buffer_load_b128(rx_lo, xoff[mt], xs, 0);
        }
        if (++cc == cchunks) {
            cc = 0;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };

    f32x4 acc[NTW][4];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf, const u32x4 (&x)[4][NPL]) {
        const char* wb = smem + buf * WSTAGE + wfo;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const bf16x8 wh = *(const bf16x8*)(wb + nt * 1024);
            bf16x8 wl = wh;
            if (WPL == 2) wl = *(const bf16x8*)(wb + W_PLANE + nt * 1024);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, x[mt][0]);
                const bf16x8 xl = __builtin_bit_cast(bf16x8, x[mt][NPL - 1]);
                mfma16<NPREC>(acc[nt][mt], wh, wl, xh, xl);
            }
        }
    };

    // ---- main loop, unrolled by two so that the register double-buffer is statically indexed.
    // Prefetches past the end are clamped (W) or read zeros from beyond the plane (X): harmless.
    u32x4 xa[4][NPL], xb[4][NPL];
    load_w(0, 0);
    load_x(xa);
    for (int kt = 0; kt + 1 < nk; kt += 2) {
        __syncthreads();   // W(kt) in LDS and X(kt) in registers (vmcnt(0)); ring slot 1 is free
        load_w(1, kt + 1);
        load_x(xb);
        compute(0, xa);
        __syncthreads();
        load_w(0, kt + 2 < nk ? kt + 2 : nk - 1);
        load_x(xa);
        compute(1, xb);
    }
    if (nk & 1) {          // odd tail: K-step nk-1 sits in ring slot 0 / xa
        __syncthreads();
        compute(0, xa);
    }

    if constexpr (POOL) {
        // ---- fused epilogue: BN + ReLU in registers -> fp16 block [256 px][BN ch] in LDS -> 7x7 max-pool outputs
        static_assert(!POOL || (NTW == 4 && PrecT<NPREC>::XPL == 1), "stem + pool: 64 channels, fp16 maps");
        __syncthreads();                        // the W ring is dead: the block aliases it
        constexpr int PP = 72;                  // block pitch in elements: 64 ch + 8 (144 B: spreads the banks, keeps 16-B alignment)
        bf16_t* blk = (bf16_t*)smem;            // [pixel 0..255][PP]
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int c0 = nt * 16 + 4 * lq;
            float sc4[4], sh4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc4[e] = p.scale ? p.scale[c0 + e] : 1.f;
                sh4[e] = p.shift ? p.shift[c0 + e] : 0.f;
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bool ok = (pvalid >> mt) & 1u;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? fmaxf(acc[nt][mt][e] * sc4[e] + sh4[e], 0.f) : 0.f;
                const int px = (4 * wave + mt) * 16 + l15;
                u32x2 pk = {pack2(f2h(v[0]), f2h(v[1])), pack2(f2h(v[2]), f2h(v[3]))};
                *(u32x2*)(blk + px * PP + c0) = pk;
            }
        }
        __syncthreads();
        bf16_t* ohi = (bf16_t*)p.o_hi;
        for (int it = tid; it < 49 * 8; it += 256) {
            const int g = it & 7, pp = it >> 3;
            const int py = pp / 7, pxx = pp - py * 7;
            const int oy = 7 * pty + py, ox = 7 * ptx + pxx;
            if (oy >= p.pool_h2 || ox >= p.pool_w2) continue;
            float best[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) best[e] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float v[8];
                    unpack8_h(*(const u32x4*)(blk + ((2 * py + ky) * 16 + 2 * pxx + kx) * PP + g * 8), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], v[e]);
                }
            const size_t off = (size_t)pimg * p.o_sn + (size_t)oy * p.o_sh + (size_t)ox * p.o_sw + p.o_base + g * 8;
            *(u32x4*)(ohi + off) = pack8_h(best);
        }
        return;
    }

    // ---- epilogue: two passes of 32 pixels per wave through LDS
    __syncthreads();
    char* er = smem + wave * (32 * EROWB);
    const int ch = lane % LPP;
    const int nglob = n0 + ch * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = p.scale ? p.scale[nglob + e] : 1.f;
        sh[e] = p.shift ? p.shift[nglob + e] : 0.f;
    }
    bf16_t* ohi = (bf16_t*)p.o_hi;
    bf16_t* olo = (bf16_t*)p.o_lo;
    const bf16_t* rhi = (const bf16_t*)p.r_hi;
    const bf16_t* rlo = (const bf16_t*)p.r_lo;
    // optional per-tile channel statistics of the stored values (train-mode BatchNorm; as igemm_kxr's: agp_conv_desc::stat_partial)
    const bool stats = p.stat_partial != nullptr;
    float st1[8], st2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                *(f32x4*)(er + (m2 * 16 + l15) * EROWB + (nt * 16 + 4 * lq) * 4) = acc[nt][pass * 2 + m2];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 32 / (64 / LPP); ++it) {
            const int ml = it * (64 / LPP) + lane / LPP;
            const int m = m0 + pass * 32 + ml;
            if (m >= p.M) continue;
            const f32x4 a = *(const f32x4*)(er + ml * EROWB + ch * 32);
            const f32x4 b = *(const f32x4*)(er + ml * EROWB + ch * 32 + 16);
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            const uint32_t img = fdiv((uint32_t)m, p.d_howo);
            const uint32_t rem = (uint32_t)m - img * p.d_howo.d;
            const uint32_t oy = fdiv(rem, p.d_wo);
            const uint32_t ox = rem - oy * p.d_wo.d;
            const size_t off = (size_t)img * p.o_sn + (size_t)oy * p.o_sh + (size_t)ox * p.o_sw + p.o_base + nglob;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (rhi) {
                float r[8];
                map_load8(rhi, rlo, off, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            map_store8(ohi, olo, off, v);
            if (stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { st1[e] += v[e]; st2[e] += v[e] * v[e]; }
            }
        }
    }
    if (stats) {
        // lanes sharing a channel group (lane % LPP) -> wave totals; the four waves (row blocks of the tile) -> tile totals, in a
        // fixed order: partial[m tile][2][N]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = LPP; o < 64; o <<= 1) {
                st1[e] += __shfl_xor(st1[e], o, 64);
                st2[e] += __shfl_xor(st2[e], o, 64);
            }
        }
        float* red = (float*)(smem + 4 * 32 * EROWB);           // [wave][BN channels][2], beyond every wave's staging rows
        if (lane < LPP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * BN + lane * 8 + e) * 2] = st1[e];
                red[(wave * BN + lane * 8 + e) * 2 + 1] = st2[e];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a += red[(w * BN + tid) * 2];
                b += red[(w * BN + tid) * 2 + 1];
            }
            const int mt = (m0 - wave * 64) / BM;
            p.stat_partial[(size_t)mt * 2 * p.N + n0 + tid] = a;
            p.stat_partial[(size_t)mt * 2 * p.N + p.N + n0 + tid] = b;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- the stem (packed 7x7/2 conv + BN + ReLU + MaxPool2d(3, 2, 1)), input patch staged in LDS.
// The direct-X kernel above fetches a pixel's 64-B tap row straight into MFMA operand registers; for the stem the tap rows
// of neighbouring output pixels overlap by 6 of 8 pixels (stride 2, 7 taps), so a 16x16 output block re-fetches its
// 37 x 38-pixel input patch (11 KB) ten times over through the texture path (28 wave-wide loads per wave, each waited
// for inside the K loop: 7.7 us per workgroup round for 0.75 us of MFMA).  Here the patch and ALL of W (7 tap rows x 64
// channels x 64 B = 28 KB) go to LDS once per workgroup by LDS-DMA, and the K loop is 7 unrolled steps of ds_read_b128
// + MFMA with no global access and no barrier.  Layout: W [ky][n-tile][16 rows x 64 B, chunk-swizzled]; patch
// [37 rows][304 B] (38 pixels x 4 channels fp16; the B fragment of lane (pixel x, k-group q) is the 16 B at
// 16 * (x + q) of row 2 * y + ky: 16 consecutive lanes read 256 contiguous bytes).  Out-of-plane patch rows / columns
// (block seams at the image border) read zeros or the neighbouring image through the buffer descriptor and only feed
// outputs that are masked out.  Epilogue as above (BN + ReLU -> fp16 block in LDS -> 7x7 pooled outputs).
constexpr int STEM_PROWB = 304;                       // patch row bytes
constexpr int STEM_PROWS = 37;
constexpr int STEM_W_BYTES = 7 * 4 * 1024;            // 28 KB
[[maybe_unused]] constexpr int STEM_PATCH_BYTES = STEM_PROWS * STEM_PROWB;
constexpr int STEM_LDS = (STEM_W_BYTES + 12 * 1024 > 256 * 72 * 2) ? STEM_W_BYTES + 12 * 1024 : 256 * 72 * 2;

// IN = 0: the packed NHWC4 halo-3 map (LDS-DMA).  IN = 1 / 2: the network's INPUT itself -- an fp32 [n][3][h][w] image with
// arbitrary element strides, or uint8 camera tiles [n][ncam][h][wcam][3] (ToTensor + Normalize on the fly, the arithmetic of
// pack_u8_cams_kernel) -- converted to fp16 on its way into the patch: no packed copy of the image is written or read
// (pack_nchw4_kernel: 98 us and 159 MB written + re-read per 64 panoramas).
// (struct StemRaw: igemm_params.hpp)

template <int IN>
__global__ void __launch_bounds__(256, 4) stem_pool_lds_kernel(IgemmParams p, StemRaw raw) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int bid = blockIdx.x;
    const int per_img = p.pool_ty * p.pool_tx;
    const int pimg = bid / per_img;
    const int r0 = bid - pimg * per_img;
    const int pty = r0 / p.pool_tx, ptx = r0 - pty * p.pool_tx;
    const int oy0 = 14 * pty - 1, ox0 = 14 * ptx - 1;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, p.w_bytes, 0x00020000);
    char* wl = smem;
    char* patch = smem + STEM_W_BYTES;
    // ---- W: tap row ky of the 64 channels = 4 KB = one LDS-DMA instruction per wave (16 rows x 64 B each)
    {
        const int row = wave * 16 + (lane >> 2);
        const int woff = row * p.Ktot * 2 + (((lane & 3) ^ swz16(row)) << 4);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(wl + ky * 4096 + wave * 1024), 16, woff, ky * 64, 0, 0);
    }
    if constexpr (IN == 0) {
    // ---- patch: 37 rows x 19 chunks of 16 B; chunk c = 64 * i + lane of instruction i (3 per wave)
        const int base = (pimg * p.x_sn + 2 * oy0 * p.x_sh + 2 * ox0 * 4) * 2;     // may be "negative": wraps past num_records -> zeros
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ci = wave * 3 + i;                   // 12 instructions cover 768 >= 703 chunks
            const int c = ci * 64 + lane;
            const int pr = c / 19, pj = c - pr * 19;
            const int off = (pr < STEM_PROWS) ? base + pr * p.x_sh * 2 + pj * 16 : -16;     // past the patch: a zero read
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(patch + ci * 1024), 16, off, 0, 0, 0);
        }
    } else {
        // ---- patch from the raw input: pixel (pr, px) of the patch is image pixel (2 oy0 + pr - 3, 2 ox0 + px - 3);
        // outside the image it is the zero padding.  All loads of a thread first (6 pixels x 3 channels), then the
        // conversions and the 8-byte LDS writes.
        constexpr int NPX = STEM_PROWS * 38, IT = (NPX + 255) / 256;
        float v[IT][3];
        int lo_[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = it * 256 + tid;
            const int pr = idx / 38, px = idx - pr * 38;
            const int iy = 2 * oy0 + pr - 3, ix = 2 * ox0 + px - 3;
            const bool inb = idx < NPX && iy >= 0 && iy < raw.h && ix >= 0 && ix < raw.w;
            lo_[it] = idx < NPX ? pr * STEM_PROWB + px * 8 : -1;
            if constexpr (IN == 1) {
                const float* src = (const float*)raw.x + (long long)pimg * raw.sn + (long long)iy * raw.sh + (long long)ix * raw.sw;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[it][c] = inb ? src[c * raw.sc] : 0.f;
            } else {
                const int cam = ix / raw.wcam, xx = ix - cam * raw.wcam;
                const uint8_t* src = (const uint8_t*)raw.x + ((((long long)pimg * raw.ncam + cam) * raw.h + iy) * raw.wcam + xx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[it][c] = inb ? ((float)src[c] / 255.f - raw.m[c]) / raw.s[c] : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (lo_[it] >= 0) {
                u32x2 pk = {pack2(f2h(v[it][0]), f2h(v[it][1])), pack2(f2h(v[it][2]), (bf16_t)0)};
                *(u32x2*)(patch + lo_[it]) = pk;
            }
        }
    }
    unsigned pvalid = 0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int cy = oy0 + 4 * wave + mt, cx = ox0 + l15;
        const bool ok = cy >= 0 && cy < p.pool_h1 && cx >= 0 && cx < p.pool_w1;
        pvalid |= (ok ? 1u : 0u) << mt;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wfo = l15 * 64 + ((lq ^ swz16(l15)) << 4);
    const int pfo = (8 * wave) * STEM_PROWB + 16 * (l15 + lq);       // block row 4 * wave (+ mt), patch row 2 * that (+ ky)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        bf16x8 wf[4], xf[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8*)(wl + ky * 4096 + nt * 1024 + wfo);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8*)(patch + pfo + (2 * mt + ky) * STEM_PROWB);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) mfma16<4>(acc[nt][mt], wf[nt], wf[nt], xf[mt], xf[mt]);
    }
    // ---- BN + ReLU in registers -> fp16 block [256 px][64 ch] in LDS -> 7x7 max-pool outputs
    __syncthreads();                        // W and the patch are dead: the block aliases them
    constexpr int PP = 72;
    bf16_t* blk = (bf16_t*)smem;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c0 = nt * 16 + 4 * lq;
        float sc4[4], sh4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc4[e] = p.scale ? p.scale[c0 + e] : 1.f;
            sh4[e] = p.shift ? p.shift[c0 + e] : 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const bool ok = (pvalid >> mt) & 1u;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][e] * sc4[e] + sh4[e];
            const int px = (4 * wave + mt) * 16 + l15;
            // ReLU + fp16 saturation in one med3, packed conversion, the block-seam mask on the packed words
            u32x2 pk = {pack2_h_lo(v[0], v[1], 0.f), pack2_h_lo(v[2], v[3], 0.f)};
            if (!ok) pk = u32x2{0u, 0u};
            *(u32x2*)(blk + px * PP + c0) = pk;
        }
    }
    __syncthreads();
    bf16_t* ohi = (bf16_t*)p.o_hi;
    for (int it = tid; it < 49 * 8; it += 256) {
        const int g = it & 7, pp = it >> 3;
        const int py = pp / 7, pxx = pp - py * 7;
        const int oy = 7 * pty + py, ox = 7 * ptx + pxx;
        if (oy >= p.pool_h2 || ox >= p.pool_w2) continue;
        u32x4 best = {0u, 0u, 0u, 0u};              // post-ReLU values: the maximum stays in packed fp16 (exact)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                best = pkmax8_h(best, *(const u32x4*)(blk + ((2 * py + ky) * 16 + 2 * pxx + kx) * PP + g * 8));
        const size_t off = (size_t)pimg * p.o_sn + (size_t)oy * p.o_sh + (size_t)ox * p.o_sw + p.o_base + g * 8;
        *(u32x4*)(ohi + off) = best;
    }
#endif
}

template <int IN>
int launch_stem_pool_lds(IgemmParams& p, const StemRaw& raw, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)stem_pool_lds_kernel<IN>, STEM_LDS, attr_done)) return AGP_E_LAUNCH;
    const int n = p.M / (p.pool_h1 * p.pool_w1);
    AGP_LAUNCH(stem_pool_lds_kernel<IN>, dim3(n * p.pool_ty * p.pool_tx), dim3(256), STEM_LDS, s, p, raw);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

template <int NTW, int NPREC>
int launch_d16(IgemmParams& p, hipStream_t s) {
    constexpr int lds = d16_lds_bytes<NTW, NPREC>();
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)igemm_d16_kernel<NTW, NPREC>, lds, attr_done)) return AGP_E_LAUNCH;
    p.MT = (p.M + 255) / 256;
    p.NT = (p.N + NTW * 16 - 1) / (NTW * 16);
    p.mt_chunk = (p.MT + 7) / 8;
    AGP_LAUNCH((igemm_d16_kernel<NTW, NPREC>), dim3(p.mt_chunk * 8 * p.NT), dim3(256), lds, s, p);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

// fused stem + max-pool launch (fp16 maps, 64 output channels)
template <int NPREC>
int launch_d16_pool(IgemmParams& p, hipStream_t s) {
    constexpr int lds = d16_lds_bytes<4, NPREC>() > 256 * 72 * 2 ? d16_lds_bytes<4, NPREC>() : 256 * 72 * 2;
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)igemm_d16_kernel<4, NPREC, 1>, lds, attr_done)) return AGP_E_LAUNCH;
    const int n = p.M / (p.pool_h1 * p.pool_w1);
    AGP_LAUNCH((igemm_d16_kernel<4, NPREC, 1>), dim3(n * p.pool_ty * p.pool_tx), dim3(256), lds, s, p);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

}  // namespace agp_igemm

int agp_internal_stem_walk(agp_igemm::IgemmParams& p, int kind, agp_igemm::StemRaw raw, hipStream_t s);     // stem_walk.hip
bool agp_internal_stem_walk_reads(const agp_igemm::StemRaw& raw, int n);
bool agp_internal_stem_walk_reads_u8(const agp_igemm::StemRaw& raw, int n);

// development build, STEM_WALK = 0: the one-workgroup-per-block stem kernels of this file instead of stem_walk.hip
static bool stem_walk_enabled() { return AGP_TUNE("STEM_WALK", 1) != 0; }

int agp_internal_conv_d16_pool(agp_igemm::IgemmParams& p, int prec, hipStream_t s) {
    using namespace agp_igemm;
    if (prec == AGP_PREC_F16W2) return launch_d16_pool<2>(p, s);
    if (prec == AGP_PREC_F16) {
        if (stem_walk_enabled()) return agp_internal_stem_walk(p, 0, StemRaw{}, s);
#if defined(AGP_TUNING)
        // the patch addressing uses 32-bit byte offsets relative to the plane; STEM_LDS = 0: the direct-X kernel
        return AGP_TUNE("STEM_LDS", 1) ? launch_stem_pool_lds<0>(p, StemRaw{}, s) : launch_d16_pool<4>(p, s);
#else
        return AGP_E_UNSUPPORTED;           // (unreachable: the walking kernel takes every fp16 stem)
#endif
    }
    return AGP_E_BADARG;
}

// Conv dispatch for the direct-X kernel (called from igemm.hip's agp_conv2d_fwd).
int agp_internal_conv_d16(agp_igemm::IgemmParams& p, int prec, hipStream_t s) {
    using namespace agp_igemm;
    const bool wide = (p.N % 128 == 0);
    if (prec == AGP_PREC_BF16X3) return wide ? launch_d16<8, 3>(p, s) : launch_d16<4, 3>(p, s);
    if (prec == AGP_PREC_F16W2) return wide ? launch_d16<8, 2>(p, s) : launch_d16<4, 2>(p, s);
    if (prec == AGP_PREC_F16) return wide ? launch_d16<8, 4>(p, s) : launch_d16<4, 4>(p, s);
    return AGP_E_BADARG;
}

// The stem reading the network's input directly (kind 1: fp32 image with strides, 2: uint8 camera tiles).
int agp_internal_stem_raw(agp_igemm::IgemmParams& p, int kind, const void* x, long long sn, long long sc, long long sh, long long sw,
                          int h, int w, int ncam, const float* mean3, const float* std3, hipStream_t s) {
    using namespace agp_igemm;
    StemRaw raw = {};
    raw.x = x; raw.sn = sn; raw.sc = sc; raw.sh = sh; raw.sw = sw; raw.h = h; raw.w = w;
    raw.ncam = ncam > 0 ? ncam : 1; raw.wcam = w / raw.ncam;
    for (int c = 0; c < 3; ++c) { raw.m[c] = mean3 ? mean3[c] : 0.f; raw.s[c] = std3 ? std3[c] : 1.f; }
    if (kind == 1 && stem_walk_enabled() && agp_internal_stem_walk_reads(raw, p.M / (p.pool_h1 * p.pool_w1)))
        return agp_internal_stem_walk(p, 1, raw, s);
    if (kind == 2 && stem_walk_enabled() && agp_internal_stem_walk_reads_u8(raw, p.M / (p.pool_h1 * p.pool_w1)))
        return agp_internal_stem_walk(p, 2, raw, s);
    if (kind == 1) return launch_stem_pool_lds<1>(p, raw, s);
    if (kind == 2) return launch_stem_pool_lds<2>(p, raw, s);
    return AGP_E_BADARG;
}
