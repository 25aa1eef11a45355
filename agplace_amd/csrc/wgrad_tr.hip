// Convolution weight gradient straight from the NHWC maps, no transposed copies.
//
//   dW[ky][kx][ci][co] = sum over output pixels  x[img][s*oy + ky - pad][s*ox + kx - pad][ci] * g[img][oy][ox][co]
//
// is a GEMM whose reduction index is the PIXEL.  In NHWC memory the 8 consecutive k-values an MFMA
// operand lane needs are strided by the channel count, so the first version of this path copied both
// maps into channel-major planes (agp_map_transpose_cp / agp_im2col_t: 60 % of a training step).
// gfx950's LDS transpose read (ds_read_b64_tr_b16) removes that: pixel strips are staged in their
// natural [pixel][32 channels] order (64-byte rows, LDS-DMA, conflict-free without a swizzle) and
// BOTH operands are read column-wise:
//   A = x^T  (rows = 32 input channels of one tap)      B = g  (columns = 32 output channels)
//   D[ci][co] += A[ci][pix] * B[pix][co]                 split-bf16: Al*Bh + Ah*Bl + Ah*Bh
// The reduction runs over the linear raster of the halo-padded gradient plane: its halo is zero, so
// halo positions contribute nothing and no index arithmetic is needed along K.
//   MODE 0 (3x3, stride 1, pad 1): the input pixel of tap (ky,kx) at raster position p is
//          p + (ky-1)*Wp + (kx-1): one staged strip per ky serves its three kx taps by a row shift.
//   MODE 1 (stride 2, 1x1, the packed 7x7 stem): one strip per A-block, gathered per lane.
//   MODE 2 (sparse convolution): as MODE 1, the input row of (tap, output row) comes from the kernel map.
// A workgroup owns NB A-blocks (taps of one 32-channel block, or 32-channel blocks of a 1x1) x 64
// output channels; split-K over raster chunks (blockIdx.z), fp32 partials reduced by a second kernel.

#include "common.hpp"

namespace agp_wgrad {

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct WgradParams {
    const void* x_hi; const void* x_lo; uint32_t x_bytes;
    const void* g_hi; const void* g_lo; uint32_t g_bytes;
    int C;                 // elements between consecutive input pixels (cin; 4 for the packed stem)
    int CK;                // channels per tap (cin; 32 for the stem)
    int N;                 // cout
    int T, KW;             // taps, kernel width (T = KH*KW)
    int Hpx, Wpx;          // padded input plane (pixels)
    int Ho, Wo;            // gradient map interior
    FastDiv d_hopwop, d_wop;
    int stride, d0;        // input padded coordinate = stride*(o) + k + d0,  d0 = pin - pad
    int64_t Kpix;          // raster length = n*(Ho+2)*(Wo+2)
    int k_chunk;           // raster positions per split (multiple of the strip length)
    int nblk_total;        // T * (CK/32)
    int rows_total;        // T * CK
    float* out;            // [split][rows_total][N]
    const int* tab;        // MODE 2: kernel map [T][tab_stride] -> input row
    int tab_stride;
    const uint32_t* gmax;  // wgrad_gather_f16_kernel: [N] fp32 bit patterns of max |g| per output channel (x_hi = the fp16 plane)
};

__device__ __forceinline__ bf16x8 tr_frag(const char* base, int byte_off) {
    // two transposed 4-row reads = the 8 consecutive k-values of this lane's row / column
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + byte_off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + byte_off + 4 * 64));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int MODE, int NB>
__global__ void __launch_bounds__(256, 3) wgrad_tr_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int P = MODE == 0 ? 64 : 32;            // raster positions per K-step
    constexpr int XR = MODE == 0 ? P + 16 : P;        // rows of one X strip
    constexpr int NXS = MODE == 0 ? 3 : NB;           // X strips
    constexpr int XS_BYTES = XR * 64, GS_BYTES = P * 64;
    constexpr int X_BYTES = NXS * 2 * XS_BYTES;       // [strip][plane][row][64 B]
    constexpr int MAXT = (NB + 1) / 2;
    constexpr int GCH = P / 16, XCH = XR / 16;        // 16-row LDS-DMA instructions per strip plane
    constexpr int NINS_G = 2 * 2 * GCH, NINS_X = NXS * 2 * XCH;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs = smem;
    char* const gs = smem + X_BYTES;                  // [co half][plane][row][64 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wt = wave >> 1;
    const int b_lo = wt ? (NB + 1) / 2 : 0;
    const int cnt = wt ? NB / 2 : (NB + 1) / 2;
    const int B0 = blockIdx.x * NB;                   // first A-block of this workgroup
    const int co0 = blockIdx.y * 64;
    const int64_t k_begin = (int64_t)blockIdx.z * p.k_chunk;
    const int nk = p.k_chunk / P;
    const int cblocks = p.CK / 32;

    const __amdgpu_buffer_rsrc_t rx_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx_lo = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_lo, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.g_hi, 0, p.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg_lo = __builtin_amdgcn_make_buffer_rsrc((void*)p.g_lo, 0, p.g_bytes, 0x00020000);

    const int lrow = lane >> 2, lchunk = (lane & 3) << 4;
    // transposed-read lane address inside a strip plane: row 8*(l>>5) + ((l&15)>>2), columns 16*((l>>4)&1) + 4*(l&3)
    const int tr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;

    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
        const int64_t p0 = k_begin + (int64_t)kt * P;
        if (p0 >= p.Kpix) break;                      // wave-uniform
        if (kt) __syncthreads();                      // previous step's fragment reads are done

        // MODE 1: raster position -> input pixel of tap (0,0), per 16-row chunk handled by this lane
        int xbase[MODE == 1 ? GCH : 1];
        bool xval[MODE == 1 ? GCH : 1];
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < GCH; ++c) {
                const int64_t pg = p0 + c * 16 + lrow;
                const uint32_t q = (uint32_t)(pg < p.Kpix ? pg : 0);
                const uint32_t img = fdiv(q, p.d_hopwop);
                const uint32_t rem = q - img * p.d_hopwop.d;
                const uint32_t yy = fdiv(rem, p.d_wop);
                const uint32_t xx = rem - yy * p.d_wop.d;
                xval[c] = pg < p.Kpix && yy >= 1 && yy <= (uint32_t)p.Ho && xx >= 1 && xx <= (uint32_t)p.Wo;
                xbase[c] = ((int)img * p.Hpx + p.stride * ((int)yy - 1) + p.d0) * p.Wpx + p.stride * ((int)xx - 1) + p.d0;
            }
        }
        for (int i = wave; i < NINS_G + NINS_X; i += 4) {
            if (i < NINS_G) {
                const int c = i % GCH, pl = (i / GCH) & 1, hf = i / (2 * GCH);
                const int64_t pix = p0 + c * 16 + lrow;
                const int64_t off64 = (pix * p.N + co0 + hf * 32) * 2 + lchunk;
                const int off = off64 < (int64_t)p.g_bytes ? (int)off64 : 0x7ffffff0;
                const int dst = __builtin_amdgcn_readfirstlane((hf * 2 + pl) * GS_BYTES + c * 1024);
                if (pl) __builtin_amdgcn_raw_ptr_buffer_load_lds(rg_lo, LDS_PTR(gs + dst), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rg_hi, LDS_PTR(gs + dst), 16, off, 0, 0, 0);
            } else {
                const int j = i - NINS_G;
                const int c = j % XCH, pl = (j / XCH) & 1, s = j / (2 * XCH);
                int off;
                if constexpr (MODE == 0) {
                    const int64_t pix = p0 + (int64_t)(s - 1) * p.Wpx - 1 + c * 16 + lrow;
                    const int64_t off64 = (pix * p.C + blockIdx.x * 32) * 2 + lchunk;
                    off = (off64 >= 0 && off64 < (int64_t)p.x_bytes) ? (int)off64 : 0x7ffffff0;
                } else if constexpr (MODE == 2) {
                    const int B = B0 + s;
                    const int cib = B / p.T, tap = B - cib * p.T;
                    const int64_t pix = p0 + c * 16 + lrow;
                    const bool ok = pix < p.Kpix && B < p.nblk_total;
                    const int row = ok ? p.tab[(size_t)tap * p.tab_stride + pix] : 0;
                    off = ok ? (row * p.C + cib * 32) * 2 + lchunk : 0x7ffffff0;
                } else {
                    const int B = B0 + s;
                    const int cib = B / p.T, tap = B - cib * p.T;
                    const int ky = tap / p.KW, kx = tap - ky * p.KW;
                    static_assert(MODE == 0 || GCH == 2, "two 16-row chunks per strip in gather mode");
                    const bool ok = (c == 0 ? xval[0] : xval[GCH - 1]) && B < p.nblk_total;
                    const int xpix = (c == 0 ? xbase[0] : xbase[GCH - 1]) + ky * p.Wpx + kx;
                    off = ok ? (xpix * p.C + cib * 32) * 2 + lchunk : 0x7ffffff0;
                }
                const int dst = __builtin_amdgcn_readfirstlane((s * 2 + pl) * XS_BYTES + c * 1024);
                if (pl) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_lo, LDS_PTR(xs + dst), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_hi, LDS_PTR(xs + dst), 16, off, 0, 0, 0);
            }
        }
        __syncthreads();                              // vmcnt(0): the strips have landed

#pragma unroll
        for (int ks = 0; ks < P / 16; ++ks) {
            const char* gb = gs + wn * 2 * GS_BYTES + ks * 16 * 64 + tr_off;
            const bf16x8 bh = tr_frag(gb, 0);
            const bf16x8 bl = tr_frag(gb, GS_BYTES);
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                if (t < cnt) {
                    const int b = b_lo + t;
                    const int strip = MODE == 0 ? b / 3 : b;
                    const int shift = MODE == 0 ? b - 3 * (b / 3) : 0;
                    const char* xb = xs + strip * 2 * XS_BYTES + (ks * 16 + shift) * 64 + tr_off;
                    const bf16x8 ah = tr_frag(xb, 0);
                    const bf16x8 al = tr_frag(xb, XS_BYTES);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // ---- fp32 partial tiles: out[split][row][co]; D layout: lane&31 = co, register r -> ci
    float* outp = p.out + (size_t)blockIdx.z * p.rows_total * p.N;
    const int co = co0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t >= cnt) continue;
        const int b = b_lo + t;
        int row0;
        if (MODE == 0) {
            row0 = b * p.CK + blockIdx.x * 32;        // tap b, channel block blockIdx.x
        } else {
            const int B = B0 + b;
            if (B >= p.nblk_total) continue;
            const int cib = B / p.T, tap = B - cib * p.T;
            row0 = tap * p.CK + cib * 32;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            outp[(size_t)(row0 + m) * p.N + co] = acc[t][r];
        }
    }
    (void)cblocks;
#endif
}

// ---- ONE-PASS fp16 weight gradient of the 3x3 stride-1 convs (round 5; agp_conv_desc::in_h16 / out_absmax).
// The three-product kernel above moves 812 MB per launch at 4.9 TB/s of fabric traffic with the MFMA pipe 0.30 busy
// (profiles/r05_pmc_train.json): it is bound by BYTES -- both bf16 planes of both operands, every (ci block, co block)
// workgroup of a raster chunk on a different XCD -- before it is bound by its three products.  tools/grad_prec_emul.py
// (fp64 oracle, operand rounding per conv role): a weight gradient from fp16(x) * fp16(g * 2^s) is 7e-4 off (bar 1e-3), any
// two-product bf16 combination 4e-3.  So:
//   x : ONE fp16 plane (written by the pass that produced the map: agp_map_affine's o_h16), LDS-DMA as before, half the bytes;
//   g : the bf16 pair, loaded into registers a K-step ahead, hi + lo summed, scaled by 2^s[co] (s from the exact per-channel
//       maximum the BatchNorm backward folded into out_absmax: max * 2^s in [2^13, 2^14), so no fp16 overflow and 13 binades
//       of full precision below the maximum) and written to LDS as fp16 -- ~56 VALU per thread and K-step beside 5 MFMAs of
//       32 cycles, and no third pass over the gradient to produce an fp16 plane of it;
//   one MFMA product (f16) per tap block; the scale is divided out of the fp32 partial tile;
//   double-buffered stages (the old kernel: load -> barrier -> compute -> barrier) and an XCD-aware grid: the workgroups of a
//   raster chunk ((ci block, co block) tiles) are neighbours on ONE XCD, so the chunk's strips come from HBM once.
struct WgradF16Params {
    const void* x; uint32_t x_bytes;             // fp16 plane of the input map
    const void* g_hi; const void* g_lo; uint32_t g_bytes;
    const uint32_t* gmax;                        // [N] fp32 bit patterns of max |g| per output channel
    int C, N, Wpx;
    int64_t Kpix;
    int k_chunk, rows_total;
    int gx, gy, splits, cpx;                     // tiles, raster chunks, chunks per XCD
    float* out;                                  // [split][rows_total][N]
};

// LDS-DMA of 16 bytes per lane as inline assembly: issued through the builtin, the compiler puts an `s_waitcnt vmcnt` that
// covers every outstanding DMA in front of the first LDS read behind it in program order (SIInsertWaitcnts cannot tell the
// stage being filled from the stage being read), so a transfer issued at the top of a K-step never overlapped that step's
// MFMAs (3-4 us per K-step, MFMA pipe 0.18 busy).  The kernel that uses this retires its DMAs itself (wait_all_vm + barrier).
// The asm writes m0 and cannot say so: hipcc 7.2 treats m0 on a clobber list as a reserved register ("may not be preserved across
// the asm statement, and clobbering them may lead to undefined behaviour").  The kernels that call this therefore must not contain
// ANY compiler-generated m0 use -- no __builtin_amdgcn_raw_ptr_buffer_load_lds / global_load_lds, no ds_*_addtid, no s_movrel:
// wgrad_f16_kernel and wgrad_gather_f16_kernel issue every LDS-DMA through this function.
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 raw_rsrc(const void* ptr, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)ptr;
    const i32x4 r = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    return r;
}
__device__ __forceinline__ void dma16_hidden(const i32x4& rsrc, uint32_t lds_addr, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void wait_all_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ void __launch_bounds__(256, 2) wgrad_f16_kernel(WgradF16Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // A workgroup owns 9 taps x 64 input channels x 64 output channels; wave = (co half, ci half) with all nine taps of its
    // 32 x 32 channel block in registers (144 accumulators): a K-step of 64 raster positions is 36 MFMAs per wave on 46 KB of
    // operands, straight-line (the compiler keeps the fragment reads several MFMAs ahead).
    constexpr int P = 64, XR = P + 16;
    constexpr int XS_BYTES = XR * 64, XH_BYTES = 3 * XS_BYTES, X_BYTES = 2 * XH_BYTES, GH_BYTES = P * 64, STAGE = X_BYTES + 2 * GH_BYTES;
    constexpr int XCH = XR / 16, NINS_X = 2 * 3 * XCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const sc_tab = (float*)(smem + 2 * STAGE);          // [64] operand scale 2^s, [64] its inverse
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // ---- XCD-aware order: XCD x owns the raster chunks [x * cpx, (x + 1) * cpx), a chunk's tiles are consecutive there
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int tiles = p.gx * p.gy;
    const int cl = j / tiles, tile = j - cl * tiles;
    const int chunk = xcd * p.cpx + cl;
    if (cl >= p.cpx || chunk >= p.splits) return;
    const int bx = tile % p.gx, by = tile / p.gx;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wc = wave >> 1;
    const int co0 = by * 64;
    const int64_t k_begin = (int64_t)chunk * p.k_chunk;
    const int nk = p.k_chunk / P;

    const i32x4 rx = raw_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rg_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.g_hi, 0, p.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg_lo = __builtin_amdgcn_make_buffer_rsrc((void*)p.g_lo, 0, p.g_bytes, 0x00020000);

    // per-channel operand scale: max * 2^s in [2^13, 2^14)
    if (tid < 64) {
        const uint32_t bits = p.gmax[co0 + tid];
        int ex = 13 - ((int)((bits >> 23) & 0xffu) - 127);
        if (bits == 0u) ex = 0;
        ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
        sc_tab[tid] = __builtin_bit_cast(float, (uint32_t)(127 + ex) << 23);
        sc_tab[64 + tid] = __builtin_bit_cast(float, (uint32_t)(127 - ex) << 23);
    }
    __syncthreads();
    const int cc = tid & 7;                      // this thread's 8-channel chunk of a gradient row (same in every K-step)
    float gsc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gsc[e] = sc_tab[cc * 8 + e];

    const int lrow = lane >> 2, lchunk = (lane & 3) << 4;
    const int tr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;
    const int g_row = tid >> 3;                  // rows g_row and g_row + 32 of a K-step's 64
    const int g_lds = (cc >> 2) * GH_BYTES + g_row * 64 + (cc & 3) * 16;

    // Per-lane byte offsets of this wave's DMA instructions and of this thread's gradient chunks for the FIRST K-step; a K-step
    // later every one of them is one 32-bit add further (the first form recomputed them from 64-bit pixel indices with bounds
    // tests: 8 vector + 5 scalar instructions per MFMA, the waves 40 % of their time issuing them).  No bounds tests: an offset
    // before the plane wraps to >= 2^31 and one past its end is >= num_records -- both read as zero (the planes are < 2^31 bytes).
    int xoff[8];
    uint32_t xdst[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int i = wave + 4 * it;
        const int c = i % XCH, sx = (i / XCH) % 3, hf = i / (3 * XCH);
        const int64_t pix = k_begin + (int64_t)(sx - 1) * p.Wpx - 1 + c * 16 + lrow;
        xoff[it] = (int)(uint32_t)((pix * p.C + bx * 64 + hf * 32) * 2 + lchunk);
        xdst[it] = __builtin_amdgcn_readfirstlane(smem_lds + hf * XH_BYTES + sx * XS_BYTES + c * 1024);
    }
    const int xstep = P * p.C * 2, gstep = P * p.N * 2;
    int goff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) goff[q] = (int)(uint32_t)(((k_begin + g_row + 32 * q) * p.N + co0 + cc * 8) * 2);

    auto issue_x = [&](int stage_off) {              // the NEXT K-step's strips (the offsets are advanced here)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (it < 7 || wave < NINS_X - 28) dma16_hidden(rx, xdst[it] + stage_off, xoff[it]);
            xoff[it] += xstep;
        }
    };
    auto load_g = [&](u32x4 (&r)[2][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            r[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rg_hi, goff[q], 0, 0);
            r[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rg_lo, goff[q], 0, 0);
            goff[q] += gstep;
        }
    };
    auto store_g = [&](const u32x4 (&r)[2][2], char* stage) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float h[8], l[8];
            unpack8(r[q][0], h);
            unpack8(r[q][1], l);
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (h[e] + l[e]) * gsc[e];
            *(u32x4*)(stage + X_BYTES + g_lds + q * 32 * 64) = pack8_h(h);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    u32x4 gr[2][2];
    if (k_begin < p.Kpix) {
        load_g(gr);
        issue_x(0);
        store_g(gr, smem);
    }
    wait_all_vm();
    __syncthreads();                                  // stage 0 has landed

    for (int kt = 0; kt < nk; ++kt) {
        const int64_t p0 = k_begin + (int64_t)kt * P;
        if (p0 >= p.Kpix) break;                      // uniform over the workgroup
        char* const cur = smem + (kt & 1) * STAGE;
        char* const nxt = smem + ((kt + 1) & 1) * STAGE;
        const bool more = kt + 1 < nk && p0 + P < p.Kpix;
        if (more) {                                   // the next K-step's operands travel while this one is multiplied
            load_g(gr);
            issue_x(((kt + 1) & 1) * STAGE);
        }
        const char* const xs = cur + wc * XH_BYTES;
        const char* const gs = cur + X_BYTES + wn * GH_BYTES;
#pragma unroll
        for (int ks = 0; ks < P / 16; ++ks) {
            const f16x8 bh = __builtin_bit_cast(f16x8, tr_frag(gs + ks * 16 * 64 + tr_off, 0));
#pragma unroll
            for (int b = 0; b < 9; ++b) {
                const int strip = b / 3, shift = b - 3 * strip;
                const f16x8 ah = __builtin_bit_cast(f16x8, tr_frag(xs + strip * XS_BYTES + (ks * 16 + shift) * 64 + tr_off, 0));
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
            }
        }
        if (more) store_g(gr, nxt);
        wait_all_vm();                                // the hidden DMAs of the next stage
        __syncthreads();                              // this step's reads are done, the next stage has landed
    }

    // ---- fp32 partial tiles: out[split][row][co], the operand scale divided out; D layout: lane&31 = co, register r -> ci
    float* outp = p.out + (size_t)chunk * p.rows_total * p.N;
    const int co = co0 + wn * 32 + (lane & 31);
    const float inv = sc_tab[64 + wn * 32 + (lane & 31)];
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        const int row0 = b * p.C + bx * 64 + wc * 32;  // tap b, this wave's 32 input channels
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            outp[(size_t)(row0 + m) * p.N + co] = acc[b][r] * inv;
        }
    }
#endif
}

// ---- ONE-PASS fp16 weight gradient of the GATHER shapes (stride-2 3x3, 1x1 downsample, the packed 7x7 stem;
// agp_conv_desc::in_h16 / out_absmax).
// wgrad_tr_kernel<1, NB> with the operands of wgrad_f16_kernel: x is ONE fp16 plane (half the gathered bytes -- the gather form
// is bound by its L2 traffic: every tap of every output position is a 64-byte piece of its own), g goes through registers
// (hi + lo, times 2^s[co] from the exact per-channel maximum, fp16 into LDS), one MFMA product instead of three, the scale
// divided out of the partial tile.  Same staging order as wgrad_tr_kernel (stage, barrier, multiply, barrier; three workgroups
// per CU overlap each other).
template <int NB>
__global__ void __launch_bounds__(256, 3) wgrad_gather_f16_kernel(WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int P = 32;
    constexpr int XS_BYTES = P * 64, GS_BYTES = P * 64;
    constexpr int X_BYTES = NB * XS_BYTES;            // [strip][row][64 B]
    constexpr int MAXT = (NB + 1) / 2;
    constexpr int XCH = P / 16;
    constexpr int NINS_X = NB * XCH;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs = smem;
    char* const gs = smem + X_BYTES;                  // [co half][row][64 B]
    float* const sc_tab = (float*)(smem + X_BYTES + 2 * GS_BYTES);      // [64] operand scale 2^s, [64] its inverse

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wt = wave >> 1;
    const int b_lo = wt ? (NB + 1) / 2 : 0;
    const int cnt = wt ? NB / 2 : (NB + 1) / 2;
    const int B0 = blockIdx.x * NB;
    const int co0 = blockIdx.y * 64;
    const int64_t k_begin = (int64_t)blockIdx.z * p.k_chunk;
    const int nk = p.k_chunk / P;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.g_hi, 0, p.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg_lo = __builtin_amdgcn_make_buffer_rsrc((void*)p.g_lo, 0, p.g_bytes, 0x00020000);

    if (tid < 64) {                                   // per-channel operand scale: max * 2^s in [2^13, 2^14)
        const uint32_t bits = p.gmax[co0 + tid];
        int ex = 13 - ((int)((bits >> 23) & 0xffu) - 127);
        if (bits == 0u) ex = 0;
        ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
        sc_tab[tid] = __builtin_bit_cast(float, (uint32_t)(127 + ex) << 23);
        sc_tab[64 + tid] = __builtin_bit_cast(float, (uint32_t)(127 - ex) << 23);
    }
    __syncthreads();
    const int cc = tid & 7, g_row = tid >> 3;         // this thread's 8-channel chunk of gradient row g_row of a K-step
    float gsc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gsc[e] = sc_tab[cc * 8 + e];
    const int g_lds = (cc >> 2) * GS_BYTES + g_row * 64 + (cc & 3) * 16;

    const int lrow = lane >> 2, lchunk = (lane & 3) << 4;
    const int tr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;

    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
        const int64_t p0 = k_begin + (int64_t)kt * P;
        if (p0 >= p.Kpix) break;                      // wave-uniform
        if (kt) __syncthreads();                      // previous step's fragment reads are done
        // the gradient rows first (their latency runs beside the strip addressing below)
        u32x4 gh, gl;
        {
            const int64_t off64 = ((p0 + g_row) * p.N + co0 + cc * 8) * 2;
            const int off = off64 < (int64_t)p.g_bytes ? (int)off64 : 0x7ffffff0;
            gh = __builtin_amdgcn_raw_buffer_load_b128(rg_hi, off, 0, 0);
            gl = __builtin_amdgcn_raw_buffer_load_b128(rg_lo, off, 0, 0);
        }
        int xbase[XCH];
        bool xval[XCH];
#pragma unroll
        for (int c = 0; c < XCH; ++c) {
            const int64_t pg = p0 + c * 16 + lrow;
            const uint32_t q = (uint32_t)(pg < p.Kpix ? pg : 0);
            const uint32_t img = fdiv(q, p.d_hopwop);
            const uint32_t rem = q - img * p.d_hopwop.d;
            const uint32_t yy = fdiv(rem, p.d_wop);
            const uint32_t xx = rem - yy * p.d_wop.d;
            xval[c] = pg < p.Kpix && yy >= 1 && yy <= (uint32_t)p.Ho && xx >= 1 && xx <= (uint32_t)p.Wo;
            xbase[c] = ((int)img * p.Hpx + p.stride * ((int)yy - 1) + p.d0) * p.Wpx + p.stride * ((int)xx - 1) + p.d0;
        }
        for (int j = wave; j < NINS_X; j += 4) {
            const int c = j % XCH, s = j / XCH;
            const int B = B0 + s;
            const int cib = B / p.T, tap = B - cib * p.T;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const bool ok = (c == 0 ? xval[0] : xval[XCH - 1]) && B < p.nblk_total;
            const int xpix = (c == 0 ? xbase[0] : xbase[XCH - 1]) + ky * p.Wpx + kx;
            const int off = ok ? (xpix * p.C + cib * 32) * 2 + lchunk : 0x7ffffff0;
            const int dst = __builtin_amdgcn_readfirstlane(s * XS_BYTES + c * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + dst), 16, off, 0, 0, 0);
        }
        {
            float f[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f[2 * i] = (bf2f((bf16_t)(gh[i] & 0xffffu)) + bf2f((bf16_t)(gl[i] & 0xffffu))) * gsc[2 * i];
                f[2 * i + 1] = (bf2f((bf16_t)(gh[i] >> 16)) + bf2f((bf16_t)(gl[i] >> 16))) * gsc[2 * i + 1];
            }
            *(u32x4*)(gs + g_lds) = pack8_h(f);
        }
        __syncthreads();                              // vmcnt(0) lgkmcnt(0): the strips have landed

#pragma unroll
        for (int ks = 0; ks < P / 16; ++ks) {
            const bf16x8 bfr = tr_frag(gs + wn * GS_BYTES + ks * 16 * 64 + tr_off, 0);
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                if (t < cnt) {
                    const bf16x8 afr = tr_frag(xs + (b_lo + t) * XS_BYTES + ks * 16 * 64 + tr_off, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr), __builtin_bit_cast(f16x8, bfr), acc[t], 0, 0, 0);
                }
            }
        }
    }

    float* outp = p.out + (size_t)blockIdx.z * p.rows_total * p.N;
    const int co = co0 + wn * 32 + (lane & 31);
    const float inv = sc_tab[64 + wn * 32 + (lane & 31)];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t >= cnt) continue;
        const int B = B0 + b_lo + t;
        if (B >= p.nblk_total) continue;
        const int cib = B / p.T, tap = B - cib * p.T;
        const int row0 = tap * p.CK + cib * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            outp[(size_t)(row0 + m) * p.N + co] = acc[t][r] * inv;
        }
    }
#endif
}

template <int NB>
int launch_wgrad_gather_f16(const WgradParams& p, dim3 grid, hipStream_t s) {
    constexpr int lds = NB * 32 * 64 + 2 * 32 * 64 + 512;
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)wgrad_gather_f16_kernel<NB>, lds, attr_done)) return AGP_E_LAUNCH;
    AGP_LAUNCH((wgrad_gather_f16_kernel<NB>), grid, dim3(256), lds, s, p);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

// param_taps > 0: `out` is the nn.Conv2d parameter layout [cout][cin][taps] (i runs over [tap][cin][cout]); the scattered 4-byte
// stores are the weight tensor once, against `splits` reads of it.  accumulate: out += (a gradient that already exists).
// A workgroup sums 64 consecutive elements: wave w adds the splits w, w + 4, w + 8, ... in that order (eight independent loads
// per trip), the four wave sums are added in wave order -- a fixed order whatever the grid.  (One thread per element walking all
// the splits: 32 us for the 36 864 elements x 512 splits of a 64-channel conv, 144 workgroups on 256 CUs.)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, int splits, int64_t count,
                                                            float* __restrict__ out, int param_taps = 0, int cin = 0, int cout = 0,
                                                            int accumulate = 0, uint32_t* zero_words = nullptr, int nzero = 0) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (zero_words && z < nzero) zero_words[z] = 0u;    // (the weight-gradient kernel, this one's predecessor, has read them)
    }
    for (int64_t base = (int64_t)blockIdx.x * 64; base < count; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + lane;
        float s = 0.f;
        if (i < count) {
            for (int k0 = w; k0 < splits; k0 += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k0 + 4 * u < splits ? k0 + 4 * u : splits - 1) * count + i];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + 4 * u < splits) s += v[u];
            }
        }
        red[w][lane] = s;
        __syncthreads();
        if (w == 0 && i < count) {
            const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
            int64_t o = i;
            if (param_taps > 0) {
                const int co = (int)(i % cout);
                const int64_t r = i / cout;
                const int ci = (int)(r % cin), tap = (int)(r / cin);
                o = ((int64_t)co * cin + ci) * param_taps + tap;
            }
            out[o] = accumulate ? out[o] + t : t;
        }
        __syncthreads();
    }
}

template <int MODE, int NB>
constexpr int wgrad_lds() {
    constexpr int P = MODE == 0 ? 64 : 32;
    constexpr int XR = MODE == 0 ? P + 16 : P;
    constexpr int NXS = MODE == 0 ? 3 : NB;
    return NXS * 2 * XR * 64 + 2 * 2 * P * 64;
}

template <int MODE, int NB>
int launch_wgrad(const WgradParams& p, dim3 grid, hipStream_t s) {
    constexpr int lds = wgrad_lds<MODE, NB>();
    static_assert(lds <= 53 * 1024, "three workgroups per CU");
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)wgrad_tr_kernel<MODE, NB>, lds, attr_done)) return AGP_E_LAUNCH;
    AGP_LAUNCH((wgrad_tr_kernel<MODE, NB>), grid, dim3(256), lds, s, p);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

struct Plan {
    int mode, nb, gx, gy, splits, k_chunk;
    int64_t kpix;
};

inline bool make_plan(const agp_conv_desc* d, Plan& pl) {
    const bool stem = d->in_w_step != d->cin;
    if (d->cout % 64 || d->cin % 32 || d->pout != 1 || d->n <= 0) return false;
    if (stem && !(d->kw == 1 && d->cin == 32)) return false;
    const int T = d->kh * d->kw;
    const bool kxr = !stem && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->pin == 1 &&
                     d->hin == d->hout && d->win == d->wout;
    const int cblocks = d->cin / 32;
    pl.mode = kxr ? 0 : 1;
    if (kxr) { pl.nb = 9; pl.gx = cblocks; }
    else if (T == 9) { pl.nb = 9; pl.gx = cblocks; }
    else if (T == 7 && cblocks == 1) { pl.nb = 7; pl.gx = 1; }
    else if (T == 1) {
        pl.nb = cblocks >= 8 ? 8 : (cblocks >= 4 ? 4 : 2);
        pl.gx = (cblocks + pl.nb - 1) / pl.nb;
    } else {
        return false;
    }
    pl.gy = d->cout / 64;
    pl.kpix = (int64_t)d->n * (d->hout + 2) * (d->wout + 2);
    const int P = pl.mode == 0 ? 64 : 32;
    const int64_t ksteps = (pl.kpix + P - 1) / P;
    const int tiles = pl.gx * pl.gy;
    // every split writes a full fp32 partial tile set: keep >= 24 K-steps behind each of them so the
    // partial traffic (and the reduce pass) stays small next to the MFMA work
    int64_t splits = (1024 + tiles - 1) / tiles;
    if (kxr && d->in_h16 && d->cin % 64 == 0) {
        // one-pass kernel: exactly one round of workgroups on the chip's 512 slots (256 CUs x 2): 1024 workgroups on 768 slots ran as one
        // full round and one a third full -- every workgroup streams the same number of K-steps, so the second round cost a
        // whole first one (96 -> 65 us on the 64-channel maps of the training step)
        const int tiles16 = (d->cin / 64) * pl.gy;       // (the one-pass kernel's tiles: 64 x 64 channels)
        splits = 512 / tiles16;
        if (splits < 1) splits = 1;
        if (splits > ksteps / 8) splits = ksteps / 8;
    } else
    if (splits > ksteps / 24) splits = ksteps / 24;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    const int64_t per = (ksteps + splits - 1) / splits;
    pl.k_chunk = (int)(per * P);
    pl.splits = (int)((ksteps + per - 1) / per);
    return true;
}

}  // namespace agp_wgrad
using namespace agp_wgrad;

extern "C" int64_t agp_conv2d_wgrad_workspace_bytes(const agp_conv_desc* d) {
    Plan pl;
    if (!d || !make_plan(d, pl)) return -1;
    return (int64_t)pl.splits * d->kh * d->kw * d->cin * d->cout * 4;
}

static int conv2d_wgrad_impl(const agp_conv_desc* d, float* gw, void* workspace, int64_t workspace_bytes, void* stream,
                             bool param_layout, int accumulate) {
    Plan pl;
    if (!d || !gw || !workspace || !d->in_hi || !d->out_hi || !d->out_lo) return AGP_E_BADARG;
    if (d->prec != AGP_PREC_BF16X3) return AGP_E_BADARG;     // gradients live on split-bf16 maps
    if ((d->in_h16 != nullptr) != (d->out_absmax != nullptr)) return AGP_E_BADARG;
    if (!make_plan(d, pl)) return AGP_E_UNSUPPORTED;
    // in_lo may be NULL only where a one-product kernel runs (they read in_h16, never in_hi / in_lo): the fast training mode keeps
    // some activations as ONE fp16 plane (train_graph.ConvBNUnit.forward, out_f16_only)
    if (!d->in_lo && !(d->in_h16 && ((pl.mode == 0 && d->cin % 64 == 0) || pl.mode == 1))) return AGP_E_BADARG;
    const int64_t rows = (int64_t)d->kh * d->kw * d->cin;
    if (workspace_bytes < (int64_t)pl.splits * rows * d->cout * 4) return AGP_E_BADARG;
    if (d->in_h16 && pl.mode == 0 && d->cin % 64 == 0) {
        // one fp16 product (wgrad_f16_kernel); the gather shapes -- stride-2 entries, 1x1 downsamples AND the packed stem -- take
        // wgrad_gather_f16_kernel below.  A 3x3 stride-1 conv with cin % 64 != 0 runs the three-product kernel; the host never
        // hands such a conv the two fields (train_graph.ConvBNUnit.wgrad_f16_ok)
        const int hp = d->hin + 2, wp = d->win + 2;
        const int64_t x_elems = (int64_t)d->n * hp * wp * d->cin, g_elems = pl.kpix * d->cout;
        if (x_elems * 2 >= (1ll << 31) || g_elems * 2 >= (1ll << 31)) return AGP_E_BADARG;
        WgradF16Params q = {};
        q.x = d->in_h16; q.x_bytes = (uint32_t)(x_elems * 2);
        q.g_hi = d->out_hi; q.g_lo = d->out_lo; q.g_bytes = (uint32_t)(g_elems * 2);
        q.gmax = d->out_absmax;
        q.C = d->cin; q.N = d->cout; q.Wpx = wp; q.Kpix = pl.kpix; q.k_chunk = pl.k_chunk; q.rows_total = (int)rows;
        q.gx = d->cin / 64; q.gy = pl.gy; q.splits = pl.splits; q.cpx = (pl.splits + 7) / 8;
        q.out = (float*)workspace;
        hipStream_t s = (hipStream_t)stream;
        constexpr int lds = 2 * (2 * 3 * 80 * 64 + 2 * 64 * 64) + 512;
        static_assert(lds <= 80 * 1024, "two workgroups per CU");
        static std::atomic<uint64_t> attr_f16{0};
        if (!agp_lds_attr((const void*)wgrad_f16_kernel, lds, attr_f16)) return AGP_E_LAUNCH;
        AGP_LAUNCH(wgrad_f16_kernel, dim3(8 * q.cpx * q.gx * q.gy), dim3(256), lds, s, q);
        AGP_CHECK_LAUNCH();
        const int64_t count = rows * d->cout;
        int blocks = (int)((count + 63) / 64);
        if (blocks > 4096) blocks = 4096;
        AGP_LAUNCH(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, pl.splits, count, gw,
                   param_layout ? d->kh * d->kw : 0, d->cin, d->cout, accumulate, d->out_absmax, d->cout);
        AGP_CHECK_LAUNCH();
        return AGP_OK;
    }
    const int hp = d->hin + 2 * d->pin, wp = d->win + 2 * d->pin;
    const int64_t x_elems = (int64_t)d->n * hp * wp * d->in_w_step;
    const int64_t g_elems = pl.kpix * d->cout;
    if (x_elems * 2 >= (1ll << 31) || g_elems * 2 >= (1ll << 31)) return AGP_E_BADARG;
    WgradParams p = {};
    p.x_hi = d->in_hi; p.x_lo = d->in_lo; p.x_bytes = (uint32_t)(x_elems * 2);
    p.g_hi = d->out_hi; p.g_lo = d->out_lo; p.g_bytes = (uint32_t)(g_elems * 2);
    p.C = d->in_w_step; p.CK = d->cin; p.N = d->cout;
    p.T = d->kh * d->kw; p.KW = d->kw;
    p.Hpx = hp; p.Wpx = wp; p.Ho = d->hout; p.Wo = d->wout;
    p.d_hopwop = make_fastdiv((uint32_t)((d->hout + 2) * (d->wout + 2)));
    p.d_wop = make_fastdiv((uint32_t)(d->wout + 2));
    p.stride = d->stride; p.d0 = d->pin - d->pad;
    p.Kpix = pl.kpix; p.k_chunk = pl.k_chunk;
    p.nblk_total = p.T * (d->cin / 32); p.rows_total = (int)rows;
    p.out = (pl.splits == 1 && !param_layout) ? gw : (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(pl.gx, pl.gy, pl.splits);
    int rc;
    // gather shapes (the packed stem included) with an fp16 operand plane and the gradient's per-channel maxima: one fp16 product
    const bool g16 = pl.mode == 1 && d->in_h16;
    if (g16) {
        p.x_hi = d->in_h16; p.x_lo = nullptr; p.gmax = d->out_absmax;
        p.out = (float*)workspace;
        if (pl.nb == 9) rc = launch_wgrad_gather_f16<9>(p, grid, s);
        else if (pl.nb == 8) rc = launch_wgrad_gather_f16<8>(p, grid, s);
        else if (pl.nb == 7) rc = launch_wgrad_gather_f16<7>(p, grid, s);
        else if (pl.nb == 4) rc = launch_wgrad_gather_f16<4>(p, grid, s);
        else rc = launch_wgrad_gather_f16<2>(p, grid, s);
        if (rc != AGP_OK) return rc;
        const int64_t count = rows * d->cout;
        int blocks = (int)((count + 63) / 64);
        if (blocks > 4096) blocks = 4096;
        AGP_LAUNCH(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, pl.splits, count, gw,
                   param_layout ? d->kh * d->kw : 0, d->cin, d->cout, accumulate, d->out_absmax, d->cout);
        AGP_CHECK_LAUNCH();
        return AGP_OK;
    }
    if (pl.mode == 0) rc = launch_wgrad<0, 9>(p, grid, s);
    else if (pl.nb == 9) rc = launch_wgrad<1, 9>(p, grid, s);
    else if (pl.nb == 8) rc = launch_wgrad<1, 8>(p, grid, s);
    else if (pl.nb == 7) rc = launch_wgrad<1, 7>(p, grid, s);
    else if (pl.nb == 4) rc = launch_wgrad<1, 4>(p, grid, s);
    else rc = launch_wgrad<1, 2>(p, grid, s);
    if (rc != AGP_OK) return rc;
    if (pl.splits > 1 || param_layout) {
        const int64_t count = rows * d->cout;
        int blocks = (int)((count + 63) / 64);
        if (blocks > 4096) blocks = 4096;
        AGP_LAUNCH(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, pl.splits, count, gw,
                   param_layout ? d->kh * d->kw : 0, d->cin, d->cout, accumulate);
        AGP_CHECK_LAUNCH();
    }
    return AGP_OK;
}

extern "C" int agp_conv2d_wgrad(const agp_conv_desc* d, float* gw, void* workspace, int64_t workspace_bytes, void* stream) {
    return conv2d_wgrad_impl(d, gw, workspace, workspace_bytes, stream, false, 0);
}

extern "C" int agp_conv2d_wgrad_param(const agp_conv_desc* d, float* gw, int accumulate, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
    if (d && d->in_w_step != d->cin) return AGP_E_UNSUPPORTED;        // the packed stem's rows are not the parameter's
    return conv2d_wgrad_impl(d, gw, workspace, workspace_bytes, stream, true, accumulate ? 1 : 0);
}

// ---- sparse convolution weight gradient: gw[tap][ci][co] = sum_i x[nbr[tap][i]][ci] * g[i][co]
extern "C" int64_t agp_sparse_conv_wgrad_workspace_bytes(int64_t n_out, int cin, int cout, int ntaps) {
    if (n_out <= 0 || cin % 32 || cout % 64 || ntaps <= 0) return -1;
    const int nblk = ntaps * (cin / 32);
    const int nb = ntaps >= 9 ? 9 : (nblk >= 8 ? 8 : (nblk >= 4 ? 4 : 2));
    const int tiles = ((nblk + nb - 1) / nb) * (cout / 64);
    const int64_t ksteps = (n_out + 31) / 32;
    int64_t splits = (1024 + tiles - 1) / tiles;
    if (splits > ksteps / 24) splits = ksteps / 24;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    return splits * ntaps * cin * cout * 4;
}

extern "C" int agp_sparse_conv_wgrad(const void* x_hi, const void* x_lo, int64_t n_in_rows, const int32_t* nbr, int64_t n_out,
                                     int cin, int cout, int ntaps, const void* g_hi, const void* g_lo, float* gw, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    if (!x_hi || !x_lo || !nbr || !g_hi || !g_lo || !gw || !workspace || n_out <= 0 || n_in_rows <= 0) return AGP_E_BADARG;
    if (cin % 32 || cout % 64 || ntaps <= 0) return AGP_E_BADARG;
    if (n_in_rows * cin * 2 >= (1ll << 31) || n_out * (int64_t)cout * 2 >= (1ll << 31)) return AGP_E_BADARG;
    const int nblk = ntaps * (cin / 32);
    const int nb = ntaps >= 9 ? 9 : (nblk >= 8 ? 8 : (nblk >= 4 ? 4 : 2));
    const int gx = (nblk + nb - 1) / nb, gy = cout / 64;
    const int64_t ksteps = (n_out + 31) / 32;
    int64_t splits = (1024 + gx * gy - 1) / (gx * gy);
    if (splits > ksteps / 24) splits = ksteps / 24;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    const int64_t per = (ksteps + splits - 1) / splits;
    splits = (ksteps + per - 1) / per;
    const int64_t rows = (int64_t)ntaps * cin;
    if (workspace_bytes < splits * rows * cout * 4) return AGP_E_BADARG;
    WgradParams p = {};
    p.x_hi = x_hi; p.x_lo = x_lo; p.x_bytes = (uint32_t)(n_in_rows * cin * 2);
    p.g_hi = g_hi; p.g_lo = g_lo; p.g_bytes = (uint32_t)(n_out * (int64_t)cout * 2);
    p.C = cin; p.CK = cin; p.N = cout; p.T = ntaps; p.KW = ntaps;
    p.Kpix = n_out; p.k_chunk = (int)(per * 32);
    p.nblk_total = nblk; p.rows_total = (int)rows;
    p.out = splits == 1 ? gw : (float*)workspace;
    p.tab = nbr; p.tab_stride = (int)n_out;
    p.d_hopwop = make_fastdiv(1); p.d_wop = make_fastdiv(1);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(gx, gy, (unsigned)splits);
    int rc;
    if (nb == 9) rc = launch_wgrad<2, 9>(p, grid, s);
    else if (nb == 8) rc = launch_wgrad<2, 8>(p, grid, s);
    else if (nb == 4) rc = launch_wgrad<2, 4>(p, grid, s);
    else rc = launch_wgrad<2, 2>(p, grid, s);
    if (rc != AGP_OK) return rc;
    if (splits > 1) {
        const int64_t count = rows * cout;
        int blocks = (int)((count + 63) / 64);
        if (blocks > 4096) blocks = 4096;
        AGP_LAUNCH(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, (int)splits, count, gw);
        AGP_CHECK_LAUNCH();
    }
    return AGP_OK;
}
