// Implicit-GEMM 3x3 stride-1 convolution with horizontal-tap reuse in LDS ("kxr").
//
// Measured on MI355X the generic kernel (igemm.hip) is bound by LDS WRITE bandwidth: every tap
// re-stages its 256-pixel X tile by LDS-DMA (~80 B/clk/CU into LDS), about as many LDS cycles as
// the MFMAs of that tap take.  For a 3x3 stride-1 pad-1 conv on the halo-padded NHWC planes the
// GEMM rows can run over the PADDED-WIDTH raster (img, y, x' in [0, W+2)): consecutive GEMM rows
// are then consecutive pixels in memory, and the input pixel of output row m at tap (ky,kx) is
//     rowbase(m) + (ky*Wp + kx) * Cin,     rowbase(m) = ((img*Hp + y)*Wp + x' - 1) * Cin,
// i.e. the three horizontal taps are the SAME staged row block read at row offsets 0, 1, 2.
// One macro-step = (ky, 32-channel chunk): stage BM+16 X rows once and the three W taps, then run
// 3 x (BK/16) x tiles x passes MFMAs from it: X LDS-DMA traffic / 3, 72 MFMAs per wave (split-bf16)
// between barriers.  Outputs at the two halo columns (x' = 0, W+1) are computed and discarded
// (2/(W+2) waste).  No double buffering: the stage is 58-66 KB, so TWO workgroups fit per CU and
// alternate -- one stages while the other computes.
#include "igemm_params.hpp"

namespace agp_igemm {

__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }

template <int WM, int WN, int NPREC>
constexpr int kxr_lds_bytes() {
    constexpr int stage = ((WM * 64 + 16) + 3 * WN * 64) * 64 * (NPREC == 3 ? 2 : 1);
    constexpr int epi = WM * WN * 32 * EPI_ROWB;
    return stage > epi ? stage : epi;
}

template <int WM, int WN, int NPREC>
__global__ void __launch_bounds__(256, 2) igemm_kxr_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * 64, BN = WN * 64, NW = 4;
    constexpr int BMX = BM + 16;                 // staged X rows (the extra block feeds kx = 1, 2)
    constexpr int ROWB = 64;                     // 32 bf16 per row
    constexpr int NPL = (NPREC == 3) ? 2 : 1;
    constexpr int X_PLANE = BMX * ROWB, W_TAP = BN * ROWB, W_PLANE = 3 * W_TAP;
    constexpr int XINS = BMX / 16;               // X LDS-DMA instructions per plane (16 rows each)
    constexpr int XI = (XINS + NW - 1) / NW;
    constexpr int WINS = 3 * BN / 16;            // W instructions per plane (3 taps)
    constexpr int WI = (WINS + NW - 1) / NW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs_hi = smem;
    char* const xs_lo = smem + X_PLANE;                  // only when NPL == 2
    char* const ws_hi = smem + X_PLANE * NPL;
    char* const ws_lo = ws_hi + W_PLANE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int nt = j % p.NT;
    const int mt = xcd * p.mt_chunk + j / p.NT;
    if (mt >= p.MT) return;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- LDS-DMA source offsets.  X instruction i covers LDS rows 16i..16i+15 = GEMM rows m0+16i+..
    const int lrow = lane >> 2, lpos = lane & 3;
    int xoff[XI], woff[WI];
#pragma unroll
    for (int q = 0; q < XI; ++q) {
        const int row = (wave + NW * q) * 16 + lrow;
        // NOT clamped to M-1: rows past the last image must read zeros (beyond the plane the
        // buffer range check returns 0), because they are the kx = 1,2 neighbours of the last rows.
        const int m = m0 + row;
        const uint32_t img = fdiv((uint32_t)m, p.d_howo);
        const uint32_t rem = (uint32_t)m - img * p.d_howo.d;
        const uint32_t y = fdiv(rem, p.d_wo);
        const uint32_t xq = rem - y * p.d_wo.d;
        const int el = (int)img * p.x_sn + (int)y * p.x_sh + (int)xq * p.x_sw + p.x_base;
        xoff[q] = el * 2 + ((lpos ^ swz32(row)) << 4);
    }
#pragma unroll
    for (int q = 0; q < WI; ++q) {
        const int ins = wave + NW * q;               // instruction index over 3 taps x BN/16
        const int tap = ins / (BN / 16), row = (ins % (BN / 16)) * 16 + lrow;
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        // tap kx adds kx*CK elements along K
        woff[q] = (n * p.Ktot + tap * p.CK) * 2 + ((lpos ^ swz32(row)) << 4);
    }
    const __amdgpu_buffer_rsrc_t rx_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(NPREC == 3 ? p.x_lo : p.x_hi), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(NPREC == 3 ? p.w_lo : p.w_hi), 0, p.w_bytes, 0x00020000);

    // ---- fragment read offsets: X rows shifted by kx, W rows per tap
    const int l31 = lane & 31, lh = lane >> 5;
    int xro[3][2], xsw[3][2], wro[2], wsw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int r = wm * 64 + t * 32 + l31 + kx;
            xro[kx][t] = r * ROWB;
            xsw[kx][t] = swz32(r);
        }
        const int wr = wn * 64 + t * 32 + l31;
        wro[t] = wr * ROWB;
        wsw[t] = swz32(wr);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // timing experiment: stagger half of the workgroups by ~half a macro-step (dbg bits)
    if (p.dbg & 0x300) {
        const int sel = (p.dbg & 0x100) ? (bid >> 8) & 1 : (bid >> 3) & 1;
        if (sel) for (int i = 0; i < (p.dbg >> 12); ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int cchunks = p.CK / 32;
    const int nsteps = 3 * cchunks;                  // (ky, cc) macro-steps
    int ky = 0, cc = 0;
    for (int st = 0; st < nsteps; ++st) {
        // stage X(ky,cc) and W(ky, kx=0..2, cc)
        const int xs = __builtin_amdgcn_readfirstlane((ky * p.x_sh + cc * 32) * 2);
        const int ws = __builtin_amdgcn_readfirstlane((ky * 3 * p.CK + cc * 32) * 2);
        if (st) __syncthreads();                     // previous macro-step's fragment reads are done
        if (!((p.dbg & 1) && st > 0)) {
#pragma unroll
        for (int q = 0; q < XI; ++q) {
            const int ins = wave + NW * q;
            if (ins < XINS) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_hi, LDS_PTR(xs_hi + ins * 1024), 16, xoff[q], xs, 0, 0);
                if (NPREC == 3)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_lo, LDS_PTR(xs_lo + ins * 1024), 16, xoff[q], xs, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < WI; ++q) {
            const int ins = wave + NW * q;
            if (ins < WINS) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_hi, LDS_PTR(ws_hi + ins * 1024), 16, woff[q], ws, 0, 0);
                if (NPREC == 3)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_lo, LDS_PTR(ws_lo + ins * 1024), 16, woff[q], ws, 0, 0);
            }
        }
        }
        if (++cc == cchunks) { cc = 0; ++ky; }
        __syncthreads();                             // vmcnt(0): the stage has landed for every wave
        if (!(p.dbg & 4))
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int xo = xro[kx][t] + (((2 * ks + lh) ^ xsw[kx][t]) << 4);
                    const int wo = kx * W_TAP + wro[t] + (((2 * ks + lh) ^ wsw[t]) << 4);
                    xh[t] = *(const bf16x8*)(xs_hi + xo);
                    wh[t] = *(const bf16x8*)(ws_hi + wo);
                    if (NPREC == 3) {
                        xl[t] = *(const bf16x8*)(xs_lo + xo);
                        wl[t] = *(const bf16x8*)(ws_lo + wo);
                    }
                }
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) {
                        if (NPREC == 3) {
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[tn], xh[tm], acc[tn][tm], 0, 0, 0);
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[tn], xl[tm], acc[tn][tm], 0, 0, 0);
                        }
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[tn], xh[tm], acc[tn][tm], 0, 0, 0);
                    }
            }
        }
    }

    // ---- epilogue (as igemm.hip), halo columns masked
    __syncthreads();
    char* er = smem + wave * (32 * EPI_ROWB);
    const int ch = lane & 7;
    const int nglob = n0 + wn * 64 + ch * 8;
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
    const uint32_t wlast = p.d_wo.d - 1;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        if (tm) __syncthreads();
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
                *(f32x4*)(er + l31 * EPI_ROWB + (tn * 32 + 8 * q + 4 * lh) * 4) = v;
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int ml = it * 8 + (lane >> 3);
            const int m = m0 + wm * 64 + tm * 32 + ml;
            if (m >= p.M) continue;
            const uint32_t img = fdiv((uint32_t)m, p.d_howo);
            const uint32_t rem = (uint32_t)m - img * p.d_howo.d;
            const uint32_t y = fdiv(rem, p.d_wo);
            const uint32_t xq = rem - y * p.d_wo.d;
            if (xq == 0 || xq == wlast) continue;     // halo column: keep the zeros
            const f32x4 a = *(const f32x4*)(er + ml * EPI_ROWB + ch * 32);
            const f32x4 b = *(const f32x4*)(er + ml * EPI_ROWB + ch * 32 + 16);
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            const size_t off = (size_t)img * p.o_sn + (size_t)y * p.o_sh + (size_t)xq * p.o_sw + p.o_base + nglob;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (rhi) {
                float r[8];
                unpack8(*(const u32x4*)(rhi + off), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
                if (rlo) {
                    unpack8(*(const u32x4*)(rlo + off), r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u32x4 h, l;
            split8(v, h, l);
            *(u32x4*)(ohi + off) = h;
            if (olo) *(u32x4*)(olo + off) = l;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int WM, int WN, int NPREC>
int launch_kxr(IgemmParams& p, hipStream_t s) {
    constexpr int lds = kxr_lds_bytes<WM, WN, NPREC>();
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)igemm_kxr_kernel<WM, WN, NPREC>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return AGP_E_LAUNCH;
        attr_set = true;
    }
    constexpr int BM = WM * 64, BN = WN * 64;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.N + BN - 1) / BN;
    p.mt_chunk = (p.MT + 7) / 8;
    AGP_LAUNCH((igemm_kxr_kernel<WM, WN, NPREC>), dim3(p.mt_chunk * 8 * p.NT), dim3(256), lds, s, p);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

}  // namespace agp_igemm

// 3x3 / stride 1 / pad 1 convs on 1-pixel-halo planes.  `p` arrives with the generic geometry;
// this rewrites it for the padded-width raster.
int agp_internal_conv_kxr(agp_igemm::IgemmParams& p, const agp_conv_desc* d, hipStream_t s) {
    using namespace agp_igemm;
    const int hp = d->hin + 2, wp = d->win + 2;
    p.M = d->n * d->hin * wp;                          // rows: (img, y, x' in [0, wp))
    p.d_howo = make_fastdiv((uint32_t)(d->hin * wp));
    p.d_wo = make_fastdiv((uint32_t)wp);
    p.x_sw = d->cin; p.x_sh = wp * d->cin; p.x_sn = hp * wp * d->cin;
    p.x_base = -d->cin;                                 // pixel (y + ky, x' + kx - 1)
    p.o_sw = d->cout; p.o_sh = wp * d->cout; p.o_sn = hp * wp * d->cout;
    p.o_base = wp * d->cout;                            // padded row y + 1, padded column x'
    const bool wide = (p.N % 128 == 0);
    if (d->prec == AGP_PREC_BF16X3) return wide ? launch_kxr<2, 2, 3>(p, s) : launch_kxr<4, 1, 3>(p, s);
    if (d->prec == AGP_PREC_BF16) return wide ? launch_kxr<2, 2, 1>(p, s) : launch_kxr<4, 1, 1>(p, s);
    return AGP_E_BADARG;
}
