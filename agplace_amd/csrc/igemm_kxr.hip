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

// -DAGP_CENSUS=1 compiles in the per-workgroup census / phase stamps read by tools/census.py
#ifndef AGP_CENSUS
#define AGP_CENSUS 0
#endif

#include <type_traits>
#include <utility>

#include "igemm_params.hpp"

namespace agp_igemm {

template <class F, int... I>
__device__ __forceinline__ void kxr_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    kxr_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}


__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }
// 16x16x32 fragments (lane = row l&15, 16-byte K chunk l>>4): conflict-free ds_read_b128 for every kx row shift
__device__ __forceinline__ int swz16(int row) { return (row >> 1) & 2; }
template <int MF> __device__ __forceinline__ int swz(int row) { return MF == 16 ? swz16(row) : swz32(row); }

// Tile BM x BN per workgroup, WM x WN waves, each wave TM x TN MFMA tiles of 32x32.
template <int BM, int BN, int WM, int WN, int NPREC, int RING>
constexpr int kxr_lds_bytes() {
    constexpr int stage = ((RING >= 2 ? 2 : 1) * (BM + 16) * PrecT<NPREC>::XPL + (RING == 4 ? 4 : (RING ? 2 : 3)) * BN * PrecT<NPREC>::WPL) * 64;
    // the waves' staging rows + the statistics epilogue's [wave][columns][2] sums behind them
    constexpr int epi = WM * WN * 32 * ((BN / WN) * 4 + 16) + WM * WN * (BN / WN) * 8;
    return (stage > epi ? stage : epi) + 2 * BN * 4;     // + the scale/shift table of the direct epilogue
}

// RING = 2: phase pipeline.  A phase = one (macro-step, kx) tap: 2 x TM x TN x products MFMAs per wave.
// The X block is double buffered and EVERY load (the next tap's W; the next macro-step's X together
// with its tap 0) is issued at the start of the phase before the one that consumes it, so no phase
// waits for a full memory round trip: one barrier per phase, nothing staged up front per macro-step.
// RING = 3: as 2, with the DMA issue placed behind the phase's first fragment reads (off the path to the first
// MFMA).  Tiles: 256 x 64 for every width since the measurements of profiles/README.md ("kxr sensitivity"):
// the W tap a workgroup re-stages per phase is what the loop is most sensitive to (8 KB instead of 16 KB
// at equal MFMAs per phase), +8 % on the 128/256-channel layers against 128 x 128 tiles although
// the X block is then staged once per 64-channel column tile (from L2).
// RING = 1: the three W taps of a macro-step go through a 2-slot ring (tap kx=2 is fetched while
// kx=1 computes): 51 KB instead of 59-66 KB per workgroup -> THREE workgroups per CU.
// workgroups per CU the register budget is sized for: 8-tile waves (TM*TN = 8) hold 128 accumulator
// registers -> 2 waves per SIMD; 4-tile waves fit 3 (W ring) or 2 workgroups of 4 waves.
template <int BM, int BN, int WM, int WN, int RING, int NPREC = 0>
constexpr int kxr_min_blocks() {
    constexpr int tiles = (BM / (WM * 32)) * (BN / (WN * 32));
    if (RING == 4) return 1;                             // one wave per SIMD, the whole register file (RING = 4 below)
    if (NPREC == 3 && RING >= 2) return 2;               // two-plane operands, double-buffered X: 70 KB of LDS
    return (WM * WN == 8) ? 2 : (tiles >= 8 ? 2 : (RING ? 3 : 2));
}

// MF = MFMA shape: 32 = 32x32x16 (two K-steps per tap phase), 16 = 16x16x32 (one): same operand bytes and MFMA
// cycles per phase, but the chip holds a higher clock under load with the 16x16x32 form (MI355X_MICROARCH.md,
// DVFS item 7), and these kernels are clock-limited: the same launch runs 1.37x faster on all-zero operands.
template <int BM, int BN, int WM, int WN, int NPREC, int RING, int MF = 32, bool LDS_EPI = false, bool Q8 = false>
__global__ void __launch_bounds__(WM* WN * 64, (kxr_min_blocks<BM, BN, WM, WN, RING, NPREC>())) igemm_kxr_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(MF == 32 || (MF == 16 && RING >= 2), "the 16x16x32 form exists for the phase-pipelined loop only");
    static_assert(!Q8 || (NPREC == 2 && RING >= 2 && MF == 32), "Q8: the fp8 lo product of the F16W2 mode, phase-pipelined 32x32 loop only");
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);     // MFMA tiles per wave
    constexpr int EROWB = TN * 32 * 4 + 16;                     // epilogue row of one wave
    constexpr int LPP = TN * 4;                                 // lanes per pixel in the read-back
    constexpr int BMX = BM + 16;                 // staged X rows (the extra block feeds kx = 1, 2)
    constexpr int ROWB = 64;                     // 32 bf16 per row
    constexpr int XPL = PrecT<NPREC>::XPL, WPL = PrecT<NPREC>::WPL;
    constexpr int X_PLANE = BMX * ROWB, W_TAP = BN * ROWB, W_PLANE = 3 * W_TAP;
    constexpr int XINS = BMX / 16;               // X LDS-DMA instructions per plane (16 rows each)
    constexpr int XI = (XINS + NW - 1) / NW;
    constexpr int WINS = (RING ? 1 : 3) * BN / 16;  // W instructions per plane per issue group
    constexpr int WI = (WINS + NW - 1) / NW;

    // DIRECT epilogue (fp16 maps, 32x32 MFMA): the W rows a wave feeds to the MFMA are permuted (bits 2 and 3 of
    // the row index swapped) so that accumulator registers 8h..8h+7 of a lane are 8 CONSECUTIVE channels
    // (16h + 8*(lane>>5) + e) of its pixel (lane & 31): 16-byte stores straight from registers, no LDS transpose,
    // no barrier before the epilogue, one output offset per 32-pixel tile instead of one per read-back iteration.
    constexpr bool DIRECT = (PrecT<NPREC>::XPL == 1) && MF == 32 && !LDS_EPI;
    constexpr int BN_TAB = kxr_lds_bytes<BM, BN, WM, WN, NPREC, RING>() - 2 * BN * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs_hi = smem;
    char* const xs_lo = smem + X_PLANE;                  // only when XPL == 2
    char* const ws_hi = smem + X_PLANE * XPL * (RING >= 2 ? 2 : 1);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    if constexpr (Q8) {
        // e4m3 holds +-448: an activation beyond that must SATURATE in the lo product's operand (its share of the result
        // is ~2^-12, so the clamp is harmless), not become NaN.  MODE.FP16_OVFL (bit 23 of HW_REG_MODE) makes the fp8
        // conversions of this wave clamp to the largest finite value (found with a ResNet50 whose unnormalised
        // random-BatchNorm activations reach 10^3: the 256-channel 3x3 convs returned garbage).
        __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    }

    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int nt = j % p.NT;
    const int mt = xcd * p.mt_chunk + j / p.NT;
    if (mt >= p.MT) return;
    const int m0 = mt * BM, n0 = nt * BN;
#if AGP_CENSUS
    unsigned long long t_begin = 0;
    const bool census = (p.dbg & 0x1000000) != 0;
    unsigned long long* stamp = nullptr;
    int nstamp = 0;
    if (census) {
        t_begin = __builtin_amdgcn_s_memrealtime();
        stamp = (unsigned long long*)p.gmin + (size_t)blockIdx.x * 64 + 4;
    }
#define AGP_STAMP() do { if (census && tid == 0 && nstamp < 56) stamp[nstamp] = __builtin_amdgcn_s_memtime(); ++nstamp; } while (0)
#else
#define AGP_STAMP() do { } while (0)
#endif

    // ---- LDS-DMA source offsets.  X instruction i covers LDS rows 16i..16i+15 = GEMM rows m0+16i+..
    const int lrow = lane >> 2, lpos = lane & 3;
    int xoff[XI], woff[WI], woffq[Q8 ? WI : 1];
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
        xoff[q] = el * 2 + ((lpos ^ swz<MF>(row)) << 4);
    }
#pragma unroll
    for (int q = 0; q < WI; ++q) {
        const int ins = wave + NW * q;               // instruction index over (3 taps x) BN/16
        const int tap = RING ? 0 : ins / (BN / 16), row = (ins % (BN / 16)) * 16 + lrow;
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        // tap kx adds kx*CK elements along K
        // (chunk-major planes [Ktot/32][N][32], agp_conv_desc::w_cm: a 64-byte K chunk is N * 64 bytes on)
        woff[q] = (p.w_cm ? n * 64 + tap * p.CK * 2 * p.N : (n * p.Ktot + tap * p.CK) * 2) + ((lpos ^ swz<MF>(row)) << 4);
        if (Q8) woffq[q] = n * p.Ktot + ((lpos ^ swz<MF>(row)) << 4);
    }
    const int wmul = __builtin_amdgcn_readfirstlane(p.w_cm ? p.N : 1);
    const __amdgpu_buffer_rsrc_t rx_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_cm ? p.w_cm : p.w_hi), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(XPL == 2 ? p.x_lo : p.x_hi), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_lo = Q8 ? __builtin_amdgcn_make_buffer_rsrc((void*)p.w_q8, 0, p.w_bytes / 2, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)(WPL == 2 ? (p.w_cm ? p.w_cm_lo : p.w_lo) : (p.w_cm ? p.w_cm : p.w_hi)), 0, p.w_bytes, 0x00020000);

    // ---- fragment read offsets: X rows shifted by kx, W rows per tap
    const int l31 = lane & 31, lh = lane >> 5;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int FT = 32 / MF;                  // MFMA tiles per 32 rows
    int xro[3][TM * FT], xsw[3][TM * FT], wro[TN * FT], wsw[TN * FT];
#pragma unroll
    for (int t = 0; t < TM * FT; ++t)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int r = wm * (TM * 32) + t * MF + (MF == 16 ? l15 : l31) + kx;
            xro[kx][t] = r * ROWB;
            xsw[kx][t] = swz<MF>(r);
        }
#pragma unroll
    for (int t = 0; t < TN * FT; ++t) {
        const int wrow = DIRECT ? ((l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1)) : (MF == 16 ? l15 : l31);
        const int wr = wn * (TN * 32) + t * MF + wrow;
        wro[t] = wr * ROWB;
        wsw[t] = swz<MF>(wr);
    }

    f32x16 acc[MF == 32 ? TN : 1][MF == 32 ? TM : 1];
    f32x4 acc4[MF == 16 ? TN * 2 : 1][MF == 16 ? TM * 2 : 1];    // 16x16 tiles: [channel tile][pixel tile]
#pragma unroll
    for (int a = 0; a < (MF == 32 ? TN : 1); ++a)
#pragma unroll
        for (int b = 0; b < (MF == 32 ? TM : 1); ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
    for (int a = 0; a < (MF == 16 ? TN * 2 : 1); ++a)
#pragma unroll
        for (int b = 0; b < (MF == 16 ? TM * 2 : 1); ++b) acc4[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Residual prefetch (single-plane fp16 maps): the epilogue's residual reads are issued before the
    // LAST macro-step's MFMAs, so that their HBM latency (1-2k cycles each, 8 of them in sequence
    // otherwise) is hidden behind compute instead of being paid per read-back iteration.
    constexpr int NIT = 32 / (64 / LPP);             // read-back iterations per 32-row pass
    constexpr bool RPF = PrecT<NPREC>::F16 && (XPL == 1) && ((DIRECT ? TM * TN * 2 : TM * NIT) <= 8);   // 32 prefetch registers at most
    u32x4 rpf[RPF ? (DIRECT ? TM * TN * 2 : TM * NIT) : 1];
    // DIRECT: the lane's pixel of tile row tm and its element offset in the output / residual planes
    size_t doff[DIRECT ? TM : 1];
    bool dvalid[DIRECT ? TM : 1];
    const bf16_t* const rhi = (const bf16_t*)p.r_hi;
    const bf16_t* const rlo = (const bf16_t*)p.r_lo;
    const int ch = lane % LPP;
    const int nglob = n0 + wn * (TN * 32) + ch * 8;
    const uint32_t wlast = p.d_wo.d - 1;
    auto out_offset = [&](int tm, int it, bool& valid) -> size_t {
        const int ml = it * (64 / LPP) + lane / LPP;
        const int m = m0 + wm * (TM * 32) + tm * 32 + ml;
        const uint32_t mm = (uint32_t)(m < p.M ? m : p.M - 1);
        const uint32_t img = fdiv(mm, p.d_howo);
        const uint32_t rem = mm - img * p.d_howo.d;
        const uint32_t y = fdiv(rem, p.d_wo);
        const uint32_t xq = rem - y * p.d_wo.d;
        valid = (m < p.M) && xq != 0 && xq != wlast;  // halo columns keep their zeros
        return (size_t)img * p.o_sn + (size_t)y * p.o_sh + (size_t)xq * p.o_sw + p.o_base + nglob;
    };

    if constexpr (DIRECT) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = m0 + wm * (TM * 32) + tm * 32 + l31;
            const uint32_t mm = (uint32_t)(m < p.M ? m : p.M - 1);
            const uint32_t img = fdiv(mm, p.d_howo);
            const uint32_t rem = mm - img * p.d_howo.d;
            const uint32_t y = fdiv(rem, p.d_wo);
            const uint32_t xq = rem - y * p.d_wo.d;
            dvalid[tm] = (m < p.M) && xq != 0 && xq != wlast;
            doff[tm] = (size_t)img * p.o_sn + (size_t)y * p.o_sh + (size_t)xq * p.o_sw + p.o_base + n0 + wn * (TN * 32) + 8 * lh;
        }
    }
    auto prefetch_residual = [&]() {
        if constexpr (DIRECT) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int j = 0; j < TN * 2; ++j)        // j = 2*tn + h: channels 16*j + 8*lh .. +7 of the wave's columns
                    rpf[tm * TN * 2 + j] = dvalid[tm] ? *(const u32x4*)(rhi + doff[tm] + 16 * j) : u32x4{0u, 0u, 0u, 0u};
        } else {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    bool valid;
                    const size_t off = out_offset(tm, it, valid);
                    rpf[tm * NIT + it] = valid ? *(const u32x4*)(rhi + off) : u32x4{0u, 0u, 0u, 0u};
                }
        }
    };
    if (DIRECT && tid < BN) {     // visible to every wave after the first barrier of the main loop
        const int n = n0 + tid < p.N ? n0 + tid : p.N - 1;
        ((float*)(smem + BN_TAB))[tid] = p.scale ? p.scale[n] : 1.f;
        ((float*)(smem + BN_TAB))[BN + tid] = p.shift ? p.shift[n] : 0.f;
    }
    AGP_STAMP();                                     // 0: prologue done
    const int cchunks = p.CK / 32;
    const int nsteps = 3 * cchunks;                  // (ky, cc) macro-steps
    int ky = 0, cc = 0;
    if constexpr (RING == 4) {
        // ---- one wave per SIMD (round 5): the three-product loop of the TRAINING convs with everything a step ahead.
        // The other forms lean on two or three workgroups per CU to hide their stage behind each other and read 0.67 LDS
        // fragments per MFMA at two planes per operand (LDS cycles ~ MFMA cycles: profiles/README.md, round 5).  Here a
        // workgroup is 256 x 128 (a wave = 64 pixels x 128 channels: 0.5 fragment reads per MFMA, half the W staging per
        // MFMA), alone on its CU with the whole register file, and the pipeline is spelled out:
        //   * a phase = one tap kx of a (ky, 32-channel chunk) macro-step = two K-steps of 24 MFMAs; the fragments of the
        //     NEXT K-step (across phase and macro-step borders) are read while this one's MFMAs run, the interleave is
        //     fixed with sched_group_barrier (one LDS read behind each of the first twelve MFMAs);
        //   * the barrier that publishes phase q + 1's operands sits in the MIDDLE of phase q; behind it a wave issues the
        //     W tap of phase q + 3 (ring of four slots) and, over three half-phases, its ten pieces of the next macro-step's
        //     X block (double buffered); counted vmcnt: 4 / 11 / 0 by tap;
        //   * dependent MFMAs (the three products of one accumulator) are eight MFMAs apart: product-major order, the same
        //     order per accumulator as mfma32<3> -- results are bit-identical to the other forms.
        // Loads past the last macro-step run out of the planes' ranges and return zeros nobody reads.
        static_assert(NPREC == 3 && MF == 32 && NW == 4 && !Q8 && !DIRECT, "RING = 4: split-bf16 operands, four waves");
        constexpr int WSLOT = WPL * W_TAP, XBUF = X_PLANE * XPL;
        const int tapb = __builtin_amdgcn_readfirstlane(p.CK * 2);
        int xo4[XI], xd4[XI];                            // this wave's X pieces, clamped to the block (a repeat lands on itself)
#pragma unroll
        for (int q = 0; q < XI; ++q) {
            int ins = wave + NW * q;
            ins = ins < XINS ? ins : XINS - 1;
            const int row = ins * 16 + lrow;
            const int m = m0 + row;
            const uint32_t img = fdiv((uint32_t)m, p.d_howo);
            const uint32_t rem = (uint32_t)m - img * p.d_howo.d;
            const uint32_t y = fdiv(rem, p.d_wo);
            const uint32_t xq = rem - y * p.d_wo.d;
            const int el = (int)img * p.x_sn + (int)y * p.x_sh + (int)xq * p.x_sw + p.x_base;
            xo4[q] = el * 2 + ((lpos ^ swz32(row)) << 4);
            xd4[q] = __builtin_amdgcn_readfirstlane(ins * 1024);
        }
        auto dma_x = [&](auto jc, int buf, int xs) {     // piece j = 2 q + plane of this wave's ten
            constexpr int j = decltype(jc)::value, q = j >> 1, pl = j & 1;
            char* dst = smem + buf * XBUF + pl * X_PLANE + xd4[q];
            if (pl) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_lo, LDS_PTR(dst), 16, xo4[q], xs, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_hi, LDS_PTR(dst), 16, xo4[q], xs, 0, 0);
        };
        auto dma_w = [&](int slot, int wbytes) {         // one tap: WI row blocks x two planes per wave
            const int so = __builtin_amdgcn_readfirstlane(wbytes * wmul);
#pragma unroll
            for (int q = 0; q < WI; ++q) {
                char* dst = ws_hi + slot * WSLOT + (wave + NW * q) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_hi, LDS_PTR(dst), 16, woff[q], so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_lo, LDS_PTR(dst + W_TAP), 16, woff[q], so, 0, 0);
            }
        };
        static_assert(XI * XPL == 10 && WI * WPL == 4, "the vmcnt counts below are written for ten X and four W pieces per wave");
        bf16x8 fxh[2][TM], fxl[2][TM], fwh[2][TN], fwl[2][TN];
        auto load_frags = [&](auto setc, const char* xb, const char* wb, auto kxc, auto ksc) {
            constexpr int S = decltype(setc)::value, kx = decltype(kxc)::value, ks = decltype(ksc)::value;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int xo = xro[kx][t] + (((2 * ks + lh) ^ xsw[kx][t]) << 4);
                fxh[S][t] = *(const bf16x8*)(xb + xo);
                fxl[S][t] = *(const bf16x8*)(xb + X_PLANE + xo);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int wo = wro[t] + (((2 * ks + lh) ^ wsw[t]) << 4);
                fwh[S][t] = *(const bf16x8*)(wb + wo);
                fwl[S][t] = *(const bf16x8*)(wb + W_TAP + wo);
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        // prologue: X(0), taps 0 .. 2 of macro-step 0
        static_for<10>([&](auto jc) { dma_x(jc, 0, 0); });
        dma_w(0, 0); dma_w(1, tapb); dma_w(2, 2 * tapb);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_frags(I0{}, smem, ws_hi, I0{}, I0{});
        int wslot = 0;                                   // ring slot of the current phase's tap (q & 3)
        for (int st = 0; st < nsteps; ++st) {
            int nky = ky, ncc = cc + 1;
            if (ncc == cchunks) { ncc = 0; ++nky; }
            const int wnext = (nky * 3 * p.CK + ncc * 32) * 2;
            const int xsn = __builtin_amdgcn_readfirstlane((nky * p.x_sh + ncc * 32) * 2);
            const char* xb = smem + (st & 1) * XBUF;
            const char* xbn = smem + ((st + 1) & 1) * XBUF;
            static_for<3>([&](auto kxc) {
                constexpr int kx = decltype(kxc)::value;
                const char* wb = ws_hi + wslot * WSLOT;
                const int ns = (wslot + 1) & 3;
                const char* wbn = ws_hi + ns * WSLOT;
                static_for<2>([&](auto ksc) {
                    constexpr int ks = decltype(ksc)::value;
                    constexpr int cur = ks;              // fragment set of this K-step (two K-steps per phase: the sets alternate)
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ks == 1) {
                        // phase q + 1's operands are published here
                        if constexpr (kx == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else if constexpr (kx == 1) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        dma_w((wslot + 3) & 3, wnext + kx * tapb);        // tap kx of the next macro-step = phase q + 3
                        if constexpr (kx == 0) { dma_x(I0{}, (st + 1) & 1, xsn); dma_x(I1{}, (st + 1) & 1, xsn); dma_x(I2{}, (st + 1) & 1, xsn);
                                                 dma_x(std::integral_constant<int, 3>{}, (st + 1) & 1, xsn); }
                        if constexpr (kx == 1) { dma_x(std::integral_constant<int, 7>{}, (st + 1) & 1, xsn); dma_x(std::integral_constant<int, 8>{}, (st + 1) & 1, xsn);
                                                 dma_x(std::integral_constant<int, 9>{}, (st + 1) & 1, xsn); }
                    } else {
                        if constexpr (kx == 1) { dma_x(std::integral_constant<int, 4>{}, (st + 1) & 1, xsn); dma_x(std::integral_constant<int, 5>{}, (st + 1) & 1, xsn);
                                                 dma_x(std::integral_constant<int, 6>{}, (st + 1) & 1, xsn); }
                    }
                    // the next K-step's fragments: this phase's second K-step, or the first of the next phase (next tap of this
                    // macro-step, or tap 0 of the next one from the other X buffer)
                    if constexpr (ks == 0) load_frags(I1{}, xb, wb, kxc, I1{});
                    else if constexpr (kx < 2) load_frags(I0{}, xb, wbn, std::integral_constant<int, kx + 1>{}, I0{});
                    else load_frags(I0{}, xbn, wbn, I0{}, I0{});
                    // 24 MFMAs, product-major
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwl[cur][tn], fxh[cur][tm], acc[tn][tm], 0, 0, 0);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwh[cur][tn], fxl[cur][tm], acc[tn][tm], 0, 0, 0);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwh[cur][tn], fxh[cur][tm], acc[tn][tm], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 3 * TM * TN; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i < 2 * (TM + TN)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (i >= 12 && i < 20) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    }
                });
                wslot = ns;
            });
            ky = nky; cc = ncc;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the loads issued past the last macro-step have landed (zeros)
        __syncthreads();                                     // before the epilogue reuses the stage
    } else
    if constexpr (RING >= 2) {
        const int tapb = __builtin_amdgcn_readfirstlane(p.CK * 2);   // bytes between consecutive kx taps
        auto load_x = [&](int buf, int ky_, int cc_) {
            const int xs = __builtin_amdgcn_readfirstlane((ky_ * p.x_sh + cc_ * 32) * 2);
            char* base = smem + buf * (X_PLANE * XPL);
#pragma unroll
            for (int q = 0; q < XI; ++q) {
                const int ins = wave + NW * q;
                if (ins < XINS) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_hi, LDS_PTR(base + ins * 1024), 16, xoff[q], xs, 0, 0);
                    if (XPL == 2)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_lo, LDS_PTR(base + X_PLANE + ins * 1024), 16, xoff[q], xs, 0, 0);
                }
            }
        };
        // Q8: a W ring slot holds the hi plane only; the e4m3 lo plane has its own 2-slot ring, one slot per PAIR of phases
        constexpr int WSLOT = Q8 ? W_TAP : WPL * W_TAP;
        char* const wq_base = ws_hi + 2 * WSLOT;
        auto load_w = [&](int slot, int wbytes) {
            const int so = __builtin_amdgcn_readfirstlane(wbytes * wmul);
#pragma unroll
            for (int q = 0; q < WI; ++q) {
                const int ins = wave + NW * q;
                if (ins < WINS) {
                    char* dst = ws_hi + slot * WSLOT + ins * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_hi, LDS_PTR(dst), 16, woff[q], so, 0, 0);
                    if (WPL == 2 && !Q8)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_lo, LDS_PTR(dst + W_TAP), 16, woff[q], so, 0, 0);
                }
            }
        };
        // e4m3 lo plane: [n][pair][64 B]; pair P = phases 2P, 2P+1 in execution order (igemm.hip, agp_conv_w_q8_layout)
        auto load_wq = [&](int pair) {
            const int so = __builtin_amdgcn_readfirstlane(pair * 64);
#pragma unroll
            for (int q = 0; q < WI; ++q) {
                const int ins = wave + NW * q;
                if (ins < WINS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_lo, LDS_PTR(wq_base + (pair & 1) * W_TAP + ins * 1024), 16, woffq[q], so, 0, 0);
            }
        };
        const int nphases = 3 * nsteps;
        const int q8_scale = 127 - p.w_q8_exp;       // E8M0 block scale undoing the 2^exp applied to the stored lo plane
        int x8[Q8 ? TM : 1][8];                      // e4m3 activations of the current phase pair (lane's K subset)
        load_x(0, 0, 0);
        load_w(0, 0);
        if (Q8) load_wq(0);
        for (int st = 0; st < nsteps; ++st) {
            int nky = ky, ncc = cc + 1;
            if (ncc == cchunks) { ncc = 0; ++nky; }
            const int wcur = (ky * 3 * p.CK + cc * 32) * 2, wnext = (nky * 3 * p.CK + ncc * 32) * 2;
            const char* xb_hi = smem + (st & 1) * (X_PLANE * XPL);
            const char* xb_lo = xb_hi + X_PLANE;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                __syncthreads();     // the loads issued one phase ago have landed; the slot / buffer written next is free
                // timing experiments: dbg 1024 = no W staging after the first tap, 2048 = no X staging after the first block
                const int ph = st * 3 + kx;          // phase index; Q8 pairs phases (2P, 2P+1)
                auto issue_loads = [&]() {
                    if (Q8 && !(ph & 1) && ph + 2 < nphases && !(p.dbg & 1024)) load_wq((ph >> 1) + 1);
                    if (kx < 2) {
                        if (!(p.dbg & 1024)) load_w((st + kx + 1) & 1, wcur + (kx + 1) * tapb);
                    } else if (st + 1 < nsteps) {
                        if (!(p.dbg & 2048)) load_x((st + 1) & 1, nky, ncc);
                        if (!(p.dbg & 1024)) load_w((st + 3) & 1, wnext);
                    }
                };
                if (RING == 2) issue_loads();
                if (RPF && rhi && st == nsteps - 1 && kx == 0) prefetch_residual();
                const char* wbase_hi = ws_hi + ((st + kx) & 1) * WSLOT;
                const char* wbase_lo = wbase_hi + W_TAP;
                if constexpr (MF == 16) {
                    // offsets: the swizzle term (row >> 1) & 2 does not depend on the 16-row tile index
                    bf16x8 xh[TM * 2], xl[TM * 2];
                    const int xq = xro[kx][0] + ((lq ^ xsw[kx][0]) << 4), wq = wro[0] + ((lq ^ wsw[0]) << 4);
#pragma unroll
                    for (int t = 0; t < TM * 2; ++t) {
                        xh[t] = *(const bf16x8*)(xb_hi + xq + t * 16 * ROWB);
                        if (XPL == 2) xl[t] = *(const bf16x8*)(xb_lo + xq + t * 16 * ROWB);
                    }
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        bf16x8 wh[TN], wl[TN];
#pragma unroll
                        for (int t = 0; t < TN; ++t) {
                            wh[t] = *(const bf16x8*)(wbase_hi + wq + (half * TN + t) * 16 * ROWB);
                            if (WPL == 2) wl[t] = *(const bf16x8*)(wbase_lo + wq + (half * TN + t) * 16 * ROWB);
                        }
                        if (RING == 3 && half == 0) issue_loads();
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                            for (int tm = 0; tm < TM * 2; ++tm)
                                mfma16<NPREC>(acc4[half * TN + tn][tm], wh[tn], wl[tn], xh[tm], xl[tm]);
                    }
                } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 xh[TM], xl[TM], wh[TN], wl[TN];
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        const int xo = xro[kx][t] + (((2 * ks + lh) ^ xsw[kx][t]) << 4);
                        xh[t] = *(const bf16x8*)(xb_hi + xo);
                        if (XPL == 2) xl[t] = *(const bf16x8*)(xb_lo + xo);
                    }
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const int wo = wro[t] + (((2 * ks + lh) ^ wsw[t]) << 4);
                        wh[t] = *(const bf16x8*)(wbase_hi + wo);
                        if (WPL == 2 && !Q8) wl[t] = *(const bf16x8*)(wbase_lo + wo);
                    }
                    // RING = 3: the DMA issue sits behind the first fragment reads, off the path to the first MFMA
                    if (RING == 3 && ks == 0) issue_loads();
                    if constexpr (Q8) {
                        // the lane's 8 fp16 activations of this K-step -> 8 e4m3 bytes (2 dwords) of its K subset of the pair
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            const f16x8 xv = __builtin_bit_cast(f16x8, xh[tm]);
                            int c[2];
                            {
                                s16x2 o = {0, 0};
                                o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, __builtin_shufflevector(xv, xv, 0, 1), 1.f, false);
                                o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, __builtin_shufflevector(xv, xv, 2, 3), 1.f, true);
                                c[0] = __builtin_bit_cast(int, o);
                                s16x2 o2 = {0, 0};
                                o2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o2, __builtin_shufflevector(xv, xv, 4, 5), 1.f, false);
                                o2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o2, __builtin_shufflevector(xv, xv, 6, 7), 1.f, true);
                                c[1] = __builtin_bit_cast(int, o2);
                            }
                            if (ph & 1) { x8[tm][4 + 2 * ks] = c[0]; x8[tm][5 + 2 * ks] = c[1]; }
                            else        { x8[tm][2 * ks] = c[0];     x8[tm][2 * ks + 1] = c[1]; }
                        }
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) mfma32<4>(acc[tn][tm], wh[tn], wl[tn], xh[tm], xl[tm]);
                    } else {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) mfma32<NPREC>(acc[tn][tm], wh[tn], wl[tn], xh[tm], xl[tm]);
                    }
                }
                if constexpr (Q8) {
                    if (ph & 1) {       // second phase of the pair: the lo product of both taps in one K = 64 block-scaled MFMA per tile
                        const char* wqb = wq_base + ((ph >> 1) & 1) * W_TAP;
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) {
                            const u32x4 q0 = *(const u32x4*)(wqb + wro[tn] + (((2 * lh) ^ wsw[tn]) << 4));
                            const u32x4 q1 = *(const u32x4*)(wqb + wro[tn] + (((2 * lh + 1) ^ wsw[tn]) << 4));
                            const i32x8 wq = {(int)q0[0], (int)q0[1], (int)q0[2], (int)q0[3], (int)q1[0], (int)q1[1], (int)q1[2], (int)q1[3]};
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) {
                                const i32x8 xq = {x8[tm][0], x8[tm][1], x8[tm][2], x8[tm][3], x8[tm][4], x8[tm][5], x8[tm][6], x8[tm][7]};
                                acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wq, xq, acc[tn][tm], 0, 0, 0, q8_scale, 0, 127);
                            }
                        }
                    }
                }
                }
            }
            ky = nky; cc = ncc;
        }
    } else {
        for (int st = 0; st < nsteps; ++st) {
            // stage X(ky,cc) and W(ky, kx=0..2, cc)
            const int xs = __builtin_amdgcn_readfirstlane((ky * p.x_sh + cc * 32) * 2);
            const int ws = __builtin_amdgcn_readfirstlane((ky * 3 * p.CK + cc * 32) * 2 * wmul);
            if (st) __syncthreads();                     // previous macro-step's fragment reads are done
            const bool noload = (p.dbg & 256) && st > 0;   // timing experiment: stage only the first macro-step
    #pragma unroll
            for (int q = 0; q < XI; ++q) {
                const int ins = wave + NW * q;
                if (ins < XINS && !noload) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_hi, LDS_PTR(xs_hi + ins * 1024), 16, xoff[q], xs, 0, 0);
                    if (XPL == 2)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_lo, LDS_PTR(xs_lo + ins * 1024), 16, xoff[q], xs, 0, 0);
                }
            }
            // W: all three taps at once (RING = 0), or taps 0 and 1 into ring slots 0 and 1
            auto load_w = [&](int slot, int tapoff) {
    #pragma unroll
                for (int q = 0; q < WI; ++q) {
                    const int ins = wave + NW * q;
                    if (ins < WINS && !noload) {
                        char* dst = ws_hi + slot * (WPL * W_TAP) + ins * 1024;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_hi, LDS_PTR(dst), 16, woff[q], ws + tapoff * wmul, 0, 0);
                        if (WPL == 2)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_lo, LDS_PTR(dst + (RING ? W_TAP : W_PLANE)), 16, woff[q], ws + tapoff * wmul, 0, 0);
                    }
                }
            };
            const int tapb = __builtin_amdgcn_readfirstlane(p.CK * 2);   // bytes between consecutive kx taps
            load_w(0, 0);
            if (RING) load_w(1, tapb);
            // macro-step order: the three row blocks (ky) of ONE 32-channel chunk back to back, then the next chunk (round 6; before:
            // ky outermost).  A tile's three ky blocks overlap in 3/4 of their rows, and 96 tiles in flight per XCD hold more than
            // its 4 MB L2: with the re-read one macro-step away instead of `cchunks` the training step runs 16.06 / 16.30 ->
            // 15.95 / 15.99 ms (A/B of two libraries on one box); the fp32 accumulation order of these kernels changes with it
            if (++ky == 3) { ky = 0; ++cc; }
            AGP_STAMP();                                 // loads issued
            __syncthreads();                             // vmcnt(0): the stage has landed for every wave
            AGP_STAMP();                                 // stage landed
            if (RPF && rhi && st == nsteps - 1) prefetch_residual();
    #pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (RING && kx == 1) {
                    __syncthreads();                     // every wave is done with ring slot 0 (tap 0)
                    load_w(0, 2 * tapb);                 // tap 2 streams in while tap 1 computes
                }
                if (RING && kx == 2) __syncthreads();    // tap 2 has landed
                const char* wbase_hi = RING ? ws_hi + (kx & 1) * (WPL * W_TAP) : ws_hi + kx * W_TAP;
                const char* wbase_lo = wbase_hi + (RING ? W_TAP : W_PLANE);
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 xh[TM], xl[TM], wh[TN], wl[TN];
    #pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        const int xo = xro[kx][t] + (((2 * ks + lh) ^ xsw[kx][t]) << 4);
                        xh[t] = *(const bf16x8*)(xs_hi + xo);
                        if (XPL == 2) xl[t] = *(const bf16x8*)(xs_lo + xo);
                    }
    #pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const int wo = wro[t] + (((2 * ks + lh) ^ wsw[t]) << 4);
                        wh[t] = *(const bf16x8*)(wbase_hi + wo);
                        if (WPL == 2) wl[t] = *(const bf16x8*)(wbase_lo + wo);
                    }
    #pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
    #pragma unroll
                        for (int tm = 0; tm < TM; ++tm) mfma32<NPREC>(acc[tn][tm], wh[tn], wl[tn], xh[tm], xl[tm]);
                }
            }
            AGP_STAMP();                                 // compute issued
        }
    }

    AGP_STAMP();                                     // K loop done
    if (p.dbg & 128) {                               // timing experiment: no epilogue at all
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < (MF == 32 ? TN : 1); ++a)
#pragma unroll
            for (int b = 0; b < (MF == 32 ? TM : 1); ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[a][b][r];
#pragma unroll
        for (int a = 0; a < (MF == 16 ? TN * 2 : 1); ++a)
#pragma unroll
            for (int b = 0; b < (MF == 16 ? TM * 2 : 1); ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) t += acc4[a][b][r];
        if (t == 1.2345e30f) ((float*)p.o_hi)[0] = t;
        return;
    }
    if constexpr (DIRECT) {
        // ---- direct epilogue: registers -> scale/shift (LDS table) -> + residual -> ReLU -> fp16 -> 16-byte stores
        const float* tab = (const float*)(smem + BN_TAB) + wn * (TN * 32) + 8 * lh;
        bf16_t* const ohi = (bf16_t*)p.o_hi;
        // tile row outermost: the 2 * TN stores of a lane pair complete its pixel's BN-channel segment back to back
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            if (!dvalid[tm]) continue;
#pragma unroll
            for (int j = 0; j < TN * 2; ++j) {           // j = 2*tn + h
                const f32x4 s0 = *(const f32x4*)(tab + 16 * j), s1 = *(const f32x4*)(tab + 16 * j + 4);
                const f32x4 t0 = *(const f32x4*)(tab + BN + 16 * j), t1 = *(const f32x4*)(tab + BN + 16 * j + 4);
                const float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                const float sh[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[j >> 1][tm][8 * (j & 1) + e] * sc[e] + sh[e];
                if (rhi) {
                    float r[8];
                    if (RPF) unpack8_h(rpf[tm * TN * 2 + j], r);
                    else unpack8_h(*(const u32x4*)(rhi + doff[tm] + 16 * j), r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *(u32x4*)(ohi + doff[tm] + 16 * j) = pack8_h(v);
            }
        }
        return;
    }
    // ---- epilogue (as igemm.hip), halo columns masked; one pass per 32-pixel tile row
    __syncthreads();
    char* er = smem + wave * (32 * EROWB);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = p.scale ? p.scale[nglob + e] : 1.f;
        sh[e] = p.shift ? p.shift[nglob + e] : 0.f;
    }
    bf16_t* ohi = (bf16_t*)p.o_hi;
    bf16_t* olo = (bf16_t*)p.o_lo;
    // optional per-tile channel statistics of the stored values (train-mode BatchNorm: sum and sum of squares)
    const bool stats = p.stat_partial != nullptr;
    // backward-statistics mode: the stored values are a gradient g at the output y = relu?(BN(z) + r) of an earlier unit; the sums
    // become (sum g*[y>0], sum g*[y>0]*zhat), zhat = (z - mean)*rstd -- the first stage of that unit's BatchNorm backward, which
    // then needs no pass of its own over g, z and y (same map geometry as this conv's output)
    const bf16_t* bz_hi = (const bf16_t*)p.bs_z_hi;
    const bf16_t* bz_lo = (const bf16_t*)p.bs_z_lo;
    const bf16_t* by_hi = (const bf16_t*)p.bs_y_hi;
    const bool bstats = stats && bz_hi != nullptr;
    float st1[8], st2[8], bmu[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        st1[e] = 0.f; st2[e] = 0.f;
        bmu[e] = bstats ? p.bs_mean[nglob + e] : 0.f;
    }
    // backward-statistics mode: the (residual, z, y) lines of tile-row item `it` as raw registers, fetched ONE ITEM AHEAD of their
    // use -- written in this order because the compiler may not move a load above the previous item's stores (the planes could
    // alias), which made every item two serial memory round trips (a dgrad launch +50 %)
    struct BsRaw { u32x4 rh, rl, zh, zl, yh; size_t off; bool valid; };
    auto bs_fetch = [&](int tm, int it) -> BsRaw {
        BsRaw r;
        r.off = out_offset(tm, it, r.valid);            // always inside the map (rows clamped, halo columns exist)
        r.rh = r.rl = r.yh = u32x4{0u, 0u, 0u, 0u};
        if (rhi) { r.rh = *(const u32x4*)(rhi + r.off); if (rlo) r.rl = *(const u32x4*)(rlo + r.off); }
        r.zh = *(const u32x4*)(bz_hi + r.off);
        r.zl = bz_lo ? *(const u32x4*)(bz_lo + r.off) : u32x4{0u, 0u, 0u, 0u};      // (z: a bf16 pair, or ONE fp16 plane)
        if (by_hi) r.yh = *(const u32x4*)(by_hi + r.off);
        return r;
    };
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        // (the staging rows are private to the wave: LDS ops of one wave execute in order, no barrier)
        if constexpr (MF == 16) {
            // D tile (channel tile a, pixel tile b): lane holds pixel l15, channels 4*lq .. 4*lq+3
#pragma unroll
            for (int a = 0; a < TN * 2; ++a)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *(f32x4*)(er + (h * 16 + l15) * EROWB + (a * 16 + 4 * lq) * 4) = acc4[a][tm * 2 + h];
        } else {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
                *(f32x4*)(er + l31 * EROWB + (tn * 32 + 8 * q + 4 * lh) * 4) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (bstats) {
            // data-gradient conv feeding a BatchNorm backward: no scale / shift / ReLU (agp_conv_desc::bstat_*)
            BsRaw cur = bs_fetch(tm, 0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                BsRaw nxt = cur;
                if (it + 1 < NIT) nxt = bs_fetch(tm, it + 1);
                if (cur.valid) {
                    const int ml = it * (64 / LPP) + lane / LPP;
                    const f32x4 a = *(const f32x4*)(er + ml * EROWB + ch * 32);
                    const f32x4 b = *(const f32x4*)(er + ml * EROWB + ch * 32 + 16);
                    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                    if (rhi) {
                        float r[8], l[8];
                        if (rlo) {
                            unpack8(cur.rh, r); unpack8(cur.rl, l);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += r[e] + l[e];
                        } else {
                            unpack8_h(cur.rh, r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += r[e];
                        }
                    }
                    map_store8(ohi, olo, cur.off, v);
                    float zz[8], zl[8];
                    if (bz_lo) { unpack8(cur.zh, zz); unpack8(cur.zl, zl); }
                    else {
                        unpack8_h(cur.zh, zz);
#pragma unroll
                        for (int e = 0; e < 8; ++e) zl[e] = 0.f;
                    }
                    const unsigned pm = by_hi ? pos_mask8_raw(cur.yh) : 0xffu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gm = ((pm >> e) & 1u) ? v[e] : 0.f;
                        st1[e] += gm;
                        st2[e] += gm * ((zz[e] + zl[e]) - bmu[e]);
                    }
                }
                cur = nxt;
            }
            continue;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int ml = it * (64 / LPP) + lane / LPP;
            bool valid;
            const size_t off = out_offset(tm, it, valid);
            if (!valid) continue;
            const f32x4 a = *(const f32x4*)(er + ml * EROWB + ch * 32);
            const f32x4 b = *(const f32x4*)(er + ml * EROWB + ch * 32 + 16);
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (rhi) {
                float r[8];
                if (RPF) unpack8_h(rpf[tm * NIT + it], r);
                else map_load8(rhi, rlo, off, r);
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
    if (bstats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) st2[e] *= p.bs_rstd[nglob + e];
    }
    if (stats) {
        // lanes sharing a channel group (lane % LPP) -> wave totals; waves sharing the columns (same wn) -> tile totals,
        // in a fixed order: partial[tile][2][N] as agp_bn_stats' own first stage writes it
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = LPP; o < 64; o <<= 1) {
                st1[e] += __shfl_xor(st1[e], o, 64);
                st2[e] += __shfl_xor(st2[e], o, 64);
            }
        }
        float* red = (float*)(smem + NW * 32 * EROWB);           // [wave][TN*32 channels][2], beyond every wave's staging rows
        if (lane < LPP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * (TN * 32) + lane * 8 + e) * 2] = st1[e];
                red[(wave * (TN * 32) + lane * 8 + e) * 2 + 1] = st2[e];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            const int wn_c = tid / (TN * 32), cc = tid % (TN * 32);
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                a += red[((wn_c * WM + w) * (TN * 32) + cc) * 2];
                b += red[((wn_c * WM + w) * (TN * 32) + cc) * 2 + 1];
            }
            p.stat_partial[(size_t)mt * 2 * p.N + n0 + tid] = a;
            p.stat_partial[(size_t)mt * 2 * p.N + p.N + n0 + tid] = b;
        }
    }
#if AGP_CENSUS
    if ((p.dbg & 0x1000000) && tid == 0 && p.gmin) {
        // census record: {hw_id, xcc_id, start, end} (100 MHz ticks) per workgroup
        __builtin_amdgcn_s_waitcnt(0);
        stamp[nstamp < 56 ? nstamp : 55] = __builtin_amdgcn_s_memtime();   // epilogue done
        unsigned long long* rec = (unsigned long long*)p.gmin + (size_t)bid * 64;
        rec[0] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        rec[1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        rec[2] = t_begin;
        rec[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int BN, int WM, int WN, int NPREC, int RING, int MF = 32, bool LDS_EPI = false, bool Q8 = false>
int launch_kxr(IgemmParams& p, hipStream_t s) {
    constexpr int lds = kxr_lds_bytes<BM, BN, WM, WN, NPREC, RING>();
    static_assert(lds <= (kxr_min_blocks<BM, BN, WM, WN, RING, NPREC>() == 3 ? 53 : (kxr_min_blocks<BM, BN, WM, WN, RING, NPREC>() == 2 ? 80 : 160)) * 1024,
                  "LDS budget of the intended workgroups per CU");
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)igemm_kxr_kernel<BM, BN, WM, WN, NPREC, RING, MF, LDS_EPI, Q8>, lds, attr_done)) return AGP_E_LAUNCH;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.N + BN - 1) / BN;
    p.mt_chunk = (p.MT + 7) / 8;
    AGP_LAUNCH((igemm_kxr_kernel<BM, BN, WM, WN, NPREC, RING, MF, LDS_EPI, Q8>), dim3(p.mt_chunk * 8 * p.NT), dim3(WM * WN * 64), lds, s, p);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

}  // namespace agp_igemm

int agp_internal_conv_kxr2(agp_igemm::IgemmParams* ps, int n, hipStream_t s);

// Rewrites the generic geometry of `p` for the padded-width raster of the 3x3 stride-1 kernels.
void agp_internal_conv_kxr_geometry(agp_igemm::IgemmParams& p, const agp_conv_desc* d) {
    using namespace agp_igemm;
    const int hp = d->hin + 2, wp = d->win + 2;
    p.img_rows = d->hin * wp;
    // conv-epilogue pooling (igemm_kxr2): every image gets a multiple of 64 raster rows, so that a 64-row wave block lies
    // in one image at an image-relative position; the extra rows are computed and dropped (< 1 % at the bench's sizes)
    const int rp = p.pool_partial ? (p.img_rows + 63) / 64 * 64 : p.img_rows;
    p.M = d->n * rp;                                   // rows: (img, y, x' in [0, wp))
    p.d_howo = make_fastdiv((uint32_t)rp);
    p.d_wo = make_fastdiv((uint32_t)wp);
    p.x_sw = d->cin; p.x_sh = wp * d->cin; p.x_sn = hp * wp * d->cin;
    p.x_base = -d->cin;                                 // pixel (y + ky, x' + kx - 1)
    p.o_sw = d->cout; p.o_sh = wp * d->cout; p.o_sn = hp * wp * d->cout;
    p.o_base = wp * d->cout;                            // padded row y + 1, padded column x'
}

// fp16 maps with one fp16 product run on the round-2 kernel (igemm_kxr2.hip).
bool agp_internal_use_kxr2(const agp_conv_desc* d) {
    // (the epilogue addresses the output plane with 32-bit element offsets)
    return AGP_TUNE("KXR2", 1) && d->prec == AGP_PREC_F16 && !d->stat_partial &&
           (int64_t)d->n * (d->hout + 2) * (d->wout + 2) * d->cout < (1ll << 31);
}

// 3x3 / stride 1 / pad 1 convs on 1-pixel-halo planes.  `p` arrives with the generic geometry.
int agp_internal_conv_kxr(agp_igemm::IgemmParams& p, const agp_conv_desc* d, hipStream_t s) {
    using namespace agp_igemm;
    agp_internal_conv_kxr_geometry(p, d);
    if (agp_internal_use_kxr2(d)) return agp_internal_conv_kxr2(&p, 1, s);
    const bool wide = (p.N % 128 == 0);
    const int var = AGP_TUNE("KXR_VARIANT", 0);
    (void)var;
    if (d->prec == AGP_PREC_BF16X3 && d->hi_only) {
        // one bf16 product on the hi planes, the split-pair epilogue (residual, statistics, out_hi / out_lo) of the three-product
        // form; 256-row tiles at every width (agp_conv2d_stat_tiles mirrors it)
        // wide: a wave owns 64 rows x 128 columns -- 16 MFMAs per phase and barrier, the inference kernel's shape (8 on 128 x 128
        // tiles of four waves: 0.20 MFMA-busy at 2.2 TB/s, bound by neither)
        return wide ? launch_kxr<256, 128, 4, 1, 1, 3, 32, true>(p, s) : launch_kxr<256, 64, 4, 1, 1, 3, 32, true>(p, s);
    }
    if (d->prec == AGP_PREC_BF16X3) {
#if defined(AGP_TUNING)
        if (var == 1) return wide ? launch_kxr<128, 128, 2, 2, 3, 0>(p, s) : launch_kxr<256, 64, 4, 1, 3, 0>(p, s);
        if (var == 7 && wide) return launch_kxr<256, 128, 4, 1, 3, 4>(p, s);
        if (var == 2 && wide) return launch_kxr<128, 128, 2, 2, 3, 2>(p, s);
        if (var == 3 && wide) return launch_kxr<128, 128, 2, 2, 3, 3>(p, s);
        if (var == 2 && !wide) return launch_kxr<128, 64, 2, 1, 3, 2>(p, s);
        if (var == 3 && !wide) return launch_kxr<128, 64, 2, 1, 3, 3>(p, s);
#endif
        // 256-channel layers (K = 2304: 24 macro-steps per tile, few tiles): the phase-pipelined loop (X double-buffered, every load
        // a phase ahead; two workgroups per CU) -- 94 -> 82 us on the panorama maps, 181 -> 175 on the tile maps of the training
        // step; the 128-channel layers are even (74 / 76, 160 / 155) and the 64-channel ones lose (82 -> 93), tools/conv_bench.py
        if (wide && p.N % 256 == 0) return launch_kxr<128, 128, 2, 2, 3, 3>(p, s);
        return wide ? launch_kxr<128, 128, 2, 2, 3, 1>(p, s) : launch_kxr<256, 64, 4, 1, 3, 1>(p, s);
    }
    if (d->prec == AGP_PREC_F16W2) {
#if defined(AGP_TUNING)
        if (var == 1) return wide ? launch_kxr<128, 128, 2, 2, 2, 0>(p, s) : launch_kxr<256, 64, 4, 1, 2, 0>(p, s);
        if (var == 6) return wide ? launch_kxr<128, 128, 2, 2, 2, 1>(p, s) : launch_kxr<256, 64, 4, 1, 2, 1>(p, s);
        if (var == 12) return wide ? launch_kxr<128, 128, 2, 2, 2, 2>(p, s) : launch_kxr<256, 64, 4, 1, 2, 2>(p, s);
        if (var == 13) return launch_kxr<256, 64, 4, 1, 2, 3, 16>(p, s);
        if (var == 15) return launch_kxr<256, 64, 4, 1, 2, 3, 32, true>(p, s);
        if (var == 12) return launch_kxr<256, 64, 4, 1, 2, 3>(p, s);
#endif
        if (p.w_q8 && p.CK % 64 == 0) return launch_kxr<256, 64, 4, 1, 2, 3, 32, false, true>(p, s);
        return launch_kxr<256, 64, 4, 1, 2, 3>(p, s);
    }
    if (d->prec == AGP_PREC_F16) {
        // (reached only where igemm_kxr2 does not take the conv: output planes past its 32-bit element offsets, stat_partial)
#if defined(AGP_TUNING)
        if (var == 1) return wide ? launch_kxr<128, 128, 2, 2, 4, 0>(p, s) : launch_kxr<256, 64, 4, 1, 4, 0>(p, s);
        if (var == 12) return wide ? launch_kxr<128, 128, 2, 2, 4, 2>(p, s) : launch_kxr<256, 64, 4, 1, 4, 2>(p, s);
#endif
        return launch_kxr<256, 64, 4, 1, 4, 3>(p, s);
    }
    return AGP_E_BADARG;
}
