// igemm_kxrw.hip -- the WIDE form of the inference hot kernel (igemm_kxr2.hip): 3x3 / stride 1 / pad 1 convolution on fp16 maps
// with one fp16 MFMA product, for layers with cout % 128 == 0 (ResNet layers 2 and 3, the stage-2 BasicBlock): 256 x 128 tiles,
// four waves of 64 pixels x 128 channels.
//
// Same implicit GEMM over the padded-width raster, same phase pipeline (W taps through a 3-slot ring, X double-buffered, LDS-DMA
// issued two / three phases ahead, raw s_barrier + counted s_waitcnt vmcnt(N)) and the same line-layout epilogue as igemm_kxr2;
// what changes is the arithmetic per staged byte and per barrier:
//   * the X row block of a (ky, channel chunk) macro-step is staged ONCE for 128 output channels (kxr2: once per 64);
//   * a phase is 16 MFMAs per wave (kxr2: 8) on 12 LDS fragment reads (kxr2: 8): 0.75 instead of 1.0 ds_read_b128 per MFMA;
//   * 60 KB of LDS and ~200 VGPRs: two workgroups per CU = two waves per SIMD, each with 512 MFMA cycles per phase.
// (The stride-2 entry kernel gained 18 % from the same change, igemm_s2.hip; round 3.)  The residual is not prefetched during the
// last macro-step (the accumulators take 128 VGPRs): tile row 0's residual loads are issued before the epilogue barrier, tile
// row 1's before tile row 0's stores.  POOL as in igemm_kxr2 (conv-epilogue pooling over image-aligned 64-row blocks).
//
// vmcnt bookkeeping (per wave, issue order; NX = 5 X pieces, a W piece = 128 rows = NWP = 2 instructions per wave):
//     prologue          : X(0)[NX]  W(0,0)[2]  W(0,1)[2]
//     phase (st,0)      : W(st,2)[2]   X(st+1)[NX]
//     phase (st,1)      : W(st+1,0)[2]
//     phase (st,2)      : W(st+1,1)[2]
//   opens (st,1): W(st,1);            younger: W(st,2) X(st+1)        -> vmcnt(NX+2)
//   opens (st,2): W(st,2);            younger: X(st+1) W(st+1,0)      -> vmcnt(NX+2)
//   opens (st+1,0): W(st+1,0) X(st+1); younger: W(st+1,1)             -> vmcnt(2)
//   last macro-step L: opens (L,1): younger W(L,2) -> vmcnt(2); opens (L,2): nothing younger -> vmcnt(0).

#include <type_traits>

#include "igemm_params.hpp"

namespace agp_igemm {

__device__ __forceinline__ int kw_swz(int row) { return (row >> 2) & 3; }   // XOR-swizzle of a row's four 16-byte chunks

constexpr int KXRW_MAXP = 4;
struct KxrwGroup {
    IgemmParams p[KXRW_MAXP];
    int mt_end[KXRW_MAXP];
    int nprob, MT, NT, mt_chunk;
    // round 6, the launch's LAST round of workgroups as HALF tiles (128 rows): row tiles [MT_full, MT) of the global sequence are
    // not in the XCD-ordered part of the grid but follow it as 2 (MT - MT_full) NT blocks from block `half_bid0` on
    int MT_full, half_bid0;
};

template <int N> __device__ __forceinline__ void kw_wait() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    else static_assert(N < 0, "add the count");
}

// Tile shapes (a wave = TM_ x TN_ MFMA tiles of 32 x 32, four waves stacked along the rows):
//   TM_ = 2, TN_ = 4: 256 rows x 128 channels -- the WIDE form, cout % 128 == 0;
//   TM_ = 4, TN_ = 2: 512 rows x  64 channels -- the TALL form for cout = 64 (ResNet layer 1): the same 16 MFMAs per phase on 12
//     fragment reads, and HALF as many tiles: layer 1's K is 576 = 18 phases, so the fixed cost of a tile (first stage in
//     flight, epilogue round trips) is a third of a 256 x 64 tile's time (timed: K 576 -> 1152 on the same map raises the kernel
//     from 650 to 800 TFLOP/s).
constexpr int KW_ROWB = 64;
template <int TM_, int TN_> struct KwShape {
    static constexpr int BM = 128 * TM_, BN = 32 * TN_, BMX = BM + 16;
    static constexpr int XBUF = BMX * KW_ROWB, WTAP = BN * KW_ROWB;
    static constexpr int LDS = 2 * XBUF + 3 * WTAP + 2 * BN * 4;
};

// One tile: rows [m0, m0 + 128 TM_) x columns [n0, n0 + 32 TN_) of problem g.p[pid].
template <bool POOL, bool SCHED, int TM_, int TN_>
__device__ __forceinline__ void kxrw_tile(const KxrwGroup& g, const int pid, const int m0, const int n0) {
#if defined(__HIP_DEVICE_COMPILE__)
    using SH = KwShape<TM_, TN_>;
    constexpr int BM = SH::BM, BN = SH::BN, NW = 4, TM = TM_, TN = TN_, ROWB = KW_ROWB;
    constexpr int X_BUF = SH::XBUF, W_TAP = SH::WTAP;
    constexpr int XINS = SH::BMX / 16;                 // LDS-DMA pieces (16 rows x 64 B) per X block: 17 / 33
    constexpr int NX = (XINS + NW - 1) / NW;           // 5 / 9 per wave; pieces beyond XINS re-issue the last one
    constexpr int NWP = BN / (NW * 16);                // 2 / 1 instructions per wave and W piece
    static_assert((NX == 5 && NWP == 2) || (NX == 9 && NWP == 1) || (NX == 3 && NWP == 2), "the vmcnt counts exist for these");
    static_assert(!POOL || ((TM == 2 || TM == 1) && TN == 4), "conv-epilogue pooling: the wide form (and its half tiles) only");
    static_assert(TM * TN == 8 || (TM == 1 && TN == 4), "16 MFMAs per phase and wave (8 in a half tile)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ws = smem + 2 * X_BUF;
    float* const tab = (float*)(ws + 3 * W_TAP);       // [scale 128][shift 128]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const IgemmParams& p = g.p[pid];

    const FastDiv d_howo = p.d_howo, d_wo = p.d_wo;
    const int pM = p.M, pN = p.N, pKtot = p.Ktot;
    const uint32_t pR = (uint32_t)p.img_rows;
    const int x_sn = p.x_sn, x_sh_ = p.x_sh, x_sw = p.x_sw, x_base = p.x_base;
    const int o_sn = p.o_sn, o_sw = p.o_sw, o_base = p.o_base;
    const float* const pscale = p.scale;
    const float* const pshift = p.shift;

    // ---- LDS-DMA source offsets (bytes)
    const int lrow = lane >> 2, lpos = lane & 3;
    int xoff[NX], woff[NWP];
#pragma unroll
    for (int q = 0; q < NX; ++q) {
        int ins = wave + NW * q;
        ins = ins < XINS ? ins : XINS - 1;
        const int row = ins * 16 + lrow;
        const int m = m0 + row;                         // not clamped: rows past the last image read zeros (buffer range check)
        const uint32_t img = fdiv((uint32_t)m, d_howo);
        const uint32_t rem = (uint32_t)m - img * d_howo.d;
        const uint32_t y = fdiv(rem, d_wo);
        const uint32_t xq = rem - y * d_wo.d;
        const int el = (int)img * x_sn + (int)y * x_sh_ + (int)xq * x_sw + x_base;
        xoff[q] = el * 2 + ((lpos ^ kw_swz(row)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int row = (wave + NW * i) * 16 + lrow;
        int n = n0 + row;
        n = n < pN ? n : pN - 1;
        woff[i] = (p.w_cm ? n * 64 : n * pKtot * 2) + ((lpos ^ kw_swz(row)) << 4);     // chunk-major W: [Ktot/32][N][32]
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_cm ? p.w_cm : p.w_hi), 0, p.w_bytes, 0x00020000);

    const int CK = __builtin_amdgcn_readfirstlane(p.CK), x_sh = __builtin_amdgcn_readfirstlane(x_sh_);
    const int cchunks = CK / 32;
    const int nsteps = 3 * cchunks;
    const int tapb = CK * 2;
    float tab_s = 1.f, tab_t = 0.f;
    if (tid < BN) {
        const int n = n0 + tid < pN ? n0 + tid : pN - 1;
        if (pscale) tab_s = pscale[n];
        if (pshift) tab_t = pshift[n];
    }
    auto load_x = [&](int buf, int ky_, int cc_) {
        const int xs = __builtin_amdgcn_readfirstlane((ky_ * x_sh + cc_ * 32) * 2);
        char* base_ = smem + buf * X_BUF;
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            int ins = wave + NW * q;
            ins = ins < XINS ? ins : XINS - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(base_ + ins * 1024), 16, xoff[q], xs, 0, 0);
        }
    };
    const int wmul = __builtin_amdgcn_readfirstlane(p.w_cm ? pN : 1);      // a 64-byte K chunk is N * 64 bytes on in the chunk-major plane
    auto load_w = [&](int slot, int wbytes) {
        const int so = __builtin_amdgcn_readfirstlane(wbytes * wmul);
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(ws + slot * W_TAP + (wave + NW * i) * 1024), 16, woff[i], so, 0, 0);
    };
    load_x(0, 0, 0);
    load_w(0, 0);
    load_w(1, tapb);

    // ---- fragment read offsets
    const int l31 = lane & 31, lh = lane >> 5;
    int xrd[3][2], wrd[2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r = wave * (TM * 32) + l31 + kx;
            xrd[kx][ks] = r * ROWB + (((2 * ks + lh) ^ kw_swz(r)) << 4);
        }
    {
        // W rows permuted (bits 2 and 3 swapped): accumulator registers 8h .. 8h+7 of a lane are 8 consecutive channels of its pixel
        const int wrow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wrd[ks] = wrow * ROWB + (((2 * ks + lh) ^ kw_swz(wrow)) << 4);
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- epilogue addressing, LINE layout: a pixel's 128 channels are 256 bytes = 16 lanes of 16 bytes; a store instruction
    // covers 4 pixels; 8 instructions per tile row of 32 pixels
    constexpr int LPP = BN / 8, PPI = 64 / LPP, NEI = 32 / PPI;
    int eoff[TM * NEI];
    const bf16_t* const rhi = (const bf16_t*)p.r_hi;
    {
        const uint32_t wlast = d_wo.d - 1;
        const int img_extra = o_sn - (int)d_howo.d * o_sw;
#pragma unroll
        for (int q = 0; q < TM * NEI; ++q) {
            const int m = m0 + wave * (TM * 32) + (q / NEI) * 32 + (q % NEI) * PPI + lane / LPP;
            const uint32_t mm = (uint32_t)(m < pM ? m : pM - 1);
            const uint32_t img = fdiv(mm, d_howo);
            const uint32_t rem = mm - img * d_howo.d;
            const uint32_t y = fdiv(rem, d_wo);
            const uint32_t xq = rem - y * d_wo.d;
            const bool ok = (m < pM) && rem < pR && xq != 0 && xq != wlast;
            eoff[q] = ok ? (int)mm * o_sw + (int)img * img_extra + o_base + n0 + 8 * (lane % LPP) : -1;
        }
    }
    float* const ppart = POOL ? p.pool_partial : nullptr;

    int ky = 0, cc = 0;
    kw_wait<NWP>();
    __builtin_amdgcn_s_barrier();
    if constexpr (SCHED) {
        // ---- the same phases with their LDS-DMA pieces spread AMONG the MFMAs (igroup pipeline: sched_group_barrier) instead
        // of in front of them: an LDS-DMA instruction costs ~60 cycles of issue between bare MFMAs against 100-185 at the head of
        // a phase beside the fragment reads (MI355X_MICROARCH.md), and in front of the MFMAs that time is on the wave's chain.
        // The loop body has no branch (one scheduling region per phase): the last macro-step is peeled.
        if (tid < BN) { tab[tid] = tab_s; tab[BN + tid] = tab_t; }
        auto phase = [&](auto KX, auto LAST, const char* xb, int st_, int nky_, int ncc_, int wcur_, int wnext_) {
            constexpr int kx = decltype(KX)::value;
            constexpr bool last = decltype(LAST)::value;
            const char* wb = ws + kx * W_TAP;
            bf16x8 xf[2][TM], wf[2][TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) xf[0][t] = *(const bf16x8*)(xb + xrd[kx][0] + t * (32 * ROWB));
#pragma unroll
            for (int t = 0; t < TN; ++t) wf[0][t] = *(const bf16x8*)(wb + wrd[0] + t * (32 * ROWB));
            constexpr int ndma = kx == 0 ? (last ? NWP : NWP + NX) : (last ? 0 : NWP);
#pragma unroll
            for (int t = 0; t < TM; ++t) xf[1][t] = *(const bf16x8*)(xb + xrd[kx][1] + t * (32 * ROWB));
#pragma unroll
            for (int t = 0; t < TN; ++t) wf[1][t] = *(const bf16x8*)(wb + wrd[1] + t * (32 * ROWB));
            // piece i of this phase's LDS-DMA list, in the order the vmcnt counts assume: W pieces first, then X(st + 1)
            auto piece = [&](int i) {
                if (kx == 0) {
                    if (i < NWP) {
                        const int so = __builtin_amdgcn_readfirstlane((wcur_ + 2 * tapb) * wmul);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(ws + 2 * W_TAP + (wave + NW * i) * 1024), 16, woff[i], so, 0, 0);
                    } else {
                        const int q = i - NWP;
                        const int xs = __builtin_amdgcn_readfirstlane((nky_ * x_sh + ncc_ * 32) * 2);
                        int ins = wave + NW * q;
                        ins = ins < XINS ? ins : XINS - 1;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(smem + ((st_ + 1) & 1) * X_BUF + ins * 1024), 16, xoff[q], xs, 0, 0);
                    }
                } else {
                    const int so = __builtin_amdgcn_readfirstlane((wnext_ + (kx - 1) * tapb) * wmul);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(ws + (kx - 1) * W_TAP + (wave + NW * i) * 1024), 16, woff[i], so, 0, 0);
                }
            };
            // all 12 fragment reads first (an LDS-DMA write may not pass an LDS read in program order), then MFMA pairs with one
            // piece behind each until the pieces are out
            int ip = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[ks][tn]),
                                                                             __builtin_bit_cast(f16x8, xf[ks][tm]), acc[tn][tm], 0, 0, 0);
                        if ((ndma > 8 || TM == 1 || (tm & 1)) && ip < ndma) { piece(ip); ++ip; }     // behind every second MFMA; every one if > 8 pieces (or 8 MFMAs)
                    }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
            if constexpr (ndma > 8) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < ndma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, TM == 1 ? 1 : 2, 0);
                    if (i < ndma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using BF = std::false_type; using BT = std::true_type;
        for (int st = 0; st < nsteps - 1; ++st) {
            int nky = ky, ncc = cc + 1;
            if (ncc == cchunks) { ncc = 0; ++nky; }
            const int wcur = (ky * 3 * CK + cc * 32) * 2, wnext = (nky * 3 * CK + ncc * 32) * 2;
            const char* xb = smem + (st & 1) * X_BUF;
            phase(I0{}, BF{}, xb, st, nky, ncc, wcur, wnext);
            kw_wait<NX + NWP>();
            __builtin_amdgcn_s_barrier();
            phase(I1{}, BF{}, xb, st, nky, ncc, wcur, wnext);
            kw_wait<NX + NWP>();
            __builtin_amdgcn_s_barrier();
            phase(I2{}, BF{}, xb, st, nky, ncc, wcur, wnext);
            kw_wait<NWP>();
            __builtin_amdgcn_s_barrier();
            ky = nky; cc = ncc;
        }
        {
            const int st = nsteps - 1;
            const int wcur = (ky * 3 * CK + cc * 32) * 2;
            const char* xb = smem + (st & 1) * X_BUF;
            phase(I0{}, BT{}, xb, st, 0, 0, wcur, 0);
            kw_wait<NWP>();
            __builtin_amdgcn_s_barrier();
            phase(I1{}, BT{}, xb, st, 0, 0, wcur, 0);
            kw_wait<0>();
            __builtin_amdgcn_s_barrier();
            phase(I2{}, BT{}, xb, st, 0, 0, wcur, 0);
        }
    } else {
    for (int st = 0; st < nsteps; ++st) {
        int nky = ky, ncc = cc + 1;
        if (ncc == cchunks) { ncc = 0; ++nky; }
        const int wcur = (ky * 3 * CK + cc * 32) * 2, wnext = (nky * 3 * CK + ncc * 32) * 2;
        const bool last = st == nsteps - 1;
        const char* xb = smem + (st & 1) * X_BUF;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const char* wb = ws + kx * W_TAP;
            bf16x8 xf[2][TM], wf[2][TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) xf[0][t] = *(const bf16x8*)(xb + xrd[kx][0] + t * (32 * ROWB));
#pragma unroll
            for (int t = 0; t < TN; ++t) wf[0][t] = *(const bf16x8*)(wb + wrd[0] + t * (32 * ROWB));
            // ---- this phase's loads (behind the first fragment reads)
            if (kx == 0) {
                load_w(2, wcur + 2 * tapb);
                if (!last) load_x((st + 1) & 1, nky, ncc);
                if (st == 0 && tid < BN) { tab[tid] = tab_s; tab[BN + tid] = tab_t; }
            } else if (!last) {
                load_w(kx - 1, wnext + (kx - 1) * tapb);
            }
#pragma unroll
            for (int t = 0; t < TM; ++t) xf[1][t] = *(const bf16x8*)(xb + xrd[kx][1] + t * (32 * ROWB));
#pragma unroll
            for (int t = 0; t < TN; ++t) wf[1][t] = *(const bf16x8*)(wb + wrd[1] + t * (32 * ROWB));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[ks][tn]),
                                                                             __builtin_bit_cast(f16x8, xf[ks][tm]), acc[tn][tm], 0, 0, 0);
            // ---- retire what the next phase reads, then open it
            if (kx == 2) {
                if (last) break;
                kw_wait<NWP>();
            } else if (!last) {
                kw_wait<NX + NWP>();
            } else if (kx == 0) {
                kw_wait<NWP>();
            } else {
                kw_wait<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
        ky = nky; cc = ncc;
    }
    }

    // ---- epilogue: accumulator layout (a lane = one pixel, 8 x 8 consecutive channels) <-> line layout through a wave-private
    // LDS strip of 32 rows x (256 + 16) bytes.  Residual of tile row 0 in flight before the barrier.
    constexpr int ERS = 2 * BN + 16;
    u32x4 rpf[NEI];
    auto load_residual = [&](int tm) {
#pragma unroll
        for (int i = 0; i < NEI; ++i) {
            const int off = eoff[tm * NEI + i];
            rpf[i] = *(const u32x4*)(rhi + (off >= 0 ? off : 0));
        }
    };
    if (rhi) load_residual(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* const strip = smem + wave * (32 * ERS);
    const int a_off = l31 * ERS + lh * 16;
    const int l_off = (lane / LPP) * ERS + (lane % LPP) * 16;
    const float* tb = tab + 8 * lh;
    bf16_t* const ohi = (bf16_t*)p.o_hi;
    const float relu_lo = p.relu ? 0.f : -65504.f;
    float psum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};       // [channel half][stat]
    const float* const ppp = POOL ? p.pool_p : nullptr;
    const float pool_pw = ppp ? ppp[0] : 1.f, pool_eps = POOL ? p.pool_eps : 0.f;
    const bool pool_cube = pool_pw == 3.f;
    const bool pool_sq = POOL && p.pool_sq;           // stat 1 = sum of squares (BatchNorm statistics), no exponent tensor
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        u32x4 rres[TN * 2];
        if (rhi) {
            // line layout -> strip -> accumulator layout; then the next tile row's residual loads go out BEFORE this row's stores
#pragma unroll
            for (int i = 0; i < NEI; ++i) *(u32x4*)(strip + l_off + i * (PPI * ERS)) = rpf[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int jj = 0; jj < TN * 2; ++jj) rres[jj] = *(const u32x4*)(strip + a_off + jj * 32);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (tm + 1 < TM) load_residual(tm + 1);
        }
        u32x4 outv[TN * 2];
#pragma unroll
        for (int jj = 0; jj < TN * 2; ++jj) {           // channels 16 jj + 8 lh .. + 7 of the tile's 128 columns
            const f32x4 s0 = *(const f32x4*)(tb + 16 * jj), s1 = *(const f32x4*)(tb + 16 * jj + 4);
            const f32x4 h0 = *(const f32x4*)(tb + BN + 16 * jj), h1 = *(const f32x4*)(tb + BN + 16 * jj + 4);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = acc[jj >> 1][tm][8 * (jj & 1) + e] * (e < 4 ? s0[e & 3] : s1[e & 3]) + (e < 4 ? h0[e & 3] : h1[e & 3]);
            if (rhi) {
                float r[8];
                unpack8_h(rres[jj], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            outv[jj] = pack8_h_lo(v, relu_lo);
        }
#pragma unroll
        for (int jj = 0; jj < TN * 2; ++jj) *(u32x4*)(strip + a_off + jj * 32) = outv[jj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        u32x4 lines[NEI];
#pragma unroll
        for (int i = 0; i < NEI; ++i) lines[i] = *(const u32x4*)(strip + l_off + i * (PPI * ERS));
#pragma unroll
        for (int i = 0; i < NEI; ++i) {
            const int off = eoff[tm * NEI + i];
            if (off >= 0) *(u32x4*)(ohi + off) = lines[i];
        }
        if constexpr (POOL) {
            if (ppart) {
                // the strip holds the tile row as stored: 32 pixels x 128 channels fp16.  Pixels that are not stored (halo columns,
                // raster rows past the image) are ZEROED in the strip first (the lanes that hold their lines, exec-masked 16-byte
                // writes), so the sweep below needs no per-element mask: a zero adds nothing to the mean and eps^p ~ 1e-18 to the
                // GeM sum.  lane = a PAIR of channels (2 lane, 2 lane + 1): one ds_read_b32 per pixel covers the 128 channels.
#pragma unroll
                for (int i = 0; i < NEI; ++i)
                    if (eoff[tm * NEI + i] < 0) *(u32x4*)(strip + l_off + i * (PPI * ERS)) = u32x4{0u, 0u, 0u, 0u};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const uint32_t* const col = (const uint32_t*)strip + lane;
#pragma unroll
                for (int p8 = 0; p8 < 32; p8 += 8) {
                    uint32_t w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = col[(p8 + u) * (ERS / 4)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float v0 = h2f((bf16_t)(w[u] & 0xffffu)), v1 = h2f((bf16_t)(w[u] >> 16));
                        psum[0][0] += v0;
                        psum[1][0] += v1;
                        if (pool_sq) {
                            psum[0][1] += v0 * v0;
                            psum[1][1] += v1 * v1;
                        } else if (ppp) {
                            const float c0 = fmaxf(v0, pool_eps), c1 = fmaxf(v1, pool_eps);
                            psum[0][1] += pool_cube ? c0 * c0 * c0 : __builtin_exp2f(pool_pw * __builtin_log2f(c0));
                            psum[1][1] += pool_cube ? c1 * c1 * c1 : __builtin_exp2f(pool_pw * __builtin_log2f(c1));
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if constexpr (POOL) {
        if (ppart) {
            // [64-row block][stat][N]: a wave of a full tile IS a block; in a half tile (32 rows per wave) the odd wave hands its sums
            // to the even wave of its pair through LDS (fixed order: even + odd)
            if constexpr (TM == 1) {
                float* const scr = (float*)(ws + 3 * W_TAP) + 2 * BN;          // behind the scale / shift table
                if (wave & 1) *(f32x4*)(scr + ((wave >> 1) * 64 + lane) * 4) = f32x4{psum[0][0], psum[0][1], psum[1][0], psum[1][1]};
                __syncthreads();
                if (!(wave & 1)) {
                    const f32x4 o = *(const f32x4*)(scr + ((wave >> 1) * 64 + lane) * 4);
                    psum[0][0] += o[0]; psum[0][1] += o[1]; psum[1][0] += o[2]; psum[1][1] += o[3];
                }
            }
            if (TM == 2 || !(wave & 1)) {
                const int block = m0 / 64 + (TM == 2 ? wave : (wave >> 1));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = n0 + 2 * lane + h;          // psum[h]: channel 2 lane + h of the tile's 128 columns
                    if (n < pN) {
                        float* o = ppart + ((size_t)block * 2) * pN + n;      // [block][stat][N]
                        o[0] = psum[h][0];
                        if (ppp || pool_sq) o[pN] = psum[h][1];
                    }
                }
            }
        }
    }
#endif
}

// block -> (problem, row tile, column tile).  XCD x owns a contiguous chunk of the global row tiles [0, MT_full); MIX: the blocks
// from half_bid0 on are the HALF tiles (128 rows) of the row tiles [MT_full, MT) -- the launch's last, partial round of workgroups.
// They carry the highest block ids, so they are dispatched last: the long tiles first, the short ones fill the end.
template <bool POOL, bool SCHED = false, int TM_ = 2, int TN_ = 4, bool MIX = false>
__global__ void __launch_bounds__(256, 2) igemm_kxrw_kernel(KxrwGroup g) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = KwShape<TM_, TN_>::BM, BN = KwShape<TM_, TN_>::BN;
    const int bid = blockIdx.x;
    const int gNT = g.NT, gnprob = g.nprob;
    const int e0 = g.mt_end[0], e1 = g.mt_end[1], e2 = g.mt_end[2];
    int mt, nt, sub = 0;
    bool half = false;
    if (MIX && bid >= g.half_bid0) {
        const int h = bid - g.half_bid0;
        nt = h % gNT;
        const int hm = h / gNT;
        mt = g.MT_full + (hm >> 1);
        sub = hm & 1;
        half = true;
        if (mt >= g.MT) return;
    } else {
        const int xcd = bid & 7, j = bid >> 3;
        nt = j % gNT;
        mt = xcd * g.mt_chunk + j / gNT;
        if (mt >= g.MT_full) return;
    }
    int pid = 0, base = 0;
    if (gnprob > 1 && mt >= e0) { pid = 1; base = e0; }
    if (gnprob > 2 && mt >= e1) { pid = 2; base = e1; }
    if (gnprob > 3 && mt >= e2) { pid = 3; base = e2; }
    mt -= base;
    const int n0 = nt * BN;
    if constexpr (MIX) {
        if (half) {
            const int m0 = mt * BM + sub * (BM / 2);
            if (m0 >= g.p[pid].M) return;                  // the second half of a problem's last, partial row tile
            kxrw_tile<POOL, SCHED, 1, TN_>(g, pid, m0, n0);
            return;
        }
    }
    kxrw_tile<POOL, SCHED, TM_, TN_>(g, pid, mt * BM, n0);
#endif
}

template <bool POOL, bool SCHED, int TM_ = 2, int TN_ = 4, bool MIX = false>
int launch_kxrw(KxrwGroup& g, hipStream_t s) {
    constexpr int lds = KwShape<TM_, TN_>::LDS;
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    static_assert(!MIX || (TM_ == 2 && KwShape<1, TN_>::LDS + 3072 <= lds), "half tiles: their stage + the pooling scratch fit the full tile's LDS");
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)igemm_kxrw_kernel<POOL, SCHED, TM_, TN_, MIX>, lds, attr_done)) return AGP_E_LAUNCH;
    const int nblocks = g.mt_chunk * 8 * g.NT + (MIX ? 2 * (g.MT - g.MT_full) * g.NT : 0);
    AGP_LAUNCH((igemm_kxrw_kernel<POOL, SCHED, TM_, TN_, MIX>), dim3(nblocks), dim3(256), lds, s, g);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

}  // namespace agp_igemm

// `ps[i]` arrive with the padded-width raster geometry of agp_internal_conv_kxr_geometry; all share N, CK, prec F16.
// N % 128 == 0: the wide form (256 x 128 tiles).  (The tall form, 512 x 64 tiles for N == 64 -- a round-3 experiment that measured
// the same alone and 4 % slower in the grouped launch -- exists in the development build only: KXR_TALL.)
static constexpr bool KXRW_MIX = true;     // false: every tile 256 rows (rounds 3-5)
int agp_internal_conv_kxrw(agp_igemm::IgemmParams* ps, int n, hipStream_t s) {
    using namespace agp_igemm;
#if defined(AGP_TUNING)
    const bool tall = ps[0].N == 64;
#else
    constexpr bool tall = false;
#endif
    if (n < 1 || n > KXRW_MAXP || (!tall && ps[0].N % 128)) return AGP_E_BADARG;
    const int bm = tall ? KwShape<4, 2>::BM : KwShape<2, 4>::BM, bn = tall ? 64 : 128;
    KxrwGroup g = {};
    g.nprob = n;
    int mt = 0;
    bool pool = false;
    for (int i = 0; i < n; ++i) {
        if (ps[i].N != ps[0].N || ps[i].CK != ps[0].CK) return AGP_E_BADARG;
        g.p[i] = ps[i];
        mt += (ps[i].M + bm - 1) / bm;
        g.mt_end[i] = mt;
        pool = pool || ps[i].pool_partial != nullptr;
    }
    g.MT = mt;
    g.NT = ps[0].N / bn;
    g.MT_full = g.MT;
    // ---- the last round of workgroups as half tiles.  512 workgroups are resident (two per CU); a launch of T tiles runs
    // ceil(T / 512) rounds and its last round holds `tail` tiles.  When that round would leave more than half of the CUs without
    // a workgroup (tail <= 128), its tiles run as twice as many 128-row tiles, each still alone on a CU: the round takes about half
    // as long (stage 2: 602 tiles = 512 + 90 -> 180 half tiles, 89.7 -> 80.1 us; a launch of <= 128 tiles -- the C1 / C2 shapes --
    // covers twice the CUs).  Measured and NOT done: a larger tail (layer 3: 714 = 512 + 202 -> 404 half tiles, two per CU) loses
    // 3-4 us per launch -- a lone 256-row workgroup already runs 1.65 x as fast as one of a pair, two half tiles per CU do not.
    bool mix = false;
    if (!tall && KXRW_MIX) {
        const int slots = 512, T = g.MT * g.NT;
        const int tail = T - (T - 1) / slots * slots;         // 1 .. slots
        if (tail <= slots / 4 && tail % g.NT == 0) {
            g.MT_full = g.MT - tail / g.NT;
            mix = true;
        }
    }
    g.mt_chunk = (g.MT_full + 7) / 8;
    g.half_bid0 = g.mt_chunk * 8 * g.NT;
#if defined(AGP_TUNING)
    const int sched = AGP_TUNE("KXRW_SCHED", 1);    // 0: LDS-DMA pieces at the head of a phase (the round-3 order) instead of among the MFMAs
    if (tall) {
        if (pool) return AGP_E_BADARG;
        return sched ? launch_kxrw<false, true, 4, 2>(g, s) : launch_kxrw<false, false, 4, 2>(g, s);
    }
    if (!sched) return pool ? launch_kxrw<true, false>(g, s) : launch_kxrw<false, false>(g, s);
#endif
    if (mix) return pool ? launch_kxrw<true, true, 2, 4, true>(g, s) : launch_kxrw<false, true, 2, 4, true>(g, s);
    return pool ? launch_kxrw<true, true>(g, s) : launch_kxrw<false, true>(g, s);
}
