// igemm_s2.hip -- the stride-2 ENTRY of a ResNet stage on fp16 maps (AGP_PREC_F16): the block's first 3x3 / stride 2 / pad 1
// convolution AND its 1x1 / stride 2 downsample (reference torchvision BasicBlock as driven by network_mm/image_fe.py:97-113:
// `conv1` and `downsample[0]` read the same input) in ONE kernel, folded BatchNorm (+ ReLU for the 3x3) in the epilogues.
//
// The generic kernel ran the two convolutions as separate tiles of one grid (igemm.hip, grouped launch): 0.21 MFMA-busy, 2.7
// TB/s, and the downsample (K = Cin: ONE K-step per tile, all prologue and epilogue) cost as much as the 3x3.  Here:
//   * implicit GEMM over the PADDED-WIDTH raster of the OUTPUT map (igemm_kxr2.hip): consecutive GEMM rows are consecutive
//     output pixels, i.e. input columns two apart.  Per (ky, 32-channel chunk) TWO row blocks are staged -- O: the input pixels
//     at padded columns 2 x' - 2, E: at 2 x' - 1 (x' = padded output column) -- and the three horizontal taps are
//     kx = 0: O rows + 0, kx = 1: E rows + 0, kx = 2: O rows + 1: 2 staged blocks serve 3 taps (the gather form staged 3);
//   * the downsample is the centre tap of the same staged E block (ky = 1): in that phase every wave issues four more MFMAs
//     on the 1x1 weights (their own LDS slot) into a second accumulator set -- no second pass over the input, no tiles of
//     their own;
//   * the phase pipeline of igemm_kxr2: W taps through a 3-slot ring, X double-buffered, every LDS-DMA issued two (W) or three
//     (X) phases ahead, raw s_barrier + counted s_waitcnt vmcnt(N); 128 x 64 tiles (four waves of 32 x 64), 53 KB LDS: three
//     workgroups per CU; up to two problems (query and database trunk) per launch.
//
// vmcnt bookkeeping (per wave, issue order; NX = X pieces per wave and macro-step, D = the 1x1 weights' chunk):
//     prologue          : X(0)[NX]  W(0,0)  W(0,1)
//     phase (st,0)      : W(st,2)   X(st+1)[NX]
//     phase (st,1)      : W(st+1,0)
//     phase (st,2)      : W(st+1,1) D(st+1)              (D re-loads a valid chunk when macro-step st+1 is not a ky = 1 step)
//   opens (st,1): W(st,1), D(st);   younger: W(st,2) X(st+1)            -> vmcnt(NX+1)
//   opens (st,2): W(st,2);          younger: X(st+1) W(st+1,0)          -> vmcnt(NX+1)
//   opens (st+1,0): W(st+1,0) X(st+1); younger: W(st+1,1) D(st+1)       -> vmcnt(2)
//   last macro-step L: (L,0) issues W(L,2) only -> opens (L,1): vmcnt(1); (L,1) issues nothing -> opens (L,2): vmcnt(0).

#include <type_traits>

#include "igemm_params.hpp"

namespace agp_igemm {

__device__ __forceinline__ int s2_swz(int row) { return (row >> 2) & 3; }   // XOR-swizzle of a row's four 16-byte chunks

constexpr int S2_MAXP = 2;
struct S2Group {
    IgemmParams p[S2_MAXP];
    int mt_end[S2_MAXP];
    int nprob, MT, NT, mt_chunk;
};

template <int N> __device__ __forceinline__ void s2_wait() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
    else static_assert(N < 0, "add the count");
}

constexpr int S2_BM = 128, S2_ROWB = 64;
constexpr int S2_BMX = S2_BM + 16;                      // rows of a staged block (offsets 0 and 1 are read)
constexpr int S2_XBLK = S2_BMX * S2_ROWB;               // one block (O or E)
constexpr int S2_XBUF = 2 * S2_XBLK;                    // O then E
template <int TN> constexpr int s2_lds() {              // X double buffer, W ring (3) + D slot, two scale / shift tables
    return 2 * S2_XBUF + 4 * (32 * TN * S2_ROWB) + 4 * (32 * TN) * 4;
}

// TN = column tiles of 32 channels per wave: 2 = 128 x 64 tiles (54 KB LDS: three workgroups per CU, 4 MFMAs per phase and wave);
// 4 = 128 x 128 tiles (71 KB: two per CU, 8 MFMAs per phase, the X blocks staged once per 128 channels).
// SCH (the default; AGP_S2_SCHED=0 turns it off): a phase's LDS-DMA pieces are issued among its MFMAs instead of in front of them,
// all of the phase's fragment reads first, the last macro-step peeled (see igemm_kxrw.hip).
template <int TN, bool SCH = false>
__global__ void __launch_bounds__(256, TN == 2 ? 3 : 2) igemm_s2_kernel(S2Group g) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = S2_BM, BN = 32 * TN, NW = 4, ROWB = S2_ROWB;
    constexpr int S2_WTAP = BN * S2_ROWB;
    constexpr int NWP = TN / 2;                         // LDS-DMA instructions per wave for one W piece (BN rows of 64 B)
    constexpr int XINS = 2 * (S2_BMX / 16);             // LDS-DMA pieces (16 rows x 64 B) per macro-step: O block then E block
    constexpr int NX = (XINS + NW - 1) / NW;            // per wave; pieces beyond XINS re-issue the last one
    static_assert(NX == 5, "the vmcnt counts below are written for NX = 5");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ws = smem + 2 * S2_XBUF;                // W ring slots 0..2, slot 3 = D (1x1 weights)
    float* const tab = (float*)(ws + 4 * S2_WTAP);      // [conv scale 64][conv shift 64][ds scale 64][ds shift 64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int gNT = g.NT, gchunk = g.mt_chunk, gMT = g.MT, gnprob = g.nprob;
    const int e0 = g.mt_end[0];
    const int nt = j % gNT;
    int mt = xcd * gchunk + j / gNT;
    if (mt >= gMT) return;
    int pid = 0;
    if (gnprob > 1 && mt >= e0) { pid = 1; mt -= e0; }
    const IgemmParams& p = g.p[pid];
    const int m0 = mt * BM, n0 = nt * BN;

    const FastDiv d_howo = p.d_howo, d_wo = p.d_wo;
    const int pM = p.M, pN = p.N, pKtot = p.Ktot;
    const int x_sn = p.x_sn, x_sh_ = p.x_sh, x_sw = p.x_sw, x_base = p.x_base;
    const int o_sn = p.o_sn, o_sw = p.o_sw, o_base = p.o_base;
    const bool has_ds = p.w2_hi != nullptr;

    // ---- LDS-DMA source offsets (bytes).  Piece i < 9 is rows 16 i .. of the O block, piece 9 + i of the E block (one input
    // pixel = `cpix` bytes further: the E column is the O column + 1).  Rows past the map / before it read zeros (range check).
    const int lrow = lane >> 2, lpos = lane & 3;
    const int cpix = p.CK * 2;                          // bytes of one input pixel
    int xoff[NX];
#pragma unroll
    for (int q = 0; q < NX; ++q) {
        int ins = wave + NW * q;
        ins = ins < XINS ? ins : XINS - 1;
        const int blk = ins >= XINS / 2 ? 1 : 0;        // 0: O, 1: E
        const int row = (ins - blk * (XINS / 2)) * 16 + lrow;
        const int m = m0 + row;
        const uint32_t img = fdiv((uint32_t)m, d_howo);
        const uint32_t rem = (uint32_t)m - img * d_howo.d;
        const uint32_t y = fdiv(rem, d_wo);
        const uint32_t xq = rem - y * d_wo.d;
        // element offset of padded input pixel (2 y + ky, 2 x' - 2) for ky = 0 (x_sh = two input rows, x_sw = two input pixels)
        const int el = (int)img * x_sn + (int)y * x_sh_ + (int)xq * x_sw + x_base;
        xoff[q] = el * 2 + blk * cpix + ((lpos ^ s2_swz(row)) << 4);
    }
    int woff_c[NWP], woff_d[NWP];                      // W piece i of a wave: rows (wave + 4 i) * 16 .. of the BN-row slot
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int row = (wave + NW * i) * 16 + lrow;
        const int n = (n0 + row) < pN ? (n0 + row) : pN - 1;
        const int wsw = (lpos ^ s2_swz(row)) << 4;
        woff_c[i] = (p.w_cm ? n * 64 : n * pKtot * 2) + wsw;     // 3x3 weights [N][3][3][CK], or chunk-major [9 CK / 32][N][32]
        woff_d[i] = (p.w2_cm ? n * 64 : n * p.CK * 2) + wsw;     // 1x1 weights [N][CK], or chunk-major [CK / 32][N][32]
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_cm ? p.w_cm : p.w_hi), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(has_ds ? (p.w2_cm ? p.w2_cm : p.w2_hi) : (p.w_cm ? p.w_cm : p.w_hi)), 0,
                                                                        has_ds ? (uint32_t)(pN * p.CK * 2) : p.w_bytes, 0x00020000);

    const int CK = __builtin_amdgcn_readfirstlane(p.CK);
    const int in_row = __builtin_amdgcn_readfirstlane(p.sy);     // elements of ONE input row (x_sh is two)
    const int cchunks = CK / 32;
    const int nsteps = 3 * cchunks;
    const int tapb = CK * 2;
    float tab_v[4] = {1.f, 0.f, 1.f, 0.f};
    if (tid < BN) {
        const int n = n0 + tid < pN ? n0 + tid : pN - 1;
        if (p.scale) tab_v[0] = p.scale[n];
        if (p.shift) tab_v[1] = p.shift[n];
        if (has_ds && p.scale2) tab_v[2] = p.scale2[n];
        if (has_ds && p.shift2) tab_v[3] = p.shift2[n];
    }
    auto load_x = [&](int buf, int ky_, int cc_) {
        const int xs = __builtin_amdgcn_readfirstlane((ky_ * in_row + cc_ * 32) * 2);
        char* base = smem + buf * S2_XBUF;
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            int ins = wave + NW * q;
            ins = ins < XINS ? ins : XINS - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(base + ins * 1024), 16, xoff[q], xs, 0, 0);
        }
    };
    // a 64-byte K chunk is N * 64 bytes on in a chunk-major plane
    const int wmul = __builtin_amdgcn_readfirstlane(p.w_cm ? pN : 1), dmul = __builtin_amdgcn_readfirstlane(p.w2_cm ? pN : 1);
    auto load_w = [&](int slot, int wbytes) {
        const int so = __builtin_amdgcn_readfirstlane(wbytes * wmul);
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(ws + slot * S2_WTAP + (wave + NW * i) * 1024), 16, woff_c[i], so, 0, 0);
    };
    auto load_d = [&](int cc_) {                        // the 1x1 weights' 32-channel chunk -> slot 3
        const int so = __builtin_amdgcn_readfirstlane(cc_ * 64 * dmul);
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, LDS_PTR(ws + 3 * S2_WTAP + (wave + NW * i) * 1024), 16,
                                                     has_ds ? woff_d[i] : woff_c[i], has_ds ? so : 0, 0, 0);
    };
    load_x(0, 0, 0);
    load_w(0, 0);
    load_w(1, tapb);

    // ---- fragment read offsets
    const int l31 = lane & 31, lh = lane >> 5;
    int xrd[3][2];                                      // [kx][ks]
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r = wave * 32 + l31 + (kx == 2 ? 1 : 0);
            xrd[kx][ks] = (kx == 1 ? S2_XBLK : 0) + r * ROWB + (((2 * ks + lh) ^ s2_swz(r)) << 4);
        }
    int wrd[2];
    {
        // W rows permuted (bits 2 and 3 swapped): accumulator registers 8h .. 8h+7 of a lane are 8 consecutive channels of its pixel
        const int wrow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wrd[ks] = wrow * ROWB + (((2 * ks + lh) ^ s2_swz(wrow)) << 4);
    }

    f32x16 acc[TN], acc2[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[a][r] = 0.f; acc2[a][r] = 0.f; }

    // ---- epilogue addressing, LINE layout (igemm_kxr2.hip): a pixel's BN channels are 2 BN bytes = LPP lanes of 16 bytes; one
    // store instruction covers 64 / LPP pixels of the wave's 32 rows
    constexpr int LPP = BN / 8, PPI = 64 / LPP, NEI = 32 / PPI;
    int eoff[NEI];
    {
        const uint32_t wlast = d_wo.d - 1;
        const int img_extra = o_sn - (int)d_howo.d * o_sw;
#pragma unroll
        for (int q = 0; q < NEI; ++q) {
            const int m = m0 + wave * 32 + q * PPI + lane / LPP;
            const uint32_t mm = (uint32_t)(m < pM ? m : pM - 1);
            const uint32_t img = fdiv(mm, d_howo);
            const uint32_t rem = mm - img * d_howo.d;
            const uint32_t y = fdiv(rem, d_wo);
            const uint32_t xq = rem - y * d_wo.d;
            const bool ok = (m < pM) && xq != 0 && xq != wlast;
            eoff[q] = ok ? (int)mm * o_sw + (int)img * img_extra + o_base + n0 + 8 * (lane % LPP) : -1;
        }
    }

    int ky = 0, cc = 0;
    s2_wait<NWP>();
    __builtin_amdgcn_s_barrier();
    if constexpr (SCH) {
        if (tid < BN) { tab[tid] = tab_v[0]; tab[BN + tid] = tab_v[1]; tab[2 * BN + tid] = tab_v[2]; tab[3 * BN + tid] = tab_v[3]; }
        auto phase = [&](auto KX, auto LAST, const char* xb, int st_, int ky_, int nky_, int ncc_, int wcur_, int wnext_) {
            constexpr int kx = decltype(KX)::value;
            constexpr bool last = decltype(LAST)::value;
            const char* wb = ws + kx * S2_WTAP;
            bf16x8 xf[2], wf[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                xf[ks] = *(const bf16x8*)(xb + xrd[kx][ks]);
#pragma unroll
                for (int t = 0; t < TN; ++t) wf[ks][t] = *(const bf16x8*)(wb + wrd[ks] + t * (32 * ROWB));
            }
            // the phase's LDS-DMA list in the order the vmcnt counts assume
            constexpr int ndma = kx == 0 ? (last ? NWP : NWP + NX) : (last ? 0 : (kx == 2 ? 2 * NWP : NWP));
            auto piece = [&](int i) {
                if (kx == 0) {
                    if (i < NWP) {
                        const int so = __builtin_amdgcn_readfirstlane((wcur_ + 2 * tapb) * wmul);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(ws + 2 * S2_WTAP + (wave + NW * i) * 1024), 16, woff_c[i], so, 0, 0);
                    } else {
                        const int q = i - NWP;
                        const int xs = __builtin_amdgcn_readfirstlane((nky_ * in_row + ncc_ * 32) * 2);
                        int ins = wave + NW * q;
                        ins = ins < XINS ? ins : XINS - 1;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(smem + ((st_ + 1) & 1) * S2_XBUF + ins * 1024), 16, xoff[q], xs, 0, 0);
                    }
                } else if (i < NWP) {
                    const int so = __builtin_amdgcn_readfirstlane((wnext_ + (kx - 1) * tapb) * wmul);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(ws + (kx - 1) * S2_WTAP + (wave + NW * i) * 1024), 16, woff_c[i], so, 0, 0);
                } else {
                    const int so = __builtin_amdgcn_readfirstlane((nky_ == 1 ? ncc_ : 0) * 64 * dmul);
                    const int k = i - NWP;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, LDS_PTR(ws + 3 * S2_WTAP + (wave + NW * k) * 1024), 16,
                                                             has_ds ? woff_d[k] : woff_c[k], has_ds ? so : 0, 0, 0);
                }
            };
            int ip = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[ks][tn]), __builtin_bit_cast(f16x8, xf[ks]),
                                                                     acc[tn], 0, 0, 0);
                    if (ip < ndma) { piece(ip); ++ip; }
                }
#pragma unroll
            for (int i = 2 * TN; i < ndma; ++i) piece(i);       // (64-channel tiles: more pieces than MFMAs in the kx = 0 phase)
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (1 + TN), 0);
#pragma unroll
            for (int i = 0; i < 2 * TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < ndma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kx == 1 && ky_ == 1 && has_ds) {
                // the downsample: centre tap of the staged E block on the 1x1 weights (slot 3)
                const char* db = ws + 3 * S2_WTAP;
                bf16x8 df[2][TN];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < TN; ++t) df[ks][t] = *(const bf16x8*)(db + wrd[ks] + t * (32 * ROWB));
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc2[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, df[ks][tn]),
                                                                          __builtin_bit_cast(f16x8, xf[ks]), acc2[tn], 0, 0, 0);
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        for (int st = 0; st < nsteps - 1; ++st) {
            int nky = ky, ncc = cc + 1;
            if (ncc == cchunks) { ncc = 0; ++nky; }
            const int wcur = (ky * 3 * CK + cc * 32) * 2, wnext = (nky * 3 * CK + ncc * 32) * 2;
            const char* xb = smem + (st & 1) * S2_XBUF;
            phase(I0{}, std::false_type{}, xb, st, ky, nky, ncc, wcur, wnext);
            s2_wait<NX + NWP>();
            __builtin_amdgcn_s_barrier();
            phase(I1{}, std::false_type{}, xb, st, ky, nky, ncc, wcur, wnext);
            s2_wait<NX + NWP>();
            __builtin_amdgcn_s_barrier();
            phase(I2{}, std::false_type{}, xb, st, ky, nky, ncc, wcur, wnext);
            s2_wait<2 * NWP>();
            __builtin_amdgcn_s_barrier();
            ky = nky; cc = ncc;
        }
        {
            const int st = nsteps - 1;
            const int wcur = (ky * 3 * CK + cc * 32) * 2;
            const char* xb = smem + (st & 1) * S2_XBUF;
            phase(I0{}, std::true_type{}, xb, st, ky, 0, 0, wcur, 0);
            s2_wait<NWP>();
            __builtin_amdgcn_s_barrier();
            phase(I1{}, std::true_type{}, xb, st, ky, 0, 0, wcur, 0);
            s2_wait<0>();
            __builtin_amdgcn_s_barrier();
            phase(I2{}, std::true_type{}, xb, st, ky, 0, 0, wcur, 0);
        }
    } else
    for (int st = 0; st < nsteps; ++st) {
        int nky = ky, ncc = cc + 1;
        if (ncc == cchunks) { ncc = 0; ++nky; }
        const int wcur = (ky * 3 * CK + cc * 32) * 2, wnext = (nky * 3 * CK + ncc * 32) * 2;
        const bool last = st == nsteps - 1;
        const char* xb = smem + (st & 1) * S2_XBUF;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const char* wb = ws + kx * S2_WTAP;
            bf16x8 xf[2], wf[2][TN];
            xf[0] = *(const bf16x8*)(xb + xrd[kx][0]);
#pragma unroll
            for (int t = 0; t < TN; ++t) wf[0][t] = *(const bf16x8*)(wb + wrd[0] + t * (32 * ROWB));
            // ---- this phase's loads (behind the first fragment reads)
            if (kx == 0) {
                load_w(2, wcur + 2 * tapb);
                if (!last) load_x((st + 1) & 1, nky, ncc);
                if (st == 0 && tid < BN) {
                    tab[tid] = tab_v[0]; tab[BN + tid] = tab_v[1]; tab[2 * BN + tid] = tab_v[2]; tab[3 * BN + tid] = tab_v[3];
                }
            } else if (!last) {
                load_w(kx - 1, wnext + (kx - 1) * tapb);
                if (kx == 2) load_d(nky == 1 ? ncc : 0);
            }
            xf[1] = *(const bf16x8*)(xb + xrd[kx][1]);
#pragma unroll
            for (int t = 0; t < TN; ++t) wf[1][t] = *(const bf16x8*)(wb + wrd[1] + t * (32 * ROWB));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[ks][tn]), __builtin_bit_cast(f16x8, xf[ks]),
                                                                     acc[tn], 0, 0, 0);
            if (kx == 1 && ky == 1 && has_ds) {
                // the downsample: centre tap of the staged E block on the 1x1 weights (slot 3)
                const char* db = ws + 3 * S2_WTAP;
                bf16x8 df[2][TN];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < TN; ++t) df[ks][t] = *(const bf16x8*)(db + wrd[ks] + t * (32 * ROWB));
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc2[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, df[ks][tn]),
                                                                          __builtin_bit_cast(f16x8, xf[ks]), acc2[tn], 0, 0, 0);
            }
            // ---- retire what the next phase reads, then open it
            if (kx == 2) {
                if (last) break;
                s2_wait<2 * NWP>();
            } else if (!last) {
                s2_wait<NX + NWP>();
            } else if (kx == 0) {
                s2_wait<NWP>();
            } else {
                s2_wait<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
        ky = nky; cc = ncc;
    }

    // ---- epilogues: accumulator layout (a lane = one pixel, 4 x 8 consecutive channels) -> wave-private LDS strip -> line layout
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int ERS = 2 * BN + 16;
    char* const strip = smem + wave * (32 * ERS);
    const int a_off = l31 * ERS + lh * 16;
    const int l_off = (lane / LPP) * ERS + (lane % LPP) * 16;
    auto epilogue = [&](const f32x16* a, const float* tb0, bf16_t* out, float relu_lo) {
        const float* tb = tb0 + 8 * lh;
        u32x4 outv[TN * 2];
#pragma unroll
        for (int jj = 0; jj < TN * 2; ++jj) {
            const f32x4 s0 = *(const f32x4*)(tb + 16 * jj), s1 = *(const f32x4*)(tb + 16 * jj + 4);
            const f32x4 h0 = *(const f32x4*)(tb + BN + 16 * jj), h1 = *(const f32x4*)(tb + BN + 16 * jj + 4);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = a[jj >> 1][8 * (jj & 1) + e] * (e < 4 ? s0[e & 3] : s1[e & 3]) + (e < 4 ? h0[e & 3] : h1[e & 3]);
            outv[jj] = pack8_h_lo(v, relu_lo);
        }
#pragma unroll
        for (int jj = 0; jj < TN * 2; ++jj) *(u32x4*)(strip + a_off + jj * 32) = outv[jj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        u32x4 lines[NEI];
#pragma unroll
        for (int i = 0; i < NEI; ++i) lines[i] = *(const u32x4*)(strip + l_off + i * (PPI * ERS));
#pragma unroll
        for (int i = 0; i < NEI; ++i)
            if (eoff[i] >= 0) *(u32x4*)(out + eoff[i]) = lines[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    epilogue(acc, tab, (bf16_t*)p.o_hi, p.relu ? 0.f : -65504.f);
    if (has_ds) epilogue(acc2, tab + 2 * BN, (bf16_t*)p.o2_hi, -65504.f);
#endif
}

}  // namespace agp_igemm

// ps[i]: the 3x3 / stride-2 conv of problem i in the generic geometry (conv_fill_params), with w2_hi / scale2 / shift2 / o2_hi
// = its 1x1 / stride-2 downsample (or NULL); all problems share N and CK.
int agp_internal_conv_s2(agp_igemm::IgemmParams* ps, const agp_conv_desc* descs, int n, hipStream_t s) {
    using namespace agp_igemm;
    if (n < 1 || n > S2_MAXP) return AGP_E_BADARG;
    S2Group g = {};
    g.nprob = n;
    int mt = 0;
    for (int i = 0; i < n; ++i) {
        const agp_conv_desc* d = descs + i;
        IgemmParams& p = ps[i];
        if (p.N != ps[0].N || p.CK != ps[0].CK) return AGP_E_BADARG;
        // raster over the padded-width OUTPUT map: rows (img, y, x' in [0, wout + 2))
        const int wpo = d->wout + 2, wpi = d->win + 2, hpi = d->hin + 2;
        p.M = d->n * d->hout * wpo;
        p.d_howo = make_fastdiv((uint32_t)(d->hout * wpo));
        p.d_wo = make_fastdiv((uint32_t)wpo);
        p.x_sw = 2 * d->cin; p.x_sh = 2 * wpi * d->cin; p.x_sn = hpi * wpi * d->cin;
        p.sy = wpi * d->cin;                              // one input row (elements)
        p.x_base = -2 * d->cin;                           // padded input pixel (2 y + ky, 2 x' - 2) at ky = 0
        p.o_sw = d->cout; p.o_sh = wpo * d->cout; p.o_sn = (d->hout + 2) * wpo * d->cout;
        p.o_base = wpo * d->cout;                         // padded row y + 1, padded column x'
        mt += (p.M + S2_BM - 1) / S2_BM;
        g.mt_end[i] = mt;
        g.p[i] = p;
    }
    g.MT = mt;
    g.mt_chunk = (g.MT + 7) / 8;
    const int wide = AGP_TUNE("S2_WIDE", 1);            // development build, 0: 64-channel tiles for every width
    const int sch = AGP_TUNE("S2_SCHED", 1);            // development build, 0: LDS-DMA pieces at the head of a phase
    auto launch = [&](auto kern, int lds, std::atomic<uint64_t>& attr) -> int {
        if (!agp_lds_attr((const void*)kern, lds, attr)) return AGP_E_LAUNCH;
        AGP_LAUNCH(kern, dim3(g.mt_chunk * 8 * g.NT), dim3(256), lds, s, g);
        return AGP_OK;
    };
    static std::atomic<uint64_t> a4s{0}, a2s{0};
    int rc;
#if defined(AGP_TUNING)
    static std::atomic<uint64_t> a4{0}, a2{0};
    if (!sch) {
        if (wide && ps[0].N % 128 == 0) { g.NT = ps[0].N / 128; rc = launch(igemm_s2_kernel<4, false>, s2_lds<4>(), a4); }
        else { g.NT = (ps[0].N + 63) / 64; rc = launch(igemm_s2_kernel<2, false>, s2_lds<2>(), a2); }
    } else
#endif
    if (wide && ps[0].N % 128 == 0) {
        g.NT = ps[0].N / 128;
        rc = launch(igemm_s2_kernel<4, true>, s2_lds<4>(), a4s);
    } else {
        g.NT = (ps[0].N + 63) / 64;
        rc = launch(igemm_s2_kernel<2, true>, s2_lds<2>(), a2s);
    }
    (void)sch;
    if (rc != AGP_OK) return rc;
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
