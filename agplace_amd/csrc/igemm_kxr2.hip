// igemm_kxr2.hip -- the inference hot kernel: 3x3 / stride 1 / pad 1 convolution on fp16 maps with ONE fp16 MFMA
// product (AGP_PREC_F16), fused folded-BatchNorm / residual / ReLU epilogue.
//
// Same implicit GEMM as igemm_kxr.hip (GEMM rows run over the PADDED-WIDTH raster, so the three horizontal taps are
// one staged X row block read at row offsets 0, 1, 2; 256 x 64 tiles, four waves of 64 x 64, direct register ->
// global epilogue through permuted W rows), rebuilt around what the round-2 ablations showed (profiles/README.md,
// "round 2"): with staging and epilogue removed the loop runs at the rate of the best known gfx950 GEMM loops, and
// the LDS-DMA staging costs 25 % although its bytes are far from any bandwidth limit -- every barrier waited
// (`vmcnt(0)`) for loads issued ONE phase earlier, i.e. for an L2 / MALL round trip that a ~0.6 us phase does not
// cover.  Here
//   * every load is issued TWO (W taps, 3-slot ring) or THREE (X row block, double buffer) phases before its first
//     use, the barriers are raw `s_barrier`s and each is preceded by a COUNTED `s_waitcnt vmcnt(N)` that retires
//     exactly the loads the next phase reads (the count per phase is derived below; every wave issues the same
//     number of LDS-DMA instructions per phase, surplus X pieces re-issue the block's last piece);
//   * one launch serves up to four PROBLEMS of the same channel shape (`Kxr2Group`): the query network's and the
//     database network's conv of one layer run as one grid -- the 64-tile database launches of a step were one-wave
//     launches at a third of the big launches' rate -- and a conv on a small map no longer pays a launch of its own;
//   * residual reads are unconditional (clamped address) so that the in-flight count does not depend on the data.
//
// vmcnt bookkeeping (per wave, in issue order; NX = X pieces per wave, W = one piece per wave and tap):
//     prologue          : X(0)[NX]  W(0,0)  W(0,1)
//     phase (st,0)      : W(st,2)   X(st+1)[NX]
//     phase (st,1)      : W(st+1,0)
//     phase (st,2)      : W(st+1,1)
//   before the barrier that opens (st,1): W(st,1) must have landed; younger: W(st,2), X(st+1)        -> vmcnt(NX+1)
//   before the barrier that opens (st,2): W(st,2);                 younger: X(st+1), W(st+1,0)       -> vmcnt(NX+1)
//   before the barrier that opens (st+1,0): W(st+1,0) and X(st+1); younger: W(st+1,1)                -> vmcnt(1)
//   last macro-step L (no X(L+1), no W beyond it; the residual reads R[NR] are issued in (L,1)):
//     opens (L,1): W(L,1); younger: W(L,2) -> vmcnt(1);   opens (L,2): W(L,2); younger: R -> vmcnt(NR or 0).
// W(st,kx) lives in ring slot kx (a phase index is 3 st + kx and the ring has 3 slots); the slot written in phase p
// was last read in phase p-1, whose reads every wave has retired (lgkmcnt(0)) before the barrier that opens p.

#include <type_traits>

#include "igemm_params.hpp"

// -DAGP_CENSUS=1: per-workgroup time stamps (100 MHz) into the buffer of tools/census.py (p.gmin, dbg bit 0x1000000)
#ifndef AGP_CENSUS
#define AGP_CENSUS 0
#endif
#if AGP_CENSUS
#define KXR2_STAMP(i) do { if (census && tid == 0) rec[4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KXR2_STAMP(i) do { } while (0)
#endif

namespace agp_igemm {

__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }   // XOR-swizzle of a row's four 16-byte chunks
// the 16x16x32 fragment shape (16 consecutive rows x one chunk per 16 lanes): X rows / permuted W rows
__device__ __forceinline__ int swz16x(int row) { return (row >> 1) & 2; }
__device__ __forceinline__ int swz16w(int row) { return (row >> 3) & 2; }

constexpr int KXR2_MAXP = 4;
struct Kxr2Group {
    IgemmParams p[KXR2_MAXP];
    int mt_end[KXR2_MAXP];      // cumulative row tiles: problem i owns global row tiles [mt_end[i-1], mt_end[i])
    int nprob, MT, NT, mt_chunk;
};

template <int N> __device__ __forceinline__ void wait_vm_lgkm() {
    // retire all but the N youngest vector-memory operations of this wave, and all of its LDS reads
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else static_assert(N < 0, "add the count");
}

template <int BM, bool PF = false>
constexpr int kxr2_lds_bytes() { return (PF ? 3 : 2) * (BM + 16) * 64 + (PF ? 4 : 3) * 64 * 64 + 2 * 64 * 4; }

// BM x 64 tile, four waves (wave w: rows 32 TM w .. ), TM x 2 MFMA tiles of 32 x 32 per wave.
// PF (fragment prefetch): the MFMA fragments of phase p+1 are read from LDS DURING phase p into a second register set,
// so a phase opens with its MFMAs instead of an LDS round trip.  The census (tools/census2.py) shows what bounds these
// kernels: a workgroup needs ~1000 cycles per phase for 256 cycles of MFMA work per SIMD -- barrier, fragment-read latency,
// MFMAs, second read latency, MFMAs, in series -- and a CU's throughput is (resident workgroups) / (that latency).
// Reading a phase ahead needs the data a phase earlier: three X buffers and a four-slot W ring (69 KB, two workgroups
// per CU), every load issued three phases (W) or five to six (X) ahead.  Bookkeeping (per wave, issue order):
//     prologue     : X(0)[NX] W(0,0) W(0,1) X(1)[NX] W(0,2);      wait vmcnt(NX+1), barrier, read fragments of (0,0)
//     phase (st,0) : reads of (st,1);   issues W(st+1,0) X(st+2)[NX]
//     phase (st,1) : reads of (st,2);   issues W(st+1,1)
//     phase (st,2) : reads of (st+1,0); issues W(st+1,2)
//   the wait that closes phase p retires what phase p+1 READS, i.e. the operands of phase p+2:
//     closes (st,0): W(st,2);           younger: W(st+1,0) X(st+2)      -> vmcnt(NX+1)   [st = L-1: 1;  st = L: 0]
//     closes (st,1): W(st+1,0) X(st+1); younger: X(st+2) W(st+1,1)      -> vmcnt(NX+1)   [st = L-1: 1;  st = L: none]
//     closes (st,2): W(st+1,1);         younger: W(st+1,2)              -> vmcnt(1)      [st = L: none]
//   W(p) lives in ring slot p & 3 (p = 3 st + kx), X(st) in buffer st % 3; the slot / buffer a phase writes was last
//   READ two phases earlier (its reads retired by lgkmcnt(0) before the barrier in between).
// POOL: problems with IgemmParams::pool_partial also reduce the map they store for the global pooling behind it (GeM /
// average pool of a stage output): per 64-row wave block and channel, the sum of the stored values and of max(x, eps)^p
// over the block's pixels.  Such a problem's raster gives every image a multiple of 64 rows (img_rows real ones, the rest
// dead: computed, never stored), so a wave block lies inside ONE image at an image-relative position: the sums do not
// depend on where in the batch an image sits (bit-identical under batch permutation / splitting).  The values are read
// back from the epilogue's LDS strip (the fp16 bits that go to memory), one channel per lane, in pixel order.
// M16: the products run on v_mfma_f32_16x16x32_f16 (16 per wave and phase on 4 x 4 tiles of 16 pixels x 16 channels) instead
// of v_mfma_f32_32x32x16_f16 (8 on 2 x 2 tiles of 32 x 32): the same cycles, LDS fragment reads and registers, but the chip
// holds a higher clock on this shape under load (MI355X_MICROARCH.md, DVFS give-back item 7).  A lane's accumulators are 16
// consecutive channels of its pixel (W rows permuted accordingly); the LDS images are XOR-swizzled for this fragment shape
// (X: chunk ^ 2 * bit 2 of the row, W: chunk ^ 2 * bit 4 of the row: conflict-free ds_read_b128 for rows 16 apart in a tile).
// NW_ = 8 (round 3 experiment, AGP_KXR2_VARIANT=8): 512-row tiles on EIGHT waves (512 threads), two workgroups per CU = four waves
// per SIMD instead of three: the X block and the W ring of a workgroup then serve twice the MFMAs (10 KB of LDS per wave instead
// of 12), which is what buys the fourth wave.  Needs <= 128 VGPRs: the residual is loaded in the epilogue, not prefetched.  Every
// wave issues the W piece of its (wave & 3) row block -- waves 4..7 duplicate waves 0..3's 4 KB -- so that all waves count the same
// number of LDS-DMA instructions per phase.
// SCH (round 3, the default of the production instantiation; AGP_KXR2_SCHED=0 turns it off): a phase's LDS-DMA pieces are issued
// AMONG its MFMAs -- one behind each -- instead of in front of them (an LDS-DMA instruction costs ~60 cycles of issue between
// bare MFMAs against 100-185 at the head of a phase, and in front of the MFMAs that time sits on the wave's chain); all fragment
// reads of the phase come first (an LDS-DMA write may not pass an LDS read in program order), the last macro-step is peeled so
// that the loop body has no branch (one scheduling region per phase).
template <int BM, int MINB, bool PF = false, bool POOL = false, bool M16 = false, int NW_ = 4, bool SCH = false>
__global__ void __launch_bounds__(NW_ * 64, MINB) igemm_kxr2_kernel(Kxr2Group g) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!SCH || (!PF && !M16 && NW_ == 4), "scheduled variant: the plain loop of the four-wave 32x32x16 kernel");
    static_assert(!(M16 && PF), "the 16x16x32 variant is built without the fragment-prefetch pipeline");
    static_assert(!M16 || BM == 256, "16x16x32 variant: 256-row tiles");
    static_assert(NW_ == 4 || (NW_ == 8 && !PF && !M16), "eight-wave variant: plain loop only");
    constexpr int BN = 64, NW = NW_, TM = BM / (NW * 32), TN = 2;
    constexpr bool RPF = NW == 4;                  // residual prefetched during the last macro-step (costs 32 VGPRs in the loop)
    constexpr int BMX = BM + 16, ROWB = 64;
    constexpr int X_BUF = BMX * ROWB, W_TAP = BN * ROWB;
    constexpr int XINS = BMX / 16;                 // LDS-DMA pieces (16 rows x 64 B) per X block
    constexpr int NX = (XINS + NW - 1) / NW;       // per wave; pieces beyond XINS re-issue the last one
    constexpr int TMP = RPF ? (TM < 2 ? TM : 2) : 1;   // tile rows whose residual is loaded per round (prefetched during the last macro-step when RPF)
    constexpr int NR = RPF ? TMP * TN * 2 : 0;     // prefetched residual reads per lane (16 bytes each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ws = smem + (PF ? 3 : 2) * X_BUF;
    float* const tab = (float*)(ws + (PF ? 4 : 3) * W_TAP);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- tile -> (problem, row tile, column tile); XCD x owns a contiguous chunk of the global row tiles
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    // the group header in one go (unconditional reads: the compiler fetches them with one wide scalar load instead of a
    // chain of dependent single-dword round trips)
    const int gNT = g.NT, gchunk = g.mt_chunk, gMT = g.MT, gnprob = g.nprob;
    const int e0 = g.mt_end[0], e1 = g.mt_end[1], e2 = g.mt_end[2];
    const int nt = j % gNT;
    int mt = xcd * gchunk + j / gNT;
    if (mt >= gMT) return;
    int pid = 0, base = 0;
    if (gnprob > 1 && mt >= e0) { pid = 1; base = e0; }
    if (gnprob > 2 && mt >= e1) { pid = 2; base = e1; }
    if (gnprob > 3 && mt >= e2) { pid = 3; base = e2; }
    mt -= base;
    const IgemmParams& p = g.p[pid];
    const int m0 = mt * BM, n0 = nt * BN;
#if AGP_CENSUS
    const bool census = (p.dbg & 0x1000000) != 0 && p.gmin;
    unsigned long long* rec = (unsigned long long*)p.gmin + (size_t)bid * 64;
    if (census && tid == 0) {
        rec[0] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        rec[1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        rec[2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // every kernel argument the prologue uses, loaded ONCE (the fields sit behind a run-time problem index: left to itself
    // the compiler re-loads them from the kernarg segment inside the unrolled address loops, ~30 scalar-load round trips
    // of ~0.1 us each per tile -- census: 3 us of prologue before the first MFMA)
    const FastDiv d_howo = p.d_howo, d_wo = p.d_wo;
    const int pM = p.M, pN = p.N, pKtot = p.Ktot;
    const uint32_t pR = (uint32_t)p.img_rows;       // real raster rows per image (< d_howo.d when the raster is padded for pooling)
    const int x_sn = p.x_sn, x_sh_ = p.x_sh, x_sw = p.x_sw, x_base = p.x_base;
    const int o_sn = p.o_sn, o_sw = p.o_sw, o_base = p.o_base;
    const float* const pscale = p.scale;
    const float* const pshift = p.shift;

    // ---- LDS-DMA source offsets (bytes).  X piece i covers LDS rows 16 i .. 16 i + 15 = GEMM rows m0 + 16 i ..
    const int lrow = lane >> 2, lpos = lane & 3;
    int xoff[NX], woff;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
        int ins = wave + NW * q;
        ins = ins < XINS ? ins : XINS - 1;
        const int row = ins * 16 + lrow;
        // NOT clamped to M-1: rows past the last image read zeros (buffer range check), they are the kx = 1, 2
        // neighbours of the last rows
        const int m = m0 + row;
        const uint32_t img = fdiv((uint32_t)m, d_howo);
        const uint32_t rem = (uint32_t)m - img * d_howo.d;
        const uint32_t y = fdiv(rem, d_wo);
        const uint32_t xq = rem - y * d_wo.d;
        const int el = (int)img * x_sn + (int)y * x_sh_ + (int)xq * x_sw + x_base;
        xoff[q] = el * 2 + ((lpos ^ (M16 ? swz16x(row) : swz32(row))) << 4);
    }
    {
        const int row = (wave & 3) * 16 + lrow;
        int n = n0 + row;
        n = n < pN ? n : pN - 1;
        woff = (p.w_cm ? n * 64 : n * pKtot * 2) + ((lpos ^ (M16 ? swz16w(row) : swz32(row))) << 4);   // chunk-major W: [Ktot/32][N][32]
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_cm ? p.w_cm : p.w_hi), 0, p.w_bytes, 0x00020000);

    // loop-invariant scalars (kept in SGPRs: no kernarg reload inside the loop)
    const int CK = __builtin_amdgcn_readfirstlane(p.CK), x_sh = __builtin_amdgcn_readfirstlane(x_sh_);
    const int cchunks = CK / 32;
    const int nsteps = 3 * cchunks;                 // (ky, 32-channel chunk) macro-steps
    const int tapb = CK * 2;                        // bytes between consecutive kx taps
    // scale / shift of the tile's 64 channels: the loads are issued BEFORE the first LDS-DMA (older in the vmcnt order)
    float tab_s = 1.f, tab_t = 0.f;
    if (tid < BN) {
        const int n = n0 + tid < pN ? n0 + tid : pN - 1;
        if (pscale) tab_s = pscale[n];
        if (pshift) tab_t = pshift[n];
    }
    auto load_x = [&](int buf, int ky_, int cc_) {
        const int xs = __builtin_amdgcn_readfirstlane((ky_ * x_sh + cc_ * 32) * 2);
        char* base = smem + buf * X_BUF;
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            int ins = wave + NW * q;
            ins = ins < XINS ? ins : XINS - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(base + ins * 1024), 16, xoff[q], xs, 0, 0);
        }
    };
    const int wmul = __builtin_amdgcn_readfirstlane(p.w_cm ? pN : 1);      // a 64-byte K chunk is N * 64 bytes on in the chunk-major plane
    auto load_w = [&](int slot, int wbytes) {
        const int so = __builtin_amdgcn_readfirstlane(wbytes * wmul);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(ws + slot * W_TAP + (wave & 3) * 1024), 16, woff, so, 0, 0);
    };
    // the first stage is in flight while the rest of the prologue (fragment / epilogue addressing, accumulators) runs
    load_x(0, 0, 0);
    load_w(0, 0);
    load_w(1, tapb);
    if constexpr (PF) {
        __builtin_amdgcn_sched_barrier(0);          // the counts rely on this order
        load_x(1, cchunks == 1 ? 1 : 0, cchunks == 1 ? 0 : 1);
        load_w(2, 2 * tapb);
    }

    // ---- fragment read offsets.  The swizzle term of a row depends on (row mod 16) only.
    const int l31 = lane & 31, lh = lane >> 5;
    int xrd[3][2];                                  // [kx][ks]: byte offset of tile row 0, K-step ks
    int wrd[2];                                     // [ks]: byte offset of column tile 0 inside a ring slot
    int xrd16[3], wrd16[4];                         // M16: [kx] byte offset of pixel tile 0; [ct] byte offset of channel tile ct
    if constexpr (!M16) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int r = wave * (TM * 32) + l31 + kx;
                xrd[kx][ks] = r * ROWB + (((2 * ks + lh) ^ swz32(r)) << 4);
            }
        // DIRECT epilogue: W rows permuted (bits 2 and 3 swapped) so that accumulator registers 8h .. 8h+7 of a lane
        // are 8 consecutive channels of its pixel
        const int wrow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wrd[ks] = wrow * ROWB + (((2 * ks + lh) ^ swz32(wrow)) << 4);
    } else {
        const int a = lane & 15, q = lane >> 4;     // fragment row, 16-byte K chunk
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // pixel tiles are 16 rows apart: the swizzle term (bit 2 of the row) is the same for all four
            const int r = wave * 64 + a + kx;
            xrd16[kx] = r * ROWB + ((q ^ swz16x(r)) << 4);
        }
        // channel tile ct, fragment row a holds channel 16 (a >> 2) + 4 ct + (a & 3) of the tile's 64: accumulator registers
        // j = 0..3 of the four channel tiles of a lane (quad q = lane >> 4) are then channels 16 q .. 16 q + 15 of its pixel
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int wrow = 16 * (a >> 2) + 4 * ct + (a & 3);
            wrd16[ct] = wrow * ROWB + ((q ^ swz16w(wrow)) << 4);
        }
    }

    f32x16 acc[TN][TM];
    f32x4 acc16[4][4];                              // M16: [channel tile][pixel tile]
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue addressing, LINE layout: in store / residual-load instruction i (0..3) of tile row tm a lane handles
    // pixel 8 i + (lane >> 3) of the 32 and 16-byte chunk (lane & 7) of the tile's 128-byte channel segment, so that one
    // wave instruction touches 8 full 128-byte lines.  (The accumulator layout -- a pixel per lane, 32-byte pieces of 32
    // different lines per instruction -- costs four times the line accesses; the vector-memory path's line rate, not
    // bytes, is what bounds these kernels: profiles/README.md, round 2, "TA model".)
    int eoff[TM * 4];                               // element offset of the lane's chunk, -1: not stored (halo column / past M)
    const bf16_t* const rhi = (const bf16_t*)p.r_hi;
    {
        const uint32_t wlast = d_wo.d - 1;
        const int img_extra = o_sn - (int)d_howo.d * o_sw;      // o_sn - H Wp Cout: the two halo rows of an image
#pragma unroll
        for (int q = 0; q < TM * 4; ++q) {
            const int m = m0 + wave * (TM * 32) + (q >> 2) * 32 + (q & 3) * 8 + (lane >> 3);
            const uint32_t mm = (uint32_t)(m < pM ? m : pM - 1);
            const uint32_t img = fdiv(mm, d_howo);
            const uint32_t rem = mm - img * d_howo.d;
            const uint32_t y = fdiv(rem, d_wo);
            const uint32_t xq = rem - y * d_wo.d;
            const bool ok = (m < pM) && rem < pR && xq != 0 && xq != wlast;        // halo columns keep their zeros
            eoff[q] = ok ? (int)mm * o_sw + (int)img * img_extra + o_base + n0 + 8 * (lane & 7) : -1;
        }
    }

    // conv-epilogue pooling: which of a tile row's 32 pixels count (a real row of its image, not a halo column) --
    // wave-uniform bit masks
    uint32_t pmask[TM] = {};
    float* const ppart = POOL ? p.pool_partial : nullptr;
    if constexpr (POOL) {
        static_assert(BM == NW * 64, "pooling blocks are the 64-row blocks of the waves");
        if (ppart) {
            const uint32_t wlast = d_wo.d - 1;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int m = m0 + wave * (TM * 32) + tm * 32 + (lane & 31);
                const uint32_t mm = (uint32_t)(m < pM ? m : pM - 1);
                const uint32_t img = fdiv(mm, d_howo);
                const uint32_t rem = mm - img * d_howo.d;
                const uint32_t y = fdiv(rem, d_wo);
                const uint32_t xq = rem - y * d_wo.d;
                pmask[tm] = (uint32_t)__builtin_amdgcn_ballot_w64((m < pM) && rem < pR && xq != 0 && xq != wlast);
            }
        }
    }

    u32x4 rpf[TMP * TN * 2];
    auto load_residual = [&](int tm0) {             // tile rows tm0 .. tm0 + TMP - 1, line layout
#pragma unroll
        for (int t = 0; t < TMP; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {           // every lane loads (element 0 when its pixel is not stored): fixed vmcnt
                const int off = eoff[(tm0 + t) * 4 + i];
                rpf[t * 4 + i] = *(const u32x4*)(rhi + (off >= 0 ? off : 0));
            }
    };
    auto prefetch_residual = [&]() { load_residual(0); };

    int ky = 0, cc = 0;
    KXR2_STAMP(0);                                  // prologue arithmetic done, first stage in flight
    if constexpr (PF) {
        wait_vm_lgkm<NX + 1>();
        __builtin_amdgcn_s_barrier();
        KXR2_STAMP(1);
        const int L = nsteps - 1;
        bf16x8 fx[2][2][TM], fw[2][2][TN];          // [register set][K-step][tile]
        auto read_frags = [&](auto setc, auto kxc, int st_) {
            constexpr int S = decltype(setc)::value, KX = decltype(kxc)::value;
            const char* xb = smem + (st_ % 3) * X_BUF;
            const char* wb = ws + ((3 * st_ + KX) & 3) * W_TAP;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int t = 0; t < TM; ++t) fx[S][ks][t] = *(const bf16x8*)(xb + xrd[KX][ks] + t * (32 * ROWB));
#pragma unroll
                for (int t = 0; t < TN; ++t) fw[S][ks][t] = *(const bf16x8*)(wb + wrd[ks] + t * (32 * ROWB));
            }
        };
        read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
        // one macro-step = three phases; P0 = register set of its first phase (macro-steps alternate 0, 1)
        auto macro_step = [&](auto p0c, int st, int ky_, int cc_) {
            constexpr int P0 = decltype(p0c)::value;
            int nky = ky_, ncc = cc_ + 1;
            if (ncc == cchunks) { ncc = 0; ++nky; }
            int n2ky = nky, n2cc = ncc + 1;
            if (n2cc == cchunks) { n2cc = 0; ++n2ky; }
            const int wnext = (nky * 3 * CK + ncc * 32) * 2;
            const bool has1 = st < L, has2 = st + 2 <= L;
            // ---- phase (st,0)
            if (true) {
                read_frags(std::integral_constant<int, P0 ^ 1>{}, std::integral_constant<int, 1>{}, st);
                if (has1) load_w((3 * st + 3) & 3, wnext);
                __builtin_amdgcn_sched_barrier(0);
                if (has2) load_x((st + 2) % 3, n2ky, n2cc);
                if (st == 0 && tid < BN) { tab[tid] = tab_s; tab[BN + tid] = tab_t; }
                __builtin_amdgcn_sched_barrier(0);      // reads and loads first, then the MFMAs (whose operands were read a phase ago)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fw[P0][ks][tn]),
                                                                                 __builtin_bit_cast(f16x8, fx[P0][ks][tm]), acc[tn][tm], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // the MFMAs stay in their phase (hipcc moves register-only MFMAs across waits and barriers)
                if (has2) wait_vm_lgkm<NX + 1>();
                else if (has1) wait_vm_lgkm<1>();
                else wait_vm_lgkm<0>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- phase (st,1)
            if (true) {
                read_frags(std::integral_constant<int, P0>{}, std::integral_constant<int, 2>{}, st);
                if (has1) load_w((3 * st + 4) & 3, wnext + tapb);
                else if (rhi) prefetch_residual();
                __builtin_amdgcn_sched_barrier(0);      // reads and loads first, then the MFMAs (whose operands were read a phase ago)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fw[P0 ^ 1][ks][tn]),
                                                                                 __builtin_bit_cast(f16x8, fx[P0 ^ 1][ks][tm]), acc[tn][tm], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (has2) wait_vm_lgkm<NX + 1>();
                else if (has1) wait_vm_lgkm<1>();
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // last macro-step: nothing left to land
                if (has1) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- phase (st,2)
            if (true) {
                if (has1) {
                    read_frags(std::integral_constant<int, P0 ^ 1>{}, std::integral_constant<int, 0>{}, st + 1);
                    load_w((3 * st + 5) & 3, wnext + 2 * tapb);
                }
                __builtin_amdgcn_sched_barrier(0);      // reads and loads first, then the MFMAs (whose operands were read a phase ago)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fw[P0][ks][tn]),
                                                                                 __builtin_bit_cast(f16x8, fx[P0][ks][tm]), acc[tn][tm], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (has1) {
                    wait_vm_lgkm<1>();
                    __builtin_amdgcn_s_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int st = 0;
        for (; st + 1 < nsteps; st += 2) {
            macro_step(std::integral_constant<int, 0>{}, st, ky, cc);
            if (++cc == cchunks) { cc = 0; ++ky; }
            macro_step(std::integral_constant<int, 1>{}, st + 1, ky, cc);
            if (++cc == cchunks) { cc = 0; ++ky; }
        }
        if (st < nsteps) macro_step(std::integral_constant<int, 0>{}, st, ky, cc);
    } else if constexpr (SCH) {
        if (tid < BN) { tab[tid] = tab_s; tab[BN + tid] = tab_t; }
        wait_vm_lgkm<1>();
        __builtin_amdgcn_s_barrier();
        auto phase = [&](auto KX, auto LAST, const char* xb, int st_, int nky_, int ncc_, int wcur_, int wnext_) {
            constexpr int kx = decltype(KX)::value;
            constexpr bool last = decltype(LAST)::value;
            const char* wb = ws + kx * W_TAP;
            bf16x8 xf[2][TM], wf[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int t = 0; t < TM; ++t) xf[ks][t] = *(const bf16x8*)(xb + xrd[kx][ks] + t * (32 * ROWB));
#pragma unroll
                for (int t = 0; t < TN; ++t) wf[ks][t] = *(const bf16x8*)(wb + wrd[ks] + t * (32 * ROWB));
            }
            constexpr int ndma = kx == 0 ? (last ? 1 : 1 + NX) : (last ? 0 : 1);
            // piece i of this phase's LDS-DMA list in the order the vmcnt counts assume: the W piece, then X(st + 1)
            auto piece = [&](int i) {
                if (kx == 0) {
                    if (i == 0) {
                        load_w(2, wcur_ + 2 * tapb);
                    } else {
                        const int q = i - 1;
                        const int xs = __builtin_amdgcn_readfirstlane((nky_ * x_sh + ncc_ * 32) * 2);
                        int ins = wave + NW * q;
                        ins = ins < XINS ? ins : XINS - 1;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(smem + ((st_ + 1) & 1) * X_BUF + ins * 1024), 16, xoff[q], xs, 0, 0);
                    }
                } else {
                    load_w(kx - 1, wnext_ + (kx - 1) * tapb);
                }
            };
            int ip = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[ks][tn]),
                                                                             __builtin_bit_cast(f16x8, xf[ks][tm]), acc[tn][tm], 0, 0, 0);
                        if (ip < ndma) { piece(ip); ++ip; }
                    }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
#pragma unroll
            for (int i = 0; i < 2 * TM * TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < ndma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        for (int st = 0; st < nsteps - 1; ++st) {
            int nky = ky, ncc = cc + 1;
            if (ncc == cchunks) { ncc = 0; ++nky; }
            const int wcur = (ky * 3 * CK + cc * 32) * 2, wnext = (nky * 3 * CK + ncc * 32) * 2;
            const char* xb = smem + (st & 1) * X_BUF;
            phase(I0{}, std::false_type{}, xb, st, nky, ncc, wcur, wnext);
            wait_vm_lgkm<NX + 1>();
            __builtin_amdgcn_s_barrier();
            phase(I1{}, std::false_type{}, xb, st, nky, ncc, wcur, wnext);
            wait_vm_lgkm<NX + 1>();
            __builtin_amdgcn_s_barrier();
            phase(I2{}, std::false_type{}, xb, st, nky, ncc, wcur, wnext);
            wait_vm_lgkm<1>();
            __builtin_amdgcn_s_barrier();
            ky = nky; cc = ncc;
        }
        {
            const int st = nsteps - 1;
            const int wcur = (ky * 3 * CK + cc * 32) * 2;
            const char* xb = smem + (st & 1) * X_BUF;
            phase(I0{}, std::true_type{}, xb, st, 0, 0, wcur, 0);
            wait_vm_lgkm<1>();
            __builtin_amdgcn_s_barrier();
            if (rhi && RPF) prefetch_residual();
            phase(I1{}, std::true_type{}, xb, st, 0, 0, wcur, 0);
            if (rhi && RPF) wait_vm_lgkm<NR>();     // (L,1): W(L,2) must have landed; younger: the residual reads
            else wait_vm_lgkm<0>();
            __builtin_amdgcn_s_barrier();
            phase(I2{}, std::true_type{}, xb, st, 0, 0, wcur, 0);
        }
    } else {
    wait_vm_lgkm<1>();
    __builtin_amdgcn_s_barrier();
    KXR2_STAMP(1);                                  // first stage landed
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
            bf16x8 xg[4], wg[4];                    // M16: one K = 32 fragment per pixel / channel tile
            if constexpr (!M16) {
#pragma unroll
                for (int t = 0; t < TM; ++t) xf[0][t] = *(const bf16x8*)(xb + xrd[kx][0] + t * (32 * ROWB));
#pragma unroll
                for (int t = 0; t < TN; ++t) wf[0][t] = *(const bf16x8*)(wb + wrd[0] + t * (32 * ROWB));
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) wg[t] = *(const bf16x8*)(wb + wrd16[t]);
#pragma unroll
                for (int t = 0; t < 2; ++t) xg[t] = *(const bf16x8*)(xb + xrd16[kx] + t * (16 * ROWB));
            }
            // ---- this phase's loads (behind the first fragment reads, off the path to the first MFMA)
            if (kx == 0) {
                load_w(2, wcur + 2 * tapb);
                if (!last) load_x((st + 1) & 1, nky, ncc);
                if (st == 0 && tid < BN) {          // scale / shift table: its loads were issued first thing and have landed by
                    tab[tid] = tab_s;               // now (census: writing it in the prologue cost 1-2 us of load latency per tile);
                    tab[BN + tid] = tab_t;          // read only in the epilogue, many barriers later
                }
            } else if (!last) {
                load_w(kx - 1, wnext + (kx - 1) * tapb);
            } else if (kx == 1 && rhi && RPF) {
                prefetch_residual();
            }
            if constexpr (!M16) {
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
            } else {
#pragma unroll
                for (int t = 2; t < 4; ++t) xg[t] = *(const bf16x8*)(xb + xrd16[kx] + t * (16 * ROWB));
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc16[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wg[ct]),
                                                                               __builtin_bit_cast(f16x8, xg[pt]), acc16[ct][pt], 0, 0, 0);
            }
            // ---- retire what the next phase reads, then open it
            if (kx == 2) {
                if (last) break;                    // the epilogue reads no staged data
                wait_vm_lgkm<1>();
            } else if (!last) {
                wait_vm_lgkm<NX + 1>();
            } else if (kx == 0) {
                wait_vm_lgkm<1>();
            } else {                                // (L,1): W(L,2) must have landed; younger: the residual reads
                if (rhi && RPF) wait_vm_lgkm<NR>();
                else wait_vm_lgkm<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
        ky = nky; cc = ncc;
    }
    }   // !PF

    KXR2_STAMP(2);                                  // main loop done
    if (p.dbg & 128) {                              // timing experiment: no epilogue at all
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[a][b][r];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) t += acc16[a][b][r];
        if (t == 1.2345e30f) ((float*)p.o_hi)[0] = t;
        return;
    }
    // ---- epilogue.  Accumulator layout (a lane = one pixel, 4 x 8 consecutive channels) <-> line layout (8 lanes = one
    // pixel's 128 bytes) through a wave-private LDS strip of 32 rows x (128 + 16) bytes: the residual goes in by lines and
    // is read back per pixel, the result goes in per pixel and leaves by lines.  The strips reuse the staging buffers,
    // hence the barrier: every wave has finished its last fragment reads.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int ERS = 144;
    char* const strip = smem + wave * (32 * ERS);
    const int a_off = l31 * ERS + lh * 16;                      // accumulator layout: chunk 2 jj + lh of the lane's pixel
    const int a_off16 = (lane & 15) * ERS + (lane >> 4) * 32;   // M16 accumulator layout: 32 bytes (16 channels) at 32 q of pixel lane & 15 (+ 16 rows for odd pixel tiles)
    const int l_off = (lane >> 3) * ERS + (lane & 7) * 16;      // line layout: + 8 i rows
    const float* tb = tab + 8 * lh;
    bf16_t* const ohi = (bf16_t*)p.o_hi;
    const float relu_lo = p.relu ? 0.f : -65504.f;
    // LDS accesses in BATCHES (all reads of a step issued before the first use): written value by value the compiler
    // serialises ~10 dependent LDS round trips per tile row -- the census showed 3.8 us of epilogue per tile, not the stores.
    // (scale / shift are re-read per tile row in two halves: holding all 64 values would cost the third wave per SIMD)
    // Residual FIRST, for every tile row, before any store is issued: loads and stores retire through ONE in-order
    // counter (vmcnt), so a wait for a residual load placed behind a tile row's stores waits for those stores' full round
    // trip (census: ~2 us per tile row).  Line layout -> strip -> accumulator layout, the prefetch registers are reused.
    static_assert(TM <= TMP || TM == 2 * TMP, "residual staging below handles one or two prefetch rounds");
    u32x4 rres[TM][TN * 2];                                     // M16: [tm][2 ph + u] = pixel tile 2 tm + ph, channels 16 q + 8 u ..
    if (rhi) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            if (!RPF || (TM > TMP && tm == TMP)) load_residual(tm);         // not prefetched / second round (512-row tiles)
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(strip + l_off + i * (8 * ERS)) = rpf[(tm % TMP) * 4 + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if constexpr (!M16) {
#pragma unroll
                for (int jj = 0; jj < TN * 2; ++jj) rres[tm][jj] = *(const u32x4*)(strip + a_off + jj * 32);
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) rres[tm][jj] = *(const u32x4*)(strip + a_off16 + (jj >> 1) * (16 * ERS) + (jj & 1) * 16);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
    float psum[2] = {0.f, 0.f};                                     // [stat]: this wave block's sums for channel n0 + lane
    const float* const ppp = POOL ? p.pool_p : nullptr;
    const float pool_pw = ppp ? ppp[0] : 1.f, pool_eps = POOL ? p.pool_eps : 0.f;
    const bool pool_cube = pool_pw == 3.f;
    const bool pool_sq = POOL && p.pool_sq;           // stat 1 = sum of squares (BatchNorm statistics), no exponent tensor
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        u32x4 outv[TN * 2];
        if constexpr (!M16) {
#pragma unroll
        for (int jp = 0; jp < TN; ++jp) {
            f32x4 sc4[2][2], sh4[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jj = 2 * jp + u;
                sc4[u][0] = *(const f32x4*)(tb + 16 * jj);
                sc4[u][1] = *(const f32x4*)(tb + 16 * jj + 4);
                sh4[u][0] = *(const f32x4*)(tb + BN + 16 * jj);
                sh4[u][1] = *(const f32x4*)(tb + BN + 16 * jj + 4);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {               // jj = 2 tn + h: channels 16 jj + 8 lh .. + 7 of the tile's 64 columns
                const int jj = 2 * jp + u;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[jj >> 1][tm][8 * (jj & 1) + e] * sc4[u][e >> 2][e & 3] + sh4[u][e >> 2][e & 3];
                if (rhi) {
                    float r[8];
                    unpack8_h(rres[tm][jj], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                outv[jj] = pack8_h_lo(v, relu_lo);         // ReLU folded into the fp16 saturation clamp (one med3 per value)
            }
        }
#pragma unroll
        for (int jj = 0; jj < TN * 2; ++jj) *(u32x4*)(strip + a_off + jj * 32) = outv[jj];
        } else {
            // a lane holds channels 16 q .. 16 q + 15 of pixel (lane & 15) of each pixel tile: channel 16 q + 4 ct + j = acc16[ct][pt][j]
            const float* tb16 = tab + 16 * (lane >> 4);
#pragma unroll
            for (int u = 0; u < 2; ++u) {               // channels 16 q + 8 u .. + 7 (scale / shift re-read per half: registers)
                f32x4 sc4[2], sh4[2];
#pragma unroll
                for (int c4 = 0; c4 < 2; ++c4) { sc4[c4] = *(const f32x4*)(tb16 + 8 * u + 4 * c4); sh4[c4] = *(const f32x4*)(tb16 + BN + 8 * u + 4 * c4); }
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc16[2 * u + (e >> 2)][2 * tm + ph][e & 3] * sc4[e >> 2][e & 3] + sh4[e >> 2][e & 3];
                    if (rhi) {
                        float r[8];
                        unpack8_h(rres[tm][2 * ph + u], r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
                    outv[2 * ph + u] = pack8_h_lo(v, relu_lo);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) *(u32x4*)(strip + a_off16 + (jj >> 1) * (16 * ERS) + (jj & 1) * 16) = outv[jj];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        u32x4 lines[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lines[i] = *(const u32x4*)(strip + l_off + i * (8 * ERS));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = eoff[tm * 4 + i];
            if (off >= 0) *(u32x4*)(ohi + off) = lines[i];
        }
        if constexpr (POOL) {
            if (ppart) {
                // the strip holds the tile row as stored: 32 pixels x 64 channels fp16; lane = channel, pixels in order
                const bf16_t* const col = (const bf16_t*)strip + lane;
                const uint32_t mk = pmask[tm];
#pragma unroll
                for (int p8 = 0; p8 < 32; p8 += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = h2f(col[(p8 + u) * (ERS / 2)]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool a = (mk >> (p8 + u)) & 1u;                                 // wave-uniform
                        psum[0] += a ? v[u] : 0.f;
                        if (pool_sq) {
                            psum[1] += a ? v[u] * v[u] : 0.f;
                        } else if (ppp) {
                            const float c = fmaxf(v[u], pool_eps);
                            const float gq = pool_cube ? c * c * c : __builtin_exp2f(pool_pw * __builtin_log2f(c));
                            psum[1] += a ? gq : 0.f;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if constexpr (POOL) {
        if (ppart && n0 + lane < pN) {
            float* o = ppart + ((size_t)(mt * NW + wave) * 2) * pN + n0 + lane;      // [block][stat][N]
            o[0] = psum[0];
            if (ppp || pool_sq) o[pN] = psum[1];
        }
    }
#if AGP_CENSUS
    if (census && tid == 0) {
        rec[4 + 3] = __builtin_amdgcn_s_memrealtime();          // epilogue issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rec[3] = __builtin_amdgcn_s_memrealtime();              // stores retired
    }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int MINB, bool PF = false, bool POOL = false, bool M16 = false, int NW = 4, bool SCH = false>
int launch_kxr2(Kxr2Group& g, hipStream_t s) {
    constexpr int lds = kxr2_lds_bytes<BM, PF>();
    static_assert(lds * (MINB * 4 / NW) <= 160 * 1024, "LDS budget of the intended workgroups per CU");
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)igemm_kxr2_kernel<BM, MINB, PF, POOL, M16, NW, SCH>, lds, attr_done)) return AGP_E_LAUNCH;
    int mt = 0;
    for (int i = 0; i < g.nprob; ++i) {
        mt += (g.p[i].M + BM - 1) / BM;
        g.mt_end[i] = mt;
    }
    g.MT = mt;
    g.NT = (g.p[0].N + 63) / 64;
    g.mt_chunk = (g.MT + 7) / 8;
    AGP_LAUNCH((igemm_kxr2_kernel<BM, MINB, PF, POOL, M16, NW, SCH>), dim3(g.mt_chunk * 8 * g.NT), dim3(NW * 64), lds, s, g);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

}  // namespace agp_igemm

int agp_internal_conv_kxrw(agp_igemm::IgemmParams* ps, int n, hipStream_t s);

// `ps[i]` arrive with the padded-width raster geometry of agp_internal_conv_kxr_geometry; all share N, CK, prec F16.
int agp_internal_conv_kxr2(agp_igemm::IgemmParams* ps, int n, hipStream_t s) {
    using namespace agp_igemm;
    if (n < 1 || n > KXR2_MAXP) return AGP_E_BADARG;
    // layers with cout % 128 == 0 run on the wide form (256 x 128 tiles, igemm_kxrw.hip)
    if (AGP_TUNE("KXR_WIDE", 1) && ps[0].N % 128 == 0 && !AGP_TUNE("KXR2_VARIANT", 0)) return agp_internal_conv_kxrw(ps, n, s);
    Kxr2Group g = {};
    g.nprob = n;
    for (int i = 0; i < n; ++i) {
        if (ps[i].N != ps[0].N || ps[i].CK != ps[0].CK) return AGP_E_BADARG;
        g.p[i] = ps[i];
    }
    bool pool = false;
    for (int i = 0; i < n; ++i) pool = pool || ps[i].pool_partial != nullptr;
#if defined(AGP_TUNING)
    if (AGP_TUNE("KXR_TALL", 0) && !pool && ps[0].N == 64 && !AGP_TUNE("KXR2_VARIANT", 0)) return agp_internal_conv_kxrw(ps, n, s);
    // experiments that were measured and NOT adopted (profiles/README.md), development build only: 512-row tiles, 8-wave
    // workgroups, the 16x16x32 form, two-slot rings, LDS-DMA pieces at the head of a phase (KXR2_SCHED = 0)
    const int var = AGP_TUNE("KXR2_VARIANT", 0);
    if (var == 8) return pool ? launch_kxr2<512, 4, false, true, false, 8>(g, s) : launch_kxr2<512, 4, false, false, false, 8>(g, s);
    if (var == 16) return pool ? launch_kxr2<256, 3, false, true, true>(g, s) : launch_kxr2<256, 3, false, false, true>(g, s);
    if (!AGP_TUNE("KXR2_SCHED", 1)) return pool ? launch_kxr2<256, 3, false, true>(g, s) : launch_kxr2<256, 3>(g, s);
    if (!pool && var == 1) return launch_kxr2<512, 2>(g, s);
    if (!pool && var == 2) return launch_kxr2<256, 2>(g, s);
    if (!pool && var == 3) return launch_kxr2<256, 2, true>(g, s);
#endif
    if (pool)                           // (agp_conv2d_pool_blocks promises this tile shape)
        return launch_kxr2<256, 3, false, true, false, 4, true>(g, s);
    return launch_kxr2<256, 3, false, false, false, 4, true>(g, s);
}
