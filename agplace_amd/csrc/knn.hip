// Exact squared-L2 kNN (replaces faiss.IndexFlatL2, reference test.py:27-32).
//
// Pass 1 (igemm.hip, EPI_GMIN): split-bf16 MFMA GEMM  acc = (-2 q) . d ; coarse distance
//         c(q,row) = |d|^2 + acc  (|q|^2 is constant per query and irrelevant for ranking);
//         the epilogue keeps only the minimum over each group of 16 database rows
//         -> gmin[group][query], never the [nq][nb] matrix.
// Pass 2: transpose to [query][group].
// Pass 3 (one workgroup per query):
//   a. radix-select the k-th smallest group minimum T_k;
//   b. every group with gmin <= T_k + 2*eps is a candidate, eps = proven bound on
//      |coarse - true|.  Proof of completeness: if T_k < d*_k - eps there would be k rows
//      (one per group) whose true distance is < d*_k, contradicting d*_k being the k-th
//      smallest true distance; and a true top-k row has coarse <= d*_k + eps <= T;
//   c. fp64 direct evaluation sum((q-d)^2) of all rows of the candidate groups from the
//      ORIGINAL fp32 vectors, rank by (distance, index) -> sorted top-k, faiss layout.

#include <type_traits>
#include <utility>

#include "common.hpp"

int agp_internal_gmin(const void* q_hi, const void* q_lo, int64_t nq, const void* db_hi,
                      const void* db_lo, const float* db_norm, int64_t nb, int64_t nb_pad, int d,
                      int prec, float* gmin, int gq_stride, hipStream_t s);

namespace agp_knn {

constexpr int KNN_TILE_ROWS = 128;   // database rows per igemm column tile (BN)
// Exact-phase entries per round (LDS: 12 bytes each; a power of two: the sort pads a round to one).  The selection kernel is a
// chain of dependent phases per query, so what it needs is QUERIES IN FLIGHT: with 4096 entries (50 KB) and 146 VGPRs three
// workgroups fit a CU and 4096 queries took 5.3 rounds of them.  2048 entries (26 KB), 8-value windows of group minima and two-row
// unrolls in the re-ranking loops (80 VGPRs, 4 dwords spilled) fit six: 0.383 -> 0.342 ms per search (four: 0.360).  A query
// with more candidates than a round holds takes more rounds, MAX_ENT_CHUNK rows at a time (a round must fit a chunk behind the
// k <= 128 running-best entries, or it would make no progress).
constexpr int MAX_ENT = 2048;
constexpr int MAX_ENT_CHUNK = 1024;
static_assert((MAX_ENT & (MAX_ENT - 1)) == 0, "the bitonic sort pads a round to the next power of two inside the entry buffer");
static_assert(MAX_ENT >= MAX_ENT_CHUNK + 128, "a round of the exact phase must be able to take one chunk");
constexpr int MAX_K = 128;

// Database planes + squared norms + the largest norm (the selection's error bound reads it at norm[nb_pad]).  A streaming pass:
// a wave takes rows in a grid-stride loop, a lane four consecutive columns (one 16-byte load, one 8-byte store per plane), and the
// largest norm leaves through ONE atomic per workgroup -- round 5's kernel issued one atomicMax per ROW on that single word: 100 000
// same-address atomics serialise in L2 at ~11 ns each, 1.14 ms for a pass that moves 150 MB.
__global__ void __launch_bounds__(256) db_prep_kernel(const float* __restrict__ xb, int64_t nb, int64_t nb_pad, int d,
                                                       bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                       float* __restrict__ norm, int f16) {
    __shared__ float s_mx[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float mx = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < nb_pad; row += nwaves) {
        float s = 0.f;
        for (int k = lane * 4; k < d; k += 256) {
            const f32x4 v = row < nb ? *(const f32x4*)(xb + row * d + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            bf16_t h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (f16) { h[e] = f2h(v[e]); l[e] = 0; }
                else split_bf16(v[e], h[e], l[e]);
                s += v[e] * v[e];
            }
            *(u32x2*)(hi + row * d + k) = u32x2{pack2(h[0], h[1]), pack2(h[2], h[3])};
            if (lo) *(u32x2*)(lo + row * d + k) = u32x2{pack2(l[0], l[1]), pack2(l[2], l[3])};
        }
        s = wave_sum(s);
        if (lane == 0) norm[row] = s;
        if (row < nb) mx = fmaxf(mx, s);
    }
    if (lane == 0) s_mx[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        if (mx > 0.f) atomicMax((unsigned int*)(norm + nb_pad), __float_as_uint(mx));      // non-negative floats order like their bits
    }
}

__global__ void zero_tail_kernel(float* p) { if (threadIdx.x < 32) p[threadIdx.x] = 0.f; }

__global__ void q_prep_kernel(const float* __restrict__ xq, int64_t n, bf16_t* __restrict__ hi,
                              bf16_t* __restrict__ lo, int f16) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        bf16_t h, l;
        if (f16) { h = f2h(-2.f * xq[i]); l = 0; }
        else split_bf16(-2.f * xq[i], h, l);
        hi[i] = h;
        lo[i] = l;
    }
}


// order-preserving float <-> uint32 keys
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---- fp16 coarse pass with the QUERY fragments resident in registers.
// The generic implicit GEMM re-stages its 256-query tile for every 128-row database tile; with
// K = d = 256 that is 4 K-steps of work per 196 KB of LDS-DMA: LDS-bound (21 % of the MFMA peak).
// Here a workgroup owns 128 queries for a whole range of database tiles: each wave keeps its 64
// queries as MFMA B-operand fragments (d/2 VGPRs) for the whole kernel and only the database rows
// stream through LDS (64-column chunks, double buffered).  Same output as EPI_GMIN:
// gmin[group][query], group = 2 * (32-row tile) + (lane >> 5).
__device__ __forceinline__ int kswz64(int row) { return (row >> 1) & 7; }
// The wait in front of the barrier that lets the NEXT LDS-DMA overwrite a ring slot: the chunk being published has landed
// (vmcnt) AND this wave's own fragment reads are complete (lgkmcnt(0)).  Without the second half the compiler schedules the last
// ds_read of a chunk in front of the barrier and its MFMA behind it: the read is still in flight when another wave's DMA -- which
// returns in a few hundred cycles when the database range sits in the XCD's L2 -- lands in that slot.  Found in round 5 (the
// bench's own parity leg): one query in ~10 % of the <= 512-query searches (coarse_f16_kernel<D, 2>: two workgroups per CU, the K
// loop unrolled across chunks) lost a neighbour once the XCD-aware order made those L2 hits the rule -- 79 of 80 calls on one box,
// none with one workgroup per CU, none with round 4's order, none in 480 calls with this wait under either order
// (tools/knn_stress.py, tests/test_gpu_knn.py::test_search_is_repeatable_across_interleaved_query_counts).
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// QW = waves along the query axis: a workgroup owns QW * 64 queries (2 * QW waves: QW query groups x 2 halves of the
// 128-row database tile).  Every database tile streams L2 -> LDS once per QUERY TILE: with 128 queries per workgroup a
// 4096 x 100k search moves 32 x 51 MB = 1.6 GB that way (the kernel ran at the L2 -> LDS rate, 0.30 ms); QW = 4 halves it.
template <int D, int QW>
__global__ void __launch_bounds__(QW * 128, (QW == 2 ? 2 : 1)) coarse_f16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ db,
                                                            const float* __restrict__ wnorm, uint32_t* __restrict__ gminT, int nq,
                                                            int nb, int nb_pad, int g_stride, int tiles_per_split, int qtiles,
                                                            int nsplits, int xcd_map) {
    // output: two planes [nq][g_stride] of uint32 keys -- plane 0 the smallest coarse distance of each 64-row group (the 64
    // rows a wave multiplies per tile: both lane halves) with the row that attains it in the low 6 bits, plane 1 (at
    // + nq * g_stride words) the second smallest
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = D / 16, KC = D / 64;
    constexpr int NW = 2 * QW, NQ = QW * 64, DI = 16 / NW;   // waves, queries per workgroup, LDS-DMA instructions per wave and stage
    constexpr int STAGE = 128 * 128;                     // 128 database rows x 64 fp16
    constexpr int FT = (QW == 2) ? 6 : 8;                // database tiles per flush of the transposed minima (LDS: 2 workgroups per CU at QW = 2)
    constexpr int GROW = FT * 2 + 1;                     // words per query row of an LDS block (2 groups per tile; +1: bank spread)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = 3;                               // LDS ring: two 64-column chunks in flight behind the one being multiplied
    float* const nrm = (float*)(smem + NST * STAGE);     // [3 slots][128] |db row|^2 of a tile (slot = tile ordinal % 3), by LDS-DMA with the tile's first chunk
    uint32_t* const gt = (uint32_t*)(smem + NST * STAGE + 2048);  // [2 planes][NQ queries][FT * 4 groups]: written [query][group] -> coalesced rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave % QW, wd = wave / QW;
    const int l31 = lane & 31, lh = lane >> 5;
    // XCD-aware order (workgroup ids go round-robin over the 8 XCDs, each with its own 4 MB L2): the workgroups that stream
    // the SAME range of database tiles -- one per query tile -- sit on ONE XCD, so a range comes from memory once per XCD that
    // owns it instead of once per query tile (a 4096 x 100k search fetched 494 MB per launch for a 51 MB database: every XCD
    // streamed the whole database for each of its two query tiles)
    int bq, bs;
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int total = qtiles * nsplits, chunk = (total + 7) >> 3;
        int L = xcd * chunk + j;                  // split-major order: XCD x owns the contiguous chunk [x * chunk, (x + 1) * chunk)
        if (!xcd_map) L = bid;                    // (development build A/B: query tiles fastest, round 4's order)
        else if (j >= chunk) return;
        if (L >= total) return;
        bs = L / qtiles;
        bq = L - bs * qtiles;
    }
    const int q0 = bq * NQ + wq * 64;

    bf16x8 qf[2][KS];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        int row = q0 + tm * 32 + l31;
        row = row < nq ? row : nq - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[tm][ks] = *(const bf16x8*)(q + (size_t)row * D + ks * 16 + 8 * lh);
    }

    const int ntiles = nb_pad / 128;
    const int t0 = bs * tiles_per_split;
    const int t1 = min(t0 + tiles_per_split, ntiles);
    if (t0 >= t1) return;
    const __amdgpu_buffer_rsrc_t rdb = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, (uint32_t)((size_t)nb_pad * D * 2), 0x00020000);
    // LDS-DMA: 16 instructions of 8 rows x 128 B per stage, 4 per wave; the 16-B chunk position is swizzled on the source side
    int woff[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) {
        const int row = (wave + NW * i) * 8 + (lane >> 3);
        woff[i] = row * D * 2 + (((lane & 7) ^ kswz64(row)) << 4);
    }
    const __amdgpu_buffer_rsrc_t rnm = __builtin_amdgcn_make_buffer_rsrc((void*)wnorm, 0, (uint32_t)((size_t)nb * 4), 0x00020000);
    // chunks are numbered linearly over this workgroup's range: chunk c = 64 columns kc = c % KC of tile t0 + c / KC (KC = 1
    // at d = 64: every chunk is a whole tile and carries its norms).  A tile's norms go to slot (tile ordinal % 3): the
    // chunk issued at step c is c + 2, i.e. up to two tiles ahead of the one whose norms are being read.  Chunks past the
    // end re-read the last tile (instruction counts stay uniform) into slots nobody reads any more.
    auto issue = [&](int st, int c) {
        const int vt = c / KC, kc = c % KC;
        const int tile = min(t0 + vt, t1 - 1);
        const int so = __builtin_amdgcn_readfirstlane((tile * 128 * D + kc * 64) * 2);
#pragma unroll
        for (int i = 0; i < DI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdb, LDS_PTR(smem + st * STAGE + (wave + NW * i) * 1024), 16, woff[i], so, 0, 0);
        if (kc == 0) {
            // the tile's 128 row norms (every wave writes the same 512 B: the instruction count per stage stays uniform);
            // rows >= nb read zeros through the descriptor and are replaced in the epilogue
            const int no = __builtin_amdgcn_readfirstlane(tile * 512);
            const int ns = __builtin_amdgcn_readfirstlane((vt % 3) * 512);
            if (lane < 32)       // 32 lanes x 16 B (an LDS-DMA lane writes at base + 16 * lane: the upper half would spill over)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rnm, LDS_PTR((char*)nrm + ns), 16, lane * 16, no, 0, 0);
        }
    };
    int aoff[2], asw[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int r = wd * 64 + tn * 32 + l31;
        aoff[tn] = r * 128;
        asw[tn] = kswz64(r);
    }
    f32x16 acc[2][2];
    const float INF = __builtin_huge_valf();
    issue(0, 0);
    issue(1, 1);
    int st = 0;                                          // ring slot of the chunk being multiplied
    int nslot = 0;                                       // norm slot of the current tile: (tile - t0) % 3
    for (int tile = t0; tile < t1; ++tile) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            // chunk (tile, kc) -- norms included, they are its last instruction -- has landed when at most the NEWER chunk's
            // instructions are outstanding (vmcnt counts in issue order; that chunk carries a norm load when it starts a tile)
            if ((kc + 1) % KC == 0) wait_vm<DI + 1>(); else wait_vm<DI>();
            __builtin_amdgcn_s_barrier();               // visible to every wave; the slot multiplied last step is free
            if (kc == 0) {
                // the accumulators start at |db row|^2 (the tile's norms landed with this chunk): dist = |d|^2 - 2 q.d
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int nl = wd * 64 + tn * 32 + 8 * qq + 4 * lh;
                        const f32x4 w4 = *(const f32x4*)(nrm + nslot * 128 + nl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float w = (tile * 128 + nl + e < nb) ? w4[e] : 3.0e38f;   // finite: its low bits get a row index
                            acc[tn][0][4 * qq + e] = w;
                            acc[tn][1][4 * qq + e] = w;
                        }
                    }
            }
            issue(st >= 1 ? st - 1 : NST - 1, (tile - t0) * KC + kc + 2);
            const char* sb = smem + st * STAGE;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                bf16x8 a[2];
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) a[tn] = *(const bf16x8*)(sb + aoff[tn] + (((2 * k4 + lh) ^ asw[tn]) << 4));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[tn]),
                                                                              __builtin_bit_cast(f16x8, qf[tm][kc * 4 + k4]),
                                                                              acc[tn][tm], 0, 0, 0);
            }
            st = st + 1 == NST ? 0 : st + 1;
        }
        nslot = nslot == 2 ? 0 : nslot + 1;
        // ---- epilogue of this database tile: smallest coarse distance of the 32 rows a lane holds (16 in each of its two
        // 32-row MFMA tiles), WHICH row it is, and the second smallest -- 3 VALU per row: the row index replaces the 6 lowest
        // mantissa bits of the distance (a perturbation of <= 2^-17 relative, far inside the coarse error bound, that makes
        // the values distinct and the minimum carry its row), then min / med3: with v1 <= v2, med3(v1, t, v2) is the new
        // second smallest whatever t is.  The two lane halves (rows 4 lh + .. of every 8) are then merged with one exchange:
        // one group = the 64 rows of the wave's half tile -- half the coarse output of 32-row groups, same pruning power
        // (the k-th smallest group minimum tracks the k-th smallest distance as long as groups >> k).
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            float v1 = INF, v2 = INF;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = __uint_as_float((__float_as_uint(acc[tn][tm][r]) & ~63u) | (uint32_t)(lh * 32 + tn * 16 + r));
                    v2 = __builtin_amdgcn_fmed3f(v1, t, v2);
                    v1 = fminf(v1, t);
                }
            const float o1 = __shfl_xor(v1, 32, 64), o2 = __shfl_xor(v2, 32, 64);
            const float m1 = fminf(v1, o1), m2 = fminf(fmaxf(v1, o1), fminf(v2, o2));
            if (lh == 0) {
                const int ql = wq * 64 + tm * 32 + l31;                 // query within the workgroup
                const int gl = ((tile - t0) % FT) * 2 + wd;             // group within the flush block
                gt[ql * GROW + gl] = __float_as_uint(m1);
                gt[(NQ + ql) * GROW + gl] = __float_as_uint(m2);
            }
        }
        // every FT tiles (and at the end) write the block out as [query][group] rows of both planes
        const int done = tile - t0 + 1;
        if (done % FT == 0 || tile + 1 == t1) {
            __syncthreads();
            const int ng = ((done - 1) % FT + 1) * 2;                   // words per query and plane in this block
            const int g0 = (t0 + (done - 1) / FT * FT) * 2;            // first group
            for (int e = tid; e < 2 * NQ * ng; e += NW * 64) {
                const int pq = e / ng, gl = e - pq * ng;                // pq = plane * NQ + query
                const int pl = pq >= NQ, ql = pq - pl * NQ;
                const int m = bq * NQ + ql;
                if (m < nq) gminT[((size_t)pl * nq + m) * g_stride + g0 + gl] = gt[pq * GROW + gl];
            }
            __syncthreads();
        }
    }
#endif
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ---- round 5: the same pass with ONE wave per SIMD and the whole 512-entry register file per wave.
// coarse_f16_kernel<D, 4> runs eight waves of 255 VGPRs: 128 hold the wave's query fragments, 64 its accumulators, nothing is left
// to read a database fragment ahead of its MFMAs (the loop is ds_read -> wait -> 2 MFMAs), and the epilogue -- 3 VALU per database
// row and query -- runs with the matrix pipe idle (counters: 0.43 MFMA-busy; MFMA cycles + VALU issue cycles + waits = the wave's
// lifetime).  Here a workgroup is FOUR waves, each with 64 of the workgroup's 256 queries and ALL 128 rows of a database tile:
//   * two accumulator sets (2 x 128 registers): the minima of tile t are taken, three or four VALU instructions behind every MFMA
//     (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles: five 4-cycle instructions hide in the gap), while tile
//     t + 1 is multiplied; `sched_group_barrier` spells that interleave out, the compiler's own clumps the MFMAs;
//   * an accumulator is never initialised: the first MFMA of a tile takes the rows' norms as its C operand straight from the
//     registers an LDS read filled (the eight-wave kernel writes 128 accumulator registers per tile with the matrix pipe idle);
//   * the database fragments of the NEXT K-step -- across chunk and tile borders -- are read while this one's MFMAs run; the
//     barrier that publishes chunk c + 1 sits in the middle of chunk c, and the four LDS-DMA pieces a wave issues per chunk
//     (a piece stalls the issuing wave for tens of cycles) go out one per K-step, five chunks ahead;
//   * the minima leave through 16-byte LDS reads and 16-byte stores (ranges start on an even tile).
// Same arithmetic per (query, row) in the same order, same groups and row tags: the output is bit-identical to
// coarse_f16_kernel<D, 4>'s.  ABL (development build): 1 = an eighth of the epilogue, 4 = no flush, 8 = no LDS-DMA after the prologue.
template <int D, int ABL = 0>
__global__ void __launch_bounds__(256, 1) coarse_f16_w4_kernel(const float* __restrict__ xq, const bf16_t* __restrict__ db,
                                                               const float* __restrict__ wnorm, uint32_t* __restrict__ gminT, int nq,
                                                               int nb, int nb_pad, int g_stride, int qtiles, int nsplits) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = D / 16, KC = D / 64;
    constexpr int NW = 4, NQ = 256, DI = 16 / NW;
    constexpr int PF = 3;                               // chunks in flight; NST = KC ring slots: chunk kc of every tile lives in slot kc,
    constexpr int STAGE = 128 * 128, NST = PF + 1, FT = 8, GROW = FT * 2 + 4;   // so every LDS address is base + immediate
    static_assert(KC == 4 && PF == 3 && DI == 4, "the step table below is written for four chunks of four K-steps, three in flight");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const nrm = (float*)(smem + NST * STAGE);                  // [4 tiles][128] row norms
    uint32_t* const gt = (uint32_t*)(smem + NST * STAGE + 2048);      // [2 planes][NQ][GROW] minima of a block of FT tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    int bq, bs;
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int total = qtiles * nsplits, chunk = (total + 7) >> 3;
        const int L = xcd * chunk + j;
        if (j >= chunk || L >= total) return;
        bs = L / qtiles;
        bq = L - bs * qtiles;
    }
    const int q0 = bq * NQ + wave * 64;

    bf16x8 qf[2][KS];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        int row = q0 + tm * 32 + l31;
        row = row < nq ? row : nq - 1;
        // the query preparation (q_prep_kernel: fp16 of -2 x) happens here, on the way into the registers
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float* p = xq + (size_t)row * D + ks * 16 + 8 * lh;
            const f32x4 x0 = *(const f32x4*)p, x1 = *(const f32x4*)(p + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                qf[tm][ks][c] = f2h(-2.f * x0[c]);
                qf[tm][ks][4 + c] = f2h(-2.f * x1[c]);
            }
        }
    }
    const int ntiles = nb_pad / 128;
    const int t0 = (int)((int64_t)bs * ntiles / nsplits);        // balanced ranges: floor or ceil of ntiles / nsplits tiles
    const int t1 = (int)((int64_t)(bs + 1) * ntiles / nsplits);
    if (t0 >= t1) return;
    const int nt = t1 - t0;
    const __amdgpu_buffer_rsrc_t rdb = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, (uint32_t)((size_t)nb_pad * D * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rnm = __builtin_amdgcn_make_buffer_rsrc((void*)wnorm, 0, (uint32_t)((size_t)nb * 4), 0x00020000);
    int woff[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) {
        const int row = (wave + NW * i) * 8 + (lane >> 3);
        woff[i] = row * D * 2 + (((lane & 7) ^ kswz64(row)) << 4);
    }
    // piece i of the wave's four of a chunk: rows (wave + 4 i) * 8 .. + 8 of tile ordinal vt (clamped: a phantom tile past the
    // range replays the last one), columns 64 kcc .. + 64, into ring slot `slot`
    auto dma_piece = [&](auto ic, int slot, int vt, int kcc) {
        constexpr int i = decltype(ic)::value;
        const int tile = min(t0 + vt, t1 - 1);
        const int so = __builtin_amdgcn_readfirstlane((tile * 128 * D + kcc * 64) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdb, LDS_PTR(smem + slot * STAGE + (wave + NW * i) * 1024), 16, woff[i], so, 0, 0);
    };
    auto dma_norms = [&](int vt) {
        const int tile = min(t0 + vt, t1 - 1);
        const int no = __builtin_amdgcn_readfirstlane(tile * 512);
        const int ns = __builtin_amdgcn_readfirstlane((vt & 3) * 512);
        if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rnm, LDS_PTR((char*)nrm + ns), 16, lane * 16, no, 0, 0);
    };
    // the ragged last tile of the database: rows past nb take a norm no real row can beat (workgroup-uniform branch)
    auto patch_norms = [&](int vt) {
        const int tile = min(t0 + vt, t1 - 1);
        if (tile * 128 + 128 > nb) {
            if (tid < 128 && tile * 128 + tid >= nb) nrm[(vt & 3) * 128 + tid] = 3.0e38f;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    };
    int aoff[4], asw[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int r = tn * 32 + l31;
        aoff[tn] = r * 128;
        asw[tn] = kswz64(r);
    }
    f32x16 acc[2][4][2];
    f32x16 W[2];                                         // norms of two 32-row blocks: the C operand of a tile's first MFMAs
    bf16x8 a[2][4];
    const float INF = __builtin_huge_valf();
    float v1 = INF, v2 = INF;                            // running minima of the epilogue piece in progress
    auto load_w = [&](f32x16& w, int vt, int tn) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 w4 = *(const f32x4*)(nrm + (vt & 3) * 128 + tn * 32 + 8 * qq + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[4 * qq + e] = w4[e];
        }
    };
    auto load_a = [&](bf16x8 (&f)[4], int slot, int kb) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) f[tn] = *(const bf16x8*)(smem + slot * STAGE + aoff[tn] + (((2 * kb + lh) ^ asw[tn]) << 4));
    };

    // slice s (0..15) of the epilogue of accumulator set B (tile ordinal it_prev): piece = s >> 2 = (tm, group), four slices of
    // 8 values each; the last slice of a piece merges the lane halves and stores the group's two keys
    auto epi_slice = [&](auto bufc, auto sc, int it_prev) {
        constexpr int B = decltype(bufc)::value, S = decltype(sc)::value;
        constexpr int piece = S >> 2, sub = S & 3;
        constexpr int tm = piece >> 1, grp = piece & 1;
        constexpr int tnl = sub >> 1, r0 = (sub & 1) * 8;
        if constexpr (sub == 0) { v1 = INF; v2 = INF; }
        // the tag's lane-half bit is the same for every value of a lane: it goes in once per piece (below), so the in-loop tag is
        // an inline constant; med3(a, b, -3.4e38) = min(a, b) for the finite values here, without the canonicalising v_max the
        // compiler puts before a v_min (and which it also re-creates from a med3 against -inf)
#pragma unroll
        for (int r = r0; r < r0 + ((ABL & 1) ? 1 : 8); ++r) {
            const float t = __uint_as_float((__float_as_uint(acc[B][grp * 2 + tnl][tm][r]) & ~63u) | (uint32_t)(tnl * 16 + r));
            v2 = __builtin_amdgcn_fmed3f(v1, t, v2);
            v1 = __builtin_amdgcn_fmed3f(v1, t, -3.4e38f);
        }
        if constexpr (sub == 3) {
            const uint32_t u1 = __float_as_uint(v1) | (uint32_t)(lh << 5), u2 = __float_as_uint(v2) | (uint32_t)(lh << 5);
            // v_permlane32_swap: [0] = the lower half's value in both halves, [1] = the upper half's
            const auto s1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
            const auto s2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
            const float a1 = __uint_as_float(s1[0]), b1 = __uint_as_float(s1[1]);
            const float a2 = __uint_as_float(s2[0]), b2 = __uint_as_float(s2[1]);
            // min / max of finite values as med3 against +-3.4e38 (no canonicalising v_max in front of them)
            const float m1 = __builtin_amdgcn_fmed3f(a1, b1, -3.4e38f), mx = __builtin_amdgcn_fmed3f(a1, b1, 3.4e38f);
            const float m2 = __builtin_amdgcn_fmed3f(mx, __builtin_amdgcn_fmed3f(a2, b2, -3.4e38f), -3.4e38f);
            // both lane halves hold both results: the lower half stores the minimum, the upper the second minimum (one
            // unmasked ds_write); the planes hold the float's bits, select_rerank makes the ordered key
            const int ql = wave * 64 + tm * 32 + l31;
            const int gl = ((it_prev + FT) % FT) * 2 + grp;
            gt[(lh * NQ + ql) * GROW + gl] = __float_as_uint(lh ? m2 : m1);
        }
    };
    // the block of FT tiles that ends with tile ordinal `last` (inclusive) goes out as [query][group] rows of both planes.  The
    // barrier is a bare one (LDS only): a __syncthreads() would drain the LDS-DMA pipeline.  No barrier behind the reads: the
    // next store into `gt` sits behind the barrier of the following tile's third K-step.
    auto flush = [&](int last) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int ng = (last % FT + 1) * 2;
        const int g0 = (t0 + last / FT * FT) * 2;
        if (ng == FT * 2 && !(g0 & 3)) {
#pragma unroll 2
            for (int e = tid; e < 2 * NQ * (FT * 2 / 4); e += NW * 64) {
                const int pq = e >> 2, g4 = (e & 3) * 4;
                const int pl = pq >= NQ, ql = pq - pl * NQ;
                const int m = bq * NQ + ql;
                const u32x4 v = *(const u32x4*)(gt + pq * GROW + g4);
                if (m < nq) *(u32x4*)(gminT + ((size_t)pl * nq + m) * g_stride + g0 + g4) = v;
            }
        } else if (ng == FT * 2) {                       // a range that starts on an odd tile: rows are 8-byte aligned
#pragma unroll 2
            for (int e = tid; e < 2 * NQ * (FT * 2 / 2); e += NW * 64) {
                const int pq = e >> 3, g2 = (e & 7) * 2;
                const int pl = pq >= NQ, ql = pq - pl * NQ;
                const int m = bq * NQ + ql;
                const u32x2 v = *(const u32x2*)(gt + pq * GROW + g2);
                if (m < nq) *(u32x2*)(gminT + ((size_t)pl * nq + m) * g_stride + g0 + g2) = v;
            }
        } else {
            for (int e = tid; e < 2 * NQ * ng; e += NW * 64) {
                const int pq = e / ng, gl = e - pq * ng;
                const int pl = pq >= NQ, ql = pq - pl * NQ;
                const int m = bq * NQ + ql;
                if (m < nq) gminT[((size_t)pl * nq + m) * g_stride + g0 + gl] = gt[pq * GROW + gl];
            }
        }
    };

    // ---- prologue: chunks 0 and 1 whole, the first two pieces of chunk 2, the norms of tile 0
    static_for<2>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        dma_piece(std::integral_constant<int, 0>{}, c, 0, c);
        if constexpr (c == 0) dma_norms(0);
        dma_piece(std::integral_constant<int, 1>{}, c, 0, c);
        dma_piece(std::integral_constant<int, 2>{}, c, 0, c);
        dma_piece(std::integral_constant<int, 3>{}, c, 0, c);
    });
    dma_piece(std::integral_constant<int, 0>{}, 2, 0, 2);
    dma_piece(std::integral_constant<int, 1>{}, 2, 0, 2);
    wait_vm<DI + 2>();                                   // chunk 0 and tile 0's norms have landed
    __builtin_amdgcn_s_barrier();
    patch_norms(0);
    load_a(a[0], 0, 0);
    load_w(W[0], 0, 0);
    load_w(W[1], 0, 1);
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][tn][tm][r] = 0.f;   // "tile -1": its minima land in a block slot tile 7 overwrites

    // K-step S = 4 kc + k4 of tile ordinal `it` into accumulator set B, with slice S of the minima of set B ^ 1 (tile it - 1).
    // No condition on `it`: an odd range ends with a phantom tile whose minima are never flushed.
    auto step = [&](auto bufc, auto sc, int it) {
        constexpr int B = decltype(bufc)::value, S = decltype(sc)::value;
        constexpr int kc = S >> 2, k4 = S & 3;
        __builtin_amdgcn_sched_barrier(0);               // keeps slice S of the epilogue (8 accumulator reads) inside step S
        if constexpr (k4 == 2) {
            // chunk c + 1 is published here; chunk c + 2 (and the norms that travel with a tile's first) may be in flight
            wait_vm<DI + (kc == 2 ? 1 : 0)>();
            __builtin_amdgcn_s_barrier();
            if constexpr (kc == 3) patch_norms(it + 1);
        }
        if constexpr (!(ABL & 8)) {
            if constexpr (k4 >= 2) {                     // chunk c + 3 -> the slot chunk c - 1 left (everybody is past the barrier)
                dma_piece(std::integral_constant<int, k4 - 2>{}, (kc + 3) & 3, it + (kc >= 1), (kc + 3) & 3);
                if constexpr (k4 == 2 && kc == 1) dma_norms(it + 1);
            } else {                                     // the other half of chunk c + 2
                dma_piece(std::integral_constant<int, k4 + 2>{}, (kc + 2) & 3, it + (kc >= 2), (kc + 2) & 3);
            }
        }
        if constexpr (k4 < 3) load_a(a[(S + 1) & 1], kc, k4 + 1);
        else load_a(a[(S + 1) & 1], (kc + 1) & 3, 0);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                acc[B][tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[S & 1][tn]),
                                                                         __builtin_bit_cast(f16x8, qf[tm][S]),
                                                                         S == 0 ? W[tn & 1] : acc[B][tn][tm], 0, 0, 0);
            if constexpr (S == 0) {
                if (tn < 2) load_w(W[tn], it, tn + 2);
            }
        }
        if constexpr (S == 15) {
            load_w(W[0], it + 1, 0);
            load_w(W[1], it + 1, 1);
        }
        epi_slice(std::integral_constant<int, B ^ 1>{}, sc, it - 1);
        // the interleave: every MFMA is followed by what hides in its 24 free issue cycles
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i == 0) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, (S == 0 || S == 15) ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
    };
    const int ne = (nt + 1) & ~1;
    for (int it = 0; it < ne; it += 2) {
        static_for<16>([&](auto sc) { step(std::integral_constant<int, 0>{}, sc, it); });
        if (!(ABL & 4) && it && it % FT == 0) flush(it - 1);          // tiles it - 8 .. it - 1 are in the LDS block now
        static_for<16>([&](auto sc) { step(std::integral_constant<int, 1>{}, sc, it + 1); });
    }
    // the minima of set 1's last tile (ordinal ne - 1 = nt - 1, or the phantom): nothing left to hide them behind
    static_for<16>([&](auto sc) { epi_slice(std::integral_constant<int, 1>{}, sc, ne - 1); });
    wait_vm<0>();                                        // the LDS-DMA issued past the range has landed before the LDS is given back
    flush(nt - 1);
#endif
}

template <int D, int ABL = 0>
int launch_coarse_f16_w4(const float* xq, const void* db, const float* wnorm, uint32_t* gminT, int64_t nq, int64_t nb, int64_t nb_pad,
                         int g_stride, hipStream_t s) {
    constexpr int NQ = 256, FT = 8;
    constexpr int lds = 4 * 128 * 128 + 2048 + 2 * NQ * (FT * 2 + 4) * 4;
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)coarse_f16_w4_kernel<D, ABL>, lds, attr_done)) return AGP_E_LAUNCH;
    const int qt = (int)((nq + NQ - 1) / NQ);
    const int ntiles = (int)(nb_pad / 128);
    int splits = (256 + qt - 1) / qt;                     // one workgroup per CU
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    AGP_LAUNCH((coarse_f16_w4_kernel<D, ABL>), dim3(8 * ((qt * splits + 7) / 8)), dim3(256), lds, s, xq, (const bf16_t*)db,
               wnorm, gminT, (int)nq, (int)nb, (int)nb_pad, g_stride, qt, splits);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

template <int D, int QW>
int launch_coarse_f16_cfg(const void* q, const void* db, const float* wnorm, uint32_t* gminT, int64_t nq, int64_t nb, int64_t nb_pad,
                          int g_stride, hipStream_t s) {
    constexpr int NQ = QW * 64;
    constexpr int FT = (QW == 2) ? 6 : 8;
    constexpr int lds = 3 * 128 * 128 + 2048 + 2 * NQ * (FT * 2 + 1) * 4;
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)coarse_f16_kernel<D, QW>, lds, attr_done)) return AGP_E_LAUNCH;
    const int qt = (int)((nq + NQ - 1) / NQ);
    const int ntiles = (int)(nb_pad / 128);
    const int target = QW == 2 ? 512 : 256;             // workgroups: two (one) per CU
    int splits = (target + qt - 1) / qt;
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    const int per = (ntiles + splits - 1) / splits;
    splits = (ntiles + per - 1) / per;
    const int xcd_map = AGP_TUNE("KNN_XCD", 1);          // development build, 0: round 4's order (query tile fastest over the XCDs)
    AGP_LAUNCH((coarse_f16_kernel<D, QW>), dim3(8 * ((qt * splits + 7) / 8)), dim3(QW * 128), lds, s, (const bf16_t*)q, (const bf16_t*)db,
               wnorm, gminT, (int)nq, (int)nb, (int)nb_pad, g_stride, per, qt, splits, xcd_map);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

// the four-wave kernel takes the fp32 queries themselves (no q_prep_kernel launch); everything else the prepared fp16 plane
// (from a dozen database tiles per workgroup on: below that its deeper pipeline fill and the phantom tile of an odd range cost
// more than the interleave saves -- 5000 rows x 777 queries: 20 us against 16)
inline bool coarse_w4_applies(int d, int64_t nq, int64_t nb_pad) {
    if (d != 256 || nq <= 512 || AGP_TUNE("KNN_QW", 4) != 4) return false;
    const int64_t qt = (nq + 255) / 256, ntiles = nb_pad / 128;
    int64_t splits = (256 + qt - 1) / qt;
    if (splits > ntiles) splits = ntiles;
    return ntiles >= 12 * splits;
}

template <int D>
int launch_coarse_f16(const float* xq, const void* q, const void* db, const float* wnorm, uint32_t* gminT, int64_t nq, int64_t nb,
                      int64_t nb_pad, int g_stride, hipStream_t s) {
    const int qw = AGP_TUNE("KNN_QW", 4);               // development build, 2: 128-query workgroups; 8: round 4's eight-wave form
    if constexpr (D == 256) {
        if (coarse_w4_applies(D, nq, nb_pad)) {
#if defined(AGP_TUNING)
            switch (AGP_TUNE("KNN_ABL", 0)) {            // bit set of the kernel's ABL (timing only, results wrong)
                case 1: return launch_coarse_f16_w4<D, 1>(xq, db, wnorm, gminT, nq, nb, nb_pad, g_stride, s);
                case 4: return launch_coarse_f16_w4<D, 4>(xq, db, wnorm, gminT, nq, nb, nb_pad, g_stride, s);
                case 8: return launch_coarse_f16_w4<D, 8>(xq, db, wnorm, gminT, nq, nb, nb_pad, g_stride, s);
                case 13: return launch_coarse_f16_w4<D, 13>(xq, db, wnorm, gminT, nq, nb, nb_pad, g_stride, s);
                default: break;
            }
#endif
            return launch_coarse_f16_w4<D>(xq, db, wnorm, gminT, nq, nb, nb_pad, g_stride, s);
        }
    }
    if ((qw == 4 || qw == 8) && nq > 512) return launch_coarse_f16_cfg<D, 4>(q, db, wnorm, gminT, nq, nb, nb_pad, g_stride, s);
    return launch_coarse_f16_cfg<D, 2>(q, db, wnorm, gminT, nq, nb, nb_pad, g_stride, s);
}

__global__ void transpose_kernel(const float* __restrict__ in, int rows, int cols, int in_stride,
                                 float* __restrict__ out, int out_stride) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * in_stride + c] : __builtin_huge_valf();
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (c < cols && r < rows) out[(size_t)c * out_stride + r] = tile[threadIdx.x][j];
    }
}

// PACKED: gminT holds two planes [nq][g_stride] over 64-row groups -- the ordered key of the group's smallest coarse distance with the row
// that attains it in the low 6 bits, and the key of the second smallest (coarse_f16_kernel).  A candidate group whose
// SECOND minimum is outside the window contributes exactly one row to the exact pass instead of all of them: the ~20
// groups inside the window of a typical query hold ~20 rows to re-evaluate instead of hundreds to gather and re-score.
// !PACKED: one float per group (generic coarse pass): every row of a candidate group is examined.
// VPT_: group minima a thread holds per window (256 * VPT_ groups): 8 when the database has <= 2048 groups (131 072 rows: the
// bench's 100k), else 32 -- 24 VGPRs less.
template <bool PACKED, int VPT_ = 32>
__global__ __launch_bounds__(256, 6) void select_rerank_kernel(
    const float* __restrict__ xq, const float* __restrict__ xb, const uint32_t* __restrict__ gminT,
    int G, int g_stride, const float* __restrict__ db_norm, int64_t nb, int64_t nb_pad, int d, int k,
    float cerr, float* __restrict__ dist, int64_t* __restrict__ idx, int dbg, const bf16_t* __restrict__ db_f16) {
    __shared__ uint32_t smin[256];
    __shared__ uint32_t s_T;
    __shared__ unsigned int s_count, s_ncand;
    __shared__ double e_d[MAX_ENT];
    __shared__ int e_i[MAX_ENT];
    __shared__ int s_nbest;

    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int GR = PACKED ? 64 : 16;                 // rows per group
    const uint32_t* gm = gminT + (size_t)q * g_stride;                                 // PACKED: plane 0 (minimum + row)
    const uint32_t* gm2 = gminT + ((size_t)gridDim.x + q) * g_stride;                  // PACKED: plane 1 (second minimum)
    // rows of group g: !PACKED 16 rows of a 32-row tile (tile = g >> 1, half h = g & 1: rows 8 (r >> 2) + 4 h + (r & 3));
    // PACKED all 64 rows of block g, indexed as the coarse kernel tags them: r = 32 h + 16 tn + r16
    auto group_row = [&](int g, int r) -> int64_t {
        const int r16 = r & 15;
        const int h = PACKED ? (r >> 5) : (g & 1);
        const int64_t t32 = PACKED ? (int64_t)g * 2 + ((r >> 4) & 1) : (int64_t)(g >> 1);
        return t32 * 32 + 8 * (r16 >> 2) + 4 * h + (r16 & 3);
    };
    const float* qv = xq + (size_t)q * d;
    const float INF = __builtin_huge_valf();
    constexpr uint32_t KMAX = 0xffffffffu;              // key of an out-of-range slot: above every value, +INF included

    // ---- a. an upper bound T' >= T_k (the k-th smallest group minimum): the kk-th smallest of the
    // 256 per-thread minima.  Each of those is one group's value, so at least kk groups are <= T',
    // hence T' >= T_k; using it keeps the candidate set a superset of the exact one.
    // Group minima are read in windows of 32 values per thread, all 32 loads issued back to back
    // (one memory round trip per window instead of one per value).
    // the query's norm and the database's largest norm feed the candidate window (step b): their loads go out FIRST, beside the
    // group minima's, instead of as one more dependent round trip behind the threshold (the selection is a chain of round trips)
    float qn = 0.f;
    for (int i = lane; i < d; i += 64) qn += qv[i] * qv[i];
    const float dmax2 = db_norm[nb_pad];
    constexpr int VPT = VPT_;
    uint32_t v[VPT];                                    // keys of the group minima (PACKED: row index in the low 4 bits)
    auto load_window = [&](int w) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int g = (w * VPT + i) * 256 + tid;
            v[i] = g < G ? fkey(__uint_as_float(gm[g])) : KMAX;
        }
    };
    const int nwin = (G + VPT * 256 - 1) / (VPT * 256);
    uint32_t mn = KMAX;
    for (int w = 0; w < nwin; ++w) {
        load_window(w);
#pragma unroll
        for (int i = 0; i < VPT; ++i) mn = v[i] < mn ? v[i] : mn;
    }
    // the kk-th smallest of the 256 minima: every wave sorts its 64 values with a shuffle-only bitonic network (21
    // compare-exchange steps), the four sorted runs go to LDS, and the candidates (the first kk of each run) find their
    // rank in the union by binary search in the other three runs -- ties ordered by (value, wave, position).  (Counting,
    // for every thread, how many of the 256 values are smaller was 1536 VALU instructions per thread: 40 us of the
    // selection's 120 us at 4096 queries.)
    {
        uint32_t sv = mn;
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)sv, stride, 64);
                const bool up = (lane & size) == 0 || size == 64;        // the last merge sorts the whole wave ascending
                const bool lower = (lane & stride) == 0;
                const uint32_t lo_ = sv < o ? sv : o, hi_ = sv < o ? o : sv;
                sv = (lower == up) ? lo_ : hi_;
            }
        smin[wave * 64 + lane] = sv;
        if (tid == 0) { s_nbest = 0; s_ncand = 0; s_T = fkey(INF); }
        __syncthreads();
        const int kk = k < G ? k : G;        // k <= 128 < 256 threads
        if (lane < kk) {
            int r = lane;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w == wave) continue;
                // number of elements of run w that sort before (sv, wave, lane): < sv, or <= sv for an earlier run
                const uint32_t* run = smin + w * 64;
                int lo_ = 0, hi_ = 64;
                while (lo_ < hi_) {
                    const int mid = (lo_ + hi_) >> 1;
                    const uint32_t o = run[mid];
                    const bool before = w < wave ? (o <= sv) : (o < sv);
                    if (before) lo_ = mid + 1; else hi_ = mid;
                }
                r += lo_;
            }
            if (r == kk - 1) s_T = sv;
        }
    }
    // ---- b. candidate threshold
    qn = sqrtf(wave_sum(qn));
    // cerr < 0 flags the fp16 coarse pass: add the underflow term (elements below 2^-14 are rounded to
    // multiples of 2^-24) and give up the pruning entirely if an operand could have saturated
    const bool f16c = cerr < 0.f;
    const float ce = fabsf(cerr);
    float eps = 2.f * ce * qn * sqrtf(dmax2) + 1.2e-7f * (float)d * dmax2 * 0.0625f + 1e-30f;
    if (f16c) {
        eps += 6e-8f * sqrtf((float)d) * (sqrtf(dmax2) + 2.f * qn) * 2.f;
        if (!(sqrtf(dmax2) < 6.0e4f) || !(2.f * qn < 6.0e4f)) eps = 3.0e38f;
    }
    __syncthreads();
    // PACKED: every stored value carries a row index in its 6 lowest mantissa bits, i.e. is off by < 2^-17 of its magnitude
    // (<= dmax2 + 2 |q| |d|max) in either direction: `pert` on the thresholds keeps the candidate set a superset
    const uint32_t sT = s_T;
    const float pert = PACKED ? 7.6293945e-6f * (dmax2 + 2.f * qn * sqrtf(dmax2)) : 0.f;
    // (fewer groups than k: the k-th smallest GROUP minimum does not exist, so no finite bound on the k-th smallest ROW)
    const float T = (sT >= fkey(INF) || G < k) ? INF : fkey_inv(sT) + 2.f * eps + 2.f * pert;
    const uint32_t Tkey = (T == INF) ? fkey(INF) : fkey(T);                      // key <= Tkey  <=>  value <= T
    // rows a candidate group contributes: all of them, or (PACKED, second minimum outside the window) the one row of its minimum
    auto group_rows = [&](int g) -> int {
        if (!PACKED) return GR;
        return fkey(__uint_as_float(gm2[g])) <= Tkey ? GR : 1;
    };

    // how many candidate ROWS are there in total?  (one window: which of this thread's candidate groups contribute ALL their rows
    // is remembered in `fullmask` -- the gather below then needs no second look at the second-minimum plane)
    uint32_t fullmask = 0u;
    {
        unsigned int c = 0;
        for (int w = 0; w < nwin; ++w) {
            if (nwin > 1) load_window(w);     // one window (G <= 8192 groups): the values of sweep (a) are still in registers
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int g = (w * VPT + i) * 256 + tid;
                if (g < G && v[i] <= Tkey) {
                    const int rows = group_rows(g);
                    if (rows == GR) fullmask |= 1u << i;
                    c += rows;
                }
            }
        }
        if (c) atomicAdd(&s_ncand, c);
    }
    __syncthreads();
    const bool one_round = (s_ncand <= (unsigned)MAX_ENT);
    const bool few = PACKED && one_round && s_ncand <= 8u * (unsigned)k;      // straight to the exact pass
    if (dbg == 1) return;

    // ---- c. exact phase in rounds of <= MAX_ENT entries
    int g_next = 0;   // uniform scan position over groups
    while (g_next < G) {
        // entries [0, nbest) carry the running best; gather candidate rows after them.
        if (tid == 0) s_count = s_nbest;
        __syncthreads();
        int g_base = g_next;
        if (one_round) {
            // common case: every candidate fits at once -> one strided sweep, no per-chunk barriers
            for (int w = 0; w < nwin; ++w) {
                if (nwin > 1) load_window(w);
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    const int g = (w * VPT + i) * 256 + tid;
                    if (g < G && v[i] <= Tkey) {
                        if (nwin == 1 ? ((fullmask >> i) & 1u) != 0u : group_rows(g) == GR) {
                            const unsigned int slot = atomicAdd(&s_count, (unsigned)GR);
#pragma unroll
                            for (int r = 0; r < GR; ++r) {
                                const int64_t n = group_row(g, r);
                                e_i[slot + r] = n < nb ? (int)n : 0x7fffffff;
                            }
                        } else {
                            const int r = (int)(((v[i] & 0x80000000u) ? v[i] : ~v[i]) & 63u);    // fkey complements negative values' bits
                            const int64_t n = group_row(g, r);
                            e_i[atomicAdd(&s_count, 1u)] = n < nb ? (int)n : 0x7fffffff;
                        }
                    }
                }
            }
            g_base = G;
            __syncthreads();
        } else {
            // degenerate inputs (many ties): CH groups (MAX_ENT_CHUNK rows) at a time, a chunk is only started while
            // it cannot overflow the entry buffer.
            constexpr int CH = MAX_ENT_CHUNK / GR;
            for (; g_base < G; g_base += CH) {
                const unsigned int cnt = s_count;
                __syncthreads();   // everyone has read cnt before anyone bumps s_count
                if (cnt + CH * GR > MAX_ENT) break;
                const int g = g_base + tid;
                if (tid < CH && g < G && fkey(__uint_as_float(gm[g])) <= Tkey) {
                    const unsigned int slot = atomicAdd(&s_count, (unsigned)GR);
#pragma unroll
                    for (int r = 0; r < GR; ++r) {
                        const int64_t n = group_row(g, r);
                        e_i[slot + r] = n < nb ? (int)n : 0x7fffffff;
                    }
                }
                __syncthreads();
            }
        }
        g_next = g_base;
        if (dbg == 2) return;
        int total = s_count;
        const int nbest0 = s_nbest;
        // ---- c'. row-level pruning (fp16 coarse pass only): only the minimum of a candidate group is
        // known to be inside the window; re-evaluate every row's COARSE distance |d|^2 - 2 q~.d~ from the
        // fp16 planes (512 B per row instead of 1 KB, fp32 math) and keep the rows with coarse <= T.
        // Any approximation within eps of the truth keeps every true top-k row (same proof as for the
        // groups), so this only removes work from the exact pass.
        if (db_f16 && T < INF && !few) {
            int* surv = (int*)(e_d + nbest0);            // e_d beyond the running best is not written before the exact pass
            if (tid == 0) s_ncand = 0;
            __syncthreads();
            // 16 lanes per row, UP rows per 16-lane group and trip: 16 * UP row gathers of the workgroup are in flight
            // per memory round trip (with one row per group the ~1000 candidate rows of a query were ~60 dependent trips)
            constexpr int UP = 2;      // (4 rows per group and trip cost 20 VGPRs more: one workgroup per CU less)
            const int sub = lane >> 4, sl = lane & 15;
            for (int e0 = nbest0 + wave * (4 * UP); e0 < total; e0 += 16 * UP) {
                int nn[UP];
                float acc[UP];
#pragma unroll
                for (int u = 0; u < UP; ++u) {
                    const int e = e0 + u * 4 + sub;
                    nn[u] = e < total ? e_i[e] : 0x7fffffff;
                    acc[u] = 0.f;
                }
                for (int i = sl * 8; i < d; i += 128) {
                    u32x4 raw[UP];
#pragma unroll
                    for (int u = 0; u < UP; ++u)
                        raw[u] = nn[u] != 0x7fffffff ? *(const u32x4*)(db_f16 + (size_t)nn[u] * d + i) : u32x4{0u, 0u, 0u, 0u};
                    const f32x4 q0 = *(const f32x4*)(qv + i), q1 = *(const f32x4*)(qv + i + 4);
                    float qq[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
                    for (int c = 0; c < 8; ++c) qq[c] = h2f(f2h(-2.f * qq[c]));
#pragma unroll
                    for (int u = 0; u < UP; ++u) {
                        float dv[8];
                        unpack8_h(raw[u], dv);
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[u] += qq[c] * dv[c];
                    }
                }
#pragma unroll
                for (int u = 0; u < UP; ++u) {
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o, 64);
                    if (sl == 0 && nn[u] != 0x7fffffff && db_norm[nn[u]] + acc[u] <= T) surv[atomicAdd(&s_ncand, 1u)] = nn[u];
                }
            }
            __syncthreads();
            const int ns = (int)s_ncand;
            for (int j = tid; j < ns; j += 256) e_i[nbest0 + j] = surv[j];
            __syncthreads();
            total = nbest0 + ns;
        }
        // exact fp64 distances for the new entries: 16 lanes per row, 16 rows in flight per wave
        // so that the row gathers overlap instead of serialising on one round trip per row.
        {
            constexpr int U = 2;
            const int sub = lane >> 4, sl = lane & 15;
            for (int e0 = nbest0 + wave * (4 * U); e0 < total; e0 += 16 * U) {
                int ee[U], nn[U];
                double acc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ee[u] = e0 + u * 4 + sub;
                    nn[u] = ee[u] < total ? e_i[ee[u]] : 0x7fffffff;
                    acc[u] = 0.0;
                }
                for (int i = sl * 4; i < d; i += 64) {
                    const f32x4 qq = *(const f32x4*)(qv + i);
                    f32x4 dd[U];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        dd[u] = nn[u] != 0x7fffffff ? *(const f32x4*)(xb + (size_t)nn[u] * d + i) : qq;
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const double t = (double)qq[c] - (double)dd[u][c];
                            acc[u] += t * t;
                        }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o, 64);
                    if (sl == 0 && ee[u] < total) e_d[ee[u]] = nn[u] != 0x7fffffff ? acc[u] : __builtin_huge_val();
                }
            }
        }
        if (dbg == 3) return;
        // sort the entries by (distance, index) -- bitonic network over the next power of two
        int P = 1;
        while (P < total) P <<= 1;
        for (int e = total + tid; e < P; e += 256) { e_d[e] = __builtin_huge_val(); e_i[e] = 0x7fffffff; }
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                for (int t = tid; t < (P >> 1); t += 256) {
                    const int lo = 2 * t - (t & (stride - 1));
                    const int hi = lo + stride;
                    const double dl = e_d[lo], dh = e_d[hi];
                    const int il = e_i[lo], ih = e_i[hi];
                    const bool gt = dl > dh || (dl == dh && il > ih);
                    const bool up = (lo & size) == 0;
                    if (gt == up) { e_d[lo] = dh; e_d[hi] = dl; e_i[lo] = ih; e_i[hi] = il; }
                }
            }
        }
        __syncthreads();
        // the first min(k, #valid) entries are the running best and stay in place for the next round
        if (tid == 0) {
            int nb_ = total < k ? total : k;
            while (nb_ > 0 && e_i[nb_ - 1] == 0x7fffffff) --nb_;
            s_nbest = nb_;
        }
        __syncthreads();
    }
    for (int e = tid; e < k; e += 256) {
        const bool ok = e < s_nbest;
        dist[(size_t)q * k + e] = ok ? (float)e_d[e] : 3.4028234663852886e38f;
        idx[(size_t)q * k + e] = ok ? (int64_t)e_i[e] : -1;
    }
}

inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

struct KnnWs {
    int64_t q_hi, q_lo, gmin, gminT, total;
    int G, gq_stride, g_stride, g_stride64;      // G: 16-row groups (generic coarse pass); the packed pass uses G / 4 groups of 64
};
inline KnnWs knn_ws(int64_t nq, int64_t nb, int d) {
    KnnWs w;
    const int64_t nb_pad = agp_knn_pad_rows(nb);
    w.G = (int)(nb_pad / 16);
    w.gq_stride = (int)((nq + 31) / 32 * 32);
    w.g_stride = (w.G + 31) / 32 * 32;
    w.g_stride64 = (w.G / 4 + 31) / 32 * 32;
    w.q_hi = 0;
    w.q_lo = align256(w.q_hi + nq * d * 2);
    w.gmin = align256(w.q_lo + nq * d * 2);
    w.gminT = align256(w.gmin + (int64_t)w.G * w.gq_stride * 4);
    // [query][16-row group] floats, or the packed pass's 2 planes [query][64-row group] -- on a tiny database both strides round
    // up to 32 words, so the two planes need MORE than the one (found in round 5: a 384-row database wrote 77 KB past the
    // workspace, which faulted only when the workspace happened to end a mapped segment)
    const int64_t per_query = w.g_stride > 2 * w.g_stride64 ? w.g_stride : 2 * w.g_stride64;
    w.total = align256(w.gminT + nq * per_query * 4);
    return w;
}

}  // namespace agp_knn
using namespace agp_knn;

extern "C" int64_t agp_knn_pad_rows(int64_t nb) {
    if (nb < 1) nb = 1;
    return (nb + KNN_TILE_ROWS - 1) / KNN_TILE_ROWS * KNN_TILE_ROWS;
}

extern "C" int agp_knn_prepare_db(const float* xb, int64_t nb, int d, int prec, void* db_hi, void* db_lo,
                                  float* db_norm, void* stream) {
    if (!db_hi || !db_norm || nb < 0 || d <= 0 || d % 32) return AGP_E_BADARG;
    if (prec != AGP_PREC_BF16X3 && prec != AGP_PREC_BF16 && prec != AGP_PREC_F16) return AGP_E_BADARG;
    if (nb > 0 && !xb) return AGP_E_BADARG;
    const int64_t nb_pad = agp_knn_pad_rows(nb);
    hipStream_t s = (hipStream_t)stream;
    AGP_LAUNCH(zero_tail_kernel, dim3(1), dim3(64), 0, s, db_norm + nb_pad);      // (a kernel, not a memset node: csrc/coords.hip)
    AGP_CHECK_LAUNCH();
    const int64_t want = (nb_pad + 3) / 4;
    AGP_LAUNCH(db_prep_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, s, xb, nb, nb_pad,
                       d, (bf16_t*)db_hi, (bf16_t*)db_lo, db_norm, prec == AGP_PREC_F16 ? 1 : 0);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int64_t agp_knn_workspace_bytes(int64_t nq, int64_t nb, int d, int k) {
    (void)k;
    if (nq < 1) nq = 1;
    return knn_ws(nq, nb, d).total;
}

static int knn_search_impl(const float* xq, int64_t nq, const float* xb, const void* db_hi,
                           const void* db_lo, const float* db_norm, int64_t nb, int d, int k, int prec,
                           float* dist, int64_t* idx, void* workspace, int64_t workspace_bytes,
                           void* stream, bool coarse_only) {
    if (nq == 0) return AGP_OK;
    if (!xq || !db_hi || !db_norm || (!coarse_only && (!dist || !idx)) || !workspace || nq < 0 || nb < 0) return AGP_E_BADARG;
    if (k < 1 || k > MAX_K || d % 32 || d <= 0) return AGP_E_BADARG;
    if (prec == AGP_PREC_BF16X3 && !db_lo) return AGP_E_BADARG;
    if (nb > 0 && !xb && !coarse_only) return AGP_E_BADARG;
    const KnnWs w = knn_ws(nq, nb, d);
    if (workspace_bytes < w.total) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int64_t nb_pad = agp_knn_pad_rows(nb);
    const int64_t nqd = nq * d;
    int g = (int)((nqd + 255) / 256);
    if (g > 4096) g = 4096;
    const int coarse = AGP_TUNE("KNN_COARSE", 1);       // development build, 0: the generic implicit GEMM for the fp16 coarse pass too
    const bool resident = prec == AGP_PREC_F16 && coarse && (d == 256 || d == 128 || d == 64) && nb_pad * (int64_t)d * 2 < (1ll << 31);
    if (!(resident && coarse_w4_applies(d, nq, nb_pad))) {
        AGP_LAUNCH(q_prep_kernel, dim3(g), dim3(256), 0, s, xq, nqd, (bf16_t*)(ws + w.q_hi),
                           (bf16_t*)(ws + w.q_lo), prec == AGP_PREC_F16 ? 1 : 0);
        AGP_CHECK_LAUNCH();
    }
    int rc;
    bool packed = false;
    if (resident) {
        // query-resident coarse kernel: writes (minimum + its row, second minimum) per group, already transposed ([query][group][2])
        uint32_t* gT = (uint32_t*)(ws + w.gminT);
        packed = true;
        if (d == 256) rc = launch_coarse_f16<256>(xq, ws + w.q_hi, db_hi, db_norm, gT, nq, nb, nb_pad, w.g_stride64, s);
        else if (d == 128) rc = launch_coarse_f16<128>(xq, ws + w.q_hi, db_hi, db_norm, gT, nq, nb, nb_pad, w.g_stride64, s);
        else rc = launch_coarse_f16<64>(xq, ws + w.q_hi, db_hi, db_norm, gT, nq, nb, nb_pad, w.g_stride64, s);
        if (rc != AGP_OK) return rc;
    } else {
        rc = agp_internal_gmin(ws + w.q_hi, ws + w.q_lo, nq, db_hi, db_lo, db_norm, nb, nb_pad, d, prec,
                               (float*)(ws + w.gmin), w.gq_stride, s);
        if (rc != AGP_OK) return rc;
        AGP_LAUNCH(transpose_kernel, dim3((unsigned)((nq + 31) / 32), (unsigned)((w.G + 31) / 32)),
                           dim3(32, 8), 0, s, (const float*)(ws + w.gmin), w.G, (int)nq, w.gq_stride,
                           (float*)(ws + w.gminT), w.g_stride);
        AGP_CHECK_LAUNCH();
    }
    // bound on |coarse - true| / (|q| |d|): split-bf16 products + fp32 accumulation (2^-13), plain bf16
    // (2^-7), or plain fp16 (operands to 2^-12 each -> 2^-10 with margin; saturation / underflow of the
    // fp16 planes are handled in select_rerank: out-of-range norms widen the window to everything)
    const float scale_d = d > 256 ? (float)d / 256.f : 1.f;
    const float cerr = (prec == AGP_PREC_BF16X3 ? 1.2207031e-4f : (prec == AGP_PREC_F16 ? 9.765625e-4f : 7.8125e-3f)) * scale_d;
    if (coarse_only) return AGP_OK;    // agp_knn_coarse_pass: query preparation + coarse pass
    const int dbg = AGP_TUNE("KNN_DBG", 0);
    const bf16_t* f16rows = (prec == AGP_PREC_F16 && d % 128 == 0 && !AGP_TUNE("KNN_NOPRUNE", 0)) ? (const bf16_t*)db_hi : nullptr;
    const float ce = prec == AGP_PREC_F16 ? -cerr : cerr;
    if (packed) {
        const int G64 = w.G / 4;
        if (G64 <= 8 * 256) {
            AGP_LAUNCH((select_rerank_kernel<true, 8>), dim3((unsigned)nq), dim3(256), 0, s, xq, xb, (const uint32_t*)(ws + w.gminT), G64,
                       w.g_stride64, db_norm, nb, nb_pad, d, k, ce, dist, idx, dbg, f16rows);
        } else {
            AGP_LAUNCH((select_rerank_kernel<true, 32>), dim3((unsigned)nq), dim3(256), 0, s, xq, xb, (const uint32_t*)(ws + w.gminT), G64,
                       w.g_stride64, db_norm, nb, nb_pad, d, k, ce, dist, idx, dbg, f16rows);
        }
    } else if (w.G <= 8 * 256) {
        AGP_LAUNCH((select_rerank_kernel<false, 8>), dim3((unsigned)nq), dim3(256), 0, s, xq, xb, (const uint32_t*)(ws + w.gminT), w.G,
                   w.g_stride, db_norm, nb, nb_pad, d, k, ce, dist, idx, dbg, f16rows);
    } else {
        AGP_LAUNCH((select_rerank_kernel<false, 32>), dim3((unsigned)nq), dim3(256), 0, s, xq, xb, (const uint32_t*)(ws + w.gminT), w.G,
                   w.g_stride, db_norm, nb, nb_pad, d, k, ce, dist, idx, dbg, f16rows);
    }
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_knn_search(const float* xq, int64_t nq, const float* xb, const void* db_hi,
                              const void* db_lo, const float* db_norm, int64_t nb, int d, int k, int prec,
                              float* dist, int64_t* idx, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    return knn_search_impl(xq, nq, xb, db_hi, db_lo, db_norm, nb, d, k, prec, dist, idx, workspace, workspace_bytes, stream, false);
}

extern "C" int agp_knn_coarse_pass(const float* xq, int64_t nq, const void* db_hi, const void* db_lo, const float* db_norm,
                                   int64_t nb, int d, int prec, void* workspace, int64_t workspace_bytes, void* stream) {
    return knn_search_impl(xq, nq, nullptr, db_hi, db_lo, db_norm, nb, d, 1, prec, nullptr, nullptr, workspace, workspace_bytes,
                           stream, true);
}
