// Implicit-GEMM convolution / distance GEMM on the gfx950 matrix cores.
//
//   C[n][m] = sum_k W[n][k] * X[m][k]        n = output channel (or database row)
//                                            m = output pixel   (or query)
//                                            k = (ky, kx, cin chunk)
//
// * X rows are gathered straight from the halo-padded NHWC activation planes by
//   LDS-DMA (buffer_load ... lds, 16 B per lane): the per-lane voffset carries the
//   pixel base, the scalar soffset carries the (ky,kx,c) tap, so the K loop does no
//   address arithmetic in vector registers and the zero halo supplies the padding.
// * The LDS image is [row][BK] bf16 with the 16-byte chunks XOR-swizzled on the
//   SOURCE side (LDS-DMA writes lane-linear), read back conflict-free with
//   ds_read_b128 as MFMA 32x32x16 bf16 fragments.
// * MFMA orientation: A = W (rows -> accumulator registers), B = X (cols -> lanes),
//   so one lane owns one pixel/query and 32 channels/database rows of it.
// * NPREC = 3 runs hi*hi + hi*lo + lo*hi on split-bf16 planes (fp32-class result),
//   NPREC = 1 is plain bf16.
// * Epilogue CONV: transpose through LDS to [pixel][channel], then scale/shift
//   (folded BN / bias), residual, ReLU, re-split, full-line 16-byte stores.
//   Epilogue GMIN: dist = |w|^2 + acc (queries pre-scaled by -2), min over the 16
//   database rows a lane holds per 32x32 tile -> gmin[group][query].

#ifndef AGP_SCHED
#define AGP_SCHED 0
#endif

#include "igemm_params.hpp"

namespace agp_igemm {


template <int BK> struct Swz;
template <> struct Swz<32> { __device__ static __forceinline__ int f(int row) { return (row >> 2) & 3; } };
template <> struct Swz<64> { __device__ static __forceinline__ int f(int row) { return (row >> 1) & 7; } };

template <int WM, int WN, int BK, int NPREC, int NST>
constexpr int igemm_lds_bytes() {
    constexpr int stage = (WM * 64 * PrecT<NPREC>::XPL + WN * 64 * PrecT<NPREC>::WPL) * BK * 2;
    // epilogue: 32 pixel rows per wave per pass (+ the waves' channel sums of the bf16-pair kernels: stat_partial)
    constexpr int epi = WM * WN * 32 * EPI_ROWB + (NPREC == 3 ? WM * WN * 64 * 2 * 4 : 0);
    return (NST * stage > epi) ? NST * stage : epi;
}

// NST = LDS pipeline depth: 2 = one K-step of prefetch (__syncthreads per step),
// >= 3 = NST-1 K-steps of LDS-DMA in flight across raw s_barriers with a counted vmcnt.
// The body is compiled in the device pass only: on the host pass hipcc (ROCm 7.2) silently
// drops the stub of a kernel template whose body holds the 32x32x16 MFMA loop.
#if defined(__HIP_DEVICE_COMPILE__)
template <int WM, int WN, int BK, int NPREC, int EPI, int NST>
__device__ __forceinline__ void igemm_body(const IgemmParams& p, const int bid) {
    constexpr int BM = WM * 64, BN = WN * 64, NW = WM * WN;
    constexpr int ROWB = BK * 2;          // bytes per LDS row
    constexpr int CPR = ROWB / 16;        // 16-B chunks per row
    constexpr int RPI = 1024 / ROWB;      // rows per wave-wide LDS-DMA instruction
    constexpr int XI = BM / RPI / NW;     // X instructions per wave per plane
    constexpr int WI = BN / RPI / NW;     // W instructions per wave per plane
    constexpr int XPL = PrecT<NPREC>::XPL, WPL = PrecT<NPREC>::WPL;
    constexpr int X_PLANE = BM * ROWB, W_PLANE = BN * ROWB;
    constexpr int STAGE = X_PLANE * XPL + W_PLANE * WPL;
    static_assert(XI >= 1 && WI >= 1, "tile too small for the wave count");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;

    // XCD-aware tile order (blocks b and b+8 share an XCD and its L2): each XCD owns a
    // CONTIGUOUS chunk of row tiles, so vertically adjacent tiles (the ky taps re-read the same
    // input rows) and the NT column tiles of one row tile hit the same L2.
    const int xcd = bid & 7, j = bid >> 3;
    const int nt = j % p.NT;
    int MTv = p.MT, chunk = p.mt_chunk;
    if (p.m_dev) {
        // capacity-mode sparse tensor: the grid covers the capacity, the VALID row tiles (device-side count) are dealt over the
        // XCDs -- with the host-side chunking 16 k valid rows of a 512 k capacity were 125 tiles on ONE XCD (32 CUs, 7 XCDs idle)
        const int64_t mv = *p.m_dev < (int64_t)p.M ? *p.m_dev : (int64_t)p.M;
        MTv = (int)((mv + BM - 1) / BM);
        chunk = (MTv + 7) / 8;
        if (j / p.NT >= chunk) return;
    }
    const int mt = xcd * chunk + j / p.NT;
    if (mt >= MTv) return;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-lane source offsets (bytes) for the LDS-DMA loads
    const int lrow = lane / CPR, lpos = lane % CPR;
    int xoff[XI], woff[WI], xm[XI], xswz[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = (wave + NW * i) * RPI + lrow;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        xm[i] = p.row_perm ? p.row_perm[m] : m;           // (sparse convolution: the output row this GEMM row computes)
        xswz[i] = (lpos ^ Swz<BK>::f(row)) << 4;
        const uint32_t img = fdiv((uint32_t)m, p.d_howo);
        const uint32_t rem = (uint32_t)m - img * p.d_howo.d;
        const uint32_t oy = fdiv(rem, p.d_wo);
        const uint32_t ox = rem - oy * p.d_wo.d;
        int el = (int)img * p.x_sn + (int)oy * p.sy * p.x_sh + (int)ox * p.sx * p.x_sw + p.x_base;
        xoff[i] = el * 2 + ((lpos ^ Swz<BK>::f(row)) << 4);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = (wave + NW * i) * RPI + lrow;
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        woff[i] = n * p.Ktot * 2 + ((lpos ^ Swz<BK>::f(row)) << 4);
    }

    const __amdgpu_buffer_rsrc_t rx_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(XPL == 2 ? p.x_lo : p.x_hi), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(WPL == 2 ? p.w_lo : p.w_hi), 0, p.w_bytes, 0x00020000);

    // K-step state: tap (ky,kx) and channel chunk, advanced incrementally
    const int cchunks = p.CK / BK;
    int nk = (p.dbg & 64) ? 1 : p.ntaps * cchunks;   // dbg 64: a single K-step (timing only)
    int kx = 0, ky = 0, cc = 0;

    // ---- sparse convolution: the taps that at least one row of this tile has a neighbour for, in tap order.  A tap that is
    // absent for the whole tile multiplies zero rows only (+0 into every accumulator): the K loop skips it, the result is
    // bit-identical.  The tile's tap set is the OR of its 128-row granules' masks (agp_sparse_tile_taps: one pass over the kernel
    // map per level, shared by the level's convolutions; an in-kernel ballot over the map columns cost every tile a memory round
    // trip and three barriers -- more than the skipped taps gave back at 4 of 27); taps are then peeled off the mask bit by bit.
    uint32_t tapmask = p.ntaps >= 32 ? 0xffffffffu : ((1u << p.ntaps) - 1u);
    if (p.tap_stride && p.tile_taps && p.ntaps <= 32 && !(p.dbg & 128)) {      // dbg 128: every tap (timing only)
        uint32_t mk = 0u;
#pragma unroll
        for (int q = 0; q < BM / 128; ++q) mk |= p.tile_taps[m0 / 128 + q];
        tapmask &= __builtin_amdgcn_readfirstlane(mk);
    }
    const bool masked = p.tap_stride && p.ntaps <= 32;
    int nact = masked ? __builtin_popcount(tapmask) : p.ntaps;
    if (masked && !(p.dbg & 64)) nk = nact * cchunks;
    uint32_t taprem = tapmask;                            // taps not yet started (bit 0 side first)
    auto tap_of = [&](uint32_t rem) { return rem ? __builtin_ctz(rem) : (tapmask ? 31 - __builtin_clz(tapmask) : 0); };

    int xnext[XI];                                       // sparse convolution: the next tap's gather-table words (one tap ahead)
    if (p.tap_stride) {
        const int* tab0 = p.xrow_tab + (size_t)(masked ? tap_of(taprem) : 0) * p.tap_stride;
#pragma unroll
        for (int i = 0; i < XI; ++i) xnext[i] = tab0[xm[i]];
    }
    auto stage_load = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        // wave-uniform by construction; readfirstlane makes that provable so the scalar
        // soffset is an SGPR and hipcc emits no waterfall loop around each LDS-DMA
        int wstep = kt;
        if (p.tap_stride) {
            // sparse convolution: every tap has its own gather table (neighbour row of each output row); kx walks the tile's
            // list of active taps (the prefetch one step past the end of the K loop stays on the last one)
            // the current tap = lowest bit of taprem (past the end: the last tap again); unmasked tables (> 32 taps): kx itself
            const int tap = masked ? tap_of(taprem) : (kx < p.ntaps ? kx : p.ntaps - 1);
            if (cc == 0) {
                // this tap's table words were requested one tap ago (xnext): reading them HERE would put a vmcnt(0) -- the whole
                // LDS-DMA ring -- in front of every tap's first stage
#pragma unroll
                for (int i = 0; i < XI; ++i) xoff[i] = (xnext[i] * p.tab_mul + p.x_base) * 2 + xswz[i];
                const int tapn = masked ? tap_of(taprem & (taprem - 1)) : (kx + 1 < p.ntaps ? kx + 1 : p.ntaps - 1);
                const int* tabn = p.xrow_tab + (size_t)tapn * p.tap_stride;
#pragma unroll
                for (int i = 0; i < XI; ++i) xnext[i] = tabn[xm[i]];
            }
            wstep = tap * cchunks + cc;
        }
        const int xs = __builtin_amdgcn_readfirstlane((ky * p.x_sh + kx * p.x_sw + cc * BK) * 2);
        const int ws = __builtin_amdgcn_readfirstlane(wstep * BK * 2);
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int ldsoff = (wave + NW * i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_hi, LDS_PTR(base + ldsoff), 16, xoff[i], xs, 0, 0);
            if (XPL == 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_lo, LDS_PTR(base + X_PLANE + ldsoff), 16, xoff[i], xs, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int ldsoff = X_PLANE * XPL + (wave + NW * i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_hi, LDS_PTR(base + ldsoff), 16, woff[i], ws, 0, 0);
            if (WPL == 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_lo, LDS_PTR(base + W_PLANE + ldsoff), 16, woff[i], ws, 0, 0);
        }
        // advance (cc, kx, ky) for the next call
        if (++cc == cchunks) {
            cc = 0;
            taprem &= taprem - 1;                         // (sparse convolution: the next active tap)
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };

    // ---- fragment read addresses (bytes within a plane)
    const int l31 = lane & 31, lh = lane >> 5;
    int xrow[2], wrow[2], xsw[2], wsw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int xr = wm * 64 + t * 32 + l31;
        const int wr = wn * 64 + t * 32 + l31;
        xrow[t] = xr * ROWB; xsw[t] = Swz<BK>::f(xr);
        wrow[t] = wr * ROWB; wsw[t] = Swz<BK>::f(wr);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto compute = [&](int buf) {
        const char* xb = smem + buf * STAGE;
        const char* wb = xb + X_PLANE * XPL;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int xo = xrow[t] + (((2 * ks + lh) ^ xsw[t]) << 4);
                const int wo = wrow[t] + (((2 * ks + lh) ^ wsw[t]) << 4);
                xh[t] = *(const bf16x8*)(xb + xo);
                wh[t] = *(const bf16x8*)(wb + wo);
                if (XPL == 2) xl[t] = *(const bf16x8*)(xb + X_PLANE + xo);
                if (WPL == 2) wl[t] = *(const bf16x8*)(wb + W_PLANE + wo);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) mfma32<NPREC>(acc[tn][tm], wh[tn], wl[tn], xh[tm], xl[tm]);
        }
    };

    if (NST == 2) {
#if AGP_SCHED
        constexpr int LPS = XI * XPL + WI * WPL;         // LDS-DMA instructions per wave per stage
        constexpr int NFR = 2 * (XPL + WPL);             // fragment reads per 16-deep k sub-step
        constexpr int NMF = 4 * PrecT<NPREC>::NPROD;     // MFMAs per k sub-step
#endif
        stage_load(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();  // stage kt landed (vmcnt(0)) and the other buffer is free
            // The prefetch is unconditional (the step past the end re-reads the last K-step into
            // the idle buffer) so that the whole body is ONE basic block and the LDS-DMA issue
            // can be interleaved with the MFMAs instead of running ahead of them.
            stage_load((kt + 1) & 1, kt + 1 < nk ? kt + 1 : nk - 1);
            compute(kt & 1);
#if AGP_SCHED
            {
                // first sub-step fragments, then one LDS-DMA issue behind each of the first MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
#pragma unroll
                for (int i = 0; i < LPS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (i == LPS / 2 - 1 && BK / 16 > 1) __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, (BK / 16) * NMF - LPS, 0);
            }
#endif
        }
    } else {
        // LPS LDS-DMA instructions per wave per stage; vmcnt counts them in issue order, so
        // "all but the newest (NST-2) stages have landed" is vmcnt((NST-2)*LPS).
        constexpr int LPS = XI * XPL + WI * WPL;
        int issued = 0;
        for (; issued < NST - 1 && issued < nk; ++issued) stage_load(issued, issued);
        int buf = 0, lbuf = NST - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = issued - kt - 1;     // stages in flight beyond stage kt (uniform)
            if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPS) : "memory");
            else if (NST > 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // stage kt visible to every wave; buffer lbuf is free
            if (issued < nk) {
                stage_load(lbuf, issued);
                ++issued;
            }
            compute(buf);
            buf = (buf + 1 == NST) ? 0 : buf + 1;
            lbuf = (lbuf + 1 == NST) ? 0 : lbuf + 1;
        }
    }

    if (EPI == EPI_GMIN) {
        // lane = query (m), registers = database rows (n). dist = |w|^2 + acc.
        const float INF = __builtin_huge_valf();
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int nb = n0 + wn * 64 + tn * 32;   // first database row of this 32-row tile
            float wn2[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb + 8 * q + 4 * lh;
#pragma unroll
                for (int e = 0; e < 4; ++e) wn2[4 * q + e] = (n + e < p.N) ? p.wnorm[n + e] : INF;
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                float v = INF;
#pragma unroll
                for (int r = 0; r < 16; ++r) v = fminf(v, wn2[r] + acc[tn][tm][r]);
                const int m = m0 + wm * 64 + tm * 32 + l31;
                const int g = (nb >> 5) * 2 + lh;
                if (m < p.M) p.gmin[(size_t)g * p.gq_stride + m] = v;
            }
        }
        return;
    }

    // ---- CONV epilogue: transpose through LDS to [pixel][channel], 32 pixel rows per pass
    if (p.dbg & 16) { if (acc[0][0][0] == 123.456f) p.gmin[0] = 1.f; return; }
    __syncthreads();  // everyone is done reading the staging buffers
    char* er = smem + wave * (32 * EPI_ROWB);
    const int ch = lane & 7;                         // 8-channel chunk within the wave's 64
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
    // optional per-tile channel statistics of the stored values (bf16-pair kernels only: train-mode BatchNorm of the convs this
    // kernel runs, the 1x1 / stride-2 ones; agp_conv_desc::stat_partial)
    const bool stats = NPREC == 3 && p.stat_partial != nullptr;
    float st1[8], st2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        if (tm) __syncthreads();      // pass 0's reads are done before pass 1 overwrites the rows
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
            int m = m0 + wm * 64 + tm * 32 + ml;
            if (m >= p.M) continue;
            if (p.row_perm) m = p.row_perm[m];
            const f32x4 a = *(const f32x4*)(er + ml * EPI_ROWB + ch * 32);
            const f32x4 b = *(const f32x4*)(er + ml * EPI_ROWB + ch * 32 + 16);
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
            if (p.dbg & 32) { if (v[0] == 1.2345678e30f) ohi[off] = 1; continue; }
            map_store8(ohi, olo, off, v);
            if constexpr (NPREC == 3) {
                if (stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { st1[e] += v[e]; st2[e] += v[e] * v[e]; }
                }
            }
        }
    }
    if constexpr (NPREC == 3) {
        if (stats) {
            // lanes sharing a channel chunk (lane & 7) -> wave totals; the WM waves of a column block -> tile totals, fixed order
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    st1[e] += __shfl_xor(st1[e], o, 64);
                    st2[e] += __shfl_xor(st2[e], o, 64);
                }
            }
            float* red = (float*)(smem + WM * WN * 32 * EPI_ROWB);      // [wave][64 channels][2]
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(wave * 64 + lane * 8 + e) * 2] = st1[e];
                    red[(wave * 64 + lane * 8 + e) * 2 + 1] = st2[e];
                }
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.N) {
                const int wn_c = tid / 64, cc = tid % 64;
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {             // wave index = wm + WM * wn
                    a += red[((wn_c * WM + w) * 64 + cc) * 2];
                    b += red[((wn_c * WM + w) * 64 + cc) * 2 + 1];
                }
                p.stat_partial[(size_t)mt * 2 * p.N + n0 + tid] = a;
                p.stat_partial[(size_t)mt * 2 * p.N + p.N + n0 + tid] = b;
            }
        }
    }
}
#endif  // __HIP_DEVICE_COMPILE__

template <int WM, int WN, int BK, int NPREC, int EPI, int NST>
__global__ void __launch_bounds__(WM* WN * 64) igemm_kernel(IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    igemm_body<WM, WN, BK, NPREC, EPI, NST>(p, blockIdx.x);
#endif
}

// Up to 4 problems of one tile configuration as ONE grid: workgroups [start[i], start[i+1]) run problem i.  Used for
// the stride-2 block entry of a ResNet stage: the 3x3/s2 conv and the 1x1/s2 downsample read the same input and are
// independent, and the latency-bound downsample (K = Cin: one or two K-steps per tile) hides between the tiles of
// the 3x3 instead of being a launch of its own (81 us alone next to 71 us for the 3x3 at L2 of the bench workload).
struct IgemmGroup {
    IgemmParams p[4];
    int start[5];
    int n;
};

template <int WM, int WN, int BK, int NPREC, int EPI, int NST>
__global__ void __launch_bounds__(WM* WN * 64) igemm_group_kernel(IgemmGroup g) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int bid = blockIdx.x;
    int prob = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.n && bid >= g.start[i]) prob = i;
    prob = __builtin_amdgcn_readfirstlane(prob);
    igemm_body<WM, WN, BK, NPREC, EPI, NST>(g.p[prob], bid - g.start[prob]);
#endif
}

// Grouped launch of the fp16 single-product configuration (the only one the product groups): every problem must
// satisfy CK % 32 == 0 and N % 128 == 0.
template <int BK, int NST>
int launch_group_f16_cfg(IgemmParams* ps, int n, hipStream_t s) {
    constexpr int WM = 2, WN = 2, NPREC = 4;
    constexpr int lds = igemm_lds_bytes<WM, WN, BK, NPREC, NST>();
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)igemm_group_kernel<WM, WN, BK, NPREC, EPI_CONV, NST>, lds, attr_done)) return AGP_E_LAUNCH;
    IgemmGroup g = {};
    g.n = n;
    int grid = 0;
    for (int i = 0; i < n; ++i) {
        IgemmParams& p = ps[i];
        p.MT = (p.M + WM * 64 - 1) / (WM * 64);
        p.NT = (p.N + WN * 64 - 1) / (WN * 64);
        p.mt_chunk = (p.MT + 7) / 8;
        g.p[i] = p;
        g.start[i] = grid;
        grid += p.mt_chunk * 8 * p.NT;
    }
    for (int i = n; i < 5; ++i) g.start[i] = grid;
    AGP_LAUNCH((igemm_group_kernel<WM, WN, BK, NPREC, EPI_CONV, NST>), dim3(grid), dim3(WM * WN * 64), lds, s, g);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

int launch_group_f16(IgemmParams* ps, int n, hipStream_t s) {
    // measured on the bench workload (serial conv-family fraction of peak / ms per step): BK 64 x 2 stages 0.275 / 2.215,
    // BK 32 x 3 stages with counted vmcnt 0.319 / 2.215 (default), BK 32 x 4 0.311 / 2.215, BK 64 x 3 0.304 / 2.30,
    // BK 32 x 2 0.316 / 2.207
#if defined(AGP_TUNING)
    const int var = AGP_TUNE("GROUP_VARIANT", 0);
    if (var == 1) return launch_group_f16_cfg<64, 2>(ps, n, s);
    if (var == 2) return launch_group_f16_cfg<32, 2>(ps, n, s);
#endif
    return launch_group_f16_cfg<32, 3>(ps, n, s);
}

template <int WM, int WN, int BK, int NPREC, int EPI, int NST>
int launch_cfg(IgemmParams& p, hipStream_t s) {
    constexpr int lds = igemm_lds_bytes<WM, WN, BK, NPREC, NST>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)igemm_kernel<WM, WN, BK, NPREC, EPI, NST>, lds, attr_done)) return AGP_E_LAUNCH;
    constexpr int BM = WM * 64, BN = WN * 64;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.N + BN - 1) / BN;
    p.mt_chunk = (p.MT + 7) / 8;
    const int grid = p.mt_chunk * 8 * p.NT;
    const int splits = 1;
    AGP_LAUNCH((igemm_kernel<WM, WN, BK, NPREC, EPI, NST>), dim3(grid, splits), dim3(WM * WN * 64), lds, s, p);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

// Tuning hook (development build only): IGEMM_VARIANT selects an alternative tile / pipeline configuration.
inline int igemm_variant() { return AGP_TUNE("IGEMM_VARIANT", 0); }

template <int EPI>
int launch_igemm(IgemmParams& p, int prec, hipStream_t s) {
    const bool wide = (p.N % 128 == 0) || (EPI == EPI_GMIN);
    const int var = igemm_variant();
    p.dbg = AGP_TUNE("IGEMM_DBG", 0);
    if (EPI == EPI_GMIN && var == 0) {
        // kNN coarse pass: 256 queries x 128 database rows per 8-wave workgroup (measured best)
        if (prec == AGP_PREC_BF16X3) return launch_cfg<4, 2, 32, 3, EPI, 2>(p, s);
        if (prec == AGP_PREC_BF16)
            return p.CK % 64 == 0 ? launch_cfg<4, 2, 64, 1, EPI, 2>(p, s) : launch_cfg<4, 2, 32, 1, EPI, 2>(p, s);
        if (prec == AGP_PREC_F16)
            return p.CK % 64 == 0 ? launch_cfg<4, 2, 64, 4, EPI, 2>(p, s) : launch_cfg<4, 2, 32, 4, EPI, 2>(p, s);
        return AGP_E_BADARG;
    }
    if (prec == AGP_PREC_BF16X3) {
#if defined(AGP_TUNING)
        if (var == 1) return wide ? launch_cfg<2, 2, 32, 3, EPI, 3>(p, s) : launch_cfg<4, 1, 32, 3, EPI, 3>(p, s);
        if (var == 2) return wide ? launch_cfg<4, 2, 32, 3, EPI, 2>(p, s) : launch_cfg<4, 1, 32, 3, EPI, 2>(p, s);
        if (var == 3) return wide ? launch_cfg<4, 2, 32, 3, EPI, 3>(p, s) : launch_cfg<4, 1, 32, 3, EPI, 3>(p, s);
        if (var == 4) return wide ? launch_cfg<2, 2, 32, 3, EPI, 4>(p, s) : launch_cfg<4, 1, 32, 3, EPI, 4>(p, s);
#endif
        return wide ? launch_cfg<2, 2, 32, 3, EPI, 2>(p, s) : launch_cfg<4, 1, 32, 3, EPI, 2>(p, s);
    } else if (prec == AGP_PREC_BF16) {
        if (p.CK % 64 == 0) {
#if defined(AGP_TUNING)
            if (var == 1) return wide ? launch_cfg<2, 2, 64, 1, EPI, 3>(p, s) : launch_cfg<4, 1, 64, 1, EPI, 3>(p, s);
#endif
            return wide ? launch_cfg<2, 2, 64, 1, EPI, 2>(p, s) : launch_cfg<4, 1, 64, 1, EPI, 2>(p, s);
        }
        return wide ? launch_cfg<2, 2, 32, 1, EPI, 2>(p, s) : launch_cfg<4, 1, 32, 1, EPI, 2>(p, s);
    }
    if constexpr (EPI == EPI_CONV) {       // fp16 modes exist for the convolutions only
        if (prec == AGP_PREC_F16W2)
            return wide ? launch_cfg<2, 2, 32, 2, EPI, 2>(p, s) : launch_cfg<4, 1, 32, 2, EPI, 2>(p, s);
        if (prec == AGP_PREC_F16) {
            // 32-deep K-steps through a 3-slot ring with counted vmcnt (see launch_group_f16: 0.319 against 0.275 of peak
            // for the conv family with 64-deep steps and one stage of prefetch)
#if defined(AGP_TUNING)
            if (var == 5) {          // the old choice
                if (p.CK % 64 == 0) return wide ? launch_cfg<2, 2, 64, 4, EPI, 2>(p, s) : launch_cfg<4, 1, 64, 4, EPI, 2>(p, s);
                return wide ? launch_cfg<2, 2, 32, 4, EPI, 2>(p, s) : launch_cfg<4, 1, 32, 4, EPI, 2>(p, s);
            }
#endif
            if (p.tap_stride && var == 0) {
                // gather-GEMM (sparse convolution), measured per shape on the voxel branch: 64 output channels run 14 % faster with
                // ONE stage of prefetch behind __syncthreads than through the 3-slot ring (a tap's gather table is read at the head
                // of a stage: the ring's counted waits drain behind it anyway); 128+ channels on 256 x 128 tiles (8 waves) when
                // that still fills the chip twice over
                if (!wide) return launch_cfg<4, 1, 32, 4, EPI, 2>(p, s);
                if ((int64_t)((p.M + 255) / 256) * (p.N / 128) >= 512) return launch_cfg<4, 2, 32, 4, EPI, 3>(p, s);
            }
            return wide ? launch_cfg<2, 2, 32, 4, EPI, 3>(p, s) : launch_cfg<4, 1, 32, 4, EPI, 3>(p, s);
        }
    }
    return AGP_E_BADARG;
}

}  // namespace agp_igemm
using namespace agp_igemm;

int agp_internal_conv_d16(agp_igemm::IgemmParams& p, int prec, hipStream_t s);
int agp_internal_conv_d16_pool(agp_igemm::IgemmParams& p, int prec, hipStream_t s);
int agp_internal_stem_raw(agp_igemm::IgemmParams& p, int kind, const void* x, long long sn, long long sc, long long sh, long long sw,
                          int h, int w, int ncam, const float* mean3, const float* std3, hipStream_t s);
int agp_internal_conv_kxr(agp_igemm::IgemmParams& p, const agp_conv_desc* d, hipStream_t s);

// Row tiles of the kernel that would run `d`, if that kernel can emit per-tile channel statistics
// (agp_conv_desc::stat_partial): the 3x3 stride-1 kernel on bf16-pair maps.  Must mirror agp_internal_conv_kxr.
extern "C" int agp_conv2d_stat_tiles(const agp_conv_desc* d) {
    if (!d || d->prec != AGP_PREC_BF16X3) return 0;
    // the packed stem on the direct-X kernel (igemm_d16: 256-row tiles of the plain [n][hout][wout] raster)
    const int force = AGP_TUNE("CONV_KERNEL", 0);          // development build: 1 = generic LDS-staged, 2 = direct-X, 3 = 3x3 kernel
    if (d->in_w_step != d->cin && !force && d->cout % 64 == 0)
        return (int)(((int64_t)d->n * d->hout * d->wout + 255) / 256);
    // everything else the generic kernel runs (1x1 and stride-2 convs): 128-row tiles for cout % 128 == 0, else 256
    if (d->in_w_step == d->cin && !force && !AGP_TUNE("IGEMM_VARIANT", 0) && d->cin % 32 == 0 && d->cout % 64 == 0 &&
        !(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1)) {
        const int bm = (d->cout % 128 == 0) ? 128 : 256;
        return (int)(((int64_t)d->n * d->hout * d->wout + bm - 1) / bm);
    }
    const bool kxr_ok = d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->pin == 1 &&
                        d->pout == 1 && d->in_w_step == d->cin && d->hout == d->hin && d->wout == d->win &&
                        (int64_t)d->n * (d->hin + 2) * (d->win + 2) * d->cin * 2 < (1ll << 31);
    if (!kxr_ok || d->cin % 32 || d->cout % 64 || (force && force != 3)) return 0;
    const int64_t m = (int64_t)d->n * d->hin * (d->win + 2);
    const int bm = (d->cout % 128 == 0 && !d->hi_only) ? 128 : 256;      // (the one-product form: 256-row tiles at every width)
    return (int)((m + bm - 1) / bm);
}

static bool conv_kxr_ok(const agp_conv_desc* d);
bool agp_internal_use_kxr2(const agp_conv_desc* d);

// 64-row blocks of agp_conv_desc::pool_partial: the AGP_PREC_F16 3x3 stride-1 kernel (igemm_kxr2, 256-row tiles of four
// 64-row wave blocks) over a raster that gives every image a multiple of 64 rows.
extern "C" int agp_conv2d_pool_blocks(const agp_conv_desc* d) {
    if (!d || d->prec != AGP_PREC_F16 || d->in_lo || d->out_lo || d->cin % 32 || d->cout % 64 || d->n <= 0) return 0;
    if (!conv_kxr_ok(d) || !agp_internal_use_kxr2(d) || AGP_TUNE("CONV_KERNEL", 0) || AGP_TUNE("NO_CONV_POOL", 0)) return 0;
    const int64_t rp = ((int64_t)d->hin * (d->win + 2) + 63) / 64 * 64;
    if ((int64_t)d->n * rp >= (1ll << 31)) return 0;
    return (int)(((int64_t)d->n * rp + 511) / 512 * 8);      // (64-row blocks of 256- or 512-row tiles: the larger count)
}

static bool conv_kxr_ok(const agp_conv_desc* d) {
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->pin == 1 &&
           d->pout == 1 && d->in_w_step == d->cin && d->hout == d->hin && d->wout == d->win &&
           (int64_t)d->n * (d->hin + 2) * (d->win + 2) * d->cin * 2 < (1ll << 31);
}

static int conv_fill_params(const agp_conv_desc* d, IgemmParams& p);
void agp_internal_conv_kxr_geometry(agp_igemm::IgemmParams& p, const agp_conv_desc* d);
bool agp_internal_use_kxr2(const agp_conv_desc* d);
int agp_internal_conv_kxr2(agp_igemm::IgemmParams* ps, int n, hipStream_t s);
int agp_internal_conv_s2(agp_igemm::IgemmParams* ps, const agp_conv_desc* descs, int n, hipStream_t s);

// Several convolutions of ONE channel shape (cin, cout, 3x3 stride 1) and precision as ONE launch: the tiles of
// every problem form one grid (igemm_kxr2.hip).  Groups the kernel cannot take run as `n` launches, in order.
extern "C" int agp_conv2d_fwd_grouped(const agp_conv_desc* descs, int n, void* stream) {
    if (!descs || n <= 0) return AGP_E_BADARG;
    bool group = n >= 2 && n <= 4;
    for (int i = 0; i < n && group; ++i) {
        const agp_conv_desc* d = descs + i;
        group = d->in_hi && d->w_hi && d->out_hi && !d->in_lo && !d->out_lo && !d->res_lo && d->n > 0 &&
                d->cin % 32 == 0 && d->cout % 64 == 0 && conv_kxr_ok(d) && agp_internal_use_kxr2(d) &&
                d->cin == descs[0].cin && d->cout == descs[0].cout && !AGP_TUNE("CONV_KERNEL", 0);
    }
    if (!group && !AGP_TUNE("CONV_KERNEL", 0) && !AGP_TUNE("NO_S2", 0)) {
        // the stride-2 entry of a ResNet stage: [3x3/s2 conv of every trunk ..., its 1x1/s2 downsample of every trunk ...] on fp16
        // maps with one product -> ONE launch of igemm_s2.hip (the downsample rides on the 3x3's staged centre tap)
        const int h = n / 2;
        bool s2 = (n == 2 || n == 4);
        for (int i = 0; i < h && s2; ++i) {
            const agp_conv_desc* c = descs + i;
            const agp_conv_desc* d = descs + h + i;
            s2 = c->prec == AGP_PREC_F16 && d->prec == AGP_PREC_F16 && c->in_hi && c->w_hi && c->out_hi && d->w_hi && d->out_hi &&
                 !c->in_lo && !c->out_lo && !d->in_lo && !d->out_lo && !c->res_hi && !d->res_hi && !c->stat_partial && !d->stat_partial &&
                 !c->pool_partial && !d->pool_partial &&
                 c->kh == 3 && c->kw == 3 && c->stride == 2 && c->pad == 1 && c->pin == 1 && c->pout == 1 && c->in_w_step == c->cin &&
                 d->kh == 1 && d->kw == 1 && d->stride == 2 && d->pad == 0 && d->pin == 1 && d->pout == 1 && d->in_w_step == d->cin &&
                 d->in_hi == c->in_hi && d->n == c->n && d->hin == c->hin && d->win == c->win && d->cin == c->cin &&
                 d->cout == c->cout && d->hout == c->hout && d->wout == c->wout && !d->relu &&
                 c->hout == (c->hin - 1) / 2 + 1 && c->wout == (c->win - 1) / 2 + 1 &&
                 c->cin % 32 == 0 && c->cout % 64 == 0 && c->n > 0 && c->cin == descs[0].cin && c->cout == descs[0].cout &&
                 (int64_t)c->n * (c->hin + 2) * (c->win + 2) * c->cin * 2 < (1ll << 31) &&
                 (int64_t)c->n * (c->hout + 2) * (c->wout + 2) * c->cout * 2 < (1ll << 31);
        }
        if (s2) {
            IgemmParams ps[2];
            for (int i = 0; i < h; ++i) {
                ps[i] = IgemmParams{};
                const int rc = conv_fill_params(descs + i, ps[i]);
                if (rc != AGP_OK) return rc;
                const agp_conv_desc* d = descs + h + i;
                ps[i].w2_hi = d->w_hi; ps[i].w2_cm = d->w_cm; ps[i].scale2 = d->scale; ps[i].shift2 = d->shift; ps[i].o2_hi = d->out_hi;
            }
            return agp_internal_conv_s2(ps, descs, h, (hipStream_t)stream);
        }
    }
    if (!group) {
        // second grouping: fp16 single-product convs of the generic kernel (1x1 and stride-2 convs) of one tile
        // configuration -- the stride-2 entry of a ResNet stage (3x3/s2 + 1x1/s2 downsample of every trunk)
        bool g2 = n >= 2 && n <= 4 && !AGP_TUNE("CONV_KERNEL", 0) && !AGP_TUNE("NO_IGEMM_GROUP", 0);
        for (int i = 0; i < n && g2; ++i) {
            const agp_conv_desc* d = descs + i;
            g2 = d->in_hi && d->w_hi && d->out_hi && !d->in_lo && !d->out_lo && !d->res_lo && d->n > 0 &&
                 d->prec == AGP_PREC_F16 && !d->stat_partial && d->cin % 64 == 0 && d->cout % 128 == 0 &&
                 !conv_kxr_ok(d) && d->in_w_step == d->cin && !(d->pin < d->pad);
        }
        if (g2) {
            IgemmParams ps[4];
            for (int i = 0; i < n; ++i) {
                ps[i] = IgemmParams{};
                const int rc = conv_fill_params(descs + i, ps[i]);
                if (rc != AGP_OK) return rc;
            }
            return launch_group_f16(ps, n, (hipStream_t)stream);
        }
        for (int i = 0; i < n; ++i) {
            const int rc = agp_conv2d_fwd(descs + i, stream);
            if (rc != AGP_OK) return rc;
        }
        return AGP_OK;
    }
    IgemmParams ps[4];
    for (int i = 0; i < n; ++i) {
        ps[i] = IgemmParams{};
        const int rc = conv_fill_params(descs + i, ps[i]);
        if (rc != AGP_OK) return rc;
        agp_internal_conv_kxr_geometry(ps[i], descs + i);
    }
    return agp_internal_conv_kxr2(ps, n, (hipStream_t)stream);
}

extern "C" int agp_conv2d_fwd(const agp_conv_desc* d, void* stream) {
    if (!d || !d->in_hi || !d->w_hi || !d->out_hi) return AGP_E_BADARG;
    // storage format follows the precision: BF16X3 = bf16 plane pairs everywhere; F16W2 / F16 = one
    // fp16 activation plane (lo pointers NULL) and an fp16 weight pair / single plane
    if (d->prec == AGP_PREC_BF16X3) {
        if ((!d->hi_only && (!d->in_lo || !d->w_lo)) || !d->out_lo || (d->res_hi && !d->res_lo)) return AGP_E_BADARG;
    } else if (d->prec == AGP_PREC_F16W2 || d->prec == AGP_PREC_F16) {
        if (d->in_lo || d->out_lo || d->res_lo) return AGP_E_BADARG;
        if (d->prec == AGP_PREC_F16W2 && !d->w_lo) return AGP_E_BADARG;
    } else {
        return AGP_E_BADARG;
    }
    if (d->cin % 32 || d->cout % 64 || d->n <= 0) return AGP_E_BADARG;
    if (d->pin < d->pad && d->in_w_step == d->cin) return AGP_E_BADARG;
    IgemmParams p = {};
    {
        const int rc = conv_fill_params(d, p);
        if (rc != AGP_OK) return rc;
    }
    // Kernel choice (development build: CONV_KERNEL = 1 generic / 2 direct-X / 3 3x3 kernel forces one where it is applicable):
    //   3x3 stride-1 pad-1 on 1-pixel-halo planes -> igemm_kxr.hip / igemm_kxr2.hip (horizontal-tap reuse in LDS)
    //   packed stem (in_w_step != cin)             -> igemm_d16.hip (X straight into registers)
    //   everything else (1x1, stride 2)            -> the generic LDS-staged kernel of this file
    const int force = AGP_TUNE("CONV_KERNEL", 0);
    const bool kxr_ok = conv_kxr_ok(d);
    const bool stem = d->in_w_step != d->cin;
#if defined(AGP_TUNING)
    if (p.dbg & 0x1000000) {   // census experiment (tools/census.py): the record buffer's address as two switch words
        const uint64_t a = ((uint64_t)(uint32_t)AGP_TUNE("CENSUS_BUF_HI", 0) << 32) | (uint32_t)AGP_TUNE("CENSUS_BUF_LO", 0);
        p.gmin = (float*)(uintptr_t)a;
        if (!p.gmin) p.dbg &= ~0x1000000;
    }
#endif
    int which = force ? force : (kxr_ok ? 3 : (stem ? 2 : 1));
    if (which == 3 && !kxr_ok) which = stem ? 2 : 1;
    if (d->hi_only && (d->prec != AGP_PREC_BF16X3 || which != 3)) return AGP_E_BADARG;      // the 3x3 stride-1 kernel's one-product form only
    // w_cm == w_hi: the caller holds chunk-major planes ONLY (training planes written that way): every kernel but the 3x3 stride-1
    // one would read them as row-major -- refuse instead
    if (d->w_cm && d->w_cm == d->w_hi && (which != 3 || !p.w_cm)) return AGP_E_BADARG;
    if (which == 3) return agp_internal_conv_kxr(p, d, (hipStream_t)stream);
    if (which == 2) return agp_internal_conv_d16(p, d->prec, (hipStream_t)stream);
    return launch_igemm<EPI_CONV>(p, d->prec, (hipStream_t)stream);
}

// The generic geometry of `d` (every conv kernel starts from it).
static int conv_fill_params(const agp_conv_desc* d, IgemmParams& p) {
    const int hp = d->hin + 2 * d->pin, wp = d->win + 2 * d->pin;
    const int wstep = d->in_w_step;
    // bytes of one plane; for the packed stem (in_w_step < cin) rows overlap, the plane
    // still has hp*wp pixels of in_w_step elements.
    const int64_t x_elems = (int64_t)d->n * hp * wp * wstep;
    const int64_t w_elems = (int64_t)d->cout * d->kh * d->kw * d->cin;
    if (x_elems * 2 >= (1ll << 32) || w_elems * 2 >= (1ll << 31)) return AGP_E_BADARG;
    p.x_hi = d->in_hi; p.x_lo = d->in_lo; p.x_bytes = (uint32_t)(x_elems * 2);
    p.w_hi = d->w_hi; p.w_lo = d->w_lo; p.w_bytes = (uint32_t)(w_elems * 2);
    if (d->prec == AGP_PREC_F16W2 && d->w_q8) { p.w_q8 = d->w_q8; p.w_q8_exp = d->w_q8_exp; }
    if (d->prec == AGP_PREC_F16 && d->w_cm) p.w_cm = d->w_cm;
    // the two-plane modes (igemm_kxr: 3x3 stride-1 convs): both planes chunk-major
    if ((d->prec == AGP_PREC_F16W2 || d->prec == AGP_PREC_BF16X3) && d->w_cm && d->kh == 3 && d->kw == 3 &&
        d->stride == 1 && d->pad == 1 && d->in_w_step == d->cin && (d->w_cm_lo || (d->hi_only && d->prec == AGP_PREC_BF16X3))) {
        p.w_cm = d->w_cm; p.w_cm_lo = d->w_cm_lo;
    }
    if (d->stat_partial) {
        if (agp_conv2d_stat_tiles(d) <= 0) return AGP_E_BADARG;      // only the kernels that can produce them
        p.stat_partial = d->stat_partial;
        if (d->bstat_z_hi) {
            // backward mode: the 3x3 stride-1 kernel only (the other kernels' tiles carry forward sums)
            // (bstat_z_lo NULL: z is ONE fp16 plane -- the output of a forward conv that ran as one fp16 product)
            if (!d->bstat_mean || !d->bstat_rstd || d->in_w_step != d->cin ||
                !(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1))
                return AGP_E_BADARG;
            p.bs_z_hi = d->bstat_z_hi; p.bs_z_lo = d->bstat_z_lo; p.bs_y_hi = d->bstat_y_hi;
            p.bs_mean = d->bstat_mean; p.bs_rstd = d->bstat_rstd;
        }
    } else if (d->bstat_z_hi) {
        return AGP_E_BADARG;
    }
    if (d->pool_partial) {
        if (agp_conv2d_pool_blocks(d) <= 0) return AGP_E_BADARG;
        if (d->pool_stat != 0 && d->pool_stat != 1) return AGP_E_BADARG;
        p.pool_partial = d->pool_partial; p.pool_p = d->pool_stat ? nullptr : d->pool_p; p.pool_eps = d->pool_eps; p.pool_sq = d->pool_stat;
    }
    p.M = d->n * d->hout * d->wout;
    p.N = d->cout;
    p.KW = d->kw; p.CK = d->cin; p.ntaps = d->kh * d->kw;
    p.Ktot = d->kh * d->kw * d->cin;
    p.d_howo = make_fastdiv((uint32_t)(d->hout * d->wout));
    p.d_wo = make_fastdiv((uint32_t)d->wout);
    p.x_sw = wstep; p.x_sh = wp * wstep; p.x_sn = hp * wp * wstep;
    p.x_base = ((d->pin - d->pad) * wp + (d->pin - d->pad)) * wstep;
    p.sy = d->stride; p.sx = d->stride;
    const int hop = d->hout + 2 * d->pout, wop = d->wout + 2 * d->pout;
    p.o_hi = d->out_hi; p.o_lo = d->out_lo;
    p.o_sw = d->cout; p.o_sh = wop * d->cout; p.o_sn = hop * wop * d->cout;
    p.o_base = (d->pout * wop + d->pout) * d->cout;
    p.r_hi = d->res_hi; p.r_lo = d->res_lo;
    p.scale = d->scale; p.shift = d->shift; p.relu = d->relu;
    p.dbg = AGP_TUNE("IGEMM_DBG", 0);
    return AGP_OK;
}

// Coarse kNN pass, called from knn.hip: W = database rows, X = queries (1x1 "conv").
int agp_internal_gmin(const void* q_hi, const void* q_lo, int64_t nq, const void* db_hi,
                      const void* db_lo, const float* db_norm, int64_t nb, int64_t nb_pad, int d,
                      int prec, float* gmin, int gq_stride, hipStream_t s) {
    IgemmParams p = {};
    if (nq * d * 2 >= (1ll << 32) || nb_pad * d * 2 >= (1ll << 31)) return AGP_E_BADARG;
    p.x_hi = q_hi; p.x_lo = q_lo; p.x_bytes = (uint32_t)(nq * d * 2);
    p.w_hi = db_hi; p.w_lo = db_lo; p.w_bytes = (uint32_t)(nb_pad * d * 2);
    p.M = (int)nq; p.N = (int)nb; p.Ktot = d; p.KW = 1; p.CK = d; p.ntaps = 1;
    p.d_howo = make_fastdiv(1); p.d_wo = make_fastdiv(1);
    p.x_sn = d; p.x_sh = 0; p.x_sw = 0; p.x_base = 0; p.sy = 1; p.sx = 1;
    p.gmin = gmin; p.wnorm = db_norm; p.gq_stride = gq_stride;
    return launch_igemm<EPI_GMIN>(p, prec, s);
}


// ---- sparse (submanifold / strided) convolution as a gather-GEMM on the generic kernel: feature rows
// [n_in + 1][cin] (the last row is zero and stands for a missing neighbour), one gather table per tap.
extern "C" int agp_sparse_conv_fwd(const void* f_hi, const void* f_lo, int64_t n_in_rows, const int32_t* nbr, int64_t n_out,
                                   int cin, int cout, int ntaps, const void* w_hi, const void* w_lo, const float* scale,
                                   const float* shift, const void* res_hi, const void* res_lo, int relu, void* out_hi,
                                   void* out_lo, int prec, const int64_t* n_dev, const int32_t* row_perm, const uint32_t* tile_taps, void* stream) {
    if (!f_hi || !nbr || !w_hi || !out_hi || n_out <= 0 || n_in_rows <= 0 || ntaps <= 0) return AGP_E_BADARG;
    if (cin % 32 || cout % 64) return AGP_E_BADARG;
    if (prec == AGP_PREC_BF16X3) { if (!f_lo || !w_lo || !out_lo || (res_hi && !res_lo)) return AGP_E_BADARG; }
    else if (prec == AGP_PREC_F16W2 || prec == AGP_PREC_F16) { if (f_lo || out_lo || res_lo || (prec == AGP_PREC_F16W2 && !w_lo)) return AGP_E_BADARG; }
    else return AGP_E_BADARG;
    const int64_t x_elems = n_in_rows * cin, w_elems = (int64_t)cout * ntaps * cin;
    if (x_elems * 2 >= (1ll << 31) || w_elems * 2 >= (1ll << 31) || n_out * (int64_t)cout * 2 >= (1ll << 31)) return AGP_E_BADARG;
    IgemmParams p = {};
    p.x_hi = f_hi; p.x_lo = f_lo; p.x_bytes = (uint32_t)(x_elems * 2);
    p.w_hi = w_hi; p.w_lo = w_lo; p.w_bytes = (uint32_t)(w_elems * 2);
    p.M = (int)n_out; p.N = cout; p.Ktot = ntaps * cin;
    p.KW = ntaps; p.CK = cin; p.ntaps = ntaps;
    p.d_howo = make_fastdiv((uint32_t)n_out); p.d_wo = make_fastdiv((uint32_t)n_out);
    p.x_sn = 0; p.x_sh = 0; p.x_sw = 0; p.x_base = 0; p.sy = 1; p.sx = 1;
    p.xrow_tab = nbr; p.tap_stride = (int)n_out; p.tab_mul = cin; p.m_dev = n_dev; p.row_perm = row_perm; p.tile_taps = tile_taps;
    p.o_hi = out_hi; p.o_lo = out_lo; p.o_sn = 0; p.o_sh = 0; p.o_sw = cout; p.o_base = 0;
    p.r_hi = res_hi; p.r_lo = res_lo; p.scale = scale; p.shift = shift; p.relu = relu;
    return launch_igemm<EPI_CONV>(p, prec, (hipStream_t)stream);
}

// ---- packed 7x7/2 stem conv + BatchNorm + ReLU + MaxPool2d(3, 2, 1) in one kernel (fp16 maps).
// `d` describes the stem conv as for agp_conv2d_fwd (cin = 32, in_w_step = 4, kw = 1, stride 2, pad 3,
// cout = 64, relu = 1) except that out_* is the POOLED map [n][hp2][wp2][64] with halo d->pout and
// hout / wout are the POOLED sizes.
static int stem_pool_impl(const agp_conv_desc* d, int kind, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int ncam,
                          const float* mean3, const float* std3, void* stream) {
    if (!d || !d->in_hi || !d->w_hi || !d->out_hi || d->in_lo || d->out_lo || d->res_hi) return AGP_E_BADARG;
    if (d->prec != AGP_PREC_F16W2 && d->prec != AGP_PREC_F16) return AGP_E_BADARG;
    if (d->prec == AGP_PREC_F16W2 && !d->w_lo) return AGP_E_BADARG;
    if (d->cin != 32 || d->in_w_step != 4 || d->kw != 1 || d->kh != 7 || d->stride != 2 || d->pad != 3 || d->pin != 3 ||
        d->cout != 64 || !d->relu || d->n <= 0)
        return AGP_E_BADARG;
    const int h1 = (d->hin + 2 * 3 - 7) / 2 + 1, w1 = (d->win + 2 * 3 - 7) / 2 + 1;
    const int h2 = (h1 + 2 - 3) / 2 + 1, w2 = (w1 + 2 - 3) / 2 + 1;
    if (d->hout != h2 || d->wout != w2) return AGP_E_BADARG;
    IgemmParams p = {};
    const int hp = d->hin + 6, wp = d->win + 6;
    const int64_t x_elems = (int64_t)d->n * hp * wp * 4;
    const int64_t w_elems = (int64_t)64 * 7 * 32;
    if (kind == 0 && x_elems * 2 >= (1ll << 32)) return AGP_E_BADARG;
    p.x_hi = d->in_hi; p.x_lo = nullptr; p.x_bytes = (uint32_t)(x_elems * 2);
    p.w_hi = d->w_hi; p.w_lo = d->w_lo; p.w_bytes = (uint32_t)(w_elems * 2);
    p.M = d->n * h1 * w1; p.N = 64; p.Ktot = 7 * 32; p.KW = 1; p.CK = 32; p.ntaps = 7;
    p.d_howo = make_fastdiv((uint32_t)(h1 * w1)); p.d_wo = make_fastdiv((uint32_t)w1);
    p.x_sw = 4; p.x_sh = wp * 4; p.x_sn = hp * wp * 4; p.x_base = 0; p.sy = 2; p.sx = 2;
    const int hop = h2 + 2 * d->pout, wop = w2 + 2 * d->pout;
    p.o_hi = d->out_hi; p.o_lo = nullptr;
    p.o_sw = 64; p.o_sh = wop * 64; p.o_sn = hop * wop * 64; p.o_base = (d->pout * wop + d->pout) * 64;
    p.scale = d->scale; p.shift = d->shift; p.relu = 1;
    p.pool_h1 = h1; p.pool_w1 = w1; p.pool_h2 = h2; p.pool_w2 = w2;
    p.pool_ty = (h2 + 6) / 7; p.pool_tx = (w2 + 6) / 7;
    if (kind == 0) return agp_internal_conv_d16_pool(p, d->prec, (hipStream_t)stream);
    if (d->prec != AGP_PREC_F16) return AGP_E_BADARG;
    if (kind == 2 && (ncam <= 0 || d->win % ncam)) return AGP_E_BADARG;
    return agp_internal_stem_raw(p, kind, d->in_hi, sn, sc, sh, sw, d->hin, d->win, ncam, mean3, std3, (hipStream_t)stream);
}

extern "C" int agp_stem_pool_fwd(const agp_conv_desc* d, void* stream) {
    return stem_pool_impl(d, 0, 0, 0, 0, 0, 1, nullptr, nullptr, stream);
}

extern "C" int agp_stem_pool_raw_fwd(const agp_conv_desc* d, int kind, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int ncam,
                                     const float* mean3, const float* std3, void* stream) {
    if (kind != 1 && kind != 2) return AGP_E_BADARG;
    return stem_pool_impl(d, kind, sn, sc, sh, sw, ncam, mean3, std3, stream);
}
