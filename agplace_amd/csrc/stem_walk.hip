// The stem as a horizontal WALK: packed 7x7/2 conv + BatchNorm + ReLU + MaxPool2d(3, 2, 1), fp16 maps (reference
// network_mm/image_fe.py:97-100 = torchvision ResNet conv1 / bn1 / relu / maxpool).
//
// stem_pool_lds_kernel (igemm_d16.hip) gives every 16 x 16 block of the conv map a workgroup of its own: 28 KB of W and an
// 11 KB input patch go to LDS, 0.75 us of MFMA follow, then a 36 KB fp16 block is laid down in LDS and pooled -- load,
// compute and epilogue of a workgroup are serial, a workgroup lives 8.6 us, and only the four workgroups of a CU overlap
// one another (MFMA pipe 0.42 busy; 27 % of the conv is recomputed at the block seams, all of W is re-staged 24 576 times).
//
// Here a workgroup OWNS a strip of 16 conv rows (14 new ones = 7 pooled rows) of one image and walks along it in steps of
// 16 conv columns:
//   * W lives in REGISTERS for the whole walk (7 tap rows x 4 channel tiles x 16 B per lane = 112 VGPRs, loaded once from
//     global memory): the K loop reads only the X fragments from LDS -- 13 ds_read_b128 per step, the compiler merges the
//     reads of a patch row shared by several (block row, tap row) pairs; with W in LDS it would be 41;
//   * the input patch of step j + 1 (37 rows x 38 pixels x 4 channels fp16) is fetched by LDS-DMA into the other half of a
//     double buffer while step j computes: one barrier per step, no load latency on the chain;
//   * no horizontal seam: the 3-wide horizontal maximum runs across lanes (DPP row rotate / shift inside the 16-lane rows of
//     the 16x16x32 accumulator layout), and the one column it needs from the previous step is CARRIED in a register
//     (lane 0 <- lane 15 of the step before).  Vertically a wave holds 4 conv rows: pooled row 2w is complete in its
//     registers, pooled row 2w + 1 needs conv row 4w + 4 = the first row of the next wave, which passes through 2 KB of LDS
//     per wave and step, read after the NEXT step's barrier (the epilogue's second half is software-pipelined behind it).
//     The 16 x 16 x 64 block never exists in LDS; the recompute is 16 / 14 (strip seams) instead of (16 / 14)^2;
//   * the pooled outputs leave as 8-byte pieces per lane (4 channels): the four channel tiles of a pixel fill its 128-byte
//     line from four instructions of the same wave (merged in L2; the stem writes 154 MB, the convs behind it 10x that).
// A strip is cut into segments when there are fewer strips than workgroup slots (2 per CU); a segment that does not start at
// the image's left edge first runs ONE step to the left of its range with its stores masked, which produces the carry.
//
// Arithmetic: the same operand layout, tap order and MFMA as stem_pool_lds_kernel, so the conv sums are bit-identical to
// that kernel's, and the maximum of post-ReLU fp16 values is exact in any order: outputs are bit-identical.
#include "igemm_params.hpp"

namespace agp_igemm {

[[maybe_unused]] constexpr int SWK_PROWB = 304;                        // patch row bytes: 38 pixels x 4 channels fp16
[[maybe_unused]] constexpr int SWK_PROWS = 37;
constexpr int SWK_PBUF = 12 * 1024;                   // 12 LDS-DMA instructions of 1 KB cover the 703 16-byte chunks of a patch
constexpr int SWK_EXCH = 2048;                        // per wave and step parity: conv row 4w of the block, 64 lanes x 32 B
constexpr int SWK_OFF_EXCH = 2 * SWK_PBUF;
constexpr int SWK_OFF_SS = SWK_OFF_EXCH + 2 * 4 * SWK_EXCH;
constexpr int SWK_OFF_STAGE = SWK_OFF_SS + 512;       // BatchNorm scale / shift (2 x 64 fp32), then (IN = 1) the fp32 staging
constexpr int SWK_STAGE_W = 5 * 1024;                 // per wave: 5 LDS-DMA instructions >= 10 rows x 3 channels x 10 chunks
constexpr int SWK_OFF_W6 = SWK_OFF_STAGE + 4 * SWK_STAGE_W;   // IN >= 1: tap row 6 of W (4 KB), then (IN = 2) the uint8 -> fp16 table (3 x 256)
constexpr int SWK_OFF_LUT = SWK_OFF_W6 + 4096;
template <int IN> constexpr int swk_lds() { return IN == 2 ? SWK_OFF_LUT + 3 * 256 * 2 : (IN == 1 ? SWK_OFF_LUT : SWK_OFF_STAGE); }

struct WalkGeo {
    int nblk;          // 16-column blocks per conv row
    int nseg, sps;     // segments per strip, steps per segment
    int items, chunk;  // work items (image, strip, segment); items per XCD
    uint32_t o_bytes;  // bytes of the pooled map (halo included)
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ u32x2 pkmax4_h(const u32x2& a, const u32x2& b) {      // see pkmax8_h: exact for post-ReLU fp16
    return __builtin_bit_cast(u32x2, __builtin_elementwise_max(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b)));
}
__device__ __forceinline__ uint32_t pkmax2_h(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// inside a row of 16 lanes: lane i <- lane (i - 1) mod 16  /  lane i <- lane i + 1 (lane 15 <- 0)
// (mov_dpp: no `old` operand to initialise; row_ror has a source for every lane, row_shl's missing one reads 0 = bound_ctrl)
__device__ __forceinline__ uint32_t dpp_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x121, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x101, 0xf, 0xf, true); }
// packed fp16 pair of clamp(a, 0, hi), clamp(b, 0, hi): ReLU + saturation (hi = 65504) or the seam mask (hi = 0) in the one med3
__device__ __forceinline__ uint32_t pack2_h_clamp(float a, float b, float hi) {
    const f32x2v v = {__builtin_amdgcn_fmed3f(a, 0.f, hi), __builtin_amdgcn_fmed3f(b, 0.f, hi)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

// IN = 2: uint8 camera tiles, see below.
// IN = 0: the packed NHWC4 halo-3 map (LDS-DMA).  IN = 1: the network's input itself, an fp32 [n][3][h][w] image with unit
// column stride and 16-byte aligned rows (StemRaw; checked by the launcher), staged by LDS-DMA and converted to fp16 NHWC4 on
// its way into the patch: no packed copy of the image is written or read.
template <int IN>
__global__ void __launch_bounds__(256, 2) stem_walk_kernel(IgemmParams p, StemRaw raw, WalkGeo g) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int item = (blockIdx.x & 7) * g.chunk + (blockIdx.x >> 3);     // neighbouring strips (shared input rows) on one XCD
    if (item >= g.items) return;
    const int seg = item % g.nseg;
    const int strip = item / g.nseg;
    const int pty = strip % p.pool_ty, pimg = strip / p.pool_ty;
    const int j0 = seg * g.sps, j1 = min(g.nblk, j0 + g.sps);
    const int js = j0 > 0 ? j0 - 1 : 0;                                  // the carry's warm-up step
    const int oy0 = 14 * pty - 1;

    char* patch = smem;
    char* exch = smem + SWK_OFF_EXCH;
    float* ss = (float*)(smem + SWK_OFF_SS);
    if (tid < 64) {
        ss[tid] = p.scale ? p.scale[tid] : 1.f;
        ss[64 + tid] = p.shift ? p.shift[tid] : 0.f;
    }
    // ---- W fragments: channel nt * 16 + l15, k = ky * 32 + 8 * lq .. + 7
    // (IN = 1 needs ~20 registers more than the 256 of two waves per SIMD leave: its last tap row's fragments live in LDS,
    // [nt][lane] x 16 B per wave = 4 more ds_read_b128 per step)
    constexpr int WREG = IN >= 1 ? 6 : 7;
    bf16x8 wf[7][4];
    char* wl6 = smem + SWK_OFF_W6 + lane * 16;       // the same bytes from every wave
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const bf16x8 f = *(const bf16x8*)((const char*)p.w_hi + ((nt * 16 + l15) * p.Ktot + ky * 32 + lq * 8) * 2);
            if (ky < WREG) wf[ky][nt] = f;
            else *(bf16x8*)(wl6 + nt * 1024) = f;
        }

    // ---- patch addressing
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_hi, 0, p.x_bytes, 0x00020000);
    int choff[3];                  // IN = 0: byte offset of this lane's chunk of instruction i inside the patch's source window
    unsigned chok = 0;
    if constexpr (IN == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = (wave * 3 + i) * 64 + lane;                    // 12 instructions cover 768 >= 703 chunks
            const int pr = c / 19, pj = c - pr * 19;
            choff[i] = pr * p.x_sh * 2 + pj * 16;
            chok |= (pr < SWK_PROWS ? 1u : 0u) << i;
        }
    }
    const int xbase0 = (pimg * p.x_sn + 2 * oy0 * p.x_sh) * 2;            // may be "negative": wraps past num_records -> zeros

    auto issue_patch = [&](int j, int buf) {
        if constexpr (IN == 0) {
            const int base = xbase0 + 256 * j;                           // 32 input pixels x 8 B per step
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int off = ((chok >> i) & 1u) ? base + choff[i] : -16;       // past the patch: a zero read
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(patch + buf * SWK_PBUF + (wave * 3 + i) * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    // ---- IN = 1: the fp32 planes themselves.  Wave w owns patch rows w, w + 4, ...: it fetches them as 16-byte chunks (4 pixels
    // of one channel; columns 32 j - 4 ... 32 j + 35, chunk-aligned) by LDS-DMA into ITS staging area while step j computes, and
    // converts them into the other patch buffer after its own `s_waitcnt vmcnt(0)`: no barrier between the two, no register
    // holds a load across the MFMAs.  Chunk q = rr * 30 + ch * 10 + g of the wave (row w + 4 rr, channel ch, column group g).
    const __amdgpu_buffer_rsrc_t rraw = __builtin_amdgcn_make_buffer_rsrc((void*)raw.x, 0, raw.bytes, 0x00020000);
    char* stage = smem + SWK_OFF_STAGE + wave * SWK_STAGE_W;
    int rq[5];                     // byte offset of chunk q = 64 i + lane at j = 0 (a multiple of 16), its g in the low 4 bits; -1 = none
    if constexpr (IN == 1) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = i * 64 + lane;
            const int rr = q / 30, rem = q - rr * 30;
            const int ch = rem / 10, gq = rem - ch * 10;
            const int pr = wave + 4 * rr, iy = 2 * oy0 + pr - 3;
            const bool ok = q < 300 && pr < SWK_PROWS && iy >= 0 && iy < raw.h;
            const long long el = (long long)pimg * raw.sn + (long long)ch * raw.sc + (long long)iy * raw.sh + 4 * gq - 4;
            rq[i] = ok ? (int)(el * 4) | gq : -1;
        }
    }
    // ---- IN = 2: uint8 camera tiles [n][ncam][h][wcam][3] (wcam % 32 == 0: a step's 32 columns lie in ONE camera, only the
    // 3-pixel fringes may come from a neighbour).  Per patch row 8 chunks of 16 bytes: c8 = 0 the chunk that ENDS at the step's
    // first column (left fringe), 1..6 the step's 96 bytes, 7 the chunk that STARTS behind them (right fringe); a wave's
    // <= 10 rows are 80 chunks = 2 LDS-DMA instructions.  ToTensor + Normalize through a 3 x 256 fp16 table built once per
    // workgroup with pack_u8_cams_kernel's arithmetic (bit-identical to packing first).
    bf16_t* const lut = (bf16_t*)(smem + SWK_OFF_LUT);
    const int rowb = raw.wcam * 3;                 // bytes of a camera tile row
    const int spc = raw.wcam >> 5;                 // steps per camera
    int riy[2];                                    // byte offset of this lane's patch row inside a camera tile (instruction i), -1 = none
    if constexpr (IN == 2) {
        for (int idx = tid; idx < 768; idx += 256) {
            const int c = idx >> 8, v = idx & 255;
            lut[idx] = f2h(((float)v / 255.f - raw.m[c]) / raw.s[c]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = i * 8 + (lane >> 3);
            const int pr = wave + 4 * rr, iy = 2 * oy0 + pr - 3;
            riy[i] = (rr < 10 && pr < SWK_PROWS && iy >= 0 && iy < raw.h) ? iy * rowb : -1;
        }
        __syncthreads();                           // the table, before the first conversion
    }
    auto issue_raw = [&](int j) {
        if constexpr (IN == 2) {
            const int cam = j / spc, jj = j - cam * spc;
            const int img0 = pimg * raw.ncam;
            const int base_m = (img0 + cam) * raw.h * rowb + jj * 96;
            const bool lv = jj > 0 || cam > 0, rv = jj + 1 < spc || cam + 1 < raw.ncam;
            const int base_l = jj > 0 ? base_m - 16 : (img0 + cam - 1) * raw.h * rowb + rowb - 16;
            const int base_r = jj + 1 < spc ? base_m + 96 : (img0 + cam + 1) * raw.h * rowb;
            const int c8 = lane & 7;
            const int cb = c8 == 0 ? base_l : (c8 == 7 ? base_r : base_m + 16 * (c8 - 1));
            const bool cv = c8 == 0 ? lv : (c8 == 7 ? rv : true);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (cv && riy[i] >= 0) ? cb + riy[i] : -16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rraw, LDS_PTR(stage + i * 1024), 16, off, 0, 0, 0);
            }
        }
        if constexpr (IN == 1) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int gq = rq[i] & 15;
                const int ix0 = 32 * j - 4 + 4 * gq;
                const bool ok = gq < 10 && ix0 >= 0 && ix0 < raw.w;
                const int off = ok ? (rq[i] & ~15) + 128 * j : -16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rraw, LDS_PTR(stage + i * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    // the wave's staged rows -> fp16 NHWC4 pixels of patch buffer `buf` (pixel px of the patch = column 32 j - 3 + px)
    auto convert_raw = [&](int buf, int j) {
        if constexpr (IN == 2) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int u = t * 64 + lane;                             // unit = (row rr, patch pixel p)
                const int rr = (u * 6899) >> 18, p_ = u - rr * 38;        // u / 38 for u < 384
                const int pr = wave + 4 * rr;
                if (u < 380 && pr < SWK_PROWS) {
                    const int iy = 2 * oy0 + pr - 3, col = 32 * j - 3 + p_;
                    const bool ok = iy >= 0 && iy < raw.h && col >= 0 && col < raw.w;
                    const int bo = rr * 128 + (p_ < 3 ? 7 + 3 * p_ : (p_ < 35 ? 16 + 3 * (p_ - 3) : 112 + 3 * (p_ - 35)));
                    const uint8_t* sb = (const uint8_t*)stage + bo;
                    const unsigned b0 = sb[0], b1 = sb[1], b2 = sb[2];
                    const bf16_t h0 = lut[b0], h1 = lut[256 + b1], h2 = lut[512 + b2];
                    const u32x2 pk = ok ? u32x2{pack2(h0, h1), pack2(h2, (bf16_t)0)} : u32x2{0u, 0u};
                    *(u32x2*)(patch + buf * SWK_PBUF + pr * SWK_PROWB + p_ * 8) = pk;
                }
            }
        }
        if constexpr (IN == 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int u = t * 64 + lane;                             // unit = (row rr, column group g): 4 pixels
                const int rr = u / 10, gq = u - rr * 10;
                const int pr = wave + 4 * rr;
                if (u < 100 && pr < SWK_PROWS) {
                    const char* sq = stage + (rr * 30 + gq) * 16;
                    const f32x4 c0 = *(const f32x4*)sq, c1 = *(const f32x4*)(sq + 160), c2 = *(const f32x4*)(sq + 320);
                    char* dst = patch + buf * SWK_PBUF + pr * SWK_PROWB + (4 * gq - 1) * 8;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if ((gq == 0 && k == 0) || (gq == 9 && k == 3)) continue;      // pixels -1 and 38 of the patch do not exist
                        *(u32x2*)(dst + k * 8) = u32x2{pack2(f2h(c0[k]), f2h(c1[k])), pack2(f2h(c2[k]), (bf16_t)0)};
                    }
                }
            }
        }
    };

    unsigned rowok = 0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int cy = oy0 + 4 * wave + mt;
        rowok |= ((cy >= 0 && cy < p.pool_h1) ? 1u : 0u) << mt;
    }
    const int pfo = (8 * wave) * SWK_PROWB + 16 * (l15 + lq);       // block row 4 * wave (+ mt), patch row 2 * that (+ ky)
    const int xo = lane * 16;                                        // this lane's 16 B in each 1 KB half of an exchange slot
    // pooled outputs leave through a buffer descriptor: a masked store gets an offset past num_records and is dropped
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.o_hi, 0, g.o_bytes, 0x00020000);
    const unsigned orow = (unsigned)(pimg * p.o_sn + p.o_base + 4 * lq) * 2u;

    uint32_t carry[2][4][2];
    u32x2 V[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { carry[r][nt][0] = 0u; carry[r][nt][1] = 0u; V[r][nt] = u32x2{0u, 0u}; }

    // second half of step jj's epilogue: the row from the next wave, the horizontal maximum, the stores
    auto part2 = [&](int jj) {
        if (wave < 3) {
            const char* ex = exch + ((jj & 1) * 4 + wave + 1) * SWK_EXCH + xo;
            const u32x4 a = *(const u32x4*)ex, b = *(const u32x4*)(ex + 1024);
            V[1][0] = pkmax4_h(V[1][0], u32x2{a[0], a[1]});
            V[1][1] = pkmax4_h(V[1][1], u32x2{a[2], a[3]});
            V[1][2] = pkmax4_h(V[1][2], u32x2{b[0], b[1]});
            V[1][3] = pkmax4_h(V[1][3], u32x2{b[2], b[3]});
        }
        const bool first = l15 == 0;
        const int ox = 8 * jj + (l15 >> 1);
        const bool st = jj >= j0 && !(l15 & 1) && ox < p.pool_w2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = 7 * pty + 2 * wave + r;
            const bool str = st && oy < p.pool_h2 && (r == 0 || wave < 3);
            const unsigned off = str ? orow + (unsigned)(oy * p.o_sh + ox * p.o_sw) * 2u : 0xffffff00u;      // (+ nt * 32 must not wrap)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                u32x2 hm;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t cur = V[r][nt][h];
                    const uint32_t rot = dpp_ror1(cur);
                    const uint32_t left = first ? carry[r][nt][h] : rot;
                    carry[r][nt][h] = rot;                           // lane 0 now holds lane 15's value = this step's column 15
                    hm[h] = pkmax2_h(pkmax2_h(left, cur), dpp_shl1(cur));
                }
                __builtin_amdgcn_raw_buffer_store_b64(hm, ro, off + nt * 32, 0, 0);
            }
        }
    };

    if constexpr (IN == 0) issue_patch(js, 0);
    else {
        issue_raw(js);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert_raw(0, js);
    }
    for (int j = js; j < j1; ++j) {
        const int b = (j - js) & 1;
        // IN = 0, COUNTED wait (round 4): the eight output stores of part2 are the youngest vector-memory operations of a wave here
        // and nothing has to wait for them: vmcnt(8) retires the patch's LDS-DMA in front of them and leaves the stores in flight
        // (packed-input stem 159 -> 153 us per 64 panoramas).  With the raw-input forms (IN >= 1: the staged rows are awaited before
        // their conversion) the same change measured no gain (183 -> 185 us): they keep the plain waits.
        if constexpr (IN == 0) {
            if (j <= js + 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // (raw: __syncthreads() would add its own vmcnt(0))
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // patch j is in LDS; every wave is done with step j - 1's patch and has published its row
        const bool more = j + 1 < j1;
        if (more) {
            if constexpr (IN == 0) issue_patch(j + 1, b ^ 1);
            else issue_raw(j + 1);
        }
        if (j > js) part2(j - 1);
        // ---- the K loop: 7 tap rows, X fragments from the patch, W from registers
        f32x4 acc[4][4];
        const char* pb = patch + b * SWK_PBUF + pfo;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            bf16x8 xf[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8*)(pb + (2 * mt + ky) * SWK_PROWB);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (ky == 0) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};       // an inline-constant C operand, not 64 v_mov
                    if (ky < WREG) {
                        mfma16<4>(acc[nt][mt], wf[ky][nt], wf[ky][nt], xf[mt], xf[mt]);
                    } else {
                        const bf16x8 w = *(const bf16x8*)(wl6 + nt * 1024);
                        mfma16<4>(acc[nt][mt], w, w, xf[mt], xf[mt]);
                    }
                }
        }
        // ---- first half of the epilogue: BN + ReLU -> packed fp16, seam mask, vertical maxima, row 4w to the previous wave
        const bool colok = 16 * j + l15 < p.pool_w1;
        float hi_[4];                       // upper clamp of a conv row's values: fp16 saturation, or 0 outside the conv map
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) hi_[mt] = (colok && ((rowok >> mt) & 1u)) ? 65504.f : 0.f;
        u32x2 r0[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 sc = *(const f32x4*)(ss + nt * 16 + 4 * lq), sh = *(const f32x4*)(ss + 64 + nt * 16 + 4 * lq);
            u32x2 r[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][e] * sc[e] + sh[e];
                r[mt] = u32x2{pack2_h_clamp(v[0], v[1], hi_[mt]), pack2_h_clamp(v[2], v[3], hi_[mt])};
            }
            V[0][nt] = pkmax4_h(pkmax4_h(r[0], r[1]), r[2]);
            V[1][nt] = pkmax4_h(r[2], r[3]);
            r0[nt] = r[0];
        }
        if (wave > 0) {
            char* ex = exch + ((j & 1) * 4 + wave) * SWK_EXCH + xo;
            *(u32x4*)ex = u32x4{r0[0][0], r0[0][1], r0[1][0], r0[1][1]};
            *(u32x4*)(ex + 1024) = u32x4{r0[2][0], r0[2][1], r0[3][0], r0[3][1]};
        }
        if constexpr (IN >= 1) {
            if (more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's staged rows of step j + 1 have landed
                convert_raw(b ^ 1, j + 1);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    part2(j1 - 1);
#endif
}

static WalkGeo walk_geo(int n, const IgemmParams& p) {
    WalkGeo g = {};
    g.nblk = (p.pool_w1 + 15) / 16;
    const long long strips = (long long)n * p.pool_ty;
    const int slots = 512;                 // 256 CUs x 2 workgroups
    // segments per strip: the cut that minimises (rounds of workgroups) x (steps of a workgroup, warm-up step included)
    long long best = -1;
    for (int ns = 1; ns <= g.nblk && ns <= 32; ++ns) {
        const int sps = (g.nblk + ns - 1) / ns;
        const int nseg = (g.nblk + sps - 1) / sps;
        const long long rounds = (strips * nseg + slots - 1) / slots;
        const long long cost = rounds * (sps + (nseg > 1 ? 1 : 0) + 1);      // + 1: a workgroup's fixed prologue / drain
        if (best < 0 || cost < best) { best = cost; g.nseg = nseg; g.sps = sps; }
    }
    g.items = (int)(strips * g.nseg);
    g.chunk = (g.items + 7) / 8;
    return g;
}

template <int IN>
static int launch_stem_walk(IgemmParams& p, const StemRaw& raw, hipStream_t s) {
    constexpr int lds = swk_lds<IN>();
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)stem_walk_kernel<IN>, lds, attr_done)) return AGP_E_LAUNCH;
    const int n = p.M / (p.pool_h1 * p.pool_w1);
    WalkGeo g = walk_geo(n, p);
    const long long o_bytes = (long long)n * p.o_sn * 2;
    if (o_bytes >= 0xffffff00ll) return AGP_E_BADARG;
    g.o_bytes = (uint32_t)o_bytes;
    AGP_LAUNCH(stem_walk_kernel<IN>, dim3(g.chunk * 8), dim3(256), lds, s, p, raw, g);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

}  // namespace agp_igemm

// Can stem_walk_kernel<1> read this fp32 image?  Unit column stride, 16-byte aligned base / strides / width (a chunk of 4
// pixels is then inside the row or outside it), everything addressable by a 32-bit byte offset.
bool agp_internal_stem_walk_reads(const agp_igemm::StemRaw& raw, int n) {
    if (raw.sw != 1 || (raw.sn | raw.sc | raw.sh) % 4 || raw.w % 4 || ((uintptr_t)raw.x & 15)) return false;
    if (raw.sn < 0 || raw.sc < 0 || raw.sh < 0) return false;
    const long long last = (long long)(n - 1) * raw.sn + 2 * raw.sc + (long long)(raw.h - 1) * raw.sh + raw.w;
    return last * 4 < (1ll << 31);
}

// ... and these uint8 camera tiles?  Tile width a multiple of 32 (a step's columns lie in one camera), 16-byte aligned base,
// 31-bit byte offsets.
bool agp_internal_stem_walk_reads_u8(const agp_igemm::StemRaw& raw, int n) {
    if (raw.wcam <= 0 || raw.wcam % 32 || raw.ncam <= 0 || raw.w != raw.ncam * raw.wcam || ((uintptr_t)raw.x & 15)) return false;
    return (long long)n * raw.ncam * raw.h * raw.wcam * 3 < (1ll << 31);
}

// kind 0: packed NHWC4 input (p.x_hi); 1: the fp32 image described by `raw` (agp_internal_stem_walk_reads must hold);
// 2: uint8 camera tiles (agp_internal_stem_walk_reads_u8 must hold)
int agp_internal_stem_walk(agp_igemm::IgemmParams& p, int kind, agp_igemm::StemRaw raw, hipStream_t s) {
    using namespace agp_igemm;
    if (kind == 0) return launch_stem_walk<0>(p, raw, s);
    if (kind == 2) {
        const int n = p.M / (p.pool_h1 * p.pool_w1);
        raw.bytes = (uint32_t)((long long)n * raw.ncam * raw.h * raw.wcam * 3);
        return launch_stem_walk<2>(p, raw, s);
    }
    if (kind == 1) {
        const int n = p.M / (p.pool_h1 * p.pool_w1);
        raw.bytes = (uint32_t)(((long long)(n - 1) * raw.sn + 2 * raw.sc + (long long)(raw.h - 1) * raw.sh + raw.w) * 4);
        return launch_stem_walk<1>(p, raw, s);
    }
    return AGP_E_BADARG;
}
