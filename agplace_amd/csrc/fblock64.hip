// fblock64.hip -- a whole ResNet BasicBlock on 64-channel fp16 maps as ONE kernel (round 4):
//     out = relu( bn2(conv3x3(relu(bn1(conv3x3(x))))) + x ),   64 -> 64 -> 64 channels, stride 1
// (torchvision BasicBlock of layer1 as the reference drives it: network_mm/image_fe.py:102, network/image_fe.py:117; eval-mode
// BatchNorm folded into per-channel scale / shift).
//
// Why.  On the layer-1 maps (64 channels, 56 x 336 per panorama) a 3x3 conv has 192..288 flop per byte of its maps: below
// the ridge of the chip (2.5 PFLOP/s over ~6.3 TB/s = ~400), and the two convs of a block moved 900 MB per launch pair at
// the bench size (input, intermediate written and re-read, residual re-read, output).  Here the intermediate map never
// leaves the CU and the residual comes from the input rows that are already in LDS: a block reads its input once and writes
// its output once (~376 MB algorithmic + the strips' halo columns).
//
// How.  A workgroup owns a column STRIP of 28 pixels and walks down it over a range of the map's rows -- all images of the
// batch are one column of "virtual rows" (image i owns rows i (H+2) .. incl. its two halo rows, which is how the planes lie
// in memory), so a walk crosses image boundaries without a pipeline drain.  LDS holds two rings of 16 rows x 32 pixel slots
// x 128 B (64 channels fp16), XOR-swizzled per slot pair: XR = input rows (strip columns -2 .. +29, fetched by LDS-DMA one
// step = 4 rows ahead), IR = intermediate rows (columns -1 .. +30).  Eight waves = two roles (two waves per SIMD, one of
// each role -- profiles/README.md, round 4: the "weights in registers, activations from LDS" loop runs at 1330 TFLOP/s
// with two waves per SIMD and at 1040-1130 with one):
//   waves 0..3 (conv1): wave (cA, pA) holds the 3x3x64 weights of 32 output channels of conv1 in REGISTERS for the whole
//     launch (36 MFMA A-fragments = 144 VGPRs) and computes, per step, two intermediate rows (one 32 x 32 MFMA tile each:
//     36 x {ds_read_b128 of an X fragment at (row + ky, slot + kx), v_mfma_f32_32x32x16_f16}); BN + ReLU + fp16 into IR
//     (zeros outside the image: that is conv2's padding);
//   waves 4..7 (conv2): the same with conv2's weights on IR rows, one step behind; epilogue BN + residual (from XR) +
//     ReLU -> global.
// One s_barrier per step (~70 MFMAs per wave) and no load on any wave's critical path.  A 32-slot tile carries 30 (conv1) /
// 28 (conv2) useful pixels: 1.14 x the algorithmic flops are issued.
// The MFMA sequence of an output element (ky, 32-channel chunk, kx, 16-channel K-step) and the epilogue arithmetic are those
// of igemm_kxr2.hip, and the intermediate is rounded to fp16 exactly as the stored map would be: the M16 = false form is
// bit-identical to two agp_conv2d_fwd launches (tests/test_gpu_kernels.py).
// M16 = true (the production form, AGP_FB_M16=0 turns it off): the same kernel on v_mfma_f32_16x16x32_f16 -- 2 x 2 tiles of 16
// channels x 16 pixels per K = 32 step, the same LDS bytes, registers and cycles, but this MFMA-bound loop holds a higher clock
// on that shape (tools/ubench/fb_loop.hip: 1470 against 1354 TFLOP/s; MI355X_MICROARCH.md, DVFS give-back item 7).  A K = 32 MFMA
// sums its 32 products in another order than two K = 16 ones, so this form is bit-identical to igemm_kxr2's own 16x16x32
// variant (AGP_KXR2_VARIANT=16), not to its default; it is checked against the 32x32x16 form to fp16 rounding and against fp64.
// POOL (layer1's last block): the per-channel sums of the stored output for the level mean (fuse_block_toshallow.py:82) are taken
// ON THE MATRIX PIPE: the row's fp16 tile sits in the wave's strip anyway; read back transposed (ds_read_b64_tr_b16: k = pixel) it
// is the B operand of two v_mfma_f32_16x16x32_f16 whose A operand is the 0 / 1 mask of the stored pixels, accumulating sum_px
// mask(px) * y[px][ch] in fp32 (exact products, fixed order) over the PAIR of rows (y, y + 1; y even in padded coordinates) the
// wave has just computed -- no vector instruction per value, 8 accumulator registers that live for the last MFMAs of a pair only.  (Per-lane sums in registers cost 16 VGPRs the kernel does not have,
// LDS float atomics 3 x the kernel's time, read-modify-write in LDS + a DPP reduction +50 us: profiles/README.md, round 4.)
// The wave then writes the 32 sums: [pair of virtual rows][strip][64].  The pairs are image-relative units (row segments start at
// even rows, images have an even number of padded rows), so the sums do not depend on where a workgroup's walk starts or where
// an image sits in the batch; agp_bblock64_pool_finish adds an image's pairs and strips in a fixed order.

#include <type_traits>
#include <utility>

#include "igemm_params.hpp"

namespace agp_fb {

constexpr int MAXP = 4;            // problems per launch (the query and the database trunk's block of one layer, ...)
constexpr int TW = 28;             // strip width (output pixels); 32 slots per ring row
constexpr int RING = 16;           // rows per ring
constexpr int ROWB = 32 * 128;     // bytes of a ring row
constexpr int TAB_OFF = 2 * RING * ROWB + 512;      // (512 B guard: slots 32, 33 of the last IR row)
constexpr int STRIP_OFF = TAB_OFF + 4 * 64 * 4;    // conv2 waves: wave-private 32 pixels x 64 B output strips (accumulator -> line layout)
template <bool POOL> constexpr int lds_bytes() { return STRIP_OFF + 4 * 2048; }

struct Problem {
    const void* x; void* out;
    const void* w1; const void* w2;          // fp16 [64][3][3][64]
    const float *s1, *t1, *s2, *t2;          // folded BatchNorm: y = conv * s + t
    float* pool;                             // optional [VR / 2][nstrips][64] pair sums of the stored output
    uint32_t bytes;                          // bytes of one map
    int VR, HP, W, pitch;                    // virtual rows n (H+2), H+2, interior width, bytes of a padded row
    int nstrips, segs;                       // column strips; row segments per strip
    int dbg;                                 // timing-only experiments (AGP_FB_DBG), 0 in production
    FastDiv d_hp, d_segs;
};
struct Group {
    Problem p[MAXP];
    int wg_end[MAXP];
    int nprob;
};

__device__ __forceinline__ int perm23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <bool POOL, bool M16>
__global__ void __launch_bounds__(512, 2) fblock64_kernel(Group g) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XR = smem;
    char* const IR = smem + RING * ROWB;
    float* const tab = (float*)(smem + TAB_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, cA = wave & 1, pA = (wave >> 1) & 1;
    // MFMA lane coordinates.  32x32x16: lane = (pixel l31, K half lh).  16x16x32: lane = (pixel a of a 16-pixel tile, K quarter kq)
    const int l31 = lane & 31, lh = lane >> 5;
    const int la = lane & 15, kq = lane >> 4;

    // ---- workgroup -> (problem, strip, row segment)
    const int bid = blockIdx.x;
    int pid = 0, lb = bid;
    {
        const int e0 = g.wg_end[0], e1 = g.wg_end[1], e2 = g.wg_end[2], np = g.nprob;
        if (np > 1 && bid >= e0) { pid = 1; lb = bid - e0; }
        if (np > 2 && bid >= e1) { pid = 2; lb = bid - e1; }
        if (np > 3 && bid >= e2) { pid = 3; lb = bid - e2; }
    }
    const Problem& p = g.p[pid];
    const int VR = p.VR, HP = p.HP, W = p.W, pitch = p.pitch, segs = p.segs;
    const FastDiv d_hp = p.d_hp;
    const int dbg = p.dbg;
    const int strip = (int)fdiv((uint32_t)lb, p.d_segs);
    const int sk = lb - strip * segs;
    // segment [Ra, Rb) of the strip's virtual rows, both even
    const int Ra = (int)(((uint32_t)sk * (uint32_t)VR / (uint32_t)segs) & ~1u);         // (segs * VR < 2^32: checked by the host)
    const int Rb = sk + 1 == segs ? VR : (int)(((uint32_t)(sk + 1) * (uint32_t)VR / (uint32_t)segs) & ~1u);
    const int x0 = strip * TW;                          // interior column of output slot 0
    const int NS = (Rb - Ra + 3) >> 2;

    // ---- this wave's weights, 36 A fragments, in the K order igemm_kxr2 runs.
    // 32x32x16: w[((ky 2 + cc) 3 + kx) 2 + ks] = row perm23(l31) of the wave's 32 channels, K = 16 ks + 8 lh .. of the tap's chunk cc;
    //           accumulator registers 8 u .. 8 u + 7 of a lane are then channels 8 lh + 16 u .. + 7 of its pixel.
    // 16x16x32: w[(((ky 2 + cc) 3 + kx) 2 + ct]: channel tile ct's row la = channel 8 (la >> 2) + 4 ct + (la & 3), K = 8 kq ..;
    //           a lane's registers of the two channel tiles are then channels 8 kq .. 8 kq + 7 of its pixel.
    f16x8 w[36];
    {
        const bf16_t* wb = (const bf16_t*)(role ? p.w2 : p.w1);
        int i = 0;
        if constexpr (!M16) {
            const bf16_t* wp = wb + (size_t)(cA * 32 + perm23(l31)) * 576 + lh * 8;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks, ++i)
                            w[i] = *(const f16x8*)(wp + (ky * 3 + kx) * 64 + cc * 32 + ks * 16);
        } else {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct, ++i)
                            w[i] = *(const f16x8*)(wb + (size_t)(cA * 32 + 8 * (la >> 2) + 4 * ct + (la & 3)) * 576 + (ky * 3 + kx) * 64 + cc * 32 + kq * 8);
        }
    }
    if (tid < 64) {
        tab[tid] = p.s1[tid];
        tab[64 + tid] = p.t1[tid];
        tab[128 + tid] = p.s2[tid];
        tab[192 + tid] = p.t2[tid];
    }

    // ---- LDS addressing.  Slot s of a ring row: 128 B at s * 128, its 16-byte chunk c stored at chunk c ^ ((s >> 1) & 7)
    // fragment reads: 32x32x16: slot l31 + kx, chunk lh + 4 cc + 2 ks (by XOR); 16x16x32: slot la + kx (+ 16: + 2048 B), chunk kq + 4 cc
    int A[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int s = (M16 ? la : l31) + kx;
        A[kx] = s * 128 + (((M16 ? kq : lh) ^ ((s >> 1) & 7)) << 4);
    }
    // epilogue: a lane ends up with TWO 16-byte chunks (8 channels each) of the wave's 64-byte channel half:
    //   32x32x16: chunks lh and 2 + lh of pixel l31;   16x16x32: chunk kq of pixels la and 16 + la
    const int px0 = M16 ? la : l31, px1 = M16 ? la + 16 : l31;
    const int ch0 = M16 ? kq : lh, ch1 = M16 ? kq : lh + 2;                                       // chunk inside the wave's half
    auto slot_off = [&](int s, int c) { return s * 128 + (((4 * cA + c) ^ ((s >> 1) & 7)) << 4); };
    const int wr0 = slot_off(px0, ch0), wr1 = slot_off(px1, ch1);                                  // IR slots: conv1's output
    const int rr0 = slot_off(px0 + 2, ch0), rr1 = slot_off(px1 + 2, ch1);                          // XR slots + 2: the residual
    const float* const tb0 = tab + role * 128 + 32 * cA + 8 * ch0;
    const float* const tb1 = tab + role * 128 + 32 * cA + 8 * ch1;
    // conv1: intermediate column x0 - 1 + px must lie inside the image, else it is conv2's zero padding
    const bool cin0 = (x0 - 1 + px0 >= 0) && (x0 - 1 + px0 < W), cin1 = (x0 - 1 + px1 >= 0) && (x0 - 1 + px1 < W);
    // conv2: output column x0 + px is stored if it belongs to the strip and to the image
    const bool sok0 = px0 < TW && x0 + px0 < W, sok1 = px1 < TW && x0 + px1 < W;
    // conv2 output through the wave's strip: the lane's two chunks go in, and come back as line layout -- store instruction i covers
    // pixels 16 i .. 16 i + 15, four lanes per pixel = 64 contiguous bytes (a store of the accumulator layout touches 32 lines with
    // 32 B each: 26 us of a 216 us launch, profiles/README.md round 4).  64-B strip rows, chunk ^ ((pixel >> 1) & 3).
    char* const ostrip = smem + STRIP_OFF + (wave & 3) * 2048;
    auto strip_off = [&](int px, int c) { return px * 64 + ((c ^ ((px >> 1) & 3)) << 4); };
    const int sw0 = strip_off(px0, ch0), sw1 = strip_off(px1, ch1);
    const int spx = lane >> 2;                                                              // line layout: pixel 16 i + spx, chunk lane & 3
    const int sr_ = strip_off(spx, lane & 3);                                               // + 1024 i  (16 i does not change the swizzle term)
    const int gline = (x0 + 1 + spx) * 128 + 64 * cA + 16 * (lane & 3);                     // + 2048 i
    const bool st_ok0 = spx < TW && x0 + spx < W, st_ok1 = spx + 16 < TW && x0 + spx + 16 < W;

    // ---- LDS-DMA of input rows (conv1 waves; wave w4 fetches slots 8 w4 .. 8 w4 + 7 of a row: 1 KB per instruction)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.bytes, 0x00020000);
    const int pc = wave & 3;
    int dvoff;
    {
        const int s = pc * 8 + (lane >> 3);
        int xp = x0 - 1 + s;                             // padded column of slot s (slot 0 = interior column x0 - 2)
        xp = xp < 0 ? 0 : (xp > W + 1 ? W + 1 : xp);     // clamped columns feed masked / unstored outputs only
        dvoff = xp * 128 + (((lane & 7) ^ ((s >> 1) & 7)) << 4);
    }
    auto issue_rows = [&](int R0, int cnt) {
#pragma unroll 4
        for (int r = 0; r < cnt; ++r) {
            const int R = R0 + r;
            const int Rc = R < 0 ? 0 : (R >= VR ? VR - 1 : R);   // rows outside the batch feed zeroed intermediate rows only
            const int so = __builtin_amdgcn_readfirstlane(Rc * pitch);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(XR + (R & (RING - 1)) * ROWB + pc * 1024), 16, dvoff, so, 0, 0);
        }
    };
    auto is_real = [&](int R) -> bool {                  // an image row (not a halo row, inside the batch); wave-uniform
        if (R < 0 || R >= VR) return false;
        const int yy = R - (int)fdiv((uint32_t)R, d_hp) * HP;
        return yy >= 1 && yy <= HP - 2;
    };
    // The accumulators of one output row: 32 pixels x this wave's 32 channels (16 registers either way)
    struct Acc {
        f32x16 m;                                        // 32x32x16
        f32x4 t[2][2];                                   // 16x16x32: [channel tile][pixel tile]
    };
    auto acc_zero = [](Acc& a) {
        if constexpr (!M16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a.m[r] = 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) a.t[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // Two rows at once: output rows R and R + 1 of `ring`'s convolution.  The rows share two of their input rows, so the pair reads
    // 4 x 12 KB-sized X fragments for its 72 (144) MFMAs instead of 6 x 12 (36 fragment reads per 36 MFMAs on two waves per SIMD
    // keep the LDS 60-75 % busy).  Each accumulator still sees its own MFMAs in igemm_kxr2's order.  `mid()` is issued among the
    // first MFMAs (deferred stores), `epi0()` when row R is complete, in front of the last MFMAs of row R + 1.
    auto tile2 = [&](const char* ring, int R, Acc& a0, Acc& a1, auto&& mid, auto&& epi0) {
        acc_zero(a0);
        acc_zero(a1);
        // The fragment reads are SOFTWARE-PIPELINED by hand, PD read groups ahead of the MFMAs that use them (group = one
        // (row q, chunk cc, tap kx[, K-step ks]): two fragments in the 16x16x32 form, one in the 32x32x16 form).  Left to itself
        // hipcc issues a group's reads right in front of its own s_waitcnt lgkmcnt(0) and MFMAs -- zero distance: the wave sits out
        // a full LDS latency every 8 MFMAs and only its SIMD partner covers it (profiles/README.md round 4: 39 % of the wave
        // cycles parked).  Everything is unrolled: buffer indices are compile-time constants.
        constexpr int NG = M16 ? 24 : 48, PD = 2, NF = M16 ? 2 : 1;
        const char* rb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rb[q] = ring + __builtin_amdgcn_readfirstlane(((R - 1 + q) & (RING - 1)) * ROWB);
        f16x8 fb[PD + 1][NF];
        auto load_group = [&](auto gc, auto slotc) {
            constexpr int g = decltype(gc)::value, sl = decltype(slotc)::value;
            if constexpr (M16) {
                constexpr int q = g / 6, cc = (g % 6) / 3, kx = g % 3;
                fb[sl][0] = *(const f16x8*)(rb[q] + (A[kx] ^ (cc << 6)));
                fb[sl][1] = *(const f16x8*)(rb[q] + (A[kx] ^ (cc << 6)) + 2048);
            } else {
                constexpr int q = g / 12, cc = (g % 12) / 6, kx = (g % 6) / 2, ks = g % 2;
                fb[sl][0] = *(const f16x8*)(rb[q] + (A[kx] ^ ((cc * 4 + ks * 2) << 4)));
            }
        };
        auto do_group = [&](auto gc, auto slotc) {
            constexpr int g = decltype(gc)::value, sl = decltype(slotc)::value;
            if constexpr (M16) {
                constexpr int q = g / 6, j = (g % 6) * 2;            // j = ((cc 3 + kx) 2: the weight fragment pair of this tap / chunk
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if constexpr (q <= 2) {
                        a0.t[ct][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[q * 12 + j + ct], fb[sl][0], a0.t[ct][0], 0, 0, 0);
                        a0.t[ct][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[q * 12 + j + ct], fb[sl][1], a0.t[ct][1], 0, 0, 0);
                    }
                    if constexpr (q >= 1) {
                        a1.t[ct][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(q - 1) * 12 + j + ct], fb[sl][0], a1.t[ct][0], 0, 0, 0);
                        a1.t[ct][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(q - 1) * 12 + j + ct], fb[sl][1], a1.t[ct][1], 0, 0, 0);
                    }
                }
            } else {
                constexpr int q = g / 12, j = g % 12;
                if constexpr (q <= 2) a0.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[q * 12 + j], fb[sl][0], a0.m, 0, 0, 0);
                if constexpr (q >= 1) a1.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(q - 1) * 12 + j], fb[sl][0], a1.m, 0, 0, 0);
            }
        };
        auto step = [&](auto gc, auto&& self) -> void {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < NG) {
                if constexpr (g + PD < NG) load_group(std::integral_constant<int, g + PD>{}, std::integral_constant<int, (g + PD) % (PD + 1)>{});
                __builtin_amdgcn_sched_barrier(0);                   // (or the machine scheduler sinks the reads back to their uses)
                do_group(gc, std::integral_constant<int, g % (PD + 1)>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (g == (M16 ? 4 : 8)) mid();
                if constexpr (g == 3 * NG / 4 - 1) epi0();           // row R is complete: its epilogue in front of row R + 1's last MFMAs
                self(std::integral_constant<int, g + 1>{}, self);
            }
        };
        load_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        load_group(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        if constexpr (PD > 2) load_group(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 0>{}, step);
    };
    // BatchNorm of the lane's two chunks: v[0..7] = chunk 0's 8 channels, v[8..15] = chunk 1's
    auto bn = [&](const Acc& a, float* v) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float* tb = u ? tb1 : tb0;
            const f32x4 s0 = *(const f32x4*)(tb), s1 = *(const f32x4*)(tb + 4);
            const f32x4 t0 = *(const f32x4*)(tb + 64), t1 = *(const f32x4*)(tb + 64 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c0 = M16 ? a.t[0][u][e] : a.m[8 * u + e], c1 = M16 ? a.t[1][u][e] : a.m[8 * u + 4 + e];
                v[8 * u + e] = c0 * s0[e] + t0[e];
                v[8 * u + 4 + e] = c1 * s1[e] + t1[e];
            }
        }
    };
    // conv1: intermediate rows R, R + 1 (a row that is not an image row is conv2's zero padding)
    auto write_ir = [&](int R, const Acc& acc, bool real) {
        char* const dst = IR + __builtin_amdgcn_readfirstlane((R & (RING - 1)) * ROWB);
        float v[16];
        bn(acc, v);
        u32x4 o0 = pack8_h_lo(v, 0.f), o1 = pack8_h_lo(v + 8, 0.f);
        if (!(real && cin0)) o0 = u32x4{0u, 0u, 0u, 0u};
        if (!(real && cin1)) o1 = u32x4{0u, 0u, 0u, 0u};
        *(u32x4*)(dst + wr0) = o0;
        *(u32x4*)(dst + wr1) = o1;
    };
    auto conv1_pair = [&](int R) {
        const bool real0 = is_real(R), real1 = is_real(R + 1);
        Acc a0, a1;
        if (real0 || real1) {
            tile2(XR, R, a0, a1, [] {}, [&] { write_ir(R, a0, real0); });
            write_ir(R + 1, a1, real1);
        } else {
            acc_zero(a0);
            write_ir(R, a0, false);
            write_ir(R + 1, a0, false);
        }
    };
    // POOL: channel sums on the matrix pipe (see the header).  A = ones (the pixels that are not stored are zeroed on their way into
    // the strip), B = the strip read transposed: block (ct, hh) = pixels 8 (lane >> 4) + 4 hh + q, channels 16 ct + 4 p .. + 3,
    // lane = 16 . + 4 q + p
    f32x4 pacc[2];
    // conv2's stores are DEFERRED: a row's line-layout registers leave for memory among later MFMAs of the wave, when the strip
    // reads have long landed (issued right behind the epilogue they cost the wave two LDS round trips per tile): row R's behind
    // row R + 1's epilogue, row R + 1's among the first MFMAs of the wave's next pair
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.bytes, 0x00020000);
    u32x4 pl0 = {0u, 0u, 0u, 0u}, pl1 = {0u, 0u, 0u, 0u};
    int poff = -1;                                       // byte offset of the pending row, -1 = nothing pending (wave-uniform)
    const int gl0 = st_ok0 ? gline : -(1 << 30), gl1 = st_ok1 ? gline + 2048 : -(1 << 30);
    auto store_lines = [&](const u32x4& l0, const u32x4& l1, int off) {
        // a masked lane's / an empty slot's offset lies past num_records: the store is dropped, no branch among the MFMAs
        const int v0 = (off >= 0 && gl0 >= 0) ? off + gl0 : -16, v1 = (off >= 0 && gl1 >= 0) ? off + gl1 : -16;
        if (!(dbg & 1)) {
            __builtin_amdgcn_raw_buffer_store_b128(l0, ro, v0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(l1, ro, v1, 0, 0);
        }
    };
    auto flush_pending = [&]() { store_lines(pl0, pl1, poff); poff = -1; };
    // BN + residual + ReLU of one row -> fp16 -> the wave's strip -> line layout registers (l0, l1)
    auto epi2 = [&](int R, const Acc& acc, bool real, u32x4& l0, u32x4& l1, bool first) {
        const char* const rsrc = XR + __builtin_amdgcn_readfirstlane((R & (RING - 1)) * ROWB);
        u32x4 r0 = {0u, 0u, 0u, 0u}, r1 = r0;
        if (!(dbg & 16)) { r0 = *(const u32x4*)(rsrc + rr0); r1 = *(const u32x4*)(rsrc + rr1); }
        float v[16], rf[16];
        bn(acc, v);
        unpack8_h(r0, rf);
        unpack8_h(r1, rf + 8);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += rf[e];
        u32x4 o0 = pack8_h_lo(v, 0.f), o1 = pack8_h_lo(v + 8, 0.f);
        // (LDS operations of one wave execute in order: the reads below see the writes above, and the next row's writes come
        // after these reads)
        if constexpr (POOL) {                            // dead columns: never stored, and they must not count
            if (!sok0) o0 = u32x4{0u, 0u, 0u, 0u};
            if (!sok1) o1 = u32x4{0u, 0u, 0u, 0u};
        }
        *(u32x4*)(ostrip + sw0) = o0;
        *(u32x4*)(ostrip + sw1) = o1;
        l0 = *(const u32x4*)(ostrip + sr_);
        l1 = *(const u32x4*)(ostrip + sr_ + 1024);
        if constexpr (POOL) {
            if (first) { pacc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; pacc[1] = pacc[0]; }
            {
                // (no branch among the MFMAs, and the transposed reads need EXEC all ones: a row that is not an image row multiplies
                // by A = 0 instead of being skipped)
                const int tpx = 8 * (lane >> 4) + ((lane >> 2) & 3), pp = lane & 3;
                const int tr0 = tpx * 64 + (((pp >> 1) ^ ((tpx >> 1) & 3)) << 4) + 8 * (pp & 1);      // hh = 1: 4 pixels on = (+ 256) ^ 32
                typedef __attribute__((ext_vector_type(4))) short s16x4;
                const uint32_t one2 = real ? 0x3C003C00u : 0u;
                const f16x8 ones = __builtin_bit_cast(f16x8, u32x4{one2, one2, one2, one2});
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ostrip + (tr0 ^ (32 * ct))));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ostrip + (((tr0 + 256) ^ 32) ^ (32 * ct))));
                    const bf16x8 bf = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, __builtin_bit_cast(f16x8, bf), pacc[ct], 0, 0, 0);
                }
            }
        }
    };
    auto conv2_pair = [&](int R) {
        const bool real0 = is_real(R), real1 = is_real(R + 1);
        if (!(real0 || real1)) {
            if constexpr (POOL) { pacc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; pacc[1] = pacc[0]; }
            return;
        }
        Acc a0, a1;
        u32x4 q0, q1;
        tile2(IR, R, a0, a1, flush_pending, [&] { epi2(R, a0, real0, q0, q1, true); });
        epi2(R + 1, a1, real1, pl0, pl1, false);
        store_lines(q0, q1, real0 ? __builtin_amdgcn_readfirstlane(R * pitch) : -1);
        poff = real1 ? __builtin_amdgcn_readfirstlane((R + 1) * pitch) : -1;
    };
    // after the pair (R, R + 1): its channel sums leave (every row of the 16 x 16 result holds them: row 0)
    auto pool_flush = [&](int R) {
        if constexpr (POOL) {
            if (p.pool != nullptr && lane < 16) {
                float* o = p.pool + ((size_t)(R >> 1) * p.nstrips + strip) * 64 + 32 * cA + lane;
                o[0] = pacc[0][0];
                o[16] = pacc[1][0];
            }
        }
    };

    // ---- prologue: input rows Ra - 2 .. Ra + 5, then the two intermediate rows in front of the first step
    if (role == 0) issue_rows(Ra - 2, 8);
    // (the BUILTIN, not inline asm: the compiler's wait-count pass must see that the weight loads have landed, or it repeats the
    // vmcnt countdown of their first use inside the step loop)
    __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (role == 0 && pA == 0) conv1_pair(Ra - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- steps: conv1 writes intermediate rows Ra + 1 + 4 t .. + 3 while conv2 turns rows Ra + 4 (t - 1) .. + 3 into output
    for (int t = 0; t <= NS; ++t) {
        if (role == 0) {
            if (t < NS) {
                if (t + 1 < NS && !(dbg & 2)) issue_rows(Ra + 4 * t + 6, 4);       // what conv1 reads in step t + 1
                const int R = Ra + 1 + 4 * t + 2 * pA;
                if (R <= Rb) conv1_pair(R);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (t >= 1) {
            const int R = Ra + 4 * (t - 1) + 2 * pA;
            if (R < Rb) {                                            // (Rb is even: R + 1 < Rb too)
                conv2_pair(R);
                pool_flush(R);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(dbg & 4)) __builtin_amdgcn_s_barrier();
    }
    if (role == 1) flush_pending();
#endif
}

}  // namespace agp_fb

using namespace agp_fb;

static int fb_num_cus() {
    static int n = -1;
    if (n < 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

static bool fb_desc_ok(const agp_bblock64_desc* d) {
    return d && d->in && d->out && d->w1 && d->w2 && d->scale1 && d->shift1 && d->scale2 && d->shift2 && d->n > 0 && d->h > 0 && d->w > 0 &&
           (d->h & 1) == 0 && (int64_t)d->n * (d->h + 2) * (d->w + 2) * 128 < (1ll << 31);
}

extern "C" int64_t agp_bblock64_pool_floats(const agp_bblock64_desc* d) {
    if (!fb_desc_ok(d)) return 0;
    return (int64_t)d->n * ((d->h + 2) / 2) * ((d->w + TW - 1) / TW) * 64;
}

extern "C" int agp_bblock64_fwd_grouped(const agp_bblock64_desc* descs, int n, void* stream) {
    if (!descs || n < 1 || n > MAXP) return AGP_E_BADARG;
    Group g = {};
    g.nprob = n;
    bool pool = false;
    int64_t total = 0;                      // strip rows of all problems
    for (int i = 0; i < n; ++i) {
        const agp_bblock64_desc* d = descs + i;
        if (!fb_desc_ok(d)) return AGP_E_BADARG;
        Problem& p = g.p[i];
        p.x = d->in; p.out = d->out; p.w1 = d->w1; p.w2 = d->w2;
        p.s1 = d->scale1; p.t1 = d->shift1; p.s2 = d->scale2; p.t2 = d->shift2;
        p.pool = d->pool_partial;
        pool = pool || d->pool_partial;
        p.HP = d->h + 2; p.VR = d->n * p.HP; p.W = d->w; p.pitch = (d->w + 2) * 128;
        p.bytes = (uint32_t)((int64_t)p.VR * p.pitch);
        p.nstrips = (d->w + TW - 1) / TW;
        p.d_hp = make_fastdiv((uint32_t)p.HP);
        p.dbg = AGP_TUNE("FB_DBG", 0);
        total += (int64_t)p.nstrips * p.VR;
    }
    // row segments per strip: about one workgroup per CU, every workgroup the same number of rows (>= 16)
    const int cus = fb_num_cus();
    int wg = 0;
    for (int i = 0; i < n; ++i) {
        Problem& p = g.p[i];
        int64_t s = (int64_t)p.VR * cus / total;
        const int64_t smax = p.VR / 16 > 1 ? p.VR / 16 : 1;
        s = s < 1 ? 1 : (s > smax ? smax : s);
        while (s > 1 && s * (int64_t)p.VR >= (1ll << 32)) --s;
        p.segs = (int)s;
        p.d_segs = make_fastdiv((uint32_t)p.segs);
        wg += p.nstrips * p.segs;
        g.wg_end[i] = wg;
    }
    const int form = descs[0].form == 0 ? 1 : 0;                // kernel template argument M16 (agp_bblock64_desc::form)
    static std::atomic<uint64_t> attr_done[2][2];
    const void* fn = pool ? (form ? (const void*)fblock64_kernel<true, true> : (const void*)fblock64_kernel<true, false>)
                          : (form ? (const void*)fblock64_kernel<false, true> : (const void*)fblock64_kernel<false, false>);
    const int lds = pool ? lds_bytes<true>() : lds_bytes<false>();
    if (!agp_lds_attr(fn, lds, attr_done[pool][form])) return AGP_E_LAUNCH;
    (void)hipGetLastError();
    if (pool && form) hipLaunchKernelGGL((fblock64_kernel<true, true>), dim3(wg), dim3(512), lds, (hipStream_t)stream, g);
    else if (pool) hipLaunchKernelGGL((fblock64_kernel<true, false>), dim3(wg), dim3(512), lds, (hipStream_t)stream, g);
    else if (form) hipLaunchKernelGGL((fblock64_kernel<false, true>), dim3(wg), dim3(512), lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((fblock64_kernel<false, false>), dim3(wg), dim3(512), lds, (hipStream_t)stream, g);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

// mean_out[i][c] = (sum of image i's pair sums [pair][strip][c]) / (h w), fixed order: 16 slices of the entries per channel
// (loads eight deep, adds in entry order), then a fixed tree over the slices
__global__ void __launch_bounds__(1024) bblock64_pool_finish_kernel(const float* __restrict__ partial, int cnt, float inv_hw,
                                                                    float* __restrict__ mean_out) {
    __shared__ float red[16][64];
    const int im = blockIdx.x, c = threadIdx.x & 63, k = threadIdx.x >> 6;
    const float* base = partial + (size_t)im * cnt * 64 + c;      // an image's entries are contiguous
    float s = 0.f;
    for (int b0 = k; b0 < cnt; b0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 16 * u;
            v[u] = b < cnt ? base[(size_t)b * 64] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    red[k][c] = s;
    __syncthreads();
    if (k == 0) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = red[2 * u][c] + red[2 * u + 1][c];
        mean_out[(size_t)im * 64 + c] = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) * inv_hw;
    }
}

extern "C" int agp_bblock64_pool_finish(const float* partial, int n, int h, int w, float* mean_out, void* stream) {
    if (!partial || !mean_out || n <= 0 || h <= 0 || w <= 0 || (h & 1)) return AGP_E_BADARG;
    AGP_LAUNCH(bblock64_pool_finish_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, partial, (h + 2) / 2 * ((w + TW - 1) / TW),
               1.f / ((float)h * (float)w), mean_out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
