// The vector path of a forward as ONE launch: a short program of row-wise operations on [b, 256] fp32 vectors
// (reference network_mm/mm.py:91-129 after the backbones: F.normalize, FuseBlockToShallow's up-dim Linears and
// Neural-ODE blocks, the stage-2 projections, Basic's fc-LayerNorm-ReLU-fc-LayerNorm, stg2fusefc, the weighted final
// sum; models_baseline/dbvanilla2d.py:17-28,81-92 MLP + normalize).  The per-op kernels of fusion.hip need ~27 launches
// for the query network's vector path, each 5-25 us of mostly launch latency on 64 rows; here a workgroup of 16 waves
// owns 16 batch rows for the whole program:
//   * up to VP_NREG vector registers live in LDS as fp32 [16 rows][256] (+4 floats of row padding: the 16 rows of a wave's
//     access fall into different banks); every op reads and writes a lane's OWN elements of a register (lane = batch row
//     l&15, features 16*wave + 4*(l>>4) + r), so registers need no barriers;
//   * LINEAR / FCODE stage their input as split-bf16 planes (double-buffered: ONE barrier per matrix product), wave w
//     multiplies W rows [16w, 16w+16) from global memory (MFMA 16x16x32 bf16, split-bf16 x3, fp32-class results, the
//     arithmetic of fusion.hip); FCODE keeps its W fragments in registers for every step of the solver;
//   * L2NORM / LAYERNORM reduce over the 256 features of a row with two shuffles + a [16 waves][16 rows] LDS exchange.
// The program (<= VP_MAXOPS ops) travels in the kernel arguments; pointers are baked into captured graphs like any other.
// Measured per op (b = 64, 4 workgroups): LINEAR 7-8 us whatever K is -- one CU pulls the whole W (256 KB as bf16 pairs)
// through its own L2 port, where the per-op kernel spread W over 16 CUs; an Euler step of FCODE 1.3 us = one CU's MFMA
// pipes at ~50 % (3 products x 256 x 256 x 16 rows per step); L2NORM / LAYERNORM / WSUM 0.2-0.5 us; launch + first
// loads 8 us.  The query head (5 LINEAR, 3 FCODE, ...) is 115-127 us, the tail 60, the database head 26.
#include <cstring>

#include "fusion_common.hpp"

namespace agp_fusion {

constexpr int VP_RS = 260;                     // floats per register row (256 + pad)
constexpr int VP_YRB = 256 * 2 + 16;           // bytes per row of a split-bf16 plane

struct VpOpD {                                 // device-side mirror of agp_vecprog_op (include/agplace_hip.h)
    int op, dst, r[6], k, act, aux, n;
    float f0;
    int pad;
    const void* p[6];
};

struct VpProgram {
    int nops, b, method, nsteps;
    int nops_a, blocks_a, b_b, pad_;   // two programs in one launch: blocks < blocks_a run ops [0, nops_a) on b rows, the others ops [nops_a, nops) on b_b rows
    float dt[48];
    VpOpD ops[AGP_VECPROG_MAXOPS];
};
static_assert(sizeof(VpProgram) <= 4096, "kernel argument budget");

__global__ __launch_bounds__(FT) void vecprog_kernel(VpProgram P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* regs = (float*)smem;                                            // [NREG][16][VP_RS]
    char* planes = smem + AGP_VECPROG_NREG * FROWS * VP_RS * 4;            // [2 buf][hi, lo][16][VP_YRB]
    float* red = (float*)(planes + 2 * 2 * FROWS * VP_YRB);                // [2 buf][16 waves][16 rows]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 15;
    const int nf = wave * 16 + (lane >> 4) * 4;
    // a second, independent program may ride in the launch on workgroups of its own (agp_vecprog_run2: the database network's
    // head beside the query network's -- two latency chains side by side instead of one behind the other)
    const bool grp_b = (int)blockIdx.x >= P.blocks_a;
    const int op_lo = grp_b ? P.nops_a : 0, op_hi = grp_b ? P.nops : P.nops_a;
    const int brow = ((int)blockIdx.x - (grp_b ? P.blocks_a : 0)) * FROWS + row;
    const bool live = brow < (grp_b ? P.b_b : P.b);
    const int roff = row * VP_RS + nf;
    int pbuf = 0, rbuf = 0;

    auto rd = [&](int r) -> f32x4 { return *(const f32x4*)(regs + r * (FROWS * VP_RS) + roff); };
    auto wr = [&](int r, const f32x4& v) { *(f32x4*)(regs + r * (FROWS * VP_RS) + roff) = v; };
    // sum over the 256 features of this lane's row (every lane of the row gets the total)
    auto rowsum = [&](float s) -> float {
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        float* rb = red + rbuf * (16 * FROWS);
        if (lane < 16) rb[wave * FROWS + row] = s;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += rb[w * FROWS + row];
        rbuf ^= 1;
        return t;
    };
    // operand of a matrix product: register r0 (+ optional add registers)
    auto operand = [&](int r0, int r1, int r2) -> f32x4 {
        f32x4 v = rd(r0);
        if (r1 >= 0) v += rd(r1);
        if (r2 >= 0) v += rd(r2);
        return v;
    };
    auto bias4 = [&](const void* b) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b) v = *(const f32x4*)((const float*)b + nf);
        return v;
    };

    for (int i = op_lo; i < op_hi; ++i) {
        // the op record BY VALUE: one wide scalar load per op (read field by field inside the branches it was a chain of
        // dependent scalar-load round trips: a LINEAR op cost 8 us whatever its size)
        // (explicit scalars, not a struct copy: a by-value copy of a run-time-indexed kernel-argument record goes to scratch)
        const VpOpD* const kop = &P.ops[i];
        struct {
            int op, dst, r[6], k, act, n;
            float f0;
            const void* p[6];
        } o;
        o.op = kop->op; o.dst = kop->dst; o.k = kop->k; o.act = kop->act; o.n = kop->n; o.f0 = kop->f0;
        o.r[0] = kop->r[0]; o.r[1] = kop->r[1]; o.r[2] = kop->r[2]; o.r[3] = kop->r[3]; o.r[4] = kop->r[4]; o.r[5] = kop->r[5];
        o.p[0] = kop->p[0]; o.p[1] = kop->p[1]; o.p[2] = kop->p[2]; o.p[3] = kop->p[3]; o.p[4] = kop->p[4]; o.p[5] = kop->p[5];
        switch (o.op) {
            case AGP_VP_LOAD: {                                  // dst <- global [b, k] (zero beyond k), times *p[1] if given
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (live && nf < o.k) v = *(const f32x4*)((const float*)o.p[0] + (size_t)brow * o.k + nf);
                if (o.p[1]) v = v * ((const float*)o.p[1])[0];
                wr(o.dst, v);
                break;
            }
            case AGP_VP_STORE: {                                 // global [b, 256] <- r[0]
                if (live) *(f32x4*)((float*)o.p[0] + (size_t)brow * 256 + nf) = rd(o.r[0]);
                break;
            }
            case AGP_VP_LINEAR: {                                // dst <- act(W (r0 + r1 + r2) + bias), W [256][k]
                const int K = o.k, nks = K / 32;
                // fragment-major planes [wave][ks][lane][8] (agp_vecprog_op, include/agplace_hip.h): a wave instruction reads 1 KB
                // of consecutive bytes.  (Row-major [256][K] made every instruction touch 16 half-lines 512 bytes apart: 7.7 us
                // per product whatever K was, the weights hot in L2.)
                const bf16_t* wrh = (const bf16_t*)o.p[0] + ((size_t)wave * nks * 64 + lane) * 8;
                const bf16_t* wrl = (const bf16_t*)o.p[1] + ((size_t)wave * nks * 64 + lane) * 8;
                bf16x8 wh[8], wl[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int kk = ks < nks ? ks : nks - 1;
                    wh[ks] = *(const bf16x8*)(wrh + kk * 512);
                    wl[ks] = *(const bf16x8*)(wrl + kk * 512);
                }
                const f32x4 bia = bias4(o.p[2]);
                char* hi = planes + pbuf * (2 * FROWS * VP_YRB);
                char* lo = hi + FROWS * VP_YRB;
                store_state(hi, lo, VP_YRB, lane, wave, operand(o.r[0], o.r[1], o.r[2]));
                __syncthreads();
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
                const int boff = row * VP_YRB + (lane >> 4) * 16;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks < nks) {
                        const bf16x8 bh = *(const bf16x8*)(hi + boff + ks * 64);
                        const bf16x8 bl = *(const bf16x8*)(lo + boff + ks * 64);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], bh, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], bl, a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], bh, a2, 0, 0, 0);
                    }
                }
                pbuf ^= 1;
                f32x4 z = (a0 + a1) + a2 + bia;
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = apply_act(z[r], o.act);
                wr(o.dst, z);
                break;
            }
            case AGP_VP_FCODE: {                                 // dst <- odeint(act(W y + bias), y0 = r0 + r1 + r2), fixed grid
                constexpr int KS = 8;
                bf16x8 wh[KS], wl[KS];
                {
                    const size_t wo = ((size_t)wave * KS * 64 + lane) * 8;      // fragment-major, see AGP_VP_LINEAR
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        wh[ks] = *(const bf16x8*)((const bf16_t*)o.p[0] + wo + ks * 512);
                        wl[ks] = *(const bf16x8*)((const bf16_t*)o.p[1] + wo + ks * 512);
                    }
                }
                const f32x4 bia = bias4(o.p[2]);
                f32x4 yv = operand(o.r[0], o.r[1], o.r[2]);
                const int act = o.act;
                auto feval = [&](const f32x4& state) -> f32x4 {
                    char* hi = planes + pbuf * (2 * FROWS * VP_YRB);
                    char* lo = hi + FROWS * VP_YRB;
                    store_state(hi, lo, VP_YRB, lane, wave, state);
                    __syncthreads();
                    f32x4 z = mfma_resident<KS>(wh, wl, hi, lo, VP_YRB, lane) + bia;
                    pbuf ^= 1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = apply_act(z[r], act);
                    return z;
                };
                const float third = 1.f / 3.f;
                for (int s = 0; s < P.nsteps; ++s) {
                    const float dt = P.dt[s];
                    if (P.method == AGP_ODE_EULER) {
                        yv = yv + dt * feval(yv);
                    } else if (P.method == AGP_ODE_MIDPOINT) {
                        const f32x4 k1 = feval(yv);
                        yv = yv + dt * feval(yv + k1 * (0.5f * dt));
                    } else {      // rk4, 3/8 rule (torchdiffeq rk4_alt_step_func)
                        const f32x4 k1 = feval(yv);
                        const f32x4 k2 = feval(yv + dt * k1 * third);
                        const f32x4 k3 = feval(yv + dt * (k2 - k1 * third));
                        const f32x4 k4 = feval(yv + dt * (k1 - k2 + k3));
                        yv = yv + (k1 + 3.f * (k2 + k3) + k4) * dt * 0.125f;
                    }
                }
                wr(o.dst, yv);
                break;
            }
            case AGP_VP_L2NORM: {                                // dst <- r0 / max(|r0|_2, 1e-12)
                const f32x4 v = rd(o.r[0]);
                const float s = rowsum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
                const float nrm = fmaxf(sqrtf(s), 1e-12f);
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = v[r] / nrm;
                wr(o.dst, y);
                break;
            }
            case AGP_VP_LAYERNORM: {                             // dst <- relu?(LN(r0) * gamma + beta + r1)
                const f32x4 v = rd(o.r[0]);
                const float mean = rowsum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
                f32x4 d;
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = v[r] - mean;
                const float var = rowsum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
                const float rstd = 1.f / sqrtf(var + o.f0);
                f32x4 gam = {1.f, 1.f, 1.f, 1.f};
                if (o.p[0]) gam = *(const f32x4*)((const float*)o.p[0] + nf);
                const f32x4 bet = bias4(o.p[1]);
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = d[r] * rstd * gam[r] + bet[r];
                if (o.r[1] >= 0) y += rd(o.r[1]);
                if (o.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
                }
                wr(o.dst, y);
                break;
            }
            case AGP_VP_WSUM: {                                  // dst <- sum_t w_t * r_t  (w_t = *p[t], 1 when NULL), in order
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    if (t < o.n) {
                        const float w = o.p[t] ? ((const float*)o.p[t])[0] : 1.f;
                        s += w * rd(o.r[t]);
                    }
                }
                wr(o.dst, s);
                break;
            }
            default: break;
        }
    }
}

}  // namespace agp_fusion
using namespace agp_fusion;

static int vecprog_check(const agp_vecprog_op& o, int ode_method, const float* ode_dt, int ode_nsteps) {
    auto reg_ok = [](int r, bool opt) { return (opt && r < 0) || (r >= 0 && r < AGP_VECPROG_NREG); };
    switch (o.op) {
        case AGP_VP_LOAD:
            if (!reg_ok(o.dst, false) || !o.p[0] || o.k <= 0 || o.k > 256 || o.k % 4) return AGP_E_BADARG;
            break;
        case AGP_VP_STORE:
            if (!reg_ok(o.r[0], false) || !o.p[0]) return AGP_E_BADARG;
            break;
        case AGP_VP_LINEAR:
        case AGP_VP_FCODE:
            if (!reg_ok(o.dst, false) || !reg_ok(o.r[0], false) || !reg_ok(o.r[1], true) || !reg_ok(o.r[2], true) || !o.p[0] || !o.p[1])
                return AGP_E_BADARG;
            if (o.op == AGP_VP_LINEAR && (o.k <= 0 || o.k > 256 || o.k % 32)) return AGP_E_BADARG;
            if (o.op == AGP_VP_FCODE && (ode_nsteps <= 0 || !ode_dt || ode_method < AGP_ODE_EULER || ode_method > AGP_ODE_RK4))
                return AGP_E_BADARG;
            if (o.act < AGP_ACT_ID || o.act > AGP_ACT_SIGMOID) return AGP_E_BADARG;
            break;
        case AGP_VP_L2NORM:
            if (!reg_ok(o.dst, false) || !reg_ok(o.r[0], false)) return AGP_E_BADARG;
            break;
        case AGP_VP_LAYERNORM:
            if (!reg_ok(o.dst, false) || !reg_ok(o.r[0], false) || !reg_ok(o.r[1], true)) return AGP_E_BADARG;
            break;
        case AGP_VP_WSUM:
            if (!reg_ok(o.dst, false) || o.n < 1 || o.n > 6) return AGP_E_BADARG;
            for (int t = 0; t < o.n; ++t)
                if (!reg_ok(o.r[t], false)) return AGP_E_BADARG;
            break;
        default: return AGP_E_BADARG;
    }
    return AGP_OK;
}

extern "C" int agp_vecprog_run2(const agp_vecprog_op* ops_a, int nops_a, int b_a, const agp_vecprog_op* ops_b, int nops_b, int b_b,
                                int ode_method, const float* ode_dt, int ode_nsteps, void* stream) {
    if (!ops_a || nops_a <= 0 || b_a <= 0 || nops_b < 0 || (nops_b > 0 && (!ops_b || b_b <= 0)) || nops_a + nops_b > AGP_VECPROG_MAXOPS ||
        ode_nsteps < 0 || ode_nsteps > 48)
        return AGP_E_BADARG;
    static_assert(sizeof(VpOpD) == sizeof(agp_vecprog_op), "agp_vecprog_op layout");
    VpProgram P = {};
    P.nops = nops_a + nops_b; P.b = b_a; P.method = ode_method; P.nsteps = ode_nsteps;
    P.nops_a = nops_a; P.blocks_a = (b_a + FROWS - 1) / FROWS; P.b_b = nops_b > 0 ? b_b : 0;
    for (int i = 0; i < ode_nsteps; ++i) P.dt[i] = ode_dt ? ode_dt[i] : 0.f;
    for (int i = 0; i < P.nops; ++i) {
        const agp_vecprog_op& o = i < nops_a ? ops_a[i] : ops_b[i - nops_a];
        const int rc = vecprog_check(o, ode_method, ode_dt, ode_nsteps);
        if (rc != AGP_OK) return rc;
        std::memcpy(&P.ops[i], &o, sizeof(VpOpD));
    }
    constexpr int lds = AGP_VECPROG_NREG * FROWS * VP_RS * 4 + 2 * 2 * FROWS * VP_YRB + 2 * 16 * FROWS * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static std::atomic<uint64_t> attr_done{0};
    if (!agp_lds_attr((const void*)vecprog_kernel, lds, attr_done)) return AGP_E_LAUNCH;
    const int blocks = P.blocks_a + (nops_b > 0 ? (b_b + FROWS - 1) / FROWS : 0);
    AGP_LAUNCH(vecprog_kernel, dim3(blocks), dim3(FT), lds, (hipStream_t)stream, P);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_vecprog_run(const agp_vecprog_op* ops, int nops, int b, int ode_method, const float* ode_dt, int ode_nsteps,
                               void* stream) {
    return agp_vecprog_run2(ops, nops, b, nullptr, 0, 0, ode_method, ode_dt, ode_nsteps, stream);
}
