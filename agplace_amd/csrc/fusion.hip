// Fusion MLPs and the fixed-step Neural-ODE integrator (reference network_mm/ffns.py,
// fuse_block_toshallow.py, stage2fuse_blockadd.py::Basic, dbvanilla2d.py::MLP).
//
// Work shape: [b, <=1024] x [256, K]^T with b = 16..64 rows -> latency-bound, not
// MFMA-bound.  One workgroup of 16 waves owns 16 batch rows and all 256
// output features: wave w owns features [16w, 16w+16) as the MFMA A operand (W rows),
// the activations are the B operand, staged in LDS as split-bf16 [row][k]; the fp32
// state lives in registers (lane = batch row l&15, features 16w + 4*(l>>4) + r).
// For the ODE, W's fragments (64 VGPRs: 8 k-steps x hi/lo) stay in registers for every
// step of the solver, so HBM/L2 sees W once per FCODE and there is ONE barrier per
// f-evaluation (the LDS activation planes are double-buffered).
// MFMA 16x16x32 bf16, split-bf16 x3 (hi*hi + hi*lo + lo*hi): fp32-class results.
#include "fusion_common.hpp"

namespace agp_fusion {

// ------------------------------------------------------------------ generic linear
// LT threads = LT/64 waves x 16 output columns per workgroup.  4 waves (64 columns): next to the conv kernels of the
// other streams (3 workgroups x 51 KB LDS per CU) a 16-wave workgroup with up to 82 KB LDS waits for most of a CU
// to drain before it can start -- in the replayed graph these launches took 21 us on average against 12 us alone.
constexpr int LT = 256;
__global__ __launch_bounds__(LT) void linear_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ add1,
                                                    const float* __restrict__ add2,
                                                    const bf16_t* __restrict__ w_hi,
                                                    const bf16_t* __restrict__ w_lo,
                                                    const float* __restrict__ bias, int b, int K, int N,
                                                    int act, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * FROWS, nblk = blockIdx.y * (LT / 4);
    const int yrb = yrow_bytes(K);
    char* yhi = smem;
    char* ylo = smem + FROWS * yrb;
    // The W fragments come straight from global memory: a whole chunk of 8 K-steps (16 loads in flight) is fetched
    // before its MFMAs, the first chunk even before x is staged, instead of one L2 round trip per K-step
    // (the launch is latency-bound: 2-4 workgroups).
    const int n = nblk + wave * 16 + (lane & 15);            // W row this lane loads
    const bf16_t* wrh = w_hi + (size_t)n * K + (lane >> 4) * 8;
    const bf16_t* wrl = w_lo + (size_t)n * K + (lane >> 4) * 8;
    const int nks = K / 32;
    bf16x8 ah[8], al[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ks = k0 + j < nks ? k0 + j : nks - 1;
            ah[j] = *(const bf16x8*)(wrh + ks * 32);
            al[j] = *(const bf16x8*)(wrl + ks * 32);
        }
    };
    fetch(0);
    // stage x (+adds) as split bf16
    // (16-byte loads, four elements per thread and iteration: K = 256 is ONE round trip for the workgroup)
    const int k4n = K / 4;
#pragma unroll 4
    for (int i = tid; i < FROWS * k4n; i += LT) {
        const int r = i / k4n, k = (i - r * k4n) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < b) {
            const size_t g = (size_t)(row0 + r) * K + k;
            v = *(const f32x4*)(x + g);
            if (add1) v += *(const f32x4*)(add1 + g);
            if (add2) v += *(const f32x4*)(add2 + g);
        }
        bf16x4 h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16_t h, l;
            split_bf16(v[e], h, l);
            h4[e] = h; l4[e] = l;
        }
        *(bf16x4*)(yhi + r * yrb + k * 2) = h4;
        *(bf16x4*)(ylo + r * yrb + k * 2) = l4;
    }
    __syncthreads();
    const int boff = (lane & 15) * yrb + (lane >> 4) * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nks; k0 += 8) {
        if (k0) fetch(k0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (k0 + j < nks) {
                const int ks = k0 + j;
                const bf16x8 bh = *(const bf16x8*)(yhi + boff + ks * 64);
                const bf16x8 bl = *(const bf16x8*)(ylo + boff + ks * 64);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bh, acc, 0, 0, 0);
            }
        }
    }
    const int brow = row0 + (lane & 15);
    const int nf = nblk + wave * 16 + (lane >> 4) * 4;
    if (brow < b) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = apply_act(acc[r] + (bias ? bias[nf + r] : 0.f), act);
        *(f32x4*)(y + (size_t)brow * N + nf) = o;
    }
}

// ------------------------------------------------------------------------- FCODE
template <int ACT>
__global__ __launch_bounds__(FT) void fcode_kernel(const float* __restrict__ x,
                                                   const float* __restrict__ add1,
                                                   const float* __restrict__ add2,
                                                   const bf16_t* __restrict__ w_hi,
                                                   const bf16_t* __restrict__ w_lo,
                                                   const float* __restrict__ bias, int b, int method,
                                                   OdeSteps steps, int nsteps, float* __restrict__ y,
                                                   float* __restrict__ traj) {
    constexpr int D = 256, KS = D / 32;
    constexpr int YRB = D * 2 + 16;
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * FROWS * YRB];   // [buf][plane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * FROWS;
    const int brow = row0 + (lane & 15);
    const int nf = wave * 16 + (lane >> 4) * 4;
    const bool live = brow < b;

    // resident W fragments (A operand): row n = 16*wave + (lane&15), k = 32*ks + 8*(lane>>4)
    bf16x8 wh[KS], wl[KS];
    {
        const size_t wo = (size_t)(wave * 16 + (lane & 15)) * D + (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wh[ks] = *(const bf16x8*)(w_hi + wo + ks * 32);
            wl[ks] = *(const bf16x8*)(w_lo + wo + ks * 32);
        }
    }
    f32x4 bia;
#pragma unroll
    for (int r = 0; r < 4; ++r) bia[r] = bias ? bias[nf + r] : 0.f;

    f32x4 yv = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const size_t g = (size_t)brow * D + nf;
        yv = *(const f32x4*)(x + g);
        if (add1) yv += *(const f32x4*)(add1 + g);
        if (add2) yv += *(const f32x4*)(add2 + g);
    }

    int buf = 0;
    auto feval = [&](const f32x4& state) -> f32x4 {
        char* hi = smem + buf * (2 * FROWS * YRB);
        char* lo = hi + FROWS * YRB;
        store_state(hi, lo, YRB, lane, wave, state);
        __syncthreads();
        f32x4 z = mfma_resident<KS>(wh, wl, hi, lo, YRB, lane);
        buf ^= 1;
        return act4<ACT>(z + bia);
    };

    const float third = 1.f / 3.f;
    const int nslot = 1 + (method == AGP_ODE_EULER ? 1 : (method == AGP_ODE_MIDPOINT ? 2 : 4));
    auto rec = [&](int s, int slot, const f32x4& v) {   // traj[s][slot][b][256]
        if (traj && live) *(f32x4*)(traj + (((size_t)s * nslot + slot) * b + brow) * D + nf) = v;
    };
    for (int s = 0; s < nsteps; ++s) {
        const float dt = steps.dt[s];
        rec(s, 0, yv);
        if (method == AGP_ODE_EULER) {
            const f32x4 k1 = feval(yv);
            rec(s, 1, k1);
            yv = yv + dt * k1;
        } else if (method == AGP_ODE_MIDPOINT) {
            const f32x4 k1 = feval(yv);
            const f32x4 k2 = feval(yv + k1 * (0.5f * dt));
            rec(s, 1, k1); rec(s, 2, k2);
            yv = yv + dt * k2;
        } else {  // rk4, 3/8 rule (torchdiffeq rk4_alt_step_func)
            const f32x4 k1 = feval(yv);
            const f32x4 k2 = feval(yv + dt * k1 * third);
            const f32x4 k3 = feval(yv + dt * (k2 - k1 * third));
            const f32x4 k4 = feval(yv + dt * (k1 - k2 + k3));
            rec(s, 1, k1); rec(s, 2, k2); rec(s, 3, k3); rec(s, 4, k4);
            yv = yv + (k1 + 3.f * (k2 + k3) + k4) * dt * 0.125f;
        }
    }
    if (live) *(f32x4*)(y + (size_t)brow * D + nf) = yv;
}

// ------------------------------------------------------------------- row-wise ops
// one wave per row
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ res, int b, int d,
                                                        float eps, int relu, float* __restrict__ y) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= b) return;
    const float* xr = x + (size_t)row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += xr[i];
    const float mean = wave_sum(s) / d;
    float v = 0.f;
    for (int i = lane; i < d; i += 64) { const float t = xr[i] - mean; v += t * t; }
    const float rstd = 1.f / sqrtf(wave_sum(v) / d + eps);
    for (int i = lane; i < d; i += 64) {
        float o = (xr[i] - mean) * rstd;
        o = o * (gamma ? gamma[i] : 1.f) + (beta ? beta[i] : 0.f);
        if (res) o += res[(size_t)row * d + i];
        if (relu) o = fmaxf(o, 0.f);
        y[(size_t)row * d + i] = o;
    }
}

__global__ __launch_bounds__(256) void l2normalize_kernel(const float* __restrict__ x, int b, int d,
                                                          float* __restrict__ y) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= b) return;
    const float* xr = x + (size_t)row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += xr[i] * xr[i];
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int i = lane; i < d; i += 64) y[(size_t)row * d + i] = xr[i] / nrm;
}

struct WsumArgs {
    const float* x[6];
    const float* w[6];
};
__global__ void wsum_kernel(WsumArgs a, int64_t n, float* __restrict__ y) {
    float w[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) w[t] = (a.x[t] && a.w[t]) ? a.w[t][0] : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 6; ++t)
            if (a.x[t]) s += w[t] * a.x[t][i];
        y[i] = s;
    }
}

}  // namespace agp_fusion
using namespace agp_fusion;

extern "C" int agp_wsum_fwd(const float* x0, const float* x1, const float* x2, const float* x3,
                            const float* x4, const float* x5, const float* w0, const float* w1,
                            const float* w2, const float* w3, const float* w4, const float* w5, int64_t n,
                            float* y, void* stream) {
    if (!x0 || !y || n <= 0) return AGP_E_BADARG;
    WsumArgs a = {{x0, x1, x2, x3, x4, x5}, {w0, w1, w2, w3, w4, w5}};
    for (int t = 1; t < 6; ++t)
        if (!a.x[t - 1]) a.x[t] = nullptr;   // the first NULL ends the list
    int g = (int)((n + 255) / 256);
    if (g > 2048) g = 2048;
    AGP_LAUNCH(wsum_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, a, n, y);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

// Backward of FCODE is implemented in fusion_bwd.hip.

extern "C" int agp_linear_fwd(const float* x, const float* add1, const float* add2, const void* w_hi,
                              const void* w_lo, const float* bias, int b, int k, int n, int act,
                              float* y, void* stream) {
    if (!x || !w_hi || !w_lo || !y || b <= 0 || k % 32 || k > MAXK || n % 256) return AGP_E_BADARG;
    if (act < AGP_ACT_ID || act > AGP_ACT_SIGMOID) return AGP_E_BADARG;
    const int lds = 2 * FROWS * (k * 2 + 16);
    AGP_LAUNCH(linear_kernel, dim3((b + FROWS - 1) / FROWS, n / (LT / 4)), dim3(LT), lds,
                       (hipStream_t)stream, x, add1, add2, (const bf16_t*)w_hi, (const bf16_t*)w_lo, bias,
                       b, k, n, act, y);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int64_t agp_fcode_traj_floats(int b, int method, int nsteps) {
    const int nst = method == AGP_ODE_EULER ? 1 : (method == AGP_ODE_MIDPOINT ? 2 : 4);
    return (int64_t)nsteps * (1 + nst) * b * 256;
}

extern "C" int agp_fcode_fwd(const float* x, const float* add1, const float* add2, const void* w_hi,
                             const void* w_lo, const float* bias, int b, int act, int method,
                             const float* dt, int nsteps, float* y, float* traj, void* stream) {
    if (!x || !w_hi || !w_lo || !y || !dt || b <= 0 || nsteps <= 0 || nsteps > 64) return AGP_E_BADARG;
    if (method < AGP_ODE_EULER || method > AGP_ODE_RK4) return AGP_E_BADARG;
    OdeSteps st;
    for (int i = 0; i < 64; ++i) st.dt[i] = i < nsteps ? dt[i] : 0.f;
    const dim3 grid((b + FROWS - 1) / FROWS), blk(FT);
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* wh = (const bf16_t*)w_hi;
    const bf16_t* wl = (const bf16_t*)w_lo;
    switch (act) {
        case AGP_ACT_ID: AGP_LAUNCH(fcode_kernel<AGP_ACT_ID>, grid, blk, 0, s, x, add1, add2, wh, wl, bias, b, method, st, nsteps, y, traj); break;
        case AGP_ACT_RELU: AGP_LAUNCH(fcode_kernel<AGP_ACT_RELU>, grid, blk, 0, s, x, add1, add2, wh, wl, bias, b, method, st, nsteps, y, traj); break;
        case AGP_ACT_TANH: AGP_LAUNCH(fcode_kernel<AGP_ACT_TANH>, grid, blk, 0, s, x, add1, add2, wh, wl, bias, b, method, st, nsteps, y, traj); break;
        case AGP_ACT_SIGMOID: AGP_LAUNCH(fcode_kernel<AGP_ACT_SIGMOID>, grid, blk, 0, s, x, add1, add2, wh, wl, bias, b, method, st, nsteps, y, traj); break;
        default: return AGP_E_BADARG;
    }
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* res,
                                 int b, int d, float eps, int relu, float* y, void* stream) {
    if (!x || !y || b <= 0 || d <= 0 || d > 4096) return AGP_E_BADARG;
    AGP_LAUNCH(layernorm_kernel, dim3((b + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma,
                       beta, res, b, d, eps, relu, y);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_l2normalize_fwd(const float* x, int b, int d, float* y, void* stream) {
    if (!x || !y || b <= 0 || d <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(l2normalize_kernel, dim3((b + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, b, d, y);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
