// Backward kernels of the fusion path: FCODE (Neural-ODE block), Linear(+act), LayerNorm, L2-normalise.
//
// The reference differentiates THROUGH the unrolled fixed-grid solver (plain `odeint`, not the
// adjoint method, network_mm/ffns.py:84), so the gradient is the exact gradient of the discrete
// scheme.  With stage inputs s_i = y + dt*sum_j A_ij k_j, k_i = act(s_i W^T + b) and
// y' = y + dt*sum_i b_i k_i, one step back-propagates (a = dL/dy'):
//     for i = last..first:  u_i  = dt*b_i*a + dt*sum_{j>i} A_ji * gs_j
//                           gz_i = u_i * act'(k_i)            (act' from the recorded output k_i)
//                           gs_i = gz_i W                      (MFMA, W^T fragments in registers)
//     dL/dy = a + sum_i gs_i ;   dW += gz_i^T s_i ;  db += sum_batch gz_i
// The forward kernel records y and every k_i, so no forward matmul is recomputed here.  All
// (gz_i, s_i) pairs are written to a [R][256] workspace and dW is one fp32 GEMM over R afterwards.
#include "fusion_common.hpp"

namespace agp_fusion {

template <int ACT>
__device__ __forceinline__ f32x4 dact_from_output(const f32x4& k) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (ACT == AGP_ACT_RELU) o[r] = k[r] > 0.f ? 1.f : 0.f;
        else if (ACT == AGP_ACT_TANH) o[r] = 1.f - k[r] * k[r];
        else if (ACT == AGP_ACT_SIGMOID) o[r] = k[r] * (1.f - k[r]);
        else o[r] = 1.f;
    }
    return o;
}

// One workgroup (16 waves) per 16 batch rows; lane owns (batch row l&15, features 16w+4(l>>4)+r).
template <int ACT>
__global__ __launch_bounds__(FT) void fcode_bwd_state_kernel(const float* __restrict__ traj,
                                                             const float* __restrict__ gy,
                                                             const bf16_t* __restrict__ wt_hi,
                                                             const bf16_t* __restrict__ wt_lo, int b,
                                                             int method, OdeSteps steps, int nsteps,
                                                             float* __restrict__ gx, float* __restrict__ GZ,
                                                             float* __restrict__ S, int Bp) {
    constexpr int D = 256, KS = D / 32;
    constexpr int YRB = D * 2 + 16;
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * FROWS * YRB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * FROWS;
    const int brow = row0 + (lane & 15);
    const int nf = wave * 16 + (lane >> 4) * 4;
    const bool live = brow < b;
    const int nst = method == AGP_ODE_EULER ? 1 : (method == AGP_ODE_MIDPOINT ? 2 : 4);
    const int nslot = 1 + nst;

    // resident W^T fragments: row k = 16*wave + (lane&15), contraction index n = 32*ks + 8*(lane>>4)
    bf16x8 wh[KS], wl[KS];
    {
        const size_t wo = (size_t)(wave * 16 + (lane & 15)) * D + (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wh[ks] = *(const bf16x8*)(wt_hi + wo + ks * 32);
            wl[ks] = *(const bf16x8*)(wt_lo + wo + ks * 32);
        }
    }
    int buf = 0;
    auto times_w = [&](const f32x4& gz) -> f32x4 {     // gs = gz W for this lane's 4 features
        char* hi = smem + buf * (2 * FROWS * YRB);
        char* lo = hi + FROWS * YRB;
        store_state(hi, lo, YRB, lane, wave, gz);
        __syncthreads();
        const f32x4 r = mfma_resident<KS>(wh, wl, hi, lo, YRB, lane);
        buf ^= 1;
        return r;
    };
    auto ld = [&](int s, int slot) -> f32x4 {
        if (!live) return f32x4{0.f, 0.f, 0.f, 0.f};
        return *(const f32x4*)(traj + (((size_t)s * nslot + slot) * b + brow) * D + nf);
    };
    auto emit = [&](int s, int i, const f32x4& gz, const f32x4& sin) {   // row r = (s*nst+i)*Bp + brow
        const size_t r = ((size_t)s * nst + i) * Bp + brow;
        *(f32x4*)(GZ + r * D + nf) = gz;
        *(f32x4*)(S + r * D + nf) = sin;
    };

    f32x4 a = live ? *(const f32x4*)(gy + (size_t)brow * D + nf) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float third = 1.f / 3.f;
    for (int s = nsteps - 1; s >= 0; --s) {
        const float dt = steps.dt[s];
        const f32x4 y = ld(s, 0);
        if (method == AGP_ODE_EULER) {
            const f32x4 k1 = ld(s, 1);
            const f32x4 gz1 = (dt * a) * dact_from_output<ACT>(k1);
            emit(s, 0, gz1, y);
            a = a + times_w(gz1);
        } else if (method == AGP_ODE_MIDPOINT) {
            const f32x4 k1 = ld(s, 1), k2 = ld(s, 2);
            const f32x4 s2 = y + k1 * (0.5f * dt);
            const f32x4 gz2 = (dt * a) * dact_from_output<ACT>(k2);
            emit(s, 1, gz2, s2);
            const f32x4 gs2 = times_w(gz2);
            const f32x4 gz1 = ((0.5f * dt) * gs2) * dact_from_output<ACT>(k1);
            emit(s, 0, gz1, y);
            a = a + gs2 + times_w(gz1);
        } else {
            const f32x4 k1 = ld(s, 1), k2 = ld(s, 2), k3 = ld(s, 3), k4 = ld(s, 4);
            const f32x4 s2 = y + dt * k1 * third;
            const f32x4 s3 = y + dt * (k2 - k1 * third);
            const f32x4 s4 = y + dt * (k1 - k2 + k3);
            const f32x4 gz4 = ((0.125f * dt) * a) * dact_from_output<ACT>(k4);
            emit(s, 3, gz4, s4);
            const f32x4 gs4 = times_w(gz4);
            const f32x4 gz3 = ((0.375f * dt) * a + dt * gs4) * dact_from_output<ACT>(k3);
            emit(s, 2, gz3, s3);
            const f32x4 gs3 = times_w(gz3);
            const f32x4 gz2 = ((0.375f * dt) * a + dt * (gs3 - gs4)) * dact_from_output<ACT>(k2);
            emit(s, 1, gz2, s2);
            const f32x4 gs2 = times_w(gz2);
            const f32x4 gz1 = ((0.125f * dt) * a + dt * (gs4 - third * gs3 + third * gs2)) * dact_from_output<ACT>(k1);
            emit(s, 0, gz1, y);
            a = a + gs4 + gs3 + gs2 + times_w(gz1);
        }
    }
    if (live) *(f32x4*)(gx + (size_t)brow * D + nf) = a;
}

// C[M][N] = sum_r A[r][M] * B[r][N]  (fp32, contraction over the ROW index of both inputs).
// 64x64 output tile per 256-thread block, 4x4 outputs per thread, 16-row chunks through LDS.
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int R, int lda,
                                                          int ldb, int ldc) {
    __shared__ float as[16][64 + 4], bs[16][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int r0 = 0; r0 < R; r0 += 16) {
        for (int i = tid; i < 16 * 64; i += 256) {
            const int rr = i >> 6, cc = i & 63;
            const int r = r0 + rr;
            as[rr][cc] = (r < R && m0 + cc < M) ? A[(size_t)r * lda + m0 + cc] : 0.f;
            bs[rr][cc] = (r < R && n0 + cc < N) ? B[(size_t)r * ldb + n0 + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = as[rr][ty * 4 + i]; bv[i] = bs[rr][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] += av[i] * bv[jj];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + jj;
            if (m < M && n < N) C[(size_t)m * ldc + n] = acc[i][jj];
        }
}

// out[c] = sum_r A[r][c].  A workgroup owns 64 columns: wave w adds the rows w, w + 4, ... (eight independent loads per trip),
// the four wave sums are added in wave order -- a fixed order.  (One thread per column walking all the rows in one dependent
// loop: 130 us for the 176 x 256 bias gradient of the database head, 0.35 ms of a training step.)
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ A, int R, int Ccols, int lda, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < Ccols) {
        for (int r0 = w; r0 < R; r0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = A[(size_t)(r0 + 4 * u < R ? r0 + 4 * u : R - 1) * lda + c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + 4 * u < R) s += v[u];
        }
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < Ccols) out[c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// gz = gy * act'(y)   (y = forward OUTPUT), written with row stride ldz (zero padding beyond n)
__global__ void dact_kernel(const float* __restrict__ y, const float* __restrict__ gy, int b, int n, int ldz,
                            int act, float* __restrict__ gz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * ldz) return;
    const int r = i / ldz, c = i - r * ldz;
    float v = 0.f;
    if (c < n) {
        v = gy[(size_t)r * n + c];
        if (act != AGP_ACT_ID) {
            const float k = y[(size_t)r * n + c];
            v *= act == AGP_ACT_RELU ? (k > 0.f ? 1.f : 0.f) : (act == AGP_ACT_TANH ? 1.f - k * k : k * (1.f - k));
        }
    }
    gz[i] = v;
}

// LayerNorm backward, one wave per row.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ y, const float* __restrict__ gy,
                                                            int b, int d, float eps, int relu, float* __restrict__ gx,
                                                            float* __restrict__ gres) {
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
    // g = gy masked by the ReLU; gxh = g*gamma
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < d; i += 64) {
        float g = gy[(size_t)row * d + i];
        if (relu && !(y[(size_t)row * d + i] > 0.f)) g = 0.f;
        const float xh = (xr[i] - mean) * rstd;
        const float gxh = g * (gamma ? gamma[i] : 1.f);
        s1 += gxh;
        s2 += gxh * xh;
        if (gres) gres[(size_t)row * d + i] = g;
    }
    s1 = wave_sum(s1) / d;
    s2 = wave_sum(s2) / d;
    for (int i = lane; i < d; i += 64) {
        float g = gy[(size_t)row * d + i];
        if (relu && !(y[(size_t)row * d + i] > 0.f)) g = 0.f;
        const float xh = (xr[i] - mean) * rstd;
        const float gxh = g * (gamma ? gamma[i] : 1.f);
        gx[(size_t)row * d + i] = rstd * (gxh - s1 - xh * s2);
    }
}

// ggamma[c] = sum_r g[r][c] * xhat[r][c], gbeta[c] = sum_r g[r][c] in a FIXED order: a workgroup owns 64 columns, wave w walks the
// rows w, w + 4, ... (the row's mean / rstd recomputed: d values per row and wave), the four wave sums are added in wave order.
// (The row kernel used to atomicAdd every element: the two gradients differed in the last bits from run to run.)
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ gy, int b, int d, float eps, int relu,
                                                                   float* __restrict__ ggamma, float* __restrict__ gbeta) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float sg = 0.f, sb = 0.f;
    for (int row = w; row < b; row += 4) {
        const float* xr = x + (size_t)row * d;
        float s = 0.f;
        for (int i = lane; i < d; i += 64) s += xr[i];
        const float mean = wave_sum(s) / d;
        float v = 0.f;
        for (int i = lane; i < d; i += 64) { const float t = xr[i] - mean; v += t * t; }
        const float rstd = 1.f / sqrtf(wave_sum(v) / d + eps);
        if (c < d) {
            float g = gy[(size_t)row * d + c];
            if (relu && !(y[(size_t)row * d + c] > 0.f)) g = 0.f;
            sg += g * ((xr[c] - mean) * rstd);
            sb += g;
        }
    }
    red[0][w][lane] = sg;
    red[1][w][lane] = sb;
    __syncthreads();
    if (w == 0 && c < d) {
        if (ggamma) ggamma[c] = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
        if (gbeta) gbeta[c] = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
    }
}

// y = x / max(|x|, 1e-12)  ->  gx = (gy - y (y.gy)) / |x|   (clamped norm: plain gy / 1e-12)
__global__ __launch_bounds__(256) void l2normalize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                              int b, int d, float* __restrict__ gx) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= b) return;
    const float* xr = x + (size_t)row * d;
    const float* gr = gy + (size_t)row * d;
    float s = 0.f, dot = 0.f;
    for (int i = lane; i < d; i += 64) { s += xr[i] * xr[i]; dot += xr[i] * gr[i]; }
    const float nrm = sqrtf(wave_sum(s));
    dot = wave_sum(dot);
    for (int i = lane; i < d; i += 64) {
        float o;
        if (nrm > 1e-12f) o = (gr[i] - xr[i] * (dot / (nrm * nrm))) / nrm;
        else o = gr[i] / 1e-12f;
        gx[(size_t)row * d + i] = o;
    }
}

inline int stages_of(int method) { return method == AGP_ODE_EULER ? 1 : (method == AGP_ODE_MIDPOINT ? 2 : 4); }

}  // namespace agp_fusion
using namespace agp_fusion;

// declared in fusion.hip
extern "C" int agp_linear_fwd(const float* x, const float* add1, const float* add2, const void* w_hi,
                              const void* w_lo, const float* bias, int b, int k, int n, int act, float* y,
                              void* stream);

static int gemm_tn(const float* A, const float* B, float* C, int M, int N, int R, int lda, int ldb, int ldc,
                   hipStream_t s) {
    AGP_LAUNCH(gemm_tn_f32_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, A, B, C, M, N, R, lda, ldb, ldc);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int64_t agp_fcode_bwd_workspace_bytes(int b, int method, int nsteps) {
    const int64_t Bp = (b + FROWS - 1) / FROWS * FROWS;
    return 2 * (int64_t)nsteps * stages_of(method) * Bp * 256 * sizeof(float);
}

extern "C" int agp_fcode_bwd(const float* traj, const float* gy, const void* wt_hi, const void* wt_lo, int b,
                             int act, int method, const float* dt, int nsteps, float* gx, float* gw, float* gb,
                             void* workspace, int64_t workspace_bytes, void* stream) {
    if (!traj || !gy || !wt_hi || !wt_lo || !dt || !gx || !workspace || b <= 0 || nsteps <= 0 || nsteps > 64)
        return AGP_E_BADARG;
    if (method < AGP_ODE_EULER || method > AGP_ODE_RK4) return AGP_E_BADARG;
    if (workspace_bytes < agp_fcode_bwd_workspace_bytes(b, method, nsteps)) return AGP_E_BADARG;
    const int Bp = (b + FROWS - 1) / FROWS * FROWS;
    const int R = nsteps * stages_of(method) * Bp;
    float* GZ = (float*)workspace;
    float* S = GZ + (size_t)R * 256;
    OdeSteps st;
    for (int i = 0; i < 64; ++i) st.dt[i] = i < nsteps ? dt[i] : 0.f;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(Bp / FROWS), blk(FT);
    const bf16_t* wh = (const bf16_t*)wt_hi;
    const bf16_t* wl = (const bf16_t*)wt_lo;
    switch (act) {
        case AGP_ACT_ID: AGP_LAUNCH(fcode_bwd_state_kernel<AGP_ACT_ID>, grid, blk, 0, s, traj, gy, wh, wl, b, method, st, nsteps, gx, GZ, S, Bp); break;
        case AGP_ACT_RELU: AGP_LAUNCH(fcode_bwd_state_kernel<AGP_ACT_RELU>, grid, blk, 0, s, traj, gy, wh, wl, b, method, st, nsteps, gx, GZ, S, Bp); break;
        case AGP_ACT_TANH: AGP_LAUNCH(fcode_bwd_state_kernel<AGP_ACT_TANH>, grid, blk, 0, s, traj, gy, wh, wl, b, method, st, nsteps, gx, GZ, S, Bp); break;
        case AGP_ACT_SIGMOID: AGP_LAUNCH(fcode_bwd_state_kernel<AGP_ACT_SIGMOID>, grid, blk, 0, s, traj, gy, wh, wl, b, method, st, nsteps, gx, GZ, S, Bp); break;
        default: return AGP_E_BADARG;
    }
    AGP_CHECK_LAUNCH();
    if (gw) {
        const int rc = gemm_tn(GZ, S, gw, 256, 256, R, 256, 256, 256, s);   // dW[n][k] = sum_r gz[r][n] s[r][k]
        if (rc != AGP_OK) return rc;
    }
    if (gb) {
        AGP_LAUNCH(colsum_kernel, dim3(4), dim3(256), 0, s, GZ, R, 256, 256, gb);
        AGP_CHECK_LAUNCH();
    }
    return AGP_OK;
}

extern "C" int64_t agp_linear_bwd_workspace_bytes(int b, int k, int n) {
    (void)k;
    const int64_t np = (n + 31) / 32 * 32;
    return (int64_t)b * np * sizeof(float);
}

// wt planes: [kpad][npad] with kpad = k rounded up to 256, npad = n rounded up to 32 (zero padded).
extern "C" int agp_linear_bwd(const float* x, const float* y, const float* gy, const void* wt_hi,
                              const void* wt_lo, int b, int k, int n, int act, float* gx, float* gw, float* gb,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    if (!gy || !workspace || b <= 0 || k <= 0 || n <= 0) return AGP_E_BADARG;
    if (act != AGP_ACT_ID && !y) return AGP_E_BADARG;
    if (workspace_bytes < agp_linear_bwd_workspace_bytes(b, k, n)) return AGP_E_BADARG;
    const int np = (n + 31) / 32 * 32, kp = (k + 255) / 256 * 256;
    if (np > 1024) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float* gz = (float*)workspace;
    AGP_LAUNCH(dact_kernel, dim3((b * np + 255) / 256), dim3(256), 0, s, y, gy, b, n, np, act, gz);
    AGP_CHECK_LAUNCH();
    if (gx) {   // gx[b][kp] = gz[b][np] (W^T)[kp][np]^T
        if (!wt_hi || !wt_lo) return AGP_E_BADARG;
        const int rc = agp_linear_fwd(gz, nullptr, nullptr, wt_hi, wt_lo, nullptr, b, np, kp, AGP_ACT_ID, gx, stream);
        if (rc != AGP_OK) return rc;
    }
    if (gw) {
        if (!x) return AGP_E_BADARG;
        const int rc = gemm_tn(gz, x, gw, n, k, b, np, k, k, s);           // dW[n][k] = sum_b gz[b][n] x[b][k]
        if (rc != AGP_OK) return rc;
    }
    if (gb) {
        AGP_LAUNCH(colsum_kernel, dim3((n + 63) / 64), dim3(256), 0, s, gz, b, n, np, gb);
        AGP_CHECK_LAUNCH();
    }
    return AGP_OK;
}

extern "C" int agp_layernorm_bwd(const float* x, const float* gamma, const float* y, const float* gy, int b, int d,
                                 float eps, int relu, float* gx, float* gres, float* ggamma, float* gbeta,
                                 void* stream) {
    if (!x || !gy || !gx || b <= 0 || d <= 0 || (relu && !y)) return AGP_E_BADARG;
    AGP_LAUNCH(layernorm_bwd_kernel, dim3((b + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, y, gy, b, d, eps,
               relu, gx, gres);
    AGP_CHECK_LAUNCH();
    if (ggamma || gbeta) {
        AGP_LAUNCH(layernorm_bwd_params_kernel, dim3((d + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, y, gy, b, d, eps, relu,
                   ggamma, gbeta);
        AGP_CHECK_LAUNCH();
    }
    return AGP_OK;
}

extern "C" int agp_l2normalize_bwd(const float* x, const float* gy, int b, int d, float* gx, void* stream) {
    if (!x || !gy || !gx || b <= 0 || d <= 0) return AGP_E_BADARG;
    AGP_LAUNCH(l2normalize_bwd_kernel, dim3((b + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gy, b, d, gx);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}


// ---- gradient of a learnable scalar fusion weight: dL/dw_t = sum_i gy[i] * x_t[i]  (reference tools/options.py:139-146,
// the xxx_learnweight flags turn MM's scalar mixing weights into trained parameters; mm.py:84,92,104,123-138).
// One workgroup, fixed-order reduction (lanes -> waves -> workgroup): deterministic.
namespace agp_fusion_bwd {
__global__ void __launch_bounds__(1024) dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                   float* __restrict__ out) {
    __shared__ float part[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        out[0] = t;
    }
}
}  // namespace agp_fusion_bwd

extern "C" int agp_dot_f32(const float* a, const float* b, int64_t n, float* out, void* stream) {
    if (!a || !b || !out || n < 0) return AGP_E_BADARG;
    AGP_LAUNCH(agp_fusion_bwd::dot_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, out);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
