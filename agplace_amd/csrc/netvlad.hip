// NetVLAD.forward (reference model/aggregation.py:126-146), one workgroup per image.
//   x^ = x / max(|x|_2 over D, 1e-12)                    (per pixel, if normalize_input)
//   a  = softmax_k( conv_w[k] . x^ )                      (per pixel)
//   V[k][:] = sum_p a[k,p] x^[:,p] - (sum_p a[k,p]) c[k]  (the reference's per-cluster loop)
//   intra-normalise rows of V, flatten, L2-normalise.
// Pixels are processed in chunks of 64 staged in LDS; thread t owns cluster t/4 and the 64
// descriptor dims (t%4)*D/4.. of V in registers.  fp32 VALU: this aggregator is dead code in
// the reference's live path, correctness is what matters here.
#include "common.hpp"

namespace agp_netvlad {

constexpr int NV_K = 64, NV_PC = 64, NV_MAXD = 512;

template <int DPT>   // descriptor dims per thread = D / 4
__global__ __launch_bounds__(256) void netvlad_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ conv_w,
                                                      const float* __restrict__ cent, int D, int hw,
                                                      int K, int normalize_input, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                       // [D][NV_PC]
    float* as = xs + D * NV_PC;           // [NV_K][NV_PC]
    float* red = as + NV_K * NV_PC;       // [256]
    const int tid = threadIdx.x, im = blockIdx.x;
    const int k = tid >> 2, ds = tid & 3;
    const float* xi = x + (size_t)im * D * hw;
    float v[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) v[j] = 0.f;
    float asum = 0.f;
    for (int p0 = 0; p0 < hw; p0 += NV_PC) {
        const int np = min(NV_PC, hw - p0);
        __syncthreads();
        for (int i = tid; i < D * NV_PC; i += 256) {
            const int dd = i / NV_PC, p = i % NV_PC;
            xs[i] = p < np ? xi[(size_t)dd * hw + p0 + p] : 0.f;
        }
        __syncthreads();
        if (normalize_input) {
            // 4 threads per pixel
            const int p = tid >> 2;
            float s = 0.f;
            for (int dd = ds; dd < D; dd += 4) { const float t = xs[dd * NV_PC + p]; s += t * t; }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
            for (int dd = ds; dd < D; dd += 4) xs[dd * NV_PC + p] *= inv;
            __syncthreads();
        }
        // logits: thread handles cluster kk = tid/4 ... use 16 (k,p) pairs per thread
        for (int i = tid; i < NV_K * NV_PC; i += 256) {
            const int kk = i / NV_PC, p = i % NV_PC;
            float s = -__builtin_huge_valf();
            if (kk < K) {
                s = 0.f;
                const float* w = conv_w + (size_t)kk * D;
                for (int dd = 0; dd < D; ++dd) s += w[dd] * xs[dd * NV_PC + p];
            }
            as[i] = s;
        }
        __syncthreads();
        if (tid < NV_PC) {   // softmax over clusters for pixel tid
            const int p = tid;
            float m = -__builtin_huge_valf();
            for (int kk = 0; kk < K; ++kk) m = fmaxf(m, as[kk * NV_PC + p]);
            float s = 0.f;
            for (int kk = 0; kk < K; ++kk) { const float e = __expf(as[kk * NV_PC + p] - m); as[kk * NV_PC + p] = e; s += e; }
            const float inv = p < np ? 1.f / s : 0.f;
            for (int kk = 0; kk < NV_K; ++kk) as[kk * NV_PC + p] = kk < K ? as[kk * NV_PC + p] * inv : 0.f;
        }
        __syncthreads();
        for (int p = 0; p < np; ++p) {
            const float a = as[k * NV_PC + p];
            asum += a;
#pragma unroll
            for (int j = 0; j < DPT; ++j) v[j] += a * xs[(ds * DPT + j) * NV_PC + p];
        }
    }
    // residual term, intra-normalisation over D (4 threads per cluster)
    float ss = 0.f;
    if (k < K) {
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            v[j] -= asum * cent[(size_t)k * D + ds * DPT + j];
            ss += v[j] * v[j];
        }
    }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < DPT; ++j) { v[j] *= inv; tot += v[j] * v[j]; }
    if (k >= K) tot = 0.f;
    __syncthreads();
    red[tid] = tot;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float ginv = 1.f / fmaxf(sqrtf(red[0]), 1e-12f);
    if (k < K) {
        float* o = out + (size_t)im * K * D + (size_t)k * D + ds * DPT;
#pragma unroll
        for (int j = 0; j < DPT; ++j) o[j] = v[j] * ginv;
    }
}

// ---- backward (reference: autograd through model/aggregation.py:126-146).  One workgroup per image, the forward's thread map
// (thread = cluster k = tid / 4, descriptor dims (tid % 4) * D / 4 ..):
//   pass 1 recomputes the image's V_k, S_k, |V_k|, G exactly as the forward does, then walks the tail backwards in registers:
//       out = Vn / G            dVn = (g - out (g . out)) / G
//       Vn_k = V_k / |V_k|      dV_k = (dVn_k - Vn_k (dVn_k . Vn_k)) / |V_k|
//       V_k = U_k - S_k c_k     dS_k = - dV_k . c_k        dc_k = - S_k dV_k
//   pass 2 walks the pixels again in chunks of 32 (x^, a recomputed):
//       da_kp = dV_k . x^_p + dS_k          dl_kp = a_kp (da_kp - sum_j a_jp da_jp)         (softmax)
//       dw_k += sum_p dl_kp x^_p            dx^_p = sum_k (a_kp dV_k + dl_kp w_k)
//       dx_p  = (dx^_p - x^_p (dx^_p . x^_p)) / |x_p|                                       (input normalisation)
// dV goes through a global scratch row (every thread needs every cluster's dV for dx^); dw / dc leave as per-image partials and are
// added over the images in image order by netvlad_reduce_kernel.  fp32 VALU like the forward: dead code in the reference's live path.
constexpr int NV_PB = 32;

template <int DPT>
__global__ __launch_bounds__(256) void netvlad_bwd_kernel(const float* __restrict__ x, const float* __restrict__ conv_w,
                                                          const float* __restrict__ cent, const float* __restrict__ gout, int D, int hw,
                                                          int K, int normalize_input, float* __restrict__ dx, float* __restrict__ dvg,
                                                          float* __restrict__ dwp, float* __restrict__ dcp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                       // [D][NV_PB]   x^ of the chunk
    float* dxs = xs + D * NV_PB;          // [D][NV_PB]   dx^ of the chunk
    float* as = dxs + D * NV_PB;          // [NV_K][NV_PB] a
    float* das = as + NV_K * NV_PB;       // [NV_K][NV_PB] da, then dl
    float* red = das + NV_K * NV_PB;      // [256]
    float* nu = red + 256;                // [NV_PB] 1 / max(|x_p|, eps), < 0 when the norm was clamped
    const int tid = threadIdx.x, im = blockIdx.x;
    const int k = tid >> 2, ds = tid & 3;
    const float* xi = x + (size_t)im * D * hw;
    const float EPS = 1e-12f;

    // the chunk's x^ and a, as in the forward
    auto stage = [&](int p0, int np) {
        __syncthreads();
        for (int i = tid; i < D * NV_PB; i += 256) {
            const int dd = i / NV_PB, p = i % NV_PB;
            xs[i] = p < np ? xi[(size_t)dd * hw + p0 + p] : 0.f;
        }
        __syncthreads();
        if (normalize_input) {
            if (tid < 4 * NV_PB) {
                const int p = tid >> 2;
                float s = 0.f;
                for (int dd = ds; dd < D; dd += 4) { const float t = xs[dd * NV_PB + p]; s += t * t; }
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                const float nrm = sqrtf(s), inv = 1.f / fmaxf(nrm, EPS);
                for (int dd = ds; dd < D; dd += 4) xs[dd * NV_PB + p] *= inv;
                if (ds == 0) nu[p] = nrm > EPS ? inv : -inv;
            }
            __syncthreads();
        }
        for (int i = tid; i < NV_K * NV_PB; i += 256) {
            const int kk = i / NV_PB, p = i % NV_PB;
            float s = -__builtin_huge_valf();
            if (kk < K) {
                s = 0.f;
                const float* w = conv_w + (size_t)kk * D;
                for (int dd = 0; dd < D; ++dd) s += w[dd] * xs[dd * NV_PB + p];
            }
            as[i] = s;
        }
        __syncthreads();
        if (tid < NV_PB) {
            const int p = tid;
            float m = -__builtin_huge_valf();
            for (int kk = 0; kk < K; ++kk) m = fmaxf(m, as[kk * NV_PB + p]);
            float s = 0.f;
            for (int kk = 0; kk < K; ++kk) { const float e = __expf(as[kk * NV_PB + p] - m); as[kk * NV_PB + p] = e; s += e; }
            const float inv = p < np ? 1.f / s : 0.f;
            for (int kk = 0; kk < NV_K; ++kk) as[kk * NV_PB + p] = kk < K ? as[kk * NV_PB + p] * inv : 0.f;
        }
        __syncthreads();
    };

    // ---- pass 1: V_k, S_k
    float v[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) v[j] = 0.f;
    float asum = 0.f;
    for (int p0 = 0; p0 < hw; p0 += NV_PB) {
        const int np = min(NV_PB, hw - p0);
        stage(p0, np);
        for (int p = 0; p < np; ++p) {
            const float a = as[k * NV_PB + p];
            asum += a;
#pragma unroll
            for (int j = 0; j < DPT; ++j) v[j] += a * xs[(ds * DPT + j) * NV_PB + p];
        }
    }
    float ss = 0.f;
    if (k < K) {
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            v[j] -= asum * cent[(size_t)k * D + ds * DPT + j];
            ss += v[j] * v[j];
        }
    }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    const float rk = sqrtf(ss), inv_r = 1.f / fmaxf(rk, EPS);
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < DPT; ++j) { v[j] *= inv_r; tot += v[j] * v[j]; }       // v = Vn
    if (k >= K) tot = 0.f;
    auto block_sum = [&](float t) {
        __syncthreads();
        red[tid] = t;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        const float r_ = red[0];
        __syncthreads();
        return r_;
    };
    const float G = sqrtf(block_sum(tot)), ginv = 1.f / fmaxf(G, EPS);
    // ---- the tail, backwards: dV_k in dv[], dS_k
    float dv[DPT];
    float go = 0.f;
    if (k < K) {
        const float* g = gout + (size_t)im * K * D + (size_t)k * D + ds * DPT;
#pragma unroll
        for (int j = 0; j < DPT; ++j) { dv[j] = g[j]; go += g[j] * v[j] * ginv; }          // g . out
    } else {
#pragma unroll
        for (int j = 0; j < DPT; ++j) dv[j] = 0.f;
    }
    const float gdot = G > EPS ? block_sum(go) : (block_sum(0.f), 0.f);
    float dk = 0.f;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        dv[j] = (dv[j] - v[j] * ginv * gdot) * ginv;       // dVn
        dk += dv[j] * v[j];
    }
    dk += __shfl_xor(dk, 1, 64);
    dk += __shfl_xor(dk, 2, 64);
    if (!(rk > EPS)) dk = 0.f;
    float dS = 0.f;
    if (k < K) {
        float* dvo = dvg + (size_t)im * K * D + (size_t)k * D + ds * DPT;
        float* dco = dcp + (size_t)im * K * D + (size_t)k * D + ds * DPT;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            dv[j] = (dv[j] - v[j] * dk) * inv_r;             // dV
            dS -= dv[j] * cent[(size_t)k * D + ds * DPT + j];
            dvo[j] = dv[j];
            dco[j] = -asum * dv[j];
        }
    }
    dS += __shfl_xor(dS, 1, 64);
    dS += __shfl_xor(dS, 2, 64);
    __threadfence_block();

    // ---- pass 2: pixels
    float dwv[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) dwv[j] = 0.f;
    const float* dvi = dvg + (size_t)im * K * D;
    float* dxi = dx + (size_t)im * D * hw;
    for (int p0 = 0; p0 < hw; p0 += NV_PB) {
        const int np = min(NV_PB, hw - p0);
        stage(p0, np);
        for (int p = 0; p < NV_PB; ++p) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < DPT; ++j) t += dv[j] * xs[(ds * DPT + j) * NV_PB + p];
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            if (ds == 0) das[k * NV_PB + p] = (k < K && p < np) ? t + dS : 0.f;
        }
        __syncthreads();
        if (tid < NV_PB) {
            const int p = tid;
            float t = 0.f;
            for (int kk = 0; kk < K; ++kk) t += as[kk * NV_PB + p] * das[kk * NV_PB + p];
            for (int kk = 0; kk < K; ++kk) das[kk * NV_PB + p] = as[kk * NV_PB + p] * (das[kk * NV_PB + p] - t);
        }
        __syncthreads();
        for (int p = 0; p < np; ++p) {
            const float dl = das[k * NV_PB + p];
#pragma unroll
            for (int j = 0; j < DPT; ++j) dwv[j] += dl * xs[(ds * DPT + j) * NV_PB + p];
        }
        for (int i = tid; i < D * NV_PB; i += 256) {
            const int dd = i / NV_PB, p = i % NV_PB;
            float acc = 0.f;
            for (int kk = 0; kk < K; ++kk)
                acc += as[kk * NV_PB + p] * dvi[(size_t)kk * D + dd] + das[kk * NV_PB + p] * conv_w[(size_t)kk * D + dd];
            dxs[i] = acc;
        }
        __syncthreads();
        if (tid < 4 * NV_PB) {
            const int p = tid >> 2;
            float dot = 0.f, scale = 1.f;
            if (normalize_input) {
                for (int dd = ds; dd < D; dd += 4) dot += dxs[dd * NV_PB + p] * xs[dd * NV_PB + p];
                dot += __shfl_xor(dot, 1, 64);
                dot += __shfl_xor(dot, 2, 64);
                const float n_ = nu[p];
                scale = fabsf(n_);
                if (n_ < 0.f) dot = 0.f;                     // the norm was clamped: x^ = x / eps is linear in x
            }
            if (p < np)
                for (int dd = ds; dd < D; dd += 4)
                    dxi[(size_t)dd * hw + p0 + p] = (dxs[dd * NV_PB + p] - xs[dd * NV_PB + p] * dot) * scale;
        }
    }
    if (k < K) {
        float* dwo = dwp + (size_t)im * K * D + (size_t)k * D + ds * DPT;
#pragma unroll
        for (int j = 0; j < DPT; ++j) dwo[j] = dwv[j];
    }
}

// dw / dc = the per-image partials added in image order
__global__ void netvlad_reduce_kernel(const float* __restrict__ dwp, const float* __restrict__ dcp, int n, int kd, float* __restrict__ dw,
                                      float* __restrict__ dc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kd) return;
    float a = 0.f, b = 0.f;
    for (int im = 0; im < n; ++im) { a += dwp[(size_t)im * kd + i]; b += dcp[(size_t)im * kd + i]; }
    dw[i] = a;
    dc[i] = b;
}

}  // namespace agp_netvlad
using namespace agp_netvlad;

extern "C" int agp_netvlad_fwd(const float* x, const float* conv_w, const float* centroids, int n, int d,
                               int hw, int k, int normalize_input, float* out, void* stream) {
    if (!x || !conv_w || !centroids || !out || n <= 0 || hw <= 0) return AGP_E_BADARG;
    if (k < 1 || k > NV_K || d > NV_MAXD || d % 4) return AGP_E_BADARG;
    const int lds = (d * NV_PC + NV_K * NV_PC + 256) * 4;
    hipStream_t s = (hipStream_t)stream;
#define NV_LAUNCH(DPT)                                                                                   \
    do {                                                                                                 \
        static std::atomic<uint64_t> set{0};                                                             \
        if (!agp_lds_attr((const void*)netvlad_kernel<DPT>, (NV_MAXD * NV_PC + NV_K * NV_PC + 256) * 4, set)) return AGP_E_LAUNCH; \
        AGP_LAUNCH(netvlad_kernel<DPT>, dim3(n), dim3(256), lds, s, x, conv_w, centroids, d, hw, \
                           k, normalize_input, out);                                                     \
    } while (0)
    switch (d) {
        case 64: NV_LAUNCH(16); break;
        case 128: NV_LAUNCH(32); break;
        case 256: NV_LAUNCH(64); break;
        case 512: NV_LAUNCH(128); break;
        default: return AGP_E_UNSUPPORTED;
    }
#undef NV_LAUNCH
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int agp_netvlad_bwd(const float* x, const float* conv_w, const float* centroids, const float* gout, int n, int d, int hw,
                               int k, int normalize_input, float* dx, float* dw, float* dc, float* workspace, void* stream) {
    if (!x || !conv_w || !centroids || !gout || !dx || !dw || !dc || !workspace || n <= 0 || hw <= 0) return AGP_E_BADARG;
    if (k < 1 || k > NV_K || d > NV_MAXD || d % 4) return AGP_E_BADARG;
    const int lds = (2 * d * NV_PB + 2 * NV_K * NV_PB + 256 + NV_PB) * 4;
    hipStream_t s = (hipStream_t)stream;
    const size_t nkd = (size_t)n * k * d;
    float* dvg = workspace; float* dwp = workspace + nkd; float* dcp = workspace + 2 * nkd;      // workspace: 3 n k d floats
#define NV_LAUNCH(DPT)                                                                                   \
    do {                                                                                                 \
        static std::atomic<uint64_t> set{0};                                                             \
        if (!agp_lds_attr((const void*)netvlad_bwd_kernel<DPT>, (2 * NV_MAXD * NV_PB + 2 * NV_K * NV_PB + 256 + NV_PB) * 4, set)) \
            return AGP_E_LAUNCH;                                                                         \
        AGP_LAUNCH(netvlad_bwd_kernel<DPT>, dim3(n), dim3(256), lds, s, x, conv_w, centroids, gout, d, hw, k, normalize_input, dx, dvg, \
                   dwp, dcp);                                                                            \
    } while (0)
    switch (d) {
        case 64: NV_LAUNCH(16); break;
        case 128: NV_LAUNCH(32); break;
        case 256: NV_LAUNCH(64); break;
        default: return AGP_E_UNSUPPORTED;
    }
#undef NV_LAUNCH
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(netvlad_reduce_kernel, dim3((k * d + 255) / 256), dim3(256), 0, s, dwp, dcp, n, k * d, dw, dc);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
