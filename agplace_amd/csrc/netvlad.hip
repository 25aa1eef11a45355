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
        static bool set = false;                                                                         \
        if (!set) {                                                                                      \
            if (hipFuncSetAttribute((const void*)netvlad_kernel<DPT>,                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (NV_MAXD * NV_PC + NV_K * NV_PC + 256) * 4) != hipSuccess) \
                return AGP_E_LAUNCH;                                                                     \
            set = true;                                                                                  \
        }                                                                                                \
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
