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


// ---- round 6: the forward on the MATRIX pipe, exact fp32: v_mfma_f32_32x32x2_f32 (fp32 operands, fp32
// accumulate, 64 FLOP / clk / SIMD = 157 TFLOP/s chip-wide; MI355X_MICROARCH.md) -- the aggregator's tests hold it to 1e-5 of fp64,
// which one 16-bit product cannot meet (2^-12 per operand) and three products of split operands would need every LDS image twice.
// Several workgroups per image (`splits` pixel ranges), four waves each, 64-pixel chunks:
//   stage   x[d][p0 .. p0 + 63] -> registers (16-byte loads along the pixels of a channel row), per-pixel |x|^2 through LDS,
//           x^ = x / max(|x|, 1e-12) into ONE fp32 LDS image xs[d][66] that serves both products;
//   GEMM 1  logits[k][p] = sum_d w[k][d] x^[d][p]: a wave = 32 clusters x 32 pixels, ITS 32 x D SLICE OF w IN REGISTERS for the whole
//           kernel (one float per lane and MFMA: D / 2 registers), B operand = one ds_read_b32 per MFMA (lanes = consecutive pixels);
//   softmax over the clusters (four threads per pixel), a[k][p] into LDS, sum_p a[k][p] by wave reductions;
//   GEMM 2  V[k][d] += sum_p a[k][p] x^[d][p]: a wave = 64 clusters x D / 4 channels, accumulators live across the chunks
//           (A operand: lanes = consecutive clusters of as[k][66]; B operand: lanes = consecutive channels of xs, bank = 2 d + p).
// The workgroup leaves its partial V and S; netvlad_finish_kernel adds the splits in order, subtracts S c, normalises.
constexpr int NVM_PC = 64, NVM_LD = 66;
template <int D>
__global__ __launch_bounds__(256, 1) void netvlad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ conv_w, int hw, int K,
                                                               int normalize_input, int chunks_per_split, float* __restrict__ vpart,
                                                               float* __restrict__ spart) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(D == 128 || D == 256, "a wave owns D / 4 = 32 or 64 channels of V");
    constexpr int TD = D / 128;                         // 32-channel tiles of V per wave
    constexpr int NI = D / 16;                          // channel rows a thread stages per chunk
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* const xs = sm;                               // [D][NVM_LD]   x^
    float* const as = xs + D * NVM_LD;                  // [64][NVM_LD]  logits, then a
    float* const red = as + 64 * NVM_LD;                // [16][64] partial |x|^2, then [4][64] softmax maxima / sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int im = blockIdx.y, split = blockIdx.x, nsplit = gridDim.x;
    const float* xi = x + (size_t)im * D * hw;
    const int nchunks = (hw + NVM_PC - 1) / NVM_PC;
    (void)chunks_per_split;
    const int c0 = split * nchunks / nsplit, c1 = (split + 1) * nchunks / nsplit;      // balanced: the ranges differ by one chunk at most

    // GEMM 1: this wave's 32 clusters x 32 pixels; its slice of w in registers
    const int kt = wave & 1, pt = wave >> 1;
    float wreg[D / 2];
    {
        const int kk = 32 * kt + l31;
#pragma unroll
        for (int j = 0; j < D / 2; ++j) wreg[j] = kk < K ? conv_w[(size_t)kk * D + 2 * j + lh] : 0.f;
    }
    f32x16 vacc[2][TD];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TD; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) vacc[a][b][r] = 0.f;
    float sacc[16];                                     // sum over this lane's pixel slot of a[k][p], k = 16 * wave + j
#pragma unroll
    for (int j = 0; j < 16; ++j) sacc[j] = 0.f;

    const int p4 = (tid & 15) * 4, dr = tid >> 4;       // staging: 4 pixels of channel rows dr + 16 i
    const bool vec = (hw & 3) == 0;
    // the chunk's x -> registers; issued a whole chunk ahead (right after the previous chunk's registers went to LDS), so the loads'
    // latency lies under the two GEMMs and the softmax instead of in front of them (one workgroup per CU: nobody else hides it)
    f32x4 xr[NI];
    auto load_chunk = [&](int c) {
        const int p0 = c * NVM_PC, np = min(NVM_PC, hw - p0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float* row = xi + (size_t)(dr + 16 * i) * hw + p0 + p4;
            if (vec && p4 + 3 < np) xr[i] = *(const f32x4*)row;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[i][e] = p4 + e < np ? row[e] : 0.f;
            }
        }
    };
    if (c0 < c1) load_chunk(c0);
    for (int c = c0; c < c1; ++c) {
        const int np = min(NVM_PC, hw - c * NVM_PC);
        // ---- stage
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) ss[e] += xr[i][e] * xr[i][e];
        __syncthreads();                                // the previous chunk's GEMM 2 has read xs / as / red
        if (normalize_input) {
            *(f32x4*)(red + dr * 64 + p4) = f32x4{ss[0], ss[1], ss[2], ss[3]};
            __syncthreads();
            f32x4 tot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 t = *(const f32x4*)(red + r * 64 + p4);
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[e] += t[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) ss[e] = 1.f / fmaxf(sqrtf(tot[e]), 1e-12f);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) ss[e] = 1.f;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float* dst = xs + (dr + 16 * i) * NVM_LD + p4;
            *(f32x2v*)dst = f32x2v{xr[i][0] * ss[0], xr[i][1] * ss[1]};
            *(f32x2v*)(dst + 2) = f32x2v{xr[i][2] * ss[2], xr[i][3] * ss[3]};
        }
        if (c + 1 < c1) load_chunk(c + 1);
        __syncthreads();
        // ---- GEMM 1: logits
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* bp = xs + lh * NVM_LD + 32 * pt + l31;
#pragma unroll
            for (int j = 0; j < D / 2; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[j], bp[2 * j * NVM_LD], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                as[(32 * kt + 8 * (r >> 2) + 4 * lh + (r & 3)) * NVM_LD + 32 * pt + l31] = acc[r];
        }
        __syncthreads();
        // ---- softmax over the clusters: thread = (pixel = lane, clusters 16 wave .. + 15)
        {
            float lg[16], m = -__builtin_huge_valf();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int kk = 16 * wave + j;
                lg[j] = kk < K ? as[kk * NVM_LD + lane] : -__builtin_huge_valf();
                m = fmaxf(m, lg[j]);
            }
            red[wave * 64 + lane] = m;
            __syncthreads();
            m = fmaxf(fmaxf(red[lane], red[64 + lane]), fmaxf(red[128 + lane], red[192 + lane]));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { lg[j] = __expf(lg[j] - m); sum += lg[j]; }       // (exp(-inf) = 0: clusters beyond K)
            red[256 + wave * 64 + lane] = sum;
            __syncthreads();
            sum = (red[256 + lane] + red[320 + lane]) + (red[384 + lane] + red[448 + lane]);
            const float inv = lane < np ? 1.f / sum : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float a = lg[j] * inv;
                as[(16 * wave + j) * NVM_LD + lane] = a;
                sacc[j] += a;                           // per pixel slot; summed over the wave's lanes ONCE, behind the last chunk
            }
        }
        __syncthreads();
        // ---- GEMM 2: V += a x^T (contraction over the chunk's 64 pixels, two per MFMA)
        {
            const float* ap = as + l31 * NVM_LD + lh;
            const float* bp = xs + (wave * (32 * TD) + l31) * NVM_LD + lh;
#pragma unroll 8
            for (int j = 0; j < NVM_PC / 2; ++j) {
                const float a0 = ap[2 * j], a1 = ap[32 * NVM_LD + 2 * j];
                float b[TD];
#pragma unroll
                for (int t = 0; t < TD; ++t) b[t] = bp[t * 32 * NVM_LD + 2 * j];
#pragma unroll
                for (int t = 0; t < TD; ++t) {
                    vacc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[t], vacc[0][t], 0, 0, 0);
                    vacc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[t], vacc[1][t], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial V [64][D] and S [64] of this (image, split)
    float* vp = vpart + ((size_t)im * nsplit + split) * 64 * D;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < TD; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                vp[(size_t)(32 * a + 8 * (r >> 2) + 4 * lh + (r & 3)) * D + wave * (32 * TD) + 32 * t + l31] = vacc[a][t][r];
    {
        float* sp = spart + ((size_t)im * nsplit + split) * 64 + 16 * wave;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float t = wave_sum(sacc[j]);
            if (lane == 0) sp[j] = t;
        }
    }
#endif
}

// V = sum over the splits (in order) - S c; intra-normalisation, flatten, L2 normalisation.  One workgroup per image, the
// thread map of netvlad_kernel's tail (thread = cluster tid / 4, channels (tid % 4) * D / 4 ..).
template <int DPT>
__global__ __launch_bounds__(256) void netvlad_finish_kernel(const float* __restrict__ vpart, const float* __restrict__ spart, int nsplit,
                                                             const float* __restrict__ cent, int D, int K, float* __restrict__ out) {
    __shared__ float red[256];
    const int tid = threadIdx.x, im = blockIdx.x, k = tid >> 2, ds = tid & 3;
    float v[DPT], asum = 0.f;
#pragma unroll
    for (int j = 0; j < DPT; ++j) v[j] = 0.f;
    for (int s_ = 0; s_ < nsplit; ++s_) {
        const float* vp = vpart + (((size_t)im * nsplit + s_) * 64 + k) * D + ds * DPT;
#pragma unroll
        for (int j = 0; j < DPT; j += 4) {
            const f32x4 t = *(const f32x4*)(vp + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j + e] += t[e];
        }
        asum += spart[((size_t)im * nsplit + s_) * 64 + k];
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
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < DPT; ++j) { v[j] *= inv; tot += v[j] * v[j]; }
    if (k >= K) tot = 0.f;
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

inline int nvm_splits(int n, int hw) {
    const int nchunks = (hw + NVM_PC - 1) / NVM_PC;
    int s = (256 + n - 1) / n;                          // one workgroup per CU (150 KB of LDS each): ONE round of them
    if (s > nchunks) s = nchunks;
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
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

extern "C" int64_t agp_netvlad_workspace_bytes(int n, int d, int hw, int k) {
    (void)k;
    if (n < 1 || hw < 1 || (d != 128 && d != 256)) return 0;        // 0: the matrix-pipe forward does not take this shape
    return (int64_t)n * nvm_splits(n, hw) * 64 * ((int64_t)d + 1) * 4;
}

extern "C" int agp_netvlad_fwd_mfma(const float* x, const float* conv_w, const float* centroids, int n, int d, int hw, int k,
                                    int normalize_input, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x || !conv_w || !centroids || !out || !workspace || n <= 0 || hw <= 0) return AGP_E_BADARG;
    if (k < 1 || k > NV_K) return AGP_E_BADARG;
    if (d != 128 && d != 256) return AGP_E_UNSUPPORTED;
    if (workspace_bytes < agp_netvlad_workspace_bytes(n, d, hw, k)) return AGP_E_BADARG;
    const int splits = nvm_splits(n, hw), nchunks = (hw + NVM_PC - 1) / NVM_PC;
    const int cps = (nchunks + splits - 1) / splits;
    float* vpart = (float*)workspace;
    float* spart = vpart + (size_t)n * splits * 64 * d;
    hipStream_t s = (hipStream_t)stream;
#define NVM_LAUNCH(DD)                                                                                                   \
    do {                                                                                                                 \
        constexpr int lds = ((DD) * NVM_LD + 64 * NVM_LD + 1024) * 4;                                                    \
        static std::atomic<uint64_t> set{0};                                                                             \
        if (!agp_lds_attr((const void*)netvlad_mfma_kernel<DD>, lds, set)) return AGP_E_LAUNCH;                          \
        AGP_LAUNCH(netvlad_mfma_kernel<DD>, dim3(splits, n), dim3(256), lds, s, x, conv_w, hw, k, normalize_input, cps, vpart, spart); \
        AGP_CHECK_LAUNCH();                                                                                              \
        AGP_LAUNCH(netvlad_finish_kernel<(DD) / 4>, dim3(n), dim3(256), 0, s, vpart, spart, splits, centroids, DD, k, out);  \
        AGP_CHECK_LAUNCH();                                                                                              \
    } while (0)
    if (d == 256) NVM_LAUNCH(256);
    else NVM_LAUNCH(128);
#undef NVM_LAUNCH
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
