// Device helpers shared by fusion.hip (forward) and fusion_bwd.hip (backward).
#pragma once
#include "common.hpp"

namespace agp_fusion {

constexpr int FT = 1024;          // threads per workgroup (16 waves)
constexpr int FROWS = 16;         // batch rows per workgroup
constexpr int MAXK = 1024;

__device__ __forceinline__ int yrow_bytes(int K) { return K * 2 + 16; }   // +16 B pad: bank spread

// acc[r] += sum_k W[n][k] * Y[batch][k] for this lane's (batch, 4 features), W resident
template <int KS>
__device__ __forceinline__ f32x4 mfma_resident(const bf16x8 (&wh)[KS], const bf16x8 (&wl)[KS],
                                               const char* yhi, const char* ylo, int yrb, int lane) {
    // three independent accumulator chains (one per product of the split-bf16 form): a single chain of 3*KS dependent
    // MFMAs is pure latency in this one-wave-per-16-features kernel (each f-evaluation of an ODE step waits for it)
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
    const int boff = (lane & 15) * yrb + (lane >> 4) * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 bh = *(const bf16x8*)(yhi + boff + ks * 64);
        const bf16x8 bl = *(const bf16x8*)(ylo + boff + ks * 64);
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], bh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], bl, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], bh, a2, 0, 0, 0);
    }
    return (a0 + a1) + a2;
}

// write this lane's 4 fp32 values (features n..n+3 of one batch row) as split bf16
__device__ __forceinline__ void store_state(char* yhi, char* ylo, int yrb, int lane, int wave,
                                            const f32x4& v) {
    const int off = (lane & 15) * yrb + (wave * 16 + (lane >> 4) * 4) * 2;
    bf16_t h[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) split_bf16(v[r], h[r], l[r]);
    *(u32x2*)(yhi + off) = u32x2{pack2(h[0], h[1]), pack2(h[2], h[3])};
    *(u32x2*)(ylo + off) = u32x2{pack2(l[0], l[1]), pack2(l[2], l[3])};
}

struct OdeSteps {
    float dt[64];
};

template <int ACT>
__device__ __forceinline__ f32x4 act4(const f32x4& z) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = apply_act(z[r], ACT);
    return o;
}


}  // namespace agp_fusion
