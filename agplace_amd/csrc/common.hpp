// Shared device helpers for the gfx950 kernels of libagplace_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/agplace_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // one MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned short bf16_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- tuning switches.  The release library reads NO process environment and keeps no switch state (stateless,
// re-entrant): AGP_TUNE(key, default) IS its default -- a compile-time constant -- and the experiment variants, guarded by
// `#if defined(AGP_TUNING)`, are not compiled.  The development build (`make tuning` -> lib/libagplace_hip_tuning.so, loaded by
// the A/B harnesses under tools/ through AGP_HIP_LIB) keeps them and exports agp_debug_set(key, value) (csrc/api.hip).
#if defined(AGP_TUNING)
int agp_tune_lookup(const char* key, int def);
#define AGP_TUNE(key, def) agp_tune_lookup(key, def)
#else
#define AGP_TUNE(key, def) (def)
#endif

// ---- bf16 helpers (round-to-nearest-even through the compiler's __bf16 cast,
//      which lowers to v_cvt_pk_bf16_f32 and keeps NaNs NaN).
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
    return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}
// split v into hi + lo (both bf16): hi = rn(v), lo = rn(v - hi)
__device__ __forceinline__ void split_bf16(float v, bf16_t& hi, bf16_t& lo) {
    hi = f2bf(v);
    lo = f2bf(v - bf2f(hi));
}
__device__ __forceinline__ uint32_t pack2(bf16_t a, bf16_t b) {
    return (uint32_t)a | ((uint32_t)b << 16);
}
// 8 packed bf16 (u32x4) -> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
    }
}
// 8 floats -> split planes (8 bf16 each)
__device__ __forceinline__ void split8(const float* f, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16_t h0, l0, h1, l1;
        split_bf16(f[2 * i], h0, l0);
        split_bf16(f[2 * i + 1], h1, l1);
        hi[i] = pack2(h0, h1);
        lo[i] = pack2(l0, l1);
    }
}

// ---- fp16 helpers.  Conversions saturate at +-65504 instead of overflowing to infinity.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;   // one f16 MFMA A/B fragment
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;        // one f8f6f4 MFMA A/B fragment (32 fp8)
__device__ __forceinline__ bf16_t f2h(float f) {
    const f16_t h = (f16_t)__builtin_fminf(__builtin_fmaxf(f, -65504.f), 65504.f);
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float h2f(bf16_t b) { return (float)__builtin_bit_cast(f16_t, b); }
__device__ __forceinline__ void split_f16(float v, bf16_t& hi, bf16_t& lo) {
    hi = f2h(v);
    lo = f2h(v - h2f(hi));
}
__device__ __forceinline__ void unpack8_h(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = v[i];
        f[2 * i] = h2f((bf16_t)(w & 0xffffu));
        f[2 * i + 1] = h2f((bf16_t)(w >> 16));
    }
}
// Two fp32 -> one packed fp16 pair: ONE v_med3_f32 per value (clamp to [lo, 65504]; lo = 0 folds a ReLU into the clamp,
// lo = -65504 is the plain saturation of f2h) and ONE v_cvt_pk_f16_f32 per pair (gfx950; round to nearest even like
// v_cvt_f16_f32) -- 1.5 VALU per element against ~4.5 for max / min / cvt / pack.  The conv epilogues are VALU-bound on
// exactly this (stem: 64 accumulators per lane, ~8 VALU each = 76 of the kernel's 260 us).
typedef __attribute__((ext_vector_type(2))) float f32x2v;
__device__ __forceinline__ uint32_t pack2_h_lo(float a, float b, float lo) {
    const f32x2v v = {__builtin_amdgcn_fmed3f(a, lo, 65504.f), __builtin_amdgcn_fmed3f(b, lo, 65504.f)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ u32x4 pack8_h_lo(const float* f, float lo) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack2_h_lo(f[2 * i], f[2 * i + 1], lo);
    return o;
}
__device__ __forceinline__ u32x4 pack8_h(const float* f) { return pack8_h_lo(f, -65504.f); }
// element-wise maximum of eight packed NON-NEGATIVE fp16 values (post-ReLU maps): for such values the IEEE order is the
// order of the bit patterns as signed 16-bit integers (-0 = 0x8000 sorts below everything), so v_pk_max_i16 is an exact
// max that no floating-point mode (denormal flushing, NaN quieting) can touch; the max-pool stays in the storage format
typedef __attribute__((ext_vector_type(8))) short s16x8;
__device__ __forceinline__ u32x4 pkmax8_h(const u32x4& a, const u32x4& b) {
    // (the whole 128-bit value in one cast: an element-by-element form compiled to ONE v_pk_max_i16 instead of four, hipcc 7.2)
    return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b)));
}

// ---- feature-map planes (ops.SplitMap).  A map is EITHER a split-bf16 pair (hi, lo != nullptr;
//      value = hi + lo, ~2^-17 relative) OR one fp16 plane (lo == nullptr; 2^-12 relative).  Every map
//      kernel goes through these three, so the storage format is decided by the caller's pointers.
__device__ __forceinline__ void map_load8(const bf16_t* hi, const bf16_t* lo, size_t off, float* v) {
    const u32x4 h = *(const u32x4*)(hi + off);
    if (lo) {
        float l[8];
        unpack8(h, v);
        unpack8(*(const u32x4*)(lo + off), l);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += l[e];
    } else {
        unpack8_h(h, v);
    }
}

// ReLU mask of a stored post-ReLU map from its hi plane alone (2 of its 4 bytes per element): y = hi + lo with
// hi = rn_bf16(y), so y > 0 <=> hi > 0 for every normal y (a positive value the forward stored never rounds to -0 / 0).
__device__ __forceinline__ unsigned pos_mask8_raw(const u32x4& r) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo16 = r[i] & 0xffffu, hi16 = r[i] >> 16;
        m |= ((lo16 & 0x7fffu) != 0 && !(lo16 & 0x8000u)) ? (1u << (2 * i)) : 0u;
        m |= ((hi16 & 0x7fffu) != 0 && !(hi16 & 0x8000u)) ? (1u << (2 * i + 1)) : 0u;
    }
    return m;
}
__device__ __forceinline__ unsigned pos_mask8(const bf16_t* hi, size_t off) { return pos_mask8_raw(*(const u32x4*)(hi + off)); }
__device__ __forceinline__ void map_store8(bf16_t* hi, bf16_t* lo, size_t off, const float* v) {
    if (lo) {
        u32x4 h, l;
        split8(v, h, l);
        *(u32x4*)(hi + off) = h;
        *(u32x4*)(lo + off) = l;
    } else {
        *(u32x4*)(hi + off) = pack8_h(v);
    }
}
// scalar element -> (hi, lo) bit patterns in the map's format
__device__ __forceinline__ void map_split1(float v, bool paired, bf16_t& hi, bf16_t& lo) {
    if (paired) split_bf16(v, hi, lo);
    else { hi = f2h(v); lo = 0; }
}

// ---- exact unsigned division by a runtime-invariant divisor (Granlund-Montgomery)
struct FastDiv {
    uint32_t m, sh1, sh2, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 1 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// A scalar gradient summed over a whole grid in a FIXED order (dL/dp of a GeM exponent): every block hands in its own total
// (thread 0), the block that arrives last adds the partials by index -- the same bits every run, where one atomicAdd per wave
// made the value depend on the order the waves happened to finish in (1e-5 relative from run to run: tools/repeat_stress.py).
// `gp` is the caller's buffer of AGP_GP_FLOATS floats (include/agplace_hip.h): [0] the result (+=), [1 .. 1 + AGP_GP_SLOTS) the
// partials, [1 + AGP_GP_SLOTS] the arrival ticket (zero between calls).  Grids beyond AGP_GP_SLOTS blocks keep the atomic form.
constexpr int AGP_GP_SLOTS = 16384;
__device__ __forceinline__ void agp_grid_sum_ordered(float block_total, float* gp) {
    const int nblocks = (int)gridDim.x;
    if (nblocks > AGP_GP_SLOTS) {
        if (threadIdx.x == 0 && block_total != 0.f) atomicAdd(gp, block_total);
        return;
    }
    __shared__ int s_last;
    unsigned* const ticket = (unsigned*)(gp + 1 + AGP_GP_SLOTS);
    if (threadIdx.x == 0) {
        __hip_atomic_store(gp + 1 + blockIdx.x, block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nblocks - 1);
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
        __threadfence();
        float t = 0.f;
        for (int i = threadIdx.x; i < nblocks; i += 64) t += __hip_atomic_load(gp + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = wave_sum(t);
        if (threadIdx.x == 0) {
            gp[0] += t;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// the block total of a per-thread value, waves added in wave order (blockDim.x = 256)
__device__ __forceinline__ float agp_block_sum_ordered(float v) {
    __shared__ float s_w[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case AGP_ACT_RELU: return fmaxf(v, 0.f);
        case AGP_ACT_TANH: return tanhf(v);
        case AGP_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: it is remembered per device (a process that drives a
// second GPU would otherwise launch kernels that need more than 64 KB of LDS without it there) in a caller-owned atomic mask.
#include <atomic>
static inline bool agp_lds_attr(const void* fn, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}

#define AGP_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return AGP_E_LAUNCH; \
    } while (0)

// Launch wrapper: drop any stale (not ours) sticky HIP error first, so that AGP_CHECK_LAUNCH
// reports only the status of this launch.
#define AGP_LAUNCH (void)hipGetLastError(); hipLaunchKernelGGL
