// Micro-benchmark (round 4 design aid, not product code): the inner loop of a fused BasicBlock kernel for 64-channel maps.
// A wave keeps the 3x3x64 weights of 32 output channels in registers (36 fragments = 144 VGPRs), reads one X fragment per MFMA
// from an LDS-resident, XOR-swizzled pixel raster (128 B per pixel, 32 pixels per raster row) and runs 36 MFMAs per 32x32 tile;
// epilogue: fma + clamp + fp16 + two ds_write_b128.  Question: which MFMA rate does "W in registers, X from LDS" sustain with
// 4 waves (one per SIMD) or 8 waves (two per SIMD), and with a barrier every TPB tiles?
// build: hipcc -O3 --offload-arch=gfx950 fb_loop.hip -o fb_loop ; run: ./fb_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2v;

__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2v v = {__builtin_amdgcn_fmed3f(a, 0.f, 65504.f), __builtin_amdgcn_fmed3f(b, 0.f, 65504.f)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

template <int NW, int TPB>
__global__ void __launch_bounds__(NW * 64, NW / 4) fb_loop(const _Float16* __restrict__ wsrc, const _Float16* __restrict__ xsrc,
                                                          float* __restrict__ out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XROWS = 16, OROWS = 16;           // input ring rows / output ring rows (4 KB each)
    char* const xs = smem;
    char* const os = smem + XROWS * 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fill the X ring with random fp16 (plain copies)
    for (int i = tid; i < XROWS * 4096 / 16; i += NW * 64) ((u32x4*)xs)[i] = ((const u32x4*)xsrc)[i + blockIdx.x % 7 * 16];
    f16x8 w[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) w[i] = *(const f16x8*)(wsrc + ((size_t)((wave & 1) * 36 + i) * 64 + lane) * 8);
    const int l31 = lane & 31, lh = lane >> 5;
    int A[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int slot = l31 + kx;
        A[kx] = slot * 128 + ((lh ^ ((slot >> 1) & 7)) << 4);
    }
    const int wo = l31 * 128 + (((2 * (wave & 1) * 2 + lh) ^ ((l31 >> 1) & 7)) << 4);
    float sc[16], sh[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { sc[e] = 1.f + 0.01f * e; sh[e] = 0.001f * e; }
    __syncthreads();
    float keep = 0.f;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
            const int row = (s * TPB + t + (wave >> 1)) & 7;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            int i = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ro = __builtin_amdgcn_readfirstlane(((row + ky) & (XROWS - 1)) * 4096);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks, ++i) {
                            const f16x8 x = *(const f16x8*)(xs + ro + (A[kx] ^ ((cc * 4 + ks * 2) << 4)));
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i], x, acc, 0, 0, 0);
                        }
            }
            u32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] = pack2(acc[2 * e] * sc[2 * e] + sh[2 * e], acc[2 * e + 1] * sc[2 * e + 1] + sh[2 * e + 1]);
                o1[e] = pack2(acc[8 + 2 * e] * sc[8 + 2 * e] + sh[8 + 2 * e], acc[8 + 2 * e + 1] * sc[8 + 2 * e + 1] + sh[8 + 2 * e + 1]);
            }
            char* orow = os + ((row + 3 * t) & (OROWS - 1)) * 4096;
            *(u32x4*)(orow + wo) = o0;
            *(u32x4*)(orow + (wo ^ 32)) = o1;
            keep += acc[3];
        }
        __syncthreads();
    }
    if (keep == 1.2345e30f) out[0] = keep;
    if (tid == 0 && blockIdx.x == 0) out[1] = ((float*)os)[5];
}


typedef __attribute__((ext_vector_type(4))) float f32x4;
// the same loop on v_mfma_f32_16x16x32_f16: 2 x 2 tiles of 16 channels x 16 pixels per K = 32 step, the same LDS bytes
template <int NW, int TPB>
__global__ void __launch_bounds__(NW * 64, NW / 4) fb_loop16(const _Float16* __restrict__ wsrc, const _Float16* __restrict__ xsrc,
                                                            float* __restrict__ out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XROWS = 16, OROWS = 16;
    char* const xs = smem;
    char* const os = smem + XROWS * 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < XROWS * 4096 / 16; i += NW * 64) ((u32x4*)xs)[i] = ((const u32x4*)xsrc)[i + blockIdx.x % 7 * 16];
    f16x8 w[36];                                      // [k-step 18][channel tile 2]
#pragma unroll
    for (int i = 0; i < 36; ++i) w[i] = *(const f16x8*)(wsrc + ((size_t)((wave & 1) * 36 + i) * 64 + lane) * 8);
    const int a = lane & 15, q = lane >> 4;
    int A[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int slot = a + 16 * pt + kx;
            A[kx][pt] = slot * 128 + ((q ^ ((slot >> 1) & 7)) << 4);
        }
    const int wo = a * 128 + (((2 * (wave & 1) * 2 + (q & 1)) ^ ((a >> 1) & 7)) << 4);
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[e] = 1.f + 0.01f * e; sh[e] = 0.001f * e; }
    __syncthreads();
    float keep = 0.f;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
            const int row = (s * TPB + t + (wave >> 1)) & 7;
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            int i = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ro = __builtin_amdgcn_readfirstlane(((row + ky) & (XROWS - 1)) * 4096);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx, ++i) {
                        const f16x8 x0 = *(const f16x8*)(xs + ro + (A[kx][0] ^ ((cc * 4) << 4)));
                        const f16x8 x1 = *(const f16x8*)(xs + ro + (A[kx][1] ^ ((cc * 4) << 4)));
                        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * i], x0, acc[0][0], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * i + 1], x0, acc[1][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * i], x1, acc[0][1], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * i + 1], x1, acc[1][1], 0, 0, 0);
                    }
            }
            char* orow = os + ((row + 3 * t) & (OROWS - 1)) * 4096;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                u32x4 o;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    o[2 * ct] = pack2(acc[ct][pt][0] * sc[0] + sh[0], acc[ct][pt][1] * sc[1] + sh[1]);
                    o[2 * ct + 1] = pack2(acc[ct][pt][2] * sc[2] + sh[2], acc[ct][pt][3] * sc[3] + sh[3]);
                }
                *(u32x4*)(orow + pt * 2048 + wo) = o;
            }
            keep += acc[0][0][3];
        }
        __syncthreads();
    }
    if (keep == 1.2345e30f) out[0] = keep;
    if (tid == 0 && blockIdx.x == 0) out[1] = ((float*)os)[5];
}

template <int NW, int TPB, bool M16 = false>
void run(const _Float16* w, const _Float16* x, float* out, int steps, const char* name) {
    const int lds = 32 * 4096;
    hipFuncSetAttribute(M16 ? (const void*)fb_loop16<NW, TPB> : (const void*)fb_loop<NW, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) { if (M16) hipLaunchKernelGGL((fb_loop16<NW, TPB>), dim3(grid), dim3(NW * 64), lds, 0, w, x, out, steps); else hipLaunchKernelGGL((fb_loop<NW, TPB>), dim3(grid), dim3(NW * 64), lds, 0, w, x, out, steps); }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        const double fl = (double)grid * NW * steps * TPB * 36 * 2.0 * 32 * 32 * 16;
        printf("%s NW=%d TPB=%d steps=%d: %.1f us  %.0f TFLOP/s (%.3f of 2.5 PF)\n", name, NW, TPB, steps, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 2500);
    }
}

int main() {
    const size_t nw = 2 * 36 * 64 * 8, nx = 40 * 4096 / 2;
    std::vector<_Float16> hw(nw), hx(nx);
    srand(1);
    for (auto& v : hw) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
    for (auto& v : hx) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    _Float16 *dw, *dx; float* dout;
    hipMalloc(&dw, nw * 2); hipMalloc(&dx, nx * 2); hipMalloc(&dout, 64);
    hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 2>(dw, dx, dout, 400, "32x32x16 two waves per SIMD");
        run<8, 2, true>(dw, dx, dout, 400, "16x16x32 two waves per SIMD");
    }
    run<4, 2>(dw, dx, dout, 400, "32x32x16 one wave per SIMD ");
    run<4, 2, true>(dw, dx, dout, 400, "16x16x32 one wave per SIMD ");
    hipDeviceSynchronize();
    printf("done %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
