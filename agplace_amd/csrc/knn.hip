// Exact squared-L2 kNN (replaces faiss.IndexFlatL2, reference test.py:27-32).
//
// Pass 1 (igemm.hip, EPI_GMIN): split-bf16 MFMA GEMM  acc = (-2 q) . d ; coarse distance
//         c(q,row) = |d|^2 + acc  (|q|^2 is constant per query and irrelevant for ranking);
//         the epilogue keeps only the minimum over each group of 16 database rows
//         -> gmin[group][query], never the [nq][nb] matrix.
// Pass 2: transpose to [query][group].
// Pass 3 (one workgroup per query):
//   a. radix-select the k-th smallest group minimum T_k;
//   b. every group with gmin <= T_k + 2*eps is a candidate, eps = proven bound on
//      |coarse - true|.  Proof of completeness: if T_k < d*_k - eps there would be k rows
//      (one per group) whose true distance is < d*_k, contradicting d*_k being the k-th
//      smallest true distance; and a true top-k row has coarse <= d*_k + eps <= T;
//   c. fp64 direct evaluation sum((q-d)^2) of all rows of the candidate groups from the
//      ORIGINAL fp32 vectors, rank by (distance, index) -> sorted top-k, faiss layout.
#include "common.hpp"

int agp_internal_gmin(const void* q_hi, const void* q_lo, int64_t nq, const void* db_hi,
                      const void* db_lo, const float* db_norm, int64_t nb, int64_t nb_pad, int d,
                      int prec, float* gmin, int gq_stride, hipStream_t s);

namespace agp_knn {

constexpr int KNN_TILE_ROWS = 128;   // database rows per igemm column tile (BN)
constexpr int MAX_ENT = 4096;        // exact-phase entries per round (LDS)
constexpr int MAX_K = 128;

__global__ void db_prep_kernel(const float* __restrict__ xb, int64_t nb, int64_t nb_pad, int d,
                               bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                               float* __restrict__ norm) {
    // one wave per row
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= nb_pad) return;
    float s = 0.f;
    for (int k = lane; k < d; k += 64) {
        const float v = row < nb ? xb[row * d + k] : 0.f;
        bf16_t h, l;
        split_bf16(v, h, l);
        hi[row * d + k] = h;
        if (lo) lo[row * d + k] = l;
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) {
        norm[row] = s;
        if (row < nb) atomicMax((unsigned int*)(norm + nb_pad), __float_as_uint(s));
    }
}

__global__ void q_prep_kernel(const float* __restrict__ xq, int64_t n, bf16_t* __restrict__ hi,
                              bf16_t* __restrict__ lo) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        bf16_t h, l;
        split_bf16(-2.f * xq[i], h, l);
        hi[i] = h;
        lo[i] = l;
    }
}

__global__ void transpose_kernel(const float* __restrict__ in, int rows, int cols, int in_stride,
                                 float* __restrict__ out, int out_stride) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * in_stride + c] : __builtin_huge_valf();
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (c < cols && r < rows) out[(size_t)c * out_stride + r] = tile[threadIdx.x][j];
    }
}

__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void select_rerank_kernel(
    const float* __restrict__ xq, const float* __restrict__ xb, const float* __restrict__ gminT,
    int G, int g_stride, const float* __restrict__ db_norm, int64_t nb, int64_t nb_pad, int d, int k,
    float cerr, float* __restrict__ dist, int64_t* __restrict__ idx) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int s_prefix, s_remain, s_count;
    __shared__ double e_d[MAX_ENT];
    __shared__ int e_i[MAX_ENT];
    __shared__ double best_d[MAX_K];
    __shared__ int best_i[MAX_K];
    __shared__ int s_nbest;

    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* gm = gminT + (size_t)q * g_stride;
    const float* qv = xq + (size_t)q * d;

    // ---- a. radix select of the kk-th smallest group minimum
    const int kk = k < G ? k : G;
    if (tid == 0) { s_prefix = 0; s_remain = kk; s_nbest = 0; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        for (int g = tid; g < G; g += 256) {
            const uint32_t key = fkey(gm[g]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t rem = s_remain, b = 0, cum = 0;
            for (; b < 256; ++b) {
                if (cum + hist[b] >= rem) break;
                cum += hist[b];
            }
            if (b > 255) b = 255;
            s_prefix = prefix | (b << shift);
            s_remain = rem - cum;
        }
        __syncthreads();
    }
    const float Tk = fkey_inv(s_prefix);

    // ---- b. candidate threshold
    float qn = 0.f;
    for (int i = lane; i < d; i += 64) qn += qv[i] * qv[i];
    qn = sqrtf(wave_sum(qn));
    const float dmax2 = db_norm[nb_pad];
    const float eps = 2.f * cerr * qn * sqrtf(dmax2) + 1.2e-7f * (float)d * dmax2 * 0.0625f + 1e-30f;
    const float T = Tk + 2.f * eps;

    // ---- c. exact phase in rounds of <= MAX_ENT entries
    int g_next = 0;   // uniform scan position over groups
    while (g_next < G) {
        // entries [0, nbest) carry the running best; gather candidate rows after them.
        // Groups are scanned 128 at a time (<= 2048 new entries per chunk), and a chunk is
        // only started while it cannot overflow the entry buffer.
        if (tid == 0) s_count = s_nbest;
        __syncthreads();
        int g_base = g_next;
        for (; g_base < G; g_base += 128) {
            const unsigned int cnt = s_count;
            __syncthreads();   // everyone has read cnt before anyone bumps s_count
            if (cnt + 128 * 16 > MAX_ENT) break;
            const int g = g_base + tid;
            if (tid < 128 && g < G && gm[g] <= T) {
                const int tile32 = g >> 1, h = g & 1;
                const unsigned int slot = atomicAdd(&s_count, 16u);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t n = (int64_t)tile32 * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    e_i[slot + r] = n < nb ? (int)n : -1;
                }
            }
            __syncthreads();
        }
        g_next = g_base;
        const int total = s_count;
        const int nbest0 = s_nbest;
        // exact fp64 distances for the new entries: one wave per row
        for (int e = nbest0 + wave; e < total; e += 4) {
            const int n = e_i[e];
            double acc = 0.0;
            if (n >= 0) {
                const float* dv = xb + (size_t)n * d;
                for (int i = lane; i < d; i += 64) {
                    const double t = (double)qv[i] - (double)dv[i];
                    acc += t * t;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == 0) e_d[e] = n >= 0 ? acc : __builtin_huge_val();
        }
        __syncthreads();
        // rank by (distance, index); rank < k survive
        for (int e = tid; e < total; e += 256) {
            const double de = e_d[e];
            const int ie = e_i[e];
            int rank = 0;
            if (ie >= 0) {
                for (int o = 0; o < total; ++o) {
                    const double d2 = e_d[o];
                    const int i2 = e_i[o];
                    rank += (i2 >= 0) && (d2 < de || (d2 == de && i2 < ie));
                }
                if (rank < k) { best_d[rank] = de; best_i[rank] = ie; }
            }
        }
        __syncthreads();
        if (tid == 0) {
            int valid = 0;
            for (int e = 0; e < total; ++e) valid += e_i[e] >= 0;
            s_nbest = valid < k ? valid : k;
        }
        __syncthreads();
        for (int e = tid; e < s_nbest; e += 256) { e_d[e] = best_d[e]; e_i[e] = best_i[e]; }
        __syncthreads();
    }
    for (int e = tid; e < k; e += 256) {
        const bool ok = e < s_nbest;
        dist[(size_t)q * k + e] = ok ? (float)best_d[e] : 3.4028234663852886e38f;
        idx[(size_t)q * k + e] = ok ? (int64_t)best_i[e] : -1;
    }
}

inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

struct KnnWs {
    int64_t q_hi, q_lo, gmin, gminT, total;
    int G, gq_stride, g_stride;
};
inline KnnWs knn_ws(int64_t nq, int64_t nb, int d) {
    KnnWs w;
    const int64_t nb_pad = agp_knn_pad_rows(nb);
    w.G = (int)(nb_pad / 16);
    w.gq_stride = (int)((nq + 31) / 32 * 32);
    w.g_stride = (w.G + 31) / 32 * 32;
    w.q_hi = 0;
    w.q_lo = align256(w.q_hi + nq * d * 2);
    w.gmin = align256(w.q_lo + nq * d * 2);
    w.gminT = align256(w.gmin + (int64_t)w.G * w.gq_stride * 4);
    w.total = align256(w.gminT + nq * (int64_t)w.g_stride * 4);
    return w;
}

}  // namespace agp_knn
using namespace agp_knn;

extern "C" int64_t agp_knn_pad_rows(int64_t nb) {
    if (nb < 1) nb = 1;
    return (nb + KNN_TILE_ROWS - 1) / KNN_TILE_ROWS * KNN_TILE_ROWS;
}

extern "C" int agp_knn_prepare_db(const float* xb, int64_t nb, int d, void* db_hi, void* db_lo,
                                  float* db_norm, void* stream) {
    if (!db_hi || !db_norm || nb < 0 || d <= 0 || d % 32) return AGP_E_BADARG;
    if (nb > 0 && !xb) return AGP_E_BADARG;
    const int64_t nb_pad = agp_knn_pad_rows(nb);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(db_norm + nb_pad, 0, 32 * sizeof(float), s) != hipSuccess) return AGP_E_LAUNCH;
    AGP_LAUNCH(db_prep_kernel, dim3((unsigned)((nb_pad + 3) / 4)), dim3(256), 0, s, xb, nb, nb_pad,
                       d, (bf16_t*)db_hi, (bf16_t*)db_lo, db_norm);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}

extern "C" int64_t agp_knn_workspace_bytes(int64_t nq, int64_t nb, int d, int k) {
    (void)k;
    if (nq < 1) nq = 1;
    return knn_ws(nq, nb, d).total;
}

extern "C" int agp_knn_search(const float* xq, int64_t nq, const float* xb, const void* db_hi,
                              const void* db_lo, const float* db_norm, int64_t nb, int d, int k, int prec,
                              float* dist, int64_t* idx, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (nq == 0) return AGP_OK;
    if (!xq || !db_hi || !db_norm || !dist || !idx || !workspace || nq < 0 || nb < 0) return AGP_E_BADARG;
    if (k < 1 || k > MAX_K || d % 32 || d <= 0) return AGP_E_BADARG;
    if (prec == AGP_PREC_BF16X3 && !db_lo) return AGP_E_BADARG;
    if (nb > 0 && !xb) return AGP_E_BADARG;
    const KnnWs w = knn_ws(nq, nb, d);
    if (workspace_bytes < w.total) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int64_t nb_pad = agp_knn_pad_rows(nb);
    const int64_t nqd = nq * d;
    int g = (int)((nqd + 255) / 256);
    if (g > 4096) g = 4096;
    AGP_LAUNCH(q_prep_kernel, dim3(g), dim3(256), 0, s, xq, nqd, (bf16_t*)(ws + w.q_hi),
                       (bf16_t*)(ws + w.q_lo));
    AGP_CHECK_LAUNCH();
    int rc = agp_internal_gmin(ws + w.q_hi, ws + w.q_lo, nq, db_hi, db_lo, db_norm, nb, nb_pad, d, prec,
                               (float*)(ws + w.gmin), w.gq_stride, s);
    if (rc != AGP_OK) return rc;
    AGP_LAUNCH(transpose_kernel, dim3((unsigned)((nq + 31) / 32), (unsigned)((w.G + 31) / 32)),
                       dim3(32, 8), 0, s, (const float*)(ws + w.gmin), w.G, (int)nq, w.gq_stride,
                       (float*)(ws + w.gminT), w.g_stride);
    AGP_CHECK_LAUNCH();
    // bound on |coarse - true| / (|q| |d|): split-bf16 products + fp32 accumulation, or plain bf16
    const float scale_d = d > 256 ? (float)d / 256.f : 1.f;
    const float cerr = (prec == AGP_PREC_BF16X3 ? 1.2207031e-4f /*2^-13*/ : 7.8125e-3f /*2^-7*/) * scale_d;
    AGP_LAUNCH(select_rerank_kernel, dim3((unsigned)nq), dim3(256), 0, s, xq, xb,
                       (const float*)(ws + w.gminT), w.G, w.g_stride, db_norm, nb, nb_pad, d, k, cerr, dist,
                       idx);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
