// Batched triplet mining: the reference builds one faiss index per query
// (8000 per cache refresh, datasets/datasets_ws_nuscenes.py:1241-1258).  The hardest negatives
// are one agp_knn_search over the sampled database rows (host: agplace_amd/mining.py); this file
// holds the best-positive search over RAGGED per-query candidate lists.
#include "common.hpp"

namespace agp_mining {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wave per query: exact fp64 squared distances from the fp32 vectors, first minimum in list
// order (faiss keeps the earlier of two equal distances)
__global__ void best_positive_kernel(const float* __restrict__ xq, int64_t nq, const float* __restrict__ xb, int64_t nb,
                                     int d, const int64_t* __restrict__ off, const int64_t* __restrict__ idx,
                                     int64_t* __restrict__ out_best, float* __restrict__ out_dist) {
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (q >= nq) return;
    const float* qv = xq + q * d;
    double best = 1.0e300;
    int64_t best_i = -1;
    for (int64_t j = off[q]; j < off[q + 1]; ++j) {
        const int64_t r = idx[j];
        double s = 0.0;
        if (r >= 0 && r < nb) {
            const float* dv = xb + r * d;
            for (int k = lane; k < d; k += 64) {
                const double t = (double)qv[k] - (double)dv[k];
                s += t * t;
            }
            s = wave_sum_f64(s);
        } else {
            s = 1.0e300;
        }
        if (s < best) { best = s; best_i = r; }
    }
    if (lane == 0) {
        out_best[q] = best_i;
        if (out_dist) out_dist[q] = best_i >= 0 ? (float)best : 3.4028234663852886e38f;
    }
}

}  // namespace agp_mining

extern "C" int agp_mine_best_positive(const float* xq, int64_t nq, const float* xb, int64_t nb, int d,
                                      const int64_t* pos_off, const int64_t* pos_idx, int64_t* out_best, float* out_dist,
                                      void* stream) {
    if (!xq || !xb || !pos_off || !out_best || nq < 0 || d <= 0) return AGP_E_BADARG;
    if (nq == 0) return AGP_OK;
    const int64_t blocks = (nq * 64 + 255) / 256;
    AGP_LAUNCH(agp_mining::best_positive_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xq, nq, xb, nb, d,
               pos_off, pos_idx, out_best, out_dist);
    AGP_CHECK_LAUNCH();
    return AGP_OK;
}
