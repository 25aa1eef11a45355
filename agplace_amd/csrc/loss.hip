// Training losses fused into a handful of launches.  Tiny tensors
// ([b*12, 256] descriptors): the reference spends ~60 ATen launches per step on them
// (train.py:51-61 TripletMarginLoss over 10 index views; compute_other_loss.py:56-113 four
// cdist + mask + BCE terms).  All arithmetic fp32 with fp64 reductions; every reduction runs in a
// fixed order (no atomics), so the losses and their gradients are bit-reproducible.
#include "common.hpp"

namespace agp_loss {

__device__ __forceinline__ double wsum64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- nn.TripletMarginLoss(margin, p=2, eps=1e-6, reduction='sum'): one wave per triplet.
//      d(a,b) = ||a - b + eps||_2 ; loss_t = max(0, d(q,p) - d(q,n) + margin)
//      per-triplet record: loss_t, 1/d(q,p), 1/d(q,n) (0 when inactive) for the gradient pass
__global__ void triplet_fwd_kernel(const float* __restrict__ f, int d, const int64_t* __restrict__ trip, int nt, float margin,
                                   float eps, float* __restrict__ rec) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (t >= nt) return;
    const float* q = f + trip[3 * t] * d;
    const float* p = f + trip[3 * t + 1] * d;
    const float* n = f + trip[3 * t + 2] * d;
    double sp = 0, sn = 0;
    for (int k = lane; k < d; k += 64) {
        const float a = q[k] - p[k] + eps, b = q[k] - n[k] + eps;
        sp += (double)a * a;
        sn += (double)b * b;
    }
    sp = wsum64(sp); sn = wsum64(sn);
    if (lane == 0) {
        const float dp = (float)sqrt(sp), dn = (float)sqrt(sn);
        const float l = dp - dn + margin;
        const bool act = l > 0.f;
        rec[3 * t] = act ? l : 0.f;
        rec[3 * t + 1] = (act && dp > 0.f) ? 1.f / dp : 0.f;
        rec[3 * t + 2] = (act && dn > 0.f) ? 1.f / dn : 0.f;
    }
}

// gradient of sum_t loss_t w.r.t. one feature row: one block per row, fixed triplet order
__global__ void triplet_bwd_kernel(const float* __restrict__ f, int d, const int64_t* __restrict__ trip, int nt, float eps,
                                   const float* __restrict__ rec, float* __restrict__ g) {
    const int row = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float acc = 0.f;
        for (int t = 0; t < nt; ++t) {
            const int64_t iq = trip[3 * t], ip = trip[3 * t + 1], in = trip[3 * t + 2];
            if (iq != row && ip != row && in != row) continue;
            const float rp = rec[3 * t + 1], rn = rec[3 * t + 2];
            if (rp == 0.f && rn == 0.f) continue;
            const float a = f[iq * d + k] - f[ip * d + k] + eps, b = f[iq * d + k] - f[in * d + k] + eps;
            if (iq == row) acc += a * rp - b * rn;
            if (ip == row) acc -= a * rp;
            if (in == row) acc += b * rn;
        }
        g[(size_t)row * d + k] = acc;
    }
}

// ---- SARE criteria (reference model/functional.py:5-27, train.py:62-74): per group g of `group` consecutive triplets
//      (1 for sare_ind, 10 for sare_joint) with the query and positive of the group's FIRST triplet,
//      loss_g = -log_softmax([-|q-p|^2, -|q-n_1|^2, ...])[0] = log(1 + sum_j exp(|q-p|^2 - |q-n_j|^2)).
//      Pass 1 (one wave per triplet): squared distances into rec[3t+1], rec[3t+2].
__global__ void sare_dist_kernel(const float* __restrict__ f, int d, const int64_t* __restrict__ trip, int nt, int group,
                                 float* __restrict__ rec) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (t >= nt) return;
    const int t0 = t / group * group;
    const float* q = f + trip[3 * t0] * d;
    const float* p = f + trip[3 * t0 + 1] * d;
    const float* n = f + trip[3 * t + 2] * d;
    double sp = 0, sn = 0;
    for (int k = lane; k < d; k += 64) {
        const float a = q[k] - p[k], b = q[k] - n[k];
        sp += (double)a * a;
        sn += (double)b * b;
    }
    sp = wsum64(sp); sn = wsum64(sn);
    if (lane == 0) { rec[3 * t + 1] = (float)sp; rec[3 * t + 2] = (float)sn; }
}

//      Pass 2 (one thread per group): the log-sum-exp and the gradient coefficients.  rec[3t] = loss (first triplet of the
//      group, 0 elsewhere); rec[3t+1] = 2 (1 - s_0) on the first triplet (0 elsewhere); rec[3t+2] = 2 s_j, s = softmax.
__global__ void sare_group_kernel(int nt, int group, float* __restrict__ rec) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int t0 = g * group;
    if (t0 >= nt) return;
    const int cnt = min(group, nt - t0);
    const double x0 = -(double)rec[3 * t0 + 1];
    double m = x0;
    for (int j = 0; j < cnt; ++j) m = fmax(m, -(double)rec[3 * (t0 + j) + 2]);
    double se = exp(x0 - m);
    for (int j = 0; j < cnt; ++j) se += exp(-(double)rec[3 * (t0 + j) + 2] - m);
    const double lse = m + log(se);
    for (int j = 0; j < cnt; ++j) {
        const double sj = exp(-(double)rec[3 * (t0 + j) + 2] - lse);
        rec[3 * (t0 + j)] = 0.f;
        rec[3 * (t0 + j) + 1] = 0.f;
        rec[3 * (t0 + j) + 2] = (float)(2.0 * sj);
    }
    rec[3 * t0] = (float)(lse - x0);
    rec[3 * t0 + 1] = (float)(2.0 * (1.0 - exp(x0 - lse)));
}

//      Gradient w.r.t. one feature row (one block per row, fixed triplet order): with a = q - p, b_j = q - n_j,
//      dL/dq = cp a - sum_j cn_j b_j,  dL/dp = -cp a,  dL/dn_j = cn_j b_j.
__global__ void sare_bwd_kernel(const float* __restrict__ f, int d, const int64_t* __restrict__ trip, int nt, int group,
                                const float* __restrict__ rec, float* __restrict__ g) {
    const int row = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float acc = 0.f;
        for (int t = 0; t < nt; ++t) {
            const int t0 = t / group * group;
            const int64_t iq = trip[3 * t0], ip = trip[3 * t0 + 1], in = trip[3 * t + 2];
            if (iq != row && ip != row && in != row) continue;
            const float cp = rec[3 * t + 1], cn = rec[3 * t + 2];
            const float a = f[iq * d + k] - f[ip * d + k], b = f[iq * d + k] - f[in * d + k];
            if (iq == row) acc += a * cp - b * cn;
            if (ip == row) acc -= a * cp;
            if (in == row) acc += b * cn;
        }
        g[(size_t)row * d + k] = acc;
    }
}

// fixed-order sum of n floats (+ optional count of entries != skip) by one block
__global__ void sum_kernel(const float* __restrict__ v, int64_t n, int stride, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i * stride];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// ---- one term of compute_other_loss: dist = cdist(X, Y); target from the east/north distance
//      (0 if < pos_thd, 1 if > neg_thd, ignored otherwise); elementwise loss on the kept pairs:
//      type 0 'bce' : BCEWithLogits(dist, target);  1 'mse': (sigmoid(dist)-target)^2;  2 'l1': |sigmoid(dist)-target|
//      pair record: loss_ij (0 when ignored), keep_ij (0/1), coef_ij = dloss_ij/ddist / dist (0 at dist = 0)
__global__ void pair_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, int m, int d,
                                const float* __restrict__ ex, const float* __restrict__ ey, float pos_thd, float neg_thd,
                                int type, float* __restrict__ lossm, float* __restrict__ keepm, float* __restrict__ coefm) {
    const int64_t pid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (pid >= (int64_t)n * m) return;
    const int i = (int)(pid / m), j = (int)(pid % m);
    double s = 0;
    for (int k = lane; k < d; k += 64) {
        const float t = x[(size_t)i * d + k] - y[(size_t)j * d + k];
        s += (double)t * t;
    }
    s = wsum64(s);
    if (lane) return;
    const float dist = (float)sqrt(s);
    const float dx = ex[2 * i] - ey[2 * j], dy = ex[2 * i + 1] - ey[2 * j + 1];
    const float ed = sqrtf(dx * dx + dy * dy);
    float tgt = -1.f;
    if (ed < pos_thd) tgt = 0.f;
    if (ed > neg_thd) tgt = 1.f;
    float l = 0.f, c = 0.f;
    if (tgt >= 0.f) {
        const float sg = 1.f / (1.f + __expf(-dist));
        if (type == 0) {            // max(z,0) - z*t + log(1 + exp(-|z|)), z = dist >= 0
            l = dist - dist * tgt + log1pf(__expf(-dist));
            c = sg - tgt;
        } else if (type == 1) {
            l = (sg - tgt) * (sg - tgt);
            c = 2.f * (sg - tgt) * sg * (1.f - sg);
        } else {
            l = fabsf(sg - tgt);
            c = (sg > tgt ? 1.f : (sg < tgt ? -1.f : 0.f)) * sg * (1.f - sg);
        }
        c = dist > 0.f ? c / dist : 0.f;
    }
    lossm[pid] = l;
    keepm[pid] = tgt >= 0.f ? 1.f : 0.f;
    coefm[pid] = c;
}

// gx[i] = sum_j coef_ij (x_i - y_j)   (block per row i);  with transpose=1: gy[j] = sum_i coef_ij (y_j - x_i)
__global__ void pair_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, int m, int d,
                                const float* __restrict__ coefm, int transpose, float* __restrict__ g) {
    const int r = blockIdx.x;
    const float* self = transpose ? y + (size_t)r * d : x + (size_t)r * d;
    const int cnt = transpose ? n : m;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float acc = 0.f;
        const float sv = self[k];
        for (int o = 0; o < cnt; ++o) {
            const float c = transpose ? coefm[(size_t)o * m + r] : coefm[(size_t)r * m + o];
            if (c != 0.f) acc += c * (sv - (transpose ? x[(size_t)o * d + k] : y[(size_t)o * d + k]));
        }
        g[(size_t)r * d + k] = acc;
    }
}

}  // namespace agp_loss
using namespace agp_loss;

extern "C" int64_t agp_triplet_loss_workspace_floats(int nt) { return 3 * (int64_t)nt; }

extern "C" int agp_triplet_loss(const float* feats, int nrows, int d, const int64_t* triplets, int nt, float margin,
                                float* loss_sum, float* grad_feats, float* workspace, void* stream) {
    if (!feats || !triplets || !loss_sum || !workspace || nrows <= 0 || d <= 0 || nt <= 0) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const float eps = 1e-6f;
    AGP_LAUNCH(triplet_fwd_kernel, dim3((nt * 64 + 255) / 256), dim3(256), 0, s, feats, d, triplets, nt, margin, eps, workspace);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sum_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, (int64_t)nt, 3, loss_sum);
    AGP_CHECK_LAUNCH();
    if (grad_feats) {
        AGP_LAUNCH(triplet_bwd_kernel, dim3(nrows), dim3(256), 0, s, feats, d, triplets, nt, eps, (const float*)workspace,
                   grad_feats);
        AGP_CHECK_LAUNCH();
    }
    return AGP_OK;
}

extern "C" int agp_sare_loss(const float* feats, int nrows, int d, const int64_t* triplets, int nt, int group,
                             float* loss_sum, float* grad_feats, float* workspace, void* stream) {
    if (!feats || !triplets || !loss_sum || !workspace || nrows <= 0 || d <= 0 || nt <= 0 || group <= 0) return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    AGP_LAUNCH(sare_dist_kernel, dim3((nt * 64 + 255) / 256), dim3(256), 0, s, feats, d, triplets, nt, group, workspace);
    AGP_CHECK_LAUNCH();
    const int ng = (nt + group - 1) / group;
    AGP_LAUNCH(sare_group_kernel, dim3((ng + 63) / 64), dim3(64), 0, s, nt, group, workspace);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sum_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, (int64_t)nt, 3, loss_sum);
    AGP_CHECK_LAUNCH();
    if (grad_feats) {
        AGP_LAUNCH(sare_bwd_kernel, dim3(nrows), dim3(256), 0, s, feats, d, triplets, nt, group, (const float*)workspace, grad_feats);
        AGP_CHECK_LAUNCH();
    }
    return AGP_OK;
}

extern "C" int64_t agp_pairdist_loss_workspace_floats(int n, int m) { return 3 * (int64_t)n * m; }

extern "C" int agp_pairdist_loss(const float* x, const float* y, int n, int m, int d, const float* ex, const float* ey,
                                 float pos_thd, float neg_thd, int type, float* loss_sum, float* count, float* gx, float* gy,
                                 float* workspace, void* stream) {
    if (!x || !y || !ex || !ey || !loss_sum || !count || !workspace || n <= 0 || m <= 0 || d <= 0 || type < 0 || type > 2)
        return AGP_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t np = (int64_t)n * m;
    float* lossm = workspace;
    float* keepm = workspace + np;
    float* coefm = workspace + 2 * np;
    AGP_LAUNCH(pair_fwd_kernel, dim3((unsigned)((np * 64 + 255) / 256)), dim3(256), 0, s, x, y, n, m, d, ex, ey, pos_thd, neg_thd,
               type, lossm, keepm, coefm);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sum_kernel, dim3(1), dim3(256), 0, s, (const float*)lossm, np, 1, loss_sum);
    AGP_CHECK_LAUNCH();
    AGP_LAUNCH(sum_kernel, dim3(1), dim3(256), 0, s, (const float*)keepm, np, 1, count);
    AGP_CHECK_LAUNCH();
    if (gx) {
        AGP_LAUNCH(pair_bwd_kernel, dim3(n), dim3(256), 0, s, x, y, n, m, d, (const float*)coefm, 0, gx);
        AGP_CHECK_LAUNCH();
    }
    if (gy) {
        AGP_LAUNCH(pair_bwd_kernel, dim3(m), dim3(256), 0, s, x, y, n, m, d, (const float*)coefm, 1, gy);
        AGP_CHECK_LAUNCH();
    }
    return AGP_OK;
}
