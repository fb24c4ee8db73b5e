// Group normalisation over mesh activations [N, V, C] (reference lib/models.py:681-712, the
// norm_type='group' branch used by res_block_decoder :744-774): G = min(32, C) groups of C/G
// channels, statistics over [C/G, V] per sample, population variance, eps inside the sqrt,
// per-channel gamma/beta.  Statistics are two-pass (mean, then centred sum of squares) like
// tf.nn.moments.  Optional fused ReLU (the reference always applies tf.nn.relu right after).
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float *red) {
    // 256 threads -> one value, broadcast to all
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const float *x, long long xs, int ldx, float eps, int G, int V,
                                                       int C, float *stats) {
    __shared__ float red[4];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;
    const float *xb = x + (long long)n * xs + g * Cg;
    const int total = V * Cg;
    float s = 0.f;
    for (int i = threadIdx.x; i < total; i += 256) s += xb[(long long)(i / Cg) * ldx + (i % Cg)];
    const float mean = block_sum(s, red) / (float)total;
    float q = 0.f;
    for (int i = threadIdx.x; i < total; i += 256) {
        const float d = xb[(long long)(i / Cg) * ldx + (i % Cg)] - mean;
        q = fmaf(d, d, q);
    }
    const float var = block_sum(q, red) / (float)total;
    if (threadIdx.x == 0) {
        stats[2 * blockIdx.x] = mean;
        stats[2 * blockIdx.x + 1] = 1.0f / sqrtf(var + eps);
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float *x, long long xs, int ldx, const float *gamma,
                                                       const float *beta, const float *stats, int G, int relu, float *y,
                                                       long long ys, int ldy, int N, int V, int C) {
    const int Cg = C / G;
    const long long total = (long long)N * V * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long nv = i / C;
        const int v = (int)(nv % V);
        const int n = (int)(nv / V);
        const float *st = stats + 2 * ((long long)n * G + c / Cg);
        float o = (x[(long long)n * xs + (long long)v * ldx + c] - st[0]) * st[1] * gamma[c] + beta[c];
        if (relu) o = o > 0.f ? o : 0.f;
        y[(long long)n * ys + (long long)v * ldy + c] = o;
    }
}

// per (n, g): per-channel sums of dy' and dy'*xhat, and the two group sums weighted by gamma.
// Thread = (row lane, channel of the group): a wave reads whole contiguous channel segments of consecutive rows
// (the per-channel strided loop this replaces touched every cache line Cg times: 290 us per launch at 862 x 544).
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const float *x, long long xs, int ldx, const float *y,
                                                           long long ys, int ldy, const float *dy, long long dys, int lddy,
                                                           const float *gamma, const float *stats, int G, int relu, int V,
                                                           int C, float *dgamma_p, float *dbeta_p, float *gstats) {
    __shared__ float r1[256], r2[256];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    float S1 = 0.f, S2 = 0.f;
    // channel tiles of up to 64 (groups are at most a few dozen channels wide in every CAPE configuration)
    for (int c0 = 0; c0 < Cg; c0 += 64) {
        const int cw = min(64, Cg - c0);
        int cp = 1;
        while (cp < cw) cp <<= 1;                       // lanes per row (power of two >= tile width)
        const int VL = 256 / cp;
        const int cc = threadIdx.x % cp, vl = threadIdx.x / cp;
        const int c = g * Cg + c0 + cc;
        float s1 = 0.f, s2 = 0.f;
        if (cc < cw) {
            for (int v = vl; v < V; v += VL) {
                float d = dy[(long long)n * dys + (long long)v * lddy + c];
                if (relu && !(y[(long long)n * ys + (long long)v * ldy + c] > 0.f)) d = 0.f;
                const float xh = (x[(long long)n * xs + (long long)v * ldx + c] - mean) * rstd;
                s1 += d;
                s2 = fmaf(d, xh, s2);
            }
        }
        __syncthreads();
        r1[threadIdx.x] = s1;
        r2[threadIdx.x] = s2;
        __syncthreads();
        if (threadIdx.x < cw) {                         // fixed-order sum over the row lanes of this channel
            float t1 = 0.f, t2 = 0.f;
            for (int l = 0; l < VL; ++l) {
                t1 += r1[l * cp + threadIdx.x];
                t2 += r2[l * cp + threadIdx.x];
            }
            const int ch = g * Cg + c0 + threadIdx.x;
            dbeta_p[(long long)n * C + ch] = t1;
            dgamma_p[(long long)n * C + ch] = t2;
            r1[threadIdx.x] = gamma[ch] * t1;
            r2[threadIdx.x] = gamma[ch] * t2;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int l = 0; l < cw; ++l) {
                S1 += r1[l];
                S2 += r2[l];
            }
    }
    if (threadIdx.x == 0) {
        gstats[2 * blockIdx.x] = S1;
        gstats[2 * blockIdx.x + 1] = S2;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float *x, long long xs, int ldx, const float *y, long long ys,
                                                           int ldy, const float *dy, long long dys, int lddy,
                                                           const float *gamma, const float *stats, const float *gstats, int G,
                                                           int relu, float *dx, long long dxs, int lddx, int N, int V, int C) {
    const int Cg = C / G;
    const float inv = 1.0f / (float)(V * Cg);
    const long long total = (long long)N * V * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long nv = i / C;
        const int v = (int)(nv % V);
        const int n = (int)(nv / V);
        const long long sg = (long long)n * G + c / Cg;
        const float mean = stats[2 * sg], rstd = stats[2 * sg + 1];
        float d = dy[(long long)n * dys + (long long)v * lddy + c];
        if (relu && !(y[(long long)n * ys + (long long)v * ldy + c] > 0.f)) d = 0.f;
        const float xh = (x[(long long)n * xs + (long long)v * ldx + c] - mean) * rstd;
        dx[(long long)n * dxs + (long long)v * lddx + c] =
            rstd * (gamma[c] * d - (gstats[2 * sg] + xh * gstats[2 * sg + 1]) * inv);
    }
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int cape_groupnorm_fwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *gamma,
                                  const float *beta, float eps, int32_t G, int32_t relu, float *y, int64_t y_sample_stride,
                                  int32_t ldy, float *stats, int32_t N, int32_t V, int32_t C, void *stream) {
    if (!x || !gamma || !beta || !y || !stats || N < 1 || V < 1 || C < 1 || G < 1 || (C % G) != 0 || ldx < C || ldy < C)
        return CAPE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    CAPE_LAUNCH(gn_stats_kernel, dim3(N * G), dim3(256), 0, st, x, (long long)x_sample_stride, ldx, eps, G, V, C, stats);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(gn_apply_kernel, dim3(grid_for((long long)N * V * C)), dim3(256), 0, st, x, (long long)x_sample_stride, ldx,
                       gamma, beta, stats, G, relu, y, (long long)y_sample_stride, ldy, N, V, C);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_groupnorm_bwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *y, int64_t y_sample_stride,
                                  int32_t ldy, const float *dy, int64_t dy_sample_stride, int32_t lddy, const float *gamma,
                                  const float *stats, int32_t G, int32_t relu, float *dx, int64_t dx_sample_stride, int32_t lddx,
                                  float *dgamma_partial, float *dbeta_partial, float *gstats, int32_t N, int32_t V, int32_t C,
                                  void *stream) {
    if (!x || !dy || !gamma || !stats || !dx || !dgamma_partial || !dbeta_partial || !gstats || N < 1 || V < 1 || C < 1 ||
        G < 1 || (C % G) != 0 || ldx < C || lddy < C || lddx < C)
        return CAPE_EINVAL;
    if (relu && (!y || ldy < C)) return CAPE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    CAPE_LAUNCH(gn_bwd_stats_kernel, dim3(N * G), dim3(256), 0, st, x, (long long)x_sample_stride, ldx, y,
                       (long long)y_sample_stride, ldy, dy, (long long)dy_sample_stride, lddy, gamma, stats, G, relu, V, C,
                       dgamma_partial, dbeta_partial, gstats);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(gn_bwd_apply_kernel, dim3(grid_for((long long)N * V * C)), dim3(256), 0, st, x, (long long)x_sample_stride,
                       ldx, y, (long long)y_sample_stride, ldy, dy, (long long)dy_sample_stride, lddy, gamma, stats, gstats, G,
                       relu, dx, (long long)dx_sample_stride, lddx, N, V, C);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
