// Optimiser step on the flat parameter / gradient / momentum buckets, and the VAE sampling + KL op.
//
// The reference clips the gradient by its global norm (5.0, lib/models.py:461) and applies
// tf.train.MomentumOptimizer (non-Nesterov: accum = momentum*accum + g; var -= lr*accum, :448-456); the dense
// kernels under 'generator' carry an L2 regulariser (:40, :378-379) whose gradient is coef*w.  On flat buckets
// that is: one two-stage sum of squares of (g + coef*w on the regularised ranges) and one fused update pass --
// 3 launches instead of the ~11 element-wise launches of an op-by-op optimiser (each dispatch costs ~5 us
// against a 5 ms step, and the fused pass moves 5 instead of 12 bucket-sized streams through HBM).
#include "common.h"

namespace {

constexpr int FLAT_BLOCKS = 1024;
constexpr int MAX_RANGES = 8;

struct Ranges {
    long long b[MAX_RANGES], e[MAX_RANGES];
    int n;
    float coef;
};

__device__ __forceinline__ float reg_coef_at(const Ranges &R, long long i) {
    float c = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_RANGES; ++k)
        if (k < R.n && i >= R.b[k] && i < R.e[k]) c = R.coef;
    return c;
}

__device__ __forceinline__ float block_sum(float v, float *red) {
    // fixed-order tree over the 256 threads (deterministic)
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float out = red[0];
    __syncthreads();
    return out;
}

// partial[b] = sum over this block's grid-stride elements of (g + coef*w)^2      (n % 4 == 0, 16-byte aligned)
// (gs: factor on the raw gradient -- 1 / world when the bucket holds the SUM over data-parallel ranks, so the mean needs no launch)
__global__ __launch_bounds__(256) void gradnorm_partial_kernel(const float *g, const float *w, long long n4, Ranges R, float gs, float *partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)FLAT_BLOCKS * 256) {
        float4 gv = reinterpret_cast<const float4 *>(g)[i];
        gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs;
        const float c = R.n ? reg_coef_at(R, 4 * i) : 0.f;     // ranges are multiples of 4 long and 4-aligned
        if (c != 0.f) {
            const float4 wv = reinterpret_cast<const float4 *>(w)[i];
            gv.x = fmaf(c, wv.x, gv.x); gv.y = fmaf(c, wv.y, gv.y); gv.z = fmaf(c, wv.z, gv.z); gv.w = fmaf(c, wv.w, gv.w);
        }
        s = fmaf(gv.x, gv.x, s); s = fmaf(gv.y, gv.y, s); s = fmaf(gv.z, gv.z, s); s = fmaf(gv.w, gv.w, s);
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// partial[b] = sum of x^2 over the listed ranges
__global__ __launch_bounds__(256) void sumsq_ranges_partial_kernel(const float *x, Ranges R, float *partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (int k = 0; k < R.n; ++k) {
        const long long b4 = R.b[k] >> 2, e4 = R.e[k] >> 2;
        for (long long i = b4 + (long long)blockIdx.x * 256 + threadIdx.x; i < e4; i += (long long)FLAT_BLOCKS * 256) {
            const float4 v = reinterpret_cast<const float4 *>(x)[i];
            s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
        }
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void flat_final_kernel(const float *partial, float scale, float *out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < FLAT_BLOCKS; i += 256) s += partial[i];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) *out = scale * t;
}

__global__ __launch_bounds__(256) void momentum_update_kernel(float *w, const float *g, float *m, long long n4, float momentum, float clip,
                                                              const float *sumsq, const float *neg_lr, Ranges R, float gs) {
    const float norm = sqrtf(*sumsq);
    const float scale = clip / fmaxf(norm, clip);       // tf.clip_by_global_norm
    const float nlr = *neg_lr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 gv = reinterpret_cast<const float4 *>(g)[i];
        float4 wv = reinterpret_cast<float4 *>(w)[i];
        float4 mv = reinterpret_cast<float4 *>(m)[i];
        gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs;
        const float c = R.n ? reg_coef_at(R, 4 * i) : 0.f;
        if (c != 0.f) {
            gv.x = fmaf(c, wv.x, gv.x); gv.y = fmaf(c, wv.y, gv.y); gv.z = fmaf(c, wv.z, gv.z); gv.w = fmaf(c, wv.w, gv.w);
        }
        mv.x = fmaf(momentum, mv.x, gv.x * scale); mv.y = fmaf(momentum, mv.y, gv.y * scale);
        mv.z = fmaf(momentum, mv.z, gv.z * scale); mv.w = fmaf(momentum, mv.w, gv.w * scale);
        wv.x = fmaf(nlr, mv.x, wv.x); wv.y = fmaf(nlr, mv.y, wv.y); wv.z = fmaf(nlr, mv.z, wv.z); wv.w = fmaf(nlr, mv.w, wv.w);
        reinterpret_cast<float4 *>(m)[i] = mv;
        reinterpret_cast<float4 *>(w)[i] = wv;
    }
}


// tf.train.AdamOptimizer (reference lib/models.py:447-449; beta1 0.9, beta2 0.999, epsilon 1e-8 are TensorFlow's defaults) on the
// same flat buckets, behind the same global-norm clip and with the same folded regulariser gradient as the momentum form:
//     t = step + 1;  lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
//     m = beta1 m + (1 - beta1) g';  v = beta2 v + (1 - beta2) g'^2;  w -= lr_t * m / (sqrt(v) + eps)       (g' = clipped g + reg)
// The step count lives on the device (state[0]; state[1] = finished-workgroup counter): every workgroup reads state[0] when it
// starts, and the LAST one to finish -- by then every workgroup has read it -- advances it, so a captured graph replays the
// right bias correction without any host-side counter.  (An integer ticket: no floating-point result depends on arrival order.)
__global__ __launch_bounds__(256) void adam_update_kernel(float *w, const float *g, float *m, float *v, long long n4, float beta1, float beta2,
                                                          float eps, float clip, const float *sumsq, const float *neg_lr, int *state, Ranges R, float gs) {
    __shared__ float s_nlr;
    if (threadIdx.x == 0) {
        const double t = (double)(state[0] + 1);
        s_nlr = (float)((double)*neg_lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    }
    __syncthreads();
    const float nlr = s_nlr;
    const float norm = sqrtf(*sumsq);
    const float scale = clip / fmaxf(norm, clip);       // tf.clip_by_global_norm
    const float c1 = 1.f - beta1, c2 = 1.f - beta2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 gv = reinterpret_cast<const float4 *>(g)[i];
        float4 wv = reinterpret_cast<float4 *>(w)[i];
        float4 mv = reinterpret_cast<float4 *>(m)[i];
        float4 vv = reinterpret_cast<float4 *>(v)[i];
        gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs;
        const float c = R.n ? reg_coef_at(R, 4 * i) : 0.f;
        if (c != 0.f) {
            gv.x = fmaf(c, wv.x, gv.x); gv.y = fmaf(c, wv.y, gv.y); gv.z = fmaf(c, wv.z, gv.z); gv.w = fmaf(c, wv.w, gv.w);
        }
        gv.x *= scale; gv.y *= scale; gv.z *= scale; gv.w *= scale;
        mv.x = fmaf(beta1, mv.x, c1 * gv.x); mv.y = fmaf(beta1, mv.y, c1 * gv.y);
        mv.z = fmaf(beta1, mv.z, c1 * gv.z); mv.w = fmaf(beta1, mv.w, c1 * gv.w);
        vv.x = fmaf(beta2, vv.x, c2 * gv.x * gv.x); vv.y = fmaf(beta2, vv.y, c2 * gv.y * gv.y);
        vv.z = fmaf(beta2, vv.z, c2 * gv.z * gv.z); vv.w = fmaf(beta2, vv.w, c2 * gv.w * gv.w);
        wv.x = fmaf(nlr, mv.x / (sqrtf(vv.x) + eps), wv.x); wv.y = fmaf(nlr, mv.y / (sqrtf(vv.y) + eps), wv.y);
        wv.z = fmaf(nlr, mv.z / (sqrtf(vv.z) + eps), wv.z); wv.w = fmaf(nlr, mv.w / (sqrtf(vv.w) + eps), wv.w);
        reinterpret_cast<float4 *>(m)[i] = mv;
        reinterpret_cast<float4 *>(v)[i] = vv;
        reinterpret_cast<float4 *>(w)[i] = wv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&state[1], 1) == (int)gridDim.x - 1) {
            state[1] = 0;
            atomicAdd(&state[0], 1);
        }
    }
}

inline int fill_ranges(Ranges &R, const int64_t *ranges, int nr, float coef) {
    if (nr < 0 || nr > MAX_RANGES || (nr > 0 && !ranges)) return CAPE_EINVAL;
    R.n = nr; R.coef = coef;
    for (int k = 0; k < nr; ++k) {
        R.b[k] = ranges[2 * k]; R.e[k] = ranges[2 * k + 1];
        if (R.b[k] < 0 || R.e[k] < R.b[k] || (R.b[k] & 3) || (R.e[k] & 3)) return CAPE_EINVAL;
    }
    return CAPE_OK;
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- VAE sampling + KL (reference lib/models.py:193-196, :371-372) on [N, nz] tensors: one block
// z is written with row stride ldz; cond (may be null) [N, Cc] is copied behind it: the decoder's input [z | cond] of
// lib/models.py:296 (tf.concat) comes out of the sampling launch itself
__global__ __launch_bounds__(256) void vae_fwd_kernel(const float *mean, const float *logvar, const float *eps, float *z, int ldz, float *kl,
                                                      int N, int nz, const float *cond, int ldc, int Cc) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < N * nz; i += 256) {
        const float lv = logvar[i], mu = mean[i];
        const float sd = expf(0.5f * lv);
        z[(i / nz) * ldz + (i % nz)] = fmaf(sd, eps[i], mu);
        s += 1.f + lv - mu * mu - sd * sd;
    }
    if (cond)
        for (int i = threadIdx.x; i < N * Cc; i += 256) z[(i / Cc) * ldz + nz + (i % Cc)] = cond[(i / Cc) * ldc + (i % Cc)];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) *kl = (-0.5f / (float)N) * t;
}

__global__ __launch_bounds__(256) void vae_bwd_kernel(const float *mean, const float *logvar, const float *eps, const float *gz, int ldgz,
                                                      const float *gkl, float *dmean, float *dlogvar, int N, int nz) {
    const float c = (gkl ? *gkl : 0.f) / (float)N;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N * nz; i += gridDim.x * 256) {
        const float lv = logvar[i], mu = mean[i];
        const float sd = expf(0.5f * lv);
        const float g = gz ? gz[(i / nz) * ldgz + (i % nz)] : 0.f;
        dmean[i] = fmaf(c, mu, g);
        dlogvar[i] = 0.5f * (g * sd * eps[i] + c * (sd * sd - 1.f));
    }
}

}  // namespace

extern "C" int64_t cape_flat_workspace_bytes(void) { return (int64_t)FLAT_BLOCKS * sizeof(float); }

extern "C" int cape_flat_gradnorm(const float *g, const float *w, int64_t n, const int64_t *reg_ranges, int32_t nranges,
                                  float reg_coef, float grad_scale, float *sumsq_out, void *workspace, int64_t workspace_bytes, void *stream) {
    if (!(grad_scale > 0.f)) return CAPE_EINVAL;
    if (!g || n < 4 || (n & 3) || !sumsq_out || !workspace || !al16(g) || (nranges > 0 && (!w || !al16(w)))) return CAPE_EINVAL;
    if (workspace_bytes < cape_flat_workspace_bytes()) return CAPE_EWORKSPACE;
    Ranges R;
    int rc = fill_ranges(R, reg_ranges, nranges, reg_coef);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    CAPE_LAUNCH(gradnorm_partial_kernel, dim3(FLAT_BLOCKS), dim3(256), 0, st, g, w, (long long)(n >> 2), R, grad_scale, (float *)workspace);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(flat_final_kernel, dim3(1), dim3(256), 0, st, (const float *)workspace, 1.0f, sumsq_out);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

__global__ void spin_kernel(long long ticks) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int cape_spin_us(int32_t us, void *stream) {
    if (us < 0 || us > 100000) return CAPE_EINVAL;
    if (us == 0) return CAPE_OK;
    CAPE_LAUNCH(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)us * 100);        // s_memrealtime: 100 MHz
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_sumsq_ranges(const float *x, const int64_t *ranges, int32_t nranges, float scale, float *out,
                                 void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !out || !workspace || nranges < 1 || !al16(x)) return CAPE_EINVAL;
    if (workspace_bytes < cape_flat_workspace_bytes()) return CAPE_EWORKSPACE;
    Ranges R;
    int rc = fill_ranges(R, ranges, nranges, 0.f);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    CAPE_LAUNCH(sumsq_ranges_partial_kernel, dim3(FLAT_BLOCKS), dim3(256), 0, st, x, R, (float *)workspace);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(flat_final_kernel, dim3(1), dim3(256), 0, st, (const float *)workspace, scale, out);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_flat_momentum_update(float *w, const float *g, float *m, int64_t n, float momentum, float clip,
                                         const float *sumsq, const float *neg_lr, const int64_t *reg_ranges,
                                         int32_t nranges, float reg_coef, float grad_scale, void *stream) {
    if (!(grad_scale > 0.f)) return CAPE_EINVAL;
    if (!w || !g || !m || n < 4 || (n & 3) || !sumsq || !neg_lr || !al16(w) || !al16(g) || !al16(m) || clip <= 0.f) return CAPE_EINVAL;
    Ranges R;
    int rc = fill_ranges(R, reg_ranges, nranges, reg_coef);
    if (rc) return rc;
    const long long n4 = n >> 2;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    CAPE_LAUNCH(momentum_update_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, g, m, n4, momentum, clip, sumsq,
                neg_lr, R, grad_scale);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_flat_adam_update(float *w, const float *g, float *m, float *v, int64_t n, float beta1, float beta2, float eps,
                                     float clip, const float *sumsq, const float *neg_lr, int32_t *state,
                                     const int64_t *reg_ranges, int32_t nranges, float reg_coef, float grad_scale, void *stream) {
    if (!(grad_scale > 0.f)) return CAPE_EINVAL;
    if (!w || !g || !m || !v || n < 4 || (n & 3) || !sumsq || !neg_lr || !state || !al16(w) || !al16(g) || !al16(m) || !al16(v) || clip <= 0.f)
        return CAPE_EINVAL;
    if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps > 0.f)) return CAPE_EINVAL;
    Ranges R;
    int rc = fill_ranges(R, reg_ranges, nranges, reg_coef);
    if (rc) return rc;
    const long long n4 = n >> 2;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    CAPE_LAUNCH(adam_update_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n4, beta1, beta2, eps, clip,
                sumsq, neg_lr, (int *)state, R, grad_scale);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_vae_sample_kl_fwd(const float *mean, const float *logvar, const float *eps, float *z, int32_t ldz, float *kl,
                                      int32_t N, int32_t nz, const float *cond, int32_t ldc, int32_t Cc, void *stream) {
    if (!mean || !logvar || !eps || !z || !kl || N < 1 || nz < 1 || ldz < nz + (cond ? Cc : 0) || (cond && (Cc < 1 || ldc < Cc))) return CAPE_EINVAL;
    CAPE_LAUNCH(vae_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean, logvar, eps, z, ldz, kl, N, nz, cond, ldc, cond ? Cc : 0);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_vae_sample_kl_bwd(const float *mean, const float *logvar, const float *eps, const float *gz, int32_t ldgz,
                                      const float *gkl, float *dmean, float *dlogvar, int32_t N, int32_t nz, void *stream) {
    if (!mean || !logvar || !eps || !dmean || !dlogvar || N < 1 || nz < 1 || (gz && ldgz < nz)) return CAPE_EINVAL;
    const int blocks = (N * nz + 255) / 256 > 64 ? 64 : (N * nz + 255) / 256;
    CAPE_LAUNCH(vae_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mean, logvar, eps, gz, ldgz, gkl, dmean, dlogvar, N, nz);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
