// Group normalisation over mesh activations [N, V, C] (reference lib/models.py:681-712, the
// norm_type='group' branch used by res_block_decoder :744-774): G = min(32, C) groups of C/G
// channels, statistics over [C/G, V] per sample, population variance, eps inside the sqrt,
// per-channel gamma/beta, optional fused ReLU (the reference always applies tf.nn.relu right after).
//
// Layout of the work (gfx950): every pass reads WHOLE rows with float4 accesses -- thread = (channel quad, row lane),
// block = (sample, row chunk) -- instead of one block per (sample, group) walking a [V, C/G] slab of 12..68-byte row
// segments (the first version: 0.8 TB/s, every cache line fetched by up to five different blocks).
//   forward : partial sums per (sample, chunk, channel) of d = x - p_c and d^2 with the pivot p_c = x[n, 0, c]
//             (single pass; the shift keeps sum(d^2) - sum(d)^2/n free of cancellation) -> per (sample, group) mean /
//             rstd in float64 -> per (sample, channel) scale a = rstd*gamma and shift b = beta - mean*a ->
//             y = relu(fma(a, x, b)).
//   backward: the ReLU mask is re-derived as fma(a, x, b) > 0 (bit-identical to the forward decision: same operands, same
//             fma), so y is not read; partial sums of d' and d'*xhat per (sample, chunk, channel) -> per (sample, group)
//             sums -> dx = A_c d' - (B_g + xhat C_g).
// The backward apply pass can add a second gradient of the input (dx_add: the residual branch of the GraphCMR block reads the
// same tensor) instead of leaving it to a separate element-wise launch.  Measured and NOT kept (round 3): turning the chunk
// partials into coefficients inside the partial-sum kernel by the last-arriving block of each sample -- with device-scope
// fences every block writes back its XCD's L2 (step 14.4 -> 22 ms), with write-through stores + L2-bypassing loads the
// finalising block's ~100 dependent loads from memory cost more than the 7 us launch they replace (14.4 -> 16.2 ms).
// Channel counts that are not a multiple of 4 (e.g. 24 + 8 + 6 condition channels): rows are still 16-byte aligned
// (ld % 4 == 0, so ld >= Cp = round_up(C, 4)); every pass works on Cp / 4 quads, the last quad loads its pad lanes (inside
// the row, values unused) and stores / accumulates only the lanes below C.  The per-channel tables (partials, coef, bcoef)
// use the stride Cp so that their float4 accesses stay aligned.
#include "common.h"

namespace {


inline int gn_rows(int N, int V) {
    int rb = 128;
    while (rb > 16 && (long long)N * ((V + rb - 1) / rb) < 1024) rb >>= 1;
    return rb;
}

// store the first min(4, left) lanes of a quad (left = channels from this quad's first to the row's end)
__device__ __forceinline__ void gn_store4(float *p, const float4 &o, int left) {
    if (left >= 4) {
        *reinterpret_cast<float4 *>(p) = o;
    } else {
        p[0] = o.x;
        if (left > 1) p[1] = o.y;
        if (left > 2) p[2] = o.z;
    }
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    return (int)(b < 1 ? 1 : b);
}

// coef layout per sample: [4][C] = a (rstd*gamma), b (beta - mean*a), r (rstd of the channel's group), mr (mean*rstd)
// MODE 0 (forward statistics): terms S = sum(x - p), Q = sum((x - p)^2)
// MODE 1 (backward):           terms s1 = sum(d'), s2 = sum(d' * xhat)
template <int MODE>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float *x, long long xs, int ldx, const float *dy, long long dys,
                                                         int lddy, const float *coef, int relu, int V, int C, int RB,
                                                         int chunks, float *part) {
    __shared__ float4 red[256];
    const int n = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int ra = ch * RB, rb = min(V, ra + RB);
    const float *xb = x + (long long)n * xs;
    const int Cp = (C + 3) & ~3;
    float *pp = part + ((long long)n * chunks + ch) * 2 * Cp;
    for (int cbase = 0; cbase < Cp; cbase += 256) {
        const int cw = min(256, Cp - cbase);
        const int c4n = cw >> 2;
        const int lanes = 256 / c4n;
        const int q = threadIdx.x % c4n, rl = threadIdx.x / c4n;
        const int c = cbase + 4 * q;
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
        if (rl < lanes) {
            if (MODE == 0) {
                const float4 p = *reinterpret_cast<const float4 *>(xb + c);           // pivot: row 0 of this sample
                for (int r = ra + rl; r < rb; r += lanes) {
                    const float4 v = *reinterpret_cast<const float4 *>(xb + (long long)r * ldx + c);
                    const float dx_ = v.x - p.x, dy_ = v.y - p.y, dz_ = v.z - p.z, dw_ = v.w - p.w;
                    t0.x += dx_; t0.y += dy_; t0.z += dz_; t0.w += dw_;
                    t1.x = fmaf(dx_, dx_, t1.x); t1.y = fmaf(dy_, dy_, t1.y); t1.z = fmaf(dz_, dz_, t1.z); t1.w = fmaf(dw_, dw_, t1.w);
                }
            } else {
                const float *cf = coef + (long long)n * 4 * Cp + c;
                const float4 a = *reinterpret_cast<const float4 *>(cf), b = *reinterpret_cast<const float4 *>(cf + Cp);
                const float4 rr = *reinterpret_cast<const float4 *>(cf + 2 * Cp), mr = *reinterpret_cast<const float4 *>(cf + 3 * Cp);
                const float *gb = dy + (long long)n * dys;
                for (int r = ra + rl; r < rb; r += lanes) {
                    const float4 v = *reinterpret_cast<const float4 *>(xb + (long long)r * ldx + c);
                    float4 d = *reinterpret_cast<const float4 *>(gb + (long long)r * lddy + c);
                    if (relu) {
                        d.x = fmaf(a.x, v.x, b.x) > 0.f ? d.x : 0.f; d.y = fmaf(a.y, v.y, b.y) > 0.f ? d.y : 0.f;
                        d.z = fmaf(a.z, v.z, b.z) > 0.f ? d.z : 0.f; d.w = fmaf(a.w, v.w, b.w) > 0.f ? d.w : 0.f;
                    }
                    t0.x += d.x; t0.y += d.y; t0.z += d.z; t0.w += d.w;
                    t1.x = fmaf(d.x, fmaf(v.x, rr.x, -mr.x), t1.x); t1.y = fmaf(d.y, fmaf(v.y, rr.y, -mr.y), t1.y);
                    t1.z = fmaf(d.z, fmaf(v.z, rr.z, -mr.z), t1.z); t1.w = fmaf(d.w, fmaf(v.w, rr.w, -mr.w), t1.w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __syncthreads();
            red[threadIdx.x] = j ? t1 : t0;
            __syncthreads();
            if (rl == 0) {
                float4 t = j ? t1 : t0;
                for (int l = 1; l < lanes; ++l) {                       // fixed order: deterministic
                    const float4 v = red[l * c4n + q];
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
                *reinterpret_cast<float4 *>(pp + (long long)j * Cp + c) = t;      // (pad lanes of the last quad: never read)
            }
        }
        __syncthreads();
    }
}

// block = one (sample, group); thread = (channel of the group, chunk lane).  Sums the chunk partials of every channel
// in float64, combines the channels of the group, writes stats (mean, rstd) and the per-channel coefficients.
__global__ __launch_bounds__(256) void gn_final_kernel(const float *part, int chunks, const float *x, long long xs,
                                                       const float *gamma, const float *beta, float eps, int G, int V, int C,
                                                       float *stats, float *coef) {
    __shared__ double sS[256], sQ[256];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;
    const int Cp = (C + 3) & ~3;
    int cp = 1;
    while (cp < Cg) cp <<= 1;                       // channel lanes (power of two >= group width, <= 256)
    const int CL = 256 / cp;                        // chunk lanes
    const int cc = threadIdx.x % cp, cl = threadIdx.x / cp;
    double S = 0.0, Q = 0.0;
    if (cc < Cg) {
        const int c = g * Cg + cc;
        for (int k = cl; k < chunks; k += CL) {
            const float *pp = part + ((long long)n * chunks + k) * 2 * Cp;
            S += (double)pp[c];
            Q += (double)pp[Cp + c];
        }
    }
    sS[threadIdx.x] = S; sQ[threadIdx.x] = Q;
    __syncthreads();
    if (cl == 0 && cc < Cg) {
        for (int l = 1; l < CL; ++l) { S += sS[l * cp + cc]; Q += sQ[l * cp + cc]; }
        sS[cc] = S; sQ[cc] = Q;                     // per-channel totals of d = x - p_c and d^2
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double T = (double)V * (double)Cg;
        double sum = 0.0;
        for (int j = 0; j < Cg; ++j) sum += sS[j] + (double)V * (double)x[(long long)n * xs + g * Cg + j];
        const double mean = sum / T;
        double m2 = 0.0;
        for (int j = 0; j < Cg; ++j) {
            const double dp = (double)x[(long long)n * xs + g * Cg + j] - mean;      // p_c - mean
            m2 += sQ[j] + 2.0 * dp * sS[j] + (double)V * dp * dp;
        }
        const double var = m2 / T > 0.0 ? m2 / T : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        stats[2 * blockIdx.x] = (float)mean;
        stats[2 * blockIdx.x + 1] = rstd;
        sS[0] = mean; sQ[0] = (double)rstd;
    }
    __syncthreads();
    if (threadIdx.x < Cg) {
        const int c = g * Cg + threadIdx.x;
        const float mean = (float)sS[0], rstd = (float)sQ[0];
        float *cf = coef + (long long)n * 4 * Cp;
        const float a = rstd * gamma[c];
        cf[c] = a;
        cf[Cp + c] = fmaf(-mean, a, beta[c]);
        cf[2 * Cp + c] = rstd;
        cf[3 * Cp + c] = mean * rstd;
    }
}

// max |o| over the valid lanes of a quad, then over the c4 consecutive lanes that hold the row; its first lane stores it
__device__ __forceinline__ float gn_quad_bound(const float4 &o, int left) {
    float m = fabsf(o.x);
    if (left > 1) m = fmaxf(m, fabsf(o.y));
    if (left > 2) m = fmaxf(m, fabsf(o.z));
    if (left > 3) m = fmaxf(m, fabsf(o.w));
    return m;
}
__device__ __forceinline__ void gn_row_bound(float *rm, long long row, const float4 &o, int left, int c4, bool first) {
    const float m = cape_group_max(gn_quad_bound(o, left), c4);
    if (first) cape_store_rowmax(rm, row, m);
}

// rm (may be null): row bounds of the output, [N*V][4] = (max_c |y|, 0, 0, 0), for the fp16 two-piece contractions that read y
// next (csrc/gemm_h2.h); passed only when a row's quads are a power-of-two lane group of one wave (gn_rm_fused)
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *x, long long xs, int ldx, const float *coef, int relu,
                                                       float *y, long long ys, int ldy, int N, int V, int C, float *rm) {
    const int Cp = (C + 3) & ~3, c4 = Cp >> 2;
    const long long total = (long long)N * V * c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        const long long nv = i / c4;
        const int v = (int)(nv % V);
        const int n = (int)(nv / V);
        const float *cf = coef + (long long)n * 4 * Cp + c;
        const float4 a = *reinterpret_cast<const float4 *>(cf), b = *reinterpret_cast<const float4 *>(cf + Cp);
        const float4 xv = *reinterpret_cast<const float4 *>(x + (long long)n * xs + (long long)v * ldx + c);
        float4 o;
        o.x = fmaf(a.x, xv.x, b.x); o.y = fmaf(a.y, xv.y, b.y); o.z = fmaf(a.z, xv.z, b.z); o.w = fmaf(a.w, xv.w, b.w);
        if (relu) {
            o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
        }
        gn_store4(y + (long long)n * ys + (long long)v * ldy + c, o, C - c);
        if (rm) gn_row_bound(rm, nv, o, C - c, c4, c == 0);
    }
}

// The same pass with WHOLE ROWS per block (256 / c4 rows, thread = one quad as above): for channel counts whose quads are no
// power-of-two lane group (the GraphCMR blocks normalise [features | condition]: 288, 544 channels) the row bound goes through
// an LDS maximum (non-negative floats order like their bit patterns; a NaN's pattern is above every finite one).  c4 <= 256.
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const float *x, long long xs, int ldx, const float *coef, int relu,
                                                            float *y, long long ys, int ldy, int N, int V, int C, float *rm) {
    __shared__ unsigned mx[64];
    const int Cp = (C + 3) & ~3, c4 = Cp >> 2;
    const int RB = 256 / c4, rl = threadIdx.x / c4;
    const long long rows = (long long)N * V;
    for (long long base = (long long)blockIdx.x * RB; base < rows; base += (long long)gridDim.x * RB) {
        if (threadIdx.x < RB) mx[threadIdx.x] = 0u;
        __syncthreads();
        const long long nv = base + rl;
        if (rl < RB && nv < rows) {
            const int v = (int)(nv % V), n = (int)(nv / V);
            {
            const int c = 4 * (threadIdx.x % c4);
            const float *cf = coef + (long long)n * 4 * Cp + c;
            const float4 a = *reinterpret_cast<const float4 *>(cf), b = *reinterpret_cast<const float4 *>(cf + Cp);
            const float4 xv = *reinterpret_cast<const float4 *>(x + (long long)n * xs + (long long)v * ldx + c);
            float4 o;
            o.x = fmaf(a.x, xv.x, b.x); o.y = fmaf(a.y, xv.y, b.y); o.z = fmaf(a.z, xv.z, b.z); o.w = fmaf(a.w, xv.w, b.w);
            if (relu) {
                o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
            }
            gn_store4(y + (long long)n * ys + (long long)v * ldy + c, o, C - c);
            atomicMax(&mx[rl], __float_as_uint(gn_quad_bound(o, C - c)));
            }
        }
        __syncthreads();
        if (threadIdx.x < RB && base + threadIdx.x < rows) cape_store_rowmax(rm, base + threadIdx.x, __uint_as_float(mx[threadIdx.x]));
    }
}

// block = one (sample, group): per-channel totals of s1 = sum d', s2 = sum d'*xhat -> per-sample gamma / beta gradient
// partials, the group sums S1 = sum_c gamma_c s1_c, S2 = sum_c gamma_c s2_c, and the coefficients of the apply pass:
// bcoef[n][3][C] = A_c = rstd*gamma_c, B_g = rstd*S1/T, C_g = rstd*S2/T  (dx = A d' - (B + xhat C)).
__global__ __launch_bounds__(256) void gn_bwd_final_kernel(const float *part, int chunks, const float *gamma, const float *stats,
                                                           int G, int V, int C, float *dgamma_p, float *dbeta_p, float *bcoef) {
    __shared__ double s1s[256], s2s[256];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;
    const int Cp = (C + 3) & ~3;
    int cp = 1;
    while (cp < Cg) cp <<= 1;
    const int CL = 256 / cp;
    const int cc = threadIdx.x % cp, cl = threadIdx.x / cp;
    double a1 = 0.0, a2 = 0.0;
    if (cc < Cg) {
        const int c = g * Cg + cc;
        for (int k = cl; k < chunks; k += CL) {
            const float *pp = part + ((long long)n * chunks + k) * 2 * Cp;
            a1 += (double)pp[c];
            a2 += (double)pp[Cp + c];
        }
    }
    s1s[threadIdx.x] = a1; s2s[threadIdx.x] = a2;
    __syncthreads();
    if (cl == 0 && cc < Cg) {
        for (int l = 1; l < CL; ++l) { a1 += s1s[l * cp + cc]; a2 += s2s[l * cp + cc]; }
        const int c = g * Cg + cc;
        dbeta_p[(long long)n * C + c] = (float)a1;
        dgamma_p[(long long)n * C + c] = (float)a2;
        s1s[cc] = a1 * (double)gamma[c]; s2s[cc] = a2 * (double)gamma[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S1 = 0.0, S2 = 0.0;
        for (int j = 0; j < Cg; ++j) { S1 += s1s[j]; S2 += s2s[j]; }
        const double T = (double)V * (double)Cg;
        const double rstd = (double)stats[2 * blockIdx.x + 1];
        s1s[0] = rstd * S1 / T; s2s[0] = rstd * S2 / T;
    }
    __syncthreads();
    if (threadIdx.x < Cg) {
        const int c = g * Cg + threadIdx.x;
        float *bc = bcoef + (long long)n * 3 * Cp;
        bc[c] = stats[2 * blockIdx.x + 1] * gamma[c];
        bc[Cp + c] = (float)s1s[0];
        bc[2 * Cp + c] = (float)s2s[0];
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float *x, long long xs, int ldx, const float *dy, long long dys,
                                                           int lddy, const float *coef, const float *bcoef, int relu, float *dx,
                                                           long long dxs, int lddx, const float *add, long long adds, int ldadd,
                                                           int N, int V, int C, float *rm) {
    const int Cp = (C + 3) & ~3, c4 = Cp >> 2;
    const long long total = (long long)N * V * c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        const long long nv = i / c4;
        const int v = (int)(nv % V);
        const int n = (int)(nv / V);
        const float *cf = coef + (long long)n * 4 * Cp + c;
        const float *bc = bcoef + (long long)n * 3 * Cp + c;
        const float4 a = *reinterpret_cast<const float4 *>(cf), b = *reinterpret_cast<const float4 *>(cf + Cp);
        const float4 rr = *reinterpret_cast<const float4 *>(cf + 2 * Cp), mr = *reinterpret_cast<const float4 *>(cf + 3 * Cp);
        const float4 A = *reinterpret_cast<const float4 *>(bc), B = *reinterpret_cast<const float4 *>(bc + Cp);
        const float4 Cc = *reinterpret_cast<const float4 *>(bc + 2 * Cp);
        const float4 xv = *reinterpret_cast<const float4 *>(x + (long long)n * xs + (long long)v * ldx + c);
        float4 d = *reinterpret_cast<const float4 *>(dy + (long long)n * dys + (long long)v * lddy + c);
        if (relu) {
            d.x = fmaf(a.x, xv.x, b.x) > 0.f ? d.x : 0.f; d.y = fmaf(a.y, xv.y, b.y) > 0.f ? d.y : 0.f;
            d.z = fmaf(a.z, xv.z, b.z) > 0.f ? d.z : 0.f; d.w = fmaf(a.w, xv.w, b.w) > 0.f ? d.w : 0.f;
        }
        float4 o;
        o.x = A.x * d.x - fmaf(fmaf(xv.x, rr.x, -mr.x), Cc.x, B.x);
        o.y = A.y * d.y - fmaf(fmaf(xv.y, rr.y, -mr.y), Cc.y, B.y);
        o.z = A.z * d.z - fmaf(fmaf(xv.z, rr.z, -mr.z), Cc.z, B.z);
        o.w = A.w * d.w - fmaf(fmaf(xv.w, rr.w, -mr.w), Cc.w, B.w);
        if (add) {                            // second gradient of the same input (residual branch), summed here
            const float4 e = *reinterpret_cast<const float4 *>(add + (long long)n * adds + (long long)v * ldadd + c);
            o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
        }
        gn_store4(dx + (long long)n * dxs + (long long)v * lddx + c, o, C - c);
        if (rm) gn_row_bound(rm, nv, o, C - c, c4, c == 0);
    }
}

// whole rows per block, as gn_apply_rows_kernel
__global__ __launch_bounds__(256) void gn_bwd_apply_rows_kernel(const float *x, long long xs, int ldx, const float *dy, long long dys,
                                                                int lddy, const float *coef, const float *bcoef, int relu, float *dx,
                                                                long long dxs, int lddx, const float *add, long long adds, int ldadd,
                                                                int N, int V, int C, float *rm) {
    __shared__ unsigned mx[64];
    const int Cp = (C + 3) & ~3, c4 = Cp >> 2;
    const int RB = 256 / c4, rl = threadIdx.x / c4;
    const long long rows = (long long)N * V;
    for (long long base = (long long)blockIdx.x * RB; base < rows; base += (long long)gridDim.x * RB) {
        if (threadIdx.x < RB) mx[threadIdx.x] = 0u;
        __syncthreads();
        const long long nv = base + rl;
        if (rl < RB && nv < rows) {
            const int v = (int)(nv % V), n = (int)(nv / V);
            {
            const int c = 4 * (threadIdx.x % c4);
            const float *cf = coef + (long long)n * 4 * Cp + c;
            const float *bc = bcoef + (long long)n * 3 * Cp + c;
            const float4 a = *reinterpret_cast<const float4 *>(cf), b = *reinterpret_cast<const float4 *>(cf + Cp);
            const float4 rr = *reinterpret_cast<const float4 *>(cf + 2 * Cp), mr = *reinterpret_cast<const float4 *>(cf + 3 * Cp);
            const float4 A = *reinterpret_cast<const float4 *>(bc), B = *reinterpret_cast<const float4 *>(bc + Cp);
            const float4 Cc = *reinterpret_cast<const float4 *>(bc + 2 * Cp);
            const float4 xv = *reinterpret_cast<const float4 *>(x + (long long)n * xs + (long long)v * ldx + c);
            float4 d = *reinterpret_cast<const float4 *>(dy + (long long)n * dys + (long long)v * lddy + c);
            if (relu) {
                d.x = fmaf(a.x, xv.x, b.x) > 0.f ? d.x : 0.f; d.y = fmaf(a.y, xv.y, b.y) > 0.f ? d.y : 0.f;
                d.z = fmaf(a.z, xv.z, b.z) > 0.f ? d.z : 0.f; d.w = fmaf(a.w, xv.w, b.w) > 0.f ? d.w : 0.f;
            }
            float4 o;
            o.x = A.x * d.x - fmaf(fmaf(xv.x, rr.x, -mr.x), Cc.x, B.x);
            o.y = A.y * d.y - fmaf(fmaf(xv.y, rr.y, -mr.y), Cc.y, B.y);
            o.z = A.z * d.z - fmaf(fmaf(xv.z, rr.z, -mr.z), Cc.z, B.z);
            o.w = A.w * d.w - fmaf(fmaf(xv.w, rr.w, -mr.w), Cc.w, B.w);
            if (add) {
                const float4 e = *reinterpret_cast<const float4 *>(add + (long long)n * adds + (long long)v * ldadd + c);
                o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
            }
            gn_store4(dx + (long long)n * dxs + (long long)v * lddx + c, o, C - c);
            atomicMax(&mx[rl], __float_as_uint(gn_quad_bound(o, C - c)));
            }
        }
        __syncthreads();
        if (threadIdx.x < RB && base + threadIdx.x < rows) cape_store_rowmax(rm, base + threadIdx.x, __uint_as_float(mx[threadIdx.x]));
    }
}

// The sums over the batch of the per-sample gamma / beta gradient partials of SEVERAL group-norm calls in one launch (the
// results are only needed at the end of the backward pass: the training step queues them, like the weight-gradient slab
// reductions).  thread = channel, samples added in index order.
struct GnParamBatch {
    cape_gn_param_item_t it[CAPE_MAX_GN_REDUCE_ITEMS];
    int blk_off[CAPE_MAX_GN_REDUCE_ITEMS + 1];
    int n;
};

__global__ __launch_bounds__(256) void gn_param_reduce_batch_kernel(GnParamBatch B) {
    int i = 0;
    while (i + 1 < B.n && (int)blockIdx.x >= B.blk_off[i + 1]) ++i;
    const cape_gn_param_item_t &it = B.it[i];
    const int c = ((int)blockIdx.x - B.blk_off[i]) * 256 + (int)threadIdx.x;
    if (c >= it.C) return;
    float sg = 0.f, sb = 0.f;
    for (int n = 0; n < it.N; ++n) {
        sg += it.dgamma_partial[(long long)n * it.C + c];
        sb += it.dbeta_partial[(long long)n * it.C + c];
    }
    it.dgamma[c] = sg;
    it.dbeta[c] = sb;
}

inline bool gn_aligned(const void *p, long long ss, int ld) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ss & 3) == 0) && ((ld & 3) == 0);
}

// the last column tile of a partial-sum pass must have a quad count that leaves at least one row lane
inline bool gn_shape_ok(int C, int G) {
    return C / G <= 256;
}

inline int gn_rows_grid(int N, int V, int C) {
    const int RB = 256 / ((C + 3) >> 2);
    return grid_for(((long long)N * V + RB - 1) / RB * 256);
}

// the apply kernels can bound whole rows when a row's float4 lanes are a power-of-two group inside one wave
inline bool gn_rm_fused(int C) {
    const int c4 = (C + 3) >> 2;
    return c4 <= 64 && (c4 & (c4 - 1)) == 0;
}

}  // namespace

extern "C" int64_t cape_groupnorm_workspace_bytes(int32_t N, int32_t V, int32_t C) {
    if (N < 1 || V < 1 || C < 1) return CAPE_EINVAL;
    const int RB = gn_rows(N, V);
    const long long chunks = (V + RB - 1) / RB;
    return (int64_t)N * chunks * 2 * ((C + 3) & ~3) * (int64_t)sizeof(float);
}

extern "C" int cape_groupnorm_fwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *gamma,
                                  const float *beta, float eps, int32_t G, int32_t relu, float *y, int64_t y_sample_stride,
                                  int32_t ldy, float *stats, float *coef, int32_t N, int32_t V, int32_t C, void *workspace,
                                  int64_t workspace_bytes, float *rowmax_out, void *stream) {
    if (!x || !gamma || !beta || !y || !stats || !coef || !workspace || N < 1 || V < 1 || C < 1 || G < 1 || (C % G) != 0 ||
        ldx < C || ldy < C)
        return CAPE_EINVAL;
    if (!gn_shape_ok(C, G) || !gn_aligned(x, x_sample_stride, ldx) || !gn_aligned(y, y_sample_stride, ldy)) return CAPE_EINVAL;
    if (workspace_bytes < cape_groupnorm_workspace_bytes(N, V, C)) return CAPE_EWORKSPACE;
    const int RB = gn_rows(N, V);
    const int chunks = (V + RB - 1) / RB;
    hipStream_t st = (hipStream_t)stream;
    CAPE_LAUNCH(gn_partial_kernel<0>, dim3(N * chunks), dim3(256), 0, st, x, (long long)x_sample_stride, ldx, (const float *)nullptr,
                0LL, 0, (const float *)nullptr, 0, V, C, RB, chunks, (float *)workspace);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(gn_final_kernel, dim3(N * G), dim3(256), 0, st, (const float *)workspace, chunks, x, (long long)x_sample_stride,
                gamma, beta, eps, G, V, C, stats, coef);
    CAPE_LAUNCH_CHECK();
    const bool rows_form = rowmax_out && !gn_rm_fused(C) && C <= 1024;
    if (rows_form)
        CAPE_LAUNCH(gn_apply_rows_kernel, dim3(gn_rows_grid(N, V, C)), dim3(256), 0, st, x, (long long)x_sample_stride, ldx,
                    (const float *)coef, relu, y, (long long)y_sample_stride, ldy, N, V, C, rowmax_out);
    else
        CAPE_LAUNCH(gn_apply_kernel, dim3(grid_for((long long)N * V * ((C + 3) / 4))), dim3(256), 0, st, x, (long long)x_sample_stride, ldx,
                    (const float *)coef, relu, y, (long long)y_sample_stride, ldy, N, V, C, gn_rm_fused(C) ? rowmax_out : (float *)nullptr);
    CAPE_LAUNCH_CHECK();
    if (rowmax_out && !rows_form && !gn_rm_fused(C)) return cape_rowmax(y, y_sample_stride, ldy, N, V, C, rowmax_out, 4, stream);
    return CAPE_OK;
}

extern "C" int cape_groupnorm_bwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *dy,
                                  int64_t dy_sample_stride, int32_t lddy, const float *gamma, const float *stats,
                                  const float *coef, int32_t G, int32_t relu, float *dx, int64_t dx_sample_stride, int32_t lddx,
                                  const float *dx_add, int64_t add_sample_stride, int32_t ldadd,
                                  float *dgamma_partial, float *dbeta_partial, float *bcoef, int32_t N, int32_t V, int32_t C,
                                  void *workspace, int64_t workspace_bytes, float *rowmax_out, void *stream) {
    if (!x || !dy || !gamma || !stats || !coef || !dx || !dgamma_partial || !dbeta_partial || !bcoef || !workspace || N < 1 ||
        V < 1 || C < 1 || G < 1 || (C % G) != 0 || ldx < C || lddy < C || lddx < C || (dx_add && ldadd < C))
        return CAPE_EINVAL;
    if (!gn_shape_ok(C, G) || !gn_aligned(x, x_sample_stride, ldx) || !gn_aligned(dy, dy_sample_stride, lddy) ||
        !gn_aligned(dx, dx_sample_stride, lddx) || (dx_add && !gn_aligned(dx_add, add_sample_stride, ldadd)))
        return CAPE_EINVAL;
    if (workspace_bytes < cape_groupnorm_workspace_bytes(N, V, C)) return CAPE_EWORKSPACE;
    const int RB = gn_rows(N, V);
    const int chunks = (V + RB - 1) / RB;
    hipStream_t st = (hipStream_t)stream;
    CAPE_LAUNCH(gn_partial_kernel<1>, dim3(N * chunks), dim3(256), 0, st, x, (long long)x_sample_stride, ldx, dy,
                (long long)dy_sample_stride, lddy, coef, relu, V, C, RB, chunks, (float *)workspace);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(gn_bwd_final_kernel, dim3(N * G), dim3(256), 0, st, (const float *)workspace, chunks, gamma, stats, G, V, C,
                dgamma_partial, dbeta_partial, bcoef);
    CAPE_LAUNCH_CHECK();
    const bool rows_form = rowmax_out && !gn_rm_fused(C) && C <= 1024;
    if (rows_form)
        CAPE_LAUNCH(gn_bwd_apply_rows_kernel, dim3(gn_rows_grid(N, V, C)), dim3(256), 0, st, x, (long long)x_sample_stride, ldx, dy,
                    (long long)dy_sample_stride, lddy, coef, (const float *)bcoef, relu, dx, (long long)dx_sample_stride, lddx, dx_add,
                    (long long)add_sample_stride, ldadd, N, V, C, rowmax_out);
    else
        CAPE_LAUNCH(gn_bwd_apply_kernel, dim3(grid_for((long long)N * V * ((C + 3) / 4))), dim3(256), 0, st, x, (long long)x_sample_stride,
                    ldx, dy, (long long)dy_sample_stride, lddy, coef, (const float *)bcoef, relu, dx, (long long)dx_sample_stride, lddx,
                    dx_add, (long long)add_sample_stride, ldadd, N, V, C, gn_rm_fused(C) ? rowmax_out : (float *)nullptr);
    CAPE_LAUNCH_CHECK();
    if (rowmax_out && !rows_form && !gn_rm_fused(C)) return cape_rowmax(dx, dx_sample_stride, lddx, N, V, C, rowmax_out, 4, stream);
    return CAPE_OK;
}

extern "C" int cape_groupnorm_param_reduce_batch(const cape_gn_param_item_t *items, int32_t nitems, void *stream) {
    if (!items || nitems < 1 || nitems > CAPE_MAX_GN_REDUCE_ITEMS) return CAPE_EINVAL;
    GnParamBatch B;
    B.n = nitems;
    int off = 0;
    for (int i = 0; i < nitems; ++i) {
        const cape_gn_param_item_t &t = items[i];
        if (!t.dgamma_partial || !t.dbeta_partial || !t.dgamma || !t.dbeta || t.N < 1 || t.C < 1) return CAPE_EINVAL;
        B.it[i] = t;
        B.blk_off[i] = off;
        off += (t.C + 255) / 256;
    }
    B.blk_off[nitems] = off;
    CAPE_LAUNCH(gn_param_reduce_batch_kernel, dim3((unsigned)off), dim3(256), 0, (hipStream_t)stream, B);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
