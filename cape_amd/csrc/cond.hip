// Rank-1 condition coefficients of ALL consumers of one condition vector in one launch.
//
// The decoder concatenates the same tiled condition vector [N, Cc] to the input of every block
// (reference lib/models.py:591-594, 606-609 fit_cond_dim + tf.concat).  Those channels are vertex-constant,
// so their contribution to layer l is  sum_j rowscale_j[r] * coef_l[n, j, f]  with
//     coef_l[n, k, f] = sum_c cond[n, c] * W_l[(Ch_l + c) * K_l + k, f]      k < K_l
//     coef_l[n, K_l, f] = sum_c cond[n, c] * Waff_l[Ch_l + c, f]             (affine blocks only)
// (cape_rank_t).  Per layer these are 16 x 64 x F products: far too small for a launch each (a dispatch
// costs ~5 us on MI355X, 9 layers x (2 forward + 5 backward) of them were ~0.4 ms of a 5 ms step), so the
// forward of all layers is one kernel here, and so are the weight-row gradients and the condition gradient.
#include "common.h"

namespace {

struct CondLayers {
    cape_cond_layer_t l[CAPE_MAX_COND_LAYERS];
    int nlayers;
    int blk_off[CAPE_MAX_COND_LAYERS + 1];   // first block of each layer: (K + has_aff) * ceil(F / 64) blocks per layer
};

constexpr int FT = 64;      // output columns per block; the 4 waves of a block split the condition channels

__device__ __forceinline__ void locate(const CondLayers &L, int b, int &li, int &r, int &f0) {
    li = 0;
    while (li + 1 < L.nlayers && b >= L.blk_off[li + 1]) ++li;
    const int lb = b - L.blk_off[li];
    const int fblocks = (L.l[li].F + FT - 1) / FT;
    r = lb / fblocks;
    f0 = (lb % fblocks) * FT;
}

// coef[n, r, f] = sum_c cond[n, c] * Wrow(c, r)[f].  Thread (fl, cg): column f0 + fl, channels c = cg, cg+4, ...;
// 16 samples at a time in registers; the four channel groups are summed through LDS in a fixed order.
__global__ __launch_bounds__(256) void cond_coef_fwd_kernel(CondLayers L, const float *cond, int ldc, int N, int Cc) {
    extern __shared__ float smem[];
    float *scond = smem;                       // [N][Cc]
    float *red = smem + N * Cc;                // [4][16][FT]
    for (int i = threadIdx.x; i < N * Cc; i += 256) scond[i] = cond[(long long)(i / Cc) * ldc + (i % Cc)];
    __syncthreads();
    int li, r, f0;
    locate(L, blockIdx.x, li, r, f0);
    const cape_cond_layer_t &Y = L.l[li];
    const int fl = threadIdx.x & (FT - 1), cg = threadIdx.x >> 6;
    const int f = f0 + fl;
    const bool fok = f < Y.F;
    const int R = Y.K + (Y.w_aff ? 1 : 0);
    const float *w = (r < Y.K) ? Y.w + (long long)r * Y.F + (fok ? f : 0) : Y.w_aff + (fok ? f : 0);
    const long long wstep = (r < Y.K) ? (long long)Y.K * Y.F : Y.F;
    for (int n0 = 0; n0 < N; n0 += 16) {
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int c = cg; c < Cc; c += 4) {
            const float wv = w[c * wstep];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fmaf(scond[min(n0 + i, N - 1) * Cc + c], wv, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(cg * 16 + i) * FT + fl] = acc[i];
        __syncthreads();
        // thread (fl, cg) finishes samples n0 + 4*cg .. +3
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = cg * 4 + j;
            const float t = ((red[(0 * 16 + i) * FT + fl] + red[(1 * 16 + i) * FT + fl]) + red[(2 * 16 + i) * FT + fl]) + red[(3 * 16 + i) * FT + fl];
            if (fok && n0 + i < N) Y.coef[((long long)(n0 + i) * R + r) * Y.F + f] = t;
        }
        __syncthreads();
    }
}

// gw[(c*K + k), f] = sum_n cond[n, c] * dcoef[n, k, f]   (and the affine rows).  Thread (fl, cg): column f0 + fl,
// channels c = cg, cg+4, ...; the column's dcoef values stay in registers.
__global__ __launch_bounds__(256) void cond_coef_dw_kernel(CondLayers L, const float *cond, int ldc, int N, int Cc) {
    extern __shared__ float smem[];
    float *scond = smem;
    for (int i = threadIdx.x; i < N * Cc; i += 256) scond[i] = cond[(long long)(i / Cc) * ldc + (i % Cc)];
    __syncthreads();
    int li, r, f0;
    locate(L, blockIdx.x, li, r, f0);
    const cape_cond_layer_t &Y = L.l[li];
    const int fl = threadIdx.x & (FT - 1), cg = threadIdx.x >> 6;
    const int f = f0 + fl;
    if (f >= Y.F) return;
    const int R = Y.K + (Y.w_aff ? 1 : 0);
    float *gw = (r < Y.K) ? Y.gw : Y.gw_aff;
    if (!gw) return;
    gw += (r < Y.K) ? (long long)r * Y.F + f : f;
    const long long wstep = (r < Y.K) ? (long long)Y.K * Y.F : Y.F;
    const float *d = Y.dcoef + (long long)r * Y.F + f;
    const long long dstep = (long long)R * Y.F;
    for (int n0 = 0; n0 < N; n0 += 16) {
        float dv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) dv[i] = (n0 + i < N) ? d[(n0 + i) * dstep] : 0.f;
        for (int c = cg; c < Cc; c += 4) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s = fmaf(scond[min(n0 + i, N - 1) * Cc + c], dv[i], s);
            if (n0 == 0) gw[c * wstep] = s;
            else gw[c * wstep] += s;
        }
    }
}

// dcond[n, c] = sum_l sum_r sum_f dcoef_l[n, r, f] * Wrow_l(c, r)[f]: block = (n, c); the (layer, r) pairs are dealt round robin to
// the four waves, each lane strides the f range in float4 steps with four independent accumulators.  The pairs are a chain
// of dependent accesses (layer descriptor -> row pointers -> data: ~0.8 us each, 24 of them in the benchmarked model); one
// wave per c walked the whole chain (19-21 us for 1.5 MB of weights), four waves a quarter of it.
__global__ __launch_bounds__(256) void cond_coef_dcond_kernel(CondLayers L, float *dcond, int ldd, int N, int Cc, int accumulate) {
    __shared__ float red[4];
    const int n = blockIdx.x / Cc, c = blockIdx.x % Cc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int pair = 0;
    for (int li = 0; li < L.nlayers; ++li) {
        const cape_cond_layer_t &Y = L.l[li];
        const int R = Y.K + (Y.w_aff ? 1 : 0);
        const bool v4 = (Y.F & 3) == 0 && ((reinterpret_cast<uintptr_t>(Y.w) | reinterpret_cast<uintptr_t>(Y.dcoef) |
                                             reinterpret_cast<uintptr_t>(Y.w_aff)) & 15) == 0;
        for (int r = 0; r < R; ++r, ++pair) {
            if ((pair & 3) != wave) continue;
            const float *w = (r < Y.K) ? Y.w + ((long long)c * Y.K + r) * Y.F : Y.w_aff + (long long)c * Y.F;
            const float *d = Y.dcoef + ((long long)n * R + r) * Y.F;
            if (v4) {
                const int F4 = Y.F >> 2;
                const float4 *w4 = reinterpret_cast<const float4 *>(w), *d4 = reinterpret_cast<const float4 *>(d);
#pragma unroll 2
                for (int q = lane; q < F4; q += 64) {
                    const float4 a = d4[q], b = w4[q];
                    s0 = fmaf(a.x, b.x, s0); s1 = fmaf(a.y, b.y, s1); s2 = fmaf(a.z, b.z, s2); s3 = fmaf(a.w, b.w, s3);
                }
            } else {
                for (int f = lane; f < Y.F; f += 64) s0 = fmaf(d[f], w[f], s0);
            }
        }
    }
    float s = (s0 + s1) + (s2 + s3);
    // fixed-order reduction: the lanes of a wave, then the four waves
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        float *dst = dcond + (long long)n * ldd + c;
        *dst = accumulate ? (*dst + t) : t;
    }
}

inline int fill_layers(CondLayers &L, const cape_cond_layer_t *layers, int nlayers, bool bwd) {
    if (!layers || nlayers < 1 || nlayers > CAPE_MAX_COND_LAYERS) return CAPE_EINVAL;
    int off = 0;
    for (int i = 0; i < nlayers; ++i) {
        const cape_cond_layer_t &y = layers[i];
        if (!y.w || y.K < 1 || y.F < 1) return CAPE_EINVAL;
        if (!bwd && !y.coef) return CAPE_EINVAL;
        if (bwd && !y.dcoef) return CAPE_EINVAL;
        L.l[i] = y;
        L.blk_off[i] = off;
        off += (y.K + (y.w_aff ? 1 : 0)) * ((y.F + FT - 1) / FT);
    }
    L.blk_off[nlayers] = off;
    L.nlayers = nlayers;
    return CAPE_OK;
}

}  // namespace

extern "C" int cape_cond_coef_fwd(const float *cond, int32_t ldc, int32_t N, int32_t Cc,
                                  const cape_cond_layer_t *layers, int32_t nlayers, void *stream) {
    if (!cond || N < 1 || Cc < 1 || ldc < Cc || (long long)N * Cc * 4 > 40 * 1024) return CAPE_EINVAL;
    CondLayers L;
    int rc = fill_layers(L, layers, nlayers, false);
    if (rc) return rc;
    CAPE_LAUNCH(cond_coef_fwd_kernel, dim3(L.blk_off[nlayers]), dim3(256), (size_t)(N * Cc + 4 * 16 * FT) * 4, (hipStream_t)stream, L, cond,
                ldc, N, Cc);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_cond_coef_bwd(const float *cond, int32_t ldc, int32_t N, int32_t Cc,
                                  const cape_cond_layer_t *layers, int32_t nlayers, float *dcond, int32_t ldd,
                                  int32_t accumulate, void *stream) {
    if (!cond || N < 1 || Cc < 1 || ldc < Cc || (long long)N * Cc * 4 > 40 * 1024) return CAPE_EINVAL;
    if (dcond && ldd < Cc) return CAPE_EINVAL;
    CondLayers L;
    int rc = fill_layers(L, layers, nlayers, true);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    bool any_gw = false;
    for (int i = 0; i < nlayers; ++i) any_gw = any_gw || layers[i].gw || layers[i].gw_aff;
    if (any_gw) {
        CAPE_LAUNCH(cond_coef_dw_kernel, dim3(L.blk_off[nlayers]), dim3(256), (size_t)N * Cc * 4, st, L, cond, ldc, N, Cc);
        CAPE_LAUNCH_CHECK();
    }
    if (dcond) {
        CAPE_LAUNCH(cond_coef_dcond_kernel, dim3(N * Cc), dim3(256), 0, st, L, dcond, ldd, N, Cc, accumulate);
        CAPE_LAUNCH_CHECK();
    }
    return CAPE_OK;
}
