// L1 reconstruction loss + SMPL edge loss and their gradients w.r.t. the prediction
// (reference lib/models.py:357-360,374-375 and lib/losses.py:9-25):
//   recon = mean |pred - gt|
//   edge  = mean_e || ((pred+ref)_i - (pred+ref)_j) - ((gt+ref)_i - (gt+ref)_j) ||_2
// The gradient is gathered per vertex from a vertex->incident-edge table (no float atomics;
// deterministic).  d||d||/dd at d == 0 is defined as 0 (TF's tf.norm gradient is NaN there).
#include "common.h"

namespace {

constexpr int LB = 256;

__device__ __forceinline__ float block_sum256(float v, float *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// per (n, e): unit difference vector -> unit[n,e,0:3], block-partial sum of lengths
__global__ __launch_bounds__(LB) void edge_fwd_kernel(const float *pred, const float *gt, const float *ref, const int *edges,
                                                      int N, int M, int E, int ldp, float *unit, float *part) {
    __shared__ float red[4];
    const long long total = (long long)N * E;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * LB + threadIdx.x; i < total; i += (long long)gridDim.x * LB) {
        const int e = (int)(i % E);
        const long long n = i / E;
        const int a = edges[2 * e], b = edges[2 * e + 1];
        float d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float pa = pred[(n * M + a) * ldp + k] + ref[a * 3 + k];
            const float pb = pred[(n * M + b) * ldp + k] + ref[b * 3 + k];
            const float ga = gt[(n * M + a) * 3 + k] + ref[a * 3 + k];
            const float gb = gt[(n * M + b) * 3 + k] + ref[b * 3 + k];
            d[k] = (pa - pb) - (ga - gb);
        }
        const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        s += len;
        const float inv = len > 0.f ? 1.f / len : 0.f;
        if (unit) {
            unit[i * 3 + 0] = d[0] * inv;
            unit[i * 3 + 1] = d[1] * inv;
            unit[i * 3 + 2] = d[2] * inv;
        }
    }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// per (n, v): L1 partial sums and the combined gradient
__global__ __launch_bounds__(LB) void vert_kernel(const float *pred, const float *gt, const float *unit, const int *vptr,
                                                  const int *vidx, int N, int M, int E, int ldp, int ldd, float cr, float ce,
                                                  float *dpred, float *part) {
    __shared__ float red[4];
    const long long total = (long long)N * M;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * LB + threadIdx.x; i < total; i += (long long)gridDim.x * LB) {
        const int v = (int)(i % M);
        const long long n = i / M;
        float g[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = pred[i * ldp + k] - gt[i * 3 + k];
            s += fabsf(d);
            g[k] = cr * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (dpred) {
            for (int t = vptr[v]; t < vptr[v + 1]; ++t) {
                const int code = vidx[t];
                const int e = code >> 1;
                const float sg = (code & 1) ? -ce : ce;
                const float *u = unit + (n * E + e) * 3;
                g[0] = fmaf(sg, u[0], g[0]);
                g[1] = fmaf(sg, u[1], g[1]);
                g[2] = fmaf(sg, u[2], g[2]);
            }
            dpred[i * ldd + 0] = g[0];
            dpred[i * ldd + 1] = g[1];
            dpred[i * ldd + 2] = g[2];
        }
    }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(LB) void loss_final_kernel(const float *part_e, int ne, float inv_e, const float *part_v, int nv,
                                                        float inv_v, float *loss_out, float w_recon, float w_edge, float *total_out,
                                                        const float *term_a, float w_a, const float *term_b) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nv; i += LB) s += part_v[i];
    s = block_sum256(s, red);
    float t = 0.f;
    for (int i = threadIdx.x; i < ne; i += LB) t += part_e[i];
    t = block_sum256(t, red);
    if (threadIdx.x == 0) {
        loss_out[0] = s * inv_v;
        loss_out[1] = t * inv_e;
        if (total_out) {
            // (+ w_a * term_a + term_b: the latent term and the regulariser value of the training loss, lib/models.py:393-394 --
            // device scalars computed earlier in the step; folding them in here saves the element-wise launches of the sum)
            float tot = w_recon * (s * inv_v) + w_edge * (t * inv_e);
            if (term_a) tot = fmaf(w_a, *term_a, tot);
            if (term_b) tot += *term_b;
            *total_out = tot;
        }
    }
}

// ---- adversarial losses on the discriminator's logits (lib/models.py:381-390) -----------------------------------------------
//   gan_g = mean_i bce(fake_i, 1 - smooth)            gan_d = mean_j bce(real_j, 1 - smooth) + mean_i bce(fake_i, smooth)
//   bce(x, t) = max(x, 0) - x t + log1p(exp(-|x|))   (tf.nn.sigmoid_cross_entropy_with_logits),  d bce / dx = sigmoid(x) - t
// One workgroup (the logits are N x 431 values): both means by a fixed-order block reduction, and the gradients of
// scale * gan_g and scale * gan_d w.r.t. every logit in the same pass -- the op-by-op form is ~25 launches of 5 us each.
// Logit (n, m) of a tensor is at p[n * ss + m * ld].  ga / gb: gradients of the two losses, rows = the fake samples first,
// then the real ones (ga is zero there), contiguous [Nf + Nr, M].
constexpr int GAN_LB = 1024;
__global__ __launch_bounds__(GAN_LB) void gan_bce_kernel(const float *fake, long long fss, int fld, const float *real, long long rss, int rld,
                                                         int Nf, int Nr, int M, float smooth, float scale, float *out, float *scaled_g,
                                                         float *scaled_d, float *ga, float *gb) {
    __shared__ float red[3][GAN_LB / 64];
    const float tr = 1.f - smooth, tf_ = smooth;
    const float cf = scale / ((float)Nf * (float)M), cr = scale / ((float)Nr * (float)M);
    float sg = 0.f, sdf = 0.f, sdr = 0.f;
    for (int i = threadIdx.x; i < Nf * M; i += GAN_LB) {
        const float x = fake[(long long)(i / M) * fss + (long long)(i % M) * fld];
        const float e = expf(-fabsf(x));                                  // exp(-|x|) in (0, 1]
        const float sp = fmaxf(x, 0.f) + log1pf(e);                         // softplus(x)
        const float sig = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);       // sigmoid(x) from the same exponential
        sg += sp - x * tr;
        sdf += sp - x * tf_;
        ga[i] = cf * (sig - tr);
        gb[i] = cf * (sig - tf_);
    }
    for (int i = threadIdx.x; i < Nr * M; i += GAN_LB) {
        const float x = real[(long long)(i / M) * rss + (long long)(i % M) * rld];
        const float e = expf(-fabsf(x));
        const float sp = fmaxf(x, 0.f) + log1pf(e);
        const float sig = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        sdr += sp - x * tr;
        ga[Nf * M + i] = 0.f;
        gb[Nf * M + i] = cr * (sig - tr);
    }
    // fixed-order reduction: xor shuffles inside each wave, then the 16 wave sums in order
    float v[3] = {sg, sdf, sdr};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        for (int sh = 1; sh < 64; sh <<= 1) v[j] += __shfl_xor(v[j], sh);
        if ((threadIdx.x & 63) == 0) red[j][threadIdx.x >> 6] = v[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < 3; ++j)
            for (int w = 0; w < GAN_LB / 64; ++w) t[j] += red[j][w];
        const float g = t[0] / ((float)Nf * (float)M), d = t[2] / ((float)Nr * (float)M) + t[1] / ((float)Nf * (float)M);
        out[0] = g;
        out[1] = d;
        *scaled_g = scale * g;
        *scaled_d = scale * d;
    }
}

inline int nblocks(long long total) {
    long long b = (total + LB - 1) / LB;
    if (b > 1024) b = 1024;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int64_t cape_recon_edge_workspace_bytes(int32_t N, int32_t M, int32_t E) {
    if (N < 1 || M < 1 || E < 1) return CAPE_EINVAL;
    return ((int64_t)N * E * 3 + 2048) * (int64_t)sizeof(float);
}

extern "C" int cape_recon_edge_loss_fwd_bwd(const float *pred, int32_t ldp, const float *gt, const float *verts_ref, const int32_t *edges,
                                            const int32_t *vert_edge_ptr, const int32_t *vert_edge_idx, int32_t N, int32_t M,
                                            int32_t E, float w_recon, float w_edge, float *loss_out, float *total_out,
                                            const float *term_a, float w_a, const float *term_b, float *dpred, int32_t ldd,
                                            void *workspace, int64_t workspace_bytes, void *stream) {
    if (!pred || !gt || !verts_ref || !edges || !loss_out || !workspace || N < 1 || M < 1 || E < 1 || ldp < 3) return CAPE_EINVAL;
    if (dpred && ldd < 3) return CAPE_EINVAL;
    if (dpred && (!vert_edge_ptr || !vert_edge_idx)) return CAPE_EINVAL;
    if (workspace_bytes < cape_recon_edge_workspace_bytes(N, M, E)) return CAPE_EWORKSPACE;
    float *ws = (float *)workspace;
    float *part_e = ws, *part_v = ws + 1024, *unit = ws + 2048;
    hipStream_t st = (hipStream_t)stream;
    const int ne = nblocks((long long)N * E), nv = nblocks((long long)N * M);
    CAPE_LAUNCH(edge_fwd_kernel, dim3(ne), dim3(LB), 0, st, pred, gt, verts_ref, edges, N, M, E, ldp, unit, part_e);
    CAPE_LAUNCH_CHECK();
    const float cr = w_recon / ((float)N * (float)M * 3.0f);
    const float ce = w_edge / ((float)N * (float)E);
    CAPE_LAUNCH(vert_kernel, dim3(nv), dim3(LB), 0, st, pred, gt, unit, vert_edge_ptr, vert_edge_idx, N, M, E, ldp, ldd, cr, ce, dpred, part_v);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(loss_final_kernel, dim3(1), dim3(LB), 0, st, part_e, ne, 1.0f / ((float)N * (float)E), part_v, nv,
                       1.0f / ((float)N * (float)M * 3.0f), loss_out, w_recon, w_edge, total_out, term_a, w_a, term_b);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_gan_bce_fwd_bwd(const float *fake, int64_t fake_sample_stride, int32_t ldf, const float *real,
                                    int64_t real_sample_stride, int32_t ldr, int32_t Nf, int32_t Nr, int32_t M, float smooth,
                                    float scale, float *loss_out, float *scaled_g, float *scaled_d, float *grad_g, float *grad_d,
                                    void *stream) {
    if (!fake || !real || !loss_out || !scaled_g || !scaled_d || !grad_g || !grad_d || Nf < 1 || Nr < 1 || M < 1 || ldf < 1 || ldr < 1) return CAPE_EINVAL;
    if ((long long)(Nf + Nr) * M >= (1LL << 31)) return CAPE_EINVAL;
    CAPE_LAUNCH(gan_bce_kernel, dim3(1), dim3(GAN_LB), 0, (hipStream_t)stream, fake, (long long)fake_sample_stride, ldf, real,
                (long long)real_sample_stride, ldr, Nf, Nr, M, smooth, scale, loss_out, scaled_g, scaled_d, grad_g, grad_d);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
