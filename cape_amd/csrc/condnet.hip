// The two condition networks of CAPE in one launch per direction (reference lib/models.py:479-511 ``condition``, called at
// :284-290): pose MLP  c1 [N, in1] -> leaky_relu(c1 W1 + b1) [N, hid] -> (.) W2 + b2 [N, out1]   (tf.layers.dense x 2) and the
// clothing-type layer  c2 [N, in2] -> c2 Wc + bc [N, out2]; the outputs are written side by side as ycat [N, out1 + out2]
// -- the concatenated condition every consumer wants (:533, :591, :663).  At N = 16 these are ~0.15 MFLOP: as six
// rocBLAS / elementwise dispatches forward and eight backward they cost ~70 us of a 3.5 ms step in dispatch latency alone.
// Forward: one workgroup per sample; backward: 16 workgroups over the rows of the widest gradient; fixed summation order.
#include "common.h"

namespace {

struct CondNetP {
    const float *c1, *c2;
    int ld1, ld2;
    const float *W1, *b1, *W2, *b2, *Wc, *bc;
    float *h;            // [N, hid] post-activation hidden layer (saved for backward)
    float *ycat;         // [N, out1 + out2]
    float *ycat2;        // NULL or a second copy of ycat (a consumer of its own: its gradient then arrives separately as dycat2)
    const float *dycat;  // backward: gradient w.r.t. ycat, rows lddy floats apart (NULL: zero)
    const float *dycat2; // backward: NULL or the gradient w.r.t. ycat2, rows lddy2 apart; the kernel reads dycat + dycat2
    int lddy, lddy2;
    float *gW1, *gb1, *gW2, *gb2, *gWc, *gbc;
    int N, in1, hid, out1, in2, out2;
};

// Forward: one workgroup per sample.  Thread = (output column, quarter of the contraction): four partial sums per
// output, combined in a fixed order through LDS -- 32 instead of 126 dependent multiply-adds per thread, weight reads
// coalesced over the columns.
__device__ __forceinline__ float quad_sum(float *red, int col, int part, float v, int ncol) {
    __syncthreads();
    red[part * ncol + col] = v;
    __syncthreads();
    return (red[col] + red[ncol + col]) + (red[2 * ncol + col] + red[3 * ncol + col]);
}

__global__ __launch_bounds__(256) void condnet_fwd_kernel(CondNetP p) {
    __shared__ float s1[512], sh[256], red[4 * 64];
    const int n = blockIdx.x;
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < p.in1; i += 256) s1[i] = p.c1[(long long)n * p.ld1 + i];
    __syncthreads();
    for (int j0 = 0; j0 < p.hid; j0 += 64) {               // hidden layer, 64 columns at a time
        const int j = j0 + col;
        float a = 0.f;
        if (j < p.hid) {
            // (the weight loads are independent: eight in flight -- a rolled loop waits ~0.5 us for every one of its ~32)
#pragma unroll 8
            for (int i = part; i < p.in1; i += 4) a = fmaf(s1[i], p.W1[(long long)i * p.hid + j], a);
        }
        a = quad_sum(red, col, part, a, 64);
        if (part == 0 && j < p.hid) {
            a += p.b1[j];
            a = a > 0.f ? a : 0.2f * a;
            sh[j] = a;
            p.h[(long long)n * p.hid + j] = a;
        }
    }
    __syncthreads();
    const int oc = p.out1 + p.out2;
    for (int f0 = 0; f0 < oc; f0 += 64) {
        const int f = f0 + col;
        float a = 0.f;
        if (f < p.out1) {
#pragma unroll 8
            for (int j = part; j < p.hid; j += 4) a = fmaf(sh[j], p.W2[(long long)j * p.out1 + f], a);
        } else if (f < oc) {
            const int g = f - p.out1;
#pragma unroll 4
            for (int i = part; i < p.in2; i += 4) a = fmaf(p.c2[(long long)n * p.ld2 + i], p.Wc[(long long)i * p.out2 + g], a);
        }
        a = quad_sum(red, col, part, a, 64);
        if (part == 0 && f < oc) {
            const float v = a + (f < p.out1 ? p.b2[f] : p.bc[f - p.out1]);
            p.ycat[(long long)n * oc + f] = v;
            if (p.ycat2) p.ycat2[(long long)n * oc + f] = v;
        }
    }
}

// Backward: every workgroup recomputes dh = (dy W2^T) * leaky'(h) for all samples (N * hid * out1 multiply-adds: cheap) and
// then owns a slice of the rows of gW1 (+ its bias row); workgroup 0 also produces the small gradients of the other layers.
__global__ __launch_bounds__(256) void condnet_bwd_kernel(CondNetP p) {
    extern __shared__ float sm[];
    const int oc = p.out1 + p.out2;
    float *sh = sm;                          // [N][hid]   h, then dh in place
    float *sd = sh + p.N * p.hid;            // [N][oc]    dycat
    for (int i = threadIdx.x; i < p.N * p.hid; i += 256) sh[i] = p.h[i];
    for (int i = threadIdx.x; i < p.N * oc; i += 256) {
        const int n = i / oc, f = i % oc;
        float v = p.dycat ? p.dycat[(long long)n * p.lddy + f] : 0.f;
        if (p.dycat2) v += p.dycat2[(long long)n * p.lddy2 + f];
        sd[i] = v;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int o = threadIdx.x; o < (p.hid + 1) * p.out1; o += 256) {
            const int j = o / p.out1, f = o % p.out1;          // j == hid: bias row
            float a = 0.f;
            for (int n = 0; n < p.N; ++n) a = fmaf(j < p.hid ? sh[n * p.hid + j] : 1.f, sd[n * oc + f], a);
            if (j < p.hid) p.gW2[o] = a;
            else p.gb2[f] = a;
        }
        for (int o = threadIdx.x; o < (p.in2 + 1) * p.out2; o += 256) {
            const int i = o / p.out2, g = o % p.out2;
            float a = 0.f;
#pragma unroll 8
            for (int n = 0; n < p.N; ++n) a = fmaf(i < p.in2 ? p.c2[(long long)n * p.ld2 + i] : 1.f, sd[n * oc + p.out1 + g], a);
            if (i < p.in2) p.gWc[o] = a;
            else p.gbc[g] = a;
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < p.N * p.hid; o += 256) {
        const int n = o / p.hid, j = o % p.hid;
        float a = 0.f;
#pragma unroll 8
        for (int f = 0; f < p.out1; ++f) a = fmaf(sd[n * oc + f], p.W2[(long long)j * p.out1 + f], a);
        const float hv = sh[o];                            // own element only: the in-place update needs no barrier
        sh[o] = hv > 0.f ? a : 0.2f * a;
    }
    __syncthreads();
    // rows [i0, i1) of gW1 (row in1 = bias): thread = one output element, sum over the samples in order
    const int rows = p.in1 + 1;
    const int per = (rows + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(rows, i0 + per);
    for (int o = threadIdx.x; o < (i1 - i0) * p.hid; o += 256) {
        const int i = i0 + o / p.hid, j = o % p.hid;
        float a = 0.f;
#pragma unroll 8
        for (int n = 0; n < p.N; ++n) a = fmaf(i < p.in1 ? p.c1[(long long)n * p.ld1 + i] : 1.f, sh[n * p.hid + j], a);
        if (i < p.in1) p.gW1[(long long)i * p.hid + j] = a;
        else p.gb1[j] = a;
    }
}

inline size_t condnet_lds(const CondNetP &p, bool bwd) {
    return bwd ? sizeof(float) * ((size_t)p.N * p.hid + (size_t)p.N * (p.out1 + p.out2)) : 0;
}

inline int condnet_check(const CondNetP &p) {
    if (!p.c1 || !p.c2 || !p.W1 || !p.b1 || !p.W2 || !p.b2 || !p.Wc || !p.bc || !p.h) return CAPE_EINVAL;
    if (p.N < 1 || p.N > 64 || p.in1 < 1 || p.hid < 1 || p.out1 < 1 || p.in2 < 1 || p.out2 < 1 || p.ld1 < p.in1 || p.ld2 < p.in2)
        return CAPE_EINVAL;
    if (condnet_lds(p, true) > 60 * 1024 || p.in1 > 512 || p.hid > 256) return CAPE_EINVAL;     // static LDS rows of the forward kernel
    return CAPE_OK;
}

}  // namespace

extern "C" int cape_condnet_fwd(const float *c1, int32_t ld1, const float *c2, int32_t ld2, const float *W1, const float *b1,
                                const float *W2, const float *b2, const float *Wc, const float *bc, float *h, float *ycat,
                                float *ycat2, int32_t N, int32_t in1, int32_t hid, int32_t out1, int32_t in2, int32_t out2, void *stream) {
    CondNetP p{};
    p.c1 = c1; p.c2 = c2; p.ld1 = ld1; p.ld2 = ld2; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.Wc = Wc; p.bc = bc;
    p.h = h; p.ycat = ycat; p.ycat2 = ycat2; p.N = N; p.in1 = in1; p.hid = hid; p.out1 = out1; p.in2 = in2; p.out2 = out2;
    if (!ycat) return CAPE_EINVAL;
    const int rc = condnet_check(p);
    if (rc) return rc;
    CAPE_LAUNCH(condnet_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, p);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_condnet_bwd(const float *c1, int32_t ld1, const float *c2, int32_t ld2, const float *W2, const float *h,
                                const float *dycat, int32_t lddy, const float *dycat2, int32_t lddy2, float *gW1, float *gb1,
                                float *gW2, float *gb2, float *gWc, float *gbc,
                                int32_t N, int32_t in1, int32_t hid, int32_t out1, int32_t in2, int32_t out2, void *stream) {
    CondNetP p{};
    p.c1 = c1; p.c2 = c2; p.ld1 = ld1; p.ld2 = ld2; p.W2 = W2; p.h = const_cast<float *>(h);
    p.dycat = dycat; p.dycat2 = dycat2; p.lddy = lddy; p.lddy2 = lddy2;
    p.gW1 = gW1; p.gb1 = gb1; p.gW2 = gW2; p.gb2 = gb2; p.gWc = gWc; p.gbc = gbc;
    p.N = N; p.in1 = in1; p.hid = hid; p.out1 = out1; p.in2 = in2; p.out2 = out2;
    p.W1 = p.b1 = p.b2 = p.Wc = p.bc = W2;      // (unused by the backward kernel; non-null for the shared check)
    if ((!dycat && !dycat2) || (dycat && lddy < out1 + out2) || (dycat2 && lddy2 < out1 + out2)) return CAPE_EINVAL;
    if (!gW1 || !gb1 || !gW2 || !gb2 || !gWc || !gbc) return CAPE_EINVAL;
    const int rc = condnet_check(p);
    if (rc) return rc;
    CAPE_LAUNCH(condnet_bwd_kernel, dim3(16), dim3(256), condnet_lds(p, true), (hipStream_t)stream, p);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
