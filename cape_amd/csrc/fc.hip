// The dense layers with one very long side (reference tf.layers.dense call sites lib/models.py:557, :560 --
// encoder fc_mean / fc_var, 862*64 = 55 168 -> nz -- and :582 -- decoder fc1, nz+cond -> 55 168): 87 % of the
// model's parameters, a batch of 16 rows.  They are weight-streaming problems (28 MB of fp32 weights per
// layer, ~0.1 GFLOP): every kernel here reads or writes each weight exactly once with coalesced accesses and
// keeps the 16 activations rows in LDS; reductions are two-stage with a fixed summation order.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int FC_MAXN = 64;        // batch rows
constexpr int FC_RS = 128;         // long-input forward: contraction rows per split
constexpr int FC_MAXMAT = 2;

struct LongArgs {
    const float *W[FC_MAXMAT];
    const float *b[FC_MAXMAT];
    const float *g[FC_MAXMAT];
    float *y[FC_MAXMAT];
    float *dW[FC_MAXMAT];
    float *db[FC_MAXMAT];
    int nmat;
};

// ---- long input, forward: partial[m][split][n][j] = sum_{i in split} x[n,i] W_m[i,j] ------------------------
// block (split, m): 64 output columns x 4 sample groups; the split's x tile [N][128] sits in LDS
__global__ __launch_bounds__(256) void fc_long_partial_kernel(LongArgs A, const float *x, int ldx, int N, int in, int out, int nsplit,
                                                             float *part) {
    extern __shared__ float xs[];      // [N][FC_RS]
    const int split = blockIdx.x, m = blockIdx.y;
    const int i0 = split * FC_RS;
    const int rows = min(FC_RS, in - i0);
    for (int t = threadIdx.x; t < N * FC_RS; t += 256) {
        const int n = t / FC_RS, il = t % FC_RS;
        xs[t] = il < rows ? x[(long long)n * ldx + i0 + il] : 0.f;
    }
    __syncthreads();
    const int jl = threadIdx.x & 63, ng = threadIdx.x >> 6;
    const float *W = A.W[m];
    for (int j0 = 0; j0 < out; j0 += 64) {
        const int j = j0 + jl;
        const bool jok = j < out;
        const float *wp = W + (long long)i0 * out + (jok ? j : 0);
        float acc[FC_MAXN / 4];
#pragma unroll
        for (int k = 0; k < FC_MAXN / 4; ++k) acc[k] = 0.f;
#pragma unroll 4
        for (int il = 0; il < rows; ++il) {
            const float w = wp[(long long)il * out];
#pragma unroll
            for (int k = 0; k < FC_MAXN / 4; ++k)
                if (ng + 4 * k < N) acc[k] = fmaf(xs[(ng + 4 * k) * FC_RS + il], w, acc[k]);
        }
        if (jok) {
#pragma unroll
            for (int k = 0; k < FC_MAXN / 4; ++k)
                if (ng + 4 * k < N) part[(((long long)m * nsplit + split) * N + (ng + 4 * k)) * out + j] = acc[k];
        }
    }
}

// y_m[n,j] = b_m[j] + sum_split partial: block = 16 outputs x 16 lanes, four independent partial sums per lane
__global__ __launch_bounds__(256) void fc_long_final_kernel(LongArgs A, const float *part, int N, int out, int nsplit) {
    __shared__ float red[16][17];
    const int ol = threadIdx.x & 15, ln = threadIdx.x >> 4;
    const long long per = (long long)N * out;
    const long long o = (long long)blockIdx.x * 16 + ol;       // index into [m][n][j]
    float s = 0.f;
    if (o < per * A.nmat) {
        const int m = (int)(o / per);
        const long long nj = o % per;
        const float *p = part + (long long)m * nsplit * per + nj;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sp = ln;
        for (; sp + 48 < nsplit; sp += 64) {
            s += p[(long long)sp * per];
            s1 += p[(long long)(sp + 16) * per];
            s2 += p[(long long)(sp + 32) * per];
            s3 += p[(long long)(sp + 48) * per];
        }
        for (; sp < nsplit; sp += 16) s += p[(long long)sp * per];
        s = (s + s1) + (s2 + s3);
    }
    red[ln][ol] = s;
    __syncthreads();
    if (ln == 0 && o < per * A.nmat) {
        const int m = (int)(o / per);
        const long long nj = o % per;
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[l][ol];
        const int j = (int)(nj % out);
        A.y[m][nj] = t + (A.b[m] ? A.b[m][j] : 0.f);
    }
}

// ---- long input, backward: block = 64 contraction rows i ------------------------------------------------------
//   dW_m[i,j] = sum_n x[n,i] g_m[n,j]      dx[n,i] = sum_m sum_j g_m[n,j] W_m[i,j]      db_m[j] = sum_n g_m[n,j] (block 0)
__global__ __launch_bounds__(256) void fc_long_bwd_kernel(LongArgs A, const float *x, int ldx, int N, int in, int out, float *dx, int lddx) {
    extern __shared__ float sm[];
    float *xs = sm;                         // [N][64]
    float *gs = xs + N * 64;                // [nmat][N][out]
    float *Ws = gs + A.nmat * N * out;      // [64][65]  (one 64-column tile of one matrix at a time)
    const int i0 = blockIdx.x * 64;
    const int rows = min(64, in - i0);
    for (int t = threadIdx.x; t < N * 64; t += 256) {
        const int n = t >> 6, il = t & 63;
        xs[t] = il < rows ? x[(long long)n * ldx + i0 + il] : 0.f;
    }
    for (int m = 0; m < A.nmat; ++m)
        for (int t = threadIdx.x; t < N * out; t += 256) gs[m * N * out + t] = A.g[m][t];
    __syncthreads();
    const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;
    // weight gradient rows (coalesced over j)
    for (int m = 0; m < A.nmat; ++m) {
        if (!A.dW[m]) continue;
        const float *g = gs + m * N * out;
        for (int j0 = 0; j0 < out; j0 += 64) {
            const int j = j0 + lo;
            if (j >= out) continue;
            for (int il = hi; il < rows; il += 4) {
                float s = 0.f;
                for (int n = 0; n < N; ++n) s = fmaf(xs[n * 64 + il], g[n * out + j], s);
                A.dW[m][(long long)(i0 + il) * out + j] = s;
            }
        }
        if (blockIdx.x == 0 && A.db[m]) {
            for (int j = threadIdx.x; j < out; j += 256) {
                float s = 0.f;
                for (int n = 0; n < N; ++n) s += g[n * out + j];
                A.db[m][j] = s;
            }
        }
    }
    // data gradient: thread (il = lo, sample group hi)
    if (dx) {
        float acc[FC_MAXN / 4];
#pragma unroll
        for (int k = 0; k < FC_MAXN / 4; ++k) acc[k] = 0.f;
        for (int m = 0; m < A.nmat; ++m) {
            const float *g = gs + m * N * out;
            for (int j0 = 0; j0 < out; j0 += 64) {
                const int jw = min(64, out - j0);
                __syncthreads();
                for (int t = threadIdx.x; t < 64 * 64; t += 256) {
                    const int il = t >> 6, jl = t & 63;
                    Ws[il * 65 + jl] = (il < rows && jl < jw) ? A.W[m][(long long)(i0 + il) * out + j0 + jl] : 0.f;
                }
                __syncthreads();
                for (int jl = 0; jl < jw; ++jl) {
                    const float w = Ws[lo * 65 + jl];
#pragma unroll
                    for (int k = 0; k < FC_MAXN / 4; ++k)
                        if (hi + 4 * k < N) acc[k] = fmaf(g[(hi + 4 * k) * out + j0 + jl], w, acc[k]);
                }
            }
        }
        if (lo < rows) {
#pragma unroll
            for (int k = 0; k < FC_MAXN / 4; ++k)
                if (hi + 4 * k < N) dx[(long long)(hi + 4 * k) * lddx + i0 + lo] = acc[k];
        }
    }
}

// ---- wide output, forward: y[n,j] = act(b[j] + sum_i x[n,i] W[i,j]); thread = output column, x [N][in] in LDS ------
__global__ __launch_bounds__(256) void fc_wide_fwd_kernel(const float *x, int ldx, int N, int in, int out, const float *W, const float *b,
                                                         int act, float *y, int ldy) {
    extern __shared__ float xs[];       // [N][in]
    for (int t = threadIdx.x; t < N * in; t += 256) xs[t] = x[(long long)(t / in) * ldx + (t % in)];
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= out) return;
    const float bj = b ? b[j] : 0.f;
    for (int n0 = 0; n0 < N; n0 += 16) {
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll 4
        for (int i = 0; i < in; ++i) {
            const float w = W[(long long)i * out + j];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = fmaf(xs[min(n0 + k, N - 1) * in + i], w, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (n0 + k < N) y[(long long)(n0 + k) * ldy + j] = cape_act(acc[k] + bj, act);
    }
}

// ---- wide output, backward (weights): dz = g * act'(y);  dW[i,j] = sum_n x[n,i] dz[n,j];  db[j] = sum_n dz[n,j] ----
__global__ __launch_bounds__(256) void fc_wide_bwd_dw_kernel(const float *x, int ldx, const float *g, int ldg, const float *y, int ldy, int act,
                                                            int N, int in, int out, float *dW, float *db) {
    extern __shared__ float xs[];       // [N][in]
    for (int t = threadIdx.x; t < N * in; t += 256) xs[t] = x[(long long)(t / in) * ldx + (t % in)];
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= out) return;
    float bsum = 0.f;
    for (int n0 = 0; n0 < N; n0 += 16) {
        float dz[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int n = n0 + k;
            dz[k] = (n < N) ? g[(long long)n * ldg + j] * (act == CAPE_ACT_NONE ? 1.f : cape_act_grad_from_out(y[(long long)n * ldy + j], act)) : 0.f;
            bsum += dz[k];
        }
        if (dW) {
            for (int i = 0; i < in; ++i) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) s = fmaf(xs[min(n0 + k, N - 1) * in + i], dz[k], s);
                float *dst = dW + (long long)i * out + j;
                *dst = (n0 == 0) ? s : (*dst + s);
            }
        }
    }
    if (db) db[j] = bsum;
}

// ---- wide output, backward (data): partial[chunk][n][i] = sum_{j in chunk of 64} dz[n,j] W[i,j] -------------------
__global__ __launch_bounds__(256) void fc_wide_bwd_dx_partial_kernel(const float *g, int ldg, const float *y, int ldy, int act, int N, int in, int out,
                                                                    const float *W, float *part) {
    extern __shared__ float sm[];
    float *Wt = sm;                   // [in][65]
    float *dzs = sm + in * 65;        // [N][64]
    const int j0 = blockIdx.x * 64;
    const int jw = min(64, out - j0);
    const int jl = threadIdx.x & 63, hi = threadIdx.x >> 6;
    for (int i = hi; i < in; i += 4) Wt[i * 65 + jl] = jl < jw ? W[(long long)i * out + j0 + jl] : 0.f;
    for (int n = hi; n < N; n += 4) {
        float v = 0.f;
        if (jl < jw) {
            v = g[(long long)n * ldg + j0 + jl];
            if (act != CAPE_ACT_NONE) v *= cape_act_grad_from_out(y[(long long)n * ldy + j0 + jl], act);
        }
        dzs[n * 64 + jl] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < N * in; o += 256) {
        const int n = o / in, i = o % in;
        const float *w = Wt + i * 65, *d = dzs + n * 64;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int q = 0; q < 64; q += 2) {
            s0 = fmaf(d[q], w[q], s0);
            s1 = fmaf(d[q + 1], w[q + 1], s1);
        }
        part[(long long)blockIdx.x * N * in + o] = s0 + s1;
    }
}

__global__ __launch_bounds__(256) void fc_wide_bwd_dx_final_kernel(const float *part, int chunks, int N, int in, float *dx, int lddx) {
    __shared__ float red[16][17];
    const int ol = threadIdx.x & 15, ln = threadIdx.x >> 4;
    const long long per = (long long)N * in;
    const long long o = (long long)blockIdx.x * 16 + ol;
    float s = 0.f;
    if (o < per) {
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int c = ln;
        for (; c + 48 < chunks; c += 64) {
            s += part[(long long)c * per + o];
            s1 += part[(long long)(c + 16) * per + o];
            s2 += part[(long long)(c + 32) * per + o];
            s3 += part[(long long)(c + 48) * per + o];
        }
        for (; c < chunks; c += 16) s += part[(long long)c * per + o];
        s = (s + s1) + (s2 + s3);
    }
    red[ln][ol] = s;
    __syncthreads();
    if (ln == 0 && o < per) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[l][ol];
        dx[(o / in) * lddx + (o % in)] = t;
    }
}

// =============================================================================================================
// Register-tiled variants for the aligned shapes of the shipped configurations (N <= 16, out % 4 == 0, 16-byte
// aligned rows): every thread owns a float4 of output columns x 16 samples, so one 16-byte weight load feeds 64
// FMAs and the activations come from LDS as broadcasts.  The kernels above remain the ragged-shape path.
// =============================================================================================================

// long input forward: block = 128 contraction rows; thread (cq = column quad 0..15, rg = row group 0..15)
__global__ __launch_bounds__(256) void fc_long_partial_v4_kernel(LongArgs A, const float *x, int ldx, int N, int in, int out, int nsplit,
                                                                float *part) {
    __shared__ float xs[16 * FC_RS];          // [16][128], rows beyond N / the tail are zero
    __shared__ float red[4 * 16 * 64];        // [wave][n][col]
    const int split = blockIdx.x, m = blockIdx.y;
    const int i0 = split * FC_RS;
    const int rows = min(FC_RS, in - i0);
    for (int t = threadIdx.x; t < 16 * FC_RS; t += 256) {
        const int n = t / FC_RS, il = t % FC_RS;
        xs[t] = (n < N && il < rows) ? x[(long long)n * ldx + i0 + il] : 0.f;
    }
    __syncthreads();
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *W = A.W[m];
    for (int j0 = 0; j0 < out; j0 += 64) {
        const int jc = j0 + 4 * cq;
        const bool jok = jc < out;
        float acc[16][4];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f;
        const float *wp = W + (long long)i0 * out + (jok ? jc : 0);
#pragma unroll 2
        for (int r = 0; r < FC_RS / 16; ++r) {
            const int il = rg + 16 * r;
            const float4 w = *reinterpret_cast<const float4 *>(wp + (long long)min(il, rows - 1) * out);   // x is 0 beyond rows
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float xv = xs[k * FC_RS + il];
                acc[k][0] = fmaf(xv, w.x, acc[k][0]); acc[k][1] = fmaf(xv, w.y, acc[k][1]);
                acc[k][2] = fmaf(xv, w.z, acc[k][2]); acc[k][3] = fmaf(xv, w.w, acc[k][3]);
            }
        }
        // sum the 16 row groups: 4 per wave through shuffles, then the 4 waves through LDS (fixed order)
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[k][c] = cape_sum_xor32(cape_sum_xor16(acc[k][c]));
            }
        __syncthreads();
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                *reinterpret_cast<float4 *>(&red[(wave * 16 + k) * 64 + 4 * cq]) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
        }
        __syncthreads();
        for (int o = threadIdx.x; o < 16 * 64; o += 256) {
            const int n = o >> 6, jl = o & 63;
            const float t = ((red[(0 * 16 + n) * 64 + jl] + red[(1 * 16 + n) * 64 + jl]) + red[(2 * 16 + n) * 64 + jl]) + red[(3 * 16 + n) * 64 + jl];
            if (n < N && j0 + jl < out) part[(((long long)m * nsplit + split) * N + n) * out + j0 + jl] = t;
        }
    }
}

// long input backward: block = 64 contraction rows
__global__ __launch_bounds__(256) void fc_long_bwd_v4_kernel(LongArgs A, const float *x, int ldx, int N, int in, int out, float *dx, int lddx) {
    extern __shared__ float sm[];
    float *xs = sm;                         // [16][64]
    float *gs = xs + 16 * 64;               // [nmat][16][out]   (rows beyond N are zero)
    const int i0 = blockIdx.x * 64;
    const int rows = min(64, in - i0);
    for (int t = threadIdx.x; t < 16 * 64; t += 256) {
        const int n = t >> 6, il = t & 63;
        xs[t] = (n < N && il < rows) ? x[(long long)n * ldx + i0 + il] : 0.f;
    }
    for (int m = 0; m < A.nmat; ++m)
        for (int t = threadIdx.x; t < 16 * out; t += 256) gs[m * 16 * out + t] = (t / out < N) ? A.g[m][t] : 0.f;
    __syncthreads();
    // ---- weight gradient: thread (cq, rg) keeps g[16][4 columns] in registers, rows il = rg, rg + 16, ...
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    for (int m = 0; m < A.nmat; ++m) {
        const float *g = gs + m * 16 * out;
        if (A.dW[m]) {
            for (int j0 = 0; j0 < out; j0 += 64) {
                const int jc = j0 + 4 * cq;
                if (jc < out) {
                    float4 gv[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) gv[k] = *reinterpret_cast<const float4 *>(&g[k * out + jc]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int il = rg + 16 * r;
                        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const float xv = xs[k * 64 + il];
                            s.x = fmaf(xv, gv[k].x, s.x); s.y = fmaf(xv, gv[k].y, s.y); s.z = fmaf(xv, gv[k].z, s.z); s.w = fmaf(xv, gv[k].w, s.w);
                        }
                        if (il < rows) *reinterpret_cast<float4 *>(&A.dW[m][(long long)(i0 + il) * out + jc]) = s;
                    }
                }
            }
        }
        if (blockIdx.x == 0 && A.db[m]) {
            for (int j = threadIdx.x; j < out; j += 256) {
                float s = 0.f;
                for (int n = 0; n < 16; ++n) s += g[n * out + j];
                A.db[m][j] = s;
            }
        }
    }
    // ---- data gradient: thread (il = row, ng = group of 4 samples) streams its own weight row in float4s
    if (dx) {
        const int il = threadIdx.x & 63, ng = threadIdx.x >> 6;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < A.nmat; ++m) {
            const float *g = gs + m * 16 * out;
            const float *wrow = A.W[m] + (long long)(i0 + min(il, rows - 1)) * out;
#pragma unroll 4
            for (int j = 0; j < out; j += 4) {
                const float4 w = *reinterpret_cast<const float4 *>(wrow + j);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 gv = *reinterpret_cast<const float4 *>(&g[(4 * ng + k) * out + j]);
                    acc[k] = fmaf(gv.x, w.x, acc[k]); acc[k] = fmaf(gv.y, w.y, acc[k]);
                    acc[k] = fmaf(gv.z, w.z, acc[k]); acc[k] = fmaf(gv.w, w.w, acc[k]);
                }
            }
        }
        if (il < rows) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * ng + k < N) dx[(long long)(4 * ng + k) * lddx + i0 + il] = acc[k];
        }
    }
}

// wide output forward: block = 128 columns; thread (cq = column quad 0..31, ig = row group 0..7)
__global__ __launch_bounds__(256) void fc_wide_fwd_v4_kernel(const float *x, int ldx, int N, int in, int out, const float *W, const float *b,
                                                            int act, float *y, int ldy) {
    extern __shared__ float sm[];
    float *xs = sm;                    // [16][in]
    float *red = sm + 16 * in;         // [4 waves][16][128]
    for (int t = threadIdx.x; t < 16 * in; t += 256) xs[t] = (t / in < N) ? x[(long long)(t / in) * ldx + (t % in)] : 0.f;
    __syncthreads();
    const int cq = threadIdx.x & 31, ig = threadIdx.x >> 5, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int jc = blockIdx.x * 128 + 4 * cq;
    const bool jok = jc < out;
    float acc[16][4];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f;
    const float *wp = W + (jok ? jc : 0);
#pragma unroll 2
    for (int i = ig; i < in; i += 8) {
        const float4 w = *reinterpret_cast<const float4 *>(wp + (long long)i * out);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float xv = xs[k * in + i];
            acc[k][0] = fmaf(xv, w.x, acc[k][0]); acc[k][1] = fmaf(xv, w.y, acc[k][1]);
            acc[k][2] = fmaf(xv, w.z, acc[k][2]); acc[k][3] = fmaf(xv, w.w, acc[k][3]);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k][c] = cape_sum_xor32(acc[k][c]);
    if (lane < 32) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            *reinterpret_cast<float4 *>(&red[(wave * 16 + k) * 128 + 4 * cq]) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 16 * 128; o += 256) {
        const int n = o >> 7, jl = o & 127;
        const int j = blockIdx.x * 128 + jl;
        if (n < N && j < out) {
            const float t = ((red[(0 * 16 + n) * 128 + jl] + red[(1 * 16 + n) * 128 + jl]) + red[(2 * 16 + n) * 128 + jl]) + red[(3 * 16 + n) * 128 + jl];
            y[(long long)n * ldy + j] = cape_act(t + (b ? b[j] : 0.f), act);
        }
    }
}

// wide output backward, weights: block = 256 columns; thread (cq = column quad 0..63, ig = row group 0..3)
__global__ __launch_bounds__(256) void fc_wide_bwd_dw_v4_kernel(const float *x, int ldx, const float *g, int ldg, const float *y, int ldy, int act,
                                                               int N, int in, int out, float *dW, float *db) {
    extern __shared__ float xt[];       // [in][16]  (sample index fastest: one ds_read_b128 = 4 samples)
    for (int t = threadIdx.x; t < 16 * in; t += 256) {
        const int n = t / in, i = t % in;
        xt[i * 16 + n] = n < N ? x[(long long)n * ldx + i] : 0.f;
    }
    __syncthreads();
    const int cq = threadIdx.x & 63, ig = threadIdx.x >> 6;
    const int jc = blockIdx.x * 256 + 4 * cq;
    if (jc >= out) return;
    float4 dz[16];
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < N) {
            v = *reinterpret_cast<const float4 *>(g + (long long)k * ldg + jc);
            if (act != CAPE_ACT_NONE) {
                const float4 yv = *reinterpret_cast<const float4 *>(y + (long long)k * ldy + jc);
                v.x *= cape_act_grad_from_out(yv.x, act); v.y *= cape_act_grad_from_out(yv.y, act);
                v.z *= cape_act_grad_from_out(yv.z, act); v.w *= cape_act_grad_from_out(yv.w, act);
            }
        }
        dz[k] = v;
        bs.x += v.x; bs.y += v.y; bs.z += v.z; bs.w += v.w;
    }
    if (db && ig == 0 && blockIdx.y == 0) *reinterpret_cast<float4 *>(db + jc) = bs;
    if (!dW) return;
    // the weight rows are shared out over gridDim.y blocks (216 column blocks alone leave CUs idle)
    const int per = (in + gridDim.y - 1) / gridDim.y;
    const int ia = blockIdx.y * per, ib = min(in, ia + per);
    for (int i = ia + ig; i < ib; i += 4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 xv = *reinterpret_cast<const float4 *>(&xt[i * 16 + 4 * q]);
            const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 d = dz[4 * q + u];
                s.x = fmaf(xa[u], d.x, s.x); s.y = fmaf(xa[u], d.y, s.y); s.z = fmaf(xa[u], d.z, s.z); s.w = fmaf(xa[u], d.w, s.w);
            }
        }
        *reinterpret_cast<float4 *>(dW + (long long)i * out + jc) = s;
    }
}

// wide output backward, data: block = 64 columns; thread (iq = quad of inputs, ng = pair of samples)
__global__ __launch_bounds__(256) void fc_wide_bwd_dx_partial_v4_kernel(const float *g, int ldg, const float *y, int ldy, int act, int N, int in,
                                                                       int out, const float *W, float *part) {
    extern __shared__ float sm[];
    const int ldw = in + 4;
    float *Wt = sm;                    // [64][in + 4]   (column-of-the-chunk major: one ds_read_b128 = 4 inputs)
    float *dzs = sm + 64 * ldw;        // [16][64]
    const int j0 = blockIdx.x * 64;
    const int jw = min(64, out - j0);
    const int jl = threadIdx.x & 63, hi = threadIdx.x >> 6;
    for (int i0 = hi; i0 < in; i0 += 32) {          // 8 weight rows per thread in flight
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = (jl < jw && i0 + 4 * u < in) ? W[(long long)(i0 + 4 * u) * out + j0 + jl] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 4 * u < in) Wt[jl * ldw + i0 + 4 * u] = wv[u];
    }
    for (int n = hi; n < 16; n += 4) {
        float v = 0.f;
        if (n < N && jl < jw) {
            v = g[(long long)n * ldg + j0 + jl];
            if (act != CAPE_ACT_NONE) v *= cape_act_grad_from_out(y[(long long)n * ldy + j0 + jl], act);
        }
        dzs[n * 64 + jl] = v;
    }
    __syncthreads();
    const int nquads = in >> 2;                 // in % 4 == 0
    for (int o = threadIdx.x; o < 8 * nquads; o += 256) {
        const int iq = o % nquads, ng = o / nquads;      // samples 2*ng, 2*ng + 1
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const float4 w = *reinterpret_cast<const float4 *>(&Wt[j * ldw + 4 * iq]);
            const float d0 = dzs[(2 * ng) * 64 + j], d1 = dzs[(2 * ng + 1) * 64 + j];
            a0.x = fmaf(d0, w.x, a0.x); a0.y = fmaf(d0, w.y, a0.y); a0.z = fmaf(d0, w.z, a0.z); a0.w = fmaf(d0, w.w, a0.w);
            a1.x = fmaf(d1, w.x, a1.x); a1.y = fmaf(d1, w.y, a1.y); a1.z = fmaf(d1, w.z, a1.z); a1.w = fmaf(d1, w.w, a1.w);
        }
        float *dst = part + (long long)blockIdx.x * N * in;
        if (2 * ng < N) *reinterpret_cast<float4 *>(dst + (long long)(2 * ng) * in + 4 * iq) = a0;
        if (2 * ng + 1 < N) *reinterpret_cast<float4 *>(dst + (long long)(2 * ng + 1) * in + 4 * iq) = a1;
    }
}

#include "fc_mfma.h"

inline bool fc_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int long_args(LongArgs &A, int nmat, const float *const *W, const float *const *b, const float *const *g, float *const *y,
                     float *const *dW, float *const *db) {
    if (nmat < 1 || nmat > FC_MAXMAT || !W) return CAPE_EINVAL;
    A.nmat = nmat;
    for (int m = 0; m < FC_MAXMAT; ++m) {
        const bool on = m < nmat;
        A.W[m] = on ? W[m] : nullptr;
        A.b[m] = (on && b) ? b[m] : nullptr;
        A.g[m] = (on && g) ? g[m] : nullptr;
        A.y[m] = (on && y) ? y[m] : nullptr;
        A.dW[m] = (on && dW) ? dW[m] : nullptr;
        A.db[m] = (on && db) ? db[m] : nullptr;
        if (on && !A.W[m]) return CAPE_EINVAL;
    }
    return CAPE_OK;
}

}  // namespace

extern "C" int64_t cape_fc_long_workspace_bytes(int32_t N, int32_t in, int32_t out, int32_t nmat) {
    if (N < 1 || N > FC_MAXN || in < 1 || out < 1 || nmat < 1 || nmat > FC_MAXMAT) return CAPE_EINVAL;
    const long long nsplit = (in + FC_RS - 1) / FC_RS;
    return (int64_t)nmat * nsplit * N * out * (int64_t)sizeof(float);
}

extern "C" int cape_fc_long_fwd(const float *x, int32_t ldx, int32_t N, int32_t in, int32_t out, int32_t nmat,
                                const float *const *W, const float *const *b, float *const *y, void *workspace,
                                int64_t workspace_bytes, void *stream) {
    if (!x || !y || !workspace || N < 1 || N > FC_MAXN || in < 1 || out < 1 || ldx < in) return CAPE_EINVAL;
    LongArgs A;
    int rc = long_args(A, nmat, W, b, nullptr, y, nullptr, nullptr);
    if (rc) return rc;
    for (int m = 0; m < nmat; ++m)
        if (!A.y[m]) return CAPE_EINVAL;
    if (workspace_bytes < cape_fc_long_workspace_bytes(N, in, out, nmat)) return CAPE_EWORKSPACE;
    const int nsplit = (in + FC_RS - 1) / FC_RS;
    hipStream_t st = (hipStream_t)stream;
    bool v4 = N <= 16 && (out & 3) == 0;
    for (int m = 0; m < nmat; ++m) v4 = v4 && fc_al16(A.W[m]);
    const bool m16 = fc_m16_on && v4 && (out == 64 || out == 128) && (in & 3) == 0 && (ldx & 3) == 0 && fc_al16(x) && fc_al16(workspace);
    if (m16 && out == 64)
        CAPE_LAUNCH((fc_long_partial_m16_kernel<1>), dim3(nsplit, nmat), dim3(256), 0, st, A, x, ldx, N, in, nsplit, (float *)workspace);
    else if (m16)
        CAPE_LAUNCH((fc_long_partial_m16_kernel<2>), dim3(nsplit, nmat), dim3(256), 0, st, A, x, ldx, N, in, nsplit, (float *)workspace);
    else if (v4)
        CAPE_LAUNCH(fc_long_partial_v4_kernel, dim3(nsplit, nmat), dim3(256), 0, st, A, x, ldx, N, in, out, nsplit, (float *)workspace);
    else
        CAPE_LAUNCH(fc_long_partial_kernel, dim3(nsplit, nmat), dim3(256), (size_t)N * FC_RS * 4, st, A, x, ldx, N, in, out, nsplit, (float *)workspace);
    CAPE_LAUNCH_CHECK();
    const long long outs = (long long)nmat * N * out;
    CAPE_LAUNCH(fc_long_final_kernel, dim3((unsigned)((outs + 15) / 16)), dim3(256), 0, st, A, (const float *)workspace, N, out, nsplit);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_fc_long_bwd(const float *x, int32_t ldx, int32_t N, int32_t in, int32_t out, int32_t nmat,
                                const float *const *W, const float *const *g, float *const *dW, float *const *db,
                                float *dx, int32_t lddx, void *stream) {
    if (!x || !g || N < 1 || N > FC_MAXN || in < 1 || out < 1 || ldx < in || (dx && lddx < in)) return CAPE_EINVAL;
    LongArgs A;
    int rc = long_args(A, nmat, W, nullptr, g, nullptr, dW, db);
    if (rc) return rc;
    for (int m = 0; m < nmat; ++m)
        if (!A.g[m]) return CAPE_EINVAL;
    bool v4 = N <= 16 && (out & 3) == 0 && (size_t)(16 * 64 + nmat * 16 * out) * 4 <= 60 * 1024;
    for (int m = 0; m < nmat; ++m) v4 = v4 && fc_al16(A.W[m]) && (!A.dW[m] || fc_al16(A.dW[m]));
    bool m16 = fc_m16_on && v4 && (out == 64 || (out == 128 && nmat == 1)) && (!dx || ((lddx & 3) == 0 && fc_al16(dx)));
    for (int m = 0; m < nmat; ++m) m16 = m16 && fc_al16(A.g[m]);
    if (m16) {
        const dim3 grid((in + FC_RS - 1) / FC_RS);
        hipStream_t st = (hipStream_t)stream;
        if (out == 128) CAPE_LAUNCH((fc_long_bwd_m16_kernel<2, 1>), grid, dim3(256), 0, st, A, x, ldx, N, in, dx, lddx);
        else if (nmat == 1) CAPE_LAUNCH((fc_long_bwd_m16_kernel<1, 1>), grid, dim3(256), 0, st, A, x, ldx, N, in, dx, lddx);
        else CAPE_LAUNCH((fc_long_bwd_m16_kernel<1, 2>), grid, dim3(256), 0, st, A, x, ldx, N, in, dx, lddx);
        CAPE_LAUNCH_CHECK();
        return CAPE_OK;
    }
    if (v4) {
        CAPE_LAUNCH(fc_long_bwd_v4_kernel, dim3((in + 63) / 64), dim3(256), (size_t)(16 * 64 + nmat * 16 * out) * 4, (hipStream_t)stream, A, x, ldx, N,
                    in, out, dx, lddx);
        CAPE_LAUNCH_CHECK();
        return CAPE_OK;
    }
    const size_t lds = ((size_t)N * 64 + (size_t)nmat * N * out + 64 * 65) * 4;
    if (lds > 60 * 1024) return CAPE_EINVAL;
    CAPE_LAUNCH(fc_long_bwd_kernel, dim3((in + 63) / 64), dim3(256), lds, (hipStream_t)stream, A, x, ldx, N, in, out, dx, lddx);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_fc_wide_fwd(const float *x, int32_t ldx, int32_t N, int32_t in, int32_t out, const float *W,
                                const float *b, int32_t act, float *y, int32_t ldy, void *stream) {
    if (!x || !W || !y || N < 1 || N > FC_MAXN || in < 1 || out < 1 || ldx < in || ldy < out) return CAPE_EINVAL;
    if (act < CAPE_ACT_NONE || act > CAPE_ACT_TANH || (long long)N * in * 4 > 48 * 1024) return CAPE_EINVAL;
    const bool m16 = fc_m16_on && N <= 16 && (out & 63) == 0 && (in == 64 || in == 128 || in == 256) && fc_al16(W) && fc_al16(x) &&
                     (ldx & 3) == 0 && fc_al16(y) && (ldy & 3) == 0 && (!b || fc_al16(b));
    if (m16 && in == 64)
        CAPE_LAUNCH((fc_wide_fwd_m16_kernel<1>), dim3(out / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, N, out, W, b, act, y, ldy);
    else if (m16 && in == 128)
        CAPE_LAUNCH((fc_wide_fwd_m16_kernel<2>), dim3(out / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, N, out, W, b, act, y, ldy);
    else if (m16)
        CAPE_LAUNCH((fc_wide_fwd_m16_kernel<4>), dim3(out / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, N, out, W, b, act, y, ldy);
    else if (N <= 16 && (out & 3) == 0 && fc_al16(W) && (size_t)(16 * in + 4 * 16 * 128) * 4 <= 60 * 1024)
        CAPE_LAUNCH(fc_wide_fwd_v4_kernel, dim3((out + 127) / 128), dim3(256), (size_t)(16 * in + 4 * 16 * 128) * 4, (hipStream_t)stream, x, ldx, N,
                    in, out, W, b, act, y, ldy);
    else
        CAPE_LAUNCH(fc_wide_fwd_kernel, dim3((out + 255) / 256), dim3(256), (size_t)N * in * 4, (hipStream_t)stream, x, ldx, N, in, out, W, b, act, y,
                    ldy);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int64_t cape_fc_wide_bwd_workspace_bytes(int32_t N, int32_t in, int32_t out) {
    if (N < 1 || N > FC_MAXN || in < 1 || out < 1) return CAPE_EINVAL;
    return (int64_t)((out + 63) / 64) * N * in * (int64_t)sizeof(float);
}

extern "C" int cape_fc_wide_bwd(const float *x, int32_t ldx, const float *g, int32_t ldg, const float *y, int32_t ldy,
                                int32_t act, int32_t N, int32_t in, int32_t out, const float *W, float *dW, float *db,
                                float *dx, int32_t lddx, void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !g || !W || N < 1 || N > FC_MAXN || in < 1 || out < 1 || ldx < in || ldg < out) return CAPE_EINVAL;
    if (act < CAPE_ACT_NONE || act > CAPE_ACT_TANH || (act != CAPE_ACT_NONE && (!y || ldy < out))) return CAPE_EINVAL;
    if ((long long)N * in * 4 > 48 * 1024 || ((long long)in * 65 + (long long)N * 64) * 4 > 60 * 1024) return CAPE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dW || db) {
        const bool v4 = N <= 16 && (out & 3) == 0 && (ldg & 3) == 0 && fc_al16(g) && (!dW || fc_al16(dW)) && (!db || fc_al16(db)) &&
                        (act == CAPE_ACT_NONE || ((ldy & 3) == 0 && fc_al16(y)));
        const bool m16 = fc_m16_on && v4 && (out & 63) == 0 && (in == 64 || in == 128 || in == 256) && (ldx & 3) == 0;
        if (m16 && in == 64)
            CAPE_LAUNCH((fc_wide_bwd_dw_m16_kernel<1>), dim3(out / 64), dim3(256), 0, st, x, ldx, g, ldg, y, ldy, act, N, out, dW, db);
        else if (m16 && in == 128)
            CAPE_LAUNCH((fc_wide_bwd_dw_m16_kernel<2>), dim3(out / 64), dim3(256), 0, st, x, ldx, g, ldg, y, ldy, act, N, out, dW, db);
        else if (m16)
            CAPE_LAUNCH((fc_wide_bwd_dw_m16_kernel<4>), dim3(out / 64), dim3(256), 0, st, x, ldx, g, ldg, y, ldy, act, N, out, dW, db);
        else if (v4)
            CAPE_LAUNCH(fc_wide_bwd_dw_v4_kernel, dim3((out + 255) / 256, dW ? 4 : 1), dim3(256), (size_t)16 * in * 4, st, x, ldx, g, ldg, y, ldy, act, N, in, out,
                        dW, db);
        else
            CAPE_LAUNCH(fc_wide_bwd_dw_kernel, dim3((out + 255) / 256), dim3(256), (size_t)N * in * 4, st, x, ldx, g, ldg, y, ldy, act, N, in, out, dW,
                        db);
        CAPE_LAUNCH_CHECK();
    }
    if (dx) {
        if (lddx < in || !workspace) return CAPE_EINVAL;
        if (workspace_bytes < cape_fc_wide_bwd_workspace_bytes(N, in, out)) return CAPE_EWORKSPACE;
        int chunks = (out + 63) / 64;
        const bool m16 = fc_m16_on && N <= 16 && (out & 63) == 0 && (in == 64 || in == 128 || in == 256) && fc_al16(W) && fc_al16(g) && (ldg & 3) == 0 &&
                         (act == CAPE_ACT_NONE || ((ldy & 3) == 0 && fc_al16(y)));
        if (m16) {
            chunks = (chunks + FC_DXC - 1) / FC_DXC;             // partial slabs: one per FC_DXC chunks of 64 columns
            if (in == 64) CAPE_LAUNCH((fc_wide_bwd_dx_m16_kernel<1>), dim3(chunks), dim3(256), 0, st, g, ldg, y, ldy, act, N, out, W, (float *)workspace);
            else if (in == 128) CAPE_LAUNCH((fc_wide_bwd_dx_m16_kernel<2>), dim3(chunks), dim3(256), 0, st, g, ldg, y, ldy, act, N, out, W, (float *)workspace);
            else CAPE_LAUNCH((fc_wide_bwd_dx_m16_kernel<4>), dim3(chunks), dim3(256), 0, st, g, ldg, y, ldy, act, N, out, W, (float *)workspace);
        } else if (N <= 16 && (in & 3) == 0 && fc_al16(workspace) && ((size_t)64 * (in + 4) + 16 * 64) * 4 <= 60 * 1024)
            CAPE_LAUNCH(fc_wide_bwd_dx_partial_v4_kernel, dim3(chunks), dim3(256), ((size_t)64 * (in + 4) + 16 * 64) * 4, st, g, ldg, y, ldy, act, N, in,
                        out, W, (float *)workspace);
        else
            CAPE_LAUNCH(fc_wide_bwd_dx_partial_kernel, dim3(chunks), dim3(256), ((size_t)in * 65 + (size_t)N * 64) * 4, st, g, ldg, y, ldy, act, N, in,
                        out, W, (float *)workspace);
        CAPE_LAUNCH_CHECK();
        const long long outs = (long long)N * in;
        CAPE_LAUNCH(fc_wide_bwd_dx_final_kernel, dim3((unsigned)((outs + 15) / 16)), dim3(256), 0, st, (const float *)workspace, chunks, N, in, dx,
                    lddx);
        CAPE_LAUNCH_CHECK();
    }
    return CAPE_OK;
}
