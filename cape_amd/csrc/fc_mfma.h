// MFMA forms of the dense layers with one very long side, for the shipped shapes (N <= 16 samples, 64-column groups,
// 16-byte aligned rows).  Included by fc.hip inside its anonymous namespace (LongArgs, FC_RS come from there).
//
// The 16 samples are one side of v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation), so the contraction needs
// no LDS broadcasts, no cross-lane sums and ~16 accumulator registers per 64 columns: every lane keeps 8-32 sixteen-byte
// weight accesses in flight, which is what streaming 28 MB of weights through a ~10 us kernel needs (the register-tiled
// kernels of fc.hip reach 1.3 TB/s on the same shapes; measured numbers in DESIGN.md section 4).
//
// Lane l of a wave holds  A[m = l % 16][k = l / 16],  B[k = l / 16][n = l % 16],  D[m = 4 (l / 16) + r][n = l % 16], r = 0..3.
// A float4 read along a matrix row is spread over FOUR instructions: component c of every lane forms one operand, so a
// lane's 16 bytes stay one access, and the operand index they stand for (column 4 n + c, or contraction row 4 k + c) is a
// bijection either way -- the other operand is read with the same rule.
#pragma once

#define FC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 fc_ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 fc_zero4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}

// g * act'(y) for four columns (y is not read without an activation)
__device__ __forceinline__ f32x4 fc_dz4(const float *g, const float *y, int act) {
    f32x4 v = fc_ld4(g);
    if (act != CAPE_ACT_NONE) {
        const f32x4 yv = fc_ld4(y);
        v[0] *= cape_act_grad_from_out(yv[0], act); v[1] *= cape_act_grad_from_out(yv[1], act);
        v[2] *= cape_act_grad_from_out(yv[2], act); v[3] *= cape_act_grad_from_out(yv[3], act);
    }
    return v;
}

// ---- long input, forward.  block (split, m) = FC_RS = 128 contraction rows, wave = two steps of 16 rows ----------------------
//   A = x[n][i0 + 4 k + c]     B = W[i0 + 4 k + c][64 T + 4 n' + c']     D = y[4 k + r][64 T + 4 n' + c']
// partial[m][split][n][j] as the register-tiled kernel writes it (fc_long_final_kernel finishes both).
template <int Q>
__global__ __launch_bounds__(256) void fc_long_partial_m16_kernel(LongArgs A, const float *x, int ldx, int N, int in, int nsplit, float *part) {
    constexpr int out = 64 * Q;
    __shared__ float red[3 * 16 * out];
    const int split = blockIdx.x, m = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 15, kk = lane >> 4;
    const float *W = A.W[m];
    f32x4 xq[2], wq[2][4][Q];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ia = split * FC_RS + 16 * (2 * wave + s) + 4 * kk;      // in % 4 == 0: a quad of rows is inside or outside
        xq[s] = (nl < N && ia < in) ? fc_ld4(x + (long long)nl * ldx + ia) : fc_zero4();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int T = 0; T < Q; ++T) wq[s][c][T] = fc_ld4(W + (long long)min(ia + c, in - 1) * out + 64 * T + 4 * nl);   // x is 0 there
    }
    f32x4 acc[Q][4];
#pragma unroll
    for (int T = 0; T < Q; ++T)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) acc[T][c2] = fc_zero4();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int T = 0; T < Q; ++T)
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) acc[T][c2] = FC_MFMA(xq[s][c], wq[s][c][T][c2], acc[T][c2]);
    // the four waves' sums in a fixed order; lane (k, n') holds y[4 k + r][64 T + 4 n' + c']: a float4 over c' per r
    if (wave > 0) {
#pragma unroll
        for (int T = 0; T < Q; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 v = {acc[T][0][r], acc[T][1][r], acc[T][2][r], acc[T][3][r]};
                *reinterpret_cast<f32x4 *>(&red[((wave - 1) * 16 + 4 * kk + r) * out + 64 * T + 4 * nl]) = v;
            }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int T = 0; T < Q; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 4 * kk + r, o = n * out + 64 * T + 4 * nl;
                f32x4 v = {acc[T][0][r], acc[T][1][r], acc[T][2][r], acc[T][3][r]};
                v = ((v + fc_ld4(&red[o])) + fc_ld4(&red[16 * out + o])) + fc_ld4(&red[2 * 16 * out + o]);
                if (n < N) *reinterpret_cast<f32x4 *>(part + (((long long)m * nsplit + split) * N + n) * out + 64 * T + 4 * nl) = v;
            }
    }
}

// ---- long input, backward.  block = 128 contraction rows, wave = two steps of 16 rows, all NM matrices --------------------------
//   dW[i0 + 4 k + r][64 T + 4 n' + c'] = sum_n x[n][i0 + m] g[n][..]     A = x[4 s + k][i0 + m]          B = g[4 s + k][64 T + 4 n' + c']
//   dx[n'][i0 + 4 k + r] = sum_j W[i0 + m][j] g[n'][j]                   A = W[i0 + m][16 u + 4 k + c]   B = g[n'][16 u + 4 k + c]
template <int Q, int NM>
__global__ __launch_bounds__(256) void fc_long_bwd_m16_kernel(LongArgs A, const float *x, int ldx, int N, int in, float *dx, int lddx) {
    constexpr int out = 64 * Q;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 15, kk = lane >> 4;
    f32x4 gq[NM][4][Q];          // weight-gradient operand: g[4 s + k][64 T + 4 n' ..]
    f32x4 gt[NM][4 * Q];         // data-gradient operand:   g[n'][16 u + 4 k ..]
#pragma unroll
    for (int m = 0; m < NM; ++m) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int T = 0; T < Q; ++T)
                gq[m][s][T] = (4 * s + kk < N) ? fc_ld4(A.g[m] + (long long)(4 * s + kk) * out + 64 * T + 4 * nl) : fc_zero4();
#pragma unroll
        for (int u = 0; u < 4 * Q; ++u) gt[m][u] = (nl < N) ? fc_ld4(A.g[m] + (long long)nl * out + 16 * u + 4 * kk) : fc_zero4();
    }
    if (blockIdx.x == 0) {
        for (int m = 0; m < NM; ++m)
            if (A.db[m])
                for (int j = threadIdx.x; j < out; j += 256) {
                    float sb = 0.f;
                    for (int n = 0; n < N; ++n) sb += A.g[m][(long long)n * out + j];
                    A.db[m][j] = sb;
                }
    }
    // both steps' loads first (x, and the weight rows the data gradient needs), then the matrix instructions and the stores
    float xa[2][4];
    f32x4 wv[2][NM][4 * Q];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int i0 = blockIdx.x * FC_RS + 16 * (2 * wave + s2);
        const int im = min(i0 + nl, in - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) xa[s2][s] = (4 * s + kk < N && i0 + nl < in) ? x[(long long)(4 * s + kk) * ldx + im] : 0.f;
        if (dx) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int u = 0; u < 4 * Q; ++u) wv[s2][m][u] = fc_ld4(A.W[m] + (long long)im * out + 16 * u + 4 * kk);
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int i0 = blockIdx.x * FC_RS + 16 * (2 * wave + s2);
        if (i0 < in) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (!A.dW[m]) continue;
#pragma unroll
                for (int T = 0; T < Q; ++T) {
                    f32x4 d[4];
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2) {
                        d[c2] = fc_zero4();
#pragma unroll
                        for (int s = 0; s < 4; ++s) d[c2] = FC_MFMA(xa[s2][s], gq[m][s][T][c2], d[c2]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i0 + 4 * kk + r;
                        f32x4 v = {d[0][r], d[1][r], d[2][r], d[3][r]};
                        if (row < in) *reinterpret_cast<f32x4 *>(A.dW[m] + (long long)row * out + 64 * T + 4 * nl) = v;
                    }
                }
            }
            if (dx) {
                f32x4 e = fc_zero4();
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int u = 0; u < 4 * Q; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c) e = FC_MFMA(wv[s2][m][u][c], gt[m][u][c], e);
                if (nl < N && i0 + 4 * kk < in) *reinterpret_cast<f32x4 *>(dx + (long long)nl * lddx + i0 + 4 * kk) = e;
            }
        }
    }
}

// ---- wide output, forward.  block = 64 output columns, wave = a quarter of the 64 NR input rows ------------------------------
//   A = x[n][i0 + 4 k + c]     B = W[i0 + 4 k + c][j0 + 4 n' + c']
template <int NR>
__global__ __launch_bounds__(256) void fc_wide_fwd_m16_kernel(const float *x, int ldx, int N, int out, const float *W, const float *b, int act,
                                                             float *y, int ldy) {
    __shared__ float red[3 * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 15, kk = lane >> 4;
    const int j0 = blockIdx.x * 64;
    f32x4 acc[4];
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) acc[c2] = fc_zero4();
    f32x4 xq[NR], wq[NR][4];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
        const int ia = wave * 16 * NR + 16 * s + 4 * kk;
        xq[s] = (nl < N) ? fc_ld4(x + (long long)nl * ldx + ia) : fc_zero4();
#pragma unroll
        for (int c = 0; c < 4; ++c) wq[s][c] = fc_ld4(W + (long long)(ia + c) * out + j0 + 4 * nl);
    }
#pragma unroll
    for (int s = 0; s < NR; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) acc[c2] = FC_MFMA(xq[s][c], wq[s][c][c2], acc[c2]);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            *reinterpret_cast<f32x4 *>(&red[((wave - 1) * 16 + 4 * kk + r) * 64 + 4 * nl]) = v;
        }
    }
    __syncthreads();
    if (wave == 0) {
        const f32x4 bv = b ? fc_ld4(b + j0 + 4 * nl) : fc_zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * kk + r, o = n * 64 + 4 * nl;
            f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            v = ((v + fc_ld4(&red[o])) + fc_ld4(&red[16 * 64 + o])) + fc_ld4(&red[2 * 16 * 64 + o]);
            v = v + bv;
            v[0] = cape_act(v[0], act); v[1] = cape_act(v[1], act); v[2] = cape_act(v[2], act); v[3] = cape_act(v[3], act);
            if (n < N) *reinterpret_cast<f32x4 *>(y + (long long)n * ldy + j0 + 4 * nl) = v;
        }
    }
}

// ---- wide output, backward (weights).  block = 64 columns, wave w = row blocks R = w, w + 4, .. (NR of them) --------------------
//   dW[16 R + 4 k + r][j0 + 4 n' + c'] = sum_n x[n][16 R + m] dz[n][..]     A = x[4 s + k][16 R + m]     B = dz[4 s + k][j0 + 4 n' + c']
template <int NR>
__global__ __launch_bounds__(256) void fc_wide_bwd_dw_m16_kernel(const float *x, int ldx, const float *g, int ldg, const float *y, int ldy, int act,
                                                                int N, int out, float *dW, float *db) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 15, kk = lane >> 4;
    const int j0 = blockIdx.x * 64;
    f32x4 dz[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int n = 4 * s + kk;
        dz[s] = (n < N) ? fc_dz4(g + (long long)n * ldg + j0 + 4 * nl, y + (long long)n * ldy + j0 + 4 * nl, act) : fc_zero4();
    }
    if (db && wave == 0) {
        f32x4 sb = ((dz[0] + dz[1]) + dz[2]) + dz[3];                    // samples 4 s + k: over s here, over k across the lanes
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = sb[c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            sb[c] = v;
        }
        if (kk == 0) *reinterpret_cast<f32x4 *>(db + j0 + 4 * nl) = sb;
    }
    if (!dW) return;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int R = wave + 4 * q;
        float xa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xa[s] = (4 * s + kk < N) ? x[(long long)(4 * s + kk) * ldx + 16 * R + nl] : 0.f;
        f32x4 d[4];
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            d[c2] = fc_zero4();
#pragma unroll
            for (int s = 0; s < 4; ++s) d[c2] = FC_MFMA(xa[s], dz[s][c2], d[c2]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 v = {d[0][r], d[1][r], d[2][r], d[3][r]};
            *reinterpret_cast<f32x4 *>(dW + (long long)(16 * R + 4 * kk + r) * out + j0 + 4 * nl) = v;
        }
    }
}

// ---- wide output, backward (data): part[blk][n][i] = sum over the block's FC_DXC chunks of 64 columns ---------------------------
//   A = dz[n][j0 + 16 u + 4 k + c]     B = W[16 R + n'][j0 + 16 u + 4 k + c]     D = part[4 k + r][16 R + n']
constexpr int FC_DXC = 2;
template <int NR>
__global__ __launch_bounds__(256) void fc_wide_bwd_dx_m16_kernel(const float *g, int ldg, const float *y, int ldy, int act, int N, int out,
                                                                const float *W, float *part) {
    constexpr int in = 64 * NR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 15, kk = lane >> 4;
    f32x4 dq[FC_DXC][4], wq[FC_DXC][NR][4];                // every access of the block's chunks in flight before the first MFMA
#pragma unroll
    for (int ch = 0; ch < FC_DXC; ++ch) {
        const bool on = (blockIdx.x * FC_DXC + ch) * 64 < out;
        const int j0 = on ? (blockIdx.x * FC_DXC + ch) * 64 : out - 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 16 * u + 4 * kk;
            dq[ch][u] = (nl < N && on) ? fc_dz4(g + (long long)nl * ldg + j, y + (long long)nl * ldy + j, act) : fc_zero4();
#pragma unroll
            for (int q = 0; q < NR; ++q) wq[ch][q][u] = fc_ld4(W + (long long)(16 * (wave + 4 * q) + nl) * out + j);
        }
    }
    f32x4 e[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) e[q] = fc_zero4();
#pragma unroll
    for (int ch = 0; ch < FC_DXC; ++ch)
#pragma unroll
        for (int q = 0; q < NR; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) e[q] = FC_MFMA(dq[ch][u][c], wq[ch][q][u][c], e[q]);
    float *dst = part + (long long)blockIdx.x * N * in;
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * kk + r < N) dst[(long long)(4 * kk + r) * in + 16 * (wave + 4 * q) + nl] = e[q][r];
}

static const bool fc_m16_on = !(getenv("CAPE_FC_MFMA") && atoi(getenv("CAPE_FC_MFMA")) == 0);      // 0: A/B against the register-tiled kernels
