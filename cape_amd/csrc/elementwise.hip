// Bandwidth-bound companions of the fused gather-GEMM (gfx950): standalone sparse operator
// application, bias/activation and their gradients, column reductions, condition-channel
// fill.  All kernels read/write [N, M, ld] fp32 views with channels contiguous and use
// float4 accesses whenever the view is 16-byte aligned.
#include "common.h"

namespace {

template <typename T> struct ViewT {
    T *p;
    long long ss;
    int ld;
};
template <typename T> struct CViewT {
    const T *p;
    long long ss;
    int ld;
};
typedef ViewT<float> View;
typedef CViewT<float> CView;

// four consecutive elements addressable as one vector access (16 bytes of fp32, 8 bytes of bf16)
__host__ __device__ inline bool aligned4(const void *p, long long ss, int ld, int C, int es = 4) {
    return ((reinterpret_cast<uintptr_t>(p) & (4 * es - 1)) == 0) && ((ss & 3) == 0) && ((ld & 3) == 0) && ((C & 3) == 0);
}

// blocks per sample of the sparse kernels (256 work items = (row, channel quad) pairs each)
__host__ __device__ inline int spmm_bps(int Mo, int cq) { return (int)(((long long)Mo * cq + 255) / 256); }

#ifndef CAPE_SPMM_UNROLL_DEFAULT
#define CAPE_SPMM_UNROLL_DEFAULT 4
#endif
#ifndef CAPE_SPMM_WIDE_DEFAULT
#define CAPE_SPMM_WIDE_DEFAULT 1
#endif
// Knobs of the vector sparse kernels (A/B switches, read once):
//   CAPE_SPMM_UNROLL = 0 (entry loop as written), 4 or 8 (entries in unrolled groups, see cape_gather_row)
//   CAPE_SPMM_WIDE   = 1: 8 channels per work item where the channel count and the alignment allow, 0: always 4
inline int spmm_unroll() {
    static const int u = getenv("CAPE_SPMM_UNROLL") ? atoi(getenv("CAPE_SPMM_UNROLL")) : CAPE_SPMM_UNROLL_DEFAULT;
    return u >= 8 ? 8 : u >= 4 ? 4 : 0;
}
inline bool spmm_wide() {
    static const int w = getenv("CAPE_SPMM_WIDE") ? atoi(getenv("CAPE_SPMM_WIDE")) : CAPE_SPMM_WIDE_DEFAULT;
    return w != 0;
}
// eight consecutive elements addressable as vector accesses (fp32: two 16-byte accesses, bf16: one)
inline bool aligned8(const void *p, long long ss, int ld, int C, int es) {
    const int a = es == 4 ? 3 : 7;          // element alignment of the row starts
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ss & a) == 0) && ((ld & a) == 0) && ((C & 7) == 0);
}
// launch KERNEL<VW, T, U> with VW = 8 / 4 (wide) and U from the knob (hint4: rows of at most 4 entries -> groups of 4)
#define CAPE_LAUNCH_SP(KERNEL, T, wide, hint4, ...)                                  \
    do {                                                                             \
        const int u_ = (hint4 && spmm_unroll()) ? 4 : spmm_unroll();                 \
        if (wide) {                                                                  \
            if (u_ == 8) CAPE_LAUNCH((KERNEL<8, T, 8>), __VA_ARGS__);                \
            else if (u_ == 4) CAPE_LAUNCH((KERNEL<8, T, 4>), __VA_ARGS__);           \
            else CAPE_LAUNCH((KERNEL<8, T, 0>), __VA_ARGS__);                        \
        } else {                                                                     \
            if (u_ == 8) CAPE_LAUNCH((KERNEL<4, T, 8>), __VA_ARGS__);                \
            else if (u_ == 4) CAPE_LAUNCH((KERNEL<4, T, 4>), __VA_ARGS__);           \
            else CAPE_LAUNCH((KERNEL<4, T, 0>), __VA_ARGS__);                        \
        }                                                                            \
    } while (0)

// One output row of a CSR operator, VW channels: acc = sum_e va[e] * x[ci[e], c..c+VW-1].
// U = 0: entry loop as written -- the index load, then the row load that depends on it, then the next entry: two
// dependent memory round trips per entry.
// U > 0: entries in groups of U, fully unrolled: all (index, value) pairs of a group are loaded first, then all row
// gathers are in flight together, so a row of <= U entries costs three dependent round trips (row pointer, entries,
// rows).  Slots past the end of the row re-read its first entry with weight 0 (same address as slot 0: one L1 line)
// and add nothing; the order of the sum is unchanged.
template <int VW, int U, typename T>
__device__ __forceinline__ void cape_gather_row(const T *xb, long long ldx, const int *rp, const int *ci, const float *va, int r,
                                                float (&acc)[VW]) {
#pragma unroll
    for (int u = 0; u < VW; ++u) acc[u] = 0.f;
    int e = rp[r];
    const int e1 = rp[r + 1];
    if constexpr (U == 0) {
        for (; e < e1; ++e) {
            const float v = va[e];
            float xv[VW];
            cape_ldv<VW>(xb + (long long)ci[e] * ldx, xv);
#pragma unroll
            for (int u = 0; u < VW; ++u) acc[u] = fmaf(v, xv[u], acc[u]);
        }
    } else {
        for (; e < e1; e += U) {
            int cols[U];
            float vals[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool ok = e + j < e1;
                const int ee = ok ? e + j : e;
                cols[j] = ci[ee];
                vals[j] = ok ? va[ee] : 0.f;
            }
            float xv[U][VW];
#pragma unroll
            for (int j = 0; j < U; ++j) cape_ldv<VW>(xb + (long long)cols[j] * ldx, xv[j]);
#pragma unroll
            for (int j = 0; j < U; ++j)
#pragma unroll
                for (int u = 0; u < VW; ++u) acc[u] = fmaf(vals[j], xv[j][u], acc[u]);
        }
    }
}

// ELL form of the same row (arrays [rows, ew], ew in {4, 8, 12}; entries in CSR order packed to the front, slots past the
// row's end hold (column of slot 0, 0.0f)): there is no row pointer to wait for, the index / value quads of the whole row
// are three independent 16-byte loads, and a row of <= 8 entries (every row of L~ but a handful) costs TWO dependent memory
// round trips instead of five (row pointer; per group of four: entries, rows).  These kernels are bound by exactly that
// chain times the occupancy (profiles/README.md), not by bytes.  Same entries in the same order as the CSR form: the sums
// are bit-identical.  An all-zero value quad ends the row (padding, or entries that contribute nothing).
template <int VW, typename T>
__device__ __forceinline__ void cape_gather_row_ell(const T *xb, long long ldx, const int *ec, const float *ev, int ew, int r,
                                                    float (&acc)[VW]) {
#pragma unroll
    for (int u = 0; u < VW; ++u) acc[u] = 0.f;
    const int4 *c4 = reinterpret_cast<const int4 *>(ec + (long long)r * ew);
    const float4 *v4 = reinterpret_cast<const float4 *>(ev + (long long)r * ew);
    int4 c0 = c4[0], c1 = c0, c2 = c0;
    float4 v0 = v4[0], v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1;
    if (ew > 4) { c1 = c4[1]; v1 = v4[1]; }
    if (ew > 8) { c2 = c4[2]; v2 = v4[2]; }
    auto live = [](const float4 &v) { return v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f; };
    auto gather4 = [&](const int4 &c, float (&xv)[4][VW]) {
        cape_ldv<VW>(xb + (long long)c.x * ldx, xv[0]); cape_ldv<VW>(xb + (long long)c.y * ldx, xv[1]);
        cape_ldv<VW>(xb + (long long)c.z * ldx, xv[2]); cape_ldv<VW>(xb + (long long)c.w * ldx, xv[3]);
    };
    auto fma4 = [&](const float4 &v, const float (&xv)[4][VW]) {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < VW; ++u) acc[u] = fmaf(vv[j], xv[j][u], acc[u]);
    };
    float xa[4][VW], xb2[4][VW];
    const bool g1 = live(v1);
    gather4(c0, xa);
    if (g1) gather4(c1, xb2);                     // both groups in flight together
    fma4(v0, xa);
    if (g1) {
        fma4(v1, xb2);
        if (live(v2)) {
            gather4(c2, xa);
            fma4(v2, xa);
        }
    }
}

// one output row, either form (ew = 0: CSR)
template <int VW, int U, typename T>
__device__ __forceinline__ void cape_gather(const T *xb, long long ldx, const int *rp, const int *ci, const float *va, int ew, int r,
                                            float (&acc)[VW]) {
    if (ew) cape_gather_row_ell<VW>(xb, ldx, ci, va, ew, r, acc);
    else cape_gather_row<VW, U>(xb, ldx, rp, ci, va, r, acc);
}

// ---- spmm: y[n,r,:] = alpha * S x[n] + beta * z[n,r,:] ------------------------------------
// work item = (sample, output row, VW channels); VW = 1 for unaligned / odd channel counts
template <int VW, typename T = float, int U = 0>
__global__ __launch_bounds__(256) void spmm_kernel(CViewT<T> x, const int *rp, const int *ci, const float *va, int ew,
                                                   float alpha, CViewT<T> z, float beta, ViewT<T> y, int N, int Mo, int C, float *rm) {
    const int cq = (C + VW - 1) / VW;
    // block -> (sample, 256 work items of that sample), all blocks of a sample on ONE XCD (spmm_grid / cape_map_block): the
    // ~7 neighbour rows an output row gathers are then served by the L2 that already holds that sample, instead of every
    // XCD fetching nearly the whole input (measured: 2.9x the algorithmic bytes in L2-miss traffic with linear blocks)
    int n, t;
    cape_map_block(blockIdx.x, N, spmm_bps(Mo, cq), n, t);
    const int i = t * 256 + (int)threadIdx.x;               // Mo * cq < 2^31 (checked by the host entry)
    if (i >= Mo * cq) return;
    const int r = i / cq;
    const int c = (i - r * cq) * VW;
    float acc[VW];
    cape_gather<VW, U>(x.p + (long long)n * x.ss + c, x.ld, rp, ci, va, ew, r, acc);
#pragma unroll
    for (int u = 0; u < VW; ++u) acc[u] *= alpha;
    if (z.p) {
        float zv[VW];
        cape_ldv<VW>(z.p + (long long)n * z.ss + (long long)r * z.ld + c, zv);
#pragma unroll
        for (int u = 0; u < VW; ++u) acc[u] = fmaf(beta, zv[u], acc[u]);
    }
    cape_stv<VW>(y.p + (long long)n * y.ss + (long long)r * y.ld + c, acc);
    if (rm) {                                               // row bound of y (the cq lanes of a row are consecutive and aligned)
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < VW; ++u) m = fmaxf(m, fabsf(acc[u]));
        m = cape_group_max(m, cq);
        if (c == 0) cape_store_rowmax(rm, (long long)n * Mo + r, m);
    }
}

// ---- several operator applications in one launch -----------------------------------------------------------
// separate mode: y_k = S_k x_k for every term (the X_k = S_k x of one Chebyshev layer, or T_k = S_k^T dz of its
// data gradient);  sum mode: y = sum_k S_k x_k (dx = sum_k S_k^T G_k).  A term without CSR is the identity.
// One work item = (sample, output row, VW channels) of ALL terms: a layer pays one dispatch instead of K, and the
// terms' gathers of one neighbourhood hit the same L1/L2 lines.
struct SpmmTerms {
    struct T {
        const void *x; long long xs; int ldx;       // elements of the launch's storage type
        const int *rp; const int *ci; const float *va;      // ew > 0: ci / va are the ELL arrays [rows, ew]
        void *y; long long ys; int ldy;
        float scale;
        int ew;
        float *rm;                                  // row bounds of y (separate mode) or null
    } t[CAPE_MAX_SPMM_TERMS];
    int n;
};

// Sum mode with the activation gradient of the layer BELOW fused in (encoder chain, reference lib/models.py:154-171 cnp): the
// summed application is that layer's incoming gradient g; its own backward would next read g and its output x = act(z) once
// more to form dz = g * act'(x) and the channel-bias gradient sum_{n,r} dz (cape_bwd_prep: three tensor passes, one launch).
// Here the epilogue multiplies by act'(x) before the store and leaves the bias sums of the block's rows as partials in the
// layout cape_bwd_prep_finalize reads ([sample][block][2][C], term 0), so that layer needs no backward-prep launch at all.
struct SpmmActGrad {
    const void *ax; long long axs; int ldax;     // x = act(z) of the layer below: same rows / channels as the output; null = off
    int act;
    float *part;
};

template <int VW, typename T = float, int U = 0>
__global__ __launch_bounds__(256) void spmm_multi_kernel(SpmmTerms P, int sum, ViewT<T> y, int N, int Mo, int C, float *rm, SpmmActGrad G) {
    const int cq = (C + VW - 1) / VW;
    int n, t;
    cape_map_block(blockIdx.x, N, spmm_bps(Mo, cq), n, t);      // see spmm_kernel
    const int i = t * 256 + (int)threadIdx.x;
    const bool live = i < Mo * cq;
    if (!live && !G.ax) return;                             // (the fused form keeps whole blocks alive for its barrier)
    const int ii = live ? i : Mo * cq - 1;
    const int r = ii / cq;
    const int c = (ii - r * cq) * VW;
    float tot[VW];
#pragma unroll
    for (int u = 0; u < VW; ++u) tot[u] = 0.f;
    for (int k = 0; k < P.n; ++k) {
        const SpmmTerms::T &Tm = P.t[k];
        const T *xb = reinterpret_cast<const T *>(Tm.x) + (long long)n * Tm.xs + c;
        float acc[VW];
        if (!Tm.rp) cape_ldv<VW>(xb + (long long)r * Tm.ldx, acc);
        else cape_gather<VW, U>(xb, Tm.ldx, Tm.rp, Tm.ci, Tm.va, Tm.ew, r, acc);
#pragma unroll
        for (int u = 0; u < VW; ++u) acc[u] *= Tm.scale;
        if (sum) {
#pragma unroll
            for (int u = 0; u < VW; ++u) tot[u] += acc[u];
        } else {
            cape_stv<VW>(reinterpret_cast<T *>(Tm.y) + (long long)n * Tm.ys + (long long)r * Tm.ldy + c, acc);
            if (Tm.rm) {
                float m = 0.f;
#pragma unroll
                for (int u = 0; u < VW; ++u) m = fmaxf(m, fabsf(acc[u]));
                m = cape_group_max(m, cq);
                if (c == 0) cape_store_rowmax(Tm.rm, (long long)n * Mo + r, m);
            }
        }
    }
    if (!sum) return;
    if (G.ax) {
        float xv[VW];
        cape_ldv<VW>(reinterpret_cast<const T *>(G.ax) + (long long)n * G.axs + (long long)r * G.ldax + c, xv);
#pragma unroll
        for (int u = 0; u < VW; ++u) tot[u] = live ? tot[u] * cape_act_grad_from_out(xv[u], G.act) : 0.f;
    }
    if (live) cape_stv<VW>(y.p + (long long)n * y.ss + (long long)r * y.ld + c, tot);
    if (rm) {
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < VW; ++u) m = fmaxf(m, fabsf(tot[u]));
        m = cape_group_max(m, cq);
        if (live && c == 0) cape_store_rowmax(rm, (long long)n * Mo + r, m);
    }
    if (G.ax) {
        // column sums over the block's 256 / cq rows (cq a power of two <= 64): inside each wave over the lanes that hold the
        // same column group (strides cq .. 32: one DPP row rotation, the 16- and 32-lane swaps -- no LDS traffic), then the four
        // waves through LDS in a fixed order.  (A first version let cq threads add 256 / cq LDS rows each: +10 us per launch.)
        __shared__ float cs[4][64 * (VW > 1 ? VW : 1)];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int u = 0; u < VW; ++u) {
            float v = tot[u];
            if (cq <= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
            if (cq <= 16) v = cape_sum_xor16(v);
            if (cq <= 32) v = cape_sum_xor32(v);
            tot[u] = v;
        }
        if (lane < cq)
#pragma unroll
            for (int u = 0; u < VW; ++u) cs[wave][lane * VW + u] = tot[u];
        __syncthreads();
        if ((int)threadIdx.x < cq) {
            float s4[VW];
#pragma unroll
            for (int u = 0; u < VW; ++u)
                s4[u] = ((cs[0][threadIdx.x * VW + u] + cs[1][threadIdx.x * VW + u]) + cs[2][threadIdx.x * VW + u]) + cs[3][threadIdx.x * VW + u];
            cape_stv<VW>(G.part + (((long long)n * spmm_bps(Mo, cq) + t) * 2) * C + (int)threadIdx.x * VW, s4);
        }
    }
}

// ---- backward-prep of an affine block fused with the operator application of its data gradient ----------------------------
// res_block_affine at one resolution (reference lib/models.py:776-793: y = relu(conv_K(x)) + conv_1(x), K = 2), differentiated by
// tf.gradients (:460): the incoming gradient g feeds the 1x1 branch as it is and the K-branch as dz = g * [conv_K(x) > 0] (the
// sign bits the forward launch wrote); the data gradient then needs T_1 = L~^T dz next to dz itself, and the rank-1 condition
// terms need sum_r rowscale_j[r] dz[n,r,:] (j < R) and sum_r rowscale_rg[r] g[n,r,:].  cape_bwd_prep + cape_spmm did that in
// two launches and two more tensor passes; here one work item (sample, row, VW channels) writes dz of its row, gathers the
// MASKED neighbour rows of g for T_1 (x_masked = bit ? x : 0, then the same fma chain as cape_spmm: T_1 is bit-identical to
// the two-launch form), and the block reduces the weighted sums of its 256 / cq rows into the partial layout
// cape_bwd_prep_finalize reads ([sample][block][R + 2][C]; slot 0, the bias sum, is not written -- these blocks have no bias).
constexpr int PS_MAXR = 2;           // rank-1 terms of the fused form (K = 2); more: the caller keeps the two launches
struct PrepSpmmP {                      // g / dz / t1: elements of the launch's storage type (fp32 or bf16)
    const void *g; long long gs; int ldg;
    const unsigned *mask; int words;
    const int *rp; const int *ci; const float *va; int ew;
    void *dz; long long dzs; int lddz;
    void *t1; long long t1s; int ldt1;
    const float *rowscale; int R, rg;               // rg < 0: no g-weighted sum
    float *part;
    float *rm_g, *rm_t1;
};

template <int VW>
__device__ __forceinline__ void cape_mask_row(float (&x)[VW], unsigned bits) {
#pragma unroll
    for (int u = 0; u < VW; ++u) x[u] = ((bits >> u) & 1u) ? x[u] : 0.f;
}

// masked forms of cape_gather_row / cape_gather_row_ell: every gathered row is multiplied by its own sign bits first
template <int VW, int U, typename T>
__device__ __forceinline__ void cape_gather_row_masked(const T *xb, long long ldx, const unsigned *mb, int words, int sh, const int *rp,
                                                       const int *ci, const float *va, int r, float (&acc)[VW]) {
#pragma unroll
    for (int u = 0; u < VW; ++u) acc[u] = 0.f;
    int e = rp[r];
    const int e1 = rp[r + 1];
    constexpr int G = U > 0 ? U : 1;
    for (; e < e1; e += G) {
        int cols[G];
        float vals[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const bool ok = e + j < e1;
            const int ee = ok ? e + j : e;
            cols[j] = ci[ee];
            vals[j] = ok ? va[ee] : 0.f;
        }
        float xv[G][VW];
        unsigned mk[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            cape_ldv<VW>(xb + (long long)cols[j] * ldx, xv[j]);
            mk[j] = mb[(long long)cols[j] * words] >> sh;
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            cape_mask_row<VW>(xv[j], mk[j]);
#pragma unroll
            for (int u = 0; u < VW; ++u) acc[u] = fmaf(vals[j], xv[j][u], acc[u]);
        }
    }
}

template <int VW, typename T>
__device__ __forceinline__ void cape_gather_row_ell_masked(const T *xb, long long ldx, const unsigned *mb, int words, int sh,
                                                           const int *ec, const float *ev, int ew, int r, float (&acc)[VW]) {
#pragma unroll
    for (int u = 0; u < VW; ++u) acc[u] = 0.f;
    const int4 *c4 = reinterpret_cast<const int4 *>(ec + (long long)r * ew);
    const float4 *v4 = reinterpret_cast<const float4 *>(ev + (long long)r * ew);
    int4 c0 = c4[0], c1 = c0, c2 = c0;
    float4 v0 = v4[0], v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1;
    if (ew > 4) { c1 = c4[1]; v1 = v4[1]; }
    if (ew > 8) { c2 = c4[2]; v2 = v4[2]; }
    auto live = [](const float4 &v) { return v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f; };
    auto gather4 = [&](const int4 &c, float (&xv)[4][VW], unsigned (&mk)[4]) __attribute__((always_inline)) {
        cape_ldv<VW>(xb + (long long)c.x * ldx, xv[0]); cape_ldv<VW>(xb + (long long)c.y * ldx, xv[1]);
        cape_ldv<VW>(xb + (long long)c.z * ldx, xv[2]); cape_ldv<VW>(xb + (long long)c.w * ldx, xv[3]);
        mk[0] = mb[(long long)c.x * words] >> sh; mk[1] = mb[(long long)c.y * words] >> sh;
        mk[2] = mb[(long long)c.z * words] >> sh; mk[3] = mb[(long long)c.w * words] >> sh;
    };
    auto fma4 = [&](const float4 &v, float (&xv)[4][VW], const unsigned (&mk)[4]) __attribute__((always_inline)) {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cape_mask_row<VW>(xv[j], mk[j]);
#pragma unroll
            for (int u = 0; u < VW; ++u) acc[u] = fmaf(vv[j], xv[j][u], acc[u]);
        }
    };
    float xa[4][VW], xb2[4][VW];
    unsigned ma[4], mb2[4];
    const bool g1 = live(v1);
    gather4(c0, xa, ma);
    if (g1) gather4(c1, xb2, mb2);                // both groups in flight together
    fma4(v0, xa, ma);
    if (g1) {
        fma4(v1, xb2, mb2);
        if (live(v2)) {
            gather4(c2, xa, ma);
            fma4(v2, xa, ma);
        }
    }
}

// one operator row applied to the masked AND to the plain rows from ONE set of loads (an up-sampling affine block applies S_0^T to
// dz and to g): acc_m as cape_gather_row*_masked, acc_p as cape_gather_row* -- same fma chains, bit-identical results
template <int VW, int U, typename T>
__device__ __forceinline__ void cape_gather_row_both(const T *xb, long long ldx, const unsigned *mb, int words, int sh, const int *rp,
                                                     const int *ci, const float *va, int r, float (&acc_m)[VW], float (&acc_p)[VW]) {
#pragma unroll
    for (int u = 0; u < VW; ++u) acc_m[u] = acc_p[u] = 0.f;
    int e = rp[r];
    const int e1 = rp[r + 1];
    constexpr int G = U > 0 ? U : 1;
    for (; e < e1; e += G) {
        int cols[G];
        float vals[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const bool ok = e + j < e1;
            const int ee = ok ? e + j : e;
            cols[j] = ci[ee];
            vals[j] = ok ? va[ee] : 0.f;
        }
        float xv[G][VW];
        unsigned mk[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            cape_ldv<VW>(xb + (long long)cols[j] * ldx, xv[j]);
            mk[j] = mb[(long long)cols[j] * words] >> sh;
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
#pragma unroll
            for (int u = 0; u < VW; ++u) acc_p[u] = fmaf(vals[j], xv[j][u], acc_p[u]);
            cape_mask_row<VW>(xv[j], mk[j]);
#pragma unroll
            for (int u = 0; u < VW; ++u) acc_m[u] = fmaf(vals[j], xv[j][u], acc_m[u]);
        }
    }
}

template <int VW, typename T>
__device__ __forceinline__ void cape_gather_row_ell_both(const T *xb, long long ldx, const unsigned *mb, int words, int sh,
                                                         const int *ec, const float *ev, int ew, int r, float (&acc_m)[VW], float (&acc_p)[VW]) {
#pragma unroll
    for (int u = 0; u < VW; ++u) acc_m[u] = acc_p[u] = 0.f;
    const int4 *c4 = reinterpret_cast<const int4 *>(ec + (long long)r * ew);
    const float4 *v4 = reinterpret_cast<const float4 *>(ev + (long long)r * ew);
    int4 c0 = c4[0], c1 = c0, c2 = c0;
    float4 v0 = v4[0], v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1;
    if (ew > 4) { c1 = c4[1]; v1 = v4[1]; }
    if (ew > 8) { c2 = c4[2]; v2 = v4[2]; }
    auto live = [](const float4 &v) { return v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f; };
    auto gather4 = [&](const int4 &c, float (&xv)[4][VW], unsigned (&mk)[4]) __attribute__((always_inline)) {
        cape_ldv<VW>(xb + (long long)c.x * ldx, xv[0]); cape_ldv<VW>(xb + (long long)c.y * ldx, xv[1]);
        cape_ldv<VW>(xb + (long long)c.z * ldx, xv[2]); cape_ldv<VW>(xb + (long long)c.w * ldx, xv[3]);
        mk[0] = mb[(long long)c.x * words] >> sh; mk[1] = mb[(long long)c.y * words] >> sh;
        mk[2] = mb[(long long)c.z * words] >> sh; mk[3] = mb[(long long)c.w * words] >> sh;
    };
    auto fma4 = [&](const float4 &v, float (&xv)[4][VW], const unsigned (&mk)[4]) __attribute__((always_inline)) {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int u = 0; u < VW; ++u) acc_p[u] = fmaf(vv[j], xv[j][u], acc_p[u]);
            cape_mask_row<VW>(xv[j], mk[j]);
#pragma unroll
            for (int u = 0; u < VW; ++u) acc_m[u] = fmaf(vv[j], xv[j][u], acc_m[u]);
        }
    };
    float xa[4][VW], xb2[4][VW];
    unsigned ma[4], mb2[4];
    const bool g1 = live(v1);
    gather4(c0, xa, ma);
    if (g1) gather4(c1, xb2, mb2);                // both groups in flight together
    fma4(v0, xa, ma);
    if (g1) {
        fma4(v1, xb2, mb2);
        if (live(v2)) {
            gather4(c2, xa, ma);
            fma4(v2, xa, ma);
        }
    }
}

// rpb: consecutive groups of 256 work items a block handles one after the other (the weighted sums of all of them are reduced
// ONCE: at 256 channels a group is only 8 rows, and one partial row per group and term would be a third of the tensor's bytes)
template <int VW, int U, typename AT = float>
__global__ __launch_bounds__(256) void bwd_prep_spmm_kernel(PrepSpmmP P, int N, int Mo, int C, int rpb, int chunks) {
    const int cq = C / VW;                                   // a power of two in 4 .. 64 (host check): the lanes of a row are one aligned group
    int n, t;
    cape_map_block(blockIdx.x, N, chunks, n, t);             // see spmm_kernel
    const AT *gbase = reinterpret_cast<const AT *>(P.g) + (long long)n * P.gs;
    const unsigned *mbase = P.mask + (long long)n * Mo * P.words;
    const int nterm = P.R + (P.rg >= 0 ? 1 : 0);
    float racc[PS_MAXR + 1][VW];
#pragma unroll
    for (int j = 0; j <= PS_MAXR; ++j)
#pragma unroll
        for (int u = 0; u < VW; ++u) racc[j][u] = 0.f;
    for (int it = 0; it < rpb; ++it) {
        const int i = (t * rpb + it) * 256 + (int)threadIdx.x;
        const bool live = i < Mo * cq;                       // (whole blocks stay alive for the reductions)
        const int ii = live ? i : Mo * cq - 1;
        const int r = ii / cq;
        const int c = (ii - r * cq) * VW;
        const AT *gb = gbase + c;
        const unsigned *mb = mbase + (c >> 5);
        const int sh = c & 31;
        float gv[VW], d[VW], acc[VW];
        cape_ldv<VW>(gb + (long long)r * P.ldg, gv);
        const unsigned mw = mb[(long long)r * P.words] >> sh;
        // the row weights of the sums: loaded unconditionally on the clamped row, together with everything else (a guarded load
        // compiles to a branch with a wait for ALL outstanding loads in front of it: one more dependent round trip, +5 us)
        float sv[PS_MAXR + 1];
#pragma unroll
        for (int j = 0; j <= PS_MAXR; ++j) sv[j] = P.rowscale ? P.rowscale[(long long)(j == P.R ? (P.rg >= 0 ? P.rg : 0) : (j < P.R ? j : 0)) * Mo + r] : 0.f;
        if (P.ew) cape_gather_row_ell_masked<VW, AT>(gb, P.ldg, mb, P.words, sh, P.ci, P.va, P.ew, r, acc);
        else cape_gather_row_masked<VW, U, AT>(gb, P.ldg, mb, P.words, sh, P.rp, P.ci, P.va, r, acc);
#pragma unroll
        for (int u = 0; u < VW; ++u) d[u] = gv[u];
        cape_mask_row<VW>(d, mw);
        if (live) {
            cape_stv<VW>(reinterpret_cast<AT *>(P.dz) + (long long)n * P.dzs + (long long)r * P.lddz + c, d);
            cape_stv<VW>(reinterpret_cast<AT *>(P.t1) + (long long)n * P.t1s + (long long)r * P.ldt1 + c, acc);
        }
        if (P.rm_t1) {
            float m = 0.f;
#pragma unroll
            for (int u = 0; u < VW; ++u) m = fmaxf(m, fabsf(acc[u]));
            m = cape_group_max(m, cq);
            if (live && c == 0) cape_store_rowmax(P.rm_t1, (long long)n * Mo + r, m);
        }
        if (P.rm_g) {                                        // bounds g, and therefore dz
            float m = 0.f;
#pragma unroll
            for (int u = 0; u < VW; ++u) m = fmaxf(m, fabsf(gv[u]));
            m = cape_group_max(m, cq);
            if (live && c == 0) cape_store_rowmax(P.rm_g, (long long)n * Mo + r, m);
        }
#pragma unroll
        for (int j = 0; j <= PS_MAXR; ++j)
            if (j < nterm) {
                const bool gterm = j == P.R;
                const float w = live ? sv[j] : 0.f;
#pragma unroll
                for (int u = 0; u < VW; ++u) racc[j][u] = fmaf(w, gterm ? gv[u] : d[u], racc[j][u]);
            }
    }
    // weighted column sums over the block's rows: inside each wave over the lanes that hold the same column group (a lane holds
    // the same group in every pass: 256 % cq == 0) without the LDS crossbar -- DPP row rotations by 4 and 8 inside the 16-lane
    // rows, the 16- and 32-lane swaps across them -- then the four waves through LDS in a fixed order, all terms behind ONE barrier
    // (first version: ds_bpermute shuffles and a barrier pair per term: 5-7 us of a 21 us launch)
    __shared__ float cs[4][PS_MAXR + 1][64 * VW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = P.R + 2;
    float *pp = P.part + ((long long)n * chunks + t) * T * C;
#pragma unroll
    for (int j = 0; j <= PS_MAXR; ++j) {
        if (j >= nterm) break;
#pragma unroll
        for (int u = 0; u < VW; ++u) {
            float v = racc[j][u];
            if (cq <= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
            if (cq <= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
            if (cq <= 16) v = cape_sum_xor16(v);
            if (cq <= 32) v = cape_sum_xor32(v);
            racc[j][u] = v;
        }
        if (lane < cq)
#pragma unroll
            for (int u = 0; u < VW; ++u) cs[wave][j][lane * VW + u] = racc[j][u];
    }
    __syncthreads();
    if ((int)threadIdx.x < cq) {
#pragma unroll
        for (int j = 0; j <= PS_MAXR; ++j) {
            if (j >= nterm) break;
            const bool gterm = j == P.R;
            float s4[VW];
#pragma unroll
            for (int u = 0; u < VW; ++u)
                s4[u] = ((cs[0][j][threadIdx.x * VW + u] + cs[1][j][threadIdx.x * VW + u]) + cs[2][j][threadIdx.x * VW + u]) + cs[3][j][threadIdx.x * VW + u];
            cape_stv<VW>(pp + (long long)(gterm ? P.R + 1 : 1 + j) * C + (int)threadIdx.x * VW, s4);
        }
    }
}

// The same idea for the affine block that ALSO up-samples (reference lib/models.py:776-793 with an unpool in front, :147-151): its
// data gradient needs T_k = S_k^T dz (k < K) and T_aff = S_0^T g at the COARSE input rows, its weight gradient contracts those
// T_k (ChebConvFn coarse_dw), and the rank-1 condition sums are column sums of them:
//     sum_r (S_k 1)[r] dz[n,r,:] = sum_j (S_k^T dz)[n,j,:]
// so dz itself is needed by nobody.  cape_bwd_prep + cape_spmm_multi (separate mode) become ONE launch: every term gathers the
// fine rows of g, the terms flagged in `masked` multiply each gathered row by its sign bits first, and the block leaves the column
// sums of every term as partials in the cape_bwd_prep_finalize layout (term k -> slot 1 + k).
constexpr int MP_MAXT = 3;            // terms of cape_spmm_multi_prep (K = 2 orders + the affine term)
template <int VW, int U, typename AT = float>
__global__ __launch_bounds__(256) void spmm_multi_prep_kernel(SpmmTerms P, unsigned masked, int pair, const unsigned *mask, int words, int Mfine,
                                                              int N, int Mo, int C, float *part, int T, int rpb, int chunks) {
    const int cq = C / VW;                                   // a power of two in 4 .. 64 (host check)
    int n, t;
    cape_map_block(blockIdx.x, N, chunks, n, t);             // see spmm_kernel
    const unsigned *mbase = mask + (long long)n * Mfine * words;
    float racc[MP_MAXT][VW];
#pragma unroll
    for (int k = 0; k < MP_MAXT; ++k)
#pragma unroll
        for (int u = 0; u < VW; ++u) racc[k][u] = 0.f;
    for (int it = 0; it < rpb; ++it) {
        const int i = (t * rpb + it) * 256 + (int)threadIdx.x;
        const bool live = i < Mo * cq;                       // (whole blocks stay alive for the reductions)
        const int ii = live ? i : Mo * cq - 1;
        const int r = ii / cq;
        const int c = (ii - r * cq) * VW;
        const unsigned *mb = mbase + (c >> 5);
        const int sh = c & 31;
        // `pair`: the last term applies the operator of term 0 to the same input, unmasked (T_aff = S_0^T g next to T_0 = S_0^T dz):
        // both from one set of gathered rows
        float accp[VW];
#pragma unroll
        for (int k = 0; k < MP_MAXT; ++k) {
            if (k >= P.n) break;
            const SpmmTerms::T &Tm = P.t[k];
            const AT *xb = reinterpret_cast<const AT *>(Tm.x) + (long long)n * Tm.xs + c;
            float acc[VW];
            if (pair && k == P.n - 1) {
#pragma unroll
                for (int u = 0; u < VW; ++u) acc[u] = accp[u];
            } else if (pair && k == 0) {
                if (Tm.ew) cape_gather_row_ell_both<VW, AT>(xb, Tm.ldx, mb, words, sh, Tm.ci, Tm.va, Tm.ew, r, acc, accp);
                else cape_gather_row_both<VW, U, AT>(xb, Tm.ldx, mb, words, sh, Tm.rp, Tm.ci, Tm.va, r, acc, accp);
            } else if ((masked >> k) & 1u) {
                if (Tm.ew) cape_gather_row_ell_masked<VW, AT>(xb, Tm.ldx, mb, words, sh, Tm.ci, Tm.va, Tm.ew, r, acc);
                else cape_gather_row_masked<VW, U, AT>(xb, Tm.ldx, mb, words, sh, Tm.rp, Tm.ci, Tm.va, r, acc);
            } else {
                cape_gather<VW, U>(xb, Tm.ldx, Tm.rp, Tm.ci, Tm.va, Tm.ew, r, acc);
            }
            if (live) cape_stv<VW>(reinterpret_cast<AT *>(Tm.y) + (long long)n * Tm.ys + (long long)r * Tm.ldy + c, acc);
            if (Tm.rm) {
                float m = 0.f;
#pragma unroll
                for (int u = 0; u < VW; ++u) m = fmaxf(m, fabsf(acc[u]));
                m = cape_group_max(m, cq);
                if (live && c == 0) cape_store_rowmax(Tm.rm, (long long)n * Mo + r, m);
            }
#pragma unroll
            for (int u = 0; u < VW; ++u) racc[k][u] += live ? acc[u] : 0.f;
        }
    }
    if (!part) return;
    // column sums over the block's rows (see bwd_prep_spmm_kernel): DPP rotations / lane swaps inside the wave, the four waves
    // through LDS in a fixed order, all terms behind one barrier
    __shared__ float cs[4][MP_MAXT][64 * VW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *pp = part + ((long long)n * chunks + t) * T * C;
#pragma unroll
    for (int k = 0; k < MP_MAXT; ++k) {
        if (k >= P.n) break;
#pragma unroll
        for (int u = 0; u < VW; ++u) {
            float v = racc[k][u];
            if (cq <= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
            if (cq <= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
            if (cq <= 16) v = cape_sum_xor16(v);
            if (cq <= 32) v = cape_sum_xor32(v);
            racc[k][u] = v;
        }
        if (lane < cq)
#pragma unroll
            for (int u = 0; u < VW; ++u) cs[wave][k][lane * VW + u] = racc[k][u];
    }
    __syncthreads();
    if ((int)threadIdx.x < cq) {
#pragma unroll
        for (int k = 0; k < MP_MAXT; ++k) {
            if (k >= P.n) break;
            float s4[VW];
#pragma unroll
            for (int u = 0; u < VW; ++u)
                s4[u] = ((cs[0][k][threadIdx.x * VW + u] + cs[1][k][threadIdx.x * VW + u]) + cs[2][k][threadIdx.x * VW + u]) + cs[3][k][threadIdx.x * VW + u];
            cape_stv<VW>(pp + (long long)(1 + k) * C + (int)threadIdx.x * VW, s4);
        }
    }
}

// ---- operator applications AFTER the dense contraction (up-sampling layers) ----------------------------------
// (S_k x) W_k = S_k (x W_k): on an up-sampling layer the contraction runs on the coarse input rows (half the
// GEMM work) and this kernel applies the operators to the F-channel products Z_k, adds the rank-1 condition terms
// and performs the layer epilogue:   single:  y = act(acc1 + bias)      dual:  y = relu(acc1) + acc2, sign bits out
//   acc1 = sum_{k not in to2} S_k Z_k + sum_{j not in rank_to2} rowscale_j[r] coef[n,j,f];   acc2 likewise.
struct CombineParams {
    SpmmTerms P;
    unsigned to2;
    int rankR;
    const float *rowscale;
    const float *coef;
    unsigned rank_to2;
    const float *bias;
    int bias_mode, act, dual;
    unsigned *mask;
    int mask_words;
    float *rm;                                      // row bounds of y or null
};

template <int VW, typename T = float, int U = 0>
__global__ __launch_bounds__(256) void spmm_combine_kernel(CombineParams Q, ViewT<T> y, int N, int Mo, int F) {
    const int cq = (F + VW - 1) / VW;
    int n, t;
    cape_map_block(blockIdx.x, N, spmm_bps(Mo, cq), n, t);      // see spmm_kernel
    const int total = Mo * cq;                              // work items of one sample (< 2^31, checked by the host entry)
    const int i = t * 256 + (int)threadIdx.x;
    const bool live = i < total;                            // whole blocks stay alive: the mask shuffles need every lane
    const int ii = live ? i : total - 1;
    const int r = ii / cq;
    const int q = ii - r * cq;
    const int c = q * VW;
    float a1[VW], a2[VW];
#pragma unroll
    for (int u = 0; u < VW; ++u) a1[u] = a2[u] = 0.f;
    for (int k = 0; k < Q.P.n; ++k) {
        const SpmmTerms::T &Tm = Q.P.t[k];
        const T *xb = reinterpret_cast<const T *>(Tm.x) + (long long)n * Tm.xs + c;
        float acc[VW];
        if (!Tm.rp) cape_ldv<VW>(xb + (long long)r * Tm.ldx, acc);
        else cape_gather<VW, U>(xb, Tm.ldx, Tm.rp, Tm.ci, Tm.va, Tm.ew, r, acc);
        if ((Q.to2 >> k) & 1u) {
#pragma unroll
            for (int u = 0; u < VW; ++u) a2[u] = fmaf(Tm.scale, acc[u], a2[u]);
        } else {
#pragma unroll
            for (int u = 0; u < VW; ++u) a1[u] = fmaf(Tm.scale, acc[u], a1[u]);
        }
    }
    for (int j = 0; j < Q.rankR; ++j) {
        const float rs = Q.rowscale[(long long)j * Mo + r];
        const float *cf = Q.coef + ((long long)n * Q.rankR + j) * F + c;
        if ((Q.rank_to2 >> j) & 1u) {
#pragma unroll
            for (int u = 0; u < VW; ++u) a2[u] = fmaf(rs, cf[u], a2[u]);
        } else {
#pragma unroll
            for (int u = 0; u < VW; ++u) a1[u] = fmaf(rs, cf[u], a1[u]);
        }
    }
    float o[VW];
    unsigned bits = 0;
#pragma unroll
    for (int u = 0; u < VW; ++u) {
        float v = a1[u];
        if (Q.dual) {
            if (v > 0.f) bits |= 1u << u;
            v = (v > 0.f ? v : 0.f) + a2[u];
        } else {
            if (Q.bias_mode == CAPE_BIAS_CHANNEL) v += Q.bias[c + u];
            else if (Q.bias_mode == CAPE_BIAS_VERTEX) v += Q.bias[(long long)r * F + c + u];
            v = cape_act(v, Q.act);
        }
        o[u] = v;
    }
    if (Q.mask) {
        // VW >= 4 only, F % 32 == 0: the 32 / VW lanes of one 32-channel word are consecutive and aligned
        constexpr int LW = VW >= 4 ? 32 / VW : 1;
        unsigned w = live ? (bits << (VW * (q & (LW - 1)))) : 0u;
#pragma unroll
        for (int s = 1; s < LW; s <<= 1) w |= __shfl_xor(w, s);
        if (live && (q & (LW - 1)) == 0) Q.mask[((long long)n * Mo + r) * Q.mask_words + (q / LW)] = w;
    }
    if (live) cape_stv<VW>(y.p + (long long)n * y.ss + (long long)r * y.ld + c, o);
    if (Q.rm) {
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < VW; ++u) m = fmaxf(m, fabsf(o[u]));
        m = cape_group_max(m, cq);                          // (dead lanes repeat the last row: same value, no harm)
        if (live && c == 0) cape_store_rowmax(Q.rm, (long long)n * Mo + r, m);
    }
}

// ---- bias + activation ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_act_kernel(CView x, const float *bias, int bias_mode, int act,
                                                       View y, int N, int M, int C) {
    const long long total = (long long)N * M * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long nm = i / C;
        const int m = (int)(nm % M);
        const int n = (int)(nm / M);
        float v = x.p[(long long)n * x.ss + (long long)m * x.ld + c];
        if (bias_mode == CAPE_BIAS_CHANNEL) v += bias[c];
        else if (bias_mode == CAPE_BIAS_VERTEX) v += bias[(long long)m * C + c];
        y.p[(long long)n * y.ss + (long long)m * y.ld + c] = cape_act(v, act);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(CView dy, CView y, int act, View dz, int N, int M, int C) {
    const int W = VEC ? 4 : 1;
    const int cq = C / W;
    const long long total = (long long)N * M * cq;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cq) * W;
        const long long nm = i / cq;
        const int m = (int)(nm % M);
        const int n = (int)(nm / M);
        if (VEC) {
            const float4 g = *reinterpret_cast<const float4 *>(dy.p + (long long)n * dy.ss + (long long)m * dy.ld + c);
            const float4 o = *reinterpret_cast<const float4 *>(y.p + (long long)n * y.ss + (long long)m * y.ld + c);
            float4 r;
            r.x = g.x * cape_act_grad_from_out(o.x, act);
            r.y = g.y * cape_act_grad_from_out(o.y, act);
            r.z = g.z * cape_act_grad_from_out(o.z, act);
            r.w = g.w * cape_act_grad_from_out(o.w, act);
            *reinterpret_cast<float4 *>(dz.p + (long long)n * dz.ss + (long long)m * dz.ld + c) = r;
        } else {
            const float g = dy.p[(long long)n * dy.ss + (long long)m * dy.ld + c];
            const float o = y.p[(long long)n * y.ss + (long long)m * y.ld + c];
            dz.p[(long long)n * dz.ss + (long long)m * dz.ld + c] = g * cape_act_grad_from_out(o, act);
        }
    }
}

__global__ __launch_bounds__(256) void mask_mul_kernel(CView dy, const unsigned *mask, View dz, int N, int M, int F) {
    const int words = (F + 31) / 32;
    const long long total = (long long)N * M * F;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int f = (int)(i % F);
        const long long nm = i / F;
        const int m = (int)(nm % M);
        const int n = (int)(nm / M);
        const unsigned w = mask[nm * words + (f >> 5)];
        const float g = dy.p[(long long)n * dy.ss + (long long)m * dy.ld + f];
        dz.p[(long long)n * dz.ss + (long long)m * dz.ld + f] = ((w >> (f & 31)) & 1u) ? g : 0.f;
    }
}

// ---- column sums -----------------------------------------------------------------------------
// stage 1: block b sums rows [b*RB, (b+1)*RB) of the flattened (n,m) row space -> part[b][c]
constexpr int COLSUM_RB = 128;

// aligned fast path (C % 4 == 0): thread = (float4 column, row lane); column groups of <= 256 floats
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(CView x, int N, int M, int C, float *part) {
    __shared__ float4 red[256];
    const long long R = (long long)N * M;
    const long long ra = (long long)blockIdx.x * COLSUM_RB;
    const long long rb = (ra + COLSUM_RB < R) ? ra + COLSUM_RB : R;
    for (int cbase = 0; cbase < C; cbase += 256) {
        const int cw = min(256, C - cbase);
        const int c4n = cw >> 2;
        const int lanes = 256 / c4n;               // row lanes
        const int q = threadIdx.x % c4n, rl = threadIdx.x / c4n;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < lanes)
            for (long long r = ra + rl; r < rb; r += lanes) {
                const long long n = r / M, m = r % M;
                const float4 v = *reinterpret_cast<const float4 *>(x.p + n * x.ss + m * x.ld + cbase + 4 * q);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        red[threadIdx.x] = s;
        __syncthreads();
        if (rl == 0) {
            for (int l = 1; l < lanes; ++l) {
                const float4 v = red[l * c4n + q];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4 *>(part + (long long)blockIdx.x * C + cbase + 4 * q) = s;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(CView x, int N, int M, int C, float *part) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const long long R = (long long)N * M;
    const long long ra = (long long)blockIdx.x * COLSUM_RB;
    const long long rb = (ra + COLSUM_RB < R) ? ra + COLSUM_RB : R;
    for (int cbase = 0; cbase < C; cbase += 64) {
        const int c = cbase + cl;
        float s = 0.f;
        if (c < C)
            for (long long r = ra + rl; r < rb; r += 4) {
                const long long n = r / M, m = r % M;
                s += x.p[n * x.ss + m * x.ld + c];
            }
        red[rl][cl] = s;
        __syncthreads();
        if (rl == 0 && c < C) part[(long long)blockIdx.x * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        __syncthreads();
    }
}

// stage 2: block = 64 columns x 4 partial lanes
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *part, int nblk, int C, int accumulate, float *out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (int b = pl; b < nblk; b += 4) s += part[(long long)b * C + c];
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < C) {
        const float t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

template <typename T = float>
__global__ __launch_bounds__(256) void sum_over_samples_kernel(CViewT<T> x, int N, int M, int C, int accumulate, float *out) {
    const long long total = (long long)M * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long m = i / C;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += cape_ld(x.p + (long long)n * x.ss + m * x.ld + c);
        out[i] = accumulate ? out[i] + s : s;
    }
}

// ---- condition channels ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_cond_kernel(const float *cond, int ldc, const float *scale, View y,
                                                        int N, int M, int C) {
    const long long total = (long long)N * M * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long nm = i / C;
        const int m = (int)(nm % M);
        const int n = (int)(nm / M);
        float v = cond[(long long)n * ldc + c];
        if (scale) v *= scale[m];
        y.p[(long long)n * y.ss + (long long)m * y.ld + c] = v;
    }
}

// dcond[n,c] (+)= sum_m scale[m] * dy[n,m,c]; one block per (n, 64-channel group)
__global__ __launch_bounds__(256) void reduce_cond_kernel(CView dy, const float *scale, float *dcond, int ldc,
                                                          int N, int M, int C, int accumulate) {
    __shared__ float red[4][64];
    const int cgroups = (C + 63) / 64;
    const int n = blockIdx.x / cgroups, cg = blockIdx.x % cgroups;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = cg * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (int m = rl; m < M; m += 4) {
            const float v = dy.p[(long long)n * dy.ss + (long long)m * dy.ld + c];
            s = scale ? fmaf(scale[m], v, s) : s + v;
        }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        const float t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        float *d = dcond + (long long)n * ldc + c;
        *d = accumulate ? *d + t : t;
    }
}

// float4 form: thread = (channel quad, row lane), 256 / quads row lanes, two independent partial sums per lane --
// the scalar form above walks all M rows with four row lanes per block (208 us at M = 6890 for 32 condition channels)
__global__ __launch_bounds__(256) void reduce_cond_vec_kernel(CView dy, const float *scale, float *dcond, int ldc,
                                                              int N, int M, int C, int accumulate) {
    __shared__ float4 red[256];
    const int cgroups = (C + 63) / 64;
    const int n = blockIdx.x / cgroups, cg = blockIdx.x % cgroups;
    const int gw = min(64, C - cg * 64);                // channels of this block's group (C % 4 == 0)
    int qp = 1;
    while (qp < gw / 4) qp <<= 1;                       // quad lanes per row (power of two)
    const int RL = 256 / qp;
    const int q = threadIdx.x % qp, rl = threadIdx.x / qp;
    const int c = cg * 64 + 4 * q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (4 * q < gw && c < C) {
        const float *p = dy.p + (long long)n * dy.ss + c;
        int m = rl;
        for (; m + RL < M; m += 2 * RL) {
            const float4 v0 = *reinterpret_cast<const float4 *>(p + (long long)m * dy.ld);
            const float4 v1 = *reinterpret_cast<const float4 *>(p + (long long)(m + RL) * dy.ld);
            const float s0 = scale ? scale[m] : 1.f, s1 = scale ? scale[m + RL] : 1.f;
            a.x = fmaf(s0, v0.x, a.x); a.y = fmaf(s0, v0.y, a.y); a.z = fmaf(s0, v0.z, a.z); a.w = fmaf(s0, v0.w, a.w);
            b.x = fmaf(s1, v1.x, b.x); b.y = fmaf(s1, v1.y, b.y); b.z = fmaf(s1, v1.z, b.z); b.w = fmaf(s1, v1.w, b.w);
        }
        if (m < M) {
            const float4 v0 = *reinterpret_cast<const float4 *>(p + (long long)m * dy.ld);
            const float s0 = scale ? scale[m] : 1.f;
            a.x = fmaf(s0, v0.x, a.x); a.y = fmaf(s0, v0.y, a.y); a.z = fmaf(s0, v0.z, a.z); a.w = fmaf(s0, v0.w, a.w);
        }
    }
    red[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    __syncthreads();
    if (threadIdx.x < qp && 4 * threadIdx.x < gw && cg * 64 + 4 * (int)threadIdx.x < C) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < RL; ++l) {
            const float4 v = red[l * qp + threadIdx.x];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        float *d = dcond + (long long)n * ldc + cg * 64 + 4 * threadIdx.x;
        if (accumulate) { t.x += d[0]; t.y += d[1]; t.z += d[2]; t.w += d[3]; }
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    }
}

// ---- out[n, j, f] = sum_r rowscale[j, r] * dz[n, r, f]   (gradient of the rank-1 condition terms)
constexpr int RSR_RB = 32;    // rows per block
constexpr int RSR_MAXR = 4;

// stage 1: block = (sample n, row chunk), thread = (column, row lane); part[n][chunk][j][f]
__global__ __launch_bounds__(256) void rowscale_partial_kernel(CView dz, const float *rowscale, int R, int N, int Mo, int F,
                                                               float *part, int chunks) {
    __shared__ float red[RSR_MAXR][256];
    const int n = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int ra = ch * RSR_RB, rb = min(Mo, ra + RSR_RB);
    for (int fbase = 0; fbase < F; fbase += 64) {
        const int fl = threadIdx.x & 63, rl = threadIdx.x >> 6;
        const int f = fbase + fl;
        float acc[RSR_MAXR] = {0.f, 0.f, 0.f, 0.f};
        if (f < F)
            for (int r = ra + rl; r < rb; r += 4) {
                const float v = dz.p[(long long)n * dz.ss + (long long)r * dz.ld + f];
#pragma unroll
                for (int j = 0; j < RSR_MAXR; ++j)
                    if (j < R) acc[j] = fmaf(rowscale[(long long)j * Mo + r], v, acc[j]);
            }
#pragma unroll
        for (int j = 0; j < RSR_MAXR; ++j) red[j][threadIdx.x] = acc[j];
        __syncthreads();
        if (rl == 0 && f < F) {
#pragma unroll
            for (int j = 0; j < RSR_MAXR; ++j)
                if (j < R)
                    part[(((long long)n * chunks + ch) * R + j) * F + f] =
                        (red[j][fl] + red[j][64 + fl]) + (red[j][128 + fl] + red[j][192 + fl]);
        }
        __syncthreads();
    }
}

// stage 2: out[n][j][f] = sum_chunk part[n][chunk][j][f]; block = 64 (j,f) elements x 4 chunk lanes
__global__ __launch_bounds__(256) void rowscale_final_kernel(const float *part, int chunks, int RF, int N, float *out) {
    __shared__ float red[4][64];
    const int el = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const int blocks_per_n = (RF + 63) / 64;
    const int n = blockIdx.x / blocks_per_n;
    const int e = (blockIdx.x % blocks_per_n) * 64 + el;
    float s = 0.f;
    if (e < RF)
        for (int c = cl; c < chunks; c += 4) s += part[((long long)n * chunks + c) * RF + e];
    red[cl][el] = s;
    __syncthreads();
    if (cl == 0 && e < RF) out[(long long)n * RF + e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

// ---- fused backward preparation: dz, bias gradient and rank-1 term gradients in one pass over g ----
constexpr int BP_RB_MAX = 128;              // rows per block (upper bound; halved until the grid has BP_BLOCKS blocks)
#ifndef CAPE_BP_BLOCKS_DEFAULT
#define CAPE_BP_BLOCKS_DEFAULT 448
#endif
#ifndef CAPE_BP_UNROLL_DEFAULT
#define CAPE_BP_UNROLL_DEFAULT 2           // rows loaded ahead per thread in bwd_prep_vec_kernel (compile-time: 1, 2, 4)
#endif
#ifndef CAPE_BP_RB_MIN_DEFAULT
#define CAPE_BP_RB_MIN_DEFAULT 16
#endif
inline int bp_rows(int N, int Mo) {
    constexpr int want = CAPE_BP_BLOCKS_DEFAULT, rbmin = CAPE_BP_RB_MIN_DEFAULT;      // (measured in round 2; no run-time knob)
    int rb = BP_RB_MAX;
    while (rb > rbmin && (long long)N * ((Mo + rb - 1) / rb) < want) rb >>= 1;
    return rb;
}
constexpr int BP_MAXT = RSR_MAXR + 2;       // reduction terms: [0]=sum dz, [1..R]=rowscale_j*dz, [R+1]=rowscale_rg*g

template <typename AT = float>
__global__ __launch_bounds__(256) void bwd_prep_kernel(CViewT<AT> g, CViewT<AT> y, int act, const unsigned *mask, ViewT<AT> dz,
                                                       const float *rowscale, int R, int rg, int want_bias, int want_g,
                                                       int N, int Mo, int F, float *part, int chunks, int RB) {
    __shared__ float red[BP_MAXT][256];
    const int n = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int ra = ch * RB, rb = min(Mo, ra + RB);
    const int words = (F + 31) / 32;
    const int T = R + 2;
    const int fl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    for (int fbase = 0; fbase < F; fbase += 64) {
        const int f = fbase + fl;
        float acc[BP_MAXT];
#pragma unroll
        for (int j = 0; j < BP_MAXT; ++j) acc[j] = 0.f;
        if (f < F)
            for (int r = ra + rl; r < rb; r += 4) {
                const float gv = cape_ld(g.p + (long long)n * g.ss + (long long)r * g.ld + f);
                float d;
                if (mask) d = ((mask[((long long)n * Mo + r) * words + (f >> 5)] >> (f & 31)) & 1u) ? gv : 0.f;
                else if (act != CAPE_ACT_NONE) d = gv * cape_act_grad_from_out(cape_ld(y.p + (long long)n * y.ss + (long long)r * y.ld + f), act);
                else d = gv;
                if (!((const void *)dz.p == (const void *)g.p && act == CAPE_ACT_NONE && !mask)) cape_st(dz.p + (long long)n * dz.ss + (long long)r * dz.ld + f, d);
                acc[0] += d;
#pragma unroll
                for (int j = 0; j < RSR_MAXR; ++j)
                    if (j < R) acc[1 + j] = fmaf(rowscale[(long long)j * Mo + r], d, acc[1 + j]);
                if (want_g) acc[BP_MAXT - 1] = fmaf(rowscale[(long long)rg * Mo + r], gv, acc[BP_MAXT - 1]);
            }
#pragma unroll
        for (int j = 0; j < BP_MAXT; ++j) red[j][threadIdx.x] = acc[j];
        __syncthreads();
        if (rl == 0 && f < F) {
            float *pp = part + ((long long)n * chunks + ch) * T * F;
            if (want_bias) pp[f] = (red[0][fl] + red[0][64 + fl]) + (red[0][128 + fl] + red[0][192 + fl]);
#pragma unroll
            for (int j = 0; j < RSR_MAXR; ++j)
                if (j < R) pp[(1 + j) * F + f] = (red[1 + j][fl] + red[1 + j][64 + fl]) + (red[1 + j][128 + fl] + red[1 + j][192 + fl]);
            if (want_g) pp[(R + 1) * F + f] = (red[BP_MAXT - 1][fl] + red[BP_MAXT - 1][64 + fl]) +
                                              (red[BP_MAXT - 1][128 + fl] + red[BP_MAXT - 1][192 + fl]);
        }
        __syncthreads();
    }
}

// narrow variant (F <= 4: the 3-channel output layer): thread = ROW, its F columns in registers.  The column-lane mapping
// above keeps 3 of 64 lanes busy there, each walking its rows one dependent access at a time (24 us for 2 MB).
template <typename AT = float>
__global__ __launch_bounds__(256) void bwd_prep_narrow_kernel(CViewT<AT> g, CViewT<AT> y, int act, const unsigned *mask, ViewT<AT> dz,
                                                              const float *rowscale, int R, int rg, int want_bias, int want_g,
                                                              int N, int Mo, int F, float *part, int chunks, int RB) {
    __shared__ float red[4][BP_MAXT * 4];
    const int n = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int ra = ch * RB, rb = min(Mo, ra + RB);
    const int T = R + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[BP_MAXT][4];
#pragma unroll
    for (int j = 0; j < BP_MAXT; ++j)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[j][f] = 0.f;
    for (int r = ra + (int)threadIdx.x; r < rb; r += 256) {
        float gv[4], rs[RSR_MAXR], rsg = 0.f;
        const unsigned mw = mask ? mask[(long long)n * Mo + r] : 0u;            // F <= 4: one sign word per row
#pragma unroll
        for (int f = 0; f < 4; ++f) gv[f] = f < F ? cape_ld(g.p + (long long)n * g.ss + (long long)r * g.ld + f) : 0.f;
#pragma unroll
        for (int j = 0; j < RSR_MAXR; ++j) rs[j] = j < R ? rowscale[(long long)j * Mo + r] : 0.f;
        if (want_g) rsg = rowscale[(long long)rg * Mo + r];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f >= F) continue;
            float d;
            if (mask) d = ((mw >> f) & 1u) ? gv[f] : 0.f;
            else if (act != CAPE_ACT_NONE) d = gv[f] * cape_act_grad_from_out(cape_ld(y.p + (long long)n * y.ss + (long long)r * y.ld + f), act);
            else d = gv[f];
            if (!((const void *)dz.p == (const void *)g.p && act == CAPE_ACT_NONE && !mask)) cape_st(dz.p + (long long)n * dz.ss + (long long)r * dz.ld + f, d);
            acc[0][f] += d;
#pragma unroll
            for (int j = 0; j < RSR_MAXR; ++j) acc[1 + j][f] = fmaf(rs[j], d, acc[1 + j][f]);
            acc[BP_MAXT - 1][f] = fmaf(rsg, gv[f], acc[BP_MAXT - 1][f]);
        }
    }
    // fixed-order sums: the lanes of a wave (xor butterflies), then the four waves
#pragma unroll
    for (int j = 0; j < BP_MAXT; ++j)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            float v = acc[j][f];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[wave][j * 4 + f] = v;
        }
    __syncthreads();
    if (threadIdx.x < BP_MAXT * 4) {
        const int j = threadIdx.x >> 2, f = threadIdx.x & 3;
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        float *pp = part + ((long long)n * chunks + ch) * T * F;
        if (f < F) {
            if (j == 0) { if (want_bias) pp[f] = t; }
            else if (j == BP_MAXT - 1) { if (want_g) pp[(R + 1) * F + f] = t; }
            else if (j - 1 < R) pp[j * F + f] = t;
        }
    }
}

// vector variant (F % VW == 0, aligned views): thread = (column group of VW channels, row lane).  The rows of a lane are
// taken UR at a time: the UR gradient rows (and their sign words / outputs) are all loaded before the first is used, so a
// thread waits for one memory round trip per UR rows instead of one per row.
template <typename AT, int VW, int UR>
__global__ __launch_bounds__(256) void bwd_prep_vec_kernel(CViewT<AT> g, CViewT<AT> y, int act, const unsigned *mask, ViewT<AT> dz,
                                                           const float *rowscale, int R, int rg, int want_bias, int want_g,
                                                           int N, int Mo, int F, float *part, int chunks, int RB, float *rm) {
    __shared__ float red[VW][256];
    const int n = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int ra = ch * RB, rb = min(Mo, ra + RB);
    const int words = (F + 31) / 32;
    const int T = R + 2;
    constexpr int FB = 64 * VW;                 // channels per column pass (256 threads x VW >= FB)
    float *pp = part + ((long long)n * chunks + ch) * T * F;
    const AT *gb = g.p + (long long)n * g.ss;
    const AT *yb = y.p ? y.p + (long long)n * y.ss : nullptr;
    AT *zb = dz.p + (long long)n * dz.ss;
    const bool wr = !((const void *)dz.p == (const void *)g.p && act == CAPE_ACT_NONE && !mask);      // dz == g unchanged: the caller wants the sums only
    const unsigned *mb = mask ? mask + (long long)n * Mo * words : nullptr;
    for (int fbase = 0; fbase < F; fbase += FB) {
        const int fw = min(FB, F - fbase);
        const int cvn = fw / VW;                // column groups of this pass (divides 256)
        const int lanes = 256 / cvn;
        const int q = threadIdx.x % cvn, rl = threadIdx.x / cvn;
        const int f = fbase + VW * q;
        float acc[BP_MAXT][VW];
#pragma unroll
        for (int j = 0; j < BP_MAXT; ++j)
#pragma unroll
            for (int u = 0; u < VW; ++u) acc[j][u] = 0.f;
        for (int r0 = ra + rl; r0 < rb; r0 += UR * lanes) {
            float gv[UR][VW], ov[UR][VW];
            unsigned mw[UR];
#pragma unroll
            for (int k = 0; k < UR; ++k) {
                const int r = r0 + k * lanes;
                const int rr = r < rb ? r : r0;             // rows past the chunk re-read the first (result unused)
                cape_ldv<VW>(gb + (long long)rr * g.ld + f, gv[k]);
                if (mb) mw[k] = mb[(long long)rr * words + (f >> 5)] >> (f & 31);
                else if (act != CAPE_ACT_NONE) cape_ldv<VW>(yb + (long long)rr * y.ld + f, ov[k]);
            }
#pragma unroll
            for (int k = 0; k < UR; ++k) {
                const int r = r0 + k * lanes;
                if (r < rb) {
                    float d[VW];
#pragma unroll
                    for (int u = 0; u < VW; ++u) {
                        if (mb) d[u] = ((mw[k] >> u) & 1u) ? gv[k][u] : 0.f;
                        else if (act != CAPE_ACT_NONE) d[u] = gv[k][u] * cape_act_grad_from_out(ov[k][u], act);
                        else d[u] = gv[k][u];
                    }
                    if (wr) cape_stv<VW>(zb + (long long)r * dz.ld + f, d);
                    if (rm) {
                        // row bound of g -- and therefore of dz (|dz| <= |g| element by element): the affine block's backward
                        // contracts BOTH, one reduction serves the two.  Entry = column pass (F <= 4 * FB); the cvn lanes of a row
                        // are consecutive and aligned
                        float m = 0.f;
#pragma unroll
                        for (int u = 0; u < VW; ++u) m = fmaxf(m, fabsf(gv[k][u]));
                        m = cape_group_max(m, cvn);
                        if (q == 0) {
                            // (each pass writes only its own entry -- different threads own a row in different passes --
                            // and pass 0 zeroes the entries no pass owns)
                            float *dst = rm + 4 * ((long long)n * Mo + r);
                            dst[fbase / FB] = m;
                            if (fbase == 0)
                                for (int e = (F + FB - 1) / FB; e < 4; ++e) dst[e] = 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < VW; ++u) acc[0][u] += d[u];
#pragma unroll
                    for (int j = 0; j < RSR_MAXR; ++j)
                        if (j < R) {
                            const float sv = rowscale[(long long)j * Mo + r];
#pragma unroll
                            for (int u = 0; u < VW; ++u) acc[1 + j][u] = fmaf(sv, d[u], acc[1 + j][u]);
                        }
                    if (want_g) {
                        const float sv = rowscale[(long long)rg * Mo + r];
#pragma unroll
                        for (int u = 0; u < VW; ++u) acc[BP_MAXT - 1][u] = fmaf(sv, gv[k][u], acc[BP_MAXT - 1][u]);
                    }
                }
            }
        }
        // reduce the row lanes term by term: first inside each wave (the lanes of one column group are cvn apart: xor
        // shuffles over cvn, 2 cvn, ... 32), then the <= 4 per-wave (or per-lane, cvn >= 64) sums through LDS
        const int le = cvn < 64 ? 4 : lanes;                   // partial sums per column group that reach LDS
        const bool writer = cvn < 64 ? (int)(threadIdx.x & 63) < cvn : true;
        const int slot = cvn < 64 ? (int)(threadIdx.x >> 6) * cvn + q : (int)threadIdx.x;
#pragma unroll
        for (int j = 0; j < BP_MAXT; ++j) {
            const bool used = (j == 0 && want_bias) || (j >= 1 && j <= R) || (j == BP_MAXT - 1 && want_g);
            if (!used) continue;
            float v[VW];
#pragma unroll
            for (int u = 0; u < VW; ++u) v[u] = acc[j][u];
            for (int sh = cvn; sh < 64; sh <<= 1)
#pragma unroll
                for (int u = 0; u < VW; ++u) v[u] += __shfl_xor(v[u], sh);
            if (writer)
#pragma unroll
                for (int u = 0; u < VW; ++u) red[u][slot] = v[u];
            __syncthreads();
            if ((int)threadIdx.x < cvn) {
                float t[VW];
#pragma unroll
                for (int u = 0; u < VW; ++u) t[u] = red[u][q];
                for (int l = 1; l < le; ++l)
#pragma unroll
                    for (int u = 0; u < VW; ++u) t[u] += red[u][l * cvn + q];
                const int trow = (j == BP_MAXT - 1) ? (R + 1) : j;
                cape_stv<VW>(pp + (long long)trow * F + f, t);
            }
            __syncthreads();
        }
    }
}

// stage 2: block = (term t, 16 columns) x 16 lanes.  t = 0: dbias[f] = sum over (n, chunk);
// t = 1..R: dcoef[n, t-1, f] = sum over chunks; t = R+1: dcoef_g[n, f] = sum over chunks.
__device__ __forceinline__ void bwd_prep_final_block(int bid, const float *part, int chunks, int N, int F, int R, float *dbias,
                                                     float *dcoef, float *dcoef_g, long long cs, long long cgs) {
    __shared__ float red[16][17];
    const int T = R + 2;
    const int fblocks = (F + 15) / 16;
    const int fl = threadIdx.x & 15, ln = threadIdx.x >> 4;
    int b = bid;
    const int fb = b % fblocks; b /= fblocks;
    const int f = fb * 16 + fl;
    // block order: [bias blocks: fblocks] then for each n: (R+1) * fblocks
    float s = 0.f;
    float *dst = nullptr;
    if (b == 0) {            // bias
        if (dbias && f < F) {
            // sixteen independent partial sums: the chain of dependent L2 loads, not bandwidth, bounds this loop (the fused
            // activation-gradient form leaves one partial per spmm_multi workgroup: 16 x 216 per column in the benchmarked model,
            // 54 round trips with four sums)
            const long long total = (long long)N * chunks;
            const long long TF = (long long)T * F;
            constexpr int NS = 16;
            float ps[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) ps[k] = 0.f;
            long long i = ln;
            for (; i + 16 * (NS - 1) < total; i += 16 * NS) {
#pragma unroll
                for (int k = 0; k < NS; ++k) ps[k] += part[(i + 16 * k) * TF + f];
            }
            for (; i < total; i += 16) s += part[i * TF + f];
#pragma unroll
            for (int k = 0; k < NS; k += 4) s += (ps[k] + ps[k + 1]) + (ps[k + 2] + ps[k + 3]);
            dst = dbias + f;
        }
    } else {
        const int idx = b - 1;
        const int n = idx / (R + 1), t = 1 + idx % (R + 1);
        float *out = (t <= R) ? (dcoef ? dcoef + (long long)n * cs + (long long)(t - 1) * F : nullptr)
                              : (dcoef_g ? dcoef_g + (long long)n * cgs : nullptr);
        if (out && f < F) {
            for (int c = ln; c < chunks; c += 16) s += part[(((long long)n * chunks + c) * T + t) * F + f];
            dst = out + f;
        }
    }
    red[ln][fl] = s;
    __syncthreads();
    if (ln == 0 && dst) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[l][fl];
        *dst = t;
    }
}

__global__ __launch_bounds__(256) void bwd_prep_final_kernel(const float *part, int chunks, int N, int F, int R, float *dbias,
                                                             float *dcoef, float *dcoef_g, long long cs, long long cgs) {
    bwd_prep_final_block(blockIdx.x, part, chunks, N, F, R, dbias, dcoef, dcoef_g, cs, cgs);
}

// the final stage of SEVERAL bwd_prep calls in one launch (their partial slabs stay in their workspaces until then):
// the per-layer finals are 5 us dispatches whose results are only needed at the end of the backward pass
struct BpFinalBatch {
    struct I {
        const float *part;
        int chunks, N, F, R;
        float *dbias, *dcoef, *dcoef_g;
        long long cs, cgs;
    } it[CAPE_MAX_BWD_PREP_ITEMS];
    int blk_off[CAPE_MAX_BWD_PREP_ITEMS + 1];
    int n;
};

__global__ __launch_bounds__(256) void bwd_prep_final_batch_kernel(BpFinalBatch B) {
    int i = 0;
    while (i + 1 < B.n && (int)blockIdx.x >= B.blk_off[i + 1]) ++i;
    const BpFinalBatch::I &I = B.it[i];
    bwd_prep_final_block((int)blockIdx.x - B.blk_off[i], I.part, I.chunks, I.N, I.F, I.R, I.dbias, I.dcoef, I.dcoef_g, I.cs, I.cgs);
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

namespace {
// ELL operands: width 4, 8 or 12, 16-byte aligned arrays (0 = the operator is given in CSR form)
inline bool ell_ok(int ew, const void *ec, const void *ev) {
    if (ew == 0) return true;
    return (ew == 4 || ew == 8 || ew == 12) && ec && ev && ((reinterpret_cast<uintptr_t>(ec) | reinterpret_cast<uintptr_t>(ev)) & 15) == 0;
}

// Row bounds of an output (cape_h2_src_t.rowmax, width 4): written by the kernel itself when the lanes of a row form one aligned
// power-of-two group inside a wave, otherwise by one standalone pass over the finished output (csrc/pieces.hip).
inline bool rm_fused(int lanes_per_row) { return lanes_per_row >= 1 && lanes_per_row <= 64 && (lanes_per_row & (lanes_per_row - 1)) == 0; }
template <typename T>
inline int rm_standalone(const T *y, int64_t ys, int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream) {
    if (sizeof(T) != 4) return CAPE_EINVAL;
    return cape_rowmax(reinterpret_cast<const float *>(y), ys, ldy, N, Mo, C, rowmax_out, 4, stream);
}

template <typename T>
int spmm_impl(const T *x, int64_t x_sample_stride, int32_t ldx, const int32_t *rowptr,
              const int32_t *colidx, const float *vals, int32_t max_row_nnz, int32_t ell_width, float alpha, const T *z,
              int64_t z_sample_stride, int32_t ldz, float beta, T *y, int64_t y_sample_stride,
              int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream) {
    if (!x || !rowptr || !colidx || !vals || !y || N < 1 || Mo < 1 || C < 1 || ldx < C || ldy < C) return CAPE_EINVAL;
    if (rowmax_out && sizeof(T) != 4) return CAPE_EINVAL;
    if (!ell_ok(ell_width, colidx, vals)) return CAPE_EINVAL;
    if (z && ldz < C) return CAPE_EINVAL;
    if ((long long)Mo * C >= (1LL << 31)) return CAPE_EINVAL;       // 32-bit work-item index per sample
    constexpr int es = (int)sizeof(T);
    CViewT<T> xv{x, x_sample_stride, ldx}, zv{z, z_sample_stride, ldz};
    ViewT<T> yv{y, y_sample_stride, ldy};
    const bool vec = aligned4(x, x_sample_stride, ldx, C, es) && aligned4(y, y_sample_stride, ldy, C, es) &&
                     (!z || aligned4(z, z_sample_stride, ldz, C, es));
    hipStream_t st = (hipStream_t)stream;
    if (vec) {
        // max_row_nnz is a hint: operators whose rows hold at most 4 entries (the up-/down-sampling matrices) take the
        // 4-wide entry groups
        const bool wide = spmm_wide() && aligned8(x, x_sample_stride, ldx, C, es) && aligned8(y, y_sample_stride, ldy, C, es) &&
                          (!z || aligned8(z, z_sample_stride, ldz, C, es));
        const bool hint4 = max_row_nnz >= 1 && max_row_nnz <= 4;
        float *rm = rm_fused(C / (wide ? 8 : 4)) ? rowmax_out : nullptr;
        CAPE_LAUNCH_SP(spmm_kernel, T, wide, hint4, dim3((unsigned)(N * spmm_bps(Mo, C / (wide ? 8 : 4)))), dim3(256), 0, st, xv, rowptr,
                       colidx, vals, ell_width, alpha, zv, beta, yv, N, Mo, C, rm);
        CAPE_LAUNCH_CHECK();
        if (rowmax_out && !rm) return rm_standalone(y, y_sample_stride, ldy, N, Mo, C, rowmax_out, stream);
        return CAPE_OK;
    } else {
        if (ell_width) return CAPE_EINVAL;                    // the scalar fallback reads CSR only
        CAPE_LAUNCH((spmm_kernel<1, T, 0>), dim3((unsigned)(N * spmm_bps(Mo, C))), dim3(256), 0, st, xv, rowptr, colidx, vals, 0, alpha, zv, beta, yv, N, Mo, C,
                    (float *)nullptr);
    }
    CAPE_LAUNCH_CHECK();
    if (rowmax_out) return rm_standalone(y, y_sample_stride, ldy, N, Mo, C, rowmax_out, stream);
    return CAPE_OK;
}
}  // namespace

extern "C" int cape_spmm(const float *x, int64_t x_sample_stride, int32_t ldx, const int32_t *rowptr,
                         const int32_t *colidx, const float *vals, int32_t max_row_nnz, int32_t ell_width, float alpha,
                         const float *z, int64_t z_sample_stride, int32_t ldz, float beta, float *y, int64_t y_sample_stride,
                         int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream) {
    return spmm_impl<float>(x, x_sample_stride, ldx, rowptr, colidx, vals, max_row_nnz, ell_width, alpha, z, z_sample_stride, ldz, beta, y,
                            y_sample_stride, ldy, N, Mo, C, rowmax_out, stream);
}

extern "C" int cape_spmm_bf16(const void *x, int64_t x_sample_stride, int32_t ldx, const int32_t *rowptr,
                              const int32_t *colidx, const float *vals, int32_t max_row_nnz, int32_t ell_width, float alpha,
                              const void *z, int64_t z_sample_stride, int32_t ldz, float beta, void *y, int64_t y_sample_stride,
                              int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream) {
    return spmm_impl<cape_bf16>((const cape_bf16 *)x, x_sample_stride, ldx, rowptr, colidx, vals, max_row_nnz, ell_width, alpha,
                                (const cape_bf16 *)z, z_sample_stride, ldz, beta, (cape_bf16 *)y, y_sample_stride, ldy, N, Mo, C,
                                rowmax_out, stream);
}

namespace {
template <typename T>
int spmm_multi_impl(const cape_spmm_term_t *terms, int32_t nterms, int32_t sum, T *y, int64_t y_sample_stride,
                    int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream, const SpmmActGrad *ag = nullptr) {
    if (!terms || nterms < 1 || nterms > CAPE_MAX_SPMM_TERMS || N < 1 || Mo < 1 || C < 1) return CAPE_EINVAL;
    if (rowmax_out && (!sum || sizeof(T) != 4)) return CAPE_EINVAL;
    if (sum && (!y || ldy < C)) return CAPE_EINVAL;
    if ((long long)Mo * C >= (1LL << 31)) return CAPE_EINVAL;       // 32-bit work-item index per sample
    constexpr int es = (int)sizeof(T);
    SpmmTerms P;
    P.n = nterms;
    bool vec = !sum || aligned4(y, y_sample_stride, ldy, C, es);
    bool wide = spmm_wide() && (!sum || aligned8(y, y_sample_stride, ldy, C, es));
    bool any_ell = false;
    for (int k = 0; k < nterms; ++k) {
        const cape_spmm_term_t &t = terms[k];
        if (!t.x || t.ldx < C) return CAPE_EINVAL;
        if (t.rowptr && (!t.colidx || !t.vals)) return CAPE_EINVAL;
        if (!sum && (!t.y || t.ldy < C)) return CAPE_EINVAL;
        P.t[k].x = t.x; P.t[k].xs = t.x_sample_stride; P.t[k].ldx = t.ldx;
        P.t[k].rp = t.rowptr; P.t[k].ci = t.colidx; P.t[k].va = t.vals;
        P.t[k].y = t.y; P.t[k].ys = t.y_sample_stride; P.t[k].ldy = t.ldy;
        P.t[k].scale = t.scale;
        P.t[k].ew = t.rowptr ? t.ell_width : 0;
        P.t[k].rm = sum ? nullptr : t.rowmax_out;
        if (t.rowmax_out && (sum || sizeof(T) != 4)) return CAPE_EINVAL;
        if (t.rowptr && !ell_ok(t.ell_width, t.colidx, t.vals)) return CAPE_EINVAL;
        any_ell = any_ell || P.t[k].ew;
        vec = vec && aligned4(t.x, t.x_sample_stride, t.ldx, C, es) && (sum || aligned4(t.y, t.y_sample_stride, t.ldy, C, es));
        wide = wide && aligned8(t.x, t.x_sample_stride, t.ldx, C, es) && (sum || aligned8(t.y, t.y_sample_stride, t.ldy, C, es));
    }
    wide = wide && vec;
    ViewT<T> yv{y, y_sample_stride, ldy};
    hipStream_t st = (hipStream_t)stream;
    SpmmActGrad G;
    G.ax = nullptr; G.axs = 0; G.ldax = 0; G.act = 0; G.part = nullptr;
    if (ag) {
        // the fused activation gradient: sum mode, vector form, the lanes of a row one aligned power-of-two group
        wide = wide && aligned8(ag->ax, ag->axs, ag->ldax, C, es);
        if (!sum || !vec || !aligned4(ag->ax, ag->axs, ag->ldax, C, es) || !rm_fused(C / (wide ? 8 : 4)) || !ag->part) return CAPE_EINVAL;
        G = *ag;
    }
    const bool fused = vec && rm_fused(C / (wide ? 8 : 4));
    if (!fused)
        for (int k = 0; k < nterms; ++k) P.t[k].rm = nullptr;
    if (vec) CAPE_LAUNCH_SP(spmm_multi_kernel, T, wide, false, dim3((unsigned)(N * spmm_bps(Mo, C / (wide ? 8 : 4)))), dim3(256), 0, st, P, sum, yv, N, Mo, C,
                            fused ? rowmax_out : (float *)nullptr, G);
    else if (any_ell) return CAPE_EINVAL;                      // the scalar fallback reads CSR only
    else CAPE_LAUNCH((spmm_multi_kernel<1, T, 0>), dim3((unsigned)(N * spmm_bps(Mo, C))), dim3(256), 0, st, P, sum, yv, N, Mo, C, (float *)nullptr, G);
    CAPE_LAUNCH_CHECK();
    if (!fused) {
        if (rowmax_out) {
            const int rc = rm_standalone(y, y_sample_stride, ldy, N, Mo, C, rowmax_out, stream);
            if (rc) return rc;
        }
        for (int k = 0; k < nterms && !sum; ++k)
            if (terms[k].rowmax_out) {
                const int rc = rm_standalone(reinterpret_cast<const T *>(terms[k].y), terms[k].y_sample_stride, terms[k].ldy, N, Mo, C,
                                             terms[k].rowmax_out, stream);
                if (rc) return rc;
            }
    }
    return CAPE_OK;
}
}  // namespace

extern "C" int cape_spmm_multi(const cape_spmm_term_t *terms, int32_t nterms, int32_t sum, float *y, int64_t y_sample_stride,
                               int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream) {
    return spmm_multi_impl<float>(terms, nterms, sum, y, y_sample_stride, ldy, N, Mo, C, rowmax_out, stream);
}

extern "C" int cape_spmm_multi_bf16(const cape_spmm_term_t *terms, int32_t nterms, int32_t sum, void *y, int64_t y_sample_stride,
                                    int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream) {
    return spmm_multi_impl<cape_bf16>(terms, nterms, sum, (cape_bf16 *)y, y_sample_stride, ldy, N, Mo, C, rowmax_out, stream);
}

// blocks per sample (= partial-sum chunks of the fused form) for a launch of C channels in the vector form it will take
static int actgrad_cq(const void *y, int64_t ys, int32_t ldy, const void *ax, int64_t axs, int32_t ldax, int32_t C, int es = 4) {
    const bool wide = spmm_wide() && aligned8(y, ys, ldy, C, es) && aligned8(ax, axs, ldax, C, es);
    return C / (wide ? 8 : 4);
}

extern "C" int32_t cape_spmm_multi_actgrad_chunks(const float *y, int64_t y_sample_stride, int32_t ldy, const float *act_x,
                                                  int64_t act_x_sample_stride, int32_t ld_act_x, int32_t Mo, int32_t C) {
    if (!y || !act_x || Mo < 1 || C < 4 || (C & 3)) return CAPE_EINVAL;
    return spmm_bps(Mo, actgrad_cq(y, y_sample_stride, ldy, act_x, act_x_sample_stride, ld_act_x, C));
}

extern "C" int32_t cape_spmm_multi_actgrad_chunks_bf16(const void *y, int64_t y_sample_stride, int32_t ldy, const void *act_x,
                                                       int64_t act_x_sample_stride, int32_t ld_act_x, int32_t Mo, int32_t C) {
    if (!y || !act_x || Mo < 1 || C < 4 || (C & 3)) return CAPE_EINVAL;
    return spmm_bps(Mo, actgrad_cq(y, y_sample_stride, ldy, act_x, act_x_sample_stride, ld_act_x, C, 2));
}

namespace {
template <typename T>
int spmm_multi_actgrad_impl(const cape_spmm_term_t *terms, int32_t nterms, T *y, int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t Mo,
                            int32_t C, float *rowmax_out, const T *act_x, int64_t act_x_sample_stride, int32_t ld_act_x, int32_t act,
                            float *bias_partials, void *stream) {
    if (!act_x || !bias_partials || ld_act_x < C || (act != CAPE_ACT_LEAKY && act != CAPE_ACT_RELU)) return CAPE_EINVAL;
    // every term must allow the 8-wide form exactly when y and act_x do (the chunk count above assumes it)
    SpmmActGrad G;
    G.ax = act_x; G.axs = act_x_sample_stride; G.ldax = ld_act_x; G.act = act; G.part = bias_partials;
    const int cq = actgrad_cq(y, y_sample_stride, ldy, act_x, act_x_sample_stride, ld_act_x, C, (int)sizeof(T));
    for (int k = 0; k < nterms && terms; ++k)
        if (cq == C / 8 && !aligned8(terms[k].x, terms[k].x_sample_stride, terms[k].ldx, C, (int)sizeof(T))) return CAPE_EINVAL;
    return spmm_multi_impl<T>(terms, nterms, 1, y, y_sample_stride, ldy, N, Mo, C, rowmax_out, stream, &G);
}
}  // namespace

extern "C" int cape_spmm_multi_actgrad(const cape_spmm_term_t *terms, int32_t nterms, float *y, int64_t y_sample_stride, int32_t ldy,
                                       int32_t N, int32_t Mo, int32_t C, float *rowmax_out, const float *act_x,
                                       int64_t act_x_sample_stride, int32_t ld_act_x, int32_t act, float *bias_partials, void *stream) {
    return spmm_multi_actgrad_impl<float>(terms, nterms, y, y_sample_stride, ldy, N, Mo, C, rowmax_out, act_x, act_x_sample_stride, ld_act_x,
                                          act, bias_partials, stream);
}

extern "C" int cape_spmm_multi_actgrad_bf16(const cape_spmm_term_t *terms, int32_t nterms, void *y, int64_t y_sample_stride, int32_t ldy,
                                            int32_t N, int32_t Mo, int32_t C, float *rowmax_out, const void *act_x,
                                            int64_t act_x_sample_stride, int32_t ld_act_x, int32_t act, float *bias_partials, void *stream) {
    return spmm_multi_actgrad_impl<cape_bf16>(terms, nterms, (cape_bf16 *)y, y_sample_stride, ldy, N, Mo, C, rowmax_out,
                                              (const cape_bf16 *)act_x, act_x_sample_stride, ld_act_x, act, bias_partials, stream);
}

namespace {
template <typename T>
int spmm_combine_impl(const cape_spmm_term_t *terms, int32_t nterms, uint32_t to_acc2, const cape_rank_t *rank,
                      const float *bias, int32_t bias_mode, int32_t act, int32_t dual, uint32_t *mask_out, T *y,
                      int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t Mo, int32_t F, float *rowmax_out, void *stream) {
    constexpr int es = (int)sizeof(T);
    if (rowmax_out && sizeof(T) != 4) return CAPE_EINVAL;
    if (!terms || nterms < 1 || nterms > CAPE_MAX_SPMM_TERMS || !y || N < 1 || Mo < 1 || F < 1 || ldy < F) return CAPE_EINVAL;
    if (bias_mode != CAPE_BIAS_NONE && !bias) return CAPE_EINVAL;
    if (act < CAPE_ACT_NONE || act > CAPE_ACT_TANH) return CAPE_EINVAL;
    if (dual && (bias_mode != CAPE_BIAS_NONE || act != CAPE_ACT_NONE)) return CAPE_EINVAL;
    if (!dual && (to_acc2 || mask_out)) return CAPE_EINVAL;
    if ((long long)Mo * F >= (1LL << 31)) return CAPE_EINVAL;       // 32-bit work-item index per sample
    CombineParams Q;
    Q.P.n = nterms;
    bool vec = aligned4(y, y_sample_stride, ldy, F, es);
    bool wide = spmm_wide() && aligned8(y, y_sample_stride, ldy, F, es);
    bool any_ell = false;
    for (int k = 0; k < nterms; ++k) {
        const cape_spmm_term_t &t = terms[k];
        if (!t.x || t.ldx < F) return CAPE_EINVAL;
        if (t.rowptr && (!t.colidx || !t.vals)) return CAPE_EINVAL;
        Q.P.t[k].x = t.x; Q.P.t[k].xs = t.x_sample_stride; Q.P.t[k].ldx = t.ldx;
        Q.P.t[k].rp = t.rowptr; Q.P.t[k].ci = t.colidx; Q.P.t[k].va = t.vals;
        Q.P.t[k].y = nullptr; Q.P.t[k].ys = 0; Q.P.t[k].ldy = 0;
        Q.P.t[k].scale = t.scale;
        Q.P.t[k].ew = t.rowptr ? t.ell_width : 0;
        if (t.rowptr && !ell_ok(t.ell_width, t.colidx, t.vals)) return CAPE_EINVAL;
        any_ell = any_ell || Q.P.t[k].ew;
        vec = vec && aligned4(t.x, t.x_sample_stride, t.ldx, F, es);
        wide = wide && aligned8(t.x, t.x_sample_stride, t.ldx, F, es);
    }
    wide = wide && vec;
    Q.to2 = to_acc2;
    Q.rankR = 0; Q.rowscale = nullptr; Q.coef = nullptr; Q.rank_to2 = 0;
    if (rank && rank->R > 0) {
        if (rank->R > CAPE_MAX_SRC || !rank->rowscale || !rank->coef || (rank->to_acc2 && !dual)) return CAPE_EINVAL;
        Q.rankR = rank->R; Q.rowscale = rank->rowscale; Q.coef = rank->coef; Q.rank_to2 = rank->to_acc2;
    }
    Q.bias = bias; Q.bias_mode = bias ? bias_mode : CAPE_BIAS_NONE; Q.act = act; Q.dual = dual ? 1 : 0;
    Q.mask = mask_out; Q.mask_words = (F + 31) / 32;
    if (mask_out && (!vec || (F & 31))) return CAPE_EINVAL;      // sign words are assembled from 8 float4 lanes
    ViewT<T> yv{y, y_sample_stride, ldy};
    hipStream_t st = (hipStream_t)stream;
    const bool fused = vec && rm_fused(F / (wide ? 8 : 4));
    Q.rm = fused ? rowmax_out : nullptr;
    if (vec) CAPE_LAUNCH_SP(spmm_combine_kernel, T, wide, false, dim3((unsigned)(N * spmm_bps(Mo, F / (wide ? 8 : 4)))), dim3(256), 0, st, Q, yv, N, Mo, F);
    else if (any_ell) return CAPE_EINVAL;
    else CAPE_LAUNCH((spmm_combine_kernel<1, T, 0>), dim3((unsigned)(N * spmm_bps(Mo, F))), dim3(256), 0, st, Q, yv, N, Mo, F);
    CAPE_LAUNCH_CHECK();
    if (rowmax_out && !fused) return rm_standalone(y, y_sample_stride, ldy, N, Mo, F, rowmax_out, stream);
    return CAPE_OK;
}
}  // namespace

extern "C" int cape_spmm_combine(const cape_spmm_term_t *terms, int32_t nterms, uint32_t to_acc2, const cape_rank_t *rank,
                                 const float *bias, int32_t bias_mode, int32_t act, int32_t dual, uint32_t *mask_out, float *y,
                                 int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t Mo, int32_t F, float *rowmax_out, void *stream) {
    return spmm_combine_impl<float>(terms, nterms, to_acc2, rank, bias, bias_mode, act, dual, mask_out, y, y_sample_stride, ldy,
                                    N, Mo, F, rowmax_out, stream);
}

extern "C" int cape_spmm_combine_bf16(const cape_spmm_term_t *terms, int32_t nterms, uint32_t to_acc2, const cape_rank_t *rank,
                                      const float *bias, int32_t bias_mode, int32_t act, int32_t dual, uint32_t *mask_out, void *y,
                                      int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t Mo, int32_t F, float *rowmax_out, void *stream) {
    return spmm_combine_impl<cape_bf16>(terms, nterms, to_acc2, rank, bias, bias_mode, act, dual, mask_out, (cape_bf16 *)y,
                                        y_sample_stride, ldy, N, Mo, F, rowmax_out, stream);
}

extern "C" int cape_bias_act_fwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *bias,
                                 int32_t bias_mode, int32_t act, float *y, int64_t y_sample_stride, int32_t ldy,
                                 int32_t N, int32_t M, int32_t C, void *stream) {
    if (!x || !y || N < 1 || M < 1 || C < 1 || ldx < C || ldy < C) return CAPE_EINVAL;
    if (bias_mode != CAPE_BIAS_NONE && !bias) return CAPE_EINVAL;
    if (act < CAPE_ACT_NONE || act > CAPE_ACT_TANH) return CAPE_EINVAL;
    CView xv{x, x_sample_stride, ldx};
    View yv{y, y_sample_stride, ldy};
    CAPE_LAUNCH(bias_act_kernel, dim3(grid_for((long long)N * M * C)), dim3(256), 0, (hipStream_t)stream, xv, bias,
                       bias_mode, act, yv, N, M, C);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_act_bwd(const float *dy, int64_t dy_sample_stride, int32_t lddy, const float *y,
                            int64_t y_sample_stride, int32_t ldy, int32_t act, float *dz, int64_t dz_sample_stride,
                            int32_t lddz, int32_t N, int32_t M, int32_t C, void *stream) {
    if (!dy || !y || !dz || N < 1 || M < 1 || C < 1 || lddy < C || ldy < C || lddz < C) return CAPE_EINVAL;
    if (act < CAPE_ACT_NONE || act > CAPE_ACT_TANH) return CAPE_EINVAL;
    CView gv{dy, dy_sample_stride, lddy}, ov{y, y_sample_stride, ldy};
    View zv{dz, dz_sample_stride, lddz};
    const bool vec = aligned4(dy, dy_sample_stride, lddy, C) && aligned4(y, y_sample_stride, ldy, C) &&
                     aligned4(dz, dz_sample_stride, lddz, C);
    hipStream_t st = (hipStream_t)stream;
    if (vec) CAPE_LAUNCH(act_bwd_kernel<true>, dim3(grid_for((long long)N * M * (C / 4))), dim3(256), 0, st, gv, ov, act, zv, N, M, C);
    else CAPE_LAUNCH(act_bwd_kernel<false>, dim3(grid_for((long long)N * M * C)), dim3(256), 0, st, gv, ov, act, zv, N, M, C);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int64_t cape_colsum_workspace_bytes(int32_t N, int32_t M, int32_t C) {
    if (N < 1 || M < 1 || C < 1) return CAPE_EINVAL;
    const long long R = (long long)N * M;
    const long long nblk = (R + COLSUM_RB - 1) / COLSUM_RB;
    return nblk * C * (int64_t)sizeof(float);
}

extern "C" int cape_colsum(const float *x, int64_t x_sample_stride, int32_t ldx, int32_t N, int32_t M, int32_t C,
                           int32_t per_vertex, int32_t accumulate, float *out, void *workspace,
                           int64_t workspace_bytes, void *stream) {
    if (!x || !out || N < 1 || M < 1 || C < 1 || ldx < C) return CAPE_EINVAL;
    CView xv{x, x_sample_stride, ldx};
    hipStream_t st = (hipStream_t)stream;
    if (per_vertex) {
        CAPE_LAUNCH((sum_over_samples_kernel<float>), dim3(grid_for((long long)M * C)), dim3(256), 0, st, xv, N, M, C, accumulate, out);
        CAPE_LAUNCH_CHECK();
        return CAPE_OK;
    }
    const long long R = (long long)N * M;
    const int nblk = (int)((R + COLSUM_RB - 1) / COLSUM_RB);
    if (!workspace || workspace_bytes < (int64_t)nblk * C * (int64_t)sizeof(float)) return CAPE_EWORKSPACE;
    if (aligned4(x, x_sample_stride, ldx, C) && (C >= 256 ? (C % 256) == 0 : (256 % (C / 4)) == 0))
        CAPE_LAUNCH(colsum_partial_vec_kernel, dim3(nblk), dim3(256), 0, st, xv, N, M, C, (float *)workspace);
    else
        CAPE_LAUNCH(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, xv, N, M, C, (float *)workspace);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, st, (const float *)workspace, nblk, C, accumulate, out);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

/* per-vertex column sums of a bf16 tensor: out[m, c] (+)= sum_n x[n, m, c]  (gradient of the [1, M, F] output bias) */
extern "C" int cape_colsum_vertex_bf16(const void *x, int64_t x_sample_stride, int32_t ldx, int32_t N, int32_t M, int32_t C,
                                       int32_t accumulate, float *out, void *stream) {
    if (!x || !out || N < 1 || M < 1 || C < 1 || ldx < C) return CAPE_EINVAL;
    CViewT<cape_bf16> xv{(const cape_bf16 *)x, x_sample_stride, ldx};
    CAPE_LAUNCH((sum_over_samples_kernel<cape_bf16>), dim3(grid_for((long long)M * C)), dim3(256), 0, (hipStream_t)stream, xv, N, M, C,
                accumulate, out);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_mask_mul(const float *dy, int64_t dy_sample_stride, int32_t lddy, const uint32_t *mask, float *dz,
                             int64_t dz_sample_stride, int32_t lddz, int32_t N, int32_t M, int32_t F, void *stream) {
    if (!dy || !mask || !dz || N < 1 || M < 1 || F < 1 || lddy < F || lddz < F) return CAPE_EINVAL;
    CView gv{dy, dy_sample_stride, lddy};
    View zv{dz, dz_sample_stride, lddz};
    CAPE_LAUNCH(mask_mul_kernel, dim3(grid_for((long long)N * M * F)), dim3(256), 0, (hipStream_t)stream, gv, mask, zv, N, M, F);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_fill_cond(const float *cond, int32_t ldc, const float *scale, float *y, int64_t y_sample_stride,
                              int32_t ldy, int32_t N, int32_t M, int32_t C, void *stream) {
    if (!cond || !y || N < 1 || M < 1 || C < 1 || ldc < C || ldy < C) return CAPE_EINVAL;
    View yv{y, y_sample_stride, ldy};
    CAPE_LAUNCH(fill_cond_kernel, dim3(grid_for((long long)N * M * C)), dim3(256), 0, (hipStream_t)stream, cond, ldc, scale, yv, N, M, C);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_reduce_cond(const float *dy, int64_t dy_sample_stride, int32_t lddy, const float *scale, float *dcond,
                                int32_t ldc, int32_t N, int32_t M, int32_t C, int32_t accumulate, void *stream) {
    if (!dy || !dcond || N < 1 || M < 1 || C < 1 || lddy < C || ldc < C) return CAPE_EINVAL;
    CView gv{dy, dy_sample_stride, lddy};
    const int cgroups = (C + 63) / 64;
    if (aligned4(dy, dy_sample_stride, lddy, C))
        CAPE_LAUNCH(reduce_cond_vec_kernel, dim3(N * cgroups), dim3(256), 0, (hipStream_t)stream, gv, scale, dcond, ldc, N, M, C, accumulate);
    else
        CAPE_LAUNCH(reduce_cond_kernel, dim3(N * cgroups), dim3(256), 0, (hipStream_t)stream, gv, scale, dcond, ldc, N, M, C, accumulate);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int64_t cape_rowscale_reduce_workspace_bytes(int32_t N, int32_t Mo, int32_t F, int32_t R) {
    if (N < 1 || Mo < 1 || F < 1 || R < 1 || R > RSR_MAXR) return CAPE_EINVAL;
    const long long chunks = (Mo + RSR_RB - 1) / RSR_RB;
    return (int64_t)N * chunks * R * F * (int64_t)sizeof(float);
}

extern "C" int cape_rowscale_reduce(const float *dz, int64_t dz_sample_stride, int32_t lddz, const float *rowscale, int32_t R,
                                    int32_t N, int32_t Mo, int32_t F, float *out, void *workspace, int64_t workspace_bytes,
                                    void *stream) {
    if (!dz || !rowscale || !out || !workspace || N < 1 || Mo < 1 || F < 1 || R < 1 || R > RSR_MAXR || lddz < F) return CAPE_EINVAL;
    if (workspace_bytes < cape_rowscale_reduce_workspace_bytes(N, Mo, F, R)) return CAPE_EWORKSPACE;
    const int chunks = (Mo + RSR_RB - 1) / RSR_RB;
    CView zv{dz, dz_sample_stride, lddz};
    hipStream_t st = (hipStream_t)stream;
    CAPE_LAUNCH(rowscale_partial_kernel, dim3(N * chunks), dim3(256), 0, st, zv, rowscale, R, N, Mo, F, (float *)workspace, chunks);
    CAPE_LAUNCH_CHECK();
    const int RF = R * F;
    CAPE_LAUNCH(rowscale_final_kernel, dim3(N * ((RF + 63) / 64)), dim3(256), 0, st, (const float *)workspace, chunks, RF, N, out);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int64_t cape_bwd_prep_workspace_bytes(int32_t N, int32_t Mo, int32_t F, int32_t R) {
    if (N < 1 || Mo < 1 || F < 1 || R < 0 || R > RSR_MAXR) return CAPE_EINVAL;
    const int RB = bp_rows(N, Mo);
    const long long chunks = (Mo + RB - 1) / RB;
    return (int64_t)N * chunks * (R + 2) * F * (int64_t)sizeof(float);
}

namespace {
template <typename T>
int bwd_prep_impl(const T *g, int64_t g_sample_stride, int32_t ldg, const T *y, int64_t y_sample_stride,
                  int32_t ldy, int32_t act, const uint32_t *mask, T *dz, int64_t dz_sample_stride, int32_t lddz,
                  float *dbias, const float *rowscale, int32_t R, float *dcoef, int32_t rg, float *dcoef_g,
                  int64_t dcoef_sample_stride, int32_t finalize, int32_t N, int32_t Mo, int32_t F, void *workspace,
                  int64_t workspace_bytes, float *rowmax_out, void *stream) {
    constexpr int es = (int)sizeof(T);
    if (rowmax_out && es != 4) return CAPE_EINVAL;
    if (!g || !dz || !workspace || N < 1 || Mo < 1 || F < 1 || ldg < F || lddz < F || R < 0 || R > RSR_MAXR) return CAPE_EINVAL;
    if (act < CAPE_ACT_NONE || act > CAPE_ACT_TANH) return CAPE_EINVAL;
    if (!mask && act != CAPE_ACT_NONE && (!y || ldy < F)) return CAPE_EINVAL;
    if ((R > 0 || dcoef_g) && !rowscale) return CAPE_EINVAL;
    if (R > 0 && !dcoef) return CAPE_EINVAL;
    if (workspace_bytes < cape_bwd_prep_workspace_bytes(N, Mo, F, R)) return CAPE_EWORKSPACE;
    const int RB = bp_rows(N, Mo);
    const int chunks = (Mo + RB - 1) / RB;
    CViewT<T> gv{g, g_sample_stride, ldg}, yv{y, y_sample_stride, ldy};
    ViewT<T> zv{dz, dz_sample_stride, lddz};
    hipStream_t st = (hipStream_t)stream;
    const bool vec = aligned4(g, g_sample_stride, ldg, F, es) && aligned4(dz, dz_sample_stride, lddz, F, es) &&
                     (mask || act == CAPE_ACT_NONE || aligned4(y, y_sample_stride, ldy, F, es)) &&
                     (F >= 256 ? (F % 256) == 0 : (256 % (F / 4)) == 0) && ((F & 31) == 0 || !mask);
    // 8 channels per thread where the rows allow it (column passes of 512 channels; F / 8 column groups must divide 256)
    const bool wide = vec && spmm_wide() && aligned8(g, g_sample_stride, ldg, F, es) && aligned8(dz, dz_sample_stride, lddz, F, es) &&
                      (mask || act == CAPE_ACT_NONE || aligned8(y, y_sample_stride, ldy, F, es)) &&
                      F >= 256 && (F >= 512 ? (F % 512) == 0 : (256 % (F / 8)) == 0);
    // measured (tools/bench_sparse.py, profiles/r02_ubench_sparse_bwd_prep.txt): two rows ahead is the best depth for the 4-wide
    // kernel; the 8-wide one pays only from 256 channels (fewer row lanes per column group below that), fp32 without
    // read-ahead (167 registers at depth 4, 129 at 2)
    const int bp_ur = (wide && es == 4) ? 1 : CAPE_BP_UNROLL_DEFAULT;
#define CAPE_BP_LAUNCH(VW_, UR_)                                                                                                    \
    CAPE_LAUNCH((bwd_prep_vec_kernel<T, VW_, UR_>), dim3(N * chunks), dim3(256), 0, st, gv, yv, act, mask, zv, rowscale, R, rg,     \
                dbias ? 1 : 0, dcoef_g ? 1 : 0, N, Mo, F, (float *)workspace, chunks, RB, rowmax_out)
    if (vec && wide) {
        if (bp_ur >= 4) CAPE_BP_LAUNCH(8, 4);
        else if (bp_ur >= 2) CAPE_BP_LAUNCH(8, 2);
        else CAPE_BP_LAUNCH(8, 1);
    } else if (vec) {
        if (bp_ur >= 4) CAPE_BP_LAUNCH(4, 4);
        else if (bp_ur >= 2) CAPE_BP_LAUNCH(4, 2);
        else CAPE_BP_LAUNCH(4, 1);
    }
#undef CAPE_BP_LAUNCH
    else if (F <= 4)
        CAPE_LAUNCH((bwd_prep_narrow_kernel<T>), dim3(N * chunks), dim3(256), 0, st, gv, yv, act, mask, zv, rowscale, R, rg, dbias ? 1 : 0,
                    dcoef_g ? 1 : 0, N, Mo, F, (float *)workspace, chunks, RB);
    else
        CAPE_LAUNCH((bwd_prep_kernel<T>), dim3(N * chunks), dim3(256), 0, st, gv, yv, act, mask, zv, rowscale, R, rg, dbias ? 1 : 0,
                    dcoef_g ? 1 : 0, N, Mo, F, (float *)workspace, chunks, RB);
    CAPE_LAUNCH_CHECK();
    if (rowmax_out && !vec) {                                  // the scalar form does not own whole rows per lane group
        const int rc = rm_standalone(dz, dz_sample_stride, lddz, N, Mo, F, rowmax_out, stream);
        if (rc) return rc;
    }
    if (finalize && (dbias || R > 0 || dcoef_g)) {
        const int fblocks = (F + 15) / 16;
        const int nblk = fblocks * (1 + N * (R + 1));
        CAPE_LAUNCH(bwd_prep_final_kernel, dim3(nblk), dim3(256), 0, st, (const float *)workspace, chunks, N, F, R, dbias, dcoef, dcoef_g,
                    dcoef_sample_stride ? (long long)dcoef_sample_stride : (long long)R * F,
                    dcoef_sample_stride ? (long long)dcoef_sample_stride : (long long)F);
        CAPE_LAUNCH_CHECK();
    }
    return CAPE_OK;
}
}  // namespace

extern "C" int cape_bwd_prep(const float *g, int64_t g_sample_stride, int32_t ldg, const float *y, int64_t y_sample_stride,
                             int32_t ldy, int32_t act, const uint32_t *mask, float *dz, int64_t dz_sample_stride, int32_t lddz,
                             float *dbias, const float *rowscale, int32_t R, float *dcoef, int32_t rg, float *dcoef_g,
                             int64_t dcoef_sample_stride, int32_t finalize, int32_t N, int32_t Mo, int32_t F, void *workspace,
                             int64_t workspace_bytes, float *rowmax_out, void *stream) {
    return bwd_prep_impl<float>(g, g_sample_stride, ldg, y, y_sample_stride, ldy, act, mask, dz, dz_sample_stride, lddz, dbias,
                                rowscale, R, dcoef, rg, dcoef_g, dcoef_sample_stride, finalize, N, Mo, F, workspace, workspace_bytes,
                                rowmax_out, stream);
}

extern "C" int cape_bwd_prep_bf16(const void *g, int64_t g_sample_stride, int32_t ldg, const void *y, int64_t y_sample_stride,
                                  int32_t ldy, int32_t act, const uint32_t *mask, void *dz, int64_t dz_sample_stride, int32_t lddz,
                                  float *dbias, const float *rowscale, int32_t R, float *dcoef, int32_t rg, float *dcoef_g,
                                  int64_t dcoef_sample_stride, int32_t finalize, int32_t N, int32_t Mo, int32_t F, void *workspace,
                                  int64_t workspace_bytes, float *rowmax_out, void *stream) {
    return bwd_prep_impl<cape_bf16>((const cape_bf16 *)g, g_sample_stride, ldg, (const cape_bf16 *)y, y_sample_stride, ldy, act, mask,
                                    (cape_bf16 *)dz, dz_sample_stride, lddz, dbias, rowscale, R, dcoef, rg, dcoef_g,
                                    dcoef_sample_stride, finalize, N, Mo, F, workspace, workspace_bytes, rowmax_out, stream);
}

// work items per row of the fused backward-prep + operator application: 8 channels each where every view allows it, else 4;
// 0 = the arguments do not allow the fused form at all.  es = bytes per element (4: fp32, 2: bf16 storage)
static int prep_spmm_cq(const void *g, int64_t gs, int32_t ldg, const void *dz, int64_t dzs, int32_t lddz, const void *t1,
                        int64_t t1s, int32_t ldt1, int32_t F, int es) {
    if ((F & 31) || !aligned4(g, gs, ldg, F, es) || !aligned4(dz, dzs, lddz, F, es) || !aligned4(t1, t1s, ldt1, F, es)) return 0;
    const bool wide = spmm_wide() && aligned8(g, gs, ldg, F, es) && aligned8(dz, dzs, lddz, F, es) && aligned8(t1, t1s, ldt1, F, es);
    const int cq = F / (wide ? 8 : 4);
    return (rm_fused(cq) && cq >= 4) ? cq : 0;
}
// groups of 256 work items a block handles in turn (its weighted sums are reduced once): measured on the five affine-block
// shapes of the benchmarked step (16 samples, 108 groups each; tools/experiments/prep_spmm_bench.py under rocprofv3):
// 21.4 / 20.9 / 19.9 / 20.9 us at 1 / 2 / 3 / 4 -- three keeps 576 blocks for the 256 CUs
static int prep_spmm_rpb(int N, int bps) {
    const long long groups = (long long)N * bps;
    return groups >= 1536 ? 3 : groups >= 1024 ? 2 : 1;
}
static int32_t prep_spmm_chunks(const void *g, int64_t gs, int32_t ldg, const void *dz, int64_t dzs, int32_t lddz, const void *t1, int64_t t1s,
                                int32_t ldt1, int32_t N, int32_t Mo, int32_t F, int es) {
    const int cq = prep_spmm_cq(g, gs, ldg, dz, dzs, lddz, t1, t1s, ldt1, F, es);
    if (!cq || Mo < 1 || N < 1) return 0;
    const int bps = spmm_bps(Mo, cq), rpb = prep_spmm_rpb(N, bps);
    return (bps + rpb - 1) / rpb;
}

extern "C" int32_t cape_bwd_prep_spmm_chunks(const float *g, int64_t g_sample_stride, int32_t ldg, const float *dz,
                                             int64_t dz_sample_stride, int32_t lddz, const float *t1, int64_t t1_sample_stride,
                                             int32_t ldt1, int32_t N, int32_t Mo, int32_t F) {
    return prep_spmm_chunks(g, g_sample_stride, ldg, dz, dz_sample_stride, lddz, t1, t1_sample_stride, ldt1, N, Mo, F, 4);
}
extern "C" int32_t cape_bwd_prep_spmm_chunks_bf16(const void *g, int64_t g_sample_stride, int32_t ldg, const void *dz,
                                                  int64_t dz_sample_stride, int32_t lddz, const void *t1, int64_t t1_sample_stride,
                                                  int32_t ldt1, int32_t N, int32_t Mo, int32_t F) {
    return prep_spmm_chunks(g, g_sample_stride, ldg, dz, dz_sample_stride, lddz, t1, t1_sample_stride, ldt1, N, Mo, F, 2);
}

namespace {
template <typename T>
int bwd_prep_spmm_impl(const T *g, int64_t g_sample_stride, int32_t ldg, const uint32_t *mask, const int32_t *rowptr, const int32_t *colidx,
                       const float *vals, int32_t ell_width, T *dz, int64_t dz_sample_stride, int32_t lddz, T *t1, int64_t t1_sample_stride,
                       int32_t ldt1, const float *rowscale, int32_t R, int32_t rg, int32_t N, int32_t Mo, int32_t F, float *partials,
                       int64_t partials_bytes, float *rowmax_g_out, float *rowmax_t1_out, void *stream) {
    constexpr int es = (int)sizeof(T);
    if (!g || !mask || !rowptr || !colidx || !vals || !dz || !t1 || N < 1 || Mo < 1 || F < 1 || ldg < F || lddz < F || ldt1 < F)
        return CAPE_EINVAL;
    if ((rowmax_g_out || rowmax_t1_out) && es != 4) return CAPE_EINVAL;
    if (R < 0 || R > PS_MAXR || ((R > 0 || rg >= 0) && (!rowscale || !partials))) return CAPE_EINVAL;
    if (!ell_ok(ell_width, colidx, vals)) return CAPE_EINVAL;
    if ((long long)Mo * F >= (1LL << 31)) return CAPE_EINVAL;
    const int cq = prep_spmm_cq(g, g_sample_stride, ldg, dz, dz_sample_stride, lddz, t1, t1_sample_stride, ldt1, F, es);
    if (!cq) return CAPE_EINVAL;
    const int bps = spmm_bps(Mo, cq), rpb = prep_spmm_rpb(N, bps);
    const int chunks = (bps + rpb - 1) / rpb;
    if ((R > 0 || rg >= 0) && partials_bytes < (int64_t)N * chunks * (R + 2) * F * (int64_t)sizeof(float)) return CAPE_EWORKSPACE;
    PrepSpmmP P;
    P.g = g; P.gs = g_sample_stride; P.ldg = ldg;
    P.mask = mask; P.words = (F + 31) / 32;
    P.rp = rowptr; P.ci = colidx; P.va = vals; P.ew = ell_width;
    P.dz = dz; P.dzs = dz_sample_stride; P.lddz = lddz;
    P.t1 = t1; P.t1s = t1_sample_stride; P.ldt1 = ldt1;
    P.rowscale = rowscale; P.R = R; P.rg = rg < 0 ? -1 : rg;
    P.part = partials;
    P.rm_g = rowmax_g_out; P.rm_t1 = rowmax_t1_out;
    const bool wide = cq * 8 == F;
    const dim3 grid((unsigned)(N * chunks));
    hipStream_t st = (hipStream_t)stream;
    const int u = spmm_unroll();
    if (wide) {
        if (u == 8) CAPE_LAUNCH((bwd_prep_spmm_kernel<8, 8, T>), grid, dim3(256), 0, st, P, N, Mo, F, rpb, chunks);
        else if (u == 4) CAPE_LAUNCH((bwd_prep_spmm_kernel<8, 4, T>), grid, dim3(256), 0, st, P, N, Mo, F, rpb, chunks);
        else CAPE_LAUNCH((bwd_prep_spmm_kernel<8, 0, T>), grid, dim3(256), 0, st, P, N, Mo, F, rpb, chunks);
    } else {
        if (u == 8) CAPE_LAUNCH((bwd_prep_spmm_kernel<4, 8, T>), grid, dim3(256), 0, st, P, N, Mo, F, rpb, chunks);
        else if (u == 4) CAPE_LAUNCH((bwd_prep_spmm_kernel<4, 4, T>), grid, dim3(256), 0, st, P, N, Mo, F, rpb, chunks);
        else CAPE_LAUNCH((bwd_prep_spmm_kernel<4, 0, T>), grid, dim3(256), 0, st, P, N, Mo, F, rpb, chunks);
    }
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
}  // namespace

extern "C" int cape_bwd_prep_spmm(const float *g, int64_t g_sample_stride, int32_t ldg, const uint32_t *mask, const int32_t *rowptr,
                                  const int32_t *colidx, const float *vals, int32_t ell_width, float *dz, int64_t dz_sample_stride,
                                  int32_t lddz, float *t1, int64_t t1_sample_stride, int32_t ldt1, const float *rowscale, int32_t R,
                                  int32_t rg, int32_t N, int32_t Mo, int32_t F, float *partials, int64_t partials_bytes,
                                  float *rowmax_g_out, float *rowmax_t1_out, void *stream) {
    return bwd_prep_spmm_impl<float>(g, g_sample_stride, ldg, mask, rowptr, colidx, vals, ell_width, dz, dz_sample_stride, lddz, t1,
                                     t1_sample_stride, ldt1, rowscale, R, rg, N, Mo, F, partials, partials_bytes, rowmax_g_out, rowmax_t1_out,
                                     stream);
}
extern "C" int cape_bwd_prep_spmm_bf16(const void *g, int64_t g_sample_stride, int32_t ldg, const uint32_t *mask, const int32_t *rowptr,
                                       const int32_t *colidx, const float *vals, int32_t ell_width, void *dz, int64_t dz_sample_stride,
                                       int32_t lddz, void *t1, int64_t t1_sample_stride, int32_t ldt1, const float *rowscale, int32_t R,
                                       int32_t rg, int32_t N, int32_t Mo, int32_t F, float *partials, int64_t partials_bytes,
                                       float *rowmax_g_out, float *rowmax_t1_out, void *stream) {
    return bwd_prep_spmm_impl<cape_bf16>((const cape_bf16 *)g, g_sample_stride, ldg, mask, rowptr, colidx, vals, ell_width, (cape_bf16 *)dz,
                                         dz_sample_stride, lddz, (cape_bf16 *)t1, t1_sample_stride, ldt1, rowscale, R, rg, N, Mo, F, partials,
                                         partials_bytes, rowmax_g_out, rowmax_t1_out, stream);
}

// work items per row of cape_spmm_multi_prep (8 channels where every view allows it, else 4); 0 = not possible
static int multi_prep_cq(const cape_spmm_term_t *terms, int32_t nterms, int32_t C, int es) {
    if (!terms || nterms < 1 || nterms > MP_MAXT || (C & 31)) return 0;
    bool wide = spmm_wide();
    for (int k = 0; k < nterms; ++k) {
        const cape_spmm_term_t &t = terms[k];
        if (!t.x || !t.y || !t.rowptr || !t.colidx || !t.vals || t.ldx < C || t.ldy < C || t.scale != 1.0f) return 0;
        if (t.rowmax_out && es != 4) return 0;
        if (!aligned4(t.x, t.x_sample_stride, t.ldx, C, es) || !aligned4(t.y, t.y_sample_stride, t.ldy, C, es)) return 0;
        if (!ell_ok(t.ell_width, t.colidx, t.vals)) return 0;
        wide = wide && aligned8(t.x, t.x_sample_stride, t.ldx, C, es) && aligned8(t.y, t.y_sample_stride, t.ldy, C, es);
    }
    const int cq = C / (wide ? 8 : 4);
    return (rm_fused(cq) && cq >= 4) ? cq : 0;
}
static int32_t multi_prep_chunks(const cape_spmm_term_t *terms, int32_t nterms, int32_t N, int32_t Mo, int32_t C, int es) {
    const int cq = multi_prep_cq(terms, nterms, C, es);
    if (!cq || Mo < 1 || N < 1) return 0;
    const int bps = spmm_bps(Mo, cq), rpb = prep_spmm_rpb(N, bps);
    return (bps + rpb - 1) / rpb;
}

extern "C" int32_t cape_spmm_multi_prep_chunks(const cape_spmm_term_t *terms, int32_t nterms, int32_t N, int32_t Mo, int32_t C) {
    return multi_prep_chunks(terms, nterms, N, Mo, C, 4);
}
extern "C" int32_t cape_spmm_multi_prep_chunks_bf16(const cape_spmm_term_t *terms, int32_t nterms, int32_t N, int32_t Mo, int32_t C) {
    return multi_prep_chunks(terms, nterms, N, Mo, C, 2);
}

namespace {
template <typename T>
int spmm_multi_prep_impl(const cape_spmm_term_t *terms, int32_t nterms, uint32_t masked_terms, const uint32_t *mask, int32_t mask_rows,
                         int32_t N, int32_t Mo, int32_t C, float *partials, int64_t partials_bytes, void *stream) {
    const int cq = multi_prep_cq(terms, nterms, C, (int)sizeof(T));
    if (!cq || N < 1 || Mo < 1 || mask_rows < 1 || (masked_terms && !mask) || (masked_terms >> nterms)) return CAPE_EINVAL;
    if ((long long)Mo * C >= (1LL << 31)) return CAPE_EINVAL;
    const int bps = spmm_bps(Mo, cq), rpb = prep_spmm_rpb(N, bps);
    const int chunks = (bps + rpb - 1) / rpb;
    const int T_ = nterms + 1;
    if (partials && partials_bytes < (int64_t)N * chunks * T_ * C * (int64_t)sizeof(float)) return CAPE_EWORKSPACE;
    SpmmTerms P;
    P.n = nterms;
    for (int k = 0; k < nterms; ++k) {
        const cape_spmm_term_t &t = terms[k];
        P.t[k].x = t.x; P.t[k].xs = t.x_sample_stride; P.t[k].ldx = t.ldx;
        P.t[k].rp = t.rowptr; P.t[k].ci = t.colidx; P.t[k].va = t.vals;
        P.t[k].y = t.y; P.t[k].ys = t.y_sample_stride; P.t[k].ldy = t.ldy;
        P.t[k].scale = 1.0f;
        P.t[k].ew = t.ell_width;
        P.t[k].rm = t.rowmax_out;
    }
    const bool wide = cq * 8 == C;
    const dim3 grid((unsigned)(N * chunks));
    hipStream_t st = (hipStream_t)stream;
    const int words = (C + 31) / 32;
    const int u = spmm_unroll();
    // the last term is the first one without its mask (same operator arrays, same input): one set of gathers serves both
    const cape_spmm_term_t &ta = terms[0], &tb = terms[nterms - 1];
    const int pair = nterms >= 2 && (masked_terms & 1u) && !((masked_terms >> (nterms - 1)) & 1u) && ta.x == tb.x &&
                     ta.x_sample_stride == tb.x_sample_stride && ta.ldx == tb.ldx && ta.rowptr == tb.rowptr && ta.colidx == tb.colidx &&
                     ta.vals == tb.vals && ta.ell_width == tb.ell_width;
#define CAPE_MP_LAUNCH(VW_, U_)                                                                                                          \
    CAPE_LAUNCH((spmm_multi_prep_kernel<VW_, U_, T>), grid, dim3(256), 0, st, P, masked_terms, pair, mask, words, mask_rows, N, Mo, C,   \
                partials, T_, rpb, chunks)
    if (wide) {
        if (u == 8) CAPE_MP_LAUNCH(8, 8);
        else if (u == 4) CAPE_MP_LAUNCH(8, 4);
        else CAPE_MP_LAUNCH(8, 0);
    } else {
        if (u == 8) CAPE_MP_LAUNCH(4, 8);
        else if (u == 4) CAPE_MP_LAUNCH(4, 4);
        else CAPE_MP_LAUNCH(4, 0);
    }
#undef CAPE_MP_LAUNCH
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
}  // namespace

extern "C" int cape_spmm_multi_prep(const cape_spmm_term_t *terms, int32_t nterms, uint32_t masked_terms, const uint32_t *mask,
                                    int32_t mask_rows, int32_t N, int32_t Mo, int32_t C, float *partials, int64_t partials_bytes,
                                    void *stream) {
    return spmm_multi_prep_impl<float>(terms, nterms, masked_terms, mask, mask_rows, N, Mo, C, partials, partials_bytes, stream);
}
extern "C" int cape_spmm_multi_prep_bf16(const cape_spmm_term_t *terms, int32_t nterms, uint32_t masked_terms, const uint32_t *mask,
                                         int32_t mask_rows, int32_t N, int32_t Mo, int32_t C, float *partials, int64_t partials_bytes,
                                         void *stream) {
    return spmm_multi_prep_impl<cape_bf16>(terms, nterms, masked_terms, mask, mask_rows, N, Mo, C, partials, partials_bytes, stream);
}

extern "C" int cape_bwd_prep_finalize(const cape_bwd_prep_item_t *items, int32_t nitems, void *stream) {
    if (!items || nitems < 1 || nitems > CAPE_MAX_BWD_PREP_ITEMS) return CAPE_EINVAL;
    BpFinalBatch B;
    B.n = nitems;
    int off = 0;
    for (int i = 0; i < nitems; ++i) {
        const cape_bwd_prep_item_t &t = items[i];
        if (!t.workspace || t.N < 1 || t.Mo < 1 || t.F < 1 || t.R < 0 || t.R > RSR_MAXR || (t.R > 0 && !t.dcoef)) return CAPE_EINVAL;
        const int RB = bp_rows(t.N, t.Mo);
        B.it[i].part = (const float *)t.workspace;
        B.it[i].chunks = t.chunks > 0 ? t.chunks : (t.Mo + RB - 1) / RB;
        B.it[i].N = t.N; B.it[i].F = t.F; B.it[i].R = t.R;
        B.it[i].dbias = t.dbias; B.it[i].dcoef = t.dcoef; B.it[i].dcoef_g = t.dcoef_g;
        B.it[i].cs = t.dcoef_sample_stride ? (long long)t.dcoef_sample_stride : (long long)t.R * t.F;
        B.it[i].cgs = t.dcoef_sample_stride ? (long long)t.dcoef_sample_stride : (long long)t.F;
        B.blk_off[i] = off;
        off += ((t.F + 15) / 16) * (1 + t.N * (t.R + 1));
    }
    B.blk_off[nitems] = off;
    CAPE_LAUNCH(bwd_prep_final_batch_kernel, dim3(off), dim3(256), 0, (hipStream_t)stream, B);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
