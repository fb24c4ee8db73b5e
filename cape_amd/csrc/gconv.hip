// Fused gather-GEMM for the Chebyshev mesh convolution (gfx950 / CDNA4, fp32 MFMA).
//
//   y[n, r, :] = epilogue( sum_s  (S_s x_s[n])[r, :] @ B_s )
//
// S_s is a small CSR operator (a precomposed T_k(L~), T_k(L~)*U, D*T_k(L~) or a transpose of
// one of those); the gathered A-tile [BM x 32] is built in LDS from coalesced float4 reads
// of the [N, M, C] activations (neighbour rows come from L2: one mesh level of one sample is
// <= 2.6 MB), the trailing dense contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32).
// This one kernel covers chebyshev5 (reference lib/models.py:69-103), the bias/activation
// epilogues (:105-127), poolwT folded in as an operator (:129-152), res_block_affine in DUAL
// mode (:776-793) and -- with transposed operators and weights -- their data gradients.
// The weight-gradient kernel (contraction over vertices) lives below.
#include "common.h"
#include "gconv_shared.h"
#include "gemm_plain.h"
#include "gemm_split.h"
#include "gemm_h2.h"
#include "narrow.h"
#include <stdlib.h>

namespace {

constexpr int KC_DEFAULT = 32;   // contraction chunk staged per iteration

// ---- A-tile staging: gathered rows -> LDS [ROWS][KC+4] -----------------------------------
template <int ROWS, int LDA, int KC, typename AT = float>
__device__ __forceinline__ void stage_gather(float *sA, const SrcDev &S, int n, int r0, int Mo,
                                              int c0, int tid) {
    constexpr int QPR = KC / 4;            // float4 columns per row of the chunk
    constexpr int RP = 256 / QPR;          // rows staged per pass
    const int q = tid % QPR;
    const int rl0 = tid / QPR;
    const int c = c0 + 4 * q;
    const AT *x0 = reinterpret_cast<const AT *>(S.x) + (long long)n * S.xs;
    const AT *xb = x0 + c;
    const int nvalid = S.C - c;   // channels available from c
    constexpr int P = ROWS / RP;
    if (!S.rp && S.vec && (nvalid >= 4 || (nvalid > 0 && c + 4 <= S.ldx) || nvalid <= 0)) {
        // plain source, aligned: issue every load of the chunk before the first LDS store.  A row
        // padded to a multiple of 4 floats lets the last (partial) float4 be loaded whole; the lanes
        // beyond C are zeroed below (e.g. the 3-channel network input in a 4-float row).
        float4 v[P];
#pragma unroll
        for (int pass = 0; pass < P; ++pass) {
            const int r = r0 + rl0 + RP * pass;
            // unconditional load from a clamped (always valid) row, zeroed by a select afterwards:
            // a branch around each load would make hipcc wait vmcnt(0) per load (serialised round trips)
            const int rc = r < Mo ? r : Mo - 1;
            v[pass] = cape_ld4((nvalid > 0 ? xb : x0) + (long long)rc * S.ldx);
        }
#pragma unroll
        for (int pass = 0; pass < P; ++pass) {
            const bool ok = (r0 + rl0 + RP * pass) < Mo;
            float4 o = v[pass];
            o.x = (ok && nvalid > 0) ? o.x : 0.f; o.y = (ok && nvalid > 1) ? o.y : 0.f;
            o.z = (ok && nvalid > 2) ? o.z : 0.f; o.w = (ok && nvalid > 3) ? o.w : 0.f;
            *reinterpret_cast<float4 *>(&sA[(rl0 + RP * pass) * LDA + 4 * q]) = o;
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < P; ++pass) {
        const int rl = rl0 + RP * pass;
        const int r = r0 + rl;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < Mo && nvalid > 0) {
            if (S.vec && nvalid >= 4) {
                // gathered source: two row entries in flight per step
                const int e1 = S.rp[r + 1];
                int e = S.rp[r];
                for (; e + 1 < e1; e += 2) {
                    const float v0 = S.va[e], v1 = S.va[e + 1];
                    const float4 xa = cape_ld4(xb + (long long)S.ci[e] * S.ldx);
                    const float4 xc = cape_ld4(xb + (long long)S.ci[e + 1] * S.ldx);
                    acc.x = fmaf(v0, xa.x, acc.x); acc.y = fmaf(v0, xa.y, acc.y);
                    acc.z = fmaf(v0, xa.z, acc.z); acc.w = fmaf(v0, xa.w, acc.w);
                    acc.x = fmaf(v1, xc.x, acc.x); acc.y = fmaf(v1, xc.y, acc.y);
                    acc.z = fmaf(v1, xc.z, acc.z); acc.w = fmaf(v1, xc.w, acc.w);
                }
                if (e < e1) {
                    const float v0 = S.va[e];
                    const float4 xa = cape_ld4(xb + (long long)S.ci[e] * S.ldx);
                    acc.x = fmaf(v0, xa.x, acc.x); acc.y = fmaf(v0, xa.y, acc.y);
                    acc.z = fmaf(v0, xa.z, acc.z); acc.w = fmaf(v0, xa.w, acc.w);
                }
            } else {
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                if (S.rp) {
                    const int e1 = S.rp[r + 1];
                    for (int e = S.rp[r]; e < e1; ++e) {
                        const float v = S.va[e];
                        const AT *xr = xb + (long long)S.ci[e] * S.ldx;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nvalid) a[u] = fmaf(v, cape_ld(xr + u), a[u]);
                    }
                } else {
                    const AT *xr = xb + (long long)r * S.ldx;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < nvalid) a[u] = cape_ld(xr + u);
                }
                acc = make_float4(a[0], a[1], a[2], a[3]);
            }
        }
        *reinterpret_cast<float4 *>(&sA[rl * LDA + 4 * q]) = acc;
    }
}

// ---- B-tile staging: weights [KC x BN] with arbitrary (row, col) strides -> LDS [KC][BN+4]
// NT = number of staging threads (the loader half of the workgroup)
template <int BN, int LDB, int NT, int KC>
__device__ __forceinline__ void stage_weights(float *sB, const float *w, long long rs, long long cs,
                                               int C, int F, int c0, int f0, int tid) {
    if (cs == 1 && ((rs & 3) == 0) && ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0)) {
        // output index contiguous in memory (forward weights): float4 along f
        constexpr int NL = (KC * (BN / 4) + NT - 1) / NT;
        float4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * NT;
            const int j4 = idx % (BN / 4), kk = idx / (BN / 4);
            const int c = c0 + kk, f = f0 + 4 * j4;
            const int cc = c < C ? c : C - 1, fc = f < F ? f : 0;      // clamped, always valid
            v[i] = *reinterpret_cast<const float4 *>(w + cc * rs + fc);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * NT;
            const int j4 = idx % (BN / 4), kk = idx / (BN / 4);
            const bool ok = (c0 + kk < C) && (f0 + 4 * j4 < F);
            float4 o = v[i];
            o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
            if (idx < KC * (BN / 4)) *reinterpret_cast<float4 *>(&sB[kk * LDB + 4 * j4]) = o;
        }
    } else if (rs == 1 && ((cs & 3) == 0) && ((C & 3) == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0)) {
        // contraction index contiguous in memory (transposed weights, data gradient): float4 along c
        constexpr int NL = ((KC / 4) * BN + NT - 1) / NT;
        float4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * NT;
            const int k4 = idx % (KC / 4), j = idx / (KC / 4);
            const int c = c0 + 4 * k4, f = f0 + j;
            const int cc = c < C ? c : 0, fc = f < F ? f : F - 1;      // clamped, always valid
            v[i] = *reinterpret_cast<const float4 *>(w + cc + fc * cs);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * NT;
            const int k4 = idx % (KC / 4), j = idx / (KC / 4);
            const bool ok = (c0 + 4 * k4 < C) && (f0 + j < F);
            if (idx < (KC / 4) * BN) {
                sB[(4 * k4 + 0) * LDB + j] = ok ? v[i].x : 0.f;
                sB[(4 * k4 + 1) * LDB + j] = ok ? v[i].y : 0.f;
                sB[(4 * k4 + 2) * LDB + j] = ok ? v[i].z : 0.f;
                sB[(4 * k4 + 3) * LDB + j] = ok ? v[i].w : 0.f;
            }
        }
    } else {
        const bool kmajor = (rs == 1 && cs != 1);
#pragma unroll 4
        for (int idx = tid; idx < KC * BN; idx += NT) {
            int kk, j;
            if (kmajor) {
                kk = idx & (KC - 1);
                j = idx / KC;
            } else {
                j = idx % BN;
                kk = idx / BN;
            }
            const int c = c0 + kk, f = f0 + j;
            float v = 0.f;
            if (c < C && f < F) v = w[c * rs + f * cs];
            sB[kk * LDB + j] = v;
        }
    }
}

// Workgroup = 4 waves; every wave stages its share of the [BM x KC] gathered A chunk and the [KC x BN]
// weight chunk, then multiplies.  3-4 resident workgroups per CU overlap each other's staging and MFMA
// phases (measured faster here than a loader/MFMA wave split and than a register-prefetch pipeline, which
// cost occupancy on the gather path; plain sources take the pipelined kernel of gemm_plain.h instead).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool DUAL, int KC = KC_DEFAULT, typename AT = float>
__global__ __launch_bounds__(256, 4) void gconv_fwd_kernel(GconvParams p) {
    constexpr int LDA = KC + 4;
    constexpr int LDB = BN + 4;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_SZ = BM * LDA, B_SZ = KC * LDB;
    constexpr int BUF_SZ = A_SZ + (DUAL ? 2 : 1) * B_SZ;
    static_assert(WAVES_M * WAVES_N == 4, "4 MFMA waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");

    __shared__ __attribute__((aligned(16))) float smem[BUF_SZ];

    const int tid = threadIdx.x;
    const int ltid = tid;
    const int lane = tid & 63, wave = (tid >> 6) & 3;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    int n, t;
    cape_map_block(blockIdx.x, p.N, p.row_tiles * p.col_tiles, n, t);
    const int r0 = (t / p.col_tiles) * BM;
    const int f0 = (t % p.col_tiles) * BN;

    f32x16 acc[TM][TN];
    f32x16 acc2[DUAL ? TM : 1][DUAL ? TN : 1];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                acc[a][b][g] = 0.f;
                if (DUAL) acc2[a][b][g] = 0.f;
            }

    int total = 0;
    for (int si = 0; si < p.nsrc; ++si) total += (p.s[si].C + KC - 1) / KC;

    // loader cursor (source, channel offset) of the NEXT chunk to stage; consumer cursor of the
    // chunk being multiplied (only its source's w2 flag matters)
    int l_si = 0, l_c0 = 0;
    int c_si = 0, c_c0 = 0;

    auto stage = [&](int buf) {
        const SrcDev &S = p.s[l_si];
        float *sA = smem + buf * BUF_SZ;
        float *sB = sA + A_SZ;
        stage_gather<BM, LDA, KC, AT>(sA, S, n, r0, p.Mo, l_c0, ltid);
        stage_weights<BN, LDB, 256, KC>(sB, S.w, S.wrs, S.wcs, S.C, p.F, l_c0, f0, ltid);
        if (DUAL && S.w2) stage_weights<BN, LDB, 256, KC>(sB + B_SZ, S.w2, S.w2rs, S.w2cs, S.C, p.F, l_c0, f0, ltid);
        l_c0 += KC;
        if (l_c0 >= S.C) { l_c0 = 0; ++l_si; }
    };

    auto compute = [&](int buf) {
            const float *sA = smem + buf * BUF_SZ;
            const float *sB = sA + A_SZ;
            const float *sB2 = sB + B_SZ;
            const bool has2 = DUAL && (p.s[c_si].w2 != nullptr);
#pragma unroll
            for (int kb = 0; kb < KC / 8; ++kb) {
                // contraction index permutation: MFMA step u of this block of 8 uses physical
                // index kb*8 + 4*lh + u for lane-half lh (same mapping for A and B).
                float4 av[TM];
                float bv[TN][4];
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    av[a] = *reinterpret_cast<const float4 *>(
                        &sA[(wm * WTM + a * 32 + li) * LDA + kb * 8 + 4 * lh]);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        bv[b][u] = sB[(kb * 8 + 4 * lh + u) * LDB + wn * WTN + b * 32 + li];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const float af = (u == 0) ? av[a].x : (u == 1) ? av[a].y : (u == 2) ? av[a].z : av[a].w;
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bv[b][u], acc[a][b], 0, 0, 0);
                    }
                }
                if (DUAL && has2) {
                    float b2v[TN][4];
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            b2v[b][u] = sB2[(kb * 8 + 4 * lh + u) * LDB + wn * WTN + b * 32 + li];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int a = 0; a < TM; ++a) {
                            const float af = (u == 0) ? av[a].x : (u == 1) ? av[a].y : (u == 2) ? av[a].z : av[a].w;
#pragma unroll
                            for (int b = 0; b < TN; ++b)
                                acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b2v[b][u], acc2[a][b], 0, 0, 0);
                        }
                    }
                }
            }
            c_c0 += KC;
            if (c_c0 >= p.s[c_si].C) { c_c0 = 0; ++c_si; }
    };

    for (int it = 0; it < total; ++it) {
        __syncthreads();
        stage(0);
        __syncthreads();
        compute(0);
    }

    gconv_epilogue<BM, BN, WAVES_M, WAVES_N, DUAL, AT>(p, acc, acc2, n, r0, f0, wm, wn, li, lh);
}

// =============================================================================================
// weight gradient:  dW_s[c, f] = sum_{n, r} A_s[n, r, c] * dz[n, r, f]
// Each workgroup owns one [CT x FT] tile of one source's dW and one (sample, row-range)
// slice of the vertex dimension; partials go to the workspace and are summed in a fixed order
// by dw_reduce_kernel (deterministic; no float atomics).
// =============================================================================================
template <int CT, int FT, typename AT = float>
__global__ __launch_bounds__(256, 4) void gconv_dw_kernel(DwParams p) {
    constexpr int RK = 32;
    constexpr int LDA = CT + 4, LDB = FT + 4;
    constexpr int WTM = CT / 2, WTN = FT / 2;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    __shared__ __attribute__((aligned(16))) float smem[RK * LDA + RK * LDB];
    float *sA = smem;
    float *sB = smem + RK * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int ntiles = p.tile_off[p.nsrc];
    int tile, split;                                  // split = group * rsplit + rs
    if (!cape_map_dw_block(blockIdx.x, ntiles, p.ngroups * p.rsplit, tile, split)) return;
    const int grp = split / p.rsplit;
    const int rs = split % p.rsplit;
    const int n_begin = grp * p.samples_per_group;
    const int n_end = min(p.N, n_begin + p.samples_per_group);
    int si = 0;
    while (si + 1 < p.nsrc && tile >= p.tile_off[si + 1]) ++si;
    const SrcDev &S = p.s[si];
    const int lt = tile - p.tile_off[si];
    const int c0 = (lt / p.ftiles) * CT;
    const int f0 = (lt % p.ftiles) * FT;
    const int ra = rs * p.rows_per_split;
    const int rb = min(p.Mo, ra + p.rows_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    for (int n = n_begin; n < n_end; ++n) {
    const AT *dzb = reinterpret_cast<const AT *>(((p.dz2_mask >> si) & 1u) ? p.dz2 : p.dz) + (long long)n * p.dzs;
    const AT *xb = reinterpret_cast<const AT *>(S.x) + (long long)n * S.xs;

    for (int rbase = ra; rbase < rb; rbase += RK) {
        __syncthreads();
        // A chunk: RK (gathered) rows x CT channels
        if (!S.rp && S.vec && (((S.C & 3) == 0) || (((S.C + 3) & ~3) <= S.ldx))) {
            // plain aligned source: all loads first, clamped rows/columns + select (partial channel tiles
            // included: a float4 column is either entirely inside [0, C) or entirely outside)
            constexpr int NA = RK * (CT / 4) / 256;
            float4 va4[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = tid + i * 256;
                const int rl = idx / (CT / 4), q = idx % (CT / 4);
                const int r = rbase + rl;
                const int rc = r < rb ? r : rb - 1;
                const int cc = (c0 + 4 * q) < S.C ? (c0 + 4 * q) : 0;
                va4[i] = cape_ld4(xb + (long long)rc * S.ldx + cc);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = tid + i * 256;
                const int rl = idx / (CT / 4), q = idx % (CT / 4);
                const bool okr = (rbase + rl) < rb;
                const int nv = S.C - (c0 + 4 * q);       // valid lanes of this float4 column
                float4 o = va4[i];
                o.x = (okr && nv > 0) ? o.x : 0.f; o.y = (okr && nv > 1) ? o.y : 0.f;
                o.z = (okr && nv > 2) ? o.z : 0.f; o.w = (okr && nv > 3) ? o.w : 0.f;
                *reinterpret_cast<float4 *>(&sA[rl * LDA + 4 * q]) = o;
            }
        } else
        for (int idx = tid; idx < RK * (CT / 4); idx += 256) {
            const int rl = idx / (CT / 4), q = idx % (CT / 4);
            const int r = rbase + rl;
            const int c = c0 + 4 * q;
            const int nvalid = S.C - c;
            float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rb && nvalid > 0) {
                if (S.vec && nvalid >= 4) {
                    if (S.rp) {
                        const int e1 = S.rp[r + 1];
                        for (int e = S.rp[r]; e < e1; ++e) {
                            const float v = S.va[e];
                            const float4 xv = cape_ld4(xb + (long long)S.ci[e] * S.ldx + c);
                            v4.x = fmaf(v, xv.x, v4.x);
                            v4.y = fmaf(v, xv.y, v4.y);
                            v4.z = fmaf(v, xv.z, v4.z);
                            v4.w = fmaf(v, xv.w, v4.w);
                        }
                    } else {
                        v4 = cape_ld4(xb + (long long)r * S.ldx + c);
                    }
                } else {
                    float a[4] = {0.f, 0.f, 0.f, 0.f};
                    if (S.rp) {
                        const int e1 = S.rp[r + 1];
                        for (int e = S.rp[r]; e < e1; ++e) {
                            const float v = S.va[e];
                            const AT *xr = xb + (long long)S.ci[e] * S.ldx + c;
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (u < nvalid) a[u] = fmaf(v, cape_ld(xr + u), a[u]);
                        }
                    } else {
                        const AT *xr = xb + (long long)r * S.ldx + c;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nvalid) a[u] = cape_ld(xr + u);
                    }
                    v4 = make_float4(a[0], a[1], a[2], a[3]);
                }
            }
            *reinterpret_cast<float4 *>(&sA[rl * LDA + 4 * q]) = v4;
        }
        // B chunk: RK rows of dz x FT channels
        if (p.dzvec && ((p.F & 3) == 0)) {
            constexpr int NB = RK * (FT / 4) / 256;
            float4 vb4[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int idx = tid + i * 256;
                const int rl = idx / (FT / 4), q = idx % (FT / 4);
                const int r = rbase + rl;
                const int rc = r < rb ? r : rb - 1;
                const int fc = (f0 + 4 * q) < p.F ? (f0 + 4 * q) : 0;
                vb4[i] = cape_ld4(dzb + (long long)rc * p.lddz + fc);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int idx = tid + i * 256;
                const int rl = idx / (FT / 4), q = idx % (FT / 4);
                const bool ok = ((rbase + rl) < rb) && ((f0 + 4 * q) < p.F);
                float4 o = vb4[i];
                o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
                *reinterpret_cast<float4 *>(&sB[rl * LDB + 4 * q]) = o;
            }
        } else
        for (int idx = tid; idx < RK * (FT / 4); idx += 256) {
            const int rl = idx / (FT / 4), q = idx % (FT / 4);
            const int r = rbase + rl;
            const int f = f0 + 4 * q;
            const int nvalid = p.F - f;
            float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rb && nvalid > 0) {
                const AT *zr = dzb + (long long)r * p.lddz + f;
                if (p.dzvec && nvalid >= 4) {
                    v4 = cape_ld4(zr);
                } else {
                    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < nvalid) a[u] = cape_ld(zr + u);
                    v4 = make_float4(a[0], a[1], a[2], a[3]);
                }
            }
            *reinterpret_cast<float4 *>(&sB[rl * LDB + 4 * q]) = v4;
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < RK / 8; ++kb) {
            float av[TM][4], bv[TN][4];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int u = 0; u < 4; ++u) av[a][u] = sA[(kb * 8 + 4 * lh + u) * LDA + wm * WTM + a * 32 + li];
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int u = 0; u < 4; ++u) bv[b][u] = sB[(kb * 8 + 4 * lh + u) * LDB + wn * WTN + b * 32 + li];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][u], bv[b][u], acc[a][b], 0, 0, 0);
        }
    }
    }   // samples of this group

    // partial slab layout: [split][part_off[si] + c*F + f]
    float *out = p.ws + (long long)split * p.slab + p.part_off[si];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = f0 + wn * WTN + b * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int c = c0 + wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (c < S.C && f < p.F) out[(long long)c * p.F + f] = acc[a][b][g];
            }
        }
}

struct DwReduceParams {
    float *w[CAPE_MAX_SRC];
    long long wrs[CAPE_MAX_SRC], wcs[CAPE_MAX_SRC];
    long long part_off[CAPE_MAX_SRC + 1];
    int nsrc, F, nsplit, accumulate;
    const float *ws;
    long long slab;
};

// block = 16 consecutive output elements x 16 split lanes; fixed summation order (deterministic)
__device__ __forceinline__ void dw_reduce_body(const DwReduceParams &p, long long block) {
    __shared__ float red[16][17];
    const long long total = p.part_off[p.nsrc];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long long i = block * 16 + el;
    float sum = 0.f;
    if (i < total) {
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sp = sl;
        for (; sp + 48 < p.nsplit; sp += 64) {
            sum += p.ws[(long long)sp * p.slab + i];
            s1 += p.ws[(long long)(sp + 16) * p.slab + i];
            s2 += p.ws[(long long)(sp + 32) * p.slab + i];
            s3 += p.ws[(long long)(sp + 48) * p.slab + i];
        }
        for (; sp < p.nsplit; sp += 16) sum += p.ws[(long long)sp * p.slab + i];
        sum = (sum + s1) + (s2 + s3);
    }
    red[sl][el] = sum;
    __syncthreads();
    if (sl == 0 && i < total) {
        int si = 0;
        while (si + 1 < p.nsrc && i >= p.part_off[si + 1]) ++si;
        const long long loc = i - p.part_off[si];
        const long long c = loc / p.F, f = loc % p.F;
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[l][el];
        float *dst = p.w[si] + c * p.wrs[si] + f * p.wcs[si];
        *dst = p.accumulate ? (*dst + t) : t;
    }
}

__global__ __launch_bounds__(256) void dw_reduce_kernel(DwReduceParams p) { dw_reduce_body(p, blockIdx.x); }

// Vector form: thread = 4 consecutive output elements (one float4 of a weight-gradient row), all splits summed
// in order with four independent partial sums -- fully coalesced slab reads.  Needs F % 4 == 0, unit column
// stride and 16-byte aligned destinations (the layer weights in the gradient bucket).
__device__ __forceinline__ void dw_reduce_vec_body(const DwReduceParams &p, long long block) {
    const long long total4 = p.part_off[p.nsrc] >> 2;
    const long long q = block * 256 + threadIdx.x;
    if (q >= total4) return;
    const long long i = q << 2;
    const float4 *src = reinterpret_cast<const float4 *>(p.ws + i);
    const long long step = p.slab >> 2;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int sp = 0;
    for (; sp + 3 < p.nsplit; sp += 4) {
        const float4 v0 = src[(long long)sp * step], v1 = src[(long long)(sp + 1) * step];
        const float4 v2 = src[(long long)(sp + 2) * step], v3 = src[(long long)(sp + 3) * step];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; sp < p.nsplit; ++sp) {
        const float4 v0 = src[(long long)sp * step];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    float4 t;
    t.x = (a0.x + a1.x) + (a2.x + a3.x); t.y = (a0.y + a1.y) + (a2.y + a3.y);
    t.z = (a0.z + a1.z) + (a2.z + a3.z); t.w = (a0.w + a1.w) + (a2.w + a3.w);
    int si = 0;
    while (si + 1 < p.nsrc && i >= p.part_off[si + 1]) ++si;
    const long long loc = i - p.part_off[si];
    const long long c = loc / p.F, f = loc % p.F;
    float4 *dst = reinterpret_cast<float4 *>(p.w[si] + c * p.wrs[si] + f);
    if (p.accumulate) {
        const float4 o = *dst;
        t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    *dst = t;
}

__global__ __launch_bounds__(256) void dw_reduce_vec_kernel(DwReduceParams p) { dw_reduce_vec_body(p, blockIdx.x); }

// The slab reductions of SEVERAL weight-gradient launches in one dispatch (their partial slabs stay in their workspaces
// until then): per layer the reduction is a ~5 us dispatch whose result is only needed at the end of the backward pass.
struct DwReduceBatch {
    DwReduceParams it[CAPE_MAX_DW_REDUCE_ITEMS];
    int blk_off[CAPE_MAX_DW_REDUCE_ITEMS + 1];
    int vec[CAPE_MAX_DW_REDUCE_ITEMS];
    int n;
};

__global__ __launch_bounds__(256) void dw_reduce_batch_kernel(DwReduceBatch B) {
    int i = 0;
    while (i + 1 < B.n && (int)blockIdx.x >= B.blk_off[i + 1]) ++i;
    const long long b = (long long)blockIdx.x - B.blk_off[i];
    if (B.vec[i]) dw_reduce_vec_body(B.it[i], b);       // (uniform per block)
    else dw_reduce_body(B.it[i], b);
}

// es = bytes per activation element (4: fp32, 2: bf16 storage)
inline int fill_src(SrcDev &d, const cape_src_t &s, int es = 4) {
    if (!s.x || s.C <= 0 || s.ldx < s.C) return CAPE_EINVAL;
    if (s.rowptr && (!s.colidx || !s.vals)) return CAPE_EINVAL;
    d.x = s.x; d.xs = s.x_sample_stride; d.ldx = s.ldx; d.C = s.C;
    d.rp = s.rowptr; d.ci = s.colidx; d.va = s.vals;
    d.w = s.w; d.wrs = s.w_rs; d.wcs = s.w_cs;
    d.w2 = s.w2; d.w2rs = s.w2_rs; d.w2cs = s.w2_cs;
    d.vec = ((s.ldx & 3) == 0) && ((s.x_sample_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(s.x) & (4 * es - 1)) == 0);
    d.wh = d.wl = d.wh2 = d.wl2 = nullptr;
    d.wp = d.wp2 = 0;
    d.rm = nullptr;
    d.rmw = 0;
    return CAPE_OK;
}

// fp16 two-piece operands of a launch (cape_h2_t, may be null): piece planes / row bounds per source, column scales, the
// row-bound output of the epilogue
inline int fill_h2(GconvParams &p, const cape_h2_t *h2, bool dual, int out_deinterleave) {
    p.wsi = p.wsi2 = nullptr;
    p.rm_out = nullptr;
    p.rm_out_w = 0;
    if (!h2) return CAPE_OK;
    if (h2->rowmax_out) {
        if (out_deinterleave > 1 || h2->rowmax_out_w < (p.F + 31) / 32 || (h2->rowmax_out_w & 3)) return CAPE_EINVAL;
        p.rm_out = h2->rowmax_out;
        p.rm_out_w = h2->rowmax_out_w;
    }
    if (!h2->src) return CAPE_OK;
    p.wsi = h2->wscale_inv;
    p.wsi2 = dual ? h2->w2scale_inv : nullptr;
    for (int i = 0; i < p.nsrc; ++i) {
        const cape_h2_src_t &h = h2->src[i];
        SrcDev &d = p.s[i];
        d.wh = h.w_hi; d.wl = h.w_lo; d.wp = h.w_pitch;
        d.wh2 = d.w2 ? h.w2_hi : nullptr; d.wl2 = d.w2 ? h.w2_lo : nullptr; d.wp2 = h.w2_pitch;
        d.rm = h.rowmax; d.rmw = h.rowmax_w;
    }
    return CAPE_OK;
}

struct DwPlan {
    int ct, ft, ctiles[CAPE_MAX_SRC], ftiles, ntiles, rsplit, rows_per_split, ngroups, samples_per_group;
    int vstart[CAPE_MAX_SRC + 1];      // plain kernel: virtual channel axis (see DwParams)
    long long slab;
};

// slots: workgroups the launch aims at.  512 = two per CU for the serial kernels (dw_split / dw_plain / dw_packed / gather: a
// second workgroup hides their staging phase); 256 = one per CU for the pipelined dw_h2_kernel, which overlaps its phases
// inside each wave -- half the partial slabs to write and to reduce: step 2.697 -> 2.670 ms in three alternating A/B pairs on one
// box, the kernel itself 33 -> 37 us (profiles/r05_exp_dw_slots.txt); the same setting costs the serial bf16 kernels 5 %.
constexpr int DW_SLOTS_SERIAL = 512, DW_SLOTS_PIPELINED = 256;
inline void plan_dw_splits(int N, int Mo, DwPlan &pl, int slots = DW_SLOTS_SERIAL) {
    // ONE round of workgroups (two per CU = 512 slots) with equal contraction lengths: split over the samples first
    // (the largest divisor of N that fits: every group gets the same number of samples), then the vertex range into
    // equal row blocks.  (An older rule rounded the split count UP, so the widest layers ran 672 workgroups of 24 chunks
    // on 512 slots -- one and a third rounds, groups of 6 / 6 / 4 samples -- where 512 of 27 chunks do: 50.7 -> 43.8 us.)
    int S = slots / pl.ntiles;            // (768 / 1024 slots for the smaller tiles measured slower)
    if (S < 1) S = 1;
    int ngroups = 1;
    for (int d = 1; d <= N && d <= S; ++d)
        if (N % d == 0) ngroups = d;
    const int maxr = (Mo + 127) / 128;
    int rsplit = S / ngroups;
    if (rsplit > maxr) rsplit = maxr;
    if (rsplit < 1) rsplit = 1;
    int rows = (Mo + rsplit - 1) / rsplit;
    rows = ((rows + 31) / 32) * 32;
    pl.rows_per_split = rows;
    pl.rsplit = (Mo + rows - 1) / rows;
    pl.samples_per_group = N / ngroups;
    pl.ngroups = ngroups;
}

// gather kernel: one [ct x ft] tile grid per source
inline void plan_dw(const cape_src_t *srcs, int nsrc, int N, int Mo, int F, DwPlan &pl) {
    int maxC = 0;
    for (int i = 0; i < nsrc; ++i) maxC = srcs[i].C > maxC ? srcs[i].C : maxC;
    pl.ct = (maxC <= 64) ? 64 : 128;
    pl.ft = (F <= 64) ? 64 : 128;
    pl.ftiles = (F + pl.ft - 1) / pl.ft;
    pl.ntiles = 0;
    pl.slab = 0;
    for (int i = 0; i < nsrc; ++i) {
        pl.ctiles[i] = (srcs[i].C + pl.ct - 1) / pl.ct;
        pl.ntiles += pl.ctiles[i] * pl.ftiles;
        pl.slab += (long long)srcs[i].C * F;
    }
    plan_dw_splits(N, Mo, pl);
}

// pipelined plain kernel: tiles over the virtual channel axis.  Sources that share dz are packed back to back, so
// the narrow layers (3 x 32 channels at 6890 vertices) run as ONE tile that reads dz once instead of three
// half-empty tiles; with dz2 in play every source starts on a tile boundary (a tile has one gradient operand).
inline void plan_dw_plain(const cape_src_t *srcs, int nsrc, int N, int Mo, int F, bool pack, DwPlan &pl) {
    int sumC = 0, maxC = 0;
    pl.slab = 0;
    for (int i = 0; i < nsrc; ++i) {
        sumC += srcs[i].C;
        maxC = srcs[i].C > maxC ? srcs[i].C : maxC;
        pl.slab += (long long)srcs[i].C * F;
    }
    const int span = pack ? sumC : maxC;
    const int n128 = (span + 127) / 128, n64 = (span + 63) / 64;
    pl.ft = (F <= 32) ? 32 : (F <= 64) ? 64 : 128;
    pl.ct = (pl.ft == 32 || n64 == 2 * n128) ? 128 : 64;      // 64-wide tiles only where they save MFMA work
    pl.ftiles = (F + pl.ft - 1) / pl.ft;
    int v = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (!pack) v = (v + pl.ct - 1) / pl.ct * pl.ct;
        pl.vstart[i] = v;
        v += srcs[i].C;
    }
    pl.vstart[nsrc] = v;
    pl.ntiles = ((v + pl.ct - 1) / pl.ct) * pl.ftiles;
    plan_dw_splits(N, Mo, pl);
}

// Plain = no gather and float4-addressable rows.  A channel count that is not a multiple of 4 qualifies when the
// row is padded to one (ld >= round_up(C, 4)): the pad lane only feeds an output row that is never stored
// (the 3-channel network input / output layers).
inline bool dw_srcs_plain(const cape_src_t *srcs, int nsrc, int es = 4) {
    for (int i = 0; i < nsrc; ++i)
        if (srcs[i].rowptr || (((srcs[i].C + 3) & ~3) > srcs[i].ldx) || (srcs[i].ldx & 3) || (srcs[i].x_sample_stride & 3) ||
            (reinterpret_cast<uintptr_t>(srcs[i].x) & (4 * es - 1)))
            return false;
    return true;
}


inline bool dw_dz_vec(const float *dz, int64_t dz_sample_stride, int32_t lddz, const float *dz2, int es = 4) {
    return ((lddz & 3) == 0) && ((dz_sample_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(dz) & (4 * es - 1)) == 0) &&
           (!dz2 || (reinterpret_cast<uintptr_t>(dz2) & (4 * es - 1)) == 0);
}

// row bounds of the weight gradient's operands (cape_h2_dw_t, may be null)
inline void fill_h2_dw(DwParams &p, const cape_h2_dw_t *h2) {
    p.dzrm = p.dz2rm = nullptr;
    p.dzrmw = p.dz2rmw = 0;
    if (!h2) return;
    for (int i = 0; i < p.nsrc; ++i) { p.s[i].rm = h2->src_rowmax[i]; p.s[i].rmw = h2->src_rowmax_w[i]; }
    p.dzrm = h2->dz_rowmax; p.dzrmw = h2->dz_rowmax_w;
    p.dz2rm = h2->dz2_rowmax; p.dz2rmw = h2->dz2_rowmax_w;
}

// Kernel choice of one weight-gradient launch (pure function of the arguments): 0 = gather form (gconv_dw_kernel),
// 1 = pipelined plain kernel on the exact-fp32 MFMA (dw_plain_kernel), 2 = packed narrow sources (dw_packed_kernel),
// 3 = plain sources on the bf16 pipe with the exact three-way operand split (dw_split_kernel).  Fills the tile plan.
inline int choose_dw(const cape_src_t *srcs, int nsrc, const float *dz, int64_t dz_sample_stride, int32_t lddz,
                     const float *dz2, uint32_t dz2_mask, int N, int Mo, int F, DwPlan &pl, bool bf16 = false) {
    static const int dwp_on = getenv("CAPE_DW_PLAIN") ? atoi(getenv("CAPE_DW_PLAIN")) : 1;      // 0: A/B against the gather kernel
    const int es = bf16 ? 2 : 4;
    const bool dzvec = dw_dz_vec(dz, dz_sample_stride, lddz, dz2, es);
    const bool plain = dwp_on && dzvec && (((F + 3) & ~3) <= lddz) && dw_srcs_plain(srcs, nsrc, es);
    if (bf16) {
        // bf16 storage: the split-pipe kernel with one plane where it applies (plain sources, whole float4-equivalent
        // columns, even F); the 3-channel ends take the pipelined fp32-MFMA kernels like the fp32 path (packed when
        // several narrow sources share a tile, plain otherwise); the generic gather kernel for everything else
        bool c4b = true;
        int sumCb = 0;
        for (int i = 0; i < nsrc; ++i) { c4b = c4b && (srcs[i].C & 3) == 0; sumCb += srcs[i].C; }
        if (plain && c4b && (F & 1) == 0) {
            plan_dw(srcs, nsrc, N, Mo, F, pl);
            return 3;
        }
        if (plain && c4b && !(dz2 && dz2_mask) && nsrc > 1 && sumCb <= 128 && F <= 64) {
            plan_dw_plain(srcs, nsrc, N, Mo, F, true, pl);
            return 2;
        }
        plan_dw(srcs, nsrc, N, Mo, F, pl);
        return plain ? 1 : 0;
    }
    int sumC = 0;
    for (int i = 0; i < nsrc; ++i) sumC += srcs[i].C;
    // packing pays where several sources fit ONE tile (narrow layers of the fine mesh levels)
    // (and the output is narrow: with F > 64 the single packed tile over-splits the rows -- measured 1.8x slower)
    bool c4 = true;
    for (int i = 0; i < nsrc; ++i) c4 = c4 && (srcs[i].C & 3) == 0;
    const bool packed = plain && c4 && !(dz2 && dz2_mask) && nsrc > 1 && sumC <= 128 && F <= 64;
    // un-packed plain launches run on the bf16 pipe (dw_split_kernel); CAPE_DW_BF16X6=0 keeps them on the fp32 MFMA
    static const int dws_on = getenv("CAPE_DW_BF16X6") ? atoi(getenv("CAPE_DW_BF16X6")) : CAPE_DW_BF16X6_DEFAULT;
    const bool dw_split = dws_on && plain && !packed && c4 && (F & 1) == 0;
    if (packed) plan_dw_plain(srcs, nsrc, N, Mo, F, true, pl);
    else plan_dw(srcs, nsrc, N, Mo, F, pl);
    return packed ? 2 : dw_split ? 3 : plain ? 1 : 0;
}

// Kernel choice of one forward launch (pure function of the arguments).
struct FwdPlan {
    int family;   // 4: plain sources with <= 8 input channels in total (fwd_narrow_in_kernel; plain epilogue only, else family 1)
                  // 0: gather-GEMM (gconv_fwd_kernel), 1: pipelined plain GEMM (gemm_plain_kernel),
                  // 2: plain GEMM on the bf16 matrix pipe with exact three-way operand split (gemm_split_kernel)
                  // 3: plain GEMM as three fp16 products on two-piece operands (gemm_h2_kernel; needs cape_h2_t operands)
    int BM, BN;
    int layout;   // family 1: 1 = weights contraction-contiguous, 0 = output-contiguous
};

inline FwdPlan plan_fwd(const GconvParams &p, const cape_src_t *srcs, bool dual, bool bf16 = false) {
    FwdPlan pl;
    static const int gp_on = getenv("CAPE_GEMM_PLAIN") ? atoi(getenv("CAPE_GEMM_PLAIN")) : 1;   // 0: A/B against the gather kernel
    pl.layout = gp_on ? gp_weight_layout(p, dual) : -1;
    if (bf16) {
        // bf16 storage: the split-pipe kernel with one plane where its staging applies (plain sources of whole 32-channel
        // chunks whose rows can be read 8 elements = 16 bytes at a time, F >= 64), the generic gather kernel otherwise
        bool ok = pl.layout >= 0 && p.F >= 64;
        for (int i = 0; i < p.nsrc && ok; ++i) {
            const long long ws = srcs[i].w_cs > srcs[i].w_rs ? srcs[i].w_cs : srcs[i].w_rs;
            ok = (srcs[i].ldx & 7) == 0 && (srcs[i].x_sample_stride & 7) == 0 && (reinterpret_cast<uintptr_t>(srcs[i].x) & 15) == 0 &&
                 (long long)p.Mo * srcs[i].ldx < (1LL << 31) && (long long)p.F * ws < (1LL << 31) &&
                 srcs[i].C % GS_KC == 0 && srcs[i].C >= GS_KC;
        }
        if (ok) {
            pl.family = 2;
            gs_tile(dual, p.N, p.Mo, p.F, pl.BM, pl.BN);
            return pl;
        }
        pl.layout = -1;
    }
    for (int i = 0; i < p.nsrc && pl.layout >= 0; ++i) {
        const long long ws = srcs[i].w_cs > srcs[i].w_rs ? srcs[i].w_cs : srcs[i].w_rs;
        if ((long long)p.Mo * srcs[i].ldx >= (1LL << 31) || (long long)p.F * ws >= (1LL << 31)) pl.layout = -1;   // 32-bit offsets
    }
    if (pl.layout >= 0 && h2_eligible(p, dual)) {
        pl.family = 3;
        pl.layout = 1;                                  // the piece planes are contraction-contiguous in every launch form
        int ktot = 0;
        for (int i = 0; i < p.nsrc; ++i) ktot += p.s[i].C;
        h2_tile(dual, p.N, p.Mo, p.F, ktot, pl.BM, pl.BN);
        return pl;
    }
    if (pl.layout >= 0) {
        // CAPE_GEMM_BF16X6=0: keep every contraction on the exact-fp32 MFMA (A/B switch)
        static const int gs_on = getenv("CAPE_GEMM_BF16X6") ? atoi(getenv("CAPE_GEMM_BF16X6")) : CAPE_GEMM_BF16X6_DEFAULT;
        if (gs_on && gs_eligible(p, dual)) {
            pl.family = 2;
            gs_tile(dual, p.N, p.Mo, p.F, pl.BM, pl.BN);
            return pl;
        }
        // at most 8 input channels over all sources (the 3-channel network inputs): the narrow side in registers (narrow.h)
        static const int narrow_on = getenv("CAPE_NARROW") ? atoi(getenv("CAPE_NARROW")) : 1;
        int sumC = 0;
        for (int i = 0; i < p.nsrc; ++i) sumC += p.s[i].C;
        if (narrow_on && !dual && sumC <= NARROW_MAXC && (p.F & 3) == 0 && p.F >= 16 && p.F <= 256 &&
            (long long)p.N * p.Mo < (1LL << 31)) {
            pl.family = 4;
            pl.BM = 256 / narrow_lpr(p.F / 4); pl.BN = p.F;
            return pl;
        }
        pl.family = 1;
        gp_tile(dual, p.F, pl.BM, pl.BN);
        return pl;
    }
    pl.family = 0;
    pl.BN = (p.F <= 32) ? 32 : (p.F <= 64 || dual) ? 64 : 128;
    pl.BM = 128;
    // small meshes: 128-row tiles leave <= 2 workgroups per CU (no overlap partner while staging);
    // 64-row tiles double the resident workgroups at the price of re-reading the weight tile
    constexpr int bm64_below = 640;
    if (pl.BN == 128 && (long long)p.N * ((p.Mo + 127) / 128) * ((p.F + 127) / 128) < bm64_below) pl.BM = 64;
    return pl;
}

inline int fill_fwd_srcs(GconvParams &p, const cape_src_t *srcs, int nsrc, bool &dual, int es = 4) {
    dual = false;
    for (int i = 0; i < nsrc; ++i) {
        if (!srcs[i].w) return CAPE_EINVAL;
        int rc = fill_src(p.s[i], srcs[i], es);
        if (rc) return rc;
        dual = dual || (srcs[i].w2 != nullptr);
    }
    p.nsrc = nsrc;
    return CAPE_OK;
}

}  // namespace

namespace {

int gconv_fwd_plan_impl(const cape_src_t *srcs, int32_t nsrc, int32_t N, int32_t Mo, int32_t F, int32_t plan[4], bool bf16,
                        const cape_h2_t *h2 = nullptr) {
    if (!srcs || nsrc < 1 || nsrc > CAPE_MAX_SRC || N < 1 || Mo < 1 || F < 1 || !plan) return CAPE_EINVAL;
    GconvParams p;
    bool dual;
    int rc = fill_fwd_srcs(p, srcs, nsrc, dual, bf16 ? 2 : 4);
    if (rc) return rc;
    p.N = N; p.Mo = Mo; p.F = F;
    rc = fill_h2(p, bf16 ? nullptr : h2, dual, 0);
    if (rc) return rc;
    const FwdPlan pl = plan_fwd(p, srcs, dual, bf16);
    plan[0] = pl.family; plan[1] = pl.BM; plan[2] = pl.BN; plan[3] = pl.family ? pl.layout : 0;
    return CAPE_OK;
}

template <typename AT>
void launch_gather_fwd(const GconvParams &p, const FwdPlan &pl, bool dual, dim3 grid, hipStream_t st) {
    const dim3 block(256);
    if (!dual) {
        if (pl.BN == 32) CAPE_LAUNCH((gconv_fwd_kernel<128, 32, 4, 1, false, KC_DEFAULT, AT>), grid, block, 0, st, p);
        else if (pl.BN == 64) CAPE_LAUNCH((gconv_fwd_kernel<128, 64, 4, 1, false, KC_DEFAULT, AT>), grid, block, 0, st, p);
        else if (pl.BM == 64) CAPE_LAUNCH((gconv_fwd_kernel<64, 128, 2, 2, false, KC_DEFAULT, AT>), grid, block, 0, st, p);
        else CAPE_LAUNCH((gconv_fwd_kernel<128, 128, 2, 2, false, KC_DEFAULT, AT>), grid, block, 0, st, p);
    } else {
        if (pl.BN == 32) CAPE_LAUNCH((gconv_fwd_kernel<128, 32, 4, 1, true, KC_DEFAULT, AT>), grid, block, 0, st, p);
        else CAPE_LAUNCH((gconv_fwd_kernel<128, 64, 4, 1, true, KC_DEFAULT, AT>), grid, block, 0, st, p);
    }
}

int gconv_fwd_impl(const cape_src_t *srcs, int32_t nsrc, float *y, int64_t y_sample_stride,
                   int32_t ldy, int32_t N, int32_t Mo, int32_t F, const float *bias,
                   int32_t bias_mode, int32_t act, uint32_t *mask_out, const cape_rank_t *rank,
                   int32_t out_deinterleave, void *stream, bool bf16, const cape_h2_t *h2 = nullptr) {
    if (!srcs || nsrc < 1 || nsrc > CAPE_MAX_SRC || !y || N < 1 || Mo < 1 || F < 1) return CAPE_EINVAL;
    if (bf16 && h2) return CAPE_EINVAL;
    const int dK = out_deinterleave > 1 ? out_deinterleave : 1;
    const int dstride = dK > 1 ? ((F / dK + 3) & ~3) : 0;
    if (dK > 1 && (F % dK != 0 || mask_out || rank || ldy < dK * dstride)) return CAPE_EINVAL;
    if (dK == 1 && ldy < F) return CAPE_EINVAL;
    if (bias_mode != CAPE_BIAS_NONE && !bias) return CAPE_EINVAL;
    if (act < CAPE_ACT_NONE || act > CAPE_ACT_TANH) return CAPE_EINVAL;
    GconvParams p;
    bool dual;
    int rc = fill_fwd_srcs(p, srcs, nsrc, dual, bf16 ? 2 : 4);
    if (rc) return rc;
    if (mask_out && !dual) return CAPE_EINVAL;
    if (dual && (bias_mode != CAPE_BIAS_NONE || act != CAPE_ACT_NONE)) return CAPE_EINVAL;
    p.y = y; p.ys = y_sample_stride; p.ldy = ldy;
    p.N = N; p.Mo = Mo; p.F = F;
    p.bias = bias; p.bias_mode = bias ? bias_mode : CAPE_BIAS_NONE; p.act = act;
    p.mask = mask_out; p.mask_words = (F + 31) / 32;
    p.rankR = 0; p.rowscale = nullptr; p.coef = nullptr; p.rank_to2 = 0;
    p.deintK = dK; p.deint_stride = dstride;
    if (rank && rank->R > 0) {
        if (rank->R > CAPE_MAX_SRC || !rank->rowscale || !rank->coef) return CAPE_EINVAL;
        if (rank->to_acc2 && !dual) return CAPE_EINVAL;
        p.rankR = rank->R; p.rowscale = rank->rowscale; p.coef = rank->coef; p.rank_to2 = rank->to_acc2;
    }
    rc = fill_h2(p, h2, dual, dK);
    if (rc) return rc;
    FwdPlan pl = plan_fwd(p, srcs, dual, bf16);
    if (pl.family == 4) {
        // the narrow kernel has the plain epilogue (bias + activation, float4 stores); anything else takes the tile kernel
        const bool ok = !p.rankR && !p.mask && dK == 1 && !p.rm_out && (ldy & 3) == 0 && (y_sample_stride & 3) == 0 &&
                        (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
        if (ok) {
            const int lpr = narrow_lpr(F / 4), RL = 256 / lpr;
            long long blocks = ((long long)N * Mo + 4 * RL - 1) / (4 * RL);       // four rows per thread and pass
            if (blocks > 1024) blocks = 1024;                                     // (per-block set-up: weights into LDS, pointer table)
            CAPE_LAUNCH(fwd_narrow_in_kernel, dim3((unsigned)blocks), dim3(256), (size_t)NARROW_MAXC * 4 * lpr * sizeof(float),
                        (hipStream_t)stream, p, lpr);
            CAPE_LAUNCH_CHECK();
            return CAPE_OK;
        }
        pl.family = 1;
        gp_tile(dual, p.F, pl.BM, pl.BN);
    }
    p.row_tiles = (Mo + pl.BM - 1) / pl.BM;
    p.col_tiles = (F + pl.BN - 1) / pl.BN;
    dim3 grid((unsigned)(N * p.row_tiles * p.col_tiles));
    hipStream_t st = (hipStream_t)stream;
    if (pl.family == 3) {
        h2_launch(p, dual, pl.BM, pl.BN, grid, st);
    } else if (pl.family == 2) {
        gs_launch(p, dual, pl.BM, pl.layout, bf16, grid, st);
    } else if (pl.family == 1) {
        gp_launch(p, dual, pl.BM, pl.BN, pl.layout, grid, st);
    } else if (bf16) {
        launch_gather_fwd<cape_bf16>(p, pl, dual, grid, st);
    } else {
        launch_gather_fwd<float>(p, pl, dual, grid, st);
    }
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

}  // namespace

extern "C" int cape_gconv_fwd_plan(const cape_src_t *srcs, int32_t nsrc, int32_t N, int32_t Mo, int32_t F, int32_t plan[4]) {
    return gconv_fwd_plan_impl(srcs, nsrc, N, Mo, F, plan, false);
}

extern "C" int cape_gconv_fwd_plan_bf16(const cape_src_t *srcs, int32_t nsrc, int32_t N, int32_t Mo, int32_t F, int32_t plan[4]) {
    return gconv_fwd_plan_impl(srcs, nsrc, N, Mo, F, plan, true);
}

extern "C" int cape_gconv_fwd(const cape_src_t *srcs, int32_t nsrc, float *y, int64_t y_sample_stride,
                              int32_t ldy, int32_t N, int32_t Mo, int32_t F, const float *bias,
                              int32_t bias_mode, int32_t act, uint32_t *mask_out, const cape_rank_t *rank,
                              int32_t out_deinterleave, void *stream) {
    return gconv_fwd_impl(srcs, nsrc, y, y_sample_stride, ldy, N, Mo, F, bias, bias_mode, act, mask_out, rank, out_deinterleave,
                          stream, false);
}

extern "C" int cape_gconv_fwd_h2(const cape_src_t *srcs, int32_t nsrc, float *y, int64_t y_sample_stride,
                                 int32_t ldy, int32_t N, int32_t Mo, int32_t F, const float *bias,
                                 int32_t bias_mode, int32_t act, uint32_t *mask_out, const cape_rank_t *rank,
                                 int32_t out_deinterleave, const cape_h2_t *h2, void *stream) {
    return gconv_fwd_impl(srcs, nsrc, y, y_sample_stride, ldy, N, Mo, F, bias, bias_mode, act, mask_out, rank, out_deinterleave,
                          stream, false, h2);
}

extern "C" int cape_gconv_fwd_plan_h2(const cape_src_t *srcs, int32_t nsrc, int32_t N, int32_t Mo, int32_t F, const cape_h2_t *h2,
                                      int32_t plan[4]) {
    return gconv_fwd_plan_impl(srcs, nsrc, N, Mo, F, plan, false, h2);
}

extern "C" int cape_gconv_fwd_bf16(const cape_src_t *srcs, int32_t nsrc, void *y, int64_t y_sample_stride,
                                   int32_t ldy, int32_t N, int32_t Mo, int32_t F, const float *bias,
                                   int32_t bias_mode, int32_t act, uint32_t *mask_out, const cape_rank_t *rank,
                                   int32_t out_deinterleave, void *stream) {
    return gconv_fwd_impl(srcs, nsrc, (float *)y, y_sample_stride, ldy, N, Mo, F, bias, bias_mode, act, mask_out, rank,
                          out_deinterleave, stream, true);
}

extern "C" int64_t cape_gconv_dw_workspace_bytes(const cape_src_t *srcs, int32_t nsrc, int32_t N,
                                                 int32_t Mo, int32_t F) {
    if (!srcs || nsrc < 1 || nsrc > CAPE_MAX_SRC || N < 1 || Mo < 1 || F < 1) return CAPE_EINVAL;
    DwPlan pl;
    plan_dw(srcs, nsrc, N, Mo, F, pl);
    long long need = pl.slab * pl.ngroups * pl.rsplit;
    {   // whichever kernel the launch ends up taking (depends on dz and on the storage type too): the larger plan
        plan_dw_plain(srcs, nsrc, N, Mo, F, true, pl);
        const long long n2 = pl.slab * pl.ngroups * pl.rsplit;
        need = n2 > need ? n2 : need;
    }
    return (int64_t)need * (int64_t)sizeof(float);
}

namespace {
// does this launch run as dw_h2_kernel (fp16 two-piece operands)?  Decided from the arguments alone, identically by the plan
// query, the contraction stage and the reduction stages (they must agree on the number of partial slabs)
inline bool dw_takes_h2(const cape_src_t *srcs, int nsrc, const float *dz2, uint32_t dz2_mask, int Mo, int lddz, const cape_h2_dw_t *h2) {
    if (!h2) return false;
    DwParams p;
    p.nsrc = nsrc;
    p.dz2_mask = dz2 ? dz2_mask : 0u;
    p.Mo = Mo; p.lddz = lddz;                                     // (what h2_dw_eligible reads besides the row bounds)
    for (int i = 0; i < nsrc; ++i) p.s[i].ldx = srcs[i].ldx;
    fill_h2_dw(p, h2);
    return h2_dw_eligible(p);
}
int gconv_dw_plan_impl(const cape_src_t *srcs, int32_t nsrc, const float *dz, int64_t dz_sample_stride,
                       int32_t lddz, const float *dz2, uint32_t dz2_mask, int32_t N, int32_t Mo, int32_t F,
                       int32_t plan[4], bool bf16, const cape_h2_dw_t *h2 = nullptr) {
    if (!srcs || nsrc < 1 || nsrc > CAPE_MAX_SRC || !dz || N < 1 || Mo < 1 || F < 1 || lddz < F || !plan) return CAPE_EINVAL;
    DwPlan pl;
    plan[0] = choose_dw(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, N, Mo, F, pl, bf16);
    // (the narrow kernels are serial kernels: they keep the 512-slot plan even when the launch carries row bounds -- ADVICE r05)
    const int nm = dw_narrow_mode(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, F, plan[0], bf16);
    if (nm) {
        plan[0] = nm;
    } else if (plan[0] == 3 && !bf16 && dw_takes_h2(srcs, nsrc, dz2, dz2_mask, Mo, lddz, h2)) {
        plan[0] = 4;
        plan_dw_splits(N, Mo, pl, DW_SLOTS_PIPELINED);
    }
    plan[1] = pl.ct; plan[2] = pl.ft; plan[3] = pl.ngroups * pl.rsplit;
    return CAPE_OK;
}
struct DwReduceParams;
int gconv_dw_stage_impl(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                        int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                        int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                        int64_t workspace_bytes, int32_t stage, void *stream, bool bf16,
                        DwReduceParams *batch_rp = nullptr, int *batch_vec = nullptr, int *batch_blocks = nullptr,
                        const cape_h2_dw_t *h2 = nullptr);
}  // namespace

extern "C" int cape_gconv_dw_plan(const cape_src_t *srcs, int32_t nsrc, const float *dz, int64_t dz_sample_stride,
                                  int32_t lddz, const float *dz2, uint32_t dz2_mask, int32_t N, int32_t Mo, int32_t F,
                                  int32_t plan[4]) {
    return gconv_dw_plan_impl(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, N, Mo, F, plan, false);
}

extern "C" int cape_gconv_dw_plan_h2(const cape_src_t *srcs, int32_t nsrc, const float *dz, int64_t dz_sample_stride,
                                     int32_t lddz, const float *dz2, uint32_t dz2_mask, int32_t N, int32_t Mo, int32_t F,
                                     const cape_h2_dw_t *h2, int32_t plan[4]) {
    return gconv_dw_plan_impl(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, N, Mo, F, plan, false, h2);
}

extern "C" int cape_gconv_dw_stage_h2(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                                      int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                                      int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                                      int64_t workspace_bytes, int32_t stage, const cape_h2_dw_t *h2, void *stream) {
    if (stage > 2) return CAPE_EINVAL;
    return gconv_dw_stage_impl(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, N, Mo, F, accumulate, workspace,
                               workspace_bytes, stage, stream, false, nullptr, nullptr, nullptr, h2);
}

extern "C" int cape_gconv_dw_plan_bf16(const cape_src_t *srcs, int32_t nsrc, const void *dz, int64_t dz_sample_stride,
                                       int32_t lddz, const void *dz2, uint32_t dz2_mask, int32_t N, int32_t Mo, int32_t F,
                                       int32_t plan[4]) {
    return gconv_dw_plan_impl(srcs, nsrc, (const float *)dz, dz_sample_stride, lddz, (const float *)dz2, dz2_mask, N, Mo, F, plan, true);
}

extern "C" int cape_gconv_dw_bf16(const cape_src_t *srcs, int32_t nsrc, const void *dz,
                                  int64_t dz_sample_stride, int32_t lddz, const void *dz2, uint32_t dz2_mask,
                                  int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                                  int64_t workspace_bytes, void *stream) {
    return gconv_dw_stage_impl(srcs, nsrc, (const float *)dz, dz_sample_stride, lddz, (const float *)dz2, dz2_mask, N, Mo, F,
                               accumulate, workspace, workspace_bytes, 0, stream, true);
}

extern "C" int cape_gconv_dw_stage_bf16(const cape_src_t *srcs, int32_t nsrc, const void *dz,
                                        int64_t dz_sample_stride, int32_t lddz, const void *dz2, uint32_t dz2_mask,
                                        int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                                        int64_t workspace_bytes, int32_t stage, void *stream) {
    if (stage > 2) return CAPE_EINVAL;
    return gconv_dw_stage_impl(srcs, nsrc, (const float *)dz, dz_sample_stride, lddz, (const float *)dz2, dz2_mask, N, Mo, F,
                               accumulate, workspace, workspace_bytes, stage, stream, true);
}

extern "C" int cape_gconv_dw(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                             int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                             int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                             int64_t workspace_bytes, void *stream) {
    return cape_gconv_dw_stage(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, N, Mo, F, accumulate, workspace,
                               workspace_bytes, 0, stream);
}

extern "C" int cape_gconv_dw_stage(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                                   int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                                   int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                                   int64_t workspace_bytes, int32_t stage, void *stream) {
    if (stage > 2) return CAPE_EINVAL;
    return gconv_dw_stage_impl(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, N, Mo, F, accumulate, workspace,
                               workspace_bytes, stage, stream, false);
}

namespace {
int gconv_dw_stage_impl(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                        int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                        int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                        int64_t workspace_bytes, int32_t stage, void *stream, bool bf16,
                        DwReduceParams *batch_rp, int *batch_vec, int *batch_blocks, const cape_h2_dw_t *h2) {
    if (stage < 0 || stage > 3 || (stage == 3 && (!batch_rp || !batch_vec || !batch_blocks))) return CAPE_EINVAL;
    if (!srcs || nsrc < 1 || nsrc > CAPE_MAX_SRC || !dz || N < 1 || Mo < 1 || F < 1 || lddz < F || !workspace)
        return CAPE_EINVAL;
    if (dz2_mask && !dz2) return CAPE_EINVAL;
    DwPlan pl;
    const int fam = choose_dw(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, N, Mo, F, pl, bf16);
    const bool packed = fam == 2, plain = fam != 0, dw_split = fam == 3;
    const int narrow = dw_narrow_mode(srcs, nsrc, dz, dz_sample_stride, lddz, dz2, dz2_mask, F, fam, bf16);
    // one workgroup per CU (256 slots) for the pipelined two-piece kernel ONLY: the narrow kernels are serial ones and keep the
    // 512-slot plan whether or not the launch carries row bounds (every stage and the plan query decide this identically)
    const bool use_h2 = !narrow && !bf16 && dw_split && dw_takes_h2(srcs, nsrc, dz2, dz2_mask, Mo, lddz, h2);
    if (use_h2) plan_dw_splits(N, Mo, pl, DW_SLOTS_PIPELINED);
    const bool dzvec = dw_dz_vec(dz, dz_sample_stride, lddz, dz2, bf16 ? 2 : 4);
    const long long need = pl.slab * pl.ngroups * pl.rsplit * (long long)sizeof(float);
    if (workspace_bytes < need) return CAPE_EWORKSPACE;
    DwParams p;
    DwReduceParams rp;
    p.nsrc = nsrc; rp.nsrc = nsrc;
    int toff = 0;
    long long poff = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (!srcs[i].w) return CAPE_EINVAL;
        int rc = fill_src(p.s[i], srcs[i], bf16 ? 2 : 4);
        if (rc) return rc;
        p.tile_off[i] = toff; p.part_off[i] = poff; rp.part_off[i] = poff;
        toff += packed ? 0 : pl.ctiles[i] * pl.ftiles;
        poff += (long long)srcs[i].C * F;
        p.vstart[i] = packed ? pl.vstart[i] : 0;
        rp.w[i] = const_cast<float *>(srcs[i].w); rp.wrs[i] = srcs[i].w_rs; rp.wcs[i] = srcs[i].w_cs;
    }
    p.tile_off[nsrc] = toff; p.part_off[nsrc] = poff; rp.part_off[nsrc] = poff;
    p.vstart[nsrc] = packed ? pl.vstart[nsrc] : 0;
    p.dz = dz; p.dz2 = dz2; p.dz2_mask = dz2 ? dz2_mask : 0u; p.dzs = dz_sample_stride; p.lddz = lddz;
    p.dzvec = dzvec;
    p.N = N; p.Mo = Mo; p.F = F; p.ftiles = pl.ftiles;
    p.rsplit = pl.rsplit; p.rows_per_split = pl.rows_per_split;
    p.ngroups = pl.ngroups; p.samples_per_group = pl.samples_per_group;
    p.ws = (float *)workspace; p.slab = pl.slab;
    fill_h2_dw(p, bf16 ? nullptr : h2);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(pl.ntiles * ((pl.ngroups * pl.rsplit + 7) / 8) * 8)), block(256);     // cape_map_dw_block
    if (stage >= 2) {
        // reduction only: the partial slabs of an earlier stage-1 call with the same arguments are in the workspace
    } else if (narrow == 5) {                                   // narrow.h: same splits and slabs, tile 0's workgroups do all sources
        CAPE_LAUNCH(dw_narrow_in_kernel, grid, block, 32768, st, p, narrow_lpr(F / 4), pl.ntiles);
    } else if (narrow == 6) {
        int sumC = 0;
        for (int i = 0; i < nsrc; ++i) sumC += srcs[i].C;
        CAPE_LAUNCH(dw_narrow_out_kernel, grid, block, 16384, st, p, narrow_lpr(sumC / 4), pl.ntiles);
    } else if (use_h2) {
        h2_dw_launch(p, pl.ct, pl.ft, grid, st);                // fp16 two-piece operands: same tiles, splits and slabs
    } else if (bf16 && dw_split) {
        if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((dw_split_kernel<64, 64, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ct == 64) CAPE_LAUNCH((dw_split_kernel<64, 128, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ft == 64) CAPE_LAUNCH((dw_split_kernel<128, 64, cape_bf16>), grid, block, 0, st, p);
        else CAPE_LAUNCH((dw_split_kernel<128, 128, cape_bf16>), grid, block, 0, st, p);
    } else if (bf16 && packed) {
        if (pl.ft == 32) CAPE_LAUNCH((dw_packed_kernel<128, 32, 4, 1, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((dw_packed_kernel<64, 64, 2, 2, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ct == 64) CAPE_LAUNCH((dw_packed_kernel<64, 128, 2, 2, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ft == 64) CAPE_LAUNCH((dw_packed_kernel<128, 64, 2, 2, cape_bf16>), grid, block, 0, st, p);
        else CAPE_LAUNCH((dw_packed_kernel<128, 128, 2, 2, cape_bf16>), grid, block, 0, st, p);
    } else if (bf16 && plain) {
        if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((dw_plain_kernel<64, 64, 2, 2, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ct == 64) CAPE_LAUNCH((dw_plain_kernel<64, 128, 2, 2, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ft == 64) CAPE_LAUNCH((dw_plain_kernel<128, 64, 2, 2, cape_bf16>), grid, block, 0, st, p);
        else CAPE_LAUNCH((dw_plain_kernel<128, 128, 2, 2, cape_bf16>), grid, block, 0, st, p);
    } else if (bf16) {
        if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((gconv_dw_kernel<64, 64, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ct == 64) CAPE_LAUNCH((gconv_dw_kernel<64, 128, cape_bf16>), grid, block, 0, st, p);
        else if (pl.ft == 64) CAPE_LAUNCH((gconv_dw_kernel<128, 64, cape_bf16>), grid, block, 0, st, p);
        else CAPE_LAUNCH((gconv_dw_kernel<128, 128, cape_bf16>), grid, block, 0, st, p);
    } else
    if (packed) {
        if (pl.ft == 32) CAPE_LAUNCH((dw_packed_kernel<128, 32, 4, 1>), grid, block, 0, st, p);
        else if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((dw_packed_kernel<64, 64, 2, 2>), grid, block, 0, st, p);
        else if (pl.ct == 64) CAPE_LAUNCH((dw_packed_kernel<64, 128, 2, 2>), grid, block, 0, st, p);
        else if (pl.ft == 64) CAPE_LAUNCH((dw_packed_kernel<128, 64, 2, 2>), grid, block, 0, st, p);
        else CAPE_LAUNCH((dw_packed_kernel<128, 128, 2, 2>), grid, block, 0, st, p);
    } else if (plain && dw_split) {
        if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((dw_split_kernel<64, 64>), grid, block, 0, st, p);
        else if (pl.ct == 64) CAPE_LAUNCH((dw_split_kernel<64, 128>), grid, block, 0, st, p);
        else if (pl.ft == 64) CAPE_LAUNCH((dw_split_kernel<128, 64>), grid, block, 0, st, p);
        else CAPE_LAUNCH((dw_split_kernel<128, 128>), grid, block, 0, st, p);
    } else if (plain) {
        if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((dw_plain_kernel<64, 64, 2, 2>), grid, block, 0, st, p);
        else if (pl.ct == 64) CAPE_LAUNCH((dw_plain_kernel<64, 128, 2, 2>), grid, block, 0, st, p);
        else if (pl.ft == 64) CAPE_LAUNCH((dw_plain_kernel<128, 64, 2, 2>), grid, block, 0, st, p);
        else CAPE_LAUNCH((dw_plain_kernel<128, 128, 2, 2>), grid, block, 0, st, p);
    } else
    if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((gconv_dw_kernel<64, 64>), grid, block, 0, st, p);
    else if (pl.ct == 64) CAPE_LAUNCH((gconv_dw_kernel<64, 128>), grid, block, 0, st, p);
    else if (pl.ft == 64) CAPE_LAUNCH((gconv_dw_kernel<128, 64>), grid, block, 0, st, p);
    else CAPE_LAUNCH((gconv_dw_kernel<128, 128>), grid, block, 0, st, p);
    CAPE_LAUNCH_CHECK();
    if (stage == 1) return CAPE_OK;
    rp.F = F; rp.nsplit = pl.ngroups * pl.rsplit; rp.accumulate = accumulate; rp.ws = (const float *)workspace; rp.slab = pl.slab;
    long long total = poff;
    int rblocks = (int)((total + 15) / 16);
    // the float4 form sums a quad's splits serially: only when the slab alone fills the chip (>= 256 blocks of quads);
    // small slabs with many splits (fine mesh levels) take the kernel that also spreads the splits over 16 lanes
    bool rvec = (F & 3) == 0 && (pl.slab & 3) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && total >= 262144;
    for (int i = 0; i < nsrc; ++i)
        rvec = rvec && srcs[i].w_cs == 1 && (srcs[i].w_rs & 3) == 0 && (reinterpret_cast<uintptr_t>(srcs[i].w) & 15) == 0;
    if (stage == 3) {                  // describe the reduction instead of launching it (cape_gconv_dw_reduce_batch)
        *batch_rp = rp;
        *batch_vec = rvec ? 1 : 0;
        *batch_blocks = rvec ? (int)((total / 4 + 255) / 256) : rblocks;
        return CAPE_OK;
    }
    if (rvec) CAPE_LAUNCH(dw_reduce_vec_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, rp);
    else CAPE_LAUNCH(dw_reduce_kernel, dim3(rblocks), dim3(256), 0, st, rp);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
}  // namespace

extern "C" int cape_gconv_dw_reduce_batch(const cape_dw_item_t *items, int32_t nitems, void *stream) {
    if (!items || nitems < 1 || nitems > CAPE_MAX_DW_REDUCE_ITEMS) return CAPE_EINVAL;
    DwReduceBatch B;
    B.n = nitems;
    int off = 0;
    for (int i = 0; i < nitems; ++i) {
        const cape_dw_item_t &t = items[i];
        int blocks = 0;
        const int rc = gconv_dw_stage_impl(t.srcs, t.nsrc, (const float *)t.dz, t.dz_sample_stride, t.lddz, (const float *)t.dz2,
                                           t.dz2_mask, t.N, t.Mo, t.F, t.accumulate, t.workspace, t.workspace_bytes, 3, stream,
                                           t.bf16 != 0, &B.it[i], &B.vec[i], &blocks, t.h2);
        if (rc) return rc;
        B.blk_off[i] = off;
        off += blocks;
    }
    B.blk_off[nitems] = off;
    CAPE_LAUNCH(dw_reduce_batch_kernel, dim3((unsigned)off), dim3(256), 0, (hipStream_t)stream, B);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
