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
#include <stdlib.h>
#ifndef CAPE_SPEC
#define CAPE_SPEC 0
#endif
#ifndef CAPE_PIPE
#define CAPE_PIPE 0
#endif

namespace {

constexpr int KC_DEFAULT = 32;   // contraction chunk staged per iteration (64 for wide, deep layers)

struct SrcDev {
    const float *x;
    long long xs;
    int ldx, C;
    const int *rp;
    const int *ci;
    const float *va;
    const float *w;
    long long wrs, wcs;
    const float *w2;
    long long w2rs, w2cs;
    int vec;   // 1: float4 gathers legal (ldx % 4 == 0, base 16B aligned)
};

struct GconvParams {
    SrcDev s[CAPE_MAX_SRC];
    int nsrc;
    float *y;
    long long ys;
    int ldy;
    int N, Mo, F;
    const float *bias;
    int bias_mode, act;
    unsigned *mask;
    int mask_words;
    int row_tiles, col_tiles;
    int rankR;
    const float *rowscale;
    const float *coef;
    unsigned rank_to2;
};

// ---- A-tile staging: gathered rows -> LDS [ROWS][KC+4] -----------------------------------
template <int ROWS, int LDA, int KC>
__device__ __forceinline__ void stage_gather(float *sA, const SrcDev &S, int n, int r0, int Mo,
                                              int c0, int tid) {
    constexpr int QPR = KC / 4;            // float4 columns per row of the chunk
    constexpr int RP = 256 / QPR;          // rows staged per pass
    const int q = tid % QPR;
    const int rl0 = tid / QPR;
    const int c = c0 + 4 * q;
    const float *xb = S.x + (long long)n * S.xs + c;
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
            v[pass] = *reinterpret_cast<const float4 *>((nvalid > 0 ? xb : S.x + (long long)n * S.xs) + (long long)rc * S.ldx);
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
                    const float4 x0 = *reinterpret_cast<const float4 *>(xb + (long long)S.ci[e] * S.ldx);
                    const float4 x1 = *reinterpret_cast<const float4 *>(xb + (long long)S.ci[e + 1] * S.ldx);
                    acc.x = fmaf(v0, x0.x, acc.x); acc.y = fmaf(v0, x0.y, acc.y);
                    acc.z = fmaf(v0, x0.z, acc.z); acc.w = fmaf(v0, x0.w, acc.w);
                    acc.x = fmaf(v1, x1.x, acc.x); acc.y = fmaf(v1, x1.y, acc.y);
                    acc.z = fmaf(v1, x1.z, acc.z); acc.w = fmaf(v1, x1.w, acc.w);
                }
                if (e < e1) {
                    const float v0 = S.va[e];
                    const float4 x0 = *reinterpret_cast<const float4 *>(xb + (long long)S.ci[e] * S.ldx);
                    acc.x = fmaf(v0, x0.x, acc.x); acc.y = fmaf(v0, x0.y, acc.y);
                    acc.z = fmaf(v0, x0.z, acc.z); acc.w = fmaf(v0, x0.w, acc.w);
                }
            } else {
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                if (S.rp) {
                    const int e1 = S.rp[r + 1];
                    for (int e = S.rp[r]; e < e1; ++e) {
                        const float v = S.va[e];
                        const float *xr = xb + (long long)S.ci[e] * S.ldx;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nvalid) a[u] = fmaf(v, xr[u], a[u]);
                    }
                } else {
                    const float *xr = xb + (long long)r * S.ldx;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < nvalid) a[u] = xr[u];
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

// Workgroup = 8 waves with SPLIT ROLES (CDNA4: MFMA and VALU/VMEM are separate pipes that co-issue
// from different waves of a SIMD): waves 0-3 only read LDS tiles and issue v_mfma_f32_32x32x2_f32,
// waves 4-7 only gather/stage the NEXT [BM x 32] A chunk and [32 x BN] weight chunk into the other
// LDS buffer.  One barrier per chunk hands the buffers over; the MFMA waves never wait on global
// memory, the loader waves are free to sit on L2 latency.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool DUAL, int KC = KC_DEFAULT>
__global__ __launch_bounds__(CAPE_SPEC ? 512 : 256, (DUAL && BN == 128 && BM == 128) ? 1 : (KC == 64 ? 3 : 4)) void gconv_fwd_kernel(GconvParams p) {
    constexpr int LDA = KC + 4;
    constexpr int LDB = BN + 4;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_SZ = BM * LDA, B_SZ = KC * LDB;
    constexpr int BUF_SZ = A_SZ + (DUAL ? 2 : 1) * B_SZ;
    static_assert(WAVES_M * WAVES_N == 4, "4 MFMA waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");

    __shared__ __attribute__((aligned(16))) float smem[((CAPE_SPEC || CAPE_PIPE) ? 2 : 1) * BUF_SZ];

    const int tid = threadIdx.x;
#if CAPE_SPEC
    const bool loader = tid >= 256;
#else
    const bool loader = false;
#endif
    const int ltid = tid & 255;
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
        stage_gather<BM, LDA, KC>(sA, S, n, r0, p.Mo, l_c0, ltid);
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

#if CAPE_SPEC
    if (loader) stage(0);
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        if (loader) {
            if (it + 1 < total) stage((it + 1) & 1);
        } else {
            compute(it & 1);
        }
        __syncthreads();
    }
#elif !CAPE_PIPE
    // All 4 waves stage, then multiply; 3-4 resident workgroups per CU overlap each other's staging
    // and MFMA phases (measured faster on MI355X than both the loader/MFMA wave split above and the
    // register-prefetch pipeline below, which cost occupancy).
    for (int it = 0; it < total; ++it) {
        __syncthreads();
        stage(0);
        __syncthreads();
        compute(0);
    }
#else
    // Software pipeline (all 4 waves stage AND multiply): the global loads of chunk it+1 are issued
    // into registers before the MFMAs of chunk it and written to the other LDS buffer after them --
    // one barrier per chunk, HBM/L2 latency hidden behind 64 MFMAs per wave.  Plain aligned sources
    // with output-contiguous weights take this path; anything else (gathered / unaligned / strided
    // weights) is staged synchronously after the multiply.
    constexpr int P = BM / 32;
    constexpr int NLB = (KC * (BN / 4)) / 256;
    float4 ra[P], rb[NLB], rb2[DUAL ? NLB : 1];
    const int q = ltid & 7, rl0 = ltid >> 3;
    auto is_fast = [&](int si) -> bool {
        const SrcDev &S = p.s[si];
        const bool wfast = (S.wcs == 1) && ((S.wrs & 3) == 0) && ((p.F & 3) == 0) && ((reinterpret_cast<uintptr_t>(S.w) & 15) == 0);
        const bool w2fast = !(DUAL && S.w2) || ((S.w2cs == 1) && ((S.w2rs & 3) == 0) && ((reinterpret_cast<uintptr_t>(S.w2) & 15) == 0));
        return (S.rp == nullptr) && S.vec && ((S.C & 3) == 0) && wfast && w2fast;
    };
    auto load_regs = [&](int si, int c0) {
        const SrcDev &S = p.s[si];
        const int c = c0 + 4 * q;
        const int cc = c < S.C ? c : 0;
        const float *xb = S.x + (long long)n * S.xs + cc;
#pragma unroll
        for (int pass = 0; pass < P; ++pass) {
            const int r = r0 + rl0 + 32 * pass;
            const int rc = r < p.Mo ? r : p.Mo - 1;
            ra[pass] = *reinterpret_cast<const float4 *>(xb + (long long)rc * S.ldx);
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int idx = ltid + i * 256;
            const int j4 = idx % (BN / 4), kk = idx / (BN / 4);
            const int cw = c0 + kk, f = f0 + 4 * j4;
            const int cwc = cw < S.C ? cw : S.C - 1, fc = f < p.F ? f : 0;
            rb[i] = *reinterpret_cast<const float4 *>(S.w + cwc * S.wrs + fc);
            if (DUAL && S.w2) rb2[i] = *reinterpret_cast<const float4 *>(S.w2 + cwc * S.w2rs + fc);
        }
    };
    auto store_regs = [&](int buf, int si, int c0) {
        const SrcDev &S = p.s[si];
        float *sA = smem + buf * BUF_SZ;
        float *sB = sA + A_SZ;
        const bool cok = (c0 + 4 * q) < S.C;
#pragma unroll
        for (int pass = 0; pass < P; ++pass) {
            const bool ok = cok && ((r0 + rl0 + 32 * pass) < p.Mo);
            float4 o = ra[pass];
            o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
            *reinterpret_cast<float4 *>(&sA[(rl0 + 32 * pass) * LDA + 4 * q]) = o;
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int idx = ltid + i * 256;
            const int j4 = idx % (BN / 4), kk = idx / (BN / 4);
            const bool ok = (c0 + kk < S.C) && (f0 + 4 * j4 < p.F);
            float4 o = rb[i];
            o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
            *reinterpret_cast<float4 *>(&sB[kk * LDB + 4 * j4]) = o;
            if (DUAL && S.w2) {
                float4 o2 = rb2[i];
                o2.x = ok ? o2.x : 0.f; o2.y = ok ? o2.y : 0.f; o2.z = ok ? o2.z : 0.f; o2.w = ok ? o2.w : 0.f;
                *reinterpret_cast<float4 *>(&sB[B_SZ + kk * LDB + 4 * j4]) = o2;
            }
        }
    };
    // prologue: chunk 0 -> buffer 0
    if (is_fast(0)) {
        load_regs(0, 0);
        store_regs(0, 0, 0);
        l_c0 += KC;
        if (l_c0 >= p.s[l_si].C) { l_c0 = 0; ++l_si; }
    } else {
        stage(0);
    }
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const bool more = (it + 1 < total);
        const bool fast = more && is_fast(l_si);
        const int n_si = l_si, n_c0 = l_c0;
        if (fast) load_regs(n_si, n_c0);
        compute(it & 1);
        if (more) {
            if (fast) {
                store_regs((it + 1) & 1, n_si, n_c0);
                l_c0 += KC;
                if (l_c0 >= p.s[l_si].C) { l_c0 = 0; ++l_si; }
            } else {
                stage((it + 1) & 1);
            }
        }
        __syncthreads();
    }
#endif
    if (loader) return;

    // ---- epilogue: C layout of 32x32 MFMA: col = lane&31, row = (g&3) + 8*(g>>2) + 4*(lane>>5)
    float *yb = p.y + (long long)n * p.ys;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = f0 + wn * WTN + b * 32 + li;
            float coef[CAPE_MAX_SRC];
#pragma unroll
            for (int j = 0; j < CAPE_MAX_SRC; ++j)
                coef[j] = (j < p.rankR && f < p.F) ? p.coef[((long long)n * p.rankR + j) * p.F + f] : 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int r = r0 + wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                const bool ok = (r < p.Mo) && (f < p.F);
                float v = acc[a][b][g];
                float v2add = 0.f;
                if (p.rankR > 0 && r < p.Mo) {
#pragma unroll
                    for (int j = 0; j < CAPE_MAX_SRC; ++j)
                        if (j < p.rankR) {
                            const float t = p.rowscale[(long long)j * p.Mo + r] * coef[j];
                            if (DUAL && ((p.rank_to2 >> j) & 1u)) v2add += t;
                            else v += t;
                        }
                }
                if (DUAL) {
                    const bool pos = ok && (v > 0.f);
                    if (p.mask) {
                        const unsigned long long bal = __ballot(pos);
                        if (li == 0 && r < p.Mo && (f0 + wn * WTN + b * 32) < p.F) {
                            const unsigned word = lh ? (unsigned)(bal >> 32) : (unsigned)bal;
                            p.mask[((long long)n * p.Mo + r) * p.mask_words + ((f0 + wn * WTN + b * 32) >> 5)] = word;
                        }
                    }
                    v = (v > 0.f ? v : 0.f) + acc2[a][b][g] + v2add;
                } else {
                    if (ok) {
                        if (p.bias_mode == CAPE_BIAS_CHANNEL) v += p.bias[f];
                        else if (p.bias_mode == CAPE_BIAS_VERTEX) v += p.bias[(long long)r * p.F + f];
                    }
                    v = cape_act(v, p.act);
                }
                if (ok) yb[(long long)r * p.ldy + f] = v;
            }
        }
    }
}

// =============================================================================================
// weight gradient:  dW_s[c, f] = sum_{n, r} A_s[n, r, c] * dz[n, r, f]
// Each workgroup owns one [CT x FT] tile of one source's dW and one (sample, row-range)
// slice of the vertex dimension; partials go to the workspace and are summed in a fixed order
// by dw_reduce_kernel (deterministic; no float atomics).
// =============================================================================================
struct DwParams {
    SrcDev s[CAPE_MAX_SRC];
    int nsrc;
    const float *dz;
    const float *dz2;
    unsigned dz2_mask;
    long long dzs;
    int lddz, dzvec;
    int N, Mo, F;
    int ftiles;
    int tile_off[CAPE_MAX_SRC + 1];   // first output tile of each source (c-tiles * ftiles)
    long long part_off[CAPE_MAX_SRC + 1];   // element offset of each source inside one partial slab
    int rsplit, rows_per_split;
    int ngroups, samples_per_group;
    float *ws;
    long long slab;   // elements per split slab
};

template <int CT, int FT>
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
    const int tile = blockIdx.x % ntiles;
    const int split = blockIdx.x / ntiles;   // split = group * rsplit + rs
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
    const float *dzb = (((p.dz2_mask >> si) & 1u) ? p.dz2 : p.dz) + (long long)n * p.dzs;
    const float *xb = S.x + (long long)n * S.xs;

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
                va4[i] = *reinterpret_cast<const float4 *>(xb + (long long)rc * S.ldx + cc);
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
                            const float4 xv = *reinterpret_cast<const float4 *>(xb + (long long)S.ci[e] * S.ldx + c);
                            v4.x = fmaf(v, xv.x, v4.x);
                            v4.y = fmaf(v, xv.y, v4.y);
                            v4.z = fmaf(v, xv.z, v4.z);
                            v4.w = fmaf(v, xv.w, v4.w);
                        }
                    } else {
                        v4 = *reinterpret_cast<const float4 *>(xb + (long long)r * S.ldx + c);
                    }
                } else {
                    float a[4] = {0.f, 0.f, 0.f, 0.f};
                    if (S.rp) {
                        const int e1 = S.rp[r + 1];
                        for (int e = S.rp[r]; e < e1; ++e) {
                            const float v = S.va[e];
                            const float *xr = xb + (long long)S.ci[e] * S.ldx + c;
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (u < nvalid) a[u] = fmaf(v, xr[u], a[u]);
                        }
                    } else {
                        const float *xr = xb + (long long)r * S.ldx + c;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nvalid) a[u] = xr[u];
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
                vb4[i] = *reinterpret_cast<const float4 *>(dzb + (long long)rc * p.lddz + fc);
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
                const float *zr = dzb + (long long)r * p.lddz + f;
                if (p.dzvec && nvalid >= 4) {
                    v4 = *reinterpret_cast<const float4 *>(zr);
                } else {
                    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < nvalid) a[u] = zr[u];
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
__global__ __launch_bounds__(256) void dw_reduce_kernel(DwReduceParams p) {
    __shared__ float red[16][17];
    const long long total = p.part_off[p.nsrc];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long long i = (long long)blockIdx.x * 16 + el;
    float sum = 0.f;
    if (i < total)
        for (int sp = sl; sp < p.nsplit; sp += 16) sum += p.ws[(long long)sp * p.slab + i];
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

inline int fill_src(SrcDev &d, const cape_src_t &s) {
    if (!s.x || s.C <= 0 || s.ldx < s.C) return CAPE_EINVAL;
    if (s.rowptr && (!s.colidx || !s.vals)) return CAPE_EINVAL;
    d.x = s.x; d.xs = s.x_sample_stride; d.ldx = s.ldx; d.C = s.C;
    d.rp = s.rowptr; d.ci = s.colidx; d.va = s.vals;
    d.w = s.w; d.wrs = s.w_rs; d.wcs = s.w_cs;
    d.w2 = s.w2; d.w2rs = s.w2_rs; d.w2cs = s.w2_cs;
    d.vec = ((s.ldx & 3) == 0) && ((s.x_sample_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(s.x) & 15) == 0);
    return CAPE_OK;
}

struct DwPlan {
    int ct, ft, ctiles[CAPE_MAX_SRC], ftiles, ntiles, rsplit, rows_per_split, ngroups, samples_per_group;
    long long slab;
};

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
    // aim for ~512 workgroups: split the vertex dimension down to 128 rows, then the batch into groups
    int S = (512 + pl.ntiles - 1) / pl.ntiles;
    if (S < 1) S = 1;
    int maxr = (Mo + 127) / 128;
    int rsplit = S < maxr ? S : maxr;
    int rows = (Mo + rsplit - 1) / rsplit;
    rows = ((rows + 31) / 32) * 32;
    pl.rows_per_split = rows;
    pl.rsplit = (Mo + rows - 1) / rows;
    int ngroups = (S + pl.rsplit - 1) / pl.rsplit;
    if (ngroups > N) ngroups = N;
    if (ngroups < 1) ngroups = 1;
    pl.samples_per_group = (N + ngroups - 1) / ngroups;
    pl.ngroups = (N + pl.samples_per_group - 1) / pl.samples_per_group;
}

}  // namespace

extern "C" int cape_gconv_fwd(const cape_src_t *srcs, int32_t nsrc, float *y, int64_t y_sample_stride,
                              int32_t ldy, int32_t N, int32_t Mo, int32_t F, const float *bias,
                              int32_t bias_mode, int32_t act, uint32_t *mask_out, const cape_rank_t *rank,
                              void *stream) {
    if (!srcs || nsrc < 1 || nsrc > CAPE_MAX_SRC || !y || N < 1 || Mo < 1 || F < 1 || ldy < F) return CAPE_EINVAL;
    if (bias_mode != CAPE_BIAS_NONE && !bias) return CAPE_EINVAL;
    if (act < CAPE_ACT_NONE || act > CAPE_ACT_TANH) return CAPE_EINVAL;
    GconvParams p;
    bool dual = false;
    for (int i = 0; i < nsrc; ++i) {
        if (!srcs[i].w) return CAPE_EINVAL;
        int rc = fill_src(p.s[i], srcs[i]);
        if (rc) return rc;
        dual = dual || (srcs[i].w2 != nullptr);
    }
    if (mask_out && !dual) return CAPE_EINVAL;
    if (dual && (bias_mode != CAPE_BIAS_NONE || act != CAPE_ACT_NONE)) return CAPE_EINVAL;
    p.nsrc = nsrc; p.y = y; p.ys = y_sample_stride; p.ldy = ldy;
    p.N = N; p.Mo = Mo; p.F = F;
    p.bias = bias; p.bias_mode = bias ? bias_mode : CAPE_BIAS_NONE; p.act = act;
    p.mask = mask_out; p.mask_words = (F + 31) / 32;
    p.rankR = 0; p.rowscale = nullptr; p.coef = nullptr; p.rank_to2 = 0;
    if (rank && rank->R > 0) {
        if (rank->R > CAPE_MAX_SRC || !rank->rowscale || !rank->coef) return CAPE_EINVAL;
        if (rank->to_acc2 && !dual) return CAPE_EINVAL;
        p.rankR = rank->R; p.rowscale = rank->rowscale; p.coef = rank->coef; p.rank_to2 = rank->to_acc2;
    }
    int BM = 128;
    static const int dual_wide = getenv("CAPE_DUAL_WIDE") ? atoi(getenv("CAPE_DUAL_WIDE")) : 0;
    const bool dualw = dual && dual_wide && F > 64;      // DUAL: 64x128 tiles (two accumulator sets = 64 AGPRs)
    const int BN = (F <= 32) ? 32 : (F <= 64 || (dual && !dualw)) ? 64 : 128;
    if (dualw) BM = 64;
    p.col_tiles = (F + BN - 1) / BN;
    // small meshes: 128-row tiles leave <= 2 workgroups per CU (no overlap partner while staging);
    // 64-row tiles double the resident workgroups at the price of re-reading the weight tile
    static const int bm64_below = getenv("CAPE_BM64_BELOW") ? atoi(getenv("CAPE_BM64_BELOW")) : 640;
    if (!dual && BN == 128 && (long long)N * ((Mo + 127) / 128) * p.col_tiles < bm64_below) BM = 64;
    // 64-wide contraction chunks (half the barriers per MFMA) when every source is a multiple of 64 wide
    static const int kc64_on = getenv("CAPE_KC64") ? atoi(getenv("CAPE_KC64")) : 0;
    bool kc64 = kc64_on && !dual && BN == 128;
    for (int i = 0; i < nsrc; ++i) kc64 = kc64 && (srcs[i].C % 64 == 0);
    if (kc64) BM = 64;
    p.row_tiles = (Mo + BM - 1) / BM;
    dim3 grid((unsigned)(N * p.row_tiles * p.col_tiles)), block(CAPE_SPEC ? 512 : 256);
    hipStream_t st = (hipStream_t)stream;
    if (!dual) {
        if (BN == 32) CAPE_LAUNCH((gconv_fwd_kernel<128, 32, 4, 1, false>), grid, block, 0, st, p);
        else if (BN == 64) CAPE_LAUNCH((gconv_fwd_kernel<128, 64, 4, 1, false>), grid, block, 0, st, p);
        else if (BM == 64 && kc64) CAPE_LAUNCH((gconv_fwd_kernel<64, 128, 2, 2, false, 64>), grid, block, 0, st, p);
        else if (BM == 64) CAPE_LAUNCH((gconv_fwd_kernel<64, 128, 2, 2, false>), grid, block, 0, st, p);
        else CAPE_LAUNCH((gconv_fwd_kernel<128, 128, 2, 2, false>), grid, block, 0, st, p);
    } else {
        if (BN == 32) CAPE_LAUNCH((gconv_fwd_kernel<128, 32, 4, 1, true>), grid, block, 0, st, p);
        else if (BN == 64) CAPE_LAUNCH((gconv_fwd_kernel<128, 64, 4, 1, true>), grid, block, 0, st, p);
        else CAPE_LAUNCH((gconv_fwd_kernel<64, 128, 2, 2, true>), grid, block, 0, st, p);
    }
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int64_t cape_gconv_dw_workspace_bytes(const cape_src_t *srcs, int32_t nsrc, int32_t N,
                                                 int32_t Mo, int32_t F) {
    if (!srcs || nsrc < 1 || nsrc > CAPE_MAX_SRC || N < 1 || Mo < 1 || F < 1) return CAPE_EINVAL;
    DwPlan pl;
    plan_dw(srcs, nsrc, N, Mo, F, pl);
    return (int64_t)pl.slab * pl.ngroups * pl.rsplit * (int64_t)sizeof(float);
}

extern "C" int cape_gconv_dw(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                             int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                             int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                             int64_t workspace_bytes, void *stream) {
    if (!srcs || nsrc < 1 || nsrc > CAPE_MAX_SRC || !dz || N < 1 || Mo < 1 || F < 1 || lddz < F || !workspace)
        return CAPE_EINVAL;
    if (dz2_mask && !dz2) return CAPE_EINVAL;
    DwPlan pl;
    plan_dw(srcs, nsrc, N, Mo, F, pl);
    const long long need = pl.slab * pl.ngroups * pl.rsplit * (long long)sizeof(float);
    if (workspace_bytes < need) return CAPE_EWORKSPACE;
    DwParams p;
    DwReduceParams rp;
    p.nsrc = nsrc; rp.nsrc = nsrc;
    int toff = 0;
    long long poff = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (!srcs[i].w) return CAPE_EINVAL;
        int rc = fill_src(p.s[i], srcs[i]);
        if (rc) return rc;
        p.tile_off[i] = toff; p.part_off[i] = poff; rp.part_off[i] = poff;
        toff += pl.ctiles[i] * pl.ftiles;
        poff += (long long)srcs[i].C * F;
        rp.w[i] = const_cast<float *>(srcs[i].w); rp.wrs[i] = srcs[i].w_rs; rp.wcs[i] = srcs[i].w_cs;
    }
    p.tile_off[nsrc] = toff; p.part_off[nsrc] = poff; rp.part_off[nsrc] = poff;
    p.dz = dz; p.dz2 = dz2; p.dz2_mask = dz2 ? dz2_mask : 0u; p.dzs = dz_sample_stride; p.lddz = lddz;
    p.dzvec = ((lddz & 3) == 0) && ((dz_sample_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(dz) & 15) == 0) &&
              (!dz2 || (reinterpret_cast<uintptr_t>(dz2) & 15) == 0);
    p.N = N; p.Mo = Mo; p.F = F; p.ftiles = pl.ftiles;
    p.rsplit = pl.rsplit; p.rows_per_split = pl.rows_per_split;
    p.ngroups = pl.ngroups; p.samples_per_group = pl.samples_per_group;
    p.ws = (float *)workspace; p.slab = pl.slab;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(pl.ntiles * pl.ngroups * pl.rsplit)), block(256);
    if (pl.ct == 64 && pl.ft == 64) CAPE_LAUNCH((gconv_dw_kernel<64, 64>), grid, block, 0, st, p);
    else if (pl.ct == 64) CAPE_LAUNCH((gconv_dw_kernel<64, 128>), grid, block, 0, st, p);
    else if (pl.ft == 64) CAPE_LAUNCH((gconv_dw_kernel<128, 64>), grid, block, 0, st, p);
    else CAPE_LAUNCH((gconv_dw_kernel<128, 128>), grid, block, 0, st, p);
    CAPE_LAUNCH_CHECK();
    rp.F = F; rp.nsplit = pl.ngroups * pl.rsplit; rp.accumulate = accumulate; rp.ws = (const float *)workspace; rp.slab = pl.slab;
    long long total = poff;
    int rblocks = (int)((total + 15) / 16);
    CAPE_LAUNCH(dw_reduce_kernel, dim3(rblocks), dim3(256), 0, st, rp);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
