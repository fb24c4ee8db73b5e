// EXPERIMENT, not built into the library (round 2): measured slower than the tiled kernels on every model shape and equal
// on the HBM-bound K = 6 launches -- profiles/r02_ubench_gemm_rows_experiment.txt.  To try it again: copy it next to
// gemm_split.h, include it from gconv.hip, make plan_fwd() return family 3 when gr_plan() says ok and call gr_launch();
// tools/bench_gemm_rows.py compares the two paths launch by launch.
// Row-streaming form of the plain-source contraction for narrow outputs (F <= 128) with a short contraction
// (sum of C <= a few hundred): the fine mesh levels (3445 / 6890 vertices) and the K = 6 single-layer benchmark.
//
// There the tiled kernel (gemm_split.h) spends its time around the multiply: every 64 x 64 tile stages 4-6 chunks of
// BOTH operands through LDS with two barriers each and ends after a few hundred MFMAs.  But with the whole output
// width in one wave tile no element of the activation operand is shared between waves, so it does not need LDS at all:
//   * the weights (all of them: sum C x F, split once into their three bf16 pieces) stay resident in LDS for the
//     lifetime of the workgroup;
//   * each wave owns 32 * TM rows and reads its A fragments straight from global memory in the MFMA operand layout
//     (lane (li, lh): row li, contraction indices 8 lh .. 8 lh + 7 of a k16 step = two 16-byte loads), splits them in
//     registers and multiplies against B fragments read from LDS;
//   * loads run one macro-step (up to four k16 steps = 64 channels) ahead of the multiplies, no barrier after the
//     weight staging.
// Same arithmetic as gemm_split_kernel: exact three-way bf16 split of both operands, the six largest of the nine
// piece products, fp32 accumulation (v_mfma_f32_32x32x16_bf16); same epilogue (gconv_shared.h).
// Reference operation: the trailing contraction of chebyshev5 (lib/models.py:99-102) and its data gradient.
#pragma once
#include "gemm_split.h"

namespace {

#ifndef GR_EXP
#define GR_EXP 0                               // experiment mask (timing only, wrong results): 1 no weight staging, 2 one product, 4 no A loads
#endif
#ifndef CAPE_GEMM_ROWS_DEFAULT
#define CAPE_GEMM_ROWS_DEFAULT 1               // CAPE_GEMM_ROWS=0: these launches stay on the tiled kernels (A/B switch)
#endif
constexpr int GR_MAX_LDS = 80 * 1024;          // two workgroups per CU
typedef float gr_f32x4 __attribute__((ext_vector_type(4)));      // native vector: register arrays of it stay in registers
constexpr int GR_STEPS = 4;                    // k16 steps per macro-step (64 channels)

struct GrExtra {
    int ktot, k2tot;                           // contraction length of the first / second weight set
};

template <int TM, int TN, bool DUAL>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(GconvParams p, GrExtra ex) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gr_smem[];
    constexpr int BN = 32 * TN;
    constexpr int BM = 128 * TM;
    const int pitch = 2 * ex.ktot + 16;        // bytes per output column of one piece plane (16-byte pad: conflict-free b128 reads)
    const int plane = BN * pitch;
    const int pitch2 = 2 * ex.k2tot + 16;
    const int plane2 = BN * pitch2;
    unsigned char *sB = gr_smem;
    unsigned char *sB2 = gr_smem + 3 * plane;  // DUAL only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // ---- weights -> three bf16 planes in LDS, [column][contraction index]; pairs of consecutive contraction indices,
    // eight pairs per thread in flight (the loads are L2 hits, but one at a time they cost a round trip each)
    auto stage = [&](const float *w, long long wrs, long long wcs, int C, unsigned char *dst, int dpitch, int dplane, int kb) {
        const int pairs = C >> 1;
        const int total = BN * pairs;
        const bool ncontig = wcs == 1;         // lanes along the contiguous axis of the weight block
        for (int e0 = tid; e0 < total; e0 += 256 * 8) {
            float w0[8], w1[8];
            int nn[8], kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u;
                const int ec = e < total ? e : e0;
                int n, kp;
                if (ncontig) { n = ec % BN; kp = ec / BN; } else { kp = ec % pairs; n = ec / pairs; }
                nn[u] = n; kk[u] = kp;
                const float *q = w + (long long)min(n, p.F - 1) * wcs + (long long)(2 * kp) * wrs;
                w0[u] = q[0];
                w1[u] = q[wrs];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (e0 + 256 * u < total) {
                    unsigned hi, mid, lo;
                    const bool live = nn[u] < p.F;           // columns past F hold zeros
                    gs_split2(live ? w0[u] : 0.f, live ? w1[u] : 0.f, hi, mid, lo);
                    unsigned char *d = dst + nn[u] * dpitch + 2 * (kb + 2 * kk[u]);
                    *reinterpret_cast<unsigned *>(d) = hi;
                    *reinterpret_cast<unsigned *>(d + dplane) = mid;
                    *reinterpret_cast<unsigned *>(d + 2 * dplane) = lo;
                }
            }
        }
    };
    {
        int kb = 0, kb2 = 0;
        for (int si = 0; si < p.nsrc; ++si) {
            const SrcDev &S = p.s[si];
#if !(GR_EXP & 1)
            stage(S.w, S.wrs, S.wcs, S.C, sB, pitch, plane, kb);
#endif
            if constexpr (DUAL) {
                if (S.w2) {
                    stage(S.w2, S.w2rs, S.w2cs, S.C, sB2, pitch2, plane2, kb2);
                    kb2 += S.C;
                }
            }
            kb += S.C;
        }
    }
    __syncthreads();

    int n, t;
    cape_map_block(blockIdx.x, p.N, p.row_tiles, n, t);
    const int r0 = t * BM;

    f32x16 acc[TM][TN];
    f32x16 acc2[DUAL ? TM : 1][DUAL ? TN : 1];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                acc[a][b][g] = 0.f;
                if constexpr (DUAL) acc2[a][b][g] = 0.f;
            }

    // rows of this wave (clamped: rows past Mo are computed on valid data and never stored)
    int rc[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) rc[a] = min(r0 + wave * (32 * TM) + a * 32 + li, p.Mo - 1);

    // ---- macro-step cursor: (source, channel offset); a macro-step never spans two sources
    struct Meta { int steps, kb, kb2; bool has2; };
    int c_si = 0, c_c0 = 0, c_kb = 0, c_kb2 = 0;
    int nmacro = 0;
    for (int si = 0; si < p.nsrc; ++si) nmacro += (p.s[si].C + 16 * GR_STEPS - 1) / (16 * GR_STEPS);

    auto load = [&](gr_f32x4 (&buf)[GR_STEPS][TM][2], Meta &m) {
        const SrcDev &S = p.s[c_si];
        const int left = S.C - c_c0;
        m.steps = left >= 16 * GR_STEPS ? GR_STEPS : left >> 4;
        m.kb = c_kb + c_c0;
        m.has2 = DUAL && S.w2 != nullptr;
        m.kb2 = c_kb2 + c_c0;
        const float *xb = S.x + (long long)n * S.xs + c_c0 + 8 * lh;
#pragma unroll
        for (int s = 0; s < GR_STEPS; ++s)
            if (s < m.steps) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const float *q = xb + (long long)rc[a] * S.ldx + 16 * s;
#if GR_EXP & 4
                    buf[s][a][0] = gr_f32x4{1.f, 2.f, 3.f, (float)rc[a]};
                    buf[s][a][1] = buf[s][a][0];
#else
                    buf[s][a][0] = *reinterpret_cast<const gr_f32x4 *>(q);
                    buf[s][a][1] = *reinterpret_cast<const gr_f32x4 *>(q + 4);
#endif
                }
            }
        c_c0 += 16 * m.steps;
        if (c_c0 >= S.C) {
            c_kb += S.C;
            if (m.has2) c_kb2 += S.C;
            c_c0 = 0;
            ++c_si;
        }
    };

    auto compute = [&](const gr_f32x4 (&buf)[GR_STEPS][TM][2], const Meta &m) {
#pragma unroll
        for (int s = 0; s < GR_STEPS; ++s)
            if (s < m.steps) {
                bf16x8 af[TM][3];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    gs_u32x4 hi, mid, lo;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        unsigned ph, pm, pl;
                        gs_split2(buf[s][a][h >> 1][2 * (h & 1)], buf[s][a][h >> 1][2 * (h & 1) + 1], ph, pm, pl);
                        hi[h] = ph; mid[h] = pm; lo[h] = pl;
                    }
                    af[a][0] = __builtin_bit_cast(bf16x8, hi);
                    af[a][1] = __builtin_bit_cast(bf16x8, mid);
                    af[a][2] = __builtin_bit_cast(bf16x8, lo);
                }
                const unsigned char *pb = sB + li * pitch + 2 * (m.kb + 16 * s + 8 * lh);
                bf16x8 bf[TN][3];
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) bf[b][pc] = *reinterpret_cast<const bf16x8 *>(pb + pc * plane + b * 32 * pitch);
                // six products, smallest first; consecutive MFMAs go to different accumulators
#pragma unroll
                for (int term = 0; term < ((GR_EXP & 2) ? 1 : 6); ++term)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][gs_ta(3, term)], bf[b][gs_tb(3, term)], acc[a][b], 0, 0, 0);
                if constexpr (DUAL) {
                    if (m.has2) {
                        const unsigned char *pb2 = sB2 + li * pitch2 + 2 * (m.kb2 + 16 * s + 8 * lh);
#pragma unroll
                        for (int b = 0; b < TN; ++b)
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) bf[b][pc] = *reinterpret_cast<const bf16x8 *>(pb2 + pc * plane2 + b * 32 * pitch2);
#pragma unroll
                        for (int term = 0; term < 6; ++term)
#pragma unroll
                            for (int a = 0; a < TM; ++a)
#pragma unroll
                                for (int b = 0; b < TN; ++b)
                                    acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][gs_ta(3, term)], bf[b][gs_tb(3, term)], acc2[a][b], 0, 0, 0);
                    }
                }
            }
    };

    gr_f32x4 bufA[GR_STEPS][TM][2], bufB[GR_STEPS][TM][2];
    Meta mA, mB;
    load(bufA, mA);
    for (int m = 0; m < nmacro; m += 2) {
        if (m + 1 < nmacro) load(bufB, mB);
        compute(bufA, mA);
        if (m + 2 < nmacro) load(bufA, mA);
        if (m + 1 < nmacro) compute(bufB, mB);
    }

    if (DUAL || p.rankR > 0 || p.bias_mode == CAPE_BIAS_VERTEX || p.act == CAPE_ACT_TANH) {
        gconv_epilogue<BM, BN, 4, 1, DUAL, float>(p, acc, acc2, n, r0, 0, wave, 0, li, lh);
        return;
    }
    // short epilogue of the common launches (see gemm_split_kernel)
    const float slope = p.act == CAPE_ACT_LEAKY ? 0.2f : 1.f;
    const bool relu = p.act == CAPE_ACT_RELU;
    float *yb = p.y + (long long)n * p.ys;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = b * 32 + li;
            const bool fok = f < p.F;
            const int fm = p.deintK > 1 ? (f % p.deintK) * p.deint_stride + f / p.deintK : f;
            const float bch = (p.bias_mode == CAPE_BIAS_CHANNEL && fok) ? p.bias[f] : 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = r0 + wave * (32 * TM) + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                float v = acc[a][b][g] + bch;
                v = v > 0.f ? v : (relu ? 0.f : slope * v);
                if (fok && row < p.Mo) yb[(long long)row * p.ldy + fm] = v;
            }
        }
}

// Eligibility and shape of the row-streaming launch (pure function of the arguments).
struct GrPlan {
    bool ok;
    int TM, TN;
    int ktot, k2tot;
    size_t lds;
};

inline GrPlan gr_plan(const GconvParams &p, bool dual, int layout) {
    GrPlan g{false, 1, 1, 0, 0, 0};
    static const int on = getenv("CAPE_GEMM_ROWS") ? atoi(getenv("CAPE_GEMM_ROWS")) : CAPE_GEMM_ROWS_DEFAULT;
    if (!on || layout < 0 || p.F > 128) return g;
    for (int i = 0; i < p.nsrc; ++i) {
        const SrcDev &S = p.s[i];
        if (S.rp || (S.C & 15) || S.C < 16 || (S.ldx & 3) || (S.xs & 3) || (reinterpret_cast<uintptr_t>(S.x) & 15)) return g;
        if ((long long)p.Mo * S.ldx >= (1LL << 31)) return g;
        g.ktot += S.C;
        if (dual && S.w2) g.k2tot += S.C;
    }
    g.TN = (p.F + 31) / 32;
    const int BN = 32 * g.TN;
    g.lds = (size_t)3 * BN * (2 * g.ktot + 16) + (dual ? (size_t)3 * BN * (2 * g.k2tot + 16) : 0);
    if (g.lds > (size_t)GR_MAX_LDS) return g;
    // worth it only where the tiled kernel runs short K-loops on many rows (fine mesh levels)
    const long long rows = (long long)p.N * p.Mo;
    static const int min_rows = getenv("CAPE_GEMM_ROWS_MIN") ? atoi(getenv("CAPE_GEMM_ROWS_MIN")) : 40000;
    if (rows < min_rows) return g;
    // 256-row workgroups (two 32-row tiles per wave share the weight fragments) while they still fill the chip
    g.TM = (dual || g.TN > 2) ? 1 : ((long long)p.N * ((p.Mo + 255) / 256) >= 448 ? 2 : 1);
    g.ok = true;
    return g;
}

template <int TM, int TN>
inline void gr_launch_t(const GconvParams &p, bool dual, const GrExtra &ex, dim3 grid, size_t lds, hipStream_t st) {
    if (dual) {
        if constexpr (TM == 1) CAPE_LAUNCH((gemm_rows_kernel<1, TN, true>), grid, dim3(256), lds, st, p, ex);
    } else {
        CAPE_LAUNCH((gemm_rows_kernel<TM, TN, false>), grid, dim3(256), lds, st, p, ex);
    }
}

inline void gr_launch(GconvParams &p, bool dual, const GrPlan &g, hipStream_t st) {
    const int BM = 128 * g.TM;
    p.row_tiles = (p.Mo + BM - 1) / BM;
    p.col_tiles = 1;
    const dim3 grid((unsigned)(p.N * p.row_tiles));
    const GrExtra ex{g.ktot, g.k2tot};
    static bool attr_done = false;
    if (!attr_done) {
        // more than 64 KB of dynamic LDS needs the attribute once per kernel
#define GR_ATTR(TM_, TN_, D_) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_rows_kernel<TM_, TN_, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, GR_MAX_LDS)
        GR_ATTR(1, 1, false); GR_ATTR(1, 2, false); GR_ATTR(1, 3, false); GR_ATTR(1, 4, false);
        GR_ATTR(2, 1, false); GR_ATTR(2, 2, false);
        GR_ATTR(1, 1, true); GR_ATTR(1, 2, true); GR_ATTR(1, 3, true); GR_ATTR(1, 4, true);
#undef GR_ATTR
        attr_done = true;
    }
    if (g.TM == 2) {
        if (g.TN == 1) gr_launch_t<2, 1>(p, dual, ex, grid, g.lds, st);
        else gr_launch_t<2, 2>(p, dual, ex, grid, g.lds, st);
    } else {
        if (g.TN == 1) gr_launch_t<1, 1>(p, dual, ex, grid, g.lds, st);
        else if (g.TN == 2) gr_launch_t<1, 2>(p, dual, ex, grid, g.lds, st);
        else if (g.TN == 3) gr_launch_t<1, 3>(p, dual, ex, grid, g.lds, st);
        else gr_launch_t<1, 4>(p, dual, ex, grid, g.lds, st);
    }
}

}  // namespace
