// Types and the output epilogue shared by the gather-GEMM kernels (gconv.hip) and the pipelined
// plain-source GEMM (gemm_plain.h).  gfx950 only.
#pragma once
#include "common.h"

namespace {

struct SrcDev {
    const float *x;   // activations; reinterpreted as cape_bf16 by the bf16-storage kernels (strides in elements)
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
    int deintK, deint_stride;
};

// Weight-gradient launch: each workgroup owns one [CT x FT] tile of one source's dW and one (sample group,
// row range) slice of the contraction; partials go to a workspace slab per split.
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
    // dw_plain_kernel: tiles run over a VIRTUAL channel axis on which source s occupies [vstart[s], vstart[s] + C_s);
    // sources are packed back to back when they share dz (small layers: one tile holds several sources), otherwise
    // each source starts on a tile boundary.  vstart[nsrc] = length of the axis.
    int vstart[CAPE_MAX_SRC + 1];
};

// Epilogue of one workgroup tile: rank-1 condition terms, bias + activation (or, in DUAL mode,
// relu(acc) + acc2 with the ReLU sign bitmask), store.  Accumulator layout of the 32x32 MFMA:
// col = lane & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5).
// AT = storage type of the OUTPUT (p.y is reinterpreted; strides are in elements of AT).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool DUAL, typename AT = float>
__device__ __forceinline__ void gconv_epilogue(const GconvParams &p,
                                               f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32],
                                               f32x16 (&acc2)[DUAL ? BM / WAVES_M / 32 : 1][DUAL ? BN / WAVES_N / 32 : 1],
                                               int n, int r0, int f0, int wm, int wn, int li, int lh) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    AT *yb = reinterpret_cast<AT *>(p.y) + (long long)n * p.ys;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = f0 + wn * WTN + b * 32 + li;
            // optional de-interleave of the output columns: column j = c*K + k is stored at channel k*stride + c
            const int fm = p.deintK > 1 ? (f % p.deintK) * p.deint_stride + f / p.deintK : f;
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
                if (ok) cape_st(&yb[(long long)r * p.ldy + fm], v);
            }
        }
    }
}

}  // namespace
