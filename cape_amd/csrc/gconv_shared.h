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
    // fp16 two-piece path (gemm_h2.h): piece planes of the weight blocks [f][c] (pitch in halfs), row-maximum bounds of x
    const unsigned short *wh, *wl;
    long long wp;
    const unsigned short *wh2, *wl2;
    long long wp2;
    const float *rm;   // [N, rows, rmw] or null
    int rmw;
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
    // fp16 two-piece path: reciprocal column scales of the weight planes (second set: DUAL); row-maximum output of ANY
    // kernel's epilogue: rm_out[(n * Mo + r) * rm_out_w + (f >> 5)] >= max |y[n, r, 32-column block of f]| (0 beyond F)
    const float *wsi, *wsi2;
    float *rm_out;
    int rm_out_w;
};

// ---- row-maximum bounds written next to an activation tensor (consumed by the fp16 two-piece contractions) ---------------
// The consumer scales every row by a power of two taken from a BOUND of the row's absolute maximum; any bound within a few
// binades of the true maximum keeps full accuracy (fp16 pieces hold 22 significant bits over 18 binades below the bound).
// The MFMA epilogues therefore reduce over GROUPS OF FOUR consecutive rows (the four accumulator registers g & 3 of a
// lane): a quarter of the cross-lane reductions, every row of the group gets the group's maximum.
// v_max over the 32 lanes (li) of each half-wave: one v_permlane16_swap merges two registers (lanes 0-15 then hold the first,
// 16-31 the second, same in the upper half), four DPP row rotations finish the 16-lane rows.
__device__ __forceinline__ float h2_max_ror(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false)));   // row_ror:8
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false)));   // row_ror:4
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false)));   // row_ror:2
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false)));   // row_ror:1
    return v;
}
// a, b: per-lane maxima of two row groups -> result: lanes 0-15 (and 32-47) = maximum of a over the half-wave's 32 lanes,
// lanes 16-31 (48-63) = that of b
__device__ __forceinline__ float h2_max_pair(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a' = [a0 b0 a2 b2], b' = [a1 b1 a3 b3]
    return h2_max_ror(fmaxf(a, b));
}
// gm[j] = this lane's maximum of |y| over rows 8 j + 4 lh + (0..3) of one 32 x 32 block (column = lane & 31).  Stores the
// block's entry of rm_out for those rows (row0 = first row of the block, cb = 32-column block index).
// The tile that owns column block 0 also zeroes the entries beyond ceil(F / 32) (the row width is padded to a multiple of 4).
__device__ __forceinline__ void h2_store_rowmax(const GconvParams &p, int n, int row0, int cb, const float (&gm)[4], int lane) {
    const int lh = lane >> 5, sel = (lane >> 4) & 1;
    const float m01 = h2_max_pair(gm[0], gm[1]), m23 = h2_max_pair(gm[2], gm[3]);
    if ((lane & 15) == 0) {
        const int nvalid = (p.F + 31) >> 5;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * h + sel;
            const float m = h ? m23 : m01;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = row0 + 8 * j + 4 * lh + i;
                if (r < p.Mo) {
                    float *dst = p.rm_out + ((long long)n * p.Mo + r) * p.rm_out_w;
                    dst[cb] = m;
                    if (cb == 0)
                        for (int e = nvalid; e < p.rm_out_w; ++e) dst[e] = 0.f;
                }
            }
        }
    }
}

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
    // fp16 two-piece form (dw_h2_kernel): row bounds of the gradient operands (the sources' are in SrcDev::rm)
    const float *dzrm, *dz2rm;
    int dzrmw, dz2rmw;
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
            float gm[4] = {0.f, 0.f, 0.f, 0.f};
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
                gm[g >> 2] = fmaxf(gm[g >> 2], ok ? fabsf(v) : 0.f);
            }
            if (p.rm_out && (f0 + wn * WTN + b * 32) < 32 * p.rm_out_w)
                h2_store_rowmax(p, n, r0 + wm * WTM + a * 32, (f0 + wn * WTN + b * 32) >> 5, gm, li + 32 * lh);
        }
    }
}

// Short epilogue for the common launches of the pipelined kernels (no rank-1 terms; channel bias or none; identity / ReLU /
// leaky ReLU as one negative-side slope).  With two workgroups per CU all tiles of a launch finish together, so the epilogue
// is not hidden behind other workgroups' multiplies: the general one (uniform branches per element) cost ~10 us there.
template <int BM, int BN, typename AT = float, int WAVES_M = 2, int WAVES_N = 2>
__device__ __forceinline__ void gconv_epilogue_short(const GconvParams &p, f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32], int n, int r0,
                                                     int f0, int wm, int wn, int li, int lh) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, TM = WTM / 32, TN = WTN / 32;
    const float slope = p.act == CAPE_ACT_LEAKY ? 0.2f : 1.f;
    const bool relu = p.act == CAPE_ACT_RELU;
    AT *yb = reinterpret_cast<AT *>(p.y) + (long long)n * p.ys;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = f0 + wn * WTN + b * 32 + li;
            const bool fok = f < p.F;
            const int fm = p.deintK > 1 ? (f % p.deintK) * p.deint_stride + f / p.deintK : f;
            const float bch = (p.bias_mode == CAPE_BIAS_CHANNEL && fok) ? p.bias[f] : 0.f;
            float gm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = r0 + wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                float v = acc[a][b][g] + bch;
                v = v > 0.f ? v : (relu ? 0.f : slope * v);
                const bool ok = fok && row < p.Mo;
                if (ok) cape_st(&yb[(long long)row * p.ldy + fm], v);
                gm[g >> 2] = fmaxf(gm[g >> 2], ok ? fabsf(v) : 0.f);
            }
            if (p.rm_out && (f0 + wn * WTN + b * 32) < 32 * p.rm_out_w)
                h2_store_rowmax(p, n, r0 + wm * WTM + a * 32, (f0 + wn * WTN + b * 32) >> 5, gm, li + 32 * lh);
        }
}

}  // namespace
