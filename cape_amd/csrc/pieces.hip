// Operand preparation of the fp16 two-piece contractions (gemm_h2.h): the weight piece planes, written once per step, and
// a standalone row-maximum pass for activation tensors whose producer did not write bounds (tensors handed in from outside
// the layer stack; the kernels inside the stack write them from their epilogues).  gfx950 only.
//
// Weight tensor of a Chebyshev layer, reference layout W[Ch*K (+ condition rows), F], row c*K + k (lib/models.py:97-101):
//   forward planes   Pf[k][f][c] = piece(W[c*K + k, f] * sf[f])      sf[f]  from max over ALL feature rows of column f
//   backward planes  Pb[c*K + k][f] = piece(W[c*K + k, f] * sb[c])   sb[c]  from max over the K rows of channel c
// Forward launches contract over c (sources k = 0..K-1 add into one accumulator column f, hence one scale per f); the data
// gradient contracts over f into output column c (de-interleaved form: column c*K + k), hence one scale per c.  Both plane
// sets are contraction-contiguous, so gemm_h2_kernel has one staging path.  The reciprocal scales are stored replicated
// (fsi[k*F + f], bsi[c*K + k]) so that every launch form indexes them by its own output column.
#include <stdlib.h>

#include "common.h"
#include "gemm_h2.h"

namespace {

// ---- standalone row maxima: out[(n*M + r)*W + 0] = max_c |x[n, r, c]|, entries 1..W-1 = 0 --------------------------------
// 16 lanes per row (one DPP row), 16 rows per 256-thread block
__global__ __launch_bounds__(256) void rowmax_kernel(const float *x, long long xs, int ldx, int N, int M, int C, float *out, int W) {
    const int l = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = row < (long long)N * M;
    const long long rr = live ? row : 0;
    const int n = (int)(rr / M), r = (int)(rr % M);
    const float *p = x + (long long)n * xs + (long long)r * ldx;
    float m = 0.f;
    if (((ldx | C) & 3) == 0 && (xs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        for (int j = l; j < (C >> 2); j += 16) {
            const float4 v = *reinterpret_cast<const float4 *>(p + 4 * j);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    } else {
        for (int j = l; j < C; j += 16) m = fmaxf(m, fabsf(p[j]));
    }
    m = h2_max_ror(m);
    if (live && l < W) out[rr * W + l] = l == 0 ? m : 0.f;
}

// ---- weight pieces ---------------------------------------------------------------------------------------------------
struct WItem {                      // device-resident copy of cape_wpiece_item_t (identical layout)
    const float *w;
    int Ch, K, F, pairK;
    const float *pw;
    unsigned short *f_hi, *f_lo, *b_hi, *b_lo;
    float *fsi, *bsi, *bsc;
    const float *fpw;
    int fprows, reserved;
    float *pc;                      // partial column maxima [row patches of 64][F]
};

// block -> (item, local block) through the prefix table off[nitems + 1].  The table is first copied to LDS by the whole block: a
// scan of it in global memory is a chain of up to nitems dependent L2 reads (13 of the first version's 19 us).
constexpr int W_MAX_ITEMS = 255;
__device__ __forceinline__ int w_find(const int *off, int nitems, int b, int &first) {
    __shared__ int soff[W_MAX_ITEMS + 1];
    for (int i = threadIdx.x; i <= nitems; i += 256) soff[i] = off[i];
    __syncthreads();
    int lo = 0, hi = nitems - 1;                       // largest i with soff[i] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (soff[mid] <= b) lo = mid; else hi = mid - 1;
    }
    first = soff[lo];
    return lo;
}

// pass 1: the maxima.  Local blocks [0, ceil(F/64) * ceil(R/64)), R = Ch*K feature rows (+ the forward partner's rows): one
// 64-row x 64-column patch each, thread = (16 row lanes) x (float4 of columns): four independent 16-byte loads per thread and a
// 16-lane LDS reduction -> partial column maxima pc[row patch][F] (no atomics; the planes pass and the last phase below take
// their maximum).  Then ceil(Ch/4) blocks of four channels (one wave per channel: its K rows, lanes over the columns).
// (The first version ran one block per 16-column strip over ALL rows: 64 dependent iterations, 18 us for the nz64 model.)
__device__ __forceinline__ int w_rpatches(const WItem &I) { return (I.Ch * I.K + (I.fpw ? I.fprows : 0) + 63) >> 6; }

__global__ __launch_bounds__(256) void wmax_kernel(const WItem *items, int nitems, const int *off) {
    int first;
    const int it = w_find(off, nitems, blockIdx.x, first);
    const WItem I = items[it];
    const int b = blockIdx.x - first;
    const int cstrips = (I.F + 63) >> 6, rp = w_rpatches(I);
    const int rows = I.Ch * I.K;
    __shared__ float4 part[16][17];
    if (b < cstrips * rp) {
        const int cs = b % cstrips, rpi = b / cstrips;
        const int fq = threadIdx.x & 15, rl = threadIdx.x >> 4, f = cs * 64 + 4 * fq;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < I.F) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = rpi * 64 + rl + 16 * u;                        // row of the concatenation [W's feature rows | partner's rows]
                const float *src = j < rows ? I.w + (long long)j * I.F : (j - rows < (I.fpw ? I.fprows : 0) ? I.fpw + (long long)(j - rows) * I.F : nullptr);
                if (src) {
                    const float4 v = *reinterpret_cast<const float4 *>(src + f);
                    m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
                }
            }
        }
        part[rl][fq] = m;
        __syncthreads();
        if (rl == 0 && f < I.F) {
#pragma unroll
            for (int l = 1; l < 16; ++l) {
                const float4 v = part[l][fq];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
            *reinterpret_cast<float4 *>(I.pc + (long long)rpi * I.F + f) = m;
        }
    } else {
        const int c = (b - cstrips * rp) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (c >= I.Ch) return;
        float m = 0.f;
        const float *p = I.w + (long long)c * I.K * I.F;                 // the K rows of channel c are contiguous: K * F floats
#pragma unroll 4
        for (int j = lane; j < I.K * I.F; j += 64) m = fmaxf(m, fabsf(p[j]));
        if (I.pw) {                                                       // the partner's rows of the same channel
            const float *pp = I.pw + (long long)c * I.pairK * I.F;
#pragma unroll 4
            for (int j = lane; j < I.pairK * I.F; j += 64) m = fmaxf(m, fabsf(pp[j]));
        }
        m = h2_max_ror(m);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < I.K) {
            float s, inv;
            h2_scale_of(m, s, inv);
            I.bsi[c * I.K + lane] = inv;
            if (lane == 0) I.bsc[c] = inv;
        }
    }
}

// reciprocal forward scale of column f from the partial maxima (a handful of L2-resident loads)
__device__ __forceinline__ float w_fscale_inv(const WItem &I, int f) {
    float m = 0.f;
    const int rp = w_rpatches(I);
#pragma unroll 4
    for (int r = 0; r < rp; ++r) m = fmaxf(m, I.pc[(long long)r * I.F + f]);
    float s, inv;
    h2_scale_of(m, s, inv);
    return inv;
}

// pass 2: the planes.  Local blocks [0, K * (Ch/32) * ceil(F/64)): forward planes, one 32-channel x 64-column tile of order k
// per block, transposed through LDS (rows of W are read along f, the planes are written along c: both sides coalesced;
// only when Ch % 32 == 0 -- other layers never take the forward kernel).  Then ceil(Ch*K*F / 2048) blocks of the
// backward planes (same order as W: thread = eight consecutive columns).
__device__ __forceinline__ int w_fwd_blocks(const WItem &I) { return (I.Ch & 31) ? 0 : I.K * (I.Ch >> 5) * ((I.F + 63) >> 6); }

__global__ __launch_bounds__(256) void wplanes_kernel(const WItem *items, int nitems, const int *off) {
    int first;
    const int it = w_find(off, nitems, blockIdx.x, first);
    const WItem I = items[it];
    const int b = blockIdx.x - first;
    const int nfb = w_fwd_blocks(I);
    if (b < nfb) {
        __shared__ unsigned short th[64][40], tl[64][40];                  // [column][channel], 80-byte rows: 16-byte aligned segments
        const int ftiles = (I.F + 63) >> 6, ctiles = I.Ch >> 5;
        const int ft = b % ftiles, ct = (b / ftiles) % ctiles, k = b / (ftiles * ctiles);
        const int c0 = ct * 32, f0 = ft * 64;
        // the tile's 64 column scales from the partial maxima, once per block (threads 0..63); the ct == 0 tiles publish them
        __shared__ float sinv[64];
        if (threadIdx.x < 64 && f0 + (int)threadIdx.x < I.F) {
            const float inv = w_fscale_inv(I, f0 + threadIdx.x);
            sinv[threadIdx.x] = inv;
            if (ct == 0) I.fsi[(long long)k * I.F + f0 + threadIdx.x] = inv;
        }
        __syncthreads();
        {
            const int cl = threadIdx.x >> 3, fq = threadIdx.x & 7, f = f0 + 8 * fq;
            if (f < I.F) {
                const float *src = I.w + ((long long)(c0 + cl) * I.K + k) * I.F + f;
                const float4 a = *reinterpret_cast<const float4 *>(src), bb = *reinterpret_cast<const float4 *>(src + 4);
                float inv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) inv[j] = sinv[8 * fq + j];
                const float v[8] = {a.x / inv[0], a.y / inv[1], a.z / inv[2], a.w / inv[3], bb.x / inv[4], bb.y / inv[5], bb.z / inv[6], bb.w / inv[7]};   // powers of two: exact
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const _Float16 h = (_Float16)v[j], l = (_Float16)(v[j] - (float)h);
                    th[8 * fq + j][cl] = __builtin_bit_cast(unsigned short, h);
                    tl[8 * fq + j][cl] = __builtin_bit_cast(unsigned short, l);
                }
            }
        }
        __syncthreads();
        {
            const int fl = threadIdx.x >> 2, cq = threadIdx.x & 3, f = f0 + fl;
            if (f < I.F) {
                const long long d = ((long long)k * I.F + f) * I.Ch + c0 + 8 * cq;
                *reinterpret_cast<uint4 *>(I.f_hi + d) = *reinterpret_cast<const uint4 *>(&th[fl][8 * cq]);
                *reinterpret_cast<uint4 *>(I.f_lo + d) = *reinterpret_cast<const uint4 *>(&tl[fl][8 * cq]);
            }
        }
        return;
    }
    const long long e = (long long)(b - nfb) * 256 + threadIdx.x;
    if (e >= (long long)I.Ch * I.K * (I.F >> 3)) return;
    const int f8 = (int)(e % (I.F >> 3));
    const long long j = e / (I.F >> 3);                                   // row c*K + k
    const float s = 1.f / I.bsi[j];
    const float4 a = *reinterpret_cast<const float4 *>(I.w + j * I.F + 8 * f8), bb = *reinterpret_cast<const float4 *>(I.w + j * I.F + 8 * f8 + 4);
    uint4 hi, lo;
    h2_split2(a.x * s, a.y * s, hi.x, lo.x);
    h2_split2(a.z * s, a.w * s, hi.y, lo.y);
    h2_split2(bb.x * s, bb.y * s, hi.z, lo.z);
    h2_split2(bb.z * s, bb.w * s, hi.w, lo.w);
    *reinterpret_cast<uint4 *>(I.b_hi + j * I.F + 8 * f8) = hi;
    *reinterpret_cast<uint4 *>(I.b_lo + j * I.F + 8 * f8) = lo;
}

}  // namespace

extern "C" int cape_rowmax(const float *x, int64_t x_sample_stride, int32_t ldx, int32_t N, int32_t M, int32_t C, float *out,
                           int32_t out_w, void *stream) {
    if (!x || !out || N < 1 || M < 1 || C < 1 || ldx < C || out_w < 4 || out_w > 16 || (out_w & 3)) return CAPE_EINVAL;
    const long long rows = (long long)N * M;
    CAPE_LAUNCH(rowmax_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x, (long long)x_sample_stride, ldx,
                N, M, C, out, out_w);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}

extern "C" int cape_weight_pieces_blocks(const cape_wpiece_item_t *host_items, int32_t nitems, int32_t *max_off, int32_t *planes_off) {
    if (!host_items || nitems < 1 || nitems > W_MAX_ITEMS || !max_off || !planes_off) return CAPE_EINVAL;
    static_assert(sizeof(WItem) == sizeof(cape_wpiece_item_t), "layout");
    max_off[0] = planes_off[0] = 0;
    for (int i = 0; i < nitems; ++i) {
        const cape_wpiece_item_t &I = host_items[i];
        if (!I.w || I.Ch < 8 || (I.Ch & 7) || I.K < 1 || I.K > 16 || I.F < 8 || (I.F & 7)) return CAPE_EINVAL;
        if (!I.f_hi || !I.f_lo || !I.b_hi || !I.b_lo || !I.fscale_inv || !I.bscale_inv || !I.bscale_c_inv) return CAPE_EINVAL;
        if (I.pair_w && (I.pair_K < 1 || I.pair_K > 16)) return CAPE_EINVAL;
        if (I.fpair_w && I.fpair_rows < 1) return CAPE_EINVAL;
        const long long bwd = (long long)I.Ch * I.K * (I.F / 8);
        const int fwd = (I.Ch & 31) ? 0 : I.K * (I.Ch / 32) * ((I.F + 63) / 64);
        const int rpatch = (I.Ch * I.K + (I.fpair_w ? I.fpair_rows : 0) + 63) / 64;
        if (!I.colmax_partial) return CAPE_EINVAL;
        max_off[i + 1] = max_off[i] + ((I.F + 63) / 64) * rpatch + (I.Ch + 3) / 4;
        planes_off[i + 1] = planes_off[i] + fwd + (int)((bwd + 255) / 256);
    }
    return CAPE_OK;
}

extern "C" int cape_weight_pieces(const cape_wpiece_item_t *dev_items, int32_t nitems, const int32_t *dev_max_off, int32_t max_blocks,
                                  const int32_t *dev_planes_off, int32_t planes_blocks, void *stream) {
    if (!dev_items || nitems < 1 || nitems > W_MAX_ITEMS || !dev_max_off || !dev_planes_off || max_blocks < 1 || planes_blocks < 1) return CAPE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    CAPE_LAUNCH(wmax_kernel, dim3((unsigned)max_blocks), dim3(256), 0, st, reinterpret_cast<const WItem *>(dev_items), nitems, dev_max_off);
    CAPE_LAUNCH_CHECK();
    CAPE_LAUNCH(wplanes_kernel, dim3((unsigned)planes_blocks), dim3(256), 0, st, reinterpret_cast<const WItem *>(dev_items), nitems,
                dev_planes_off);
    CAPE_LAUNCH_CHECK();
    return CAPE_OK;
}
