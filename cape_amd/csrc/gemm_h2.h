// fp32 contraction of PLAIN sources as THREE fp16 MFMA products per multiply-add ("h2": two half-precision pieces).
// Same contraction and epilogues as gemm_plain.h / gemm_split.h (y[n] = epilogue(sum_s X_s[n] @ B_s), reference
// lib/models.py:99-102; DUAL: the affine block of :776-793), at half the matrix-pipe work of the six-product bf16 form and
// a third of its staging arithmetic:
//     x * s = hi + lo,   hi = fp16(x * s),  lo = fp16(x * s - hi)      (both round to nearest: 22 significant bits)
//     a * b ~ (lo_a hi_b + hi_a lo_b + hi_a hi_b) / (s_a s_b)          (dropped lo_a lo_b: 2^-22 relative)
// with s a power of two per activation ROW (from a bound of the row's absolute maximum that the kernel PRODUCING the tensor
// wrote next to it: SrcDev::rm, see gconv_shared.h) and per weight COLUMN (planes prepared once per step by
// cape_weight_pieces: SrcDev::wh / wl, contraction index contiguous for every launch form, so there is one kernel for the
// forward and the data-gradient layout).  Measured against float64 the result is as accurate as the bf16 six-product form and
// an fp32 FMA chain (tools/ubench/gemm_h2.hip, profiles/r04_ubench_h2*.txt: rms 3.6e-7 of the row rms at K = 1024 on rows
// spanning 23 binades; fp32 chain 5.5e-7).  Range: fp16 keeps 22 bits for elements down to 2^-16 of the row bound and
// degrades gracefully below (absolute error <= 2^-38 of the bound; tests/test_h2_numerics.py); a zero / denormal row
// bound is clamped.
//
// Structure (tools/ubench/gemm_h2.hip, AMODE 2): workgroup tile BM x BN, 4 waves as 2 x 2, k32 chunks, two LDS stages, ONE
// barrier per chunk.  Weight pieces come in by LDS-DMA one chunk ahead (no registers, no VALU, no ds_write; L2-resident);
// the activations through a buffer resource TWO chunks ahead in registers (the long-latency stream), split after the
// multiply phase of the chunk in between.  vmcnt retires in order: the DMA of chunk i+1 is issued before the register loads
// of chunk i+2, so "all but the newest PA*2" covers it.  LDS rows are 64 bytes with the 16-byte segments XOR-swizzled by
// (row >> 2) & 3 (conflict-free ds_read_b128 for the 32x32x16 fragment pattern; the DMA applies it on the SOURCE address).
// Against gemm_split_kernel on the model's shapes (same box): 1.25-1.55x on the 862 / 1723-vertex levels, 1.1-1.25x on the
// fine ones (profiles/r04_ubench_h2_pf2.txt); the loads alone (448 MB through L2 for the widest layer) take 39 us of its
// 57 -- the 128 x 128 tile's L2 traffic, not the matrix pipe, is the next bound.
#pragma once
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "gconv_shared.h"

namespace {

typedef _Float16 h2_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2_half2 __attribute__((ext_vector_type(2)));
typedef unsigned h2_u32x4 __attribute__((ext_vector_type(4)));

constexpr int H2_KC = 32;          // contraction indices per staged chunk = two k16 MFMA steps
constexpr int H2_ROW = 64;         // bytes per LDS row of one piece plane

__device__ __forceinline__ unsigned h2_bits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float h2_float(unsigned v) { return __builtin_bit_cast(float, v); }

// two scaled fp32 values -> their fp16 pieces, packed pairwise (first element in the low half)
__device__ __forceinline__ void h2_split2(float x0, float x1, unsigned &hi, unsigned &lo) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    const h2_half2 H = {h0, h1}, L = {l0, l1};
    hi = __builtin_bit_cast(unsigned, H);
    lo = __builtin_bit_cast(unsigned, L);
}

// The same split of two values TIMES a power of two in four instructions: both conversions fold the multiplication (the product
// is exact in fp32, so rounding x * s to fp16 once equals rounding the fp32 product) and the residual x * s - hi is formed in
// fp32 inside the fused op before it is rounded -- bit-identical to h2_split2(x0 * s, x1 * s, ...).  hipcc's own selection for
// that expression is seven instructions per pair (v_mul x2, v_fma_mixlo x3, v_fma_mixhi, v_cvt_pk): it packs hi from the
// products instead of writing both halves of one register.  (Not volatile: the scheduler may place it.)
__device__ __forceinline__ void h2_split2s(float x0, float x1, float s, unsigned &hi, unsigned &lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l) : "v"(x0), "v"(x1), "v"(s));
    hi = h;
    lo = l;
}

// power of two that puts a bound m of the absolute maximum into [2^13, 2^14), and its reciprocal.  Zero / denormal bounds
// are clamped (their rows hold nothing a half can represent anyway); inf / NaN bounds give inf / NaN results like the split
// kernels do for such operands.
__device__ __host__ __forceinline__ void h2_scale_of(float m, float &s, float &inv) {
    unsigned u;
    __builtin_memcpy(&u, &m, 4);
    int e = (int)((u >> 23) & 255u);
    e = e < 14 ? 14 : (e > 253 ? 253 : e);
    const unsigned us = (unsigned)(267 - e) << 23, ui = (unsigned)(e - 13) << 23;
    __builtin_memcpy(&s, &us, 4);
    __builtin_memcpy(&inv, &ui, 4);
}

// one 1 KB LDS-DMA piece: lane L's 16 bytes land at lds_dst + 16 L (M0 = wave-uniform LDS byte address); source address =
// 64-bit scalar base + 32-bit per-lane byte offset.  Untracked by hipcc: completion is counted by hand (vmcnt).
__device__ __forceinline__ void h2_glds16(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// BM x BN in {128 x 128, 64 x 64} (single), 128 x 64 (DUAL: second weight set / second accumulator tile)
template <int BM, int BN, bool DUAL>
__global__ __launch_bounds__(256, 2) void gemm_h2_kernel(GconvParams p) {
    constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
    constexpr int PA = BM / 64;                          // A staging passes: 64 rows x 4 eight-float segments per pass
    constexpr int APL = BM * H2_ROW, BPL = BN * H2_ROW;  // bytes of one piece plane
    constexpr int NB = DUAL ? 2 : 1;
    constexpr int STAGE = 2 * APL + 2 * NB * BPL;
    constexpr int BPW = (2 * BPL / 1024) / 4;            // DMA pieces (16 rows x 64 B) per wave and weight set
    static_assert(TM >= 1 && TN >= 1 && BPW >= 1, "tile");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    __shared__ float inv_row[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int q = tid & 3, r = tid >> 2;

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
                if constexpr (DUAL) acc2[a][b][g] = 0.f;
            }

    int total = 0;
    for (int si = 0; si < p.nsrc; ++si) total += p.s[si].C / H2_KC;

    int rc[PA];
    float sa[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) rc[i] = min(r0 + r + 64 * i, p.Mo - 1);
    // ---- DMA lanes: piece j of this wave covers 16 rows x 64 B of one plane; lane -> row drow, LDS slot lane & 3,
    //      source segment (lane & 3) ^ ((row >> 2) & 3)
    const int drow = lane >> 2, dseg = (lane & 3) ^ ((drow >> 2) & 3);
    const unsigned lds0 = (unsigned)(size_t)smem;
    int bcol[BPW];                                       // output column (clamped) of this lane's row in piece j
    unsigned bdst[BPW];                                  // LDS byte offset of piece j inside a stage (first weight set)
    int bplane[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int i = wave * BPW + j, plane = i / (BN / 16), rb = i % (BN / 16);
        bcol[j] = min(f0 + rb * 16 + drow, p.F - 1);
        bdst[j] = 2 * APL + plane * BPL + rb * 1024;
        bplane[j] = plane;
    }

    // ---- cursors over the chunk sequence (source, channel offset): B = DMA of the weight pieces (one chunk ahead of the
    //      multiply), A = register loads of the activations (two ahead)
    struct Cur { int si, c0; };
    Cur cb = {0, 0}, ca = {0, 0};
    unsigned bvoff[BPW], bvoff2[DUAL ? BPW : 1];
    // the fields of the current source live in registers between source changes: a scalar load of p.s[si] inside the chunk
    // loop shares its wait counter (lgkmcnt) with the LDS fragment reads and would drain them once per chunk
    const unsigned short *b_wh = nullptr, *b_wl = nullptr, *b_wh2 = nullptr, *b_wl2 = nullptr;
    int b_C = 0, a_C = 0;
    auto open_b = [&]() {
        const SrcDev &S = p.s[cb.si];
        b_wh = S.wh; b_wl = S.wl; b_C = S.C;
        if constexpr (DUAL) { b_wh2 = S.wh2; b_wl2 = S.wl2; }
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            bvoff[j] = (unsigned)(((long long)bcol[j] * S.wp + 8 * dseg) * 2);
            if constexpr (DUAL) bvoff2[j] = (unsigned)(((long long)bcol[j] * S.wp2 + 8 * dseg) * 2);
        }
    };
    int avoff[PA];
    __amdgpu_buffer_rsrc_t arsrc;
    auto open_a = [&]() {
        const SrcDev &S = p.s[ca.si];
        a_C = S.C;
        arsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(S.x + (long long)n * S.xs), 0, 0x7FFFFFFC, 0x00020000);
#pragma unroll
        for (int i = 0; i < PA; ++i) avoff[i] = (rc[i] * S.ldx + 8 * q) * 4;
    };
    unsigned has2 = 0;                                   // (DUAL) bit b: the chunk staged in LDS buffer b carries a second weight set
    auto dma = [&](int buf) {
        const unsigned dst = lds0 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            h2_glds16((bplane[j] ? b_wl : b_wh) + cb.c0, bvoff[j], dst + bdst[j]);
        if constexpr (DUAL) {
            const bool two = b_wh2 != nullptr;
            has2 = (has2 & ~(1u << buf)) | ((two ? 1u : 0u) << buf);
            if (two) {
#pragma unroll
                for (int j = 0; j < BPW; ++j)
                    h2_glds16((bplane[j] ? b_wl2 : b_wh2) + cb.c0, bvoff2[j], dst + bdst[j] + 2 * BPL);
            }
        }
        cb.c0 += H2_KC;
        if (cb.c0 >= b_C) {
            cb.c0 = 0;
            ++cb.si;
            if (cb.si < p.nsrc) open_b();
        }
    };
    auto load_a = [&](float4 (&ra)[PA][2]) {
        const int so = ca.c0 * 4;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            ra[i][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, avoff[i], so, 0));
            ra[i][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, avoff[i] + 16, so, 0));
        }
        ca.c0 += H2_KC;
        if (ca.c0 >= a_C) {
            ca.c0 = 0;
            ++ca.si;
            if (ca.si < p.nsrc) open_a();
        }
    };
    auto store_a = [&](int buf, const float4 (&ra)[PA][2]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = r + 64 * i;
            const float s = sa[i];
            uint4 hi, lo;
            h2_split2s(ra[i][0].x, ra[i][0].y, s, hi.x, lo.x);
            h2_split2s(ra[i][0].z, ra[i][0].w, s, hi.y, lo.y);
            h2_split2s(ra[i][1].x, ra[i][1].y, s, hi.z, lo.z);
            h2_split2s(ra[i][1].z, ra[i][1].w, s, hi.w, lo.w);
            unsigned char *d = smem + buf * STAGE + row * H2_ROW + 16 * (q ^ ((row >> 2) & 3));
            *reinterpret_cast<uint4 *>(d) = hi;
            *reinterpret_cast<uint4 *>(d + APL) = lo;
        }
    };

    // ---- multiply one staged chunk.  Lane (li, lh) of v_mfma_f32_32x32x16_f16 supplies row / column li and the contraction
    //      indices 8 lh .. 8 lh + 7 of the k16 step: one 16-byte LDS read per operand piece; the reads of step 1 ride between
    //      the MFMAs of step 0 (a ds_read_b128 holds its wave's issue port ~29 cycles: DESIGN.md section 4, round 3)
    const int fsw = (li >> 2) & 3;                       // swizzle term of this lane's fragment rows (tile offsets are multiples of 32)
    auto compute = [&](int buf, auto W2) {
        constexpr bool w2 = decltype(W2)::value;
        const unsigned char *pa = smem + buf * STAGE + (wm * WTM + li) * H2_ROW;
        const unsigned char *pb = smem + buf * STAGE + 2 * APL + (wn * WTN + li) * H2_ROW;
        h2_half8 af[2][TM][2], bf[2][TN][2], bf2[2][w2 ? TN : 1][w2 ? 2 : 1];
        auto rd = [&](int ks) {
            const int so = 16 * ((2 * ks + lh) ^ fsw);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) af[ks][a][pc] = *reinterpret_cast<const h2_half8 *>(pa + pc * APL + a * 32 * H2_ROW + so);
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    bf[ks][b][pc] = *reinterpret_cast<const h2_half8 *>(pb + pc * BPL + b * 32 * H2_ROW + so);
                    if constexpr (w2) bf2[ks][b][pc] = *reinterpret_cast<const h2_half8 *>(pb + (2 + pc) * BPL + b * 32 * H2_ROW + so);
                }
        };
        auto mm = [&](int ks) {
#pragma unroll
            for (int term = 0; term < 3; ++term)         // lo*hi, hi*lo, hi*hi: small products first
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][a][term == 0 ? 1 : 0], bf[ks][b][term == 1 ? 1 : 0],
                                                                           acc[a][b], 0, 0, 0);
                        if constexpr (w2)
                            acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][a][term == 0 ? 1 : 0], bf2[ks][b][term == 1 ? 1 : 0],
                                                                                acc2[a][b], 0, 0, 0);
                    }
        };
        constexpr int NM = 3 * TM * TN * (w2 ? 2 : 1), NR = 2 * TM + 2 * TN * (w2 ? 2 : 1);
        rd(0);
        __builtin_amdgcn_sched_barrier(0);
        rd(1);
        mm(0);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, NM / NR > 0 ? NM / NR : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
    };
    auto compute_buf = [&](int buf) {
        if constexpr (DUAL) {
            if ((has2 >> buf) & 1u) compute(buf, std::true_type{});
            else compute(buf, std::false_type{});
        } else {
            compute(buf, std::false_type{});
        }
    };

    float4 ra[PA][2], rb[PA][2];
    open_b();
    open_a();
    dma(0);
    load_a(ra);
    if (total > 1) load_a(rb);
    // ---- row scales: common to all sources (they add into one accumulator): bound = max over sources and column blocks.
    //      Read AFTER the first chunks' loads are in flight: one memory round trip less in front of the first split.
    {
        // (all loads first -- predicated over the CAPE_MAX_SRC slots -- then the maxima: inside a source loop hipcc waits
        // for every load before issuing the next, one L2 round trip per source and row)
        float4 bv[PA][CAPE_MAX_SRC];
#pragma unroll
        for (int i = 0; i < PA; ++i)
#pragma unroll
            for (int si = 0; si < CAPE_MAX_SRC; ++si) {
                bv[i][si] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (si < p.nsrc && 4 * q < p.s[si].rmw)
                    bv[i][si] = *reinterpret_cast<const float4 *>(p.s[si].rm + ((long long)n * p.Mo + rc[i]) * p.s[si].rmw + 4 * q);
            }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            float m = 0.f;
#pragma unroll
            for (int si = 0; si < CAPE_MAX_SRC; ++si)
                m = fmaxf(m, fmaxf(fmaxf(bv[i][si].x, bv[i][si].y), fmaxf(bv[i][si].z, bv[i][si].w)));
            // the four lanes of a row (q = 0..3) hold different column blocks: quad all-reduce
            m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, false)));   // quad_perm [1,0,3,2]
            m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, false)));   // quad_perm [2,3,0,1]
            float inv;
            h2_scale_of(m, sa[i], inv);
            if (q == 0) inv_row[r + 64 * i] = inv;
        }
    }

    store_a(0, ra);
    if (total > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // steady = true: chunks it+1 and it+2 exist -- no branches in the body, so hipcc's own vmcnt bookkeeping for the register
    // loads stays exact (with the conditions inside, the merge of the paths made it wait for the newest loads as well)
    auto body = [&](int it, float4 (&rfree)[PA][2], const float4 (&rnext)[PA][2], auto steady) {
        constexpr bool ST = decltype(steady)::value;
        const int buf = it & 1;
        const bool m1 = ST || it + 1 < total, m2 = ST || it + 2 < total;
        if (m1) dma(buf ^ 1);
        if (m2) load_a(rfree);
        compute_buf(buf);
        if (m1) store_a(buf ^ 1, rnext);
        if (m2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    int it = 0;
    for (; it + 3 < total; it += 2) {
        body(it, ra, rb, std::true_type{});
        body(it + 1, rb, ra, std::true_type{});
    }
    for (; it < total; it += 2) {
        body(it, ra, rb, std::false_type{});
        if (it + 1 < total) body(it + 1, rb, ra, std::false_type{});
    }

    // ---- undo the scales (all powers of two: exact), then the shared epilogues
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = min(f0 + wn * WTN + b * 32 + li, p.F - 1);
            const float wi = p.wsi[f];
            float wi2 = 0.f;
            if constexpr (DUAL) wi2 = p.wsi2[f];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float ir = inv_row[wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh];
                acc[a][b][g] *= ir * wi;
                if constexpr (DUAL) acc2[a][b][g] *= ir * wi2;
            }
        }
    if (DUAL || p.rankR > 0 || p.bias_mode == CAPE_BIAS_VERTEX || p.act == CAPE_ACT_TANH) {
        gconv_epilogue<BM, BN, 2, 2, DUAL, float>(p, acc, acc2, n, r0, f0, wm, wn, li, lh);
        return;
    }
    gconv_epilogue_short<BM, BN, float>(p, acc, n, r0, f0, wm, wn, li, lh);
}

// =============================================================================================================
// Weight gradient of plain sources on the same arithmetic:  dW_s[c, f] = sum_{n, r} X_s[n, r, c] * dz[n, r, f].
// Decomposition, staging and partial-slab output of dw_split_kernel (gemm_split.h): the contraction runs over the vertices,
// BOTH operands are activations and are transposed in the stage (a thread reads 8 consecutive rows of its 1-2 channels,
// splits them and writes one 16-byte row segment per piece plane [channel][row]).  The scales must be constant along the
// contraction, i.e. over the rows of a workgroup's (sample group, row range): one power of two per operand and workgroup,
// from the maximum of the row bounds of its range (read in a prologue: <= 4 floats per row).  A row far below the range's
// maximum loses relative precision, but its contribution to the sum is smaller by the same factor: the error of an element
// is <= 2^-22 |x| + 2^-40 max|x|, against fp32's own 2^-24 |x z| per term.
// =============================================================================================================
// Round 5: a workgroup is TWO groups of four waves (KG = 2, 512 threads).  Each group runs the pipeline above on its own half
// of the workgroup's row range with its own LDS stages -- the two groups share a CU the way two workgroups used to -- and the
// halves' accumulators are added through LDS before ONE partial slab is written: half the slabs to write and to reduce
// (the slab store of 512 workgroups was most of the launch's 11 us floor, profiles/r05_exp_dw_h2_phases.txt) at the
// per-CU concurrency of the two-workgroup form.  Fixed order (group 0 + group 1): bit-reproducible.
constexpr int H2_DW_KG = 2;
// V4 (round 6): the operands are read with 16-byte loads -- a thread takes FOUR consecutive channels of 8 (4 for a 64-wide
// operand) consecutive rows, waves 0-1 of a group the source tile, waves 2-3 the gradient tile -- instead of one channel of
// eight rows with 4-byte loads: a quarter of the load instructions (32 -> 8 per thread and chunk) and of their address
// arithmetic.  For the LDS stores of that pattern to spread over the banks (lanes = channel quads: with the channels of a quad
// in consecutive 64-byte rows every store of a wave would fall on four 16-byte slots) the rows of a quad are ROTATED by
// (quad >> 2) & 3 inside the stage; the accumulator rows / columns come out in the same rotated order and the slab store
// undoes it.  CAPE_DW_V4=0 keeps the 4-byte form (A/B).
template <int CT, int FT, bool V4 = true>
__global__ __launch_bounds__(256 * H2_DW_KG, 1) void dw_h2_kernel(DwParams p) {
    constexpr int RK = 32, KG = H2_DW_KG;
    constexpr int WTM = CT / 2, WTN = FT / 2;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int CPA = CT / 64, CPB = FT / 64;       // channels per thread (8 rows each): lane, lane + 64
    static_assert(TM >= 1 && TN >= 1 && (CT == 64 || CT == 128) && (FT == 64 || FT == 128), "tile");
    // TWO stages of unpadded 64-byte rows, 16-byte segments XOR-swizzled by (row >> 2) & 3 (the
    // fragment-read pattern of gemm_h2_kernel: conflict-free ds_read_b128; the staging writes are 2-way, inside the
    // instruction's own issue time)
    constexpr int ROW = 64, APL2 = CT * ROW, BPL2 = FT * ROW, STAGE2 = 2 * (APL2 + BPL2);
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[KG * 2 * STAGE2];
    __shared__ float red[2][4 * KG];
    static_assert(2 * STAGE2 >= TM * TN * 16 * 256 * 4, "a group's stages hold its accumulator for the final sum");

    const int kg = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);      // contraction half of this wave's group
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;       // everything below is per GROUP of four waves
    unsigned char *smem = smem_all + kg * 2 * STAGE2;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int rg = tid >> 6;
    const int ca = tid & 63, fb = tid & 63;

    const int ntiles = p.tile_off[p.nsrc];
    int tile, split;
    if (!cape_map_dw_block(blockIdx.x, ntiles, p.ngroups * p.rsplit, tile, split)) return;
    const int grp = split / p.rsplit;
    const int rs = split % p.rsplit;
    const int n_begin = grp * p.samples_per_group;
    const int n_end = min(p.N, n_begin + p.samples_per_group);
    int si = 0;
    while (si + 1 < p.nsrc && tile >= p.tile_off[si + 1]) ++si;
    // (the source is indexed at run time: its fields are read through the kernel-argument segment -- scalar loads with a register
    // offset -- and copied; a reference into the by-value parameter can make hipcc copy the whole block to scratch memory)
    typedef const __attribute__((address_space(4))) SrcDev *SrcTab;
    const SrcTab src_tab = (SrcTab)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(DwParams, s));
    struct { const float *x, *rm; long long xs; int ldx, C, rmw; } S;
    S.x = src_tab[si].x; S.rm = src_tab[si].rm; S.xs = src_tab[si].xs; S.ldx = src_tab[si].ldx; S.C = src_tab[si].C; S.rmw = src_tab[si].rmw;
    const int lt = tile - p.tile_off[si];
    const int c0 = (lt / p.ftiles) * CT;
    const int f0 = (lt % p.ftiles) * FT;
    const int wra = rs * p.rows_per_split;                        // the WORKGROUP's row range ...
    const int rb = min(p.Mo, wra + p.rows_per_split);
    // ... and this group's part of it: whole chunks, the same number for every group (a part that reaches past rb reads zeros
    // there -- the buffer resources end at row rb -- so the groups run the same trip count and share their barriers)
    const int chunks = ((rb - wra + RK - 1) / RK + KG - 1) / KG;
    const int ra = wra + kg * chunks * RK;
    const bool two = ((p.dz2_mask >> si) & 1u) != 0;
    const float *dz0 = two ? p.dz2 : p.dz;

    // ---- scales of the workgroup's range (one pair for both groups: their accumulators are added unscaled)
    float sx, sz, inv;
    {
        const float *rmz = two ? p.dz2rm : p.dzrm;
        const int wz = two ? p.dz2rmw : p.dzrmw;
        float mx = 0.f, mz = 0.f;
        for (int n = n_begin; n < n_end; ++n) {
            const float *px = S.rm + ((long long)n * p.Mo + wra) * S.rmw;
            const float *pz = rmz + ((long long)n * p.Mo + wra) * wz;
            // (row widths are multiples of 4 and the arrays 16-byte aligned: float4 loads, four in flight per thread)
            const float4 *px4 = reinterpret_cast<const float4 *>(px), *pz4 = reinterpret_cast<const float4 *>(pz);
            const int nx = (rb - wra) * (S.rmw >> 2), nz = (rb - wra) * (wz >> 2);
#pragma unroll 4
            for (int i = (int)threadIdx.x; i < nx; i += 256 * KG) { const float4 v = px4[i]; mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))); }
#pragma unroll 4
            for (int i = (int)threadIdx.x; i < nz; i += 256 * KG) { const float4 v = pz4[i]; mz = fmaxf(mz, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))); }
        }
        mx = h2_max_ror(mx); mz = h2_max_ror(mz);
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mz = fmaxf(mz, __shfl_xor(mz, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32)); mz = fmaxf(mz, __shfl_xor(mz, 32));
        if (lane == 0) { red[0][kg * 4 + wave] = mx; red[1][kg * 4 + wave] = mz; }
        __syncthreads();
        mx = mz = 0.f;
#pragma unroll
        for (int w = 0; w < 4 * KG; ++w) { mx = fmaxf(mx, red[0][w]); mz = fmaxf(mz, red[1][w]); }
        float ix, iz;
        h2_scale_of(mx, sx, ix);
        h2_scale_of(mz, sz, iz);
        inv = ix * iz;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    const int total = (n_end - n_begin) * chunks;
    const int rend = ra + chunks * RK;               // end of this group's part (the cursor wraps here; rows >= rb read zeros)
    int l_n = n_begin, l_r = ra;
    float xa[V4 ? 1 : CPA][8], xz[V4 ? 1 : CPB][8];
    // ---- V4: this thread's operand (waves 0-1: x, waves 2-3: dz), channel quad and row group
    constexpr int QA = CT / 4, QB = FT / 4;                                // channel quads of a tile
    constexpr int RPA = QA / 4, RPB = QB / 4;                              // rows per thread: 32 rows / (128 threads / quads)
    constexpr int RPM = RPA > RPB ? RPA : RPB;
    const bool is_b = V4 && __builtin_amdgcn_readfirstlane(tid >> 6) >= 2;   // (wave-uniform)
    const int t_op = tid & 127;
    const int qn = is_b ? QB : QA;
    const int cq = t_op % qn, rgi = t_op / qn;
    float v4[RPM][4];                                                     // (dead in the !V4 instantiations)
    int voff4 = 0;

    // ---- operand streams through buffer resources (one per operand and sample: base = the sample, num_records = the bytes
    //      below row rb).  Round 5: the loop used to form a 64-bit address per element (v_mad_i64_i32 + v_lshl_add_u64 + moves:
    //      ~160 of its ~460 VALU instructions per chunk) and to mask the rows past rb with selects that also kept hipcc from
    //      fusing the scale, the conversion and the subtraction of the split (v_fma_mixlo_f16) -- with 24 MFMAs per chunk the
    //      kernel was bound by VALU issue (3.7 cycles per VALU instruction over the whole launch), not by the matrix pipe.
    //      A buffer load takes a 32-bit per-lane byte offset against a scalar base and returns 0 past num_records, so rows
    //      >= rb read as zeros without a select and an address costs one v_add per row and chunk.  Channel columns past
    //      S.C / p.F (edge tiles) read whatever lies there inside the row range: they only reach accumulator rows / columns
    //      the store below drops (an MFMA output element depends on its own operand row and column alone).
    int xoff[8], zoff[8];                            // byte offset of this thread's row j, first channel, inside a chunk
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xoff[j] = ((8 * rg + j) * S.ldx + c0 + ca) * 4;
        zoff[j] = ((8 * rg + j) * p.lddz + f0 + fb) * 4;
    }
    __amdgpu_buffer_rsrc_t rx, rz;
    auto open_sample = [&]() {
        // (bounded at the last row's last channel, not at the end of its padded pitch: the overhang of an edge tile reads 0)
        rx = __builtin_amdgcn_make_buffer_rsrc((void *)(S.x + (long long)l_n * S.xs), 0, ((rb - 1) * S.ldx + S.C) * 4, 0x00020000);
        rz = __builtin_amdgcn_make_buffer_rsrc((void *)(dz0 + (long long)l_n * p.dzs), 0, ((rb - 1) * p.lddz + p.F) * 4, 0x00020000);
    };
    // (the row advance goes into the per-lane offset, which IS range-checked; a scalar offset operand would not be)
    // (operand-dependent values are chosen by masks on VALUES: a select between two kernel-argument addresses makes hipcc copy
    // the whole parameter block to scratch memory)
    const int mb = -(int)is_b;
    const int ld_x = S.ldx, ld_z = p.lddz;
    if constexpr (V4) voff4 = ((((rgi * RPB) * ld_z + f0 + 4 * cq) * 4) & mb) | ((((rgi * RPA) * ld_x + c0 + 4 * cq) * 4) & ~mb);
    auto load4 = [&]() __attribute__((always_inline)) {      // ... and advances the chunk cursor
        auto ld = [&](__amdgpu_buffer_rsrc_t rs, int base, int pitch, int j) __attribute__((always_inline)) {
            const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, base + j * pitch, 0, 0));
            v4[j][0] = t.x; v4[j][1] = t.y; v4[j][2] = t.z; v4[j][3] = t.w;
        };
        if (!is_b) {
            const int bx = l_r * ld_x * 4 + voff4, pitch = ld_x * 4;
#pragma unroll
            for (int j = 0; j < RPA; ++j) ld(rx, bx, pitch, j);
        } else {
            const int bz = l_r * ld_z * 4 + voff4, pitch = ld_z * 4;
#pragma unroll
            for (int j = 0; j < RPB; ++j) ld(rz, bz, pitch, j);
        }
        l_r += RK;
        if (l_r >= rend) {
            l_r = ra;
            ++l_n;
            if (l_n < n_end) open_sample();
        }
    };
    auto load_a = [&]() {
        const int bx = l_r * S.ldx * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int ch = 0; ch < CPA; ++ch)
                xa[ch][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff[j] + bx + 256 * ch, 0, 0));
    };
    auto load_b = [&]() {                            // ... and advances the chunk cursor: call after load_a
        const int bz = l_r * p.lddz * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int ch = 0; ch < CPB; ++ch)
                xz[ch][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, zoff[j] + bz + 256 * ch, 0, 0));
        l_r += RK;
        if (l_r >= rend) {
            l_r = ra;
            ++l_n;
            if (l_n < n_end) open_sample();
        }
    };
    auto store8 = [&](unsigned char *dst, int plane, const float (&v)[8], float s) {
        uint4 hi, lo;
        h2_split2s(v[0], v[1], s, hi.x, lo.x);
        h2_split2s(v[2], v[3], s, hi.y, lo.y);
        h2_split2s(v[4], v[5], s, hi.z, lo.z);
        h2_split2s(v[6], v[7], s, hi.w, lo.w);
        *reinterpret_cast<uint4 *>(dst) = hi;
        *reinterpret_cast<uint4 *>(dst + plane) = lo;
    };
    // ---- software-pipelined chunk loop (round 5).  Measured on the serial loop this replaces (load chunk it + 1 -> multiply
    //      chunk it -> barrier -> split + store chunk it + 1 -> barrier) with phases knocked out (profiles/
    //      r05_exp_dw_h2_phases.txt; average of the nine 128 x 128 launches of the step): everything 33 us; prologue + barriers +
    //      slab store alone 11; on top of that floor the global loads +14, the LDS reads + MFMAs +13, the split + LDS stores +10
    //      -- 47 us if nothing overlapped, i.e. two workgroups per CU hid only a third of it, and neither four times fewer VALU
    //      instructions nor a longer load lead moved the total (profiles/r05_exp_dw_h2_load_lead.txt).  Now the three run inside
    //      ONE instruction stream per wave: the MFMAs of chunk it with the fragment reads of their second k16 step AND the
    //      split + LDS stores of chunk it + 1 (into the other stage) placed between them (sched_group_barrier: the matrix pipe
    //      takes 32 cycles per MFMA, an MFMA issues in 4-8), each operand's loads of chunk it + 2 issued as soon as its registers
    //      are staged, one barrier per chunk.
    unsigned char *st0 = smem, *st1 = smem + STAGE2;
    const int fsw = (li >> 2) & 3;
    auto stage_a = [&](unsigned char *st) {
#pragma unroll
        for (int ch = 0; ch < CPA; ++ch) {
            const int row = ca + 64 * ch;
            store8(st + row * ROW + 16 * (rg ^ ((row >> 2) & 3)), APL2, xa[ch], sx);
        }
    };
    auto stage_b = [&](unsigned char *st) {
#pragma unroll
        for (int ch = 0; ch < CPB; ++ch) {
            const int row = fb + 64 * ch;
            store8(st + 2 * APL2 + row * ROW + 16 * (rg ^ ((row >> 2) & 3)), BPL2, xz[ch], sz);
        }
    };
    // V4: channel i of the thread's quad, rows rgi * RP .. + RP - 1 -> one 16-byte (RP = 8) or 8-byte (RP = 4) piece per plane at
    // LDS row 4 cq + ((i + (cq >> 2)) & 3) (the rotation), segment (rows / 8) ^ (cq & 3)
    auto stage4 = [&](unsigned char *st) __attribute__((always_inline)) {
        const int rot = cq >> 2, swz = cq & 3;
        auto put = [&](unsigned char *base, int plane, float s, auto RP_) __attribute__((always_inline)) {
            constexpr int RP = decltype(RP_)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * cq + ((i + rot) & 3);
                if constexpr (RP == 8) {
                    uint4 hi, lo;
                    h2_split2s(v4[0][i], v4[1][i], s, hi.x, lo.x);
                    h2_split2s(v4[2][i], v4[3][i], s, hi.y, lo.y);
                    h2_split2s(v4[4][i], v4[5][i], s, hi.z, lo.z);
                    h2_split2s(v4[6][i], v4[7][i], s, hi.w, lo.w);
                    unsigned char *d = base + row * ROW + 16 * (rgi ^ swz);
                    *reinterpret_cast<uint4 *>(d) = hi;
                    *reinterpret_cast<uint4 *>(d + plane) = lo;
                } else {
                    uint2 hi, lo;
                    h2_split2s(v4[0][i], v4[1][i], s, hi.x, lo.x);
                    h2_split2s(v4[2][i], v4[3][i], s, hi.y, lo.y);
                    unsigned char *d = base + row * ROW + 16 * ((rgi >> 1) ^ swz) + 8 * (rgi & 1);
                    *reinterpret_cast<uint2 *>(d) = hi;
                    *reinterpret_cast<uint2 *>(d + plane) = lo;
                }
            }
        };
        if (!is_b) put(st, APL2, sx, std::integral_constant<int, RPA>{});
        else put(st + 2 * APL2, BPL2, sz, std::integral_constant<int, RPB>{});
    };
    h2_half8 af[2][TM][2], bf[2][TN][2];
    auto rd = [&](const unsigned char *st, int ks) __attribute__((always_inline)) {
        const unsigned char *pa = st + (wm * WTM + li) * ROW, *pb = st + 2 * APL2 + (wn * WTN + li) * ROW;
        const int so = 16 * ((2 * ks + lh) ^ fsw);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) af[ks][a][pc] = *reinterpret_cast<const h2_half8 *>(pa + pc * APL2 + a * 32 * ROW + so);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) bf[ks][b][pc] = *reinterpret_cast<const h2_half8 *>(pb + pc * BPL2 + b * 32 * ROW + so);
    };
    auto mm = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int term = 0; term < 3; ++term)         // lo*hi, hi*lo, hi*hi: small products first
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][a][term == 0 ? 1 : 0], bf[ks][b][term == 1 ? 1 : 0],
                                                                       acc[a][b], 0, 0, 0);
    };
    constexpr int NM = 3 * TM * TN, NR = 2 * (TM + TN);
    // VALU instructions of one operand's split as hipcc emits it (per four values: v_pk_mul, v_cvt_pk_f16 x 2, v_cvt_f32_f16 x 2,
    // v_pk_fma): 12 per channel of 8 rows, plus the row offsets of the loads that follow; dealt over the MFMAs of a k16 step
    constexpr int VA = (8 * CPA + 8 + NM - 1) / NM, VB = (8 * CPB + 8 + NM - 1) / NM;     // (h2_split2s: 16 per channel of 8 rows... 8 per 4 values)
    // one chunk: stage `cur` holds chunk it, the registers chunk it + 1 (NEXT), stage `nxt` is free; MORE: chunk it + 2 exists
    // V4: one operand per thread -- its split + stores of chunk it + 1 ride behind the second k16 step's MFMAs (the first step
    // carries the fragment reads), its loads of chunk it + 2 follow
    constexpr int V4S = 8 * RPM + 8;                     // VALU instructions of stage4 (per thread), dealt over NM MFMAs
    auto chunk4 = [&](const unsigned char *cur, unsigned char *nxt, auto next, auto more) __attribute__((always_inline)) {
        constexpr bool NX = decltype(next)::value, MO = decltype(more)::value;
        rd(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd(cur, 1);
        mm(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        if constexpr (NX) stage4(nxt);
        if constexpr (MO) load4();
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (NX) {
                __builtin_amdgcn_sched_group_barrier(0x002, (V4S + NM - 1) / NM, 0);
                if (i >= NM - 8) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto chunk = [&](const unsigned char *cur, unsigned char *nxt, auto next, auto more) __attribute__((always_inline)) {
        constexpr bool NX = decltype(next)::value, MO = decltype(more)::value;
        if constexpr (V4) { chunk4(cur, nxt, next, more); return; }
        rd(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd(cur, 1);
        mm(0);
        if constexpr (NX) stage_a(nxt);
        if constexpr (MO) load_a();
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (NX) {
                __builtin_amdgcn_sched_group_barrier(0x002, VA, 0);
                if (i >= NM - 2 * CPA) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        if constexpr (NX) stage_b(nxt);
        if constexpr (MO) load_b();
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (NX) {
                __builtin_amdgcn_sched_group_barrier(0x002, VB, 0);
                if (i >= NM - 2 * CPB) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (total > 0) {
        open_sample();
        if constexpr (V4) {
            load4();
            stage4(st0);
            if (total > 1) load4();
        } else {
            load_a();
            load_b();
            stage_a(st0);
            stage_b(st0);
            if (total > 1) { load_a(); load_b(); }
        }
        __syncthreads();
        int it = 0;
        // steady state: chunks it + 1, it + 2, it + 3 exist -- no branches inside, so hipcc's own vmcnt bookkeeping stays exact
        for (; it + 3 < total; it += 2) {
            chunk(st0, st1, std::true_type{}, std::true_type{});      // multiply chunk it, stage it + 1, load it + 2
            __syncthreads();
            chunk(st1, st0, std::true_type{}, std::true_type{});
            __syncthreads();
        }
        for (; it < total; ++it) {
            const unsigned char *cur = (it & 1) ? st1 : st0;
            unsigned char *nxt = (it & 1) ? st0 : st1;
            if (it + 2 < total) chunk(cur, nxt, std::true_type{}, std::true_type{});
            else if (it + 1 < total) chunk(cur, nxt, std::true_type{}, std::false_type{});
            else chunk(cur, nxt, std::false_type{}, std::false_type{});
            __syncthreads();
        }
    }

    // ---- the halves' sum, in a fixed order: group 1 parks its accumulator in its own (now idle) stages, group 0 adds it
    //      (element k of thread t at [k][t]: consecutive lanes, consecutive words)
    if constexpr (KG == 2) {
        float *park = reinterpret_cast<float *>(smem_all + 2 * STAGE2);
        if (kg == 1) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int g = 0; g < 16; ++g) park[((a * TN + b) * 16 + g) * 256 + tid] = acc[a][b][g];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[a][b][g] += park[((a * TN + b) * 16 + g) * 256 + tid];
    }
    float *out = p.ws + (long long)split * p.slab + p.part_off[si];
    // V4: stage row R holds channel 4 q + ((R & 3) - (q >> 2) & 3), q = R >> 2 (accumulator rows and columns alike)
    auto unrot = [](int R) { return V4 ? (R & ~3) | (((R & 3) - (R >> 4)) & 3) : R; };
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = f0 + unrot(wn * WTN + b * 32 + li);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int c = c0 + unrot(wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh);
                if (c < S.C && f < p.F) out[(long long)c * p.F + f] = acc[a][b][g] * inv;
            }
        }
}

// Eligibility on top of the plain-source conditions of plan_fwd: fp32 storage, every source with piece planes and row bounds,
// whole 32-channel chunks, rows addressable through a 32-bit buffer offset, an output of at least 64 columns.
inline bool h2_eligible(const GconvParams &p, bool dual) {
    static const int on = getenv("CAPE_GEMM_H2") ? atoi(getenv("CAPE_GEMM_H2")) : 1;      // 0: A/B against the six-product bf16 form
    if (!on || p.F < 64 || !p.wsi || (dual && !p.wsi2)) return false;
    // short contractions (one to four chunks) are latency-bound launches in which this kernel's longer prologue (row bounds,
    // DMA set-up, two LDS stages -> four instead of five workgroups per CU) costs more than the halved matrix work returns:
    // 0.70-0.85x of the six-product form at 64 / 128 contraction indices with 64 output columns (profiles/r04_h2_bench_*.txt)
    int ktot = 0;
    for (int i = 0; i < p.nsrc; ++i) ktot += p.s[i].C;
    if (!(ktot >= 256 || (ktot >= 128 && p.F >= 128))) return false;
    for (int i = 0; i < p.nsrc; ++i) {
        const SrcDev &S = p.s[i];
        if (S.rp || !S.wh || !S.wl || !S.rm || S.rmw < 4 || S.rmw > 16 || (S.rmw & 3)) return false;
        if (S.C % H2_KC != 0 || S.C < H2_KC) return false;
        if ((S.ldx & 3) || (S.xs & 3) || (reinterpret_cast<uintptr_t>(S.x) & 15)) return false;
        if ((long long)p.Mo * S.ldx >= (1LL << 29) || (long long)p.F * S.wp >= (1LL << 30)) return false;
        if ((S.wp & 7) || (reinterpret_cast<uintptr_t>(S.wh) & 15) || (reinterpret_cast<uintptr_t>(S.wl) & 15)) return false;
        if (S.w2) {
            if (!dual || !S.wh2 || !S.wl2 || (S.wp2 & 7) || (long long)p.F * S.wp2 >= (1LL << 30)) return false;
            if ((reinterpret_cast<uintptr_t>(S.wh2) & 15) || (reinterpret_cast<uintptr_t>(S.wl2) & 15)) return false;
        }
    }
    return true;
}

inline bool h2x_wanted(bool dual, int N, int Mo, int F, int Ktot);        // gemm_h2x.h: the wide tile, 128 x 256
inline void h2_tile(bool dual, int N, int Mo, int F, int Ktot, int &BM, int &BN) {
    if (h2x_wanted(dual, N, Mo, F, Ktot)) { BM = 128; BN = 256; return; }
    if (dual) { BM = 128; BN = 64; return; }
    // CAPE_H2_TILE=BMxBN (experiments, tools/ubench/h2_bench.cpp): force one of the four single-accumulator tiles
    static const char *force = getenv("CAPE_H2_TILE");
    if (force && sscanf(force, "%dx%d", &BM, &BN) == 2 && (BM == 64 || BM == 128) && (BN == 64 || BN == 128)) {
        if (BN == 128 && F < 128) BN = 64;
        return;
    }
    // measured over the model's shapes (tools/ubench/h2_bench.cpp, profiles/r04_h2_bench_tiles.txt): the 128 x 128 tile wins
    // only where the contraction is long (its fewer L2 re-reads matter) and still gives every CU its two workgroups; short
    // contractions are latency-bound launches of one or two rounds of tiles and want as many workgroups in flight as fit
    const long long big = (long long)N * ((Mo + 127) / 128) * ((F + 127) / 128);
    if (F >= 128 && big >= 384 && Ktot >= 512) { BM = 128; BN = 128; }
    else { BM = 64; BN = 64; }
}

// weight gradient: the dw_split plan (family 3) with row bounds on every operand
inline bool h2_dw_eligible(const DwParams &p) {
    static const int on = getenv("CAPE_DW_H2") ? atoi(getenv("CAPE_DW_H2")) : 1;          // 0: A/B against dw_split_kernel
    if (!on || !p.dzrm || p.dzrmw < 4 || (p.dzrmw & 3) || (p.dz2_mask && (!p.dz2rm || p.dz2rmw < 4 || (p.dz2rmw & 3)))) return false;
    // (the operands are read through 32-bit buffer offsets: a sample's rows, plus an edge tile's overhang, within 2^31 bytes)
    if (((long long)p.Mo * p.lddz + 512) * 4 >= (1LL << 31)) return false;
    for (int i = 0; i < p.nsrc; ++i) {
        if (!p.s[i].rm || p.s[i].rmw < 4 || (p.s[i].rmw & 3)) return false;
        if (((long long)p.Mo * p.s[i].ldx + 512) * 4 >= (1LL << 31)) return false;
    }
    return true;
}

inline void h2_dw_launch(const DwParams &p, int ct, int ft, dim3 grid, hipStream_t st) {
    const dim3 block(256 * H2_DW_KG);
    static const int v4 = getenv("CAPE_DW_V4") ? atoi(getenv("CAPE_DW_V4")) : 1;
    // (16-byte loads: every operand with 16-byte aligned rows)
    bool ok = v4 && (p.lddz & 3) == 0 && (p.dzs & 3) == 0 && (reinterpret_cast<uintptr_t>(p.dz) & 15) == 0 &&
              (!p.dz2 || (reinterpret_cast<uintptr_t>(p.dz2) & 15) == 0);
    for (int i = 0; i < p.nsrc && ok; ++i)
        ok = (p.s[i].ldx & 3) == 0 && (p.s[i].xs & 3) == 0 && (reinterpret_cast<uintptr_t>(p.s[i].x) & 15) == 0;
    // (128 x 128 tiles only: with a 64-wide operand the two wave pairs of a group get unequal shares -- measured 12.8 -> 18.4 us on
    // 128 x 64, 17.3 -> 26.2 on 64 x 128 -- and 64 x 64 gains nothing; 128 x 128: 64.1 -> 62.6, 37.1 -> 36.3, 34.3 -> 33.6 us)
    if (ok && ct == 128 && ft == 128) {
        CAPE_LAUNCH((dw_h2_kernel<128, 128, true>), grid, block, 0, st, p);
        return;
    }
    if (ct == 64 && ft == 64) CAPE_LAUNCH((dw_h2_kernel<64, 64, false>), grid, block, 0, st, p);
    else if (ct == 64) CAPE_LAUNCH((dw_h2_kernel<64, 128, false>), grid, block, 0, st, p);
    else if (ft == 64) CAPE_LAUNCH((dw_h2_kernel<128, 64, false>), grid, block, 0, st, p);
    else CAPE_LAUNCH((dw_h2_kernel<128, 128, false>), grid, block, 0, st, p);
}

inline void h2x_launch(const GconvParams &p, dim3 grid, hipStream_t st);
inline void h2_launch(const GconvParams &p, bool dual, int BM, int BN, dim3 grid, hipStream_t st) {
    if (!dual && BM == 128 && BN == 256) h2x_launch(p, grid, st);
    else if (dual) CAPE_LAUNCH((gemm_h2_kernel<128, 64, true>), grid, dim3(256), 0, st, p);
    else if (BM == 128 && BN == 128) CAPE_LAUNCH((gemm_h2_kernel<128, 128, false>), grid, dim3(256), 0, st, p);
    else if (BM == 128) CAPE_LAUNCH((gemm_h2_kernel<128, 64, false>), grid, dim3(256), 0, st, p);
    else if (BN == 128) CAPE_LAUNCH((gemm_h2_kernel<64, 128, false>), grid, dim3(256), 0, st, p);
    else CAPE_LAUNCH((gemm_h2_kernel<64, 64, false>), grid, dim3(256), 0, st, p);
}

}  // namespace

#include "gemm_h2x.h"
