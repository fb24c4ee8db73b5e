// fp32 GEMM for PLAIN sources on the bf16 matrix pipe ("bf16x6"): the same contraction as gemm_plain.h
// (y[n] = epilogue(sum_s X_s[n] @ B_s), reference lib/models.py:99-102), with every fp32 operand split EXACTLY
// into three bf16 pieces while it is staged into LDS,
//     x = hi + mid + lo      (8 + 8 + 8 significand bits; truncation split, both subtractions exact)
// and six of the nine cross products accumulated in the fp32 accumulator of v_mfma_f32_32x32x16_bf16
// (smallest first):  hi*lo + lo*hi + mid*mid + hi*mid + mid*hi + hi*hi.
// The three dropped products (mid*lo, lo*mid, lo*lo) amount to 2^-24 of the exact product in the rms -- one fp32
// rounding -- and at most 2^-21 (truncation split: |mid| < 2^-7 |x|, |lo| < 2^-15 |x|; tests/test_bf16_split_numerics.py);
// measured against float64 a contraction is as accurate as an fp32 FMA chain (tools/ubench/gemm_bf16x3.hip:
// rms error 4.96e-07 vs 5.74e-07 of rms(ref) at K = 1024; all nine products: 4.95e-07).  The identity x = hi+mid+lo
// is exact for every finite x; lo is a bf16 number for |x| >= 2^-110 (below, <= 2^-133 is lost).  The bf16 pipe sustains
// ~1.7-1.85 PFLOP/s on this chip (same ubench), /6 = ~290 TFLOP/s fp32-equivalent against the 142 TFLOP/s the
// exact-fp32 MFMA sustains.
//
// Scope: non-DUAL launches whose sources all have C % 32 == 0 (whole 32-wide chunks, no masking); weights in
// either layout (contraction-contiguous, or output-contiguous with a transposing stage).  Same block mapping,
// epilogue (rank-1 terms, bias, activation, de-interleave) and software pipeline as gemm_plain_kernel.
// Not bit-identical to the fp32-MFMA kernels (different summation tree); inf inputs give NaN (inf - inf in the split).
#pragma once
#include <type_traits>

#include "gconv_shared.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gs_u32x4 __attribute__((ext_vector_type(4)));      // native vector: stays in registers when held in arrays

#ifndef CAPE_GEMM_BF16X6_DEFAULT
#define CAPE_GEMM_BF16X6_DEFAULT 1
#endif
#ifndef CAPE_DW_BF16X6_DEFAULT
#define CAPE_DW_BF16X6_DEFAULT 1      // weight gradient on the bf16 pipe (dw_split_kernel); CAPE_DW_BF16X6=0 -> exact-fp32 MFMA
#endif

constexpr int GS_KC = 32;       // contraction indices per staged chunk = two k16 MFMA steps
constexpr int GS_PITCH = 80;    // bytes per LDS row of one piece plane: 32 bf16 + 16 B pad (conflict-free ds_read_b128)
__device__ __forceinline__ int gs_seg(int, int seg) { return seg; }
constexpr int GS_BIG_MINB = 2;
// (Measured and dropped, kept under tools/ubench: unpadded 64-byte rows with XOR-swizzled segments at three workgroups per
// CU, a round-to-nearest split, de-phased workgroups, pre-split operands, a double-buffered single-barrier K-loop, an
// explicit two-group ping-pong -- all within +-5 % of this kernel; profiles/r02_ubench_split_variants.txt,
// profiles/r03_ubench_v4_*.txt.)

__device__ __forceinline__ unsigned gs_bits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float gs_float(unsigned v) { return __builtin_bit_cast(float, v); }

// two fp32 -> their three bf16 pieces, packed pairwise (first element in the low half)
__device__ __forceinline__ void gs_split2(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    const unsigned h0 = gs_bits(x0) & 0xFFFF0000u, h1 = gs_bits(x1) & 0xFFFF0000u;
    const float r0 = x0 - gs_float(h0), r1 = x1 - gs_float(h1);                 // exact
    const unsigned m0 = gs_bits(r0) & 0xFFFF0000u, m1 = gs_bits(r1) & 0xFFFF0000u;
    const float s0 = r0 - gs_float(m0), s1 = r1 - gs_float(m1);                 // exact, <= 8 significant bits left
    hi = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    mid = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    lo = __builtin_amdgcn_perm(gs_bits(s1), gs_bits(s0), 0x07060302u);
}

// eight consecutive contraction indices -> one 16-byte row segment per piece plane
__device__ __forceinline__ void gs_store8(unsigned char *dst, int plane, const float (&v)[8]) {
    uint4 hi, mid, lo;
    gs_split2(v[0], v[1], hi.x, mid.x, lo.x);
    gs_split2(v[2], v[3], hi.y, mid.y, lo.y);
    gs_split2(v[4], v[5], hi.z, mid.z, lo.z);
    gs_split2(v[6], v[7], hi.w, mid.w, lo.w);
    *reinterpret_cast<uint4 *>(dst) = hi;
    *reinterpret_cast<uint4 *>(dst + plane) = mid;
    *reinterpret_cast<uint4 *>(dst + 2 * plane) = lo;
}

// NP = 3: the exact three-way split of fp32 operands above; NP = 1: bf16-storage launches (cape_*_bf16 entry points),
// one round-to-nearest bf16 value per element (exact when the element already is a bf16 number) and ONE MFMA product
// per multiply-add.
// piece indices (0 = hi, 1 = mid, 2 = lo) of the six products, smallest first: A = {0,2,1,0,1,0}, B = {2,0,1,1,0,0};
// the single product of the one-plane (bf16 storage) form is (0, 0)
__device__ __forceinline__ constexpr int gs_ta(int np, int t) { return np == 1 ? 0 : (t == 1 ? 2 : ((t == 2 || t == 4) ? 1 : 0)); }
__device__ __forceinline__ constexpr int gs_tb(int np, int t) { return np == 1 ? 0 : (t == 0 ? 2 : ((t == 2 || t == 3) ? 1 : 0)); }

// Weights of the bf16-storage launches keep TWO bf16 planes, w = hi + mid (hi = the top 16 bits, mid = the remainder
// rounded to nearest: 16 significant bits), multiplied as two MFMA products against the one-plane activations.  The
// fp32 master weights rounded to a single bf16 were the dominant error of the bf16 model: the same rounded weight meets
// every vertex and sample, so its error does not average out -- prediction error 2.0e-2 with one plane against 6.3e-3
// when only the activations are bf16 (tools/diag_bf16_error.py, profiles/r03_diag_bf16_error.txt).  The launches are
// HBM-bound, the second product costs matrix-pipe time that was idle.
__device__ __forceinline__ void gs_store8_w2(unsigned char *dst, int plane, const float (&v)[8]) {
    uint4 hi, mid;
    unsigned *ph = &hi.x, *pm = &mid.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned h0 = gs_bits(v[2 * j]) & 0xFFFF0000u, h1 = gs_bits(v[2 * j + 1]) & 0xFFFF0000u;
        const float r0 = v[2 * j] - gs_float(h0), r1 = v[2 * j + 1] - gs_float(h1);        // exact
        ph[j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
        pm[j] = cape_pack_bf16(r0, r1);
    }
    *reinterpret_cast<uint4 *>(dst) = hi;
    *reinterpret_cast<uint4 *>(dst + plane) = mid;
}

// planes of the A (activation) and B (weight) operand and the MFMA products per multiply-add:
//   fp32 storage: 3 / 3 planes, 6 products;  bf16 storage: 1 / NPB planes (NPB = 2 for weights, 1 when B is an activation
//   too -- the weight gradient), NPB products
__device__ __forceinline__ constexpr int gs_nt(int npa, int npb) { return npa == 3 ? 6 : npb; }
__device__ __forceinline__ constexpr int gs_pa(int npa, int t) { return npa == 3 ? gs_ta(3, t) : 0; }
__device__ __forceinline__ constexpr int gs_pb(int npa, int npb, int t) { return npa == 3 ? gs_tb(3, t) : (npb == 2 ? 1 - t : 0); }   // mid first

template <int NP>
__device__ __forceinline__ void gs_store8_np(unsigned char *dst, int plane, const float (&v)[8]) {
    if constexpr (NP == 3) {
        gs_store8(dst, plane, v);
    } else {
        uint4 h;
        h.x = cape_pack_bf16(v[0], v[1]); h.y = cape_pack_bf16(v[2], v[3]);
        h.z = cape_pack_bf16(v[4], v[5]); h.w = cape_pack_bf16(v[6], v[7]);
        *reinterpret_cast<uint4 *>(dst) = h;
    }
}

// Scheduling pipeline of one k16 step: NR fragment reads (of the NEXT step) spread over the NM MFMAs of this one, one read
// after every NM / NR MFMAs (IGroupLP: 0x008 = MFMA, 0x100 = DS read).  Applies to the instructions of the enclosing
// scheduling region, i.e. the straight-line code since the last sched_barrier.
template <int NM, int NR>
__device__ __forceinline__ void gs_interleave() {
    constexpr int PER = NM / NR > 0 ? NM / NR : 1;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
}

template <int NPB>
__device__ __forceinline__ void gs_store_w(unsigned char *dst, int plane, const float (&v)[8]) {
    if constexpr (NPB == 3) gs_store8(dst, plane, v);
    else gs_store8_w2(dst, plane, v);
}

// Workgroup tile BM x BN, 4 waves as 2 x 2, wave tile (BM/2) x (BN/2).  DUAL: sources may carry a second weight set
// (w2) accumulated into a second tile, combined as relu(acc) + acc2 by the shared epilogue (res_block_affine,
// reference lib/models.py:776-793); its LDS holds a third group of planes, so the DUAL tile is 128 x 64.
// AT = float: fp32 activations, three bf16 planes per operand, six products.  AT = cape_bf16: activations (sources and
// output) stored as bf16: one activation plane; weights still fp32 in HBM, staged as two bf16 planes hi + mid (gs_store8_w2),
// two products per multiply-add, fp32 accumulate.
template <int BM, int BN, bool BKC, bool DUAL = false, typename AT = float>
__global__ __launch_bounds__(256, (BM * BN * (DUAL ? 2 : 1) >= 128 * 128) ? GS_BIG_MINB : 4) void gemm_split_kernel(GconvParams p) {
    constexpr bool BF = cape_is_bf16<AT>::value;
    constexpr int NP = BF ? 1 : 3;              // planes of the activation operand
    constexpr int NPB = BF ? 2 : 3;             // planes of the weight operand (bf16 storage: hi + mid, see gs_store8_w2)
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int PA = BM / 64, PB = BN / 64;      // staging passes of the k-contiguous form: 64 rows x 4 eight-float groups
    constexpr int KPT = BN / 8;                     // [k][n] weight staging: one output column, KPT consecutive k per thread
    constexpr int APLANE = BM * GS_PITCH, BPLANE = BN * GS_PITCH;
    static_assert(TM >= 1 && TN >= 1 && (BN == 64 || BN == 128), "tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NP * APLANE + NPB * (DUAL ? 2 : 1) * BPLANE];
    unsigned char *sA = smem, *sB = smem + NP * APLANE;
    unsigned char *sB2 = sB + NPB * BPLANE;         // DUAL only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int q = tid & 3, r = tid >> 2;
    const int bcol = tid % BN, kg = tid / BN;       // [k][n] staging coordinates

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
    for (int si = 0; si < p.nsrc; ++si) total += p.s[si].C / GS_KC;

    // clamped rows / columns (tile parts beyond Mo / F are computed on valid data and never stored)
    int rc[PA], fc[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) rc[i] = min(r0 + r + 64 * i, p.Mo - 1);
#pragma unroll
    for (int i = 0; i < PB; ++i) fc[i] = min(f0 + r + 64 * i, p.F - 1);
    const int fcol = min(f0 + bcol, p.F - 1);

    // ---- loader cursor: source l_si, channel offset l_c0 (every source is a whole number of chunks)
    int l_si = 0, l_c0 = 0, l_C = 0;
    const AT *l_x = nullptr;
    const float *l_w = nullptr, *l_w2 = nullptr;
    long long l_wrs = 0, l_w2rs = 0;
    int arow[PA], brow[PB], brow2[DUAL ? PB : 1];
    auto open_source = [&]() {
        const SrcDev &S = p.s[l_si];
        l_C = S.C;
        l_x = reinterpret_cast<const AT *>(S.x) + (long long)n * S.xs;
        l_w = S.w;
        l_wrs = S.wrs;
        if constexpr (DUAL) {
            l_w2 = S.w2;
            l_w2rs = S.w2rs;
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) arow[i] = rc[i] * S.ldx;
        if constexpr (BKC) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                brow[i] = fc[i] * (int)S.wcs;
                if constexpr (DUAL) brow2[i] = fc[i] * (int)S.w2cs;
            }
        }
    };

    float4 ra[PA][2];
    gs_u32x4 rab[PA];                                 // bf16 storage: 8 consecutive elements = one 16-byte load, staged as is
    float rbv[BKC ? PB : 1][BKC ? 8 : KPT];
    float rbv2[DUAL ? (BKC ? PB : 1) : 1][DUAL ? (BKC ? 8 : KPT) : 1];
    bool s_has2 = false;                            // of the chunk held in the staging registers
    auto load_regs = [&]() {
        const int c = l_c0 + 8 * q;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if constexpr (BF) {
                rab[i] = *reinterpret_cast<const gs_u32x4 *>(l_x + arow[i] + c);
            } else {
                ra[i][0] = *reinterpret_cast<const float4 *>(l_x + arow[i] + c);
                ra[i][1] = *reinterpret_cast<const float4 *>(l_x + arow[i] + c + 4);
            }
        }
        if constexpr (BKC) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const float4 u = *reinterpret_cast<const float4 *>(l_w + brow[i] + c);
                const float4 v = *reinterpret_cast<const float4 *>(l_w + brow[i] + c + 4);
                rbv[i][0] = u.x; rbv[i][1] = u.y; rbv[i][2] = u.z; rbv[i][3] = u.w;
                rbv[i][4] = v.x; rbv[i][5] = v.y; rbv[i][6] = v.z; rbv[i][7] = v.w;
            }
        } else {
            const float *wk = l_w + (long long)(l_c0 + kg * KPT) * l_wrs + fcol;
#pragma unroll
            for (int j = 0; j < KPT; ++j) rbv[0][j] = wk[(long long)j * l_wrs];
        }
        if constexpr (DUAL) {
            s_has2 = l_w2 != nullptr;
            if (s_has2) {
                if constexpr (BKC) {
#pragma unroll
                    for (int i = 0; i < PB; ++i) {
                        const float4 u = *reinterpret_cast<const float4 *>(l_w2 + brow2[i] + c);
                        const float4 v = *reinterpret_cast<const float4 *>(l_w2 + brow2[i] + c + 4);
                        rbv2[i][0] = u.x; rbv2[i][1] = u.y; rbv2[i][2] = u.z; rbv2[i][3] = u.w;
                        rbv2[i][4] = v.x; rbv2[i][5] = v.y; rbv2[i][6] = v.z; rbv2[i][7] = v.w;
                    }
                } else {
                    const float *wk2 = l_w2 + (long long)(l_c0 + kg * KPT) * l_w2rs + fcol;
#pragma unroll
                    for (int j = 0; j < KPT; ++j) rbv2[0][j] = wk2[(long long)j * l_w2rs];
                }
            }
        }
        l_c0 += GS_KC;
        if (l_c0 >= l_C) {
            l_c0 = 0;
            ++l_si;
            if (l_si < p.nsrc) open_source();
        }
    };

    auto store_regs = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            unsigned char *dst = sA + (r + 64 * i) * GS_PITCH + 16 * gs_seg(r + 64 * i, q);
            if constexpr (BF) {
                *reinterpret_cast<gs_u32x4 *>(dst) = rab[i];
            } else {
                const float v[8] = {ra[i][0].x, ra[i][0].y, ra[i][0].z, ra[i][0].w, ra[i][1].x, ra[i][1].y, ra[i][1].z, ra[i][1].w};
                gs_store8(dst, APLANE, v);
            }
        }
        if constexpr (BKC) {
#pragma unroll
            for (int i = 0; i < PB; ++i) gs_store_w<NPB>(sB + (r + 64 * i) * GS_PITCH + 16 * gs_seg(r + 64 * i, q), BPLANE, rbv[i]);
        } else {
#pragma unroll
            for (int g = 0; g < KPT / 8; ++g) {
                const float v[8] = {rbv[0][8 * g + 0], rbv[0][8 * g + 1], rbv[0][8 * g + 2], rbv[0][8 * g + 3],
                                    rbv[0][8 * g + 4], rbv[0][8 * g + 5], rbv[0][8 * g + 6], rbv[0][8 * g + 7]};
                gs_store_w<NPB>(sB + bcol * GS_PITCH + 16 * gs_seg(bcol, kg * (KPT / 8) + g), BPLANE, v);
            }
        }
        if constexpr (DUAL) {
            if (s_has2) {
                if constexpr (BKC) {
#pragma unroll
                    for (int i = 0; i < PB; ++i) gs_store_w<NPB>(sB2 + (r + 64 * i) * GS_PITCH + 16 * gs_seg(r + 64 * i, q), BPLANE, rbv2[i]);
                } else {
#pragma unroll
                    for (int g = 0; g < KPT / 8; ++g) {
                        const float v[8] = {rbv2[0][8 * g + 0], rbv2[0][8 * g + 1], rbv2[0][8 * g + 2], rbv2[0][8 * g + 3],
                                            rbv2[0][8 * g + 4], rbv2[0][8 * g + 5], rbv2[0][8 * g + 6], rbv2[0][8 * g + 7]};
                        gs_store_w<NPB>(sB2 + bcol * GS_PITCH + 16 * gs_seg(bcol, kg * (KPT / 8) + g), BPLANE, v);
                    }
                }
            }
        }
    };

    // ---- multiply one staged chunk.  Lane (li, lh) of v_mfma_f32_32x32x16_bf16 supplies row/column li and the
    // contraction indices 8*lh .. 8*lh+7 of the k16 step: one 16-byte LDS read per operand piece.
    // The fragment reads of k16 step 1 are issued one per MFMA group INSIDE step 0 (gs_interleave): a ds_read_b128 holds its
    // wave's issue port for ~29 cycles, so the twelve reads of a step in a row stall that wave's MFMA stream for ~350
    // cycles -- measured on the producer/consumer prototype (profiles/r03_ubench_v5_*.txt: 2448 -> 1780 cycles per chunk
    // for a multiply-only wave) and on this kernel's structure (profiles/r03_ubench_ilv.txt: +6..11 % on the 128 x 128
    // tiles, +2..4 % on the 64 x 64 ones).  hipcc by itself sinks every read next to its first use.
    auto compute = [&](bool has2) {
        // rows wm*WTM + a*32 + li: all tile offsets are multiples of 32, so the swizzle term (row >> 2) & 3 is that of li
        const unsigned char *pa = sA + (wm * WTM + li) * GS_PITCH;
        const unsigned char *pb = sB + (wn * WTN + li) * GS_PITCH;
        constexpr int NT = gs_nt(NP, NPB);
        static_assert(GS_KC == 32, "two k16 steps per chunk");
        bf16x8 af[2][TM][NP], bf[2][TN][NPB], bf2[2][DUAL ? TN : 1][DUAL ? NPB : 1];
        auto rd = [&](int ks, auto W2) {
            const int so = 16 * gs_seg(li, lh + 2 * ks);             // byte offset of this lane's 16-byte segment
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    af[ks][a][pc] = *reinterpret_cast<const bf16x8 *>(pa + pc * APLANE + a * 32 * GS_PITCH + so);
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int pc = 0; pc < NPB; ++pc)
                    bf[ks][b][pc] = *reinterpret_cast<const bf16x8 *>(pb + pc * BPLANE + b * 32 * GS_PITCH + so);
            if constexpr (decltype(W2)::value) {
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int pc = 0; pc < NPB; ++pc)
                        bf2[ks][b][pc] = *reinterpret_cast<const bf16x8 *>(pb + NPB * BPLANE + pc * BPLANE + b * 32 * GS_PITCH + so);
            }
        };
        // piece indices (0 = hi, 1 = mid, 2 = lo) of the six products, smallest first (bf16 storage: activation x weight-mid, then activation x weight-hi)
        auto mm = [&](int ks, auto W2) {
#pragma unroll
            for (int term = 0; term < NT; ++term)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][a][gs_pa(NP, term)], bf[ks][b][gs_pb(NP, NPB, term)], acc[a][b], 0, 0, 0);
            if constexpr (decltype(W2)::value) {
#pragma unroll
                for (int term = 0; term < NT; ++term)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][a][gs_pa(NP, term)], bf2[ks][b][gs_pb(NP, NPB, term)], acc2[a][b], 0, 0, 0);
            }
        };
        auto chunk = [&](auto W2) {
            constexpr bool w2 = decltype(W2)::value;
            rd(0, W2);
            __builtin_amdgcn_sched_barrier(0);
            rd(1, W2);
            mm(0, W2);
            gs_interleave<NT * TM * TN * (w2 ? 2 : 1), NP * TM + NPB * TN * (w2 ? 2 : 1)>();
            __builtin_amdgcn_sched_barrier(0);
            mm(1, W2);
        };
        if constexpr (DUAL) {
            if (has2) chunk(std::true_type{});
            else chunk(std::false_type{});
        } else {
            chunk(std::false_type{});
        }
    };

    open_source();
    load_regs();
    store_regs();
    bool c_has2 = s_has2;               // of the chunk staged in LDS
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const bool more = it + 1 < total;
        if (more) load_regs();          // chunk it+1: global -> registers, in flight during the MFMAs below
        compute(c_has2);
        __syncthreads();
        if (more) store_regs();         // split + LDS store
        c_has2 = s_has2;
        __syncthreads();
    }

    if (DUAL || p.rankR > 0 || p.bias_mode == CAPE_BIAS_VERTEX || p.act == CAPE_ACT_TANH) {
        gconv_epilogue<BM, BN, 2, 2, DUAL, AT>(p, acc, acc2, n, r0, f0, wm, wn, li, lh);
        return;
    }
    gconv_epilogue_short<BM, BN, AT>(p, acc, n, r0, f0, wm, wn, li, lh);
}

// =============================================================================================================
// Weight gradient of plain sources on the bf16 pipe:  dW_s[c, f] = sum_{n, r} X_s[n, r, c] * dz[n, r, f].
// Same decomposition and partial-slab output as dw_plain_kernel (gemm_plain.h); the contraction runs over the
// vertices, so BOTH operands are transposed in the stage: a thread reads 8 consecutive rows of its 1-2 channels
// (loads coalesced over the channels), splits them and writes one 16-byte row segment per piece plane [channel][row].
// Requires: sources plain, C % 4 == 0, 16-byte aligned rows; dz likewise with F % 2 == 0.
// =============================================================================================================
template <int CT, int FT, typename AT = float>
__global__ __launch_bounds__(256, (CT * FT >= 128 * 128) ? GS_BIG_MINB : 3) void dw_split_kernel(DwParams p) {
    constexpr int NP = cape_is_bf16<AT>::value ? 1 : 3;       // see gemm_split_kernel
    constexpr int RK = 32;
    constexpr int WTM = CT / 2, WTN = FT / 2;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int CPA = CT / 64, CPB = FT / 64;      // channels per thread (8 rows each): 64 lanes x CP channels = one tile row
    constexpr int APLANE = CT * GS_PITCH, BPLANE = FT * GS_PITCH;
    static_assert(TM >= 1 && TN >= 1 && (CT == 64 || CT == 128) && (FT == 64 || FT == 128), "tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NP * (APLANE + BPLANE)];
    unsigned char *sA = smem, *sB = smem + NP * APLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int rg = tid >> 6;                          // rows 8*rg .. 8*rg+7 of the 32-row chunk
    const int ca = (tid & 63) * CPA, fb = (tid & 63) * CPB;

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

    // columns beyond C / F are read from column 0 and feed output rows / columns that are never stored
    const int a_col = (c0 + ca < S.C) ? c0 + ca : 0;
    const int b_col = (f0 + fb < p.F) ? f0 + fb : 0;
    const AT *dz0 = reinterpret_cast<const AT *>(((p.dz2_mask >> si) & 1u) ? p.dz2 : p.dz);

    const int chunks = (rb - ra + RK - 1) / RK;
    const int total = (n_end - n_begin) * chunks;
    int l_n = n_begin, l_r = ra;                      // loader cursor
    float xa[CPA][8], xz[CPB][8];
    unsigned ok = 0;

    auto load_regs = [&]() {
        const AT *xb = reinterpret_cast<const AT *>(S.x) + (long long)l_n * S.xs + a_col;
        const AT *zb = dz0 + (long long)l_n * p.dzs + b_col;
        ok = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = l_r + 8 * rg + j;
            ok |= (r < rb ? 1u : 0u) << j;
            const int rr = min(r, rb - 1);
            if constexpr (CPA == 2) {
                const float2 v = cape_ld2(xb + (long long)rr * S.ldx);
                xa[0][j] = v.x; xa[1][j] = v.y;
            } else {
                xa[0][j] = cape_ld(xb + (long long)rr * S.ldx);
            }
            if constexpr (CPB == 2) {
                const float2 v = cape_ld2(zb + (long long)rr * p.lddz);
                xz[0][j] = v.x; xz[1][j] = v.y;
            } else {
                xz[0][j] = cape_ld(zb + (long long)rr * p.lddz);
            }
        }
        l_r += RK;
        if (l_r >= rb) { l_r = ra; ++l_n; }
    };
    auto store_regs = [&]() {
        // rows beyond the split's range contribute zero (they were read from the clamped last row)
#pragma unroll
        for (int ch = 0; ch < CPA; ++ch) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ((ok >> j) & 1u) ? xa[ch][j] : 0.f;
            gs_store8_np<NP>(sA + (ca + ch) * GS_PITCH + 16 * gs_seg(ca + ch, rg), APLANE, v);
        }
#pragma unroll
        for (int ch = 0; ch < CPB; ++ch) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ((ok >> j) & 1u) ? xz[ch][j] : 0.f;
            gs_store8_np<NP>(sB + (fb + ch) * GS_PITCH + 16 * gs_seg(fb + ch, rg), BPLANE, v);
        }
    };
    auto compute = [&]() {                              // (fragment reads interleaved as in gemm_split_kernel)
        const unsigned char *pa = sA + (wm * WTM + li) * GS_PITCH;
        const unsigned char *pb = sB + (wn * WTN + li) * GS_PITCH;
        constexpr int NT = NP == 3 ? 6 : 1;
        static_assert(RK == 32, "two k16 steps per chunk");
        bf16x8 af[2][TM][NP], bf[2][TN][NP];
        auto rd = [&](int ks) {
            const int so = 16 * gs_seg(li, lh + 2 * ks);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    af[ks][a][pc] = *reinterpret_cast<const bf16x8 *>(pa + pc * APLANE + a * 32 * GS_PITCH + so);
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    bf[ks][b][pc] = *reinterpret_cast<const bf16x8 *>(pb + pc * BPLANE + b * 32 * GS_PITCH + so);
        };
        auto mm = [&](int ks) {
#pragma unroll
            for (int term = 0; term < NT; ++term)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][a][gs_ta(NP, term)], bf[ks][b][gs_tb(NP, term)], acc[a][b], 0, 0, 0);
        };
        rd(0);
        __builtin_amdgcn_sched_barrier(0);
        rd(1);
        mm(0);
        gs_interleave<NT * TM * TN, NP * (TM + TN)>();
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
    };

    if (total > 0) {
        load_regs();
        store_regs();
        __syncthreads();
        for (int it = 0; it < total; ++it) {
            const bool more = it + 1 < total;
            if (more) load_regs();
            compute();
            __syncthreads();
            if (more) store_regs();
            __syncthreads();
        }
    }

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

// Eligibility (on top of gp_weight_layout() >= 0): no second weight set, whole chunks, an output wide enough for
// the 64-column MFMA tile pair.
inline bool gs_eligible(const GconvParams &p, bool dual) {
    // DUAL launches (affine blocks' forward) take the 128 x 64 DUAL tile; CAPE_GEMM_BF16X6_DUAL=0 keeps them on the fp32 MFMA
    static const int dual_on = getenv("CAPE_GEMM_BF16X6_DUAL") ? atoi(getenv("CAPE_GEMM_BF16X6_DUAL")) : 1;
    if ((dual && !dual_on) || p.F < 64) return false;
    for (int i = 0; i < p.nsrc; ++i)
        if (p.s[i].C % GS_KC != 0 || p.s[i].C < GS_KC) return false;
    return true;
}

// 128 x 128 tiles (2 workgroups per CU: 61 KB LDS, ~200 VGPRs) when they still give every CU its two workgroups,
// 64 x 64 (5 per CU) otherwise -- the faster choice on every layer shape of the model in tools/ubench/gemm_bf16x3.hip.
inline void gs_tile(bool dual, int N, int Mo, int F, int &BM, int &BN) {
    if (dual) { BM = 128; BN = 64; return; }
    const long long big = (long long)N * ((Mo + 127) / 128) * ((F + 127) / 128);
    if (F >= 128 && big >= 384) { BM = 128; BN = 128; }
    else { BM = 64; BN = 64; }
}

template <typename AT>
inline void gs_launch_t(const GconvParams &p, bool dual, int BM, int layout, dim3 grid, hipStream_t st) {
    if (dual) {
        if (layout == 1) CAPE_LAUNCH((gemm_split_kernel<128, 64, true, true, AT>), grid, dim3(256), 0, st, p);
        else CAPE_LAUNCH((gemm_split_kernel<128, 64, false, true, AT>), grid, dim3(256), 0, st, p);
    } else if (BM == 128) {
        if (layout == 1) CAPE_LAUNCH((gemm_split_kernel<128, 128, true, false, AT>), grid, dim3(256), 0, st, p);
        else CAPE_LAUNCH((gemm_split_kernel<128, 128, false, false, AT>), grid, dim3(256), 0, st, p);
    } else {
        if (layout == 1) CAPE_LAUNCH((gemm_split_kernel<64, 64, true, false, AT>), grid, dim3(256), 0, st, p);
        else CAPE_LAUNCH((gemm_split_kernel<64, 64, false, false, AT>), grid, dim3(256), 0, st, p);
    }
}

inline void gs_launch(const GconvParams &p, bool dual, int BM, int layout, bool bf16, dim3 grid, hipStream_t st) {
    if (bf16) gs_launch_t<cape_bf16>(p, dual, BM, layout, grid, st);
    else gs_launch_t<float>(p, dual, BM, layout, grid, st);
}

}  // namespace
