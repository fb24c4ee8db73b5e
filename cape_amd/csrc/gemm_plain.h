// Software-pipelined GEMM for PLAIN sources (no CSR gather): the second pass of the two-pass
// Chebyshev convolution  y[n] = epilogue( sum_s X_s[n] @ B_s )  and its data gradient, where every
// X_s is a materialised [N, Mo, C_s] activation (reference lib/models.py:99-102 -- the trailing
// [N*M, Fin*K] x [Fin*K, Fout] contraction of chebyshev5).
//
// Same tiles, MFMA (v_mfma_f32_32x32x2_f32, exact fp32) and epilogue as gconv_fwd_kernel, but the
// global loads of chunk i+1 are issued into registers BEFORE the MFMAs of chunk i and land in LDS
// after them, so a workgroup hides its own HBM/L2 latency instead of relying on co-resident
// workgroups alone.  Weight tiles are accepted in two layouts: contraction-contiguous (BKC: [n][k], read
// back as one ds_read_b128 per fragment, like the activations) or output-contiguous ([k][n]).
#pragma once
#include "gconv_shared.h"

namespace {

constexpr int GP_KC = 32;            // contraction indices per chunk
constexpr int GP_LD = GP_KC + 4;     // LDS row pitch (floats) of the k-contiguous tiles

template <int BM, int BN, int WAVES_M, int WAVES_N, bool DUAL, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_plain_kernel(GconvParams p) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;
    constexpr int LDBN = BN + 4;
    constexpr int A_SZ = BM * GP_LD;
    constexpr int B_SZ = BKC ? BN * GP_LD : GP_KC * LDBN;
    constexpr int BUF_SZ = A_SZ + (DUAL ? 2 : 1) * B_SZ;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    __shared__ __attribute__((aligned(16))) float smem[BUF_SZ];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    const int q = tid & 7, t8 = tid >> 3;        // float4 column / row of the k-contiguous staging
    const int j4 = tid % (BN / 4), kr = tid / (BN / 4);   // [k][n] weight staging: float4 column / first k row
    constexpr int KSTEP = 256 / (BN / 4);        // k rows covered per pass of the [k][n] staging

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
    for (int si = 0; si < p.nsrc; ++si) total += (p.s[si].C + GP_KC - 1) / GP_KC;

    // clamped rows of this thread (the rows beyond Mo / columns beyond F are computed on valid
    // finite data of the last row / column and never stored by the epilogue)
    int rc[PA], fc[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) rc[i] = min(r0 + t8 + 32 * i, p.Mo - 1);
#pragma unroll
    for (int i = 0; i < PB; ++i) fc[i] = min(f0 + t8 + 32 * i, p.F - 1);
    const int fcol = (f0 + 4 * j4) < p.F ? (f0 + 4 * j4) : 0;     // [k][n] staging (F % 4 == 0)

    // ---- loader cursor: source l_si, channel offset l_c0; per-source state cached in registers
    int l_si = 0, l_c0 = 0, l_C = 0;
    const float *l_x = nullptr, *l_w = nullptr, *l_w2 = nullptr;
    int arow[PA], brow[PB], brow2[DUAL ? PB : 1];
    long long l_ws = 0, l_w2s = 0;
    auto open_source = [&]() {
        const SrcDev &S = p.s[l_si];
        l_C = S.C;
        l_x = S.x + (long long)n * S.xs;
        l_w = S.w;
        l_w2 = DUAL ? S.w2 : nullptr;
        l_ws = BKC ? S.wcs : S.wrs;
        l_w2s = BKC ? S.w2cs : S.w2rs;
#pragma unroll
        for (int i = 0; i < PA; ++i) arow[i] = rc[i] * S.ldx;
        if (BKC) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                brow[i] = fc[i] * (int)l_ws;
                if (DUAL) brow2[i] = fc[i] * (int)l_w2s;
            }
        }
    };

    float4 ra[PA], rb[PB], rb2[DUAL ? PB : 1];
    bool s_cok = false, s_has2 = false;       // of the chunk held in ra/rb
    unsigned s_kok = 0;
    int s_nv = 0;

    auto load_regs = [&]() {
        const int c = l_c0 + 4 * q;
        s_cok = c < l_C;
        s_nv = l_C - c;                    // valid lanes of this thread's activation float4 (row-padded sources: C % 4 != 0)
        const int cc = s_cok ? c : 0;
        s_has2 = DUAL && (l_w2 != nullptr);
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = *reinterpret_cast<const float4 *>(l_x + arow[i] + cc);
        if (BKC) {
#pragma unroll
            for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const float4 *>(l_w + brow[i] + cc);
            if (DUAL && s_has2) {
#pragma unroll
                for (int i = 0; i < PB; ++i) rb2[i] = *reinterpret_cast<const float4 *>(l_w2 + brow2[i] + cc);
            }
        } else {
            s_kok = 0;
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int k = l_c0 + kr + KSTEP * i;
                const bool ok = k < l_C;
                s_kok |= (ok ? 1u : 0u) << i;
                const int kc = ok ? k : 0;
                rb[i] = *reinterpret_cast<const float4 *>(l_w + (long long)kc * l_ws + fcol);
                if (DUAL && s_has2) rb2[i] = *reinterpret_cast<const float4 *>(l_w2 + (long long)kc * l_w2s + fcol);
            }
        }
        l_c0 += GP_KC;
        if (l_c0 >= l_C) {
            l_c0 = 0;
            ++l_si;
            if (l_si < p.nsrc) open_source();
        }
    };

    auto zsel = [](float4 v, bool ok) {
        float4 o;
        o.x = ok ? v.x : 0.f; o.y = ok ? v.y : 0.f; o.z = ok ? v.z : 0.f; o.w = ok ? v.w : 0.f;
        return o;
    };

    auto store_regs = [&]() {
        float *sA = smem;
        float *sB = sA + A_SZ;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            float4 o = ra[i];
            o.x = s_nv > 0 ? o.x : 0.f; o.y = s_nv > 1 ? o.y : 0.f; o.z = s_nv > 2 ? o.z : 0.f; o.w = s_nv > 3 ? o.w : 0.f;
            *reinterpret_cast<float4 *>(&sA[(t8 + 32 * i) * GP_LD + 4 * q]) = o;
        }
        if (BKC) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                *reinterpret_cast<float4 *>(&sB[(t8 + 32 * i) * GP_LD + 4 * q]) = zsel(rb[i], s_cok);
                if (DUAL && s_has2)
                    *reinterpret_cast<float4 *>(&sB[B_SZ + (t8 + 32 * i) * GP_LD + 4 * q]) = zsel(rb2[i], s_cok);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const bool ok = (s_kok >> i) & 1u;
                *reinterpret_cast<float4 *>(&sB[(kr + KSTEP * i) * LDBN + 4 * j4]) = zsel(rb[i], ok);
                if (DUAL && s_has2)
                    *reinterpret_cast<float4 *>(&sB[B_SZ + (kr + KSTEP * i) * LDBN + 4 * j4]) = zsel(rb2[i], ok);
            }
        }
    };

    // ---- multiply one staged chunk: fragments of k-block kb+1 are read while kb is on the MFMA pipe.
    // Contraction index permutation inside a block of 8: MFMA step u uses physical index
    // 8*kb + 4*lh + u for lane half lh (same for A and B), so both fragments are one 16-byte read.
    auto compute = [&](bool has2) {
        const float *sA = smem;
        const float *sB = sA + A_SZ;
        const float *sB2 = sB + B_SZ;
        const float *pa = sA + (wm * WTM + li) * GP_LD + 4 * lh;
        const float *pb = BKC ? sB + (wn * WTN + li) * GP_LD + 4 * lh : sB + (4 * lh) * LDBN + wn * WTN + li;
        const float *pb2 = BKC ? sB2 + (wn * WTN + li) * GP_LD + 4 * lh : sB2 + (4 * lh) * LDBN + wn * WTN + li;
        float4 af[2][TM], bf[2][TN], bf2[2][DUAL ? TN : 1];
        auto frags = [&](int s, int kb) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[s][a] = *reinterpret_cast<const float4 *>(pa + a * 32 * GP_LD + kb * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if (BKC) {
                    bf[s][b] = *reinterpret_cast<const float4 *>(pb + b * 32 * GP_LD + kb * 8);
                } else {
                    bf[s][b].x = pb[(kb * 8 + 0) * LDBN + b * 32];
                    bf[s][b].y = pb[(kb * 8 + 1) * LDBN + b * 32];
                    bf[s][b].z = pb[(kb * 8 + 2) * LDBN + b * 32];
                    bf[s][b].w = pb[(kb * 8 + 3) * LDBN + b * 32];
                }
            }
            if (DUAL && has2) {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    if (BKC) {
                        bf2[s][b] = *reinterpret_cast<const float4 *>(pb2 + b * 32 * GP_LD + kb * 8);
                    } else {
                        bf2[s][b].x = pb2[(kb * 8 + 0) * LDBN + b * 32];
                        bf2[s][b].y = pb2[(kb * 8 + 1) * LDBN + b * 32];
                        bf2[s][b].z = pb2[(kb * 8 + 2) * LDBN + b * 32];
                        bf2[s][b].w = pb2[(kb * 8 + 3) * LDBN + b * 32];
                    }
                }
            }
        };
        frags(0, 0);
#pragma unroll
        for (int kb = 0; kb < GP_KC / 8; ++kb) {
            const int s = kb & 1;
            if (kb + 1 < GP_KC / 8) frags(s ^ 1, kb + 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const float av = (u == 0) ? af[s][a].x : (u == 1) ? af[s][a].y : (u == 2) ? af[s][a].z : af[s][a].w;
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const float bv = (u == 0) ? bf[s][b].x : (u == 1) ? bf[s][b].y : (u == 2) ? bf[s][b].z : bf[s][b].w;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                    }
                }
            }
            if (DUAL && has2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const float av = (u == 0) ? af[s][a].x : (u == 1) ? af[s][a].y : (u == 2) ? af[s][a].z : af[s][a].w;
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            const float bv = (u == 0) ? bf2[s][b].x : (u == 1) ? bf2[s][b].y : (u == 2) ? bf2[s][b].z : bf2[s][b].w;
                            acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2[a][b], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };

    // One LDS buffer, two barriers per chunk: measured faster on MI355X than double-buffered LDS with one
    // barrier (half the LDS -> twice the resident workgroups, which is what hides the store/barrier bubbles), and
    // than an LDS-DMA variant (global_load_lds_dwordx4 into XOR-swizzled unpadded tiles, two buffers, one barrier,
    // issued through inline asm so that hipcc does not drain it before the multiply): bit-identical, 2-10 % slower.
    open_source();
    load_regs();
    store_regs();
    bool c_has2 = s_has2;
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const bool more = it + 1 < total;
        if (more) load_regs();          // chunk it+1: global -> registers, in flight during the MFMAs below
        compute(c_has2);
        __syncthreads();
        if (more) store_regs();
        c_has2 = s_has2;
        __syncthreads();
    }

    gconv_epilogue<BM, BN, WAVES_M, WAVES_N, DUAL>(p, acc, acc2, n, r0, f0, wm, wn, li, lh);
}

// =============================================================================================================
// Weight gradient of plain sources, software-pipelined:  dW_s[c, f] = sum_{n, r} X_s[n, r, c] * dz[n, r, f].
// Same decomposition as gconv_dw_kernel (one [CT x FT] tile x one (sample group, row range) split per
// workgroup, partial slabs reduced by dw_reduce_*), but the [32 x CT] activation chunk and the [32 x FT]
// gradient chunk of iteration i+1 are in flight in registers while iteration i is on the MFMA pipe.
// Requires: source plain, 16-byte aligned, C % 4 == 0; dz 16-byte aligned, F % 4 == 0.
// =============================================================================================================
// AT = storage type of the sources and of dz (float, or cape_bf16: rows read 8 bytes at a time and widened, the
// multiply stays on the fp32 MFMA -- exact for bf16 values).
template <int CT, int FT, int WAVES_M, int WAVES_N, typename AT = float>
__global__ __launch_bounds__(256, 2) void dw_plain_kernel(DwParams p) {
    constexpr int RK = 32;
    constexpr int LDA = CT + 4, LDB = FT + 4;
    constexpr int WTM = CT / WAVES_M, WTN = FT / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NA = RK * (CT / 4) / 256, NB = (RK * (FT / 4) + 255) / 256;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "4 waves, each at least one 32x32 MFMA tile");
    __shared__ __attribute__((aligned(16))) float smem[RK * LDA + RK * LDB];
    float *sA = smem;
    float *sB = smem + RK * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
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

    // per-thread staging coordinates (fixed for the whole kernel)
    int a_rl[NA], a_col[NA], b_rl[NB], b_col[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + i * 256;
        a_rl[i] = idx / (CT / 4);
        const int c = c0 + 4 * (idx % (CT / 4));
        a_col[i] = c < S.C ? c : 0;              // columns beyond C feed output rows that are never stored
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + i * 256;
        b_rl[i] = (idx / (FT / 4)) % RK;
        const int f = f0 + 4 * (idx % (FT / 4));
        b_col[i] = f < p.F ? f : 0;
    }
    const AT *dz0 = reinterpret_cast<const AT *>(((p.dz2_mask >> si) & 1u) ? p.dz2 : p.dz);

    const int chunks = (rb - ra + RK - 1) / RK;
    const int total = (n_end - n_begin) * chunks;
    int l_n = n_begin, l_r = ra;              // loader cursor
    float4 ra4[NA], rb4[NB];
    unsigned okA = 0, okB = 0;

    auto load_regs = [&]() {
        const AT *xb = reinterpret_cast<const AT *>(S.x) + (long long)l_n * S.xs;
        const AT *zb = dz0 + (long long)l_n * p.dzs;
        okA = okB = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = l_r + a_rl[i];
            okA |= (r < rb ? 1u : 0u) << i;
            ra4[i] = cape_ld4(xb + (long long)min(r, rb - 1) * S.ldx + a_col[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r = l_r + b_rl[i];
            okB |= (r < rb ? 1u : 0u) << i;
            rb4[i] = cape_ld4(zb + (long long)min(r, rb - 1) * p.lddz + b_col[i]);
        }
        l_r += RK;
        if (l_r >= rb) { l_r = ra; ++l_n; }
    };
    auto store_regs = [&]() {
        // rows beyond the split's range must contribute zero (they are clamped, finite data in both tiles)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = (okA >> i) & 1u;
            float4 o = ra4[i];
            o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
            *reinterpret_cast<float4 *>(&sA[a_rl[i] * LDA + 4 * ((tid + i * 256) % (CT / 4))]) = o;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const bool ok = (okB >> i) & 1u;
            float4 o = rb4[i];
            o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
            if (tid + i * 256 < RK * (FT / 4)) *reinterpret_cast<float4 *>(&sB[b_rl[i] * LDB + 4 * ((tid + i * 256) % (FT / 4))]) = o;
        }
    };
    auto compute = [&]() {
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

// The packed variant (dw_packed_kernel) lays several narrow sources side by side on one tile's channel axis
// (virtual axis, DwParams::vstart): at 6890 vertices the layers are 3 x 32 channels wide, and three half-empty
// tiles that each re-read dz become one.  It carries per-slot source pointers; layers wide enough to fill
// their own tiles use dw_plain_kernel (one source per tile, scalar base addresses -- measured 14 % faster there).
template <int CT, int FT, int WAVES_M, int WAVES_N, typename AT = float>
__global__ __launch_bounds__(256, 2) void dw_packed_kernel(DwParams p) {
    constexpr int RK = 32;
    constexpr int LDA = CT + 4, LDB = FT + 4;
    constexpr int WTM = CT / WAVES_M, WTN = FT / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NA = RK * (CT / 4) / 256, NB = (RK * (FT / 4) + 255) / 256;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "4 waves, each at least one 32x32 MFMA tile");
    __shared__ __attribute__((aligned(16))) float smem[RK * LDA + RK * LDB];
    float *sA = smem;
    float *sB = smem + RK * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    const int V = p.vstart[p.nsrc];
    const int ntiles = ((V + CT - 1) / CT) * p.ftiles;
    int tile, split;                                  // split = group * rsplit + rs
    if (!cape_map_dw_block(blockIdx.x, ntiles, p.ngroups * p.rsplit, tile, split)) return;
    const int grp = split / p.rsplit;
    const int rs = split % p.rsplit;
    const int n_begin = grp * p.samples_per_group;
    const int n_end = min(p.N, n_begin + p.samples_per_group);
    const int v0 = (tile / p.ftiles) * CT;
    const int f0 = (tile % p.ftiles) * FT;
    const int ra = rs * p.rows_per_split;
    const int rb = min(p.Mo, ra + p.rows_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    // per-thread staging coordinates (fixed for the whole kernel): slot i of the A chunk reads 4 channels of ONE source
    const AT *a_ptr[NA];
    long long a_xs[NA];
    int a_ld[NA], a_rl[NA], b_rl[NB], b_col[NB];
    int first_src = 0;
    while (first_src + 1 < p.nsrc && v0 >= p.vstart[first_src + 1]) ++first_src;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + i * 256;
        a_rl[i] = idx / (CT / 4);
        const int vc = v0 + 4 * (idx % (CT / 4));
        int si = 0;
        while (si + 1 < p.nsrc && vc >= p.vstart[si + 1]) ++si;
        const bool ok = vc - p.vstart[si] < p.s[si].C;      // padding columns read column 0: their outputs are never stored
        a_ptr[i] = reinterpret_cast<const AT *>(p.s[si].x) + (ok ? vc - p.vstart[si] : 0);
        a_xs[i] = p.s[si].xs;
        a_ld[i] = p.s[si].ldx;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + i * 256;
        b_rl[i] = (idx / (FT / 4)) % RK;
        const int f = f0 + 4 * (idx % (FT / 4));
        b_col[i] = f < p.F ? f : 0;
    }
    const AT *dz0 = reinterpret_cast<const AT *>(((p.dz2_mask >> first_src) & 1u) ? p.dz2 : p.dz);

    const int chunks = (rb - ra + RK - 1) / RK;
    const int total = (n_end - n_begin) * chunks;
    int l_n = n_begin, l_r = ra;              // loader cursor
    float4 ra4[NA], rb4[NB];
    unsigned okA = 0, okB = 0;

    auto load_regs = [&]() {
        const AT *zb = dz0 + (long long)l_n * p.dzs;
        okA = okB = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = l_r + a_rl[i];
            okA |= (r < rb ? 1u : 0u) << i;
            ra4[i] = cape_ld4(a_ptr[i] + (long long)l_n * a_xs[i] + (long long)min(r, rb - 1) * a_ld[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r = l_r + b_rl[i];
            okB |= (r < rb ? 1u : 0u) << i;
            rb4[i] = cape_ld4(zb + (long long)min(r, rb - 1) * p.lddz + b_col[i]);
        }
        l_r += RK;
        if (l_r >= rb) { l_r = ra; ++l_n; }
    };
    auto store_regs = [&]() {
        // rows beyond the split's range must contribute zero (they are clamped, finite data in both tiles)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = (okA >> i) & 1u;
            float4 o = ra4[i];
            o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
            *reinterpret_cast<float4 *>(&sA[a_rl[i] * LDA + 4 * ((tid + i * 256) % (CT / 4))]) = o;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const bool ok = (okB >> i) & 1u;
            float4 o = rb4[i];
            o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
            if (tid + i * 256 < RK * (FT / 4)) *reinterpret_cast<float4 *>(&sB[b_rl[i] * LDB + 4 * ((tid + i * 256) % (FT / 4))]) = o;
        }
    };
    auto compute = [&]() {
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

    // partial slab layout: [split][part_off[source] + c*F + f]
    float *out = p.ws + (long long)split * p.slab;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = f0 + wn * WTN + b * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int vc = v0 + wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                int si = 0;
                while (si + 1 < p.nsrc && vc >= p.vstart[si + 1]) ++si;
                const int c = vc - p.vstart[si];
                if (vc < V && c < p.s[si].C && f < p.F) out[p.part_off[si] + (long long)c * p.F + f] = acc[a][b][g];
            }
        }
}

// Layout class of one launch's weight operands: 1 = contraction-contiguous, 0 = output-contiguous,
// -1 = not eligible for the plain kernel.
inline int gp_weight_layout(const GconvParams &p, bool dual) {
    bool kc = true, nc = true;
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    for (int i = 0; i < p.nsrc; ++i) {
        const SrcDev &S = p.s[i];
        if (S.rp || !S.vec || (((S.C + 3) & ~3) > S.ldx) || !al16(S.w)) return -1;
        // contraction-contiguous weight rows are read as float4 along the contraction: whole float4s only
        kc = kc && (S.C & 3) == 0 && S.wrs == 1 && (S.wcs & 3) == 0 && S.wcs < (1LL << 20);
        nc = nc && S.wcs == 1 && (S.wrs & 3) == 0 && (p.F & 3) == 0;
        if (dual && S.w2) {
            if (!al16(S.w2)) return -1;
            kc = kc && S.w2rs == 1 && (S.w2cs & 3) == 0 && S.w2cs < (1LL << 20);
            nc = nc && S.w2cs == 1 && (S.w2rs & 3) == 0;
        }
    }
    return kc ? 1 : (nc ? 0 : -1);
}

template <int BM, int BN, int WM, int WN, bool DUAL>
inline void gp_launch_tile(const GconvParams &p, int layout, dim3 grid, hipStream_t st) {
    if (layout == 1) CAPE_LAUNCH((gemm_plain_kernel<BM, BN, WM, WN, DUAL, true>), grid, dim3(256), 0, st, p);
    else CAPE_LAUNCH((gemm_plain_kernel<BM, BN, WM, WN, DUAL, false>), grid, dim3(256), 0, st, p);
}

// Tile shape of the pipelined kernel for an [N x Mo x F] output.  Small tiles on purpose: at batch 16 the coarse
// mesh levels are only a few hundred 128x128 tiles for 256 CUs, and the makespan is set by the CU that got one
// tile more than its neighbours; 64x64 tiles (7 resident workgroups per CU at 68 VGPRs / 18 KB LDS) cut that
// quantisation loss from ~15 % to ~4 % and measured 5-10 % faster on every non-DUAL layer shape of the model.
inline void gp_tile(bool dual, int F, int &BM, int &BN) {
    if (F <= 32) { BM = 128; BN = 32; }
    else if (dual) { BM = 128; BN = 64; }
    else { BM = 64; BN = 64; }
}

inline void gp_launch(const GconvParams &p, bool dual, int BM, int BN, int layout, dim3 grid, hipStream_t st) {
    if (!dual) {
        if (BN == 32) gp_launch_tile<128, 32, 4, 1, false>(p, layout, grid, st);
        else gp_launch_tile<64, 64, 2, 2, false>(p, layout, grid, st);
    } else {
        if (BN == 32) gp_launch_tile<128, 32, 4, 1, true>(p, layout, grid, st);
        else gp_launch_tile<128, 64, 4, 1, true>(p, layout, grid, st);
    }
}

}  // namespace
