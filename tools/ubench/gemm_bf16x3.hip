// Experiment: fp32 GEMM on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the f32-MFMA rate) by splitting every
// fp32 operand EXACTLY into three bf16 pieces (x = hi + mid + lo, 8 + 8 + 8 significand bits, truncation split) while
// it is staged into LDS, and accumulating the cross products in the MFMA's fp32 accumulator:
//   NT = 6:  hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi   (dropped terms <= 2^-24 relative: fp32-level)
//   NT = 9:  all nine products (every product of two fp32 inputs exact before accumulation)
//   NT = 3:  hi*hi + hi*mid + mid*hi                              (~2^-16: tf32x3-like, for reference)
// Shape of the work: the trailing contraction of chebyshev5 (reference lib/models.py:99-102) in its data-gradient
// layout -- C[n] = A[n] (Mo x K) * B^T with B stored [F][K] (contraction-contiguous), per sample n.
// Standalone (no library):  hipcc --offload-arch=gfx950 -O3 gemm_bf16x3.hip -o gemm_bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;              // contraction indices per staged chunk (two k16 MFMA steps)
constexpr int PITCH = 80;           // bytes per LDS row of one piece plane: 32 bf16 + 16 B pad (conflict-free b128 reads)

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bitsf(unsigned v) { return __builtin_bit_cast(float, v); }

// two fp32 -> their three bf16 pieces, packed pairwise (element 0 in the low half)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    const unsigned h0 = fbits(x0) & 0xFFFF0000u, h1 = fbits(x1) & 0xFFFF0000u;
    const float r0 = x0 - bitsf(h0), r1 = x1 - bitsf(h1);                 // exact
    const unsigned m0 = fbits(r0) & 0xFFFF0000u, m1 = fbits(r1) & 0xFFFF0000u;
    const float s0 = r0 - bitsf(m0), s1 = r1 - bitsf(m1);                 // exact, <= 8 significant bits left
    hi = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    mid = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    lo = __builtin_amdgcn_perm(fbits(s1), fbits(s0), 0x07060302u);
}

__device__ __forceinline__ void split8(const float4 &u, const float4 &v, uint4 &hi, uint4 &mid, uint4 &lo) {
    split2(u.x, u.y, hi.x, mid.x, lo.x);
    split2(u.z, u.w, hi.y, mid.y, lo.y);
    split2(v.x, v.y, hi.z, mid.z, lo.z);
    split2(v.z, v.w, hi.w, mid.w, lo.w);
}

// piece indices (0 = hi, 1 = mid, 2 = lo) of the A and B factor of term t, smallest products first
__host__ __device__ constexpr int term_count(int nt) { return nt; }
__host__ __device__ constexpr int term_pa(int nt, int t) {
    return nt == 3 ? (t == 0 ? 0 : t == 1 ? 1 : 0)
         : nt == 6 ? (t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0)
                   : (t == 0 ? 2 : t == 1 ? 1 : t == 2 ? 2 : t == 3 ? 0 : t == 4 ? 2 : t == 5 ? 1 : t == 6 ? 0 : t == 7 ? 1 : 0);
}
__host__ __device__ constexpr int term_pb(int nt, int t) {
    return nt == 3 ? (t == 0 ? 1 : t == 1 ? 0 : 0)
         : nt == 6 ? (t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : t == 4 ? 0 : 0)
                   : (t == 0 ? 2 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 2 : t == 4 ? 0 : t == 5 ? 1 : t == 6 ? 1 : t == 7 ? 0 : 0);
}

// block -> (sample, tile): all tiles of a sample on one XCD (blocks are dispatched round robin over the 8 XCDs)
__device__ __forceinline__ void map_block(int b, int N, int T, int &n, int &t) {
    if ((N & 7) == 0) {
        const int per = N >> 3, local = b >> 3;
        n = (b & 7) * per + local / T;
        t = local % T;
    } else {
        n = b / T;
        t = b % T;
    }
}

// WG tile BM x BN, 4 waves as 2 x 2, wave tile (BM/2) x (BN/2)
// PF = chunks of global loads in flight during a multiply (1 or 2); DBG: 1 = no global loads inside the loop, 2 = no split
// arithmetic (raw bits stored; wrong results), 3 = both -- timing decomposition only
template <int BM, int BN, int NT, int MINB, int PF, int DBG>
__global__ __launch_bounds__(256, MINB) void gemm_bf16x3_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                                float *__restrict__ C, int N, int Mo, int K, int F,
                                                                int row_tiles, int col_tiles) {
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int PA = BM / 64, PB = BN / 64;            // staging passes (64 rows x 4 eight-float groups per pass)
    constexpr int APLANE = BM * PITCH, BPLANE = BN * PITCH;
    constexpr bool NEED_LO = NT != 3;
    constexpr int NP = NEED_LO ? 3 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NP * (APLANE + BPLANE)];
    unsigned char *sA = smem, *sB = smem + NP * APLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int q = tid & 3, r = tid >> 2;

    int n, t;
    map_block(blockIdx.x, N, row_tiles * col_tiles, n, t);
    const int r0 = (t / col_tiles) * BM, f0 = (t % col_tiles) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    const float *ap[PA], *bp[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) ap[i] = A + ((long long)n * Mo + min(r0 + r + 64 * i, Mo - 1)) * K + 8 * q;
#pragma unroll
    for (int i = 0; i < PB; ++i) bp[i] = B + (long long)min(f0 + r + 64 * i, F - 1) * K + 8 * q;

    float4 ra[PF][PA][2], rb[PF][PB][2];
    auto load_regs = [&](auto slot, int k0) {
        constexpr int S = decltype(slot)::value;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            ra[S][i][0] = *reinterpret_cast<const float4 *>(ap[i] + k0);
            ra[S][i][1] = *reinterpret_cast<const float4 *>(ap[i] + k0 + 4);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            rb[S][i][0] = *reinterpret_cast<const float4 *>(bp[i] + k0);
            rb[S][i][1] = *reinterpret_cast<const float4 *>(bp[i] + k0 + 4);
        }
    };
    auto split = [&](const float4 &u, const float4 &v, uint4 &hi, uint4 &mid, uint4 &lo) {
        if (DBG & 2) {
            hi = uint4{fbits(u.x), fbits(u.y), fbits(u.z), fbits(u.w)};
            mid = uint4{fbits(v.x), fbits(v.y), fbits(v.z), fbits(v.w)};
            lo = hi;
        } else {
            split8(u, v, hi, mid, lo);
        }
    };
    auto store_regs = [&](auto slot) {
        constexpr int S = decltype(slot)::value;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            uint4 hi, mid, lo;
            split(ra[S][i][0], ra[S][i][1], hi, mid, lo);
            unsigned char *d = sA + (r + 64 * i) * PITCH + 16 * q;
            *reinterpret_cast<uint4 *>(d) = hi;
            *reinterpret_cast<uint4 *>(d + APLANE) = mid;
            if (NEED_LO) *reinterpret_cast<uint4 *>(d + 2 * APLANE) = lo;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            uint4 hi, mid, lo;
            split(rb[S][i][0], rb[S][i][1], hi, mid, lo);
            unsigned char *d = sB + (r + 64 * i) * PITCH + 16 * q;
            *reinterpret_cast<uint4 *>(d) = hi;
            *reinterpret_cast<uint4 *>(d + BPLANE) = mid;
            if (NEED_LO) *reinterpret_cast<uint4 *>(d + 2 * BPLANE) = lo;
        }
    };
    auto compute = [&]() {
        const unsigned char *pa = sA + (wm * WTM + li) * PITCH + 16 * lh;
        const unsigned char *pb = sB + (wn * WTN + li) * PITCH + 16 * lh;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    af[a][p] = *reinterpret_cast<const bf16x8 *>(pa + p * APLANE + a * 32 * PITCH + 32 * ks);
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bf[b][p] = *reinterpret_cast<const bf16x8 *>(pb + p * BPLANE + b * 32 * PITCH + 32 * ks);
#pragma unroll
            for (int term = 0; term < NT; ++term)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][term_pa(NT, term)], bf[b][term_pb(NT, term)],
                                                                            acc[a][b], 0, 0, 0);
        }
    };

    auto sync = [&]() { if (!(DBG & 8)) __syncthreads(); };
    const int total = K / KC;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, PF - 1>;
    load_regs(S0{}, 0);
    store_regs(S0{});
    __syncthreads();
    if (PF == 1) {
        for (int it = 0; it < total; ++it) {
            const bool more = it + 1 < total;
            if (more && !(DBG & 1)) load_regs(S0{}, (it + 1) * KC);
            compute();
            sync();
            if (more) store_regs(S0{});
            sync();
        }
    } else {
        // LDS holds chunk `it`; slot A carries chunk it+1, slot B chunk it+2 (loads of two chunks in flight)
        if (1 < total) load_regs(S0{}, KC);
        for (int it = 0; it < total; it += 2) {
            if (it + 2 < total && !(DBG & 1)) load_regs(S1{}, (it + 2) * KC);
            compute();
            sync();
            if (it + 1 < total) store_regs(S0{});
            sync();
            if (it + 1 < total) {
                if (it + 3 < total && !(DBG & 1)) load_regs(S0{}, (it + 3) * KC);
                compute();
                sync();
                if (it + 2 < total) store_regs(S1{});
                sync();
            }
        }
    }

    // C/D layout of the 32x32 MFMA tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *cn = C + (long long)n * Mo * F;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = f0 + wn * WTN + b * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = r0 + wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (row < Mo && col < F) cn[(long long)row * F + col] = acc[a][b][g];
            }
        }
}

// sustained v_mfma_f32_32x32x16_bf16 rate, no memory traffic (NACC independent accumulators per wave)
template <int NACC>
__global__ __launch_bounds__(256) void mfma_bf16_loop(float *out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) acc[a][g] = 0.f;
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(1e-3f * (1 + (threadIdx.x + j) % 7)); bv[j] = (__bf16)(1e-3f * (2 + (threadIdx.x + j) % 5)); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[a], 0, 0, 0);
    }
    float sum = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) sum += acc[a][g];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

static void peak(int blocks, int iters, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_bf16_loop<4><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_bf16_loop<4><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 4 * 32768.0;
    printf("bf16 MFMA peak: %d blocks (%.0f waves/SIMD) x %d iters: %.3f ms  %.0f TFLOP/s (/6 = %.0f fp32-equivalent)\n", blocks,
           blocks / 256.0, iters, ms, fl / ms / 1e9, fl / ms / 1e9 / 6);
}

// ---------------------------------------------------------------------------------------------------------------
// Second-generation variants (six products, one chunk of prefetch), parameterised for the next measurements:
//   WM x WN waves per workgroup (64 * WM * WN threads), wave tile (BM / WM) x (BN / WN);
//   RN   1: round-to-nearest pieces through v_cvt_pk_bf16_f32 (9 VALU ops per pair instead of 11, dropped terms 2^-23);
//   SWZ  1: unpadded 64-byte LDS rows with the 16-byte segment index XOR-ed by (row >> 2) & 3 (conflict-free for the
//           ds_read_b128 lane groups {0-3,12-15,20-27}...) -- 48 KB instead of 60 KB per 128 x 128 tile, i.e. three
//           workgroups per CU if the registers allow (MINB = 3).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <bool RN>
__device__ __forceinline__ void split2v(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    if constexpr (RN) {
        const f32x2_t x = {x0, x1};
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
        const f32x2_t r = {x0 - bitsf(hi << 16), x1 - bitsf(hi & 0xFFFF0000u)};
        mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
        const f32x2_t s = {r[0] - bitsf(mid << 16), r[1] - bitsf(mid & 0xFFFF0000u)};
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf16x2_t));
    } else {
        split2(x0, x1, hi, mid, lo);
    }
}

// operand already split by its producer: three bf16 planes [3][rows][K] in HBM (1.5x the bytes, no split in the GEMM)
template <bool RN>
__global__ void presplit_kernel(const float *__restrict__ x, unsigned *__restrict__ planes, long long n_pairs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    unsigned hi, mid, lo;
    split2v<RN>(x[2 * i], x[2 * i + 1], hi, mid, lo);
    planes[i] = hi;
    planes[n_pairs + i] = mid;
    planes[2 * n_pairs + i] = lo;
}

// PRE_A / PRE_B: that operand is read as pre-split planes (A3 / B3) instead of fp32
template <int BM, int BN, int WM, int WN, int MINB, bool RN, bool SWZ, bool PRE_A = false, bool PRE_B = false>
__global__ __launch_bounds__(64 * WM * WN, MINB) void gemm_v2_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                                      float *__restrict__ C, int N, int Mo, int K, int F,
                                                                      int row_tiles, int col_tiles,
                                                                      const unsigned short *__restrict__ A3 = nullptr,
                                                                      const unsigned short *__restrict__ B3 = nullptr) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int RPP = NTH / 4;                          // rows per staging pass (4 eight-float groups per row)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int LP = SWZ ? 64 : PITCH;                  // LDS row pitch in bytes
    constexpr int APLANE = BM * LP, BPLANE = BN * LP;
    static_assert(TM >= 1 && TN >= 1 && PA >= 1 && PB >= 1, "tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * (APLANE + BPLANE)];
    unsigned char *sA = smem, *sB = smem + 3 * APLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int q = tid & 3, r = tid >> 2;
    auto seg = [](int row, int s) { return SWZ ? (s ^ ((row >> 2) & 3)) : s; };

    int n, t;
    map_block(blockIdx.x, N, row_tiles * col_tiles, n, t);
    const int r0 = (t / col_tiles) * BM, f0 = (t % col_tiles) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    const float *ap[PA], *bp[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) ap[i] = A + ((long long)n * Mo + min(r0 + r + RPP * i, Mo - 1)) * K + 8 * q;
#pragma unroll
    for (int i = 0; i < PB; ++i) bp[i] = B + (long long)min(f0 + r + RPP * i, F - 1) * K + 8 * q;

    float4 ra[PA][2], rb[PB][2];
    uint4 pa3[PRE_A ? PA : 1][3], pb3[PRE_B ? PB : 1][3];            // pre-split operands: 8 bf16 per plane per pass
    const long long a_plane = (long long)N * Mo * K, b_plane = (long long)F * K;   // elements per plane
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if constexpr (PRE_A) {
                const long long e = (ap[i] - A) + k0;
#pragma unroll
                for (int p = 0; p < 3; ++p) pa3[i][p] = *reinterpret_cast<const uint4 *>(A3 + p * a_plane + e);
            } else {
                ra[i][0] = *reinterpret_cast<const float4 *>(ap[i] + k0);
                ra[i][1] = *reinterpret_cast<const float4 *>(ap[i] + k0 + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if constexpr (PRE_B) {
                const long long e = (bp[i] - B) + k0;
#pragma unroll
                for (int p = 0; p < 3; ++p) pb3[i][p] = *reinterpret_cast<const uint4 *>(B3 + p * b_plane + e);
            } else {
                rb[i][0] = *reinterpret_cast<const float4 *>(bp[i] + k0);
                rb[i][1] = *reinterpret_cast<const float4 *>(bp[i] + k0 + 4);
            }
        }
    };
    auto store3 = [&](unsigned char *base, int plane, int row, uint4 v0, uint4 v1, uint4 v2) {
        unsigned char *d = base + row * LP + 16 * seg(row, q);
        *reinterpret_cast<uint4 *>(d) = v0;
        *reinterpret_cast<uint4 *>(d + plane) = v1;
        *reinterpret_cast<uint4 *>(d + 2 * plane) = v2;
    };
    auto store8 = [&](unsigned char *base, int plane, int row, const float4 &u, const float4 &v) {
        uint4 hi, mid, lo;
        split2v<RN>(u.x, u.y, hi.x, mid.x, lo.x);
        split2v<RN>(u.z, u.w, hi.y, mid.y, lo.y);
        split2v<RN>(v.x, v.y, hi.z, mid.z, lo.z);
        split2v<RN>(v.z, v.w, hi.w, mid.w, lo.w);
        unsigned char *d = base + row * LP + 16 * seg(row, q);
        *reinterpret_cast<uint4 *>(d) = hi;
        *reinterpret_cast<uint4 *>(d + plane) = mid;
        *reinterpret_cast<uint4 *>(d + 2 * plane) = lo;
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if constexpr (PRE_A) store3(sA, APLANE, r + RPP * i, pa3[i][0], pa3[i][1], pa3[i][2]);
            else store8(sA, APLANE, r + RPP * i, ra[i][0], ra[i][1]);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if constexpr (PRE_B) store3(sB, BPLANE, r + RPP * i, pb3[i][0], pb3[i][1], pb3[i][2]);
            else store8(sB, BPLANE, r + RPP * i, rb[i][0], rb[i][1]);
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row = wm * WTM + a * 32 + li;
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[a][p] = *reinterpret_cast<const bf16x8 *>(sA + p * APLANE + row * LP + 16 * seg(row, lh + 2 * ks));
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int row = wn * WTN + b * 32 + li;
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bf[b][p] = *reinterpret_cast<const bf16x8 *>(sB + p * BPLANE + row * LP + 16 * seg(row, lh + 2 * ks));
            }
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][term_pa(6, term)], bf[b][term_pb(6, term)],
                                                                            acc[a][b], 0, 0, 0);
        }
    };

    const int total = K / KC;
    load_regs(0);
    store_regs();
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const bool more = it + 1 < total;
        if (more) load_regs((it + 1) * KC);
        compute();
        __syncthreads();
        if (more) store_regs();
        __syncthreads();
    }

    float *cn = C + (long long)n * Mo * F;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = f0 + wn * WTN + b * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = r0 + wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (row < Mo && col < F) cn[(long long)row * F + col] = acc[a][b][g];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Third generation (round 2): double-buffered LDS (2 x 60 KB -> one workgroup per CU), ONE barrier per chunk, and the
// operand split of chunk it+1 placed between the fragment reads and the MFMAs of chunk it in program order, so that the
// wave's own VALU / LDS-write work can run in the shadow of its MFMAs.  SG: 0 = leave the interleaving to the compiler,
// 1 = sched_group_barrier pattern (1 MFMA : 4 VALU, a DS write every 4th group), 2 = same with 6 VALU per MFMA.
// ---------------------------------------------------------------------------------------------------------------
template <int SG>
__global__ __launch_bounds__(256, 1) void gemm_v3_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                        float *__restrict__ C, int N, int Mo, int K, int F,
                                                        int row_tiles, int col_tiles) {
    constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2, PA = 2, PB = 2, LP = PITCH;
    constexpr int APLANE = BM * LP, BPLANE = BN * LP, BUF = 3 * (APLANE + BPLANE);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int q = tid & 3, r = tid >> 2;
    int n, t;
    map_block(blockIdx.x, N, row_tiles * col_tiles, n, t);
    const int r0 = (t / col_tiles) * BM, f0 = (t % col_tiles) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    const float *ap[PA], *bp[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) ap[i] = A + ((long long)n * Mo + min(r0 + r + 64 * i, Mo - 1)) * K + 8 * q;
#pragma unroll
    for (int i = 0; i < PB; ++i) bp[i] = B + (long long)min(f0 + r + 64 * i, F - 1) * K + 8 * q;

    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v ra[PA][2], rb[PB][2];
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            ra[i][0] = *reinterpret_cast<const f32x4v *>(ap[i] + k0);
            ra[i][1] = *reinterpret_cast<const f32x4v *>(ap[i] + k0 + 4);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            rb[i][0] = *reinterpret_cast<const f32x4v *>(bp[i] + k0);
            rb[i][1] = *reinterpret_cast<const f32x4v *>(bp[i] + k0 + 4);
        }
    };
    auto store8 = [&](unsigned char *base, int plane, int row, const f32x4v &u, const f32x4v &v) {
        u32x4 hi, mid, lo;
        unsigned h, m, l;
        split2(u[0], u[1], h, m, l); hi[0] = h; mid[0] = m; lo[0] = l;
        split2(u[2], u[3], h, m, l); hi[1] = h; mid[1] = m; lo[1] = l;
        split2(v[0], v[1], h, m, l); hi[2] = h; mid[2] = m; lo[2] = l;
        split2(v[2], v[3], h, m, l); hi[3] = h; mid[3] = m; lo[3] = l;
        unsigned char *d = base + row * LP + 16 * q;
        *reinterpret_cast<u32x4 *>(d) = hi;
        *reinterpret_cast<u32x4 *>(d + plane) = mid;
        *reinterpret_cast<u32x4 *>(d + 2 * plane) = lo;
    };
    // half 0: the A rows, half 1: the B rows of the chunk held in the staging registers
    auto store_half = [&](unsigned char *buf, int half) {
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < PA; ++i) store8(buf, APLANE, r + 64 * i, ra[i][0], ra[i][1]);
        } else {
#pragma unroll
            for (int i = 0; i < PB; ++i) store8(buf + 3 * APLANE, BPLANE, r + 64 * i, rb[i][0], rb[i][1]);
        }
    };
    // two fragment register sets: the reads of k-step s+1 are in flight during the MFMAs of k-step s (SG >= 10: the
    // fragment-prefetch schedule; below 10: reads issued right before their MFMAs, as in the second generation)
    bf16x8 afr[2][TM][3], bfr[2][TN][3];
    auto frags_to = [&](int set, const unsigned char *buf, int ks) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                afr[set][a][p] = *reinterpret_cast<const bf16x8 *>(buf + p * APLANE + (wm * WTM + a * 32 + li) * LP + 16 * (lh + 2 * ks));
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bfr[set][b][p] = *reinterpret_cast<const bf16x8 *>(buf + 3 * APLANE + p * BPLANE + (wn * WTN + b * 32 + li) * LP + 16 * (lh + 2 * ks));
    };
    auto mfmas_of = [&](int set) {
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[set][a][term_pa(6, term)], bfr[set][b][term_pb(6, term)], acc[a][b], 0, 0, 0);
    };
    auto frags = [&](const unsigned char *buf, int ks) { frags_to(0, buf, ks); };
    auto mfmas = [&]() { mfmas_of(0); };
    auto pattern = [&]() {
        if constexpr (SG > 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, SG == 1 ? 4 : 6, 0);      // VALU of the split
                if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // a DS write now and then
            }
        }
    };

    const int total = K / KC;
    load_regs(0);
    store_half(smem3, 0);
    store_half(smem3, 1);
    if (total > 1) load_regs(KC);
    __syncthreads();
    if constexpr (SG >= 10) {
        frags_to(0, smem3, 0);
        for (int it = 0; it + 1 < total; ++it) {
            unsigned char *cur = smem3 + (it & 1) * BUF, *nxt = smem3 + ((it + 1) & 1) * BUF;
            frags_to(1, cur, 1);
            store_half(nxt, 0);
            mfmas_of(0);
            store_half(nxt, 1);
            load_regs(min(it + 2, total - 1) * KC);
            mfmas_of(1);
            __syncthreads();
            frags_to(0, nxt, 0);
        }
        {
            unsigned char *cur = smem3 + ((total - 1) & 1) * BUF;
            frags_to(1, cur, 1);
            mfmas_of(0);
            mfmas_of(1);
        }
    } else {
    // branch-free steady state (stores unconditional, the look-ahead load clamped to the last chunk) so that the split, the
    // LDS writes and the MFMAs of one iteration sit in ONE scheduling region; the last chunk is peeled
    for (int it = 0; it + 1 < total; ++it) {
        unsigned char *cur = smem3 + (it & 1) * BUF, *nxt = smem3 + ((it + 1) & 1) * BUF;
        frags(cur, 0);
        store_half(nxt, 0);
        mfmas();
        pattern();
        frags(cur, 1);
        store_half(nxt, 1);
        load_regs(min(it + 2, total - 1) * KC);
        mfmas();
        pattern();
        __syncthreads();
    }
    {
        unsigned char *cur = smem3 + ((total - 1) & 1) * BUF;
        frags(cur, 0);
        mfmas();
        frags(cur, 1);
        mfmas();
    }
    }

    float *cn = C + (long long)n * Mo * F;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = f0 + wn * WTN + b * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = r0 + wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (row < Mo && col < F) cn[(long long)row * F + col] = acc[a][b][g];
            }
        }
}

struct Shape;
template <int SG>
static double run_v3(const Shape &s, const float *A, const float *B, float *C, int iters);

// ---------------------------------------------------------------------------------------------------------------
// bf16-STORAGE contraction (BASELINE configs[4]: activations and weights kept in bf16, fp32 accumulate): one MFMA product
// per multiply-add, operands copied global -> LDS as they are (64 contraction indices = 128 bytes per row and chunk,
// swizzled 16-byte segments), fp32 or bf16 output.  A projection of what the bf16 path of the library would reach.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int MINB, bool OUT16>
__global__ __launch_bounds__(64 * WM * WN, MINB) void gemm_bf16_kernel(const unsigned short *__restrict__ A,
                                                                        const unsigned short *__restrict__ B, void *__restrict__ Cv,
                                                                        int N, int Mo, int K, int F, int row_tiles, int col_tiles) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int KB = 64;                                // contraction indices per chunk
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int RPP = NTH / 8;                          // rows per staging pass (8 sixteen-byte segments per row)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int LP = 128;                               // bytes per LDS row, unpadded: segment s of row r at s ^ (r & 7)
    static_assert(TM >= 1 && TN >= 1 && PA >= 1 && PB >= 1, "tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * LP];
    unsigned char *sA = smem, *sB = smem + BM * LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int q = tid & 7, r = tid >> 3;
    int n, t;
    map_block(blockIdx.x, N, row_tiles * col_tiles, n, t);
    const int r0 = (t / col_tiles) * BM, f0 = (t % col_tiles) * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
    long long ao[PA], bo[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) ao[i] = ((long long)n * Mo + min(r0 + r + RPP * i, Mo - 1)) * K + 8 * q;
#pragma unroll
    for (int i = 0; i < PB; ++i) bo[i] = (long long)min(f0 + r + RPP * i, F - 1) * K + 8 * q;
    u32x4 ra[PA], rb[PB];
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = *reinterpret_cast<const u32x4 *>(A + ao[i] + k0);
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const u32x4 *>(B + bo[i] + k0);
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = r + RPP * i;
            *reinterpret_cast<u32x4 *>(sA + row * LP + 16 * (q ^ (row & 7))) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = r + RPP * i;
            *reinterpret_cast<u32x4 *>(sB + row * LP + 16 * (q ^ (row & 7))) = rb[i];
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < KB / 16; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row = wm * WTM + a * 32 + li;
                af[a] = *reinterpret_cast<const bf16x8 *>(sA + row * LP + 16 * ((lh + 2 * ks) ^ (row & 7)));
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int row = wn * WTN + b * 32 + li;
                bf[b] = *reinterpret_cast<const bf16x8 *>(sB + row * LP + 16 * ((lh + 2 * ks) ^ (row & 7)));
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    };
    const int total = K / KB;
    load_regs(0);
    store_regs();
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const bool more = it + 1 < total;
        if (more) load_regs((it + 1) * KB);
        compute();
        __syncthreads();
        if (more) store_regs();
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = f0 + wn * WTN + b * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = r0 + wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (row < Mo && col < F) {
                    const long long o = ((long long)n * Mo + row) * F + col;
                    if constexpr (OUT16) reinterpret_cast<__bf16 *>(Cv)[o] = (__bf16)acc[a][b][g];
                    else reinterpret_cast<float *>(Cv)[o] = acc[a][b][g];
                }
            }
        }
}

__global__ void to_bf16_kernel(const float *__restrict__ x, __bf16 *__restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (__bf16)x[i];
}

template <int BM, int BN, int WM, int WN, int MINB, bool RN, bool SWZ, bool PRE_A = false, bool PRE_B = false>
static double run_v2(const struct Shape &s, const float *A, const float *B, float *C, int iters);

struct Shape { int N, Mo, K, F; };

static void fill(std::vector<float> &h, unsigned seed, float scale) {
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < h.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        // full 24-bit mantissas, mixed magnitudes (a few octaves) so that the low pieces matter
        const float m = ((int)(s >> 8) % 16777216) / 16777216.0f - 0.5f;
        s = s * 1664525u + 1013904223u;
        h[i] = scale * m * (1 << ((s >> 28) & 3));
    }
}

template <int BM, int BN, int NT, int MINB, int PF = 1, int DBG = 0>
static double run(const Shape &s, const float *A, const float *B, float *C, int iters) {
    const int rt = (s.Mo + BM - 1) / BM, ct = (s.F + BN - 1) / BN;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { gemm_bf16x3_kernel<BM, BN, NT, MINB, PF, DBG><<<s.N * rt * ct, 256>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct); };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 1e3 * ms / iters;
}

template <int BM, int BN, int WM, int WN, int MINB, bool RN, bool SWZ, bool PRE_A, bool PRE_B>
static double run_v2(const Shape &s, const float *A, const float *B, float *C, int iters) {
    const int rt = (s.Mo + BM - 1) / BM, ct = (s.F + BN - 1) / BN;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned short *A3 = nullptr, *B3 = nullptr;
    const long long na = (long long)s.N * s.Mo * s.K, nb = (long long)s.F * s.K;
    if (PRE_A) {          // the producer-side split is not part of the timed region (it would ride in the producer's epilogue)
        hipMalloc(&A3, na * 6);
        presplit_kernel<RN><<<(unsigned)((na / 2 + 255) / 256), 256>>>(A, (unsigned *)A3, na / 2);
    }
    if (PRE_B) {
        hipMalloc(&B3, nb * 6);
        presplit_kernel<RN><<<(unsigned)((nb / 2 + 255) / 256), 256>>>(B, (unsigned *)B3, nb / 2);
    }
    auto launch = [&]() {
        gemm_v2_kernel<BM, BN, WM, WN, MINB, RN, SWZ, PRE_A, PRE_B><<<s.N * rt * ct, 64 * WM * WN>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct, A3, B3);
    };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (A3) hipFree(A3);
    if (B3) hipFree(B3);
    return 1e3 * ms / iters;
}

template <int SG>
static double run_v3(const Shape &s, const float *A, const float *B, float *C, int iters) {
    const int rt = (s.Mo + 127) / 128, ct = (s.F + 127) / 128;
    const int lds = 2 * 3 * (128 * PITCH + 128 * PITCH);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_v3_kernel<SG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { gemm_v3_kernel<SG><<<s.N * rt * ct, 256, lds>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct); };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 1e3 * ms / iters;
}

// error of rows [ra, rb) of sample n against a float64 reference; also the error an fp32 sequential dot makes
static void check(const Shape &s, const std::vector<float> &hA, const std::vector<float> &hB, const float *dC, int n, int ra, int rb,
                  double &err_max, double &err_rms, double &f32_rms) {
    std::vector<float> hC((size_t)(rb - ra) * s.F);
    hipMemcpy(hC.data(), dC + ((size_t)n * s.Mo + ra) * s.F, hC.size() * 4, hipMemcpyDeviceToHost);
    double se = 0, sf = 0, sr = 0;
    err_max = 0;
    for (int r = ra; r < rb; ++r)
        for (int f = 0; f < s.F; ++f) {
            const float *a = &hA[((size_t)n * s.Mo + r) * s.K], *b = &hB[(size_t)f * s.K];
            double ref = 0;
            float f32 = 0.f;
            for (int k = 0; k < s.K; ++k) { ref += (double)a[k] * (double)b[k]; f32 = fmaf(a[k], b[k], f32); }
            const double e = (double)hC[(size_t)(r - ra) * s.F + f] - ref;
            se += e * e; sf += ((double)f32 - ref) * ((double)f32 - ref); sr += ref * ref;
            if (fabs(e) > err_max) err_max = fabs(e);
        }
    const double cnt = (double)(rb - ra) * s.F;
    const double scale = sqrt(sr / cnt);
    err_max /= scale; err_rms = sqrt(se / cnt) / scale; f32_rms = sqrt(sf / cnt) / scale;
}

int main(int argc, char **argv) {
    std::vector<Shape> shapes = {{16, 862, 1024, 512}, {16, 862, 768, 512}, {16, 862, 512, 512}, {16, 862, 512, 256},
                                 {16, 1723, 256, 256}};
    if (argc > 1 && !strcmp(argv[1], "short"))       // the short-contraction launches of the step (1x1 convs, narrow levels)
        shapes = {{16, 862, 64, 512}, {16, 862, 256, 512}, {16, 1723, 128, 256}, {16, 1723, 256, 256}, {16, 3445, 64, 128},
                  {16, 3445, 128, 128}, {16, 862, 512, 64}};
    const int iters = 20;
    {
        float *o; hipMalloc(&o, 2048 * 256 * 4);
        peak(256, 20000, o); peak(512, 10000, o); peak(512, 50, o); peak(1024, 5000, o);
        hipFree(o);
    }
    if (argc > 1 && !strcmp(argv[1], "bf16")) {
        printf("%-22s %22s %22s %22s   %s\n", "shape (N Mo K F)", "128x128 fp32 out", "128x128 bf16 out", "64x64 bf16 out", "rms err vs float64 of the fp32 inputs");
        for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 512, 256}, {16, 1723, 256, 256}, {16, 3445, 128, 128},
                                                 {16, 6890, 64, 64}}) {
            std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
            fill(hA, 7, 1.0f);
            fill(hB, 100, 0.05f);
            float *A, *B, *C;
            __bf16 *A16, *B16;
            const long long na = (long long)hA.size(), nb = (long long)hB.size(), nc = (long long)s.N * s.Mo * s.F;
            hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, nc * 4); hipMalloc(&A16, na * 2); hipMalloc(&B16, nb * 2);
            hipMemcpy(A, hA.data(), na * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), nb * 4, hipMemcpyHostToDevice);
            to_bf16_kernel<<<(unsigned)((na + 255) / 256), 256>>>(A, A16, na);
            to_bf16_kernel<<<(unsigned)((nb + 255) / 256), 256>>>(B, B16, nb);
            const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
            auto timeit = [&](auto launch) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                launch();
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int i = 0; i < iters; ++i) launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                return 1e3 * ms / iters;
            };
            const int rt = (s.Mo + 127) / 128, ct = (s.F + 127) / 128, rt6 = (s.Mo + 63) / 64, ct6 = (s.F + 63) / 64;
            double emax, erms, f32rms;
            const double u0 = timeit([&]() { gemm_bf16_kernel<128, 128, 2, 2, 2, false><<<s.N * rt * ct, 256>>>((const unsigned short *)A16, (const unsigned short *)B16, C, s.N, s.Mo, s.K, s.F, rt, ct); });
            check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax, erms, f32rms);
            const double u1 = timeit([&]() { gemm_bf16_kernel<128, 128, 2, 2, 2, true><<<s.N * rt * ct, 256>>>((const unsigned short *)A16, (const unsigned short *)B16, C, s.N, s.Mo, s.K, s.F, rt, ct); });
            const double u2 = timeit([&]() { gemm_bf16_kernel<64, 64, 2, 2, 4, true><<<s.N * rt6 * ct6, 256>>>((const unsigned short *)A16, (const unsigned short *)B16, C, s.N, s.Mo, s.K, s.F, rt6, ct6); });
            char name[64];
            snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
            const double by32 = 2.0 * (na + nb) + 4.0 * nc, by16 = 2.0 * (na + nb + nc);
            printf("%-22s %7.1fus %5.0fTF %4.1fTB/s %7.1fus %5.0fTF %4.1fTB/s %7.1fus %5.0fTF %4.1fTB/s   %.2e\n", name, u0, fl / u0 / 1e6,
                   by32 / u0 / 1e6, u1, fl / u1 / 1e6, by16 / u1 / 1e6, u2, fl / u2 / 1e6, by16 / u2 / 1e6, erms);
            hipFree(A); hipFree(B); hipFree(C); hipFree(A16); hipFree(B16);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "v3")) {
        const char *vn[] = {"v2 128x128 trunc (ref)", "v3 dbuf 1 barrier", "v3 + sched 1:4", "v3 + frag prefetch"};
        constexpr int NV3 = 4;
        printf("%-22s", "shape (N Mo K F)");
        for (int i = 0; i < NV3; ++i) printf(" %24s", vn[i]);
        printf("\n");
        for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 512, 512}, {16, 862, 512, 256}, {16, 1723, 256, 256},
                                                 {16, 3445, 128, 128}}) {
            std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
            fill(hA, 7, 1.0f);
            fill(hB, 100, 0.05f);
            float *A, *B, *C;
            hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)s.N * s.Mo * s.F * 4);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
            const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
            double us[NV3], emax[NV3], erms[NV3], f32rms = 0;
            auto chk = [&](int i) { check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax[i], erms[i], f32rms); };
            auto clr = [&]() { hipMemset(C, 0xFF, (size_t)s.N * s.Mo * s.F * 4); };
            clr(); us[0] = run_v2<128, 128, 2, 2, 2, false, false>(s, A, B, C, iters); chk(0);
            clr(); us[1] = run_v3<0>(s, A, B, C, iters); chk(1);
            clr(); us[2] = run_v3<1>(s, A, B, C, iters); chk(2);
            clr(); us[3] = run_v3<10>(s, A, B, C, iters); chk(3);
            char name[64];
            snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
            printf("%-22s", name);
            for (int i = 0; i < NV3; ++i) printf(" %14.1fus %5.1fTF", us[i], fl / us[i] / 1e6);
            printf("\n%-22s", "  rms err/rms(ref)");
            for (int i = 0; i < NV3; ++i) printf(" %24.2e", erms[i]);
            printf("   fp32 fma chain: %.2e\n", f32rms);
            hipFree(A); hipFree(B); hipFree(C);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "v2")) {
        const char *vn[] = {"128x128 2x2 trunc", "128x128 2x2 RN", "2x2 RN swz occ3", "4x2 RN (8 waves)", "2x4 RN (8 waves)", "64x64 2x2 RN occ5",
                            "swz occ2 B presplit", "swz occ2 A+B presplit"};
        constexpr int NV2 = 8;
        printf("%-22s", "shape (N Mo K F)");
        for (int i = 0; i < NV2; ++i) printf(" %18s", vn[i]);
        printf("\n");
        for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 512, 512}, {16, 862, 512, 256}, {16, 1723, 256, 256},
                                                 {16, 3445, 128, 128}, {16, 3445, 192, 64}}) {
            std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
            fill(hA, 7, 1.0f);
            fill(hB, 100, 0.05f);
            float *A, *B, *C;
            hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)s.N * s.Mo * s.F * 4);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
            const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
            double us[NV2], emax[NV2], erms[NV2], f32rms = 0;
            auto chk = [&](int i) { check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax[i], erms[i], f32rms); };
            auto clr = [&]() { hipMemset(C, 0xFF, (size_t)s.N * s.Mo * s.F * 4); };
            clr(); us[0] = run_v2<128, 128, 2, 2, 2, false, false>(s, A, B, C, iters); chk(0);
            clr(); us[1] = run_v2<128, 128, 2, 2, 2, true, false>(s, A, B, C, iters); chk(1);
            clr(); us[2] = run_v2<128, 128, 2, 2, 3, true, true>(s, A, B, C, iters); chk(2);
            clr(); us[3] = run_v2<128, 128, 4, 2, 1, true, false>(s, A, B, C, iters); chk(3);
            clr(); us[4] = run_v2<128, 128, 2, 4, 1, true, false>(s, A, B, C, iters); chk(4);
            clr(); us[5] = run_v2<64, 64, 2, 2, 5, true, false>(s, A, B, C, iters); chk(5);
            clr(); us[6] = run_v2<128, 128, 2, 2, 2, true, true, false, true>(s, A, B, C, iters); chk(6);
            clr(); us[7] = run_v2<128, 128, 2, 2, 2, true, true, true, true>(s, A, B, C, iters); chk(7);
            char name[64];
            snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
            printf("%-22s", name);
            for (int i = 0; i < NV2; ++i) printf(" %8.1fus %5.1fTF", us[i], fl / us[i] / 1e6);
            printf("\n%-22s", "  rms err/rms(ref)");
            for (int i = 0; i < NV2; ++i) printf(" %18.2e", erms[i]);
            printf("   fp32 fma chain: %.2e\n", f32rms);
            hipFree(A); hipFree(B); hipFree(C);
        }
        return 0;
    }
    const char *names[] = {"128x128 pf1", "neither", "neither+nobar", "nobar only", "x3 neither", "x9 neither", "64x64 neither", "128x64 neither"};
    constexpr int NV = 8;
    printf("%-22s", "shape (N Mo K F)");
    for (int i = 0; i < NV; ++i) printf(" %16s", names[i]);
    printf("\n");
    for (const Shape &s : shapes) {
        std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
        fill(hA, 7, 1.0f);
        fill(hB, 100, 0.05f);
        float *A, *B, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)s.N * s.Mo * s.F * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
        double us[NV], emax[NV], erms[NV], f32rms = 0;
        auto chk = [&](int i) { check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax[i], erms[i], f32rms); };
        auto clr = [&]() { hipMemset(C, 0xFF, (size_t)s.N * s.Mo * s.F * 4); };
        clr(); us[0] = run<128, 128, 6, 2, 1, 0>(s, A, B, C, iters); chk(0);
        clr(); us[1] = run<128, 128, 6, 2, 1, 3>(s, A, B, C, iters); chk(1);
        clr(); us[2] = run<128, 128, 6, 2, 1, 11>(s, A, B, C, iters); chk(2);
        clr(); us[3] = run<128, 128, 6, 2, 1, 8>(s, A, B, C, iters); chk(3);
        clr(); us[4] = run<128, 128, 3, 2, 1, 3>(s, A, B, C, iters); chk(4);
        clr(); us[5] = run<128, 128, 9, 2, 1, 3>(s, A, B, C, iters); chk(5);
        clr(); us[6] = run<64, 64, 6, 5, 1, 3>(s, A, B, C, iters); chk(6);
        clr(); us[7] = run<128, 64, 6, 2, 1, 3>(s, A, B, C, iters); chk(7);
        char name[64];
        snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
        printf("%-22s", name);
        for (int i = 0; i < NV; ++i) printf(" %6.1fus %5.1fTF", us[i], fl / us[i] / 1e6);
        printf("\n%-22s", "  rms err/rms(ref)");
        for (int i = 0; i < NV; ++i) printf(" %16.2e", erms[i]);
        printf("   fp32 fma chain: %.2e\n", f32rms);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
