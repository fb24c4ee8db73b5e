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
// RND: 1 = operands with full random mantissas and mixed exponents, different in every lane (switching activity of real
// data: the chip clocks to its power budget); ticks: s_memtime before / after the loop of block 0, wave 0
template <int NACC, int RND = 0>
__global__ __launch_bounds__(256) void mfma_bf16_loop(float *out, int iters, unsigned long long *ticks = nullptr) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) acc[a][g] = 0.f;
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(1e-3f * (1 + (threadIdx.x + j) % 7)); bv[j] = (__bf16)(1e-3f * (2 + (threadIdx.x + j) % 5)); }
    if (RND) {
        unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            av[j] = (__bf16)(((int)(s >> 8) % 65536 - 32768) * (1.0f / 32768) * 1e-3f * (1 << ((s >> 28) & 3)));
            s = s * 1664525u + 1013904223u;
            bv[j] = (__bf16)(((int)(s >> 8) % 65536 - 32768) * (1.0f / 32768) * 1e-3f * (1 << ((s >> 28) & 3)));
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[a], 0, 0, 0);
    }
    float sum = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) sum += acc[a][g];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (ticks && blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int RND>
static void peak_ticks(int blocks, int iters, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long *tk, h = 0;
    hipMalloc(&tk, 8);
    mfma_bf16_loop<4, RND><<<blocks, 256>>>(out, iters, tk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_bf16_loop<4, RND><<<blocks, 256>>>(out, iters, tk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
    const double fl = (double)blocks * 4 * iters * 4 * 32768.0;
    printf("bf16 MFMA loop, %s operands: %d blocks x %d iters: %.3f ms  %.0f TFLOP/s;  %.1f s_memtime ticks per MFMA per SIMD, tick rate %.2f GHz\n",
           RND ? "random" : "constant", blocks, iters, ms, fl / ms / 1e9, (double)h / (4.0 * iters * (blocks / 256.0)), (double)h / ms / 1e6);
    hipFree(tk);
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
template <int BM, int BN, int WM, int WN, int MINB, bool RN, bool SWZ, bool PRE_A = false, bool PRE_B = false, int ILV = 0>
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
        if constexpr (ILV != 0) {
            // ILV: the fragment reads of k16 step 1 sit one per two MFMAs inside step 0 (a ds_read_b128 blocks its wave's
            // issue for ~29 cycles; twelve in a row stall the wave's MFMA stream for ~350 cycles, r03_ubench_v5c.txt)
            bf16x8 fa[2][TM][3], fb[2][TN][3];
            auto rd = [&](int ks) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int row = wm * WTM + a * 32 + li;
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        fa[ks][a][p] = *reinterpret_cast<const bf16x8 *>(sA + p * APLANE + row * LP + 16 * seg(row, lh + 2 * ks));
                }
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int row = wn * WTN + b * 32 + li;
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        fb[ks][b][p] = *reinterpret_cast<const bf16x8 *>(sB + p * BPLANE + row * LP + 16 * seg(row, lh + 2 * ks));
                }
            };
            auto mm = [&](int ks) {
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][a][term_pa(6, term)], fb[ks][b][term_pb(6, term)],
                                                                                acc[a][b], 0, 0, 0);
            };
            if constexpr (ILV == 2 && TM == 2 && TN == 2) {
                // ILV 2: only the four fragments of the first product are read up front; the other eight of step 0 ride one
                // per MFMA inside products 0 and 1, the twelve of step 1 inside products 2 .. 4
                constexpr int oa[3] = {0, 2, 1}, ob[3] = {2, 0, 1};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int st = 0; st < 3; ++st) {
#pragma unroll
                        for (int a = 0; a < TM; ++a) {
                            const int row = wm * WTM + a * 32 + li;
                            fa[ks][a][oa[st]] = *reinterpret_cast<const bf16x8 *>(sA + oa[st] * APLANE + row * LP + 16 * seg(row, lh + 2 * ks));
                        }
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            const int row = wn * WTN + b * 32 + li;
                            fb[ks][b][ob[st]] = *reinterpret_cast<const bf16x8 *>(sB + ob[st] * BPLANE + row * LP + 16 * seg(row, lh + 2 * ks));
                        }
                    }
                mm(0);
                mm(1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int i = 0; i < 20; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 28, 0);
                return;
            }
            rd(0);
            __builtin_amdgcn_sched_barrier(0);
            rd(1);
            mm(0);
#pragma unroll
            for (int i = 0; i < 3 * (TM + TN); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mm(1);
            return;
        }
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
// Fourth generation (round 3): explicit PING-PONG.  512 threads = two groups of four waves; each group owns one 128 x 128
// output tile and its own single-buffered piece planes (2 x 60 KB of LDS -> one workgroup per CU, two waves per SIMD, one
// of each group).  The groups run in anti-phase, locked by workgroup barriers: while one group multiplies its staged chunk
// (MFMA + LDS fragment reads), the other splits and stores its next chunk (VALU + LDS writes); the global loads of a
// chunk are issued at the start of the multiply phase before the one that stages them.  Rationale: two independent
// 256-thread workgroups per CU (v2) fall into lockstep -- when both multiply they share the matrix pipe and finish
// together, then both stage while the pipe idles -- so the stage time is never hidden (MI355X_MICROARCH.md, "Two waves per
// SIMD": the matrix pipe is per SIMD and fully paced; a partner's VALU costs the multiplying wave little).
// PRIO: 1 = s_setprio 1 during the multiply phase.
// ---------------------------------------------------------------------------------------------------------------
template <int PRIO>
__global__ __launch_bounds__(512, 1) void gemm_v4_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                         float *__restrict__ C, int N, int Mo, int K, int F, int row_tiles,
                                                         int col_tiles, unsigned long long *ts = nullptr) {
    constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2, PA = 2, PB = 2;
    constexpr int APLANE = BM * PITCH, BPLANE = BN * PITCH, GROUP = 3 * (APLANE + BPLANE);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    unsigned char *sA = smem4 + grp * GROUP, *sB = sA + 3 * APLANE;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int q = tid & 3, r = tid >> 2;

    // the two tiles of a workgroup are consecutive tiles of one sample's list on this block's XCD
    const int T = row_tiles * col_tiles;
    int n, t;
    bool valid;
    if ((N & 7) == 0) {
        const int per = N >> 3, l = (blockIdx.x >> 3) * 2 + grp;
        valid = l < per * T && !((PRIO & 4) && grp == 1);      // PRIO & 4: solo, the second group idles (timing only)
        n = (blockIdx.x & 7) * per + (valid ? l / T : 0);
        t = valid ? l % T : 0;
    } else {
        const int l = blockIdx.x * 2 + grp;
        valid = l < N * T;
        n = valid ? l / T : 0;
        t = valid ? l % T : 0;
    }
    // row tile fastest: the two groups of a workgroup read the same weight columns
    const int r0 = (t % row_tiles) * BM, f0 = (t / row_tiles) * BN;

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

    float4 ra[PA][2], rb[PB][2];
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            ra[i][0] = *reinterpret_cast<const float4 *>(ap[i] + k0);
            ra[i][1] = *reinterpret_cast<const float4 *>(ap[i] + k0 + 4);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            rb[i][0] = *reinterpret_cast<const float4 *>(bp[i] + k0);
            rb[i][1] = *reinterpret_cast<const float4 *>(bp[i] + k0 + 4);
        }
    };
    auto store8 = [&](unsigned char *base, int plane, int row, const float4 &u, const float4 &v) {
        uint4 hi, mid, lo;
        split8(u, v, hi, mid, lo);
        unsigned char *d = base + row * PITCH + 16 * q;
        *reinterpret_cast<uint4 *>(d) = hi;
        *reinterpret_cast<uint4 *>(d + plane) = mid;
        *reinterpret_cast<uint4 *>(d + 2 * plane) = lo;
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) store8(sA, APLANE, r + 64 * i, ra[i][0], ra[i][1]);
#pragma unroll
        for (int i = 0; i < PB; ++i) store8(sB, BPLANE, r + 64 * i, rb[i][0], rb[i][1]);
    };
    auto compute = [&]() {
        if constexpr (PRIO & 8) {
            // PRIO & 8: fragment reads issued a whole k16 step ahead, in the order the products consume them (LDS returns in
            // order: the first product starts after four reads, the rest of the stream lands behind the MFMAs); scheduling
            // barriers keep the compiler from sinking each read next to its first use (its default: read 4, wait, 4 MFMAs ...)
            bf16x8 af[2][TM][3], bf[2][TN][3];
            constexpr int oa[3] = {0, 2, 1}, ob[3] = {2, 0, 1};          // piece of A / B first needed by product 0, 1, 2
            auto issue = [&](int ks) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        af[ks][a][oa[s]] = *reinterpret_cast<const bf16x8 *>(sA + oa[s] * APLANE + (wm * WTM + a * 32 + li) * PITCH + 16 * (lh + 2 * ks));
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        bf[ks][b][ob[s]] = *reinterpret_cast<const bf16x8 *>(sB + ob[s] * BPLANE + (wn * WTN + b * 32 + li) * PITCH + 16 * (lh + 2 * ks));
                    __builtin_amdgcn_sched_barrier(0);        // keep the sets in consumption order
                }
            };
            auto terms = [&](int ks, int t0, int t1) {
#pragma unroll
                for (int term = t0; term < t1; ++term)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][a][term_pa(6, term)], bf[ks][b][term_pb(6, term)],
                                                                                acc[a][b], 0, 0, 0);
            };
            issue(0);
            __builtin_amdgcn_sched_barrier(0);
            terms(0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            issue(1);
            __builtin_amdgcn_sched_barrier(0);
            terms(0, 1, 6);
            terms(1, 0, 6);
            return;
        }
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row = wm * WTM + a * 32 + li;
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[a][p] = *reinterpret_cast<const bf16x8 *>(sA + p * APLANE + row * PITCH + 16 * (lh + 2 * ks));
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int row = wn * WTN + b * 32 + li;
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bf[b][p] = *reinterpret_cast<const bf16x8 *>(sB + p * BPLANE + row * PITCH + 16 * (lh + 2 * ks));
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

    // local step j of a group: even = stage chunk j/2, odd = multiply chunk j/2; group g performs step j in phase j + g
    const int total = K / KC;
    if (valid) load_regs(0);
    // PRIO & 2: phase timestamps of wave 0 of each group of block 8 (s_memtime): [group][phase][start, work done, barrier passed]
    const bool stamp = (PRIO & 2) && ts != nullptr && blockIdx.x == 8 && tid == 0;
    for (int ph = 0; ph <= 2 * total; ++ph) {
        const int j = ph - grp;
        unsigned long long t0 = 0, t1 = 0;
        if (PRIO & 2) t0 = __builtin_amdgcn_s_memtime();
        if (valid && j >= 0 && j < 2 * total) {
            const int it = j >> 1;
            if (j & 1) {
                if (it + 1 < total) load_regs((it + 1) * KC);
                if (PRIO & 1) __builtin_amdgcn_s_setprio(1);
                compute();
                if (PRIO & 1) __builtin_amdgcn_s_setprio(0);
            } else {
                store_regs();
            }
        }
        if (PRIO & 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
        }
        __syncthreads();
        if (stamp && ph < 64) {
            unsigned long long *o = ts + ((long long)grp * 64 + ph) * 3;
            o[0] = t0; o[1] = t1; o[2] = __builtin_amdgcn_s_memtime();
        }
    }

    if (!valid) return;
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
// Fifth generation (round 3): PRODUCER / CONSUMER specialisation.  512 threads, one 128 x 128 tile per workgroup, one
// workgroup per CU.  Waves 0-3 (one per SIMD) only multiply: fragment reads + MFMAs, the fragments of the next k16 step
// (also across the chunk boundary) always in flight behind the MFMAs of the current one.  Waves 4-7 (their SIMD partners)
// only stage: global loads two chunks ahead, operand split, LDS stores.  Three chunk buffers in LDS (unpadded 64-byte rows,
// 16-byte segments XOR-swizzled: 3 x 48 KB) so that the chunk the consumers prefetch from is always complete; ONE
// workgroup barrier per chunk.  Measured background (r03_ubench_v4_*.txt): a wave that multiplies AND stages spends 2170
// cycles on a 1536-cycle MFMA stream even alone on its SIMD (exposed first fragment reads, loop overhead), and 2400-2600
// next to a staging partner; the matrix pipe issues one 32x32x16 MFMA per 32 cycles at 1.75-1.87 GHz under load.
// VAR & 1: weights pre-split into bf16 planes in HBM (B3);  VAR & 2: phase timestamps.
// ---------------------------------------------------------------------------------------------------------------
template <int VAR>
__global__ __launch_bounds__((VAR & 8) ? 768 : 512, 1) void gemm_v5_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                         float *__restrict__ C, int N, int Mo, int K, int F, int row_tiles,
                                                         int col_tiles, const unsigned short *__restrict__ B3 = nullptr,
                                                         unsigned long long *ts = nullptr) {
    constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2;
    constexpr int NPT = (VAR & 8) ? 512 : 256, RPP = NPT / 4, PA = BM / RPP, PB = BN / RPP;      // producer threads, rows per staging pass
    constexpr int LP = 64, APLANE = BM * LP, BPLANE = BN * LP, BUF = 3 * (APLANE + BPLANE);
    constexpr bool PRE_B = (VAR & 1) != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem5[];
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) ? 1 : 0;      // 0: consumer (waves 0-3), 1: producer (waves 4-7 [-11])
    const int tid = role ? (int)threadIdx.x - 256 : (int)threadIdx.x;
    auto seg = [](int row, int sg) { return sg ^ ((row >> 2) & 3); };

    int n, t;
    map_block(blockIdx.x, N, row_tiles * col_tiles, n, t);
    const int r0 = (t / col_tiles) * BM, f0 = (t % col_tiles) * BN;
    const int total = K / KC;
    const bool stamp = (VAR & 2) && ts != nullptr && blockIdx.x == 8 && tid == 0;

    if (role == 1) {
        if (VAR & 4) __builtin_amdgcn_s_setprio(3);
        // ---------------- producer: chunk c -> buffer c % 3, global loads two chunks ahead ----------------
        const int q = tid & 3, r = tid >> 2;
        const float *ap[PA], *bp[PB];
#pragma unroll
        for (int i = 0; i < PA; ++i) ap[i] = A + ((long long)n * Mo + min(r0 + r + RPP * i, Mo - 1)) * K + 8 * q;
#pragma unroll
        for (int i = 0; i < PB; ++i) bp[i] = B + (long long)min(f0 + r + RPP * i, F - 1) * K + 8 * q;
        const long long b_plane = (long long)F * K;
        float4 ra[2][PA][2], rb[2][PB][2];
        uint4 pb3[2][PRE_B ? PB : 1][3];
        auto load_regs = [&](int set, int k0) {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                ra[set][i][0] = *reinterpret_cast<const float4 *>(ap[i] + k0);
                ra[set][i][1] = *reinterpret_cast<const float4 *>(ap[i] + k0 + 4);
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                if constexpr (PRE_B) {
                    const long long e = (bp[i] - B) + k0;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) pb3[set][i][pl] = *reinterpret_cast<const uint4 *>(B3 + pl * b_plane + e);
                } else {
                    rb[set][i][0] = *reinterpret_cast<const float4 *>(bp[i] + k0);
                    rb[set][i][1] = *reinterpret_cast<const float4 *>(bp[i] + k0 + 4);
                }
            }
        };
        auto store_regs = [&](int set, unsigned char *buf) {
            unsigned char *sA = buf, *sB = buf + 3 * APLANE;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                uint4 hi, mid, lo;
                split8(ra[set][i][0], ra[set][i][1], hi, mid, lo);
                const int row = r + RPP * i;
                unsigned char *d = sA + row * LP + 16 * seg(row, q);
                *reinterpret_cast<uint4 *>(d) = hi;
                *reinterpret_cast<uint4 *>(d + APLANE) = mid;
                *reinterpret_cast<uint4 *>(d + 2 * APLANE) = lo;
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                uint4 hi, mid, lo;
                if constexpr (PRE_B) { hi = pb3[set][i][0]; mid = pb3[set][i][1]; lo = pb3[set][i][2]; }
                else split8(rb[set][i][0], rb[set][i][1], hi, mid, lo);
                const int row = r + RPP * i;
                unsigned char *d = sB + row * LP + 16 * seg(row, q);
                *reinterpret_cast<uint4 *>(d) = hi;
                *reinterpret_cast<uint4 *>(d + BPLANE) = mid;
                *reinterpret_cast<uint4 *>(d + 2 * BPLANE) = lo;
            }
        };
        if (!(VAR & 16)) load_regs(0, 0);
        if (total > 1 && !(VAR & 16)) load_regs(1, KC);
        int bo = 0;
        for (int c = 0; c < total; c += 2) {
            unsigned long long t0 = 0, t1 = 0;
            if (VAR & 2) t0 = __builtin_amdgcn_s_memtime();
            if (!(VAR & 16)) store_regs(0, smem5 + bo);
            if (c + 2 < total && !(VAR & 16)) load_regs(0, (c + 2) * KC);
            if (VAR & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t1 = __builtin_amdgcn_s_memtime(); }
            __syncthreads();
            if (stamp && c < 64) { ts[(64 + c) * 3] = t0; ts[(64 + c) * 3 + 1] = t1; ts[(64 + c) * 3 + 2] = __builtin_amdgcn_s_memtime(); }
            bo = (bo == 2 * BUF) ? 0 : bo + BUF;
            if (c + 1 < total) {
                if (VAR & 2) t0 = __builtin_amdgcn_s_memtime();
                if (!(VAR & 16)) store_regs(1, smem5 + bo);
                if (c + 3 < total && !(VAR & 16)) load_regs(1, (c + 3) * KC);
                if (VAR & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t1 = __builtin_amdgcn_s_memtime(); }
                __syncthreads();
                if (stamp && c + 1 < 64) { ts[(64 + c + 1) * 3] = t0; ts[(64 + c + 1) * 3 + 1] = t1; ts[(64 + c + 1) * 3 + 2] = __builtin_amdgcn_s_memtime(); }
                bo = (bo == 2 * BUF) ? 0 : bo + BUF;
            }
        }
        __syncthreads();          // barriers #total, #total + 1: the consumers lag one chunk behind
        __syncthreads();
        return;
    }

    // ---------------- consumer ----------------
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
    // byte offsets of this lane's row segments inside a buffer (rows wm*64 + a*32 + li: the swizzle term is that of li)
    const int sw = (li >> 2) & 3;
    const int offA = (wm * WTM + li) * LP, offB = 3 * APLANE + (wn * WTN + li) * LP;
    constexpr int oa[3] = {0, 2, 1}, ob[3] = {2, 0, 1};          // piece of A / B first needed by product 0, 1, 2
    bf16x8 f0a[TM][3], f0b[TN][3], f1a[TM][3], f1b[TN][3];
    auto issue = [&](bf16x8 (&fa)[TM][3], bf16x8 (&fb)[TN][3], const unsigned char *buf, int ks) {
        const int so = 16 * ((lh + 2 * ks) ^ sw);
#pragma unroll
        for (int st = 0; st < 3; ++st) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
                fa[a][oa[st]] = *reinterpret_cast<const bf16x8 *>(buf + offA + oa[st] * APLANE + a * 32 * LP + so);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                fb[b][ob[st]] = *reinterpret_cast<const bf16x8 *>(buf + offB + ob[st] * BPLANE + b * 32 * LP + so);
            if (!(VAR & 64)) __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto terms = [&](bf16x8 (&fa)[TM][3], bf16x8 (&fb)[TN][3], int t0, int t1) {
#pragma unroll
        for (int term = t0; term < t1; ++term)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][term_pa(6, term)], fb[b][term_pb(6, term)], acc[a][b], 0, 0, 0);
    };
    // VAR & 64: one fragment read in the shadow of every second MFMA (a ds_read_b128 blocks its wave's issue for ~29 cycles:
    // twelve in a row leave the matrix pipe idle for ~350 cycles, measured r03_ubench_v5c.txt)
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
        }
    };
    __syncthreads();              // #0: chunk 0 staged
    __syncthreads();              // #1: chunk 1 staged
    int bo = 0;
    issue(f0a, f0b, smem5, 0);
    if (VAR & 32) issue(f1a, f1b, smem5, 1);
    for (int c = 0; c < total; ++c) {
        unsigned long long t0 = 0, t1 = 0;
        if (VAR & 2) t0 = __builtin_amdgcn_s_memtime();
        const unsigned char *buf = smem5 + bo;
        bo = (bo == 2 * BUF) ? 0 : bo + BUF;
        if constexpr ((VAR & 64) != 0) {
            issue(f1a, f1b, buf, 1);
            terms(f0a, f0b, 0, 6);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            issue(f0a, f0b, smem5 + bo, 0);          // (the last chunk prefetches from a buffer nobody uses)
            terms(f1a, f1b, 0, 6);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
        } else {
            if (!(VAR & 32)) issue(f1a, f1b, buf, 1);     // k16 step 1 of this chunk, behind the MFMAs of step 0
            __builtin_amdgcn_sched_barrier(0);
            terms(f0a, f0b, 0, 6);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < total && !(VAR & 32)) issue(f0a, f0b, smem5 + bo, 0);      // step 0 of the NEXT chunk (complete since the last barrier)
            __builtin_amdgcn_sched_barrier(0);
            terms(f1a, f1b, 0, 6);
        }
        if (VAR & 2) t1 = __builtin_amdgcn_s_memtime();
        __syncthreads();          // #(c + 2)
        if (stamp && c < 64) { ts[c * 3] = t0; ts[c * 3 + 1] = t1; ts[c * 3 + 2] = __builtin_amdgcn_s_memtime(); }
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
// Sixth generation (round 3): k16 chunks in two LDS half-buffers, EVERYTHING of the next chunks interleaved into the MFMA
// stream of the current one -- the fragment reads of chunk i+1 (from half (i+1)&1), the operand split of chunk i+2 (from
// registers) and its LDS stores (into half i&1, whose fragments were read an iteration ago), the global loads of chunk
// i+3.  One barrier per k16.  Same 128 x 128 tile / 2 x 2 waves / two workgroups per CU as the library kernel; 32-byte LDS
// rows with the segment XOR-ed by (row >> 3) & 1 (conflict-free for the ds_read_b128 lane groups): 48 KB per workgroup.
// SG: 0 = leave the order to the compiler, 1 = IGroupLP pipeline (per MFMA: 4 VALU, a read every second, a store every fourth)
// ---------------------------------------------------------------------------------------------------------------
template <int SG>
__global__ __launch_bounds__(256, 2) void gemm_v6_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                         float *__restrict__ C, int N, int Mo, int K, int F, int row_tiles,
                                                         int col_tiles) {
    constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2;
    constexpr int LP = 32, APL = BM * LP, BPL = BN * LP, HALF = 3 * (APL + BPL);          // one k16 chunk: 24 KB
    __shared__ __attribute__((aligned(16))) unsigned char smem6[2 * HALF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int q = tid & 1, r = tid >> 1;                       // staging: row r, eight consecutive k = segment q
    auto seg = [](int row, int sg) { return sg ^ ((row >> 3) & 1); };

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

    const float *ap = A + ((long long)n * Mo + min(r0 + r, Mo - 1)) * K + 8 * q;
    const float *bp = B + (long long)min(f0 + r, F - 1) * K + 8 * q;
    const int total = K / 16;
    struct Regs { float4 a0, a1, b0, b1; };
    auto load_regs = [&](Regs &R, int c) {
        const int k0 = 16 * min(c, total - 1);                  // chunks beyond the end re-read the last one (never used)
        R.a0 = *reinterpret_cast<const float4 *>(ap + k0); R.a1 = *reinterpret_cast<const float4 *>(ap + k0 + 4);
        R.b0 = *reinterpret_cast<const float4 *>(bp + k0); R.b1 = *reinterpret_cast<const float4 *>(bp + k0 + 4);
    };
    auto store_regs = [&](const Regs &R, unsigned char *h) {
        uint4 hi, mid, lo;
        unsigned char *da = h + r * LP + 16 * seg(r, q);
        split8(R.a0, R.a1, hi, mid, lo);
        *reinterpret_cast<uint4 *>(da) = hi; *reinterpret_cast<uint4 *>(da + APL) = mid; *reinterpret_cast<uint4 *>(da + 2 * APL) = lo;
        unsigned char *db = h + 3 * APL + r * LP + 16 * seg(r, q);
        split8(R.b0, R.b1, hi, mid, lo);
        *reinterpret_cast<uint4 *>(db) = hi; *reinterpret_cast<uint4 *>(db + BPL) = mid; *reinterpret_cast<uint4 *>(db + 2 * BPL) = lo;
    };
    struct Frag { bf16x8 a[TM][3], b[TN][3]; };
    auto read_frag = [&](Frag &f, const unsigned char *h) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int row = wm * WTM + a * 32 + li;
#pragma unroll
            for (int p = 0; p < 3; ++p) f.a[a][p] = *reinterpret_cast<const bf16x8 *>(h + p * APL + row * LP + 16 * seg(row, lh));
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int row = wn * WTN + b * 32 + li;
#pragma unroll
            for (int p = 0; p < 3; ++p) f.b[b][p] = *reinterpret_cast<const bf16x8 *>(h + 3 * APL + p * BPL + row * LP + 16 * seg(row, lh));
        }
    };
    auto mm = [&](const Frag &f) {
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[a][term_pa(6, term)], f.b[b][term_pb(6, term)], acc[a][b], 0, 0, 0);
    };

    Regs R0, R1;
    Frag F0, F1;
    load_regs(R0, 0);
    load_regs(R1, 1);
    store_regs(R0, smem6);
    store_regs(R1, smem6 + HALF);
    load_regs(R0, 2);                                           // chunk i + 2 of iteration i = 0
    __syncthreads();
    read_frag(F0, smem6);
    // iteration i: F_cur = chunk i, R_cur = chunk i + 2 (to be staged into half i & 1), loads of chunk i + 3 go out first
    auto iter = [&](Frag &Fc, Frag &Fn, Regs &Rc, Regs &Rn, int i) {
        load_regs(Rn, i + 3);
        read_frag(Fn, smem6 + ((i + 1) & 1) * HALF);
        store_regs(Rc, smem6 + (i & 1) * HALF);
        mm(Fc);
        if (SG == 1) {
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // MFMA
                if (m < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // the four global loads first
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                        // VALU (operand split)
                if ((m & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // fragment read
                if ((m & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // LDS store
            }
        }
        __syncthreads();
    };
    for (int i = 0; i < total; i += 2) {
        iter(F0, F1, R0, R1, i);
        if (i + 1 < total) iter(F1, F0, R1, R0, i + 1);
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
// Seventh generation (round 3): TWO fp16 pieces per operand and THREE products per multiply-add instead of three bf16
// pieces and six products.  x*s = hi + lo with hi = fp16(x*s), lo = fp16(x*s - hi) (both round-to-nearest): 22 significant
// bits + the sign trick of RN leave a representation error of 2^-24 |x|, and the dropped lo*lo product is 2^-24 relative --
// the same class as the bf16 form (numpy emulation: rms 8.3e-8 vs 6.8e-8, an fp32 FMA chain 5.7e-7).  fp16's 5-bit exponent
// needs the operands in range: every A row is scaled by a power of two chosen from the row's absolute maximum (a pre-pass of
// the workgroup over its 128 rows: the tile is read twice, the second time from L2), the weights by one power of two per
// launch (SB, from the tensor's maximum); the epilogue multiplies row r by 1 / (s_r SB) -- all exact.
// Pipeline of v6 (k16 chunks, two LDS half-buffers of 16 KB, one barrier per chunk); PRE 0 = no pre-pass (unit row scales:
// what the contraction alone costs), 1 = with it.
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2h(float x0, float x1, unsigned &hi, unsigned &lo) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    half2v H = {h0, h1}, L = {l0, l1};
    hi = __builtin_bit_cast(unsigned, H);
    lo = __builtin_bit_cast(unsigned, L);
}
__device__ __forceinline__ void split8h(const float4 &u, const float4 &v, float sc, uint4 &hi, uint4 &lo) {
    split2h(u.x * sc, u.y * sc, hi.x, lo.x);
    split2h(u.z * sc, u.w * sc, hi.y, lo.y);
    split2h(v.x * sc, v.y * sc, hi.z, lo.z);
    split2h(v.z * sc, v.w * sc, hi.w, lo.w);
}

template <int PRE, int MINB>
__global__ __launch_bounds__(256, MINB) void gemm_v7_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                            float *__restrict__ C, int N, int Mo, int K, int F, int row_tiles,
                                                            int col_tiles, float SB, const float *__restrict__ rowmax) {
    constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2;
    constexpr int LP = 32, APL = BM * LP, BPL = BN * LP, HALF = 2 * (APL + BPL);          // one k16 chunk: 16 KB
    __shared__ __attribute__((aligned(16))) unsigned char smem7[2 * HALF];
    __shared__ float inv_scale[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int q = tid & 1, r = tid >> 1;                       // staging: row r, eight consecutive k = segment q
    auto seg = [](int row, int sg) { return sg ^ ((row >> 3) & 1); };

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

    const float *ap = A + ((long long)n * Mo + min(r0 + r, Mo - 1)) * K + 8 * q;
    const float *bp = B + (long long)min(f0 + r, F - 1) * K + 8 * q;
    const int total = K / 16;
    float sa = 1.f;
    if (PRE == 2) {
        // the row's absolute maximum comes with the tensor (written by the kernel that produced it): one 4-byte load
        const float m = rowmax[(long long)n * Mo + min(r0 + r, Mo - 1)];
        int e = (int)((fbits(m) >> 23) & 255);
        e = max(e, 14);
        sa = bitsf((unsigned)(267 - e) << 23);
        if (q == 0) inv_scale[r] = bitsf((unsigned)(e - 13) << 23) / SB;
    } else if (PRE) {
        // absolute maximum of row r (this thread: its half of every k16 chunk), then the power of two that puts it in [2^13, 2^14)
        float m0 = 0.f, m1 = 0.f;
#pragma unroll 4
        for (int c = 0; c < total; ++c) {
            const float4 u = *reinterpret_cast<const float4 *>(ap + 16 * c), v = *reinterpret_cast<const float4 *>(ap + 16 * c + 4);
            m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
            m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        float m = fmaxf(m0, m1);
        m = fmaxf(m, __shfl_xor(m, 1));
        int e = (int)((fbits(m) >> 23) & 255);                  // biased exponent (0 for a zero / denormal row)
        e = max(e, 14);
        sa = bitsf((unsigned)(267 - e) << 23);                  // 2^(13 - (e - 127))
        if (q == 0) inv_scale[r] = bitsf((unsigned)(e - 13) << 23) / SB;
    } else if (q == 0) {
        inv_scale[r] = 1.f / SB;
    }
    struct Regs { float4 a0, a1, b0, b1; };
    auto load_regs = [&](Regs &R, int c) {
        const int k0 = 16 * min(c, total - 1);                  // chunks beyond the end re-read the last one (never used)
        R.a0 = *reinterpret_cast<const float4 *>(ap + k0); R.a1 = *reinterpret_cast<const float4 *>(ap + k0 + 4);
        R.b0 = *reinterpret_cast<const float4 *>(bp + k0); R.b1 = *reinterpret_cast<const float4 *>(bp + k0 + 4);
    };
    auto store_regs = [&](const Regs &R, unsigned char *h) {
        uint4 hi, lo;
        unsigned char *da = h + r * LP + 16 * seg(r, q);
        split8h(R.a0, R.a1, sa, hi, lo);
        *reinterpret_cast<uint4 *>(da) = hi; *reinterpret_cast<uint4 *>(da + APL) = lo;
        unsigned char *db = h + 2 * APL + r * LP + 16 * seg(r, q);
        split8h(R.b0, R.b1, SB, hi, lo);
        *reinterpret_cast<uint4 *>(db) = hi; *reinterpret_cast<uint4 *>(db + BPL) = lo;
    };
    struct Frag { half8 a[TM][2], b[TN][2]; };
    auto read_frag = [&](Frag &f, const unsigned char *h) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int row = wm * WTM + a * 32 + li;
#pragma unroll
            for (int p = 0; p < 2; ++p) f.a[a][p] = *reinterpret_cast<const half8 *>(h + p * APL + row * LP + 16 * seg(row, lh));
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int row = wn * WTN + b * 32 + li;
#pragma unroll
            for (int p = 0; p < 2; ++p) f.b[b][p] = *reinterpret_cast<const half8 *>(h + 2 * APL + p * BPL + row * LP + 16 * seg(row, lh));
        }
    };
    auto mm = [&](const Frag &f) {
#pragma unroll
        for (int term = 0; term < 3; ++term)                     // lo*hi, hi*lo, hi*hi: small products first
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[a][term == 0 ? 1 : 0], f.b[b][term == 1 ? 1 : 0], acc[a][b], 0, 0, 0);
    };

    Regs R0, R1;
    Frag F0, F1;
    load_regs(R0, 0);
    load_regs(R1, 1);
    store_regs(R0, smem7);
    store_regs(R1, smem7 + HALF);
    load_regs(R0, 2);
    __syncthreads();
    read_frag(F0, smem7);
    auto iter = [&](Frag &Fc, Frag &Fn, Regs &Rc, Regs &Rn, int i) {
        load_regs(Rn, i + 3);
        read_frag(Fn, smem7 + ((i + 1) & 1) * HALF);
        store_regs(Rc, smem7 + (i & 1) * HALF);
        mm(Fc);
        __syncthreads();
    };
    for (int i = 0; i < total; i += 2) {
        iter(F0, F1, R0, R1, i);
        if (i + 1 < total) iter(F1, F0, R1, R0, i + 1);
    }

    float *cn = C + (long long)n * Mo * F;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = f0 + wn * WTN + b * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int rl = wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                const int row = r0 + rl;
                if (row < Mo && col < F) cn[(long long)row * F + col] = acc[a][b][g] * inv_scale[rl];
            }
        }
}

template <int PRE, int MINB>
static double run_v7(const Shape &s, const float *A, const float *B, float *C, int iters, float SB, const float *rowmax = nullptr);

template <int SG>
static double run_v6(const Shape &s, const float *A, const float *B, float *C, int iters);

template <int PRIO>
static double run_v4(const Shape &s, const float *A, const float *B, float *C, int iters);

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

template <int BM, int BN, int WM, int WN, int MINB, bool RN, bool SWZ, bool PRE_A = false, bool PRE_B = false, int ILV = 0>
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

template <int BM, int BN, int WM, int WN, int MINB, bool RN, bool SWZ, bool PRE_A, bool PRE_B, int ILV>
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
        gemm_v2_kernel<BM, BN, WM, WN, MINB, RN, SWZ, PRE_A, PRE_B, ILV><<<s.N * rt * ct, 64 * WM * WN>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct, A3, B3);
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

template <int PRIO>
static double run_v4(const Shape &s, const float *A, const float *B, float *C, int iters) {
    const int rt = (s.Mo + 127) / 128, ct = (s.F + 127) / 128, T = rt * ct;
    const int lds = 2 * 3 * (128 * PITCH + 128 * PITCH);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_v4_kernel<PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = (s.N & 7) == 0 ? 8 * (((s.N >> 3) * T + 1) / 2) : (s.N * T + 1) / 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { gemm_v4_kernel<PRIO><<<grid, 512, lds>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct); };
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

template <int VAR>
static double run_v5(const Shape &s, const float *A, const float *B, float *C, int iters, bool timing = false) {
    const int rt = (s.Mo + 127) / 128, ct = (s.F + 127) / 128;
    const int lds = 3 * 3 * (128 * 64 + 128 * 64);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_v5_kernel<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned short *B3 = nullptr;
    const long long nb = (long long)s.F * s.K;
    if (VAR & 1) {
        hipMalloc(&B3, nb * 6);
        presplit_kernel<false><<<(unsigned)((nb / 2 + 255) / 256), 256>>>(B, (unsigned *)B3, nb / 2);
    }
    unsigned long long *ts = nullptr;
    if (VAR & 2) { hipMalloc(&ts, 2 * 64 * 3 * 8); hipMemset(ts, 0, 2 * 64 * 3 * 8); }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { gemm_v5_kernel<VAR><<<s.N * rt * ct, (VAR & 8) ? 768 : 512, lds>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct, B3, ts); };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (VAR & 2) {
        std::vector<unsigned long long> h(2 * 64 * 3);
        hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
        double cw = 0, cb = 0, pw = 0, pb = 0;
        for (int c = 6; c < 26; ++c) {
            cw += (double)(h[c * 3 + 1] - h[c * 3]); cb += (double)(h[c * 3 + 2] - h[c * 3 + 1]);
            pw += (double)(h[(64 + c) * 3 + 1] - h[(64 + c) * 3]); pb += (double)(h[(64 + c) * 3 + 2] - h[(64 + c) * 3 + 1]);
        }
        printf("  v5 timing (cycles per chunk, chunks 6..25): consumer multiply %.0f + barrier %.0f;  producer stage %.0f + barrier %.0f;  chunk period %.0f\n",
               cw / 20, cb / 20, pw / 20, pb / 20, (double)(h[26 * 3] - h[6 * 3]) / 20.0);
        hipFree(ts);
    }
    if (B3) hipFree(B3);
    return 1e3 * ms / iters;
}

template <int SG>
static double run_v6(const Shape &s, const float *A, const float *B, float *C, int iters) {
    const int rt = (s.Mo + 127) / 128, ct = (s.F + 127) / 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { gemm_v6_kernel<SG><<<s.N * rt * ct, 256>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct); };
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

template <int PRE, int MINB>
static double run_v7(const Shape &s, const float *A, const float *B, float *C, int iters, float SB, const float *rowmax) {
    const int rt = (s.Mo + 127) / 128, ct = (s.F + 127) / 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { gemm_v7_kernel<PRE, MINB><<<s.N * rt * ct, 256>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct, SB, rowmax); };
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

// phase timing of the ping-pong kernel: durations of the multiply / stage phases of one wave per group
template <int PRIO>
static void time_v4(const Shape &s, const float *A, const float *B, float *C, const char *label) {
    const int rt = (s.Mo + 127) / 128, ct = (s.F + 127) / 128, T = rt * ct;
    const int lds = 2 * 3 * (128 * PITCH + 128 * PITCH);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_v4_kernel<PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = (s.N & 7) == 0 ? 8 * (((s.N >> 3) * T + 1) / 2) : (s.N * T + 1) / 2;
    unsigned long long *ts;
    hipMalloc(&ts, 2 * 64 * 3 * 8);
    hipMemset(ts, 0, 2 * 64 * 3 * 8);
    for (int i = 0; i < 3; ++i) gemm_v4_kernel<PRIO><<<grid, 512, lds>>>(A, B, C, s.N, s.Mo, s.K, s.F, rt, ct, ts);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * 64 * 3);
    hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%s: phase timing (s_memtime ticks), wave 0 of each group of one workgroup; phases 8..27\n", label);
    for (int g = 0; g < 2; ++g) {
        double w[2] = {0, 0}, b[2] = {0, 0};
        int cnt[2] = {0, 0};
        for (int ph = 8; ph < 28; ++ph) {
            const unsigned long long *o = &h[((size_t)g * 64 + ph) * 3];
            const int kind = ((ph - g) & 1);                 // 1 = multiply, 0 = stage
            w[kind] += (double)(o[1] - o[0]);
            b[kind] += (double)(o[2] - o[1]);
            ++cnt[kind];
        }
        printf("  group %d: multiply phase work %.0f + barrier wait %.0f ticks; stage phase work %.0f + barrier wait %.0f ticks\n", g,
               w[1] / cnt[1], b[1] / cnt[1], w[0] / cnt[0], b[0] / cnt[0]);
    }
    const unsigned long long *o0 = &h[(0 * 64 + 8) * 3], *o1 = &h[(0 * 64 + 28) * 3];
    printf("  20 phases of group 0: %.0f ticks per phase\n", (double)(o1[0] - o0[0]) / 20.0);
    hipFree(ts);
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
        peak_ticks<0>(256, 20000, o); peak_ticks<1>(256, 20000, o); peak_ticks<0>(512, 10000, o); peak_ticks<1>(512, 10000, o);
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
    if (argc > 1 && !strcmp(argv[1], "v7")) {
        const char *vn[] = {"v2 128x128 interleaved", "v7 f16x3 no pre-pass", "v7 f16x3 + row scales", "v7 + row maxima given"};
        constexpr int NV7 = 4;
        printf("%-22s", "shape (N Mo K F)");
        for (int i = 0; i < NV7; ++i) printf(" %24s", vn[i]);
        printf("\n");
        for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 768, 512}, {16, 862, 512, 512}, {16, 862, 512, 256},
                                                 {16, 1723, 256, 256}, {16, 1723, 512, 128}, {16, 3445, 128, 128}, {16, 3445, 256, 128},
                                                 {16, 6890, 128, 128}}) {
            std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
            fill(hA, 7, 1.0f);
            fill(hB, 100, 0.05f);
            // rows of very different magnitude (the per-row scales must absorb them)
            for (int rr = 0; rr < s.N * s.Mo; ++rr) {
                const float sc = ldexpf(1.f, -((rr * 7) % 23));
                for (int k = 0; k < s.K; ++k) hA[(size_t)rr * s.K + k] *= sc;
            }
            float bmax = 0.f;
            for (float v : hB) bmax = fmaxf(bmax, fabsf(v));
            int be;
            frexpf(bmax, &be);                                  // bmax in [2^(be-1), 2^be)
            const float SB = ldexpf(1.f, 14 - be);              // -> [2^13, 2^14)
            float *A, *B, *C;
            hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)s.N * s.Mo * s.F * 4);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
            const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
            double us[NV7], emax[NV7], erms[NV7], f32rms = 0;
            auto chk = [&](int i) {
                check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax[i], erms[i], f32rms);
                double em2, er2, f2;
                check(s, hA, hB, C, 0, 120, 136, em2, er2, f2);
                if (er2 > erms[i]) erms[i] = er2;
            };
            auto clr = [&]() { hipMemset(C, 0xFF, (size_t)s.N * s.Mo * s.F * 4); };
            clr(); us[0] = run_v2<128, 128, 2, 2, 2, false, false, false, false, 1>(s, A, B, C, iters); chk(0);
            clr(); us[1] = run_v7<0, 2>(s, A, B, C, iters, SB); chk(1);
            clr(); us[2] = run_v7<1, 2>(s, A, B, C, iters, SB); chk(2);
            // row maxima as a producer kernel would deliver them (here from the host)
            std::vector<float> hR((size_t)s.N * s.Mo);
            for (size_t rr = 0; rr < hR.size(); ++rr) {
                float m = 0.f;
                for (int k = 0; k < s.K; ++k) m = fmaxf(m, fabsf(hA[rr * s.K + k]));
                hR[rr] = m;
            }
            float *R;
            hipMalloc(&R, hR.size() * 4);
            hipMemcpy(R, hR.data(), hR.size() * 4, hipMemcpyHostToDevice);
            clr(); us[3] = run_v7<2, 2>(s, A, B, C, iters, SB, R); chk(3);
            hipFree(R);
            char name[64];
            snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
            printf("%-22s", name);
            for (int i = 0; i < NV7; ++i) printf(" %14.1fus %5.1fTF", us[i], fl / us[i] / 1e6);
            printf("\n%-22s", "  rms err/rms(ref)");
            for (int i = 0; i < NV7; ++i) printf(" %24.2e", erms[i]);
            printf("   fp32 fma chain: %.2e\n", f32rms);
            hipFree(A); hipFree(B); hipFree(C);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "v6")) {
        const char *vn[] = {"v2 128x128 (ref)", "v2 128x128 interleaved", "v6 k16 pipeline", "v6 + IGroupLP pipeline"};
        constexpr int NV6 = 4;
        printf("%-22s", "shape (N Mo K F)");
        for (int i = 0; i < NV6; ++i) printf(" %24s", vn[i]);
        printf("\n");
        for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 768, 512}, {16, 862, 512, 512}, {16, 862, 512, 256},
                                                 {16, 1723, 256, 256}, {16, 1723, 512, 128}, {16, 3445, 128, 128}, {16, 3445, 256, 128}}) {
            std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
            fill(hA, 7, 1.0f);
            fill(hB, 100, 0.05f);
            float *A, *B, *C;
            hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)s.N * s.Mo * s.F * 4);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
            const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
            double us[NV6], emax[NV6], erms[NV6], f32rms = 0;
            auto chk = [&](int i) {
                check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax[i], erms[i], f32rms);
                double em2, er2, f2;
                check(s, hA, hB, C, 0, 120, 136, em2, er2, f2);
                if (er2 > erms[i]) erms[i] = er2;
            };
            auto clr = [&]() { hipMemset(C, 0xFF, (size_t)s.N * s.Mo * s.F * 4); };
            clr(); us[0] = run_v2<128, 128, 2, 2, 2, false, false>(s, A, B, C, iters); chk(0);
            clr(); us[1] = run_v2<128, 128, 2, 2, 2, false, false, false, false, 1>(s, A, B, C, iters); chk(1);
            clr(); us[2] = run_v6<0>(s, A, B, C, iters); chk(2);
            clr(); us[3] = run_v6<1>(s, A, B, C, iters); chk(3);
            char name[64];
            snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
            printf("%-22s", name);
            for (int i = 0; i < NV6; ++i) printf(" %14.1fus %5.1fTF", us[i], fl / us[i] / 1e6);
            printf("\n%-22s", "  rms err/rms(ref)");
            for (int i = 0; i < NV6; ++i) printf(" %24.2e", erms[i]);
            printf("   fp32 fma chain: %.2e\n", f32rms);
            hipFree(A); hipFree(B); hipFree(C);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "ilv")) {
        const char *vn[] = {"v2 128x128 (ref)", "v2 128x128 interleaved", "v2 64x64 occ5 (ref)", "v2 64x64 interleaved", "128x128 ilv 2 (4 up front)"};
        constexpr int NVI = 5;
        printf("%-22s", "shape (N Mo K F)");
        for (int i = 0; i < NVI; ++i) printf(" %24s", vn[i]);
        printf("\n");
        for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 768, 512}, {16, 862, 512, 512}, {16, 862, 512, 256},
                                                 {16, 1723, 256, 256}, {16, 1723, 512, 128}, {16, 3445, 128, 128}, {16, 3445, 256, 128},
                                                 {16, 3445, 192, 64}, {16, 6890, 64, 64}}) {
            std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
            fill(hA, 7, 1.0f);
            fill(hB, 100, 0.05f);
            float *A, *B, *C;
            hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)s.N * s.Mo * s.F * 4);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
            const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
            double us[NVI], emax[NVI], erms[NVI], f32rms = 0;
            auto chk = [&](int i) { check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax[i], erms[i], f32rms); };
            auto clr = [&]() { hipMemset(C, 0xFF, (size_t)s.N * s.Mo * s.F * 4); };
            clr(); us[0] = run_v2<128, 128, 2, 2, 2, false, false>(s, A, B, C, iters); chk(0);
            clr(); us[1] = run_v2<128, 128, 2, 2, 2, false, false, false, false, 1>(s, A, B, C, iters); chk(1);
            clr(); us[2] = run_v2<64, 64, 2, 2, 5, false, false>(s, A, B, C, iters); chk(2);
            clr(); us[3] = run_v2<64, 64, 2, 2, 5, false, false, false, false, 1>(s, A, B, C, iters); chk(3);
            clr(); us[4] = run_v2<128, 128, 2, 2, 2, false, false, false, false, 2>(s, A, B, C, iters); chk(4);
            char name[64];
            snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
            printf("%-22s", name);
            for (int i = 0; i < NVI; ++i) printf(" %14.1fus %5.1fTF", us[i], fl / us[i] / 1e6);
            printf("\n%-22s", "  rms err/rms(ref)");
            for (int i = 0; i < NVI; ++i) printf(" %24.2e", erms[i]);
            printf("   fp32 fma chain: %.2e\n", f32rms);
            hipFree(A); hipFree(B); hipFree(C);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "v5")) {
        const char *vn[] = {"v2 128x128 trunc (ref)", "v5 producer/consumer", "v5 + presplit weights", "v5i interleaved reads", "v5i + presplit", "v5i 8 prod + presplit"};
        constexpr int NV5 = 6;
        printf("%-22s", "shape (N Mo K F)");
        for (int i = 0; i < NV5; ++i) printf(" %24s", vn[i]);
        printf("\n");
        for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 768, 512}, {16, 862, 512, 512}, {16, 862, 512, 256},
                                                 {16, 1723, 256, 256}, {16, 1723, 512, 128}, {16, 3445, 128, 128}, {16, 3445, 256, 128}}) {
            std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
            fill(hA, 7, 1.0f);
            fill(hB, 100, 0.05f);
            float *A, *B, *C;
            hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)s.N * s.Mo * s.F * 4);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
            const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
            double us[NV5], emax[NV5], erms[NV5], f32rms = 0;
            auto chk = [&](int i) {
                check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax[i], erms[i], f32rms);
                double em2, er2, f2;
                check(s, hA, hB, C, 0, 120, 136, em2, er2, f2);          // rows across the first tile boundary of sample 0
                if (er2 > erms[i]) erms[i] = er2;
            };
            auto clr = [&]() { hipMemset(C, 0xFF, (size_t)s.N * s.Mo * s.F * 4); };
            clr(); us[0] = run_v2<128, 128, 2, 2, 2, false, false>(s, A, B, C, iters); chk(0);
            clr(); us[1] = run_v5<0>(s, A, B, C, iters); chk(1);
            clr(); us[2] = run_v5<1>(s, A, B, C, iters); chk(2);
            clr(); us[3] = run_v5<64>(s, A, B, C, iters); chk(3);
            clr(); us[4] = run_v5<65>(s, A, B, C, iters); chk(4);
            clr(); us[5] = run_v5<73>(s, A, B, C, iters); chk(5);
            if (s.K == 1024) {
                run_v5<2>(s, A, B, C, 2);
                run_v5<3>(s, A, B, C, 2);
                run_v5<6>(s, A, B, C, 2);
                run_v5<10>(s, A, B, C, 2);
                run_v5<11>(s, A, B, C, 2);
                printf("  interleaved reads (1 per 2 MFMAs): base / presplit / 8 producers + presplit / producers idle:\n");
                run_v5<2 | 64>(s, A, B, C, 2);
                run_v5<3 | 64>(s, A, B, C, 2);
                run_v5<11 | 64>(s, A, B, C, 2);
                run_v5<2 | 16 | 64>(s, A, B, C, 2);
                printf("  floors: producers idle (consumer reads stale LDS) / consumer without fragment reads / both:\n");
                run_v5<2 | 16>(s, A, B, C, 2);
                run_v5<2 | 32>(s, A, B, C, 2);
                run_v5<2 | 16 | 32>(s, A, B, C, 2);
            }
            char name[64];
            snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
            printf("%-22s", name);
            for (int i = 0; i < NV5; ++i) printf(" %14.1fus %5.1fTF", us[i], fl / us[i] / 1e6);
            printf("\n%-22s", "  rms err/rms(ref)");
            for (int i = 0; i < NV5; ++i) printf(" %24.2e", erms[i]);
            printf("   fp32 fma chain: %.2e\n", f32rms);
            hipFree(A); hipFree(B); hipFree(C);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "v4")) {
        const char *vn[] = {"v2 128x128 trunc (ref)", "v4 ping-pong", "v4 ping-pong + setprio", "v4 pp + ordered reads", "v4 pp + ordered + prio"};
        constexpr int NV4 = 5;
        printf("%-22s", "shape (N Mo K F)");
        for (int i = 0; i < NV4; ++i) printf(" %24s", vn[i]);
        printf("\n");
        for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 768, 512}, {16, 862, 512, 512}, {16, 862, 512, 256},
                                                 {16, 1723, 256, 256}, {16, 1723, 512, 128}, {16, 3445, 128, 128}, {16, 3445, 256, 128}}) {
            std::vector<float> hA((size_t)s.N * s.Mo * s.K), hB((size_t)s.F * s.K);
            fill(hA, 7, 1.0f);
            fill(hB, 100, 0.05f);
            float *A, *B, *C;
            hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)s.N * s.Mo * s.F * 4);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
            const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
            double us[NV4], emax[NV4], erms[NV4], f32rms = 0;
            auto chk = [&](int i) {
                check(s, hA, hB, C, s.N - 1, s.Mo - 24, s.Mo, emax[i], erms[i], f32rms);
                double em2, er2, f2;
                check(s, hA, hB, C, 0, 120, 136, em2, er2, f2);          // rows across the first tile boundary of sample 0
                if (er2 > erms[i]) erms[i] = er2;
            };
            auto clr = [&]() { hipMemset(C, 0xFF, (size_t)s.N * s.Mo * s.F * 4); };
            clr(); us[0] = run_v2<128, 128, 2, 2, 2, false, false>(s, A, B, C, iters); chk(0);
            clr(); us[1] = run_v4<0>(s, A, B, C, iters); chk(1);
            clr(); us[2] = run_v4<1>(s, A, B, C, iters); chk(2);
            clr(); us[3] = run_v4<8>(s, A, B, C, iters); chk(3);
            clr(); us[4] = run_v4<9>(s, A, B, C, iters); chk(4);
            if (s.K == 1024) {
                time_v4<2>(s, A, B, C, "ping-pong");
                time_v4<6>(s, A, B, C, "solo (second group idle)");
                time_v4<10>(s, A, B, C, "ping-pong + ordered reads");
                time_v4<14>(s, A, B, C, "solo + ordered reads");
            }
            char name[64];
            snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
            printf("%-22s", name);
            for (int i = 0; i < NV4; ++i) printf(" %14.1fus %5.1fTF", us[i], fl / us[i] / 1e6);
            printf("\n%-22s", "  rms err/rms(ref)");
            for (int i = 0; i < NV4; ++i) printf(" %24.2e", erms[i]);
            printf("   fp32 fma chain: %.2e\n", f32rms);
            hipFree(A); hipFree(B); hipFree(C);
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
