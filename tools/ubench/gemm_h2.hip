// Experiment (round 4): the fp32 contraction of chebyshev5 (reference lib/models.py:99-102) as THREE fp16 MFMA products per
// multiply-add on two-piece operands  x * s = hi + lo  (both pieces round-to-nearest fp16, s a power of two per A row / per
// weight tensor), with the WEIGHT pieces split once per step outside the kernel (planes [F][K] of fp16) and brought into LDS
// by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU, no ds_write), and the ACTIVATION operand either
//   AMODE 1: fp32 in HBM, split while staged, row maxima given (what the producing kernel writes next to the tensor), or
//   AMODE 0: pre-split planes as well (pure three-product fp16 GEMM: the ceiling of this tile structure).
// Round 3's v7 (tools/ubench/gemm_bf16x3.hip) split BOTH operands in the kernel: 1.2-1.35x over the six-product bf16 form
// on the wide layers.  The non-MFMA instructions per MFMA decide whether the matrix pipe can stay busy (MI355X_MICROARCH:
// <= 5 single-issue instructions hide behind one 32-cycle MFMA): v7 issues ~8.6 per MFMA, AMODE 1 ~4, AMODE 0 ~1.8.
// Tile 128 x 128, 4 waves as 2 x 2, k32 chunks, two LDS stages of 32 KB (A_hi, A_lo, B_hi, B_lo: 128 rows x 64 B each, the
// 16-byte segments XOR-swizzled by (row >> 2) & 3 -- applied on the SOURCE address of the DMA lanes), ONE barrier per chunk.
//   hipcc --offload-arch=gfx950 -O3 gemm_h2.hip -o gemm_h2 && ./gemm_h2 [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bitsf(unsigned v) { return __builtin_bit_cast(float, v); }

// all tiles of a sample on one XCD (block b runs on XCD b % 8), as the library's cape_map_block
__device__ __forceinline__ void map_block(int b, int N, int tiles, int &n, int &t) {
    const int per = tiles;                                    // tiles per sample
    const int x = b & 7, j = b >> 3;                          // XCD, index within the XCD
    const int ns = (N + 7) / 8;                               // samples per XCD (N % 8 == 0 here)
    n = x * ns + j / per;
    t = j % per;
    if (n >= N) { n = N - 1; t = -1; }
}

__device__ __forceinline__ void split2h(float x0, float x1, unsigned &hi, unsigned &lo) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    half2v H = {h0, h1}, L = {l0, l1};
    hi = __builtin_bit_cast(unsigned, H);
    lo = __builtin_bit_cast(unsigned, L);
}

// one 1 KB LDS-DMA piece: lane L's 16 bytes land at lds_base + 16 L (M0 = wave-uniform base)
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int BM = 128, BN = 128, KC = 32;
constexpr int PLANE = 128 * 64;                  // bytes of one piece plane of one operand in a stage
constexpr int STAGE = 4 * PLANE;                 // A_hi, A_lo, B_hi, B_lo

template <int AMODE, int DIAG>        // DIAG 0: full kernel; 1: no loads after the prologue (MFMA + LDS reads only); 2: no MFMAs (loads + staging only)
__global__ __launch_bounds__(256, 2) void gemm_h2_kernel(const float *__restrict__ A, const _Float16 *__restrict__ Ah,
                                                            const _Float16 *__restrict__ Al, const float *__restrict__ rowmax,
                                                            const float *__restrict__ ascale,
                                                            const _Float16 *__restrict__ Bh, const _Float16 *__restrict__ Bl,
                                                            float *__restrict__ C, int N, int Mo, int K, int F, int row_tiles,
                                                            int col_tiles, float invSB) {
    constexpr int TM = 2, TN = 2;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    __shared__ float inv_scale[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    int n, t;
    map_block(blockIdx.x, N, row_tiles * col_tiles, n, t);
    if (t < 0) return;
    const int r0 = (t / col_tiles) * BM, f0 = (t % col_tiles) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    // ---- DMA lanes: piece i = wave * 4 + j covers 16 rows x 64 B of one plane; lane L -> row L / 4, LDS slot L % 4,
    //      source segment (L % 4) ^ ((row >> 2) & 3)
    const int drow = lane >> 2, dseg = (lane & 3) ^ ((drow >> 2) & 3);
    const unsigned lds0 = (unsigned)(size_t)smem;           // LDS byte address of the stage ring
    const _Float16 *bsrc[4];
    unsigned bdst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = wave * 4 + j, plane = i >> 3, rb = i & 7;
        const int f = min(f0 + rb * 16 + drow, F - 1);
        // AMODE 3: the weight planes are stored chunk-major, [K/32][F][32]: the 128 x 64 B of a tile's chunk are 8 KB contiguous
        bsrc[j] = (plane ? Bl : Bh) + (AMODE == 3 ? (long long)f * 32 + 8 * dseg : (long long)f * K + 8 * dseg);
        bdst[j] = lds0 + (2 + plane) * PLANE + rb * 1024;
    }
    const _Float16 *asrc[4];
    unsigned adst[4];
    if (AMODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave * 4 + j, plane = i >> 3, rb = i & 7;
            const int r = min(r0 + rb * 16 + drow, Mo - 1);
            asrc[j] = (plane ? Al : Ah) + ((long long)n * Mo + r) * K + 8 * dseg;
            adst[j] = lds0 + plane * PLANE + rb * 1024;
        }
    }
    // ---- register staging of fp32 A (AMODE 1): rows r + 64 i, eight consecutive k = segment q
    const int q = tid & 3, r = tid >> 2;
    const float *ap[2];
    float sa[2] = {1.f, 1.f};
    if (AMODE >= 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = min(r0 + r + 64 * i, Mo - 1);
            ap[i] = A + ((long long)n * Mo + row) * K + 8 * q;
            const float m = rowmax[(long long)n * Mo + row];
            int e = (int)((fbits(m) >> 23) & 255);
            e = max(e, 14);
            sa[i] = bitsf((unsigned)(267 - e) << 23);               // 2^(13 - (e - 127)): row maximum -> [2^13, 2^14)
            if (q == 0) inv_scale[r + 64 * i] = bitsf((unsigned)(e - 13) << 23) * invSB;
        }
    } else if (tid < BM) {
        inv_scale[tid] = invSB / ascale[(long long)n * Mo + min(r0 + tid, Mo - 1)];
    }
    const int total = K / KC;
    float4 ra[2][2], rb2[2][2];           // rb2: second register set of the two-chunk-deep A prefetch (AMODE 2)
    auto dma = [&](int c, int buf) {
        const int k0 = KC * c;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(bsrc[j] + (AMODE == 3 ? (long long)c * F * 32 : (long long)k0), bdst[j] + buf * STAGE);
        if (AMODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(asrc[j] + k0, adst[j] + buf * STAGE);
        }
    };
    // AMODE 2 reads A through a buffer resource: per-lane byte offset fixed for the whole kernel, the chunk offset in an SGPR --
    // no per-chunk VGPR address arithmetic, hence no register the compiler would have to protect with an early s_waitcnt
    // (with flat addresses it recomputes them into registers that alias a pending load and waits vmcnt(1) BEFORE issuing the
    // next chunk's loads: that wait also covers the DMA pieces just issued and serialises the two streams)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(A + (long long)n * Mo * K), 0, Mo * K * 4, 0x00020000);
    int avoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) avoff[i] = (min(r0 + r + 64 * i, Mo - 1) * K + 8 * q) * 4;
    const int r8 = tid >> 3, q8 = tid & 7;                    // AMODE 3: rows r8 + 32 i, k = 4 q8 .. 4 q8 + 3: 8 lanes = one 128-byte line
    int avoff3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) avoff3[i] = (min(r0 + r8 + 32 * i, Mo - 1) * K + 4 * q8) * 4;
    float sa3[4] = {1.f, 1.f, 1.f, 1.f};
    if (AMODE == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(r0 + r8 + 32 * i, Mo - 1);
            const float m = rowmax[(long long)n * Mo + row];
            int e = (int)((fbits(m) >> 23) & 255);
            e = max(e, 14);
            sa3[i] = bitsf((unsigned)(267 - e) << 23);
            if (q8 == 0) inv_scale[r8 + 32 * i] = bitsf((unsigned)(e - 13) << 23) * invSB;
        }
    }
    auto load_a = [&](int c, float4 (&ra)[2][2]) {
        if (AMODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ra[i >> 1][i & 1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, avoff3[i], KC * c * 4, 0));
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (AMODE == 2) {
                ra[i][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, avoff[i], KC * c * 4, 0));
                ra[i][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, avoff[i] + 16, KC * c * 4, 0));
            } else {
                ra[i][0] = *reinterpret_cast<const float4 *>(ap[i] + KC * c);
                ra[i][1] = *reinterpret_cast<const float4 *>(ap[i] + KC * c + 4);
            }
        }
    };
    auto store_a = [&](int buf, const float4 (&ra)[2][2]) {
        if (AMODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = r8 + 32 * i;
                const float4 v = ra[i >> 1][i & 1];
                const float s = sa3[i];
                uint2 hi, lo;
                split2h(v.x * s, v.y * s, hi.x, lo.x);
                split2h(v.z * s, v.w * s, hi.y, lo.y);
                unsigned char *d = smem + buf * STAGE + row * 64 + 16 * ((q8 >> 1) ^ ((row >> 2) & 3)) + 8 * (q8 & 1);
                *reinterpret_cast<uint2 *>(d) = hi;
                *reinterpret_cast<uint2 *>(d + PLANE) = lo;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r + 64 * i;
            uint4 hi, lo;
            const float s = sa[i];
            split2h(ra[i][0].x * s, ra[i][0].y * s, hi.x, lo.x);
            split2h(ra[i][0].z * s, ra[i][0].w * s, hi.y, lo.y);
            split2h(ra[i][1].x * s, ra[i][1].y * s, hi.z, lo.z);
            split2h(ra[i][1].z * s, ra[i][1].w * s, hi.w, lo.w);
            unsigned char *d = smem + buf * STAGE + row * 64 + 16 * (q ^ ((row >> 2) & 3));
            *reinterpret_cast<uint4 *>(d) = hi;
            *reinterpret_cast<uint4 *>(d + PLANE) = lo;
        }
    };
    const int fsw = (li >> 2) & 3;                               // swizzle term of this lane's fragment rows (tile offsets are multiples of 32)
    auto compute = [&](int buf) {
        const unsigned char *pa = smem + buf * STAGE + (wm * 64 + li) * 64;
        const unsigned char *pb = smem + buf * STAGE + 2 * PLANE + (wn * 64 + li) * 64;
        half8 af[2][TM][2], bf[2][TN][2];
        auto rd = [&](int ks) {
            const int so = 16 * ((2 * ks + lh) ^ fsw);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p) af[ks][a][p] = *reinterpret_cast<const half8 *>(pa + p * PLANE + a * 32 * 64 + so);
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[ks][b][p] = *reinterpret_cast<const half8 *>(pb + p * PLANE + b * 32 * 64 + so);
        };
        auto mm = [&](int ks) {
#pragma unroll
            for (int term = 0; term < 3; ++term)                     // lo*hi, hi*lo, hi*hi: small products first
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][a][term == 0 ? 1 : 0], bf[ks][b][term == 1 ? 1 : 0],
                                                                           acc[a][b], 0, 0, 0);
        };
        if (DIAG == 2) return;
        rd(0);
        __builtin_amdgcn_sched_barrier(0);
        rd(1);
        mm(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {                              // the 8 reads of step 1 ride between the 12 MFMAs of step 0
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
    };

    if (AMODE >= 2) {
        // A two chunks ahead in registers (the long-latency stream: activations come from HBM / the infinity cache), the
        // weight pieces one chunk ahead by DMA (L2-resident).  vmcnt retires in order: the DMA of chunk it+1 is issued BEFORE
        // the A loads of chunk it+2, so "all but the last four" covers it and leaves chunk it+2 in flight across the barrier.
        dma(0, 0);
        load_a(0, ra);
        if (total > 1) load_a(1, rb2);
        store_a(0, ra);
        if (total > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // steady = true: chunks it+1 and it+2 exist -- no branches in the loop body, so the compiler's own vmcnt bookkeeping for
        // the register loads is exact (with the conditions inside, the merge of the paths made it wait for the newest loads too)
        auto body = [&](int it, float4 (&rfree)[2][2], const float4 (&rnext)[2][2], auto steady) {
            constexpr bool ST = decltype(steady)::value;
            const int buf = it & 1;
            const bool m1 = ST || (it + 1 < total && DIAG != 1), m2 = ST || (it + 2 < total && DIAG != 1);
            if (m1) dma(it + 1, buf ^ 1);
            if (m2) load_a(it + 2, rfree);
            compute(buf);
            if (m1) store_a(buf ^ 1, rnext);
            if (m2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        int it = 0;
        if (DIAG != 1)
            for (; it + 3 < total; it += 2) {
                body(it, ra, rb2, std::true_type{});
                body(it + 1, rb2, ra, std::true_type{});
            }
        for (; it < total; it += 2) {
            body(it, ra, rb2, std::false_type{});
            if (it + 1 < total) body(it + 1, rb2, ra, std::false_type{});
        }
    } else {
    dma(0, 0);
    if (AMODE == 1) { load_a(0, ra); store_a(0, ra); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < total && DIAG != 1;
        if (more) {
            dma(it + 1, buf ^ 1);
            if (AMODE == 1) load_a(it + 1, ra);
        }
        compute(buf);
        if (more && AMODE == 1) store_a(buf ^ 1, ra);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    }

    float *cn = C + (long long)n * Mo * F;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = f0 + wn * 64 + b * 32 + li;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int rl = wm * 64 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                const int row = r0 + rl;
                if (row < Mo && col < F) cn[(long long)row * F + col] = acc[a][b][g] * inv_scale[rl];
            }
        }
}

struct Shape { int N, Mo, K, F; };

static void fill(std::vector<float> &v, unsigned seed, float scale) {
    unsigned s = seed * 2654435761u + 12345u;
    for (auto &x : v) {
        s = s * 1664525u + 1013904223u;
        const float u = ((s >> 8) & 0xFFFF) / 65536.f, w = ((s >> 3) & 0xFFF) / 4096.f;
        x = scale * (u - 0.5f) * 3.4f + scale * 1e-4f * w;         // full-mantissa values
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    printf("%-20s %22s %22s %22s %22s %22s %22s\n", "shape (N Mo K F)", "A fp32 split in kernel", "A pre-split (pure)", "split, A 2 chunks ahead", "(same): no MFMA", "+ full-line loads", "(same): no MFMA");
    for (const Shape &s : std::vector<Shape>{{16, 862, 1024, 512}, {16, 862, 768, 512}, {16, 862, 512, 512}, {16, 862, 512, 256},
                                             {16, 1723, 256, 256}, {16, 1723, 512, 128}, {16, 3445, 128, 128}, {16, 3445, 256, 128},
                                             {16, 6890, 128, 128}}) {
        const size_t na = (size_t)s.N * s.Mo * s.K, nb = (size_t)s.F * s.K, nc = (size_t)s.N * s.Mo * s.F, nr = (size_t)s.N * s.Mo;
        std::vector<float> hA(na), hB(nb), hR(nr), hS(nr);
        fill(hA, 7, 1.0f);
        fill(hB, 100, 0.05f);
        for (size_t rr = 0; rr < nr; ++rr) {                         // rows of very different magnitude
            const float sc = ldexpf(1.f, -(int)((rr * 7) % 23));
            float m = 0.f;
            for (int k = 0; k < s.K; ++k) { hA[rr * s.K + k] *= sc; m = fmaxf(m, fabsf(hA[rr * s.K + k])); }
            hR[rr] = m;
            int e;
            frexpf(m, &e);
            hS[rr] = ldexpf(1.f, 14 - e);                            // row maximum -> [2^13, 2^14)
        }
        float bmax = 0.f;
        for (float v : hB) bmax = fmaxf(bmax, fabsf(v));
        int be;
        frexpf(bmax, &be);
        const float SB = ldexpf(1.f, 14 - be);
        std::vector<_Float16> hAh(na), hAl(na), hBh(nb), hBl(nb);
        for (size_t i = 0; i < na; ++i) {
            const float v = hA[i] * hS[i / s.K];
            hAh[i] = (_Float16)v;
            hAl[i] = (_Float16)(v - (float)hAh[i]);
        }
        for (size_t i = 0; i < nb; ++i) {
            const float v = hB[i] * SB;
            hBh[i] = (_Float16)v;
            hBl[i] = (_Float16)(v - (float)hBh[i]);
        }
        std::vector<_Float16> hBh2(nb), hBl2(nb);                  // chunk-major [K/32][F][32]
        for (int f = 0; f < s.F; ++f)
            for (int k = 0; k < s.K; ++k) {
                const size_t d = ((size_t)(k / 32) * s.F + f) * 32 + (k % 32);
                hBh2[d] = hBh[(size_t)f * s.K + k];
                hBl2[d] = hBl[(size_t)f * s.K + k];
            }
        float *A, *R, *S, *C;
        _Float16 *Ah, *Al, *Bh, *Bl, *Bh2, *Bl2;
        hipMalloc(&Bh2, nb * 2); hipMalloc(&Bl2, nb * 2);
        hipMemcpy(Bh2, hBh2.data(), nb * 2, hipMemcpyHostToDevice); hipMemcpy(Bl2, hBl2.data(), nb * 2, hipMemcpyHostToDevice);
        hipMalloc(&A, na * 4); hipMalloc(&R, nr * 4); hipMalloc(&S, nr * 4); hipMalloc(&C, nc * 4);
        hipMalloc(&Ah, na * 2); hipMalloc(&Al, na * 2); hipMalloc(&Bh, nb * 2); hipMalloc(&Bl, nb * 2);
        hipMemcpy(A, hA.data(), na * 4, hipMemcpyHostToDevice);
        hipMemcpy(R, hR.data(), nr * 4, hipMemcpyHostToDevice);
        hipMemcpy(S, hS.data(), nr * 4, hipMemcpyHostToDevice);
        hipMemcpy(Ah, hAh.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(Al, hAl.data(), na * 2, hipMemcpyHostToDevice);
        hipMemcpy(Bh, hBh.data(), nb * 2, hipMemcpyHostToDevice); hipMemcpy(Bl, hBl.data(), nb * 2, hipMemcpyHostToDevice);
        const int rt = (s.Mo + 127) / 128, ct = (s.F + 127) / 128;
        const dim3 grid(s.N * rt * ct);
        const double fl = 2.0 * s.N * s.Mo * (double)s.K * s.F;
        double us[6], err[6];
        std::vector<float> hC(nc);
        auto run = [&](int v) {
            auto launch = [&]() {
                if (v == 0) gemm_h2_kernel<1, 0><<<grid, 256>>>(A, Ah, Al, R, S, Bh, Bl, C, s.N, s.Mo, s.K, s.F, rt, ct, 1.f / SB);
                if (v == 1) gemm_h2_kernel<0, 0><<<grid, 256>>>(A, Ah, Al, R, S, Bh, Bl, C, s.N, s.Mo, s.K, s.F, rt, ct, 1.f / SB);
                if (v == 2) gemm_h2_kernel<2, 0><<<grid, 256>>>(A, Ah, Al, R, S, Bh, Bl, C, s.N, s.Mo, s.K, s.F, rt, ct, 1.f / SB);
                if (v == 3) gemm_h2_kernel<2, 2><<<grid, 256>>>(A, Ah, Al, R, S, Bh, Bl, C, s.N, s.Mo, s.K, s.F, rt, ct, 1.f / SB);
                if (v == 4) gemm_h2_kernel<3, 0><<<grid, 256>>>(A, Ah, Al, R, S, Bh2, Bl2, C, s.N, s.Mo, s.K, s.F, rt, ct, 1.f / SB);
                if (v == 5) gemm_h2_kernel<3, 2><<<grid, 256>>>(A, Ah, Al, R, S, Bh2, Bl2, C, s.N, s.Mo, s.K, s.F, rt, ct, 1.f / SB);
            };
            hipMemset(C, 0xFF, nc * 4);
            launch();
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            us[v] = 1e3 * ms / iters;
            hipMemcpy(hC.data(), C, nc * 4, hipMemcpyDeviceToHost);
            // rms error relative to the rms of the exact result, over sampled rows of the first / last sample
            double se = 0, sr = 0;
            for (int pick = 0; pick < 48; ++pick) {
                const int nn = pick < 24 ? 0 : s.N - 1;
                const int row = pick < 24 ? (pick < 12 ? pick * 11 : 120 + pick) : s.Mo - 1 - (pick - 24);
                const float *ar = &hA[((size_t)nn * s.Mo + row) * s.K];
                double rs = 0, es = 0;
                for (int f = 0; f < s.F; ++f) {
                    double ref = 0;
                    for (int k = 0; k < s.K; ++k) ref += (double)ar[k] * (double)hB[(size_t)f * s.K + k];
                    const double d = (double)hC[((size_t)nn * s.Mo + row) * s.F + f] - ref;
                    rs += ref * ref; es += d * d;
                }
                se += es / (rs + 1e-300); sr += 1.0;             // every row weighs the same, whatever its magnitude
            }
            err[v] = sqrt(se / sr);
        };
        for (int v = 0; v < 6; ++v) run(v);
        char name[64];
        snprintf(name, sizeof name, "%d %d %d %d", s.N, s.Mo, s.K, s.F);
        printf("%-20s", name);
        for (int v = 0; v < 6; ++v) printf(" %8.1fus %6.1fTF %s", us[v], fl / us[v] / 1e6, "");
        printf("\n%-20s", "  rms err per row");
        for (int v = 0; v < 5; ++v) printf(" %22.2e", v == 3 ? 0.0 : err[v]);
        printf("\n");
        hipFree(A); hipFree(R); hipFree(S); hipFree(C); hipFree(Ah); hipFree(Al); hipFree(Bh); hipFree(Bl); hipFree(Bh2); hipFree(Bl2);
    }
    return 0;
}
