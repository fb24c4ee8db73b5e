// The two-piece fp16 contraction of gemm_h2.h (same arithmetic, same operands, same epilogues; reference lib/models.py:99-102)
// on a WIDE workgroup tile: 128 rows x 256 output columns, 512 threads = 8 waves as 2 (rows) x 4 (columns), 64 x 64 accumulator
// elements per wave, ONE workgroup per CU.  Single-accumulator launches only: the affine blocks' DUAL form keeps the 128 x 64
// kernel (with two accumulator sets AND both fragment generations live it does not fit the 256 registers of a 512-thread
// workgroup -- hipcc spilled ~1000 values in the prototype).
//
// Why (round 6, profiles/r06_l2_hits.json, profiles/r06_h2x_probe.txt): on the 128 x 128 kernel the counters show an L2 hit rate
// of 0.84 and an average L1 -> L2 read latency of 228 cycles -- the memory side is neither missing nor saturated -- while
// every one of the two workgroups that share a CU stages (loads, scales, splits, stores) its own copy of the SAME activation
// rows when the tiles are column neighbours, and after every barrier both waves of a SIMD wait for their first fragments with
// the matrix pipe idle.  Here
//   * the activation tile is staged once for 256 columns: half the register loads, half the split arithmetic and half the
//     LDS stores per MFMA, 0.75 x the operand bytes through L2 -> CU;
//   * THREE LDS stages of one k32 chunk each (3 x 48 KB): the weight pieces of chunk i + 2 arrive by LDS-DMA and the activation
//     pieces of chunk i + 2 are split and stored while chunk i multiplies, so that stage i + 1 is complete and visible one whole
//     chunk before it is multiplied -- its first k16 fragments are read BEFORE the barrier that ends chunk i, and the MFMAs of
//     chunk i + 1 start straight after the barrier instead of behind an LDS round trip;
//   * the two waves of a SIMD belong to the same workgroup and hit the same barrier, so whatever a wave does between its MFMAs
//     its partner does at the same time: a chunk is SIX groups of four MFMAs (one per accumulator: no dependent issue), pinned
//     in this order, and everything else -- fragment reads, DMA issue, activation loads, cursor arithmetic, the split and the
//     LDS stores -- is dealt over the groups, so that no stretch of the chunk is without MFMAs (the first version issued the
//     loads and their scalar bookkeeping in one block behind the barrier, ~45 instructions per wave with the pipe idle: 52.3 us
//     against 49.1 us for this order on 16 x 862 x 1024 -> 512);
//   * every load of the loop is inline asm and counted by hand (vmcnt retires in order): hipcc's own bookkeeping does not see the
//     DMA pieces, so its wait in front of a register load's first use would also drain DMA issued a few hundred cycles earlier.
// What the phase knock-outs of tools/ubench/h2x_probe.hip say about the rest (same shape, 224 workgroups, 32 chunks): MFMAs
// alone 21.7 us (the pipe at its clock-limited rate), + fragment reads 28, + loads, split and stores 37 for the loop; prologue
// 3.5-4.7 us; the output tile's store 6-7 us at the END of every workgroup at once -- bound by the HBM write path (28 MB), not by
// store issue: parking the tile in the dead stages and storing rows with 16-byte accesses took 9.9 us.
#pragma once
#include <stddef.h>
#include "gemm_h2.h"

namespace {

// Four 1 KB LDS-DMA pieces to consecutive LDS kilobytes in ONE statement: M0 (the LDS destination base) is written once and the
// pieces use the instruction offset 0 / 1024 / 2048 / 3072, which the hardware adds to BOTH the LDS address and the global
// address -- so piece j's per-lane source offset arrives with j * 1024 subtracted (gemm_h2x_kernel: bvoff).  Untracked by
// hipcc: completion is counted by hand (vmcnt).
__device__ __forceinline__ void h2x_glds4(const void *sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5\n\t"
                 "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, %5 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_dst) : "memory");
}

// KO: phase knock-outs for tools/ubench/h2x_probe.hip (0 in the library): 1 no MFMAs, 2 no weight DMA, 4 no activation loads,
// 8 no split + LDS stores, 16 no fragment reads, 32 no barriers, 64 no epilogue.  The results are then meaningless; only the
// time is read.  EXP: scheduling experiment of the probe (bit 1: the loads and their bookkeeping in one block behind the
// barrier, the first version's order).
template <int KO = 0, int EXP = 0>
__global__ __launch_bounds__(512, 2) void gemm_h2x_kernel(GconvParams p) {
    constexpr int BM = 128, BN = 256;
    constexpr int WTM = 64, WTN = 64, TM = 2, TN = 2;                       // per wave: 64 rows x 64 columns
    constexpr int APL = BM * H2_ROW, BPL = BN * H2_ROW;                      // bytes of one piece plane
    constexpr int STAGE = 2 * APL + 2 * BPL;                                 // 48 KB
    constexpr int PP = BN / 16;                                              // DMA pieces (16 rows x 64 B) per plane
    constexpr int BPW = 4;                                                   // pieces per wave and chunk (8 waves x 4 = 2 PP)
    static_assert(2 * PP == 8 * BPW, "tile");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * STAGE + BM * 4];
    float *inv_row = reinterpret_cast<float *>(smem + 3 * STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
    const int q = tid & 3, r = tid >> 2;                                     // A staging: row r, floats 8 q .. 8 q + 7 of the chunk

    int n, t;
    cape_map_block(blockIdx.x, p.N, p.row_tiles * p.col_tiles, n, t);
    const int r0 = (t / p.col_tiles) * BM;
    const int f0 = (t % p.col_tiles) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    // the source table is indexed by RUNTIME cursors: read it through the kernel-argument segment itself (scalar loads with a
    // register offset) -- dynamic indexing of the by-value parameter can make hipcc copy the whole block to scratch memory
    typedef const __attribute__((address_space(4))) SrcDev *SrcTab;
    const SrcTab src_tab = (SrcTab)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(GconvParams, s));
    int total = 0;
    for (int si = 0; si < p.nsrc; ++si) total += src_tab[si].C / H2_KC;

    const int rc = min(r0 + r, p.Mo - 1);
    float sa;
    // ---- weight DMA: a wave's four pieces lie in ONE piece plane (waves 0-3 hi, 4-7 lo): piece j covers 16 columns x 64 B of
    //      the chunk; lane -> column drow of the piece, 16-byte segment (lane & 3) ^ swizzle applied on the SOURCE address
    const int drow = lane >> 2, dseg = (lane & 3) ^ ((drow >> 2) & 3);
    const int wplane = (wave * BPW) / PP;
    const unsigned lds0 = (unsigned)(size_t)smem;
    int bcol[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) bcol[j] = min(f0 + ((wave * BPW) % PP + j) * 16 + drow, p.F - 1);
    // (the four pieces land in consecutive kilobytes of the wave's plane)
    const unsigned bdst = lds0 + 2 * APL + wplane * BPL + ((wave * BPW) % PP) * 1024;

    // ---- cursors over the chunk sequence: B = DMA of the weight pieces (two chunks ahead of the multiply), A = register loads
    //      of the activations (three ahead).  The fields of the current source live in registers between source changes.
    int b_si = 0, a_si = 0, b_left = 0, a_left = 0;      // chunks left in the current source
    unsigned bvoff[BPW];
    const unsigned short *b_ptr = nullptr;               // this wave's plane of the current source at the current chunk
    auto open_b = [&]() {
        const auto &S = src_tab[b_si];
        b_left = S.C / H2_KC;
        // (both planes are loaded and the choice is made on the VALUES by a mask, not between the two argument addresses)
        const unsigned long long mp = 0ull - (unsigned long long)wplane;
        const unsigned long long h1 = (unsigned long long)(size_t)S.wh, l1 = (unsigned long long)(size_t)S.wl;
        b_ptr = reinterpret_cast<const unsigned short *>((size_t)(h1 ^ ((h1 ^ l1) & mp)));
        const long long pitch = S.wp;
#pragma unroll
        for (int j = 0; j < BPW; ++j) bvoff[j] = (unsigned)(((long long)bcol[j] * pitch + 8 * dseg) * 2) - 1024u * j;
    };
    int avoff = 0, a_so = 0;
    h2_u32x4 adesc;
    auto open_a = [&]() {
        const auto &S = src_tab[a_si];
        a_left = S.C / H2_KC;
        a_so = 0;
        const unsigned long long ab = (unsigned long long)(size_t)(S.x + (long long)n * S.xs);
        adesc = h2_u32x4{(unsigned)ab, (unsigned)(ab >> 32) & 0xFFFFu, 0x7FFFFFFCu, 0x00020000u};
        avoff = (rc * S.ldx + 8 * q) * 4;
    };
    // Stages are addressed by their byte offset (0, STAGE, 2 STAGE); the roles cur / nxt / fre rotate through them.
    auto dma = [&](int so) {
        if constexpr (!(KO & 2)) h2x_glds4(b_ptr, bvoff[0], bvoff[1], bvoff[2], bvoff[3], bdst + so);
    };
    auto next_b = [&]() {
        b_ptr += H2_KC;
        if (--b_left == 0 && ++b_si < p.nsrc) open_b();
    };
    // (the destinations ARE the staging registers: a copy made before the data lands would leave the landing registers free for
    // the compiler to reuse)
    auto load_a = [&](f32x4 (&ra)[2]) {
        if constexpr (!(KO & 4))
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:16"
                         : "=&v"(ra[0]), "=&v"(ra[1]) : "v"(avoff), "s"(adesc), "s"(a_so) : "memory");
        else
            asm volatile("" : "+v"(ra[0]), "+v"(ra[1]) : "s"(a_so));
    };
    auto next_a = [&]() {
        a_so += H2_KC * 4;
        if (--a_left == 0 && ++a_si < p.nsrc) open_a();
    };
    // wait for a register set: `newer` = operations issued after its loads that may still be in flight (6, 2, or 0 = drain)
    auto wait_a = [&](f32x4 (&ra)[2], int newer) {
        if constexpr (KO & 4) return;
        if (newer == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(ra[0]), "+v"(ra[1]));
        else if (newer == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(ra[0]), "+v"(ra[1]));
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]));
    };
    uint4 sp_hi, sp_lo;
    auto split_a = [&](const f32x4 (&ra)[2], int half) {  // half 0: floats 0..3 of the row segment, 1: floats 4..7
        if constexpr (KO & 8) return;
        if (half == 0) {
            h2_split2s(ra[0][0], ra[0][1], sa, sp_hi.x, sp_lo.x);
            h2_split2s(ra[0][2], ra[0][3], sa, sp_hi.y, sp_lo.y);
        } else {
            h2_split2s(ra[1][0], ra[1][1], sa, sp_hi.z, sp_lo.z);
            h2_split2s(ra[1][2], ra[1][3], sa, sp_hi.w, sp_lo.w);
        }
    };
    unsigned char *const a_dst = smem + r * H2_ROW + 16 * (q ^ ((r >> 2) & 3));
    auto store_a = [&](int st) {
        if constexpr (KO & 8) return;
        *reinterpret_cast<uint4 *>(a_dst + st) = sp_hi;
        *reinterpret_cast<uint4 *>(a_dst + st + APL) = sp_lo;
    };

    // ---- fragments.  Lane (li, lh) of v_mfma_f32_32x32x16_f16 supplies row / column li and the contraction indices
    //      8 lh .. 8 lh + 7 of a k16 step: one 16-byte LDS read per operand piece.  Slots 0 .. 3: A (row block, piece),
    //      4 .. 7: B (column block, piece)
    const int fsw = (li >> 2) & 3;
    const unsigned char *const pa0 = smem + (wm * WTM + li) * H2_ROW;
    const unsigned char *const pb0 = smem + 2 * APL + (wn * WTN + li) * H2_ROW;
    struct Frag { h2_half8 a[TM][2], b[TN][2]; };
    auto rd = [&](Frag &F, int st, int ks, int s0 = 0, int s1 = 8) {
        if constexpr (KO & 16) return;
        const int so = st + 16 * ((2 * ks + lh) ^ fsw);
#pragma unroll
        for (int s = s0; s < s1; ++s) {
            const int x = (s >> 1) & 1, pc = s & 1;
            if (s < 4) F.a[x][pc] = *reinterpret_cast<const h2_half8 *>(pa0 + pc * APL + x * 32 * H2_ROW + so);
            else F.b[x][pc] = *reinterpret_cast<const h2_half8 *>(pb0 + pc * BPL + x * 32 * H2_ROW + so);
        }
    };
    auto mm = [&](const Frag &F, int t0 = 0, int t1 = 3) {
#pragma unroll
        for (int term = t0; term < t1; ++term)           // lo*hi, hi*lo, hi*hi: small products first
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const h2_half8 &fa = F.a[a][term == 0 ? 1 : 0], &fb = F.b[b][term == 1 ? 1 : 0];
                    if constexpr (KO & 1) asm volatile("" ::"v"(fa), "v"(fb));
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[a][b], 0, 0, 0);
                }
    };

    Frag F0, F1;
    f32x4 rA[2], rB[2];
    if constexpr (KO != 0) {                             // knocked-out producers leave these undefined: pin them to registers
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) { asm volatile("" : "=v"(F0.a[a][pc])); asm volatile("" : "=v"(F1.a[a][pc])); asm volatile("" : "=v"(F0.b[a][pc])); asm volatile("" : "=v"(F1.b[a][pc])); }
        asm volatile("" : "=v"(rA[0]), "=v"(rA[1]), "=v"(rB[0]), "=v"(rB[1]));
        sp_hi = sp_lo = make_uint4(0, 0, 0, 0);
    }
    // ---- one chunk: stage `cur` holds chunk it (its k16 step 0 fragments are in F0), stage `nxt` chunk it + 1 (complete and
    //      visible), `fre` is free; registers rsp hold the activations of chunk it + 2, rld receives chunk it + 3.
    //      steady: chunks it + 1 .. it + 3 exist (static wait counts, the grouped schedule); otherwise the simple order with
    //      conservative waits (at most four chunks per workgroup)
    auto chunk = [&](int it, int cur, int nxt, int fre, f32x4 (&rld)[2], f32x4 (&rsp)[2], auto steady) {
        constexpr bool ST = decltype(steady)::value;
        constexpr int NA = TM * TN;                      // MFMAs per group (one product term): one per accumulator
        if constexpr (ST && !(EXP & 1)) {
            // G0 .. G2: k16 step 0 from F0, the reads of step 1 (F1) between them; DMA of chunk it + 2 after G0, the register
            // loads of chunk it + 3 after G1
            rd(F1, cur, 1, 0, 3);
            mm(F0, 0, 1);
#pragma unroll
            for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            dma(fre);
            __builtin_amdgcn_sched_barrier(0);
            rd(F1, cur, 1, 3, 6);
            mm(F0, 1, 2);
#pragma unroll
            for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            load_a(rld);
            __builtin_amdgcn_sched_barrier(0);
            rd(F1, cur, 1, 6, 8);
            mm(F0, 2, 3);
            next_b();
#pragma unroll
            for (int i = 0; i < 2; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            // G3 .. G5: step 1 from F1; the first fragments of chunk it + 1 (F0 is free now) and the split of chunk it + 2's
            // activations (loaded a chunk ago: six newer operations in the queue) between them, the LDS stores last
            wait_a(rsp, (KO & 2) ? 2 : 6);
            rd(F0, nxt, 0, 0, 4);
            split_a(rsp, 0);
            mm(F1, 0, 1);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            rd(F0, nxt, 0, 4, 8);
            split_a(rsp, 1);
            mm(F1, 1, 2);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            store_a(fre);
            mm(F1, 2, 3);
            next_a();
#pragma unroll
            for (int i = 0; i < 2; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if constexpr (!(KO & 32)) __syncthreads();
            return;
        }
        const bool m1 = ST || it + 1 < total, m2 = ST || it + 2 < total, m3 = ST || it + 3 < total;
        if (m2) { dma(fre); next_b(); }
        if (m3) { load_a(rld); next_a(); }
        rd(F1, cur, 1);
        mm(F0);
        __builtin_amdgcn_sched_barrier(0);
        if (m1) rd(F0, nxt, 0);
        if (m2) {
            wait_a(rsp, ST ? ((KO & 2) ? 2 : 6) : 0);
            split_a(rsp, 0);
            split_a(rsp, 1);
            store_a(fre);
        }
        mm(F1);
        __builtin_amdgcn_sched_barrier(0);
        if (ST) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(KO & 32)) __syncthreads();
    };

    // ---- prologue: stages 0 and 1 complete, chunk 2's activations in flight
    open_b();
    open_a();
    dma(0);
    next_b();
    if (total > 1) { dma(STAGE); next_b(); }
    load_a(rA);
    next_a();
    if (total > 1) { load_a(rB); next_a(); }
    {
        // row scales, common to all sources (they add into one accumulator): bound = max over sources and column blocks.
        // All loads first -- predicated over the CAPE_MAX_SRC slots -- then the maxima (gemm_h2_kernel)
        float4 bv[CAPE_MAX_SRC];
#pragma unroll
        for (int si = 0; si < CAPE_MAX_SRC; ++si) {
            bv[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (si < p.nsrc && 4 * q < p.s[si].rmw)
                bv[si] = *reinterpret_cast<const float4 *>(p.s[si].rm + ((long long)n * p.Mo + rc) * p.s[si].rmw + 4 * q);
        }
        float m = 0.f;
#pragma unroll
        for (int si = 0; si < CAPE_MAX_SRC; ++si) m = fmaxf(m, fmaxf(fmaxf(bv[si].x, bv[si].y), fmaxf(bv[si].z, bv[si].w)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, false)));   // quad_perm [1,0,3,2]
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, false)));   // quad_perm [2,3,0,1]
        float inv;
        h2_scale_of(m, sa, inv);
        if (q == 0) inv_row[r] = inv;
    }
    // (counted waits are safe with MORE operations behind the awaited ones than assumed -- the row-bound loads above -- never
    // with fewer)
    wait_a(rA, total > 1 ? 2 : 0);
    split_a(rA, 0); split_a(rA, 1);
    store_a(0);
    if (total > 2) { load_a(rA); next_a(); }
    if (total > 1) {
        wait_a(rB, total > 2 ? 2 : 0);
        split_a(rB, 0); split_a(rB, 1);
        store_a(STAGE);
    }
    if (total > 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    rd(F0, 0, 0);

    // ---- chunk loop; the stage roles rotate (cur -> fre -> nxt -> cur), the register sets alternate (static roles: a
    //      runtime choice between the two sets would put them in scratch memory)
    int cur = 0, nxt = STAGE, fre = 2 * STAGE, it = 0;
    for (; it + 4 < total; it += 2) {
        chunk(it, cur, nxt, fre, rB, rA, std::true_type{});            // splits chunk it + 2 (rA), loads it + 3 into rB
        chunk(it + 1, nxt, fre, cur, rA, rB, std::true_type{});
        const int c = cur; cur = fre; fre = nxt; nxt = c;
    }
    for (; it < total; it += 2) {
        chunk(it, cur, nxt, fre, rB, rA, std::false_type{});
        if (it + 1 < total) chunk(it + 1, nxt, fre, cur, rA, rB, std::false_type{});
        const int c = cur; cur = fre; fre = nxt; nxt = c;
    }

    if constexpr (KO & 64) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) asm volatile("" ::"v"(acc[a][b][0]), "v"(acc[a][b][15]));
        return;
    }
    // ---- undo the scales (all powers of two: exact), then the shared epilogues
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int f = min(f0 + wn * WTN + b * 32 + li, p.F - 1);
            const float wi = p.wsi[f];
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] *= inv_row[wm * WTM + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh] * wi;
        }
    if (p.rankR > 0 || p.bias_mode == CAPE_BIAS_VERTEX || p.act == CAPE_ACT_TANH) {
        f32x16 none[1][1];
        gconv_epilogue<BM, BN, 2, 4, false, float>(p, acc, none, n, r0, f0, wm, wn, li, lh);
        return;
    }
    gconv_epilogue_short<BM, BN, float, 2, 4>(p, acc, n, r0, f0, wm, wn, li, lh);
}

// The wide tile pays where one round of workgroups still fills the chip (one workgroup per CU) and the contraction is long
// enough for its deeper pipeline: >= 8 chunks, >= 160 tiles.  CAPE_H2X=0: off (A/B against the 128 x 128 / 64 x 64 kernels),
// 2: wherever the tile shape applies.
inline bool h2x_wanted(bool dual, int N, int Mo, int F, int Ktot) {
    static const int mode = getenv("CAPE_H2X") ? atoi(getenv("CAPE_H2X")) : 1;
    if (!mode || dual || F < 192) return false;
    if (mode == 2) return true;
    const long long tiles = (long long)N * ((Mo + 127) / 128) * ((F + 255) / 256);
    return Ktot >= 256 && tiles >= 160;
}

inline void h2x_launch(const GconvParams &p, dim3 grid, hipStream_t st) { CAPE_LAUNCH((gemm_h2x_kernel<>), grid, dim3(512), 0, st, p); }

}  // namespace
