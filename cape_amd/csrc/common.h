// Shared device/host helpers for libcape_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cape_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activation storage types -------------------------------------------------------------------------------
// Activations [N, M, ld] are stored as fp32 (the parity path) or as bf16 (BASELINE configs[4]: bf16 storage, fp32
// accumulate; the "_bf16" entry points of cape_hip.h).  Kernels that exist for both are templates over the element
// type AT in {float, cape_bf16}; every load widens to fp32, every store rounds to nearest even (v_cvt_pk_bf16_f32).
typedef unsigned short cape_bf16;
typedef __bf16 cape_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float cape_f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cape_pack_bf16(float a, float b) {      // a -> low half, b -> high half
    const cape_f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cape_bf16x2_t));
}
__device__ __forceinline__ float cape_ld(const float *p) { return *p; }
__device__ __forceinline__ float cape_ld(const cape_bf16 *p) { return __builtin_bit_cast(float, (unsigned)(*p) << 16); }
__device__ __forceinline__ void cape_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void cape_st(cape_bf16 *p, float v) { *p = (cape_bf16)(cape_pack_bf16(v, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float4 cape_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 cape_ld4(const cape_bf16 *p) {           // 8-byte aligned
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    float4 r;
    r.x = __builtin_bit_cast(float, u.x << 16); r.y = __builtin_bit_cast(float, u.x & 0xFFFF0000u);
    r.z = __builtin_bit_cast(float, u.y << 16); r.w = __builtin_bit_cast(float, u.y & 0xFFFF0000u);
    return r;
}
__device__ __forceinline__ void cape_st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void cape_st4(cape_bf16 *p, float4 v) {
    uint2 u;
    u.x = cape_pack_bf16(v.x, v.y); u.y = cape_pack_bf16(v.z, v.w);
    *reinterpret_cast<uint2 *>(p) = u;
}
// two consecutive elements (4-byte aligned for bf16, 8-byte for fp32)
__device__ __forceinline__ float2 cape_ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
__device__ __forceinline__ float2 cape_ld2(const cape_bf16 *p) {
    const unsigned u = *reinterpret_cast<const unsigned *>(p);
    return make_float2(__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xFFFF0000u));
}
// v + (value of lane ^ 16) and v + (value of lane ^ 32) without the LDS crossbar: the gfx950 row / half swaps
// (__shfl_xor compiles to ds_bpermute_b32; used by the dense-layer kernels, which reduce 64 accumulators per thread)
// Written as inline asm: the ROCm 7.2 builtins (__builtin_amdgcn_permlane16_swap / 32_swap) return the first register
// twice (the generated code adds vdst to itself), so the exchanged half is lost; tools/ubench/permlane_check.hip verifies
// both helpers against __shfl_xor.  v_permlane16_swap a, b:  a' = [a0 b0 a2 b2], b' = [a1 b1 a3 b3] (rows of 16 lanes);
// v_permlane32_swap a, b:  a' = [a.lo b.lo], b' = [a.hi b.hi].  With a = b = v the sum a' + b' is v + v(lane ^ 16 / 32).
__device__ __forceinline__ float cape_sum_xor16(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float cape_sum_xor32(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// max over aligned groups of G consecutive lanes (G a power of two <= 64, wave-uniform): DPP inside the 16-lane rows
// (quad permutations, half-row mirror, row mirror -- after each step all lanes of the merged group hold its maximum, so any
// lane of the partner group is a valid source), ds_bpermute across them.
#define CAPE_DPP_MAX(v, CTRL) fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, false)))
__device__ __forceinline__ float cape_group_max(float m, int G) {
    if (G >= 2) m = CAPE_DPP_MAX(m, 0xB1);            // quad_perm [1,0,3,2]
    if (G >= 4) m = CAPE_DPP_MAX(m, 0x4E);            // quad_perm [2,3,0,1]
    if (G >= 8) m = CAPE_DPP_MAX(m, 0x141);           // row_half_mirror
    if (G >= 16) m = CAPE_DPP_MAX(m, 0x140);          // row_mirror
    if (G >= 32) m = fmaxf(m, __shfl_xor(m, 16));
    if (G >= 64) m = fmaxf(m, __shfl_xor(m, 32));
    return m;
}
// row bounds written by the kernels that own whole rows (one float4 per row: [bound, 0, 0, 0]; consumed by gemm_h2.h)
__device__ __forceinline__ void cape_store_rowmax(float *rm, long long row, float m) {
    *reinterpret_cast<float4 *>(rm + 4 * row) = make_float4(m, 0.f, 0.f, 0.f);
}

// VW (1, 4 or 8) consecutive elements, widened to fp32 / rounded back.  Alignment: VW elements of fp32 up to 16 bytes
// (VW = 8: two 16-byte accesses), VW elements of bf16 (8 -> one 16-byte access).
typedef unsigned cape_u32x4 __attribute__((ext_vector_type(4)));
template <int VW>
__device__ __forceinline__ void cape_ldv(const float *p, float (&o)[VW]) {
    if constexpr (VW == 1) {
        o[0] = *p;
    } else {
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const float4 v = *reinterpret_cast<const float4 *>(p + 4 * h);
            o[4 * h] = v.x; o[4 * h + 1] = v.y; o[4 * h + 2] = v.z; o[4 * h + 3] = v.w;
        }
    }
}
template <int VW>
__device__ __forceinline__ void cape_ldv(const cape_bf16 *p, float (&o)[VW]) {
    if constexpr (VW == 1) {
        o[0] = cape_ld(p);
    } else if constexpr (VW == 4) {
        const uint2 u = *reinterpret_cast<const uint2 *>(p);
        o[0] = __builtin_bit_cast(float, u.x << 16); o[1] = __builtin_bit_cast(float, u.x & 0xFFFF0000u);
        o[2] = __builtin_bit_cast(float, u.y << 16); o[3] = __builtin_bit_cast(float, u.y & 0xFFFF0000u);
    } else {
        const cape_u32x4 u = *reinterpret_cast<const cape_u32x4 *>(p);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            o[2 * h] = __builtin_bit_cast(float, u[h] << 16);
            o[2 * h + 1] = __builtin_bit_cast(float, u[h] & 0xFFFF0000u);
        }
    }
}
template <int VW>
__device__ __forceinline__ void cape_stv(float *p, const float (&v)[VW]) {
    if constexpr (VW == 1) {
        *p = v[0];
    } else {
#pragma unroll
        for (int h = 0; h < VW / 4; ++h)
            *reinterpret_cast<float4 *>(p + 4 * h) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
    }
}
template <int VW>
__device__ __forceinline__ void cape_stv(cape_bf16 *p, const float (&v)[VW]) {
    if constexpr (VW == 1) {
        cape_st(p, v[0]);
    } else if constexpr (VW == 4) {
        uint2 u;
        u.x = cape_pack_bf16(v[0], v[1]); u.y = cape_pack_bf16(v[2], v[3]);
        *reinterpret_cast<uint2 *>(p) = u;
    } else {
        cape_u32x4 u;
#pragma unroll
        for (int h = 0; h < 4; ++h) u[h] = cape_pack_bf16(v[2 * h], v[2 * h + 1]);
        *reinterpret_cast<cape_u32x4 *>(p) = u;
    }
}
template <typename AT> struct cape_is_bf16 { static constexpr bool value = false; };
template <> struct cape_is_bf16<cape_bf16> { static constexpr bool value = true; };

// hipGetLastError() is sticky per thread: clear whatever another runtime user (e.g. torch's device
// probing) left behind before launching, so that CAPE_LAUNCH_CHECK reports OUR launch only.
#define CAPE_LAUNCH(...)            \
    do {                            \
        (void)hipGetLastError();    \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define CAPE_LAUNCH_CHECK()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

__device__ __forceinline__ float cape_act(float v, int act) {
    switch (act) {
        case CAPE_ACT_LEAKY: return v > 0.f ? v : 0.2f * v;
        case CAPE_ACT_RELU: return v > 0.f ? v : 0.f;
        case CAPE_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float cape_act_grad_from_out(float y, int act) {
    switch (act) {
        case CAPE_ACT_LEAKY: return y > 0.f ? 1.f : 0.2f;
        case CAPE_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case CAPE_ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

// XCD-aware block -> (sample, tile) mapping.  Blocks are observed to be dispatched round
// robin over the 8 XCDs (block b -> XCD b % 8); keeping all tiles of one sample on one XCD
// lets the neighbour-row gathers of that sample hit a single 4 MiB L2.  Speed only: any
// mapping is correct.
__device__ __forceinline__ void cape_map_block(int b, int N, int T, int &n, int &t) {
    if ((N & 7) == 0) {
        const int per = N >> 3;
        const int local = b >> 3;
        n = (b & 7) * per + local / T;
        t = local % T;
    } else {
        n = b / T;
        t = b % T;
    }
}

// Weight-gradient launches: block -> (output tile, contraction split).  All tiles of one split read the same rows of
// the sources and of dz; with blocks dispatched round robin over the 8 XCDs (block b -> XCD b % 8) the natural order
// (tile fastest) spreads those re-reads over all eight L2s -- measured 173 MB of L2-miss traffic per launch against
// 58 MB algorithmic on the widest layer.  Here XCD x runs ALL tiles of split 8*g + x, so the re-reads of a row range
// hit one L2.  The grid is rounded up to whole groups of 8 splits; blocks of a split beyond nsplit exit at once.
__device__ __forceinline__ bool cape_map_dw_block(int b, int ntiles, int nsplit, int &tile, int &split) {
    const int per = ntiles << 3;
    const int g = b / per, loc = b - g * per;
    split = (g << 3) + (loc & 7);
    tile = loc >> 3;
    return split < nsplit;
}
