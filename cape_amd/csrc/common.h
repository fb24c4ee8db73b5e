// Shared device/host helpers for libcape_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cape_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
