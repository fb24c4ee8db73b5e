// Sustained v_mfma_f32_32x32x2_f32 rate on the whole chip (no memory traffic): the DVFS-limited
// ceiling the fp32 gather-GEMM kernels are priced against.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) acc[a][g] = 0.f;
    float av = seed * (1 + (threadIdx.x % 7)), bv = seed * (2 + (threadIdx.x % 5));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
        av = -av;
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int g = 0; g < 16; ++g) s += acc[a][g];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int blocks, int iters, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1e-3f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1e-3f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * NACC * 4096.0;
        printf("blocks %5d (%.1f waves/SIMD) nacc %d iters %d: %.3f ms  %.1f TFLOP/s  -> implied clock %.2f GHz\n", blocks,
               blocks / 256.0, NACC, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 2.4);
    }
}

int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    run<4>(256, 20000, out);     // 1 wave per SIMD, ~50 ms at peak
    run<4>(512, 10000, out);
    run<4>(1024, 5000, out);
    run<2>(1024, 10000, out);
    run<4>(256, 400, out);       // short burst (~1 ms): clock before the power manager reacts?
    run<4>(1024, 100, out);      // ~60 us like a real launch
    return 0;
}
