// Checks cape_sum_xor16 / cape_sum_xor32 (common.h) against __shfl_xor on one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../cape_amd/csrc/common.h"
__global__ void k(const float *in, float *o16, float *o32, float *r16, float *r32) {
    const float v = in[threadIdx.x];
    o16[threadIdx.x] = cape_sum_xor16(v);
    o32[threadIdx.x] = cape_sum_xor32(v);
    r16[threadIdx.x] = v + __shfl_xor(v, 16);
    r32[threadIdx.x] = v + __shfl_xor(v, 32);
}
int main() {
    float h[64], *d;
    for (int i = 0; i < 64; ++i) h[i] = (float)(i * i + 1);
    hipMalloc(&d, 5 * 64 * sizeof(float));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, d + 64, d + 128, d + 192, d + 256);
    float o[4][64];
    hipMemcpy(o, d + 64, sizeof(o), hipMemcpyDeviceToHost);
    int bad16 = 0, bad32 = 0;
    for (int i = 0; i < 64; ++i) { bad16 += o[0][i] != o[2][i]; bad32 += o[1][i] != o[3][i]; }
    printf("xor16 mismatches %d, xor32 mismatches %d\n", bad16, bad32);
    for (int i = 0; i < 64; i += 8) printf("lane %2d: swap16 %8.0f ref %8.0f | swap32 %8.0f ref %8.0f\n", i, o[0][i], o[2][i], o[1][i], o[3][i]);
    return bad16 + bad32;
}
