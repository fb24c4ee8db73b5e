// What does an LDS fragment read cost the MFMA stream of its SIMD?  (round 3 micro-benchmark, gfx950)
//
// One workgroup per CU, W waves per SIMD.  "mixed": every wave runs  [ R reads ; M MFMAs ]  in one basic block with the reads
// pinned between the MFMAs.  "apart": waves 0..3 (one per SIMD) only multiply, waves 4..7 only read -- if the multiplying wave
// still slows down, the cost is a SIMD-level hazard (register-file write-back, issue arbitration), not the reading wave's
// own issue port.  Reads never feed the MFMAs (their sum is stored at the end), so no waitcnt sits in the loop except the one
// the compiler needs for register reuse.
//
//   hipcc --offload-arch=gfx950 -O3 lds_mfma.hip -o lds_mfma && ./lds_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// KIND: 0 no reads, 1 ds_read_b128, 2 two ds_read_b64, 3 four ds_read_b32;  R reads (of 16 B per lane) per 8 MFMAs
template <int KIND, int R, int ROLE>   // ROLE 0: every wave does both; 1: waves < 4 multiply, waves >= 4 read
__global__ __launch_bounds__(512, 1) void lds_mfma_kernel(float *out, long long *ticks, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float *>(sm)[i] = (float)(i & 255);
    __syncthreads();
    const bool do_mm = ROLE == 0 || wave < 4, do_rd = ROLE == 0 || wave >= 4;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int g = 0; g < 16; ++g) acc[a][g] = 0.f;
    bf16x8 av, bv;
    for (int g = 0; g < 8; ++g) { av[g] = (__bf16)(0.001f * (lane + g)); bv[g] = (__bf16)(0.002f * (lane - g)); }
    float4 sum = {0, 0, 0, 0};
    // conflict-free 32-byte rows with the segment swizzle of the k16 layout
    const int li = lane & 31, lh = lane >> 5;
    const unsigned char *base = sm + (wave & 3) * 8192 + li * 32 + 16 * (lh ^ ((li >> 3) & 1));
    long long t0 = __builtin_readcyclecounter();
    t0 = __builtin_amdgcn_s_memtime();
    float4 v[R > 0 ? R : 1];
    for (int r = 0; r < (R > 0 ? R : 1); ++r) v[r] = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        // the values read an iteration ago are consumed first (software-pipelined, as in the GEMM kernels)
        if (do_rd) {
#pragma unroll
            for (int r = 0; r < R; ++r) { sum.x += v[r].x; sum.y += v[r].y; sum.z += v[r].z; sum.w += v[r].w; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (do_rd) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned char *p = base + r * 1024;
                if (KIND == 1) v[r] = *reinterpret_cast<const float4 *>(p);
                if (KIND == 2) {
                    float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(p + 8);
                    v[r] = make_float4(a.x, a.y, b.x, b.y);
                }
                if (KIND == 3) {
                    const float *q = reinterpret_cast<const float *>(p);
                    v[r] = make_float4(q[0], q[1], q[2], q[3]);
                }
            }
        }
        if (do_mm) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m & 3], 0, 0, 0);
        }
        if (ROLE == 0 && KIND != 0) {
            // spread the reads between the MFMAs
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m < R) __builtin_amdgcn_sched_group_barrier(0x100, KIND == 1 ? 1 : (KIND == 2 ? 2 : 4), 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (do_rd)
        for (int r = 0; r < R; ++r) { sum.x += v[r].x; sum.y += v[r].y; sum.z += v[r].z; sum.w += v[r].w; }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = sum.x + sum.y + sum.z + sum.w;
    for (int a = 0; a < 4; ++a)
        for (int g = 0; g < 16; ++g) s += acc[a][g];
    out[(long long)blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int R, int ROLE>
static void run(const char *name, int waves, float *out, long long *ticks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    lds_mfma_kernel<KIND, R, ROLE><<<256, 64 * waves>>>(out, ticks, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    lds_mfma_kernel<KIND, R, ROLE><<<256, 64 * waves>>>(out, ticks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
    double tmm = 0, trd = 0; int nmm = 0, nrd = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < waves; ++w) {
            if (ROLE == 0 || w < 4) { tmm += h[b * 8 + w]; ++nmm; }
            if (ROLE == 1 && w >= 4) { trd += h[b * 8 + w]; ++nrd; }
        }
    // s_memtime runs at 100 MHz-ish constant clock on some parts; report both ticks and the event time
    const double us = 1e3 * ms;
    const int mm_waves_per_simd = ROLE == 0 ? waves / 4 : 1;
    const double mfma_per_simd = 8.0 * iters * mm_waves_per_simd;
    printf("%-44s %d waves: %8.1f us  %6.1f ns per MFMA per SIMD   (multiplying waves %.0f ticks%s", name, waves, us,
           1e3 * us / mfma_per_simd, tmm / nmm, nrd ? ", " : ")\n");
    if (nrd) printf("reading waves %.0f ticks)\n", trd / nrd);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    float *out; long long *ticks;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&ticks, 256 * 8 * 8);
    const int it = 4000;
    printf("8 MFMA 32x32x16 bf16 per iteration per multiplying wave (256 cycles of the SIMD's matrix pipe); R reads of 16 B per lane\n");
    run<0, 0, 0>("no reads", 4, out, ticks, it);
    run<0, 0, 0>("no reads", 8, out, ticks, it);
    run<1, 2, 0>("mixed, 2 ds_read_b128 per 8 MFMA", 4, out, ticks, it);
    run<1, 4, 0>("mixed, 4 ds_read_b128 per 8 MFMA", 4, out, ticks, it);
    run<1, 8, 0>("mixed, 8 ds_read_b128 per 8 MFMA", 4, out, ticks, it);
    run<1, 4, 0>("mixed, 4 ds_read_b128 per 8 MFMA", 8, out, ticks, it);
    run<1, 8, 0>("mixed, 8 ds_read_b128 per 8 MFMA", 8, out, ticks, it);
    run<2, 4, 0>("mixed, 4 x 2 ds_read_b64 per 8 MFMA", 4, out, ticks, it);
    run<2, 4, 0>("mixed, 4 x 2 ds_read_b64 per 8 MFMA", 8, out, ticks, it);
    run<3, 4, 0>("mixed, 4 x 4 ds_read_b32 per 8 MFMA", 4, out, ticks, it);
    run<1, 2, 1>("apart, 2 ds_read_b128 per iteration", 8, out, ticks, it);
    run<1, 4, 1>("apart, 4 ds_read_b128 per iteration", 8, out, ticks, it);
    run<1, 8, 1>("apart, 8 ds_read_b128 per iteration", 8, out, ticks, it);
    run<2, 4, 1>("apart, 4 x 2 ds_read_b64 per iteration", 8, out, ticks, it);
    run<3, 4, 1>("apart, 4 x 4 ds_read_b32 per iteration", 8, out, ticks, it);
    return 0;
}
