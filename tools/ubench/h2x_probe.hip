// Phase knock-outs of the wide two-piece contraction kernel (cape_amd/csrc/gemm_h2x.h, included directly: the kernel template
// with its KO parameter) on one layer shape: which of {MFMAs, weight DMA, activation loads, split + LDS stores, fragment reads,
// barriers} the launch time is made of.  Weight planes / row bounds are synthetic (timing only; correctness is h2_bench's job).
//   hipcc -O3 --offload-arch=gfx950 -I../../include -I../../cape_amd/csrc h2x_probe.hip -o h2x_probe
//   ./h2x_probe [Mo C nsrc F iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "gemm_h2.h"

static void *dev_fill(size_t bytes, unsigned seed, int mode) {          // mode 0: floats in [-1, 1); 1: fp16 bit patterns of small values
    std::vector<unsigned char> h(bytes);
    unsigned s = seed * 2654435761u + 12345u;
    if (mode == 0) {
        float *f = reinterpret_cast<float *>(h.data());
        for (size_t i = 0; i < bytes / 4; ++i) { s = s * 1664525u + 1013904223u; f[i] = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
    } else {
        unsigned short *u = reinterpret_cast<unsigned short *>(h.data());
        for (size_t i = 0; i < bytes / 2; ++i) { s = s * 1664525u + 1013904223u; u[i] = (unsigned short)(0x2000u + ((s >> 9) & 0x1FFFu) + ((s >> 3) & 0x8000u)); }
    }
    void *d; hipMalloc(&d, bytes); hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    return d;
}

template <int KO, int EXP>
static double run(const GconvParams &p, dim3 grid, int iters) {
    static hipEvent_t e0, e1;
    static bool init = false;
    if (!init) { hipEventCreate(&e0); hipEventCreate(&e1); init = true; }
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_h2x_kernel<KO, EXP>), grid, dim3(512), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return 1e3 * ms / iters;
}

int main(int argc, char **argv) {
    const int N = 16, Mo = argc > 1 ? atoi(argv[1]) : 862, C = argc > 2 ? atoi(argv[2]) : 512, nsrc = argc > 3 ? atoi(argv[3]) : 2;
    const int F = argc > 4 ? atoi(argv[4]) : 512, iters = argc > 5 ? atoi(argv[5]) : 30;
    GconvParams p;
    memset(&p, 0, sizeof p);
    for (int i = 0; i < nsrc; ++i) {
        SrcDev &S = p.s[i];
        S.x = (const float *)dev_fill((size_t)N * Mo * C * 4, 7 + i, 0); S.xs = (long long)Mo * C; S.ldx = C; S.C = C;
        S.wh = (const unsigned short *)dev_fill((size_t)F * C * 2, 100 + i, 1); S.wl = (const unsigned short *)dev_fill((size_t)F * C * 2, 200 + i, 1); S.wp = C;
        float *rm; hipMalloc(&rm, (size_t)N * Mo * 16);
        std::vector<float> h((size_t)N * Mo * 4, 0.f);
        for (size_t r = 0; r < (size_t)N * Mo; ++r) h[4 * r] = 1.0f;
        hipMemcpy(rm, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        S.rm = rm; S.rmw = 4;
    }
    p.nsrc = nsrc;
    float *y; hipMalloc(&y, (size_t)N * Mo * F * 4);
    p.y = y; p.ys = (long long)Mo * F; p.ldy = F; p.N = N; p.Mo = Mo; p.F = F;
    p.bias = (const float *)dev_fill((size_t)F * 4, 5, 0); p.bias_mode = CAPE_BIAS_CHANNEL; p.act = CAPE_ACT_LEAKY;
    p.deintK = 1;
    p.wsi = (const float *)dev_fill((size_t)F * 4, 6, 0);
    if (argc > 6 && atoi(argv[6])) {                 // row bounds of the output as well (what the model's layers ask for)
        p.rm_out_w = ((F + 31) / 32 + 3) / 4 * 4;
        hipMalloc(&p.rm_out, (size_t)N * Mo * p.rm_out_w * 4);
    }
    p.row_tiles = (Mo + 127) / 128; p.col_tiles = (F + 255) / 256;
    const dim3 grid((unsigned)(N * p.row_tiles * p.col_tiles));
    const double fl = 2.0 * N * Mo * (double)C * nsrc * F;
    printf("Mo %d C %dx%d F %d: %u workgroups of 128 x 256, %d chunks\n", Mo, C, nsrc, F, grid.x, C * nsrc / 32);
    // variants interleaved over several rounds in ONE process (the chip's clock moves with load and time: single passes of
    // different variants are not comparable); median and minimum per variant
    struct Row { const char *name; double (*fn)(const GconvParams &, dim3, int); std::vector<double> us; };
    std::vector<Row> rows = {
#define V(name, ko) {name, run<ko, 0>, {}}
#ifdef PROBE_FEW
        V("full kernel", 0), {"full kernel, loads in one block behind the barrier", run<0, 1>, {}}, V("full kernel without epilogue", 64),
        V("prologue + epilogue + barriers only", 1 + 2 + 4 + 8 + 16), V("prologue + barriers only", 1 + 2 + 4 + 8 + 16 + 64),
#else
        V("full kernel", 0), {"full kernel, loads in one block behind the barrier", run<0, 1>, {}}, V("no MFMAs", 1), V("no weight DMA", 2), V("no activation loads", 4),
        V("no global loads at all", 6), V("no split + LDS stores", 8), V("no fragment reads", 16),
        V("MFMAs + fragment reads + barriers only", 2 + 4 + 8), V("MFMAs + barriers only", 2 + 4 + 8 + 16), V("loads + split + stores + barriers only", 1 + 16),
        V("prologue + epilogue + barriers only", 1 + 2 + 4 + 8 + 16), V("full kernel without epilogue", 64),
#endif
    };
    const int rounds = 7;
    for (auto &r : rows) { if (getenv("PROBE_VERBOSE")) { printf("warm-up %s\n", r.name); fflush(stdout); } r.fn(p, grid, 3); hipDeviceSynchronize(); }   // warm-up, code object load
    for (int k = 0; k < rounds; ++k)
        for (auto &r : rows) r.us.push_back(r.fn(p, grid, iters));
    for (auto &r : rows) {
        std::sort(r.us.begin(), r.us.end());
        printf("  %-50s median %7.1f us  min %7.1f us  %6.1f TF\n", r.name, r.us[rounds / 2], r.us[0], fl / r.us[rounds / 2] / 1e6);
    }
    return 0;
}
