// Standalone timing of cape_gconv_fwd on plain sources at the layer shapes of CAPE-affineconv_nz64
// (no torch; links libcape_hip.so).  Kernel selection knobs are read from the environment by the library
// (CAPE_GEMM_PLAIN, CAPE_GP_NBUF, CAPE_GP_BM64_BELOW): run once per setting and compare the columns.
//   hipcc -O2 gemm_bench.cpp -I../../include -L../../cape_amd -lcape_hip -Wl,-rpath,'$ORIGIN/../../cape_amd' -o gemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "cape_hip.h"

struct Shape { int N, Mo, nsrc, C, F, dual, kc; };   // nsrc sources of C channels each; kc = weights k-contiguous

static float *dev_rand(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
    float *d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char **argv) {
    std::vector<Shape> shapes = {
        {16, 862, 2, 512, 512, 0, 0}, {16, 862, 3, 256, 512, 0, 0}, {16, 862, 2, 256, 512, 0, 0}, {16, 862, 1, 512, 256, 0, 0},
        {16, 862, 3, 256, 256, 0, 0}, {16, 862, 1, 256, 256, 0, 0}, {16, 862, 1, 512, 64, 0, 0}, {16, 862, 1, 64, 512, 0, 0},
        {16, 862, 2, 512, 256, 1, 0}, {16, 862, 2, 256, 256, 1, 0},
        {16, 1723, 2, 128, 256, 0, 0}, {16, 1723, 3, 128, 128, 0, 0}, {16, 1723, 2, 256, 128, 1, 0}, {16, 1723, 2, 128, 128, 1, 0},
        {16, 3445, 2, 64, 128, 0, 0}, {16, 3445, 3, 64, 64, 0, 0}, {16, 3445, 2, 128, 64, 1, 0}, {16, 3445, 2, 64, 64, 1, 0},
        {16, 6890, 3, 32, 32, 0, 0}, {16, 6890, 2, 64, 32, 1, 0}, {16, 6890, 2, 32, 32, 1, 0},
    };
    int iters = 20;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double tot_us = 0, tot_fl = 0;
    for (int layout = 0; layout < 2; ++layout)
    for (const Shape &s : shapes) {
        const size_t xs = (size_t)s.Mo * s.C;
        std::vector<float *> xs_d, w_d, w2_d;
        cape_src_t srcs[CAPE_MAX_SRC];
        memset(srcs, 0, sizeof(srcs));
        for (int i = 0; i < s.nsrc; ++i) {
            float *x = dev_rand((size_t)s.N * xs, 7 + i, 1.0f);
            float *w = dev_rand((size_t)s.C * s.F, 100 + i, 0.05f);
            xs_d.push_back(x); w_d.push_back(w);
            srcs[i].x = x; srcs[i].x_sample_stride = (int64_t)xs; srcs[i].ldx = s.C; srcs[i].C = s.C;
            srcs[i].w = w;
            // the same buffer read as W[c][f] (output-contiguous) or as W^T[f][c] (contraction-contiguous): different
            // math for the two layouts, identical work
            if (layout) { srcs[i].w_rs = 1; srcs[i].w_cs = s.C; } else { srcs[i].w_rs = s.F; srcs[i].w_cs = 1; }
            if (s.dual && i == 0) {
                float *w2 = dev_rand((size_t)s.C * s.F, 200 + i, 0.05f);
                w2_d.push_back(w2);
                srcs[i].w2 = w2; srcs[i].w2_rs = srcs[i].w_rs; srcs[i].w2_cs = srcs[i].w_cs;
            }
        }
        float *y; hipMalloc(&y, (size_t)s.N * s.Mo * s.F * 4);
        float *bias = dev_rand(s.F, 5, 0.1f);
        unsigned *mask = nullptr;
        if (s.dual) hipMalloc(&mask, (size_t)s.N * s.Mo * ((s.F + 31) / 32) * 4);
        auto run = [&]() {
            return cape_gconv_fwd(srcs, s.nsrc, y, (int64_t)s.Mo * s.F, s.F, s.N, s.Mo, s.F, s.dual ? nullptr : bias,
                                  s.dual ? CAPE_BIAS_NONE : CAPE_BIAS_CHANNEL, s.dual ? CAPE_ACT_NONE : CAPE_ACT_LEAKY, mask, nullptr, 0, nullptr);
        };
        int rc = run();
        if (rc) { printf("rc %d\n", rc); return 1; }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = 1e3 * ms / iters;
        const double fl = 2.0 * s.N * s.Mo * (double)s.C * s.nsrc * s.F + (s.dual ? 2.0 * s.N * s.Mo * (double)s.C * s.F : 0.0);
        // checksum of a strided sample of the output
        std::vector<float> h((size_t)s.N * s.Mo * s.F);
        hipMemcpy(h.data(), y, h.size() * 4, hipMemcpyDeviceToHost);
        double cs = 0, ca = 0;
        for (size_t i = 0; i < h.size(); i += 97) { cs += h[i]; ca += h[i] < 0 ? -h[i] : h[i]; }
        printf("%s Mo%5d F%4d C%4dx%d%s  %8.1f us %6.1f TF  sum %.6e abs %.6e\n", layout ? "kc" : "nc", s.Mo, s.F, s.C, s.nsrc,
               s.dual ? " dual" : "     ", us, fl / us / 1e6, cs, ca);
        tot_us += us; tot_fl += fl;
        for (float *p : xs_d) hipFree(p);
        for (float *p : w_d) hipFree(p);
        for (float *p : w2_d) hipFree(p);
        hipFree(y); hipFree(bias); if (mask) hipFree(mask);
    }
    printf("TOTAL %.1f us  %.1f TF\n", tot_us, tot_fl / tot_us / 1e6);
    return 0;
}
