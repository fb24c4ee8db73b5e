// Standalone timing of cape_gconv_dw (weight gradient of plain sources) at the layer shapes of
// CAPE-affineconv_nz64; knobs via environment (CAPE_DW_PLAIN, CAPE_DW_CT, CAPE_DW_FT, CAPE_DW_WGS).
//   hipcc -O2 dw_bench.cpp -I../../include -L../../cape_amd -lcape_hip -Wl,-rpath,'$ORIGIN/../../cape_amd' -o dw_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "cape_hip.h"

struct Shape { int N, Mo, nsrc, C, F; };

static float *dev_rand(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
    float *d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}

int main() {
    std::vector<Shape> shapes = {
        {16, 862, 2, 512, 512}, {16, 862, 3, 512, 256}, {16, 862, 3, 256, 256}, {16, 862, 2, 256, 512}, {16, 862, 2, 256, 256},
        {16, 862, 1, 512, 64}, {16, 862, 1, 64, 512},
        {16, 1723, 3, 256, 128}, {16, 1723, 3, 128, 128}, {16, 1723, 2, 128, 256},
        {16, 3445, 3, 128, 64}, {16, 3445, 3, 64, 64}, {16, 3445, 2, 64, 128},
        {16, 6890, 3, 64, 32}, {16, 6890, 3, 32, 32}, {16, 6890, 2, 32, 64},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double tot_us = 0, tot_fl = 0;
    const int iters = 20;
    for (const Shape &s : shapes) {
        cape_src_t srcs[CAPE_MAX_SRC];
        memset(srcs, 0, sizeof(srcs));
        std::vector<float *> bufs;
        for (int i = 0; i < s.nsrc; ++i) {
            float *x = dev_rand((size_t)s.N * s.Mo * s.C, 7 + i, 1.0f);
            float *w; hipMalloc(&w, (size_t)s.C * s.F * 4);
            bufs.push_back(x); bufs.push_back(w);
            srcs[i].x = x; srcs[i].x_sample_stride = (int64_t)s.Mo * s.C; srcs[i].ldx = s.C; srcs[i].C = s.C;
            srcs[i].w = w; srcs[i].w_rs = s.F; srcs[i].w_cs = 1;
        }
        float *dz = dev_rand((size_t)s.N * s.Mo * s.F, 99, 1.0f);
        int64_t need = cape_gconv_dw_workspace_bytes(srcs, s.nsrc, s.N, s.Mo, s.F);
        void *ws; hipMalloc(&ws, need);
        auto run = [&]() { return cape_gconv_dw(srcs, s.nsrc, dz, (int64_t)s.Mo * s.F, s.F, nullptr, 0, s.N, s.Mo, s.F, 0, ws, need, nullptr); };
        int rc = run();
        if (rc) { printf("rc %d\n", rc); return 1; }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = 1e3 * ms / iters;
        const double fl = 2.0 * s.N * s.Mo * (double)s.C * s.nsrc * s.F;
        std::vector<float> h((size_t)s.C * s.F);
        double cs = 0, ca = 0;
        for (int i = 0; i < s.nsrc; ++i) {
            hipMemcpy(h.data(), srcs[i].w, h.size() * 4, hipMemcpyDeviceToHost);
            for (size_t k = 0; k < h.size(); k += 7) { cs += h[k]; ca += h[k] < 0 ? -h[k] : h[k]; }
        }
        printf("dw Mo%5d F%4d C%4dx%d       %8.1f us %6.1f TF  sum %.6e abs %.6e\n", s.Mo, s.F, s.C, s.nsrc, us, fl / us / 1e6, cs, ca);
        tot_us += us; tot_fl += fl;
        for (float *p : bufs) hipFree(p);
        hipFree(dz); hipFree(ws);
    }
    printf("TOTAL %.1f us  %.1f TF\n", tot_us, tot_fl / tot_us / 1e6);
    return 0;
}
