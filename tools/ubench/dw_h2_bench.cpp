// Standalone timing of the two-piece weight gradient (cape_gconv_dw_stage_h2: dw_h2_kernel + its slab reduction, timed
// separately) at the layer shapes of CAPE-affineconv_nz64, through the C-ABI only (no torch): cape_rowmax -> row bounds.
//   dw_h2_bench [iters] [shape indices, comma separated]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "cape_hip.h"

struct Shape { int N, Mo, nsrc, C, F; };

static float *dev_rand(size_t n, unsigned seed, float scale, std::vector<float> *keep = nullptr) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
    float *d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    if (keep) keep->swap(h);
    return d;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    std::vector<Shape> shapes = {
        {16, 862, 2, 512, 512}, {16, 862, 3, 512, 256}, {16, 862, 3, 256, 256}, {16, 862, 2, 256, 512}, {16, 862, 2, 256, 256},
        {16, 862, 1, 512, 64}, {16, 862, 1, 64, 512},
        {16, 1723, 3, 256, 128}, {16, 1723, 3, 128, 128}, {16, 1723, 2, 128, 256},
        {16, 3445, 3, 128, 64}, {16, 3445, 3, 64, 64}, {16, 3445, 2, 64, 128},
    };
    if (argc > 2) {
        std::vector<Shape> keep;
        for (char *t = strtok(argv[2], ","); t; t = strtok(nullptr, ",")) keep.push_back(shapes.at(atoi(t)));
        shapes = keep;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double tot[2] = {0, 0};
    for (const Shape &s : shapes) {
        cape_src_t srcs[CAPE_MAX_SRC];
        cape_h2_dw_t h2;
        memset(srcs, 0, sizeof(srcs)); memset(&h2, 0, sizeof h2);
        std::vector<float *> bufs;
        std::vector<float> hx0, hz;
        for (int i = 0; i < s.nsrc; ++i) {
            float *x = dev_rand((size_t)s.N * s.Mo * s.C, 7 + i, 1.0f, i == 0 ? &hx0 : nullptr);
            float *w, *rm; hipMalloc(&w, (size_t)s.C * s.F * 4); hipMalloc(&rm, (size_t)s.N * s.Mo * 16);
            cape_rowmax(x, (int64_t)s.Mo * s.C, s.C, s.N, s.Mo, s.C, rm, 4, nullptr);
            bufs.push_back(x); bufs.push_back(w); bufs.push_back(rm);
            srcs[i].x = x; srcs[i].x_sample_stride = (int64_t)s.Mo * s.C; srcs[i].ldx = s.C; srcs[i].C = s.C;
            srcs[i].w = w; srcs[i].w_rs = s.F; srcs[i].w_cs = 1;
            h2.src_rowmax[i] = rm; h2.src_rowmax_w[i] = 4;
        }
        float *dz = dev_rand((size_t)s.N * s.Mo * s.F, 99, 1.0f, &hz), *dzrm;
        hipMalloc(&dzrm, (size_t)s.N * s.Mo * 16);
        cape_rowmax(dz, (int64_t)s.Mo * s.F, s.F, s.N, s.Mo, s.F, dzrm, 4, nullptr);
        h2.dz_rowmax = dzrm; h2.dz_rowmax_w = 4;
        int64_t need = cape_gconv_dw_workspace_bytes(srcs, s.nsrc, s.N, s.Mo, s.F);
        void *ws; hipMalloc(&ws, need);
        int32_t plan[4];
        cape_gconv_dw_plan_h2(srcs, s.nsrc, dz, (int64_t)s.Mo * s.F, s.F, nullptr, 0, s.N, s.Mo, s.F, &h2, plan);
        double us[2];
        for (int stage = 1; stage <= 2; ++stage) {
            auto run = [&]() { return cape_gconv_dw_stage_h2(srcs, s.nsrc, dz, (int64_t)s.Mo * s.F, s.F, nullptr, 0, s.N, s.Mo, s.F, 0, ws, need, stage, &h2, nullptr); };
            int rc = run();
            if (rc) { printf("rc %d\n", rc); return 1; }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) run();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            us[stage - 1] = 1e3 * ms / iters;
            tot[stage - 1] += us[stage - 1];
        }
        // float64 check of a few elements of source 0's block
        std::vector<float> h((size_t)s.C * s.F);
        hipMemcpy(h.data(), srcs[0].w, h.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int t = 0; t < 24; ++t) {
            const int c = (t * 37 + 5) % s.C, f = (t * 101 + 3) % s.F;
            double ref = 0;
            for (size_t r = 0; r < (size_t)s.N * s.Mo; ++r) ref += (double)hx0[r * s.C + c] * hz[r * s.F + f];
            worst = fmax(worst, fabs(ref - h[(size_t)c * s.F + f])); scale = fmax(scale, fabs(ref));
        }
        const double fl = 2.0 * s.N * s.Mo * (double)s.C * s.nsrc * s.F;
        printf("dw Mo%5d F%4d C%4dx%d  fam %d tile %3dx%3d slabs %3d  contraction %7.1f us %6.1f TF  reduce %6.1f us   err %.1e of %.1e\n", s.Mo, s.F, s.C,
               s.nsrc, plan[0], plan[1], plan[2], plan[3], us[0], fl / us[0] / 1e6, us[1], worst, scale);
        for (float *p : bufs) hipFree(p);
        hipFree(dz); hipFree(dzrm); hipFree(ws);
    }
    printf("TOTAL contraction %.1f us  reduce %.1f us\n", tot[0], tot[1]);
    return 0;
}
