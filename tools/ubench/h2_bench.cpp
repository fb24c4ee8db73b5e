// Library-level sweep of the fp16 two-piece contraction (cape_gconv_fwd_h2: gemm_h2_kernel) against the six-product bf16 form
// (cape_gconv_fwd: gemm_split_kernel) over the layer shapes of CAPE-affineconv_nz64, through the C-ABI only (no torch):
// cape_rowmax -> row bounds, cape_weight_pieces -> piece planes, then the contraction.  CAPE_H2_TILE=BMxBN forces a tile.
//   hipcc -O2 h2_bench.cpp -I../../include -L../../cape_amd -lcape_hip -Wl,-rpath,'$ORIGIN/../../cape_amd' -o h2_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "cape_hip.h"

struct Shape { int N, Mo, K, C, F, dual; };      // K sources of C channels each (polynomial order K), F output columns

static float *dev_rand(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
    float *d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}

struct Pieces { uint16_t *fh, *fl, *bh, *bl; float *fsi, *bsi, *bsc, *pc; };

static Pieces make_pieces(const float *W, int Ch, int K, int F, const float *pair, int pairK) {
    Pieces P;
    const size_t n = (size_t)Ch * K * F;
    hipMalloc(&P.fh, n * 2); hipMalloc(&P.fl, n * 2); hipMalloc(&P.bh, n * 2); hipMalloc(&P.bl, n * 2);
    hipMalloc(&P.fsi, (size_t)K * F * 4); hipMalloc(&P.bsi, (size_t)Ch * K * 4); hipMalloc(&P.bsc, (size_t)Ch * 4);
    hipMalloc(&P.pc, (size_t)((Ch * K + 63) / 64) * F * 4);
    cape_wpiece_item_t it;
    memset(&it, 0, sizeof it);
    it.w = W; it.Ch = Ch; it.K = K; it.F = F; it.pair_w = pair; it.pair_K = pairK;
    it.f_hi = P.fh; it.f_lo = P.fl; it.b_hi = P.bh; it.b_lo = P.bl; it.fscale_inv = P.fsi; it.bscale_inv = P.bsi; it.bscale_c_inv = P.bsc; it.colmax_partial = P.pc;
    int32_t mo[2], po[2];
    if (cape_weight_pieces_blocks(&it, 1, mo, po)) { printf("pieces: bad item\n"); exit(1); }
    cape_wpiece_item_t *dit; int32_t *dmo, *dpo;
    hipMalloc(&dit, sizeof it); hipMalloc(&dmo, 8); hipMalloc(&dpo, 8);
    hipMemcpy(dit, &it, sizeof it, hipMemcpyHostToDevice); hipMemcpy(dmo, mo, 8, hipMemcpyHostToDevice); hipMemcpy(dpo, po, 8, hipMemcpyHostToDevice);
    if (cape_weight_pieces(dit, 1, dmo, mo[1], dpo, po[1], nullptr)) { printf("pieces: launch failed\n"); exit(1); }
    hipDeviceSynchronize();
    hipFree(dit); hipFree(dmo); hipFree(dpo);
    return P;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 30;
    std::vector<Shape> shapes = {
        {16, 862, 2, 512, 512, 0}, {16, 862, 2, 256, 512, 0}, {16, 862, 1, 512, 256, 0}, {16, 862, 1, 512, 512, 0}, {16, 862, 1, 512, 64, 0},
        {16, 862, 2, 512, 256, 1}, {16, 862, 2, 256, 256, 1},
        {16, 1723, 2, 128, 256, 0}, {16, 1723, 2, 256, 256, 0}, {16, 1723, 1, 256, 256, 0}, {16, 1723, 1, 256, 128, 0}, {16, 1723, 2, 256, 128, 1},
        {16, 1723, 2, 128, 128, 1},
        {16, 3445, 2, 64, 128, 0}, {16, 3445, 2, 128, 128, 0}, {16, 3445, 1, 128, 256, 0}, {16, 3445, 1, 128, 128, 0}, {16, 3445, 2, 128, 64, 1},
        {16, 3445, 2, 64, 64, 1},
        {16, 6890, 2, 64, 64, 0}, {16, 6890, 1, 64, 128, 0}, {16, 6890, 1, 64, 64, 0}, {16, 6890, 1, 32, 64, 0},
    };
    if (argc > 2 && atoi(argv[2]) > 0)               // batch override: how do the short launches behave over more rounds of workgroups?
        for (Shape &s : shapes) s.N = atoi(argv[2]);
    if (argc > 3) {                                  // shape filter: comma-separated indices into the list above (counter passes)
        std::vector<Shape> keep;
        for (char *t = strtok(argv[3], ","); t; t = strtok(nullptr, ",")) keep.push_back(shapes.at(atoi(t)));
        shapes = keep;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double tot[2] = {0, 0};
    for (const Shape &s : shapes) {
        const size_t xs = (size_t)s.Mo * s.C;
        float *W = dev_rand((size_t)s.C * s.K * s.F, 100, 0.05f);              // row c*K + k
        float *Wa = s.dual ? dev_rand((size_t)s.C * s.F, 200, 0.05f) : nullptr;
        Pieces P = make_pieces(W, s.C, s.K, s.F, Wa, 1), Pa;
        if (s.dual) Pa = make_pieces(Wa, s.C, 1, s.F, W, s.K);
        cape_src_t srcs[CAPE_MAX_SRC];
        cape_h2_src_t hs[CAPE_MAX_SRC];
        memset(srcs, 0, sizeof srcs); memset(hs, 0, sizeof hs);
        std::vector<float *> xd, rd;
        for (int k = 0; k < s.K; ++k) {
            float *x = dev_rand((size_t)s.N * xs, 7 + k, 1.0f), *rm;
            hipMalloc(&rm, (size_t)s.N * s.Mo * 16);
            cape_rowmax(x, (int64_t)xs, s.C, s.N, s.Mo, s.C, rm, 4, nullptr);
            xd.push_back(x); rd.push_back(rm);
            srcs[k].x = x; srcs[k].x_sample_stride = (int64_t)xs; srcs[k].ldx = s.C; srcs[k].C = s.C;
            srcs[k].w = W + (size_t)k * s.F; srcs[k].w_rs = (int64_t)s.K * s.F; srcs[k].w_cs = 1;
            hs[k].w_hi = P.fh + (size_t)k * s.F * s.C; hs[k].w_lo = P.fl + (size_t)k * s.F * s.C; hs[k].w_pitch = s.C;
            hs[k].rowmax = rm; hs[k].rowmax_w = 4;
            if (s.dual && k == 0) {
                srcs[k].w2 = Wa; srcs[k].w2_rs = s.F; srcs[k].w2_cs = 1;
                hs[k].w2_hi = Pa.fh; hs[k].w2_lo = Pa.fl; hs[k].w2_pitch = s.C;
            }
        }
        cape_h2_t h2;
        memset(&h2, 0, sizeof h2);
        h2.src = hs; h2.wscale_inv = P.fsi; h2.w2scale_inv = s.dual ? Pa.fsi : nullptr;
        float *y[2], *rmo; hipMalloc(&y[0], (size_t)s.N * s.Mo * s.F * 4); hipMalloc(&y[1], (size_t)s.N * s.Mo * s.F * 4);
        hipMalloc(&rmo, (size_t)s.N * s.Mo * 16 * 4);
        h2.rowmax_out = rmo; h2.rowmax_out_w = (((s.F + 31) / 32) + 3) / 4 * 4;
        float *bias = dev_rand(s.F, 5, 0.1f);
        unsigned *mask = nullptr;
        if (s.dual) hipMalloc(&mask, (size_t)s.N * s.Mo * ((s.F + 31) / 32) * 4);
        double us[2];
        int fam[2];
        for (int v = 0; v < 2; ++v) {
            auto run = [&]() {
                return cape_gconv_fwd_h2(srcs, s.K, y[v], (int64_t)s.Mo * s.F, s.F, s.N, s.Mo, s.F, s.dual ? nullptr : bias,
                                         s.dual ? CAPE_BIAS_NONE : CAPE_BIAS_CHANNEL, s.dual ? CAPE_ACT_NONE : CAPE_ACT_LEAKY, mask, nullptr, 0,
                                         v ? &h2 : nullptr, nullptr);
            };
            int32_t plan[4];
            cape_gconv_fwd_plan_h2(srcs, s.K, s.N, s.Mo, s.F, v ? &h2 : nullptr, plan);
            fam[v] = plan[0] * 1000000 + plan[1] * 1000 + plan[2];
            int rc = run();
            if (rc) { printf("rc %d\n", rc); return 1; }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) run();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            us[v] = 1e3 * ms / iters;
            tot[v] += us[v];
        }
        std::vector<float> h0((size_t)s.N * s.Mo * s.F), h1(h0.size());
        hipMemcpy(h0.data(), y[0], h0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), y[1], h1.size() * 4, hipMemcpyDeviceToHost);
        double num = 0, den = 0;
        for (size_t i = 0; i < h0.size(); i += 13) { const double d = (double)h1[i] - h0[i]; num += d * d; den += (double)h0[i] * h0[i]; }
        const double fl = 2.0 * s.N * s.Mo * (double)s.C * s.K * s.F + (s.dual ? 2.0 * s.N * s.Mo * (double)s.C * s.F : 0.0);
        printf("Mo%5d C%4dx%d F%4d%s  split fam %7d %7.1f us %6.1f TF | h2 fam %7d %7.1f us %6.1f TF  x%.2f  rel diff %.1e\n", s.Mo, s.C, s.K, s.F,
               s.dual ? " dual" : "     ", fam[0], us[0], fl / us[0] / 1e6, fam[1], us[1], fl / us[1] / 1e6, us[0] / us[1], sqrt(num / (den + 1e-300)));
        for (float *p : xd) hipFree(p);
        for (float *p : rd) hipFree(p);
        hipFree(W); if (Wa) hipFree(Wa); hipFree(y[0]); hipFree(y[1]); hipFree(rmo); hipFree(bias); if (mask) hipFree(mask);
        hipFree(P.fh); hipFree(P.fl); hipFree(P.bh); hipFree(P.bl); hipFree(P.fsi); hipFree(P.bsi); hipFree(P.bsc);
        if (s.dual) { hipFree(Pa.fh); hipFree(Pa.fl); hipFree(Pa.bh); hipFree(Pa.bl); hipFree(Pa.fsi); hipFree(Pa.bsi); hipFree(Pa.bsc); }
    }
    printf("TOTAL split %.1f us   h2 %.1f us   x%.2f\n", tot[0], tot[1], tot[0] / tot[1]);
    return 0;
}
