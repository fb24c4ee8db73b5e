// Weight gradient of the 3-channel output layer (2 x 32 channels -> 3; reference lib/models.py:611-616) at the full mesh
// resolution.  On the MFMA tile kernels this launch wastes > 90 % of every tile (22 us for the 30 MB it touches); here the
// narrow side lives in registers, the channels are spread over the lanes in float4 quads, rows over the remaining lanes:
// 12 us.  Exact fp32 FMA chains, fixed summation order.  (The mirror-image forms for the 3-channel INPUT layer -- forward and
// weight gradient with the <= 8 input channels in registers -- were built the same way and measured no faster than the tile
// kernels, 22-25 us; not kept.)  Included by gconv.hip inside its anonymous namespace.
#pragma once

// ---- weight gradient, narrow output: F <= 4, every C_s a multiple of 4 -----------------------------------------------------------
// thread = (row lane, quad of channels on the concatenated channel axis) with its 4 x F sums in registers.
__global__ __launch_bounds__(256) void dw_narrow_out_kernel(DwParams p, int lpr, int ntiles) {
    extern __shared__ float red[];                    // [RL][lpr][16]
    int tile, split;
    if (!cape_map_dw_block(blockIdx.x, ntiles, p.ngroups * p.rsplit, tile, split) || tile != 0) return;
    const int grp = split / p.rsplit, rs = split % p.rsplit;
    const int n_begin = grp * p.samples_per_group, n_end = min(p.N, n_begin + p.samples_per_group);
    const int ra = rs * p.rows_per_split, rb = min(p.Mo, ra + p.rows_per_split);
    const int RL = 256 / lpr;
    const int cq = threadIdx.x % lpr, rl = threadIdx.x / lpr;
    // this thread's channel quad: source si, first channel c0 (quads beyond the last source idle)
    int si = 0, c0 = 4 * cq;
    while (si < p.nsrc && c0 >= p.s[si].C) { c0 -= p.s[si].C; ++si; }
    const bool con = si < p.nsrc;
    const SrcDev &S = p.s[con ? si : 0];
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
    for (int n = n_begin; n < n_end; ++n) {
        const float *zb = p.dz + (long long)n * p.dzs;
        const float *xb = S.x + (long long)n * S.xs + (con ? c0 : 0);
        constexpr int UR = 4;
        for (int r0 = ra + rl; r0 < rb; r0 += UR * RL) {
            float d[UR][4];
            float4 xq[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = r0 + u * RL;
                const float okf = (r < rb && con) ? 1.f : 0.f;          // unconditional loads on a clamped row, zeroed afterwards
                const int rc = min(r, rb - 1);
#pragma unroll
                for (int f = 0; f < 4; ++f) d[u][f] = zb[(long long)rc * p.lddz + min(f, p.F - 1)] * (f < p.F ? okf : 0.f);
                xq[u] = *reinterpret_cast<const float4 *>(xb + (long long)rc * S.ldx);
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const float xa[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[i][f] = fmaf(xa[i], d[u][f], acc[i][f]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4 *>(&red[((rl * lpr + cq) * 4 + i) * 4]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    __syncthreads();
    if (rl == 0 && con) {
        float *out = p.ws + (long long)split * p.slab + p.part_off[si];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int l = 0; l < RL; ++l) {
                const float4 v = *reinterpret_cast<const float4 *>(&red[((l * lpr + cq) * 4 + i) * 4]);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const float sv[4] = {s.x, s.y, s.z, s.w};
            for (int f = 0; f < p.F; ++f) out[(long long)(c0 + i) * p.F + f] = sv[f];
        }
    }
}

inline int narrow_lpr(int quads) {
    int l = 1;
    while (l < quads) l <<= 1;
    return l;
}

// which narrow form (0 = none, 6 = narrow output) a weight-gradient launch of family ``fam`` takes
inline int dw_narrow_mode(const cape_src_t *srcs, int nsrc, const float *dz, int64_t dz_sample_stride, int32_t lddz, const float *dz2,
                          uint32_t dz2_mask, int F, int fam, bool bf16) {
    static const int on = getenv("CAPE_NARROW") ? atoi(getenv("CAPE_NARROW")) : 1;       // 0: A/B against the tile kernels
    if (!on || bf16 || fam == 0 || (dz2 && dz2_mask)) return 0;          // plain sources only (families 1..3), one gradient operand
    int sumC = 0;
    bool c4 = true, xal = true;
    for (int i = 0; i < nsrc; ++i) {
        sumC += srcs[i].C;
        c4 = c4 && (srcs[i].C & 3) == 0;
        xal = xal && (srcs[i].ldx & 3) == 0 && (srcs[i].x_sample_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(srcs[i].x) & 15) == 0;
    }
    if (F <= 4 && c4 && xal && sumC >= 16 && sumC <= 256) return 6;
    return 0;
}
