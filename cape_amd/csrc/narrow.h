// Weight gradient of the 3-channel output layer (2 x 32 channels -> 3; reference lib/models.py:611-616) at the full mesh
// resolution.  On the MFMA tile kernels this launch wastes > 90 % of every tile (22 us for the 30 MB it touches); here the
// narrow side lives in registers, the channels are spread over the lanes in float4 quads, rows over the remaining lanes:
// 12 us.  Exact fp32 FMA chains, fixed summation order.  The mirror-image forms serve the 3-channel INPUT layer (2 x 3 channels
// -> 64; :541): forward and weight gradient with the <= 8 input channels in registers.  Included by gconv.hip inside its
// anonymous namespace.
#pragma once

constexpr int NARROW_MAXC = 8;        // channels (over all sources) the narrow-input forms keep in registers

// channel j of the concatenated channel axis -> base pointer / strides of its source.  Slots beyond the sources point at
// source 0 (always a valid address; their products meet zero weights / are never stored): every load below is unconditional,
// because a guarded load compiles to a branch with a wait behind it -- one memory round trip per load instead of one per pass.
struct NarrowTab {
    const float *xp[NARROW_MAXC];
    long long xss[NARROW_MAXC];
    int xld[NARROW_MAXC];
};
template <typename P>
__device__ __forceinline__ void narrow_table(const P &p, NarrowTab &t) {
#pragma unroll
    for (int j = 0; j < NARROW_MAXC; ++j) {
        int jj = j, si = 0;
        while (si < p.nsrc && jj >= p.s[si].C) { jj -= p.s[si].C; ++si; }
        const bool on = si < p.nsrc;
        t.xp[j] = on ? p.s[si].x + jj : p.s[0].x;
        t.xss[j] = on ? p.s[si].xs : p.s[0].xs;
        t.xld[j] = on ? p.s[si].ldx : p.s[0].ldx;
    }
}

// ---- forward, narrow input: y[n, r, f] = act(bias + sum_s sum_c x_s[n, r, c] W_s[c, f]), sum_s C_s <= 8 ------------------------
// thread = (row, quad of output columns); the input values of a row are broadcast loads, the weights sit in LDS (zero rows for
// the unused channel slots); four rows per thread and pass, all their loads issued before the first use.
__global__ __launch_bounds__(256) void fwd_narrow_in_kernel(GconvParams p, int lpr) {
    extern __shared__ float wl[];                     // [NARROW_MAXC][4 * lpr]
    const int FS = 4 * lpr;
    for (int o = threadIdx.x; o < NARROW_MAXC * FS; o += 256) wl[o] = 0.f;
    __syncthreads();
    int sc = 0;
    for (int si = 0; si < p.nsrc; ++si) {
        const SrcDev &S = p.s[si];
        for (int o = threadIdx.x; o < S.C * FS; o += 256) {
            const int c = o / FS, f = o % FS;
            if (f < p.F) wl[(sc + c) * FS + f] = S.w[(long long)c * S.wrs + (long long)f * S.wcs];
        }
        sc += S.C;
    }
    __syncthreads();
    NarrowTab T;
    narrow_table(p, T);
    const int fq = threadIdx.x % lpr, rl = threadIdx.x / lpr, RL = 256 / lpr;
    const int f = 4 * fq;
    const unsigned rows = (unsigned)p.N * (unsigned)p.Mo;          // < 2^31 (host check)
    float4 wq[NARROW_MAXC];
#pragma unroll
    for (int j = 0; j < NARROW_MAXC; ++j) wq[j] = *reinterpret_cast<const float4 *>(&wl[j * FS + f]);
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias_mode == CAPE_BIAS_CHANNEL && f < p.F) bq = *reinterpret_cast<const float4 *>(p.bias + f);
    constexpr int UR = 4;
    for (unsigned nr0 = blockIdx.x * RL * UR + rl; nr0 < rows; nr0 += gridDim.x * RL * UR) {
        float xv[UR][NARROW_MAXC];
        unsigned nq[UR], rq[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const unsigned nr = min(nr0 + u * RL, rows - 1);
            nq[u] = nr / (unsigned)p.Mo;
            rq[u] = nr - nq[u] * (unsigned)p.Mo;
#pragma unroll
            for (int j = 0; j < NARROW_MAXC; ++j) xv[u][j] = T.xp[j][(long long)nq[u] * T.xss[j] + (long long)rq[u] * T.xld[j]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            float4 a = bq;
            if (p.bias_mode == CAPE_BIAS_VERTEX && f < p.F) a = *reinterpret_cast<const float4 *>(p.bias + (long long)rq[u] * p.F + f);
#pragma unroll
            for (int j = 0; j < NARROW_MAXC; ++j) {
                a.x = fmaf(xv[u][j], wq[j].x, a.x); a.y = fmaf(xv[u][j], wq[j].y, a.y);
                a.z = fmaf(xv[u][j], wq[j].z, a.z); a.w = fmaf(xv[u][j], wq[j].w, a.w);
            }
            a.x = cape_act(a.x, p.act); a.y = cape_act(a.y, p.act); a.z = cape_act(a.z, p.act); a.w = cape_act(a.w, p.act);
            if (f < p.F && nr0 + u * RL < rows)
                *reinterpret_cast<float4 *>(p.y + (long long)nq[u] * p.ys + (long long)rq[u] * p.ldy + f) = a;
        }
    }
}

// ---- weight gradient, narrow input: dW_s[c, f] = sum_{n, r} x_s[n, r, c] dz[n, r, f], sum_s C_s <= 8 ------------------------------
// block = one split (sample group x row range) of the plan, all sources; thread = (row lane, quad of columns) with the channel sums
// of its quad in registers; the row lanes of a block are summed through LDS in index order; slab layout of the plan.
__global__ __launch_bounds__(256) void dw_narrow_in_kernel(DwParams p, int lpr, int ntiles) {
    extern __shared__ float red[];                    // [RL][NARROW_MAXC][4 * lpr]
    int tile, split;
    if (!cape_map_dw_block(blockIdx.x, ntiles, p.ngroups * p.rsplit, tile, split) || tile != 0) return;
    const int grp = split / p.rsplit, rs = split % p.rsplit;
    const int n_begin = grp * p.samples_per_group, n_end = min(p.N, n_begin + p.samples_per_group);
    const int ra = rs * p.rows_per_split, rb = min(p.Mo, ra + p.rows_per_split);
    const int FS = 4 * lpr, RL = 256 / lpr;
    const int fq = threadIdx.x % lpr, rl = threadIdx.x / lpr;
    const int f = 4 * fq;
    const bool fon = f < p.F;
    NarrowTab T;
    narrow_table(p, T);
    float acc[NARROW_MAXC][4];
#pragma unroll
    for (int j = 0; j < NARROW_MAXC; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f;
    constexpr int UR = 4;
    for (int n = n_begin; n < n_end; ++n) {
        const float *zb = p.dz + (long long)n * p.dzs + (fon ? f : 0);
        for (int r0 = ra + rl; r0 < rb; r0 += UR * RL) {
            float4 d[UR];
            float xv[UR][NARROW_MAXC];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int rc = min(r0 + u * RL, rb - 1);
                d[u] = *reinterpret_cast<const float4 *>(zb + (long long)rc * p.lddz);
#pragma unroll
                for (int j = 0; j < NARROW_MAXC; ++j) xv[u][j] = T.xp[j][(long long)n * T.xss[j] + (long long)rc * T.xld[j]];
            }
            __builtin_amdgcn_sched_barrier(0);         // keep the pass's loads together (the scheduler sinks them to their uses)
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const float okf = (r0 + u * RL < rb && fon) ? 1.f : 0.f;
                const float4 dd = make_float4(d[u].x * okf, d[u].y * okf, d[u].z * okf, d[u].w * okf);
#pragma unroll
                for (int j = 0; j < NARROW_MAXC; ++j) {
                    acc[j][0] = fmaf(xv[u][j], dd.x, acc[j][0]); acc[j][1] = fmaf(xv[u][j], dd.y, acc[j][1]);
                    acc[j][2] = fmaf(xv[u][j], dd.z, acc[j][2]); acc[j][3] = fmaf(xv[u][j], dd.w, acc[j][3]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NARROW_MAXC; ++j)          // (slots beyond the sources hold sums of source 0: never read below)
        *reinterpret_cast<float4 *>(&red[(rl * NARROW_MAXC + j) * FS + f]) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    __syncthreads();
    float *out = p.ws + (long long)split * p.slab;
    int j0 = 0;
    for (int si = 0; si < p.nsrc; ++si) {
        const int Cs = p.s[si].C;
        for (int o = threadIdx.x; o < Cs * lpr; o += 256) {
            const int c = o / lpr, q = o % lpr;
            if (4 * q >= p.F) continue;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int l = 0; l < RL; ++l) {
                const float4 v = *reinterpret_cast<const float4 *>(&red[(l * NARROW_MAXC + j0 + c) * FS + 4 * q]);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4 *>(out + p.part_off[si] + (long long)c * p.F + 4 * q) = s;
        }
        j0 += Cs;
    }
}

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

// which narrow form (0 = none, 5 = narrow input, 6 = narrow output) a weight-gradient launch of family ``fam`` takes
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
    const bool dzal = (lddz & 3) == 0 && (dz_sample_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0;
    if (sumC <= NARROW_MAXC && (F & 3) == 0 && F >= 16 && F <= 256 && dzal) return 5;
    if (F <= 4 && c4 && xal && sumC >= 16 && sumC <= 256) return 6;
    return 0;
}
