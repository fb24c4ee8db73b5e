/*
 * cape_hip.h -- C-ABI of libcape_hip.so: the MI355X (gfx950) implementation of CAPE's
 * Chebyshev mesh-convolution hot path.
 *
 * Every entry point is extern "C", takes caller-owned DEVICE pointers + explicit sizes +
 * a hipStream_t (passed as void*), performs no allocation and no synchronisation, and
 * returns 0 on success, a negative CAPE_E* code for argument errors or a positive
 * hipError_t for launch errors.  Process-wide state: no buffers, no caches, nothing a caller
 * can change through the ABI -- but the library does latch a set of kernel-SELECTION
 * switches from the environment on first use (CAPE_GEMM_H2, CAPE_DW_H2, CAPE_H2_TILE, CAPE_H2X, CAPE_DW_V4,
 * CAPE_GEMM_BF16X6[_DUAL], CAPE_DW_BF16X6, CAPE_GEMM_PLAIN, CAPE_DW_PLAIN, CAPE_NARROW,
 * CAPE_FC_MFMA, CAPE_SPMM_UNROLL, CAPE_SPMM_WIDE; INTEGRATION.md lists them): they choose
 * among kernel families that compute the same function and are all held to the same parity
 * tests; with none of them set (the default every test and bench line runs) behaviour
 * depends on the arguments alone.  Results are deterministic (no floating-point atomics; the
 * one integer ticket, in cape_flat_adam_update, orders no floating-point operation).  Tensors are fp32, row-major [N, M, ld] with channels
 * contiguous (ld >= C is the row stride in elements), i.e. the reference's [N, M, F]
 * placeholder layout (reference lib/models.py:272-282) without its [M, F*N] shuffles
 * (lib/models.py:81-83, 97-99, 147-151).
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   cape_gconv_fwd      lib/models.py:69-103  chebyshev5   (tf.sparse_tensor_dense_matmul x(K-1)
 *                                                           + tf.matmul), fused with
 *                       lib/models.py:105-127 b1leakyrelu / b1relu / b1tanh / b2relu,
 *                       lib/models.py:129-152 poolwT       (pool folded in as output-row CSR,
 *                                                           unpool folded in as input CSR),
 *                       lib/models.py:776-793 res_block_affine (DUAL accumulator mode),
 *                       lib/models.py:611-616 per-vertex output bias;
 *                       and, with transposed operators/weights, the data gradient that
 *                       tf.gradients (lib/models.py:460,465) derives for those ops.
 *   cape_gconv_dw       the weight gradient of the same ops (tf.gradients, :460).
 *   cape_cheb_fused_fwd / cape_cheb_fused_bwd
 *                       lib/models.py:69-103 for polynomial orders above the precomposed-operator limit: the explicit
 *                       recurrence of :88-96 kept on chip (one launch per direction; BASELINE configs[1]).
 *   cape_spmm           lib/models.py:91,94,149 SparseTensorDenseMatMul as a standalone op
 *                       (general-K Chebyshev recurrence, non-selection pool matrices).
 *   cape_bias_act_fwd / cape_act_bwd / cape_colsum
 *                       lib/models.py:105-127 standalone and their gradients.
 *   cape_mask_mul       gradient of tf.nn.relu in res_block_affine (lib/models.py:785).
 *   cape_bwd_prep / cape_bwd_prep_spmm / cape_spmm_multi_prep
 *                       the non-GEMM part of a conv layer's backward under tf.gradients (:460): activation / ReLU-mask
 *                       gradient (:105-127, :785), bias and rank-1 condition sums, and -- the two fused forms, for
 *                       res_block_affine (:776-793) -- the operator applications S_k^T dz of its data gradient.
 *   cape_fill_cond      lib/models.py:813-832 fit_cond_dim + tf.concat (:535,593,608,665).
 *   cape_groupnorm_fwd / cape_groupnorm_bwd
 *                       lib/models.py:681-712 gn (norm_type='group') and its gradient.
 *   cape_gan_bce_fwd_bwd lib/models.py:381-390 the adversarial loss terms and their gradients w.r.t. the logits.
 *   cape_recon_edge_loss_fwd_bwd
 *                       lib/models.py:357-375 L1 reconstruction + lib/losses.py:9-25 edge loss
 *                       and their gradients w.r.t. the prediction.
 *   cape_csr_validate   host-side structural check of an operator before upload.
 */
#ifndef CAPE_HIP_H
#define CAPE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPE_ABI_VERSION 13
#define CAPE_MAX_SRC 8

/* error codes (negative = argument error; positive values are hipError_t) */
#define CAPE_OK 0
#define CAPE_EINVAL (-1)       /* bad size / null pointer / unsupported combination */
#define CAPE_EUNSORTED (-2)    /* CSR column indices not strictly increasing in a row */
#define CAPE_ERANGE (-3)       /* CSR index out of range / rowptr not monotone */
#define CAPE_EWORKSPACE (-4)   /* workspace too small */

/* activation selectors (lib/models.py:105-127) */
#define CAPE_ACT_NONE 0
#define CAPE_ACT_LEAKY 1       /* tf.nn.leaky_relu, alpha = 0.2 */
#define CAPE_ACT_RELU 2
#define CAPE_ACT_TANH 3

/* bias selectors */
#define CAPE_BIAS_NONE 0
#define CAPE_BIAS_CHANNEL 1    /* [1,1,F]  (b1*)                */
#define CAPE_BIAS_VERTEX 2     /* [1,M,F]  (b2relu, :615 output) */

/*
 * One operand ("source") of the fused gather-GEMM:
 *   A[n, r, c] = sum_e vals[e] * x[n, colidx[e], c]   e in [rowptr[r], rowptr[r+1])
 *              = x[n, r, c]                            when rowptr == NULL (identity)
 *   acc1[n, r, f] += sum_c A[n, r, c] * w [c*w_rs  + f*w_cs ]
 *   acc2[n, r, f] += sum_c A[n, r, c] * w2[c*w2_rs + f*w2_cs]   (only if w2 != NULL)
 * For the forward Chebyshev term k of a layer with reference weight W[Fin*K, Fout]
 * (row index fin*K + k, lib/models.py:97-101):  w = W + k*Fout, w_rs = K*Fout, w_cs = 1.
 * For its data gradient: x = dZ, CSR = transposed operator, w = W + k*Fout, w_rs = 1,
 * w_cs = K*Fout.
 */
typedef struct cape_src {
    const float *x;          /* device [N, Mi, ldx] (pointer may include a channel offset) */
    int64_t x_sample_stride; /* elements between consecutive samples                       */
    int32_t ldx;             /* row stride in elements                                     */
    int32_t C;               /* channels contracted from this source                       */
    const int32_t *rowptr;   /* device, Mo+1 entries, or NULL for identity                 */
    const int32_t *colidx;   /* device, nnz                                                */
    const float *vals;       /* device, nnz                                                */
    const float *w;          /* device weight block 1                                      */
    int64_t w_rs, w_cs;
    const float *w2;         /* device weight block 2 (DUAL mode) or NULL                  */
    int64_t w2_rs, w2_cs;
} cape_src_t;

/*
 * Rank-1 epilogue terms of the fused gather-GEMM: vertex-constant input channels (the tiled
 * condition vector, lib/models.py:813-832 fit_cond_dim + tf.concat at :593,608,665) need no
 * per-vertex contraction because S_k (1 c^T) = (S_k 1) c^T.  Their contribution to the
 * pre-activation is   sum_j rowscale[j, r] * coef[n, j, f]   with rowscale_j = S_k 1 (row sums of the
 * operator) and coef_j = cond @ W_k[cond rows] (a tiny dense product done by the caller).
 */
typedef struct cape_rank {
    int32_t R;             /* number of terms, 0..CAPE_MAX_SRC                                  */
    const float *rowscale; /* device [R, Mo]                                                    */
    const float *coef;     /* device [N, R, F]                                                  */
    uint32_t to_acc2;      /* bit j set: term j goes to the second accumulator (DUAL mode)      */
} cape_rank_t;

/*
 * Condition coefficients of all consumers of one condition vector (cape_rank_t.coef of every layer) in one
 * launch, and their gradients.  Per layer (pointers are to the CONDITION rows of the layer's weights):
 *   w      [Cc*K, F]  = W + Ch*K*F   (row c*K + k; lib/models.py:97-101 layout restricted to the tiled channels)
 *   w_aff  NULL or [Cc, F] = W_affine + Ch*F      (res_block_affine, :776-793)
 *   coef   out [N, K (+1), F]   coef[n,k,f] = sum_c cond[n,c] w[c*K+k, f];  coef[n,K,f] = sum_c cond[n,c] w_aff[c,f]
 *   dcoef  in  [N, K (+1), F]   gradient of the loss w.r.t. coef (cape_bwd_prep writes it)
 *   gw, gw_aff  out (may be NULL): gradient rows, same layout as w / w_aff
 */
#define CAPE_MAX_COND_LAYERS 16
typedef struct cape_cond_layer {
    const float *w;
    const float *w_aff;
    float *coef;
    const float *dcoef;
    float *gw;
    float *gw_aff;
    int32_t K, F;
} cape_cond_layer_t;

int cape_abi_version(void);

/* Measurement utility: keep the stream busy for about ``us`` microseconds (one wave spinning on the 100 MHz wall clock, no
 * memory traffic).  bench.py's per-launch timing enqueues it in front of every bracketed launch, so that the start event, the
 * launch and the stop event are all queued while it runs and the bracket measures the kernel, not the host's launch path. */
int cape_spin_us(int32_t us, void *stream);

/* Host-side structural check of a CSR operator (host pointers). */
int cape_csr_validate(int32_t rows, int32_t cols, int64_t nnz, const int32_t *rowptr,
                      const int32_t *colidx);

/*
 * Fused gather-GEMM forward.
 *   single mode (mask_out == NULL, no source has w2):
 *        y[n,r,f] = act(acc1 + bias)
 *   DUAL mode (at least one source has w2)  -- res_block_affine, lib/models.py:776-793:
 *        y[n,r,f] = relu(acc1) + acc2 ; if mask_out != NULL bit (f%32) of
 *        mask_out[(n*Mo + r)*ceil(F/32) + f/32] = (acc1 > 0)
 * bias: NULL or [F] (CAPE_BIAS_CHANNEL) or [Mo,F] (CAPE_BIAS_VERTEX).
 * rank: NULL or rank-1 terms added to the accumulators before the epilogue.
 * out_deinterleave = K > 1 (single mode, no rank/mask, F % K == 0): output column j = c*K + k is stored at channel
 *   k * round_up(F/K, 4) + c, i.e. the reference's fin*K + k weight-row order (lib/models.py:97-101) comes out as
 *   K contiguous channel blocks -- ONE launch  G = dz W^T  for all K orders of a data gradient, each block then
 *   being the operand of its S_k^T (ldy >= K * round_up(F/K, 4)).  0 / 1 = off.
 * Arithmetic: fp32 in, fp32 out, fp32 accumulation.  Single-mode launches on plain (no CSR) sources whose channel
 *   counts are multiples of 32, with F >= 64, multiply on the bf16 matrix pipe: each fp32 operand is split exactly
 *   into three bf16 pieces and six cross products per multiply-add are accumulated in fp32 -- as accurate as an fp32
 *   FMA chain, not bit-identical to the other kernels' exact-fp32 MFMA, and an inf operand gives
 *   NaN.  Environment CAPE_GEMM_BF16X6=0 (read once per process) keeps every launch on the exact-fp32 MFMA.
 */
int cape_gconv_fwd(const cape_src_t *srcs, int32_t nsrc, float *y, int64_t y_sample_stride,
                   int32_t ldy, int32_t N, int32_t Mo, int32_t F, const float *bias,
                   int32_t bias_mode, int32_t act, uint32_t *mask_out, const cape_rank_t *rank,
                   int32_t out_deinterleave, void *stream);

/*
 * fp16 two-piece operands ("h2") of the trailing contraction (reference lib/models.py:99-102) -- the default arithmetic of
 * every eligible fp32 launch since ABI 11: each fp32 operand, scaled by a power of two, is the sum of two fp16 numbers
 * (22 significant bits) and three cross products per multiply-add are accumulated in fp32 on v_mfma_f32_32x32x16_f16; as
 * accurate as an fp32 FMA chain, not bit-identical to the other kernels.  The scales come from
 *   - per activation row: a BOUND of the row's absolute maximum, cape_h2_src_t.rowmax [N, rows, rowmax_w] (rowmax_w a
 *     multiple of 4; the bound of a row is the maximum over its rowmax_w entries), written by the kernel that produced the
 *     tensor (rowmax_out below and of cape_spmm* / cape_bwd_prep) or by cape_rowmax;
 *   - per weight column: the piece planes and reciprocal scales cape_weight_pieces prepares once per step.
 * Launches whose sources are plain (no CSR), whole 32-channel chunks, 16-byte aligned, with F >= 64 and all of the above
 * given take gemm_h2_kernel; everything else is computed as before (the row-bound output works with every kernel).
 */
typedef struct cape_h2_src {
    const uint16_t *w_hi, *w_lo;   /* fp16 pieces of weight block 1: element (f, c) at [f * w_pitch + c] (contraction contiguous) */
    int64_t w_pitch;
    const uint16_t *w2_hi, *w2_lo; /* of weight block 2 (DUAL), NULL otherwise                                                  */
    int64_t w2_pitch;
    const float *rowmax;           /* row bounds of x: [N, rows, rowmax_w]                                                      */
    int32_t rowmax_w;
} cape_h2_src_t;
typedef struct cape_h2 {
    const cape_h2_src_t *src;      /* nsrc entries, or NULL (then only rowmax_out is used)                                      */
    const float *wscale_inv;       /* [F] reciprocal column scales of the block-1 planes                                        */
    const float *w2scale_inv;      /* [F] of the block-2 planes (DUAL)                                                          */
    float *rowmax_out;             /* NULL or [N, Mo, rowmax_out_w]: entry j of row r = bound of |y[n, r, 32 j .. 32 j + 31]|    */
    int32_t rowmax_out_w;          /* multiple of 4, >= ceil(F / 32); entries beyond ceil(F / 32) are written as 0              */
} cape_h2_t;
/* cape_gconv_fwd with the operands above (h2 == NULL: identical to cape_gconv_fwd).  fp32 storage only. */
int cape_gconv_fwd_h2(const cape_src_t *srcs, int32_t nsrc, float *y, int64_t y_sample_stride,
                      int32_t ldy, int32_t N, int32_t Mo, int32_t F, const float *bias,
                      int32_t bias_mode, int32_t act, uint32_t *mask_out, const cape_rank_t *rank,
                      int32_t out_deinterleave, const cape_h2_t *h2, void *stream);
/* plan query of cape_gconv_fwd_h2: plan[0] = 3 for gemm_h2_kernel, otherwise as cape_gconv_fwd_plan */
int cape_gconv_fwd_plan_h2(const cape_src_t *srcs, int32_t nsrc, int32_t N, int32_t Mo, int32_t F, const cape_h2_t *h2,
                           int32_t plan[4]);

/* Row bounds of a tensor whose producer wrote none: out[(n*M + r)*out_w] = max_c |x[n, r, c]|, the other entries 0. */
int cape_rowmax(const float *x, int64_t x_sample_stride, int32_t ldx, int32_t N, int32_t M, int32_t C, float *out,
                int32_t out_w, void *stream);

/*
 * Piece planes of Chebyshev-layer weights W[Ch*K (+ further rows), F], row c*K + k (lib/models.py:97-101), for all layers
 * of a model in two launches (maxima, planes):
 *   forward   f_hi / f_lo [K][F][Ch]:  piece(W[c*K+k, f] * sf[f]),  fscale_inv[k*F + f] = 1 / sf[f]   (replicated over k)
 *   backward  b_hi / b_lo [Ch*K][F]:   piece(W[c*K+k, f] * sb[c]),  bscale_inv[c*K + k] = bscale_c_inv[c] = 1 / sb[c]
 * Forward source k of a layer: w_hi = f_hi + k*F*Ch, w_pitch = Ch, wscale_inv = fscale_inv (the coarse form with K*F output
 * columns reads all K blocks as one [K*F][Ch] plane).  Data gradient: source k: w_hi = b_hi + k*F, w_pitch = K*F,
 * wscale_inv = bscale_c_inv (the de-interleaved single launch: w_hi = b_hi, w_pitch = F, column j = c*K + k, bscale_inv).
 * Ch and F must be multiples of 8.  The item table and the two block-offset tables live in DEVICE memory (they are static
 * per model); cape_weight_pieces_blocks computes the offset tables on the host.
 */
typedef struct cape_wpiece_item {
    const float *w;
    int32_t Ch, K, F, pair_K;
    const float *pair_w;           /* NULL, or a second weight tensor [Ch*pair_K (+ ...), F] whose data-gradient term adds into the
                                    * same accumulator (res_block_affine: graph_conv + affine, lib/models.py:780-789): the
                                    * backward scale of channel c then covers the rows of BOTH tensors                     */
    uint16_t *f_hi, *f_lo, *b_hi, *b_lo;
    float *fscale_inv, *bscale_inv;
    float *bscale_c_inv;           /* [Ch]: 1 / sb[c] once per channel (the multi-source data-gradient form's column index) */
    const float *fpair_w;          /* NULL, or a second weight tensor [fpair_rows, F] whose FORWARD product adds into the same
                                    * accumulator (the two-source tail of res_block_decoder, lib/models.py:763-774): the forward
                                    * scale of column f then covers the columns of both tensors                             */
    int32_t fpair_rows, reserved;
    float *colmax_partial;         /* scratch [ceil((Ch*K + fpair_rows) / 64)][F]: partial column maxima of the first pass      */
} cape_wpiece_item_t;
int cape_weight_pieces_blocks(const cape_wpiece_item_t *host_items, int32_t nitems, int32_t *max_off, int32_t *planes_off);
int cape_weight_pieces(const cape_wpiece_item_t *dev_items, int32_t nitems, const int32_t *dev_max_off, int32_t max_blocks,
                       const int32_t *dev_planes_off, int32_t planes_blocks, void *stream);

/* Which kernel cape_gconv_fwd would run for these arguments (pure query, no launch):
 * plan[0] = family (0: gather-GEMM gconv_fwd_kernel, 1: pipelined plain-source gemm_plain_kernel, 2: the same on
 * the bf16 pipe with the exact three-way operand split, gemm_split_kernel),
 * plan[1], plan[2] = workgroup tile rows x columns, plan[3] = weight layout of families 1 and 2
 * (1: contraction-contiguous, 0: output-contiguous).  Used by bench.py to attribute per-launch work to
 * the kernel names rocprofv3 reports. */
int cape_gconv_fwd_plan(const cape_src_t *srcs, int32_t nsrc, int32_t N, int32_t Mo, int32_t F, int32_t plan[4]);

/*
 * Gradient of the rank-1 terms w.r.t. coef:  out[n, j, f] = sum_r rowscale[j, r] * dz[n, r, f]
 * (deterministic two-stage reduction; workspace >= cape_rowscale_reduce_workspace_bytes).
 */
int64_t cape_rowscale_reduce_workspace_bytes(int32_t N, int32_t Mo, int32_t F, int32_t R);
int cape_rowscale_reduce(const float *dz, int64_t dz_sample_stride, int32_t lddz,
                         const float *rowscale, int32_t R, int32_t N, int32_t Mo, int32_t F,
                         float *out, void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Weight gradient of the fused gather-GEMM:
 *   dW_s[c*w_rs + f*w_cs] (+)= sum_{n,r} A_s[n,r,c] * dz_s[n,r,f]
 * for every source s (w is the *output* gradient block here and is written, w2 is ignored).
 * dz_s = dz2 for the sources whose bit is set in dz2_mask (dz2 has the layout of dz), dz otherwise:
 * res_block_affine's two weight blocks contract the same inputs against different gradients (the
 * ReLU-masked one and the raw one) and are produced by ONE launch.  accumulate != 0 adds to the existing
 * contents.  Uses a caller-provided workspace of at least cape_gconv_dw_workspace_bytes(...) bytes.
 */
int64_t cape_gconv_dw_workspace_bytes(const cape_src_t *srcs, int32_t nsrc, int32_t N,
                                      int32_t Mo, int32_t F);
int cape_gconv_dw(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                  int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                  int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                  int64_t workspace_bytes, void *stream);

/* The two stages of cape_gconv_dw separately (stage 1: contraction into the partial slabs of the workspace; stage 2:
 * their fixed-order reduction into the gradient blocks; stage 0 = both, identical to cape_gconv_dw).  Lets a caller
 * bracket each kernel with its own events (bench.py's per-kernel roofline). */
int cape_gconv_dw_stage(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                        int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                        int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                        int64_t workspace_bytes, int32_t stage, void *stream);

/* The weight gradient with fp16 two-piece operands (see cape_h2_t): row bounds of every source and of dz / dz2; launches that
 * would take dw_split_kernel (plan family 3) then run dw_h2_kernel (family 4) on the same tiles with HALF the contraction
 * splits (one workgroup per CU: the kernel overlaps loads, operand split and MFMAs inside each wave; half the partial slabs).
 * The reduction must count the same slabs: stage 2 through cape_gconv_dw_stage_h2 with the same h2, batched reductions with
 * cape_dw_item_t::h2 set.  h2 == NULL: identical to cape_gconv_dw_stage / _plan. */
typedef struct cape_h2_dw {
    const float *src_rowmax[CAPE_MAX_SRC];
    int32_t src_rowmax_w[CAPE_MAX_SRC];
    const float *dz_rowmax;
    int32_t dz_rowmax_w;
    const float *dz2_rowmax;
    int32_t dz2_rowmax_w;
} cape_h2_dw_t;
int cape_gconv_dw_stage_h2(const cape_src_t *srcs, int32_t nsrc, const float *dz,
                           int64_t dz_sample_stride, int32_t lddz, const float *dz2, uint32_t dz2_mask,
                           int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                           int64_t workspace_bytes, int32_t stage, const cape_h2_dw_t *h2, void *stream);
int cape_gconv_dw_plan_h2(const cape_src_t *srcs, int32_t nsrc, const float *dz, int64_t dz_sample_stride, int32_t lddz,
                          const float *dz2, uint32_t dz2_mask, int32_t N, int32_t Mo, int32_t F, const cape_h2_dw_t *h2,
                          int32_t plan[4]);

/* The reductions (stage 2) of up to CAPE_MAX_DW_REDUCE_ITEMS earlier stage-1 calls in ONE launch: each item repeats the
 * arguments of its cape_gconv_dw_stage(..., stage = 1, ...) call (bf16 != 0: it was the _bf16 entry); the workspaces must
 * still hold the partial slabs.  The training step defers every layer's reduction to the end of the backward pass. */
#define CAPE_MAX_DW_REDUCE_ITEMS 12
typedef struct cape_dw_item {
    const cape_src_t *srcs;
    int32_t nsrc;
    const void *dz;
    int64_t dz_sample_stride;
    int32_t lddz;
    const void *dz2;
    uint32_t dz2_mask;
    int32_t N, Mo, F, accumulate, bf16;
    void *workspace;
    int64_t workspace_bytes;
    const struct cape_h2_dw *h2;      /* the operands' row bounds when the contraction went through cape_gconv_dw_stage_h2 (NULL
                                         otherwise): the two-piece kernel writes fewer, fatter partial slabs, and the reduction
                                         has to count the same ones */
} cape_dw_item_t;
int cape_gconv_dw_reduce_batch(const cape_dw_item_t *items, int32_t nitems, void *stream);

/* Which kernel cape_gconv_dw would run for these arguments (pure query, no launch): plan[0] = family (0: gather form
 * gconv_dw_kernel, 1: dw_plain_kernel on the exact-fp32 MFMA, 2: dw_packed_kernel, 3: dw_split_kernel on the bf16 pipe
 * with the exact three-way operand split), plan[1], plan[2] = tile channels x output columns, plan[3] = number of
 * partial slabs the fixed-order reduction sums.  The parity tests use it to prove that every kernel the benchmarked
 * step launches is exercised by a passing comparison against the oracle. */
int cape_gconv_dw_plan(const cape_src_t *srcs, int32_t nsrc, const float *dz, int64_t dz_sample_stride, int32_t lddz,
                       const float *dz2, uint32_t dz2_mask, int32_t N, int32_t Mo, int32_t F, int32_t plan[4]);

/*
 * One pass over the incoming gradient g [N, Mo, F] that produces everything the backward of a conv
 * layer needs besides the GEMMs (replaces cape_act_bwd / cape_mask_mul + cape_colsum +
 * cape_rowscale_reduce, i.e. 3-7 launches and as many passes over g):
 *   dz[n,r,f]      = g * act'(y)            (y = activation OUTPUT; act = CAPE_ACT_*)            or
 *                  = g if mask bit else 0   (mask != NULL: ReLU of res_block_affine, lib/models.py:785)
 *   dbias[f]       = sum_{n,r} dz           (dbias != NULL; channel bias of lib/models.py:105-121)
 *   dcoef[n,j,f]   = sum_r rowscale[j,r] * dz[n,r,f]   j < R      (rank-1 condition terms)
 *   dcoef_g[n,f]   = sum_r rowscale[rg,r] * g[n,r,f]              (dcoef_g != NULL: affine branch)
 * dcoef_sample_stride: elements between samples of BOTH dcoef and dcoef_g (so that the two can be slices of one
 * [N, R+1, F] buffer, the layout cape_cond_coef_bwd reads); 0 = contiguous (R*F and F).
 * dz may alias g (in place); with act = CAPE_ACT_NONE and mask = NULL an aliased dz is not even stored (dz IS g: the call
 * is made for its sums).  Deterministic two-stage reductions; workspace >= cape_bwd_prep_workspace_bytes.
 */
int64_t cape_bwd_prep_workspace_bytes(int32_t N, int32_t Mo, int32_t F, int32_t R);
int cape_bwd_prep(const float *g, int64_t g_sample_stride, int32_t ldg, const float *y,
                  int64_t y_sample_stride, int32_t ldy, int32_t act, const uint32_t *mask, float *dz,
                  int64_t dz_sample_stride, int32_t lddz, float *dbias, const float *rowscale,
                  int32_t R, float *dcoef, int32_t rg, float *dcoef_g, int64_t dcoef_sample_stride, int32_t finalize,
                  int32_t N, int32_t Mo, int32_t F, void *workspace, int64_t workspace_bytes, float *rowmax_out, void *stream);

/* finalize = 0 above leaves the reductions as partial slabs in the workspace; this entry finishes up to
 * CAPE_MAX_BWD_PREP_ITEMS of them in ONE launch (same N, Mo, F, R and destinations as the deferred calls, whose
 * workspaces must still be intact).  dz is always complete after cape_bwd_prep itself. */
#define CAPE_MAX_BWD_PREP_ITEMS 16
typedef struct cape_bwd_prep_item {
    const void *workspace;
    int32_t N, Mo, F, R;
    float *dbias;
    float *dcoef;
    float *dcoef_g;
    int64_t dcoef_sample_stride;
    int32_t chunks;            /* 0: the partial layout of cape_bwd_prep itself; > 0: partials of another producer with this many
                                * chunks per sample (cape_spmm_multi_actgrad)                                                  */
} cape_bwd_prep_item_t;
int cape_bwd_prep_finalize(const cape_bwd_prep_item_t *items, int32_t nitems, void *stream);

/* y[n,r,:] = alpha * sum_e vals[e]*x[n,colidx[e],:] + beta * z[n,r,:]   (z may be NULL, may alias y).
 * max_row_nnz: upper bound on the entries of any row if the caller knows it, 0 = unknown.  A hint only: it picks
 * the width of the unrolled entry groups (4 for the up-/down-sampling matrices, 8 otherwise); any row length is
 * handled either way.
 * ell_width: 0 = colidx / vals are the CSR arrays.  4, 8 or 12 = they are the ELL form of the same operator: [rows, ell_width]
 * arrays (16-byte aligned), every row's entries in CSR order packed to the front, the slots past its end holding
 * (column of slot 0, 0.0f); rowptr is then not read (but must still be non-NULL: NULL means "identity" in the term lists
 * below).  Same sums bit for bit; two dependent memory round trips per row of <= 8 entries instead of five. */
int cape_spmm(const float *x, int64_t x_sample_stride, int32_t ldx, const int32_t *rowptr,
              const int32_t *colidx, const float *vals, int32_t max_row_nnz, int32_t ell_width, float alpha, const float *z,
              int64_t z_sample_stride, int32_t ldz, float beta, float *y,
              int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t Mo, int32_t C,
              float *rowmax_out, void *stream);

/*
 * Several operator applications in one launch (all operators have Mo rows, all operands C channels):
 *   sum = 0:  terms[k].y = S_k x_k for every k   (X_k = S_k x of one layer; T_k = S_k^T dz of its data gradient)
 *   sum = 1:  y = sum_k scale_k S_k x_k          (dx = sum_k S_k^T G_k; a Clenshaw step 2 L~^T b_{k+1} + G_k - b_{k+2})
 * A term with rowptr == NULL is the identity (its input then has Mo rows).  y / terms[k].y must not alias an input.
 */
#define CAPE_MAX_SPMM_TERMS 4
/* rowmax_out (cape_spmm, cape_spmm_multi in sum mode, cape_spmm_combine, cape_bwd_prep; fp32 storage only): NULL or
 * [N, Mo, 4]: the bound of max_c |out[n, r, c]| of the row the call writes (y / dz; cape_bwd_prep bounds its INPUT g, which
 * bounds dz as well), as cape_h2_src_t.rowmax with
 * rowmax_w = 4 -- written by the same kernel where the lanes of a row form one power-of-two group, by a standalone pass
 * (cape_rowmax) inside the call otherwise. */
typedef struct cape_spmm_term {
    const float *x;
    int64_t x_sample_stride;
    int32_t ldx;
    const int32_t *rowptr;
    const int32_t *colidx;
    const float *vals;
    float *y;
    int64_t y_sample_stride;
    int32_t ldy;
    float scale;             /* the term is scale * S_k x_k (1.0f for a plain application) */
    int32_t ell_width;       /* 0: colidx / vals in CSR form; 4 / 8 / 12: ELL form (see cape_spmm)      */
    float *rowmax_out;       /* separate mode of cape_spmm_multi (fp32): NULL or [N, Mo, 4] row bounds of y (cape_h2_src_t.rowmax) */
} cape_spmm_term_t;
int cape_spmm_multi(const cape_spmm_term_t *terms, int32_t nterms, int32_t sum, float *y, int64_t y_sample_stride,
                    int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream);

/*
 * cape_spmm_multi in sum mode with the activation gradient of the layer BELOW fused in (fp32, vector form): the summed operator
 * application is that layer's incoming gradient g (reference cnp chain, lib/models.py:154-171: conv -> bias + (leaky-)ReLU ->
 * pool, differentiated by tf.gradients :460); act_x [N, Mo, >= C] is that layer's OUTPUT.  Writes
 *     y[n,r,c] = (sum_k scale_k S_k x_k)[n,r,c] * act'(act_x[n,r,c])            (= dz of the layer below)
 * and the bias-gradient partial sums of y over the rows of every block: bias_partials [N, chunks, 2, C] floats (term 0) in the
 * layout cape_bwd_prep_finalize reads with cape_bwd_prep_item_t.chunks = cape_spmm_multi_actgrad_chunks(...), R = 0.
 * That layer then needs no cape_bwd_prep launch.  act in {CAPE_ACT_LEAKY, CAPE_ACT_RELU}.
 */
int32_t cape_spmm_multi_actgrad_chunks(const float *y, int64_t y_sample_stride, int32_t ldy, const float *act_x,
                                       int64_t act_x_sample_stride, int32_t ld_act_x, int32_t Mo, int32_t C);
int cape_spmm_multi_actgrad(const cape_spmm_term_t *terms, int32_t nterms, float *y, int64_t y_sample_stride, int32_t ldy,
                            int32_t N, int32_t Mo, int32_t C, float *rowmax_out, const float *act_x,
                            int64_t act_x_sample_stride, int32_t ld_act_x, int32_t act, float *bias_partials, void *stream);
/* the same for bf16 activation storage (x_k, y, act_x bf16; act' applied and the bias sums taken in fp32 before the one rounding
 * of y; rowmax_out must be NULL) */
int32_t cape_spmm_multi_actgrad_chunks_bf16(const void *y, int64_t y_sample_stride, int32_t ldy, const void *act_x,
                                            int64_t act_x_sample_stride, int32_t ld_act_x, int32_t Mo, int32_t C);
int cape_spmm_multi_actgrad_bf16(const cape_spmm_term_t *terms, int32_t nterms, void *y, int64_t y_sample_stride, int32_t ldy,
                                 int32_t N, int32_t Mo, int32_t C, float *rowmax_out, const void *act_x,
                                 int64_t act_x_sample_stride, int32_t ld_act_x, int32_t act, float *bias_partials, void *stream);

/*
 * Backward-prep of an affine block fused with the operator application of its data gradient (fp32, vector form, square
 * operator).  Reference: res_block_affine at one resolution, lib/models.py:776-793 (y = relu(conv_K(x)) + conv_1(x), K = 2),
 * differentiated by tf.gradients (:460): with g the incoming gradient and mask the sign bits of conv_K(x) the forward launch
 * wrote ([N, Mo, F/32] words, cape_gconv_fwd dual mode),
 *     dz[n,r,c] = mask bit ? g[n,r,c] : 0                                   (what cape_bwd_prep writes)
 *     t1[n,r,:] = sum_e vals[e] * dz[n,colidx[e],:]                          (what cape_spmm(dz) writes: T_1 = L~^T dz)
 * t1 is bit-identical to the two-launch form (the gathered rows are masked first, then the same fma chain).  The rank-1
 * condition sums go to `partials` in the layout cape_bwd_prep_finalize reads with cape_bwd_prep_item_t.chunks =
 * cape_bwd_prep_spmm_chunks(...):  slot 1 + j: sum_r rowscale[j, r] dz[n,r,:] (j < R);  slot R + 1: sum_r rowscale[rg, r] g[n,r,:]
 * (rg < 0: none);  slot 0 (the bias sum) is NOT written -- pass dbias = NULL to the finalisation.  partials:
 * N * chunks * (R + 2) * F floats; R <= 2.  rowmax_g_out / rowmax_t1_out: NULL or [N, Mo, 4] row bounds of g (which bound dz) / of t1.
 * Needs F % 32 == 0, 16-byte aligned views whose rows split into a power-of-two number (<= 64) of 4- or 8-channel work items;
 * CAPE_EINVAL otherwise (the caller then takes cape_bwd_prep + cape_spmm).  dz and t1 must not alias g.
 */
int32_t cape_bwd_prep_spmm_chunks(const float *g, int64_t g_sample_stride, int32_t ldg, const float *dz, int64_t dz_sample_stride,
                                  int32_t lddz, const float *t1, int64_t t1_sample_stride, int32_t ldt1, int32_t N, int32_t Mo,
                                  int32_t F);
int cape_bwd_prep_spmm(const float *g, int64_t g_sample_stride, int32_t ldg, const uint32_t *mask, const int32_t *rowptr,
                       const int32_t *colidx, const float *vals, int32_t ell_width, float *dz, int64_t dz_sample_stride,
                       int32_t lddz, float *t1, int64_t t1_sample_stride, int32_t ldt1, const float *rowscale, int32_t R,
                       int32_t rg, int32_t N, int32_t Mo, int32_t F, float *partials, int64_t partials_bytes,
                       float *rowmax_g_out, float *rowmax_t1_out, void *stream);

/*
 * The up-sampling form of the same block (res_block_affine behind an unpool, lib/models.py:776-793 + :147-151, under
 * tf.gradients :460): all operator applications of its data gradient, T_k = S_k^T dz (k < K) and T_aff = S_0^T g at the coarse
 * input rows, with dz = mask bit ? g : 0 formed on the fly from the gathered FINE rows of g (terms whose bit is set in
 * masked_terms; mask [N, mask_rows, C/32] words, mask_rows = rows of the terms' inputs), and the column sums of every output
 *     partials[n, block, 1 + k, :] = sum over the block's rows j of terms[k].y[n, j, :]
 * in the layout cape_bwd_prep_finalize reads (chunks = cape_spmm_multi_prep_chunks(...), R = nterms - 1, dcoef_g = the last
 * term; slot 0 not written): sum_r (S_k 1)[r] dz[n,r,:] = sum_j (S_k^T dz)[n,j,:], so the rank-1 condition gradients need no
 * pass over dz, and dz is never written -- the weight gradient of such a block contracts the T_k.  Replaces cape_bwd_prep +
 * cape_spmm_multi(sum = 0).  Every term: a CSR / ELL operator with Mo rows, scale 1, y != NULL (rowmax_out optional); fp32,
 * nterms <= 3, C % 32 == 0, 16-byte aligned views with a power-of-two number (4 .. 64) of work items per row, else CAPE_EINVAL.
 * partials (NULL: no sums): N * chunks * (nterms + 1) * C floats.
 */
int32_t cape_spmm_multi_prep_chunks(const cape_spmm_term_t *terms, int32_t nterms, int32_t N, int32_t Mo, int32_t C);
int cape_spmm_multi_prep(const cape_spmm_term_t *terms, int32_t nterms, uint32_t masked_terms, const uint32_t *mask,
                         int32_t mask_rows, int32_t N, int32_t Mo, int32_t C, float *partials, int64_t partials_bytes,
                         void *stream);
/* both fused forms for bf16 activation storage (g, dz, t1 / the terms' x and y bf16; sums in fp32; no row bounds: the rowmax
 * arguments must be NULL) */
int32_t cape_bwd_prep_spmm_chunks_bf16(const void *g, int64_t g_sample_stride, int32_t ldg, const void *dz, int64_t dz_sample_stride,
                                       int32_t lddz, const void *t1, int64_t t1_sample_stride, int32_t ldt1, int32_t N, int32_t Mo,
                                       int32_t F);
int cape_bwd_prep_spmm_bf16(const void *g, int64_t g_sample_stride, int32_t ldg, const uint32_t *mask, const int32_t *rowptr,
                            const int32_t *colidx, const float *vals, int32_t ell_width, void *dz, int64_t dz_sample_stride,
                            int32_t lddz, void *t1, int64_t t1_sample_stride, int32_t ldt1, const float *rowscale, int32_t R,
                            int32_t rg, int32_t N, int32_t Mo, int32_t F, float *partials, int64_t partials_bytes,
                            float *rowmax_g_out, float *rowmax_t1_out, void *stream);
int32_t cape_spmm_multi_prep_chunks_bf16(const cape_spmm_term_t *terms, int32_t nterms, int32_t N, int32_t Mo, int32_t C);
int cape_spmm_multi_prep_bf16(const cape_spmm_term_t *terms, int32_t nterms, uint32_t masked_terms, const uint32_t *mask,
                              int32_t mask_rows, int32_t N, int32_t Mo, int32_t C, float *partials, int64_t partials_bytes,
                              void *stream);


/*
 * Operators applied AFTER the dense contraction, with the layer epilogue -- for up-sampling layers, where
 * (S_k x) W_k = S_k (x W_k) lets the contraction run on the coarse input rows:
 *   acc1 = sum_{k: bit k of to_acc2 clear} S_k x_k + rank-1 terms (bits of rank->to_acc2 clear);  acc2 = the others
 *   dual = 0:  y = act(acc1 + bias)          dual = 1:  y = relu(acc1) + acc2, sign bits of acc1 to mask_out
 * (same epilogue semantics as cape_gconv_fwd; mask_out needs F % 32 == 0 and 16-byte aligned operands).
 */
int cape_spmm_combine(const cape_spmm_term_t *terms, int32_t nterms, uint32_t to_acc2, const cape_rank_t *rank,
                      const float *bias, int32_t bias_mode, int32_t act, int32_t dual, uint32_t *mask_out, float *y,
                      int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t Mo, int32_t F, float *rowmax_out, void *stream);

/* y = act(x + bias) over [N, M, C] views (y may alias x). */
int cape_bias_act_fwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *bias,
                      int32_t bias_mode, int32_t act, float *y, int64_t y_sample_stride,
                      int32_t ldy, int32_t N, int32_t M, int32_t C, void *stream);

/* dz = dy * act'(.) evaluated from the activation OUTPUT y (dz may alias dy). */
int cape_act_bwd(const float *dy, int64_t dy_sample_stride, int32_t lddy, const float *y,
                 int64_t y_sample_stride, int32_t ldy, int32_t act, float *dz,
                 int64_t dz_sample_stride, int32_t lddz, int32_t N, int32_t M, int32_t C,
                 void *stream);

/* out[c] (+)= sum_{n (if reduce_n), m} x[n,m,c];  reduce_n=0 gives out[m,c] = sum_n (vertex bias grad). */
int64_t cape_colsum_workspace_bytes(int32_t N, int32_t M, int32_t C);
int cape_colsum(const float *x, int64_t x_sample_stride, int32_t ldx, int32_t N, int32_t M,
                int32_t C, int32_t per_vertex, int32_t accumulate, float *out, void *workspace,
                int64_t workspace_bytes, void *stream);

/* dgc[n,r,f] = dy[n,r,f] if mask bit set else 0 (gradient through the relu of the affine block). */
int cape_mask_mul(const float *dy, int64_t dy_sample_stride, int32_t lddy, const uint32_t *mask,
                  float *dz, int64_t dz_sample_stride, int32_t lddz, int32_t N, int32_t M,
                  int32_t F, void *stream);

/* y[n,m,c] = scale[m] * cond[n,c]  (scale NULL = 1): the tiled condition channels. */
int cape_fill_cond(const float *cond, int32_t ldc, const float *scale, float *y,
                   int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t M, int32_t C,
                   void *stream);

/* dcond[n,c] (+)= sum_m scale[m] * dy[n,m,c] */
int cape_reduce_cond(const float *dy, int64_t dy_sample_stride, int32_t lddy, const float *scale,
                     float *dcond, int32_t ldc, int32_t N, int32_t M, int32_t C,
                     int32_t accumulate, void *stream);

/*
 * Group norm over [C/G, V] per sample (lib/models.py:681-712: G = min(32, C) chosen by the caller, population variance,
 * eps inside the sqrt) with per-channel gamma / beta and an optional fused ReLU (:751-760):
 *   y = relu?( (x - mean_{n,g}) * rstd_{n,g} * gamma_c + beta_c )
 * Every pass reads whole rows (float4): requires C % 4 == 0 and 16-byte aligned, 4-float-padded views.
 * Outputs kept for the backward pass: stats [N, G, 2] = (mean, rstd) and coef [N, 4, C] = (a = rstd*gamma,
 * b = beta - mean*a, rstd, mean*rstd) per channel; the forward evaluates y = relu?(fma(a, x, b)) and the backward
 * re-derives the ReLU mask from the same fma, so it never reads y.  Statistics: one pass of pivot-shifted sums
 * (pivot = the sample's first row), combined per group in float64.  workspace >= cape_groupnorm_workspace_bytes.
 * rowmax_out: NULL or [N*V][4] floats that receive the row bounds (max_c |y|, 0, 0, 0) of the output -- of dx in the backward
 * -- for the fp16 two-piece contractions that read it next (cape_gconv_fwd_h2): written by the apply pass itself when a row is
 * a power-of-two lane group (C <= 256, C/4 a power of two), by a wave-per-row form of the same pass otherwise.
 */
int64_t cape_groupnorm_workspace_bytes(int32_t N, int32_t V, int32_t C);
int cape_groupnorm_fwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *gamma,
                       const float *beta, float eps, int32_t G, int32_t relu, float *y,
                       int64_t y_sample_stride, int32_t ldy, float *stats, float *coef, int32_t N, int32_t V,
                       int32_t C, void *workspace, int64_t workspace_bytes, float *rowmax_out, void *stream);
/* dx (+ dx_add when not NULL: a second gradient of the same input -- the residual branch of res_block_decoder,
 * lib/models.py:744-774 -- summed in the apply pass instead of by a separate element-wise launch) plus per-sample partial
 * parameter gradients dgamma_partial / dbeta_partial [N, C] (the caller sums them over N); stats / coef are the forward
 * outputs; bcoef is a [N, 3, round_up(C, 4)] scratch buffer. */
int cape_groupnorm_bwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *dy,
                       int64_t dy_sample_stride, int32_t lddy, const float *gamma,
                       const float *stats, const float *coef, int32_t G, int32_t relu, float *dx,
                       int64_t dx_sample_stride, int32_t lddx, const float *dx_add, int64_t add_sample_stride,
                       int32_t ldadd, float *dgamma_partial, float *dbeta_partial, float *bcoef, int32_t N,
                       int32_t V, int32_t C, void *workspace, int64_t workspace_bytes, float *rowmax_out, void *stream);

/* dgamma[c] = sum_n dgamma_partial[n, c] (likewise dbeta) for up to CAPE_MAX_GN_REDUCE_ITEMS earlier cape_groupnorm_bwd calls in
 * ONE launch, samples added in index order (the training step needs the parameter gradients only at the end of the backward
 * pass and queues these sums, as it does the weight-gradient slab reductions). */
#define CAPE_MAX_GN_REDUCE_ITEMS 32
typedef struct cape_gn_param_item {
    const float *dgamma_partial;   /* [N, C] */
    const float *dbeta_partial;    /* [N, C] */
    float *dgamma;                 /* [C]    */
    float *dbeta;                  /* [C]    */
    int32_t N, C;
} cape_gn_param_item_t;
int cape_groupnorm_param_reduce_batch(const cape_gn_param_item_t *items, int32_t nitems, void *stream);

/*
 * L1 reconstruction + edge loss (lib/models.py:357-375, lib/losses.py:9-25) and gradient:
 *   loss_out[0] = mean |pred - gt| ; loss_out[1] = mean_e || (p_i - p_j) - (g_i - g_j) ||
 *   *total_out = w_recon * loss_out[0] + w_edge * loss_out[1] [+ w_a * *term_a] [+ *term_b]     (total_out may be NULL)
 *   term_a / term_b: NULL or device scalars computed earlier in the step (the latent term and the regulariser value of the
 *   training loss, lib/models.py:393-394): the weighted sum of the loss is then complete after this launch
 *   dpred = w_recon * d(recon)/dpred + w_edge * d(edge)/dpred      (dpred may be NULL)
 * edges: device int32 [E,2].  workspace >= cape_recon_edge_workspace_bytes(N, M, E).
 * pred rows are ldp floats apart and dpred rows ldd (>= 3: the layer stack's outputs are padded to 16-byte rows, ld 4;
 * samples follow each other after M rows); gt is dense [N, M, 3].
 */
int64_t cape_recon_edge_workspace_bytes(int32_t N, int32_t M, int32_t E);
int cape_recon_edge_loss_fwd_bwd(const float *pred, int32_t ldp, const float *gt, const float *verts_ref,
                                 const int32_t *edges, const int32_t *vert_edge_ptr,
                                 const int32_t *vert_edge_idx, int32_t N, int32_t M, int32_t E,
                                 float w_recon, float w_edge, float *loss_out, float *total_out,
                                 const float *term_a, float w_a, const float *term_b, float *dpred,
                                 int32_t ldd, void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Adversarial losses on the discriminator's logits (lib/models.py:381-390, tf.nn.sigmoid_cross_entropy_with_logits with
 * label smoothing) and their gradients, one launch:
 *   loss_out[0] = gan_g = mean bce(fake, 1 - smooth)
 *   loss_out[1] = gan_d = mean bce(real, 1 - smooth) + mean bce(fake, smooth);   *scaled_g / *scaled_d = scale * the two
 *   grad_g / grad_d [Nf + Nr, M] contiguous = d(scale * gan_g) / d logit and d(scale * gan_d) / d logit, the Nf fake samples
 *   first, then the Nr real ones (grad_g is zero there).  Logit (n, m) of a tensor is at p[n * sample_stride + m * ld]
 *   (the prediction map [N, 431, 1] of discriminator, :676-678, as a row-padded view).  Fixed-order sums.
 */
int cape_gan_bce_fwd_bwd(const float *fake, int64_t fake_sample_stride, int32_t ldf, const float *real,
                         int64_t real_sample_stride, int32_t ldr, int32_t Nf, int32_t Nr, int32_t M, float smooth,
                         float scale, float *loss_out, float *scaled_g, float *scaled_d, float *grad_g, float *grad_d,
                         void *stream);

/* coef of every layer <- cond [N, Cc] (row stride ldc).  N * Cc <= 10240. */
int cape_cond_coef_fwd(const float *cond, int32_t ldc, int32_t N, int32_t Cc,
                       const cape_cond_layer_t *layers, int32_t nlayers, void *stream);

/* gw / gw_aff of every layer <- cond^T dcoef;  dcond[n,c] (+)= sum over layers, k, f of dcoef * w
 * (dcond may be NULL).  Deterministic (fixed summation order). */
int cape_cond_coef_bwd(const float *cond, int32_t ldc, int32_t N, int32_t Cc,
                       const cape_cond_layer_t *layers, int32_t nlayers, float *dcond, int32_t ldd,
                       int32_t accumulate, void *stream);

/*
 * Optimiser step on flat fp32 buckets (parameters w, gradients g, momentum m; n % 4 == 0, 16-byte aligned):
 * tf.clip_by_global_norm(5.0) + tf.train.MomentumOptimizer (lib/models.py:448-461) with the dense kernels' L2
 * regulariser gradient (:40, :378-379) folded in.  reg_ranges: HOST array of nranges [begin, end) element
 * ranges (multiples of 4) on which the effective gradient is g + reg_coef * w.  grad_scale (> 0) multiplies the raw bucket
 * first, everywhere "g" appears below: 1 for a single process, 1 / world when the bucket holds the SUM of the data-parallel
 * ranks' gradients after the all-reduce (the mean then costs no launch of its own; clipping sees the mean, as it must).
 *   cape_flat_gradnorm:        *sumsq_out = sum (g + reg)^2            (two deterministic launches)
 *   cape_flat_momentum_update: s = clip / max(sqrt(*sumsq), clip);  m = momentum*m + s*(g + reg);  w += (*neg_lr)*m
 *   cape_flat_adam_update:     tf.train.AdamOptimizer (lib/models.py:447-449) behind the same clip and regulariser:
 *                              t = state[0] + 1;  lr_t = -(*neg_lr) * sqrt(1 - beta2^t) / (1 - beta1^t);  g' = s*(g + reg);
 *                              m = beta1*m + (1-beta1)*g';  v = beta2*v + (1-beta2)*g'^2;  w -= lr_t * m / (sqrt(v) + eps).
 *                              state: two DEVICE int32 {step count, 0}; the call advances state[0] by one (TF's beta powers
 *                              as a counter), so replays of a captured graph apply the right bias correction.
 *   cape_sumsq_ranges:         *out = scale * sum over the ranges of x^2  (the regulariser's VALUE)
 * sumsq / neg_lr / out are DEVICE scalars (nothing is read back: the sequence is graph-capturable).
 */
int64_t cape_flat_workspace_bytes(void);
int cape_flat_gradnorm(const float *g, const float *w, int64_t n, const int64_t *reg_ranges, int32_t nranges,
                       float reg_coef, float grad_scale, float *sumsq_out, void *workspace, int64_t workspace_bytes, void *stream);
int cape_flat_momentum_update(float *w, const float *g, float *m, int64_t n, float momentum, float clip,
                              const float *sumsq, const float *neg_lr, const int64_t *reg_ranges,
                              int32_t nranges, float reg_coef, float grad_scale, void *stream);
int cape_flat_adam_update(float *w, const float *g, float *m, float *v, int64_t n, float beta1, float beta2, float eps,
                          float clip, const float *sumsq, const float *neg_lr, int32_t *state,
                          const int64_t *reg_ranges, int32_t nranges, float reg_coef, float grad_scale, void *stream);
int cape_sumsq_ranges(const float *x, const int64_t *ranges, int32_t nranges, float scale, float *out,
                      void *workspace, int64_t workspace_bytes, void *stream);

/*
 * VAE sampling + KL term (lib/models.py:193-196, :371-372); mean / logvar / eps contiguous [N, nz]:
 *   z = mean + exp(0.5*logvar) * eps ;  *kl = (-0.5/N) * sum(1 + logvar - mean^2 - exp(logvar))
 * z is written with row stride ldz; cond (NULL or [N, Cc], row stride ldc) is copied behind it, so that z IS the decoder's input
 * [z | cond] of lib/models.py:296 (tf.concat) when ldz >= nz + Cc.
 * backward: dmean = gz + (gkl/N)*mean ; dlogvar = 0.5*(gz*std*eps + (gkl/N)*(exp(logvar) - 1)).  gz (row stride ldgz: the first nz
 * columns of the gradient of [z | cond]) / gkl may be NULL.
 */
int cape_vae_sample_kl_fwd(const float *mean, const float *logvar, const float *eps, float *z, int32_t ldz, float *kl,
                           int32_t N, int32_t nz, const float *cond, int32_t ldc, int32_t Cc, void *stream);
int cape_vae_sample_kl_bwd(const float *mean, const float *logvar, const float *eps, const float *gz, int32_t ldgz,
                           const float *gkl, float *dmean, float *dlogvar, int32_t N, int32_t nz, void *stream);

/*
 * Dense layers with one very long side (tf.layers.dense, lib/models.py:557, :560, :582): weight-streaming
 * kernels, N <= 64 batch rows, W row-major [in, out].  Pointer arrays are HOST arrays of device pointers.
 *
 * long input (in >> out; nmat = 1 or 2 matrices sharing x, e.g. fc_mean / fc_var):
 *   fwd:  y_m[n,j] = b_m[j] + sum_i x[n,i] W_m[i,j]                 (two deterministic launches)
 *   bwd:  dW_m[i,j] = sum_n x[n,i] g_m[n,j];  db_m[j] = sum_n g_m[n,j];  dx[n,i] = sum_m sum_j g_m[n,j] W_m[i,j]
 *         (g_m contiguous [N,out]; dW / db arrays or entries may be NULL; dx may be NULL)
 * wide output (out >> in, N*in <= 12288, in <= 200):
 *   fwd:  y[n,j] = act(b[j] + sum_i x[n,i] W[i,j])
 *   bwd:  dz = g * act'(y) (y = activation output);  dW[i,j] = sum_n x[n,i] dz[n,j];  db[j] = sum_n dz[n,j];
 *         dx[n,i] = sum_j dz[n,j] W[i,j]   (two-stage, workspace >= cape_fc_wide_bwd_workspace_bytes)
 */
int64_t cape_fc_long_workspace_bytes(int32_t N, int32_t in, int32_t out, int32_t nmat);
int cape_fc_long_fwd(const float *x, int32_t ldx, int32_t N, int32_t in, int32_t out, int32_t nmat,
                     const float *const *W, const float *const *b, float *const *y, void *workspace,
                     int64_t workspace_bytes, void *stream);
int cape_fc_long_bwd(const float *x, int32_t ldx, int32_t N, int32_t in, int32_t out, int32_t nmat,
                     const float *const *W, const float *const *g, float *const *dW, float *const *db,
                     float *dx, int32_t lddx, void *stream);
int cape_fc_wide_fwd(const float *x, int32_t ldx, int32_t N, int32_t in, int32_t out, const float *W,
                     const float *b, int32_t act, float *y, int32_t ldy, void *stream);
int64_t cape_fc_wide_bwd_workspace_bytes(int32_t N, int32_t in, int32_t out);
int cape_fc_wide_bwd(const float *x, int32_t ldx, const float *g, int32_t ldg, const float *y, int32_t ldy,
                     int32_t act, int32_t N, int32_t in, int32_t out, const float *W, float *dW, float *db,
                     float *dx, int32_t lddx, void *workspace, int64_t workspace_bytes, void *stream);

/*
 * The two condition networks in one launch per direction (lib/models.py:479-511 ``condition`` as called at :284-290):
 *   h    = leaky_relu(c1 W1 + b1, 0.2)   [N, hid]     (tf.layers.dense, pose MLP layer 1)
 *   ycat = [ h W2 + b2 | c2 Wc + bc ]    [N, out1 + out2]   (pose MLP layer 2 | clothing-type layer, concatenated the
 *          way every consumer concatenates them, :533, :591, :663)
 * W* row-major [in, out]; c1 / c2 rows ld1 / ld2 apart; h and ycat contiguous; N <= 64.  ycat2: NULL or a second buffer that
 * receives the same values -- a consumer of its own (the decoder input [z | ycat], :296), whose gradient then reaches the
 * backward separately instead of through an element-wise sum.  Backward: dycat / dycat2 [N, out1 + out2], rows lddy / lddy2
 * floats apart, either may be NULL (zero), the kernel differentiates dycat + dycat2 -> gradients of all six variables
 * (written, not accumulated).  Fixed summation order (deterministic).
 */
int cape_condnet_fwd(const float *c1, int32_t ld1, const float *c2, int32_t ld2, const float *W1, const float *b1,
                     const float *W2, const float *b2, const float *Wc, const float *bc, float *h, float *ycat,
                     float *ycat2, int32_t N, int32_t in1, int32_t hid, int32_t out1, int32_t in2, int32_t out2, void *stream);
int cape_condnet_bwd(const float *c1, int32_t ld1, const float *c2, int32_t ld2, const float *W2, const float *h,
                     const float *dycat, int32_t lddy, const float *dycat2, int32_t lddy2, float *gW1, float *gb1,
                     float *gW2, float *gb2, float *gWc, float *gbc, int32_t N, int32_t in1, int32_t hid, int32_t out1, int32_t in2, int32_t out2, void *stream);

/*
 * ---- bf16 storage variants (BASELINE configs[4]: "bf16 weights/activations", SURVEY section 8(b) "_bf16") -----------------
 * Same operators and the same argument meaning as the entry points of the same name without the suffix, with every
 * ACTIVATION tensor -- sources x, outputs y, gradients g / dz / dz2, the terms of the sparse kernels -- stored as bf16
 * (uint16_t bit patterns; cape_src_t.x, cape_spmm_term_t.x / .y then point to bf16 data; all strides are in ELEMENTS).
 * Weights, biases, rank-1 coefficients, CSR values, weight / bias / coefficient gradients and every reduction workspace
 * stay fp32 (fp32 master weights; the contraction kernels round a weight to bf16 while staging it, i.e. they compute
 * with bf16 weights).  Arithmetic: bf16 x bf16 products accumulated in fp32 (v_mfma_f32_32x32x16_bf16, ONE product per
 * multiply-add) where the launch is eligible for the matrix-pipe kernel -- plain sources of whole 32-channel chunks, rows
 * 16-byte aligned (ldx % 8 == 0), F >= 64 -- and fp32 arithmetic on widened values elsewhere; every store rounds to nearest
 * even.  Tolerance of the path against the fp32 one: <= 2e-2 relative, tests/test_gpu_bf16.py.
 * The reference's placeholders are fp32 (lib/models.py:272-282): this storage type is the north_star's own addition.
 */
int cape_gconv_fwd_bf16(const cape_src_t *srcs, int32_t nsrc, void *y, int64_t y_sample_stride,
                        int32_t ldy, int32_t N, int32_t Mo, int32_t F, const float *bias,
                        int32_t bias_mode, int32_t act, uint32_t *mask_out, const cape_rank_t *rank,
                        int32_t out_deinterleave, void *stream);
int cape_gconv_fwd_plan_bf16(const cape_src_t *srcs, int32_t nsrc, int32_t N, int32_t Mo, int32_t F, int32_t plan[4]);
/* workspace size: cape_gconv_dw_workspace_bytes (the partial slabs are fp32 in both storage types) */
int cape_gconv_dw_bf16(const cape_src_t *srcs, int32_t nsrc, const void *dz,
                       int64_t dz_sample_stride, int32_t lddz, const void *dz2, uint32_t dz2_mask,
                       int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                       int64_t workspace_bytes, void *stream);
int cape_gconv_dw_stage_bf16(const cape_src_t *srcs, int32_t nsrc, const void *dz,
                             int64_t dz_sample_stride, int32_t lddz, const void *dz2, uint32_t dz2_mask,
                             int32_t N, int32_t Mo, int32_t F, int32_t accumulate, void *workspace,
                             int64_t workspace_bytes, int32_t stage, void *stream);
int cape_gconv_dw_plan_bf16(const cape_src_t *srcs, int32_t nsrc, const void *dz, int64_t dz_sample_stride, int32_t lddz,
                            const void *dz2, uint32_t dz2_mask, int32_t N, int32_t Mo, int32_t F, int32_t plan[4]);
int cape_spmm_bf16(const void *x, int64_t x_sample_stride, int32_t ldx, const int32_t *rowptr,
                   const int32_t *colidx, const float *vals, int32_t max_row_nnz, int32_t ell_width, float alpha,
                   const void *z, int64_t z_sample_stride, int32_t ldz, float beta, void *y, int64_t y_sample_stride,
                   int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream);
int cape_spmm_multi_bf16(const cape_spmm_term_t *terms, int32_t nterms, int32_t sum, void *y, int64_t y_sample_stride,
                         int32_t ldy, int32_t N, int32_t Mo, int32_t C, float *rowmax_out, void *stream);
int cape_spmm_combine_bf16(const cape_spmm_term_t *terms, int32_t nterms, uint32_t to_acc2, const cape_rank_t *rank,
                           const float *bias, int32_t bias_mode, int32_t act, int32_t dual, uint32_t *mask_out, void *y,
                           int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t Mo, int32_t F, float *rowmax_out, void *stream);
/* workspace size and cape_bwd_prep_finalize as for cape_bwd_prep (fp32 partials) */
int cape_bwd_prep_bf16(const void *g, int64_t g_sample_stride, int32_t ldg, const void *y, int64_t y_sample_stride,
                       int32_t ldy, int32_t act, const uint32_t *mask, void *dz, int64_t dz_sample_stride, int32_t lddz,
                       float *dbias, const float *rowscale, int32_t R, float *dcoef, int32_t rg, float *dcoef_g,
                       int64_t dcoef_sample_stride, int32_t finalize, int32_t N, int32_t Mo, int32_t F, void *workspace,
                       int64_t workspace_bytes, float *rowmax_out, void *stream);
/* out[m, c] (+)= sum_n x[n, m, c]: gradient of the per-vertex output bias [1, M, F] (lib/models.py:615) from a bf16 dz */
int cape_colsum_vertex_bf16(const void *x, int64_t x_sample_stride, int32_t ldx, int32_t N, int32_t M, int32_t C,
                            int32_t accumulate, float *out, void *stream);

/*
 * General-K Chebyshev graph convolution with the recurrence ON CHIP -- reference lib/models.py:69-103 in the form of its
 * explicit recurrence (:88-96: x_k = 2 L~ x_{k-1} - x_{k-2}), for plain layers (no pool / unpool, no bias) of polynomial
 * order 2..8; BASELINE configs[1] is one such layer (K = 6, 64 x 6890 x 16 -> 32).  One workgroup per (sample, vertex
 * patch) keeps the running pair of the recurrence for the patch and its (K-1)-ring halo in LDS and contracts
 * y[patch] += T_k[patch] W_k from there; the K-stack [M, Fin*K] the reference concatenates (:85-99) never exists in HBM.
 *   y[n] = sum_k T_k(L~) x[n] W_k,   W in the reference layout [Cin*K, Fout], row c*K + k  (:99-102)
 * The patch plan (cape_amd.graph.ChebPatchPlan) is host data uploaded once per (Laplacian, K, Cin):
 *   pinfo [P][16] int32: [0] offset into vid, [1] first ELL row of the patch, [2] unused,
 *                        [3 + j] number of local vertices within ring <= j of the patch (j = 0 .. K-1; local indices are
 *                        sorted by ring, [3] = the patch itself, at most 256)
 *   vid    global vertex of every local index;  ell_col / ell_val [rows][12] (16-byte aligned): the rows of L~ of the
 *          local vertices within ring <= K-2 in ELL form, column = LOCAL index (all their neighbours lie within ring
 *          <= K-1), real entries first, padded with (own local index, 0); rows longer than 12 entries are not supported
 *   rmax   largest patch + halo (multiple of 4): 2 * rmax * (Cin + 4) floats of LDS (+ 8 * Cin * Fout backward)
 * Supported: fp32, Cin in {8, 16, 24, 32}, Fout in {32, 64}, rows 16-byte aligned (cape_cheb_fused_supported).
 * The backward entry needs L~ symmetric (it is: lib/mesh_sampling.py:10-38 builds I - D^-1/2 A D^-1/2); it recomputes the
 * recurrence for dW = sum_n sum_k T_k(L~) x[n]^T dy[n] (per-workgroup partials in ``workspace``, reduced in a fixed order)
 * and evaluates dx[n] = sum_k T_k(L~) dy[n] W_k^T by Clenshaw's recurrence.  Deterministic, no atomics.  Either of dx / dW
 * may be NULL: that half is skipped (a data-gradient-only sweep through the layer; an input that needs no gradient).
 */
int cape_cheb_fused_supported(int32_t Cin, int32_t Fout, int32_t K);
int cape_cheb_fused_fwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *W, float *y,
                        int64_t y_sample_stride, int32_t ldy, int32_t N, int32_t M, int32_t Cin, int32_t Fout, int32_t K,
                        int32_t P, const int32_t *pinfo, const int32_t *vid, const int32_t *ell_col, const float *ell_val,
                        int32_t rmax, void *stream);
int64_t cape_cheb_fused_bwd_workspace_bytes(int32_t N, int32_t Cin, int32_t Fout, int32_t K, int32_t P);
int cape_cheb_fused_bwd(const float *x, int64_t x_sample_stride, int32_t ldx, const float *dy, int64_t dy_sample_stride,
                        int32_t lddy, const float *W, float *dx, int64_t dx_sample_stride, int32_t lddx, float *dW,
                        int32_t accumulate, int32_t N, int32_t M, int32_t Cin, int32_t Fout, int32_t K, int32_t P,
                        const int32_t *pinfo, const int32_t *vid, const int32_t *ell_col, const float *ell_val,
                        int32_t rmax, void *workspace, int64_t workspace_bytes, void *stream);
/* diagnostic: device buffer of >= 64 uint64 stamped with s_memtime at the forward kernel's phase boundaries (NULL = off) */
int cape_cheb_fused_debug_timestamps(void *ts);

#ifdef __cplusplus
}
#endif
#endif /* CAPE_HIP_H */
