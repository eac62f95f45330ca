/*
 * xpretrain_hip.h -- C ABI of libxpretrain_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * CLIP-ViP video-text contrastive hot path of microsoft/XPretrain.
 *
 * The reference has NO native/FFI layer (pure Python on torch/cuDNN/cuBLAS/apex/Horovod),
 * so every entry point below cites the reference *Python* call site whose arithmetic it replaces
 * (paths relative to /root/reference/CLIP-ViP/src).  The Python host side
 * (xpretrain_amd/modeling/) binds these through ctypes -- see INTEGRATION.md.
 *
 * Conventions (SURVEY.md §8b "C-ABI the new extension must export"):
 *   - plain pointers + sizes; no torch types.  Every pointer is a DEVICE pointer unless named host_*.
 *   - never allocates, never synchronises, never owns memory; launches only on `stream`
 *     (a hipStream_t passed as void*); re-entrant (forward thread + autograd thread).
 *     The three deviations an integrator must know, all documented at the entry point concerned: (1) xp_encoder_layer_bwd runs a
 *     layer's weight-gradient GEMMs on ONE library-owned stream per device (xp_side_stream(), created on first use, ordered against
 *     `stream` by library-owned events, joined before the call returns); (2) xp_set_cu_budget() is process-global planning state;
 *     (3) xp_attn_bwd / xp_attn_bwd2 enqueue one 4-byte hipMemsetAsync on `stream` (the fused backward's problem counter, inside the
 *     caller's workspace).
 *   - returns 0 on success, a negative XP_ERR_* otherwise; xp_last_error() gives the thread-local message.
 *   - `dtype`: element type of activations / GEMM operands (XP_BF16 or XP_F32).  Parameters that are read
 *     directly from the fp32 master copy (biases, LayerNorm affine, embedding tables) are always float.
 *   - sizes are int64_t elements; `ld*` are row strides in elements.
 */
#ifndef XPRETRAIN_HIP_H
#define XPRETRAIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XP_ABI_VERSION 1

enum { XP_BF16 = 0, XP_F32 = 1 };
enum { XP_OK = 0, XP_ERR_ARG = -1, XP_ERR_LAUNCH = -2, XP_ERR_UNSUPPORTED = -3 };

int xp_abi_version(void);
const char* xp_last_error(void);

/* ------------------------------------------------------------------------------------------------ GEMM
 * C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ).  One MFMA kernel family for every dense contraction on
 * the path: q/k/v/out projections (modeling/CLIP_ViP.py:341-343,379), MLP fc1/fc2 (:393-395), the
 * patch-embedding conv as a GEMM (:157-159,178), visual/text projections (:1142,1145), and their
 * autograd backward products (dX = dY.W, dW = dY^T.X).
 *
 * Operand storage: a_kstrided/b_kstrided == 0: row index major, k contiguous (A[m*lda + k]);
 *                  == 1: k major (A[k*lda + m]) -- read through LDS transpose loads, no copy.
 */
enum {
  XP_EPI_NONE = 0,       /* C = acc                                                              */
  XP_EPI_BIAS = 1,       /* C = acc + bias[n]                                                    */
  XP_EPI_BIAS_QSCALE = 2,/* C = (acc + bias[n]) * (n < scale_cols ? scale : 1)   q*dh^-0.5 (:341)  */
  XP_EPI_BIAS_GELU = 3,  /* aux = acc + bias[n];  C = aux * sigmoid(1.702 aux)   quick_gelu (:394); aux == NULL: the
                          * pre-activation is not stored (forward-only passes: retrieval / inference)        */
  XP_EPI_BIAS_RESID = 4, /* C = acc + bias[n] + resid[m,n]                       (:455,:460)       */
  XP_EPI_GELU_BWD = 5,   /* C = acc * d/dx quick_gelu(resid[m,n])                                 */
  XP_EPI_PATCH = 6,      /* C = acc + tab1[t,n] + tab2[l,n], t=(m%c_grp)/tab_L, l=m%tab_L (:182-185) */
  XP_EPI_SCALE = 7       /* C = acc * scale                                                      */
};

typedef struct XpGemmDesc {
  const void* A; const void* B; void* C;
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  int32_t a_kstrided, b_kstrided;
  int32_t in_dtype, out_dtype;          /* XP_BF16 / XP_F32 (out may be F32 with BF16 inputs)     */
  int32_t epilogue;
  int32_t split_k;                      /* >1: writes split_k fp32 slabs [split][M][N] to C, EPI_NONE */
  /* optional row remap  r -> (r / grp) * grp_stride + off + r % grp   (grp == 0: identity)        */
  int64_t a_grp, a_grp_stride, a_off;   /* applied to A's m index (row, or k-row when a_kstrided)  */
  int64_t c_grp, c_grp_stride, c_off;   /* applied to the output row (C, aux, resid)               */
  const float* bias;
  float scale; int64_t scale_cols;
  const void* resid; int64_t ldr;       /* dtype = in_dtype                                        */
  void* aux; int64_t ldaux;             /* dtype = out_dtype                                       */
  const float* tab1; const float* tab2; int64_t tab_L;
  /* optional: column sums of the FINISHED outputs (fp32, before rounding), one partial row per half tile height of rows:
   * colsum_partials[r*N + n], r < xp_gemm_colsum_rows(desc).  This is the bias gradient of the Linear whose output
   * gradient this GEMM produces (autograd computes it as grad.sum(0), CLIP_ViP.py:383-396) without re-reading it.
   * Only where xp_gemm_colsum_rows() > 0 (large bf16 problems, EPI_NONE / EPI_GELU_BWD); finish with
   * xp_reduce_rows_batch. */
  float* colsum_partials;
  /* reserved (round 3 selected a second kernel set of the 256-wide family here; since round 4 there is one): ignored */
  int32_t reserved0;
  int32_t side_M;
  /* optional, EPI_BIAS_RESID: fp32 "side rows" of the residual stream.  Output rows m with m % side_S < side_M (the video tower's
   * proxy tokens: rows [0, M) of every sample of S tokens, CLIP_ViP.py:187-197) take their residual operand from
   * resid_side[((m / side_S) * side_M + m % side_S) * N + n] (fp32) instead of resid, and their fp32 result is ALSO written to
   * out_side (same indexing) next to the rounded C row.  The pooled feature is token 0 of the last layer: keeping the residual
   * stream of these few rows in fp32 halves the feature error of the bf16 path (tools/residual_precision_experiment.py). */
  const float* resid_side; float* out_side; int64_t side_S;
  /* optional: A is not a matrix in memory but the patch matrix of a frame tensor, gathered by the operand loader -- the conv-as-GEMM
   * of CLIPVisionViPEmbeddings.patch_embedding (CLIP_ViP.py:157-159,178) without the im2col round trip.  a_frames: [BT,3,H,W], fp32 or
   * (a_frames_u8 = 1) decoded uint8 with the collate arithmetic (x / 255 - fr_mean[c]) / fr_std[c] (datasets/dataloader.py:209-233)
   * fused in; M = BT * (H/P) * (W/P) patches in (bt, gy, gx) order, K = 3*P*P in (c, dy, dx) order; A / lda are ignored.  The gathered
   * values are rounded to bf16 exactly as xp_im2col / xp_im2col_u8 round them (same result as the materialised matrix, bit for
   * bit).  bf16 compute, P % 8 == 0, no split-K. */
  const void* a_frames; int32_t a_frames_u8; int32_t fr_H, fr_W, fr_P; float fr_mean[3], fr_std[3];
} XpGemmDesc;

int xp_gemm(const XpGemmDesc* desc, void* stream);

/* Split-K factor xp_gemm should be given for a weight-gradient-shaped problem (dW = dY^T X as computed by the
 * backward of nn.Linear inside CLIPEncoderLayer, modeling/CLIP_ViP.py:383-396,444-460): fills one resident round
 * of workgroups of the kernel family xp_gemm will pick for `desc` (desc->split_k and the data pointers are
 * ignored).  The value is always accepted by xp_gemm (whole k-steps per slab, no empty slab); 1 = no split. */
int32_t xp_gemm_auto_split(const XpGemmDesc* desc);
/* The same for a split-K launch that has SLACK: it runs on another stream beside the caller's work and nothing waits for it soon --
 * the first three weight-gradient GEMMs of an encoder layer's backward (fc2, fc1, out_proj: issued on the weight-gradient stream
 * while the dX chain of the layer still has attention and two GEMMs to go); the last one (q/k/v) is what the layer's join waits
 * for and takes xp_gemm_auto_split.  Fewer, longer slabs: less fp32 slab traffic beside the work it shares the chip with. */
int32_t xp_gemm_auto_split_slack(const XpGemmDesc* desc);
/* CUs the split-K planning may fill (64..256, default 256 or $XPRETRAIN_CU_BUDGET).  A data-parallel run lowers it by the number
 * of workgroups its collective library keeps resident during the backward pass (hvd.DistributedOptimizer's all-reduce,
 * run_pretrain.py:224-227,379; RCCL's gfx950 kernels own a CU per workgroup) so that a dW launch still fits one round. */
int xp_set_cu_budget(int32_t cus);
int32_t xp_get_cu_budget(void);
/* number of partial rows xp_gemm writes to desc->colsum_partials, or 0 if the fused column sums are not available
 * for this problem (desc->colsum_partials itself is ignored here) */
int64_t xp_gemm_colsum_rows(const XpGemmDesc* desc);
/* Output-tile height (rows) of the kernel family xp_gemm will run `desc` with: 256 for the 256-wide ping-pong family (the four
 * Linear layers of CLIPEncoderLayer at video-tower sizes, modeling/CLIP_ViP.py:341-343,379,393-395), 128 for the 128x128 family.
 * desc->split_k is honoured; the data pointers are ignored. */
int32_t xp_gemm_tile_rows(const XpGemmDesc* desc);

/* out[i] (+)= sum_z slabs[z*n + i], fp32; accumulate != 0 adds into out (gradient accumulation). */
int xp_splitk_reduce(const float* slabs, float* out, int64_t n, int32_t splits, int32_t accumulate, void* stream);

/* Bias gradient: out[n] (+)= sum_m X[m*ldx + n]; workspace >= xp_colsum_workspace_bytes(). */
size_t xp_colsum_workspace_bytes(int64_t rows, int64_t cols);
int xp_colsum(const void* X, int64_t rows, int64_t cols, int64_t ldx, int32_t dtype, float* out,
              int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Deferred form: first level only.  partials[r*cols + n] = sum over the r-th chunk of rows of X[.][n];
 * xp_colsum_partial_rows(rows, cols) chunks.  Finish with xp_reduce_rows_batch. */
int64_t xp_colsum_partial_rows(int64_t rows, int64_t cols);
int xp_colsum_partials(const void* X, int64_t rows, int64_t cols, int64_t ldx, int32_t dtype, float* partials,
                       size_t partials_bytes, void* stream);

/* Batched deterministic column sums of up to XP_REDUCE_MAX_SEGS fp32 partial-row arrays in TWO launches:
 * out[c] (+)= sum_r in[r*stride + c], r < nrows, c < width.  Used to finish all bias and LayerNorm-parameter
 * gradients of one CLIPEncoderLayer backward (modeling/CLIP_ViP.py:444-460) at once.  segs_host is a HOST array. */
#define XP_REDUCE_MAX_SEGS 16
typedef struct {
  const float* in; float* out;
  int64_t stride;                        /* row pitch of `in`, elements                                  */
  int32_t nrows, width;
  int32_t accumulate, reserved;
} XpReduceSeg;
size_t xp_reduce_rows_batch_workspace_bytes(const XpReduceSeg* segs_host, int32_t n);
int xp_reduce_rows_batch(const XpReduceSeg* segs_host, int32_t n, void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------------------- LayerNorm
 * nn.LayerNorm(eps=1e-5) of modeling/CLIP_ViP.py:404-406,855-857,720 (pre_layrnorm, layer_norm1/2,
 * post_layernorm, final_layer_norm).  Statistics in fp32; mean/rstd saved for the backward.
 */
int xp_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                     float* mean, float* rstd, int64_t rows, int64_t cols, float eps, int32_t dtype, void* stream);
/* The same with fp32 side rows of the residual stream (see XpGemmDesc::resid_side): row r with r % side_S < side_M is READ from
 * x_side[((r / side_S) * side_stride + r % side_S) * cols] (fp32) instead of x when x_side != NULL, and its fp32 result is ALSO
 * written to y_side (same indexing) when y_side != NULL (pre_layrnorm, CLIP_ViP.py:881: its output is the residual stream). */
int xp_layernorm_fwd_side(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                          float* mean, float* rstd, int64_t rows, int64_t cols, float eps, int32_t dtype,
                          const float* x_side, float* y_side, int64_t side_S, int32_t side_M, int32_t side_stride, void* stream);
size_t xp_layernorm_bwd_workspace_bytes(int64_t rows, int64_t cols);
/* Deferred form: dx is final, the parameter gradients stay as xp_layernorm_bwd_partial_rows(rows) partial rows in
 * the workspace -- [dgamma(cols) | dbeta(cols)] (pitch 2*cols); with with_dx_colsum == 1
 * [dgamma | dbeta | colsum(dx)] (pitch 3*cols; the column sums of the dx just written = the bias gradient of the
 * Linear in front of this residual add); with with_dx_colsum == 2 (needs dres) [dgamma | dbeta | colsum(dx) | colsum(dres)]
 * (pitch 4*cols; in CLIPEncoderLayer's second LayerNorm, CLIP_ViP.py:455-458, dres is the layer's incoming gradient, whose
 * column sums are fc2's bias gradient) -- for xp_reduce_rows_batch. */
int64_t xp_layernorm_bwd_partial_rows(int64_t rows);
int xp_layernorm_bwd_partials(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                              const float* mean, const float* rstd, const void* dres, int64_t lddres,
                              void* dx, int64_t lddx, int32_t with_dx_colsum, int64_t rows, int64_t cols,
                              int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);
/* dx = (dres ? dres : 0) + LN'(dy);  dgamma/dbeta (+)= column sums.  dres may alias dx. */
int xp_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                     const float* mean, const float* rstd, const void* dres, int64_t lddres,
                     void* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                     int64_t rows, int64_t cols, int32_t dtype,
                     void* workspace, size_t workspace_bytes, void* stream);
/* The two backward forms with the fp32 side rows of x the forward normalised (xp_layernorm_fwd_side's x_side, same indexing): those
 * rows' x-hat is recomputed from the fp32 values, so the backward differentiates exactly the function the forward computed. */
int xp_layernorm_bwd_partials_side(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                                   const float* mean, const float* rstd, const void* dres, int64_t lddres,
                                   void* dx, int64_t lddx, int32_t with_dx_colsum, int64_t rows, int64_t cols,
                                   int32_t dtype, const float* x_side, int64_t side_S, int32_t side_M, int32_t side_stride,
                                   void* workspace, size_t workspace_bytes, void* stream);
int xp_layernorm_bwd_side(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                          const float* mean, const float* rstd, const void* dres, int64_t lddres,
                          void* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                          int64_t rows, int64_t cols, int32_t dtype,
                          const float* x_side, int64_t side_S, int32_t side_M, int32_t side_stride,
                          void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------------------- Attention
 * Fused attention on the packed projection output qkv[B, S, 3, H, 64] (row stride ldqkv elements; q is
 * already scaled).  head_dim is fixed at 64 (every CLIP ViT-B/L and text tower).
 *
 *  mode XP_ATTN_PROXY : CLIPAttention.forward2 (modeling/CLIP_ViP.py:332-381).  S = M + N*L; frame-n
 *                       queries attend [M proxies | frame n]; proxy queries attend everything.
 *  mode XP_ATTN_CAUSAL: CLIPAttention.forward text path (:266-330) with the additive -inf causal mask
 *                       (:788-797) and the finfo.min padding mask (_expand_mask :50-61); pad_mask is
 *                       int64 [B,S] (1 = keep) or NULL.  M,N,L ignored (pass 0,1,S).
 *
 * out[B,S,H*64] (row stride ldo).  stats[B,H,S,2] fp32 = (row max, log row sum) for the backward.
 */
enum { XP_ATTN_PROXY = 0, XP_ATTN_CAUSAL = 1 };
size_t xp_attn_workspace_bytes(int32_t mode, int64_t B, int64_t H, int64_t M, int64_t N, int64_t L);
int xp_attn_fwd(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, float* stats,
                const int64_t* pad_mask, int32_t mode, int64_t B, int64_t H, int64_t S,
                int64_t M, int64_t N, int64_t L, int32_t dtype,
                void* workspace, size_t workspace_bytes, void* stream);
/* dqkv[B,S,3,H,64] (same layout as qkv; dq already multiplied by q_scale so it is the gradient of the
 * un-scaled projection output).  PROXY problems that fit one LDS group (M + L <= 208, M <= 16, no padding mask: 224^2 frames at patch
 * 16, any frame count) run as ONE persistent launch for dQ, dK and dV (attn_bwd5_kernel, round 6); its workgroups take problems from
 * a 4-byte device counter at the end of `workspace`, which this call resets with hipMemsetAsync on `stream` (the one operation of the
 * library besides kernel launches, event records and waits).  Everything else runs the dQ kernel, then the dK/dV kernel. */
int xp_attn_bwd(const void* qkv, int64_t ldqkv, const void* out, const void* dout, int64_t ldo,
                const float* stats, const int64_t* pad_mask, void* dqkv, float q_scale,
                int32_t mode, int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L, int32_t dtype,
                void* workspace, size_t workspace_bytes, void* stream);
/* The same with the bias gradients of q_proj / k_proj / v_proj on the way (autograd computes them as dqkv.sum(0),
 * CLIP_ViP.py:341-343): every backward workgroup also leaves the column sums of the dqkv rows it stores (as stored, i.e. rounded)
 * as one partial row: dqkv_colsum_partials[r * 3*H*64 + c], r < xp_attn_bwd_colsum_rows(...) -- finish with xp_reduce_rows_batch.
 * Saves the separate pass over dqkv.  xp_attn_bwd_colsum_rows() == 0 (fp32 mode): not available, pass NULL. */
int64_t xp_attn_bwd_colsum_rows(int32_t mode, int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L, int32_t dtype);
int xp_attn_bwd2(const void* qkv, int64_t ldqkv, const void* out, const void* dout, int64_t ldo,
                 const float* stats, const int64_t* pad_mask, void* dqkv, float q_scale,
                 int32_t mode, int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L, int32_t dtype,
                 void* workspace, size_t workspace_bytes, float* dqkv_colsum_partials, void* stream);

/* --------------------------------------------------------------------------------- Embeddings / glue
 * CLIPVisionViPEmbeddings.forward (modeling/CLIP_ViP.py:168-197).
 */
/* frames fp32 [BT,3,H,W] -> patch matrix [BT*gh*gw, 3*P*P] (dtype), k = (c,py,px): the conv-as-GEMM A operand */
int xp_im2col(const float* video, void* patches, int64_t BT, int64_t H, int64_t W, int64_t P, int32_t dtype, void* stream);
/* Same patch matrix straight from DECODED uint8 frames [BT,3,H,W]: (x/255 - mean[c]) / std[c] -- the host collate's
 * /255 + transforms.Normalize (datasets/dataloader.py:209-233) resp. ImageNorm on the device
 * (datasets/data_utils.py:256-281) -- fused with the cast and the patch gather.  mean/std: HOST
 * float[3]. */
int xp_im2col_u8(const uint8_t* frames, const float* mean3_host, const float* std3_host, void* patches, int64_t BT,
                 int64_t H, int64_t W, int64_t P, int32_t dtype, void* stream);
/* proxy rows: x[b, 0] = class_emb + pos[0]; x[b, 1+i] = added_cls[i] + pos[0]   (:187-191) */
int xp_vip_proxy_rows(const float* class_emb, const float* added_cls, const float* pos, void* x,
                      int64_t B, int64_t S, int64_t M, int64_t D, int32_t dtype, void* stream);
/* gradients of class_embedding, added_cls, position_embedding[1+L], time table [T,D] from dx[B,S,D] */
size_t xp_vip_embed_bwd_workspace_bytes(int64_t B, int64_t T, int64_t L, int64_t D);
int xp_vip_embed_bwd(const void* dx, float* d_class, float* d_added, float* d_pos, float* d_time,
                     int64_t B, int64_t M, int64_t T, int64_t L, int64_t D, int32_t dtype, int32_t accumulate,
                     void* workspace, size_t workspace_bytes, void* stream);
/* CLIPTextEmbeddings.forward (:210-227): x[b,t] = tok[ids[b,t]] + pos[t] */
int xp_text_embed_fwd(const int64_t* ids, const float* tok, const float* pos, void* x,
                      int64_t B, int64_t Lt, int64_t D, int64_t vocab, int32_t dtype, void* stream);
/* d_tok[id] (+)= sum of dx[b,t] over the occurrences of id in (b,t) order (fixed order, no atomics: bit-reproducible);
 * d_pos[t] (+)= sum_b dx[b,t] */
int xp_text_embed_bwd(const int64_t* ids, const void* dx, float* d_tok, float* d_pos,
                      int64_t B, int64_t Lt, int64_t D, int64_t vocab, int32_t dtype, int32_t accumulate, void* stream);
/* pooled[b] = x[b, idx[b]] (text: idx = ids.argmax(-1), :776 ; vision: idx = 0, :892) and its scatter */
int xp_argmax_rows(const int64_t* ids, int64_t* idx, int64_t B, int64_t Lt, void* stream);
int xp_gather_rows(const void* x, const int64_t* idx, void* out, int64_t B, int64_t S, int64_t D, int32_t dtype, void* stream);
int xp_scatter_rows(const void* dout, const int64_t* idx, void* dx, int64_t B, int64_t S, int64_t D, int32_t dtype, void* stream);
/* x / ||x||_2 per row (:1148-1149); in `dtype`, out fp32 features; inv_norm saved */
int xp_l2norm_fwd(const void* x, float* y, float* inv_norm, int64_t rows, int64_t cols, int32_t dtype, void* stream);
int xp_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, void* dx, int64_t rows, int64_t cols,
                  int32_t dtype, void* stream);
/* fp32 master -> compute dtype copy */
int xp_cast(const float* src, void* dst, int64_t n, int32_t dtype, void* stream);
int xp_cast_back(const void* src, float* dst, int64_t n, int32_t dtype, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------- Optimizer
 * On-device optimizer step: torch.nn.utils.clip_grad_norm_ as the training loops call it
 * (tasks/run_video_retrieval.py:390-392; coef = min(1, max_norm / (||g||_2 + 1e-6))) fused with AdamW.step
 * (optimization/adamw.py:40-103: denom = sqrt(v) + eps, step_size = lr * sqrt(1-b2^t) / (1-b1^t) when
 * correct_bias, decoupled weight decay p -= lr * wd * p AFTER the Adam update), over every parameter tensor in
 * one launch.  fp32 masters / moments / gradients; each tensor may name a shadow copy (bf16 or f32) that is
 * rewritten in the same pass (the compute-dtype weights the GEMMs read).  The clipped gradients themselves are
 * NOT written back (nothing on the path reads them after the step).
 *
 * Work is cut into chunks of XP_OPT_CHUNK elements of one tensor; `chunk_map` holds (tensor index, chunk index)
 * int32 pairs, one per workgroup, built once by the caller.  `table` and `chunk_map` are DEVICE arrays;
 * `grads_host`, `group_of_host` and `groups_host` are HOST arrays (copied into the kernel-argument block, so a
 * call takes at most XP_OPT_MAX_TENSORS tensors and XP_OPT_MAX_GROUPS hyper-parameter groups; callers with more
 * tensors issue several calls that share one partials array). */
#define XP_OPT_CHUNK 65536
#define XP_OPT_MAX_TENSORS 256
#define XP_OPT_MAX_GROUPS 16
typedef struct {
  void* p; void* m; void* v;            /* fp32 parameter, exp_avg, exp_avg_sq                          */
  void* shadow;                         /* optional copy of p in shadow_dtype (NULL: none)              */
  int64_t numel;
  int32_t shadow_dtype;                 /* XP_BF16 / XP_F32                                             */
  int32_t reserved;
} XpAdamTensor;
typedef struct {
  float lr, beta1, beta2, eps, weight_decay;
  float step_size;                      /* lr * sqrt(1-beta2^t) / (1-beta1^t) (or lr if !correct_bias)  */
} XpAdamGroup;
/* partials[c] = sum of squares of chunk c's gradient elements (fixed order, deterministic). */
int xp_grad_sqnorm_partials(const XpAdamTensor* table, const int32_t* chunk_map, int32_t n_chunks,
                            const void* const* grads_host, int32_t n_tensors, float* partials, void* stream);
/* norm_partials (nullable): ALL chunks' partial sums of squares (n_norm_partials of them); max_norm <= 0: no
 * clipping; grad_norm_out (nullable): receives ||g||_2 before clipping. */
int xp_adamw_step(const XpAdamTensor* table, const int32_t* chunk_map, int32_t n_chunks,
                  const void* const* grads_host, const uint8_t* group_of_host, int32_t n_tensors,
                  const XpAdamGroup* groups_host, int32_t n_groups, const float* norm_partials,
                  int32_t n_norm_partials, float max_norm, float* grad_norm_out, void* stream);

/* ------------------------------------------------------------------------------------------- Loss
 * NCELearnableTempLoss.forward (optimization/loss.py:134-141) on gathered unit-norm features
 * vis[n,d], txt[n,d] (fp32) and the LOG scale parameter: loss = CE(s V T^T, diag) + CE(s T V^T, diag).
 * One launch computes the loss AND d loss / d{vis, txt, log_scale} (the backward just scales them).
 */
size_t xp_nce_loss_workspace_bytes(int64_t n, int64_t d);
int xp_nce_loss(const float* vis, const float* txt, const float* log_scale, float* loss,
                float* d_vis, float* d_txt, float* d_log_scale, int64_t n, int64_t d,
                void* workspace, size_t workspace_bytes, void* stream);

/* NCELearnableTempLoss_vsc_fc.forward (optimization/loss.py:296-324, the pre-training default,
 * pretrain_vip_base_16.json:75) on gathered unit-norm features vis/txt/img/cap [n,d] (fp32): six cross-entropies
 * over video-subtitle, video-caption (off-diagonal negatives of both re-packed behind either positive, :307-314)
 * and frame-caption logits.  One call computes the loss and d loss / d{vis, txt, img, cap, log_scale}. */
size_t xp_vsc_fc_loss_workspace_bytes(int64_t n, int64_t d);
int xp_vsc_fc_loss(const float* vis, const float* txt, const float* img, const float* cap, const float* log_scale,
                   float* loss, float* d_vis, float* d_txt, float* d_img, float* d_cap, float* d_log_scale,
                   int64_t n, int64_t d, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------- Retrieval evaluation
 * validate() of tasks/run_video_retrieval.py:123-203 on gathered unit-norm features (fp32):
 *   sim = text . vis^T                     cal_cossim, utils/metrics.py:3-5
 *   sim *= softmax(theta * sim, axis 0)    DSL re-rank, run_video_retrieval.py:170-171 / np_softmax metrics.py:7-39
 *                                          (multiply == 0: sim = softmax(theta * sim, axis 0) only)
 *   rank of each query's labelled entry    compute_metrics / compute_metrics_multi, metrics.py:41-69
 * xp_retrieval_ranks: query i = row i (transpose = 0) or column i (transpose != 0, the v2t direction, which the
 * reference evaluates on sim.T); label = labels[i] or i; greater[i] / equal[i] = number of candidates with a score
 * above / equal to the labelled one (the label sits at sorted positions greater .. greater+equal-1). */
int xp_sim_matrix(const float* a, const float* b, float* sim, int64_t na, int64_t nb, int64_t d, void* stream);
int xp_dsl_rerank(float* sim, int64_t n, int64_t m, float theta, int32_t multiply, void* stream);
int xp_retrieval_ranks(const float* sim, const int64_t* labels, int64_t n, int64_t m, int32_t transpose,
                       int32_t* greater, int32_t* equal, void* stream);

/* ------------------------------------------------------------------------------------ Encoder layer
 * One CLIPEncoderLayer.forward / backward per call (modeling/CLIP_ViP.py:444-460: x + Attn(LN1(x)), then x + MLP(LN2(x));
 * attention = CLIPAttention.forward2 :332-381 (XP_ATTN_PROXY) or .forward :266-330 (XP_ATTN_CAUSAL); MLP = CLIPMLP :392-396).
 * Host-side sequencing of the entry points above -- launch for launch the same arithmetic as calling them one by one -- so
 * that a training step costs ~50 native calls instead of ~800 foreign-function calls.  Activations are [rows = B*S, D]
 * row-major in `dtype`; weights are the `dtype` copies in the reference's [out, in] layout with q/k/v concatenated row-wise
 * (Wqkv [3D, D], bqkv [3D]); biases, LayerNorm parameters, statistics and all parameter gradients are float. */
typedef struct XpLayerDims {
  int64_t rows, D, Dff, B, S, heads;    /* rows == B*S, D == heads*64                                               */
  int64_t M, N, L;                      /* XP_ATTN_PROXY: S == M + N*L; XP_ATTN_CAUSAL: pass 0, 1, S                 */
  int32_t attn_mode, dtype;
  float q_scale, ln_eps;                /* head_dim^-0.5 (:341), 1e-5                                                */
} XpLayerDims;

typedef struct XpLayerFwd {
  XpLayerDims dims;
  const void* x; const void* Wqkv; const void* Wo; const void* W1; const void* W2;
  const float* ln1_w; const float* ln1_b; const float* bqkv; const float* bo;
  const float* ln2_w; const float* ln2_b; const float* b1; const float* b2;
  const int64_t* pad_mask;              /* [B,S] 1/0 or NULL (text tower)                                            */
  /* outputs -- everything the backward needs stays in caller-owned buffers; pre == NULL: forward-only pass, the MLP
   * pre-activation (needed by the backward alone) is not written */
  void* h1; void* qkv; void* attn_o; void* x2; void* h2; void* pre; void* act; void* x3;
  float* mean1; float* rstd1; float* mean2; float* rstd2; float* stats;   /* stats [B,heads,S,2]                    */
  void* workspace; size_t workspace_bytes;                                /* >= xp_encoder_layer_fwd_workspace_bytes */
  /* optional (bf16): fp32 side rows of the residual stream -- rows r with r % side_S < side_M, stored at side row
   * (r / side_S) * side_M + r % side_S of a [.., D] fp32 buffer: the M proxy tokens of every video sample (side_S = S, side_M = M),
   * or the whole stream of the small text tower (side_S = side_M = 1).  side_in = those rows of x in fp32 (read by LayerNorm 1 and as
   * out_proj's residual operand), side_out = those rows of x3 in fp32, side_x2 = those rows of the intermediate x2 (kept for the
   * backward's second LayerNorm; NULL: they live in the workspace, forward-only pass).  side_in NULL: plain bf16 stream.  See
   * XpGemmDesc::resid_side. */
  const float* side_in; float* side_out; int64_t side_S; int32_t side_M; int32_t reserved;
  float* side_x2;
} XpLayerFwd;
size_t xp_encoder_layer_fwd_workspace_bytes(const XpLayerDims* dims);
int xp_encoder_layer_fwd(const XpLayerFwd* args, void* stream);

typedef struct XpLayerBwd {
  XpLayerDims dims;
  /* saved by the forward */
  const void* x; const void* h1; const void* qkv; const void* attn_o; const void* x2; const void* h2; const void* pre;
  const void* act; const void* Wqkv; const void* Wo; const void* W1; const void* W2;
  const float* ln1_w; const float* ln2_w; const float* mean1; const float* rstd1; const float* mean2; const float* rstd2;
  const float* stats; const int64_t* pad_mask;
  const void* dx3;                      /* gradient of the layer output                                               */
  void* dx;                             /* gradient of the layer input                                                */
  /* parameter gradients (float); a NULL pointer skips that gradient's kernels (frozen parameters, VidCLIP.py:96-103) */
  float* dln1_w; float* dln1_b; float* dwqkv; float* dbqkv; float* dwo; float* dbo;
  float* dln2_w; float* dln2_b; float* dw1; float* db1; float* dw2; float* db2;
  void* workspace; size_t workspace_bytes;                                /* >= xp_encoder_layer_bwd_workspace_bytes */
  /* optional: the forward's fp32 side rows of x (side_in) and x2 (side_x2), read by the two LayerNorm backward passes */
  const float* side_in; const float* side_x2; int64_t side_S; int32_t side_M; int32_t reserved;
} XpLayerBwd;
size_t xp_encoder_layer_bwd_workspace_bytes(const XpLayerDims* dims);
int xp_encoder_layer_bwd(const XpLayerBwd* args, void* stream);
/* The library's second stream of the current device (the one xp_encoder_layer_bwd issues the weight-gradient GEMMs on; created at the
 * device's highest priority on first use), or NULL when XPRETRAIN_WGRAD_STREAM=0.  It is idle outside xp_encoder_layer_bwd calls: the
 * host side runs the second half-batch chain of the video tower's forward on it instead of creating one more stream (a process gets few
 * hardware queues). */
void* xp_side_stream(void);

/* -------------------------------------------------------------------------------------- Diagnostics
 * Hardware-layout probes used by tests/test_probe_gpu.py to pin the MFMA / LDS-transpose lane maps
 * this library relies on (out buffers are small device arrays; see csrc/probe.hip). */
/* cycle-stamp trace of the 128x128 GEMM main loop (tools/gemm_trace.py); pass NULL to disable.  buffer: >= 1 KiB */
int xp_debug_set_gemm_trace(void* device_buffer);
/* In-step timing of ONE GEMM shape: while armed, every xp_gemm call with these (M, N, K, epilogue, operand layouts, split_k) is
 * bracketed by a pair of HIP events on the stream it is launched on (up to max_launches calls); ..._read disarms, waits for the events and
 * returns the elapsed times in ms (return value: number of bracketed launches, -1 on error).  bench.py times the dominant kernel of
 * the step with it INSIDE training steps (isolated launches run at a different power / cache operating point). */
int xp_debug_gemm_timer_arm(int64_t M, int64_t N, int64_t K, int32_t epilogue, int32_t a_kstrided, int32_t b_kstrided, int32_t split_k,
                            int32_t max_launches);
int32_t xp_debug_gemm_timer_read(float* ms, int32_t cap);
/* cycle stamps of one attention-forward workgroup (start, loads issued, loads landed, loop end); NULL disables */
int xp_debug_set_attn_trace(void* device_buffer);
/* resident workgroups per CU of the default bf16 GEMM kernel at `lds_bytes` of dynamic LDS (occupancy API) */
int xp_debug_gemm_occupancy(int lds_bytes);
int xp_probe_mfma_bf16(const void* a, const void* b, float* c, void* stream);
int xp_probe_mfma_f32(const float* a, const float* b, float* c, void* stream);
/* packed-fp32 self-check (csrc/probe.hip): err[(variant*64 + lane)*2 + half] += mismatches between one v_pk_*_f32 form and
 * the same arithmetic in unpacked VALU instructions; 15 variants */
int xp_probe_pk_f32(void* err /*15*64*2 u32, zeroed by the caller*/, int32_t iters, int32_t blocks, uint32_t seed, void* stream);
/* memory-bound copy of nbytes (multiple of 16) repeated iters times by `blocks` 256-thread workgroups: a one-GPU stand-in for a
 * collective's channel kernels beside the backward pass (hvd.DistributedOptimizer's all-reduce, run_pretrain.py:224-227) */
int xp_probe_stream_copy(void* dst, const void* src, int64_t nbytes, int32_t blocks, int32_t iters, void* stream);
/* the same copy in a kernel with the register / LDS footprint of RCCL's gfx950 collective kernel (one wave per SIMD, 19,744 B LDS) */
int xp_probe_stream_copy_fat(void* dst, const void* src, int64_t nbytes, int32_t blocks, int32_t iters, void* stream);
int xp_probe_tr16(const void* in /*4096 u16*/, const int32_t* lane_byte_off /*64*/, void* out /*64*4 u16*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif
