// One CLIPEncoderLayer pass per C-ABI call (modeling/CLIP_ViP.py:444-460 with CLIPAttention.forward2 :332-381 or
// .forward :266-330 and CLIPMLP :392-396): the 8 forward / ~21 backward kernel launches of a layer are issued from native
// code.  Pure host-side sequencing over the public entry points of this library (xp_layernorm_*, xp_gemm, xp_attn_*,
// xp_colsum_partials, xp_splitk_reduce, xp_reduce_rows_batch) -- no kernel of its own -- so the arithmetic is identical,
// launch for launch, to driving those entry points one by one (the Python op-by-op path, functional.EncoderLayerFn with
// XPRETRAIN_LAYER_CALLS=0; tests compare the two bit for bit).  Why: ~810 launches per training step cost 13.6 ms of
// Python / ctypes time against 16.7 ms of GPU time (BENCH_r01); from C++ a launch costs 3-4 us.
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <mutex>

namespace {

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct Carver {            // bump allocator over the caller's workspace
  char* base; size_t off, cap;
  void* take(size_t n) { void* p = base ? base + off : nullptr; off += align256(n); return p; }
};

XpGemmDesc gemm_desc(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int dtype) {
  XpGemmDesc d;
  memset(&d, 0, sizeof(d));
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K;
  d.lda = K; d.ldb = K; d.ldc = N; d.ldr = N; d.ldaux = N;
  d.in_dtype = dtype; d.out_dtype = dtype; d.split_k = 1; d.scale = 1.0f;
  return d;
}

// dW[n_out, n_in] = dY[rows, n_out]^T . X[rows, n_in], fp32: both operands k-strided, split-K chosen by the library
// slack: nothing waits for this launch soon (xp_gemm_auto_split_slack: fewer, longer slabs)
int wgrad(const void* dy, const void* x, float* dw, int64_t rows, int64_t n_out, int64_t n_in, int dtype, float* slabs,
          size_t slab_bytes, void* st, bool slack) {
  XpGemmDesc d = gemm_desc(dy, x, dw, n_out, n_in, rows, dtype);
  d.a_kstrided = d.b_kstrided = 1; d.lda = n_out; d.ldb = n_in; d.out_dtype = XP_F32;
  const int split = slack ? xp_gemm_auto_split_slack(&d) : xp_gemm_auto_split(&d);
  if (split <= 1) return xp_gemm(&d, st);
  XP_REQUIRE(slab_bytes >= (size_t)split * n_out * n_in * sizeof(float), "xp_encoder_layer_bwd: split-K slab space too small");
  d.C = slabs; d.split_k = split;
  int rc = xp_gemm(&d, st);
  if (rc) return rc;
  if (xp_debug_flag("skip_splitk_reduce")) return XP_OK;      // measurement only (wrong gradients): what the four reduces of a layer cost the step
  return xp_splitk_reduce(slabs, dw, n_out * n_in, split, 0, st);
}

size_t wgrad_slab_bytes(int64_t rows, int64_t n_out, int64_t n_in, int dtype) {
  XpGemmDesc d = gemm_desc(nullptr, nullptr, nullptr, n_out, n_in, rows, dtype);
  d.a_kstrided = d.b_kstrided = 1; d.lda = n_out; d.ldb = n_in; d.out_dtype = XP_F32;
  const int a = xp_gemm_auto_split(&d), b = xp_gemm_auto_split_slack(&d), split = a > b ? a : b;
  return split <= 1 ? 0 : (size_t)split * n_out * n_in * sizeof(float);
}

struct Defer {             // the layer's deferred second-level reductions (bias / LayerNorm-parameter gradients)
  XpReduceSeg segs[XP_REDUCE_MAX_SEGS]; int n = 0;
  void add(const float* in, float* out, int64_t stride, int nrows, int width) {
    if (!out) return;
    XpReduceSeg& s = segs[n++];
    s.in = in; s.out = out; s.stride = stride; s.nrows = nrows; s.width = width; s.accumulate = 0; s.reserved = 0;
  }
};

// ---- weight-gradient GEMMs on a second stream (default; XPRETRAIN_WGRAD_STREAM=0 puts them back on the caller's stream) ------
// The four dW GEMMs of a layer (+ their split-K reduces) are off the critical path of the backward pass: nothing in the layer
// reads them.  On a stream of their own they run BESIDE the dX chain (GEMM -> LayerNorm -> GEMM -> attention -> GEMM -> LayerNorm):
// their workgroups take the CUs the 222-tile dX GEMMs leave idle, the tails / launch boundaries of either stream, and overlap the
// HBM-bound LayerNorm / attention / reduce kernels with MFMA work.  Ordering is by events only; the main stream joins the side
// stream before the call returns, so buffer lifetimes (workspace reuse by the next layer, the caching allocator) are unchanged.
// Measured in the step (interleaved whole-step A/B on one box, profiles/r04a_in_step_ab_wgrad_stream_chunk_major.txt):
// 16.57 -> 16.22 ms per step; results are bit-identical (same kernels, same arguments).
struct WgradSide {
  hipStream_t side = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // main-stream progress marks: entry, dpre, dx2, dqkv
  hipEvent_t done = nullptr;                                  // side stream: last dW of the call finished
  bool ok = false;
};
WgradSide* wgrad_side() {
  static const bool on = !getenv("XPRETRAIN_WGRAD_STREAM") || atoi(getenv("XPRETRAIN_WGRAD_STREAM")) != 0;      // default on
  if (!on) return nullptr;
  static std::mutex mu;
  static WgradSide per_dev[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  WgradSide& w = per_dev[dev];
  if (!w.side) {
    // DEFAULT priority.  Highest priority was worth 0.05 ms per step on one GPU (profiles/r04j_in_step_ab_wgrad_priority_split.txt) and
    // cost 15-30 ms per step as soon as a process group's collectives ran beside it (a ONE-rank RCCL group on one GPU: 36-52 ms per
    // step against 21 with the default priority, profiles/r04v_forced_one_rank_collectives_stream_matrix.txt) -- a high-priority HSA
    // queue beside the collective library's streams starves the queues the critical chain runs on.
    bool good = hipStreamCreateWithPriority(&w.side, hipStreamNonBlocking, 0) == hipSuccess;
    for (int i = 0; i < 4 && good; ++i) good = hipEventCreateWithFlags(&w.ev[i], hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&w.done, hipEventDisableTiming) == hipSuccess;
    w.ok = good;
  }
  return w.ok ? &w : nullptr;
}

int check_dims(const char* name, const XpLayerDims& d) {
  XP_REQUIRE(d.rows > 0 && d.D > 0 && d.Dff > 0 && d.B > 0 && d.S > 0 && d.heads > 0, "%s: empty dimension", name);
  XP_REQUIRE(d.rows == d.B * d.S && d.D == d.heads * 64, "%s: rows != B*S or D != heads*64", name);
  XP_REQUIRE(d.dtype == XP_BF16 || d.dtype == XP_F32, "%s: bad dtype %d", name, d.dtype);
  return XP_OK;
}

}  // namespace

extern "C" void* xp_side_stream(void) {
  WgradSide* w = wgrad_side();
  return w ? (void*)w->side : nullptr;
}

extern "C" size_t xp_encoder_layer_fwd_workspace_bytes(const XpLayerDims* d) {
  if (!d) return 0;
  // + the fp32 side rows of x2 (video: the proxy tokens; text: every row), used when XpLayerFwd::side_in is given
  const size_t side_rows = d->attn_mode == XP_ATTN_PROXY ? (size_t)d->B * d->M : (size_t)d->rows;
  return align256(xp_attn_workspace_bytes(d->attn_mode, d->B, d->heads, d->M, d->N, d->L)) + align256(side_rows * d->D * sizeof(float)) + 256;
}

extern "C" int xp_encoder_layer_fwd(const XpLayerFwd* a, void* st) {
  XP_REQUIRE(a, "xp_encoder_layer_fwd: null argument");
  const XpLayerDims& d = a->dims;
  int rc = check_dims("xp_encoder_layer_fwd", d);
  if (rc) return rc;
  XP_REQUIRE(a->x && a->Wqkv && a->Wo && a->W1 && a->W2 && a->ln1_w && a->ln1_b && a->bqkv && a->bo && a->ln2_w && a->ln2_b &&
             a->b1 && a->b2 && a->h1 && a->qkv && a->attn_o && a->x2 && a->h2 && a->act && a->x3 && a->mean1 &&
             a->rstd1 && a->mean2 && a->rstd2 && a->stats, "xp_encoder_layer_fwd: null pointer");
  const int64_t rows = d.rows, D = d.D, Dff = d.Dff;
  const int dt = d.dtype;
  // fp32 side rows of the residual stream (the M proxy tokens of every sample): x rows in side_in, x2 rows in the workspace,
  // x3 rows in side_out
  const bool sided = a->side_in != nullptr;
  XP_REQUIRE(!sided || (a->side_out && dt == XP_BF16 && a->side_S > 0 && a->side_M > 0 && a->side_M <= a->side_S),
             "xp_encoder_layer_fwd: side rows need side_in and side_out, bf16 and 0 < side_M <= side_S");
  const size_t attn_ws = align256(xp_attn_workspace_bytes(d.attn_mode, d.B, d.heads, d.M, d.N, d.L));
  const int64_t sS = a->side_S;
  const int32_t sM = a->side_M;
  XP_REQUIRE(!sided || a->workspace_bytes >= attn_ws + align256((size_t)(cdiv(rows, sS) * sM) * D * sizeof(float)),
             "xp_encoder_layer_fwd: workspace too small for the side rows");
  float* side_x2 = !sided ? nullptr : a->side_x2 ? a->side_x2 : reinterpret_cast<float*>(static_cast<char*>(a->workspace) + attn_ws);
  // h1 = LN1(x)
  if ((rc = xp_layernorm_fwd_side(a->x, D, a->ln1_w, a->ln1_b, a->h1, D, a->mean1, a->rstd1, rows, D, d.ln_eps, dt,
                                  a->side_in, nullptr, sS, sM, sM, st))) return rc;
  // qkv = (h1 Wqkv^T + b), q columns scaled by dh^-0.5 (:341)
  XpGemmDesc g = gemm_desc(a->h1, a->Wqkv, a->qkv, rows, 3 * D, D, dt);
  g.epilogue = XP_EPI_BIAS_QSCALE; g.bias = a->bqkv; g.scale = d.q_scale; g.scale_cols = D;
  if ((rc = xp_gemm(&g, st))) return rc;
  if ((rc = xp_attn_fwd(a->qkv, 3 * D, a->attn_o, D, a->stats, a->pad_mask, d.attn_mode, d.B, d.heads, d.S, d.M, d.N, d.L, dt,
                        a->workspace, a->workspace_bytes, st))) return rc;
  // x2 = x + attn_o Wo^T + bo
  g = gemm_desc(a->attn_o, a->Wo, a->x2, rows, D, D, dt);
  g.epilogue = XP_EPI_BIAS_RESID; g.bias = a->bo; g.resid = a->x;
  if (sided) { g.resid_side = a->side_in; g.out_side = side_x2; g.side_S = sS; g.side_M = sM; }
  if ((rc = xp_gemm(&g, st))) return rc;
  if ((rc = xp_layernorm_fwd_side(a->x2, D, a->ln2_w, a->ln2_b, a->h2, D, a->mean2, a->rstd2, rows, D, d.ln_eps, dt,
                                  side_x2, nullptr, sS, sM, sM, st))) return rc;
  // pre = h2 W1^T + b1 ; act = quick_gelu(pre)
  g = gemm_desc(a->h2, a->W1, a->act, rows, Dff, D, dt);
  g.epilogue = XP_EPI_BIAS_GELU; g.bias = a->b1; g.aux = a->pre;
  if (xp_debug_flag("fc1_no_pre")) g.aux = nullptr;      // measurement only (tools/fc1_one_output.py): the backward then reads garbage
  if ((rc = xp_gemm(&g, st))) return rc;
  // x3 = x2 + act W2^T + b2
  g = gemm_desc(a->act, a->W2, a->x3, rows, D, Dff, dt);
  g.epilogue = XP_EPI_BIAS_RESID; g.bias = a->b2; g.resid = a->x2;
  if (sided) { g.resid_side = side_x2; g.out_side = a->side_out; g.side_S = sS; g.side_M = sM; }
  return xp_gemm(&g, st);
}

// workspace layout of the backward: [dpre | dh2 | dx2 | dattn | dqkv | dh1] activations-gradient temporaries, split-K slabs,
// six deferred partial-row slots, the batched-reduce scratch, the attention workspace
namespace {
struct BwdPlan {
  size_t esz, dpre, dh, dqkv, slabs, cs_pre, cs_dx3, ln2, cs_qkv, ln1, red, attn, total;
  int64_t cs_pre_rows, cs_dx3_rows, cs_qkv_rows, ln_rows;
  bool cs_qkv_fused;
};
BwdPlan plan_bwd(const XpLayerDims& d) {
  BwdPlan p;
  memset(&p, 0, sizeof(p));
  const int64_t rows = d.rows, D = d.D, Dff = d.Dff;
  p.esz = d.dtype == XP_BF16 ? 2 : 4;
  p.dpre = align256(rows * Dff * p.esz); p.dh = align256(rows * D * p.esz); p.dqkv = align256(rows * 3 * D * p.esz);
  size_t s = wgrad_slab_bytes(rows, D, Dff, d.dtype);
  size_t t = wgrad_slab_bytes(rows, Dff, D, d.dtype); if (t > s) s = t;
  t = wgrad_slab_bytes(rows, D, D, d.dtype); if (t > s) s = t;
  t = wgrad_slab_bytes(rows, 3 * D, D, d.dtype); if (t > s) s = t;
  p.slabs = align256(s);
  // fc1's bias gradient: fused into the dX GEMM epilogue where the library offers it, else a column-sum pass over dpre
  XpGemmDesc g = gemm_desc(nullptr, nullptr, nullptr, rows, Dff, D, d.dtype);
  g.b_kstrided = 1; g.ldb = Dff; g.epilogue = XP_EPI_GELU_BWD; g.ldr = Dff;
  g.resid = &g;                          // (only tested for non-NULL by the planning queries)
  p.cs_pre_rows = xp_gemm_colsum_rows(&g);
  const int64_t pre_rows = p.cs_pre_rows > 0 ? p.cs_pre_rows : xp_colsum_partial_rows(rows, Dff);
  p.cs_pre = align256(pre_rows * Dff * sizeof(float));
  p.cs_dx3_rows = 0; p.cs_dx3 = 0;           // fc2's bias gradient = column sums of dx3: taken by the second LayerNorm's backward
  // the q/k/v bias gradients: out of the attention backward kernels where they offer it, else a column-sum pass over dqkv
  p.cs_qkv_rows = xp_attn_bwd_colsum_rows(d.attn_mode, d.B, d.heads, d.S, d.M, d.N, d.L, d.dtype);
  p.cs_qkv_fused = p.cs_qkv_rows > 0;
  if (!p.cs_qkv_fused) p.cs_qkv_rows = xp_colsum_partial_rows(rows, 3 * D);
  p.cs_qkv = align256(p.cs_qkv_rows * 3 * D * sizeof(float));
  p.ln_rows = xp_layernorm_bwd_partial_rows(rows);
  p.ln2 = p.ln1 = align256(xp_layernorm_bwd_workspace_bytes(rows, D));
  p.red = align256((size_t)XP_REDUCE_MAX_SEGS * 32 * (size_t)(3 * D > Dff ? 3 * D : Dff) * sizeof(float) + 16);
  p.attn = align256(xp_attn_workspace_bytes(d.attn_mode, d.B, d.heads, d.M, d.N, d.L));
  p.total = p.dpre + 4 * p.dh + p.dqkv + p.slabs + p.cs_pre + p.cs_dx3 + p.ln2 + p.cs_qkv + p.ln1 + p.red + p.attn + 256;
  return p;
}
}  // namespace

extern "C" size_t xp_encoder_layer_bwd_workspace_bytes(const XpLayerDims* d) {
  if (!d || d->rows <= 0) return 0;
  return plan_bwd(*d).total;
}

extern "C" int xp_encoder_layer_bwd(const XpLayerBwd* a, void* st) {
  XP_REQUIRE(a, "xp_encoder_layer_bwd: null argument");
  const XpLayerDims& d = a->dims;
  int rc = check_dims("xp_encoder_layer_bwd", d);
  if (rc) return rc;
  XP_REQUIRE(a->x && a->h1 && a->qkv && a->attn_o && a->x2 && a->h2 && a->pre && a->act && a->Wqkv && a->Wo && a->W1 && a->W2 &&
             a->ln1_w && a->ln2_w && a->mean1 && a->rstd1 && a->mean2 && a->rstd2 && a->stats && a->dx3 && a->dx,
             "xp_encoder_layer_bwd: null pointer");
  XP_REQUIRE((!a->side_in && !a->side_x2) || (a->side_in && a->side_x2 && d.dtype == XP_BF16 && a->side_S > 0 && a->side_M > 0 &&
                                               a->side_M <= a->side_S),
             "xp_encoder_layer_bwd: side rows need side_in and side_x2, bf16 and 0 < side_M <= side_S");
  const BwdPlan p = plan_bwd(d);
  XP_REQUIRE(a->workspace && a->workspace_bytes >= p.total, "xp_encoder_layer_bwd: workspace too small (%zu < %zu)",
             a->workspace_bytes, p.total);
  Carver ws{(char*)a->workspace, 0, a->workspace_bytes};
  void* dpre = ws.take(p.dpre); void* dh2 = ws.take(p.dh); void* dx2 = ws.take(p.dh); void* dattn = ws.take(p.dh);
  void* dqkv = ws.take(p.dqkv); void* dh1 = ws.take(p.dh);
  float* slabs = (float*)ws.take(p.slabs);
  float* cs_pre = (float*)ws.take(p.cs_pre); float* cs_dx3 = (float*)ws.take(p.cs_dx3);
  float* ln2_part = (float*)ws.take(p.ln2); float* cs_qkv = (float*)ws.take(p.cs_qkv); float* ln1_part = (float*)ws.take(p.ln1);
  void* red_ws = ws.take(p.red); void* attn_ws = ws.take(p.attn);
  const int64_t rows = d.rows, D = d.D, Dff = d.Dff;
  const int dt = d.dtype;
  Defer df;
  // dW GEMMs beside the dX chain (video tower only: the text tower is 256 rows on a side stream of its own already)
  WgradSide* wsd = (d.attn_mode == XP_ATTN_PROXY && rows >= 4096) ? wgrad_side() : nullptr;
  hipStream_t mst = (hipStream_t)st;
  void* wst = wsd ? (void*)wsd->side : st;
  auto mark = [&](int i) -> int {           // side stream: everything the main stream has enqueued so far must finish first
    if (!wsd) return XP_OK;
    if (hipEventRecord(wsd->ev[i], mst) != hipSuccess || hipStreamWaitEvent(wsd->side, wsd->ev[i], 0) != hipSuccess) {
      xp_set_error("xp_encoder_layer_bwd: event hand-off to the weight-gradient stream failed");
      return XP_ERR_LAUNCH;
    }
    return XP_OK;
  };
  if ((rc = mark(0))) return rc;

  // ---- MLP: x3 = x2 + fc2(quick_gelu(fc1(LN2(x2))))
  XpGemmDesc g = gemm_desc(a->dx3, a->W2, dpre, rows, Dff, D, dt);            // dpre = (dx3 . W2) * quick_gelu'(pre)
  g.b_kstrided = 1; g.ldb = Dff; g.epilogue = XP_EPI_GELU_BWD; g.resid = a->pre; g.ldr = Dff;
  if (a->db1 && p.cs_pre_rows > 0) g.colsum_partials = cs_pre;
  if ((rc = xp_gemm(&g, st))) return rc;
  if (a->db1) {
    if (p.cs_pre_rows > 0) df.add(cs_pre, a->db1, Dff, (int)p.cs_pre_rows, (int)Dff);
    else {
      const int64_t r = xp_colsum_partial_rows(rows, Dff);
      if ((rc = xp_colsum_partials(dpre, rows, Dff, Dff, dt, cs_pre, p.cs_pre, st))) return rc;
      df.add(cs_pre, a->db1, Dff, (int)r, (int)Dff);
    }
  }
  if (a->dw2 && (rc = wgrad(a->dx3, a->act, a->dw2, rows, D, Dff, dt, slabs, p.slabs, wst, true))) return rc;
  if ((rc = mark(1))) return rc;                                              // dpre is ready for dW1
  g = gemm_desc(dpre, a->W1, dh2, rows, D, Dff, dt);                          // dh2 = dpre . W1
  g.b_kstrided = 1; g.ldb = D;
  if ((rc = xp_gemm(&g, st))) return rc;
  if (a->dw1 && (rc = wgrad(dpre, a->h2, a->dw1, rows, Dff, D, dt, slabs, p.slabs, wst, true))) return rc;
  // dx2 = dx3 + LN2'(dh2); partial rows [dgamma | dbeta | colsum(dx2) | colsum(dx3)] -- out_proj's and fc2's bias gradients
  if ((rc = xp_layernorm_bwd_partials_side(dh2, D, a->x2, D, a->ln2_w, a->mean2, a->rstd2, a->dx3, D, dx2, D, 2, rows, D, dt,
                                           a->side_x2, a->side_S, a->side_M, a->side_M, ln2_part, p.ln2, st))) return rc;
  df.add(ln2_part, a->dln2_w, 4 * D, (int)p.ln_rows, (int)D);
  df.add(ln2_part + D, a->dln2_b, 4 * D, (int)p.ln_rows, (int)D);
  df.add(ln2_part + 2 * D, a->dbo, 4 * D, (int)p.ln_rows, (int)D);
  df.add(ln2_part + 3 * D, a->db2, 4 * D, (int)p.ln_rows, (int)D);
  if ((rc = mark(2))) return rc;                                              // dx2 is ready for dWo
  // ---- attention: x2 = x + out_proj(attn(qkv(LN1(x))))
  g = gemm_desc(dx2, a->Wo, dattn, rows, D, D, dt);                           // dattn = dx2 . Wo
  g.b_kstrided = 1; g.ldb = D;
  if ((rc = xp_gemm(&g, st))) return rc;
  if (a->dwo && (rc = wgrad(dx2, a->attn_o, a->dwo, rows, D, D, dt, slabs, p.slabs, wst, true))) return rc;
  if ((rc = xp_attn_bwd2(a->qkv, 3 * D, a->attn_o, dattn, D, a->stats, a->pad_mask, dqkv, d.q_scale, d.attn_mode, d.B, d.heads,
                         d.S, d.M, d.N, d.L, dt, attn_ws, p.attn, (a->dbqkv && p.cs_qkv_fused) ? cs_qkv : nullptr, st))) return rc;
  if ((rc = mark(3))) return rc;                                              // dqkv is ready for dWqkv
  g = gemm_desc(dqkv, a->Wqkv, dh1, rows, D, 3 * D, dt);                      // dh1 = dqkv . Wqkv
  g.b_kstrided = 1; g.ldb = D;
  if ((rc = xp_gemm(&g, st))) return rc;
  if (a->dwqkv && (rc = wgrad(dqkv, a->h1, a->dwqkv, rows, 3 * D, D, dt, slabs, p.slabs, wst, false))) return rc;
  if (a->dbqkv) {
    if (!p.cs_qkv_fused && (rc = xp_colsum_partials(dqkv, rows, 3 * D, 3 * D, dt, cs_qkv, p.cs_qkv, st))) return rc;
    df.add(cs_qkv, a->dbqkv, 3 * D, (int)p.cs_qkv_rows, (int)(3 * D));
  }
  if ((rc = xp_layernorm_bwd_partials_side(dh1, D, a->x, D, a->ln1_w, a->mean1, a->rstd1, dx2, D, a->dx, D, 0, rows, D, dt,
                                           a->side_in, a->side_S, a->side_M, a->side_M, ln1_part, p.ln1, st))) return rc;
  df.add(ln1_part, a->dln1_w, 2 * D, (int)p.ln_rows, (int)D);
  df.add(ln1_part + D, a->dln1_b, 2 * D, (int)p.ln_rows, (int)D);
  if (df.n) {
    XP_REQUIRE(xp_reduce_rows_batch_workspace_bytes(df.segs, df.n) <= p.red, "xp_encoder_layer_bwd: reduce scratch too small");
    if ((rc = xp_reduce_rows_batch(df.segs, df.n, red_ws, p.red, st))) return rc;
  }
  if (wsd && !xp_debug_flag("no_wgrad_join")) {      // join: the weight gradients (and every workspace the side stream read) belong to the main stream again
    // (XPRETRAIN_DEBUG=no_wgrad_join: measurement only -- races on the shared workspace -- what a lazy join could be worth at most)
    if (hipEventRecord(wsd->done, wsd->side) != hipSuccess || hipStreamWaitEvent(mst, wsd->done, 0) != hipSuccess) {
      xp_set_error("xp_encoder_layer_bwd: joining the weight-gradient stream failed");
      return XP_ERR_LAUNCH;
    }
  }
  return XP_OK;
}
