// NCELearnableTempLoss (optimization/loss.py:134-141) forward AND gradient in one call, fp32 throughout:
//   A = exp(ls) * V T^T                     [n,n]   (V, T: gathered unit-norm features [n,d])
//   loss = mean_i(lse_j A_ij - A_ii) + mean_j(lse_i A_ij - A_jj)
//   G = dloss/dA = (softmax_rows(A) + softmax_cols(A) - 2 I) / n
//   dV = exp(ls) G T ;  dT = exp(ls) G^T V ;  d ls = sum(G * A)
// n = world_size * local_batch is small (64 at 8 GPUs x 8 pairs), so this is latency- not FLOP-bound:
// plain fp32 FMA tiles (bit-stable, no bf16 rounding on the logits that are multiplied by ~100).
#include "common.h"

namespace {

// C[i][j] = alpha * sum_k X(i,k) Y(k,j), generic strides; 32x32 tile, 256 threads, 2x2 per thread.
__global__ __launch_bounds__(256) XP_NO_PK_F32 void sgemm_strided_kernel(const float* __restrict__ X, int64_t sxi, int64_t sxk,
                                                            const float* __restrict__ Y, int64_t syk, int64_t syj,
                                                            float* __restrict__ C, int64_t ldc, int I, int J, int K,
                                                            const float* __restrict__ log_scale, int accumulate = 0) {
  __shared__ float xs[32][33], ys[32][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int k0 = 0; k0 < K; k0 += 32) {
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const int r = e >> 5, c = e & 31;
      // xs[r][c] = X(i0 + r, k0 + c) ; ys[r][c] = Y(k0 + r, j0 + c)
      xs[r][c] = (i0 + r < I && k0 + c < K) ? X[(int64_t)(i0 + r) * sxi + (int64_t)(k0 + c) * sxk] : 0.f;
      ys[r][c] = (k0 + r < K && j0 + c < J) ? Y[(int64_t)(k0 + r) * syk + (int64_t)(j0 + c) * syj] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float a0 = xs[ty][k], a1 = xs[ty + 16][k], b0 = ys[k][tx], b1 = ys[k][tx + 16];
      acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
    }
    __syncthreads();
  }
  const float alpha = log_scale ? expf(*log_scale) : 1.0f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
      if (i < I && j < J) {
        float* c = C + (int64_t)i * ldc + j;
        *c = accumulate ? *c + alpha * acc[a][b] : alpha * acc[a][b];
      }
    }
}

// Small similarity matrices (n <= 128: the local batch of one to a few GPUs): C[i][j] = alpha * <X[i,:], Y[j,:]> with ONE WAVE per
// (i, j) -- 16-byte loads along d, 8 independent accumulator lanes per wave step, a fixed butterfly.  The 32x32-tile kernel above
// runs a [8 x 512] x [512 x 8] problem as one workgroup walking 16 dependent k-steps of strided loads: 39 us of load latency on the
// serial stretch between the forward and the backward (nothing else runs there); this one takes one load round trip.
__global__ __launch_bounds__(256) XP_NO_PK_F32 void logits_small_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                        float* __restrict__ C, int n, int d,
                                                                        const float* __restrict__ log_scale) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n * n) return;
  const int i = pair / n, j = pair - i * n;
  const float* x = X + (int64_t)i * d;
  const float* y = Y + (int64_t)j * d;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int k = lane * 4; k < d; k += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + k), b = *reinterpret_cast<const f32x4*>(y + k);
    s0 += a[0] * b[0]; s1 += a[1] * b[1]; s2 += a[2] * b[2]; s3 += a[3] * b[3];
  }
  float s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) C[(int64_t)i * n + j] = (log_scale ? expf(*log_scale) : 1.0f) * s;
}

// logits A[n][n] = e^ls V T^T: the small form up to n = 128 (d a multiple of 4, 16-byte aligned rows), the tiled form beyond
static void launch_logits(const float* vis, const float* txt, float* A, int64_t n, int64_t d, const float* log_scale, hipStream_t st) {
  if (n <= 128 && d % 4 == 0 && ((uintptr_t)vis & 15) == 0 && ((uintptr_t)txt & 15) == 0) {
    logits_small_kernel<<<(unsigned)cdiv(n * n, 4), 256, 0, st>>>(vis, txt, A, (int)n, (int)d, log_scale);
  } else {
    dim3 gnn((unsigned)cdiv(n, 32), (unsigned)cdiv(n, 32));
    sgemm_strided_kernel<<<gnn, 256, 0, st>>>(vis, d, 1, txt, 1, d, A, n, (int)n, (int)n, (int)d, log_scale);
  }
}

// blocks 0..n-1: row i -> lse_r[i]; blocks n..2n-1: column j -> lse_c[j]   (one wave each)
__global__ void lse_kernel(const float* __restrict__ A, float* __restrict__ lse_r, float* __restrict__ lse_c, int n) {
  const int lane = threadIdx.x;
  const bool col = blockIdx.x >= n;
  const int idx = col ? blockIdx.x - n : blockIdx.x;
  const int64_t step = col ? n : 1, base = col ? idx : (int64_t)idx * n;
  float m = -INFINITY;
  for (int t = lane; t < n; t += 64) m = fmaxf(m, A[base + t * step]);
  m = wave_max(m);
  float s = 0.f;
  for (int t = lane; t < n; t += 64) s += expf(A[base + t * step] - m);
  s = wave_sum(s);
  if (lane == 0) (col ? lse_c : lse_r)[idx] = m + logf(s);
}

// block i: G[i][:] ; part[i] = (loss_i, dls_i)
__global__ void grad_kernel(const float* __restrict__ A, const float* __restrict__ lse_r, const float* __restrict__ lse_c,
                            float* __restrict__ G, float* __restrict__ part, int n) {
  const int lane = threadIdx.x, i = blockIdx.x;
  const float inv = 1.0f / (float)n, lr = lse_r[i];
  float dls = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float a = A[(int64_t)i * n + j];
    float g = (expf(a - lr) + expf(a - lse_c[j]) - (i == j ? 2.0f : 0.0f)) * inv;
    G[(int64_t)i * n + j] = g;
    dls += g * a;
  }
  dls = wave_sum(dls);
  if (lane == 0) {
    const float aii = A[(int64_t)i * n + i];
    part[2 * i] = ((lr - aii) + (lse_c[i] - aii)) * inv;
    part[2 * i + 1] = dls;
  }
}

__global__ void finish_kernel(const float* __restrict__ part, float* loss, float* d_ls, int n) {
  const int lane = threadIdx.x;
  float a = 0.f, b = 0.f;
  for (int i = lane; i < n; i += 64) { a += part[2 * i]; b += part[2 * i + 1]; }
  a = wave_sum(a); b = wave_sum(b);
  if (lane == 0) { *loss = a; *d_ls = b; }
}

// ---- NCELearnableTempLoss_vsc_fc (optimization/loss.py:288-324) ---------------------------------------------------------
//   S1 = s V T^T (video-subtitle), S2 = s V C^T (video-caption), S3 = s I C^T (frame-caption), s = exp(ls)
//   loss = mean_i [ (c1_i - S1_ii) + (c2_i - S2_ii) + (ra_i - S1_ii) + (rb_i - S2_ii) + (c3_i - S3_ii) + (r3_i - S3_ii) ]
//   c*_j = lse over column j;  r3_i = lse over row i of S3;
//   ra_i = lse(S1[i,:] U S2[i,j!=i])   (loss.py:311 [pos, neg, neg_2] with pos = S1_ii)
//   rb_i = lse(S1[i,j!=i] U S2[i,:])   (loss.py:312 with pos = S2_ii)
// blocks 0..n-1: row i -> ra, rb, r3 ; blocks n..2n-1: column j -> c1, c2, c3   (one wave each)
__global__ void vsc_lse_kernel(const float* __restrict__ S1, const float* __restrict__ S2, const float* __restrict__ S3,
                               float* __restrict__ st, int n) {
  const int lane = threadIdx.x;
  float* ra = st; float* rb = st + n; float* r3 = st + 2 * n; float* c1 = st + 3 * n; float* c2 = st + 4 * n; float* c3 = st + 5 * n;
  if ((int)blockIdx.x < n) {
    const int i = blockIdx.x;
    const float* a = S1 + (int64_t)i * n; const float* b = S2 + (int64_t)i * n; const float* c = S3 + (int64_t)i * n;
    float mab = -INFINITY, m3 = -INFINITY;
    for (int j = lane; j < n; j += 64) { mab = fmaxf(mab, fmaxf(a[j], b[j])); m3 = fmaxf(m3, c[j]); }
    mab = wave_max(mab); m3 = wave_max(m3);
    float sa = 0.f, sb = 0.f, s3 = 0.f;
    for (int j = lane; j < n; j += 64) {
      const float ea = expf(a[j] - mab), eb = expf(b[j] - mab);
      sa += ea + (j == i ? 0.f : eb);
      sb += (j == i ? 0.f : ea) + eb;
      s3 += expf(c[j] - m3);
    }
    sa = wave_sum(sa); sb = wave_sum(sb); s3 = wave_sum(s3);
    if (lane == 0) { ra[i] = mab + logf(sa); rb[i] = mab + logf(sb); r3[i] = m3 + logf(s3); }
  } else {
    const int j = blockIdx.x - n;
    float m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
    for (int i = lane; i < n; i += 64) {
      m1 = fmaxf(m1, S1[(int64_t)i * n + j]); m2 = fmaxf(m2, S2[(int64_t)i * n + j]); m3 = fmaxf(m3, S3[(int64_t)i * n + j]);
    }
    m1 = wave_max(m1); m2 = wave_max(m2); m3 = wave_max(m3);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = lane; i < n; i += 64) {
      s1 += expf(S1[(int64_t)i * n + j] - m1); s2 += expf(S2[(int64_t)i * n + j] - m2); s3 += expf(S3[(int64_t)i * n + j] - m3);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if (lane == 0) { c1[j] = m1 + logf(s1); c2[j] = m2 + logf(s2); c3[j] = m3 + logf(s3); }
  }
}

// block i: rows i of G1, G2, G3 (d loss / d S*) ; part[i] = (loss_i, dls_i)
__global__ void vsc_grad_kernel(const float* __restrict__ S1, const float* __restrict__ S2, const float* __restrict__ S3,
                                const float* __restrict__ st, float* __restrict__ G1, float* __restrict__ G2,
                                float* __restrict__ G3, float* __restrict__ part, int n) {
  const int lane = threadIdx.x, i = blockIdx.x;
  const float* ra = st; const float* rb = st + n; const float* r3 = st + 2 * n;
  const float* c1 = st + 3 * n; const float* c2 = st + 4 * n; const float* c3 = st + 5 * n;
  const float inv = 1.0f / (float)n, rai = ra[i], rbi = rb[i], r3i = r3[i];
  float dls = 0.f;
  for (int j = lane; j < n; j += 64) {
    const int64_t o = (int64_t)i * n + j;
    const float a = S1[o], b = S2[o], c = S3[o];
    const bool dg = i == j;
    const float g1 = (expf(a - c1[j]) + expf(a - rai) + (dg ? -2.0f : expf(a - rbi))) * inv;
    const float g2 = (expf(b - c2[j]) + expf(b - rbi) + (dg ? -2.0f : expf(b - rai))) * inv;
    const float g3 = (expf(c - c3[j]) + expf(c - r3i) - (dg ? 2.0f : 0.0f)) * inv;
    G1[o] = g1; G2[o] = g2; G3[o] = g3;
    dls += g1 * a + g2 * b + g3 * c;
  }
  dls = wave_sum(dls);
  if (lane == 0) {
    const int64_t d = (int64_t)i * n + i;
    part[2 * i] = ((c1[i] - S1[d]) + (c2[i] - S2[d]) + (rai - S1[d]) + (rbi - S2[d]) + (c3[i] - S3[d]) + (r3i - S3[d])) * inv;
    part[2 * i + 1] = dls;
  }
}

// ---- retrieval evaluation (tasks/run_video_retrieval.py:150-171, utils/metrics.py) --------------------------------
// DSL re-rank: sim[i][j] *= softmax_i(theta * sim[i][j])   (np_softmax(sim * 100, axis=0), metrics.py:7-39); one wave per column
__global__ void dsl_rerank_kernel(float* __restrict__ sim, int n, int m, float theta, int multiply) {
  const int lane = threadIdx.x, j = blockIdx.x;
  float mx = -INFINITY;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sim[(int64_t)i * m + j] * theta);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < n; i += 64) sum += expf(sim[(int64_t)i * m + j] * theta - mx);
  sum = wave_sum(sum);
  for (int i = lane; i < n; i += 64) {
    const float x = sim[(int64_t)i * m + j];
    const float sm = expf(x * theta - mx) / sum;
    sim[(int64_t)i * m + j] = multiply ? x * sm : sm;
  }
}

// per query row i: how many entries beat / tie the labelled entry (compute_metrics: position of the label in the
// descending sort, metrics.py:41-53; ties occupy `equal` consecutive positions there).  transpose != 0 ranks columns.
__global__ void retrieval_ranks_kernel(const float* __restrict__ sim, const int64_t* __restrict__ labels, int n, int m,
                                       int transpose, int* __restrict__ greater, int* __restrict__ equal) {
  const int lane = threadIdx.x, i = blockIdx.x;
  const int cnt = transpose ? n : m;                       // candidates per query
  const int64_t qs = transpose ? 1 : m, cs = transpose ? m : 1;
  const int64_t lab = labels ? labels[i] : i;
  const float d = sim[(int64_t)i * qs + lab * cs];
  int g = 0, e = 0;
  for (int j = lane; j < cnt; j += 64) {
    const float x = sim[(int64_t)i * qs + (int64_t)j * cs];
    g += x > d; e += x == d;
  }
  g = (int)wave_sum((float)g); e = (int)wave_sum((float)e);
  if (lane == 0) { greater[i] = g; equal[i] = e; }
}

}  // namespace

extern "C" int xp_sim_matrix(const float* a, const float* b, float* sim, int64_t na, int64_t nb, int64_t d, void* stream) {
  XP_REQUIRE(a && b && sim && na > 0 && nb > 0 && d > 0, "xp_sim_matrix: bad arguments");
  dim3 g((unsigned)cdiv(nb, 32), (unsigned)cdiv(na, 32));
  sgemm_strided_kernel<<<g, 256, 0, (hipStream_t)stream>>>(a, d, 1, b, 1, d, sim, nb, (int)na, (int)nb, (int)d, nullptr);
  XP_CHECK_LAUNCH("xp_sim_matrix");
  return XP_OK;
}

extern "C" int xp_dsl_rerank(float* sim, int64_t n, int64_t m, float theta, int32_t multiply, void* stream) {
  XP_REQUIRE(sim && n > 0 && m > 0, "xp_dsl_rerank: bad arguments");
  dsl_rerank_kernel<<<(unsigned)m, 64, 0, (hipStream_t)stream>>>(sim, (int)n, (int)m, theta, multiply);
  XP_CHECK_LAUNCH("xp_dsl_rerank");
  return XP_OK;
}

extern "C" int xp_retrieval_ranks(const float* sim, const int64_t* labels, int64_t n, int64_t m, int32_t transpose,
                                  int32_t* greater, int32_t* equal, void* stream) {
  XP_REQUIRE(sim && greater && equal && n > 0 && m > 0 && n < (1 << 24) && m < (1 << 24), "xp_retrieval_ranks: bad arguments");
  XP_REQUIRE(labels || n == m || (transpose ? m <= n : n <= m), "xp_retrieval_ranks: diagonal labels need a label for every query");
  const int64_t queries = transpose ? m : n;
  retrieval_ranks_kernel<<<(unsigned)queries, 64, 0, (hipStream_t)stream>>>(sim, labels, (int)n, (int)m, transpose, greater, equal);
  XP_CHECK_LAUNCH("xp_retrieval_ranks");
  return XP_OK;
}

extern "C" size_t xp_vsc_fc_loss_workspace_bytes(int64_t n, int64_t d) {
  (void)d;
  return (size_t)(6 * n * n + 8 * n) * sizeof(float);   // S1..S3, G1..G3, 6 lse vectors, part[2n]
}

extern "C" int xp_vsc_fc_loss(const float* vis, const float* txt, const float* img, const float* cap, const float* log_scale,
                              float* loss, float* d_vis, float* d_txt, float* d_img, float* d_cap, float* d_log_scale,
                              int64_t n, int64_t d, void* workspace, size_t workspace_bytes, void* stream) {
  XP_REQUIRE(vis && txt && img && cap && log_scale && loss && d_vis && d_txt && d_img && d_cap && d_log_scale,
             "xp_vsc_fc_loss: null pointer");
  XP_REQUIRE(n > 0 && d > 0 && n <= 16384, "xp_vsc_fc_loss: bad sizes n=%lld d=%lld", (long long)n, (long long)d);
  XP_REQUIRE(workspace && workspace_bytes >= xp_vsc_fc_loss_workspace_bytes(n, d), "xp_vsc_fc_loss: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* S1 = (float*)workspace; float* S2 = S1 + n * n; float* S3 = S2 + n * n;
  float* G1 = S3 + n * n; float* G2 = G1 + n * n; float* G3 = G2 + n * n;
  float* stats = G3 + n * n;
  float* part = stats + 6 * n;
  const int N = (int)n, D = (int)d;
  dim3 gnd((unsigned)cdiv(d, 32), (unsigned)cdiv(n, 32));
  launch_logits(vis, txt, S1, n, d, log_scale, st);
  launch_logits(vis, cap, S2, n, d, log_scale, st);
  launch_logits(img, cap, S3, n, d, log_scale, st);
  XP_CHECK_LAUNCH("xp_vsc_fc_loss(logits)");
  vsc_lse_kernel<<<(unsigned)(2 * n), 64, 0, st>>>(S1, S2, S3, stats, N);
  XP_CHECK_LAUNCH("xp_vsc_fc_loss(lse)");
  vsc_grad_kernel<<<(unsigned)n, 64, 0, st>>>(S1, S2, S3, stats, G1, G2, G3, part, N);
  XP_CHECK_LAUNCH("xp_vsc_fc_loss(grad)");
  finish_kernel<<<1, 64, 0, st>>>(part, loss, d_log_scale, N);
  XP_CHECK_LAUNCH("xp_vsc_fc_loss(finish)");
  // dV = s (G1 T + G2 C) ; dT = s G1^T V ; dC = s (G2^T V + G3^T I) ; dI = s G3 C
  sgemm_strided_kernel<<<gnd, 256, 0, st>>>(G1, n, 1, txt, d, 1, d_vis, d, N, D, N, log_scale, 0);
  sgemm_strided_kernel<<<gnd, 256, 0, st>>>(G2, n, 1, cap, d, 1, d_vis, d, N, D, N, log_scale, 1);
  sgemm_strided_kernel<<<gnd, 256, 0, st>>>(G1, 1, n, vis, d, 1, d_txt, d, N, D, N, log_scale, 0);
  sgemm_strided_kernel<<<gnd, 256, 0, st>>>(G2, 1, n, vis, d, 1, d_cap, d, N, D, N, log_scale, 0);
  sgemm_strided_kernel<<<gnd, 256, 0, st>>>(G3, 1, n, img, d, 1, d_cap, d, N, D, N, log_scale, 1);
  sgemm_strided_kernel<<<gnd, 256, 0, st>>>(G3, n, 1, cap, d, 1, d_img, d, N, D, N, log_scale, 0);
  XP_CHECK_LAUNCH("xp_vsc_fc_loss(grads)");
  return XP_OK;
}

extern "C" size_t xp_nce_loss_workspace_bytes(int64_t n, int64_t d) {
  (void)d;
  return (size_t)(2 * n * n + 4 * n) * sizeof(float);   // A, G, lse_r, lse_c, part[2n]
}

extern "C" int xp_nce_loss(const float* vis, const float* txt, const float* log_scale, float* loss,
                           float* d_vis, float* d_txt, float* d_log_scale, int64_t n, int64_t d,
                           void* workspace, size_t workspace_bytes, void* stream) {
  XP_REQUIRE(vis && txt && log_scale && loss && d_vis && d_txt && d_log_scale, "xp_nce_loss: null pointer");
  XP_REQUIRE(n > 0 && d > 0 && n <= 16384, "xp_nce_loss: bad sizes n=%lld d=%lld", (long long)n, (long long)d);
  XP_REQUIRE(workspace && workspace_bytes >= xp_nce_loss_workspace_bytes(n, d), "xp_nce_loss: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* A = (float*)workspace;
  float* G = A + n * n;
  float* lse_r = G + n * n;
  float* lse_c = lse_r + n;
  float* part = lse_c + n;
  const int N = (int)n, D = (int)d;
  dim3 gnd((unsigned)cdiv(d, 32), (unsigned)cdiv(n, 32));
  // A[i][j] = e^ls sum_k V[i][k] T[j][k]
  launch_logits(vis, txt, A, n, d, log_scale, st);
  XP_CHECK_LAUNCH("xp_nce_loss(logits)");
  lse_kernel<<<(unsigned)(2 * n), 64, 0, st>>>(A, lse_r, lse_c, N);
  XP_CHECK_LAUNCH("xp_nce_loss(lse)");
  grad_kernel<<<(unsigned)n, 64, 0, st>>>(A, lse_r, lse_c, G, part, N);
  XP_CHECK_LAUNCH("xp_nce_loss(grad)");
  finish_kernel<<<1, 64, 0, st>>>(part, loss, d_log_scale, N);
  XP_CHECK_LAUNCH("xp_nce_loss(finish)");
  // dV[i][k] = e^ls sum_j G[i][j] T[j][k] ;  dT[j][k] = e^ls sum_i G[i][j] V[i][k]
  sgemm_strided_kernel<<<gnd, 256, 0, st>>>(G, n, 1, txt, d, 1, d_vis, d, N, D, N, log_scale);
  XP_CHECK_LAUNCH("xp_nce_loss(dV)");
  sgemm_strided_kernel<<<gnd, 256, 0, st>>>(G, 1, n, vis, d, 1, d_txt, d, N, D, N, log_scale);
  XP_CHECK_LAUNCH("xp_nce_loss(dT)");
  return XP_OK;
}
