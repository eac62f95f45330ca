// Embedding / pooling / normalisation glue kernels of the CLIP-ViP path (all HBM-bound, vectorised
// 8-16 B per lane, fp32 math): im2col for the patch conv-as-GEMM, video-proxy token rows and the
// embedding-table gradients, CLIP text embeddings, EOT-argmax pooling gather/scatter, L2 normalise,
// fp32 <-> compute-dtype casts.  Reference: modeling/CLIP_ViP.py:168-197, 210-227, 776, 1148-1149.
#include "common.h"

namespace {

constexpr int TPB = 256;
inline int grid_for(int64_t work) { int64_t b = cdiv(work, TPB); return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

// ---- im2col: frames fp32 [BT,3,H,W] -> [BT*gh*gw, 3*P*P], k = (c,py,px); 8 consecutive px per thread ----
template <typename T>
__global__ void im2col_kernel(const float* __restrict__ v, T* __restrict__ out, int64_t BT, int H, int W, int P) {
  const int gh = H / P, gw = W / P, K = 3 * P * P, KC = K / 8;
  const int64_t total = BT * gh * gw * KC;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / KC;
    const int k = (int)(idx % KC) * 8;
    const int c = k / (P * P), py = (k % (P * P)) / P, px = k % P;
    const int64_t bt = m / (gh * gw);
    const int gy = (int)((m % (gh * gw)) / gw), gx = (int)(m % gw);
    const float* src = v + ((bt * 3 + c) * H + gy * P + py) * (int64_t)W + gx * P + px;
    const f32x4 a = load4(src), b = load4(src + 4);
    T* dst = out + m * K + k;
    store4(dst, a); store4(dst + 4, b);
  }
}

// ---- im2col straight from decoded uint8 frames [BT,3,H,W]: (x / 255 - mean[c]) / std[c] (the collate / ImageNorm
// arithmetic, datasets/dataloader.py:209-233, data_utils.py:256-281) fused with the cast and the patch gather: the
// frame read drops from 4 B to 1 B per pixel.  8 consecutive px (8 bytes) per thread.
struct Norm3 { float mean[3], std[3]; };
template <typename T>
__global__ void im2col_u8_kernel(const unsigned char* __restrict__ v, T* __restrict__ out, int64_t BT, int H, int W, int P, Norm3 nm) {
  const int gh = H / P, gw = W / P, K = 3 * P * P, KC = K / 8;
  const int64_t total = BT * gh * gw * KC;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / KC;
    const int k = (int)(idx % KC) * 8;
    const int c = k / (P * P), py = (k % (P * P)) / P, px = k % P;
    const int64_t bt = m / (gh * gw);
    const int gy = (int)((m % (gh * gw)) / gw), gx = (int)(m % gw);
    const u32x2 raw = *reinterpret_cast<const u32x2*>(v + ((bt * 3 + c) * H + gy * P + py) * (int64_t)W + gx * P + px);
    const float mean = nm.mean[c], sd = nm.std[c];
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // same operation order as the reference: divide by 255, subtract the mean, divide by the std
      a[e] = ((float)((raw[0] >> (8 * e)) & 0xFF) / 255.0f - mean) / sd;
      b[e] = ((float)((raw[1] >> (8 * e)) & 0xFF) / 255.0f - mean) / sd;
    }
    T* dst = out + m * K + k;
    store4(dst, a); store4(dst + 4, b);
  }
}

// ---- proxy rows: x[b,0] = class + pos[0]; x[b,1+i] = added[i] + pos[0] ----------------------------------
template <typename T>
__global__ void vip_proxy_rows_kernel(const float* __restrict__ cls, const float* __restrict__ added,
                                      const float* __restrict__ pos, T* __restrict__ x, int64_t B, int64_t S, int M, int D) {
  const int D4 = D / 4;
  const int64_t total = B * M * D4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx % D4) * 4;
    const int m = (int)((idx / D4) % M);
    const int64_t b = idx / ((int64_t)D4 * M);
    const f32x4 e = m == 0 ? load4(cls + d) : load4(added + (int64_t)(m - 1) * D + d);
    store4(x + (b * S + m) * D + d, e + load4(pos + d));
  }
}

// ---- gradients of the vision embedding tables from dx[B,S,D] --------------------------------------------
// blocks 0..L-1   : d_pos[1+l] = sum_{b,t} dx[b, M+t*L+l]
// blocks L..L+M-1 : proxy m: d_class / d_added[m-1] = sum_b dx[b,m]
// block  L+M      : d_pos[0] = sum_b sum_{m<M} dx[b,m]
template <typename T>
__global__ __launch_bounds__(256) void vip_embed_bwd_pos_kernel(const T* __restrict__ dx, float* d_class, float* d_added,
                                                                float* d_pos, int64_t B, int M, int Tn, int L, int D, int acc) {
  const int d = threadIdx.x * 4;
  if (d >= D) return;
  const int64_t S = M + (int64_t)Tn * L;
  const int blk = blockIdx.x;
  f32x4 s = {0, 0, 0, 0};
  float* out;
  if (blk < L) {
    // B * Tn rows, summed in (b, t) order; eight loads in flight per lane (the additions keep their order: same bits as one by one)
    const int64_t n = B * Tn;
    int64_t q = 0;
    for (; q + 8 <= n; q += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t b = (q + u) / Tn, t = (q + u) - b * Tn;
        v[u] = load4(dx + (b * S + M + t * L + blk) * D + d);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; q < n; ++q) {
      const int64_t b = q / Tn, t = q - b * Tn;
      s += load4(dx + (b * S + M + t * L + blk) * D + d);
    }
    out = d_pos + (int64_t)(1 + blk) * D + d;
  } else if (blk < L + M) {
    const int m = blk - L;
    for (int64_t b = 0; b < B; ++b) s += load4(dx + (b * S + m) * D + d);
    out = (m == 0 ? d_class : d_added + (int64_t)(m - 1) * D) + d;
  } else {
    for (int64_t b = 0; b < B; ++b)
      for (int m = 0; m < M; ++m) s += load4(dx + (b * S + m) * D + d);
    out = d_pos + d;
  }
  if (acc) s += load4(out);
  store4(out, s);
}
// part[b][t][D] = sum_l dx[b, M+t*L+l]
template <typename T>
__global__ __launch_bounds__(256) void vip_embed_bwd_time_kernel(const T* __restrict__ dx, float* __restrict__ part,
                                                                 int M, int Tn, int L, int D) {
  const int d = threadIdx.x * 4;
  if (d >= D) return;
  const int64_t S = M + (int64_t)Tn * L;
  const int64_t b = blockIdx.x / Tn;
  const int t = blockIdx.x % Tn;
  f32x4 s = {0, 0, 0, 0};
  const T* p = dx + (b * S + M + (int64_t)t * L) * D + d;
  int l = 0;
  for (; l + 8 <= L; l += 8) {           // eight loads in flight per lane; the additions keep their order
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = load4(p + (int64_t)(l + u) * D);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; l < L; ++l) s += load4(p + (int64_t)l * D);
  store4(part + (int64_t)blockIdx.x * D + d, s);
}

// ---- text embeddings ----------------------------------------------------------------------------------
template <typename T>
__global__ void text_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                      const float* __restrict__ pos, T* __restrict__ x, int64_t rows, int Lt, int D) {
  const int D4 = D / 4;
  const int64_t total = rows * D4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx % D4) * 4;
    const int64_t r = idx / D4;
    const int t = (int)(r % Lt);
    store4(x + r * D + d, load4(tok + ids[r] * D + d) + load4(pos + (int64_t)t * D + d));
  }
}
// Token-table gradient without atomics: one workgroup per token occurrence r; only the FIRST occurrence of a token id
// works, and adds the dx rows of all its occurrences in row order (fixed order -> bit-reproducible; the EOT id that pads
// every caption has ~100 occurrences).  Occurrences are found 256 rows at a time (one compare per thread, ballot per
// wave), then every thread walks the set bits of the four masks in order.
template <typename T>
__global__ __launch_bounds__(256) void text_embed_bwd_tok_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dx,
                                                                 float* d_tok, int64_t rows, int D, int acc) {
  __shared__ unsigned long long masks[4];
  const int64_t r = blockIdx.x, id = ids[r];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int earlier = 0;
  for (int64_t q = threadIdx.x; q < r; q += 256) earlier |= ids[q] == id;
  if (__syncthreads_or(earlier)) return;                       // block-uniform
  const int D4 = D / 4;
  for (int d40 = 0; d40 < D4; d40 += 256) {
    const int d = (d40 + threadIdx.x) * 4;
    const bool ok = d < D;
    f32x4 s = ok ? load4(dx + r * D + d) : f32x4{0, 0, 0, 0};
    for (int64_t base = r + 1; base < rows; base += 256) {
      const int64_t q = base + threadIdx.x;
      const unsigned long long m = __ballot(q < rows && ids[q] == id);
      __syncthreads();                                          // previous chunk's masks are consumed
      if (lane == 0) masks[wave] = m;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        unsigned long long mm = masks[w];
        while (mm) {
          const int j = __builtin_ctzll(mm);
          mm &= mm - 1;
          if (ok) s += load4(dx + (base + w * 64 + j) * D + d);
        }
      }
    }
    if (ok) {
      float* o = d_tok + id * D + d;
      if (acc) s += load4(o);
      store4(o, s);
    }
  }
}
template <typename T>
__global__ void text_embed_bwd_pos_kernel(const T* __restrict__ dx, float* d_pos, int64_t B, int Lt, int D, int acc) {
  const int D4 = D / 4;
  const int64_t total = (int64_t)Lt * D4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx % D4) * 4;
    const int t = (int)(idx / D4);
    f32x4 s = {0, 0, 0, 0};
    for (int64_t b = 0; b < B; ++b) s += load4(dx + (b * Lt + t) * D + d);
    float* o = d_pos + (int64_t)t * D + d;
    if (acc) s += load4(o);
    store4(o, s);
  }
}

// ---- pooling ------------------------------------------------------------------------------------------
__global__ void argmax_rows_kernel(const int64_t* __restrict__ ids, int64_t* __restrict__ idx, int64_t B, int Lt) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t best = ids[b * Lt], bi = 0;
  for (int t = 1; t < Lt; ++t) { const int64_t v = ids[b * Lt + t]; if (v > best) { best = v; bi = t; } }   // first max
  idx[b] = bi;
}
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx, T* __restrict__ out,
                                   int64_t B, int64_t S, int D) {
  const int D4 = D / 4;
  const int64_t total = B * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4) * 4;
    const int64_t b = i / D4;
    const int64_t s = idx ? idx[b] : 0;
    store4(out + b * D + d, load4(x + (b * S + s) * D + d));
  }
}
template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ dout, const int64_t* __restrict__ idx, T* __restrict__ dx,
                                    int64_t B, int64_t S, int D) {
  const int D4 = D / 4;
  const int64_t total = B * S * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4) * 4;
    const int64_t r = i / D4, b = r / S, s = r % S;
    const int64_t hit = idx ? idx[b] : 0;
    f32x4 v = {0, 0, 0, 0};
    if (s == hit) v = load4(dout + b * D + d);
    store4(dx + r * D + d, v);
  }
}

// ---- L2 normalise (one wave per row) --------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ inv, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane * 4; c < cols; c += 256) { const f32x4 v = load4(x + row * cols + c); s += v[0]*v[0] + v[1]*v[1] + v[2]*v[2] + v[3]*v[3]; }
  const float r = 1.0f / sqrtf(wave_sum(s));
  for (int c = lane * 4; c < cols; c += 256) store4(y + row * cols + c, load4(x + row * cols + c) * r);
  if (lane == 0) inv[row] = r;
}
// dx = inv * (dy - y * <y, dy>)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv, T* __restrict__ dx, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 a = load4(dy + row * cols + c), b = load4(y + row * cols + c);
    s += a[0]*b[0] + a[1]*b[1] + a[2]*b[2] + a[3]*b[3];
  }
  s = wave_sum(s);
  const float r = inv[row];
  for (int c = lane * 4; c < cols; c += 256)
    store4(dx + row * cols + c, (load4(dy + row * cols + c) - load4(y + row * cols + c) * s) * r);
}

// ---- casts --------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    store4(dst + i * 4, load4(src + i * 4));
}
template <typename T>
__global__ void cast_back_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n4, int acc) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = load4(src + i * 4);
    if (acc) v += load4(dst + i * 4);
    store4(dst + i * 4, v);
  }
}

}  // namespace

#define DISPATCH(dtype, CALL_BF16, CALL_F32, name)                         \
  if ((dtype) == XP_BF16) { CALL_BF16; }                                   \
  else if ((dtype) == XP_F32) { CALL_F32; }                                \
  else { xp_set_error(name ": bad dtype %d", (int)(dtype)); return XP_ERR_ARG; }

extern "C" int xp_im2col(const float* video, void* patches, int64_t BT, int64_t H, int64_t W, int64_t P, int32_t dtype, void* stream) {
  XP_REQUIRE(video && patches && BT > 0, "xp_im2col: null/empty");
  XP_REQUIRE(P % 8 == 0 && H % P == 0 && W % P == 0, "xp_im2col: need P%%8==0 and H,W multiples of P (H=%lld W=%lld P=%lld)",
             (long long)H, (long long)W, (long long)P);
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(BT * (H / P) * (W / P) * (3 * P * P / 8));
  DISPATCH(dtype, (im2col_kernel<bf16_t><<<g, TPB, 0, st>>>(video, (bf16_t*)patches, BT, (int)H, (int)W, (int)P)),
           (im2col_kernel<float><<<g, TPB, 0, st>>>(video, (float*)patches, BT, (int)H, (int)W, (int)P)), "xp_im2col");
  XP_CHECK_LAUNCH("xp_im2col");
  return XP_OK;
}

extern "C" int xp_im2col_u8(const uint8_t* frames, const float* mean3_host, const float* std3_host, void* patches, int64_t BT,
                            int64_t H, int64_t W, int64_t P, int32_t dtype, void* stream) {
  XP_REQUIRE(frames && patches && mean3_host && std3_host && BT > 0, "xp_im2col_u8: null/empty");
  XP_REQUIRE(P % 8 == 0 && H % P == 0 && W % P == 0, "xp_im2col_u8: need P%%8==0 and H,W multiples of P (H=%lld W=%lld P=%lld)",
             (long long)H, (long long)W, (long long)P);
  XP_REQUIRE(((uintptr_t)frames & 7) == 0, "xp_im2col_u8: frames must be 8-byte aligned");
  Norm3 nm;
  for (int c = 0; c < 3; ++c) {
    XP_REQUIRE(std3_host[c] > 0.f, "xp_im2col_u8: std[%d] must be positive", c);
    nm.mean[c] = mean3_host[c]; nm.std[c] = std3_host[c];
  }
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(BT * (H / P) * (W / P) * (3 * P * P / 8));
  DISPATCH(dtype, (im2col_u8_kernel<bf16_t><<<g, TPB, 0, st>>>(frames, (bf16_t*)patches, BT, (int)H, (int)W, (int)P, nm)),
           (im2col_u8_kernel<float><<<g, TPB, 0, st>>>(frames, (float*)patches, BT, (int)H, (int)W, (int)P, nm)), "xp_im2col_u8");
  XP_CHECK_LAUNCH("xp_im2col_u8");
  return XP_OK;
}

extern "C" int xp_vip_proxy_rows(const float* class_emb, const float* added_cls, const float* pos, void* x,
                                 int64_t B, int64_t S, int64_t M, int64_t D, int32_t dtype, void* stream) {
  XP_REQUIRE(class_emb && pos && x && (M == 1 || added_cls), "xp_vip_proxy_rows: null pointer");
  XP_REQUIRE(B > 0 && M >= 1 && M <= S && D % 4 == 0, "xp_vip_proxy_rows: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(B * M * D / 4);
  DISPATCH(dtype, (vip_proxy_rows_kernel<bf16_t><<<g, TPB, 0, st>>>(class_emb, added_cls, pos, (bf16_t*)x, B, S, (int)M, (int)D)),
           (vip_proxy_rows_kernel<float><<<g, TPB, 0, st>>>(class_emb, added_cls, pos, (float*)x, B, S, (int)M, (int)D)), "xp_vip_proxy_rows");
  XP_CHECK_LAUNCH("xp_vip_proxy_rows");
  return XP_OK;
}

extern "C" size_t xp_vip_embed_bwd_workspace_bytes(int64_t B, int64_t T, int64_t L, int64_t D) {
  (void)L;
  return (size_t)(B * T * D) * sizeof(float);
}

extern "C" int xp_vip_embed_bwd(const void* dx, float* d_class, float* d_added, float* d_pos, float* d_time,
                                int64_t B, int64_t M, int64_t T, int64_t L, int64_t D, int32_t dtype, int32_t accumulate,
                                void* workspace, size_t workspace_bytes, void* stream) {
  XP_REQUIRE(dx && d_class && d_pos && (M == 1 || d_added), "xp_vip_embed_bwd: null pointer");
  XP_REQUIRE(D % 4 == 0 && D <= 1024 && B > 0 && T > 0 && L > 0 && M >= 1, "xp_vip_embed_bwd: bad sizes (D<=1024, D%%4==0)");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           (vip_embed_bwd_pos_kernel<bf16_t><<<(unsigned)(L + M + 1), 256, 0, st>>>((const bf16_t*)dx, d_class, d_added, d_pos, B, (int)M, (int)T, (int)L, (int)D, accumulate)),
           (vip_embed_bwd_pos_kernel<float><<<(unsigned)(L + M + 1), 256, 0, st>>>((const float*)dx, d_class, d_added, d_pos, B, (int)M, (int)T, (int)L, (int)D, accumulate)),
           "xp_vip_embed_bwd");
  XP_CHECK_LAUNCH("xp_vip_embed_bwd(pos)");
  if (d_time) {
    XP_REQUIRE(workspace && workspace_bytes >= xp_vip_embed_bwd_workspace_bytes(B, T, L, D), "xp_vip_embed_bwd: workspace too small");
    float* part = (float*)workspace;
    DISPATCH(dtype,
             (vip_embed_bwd_time_kernel<bf16_t><<<(unsigned)(B * T), 256, 0, st>>>((const bf16_t*)dx, part, (int)M, (int)T, (int)L, (int)D)),
             (vip_embed_bwd_time_kernel<float><<<(unsigned)(B * T), 256, 0, st>>>((const float*)dx, part, (int)M, (int)T, (int)L, (int)D)),
             "xp_vip_embed_bwd");
    XP_CHECK_LAUNCH("xp_vip_embed_bwd(time)");
    return xp_splitk_reduce(part, d_time, T * D, (int32_t)B, accumulate, stream);
  }
  return XP_OK;
}

extern "C" int xp_text_embed_fwd(const int64_t* ids, const float* tok, const float* pos, void* x,
                                 int64_t B, int64_t Lt, int64_t D, int64_t vocab, int32_t dtype, void* stream) {
  XP_REQUIRE(ids && tok && pos && x && B > 0 && Lt > 0 && D % 4 == 0 && vocab > 0, "xp_text_embed_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(B * Lt * D / 4);
  DISPATCH(dtype, (text_embed_fwd_kernel<bf16_t><<<g, TPB, 0, st>>>(ids, tok, pos, (bf16_t*)x, B * Lt, (int)Lt, (int)D)),
           (text_embed_fwd_kernel<float><<<g, TPB, 0, st>>>(ids, tok, pos, (float*)x, B * Lt, (int)Lt, (int)D)), "xp_text_embed_fwd");
  XP_CHECK_LAUNCH("xp_text_embed_fwd");
  return XP_OK;
}

extern "C" int xp_text_embed_bwd(const int64_t* ids, const void* dx, float* d_tok, float* d_pos,
                                 int64_t B, int64_t Lt, int64_t D, int64_t vocab, int32_t dtype, int32_t accumulate, void* stream) {
  XP_REQUIRE(ids && dx && d_tok && d_pos && B > 0 && Lt > 0 && D % 4 == 0 && vocab > 0, "xp_text_embed_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(d_tok, 0, (size_t)vocab * D * sizeof(float), st);
    XP_REQUIRE(e == hipSuccess, "xp_text_embed_bwd: memset failed: %s", hipGetErrorString(e));
  }
  const unsigned g = (unsigned)(B * Lt);            // one workgroup per token occurrence (fixed-order sums, no atomics)
  DISPATCH(dtype, (text_embed_bwd_tok_kernel<bf16_t><<<g, 256, 0, st>>>(ids, (const bf16_t*)dx, d_tok, B * Lt, (int)D, accumulate)),
           (text_embed_bwd_tok_kernel<float><<<g, 256, 0, st>>>(ids, (const float*)dx, d_tok, B * Lt, (int)D, accumulate)), "xp_text_embed_bwd");
  XP_CHECK_LAUNCH("xp_text_embed_bwd(tok)");
  const int g2 = grid_for(Lt * D / 4);
  DISPATCH(dtype, (text_embed_bwd_pos_kernel<bf16_t><<<g2, TPB, 0, st>>>((const bf16_t*)dx, d_pos, B, (int)Lt, (int)D, accumulate)),
           (text_embed_bwd_pos_kernel<float><<<g2, TPB, 0, st>>>((const float*)dx, d_pos, B, (int)Lt, (int)D, accumulate)), "xp_text_embed_bwd");
  XP_CHECK_LAUNCH("xp_text_embed_bwd(pos)");
  return XP_OK;
}

extern "C" int xp_argmax_rows(const int64_t* ids, int64_t* idx, int64_t B, int64_t Lt, void* stream) {
  XP_REQUIRE(ids && idx && B > 0 && Lt > 0, "xp_argmax_rows: bad arguments");
  argmax_rows_kernel<<<(unsigned)cdiv(B, 64), 64, 0, (hipStream_t)stream>>>(ids, idx, B, (int)Lt);
  XP_CHECK_LAUNCH("xp_argmax_rows");
  return XP_OK;
}

extern "C" int xp_gather_rows(const void* x, const int64_t* idx, void* out, int64_t B, int64_t S, int64_t D, int32_t dtype, void* stream) {
  XP_REQUIRE(x && out && B > 0 && S > 0 && D % 4 == 0, "xp_gather_rows: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(B * D / 4);
  DISPATCH(dtype, (gather_rows_kernel<bf16_t><<<g, TPB, 0, st>>>((const bf16_t*)x, idx, (bf16_t*)out, B, S, (int)D)),
           (gather_rows_kernel<float><<<g, TPB, 0, st>>>((const float*)x, idx, (float*)out, B, S, (int)D)), "xp_gather_rows");
  XP_CHECK_LAUNCH("xp_gather_rows");
  return XP_OK;
}

extern "C" int xp_scatter_rows(const void* dout, const int64_t* idx, void* dx, int64_t B, int64_t S, int64_t D, int32_t dtype, void* stream) {
  XP_REQUIRE(dout && dx && B > 0 && S > 0 && D % 4 == 0, "xp_scatter_rows: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(B * S * D / 4);
  DISPATCH(dtype, (scatter_rows_kernel<bf16_t><<<g, TPB, 0, st>>>((const bf16_t*)dout, idx, (bf16_t*)dx, B, S, (int)D)),
           (scatter_rows_kernel<float><<<g, TPB, 0, st>>>((const float*)dout, idx, (float*)dx, B, S, (int)D)), "xp_scatter_rows");
  XP_CHECK_LAUNCH("xp_scatter_rows");
  return XP_OK;
}

extern "C" int xp_l2norm_fwd(const void* x, float* y, float* inv_norm, int64_t rows, int64_t cols, int32_t dtype, void* stream) {
  XP_REQUIRE(x && y && inv_norm && rows > 0 && cols > 0 && cols % 4 == 0, "xp_l2norm_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)cdiv(rows, 4);
  DISPATCH(dtype, (l2norm_fwd_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)x, y, inv_norm, rows, (int)cols)),
           (l2norm_fwd_kernel<float><<<g, 256, 0, st>>>((const float*)x, y, inv_norm, rows, (int)cols)), "xp_l2norm_fwd");
  XP_CHECK_LAUNCH("xp_l2norm_fwd");
  return XP_OK;
}

extern "C" int xp_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, void* dx, int64_t rows, int64_t cols,
                             int32_t dtype, void* stream) {
  XP_REQUIRE(dy && y && inv_norm && dx && rows > 0 && cols > 0 && cols % 4 == 0, "xp_l2norm_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)cdiv(rows, 4);
  DISPATCH(dtype, (l2norm_bwd_kernel<bf16_t><<<g, 256, 0, st>>>(dy, y, inv_norm, (bf16_t*)dx, rows, (int)cols)),
           (l2norm_bwd_kernel<float><<<g, 256, 0, st>>>(dy, y, inv_norm, (float*)dx, rows, (int)cols)), "xp_l2norm_bwd");
  XP_CHECK_LAUNCH("xp_l2norm_bwd");
  return XP_OK;
}

extern "C" int xp_cast(const float* src, void* dst, int64_t n, int32_t dtype, void* stream) {
  XP_REQUIRE(src && dst && n > 0 && n % 4 == 0, "xp_cast: need n %% 4 == 0 (n=%lld)", (long long)n);
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(n / 4);
  DISPATCH(dtype, (cast_kernel<bf16_t><<<g, TPB, 0, st>>>(src, (bf16_t*)dst, n / 4)),
           (cast_kernel<float><<<g, TPB, 0, st>>>(src, (float*)dst, n / 4)), "xp_cast");
  XP_CHECK_LAUNCH("xp_cast");
  return XP_OK;
}

extern "C" int xp_cast_back(const void* src, float* dst, int64_t n, int32_t dtype, int32_t accumulate, void* stream) {
  XP_REQUIRE(src && dst && n > 0 && n % 4 == 0, "xp_cast_back: need n %% 4 == 0");
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(n / 4);
  DISPATCH(dtype, (cast_back_kernel<bf16_t><<<g, TPB, 0, st>>>((const bf16_t*)src, dst, n / 4, accumulate)),
           (cast_back_kernel<float><<<g, TPB, 0, st>>>((const float*)src, dst, n / 4, accumulate)), "xp_cast_back");
  XP_CHECK_LAUNCH("xp_cast_back");
  return XP_OK;
}
