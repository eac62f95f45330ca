// LayerNorm forward / backward for gfx950 -- HBM-bound row kernels.
//
// One wave64 per row: the row lives in registers (4 elements per lane per 256-column slab, 8/16-byte
// loads, 512 B contiguous per wave instruction), statistics by DPP/shuffle wave reductions in fp32,
// two-pass variance.  Backward fuses the residual-gradient add (dx = dres + LN'(dy)) and accumulates
// the per-column dgamma/dbeta in registers across the rows a wave owns; one partial row per block is
// reduced by xp_splitk_reduce's kernel.
#include "common.h"

namespace {

constexpr int MAXJ = 4;          // cols <= 1024
constexpr int WAVES = 4;

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y, int64_t ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int cols, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nj = (cols + 255) >> 8;
  f32x4 gm[MAXJ], bt[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = j * 256 + lane * 4;
    if (j < nj && c < cols) { gm[j] = load4(gamma + c); bt[j] = load4(beta + c); }
    else { gm[j] = f32x4{0, 0, 0, 0}; bt[j] = f32x4{0, 0, 0, 0}; }
  }
  const float inv = 1.0f / (float)cols;
  for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < rows; row += (int64_t)gridDim.x * WAVES) {
    f32x4 v[MAXJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = j * 256 + lane * 4;
      if (j < nj && c < cols) { v[j] = load4(x + row * ldx + c); s += v[j][0] + v[j][1] + v[j][2] + v[j][3]; }
      else v[j] = f32x4{0, 0, 0, 0};
    }
    const float mu = wave_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = j * 256 + lane * 4;
      if (j < nj && c < cols) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) * inv + eps);
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = j * 256 + lane * 4;
      if (j < nj && c < cols) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rs * gm[j][e] + bt[j][e];
        store4(y + row * ldy + c, o);
      }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* dres, int64_t lddres,
                                                     T* dx, int64_t lddx, float* __restrict__ part,
                                                     int64_t rows, int cols) {
  __shared__ float red[WAVES][2][MAXJ * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nj = (cols + 255) >> 8;
  f32x4 gm[MAXJ], dg[MAXJ], db[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = j * 256 + lane * 4;
    gm[j] = (j < nj && c < cols) ? load4(gamma + c) : f32x4{0, 0, 0, 0};
    dg[j] = f32x4{0, 0, 0, 0}; db[j] = f32x4{0, 0, 0, 0};
  }
  const float inv = 1.0f / (float)cols;
  for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < rows; row += (int64_t)gridDim.x * WAVES) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[MAXJ], gy[MAXJ];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = j * 256 + lane * 4;
      if (j < nj && c < cols) {
        const f32x4 xv = load4(x + row * ldx + c), dv = load4(dy + row * lddy + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[j][e] = (xv[e] - mu) * rs;
          gy[j][e] = dv[e] * gm[j][e];
          c1 += gy[j][e]; c2 += gy[j][e] * xh[j][e];
          dg[j][e] += dv[e] * xh[j][e]; db[j][e] += dv[e];
        }
      } else { xh[j] = f32x4{0, 0, 0, 0}; gy[j] = f32x4{0, 0, 0, 0}; }
    }
    c1 = wave_sum(c1) * inv; c2 = wave_sum(c2) * inv;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = j * 256 + lane * 4;
      if (j < nj && c < cols) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gy[j][e] - c1 - xh[j][e] * c2);
        if (dres) o += load4(dres + row * lddres + c);
        store4(dx + row * lddx + c, o);
      }
    }
  }
  // block-level reduce of dgamma / dbeta partials, one partial row pair per block
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = j * 256 + lane * 4;
    if (j < nj && c < cols) { store4(&red[wave][0][c], dg[j]); store4(&red[wave][1][c], db[j]); }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { a += red[w][0][c]; b += red[w][1][c]; }
    part[((int64_t)blockIdx.x * 2 + 0) * cols + c] = a;
    part[((int64_t)blockIdx.x * 2 + 1) * cols + c] = b;
  }
}

// out_g[c] (+)= sum_b part[b][0][c]; out_b[c] (+)= sum_b part[b][1][c]
// 64 columns per block; 4 waves each sum a quarter of the partial rows (coalesced 256-byte reads), LDS combine.
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, float* dgamma, float* dbeta,
                                                              int nblocks, int cols, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;                 // index into the concatenated [2*cols] (gamma | beta)
  float s = 0.f;
  if (c < 2 * cols) {
    const int which = c / cols, col = c % cols;
    for (int b = w; b < nblocks; b += 4) s += part[((int64_t)b * 2 + which) * cols + col];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < 2 * cols) {
    const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    const int which = c / cols, col = c % cols;
    float* out = which ? dbeta : dgamma;
    out[col] = accumulate ? out[col] + t : t;
  }
}

inline int bwd_blocks(int64_t rows) { int64_t b = cdiv(rows, WAVES * 4); return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }

}  // namespace

extern "C" int xp_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                                float* mean, float* rstd, int64_t rows, int64_t cols, float eps, int32_t dtype,
                                void* stream) {
  XP_REQUIRE(x && gamma && beta && y && mean && rstd, "xp_layernorm_fwd: null pointer");
  XP_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= MAXJ * 256, "xp_layernorm_fwd: cols=%lld unsupported (need %%4==0, <=%d)",
             (long long)cols, MAXJ * 256);
  XP_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "xp_layernorm_fwd: ld must be a multiple of 4");
  const int blocks = (int)(cdiv(rows, WAVES) < 4096 ? cdiv(rows, WAVES) : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == XP_BF16)
    ln_fwd_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd, rows, (int)cols, eps);
  else if (dtype == XP_F32)
    ln_fwd_kernel<float><<<blocks, 256, 0, st>>>((const float*)x, ldx, gamma, beta, (float*)y, ldy, mean, rstd, rows, (int)cols, eps);
  else XP_REQUIRE(false, "xp_layernorm_fwd: bad dtype %d", dtype);
  XP_CHECK_LAUNCH("xp_layernorm_fwd");
  return XP_OK;
}

extern "C" size_t xp_layernorm_bwd_workspace_bytes(int64_t rows, int64_t cols) {
  return (size_t)bwd_blocks(rows) * 2 * cols * sizeof(float);
}

extern "C" int xp_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                                const float* mean, const float* rstd, const void* dres, int64_t lddres,
                                void* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                                int64_t rows, int64_t cols, int32_t dtype,
                                void* workspace, size_t workspace_bytes, void* stream) {
  XP_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "xp_layernorm_bwd: null pointer");
  XP_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= MAXJ * 256, "xp_layernorm_bwd: cols=%lld unsupported", (long long)cols);
  XP_REQUIRE(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!dres || lddres % 4 == 0), "xp_layernorm_bwd: ld must be a multiple of 4");
  XP_REQUIRE(workspace && workspace_bytes >= xp_layernorm_bwd_workspace_bytes(rows, cols), "xp_layernorm_bwd: workspace too small");
  const int blocks = bwd_blocks(rows);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == XP_BF16)
    ln_bwd_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, gamma, mean, rstd,
                                                  (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, part, rows, (int)cols);
  else if (dtype == XP_F32)
    ln_bwd_kernel<float><<<blocks, 256, 0, st>>>((const float*)dy, lddy, (const float*)x, ldx, gamma, mean, rstd,
                                                 (const float*)dres, lddres, (float*)dx, lddx, part, rows, (int)cols);
  else XP_REQUIRE(false, "xp_layernorm_bwd: bad dtype %d", dtype);
  XP_CHECK_LAUNCH("xp_layernorm_bwd");
  ln_param_reduce_kernel<<<(unsigned)cdiv(2 * cols, 64), 256, 0, st>>>(part, dgamma, dbeta, blocks, (int)cols, accumulate);
  XP_CHECK_LAUNCH("xp_layernorm_bwd(reduce)");
  return XP_OK;
}
