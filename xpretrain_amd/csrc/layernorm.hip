// LayerNorm forward / backward for gfx950 -- HBM-bound row kernels.
//
// One wave64 per row: the row lives in registers (4 elements per lane per 256-column slab, 8/16-byte
// loads, 512 B contiguous per wave instruction), statistics by DPP/shuffle wave reductions in fp32,
// two-pass variance.  Backward fuses the residual-gradient add (dx = dres + LN'(dy)) and accumulates
// the per-column dgamma/dbeta in registers across the rows a wave owns; one partial row per block is
// reduced by xp_splitk_reduce's kernel.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int MAXJ = 4;          // cols <= 1024
constexpr int WAVES = 4;

// fp32 side rows of the residual stream (XpGemmDesc::resid_side): row r is a side row iff r % S < M; its side index is
// (r / S) * stride + r % S.  xin: the row is READ from there (fp32) instead of x; yout: the fp32 result is ALSO written there
// (the LayerNorm whose output is itself the residual stream: pre_layrnorm).
struct LnSide { const float* xin; float* yout; unsigned S, M, stride; };

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y, int64_t ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int cols, float eps, LnSide side) {
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
    int64_t srow = -1;                               // wave-uniform: one row per wave
    if (side.xin || side.yout) {                     // 32-bit arithmetic (rows < 2^31, checked by the launcher)
      const unsigned r = (unsigned)row, q = r / side.S, rem = r - q * side.S;
      if (rem < side.M) srow = (int64_t)q * side.stride + rem;
    }
    const float* xs = (srow >= 0 && side.xin) ? side.xin + srow * cols : nullptr;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = j * 256 + lane * 4;
      if (j < nj && c < cols) { v[j] = xs ? load4(xs + c) : load4(x + row * ldx + c); s += v[j][0] + v[j][1] + v[j][2] + v[j][3]; }
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
        if (srow >= 0 && side.yout) store4(side.yout + srow * cols + c, o);
      }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// bf16 rows with 16-byte accesses: HALF a wave per row (32 lanes x 8 elements per 256-column slab), two adjacent rows per wave
// in flight, the two statistics of both rows reduced by one 5-step butterfly each (the halves never mix).  The wave-per-row
// kernel above moves 8 bytes per lane and access; at 18848 x 768 it reached 3.9 TB/s.
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int NJ>
__global__ __launch_bounds__(256) void ln_fwd_h_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16_t* __restrict__ y, int64_t ldy,
                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                       int64_t rows, int cols, float eps, LnSide side) {
  // gamma / beta live in LDS (48 registers otherwise at 768 columns: the row data of two rows in flight is what the register
  // file should hold)
  __shared__ __attribute__((aligned(16))) float sgb[2][NJ * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31, half = lane >> 5;
  for (int c = threadIdx.x * 4; c < NJ * 256; c += 256 * 4) {
    const bool in = c < cols;
    store4(&sgb[0][c], in ? load4(gamma + c) : f32x4{0, 0, 0, 0});
    store4(&sgb[1][c], in ? load4(beta + c) : f32x4{0, 0, 0, 0});
  }
  __syncthreads();
  const float inv = 1.0f / (float)cols;
  for (int64_t row0 = ((int64_t)blockIdx.x * WAVES + wave) * 2; row0 < rows; row0 += (int64_t)gridDim.x * WAVES * 2) {
    const int64_t row = row0 + half;
    const bool ok = row < rows;
    int64_t srow = -1;                               // uniform per half-wave
    if ((side.xin || side.yout) && ok) {             // 32-bit arithmetic (rows < 2^31, checked by the launcher)
      const unsigned r = (unsigned)row, q = r / side.S, rem = r - q * side.S;
      if (rem < side.M) srow = (int64_t)q * side.stride + rem;
    }
    const float* xs = (srow >= 0 && side.xin) ? side.xin + srow * cols : nullptr;
    f32x8 v[NJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = j * 256 + hl * 8;
      if (ok && c < cols) {
        v[j] = xs ? load8(xs + c) : load8(x + row * ldx + c);
        s += (v[j].lo[0] + v[j].lo[1]) + (v[j].lo[2] + v[j].lo[3]) + (v[j].hi[0] + v[j].hi[1]) + (v[j].hi[2] + v[j].hi[3]);
      } else v[j] = f32x8{f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    }
    const float mu = half_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = j * 256 + hl * 8;
      if (c < cols) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d0 = v[j].lo[e] - mu, d1 = v[j].hi[e] - mu; q += d0 * d0 + d1 * d1; }
      }
    }
    const float rs = rsqrtf(half_sum(q) * inv + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = j * 256 + hl * 8;
      if (ok && c < cols) {
        const f32x8 gm = load8(&sgb[0][c]), bt = load8(&sgb[1][c]);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o.lo[e] = (v[j].lo[e] - mu) * rs * gm.lo[e] + bt.lo[e];
          o.hi[e] = (v[j].hi[e] - mu) * rs * gm.hi[e] + bt.hi[e];
        }
        store8(y + row * ldy + c, o);
        if (srow >= 0 && side.yout) store8(side.yout + srow * cols + c, o);
      }
    }
    if (hl == 0 && ok) { mean[row] = mu; rstd[row] = rs; }
  }
}

template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef f32x4 type; };
template <> struct Raw4<bf16_t> { typedef bf16x4 type; };
__device__ __forceinline__ f32x4 cvt4(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 cvt4(bf16x4 v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }

// DXS = 1: also accumulate the column sums of the dx this kernel writes (the bias gradient of the Linear whose output feeds
// the residual stream here); the partial rows are then [dgamma | dbeta | dxsum].  DXS = 2: additionally the column sums of the
// residual gradient `dres` it reads anyway (in an encoder layer's second LayerNorm that is dx3: the bias gradient of fc2, which
// otherwise costs a separate pass over dx3): [dgamma | dbeta | dxsum | dressum].
template <typename T, int NJ, int DXS>
__global__ __launch_bounds__(256) XP_NO_PK_F32 void ln_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* dres, int64_t lddres,
                                                     T* dx, int64_t lddx, float* __restrict__ part,
                                                     int64_t rows, int cols, LnSide side) {
  constexpr int NV = 2 + DXS;
  __shared__ float red[WAVES][NV][NJ * 256];
  typedef typename Raw4<T>::type raw_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gm[NJ], dg[NJ], db[NJ], dxs[DXS ? NJ : 1], drs[DXS == 2 ? NJ : 1];
  if constexpr (DXS != 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) dxs[j] = f32x4{0, 0, 0, 0};
  }
  if constexpr (DXS == 2) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) drs[j] = f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = j * 256 + lane * 4;
    gm[j] = c < cols ? load4(gamma + c) : f32x4{0, 0, 0, 0};
    dg[j] = f32x4{0, 0, 0, 0}; db[j] = f32x4{0, 0, 0, 0};
  }
  const float inv = 1.0f / (float)cols;
  // two rows per iteration: all loads of both rows are issued before the first reduction (memory-level parallelism);
  // rows stay packed in their storage type until used
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const raw_t zero = __builtin_bit_cast(raw_t, typename Raw4<T>::type{});
  for (int64_t row0 = (int64_t)blockIdx.x * WAVES + wave; row0 < rows; row0 += 2 * stride) {
    const int64_t rws[2] = {row0, row0 + stride};
    raw_t xr[2][NJ], dr[2][NJ], rr[2][NJ];
    float mu[2], rs[2];
    const float* xs[2] = {nullptr, nullptr};          // fp32 side row of x (wave-uniform): the forward normalised THAT row
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool ok = rws[u] < rows;
      mu[u] = ok ? mean[rws[u]] : 0.f; rs[u] = ok ? rstd[rws[u]] : 0.f;
      if (side.xin && ok) {
        const unsigned r = (unsigned)__builtin_amdgcn_readfirstlane((int)rws[u]), q = r / side.S, rem = r - q * side.S;
        if (rem < side.M) xs[u] = side.xin + ((int64_t)q * side.stride + rem) * cols;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = j * 256 + lane * 4;
        if (ok && c < cols) {
          xr[u][j] = *reinterpret_cast<const raw_t*>(x + rws[u] * ldx + c);
          dr[u][j] = *reinterpret_cast<const raw_t*>(dy + rws[u] * lddy + c);
          rr[u][j] = dres ? *reinterpret_cast<const raw_t*>(dres + rws[u] * lddres + c) : zero;
        } else { xr[u][j] = zero; dr[u][j] = zero; rr[u][j] = zero; }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (rws[u] >= rows) continue;
      float c1 = 0.f, c2 = 0.f;
      f32x4 xh[NJ], gy[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int cx = j * 256 + lane * 4;
        const f32x4 xv = (xs[u] && cx < cols) ? load4(xs[u] + cx) : cvt4(xr[u][j]), dv = cvt4(dr[u][j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[j][e] = (xv[e] - mu[u]) * rs[u];
          gy[j][e] = dv[e] * gm[j][e];
          c1 += gy[j][e]; c2 += gy[j][e] * xh[j][e];
          dg[j][e] += dv[e] * xh[j][e]; db[j][e] += dv[e];
        }
      }
      // columns >= cols hold x = 0 -> xh = -mu*rs there, but gamma (hence gy) is 0 and dy is 0: they add nothing
      c1 = wave_sum(c1) * inv; c2 = wave_sum(c2) * inv;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < cols) {
          const f32x4 rv = cvt4(rr[u][j]);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs[u] * (gy[j][e] - c1 - xh[j][e] * c2) + rv[e];
          store4(dx + rws[u] * lddx + c, o);
          if constexpr (DXS != 0) dxs[j] += o;
          if constexpr (DXS == 2) drs[j] += rv;
        }
      }
    }
  }
  // block-level reduce of dgamma / dbeta partials, one partial row pair per block
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < cols) {
      store4(&red[wave][0][c], dg[j]); store4(&red[wave][1][c], db[j]);
      if constexpr (DXS != 0) store4(&red[wave][2][c], dxs[j]);
      if constexpr (DXS == 2) store4(&red[wave][3][c], drs[j]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) a += red[w][q][c];
      part[((int64_t)blockIdx.x * NV + q) * cols + c] = a;
    }
  }
}

// Sum `nsum` consecutive partial rows (each 2*cols wide: dgamma | dbeta) per blockIdx.y; 64 columns per block,
// 4 waves take every 4th row (coalesced 256-byte reads, 4 loads in flight), LDS combine.  Called twice
// (1024 -> 32 -> 1 rows) so no thread walks more than 8 rows and the sum order is fixed (deterministic).
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ in, float* __restrict__ out, int nrows,
                                                              int nsum, int width, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * nsum, r1 = r0 + nsum < nrows ? r0 + nsum : nrows;
  float s0 = 0.f, s1 = 0.f;
  if (c < width) {
    int r = r0 + w;
    for (; r + 4 < r1; r += 8) { s0 += in[(int64_t)r * width + c]; s1 += in[(int64_t)(r + 4) * width + c]; }
    if (r < r1) s0 += in[(int64_t)r * width + c];
  }
  red[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && c < width) {
    const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float* o = out + (int64_t)blockIdx.y * width + c;
    *o = accumulate ? *o + t : t;
  }
}

__global__ __launch_bounds__(256) void ln_param_reduce2_kernel(const float* __restrict__ in, float* dgamma, float* dbeta,
                                                               int nrows, int cols, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, width = 2 * cols;
  float s = 0.f;
  if (c < width)
    for (int r = w; r < nrows; r += 4) s += in[(int64_t)r * width + c];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < width) {
    const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float* o = c < cols ? dgamma + c : dbeta + (c - cols);
    *o = accumulate ? *o + t : t;
  }
}

inline int bwd_blocks(int64_t rows) { int64_t b = cdiv(rows, WAVES * 4); return (int)(b < 1 ? 1 : (b > 512 ? 512 : b)); }

}  // namespace

extern "C" int xp_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                                float* mean, float* rstd, int64_t rows, int64_t cols, float eps, int32_t dtype,
                                void* stream) {
  return xp_layernorm_fwd_side(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, cols, eps, dtype, nullptr, nullptr, 0, 0, 0, stream);
}

extern "C" int xp_layernorm_fwd_side(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                                     float* mean, float* rstd, int64_t rows, int64_t cols, float eps, int32_t dtype,
                                     const float* x_side, float* y_side, int64_t side_S, int32_t side_M, int32_t side_stride,
                                     void* stream) {
  XP_REQUIRE(x && gamma && beta && y && mean && rstd, "xp_layernorm_fwd: null pointer");
  XP_REQUIRE((!x_side && !y_side) || (side_S > 0 && side_M > 0 && side_M <= side_S && side_stride >= side_M && rows < ((int64_t)1 << 31) &&
                                      side_S < ((int64_t)1 << 31)),
             "xp_layernorm_fwd_side: side rows need 0 < side_M <= side_S and side_stride >= side_M");
  const LnSide side{x_side, y_side, (unsigned)(side_S > 0 ? side_S : 1), (unsigned)side_M, (unsigned)side_stride};
  XP_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= MAXJ * 256, "xp_layernorm_fwd: cols=%lld unsupported (need %%4==0, <=%d)",
             (long long)cols, MAXJ * 256);
  XP_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "xp_layernorm_fwd: ld must be a multiple of 4");
  const int blocks = (int)(cdiv(rows, WAVES) < 4096 ? cdiv(rows, WAVES) : 4096);
  hipStream_t st = (hipStream_t)stream;
  const bool half_rows = true;
  const bool wide16 = dtype == XP_BF16 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 &&
                      (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)x_side | (uintptr_t)y_side) & 15) == 0;
  if (half_rows && wide16) {
    const int hb = (int)(cdiv(rows, 2 * WAVES) < 2048 ? cdiv(rows, 2 * WAVES) : 2048);
    const int nj = (int)cdiv(cols, 256);
#define XP_LN_FWD_H(NJ) ln_fwd_h_kernel<NJ><<<hb, 256, 0, st>>>((const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd, rows, (int)cols, eps, side)
    if (nj == 1) XP_LN_FWD_H(1); else if (nj == 2) XP_LN_FWD_H(2); else if (nj == 3) XP_LN_FWD_H(3); else XP_LN_FWD_H(4);
#undef XP_LN_FWD_H
  } else if (dtype == XP_BF16)
    ln_fwd_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd, rows, (int)cols, eps, side);
  else if (dtype == XP_F32)
    ln_fwd_kernel<float><<<blocks, 256, 0, st>>>((const float*)x, ldx, gamma, beta, (float*)y, ldy, mean, rstd, rows, (int)cols, eps, side);
  else XP_REQUIRE(false, "xp_layernorm_fwd: bad dtype %d", dtype);
  XP_CHECK_LAUNCH("xp_layernorm_fwd");
  return XP_OK;
}

extern "C" size_t xp_layernorm_bwd_workspace_bytes(int64_t rows, int64_t cols) {
  return (size_t)(bwd_blocks(rows) + 32) * 4 * cols * sizeof(float);
}

namespace {
int ln_bwd_launch(const char* name, const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* mean,
                  const float* rstd, const void* dres, int64_t lddres, void* dx, int64_t lddx, int64_t rows, int64_t cols,
                  int32_t dtype, int dxs, void* workspace, size_t workspace_bytes, hipStream_t st, const float* x_side = nullptr,
                  int64_t side_S = 0, int32_t side_M = 0, int32_t side_stride = 0) {
  XP_REQUIRE(dy && x && gamma && mean && rstd && dx, "%s: null pointer", name);
  XP_REQUIRE(!x_side || (side_S > 0 && side_M > 0 && side_M <= side_S && side_stride >= side_M && rows < ((int64_t)1 << 31) &&
                         side_S < ((int64_t)1 << 31)),
             "%s: side rows need 0 < side_M <= side_S and side_stride >= side_M", name);
  const LnSide side{x_side, nullptr, (unsigned)(side_S > 0 ? side_S : 1), (unsigned)side_M, (unsigned)side_stride};
  XP_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= MAXJ * 256, "%s: cols=%lld unsupported", name, (long long)cols);
  XP_REQUIRE(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!dres || lddres % 4 == 0), "%s: ld must be a multiple of 4", name);
  XP_REQUIRE(workspace && workspace_bytes >= xp_layernorm_bwd_workspace_bytes(rows, cols), "%s: workspace too small", name);
  XP_REQUIRE(dtype == XP_BF16 || dtype == XP_F32, "%s: bad dtype %d", name, dtype);
  XP_REQUIRE(dxs >= 0 && dxs <= 2 && (dxs != 2 || dres), "%s: with_dx_colsum must be 0, 1 or 2 (2 needs dres)", name);
  const int blocks = bwd_blocks(rows);
  float* part = (float*)workspace;
  const int nj = (int)cdiv(cols, 256);
#define XP_LN_BWD(T, NJ, D)                                                                                           \
  ln_bwd_kernel<T, NJ, D><<<blocks, 256, 0, st>>>((const T*)dy, lddy, (const T*)x, ldx, gamma, mean, rstd, (const T*)dres, \
                                                  lddres, (T*)dx, lddx, part, rows, (int)cols, side)
#define XP_LN_BWD_NJ(T, D)                                                                                            \
  do { if (nj == 1) XP_LN_BWD(T, 1, D); else if (nj == 2) XP_LN_BWD(T, 2, D); else if (nj == 3) XP_LN_BWD(T, 3, D); else XP_LN_BWD(T, 4, D); } while (0)
  // (16-byte accesses -- 8 elements per lane -- were measured in round 6 and are no faster, alone or in the step: the kernel moves
  //  4.8-5.3 TB/s as it is; tools/experiments/ln_bwd_wide.hip, profiles/r06f_*)
  if (dtype == XP_BF16) { if (dxs == 2) XP_LN_BWD_NJ(bf16_t, 2); else if (dxs) XP_LN_BWD_NJ(bf16_t, 1); else XP_LN_BWD_NJ(bf16_t, 0); }
  else                  { if (dxs == 2) XP_LN_BWD_NJ(float, 2);  else if (dxs) XP_LN_BWD_NJ(float, 1);  else XP_LN_BWD_NJ(float, 0); }
#undef XP_LN_BWD_NJ
#undef XP_LN_BWD
  XP_CHECK_LAUNCH(name);
  return XP_OK;
}
}  // namespace

extern "C" int xp_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                                const float* mean, const float* rstd, const void* dres, int64_t lddres,
                                void* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                                int64_t rows, int64_t cols, int32_t dtype,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return xp_layernorm_bwd_side(dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, dgamma, dbeta, accumulate, rows, cols, dtype,
                               nullptr, 0, 0, 0, workspace, workspace_bytes, stream);
}

extern "C" int xp_layernorm_bwd_side(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                                     const float* mean, const float* rstd, const void* dres, int64_t lddres,
                                     void* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                                     int64_t rows, int64_t cols, int32_t dtype,
                                     const float* x_side, int64_t side_S, int32_t side_M, int32_t side_stride,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  XP_REQUIRE(dgamma && dbeta, "xp_layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int rc = ln_bwd_launch("xp_layernorm_bwd", dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, rows, cols, dtype, 0,
                         workspace, workspace_bytes, st, x_side, side_S, side_M, side_stride);
  if (rc) return rc;
  // two-level deterministic reduce of the per-block partial rows: blocks -> <=32 -> 1; dgamma/dbeta may be two
  // separate buffers, so the last level runs once per output
  const int blocks = bwd_blocks(rows);
  float* part = (float*)workspace;
  const int width = 2 * (int)cols, lvl = (int)cdiv(blocks, 32);
  float* part2 = part + (int64_t)blocks * width;
  ln_param_reduce_kernel<<<dim3((unsigned)cdiv(width, 64), (unsigned)cdiv(blocks, lvl)), 256, 0, st>>>(part, part2, blocks, lvl, width, 0);
  XP_CHECK_LAUNCH("xp_layernorm_bwd(reduce1)");
  const int n2 = (int)cdiv(blocks, lvl);
  // level 2 on the gamma half and the beta half (row stride = width)
  ln_param_reduce2_kernel<<<(unsigned)cdiv(width, 64), 256, 0, st>>>(part2, dgamma, dbeta, n2, (int)cols, accumulate);
  XP_CHECK_LAUNCH("xp_layernorm_bwd(reduce2)");
  return XP_OK;
}

extern "C" int64_t xp_layernorm_bwd_partial_rows(int64_t rows) { return bwd_blocks(rows); }

extern "C" int xp_layernorm_bwd_partials(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                                         const float* mean, const float* rstd, const void* dres, int64_t lddres,
                                         void* dx, int64_t lddx, int32_t with_dx_colsum, int64_t rows, int64_t cols,
                                         int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
  return ln_bwd_launch("xp_layernorm_bwd_partials", dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, rows, cols, dtype,
                       with_dx_colsum, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int xp_layernorm_bwd_partials_side(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                                              const float* mean, const float* rstd, const void* dres, int64_t lddres,
                                              void* dx, int64_t lddx, int32_t with_dx_colsum, int64_t rows, int64_t cols,
                                              int32_t dtype, const float* x_side, int64_t side_S, int32_t side_M, int32_t side_stride,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  return ln_bwd_launch("xp_layernorm_bwd_partials_side", dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, rows, cols, dtype,
                       with_dx_colsum, workspace, workspace_bytes, (hipStream_t)stream, x_side, side_S, side_M, side_stride);
}
