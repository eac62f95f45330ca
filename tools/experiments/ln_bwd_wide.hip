// EXPERIMENT (round 6, not built): LayerNorm backward with 16-byte accesses -- one wave per row, 8 elements per lane, 512-column slabs.
// Parity-green (22 LayerNorm tests + a wide-vs-narrow cross-check), and no faster: at 18848 x 768 the 8-byte kernel already moves
// 4.8 / 5.3 TB/s (LN2 / LN1 form: 24.0 / 21.7 us), this one 24.3 / 22.1 us; in the training step (interleaved whole-step A/B, one
// box, profiles/r06f_in_step_ab_layernorm_backward_16_byte.txt) 14.607 vs 14.545 ms.  The kernel is HBM-bound in isolation and, in
// the step, bound by the CUs the weight-gradient GEMMs of the other stream leave it -- not by bytes per instruction.  A half-wave-per-
// row variant (the forward's layout) was tried first: with four column-sum accumulators a lane owns 24 columns = 96 accumulator
// registers, and hipcc spills at 768 columns.  Drop-in for csrc/layernorm.hip (next to ln_bwd_kernel; launcher: ns = cdiv(cols, 512)).
// bf16 rows with 16-byte accesses (round 6): one wave per row as above, but 8 elements per lane and 512-column slabs (at 768 columns
// the second slab uses half the wave) -- 2 load instructions per operand and row instead of 3, 16 bytes per lane.  Half a wave per
// row (the forward's layout) does not fit here: with the four column-sum accumulators a lane would own 24 columns = 96 accumulator
// registers, and the kernel spills.  Same arithmetic per element and same row-to-wave assignment as ln_bwd_kernel; the two row sums
// (c1, c2) add their columns in another lane order, so results agree to the last bits, not bit for bit
// (tests/test_layernorm_gpu.py compares the two).  In the training step this kernel runs on whatever CUs the
// weight-gradient GEMMs of the other stream leave free (240 registers x 8 waves: nothing shares a CU with them), so bytes per
// instruction and per CU decide its duration there.
__device__ __forceinline__ f32x8 ln_widen(bf16x8 b) {
  return f32x8{f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]}, f32x4{(float)b[4], (float)b[5], (float)b[6], (float)b[7]}};
}
template <int NS, int DXS>
__global__ __launch_bounds__(256) XP_NO_PK_F32 void ln_bwd_w_kernel(const bf16_t* __restrict__ dy, int64_t lddy, const bf16_t* __restrict__ x,
                                                                   int64_t ldx, const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const bf16_t* dres, int64_t lddres, bf16_t* dx, int64_t lddx,
                                                                   float* __restrict__ part, int64_t rows, int cols, LnSide side) {
  constexpr int NV = 2 + DXS;
  __shared__ __attribute__((aligned(16))) float red[WAVES][NV][NS * 512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gml[NS], gmh[NS], dgl[NS], dgh[NS], dbl[NS], dbh[NS], dxl[DXS ? NS : 1], dxh[DXS ? NS : 1], drl[DXS == 2 ? NS : 1], drh[DXS == 2 ? NS : 1];
  const f32x4 z4 = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int c = j * 512 + lane * 8;
    const bool in = c < cols;
    gml[j] = in ? load4(gamma + c) : z4; gmh[j] = in ? load4(gamma + c + 4) : z4;
    dgl[j] = z4; dgh[j] = z4; dbl[j] = z4; dbh[j] = z4;
  }
  if constexpr (DXS != 0) {
#pragma unroll
    for (int j = 0; j < NS; ++j) { dxl[j] = z4; dxh[j] = z4; }
  }
  if constexpr (DXS == 2) {
#pragma unroll
    for (int j = 0; j < NS; ++j) { drl[j] = z4; drh[j] = z4; }
  }
  const float inv = 1.0f / (float)cols;
  // two rows per iteration: all loads of both rows are issued before the first reduction; rows stay packed until used
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const bf16x8 zero = __builtin_bit_cast(bf16x8, u32x4{0, 0, 0, 0});
  for (int64_t row0 = (int64_t)blockIdx.x * WAVES + wave; row0 < rows; row0 += 2 * stride) {
    const int64_t rws[2] = {row0, row0 + stride};
    bf16x8 xr[2][NS], dr[2][NS], rr[2][NS];
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
      for (int j = 0; j < NS; ++j) {
        const int c = j * 512 + lane * 8;
        if (ok && c < cols) {
          xr[u][j] = *reinterpret_cast<const bf16x8*>(x + rws[u] * ldx + c);
          dr[u][j] = *reinterpret_cast<const bf16x8*>(dy + rws[u] * lddy + c);
          rr[u][j] = dres ? *reinterpret_cast<const bf16x8*>(dres + rws[u] * lddres + c) : zero;
        } else { xr[u][j] = zero; dr[u][j] = zero; rr[u][j] = zero; }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (rws[u] >= rows) continue;
      float c1 = 0.f, c2 = 0.f;
      f32x4 xhl[NS], xhh[NS], gyl[NS], gyh[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int cx = j * 512 + lane * 8;
        const f32x8 xv = (xs[u] && cx < cols) ? load8(xs[u] + cx) : ln_widen(xr[u][j]), dv = ln_widen(dr[u][j]);
        // (element order 0..7 of the lane's 8 columns: the same c1 / c2 summation order per lane as two 4-element lanes would not be --
        //  the wave-level sums differ from ln_bwd_kernel in the last bit; everything per column is identical)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xhl[j][e] = (xv.lo[e] - mu[u]) * rs[u];
          gyl[j][e] = dv.lo[e] * gml[j][e];
          c1 += gyl[j][e]; c2 += gyl[j][e] * xhl[j][e];
          dgl[j][e] += dv.lo[e] * xhl[j][e]; dbl[j][e] += dv.lo[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xhh[j][e] = (xv.hi[e] - mu[u]) * rs[u];
          gyh[j][e] = dv.hi[e] * gmh[j][e];
          c1 += gyh[j][e]; c2 += gyh[j][e] * xhh[j][e];
          dgh[j][e] += dv.hi[e] * xhh[j][e]; dbh[j][e] += dv.hi[e];
        }
      }
      // columns >= cols hold x = 0 -> xh = -mu*rs there, but gamma (hence gy) is 0 and dy is 0: they add nothing
      c1 = wave_sum(c1) * inv; c2 = wave_sum(c2) * inv;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int c = j * 512 + lane * 8;
        if (c < cols) {
          const f32x8 rv = ln_widen(rr[u][j]);
          f32x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o.lo[e] = rs[u] * (gyl[j][e] - c1 - xhl[j][e] * c2) + rv.lo[e];
            o.hi[e] = rs[u] * (gyh[j][e] - c1 - xhh[j][e] * c2) + rv.hi[e];
          }
          store8(dx + rws[u] * lddx + c, o);
          if constexpr (DXS != 0) { dxl[j] += o.lo; dxh[j] += o.hi; }
          if constexpr (DXS == 2) { drl[j] += rv.lo; drh[j] += rv.hi; }
        }
      }
    }
  }
  // block-level reduce of the partials, one partial row set per block
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int c = j * 512 + lane * 8;
    if (c < cols) {
      store4(&red[wave][0][c], dgl[j]); store4(&red[wave][0][c + 4], dgh[j]);
      store4(&red[wave][1][c], dbl[j]); store4(&red[wave][1][c + 4], dbh[j]);
      if constexpr (DXS != 0) { store4(&red[wave][2][c], dxl[j]); store4(&red[wave][2][c + 4], dxh[j]); }
      if constexpr (DXS == 2) { store4(&red[wave][3][c], drl[j]); store4(&red[wave][3][c + 4], drh[j]); }
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

