// Shared between the GEMM kernel families (gemm.hip: 128x128 tiles; gemm256.hip: 256x256 deep-pipelined tiles).
#pragma once
#include "common.h"

namespace xpgemm {

struct Remap {
  int64_t grp, stride, off;
  __device__ __forceinline__ int64_t operator()(int64_t r) const {
    return grp == 0 ? r : (r / grp) * stride + off + (r % grp);
  }
};

struct KParams {
  const void* A; const void* B; void* C;
  int64_t M, N, K, lda, ldb, ldc;
  Remap amap, cmap;
  int epilogue, out_f32;
  int64_t k_per_split;
  const float* bias; float scale; int64_t scale_cols;
  const void* resid; int64_t ldr;
  void* aux; int64_t ldaux;
  const float* tab1; const float* tab2; int64_t tab_L;
  int tiles_m, tiles_n;
  int group_n;               // tile columns per L2 super-tile group (see tile_of)
  int xcd_remap;             // 1: consecutive tile ids -> same XCD (default); 0: hardware round-robin (A/B switch)
  int wide;                  // 1: N, ldc, ldr, ldaux all multiples of 8 -> 8 columns per lane, 16-byte bf16 stores
  unsigned long long* dbg;   // optional cycle-stamp trace buffer (xp_debug_set_gemm_trace), else null
};


// Per-lane epilogue constants: the lane owns columns n .. n+3 of every row it stores.
struct EpiLane {
  f32x4 bias; float colscale;
  __device__ __forceinline__ EpiLane(const KParams& p, int64_t n) {
    const int ep = p.epilogue;
    const bool has_bias = ep == XP_EPI_BIAS || ep == XP_EPI_BIAS_QSCALE || ep == XP_EPI_BIAS_GELU || ep == XP_EPI_BIAS_RESID;
    bias = has_bias ? load4(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    colscale = ep == XP_EPI_SCALE ? p.scale : ((ep == XP_EPI_BIAS_QSCALE && n < p.scale_cols) ? p.scale : 1.0f);
  }
};

// Finish and store 4 consecutive outputs of row m (all math fp32; coalesced across the lanes of a row).
template <typename T>
__device__ __forceinline__ void epi_row(const KParams& p, const EpiLane& el, f32x4 v, int64_t m, int64_t n, float* Cf, T* Ct) {
  const int ep = p.epilogue;
  v = (v + el.bias) * el.colscale;
  const int64_t crow = p.cmap(m);
  if (ep == XP_EPI_BIAS_GELU) {
    if (p.out_f32) store4(reinterpret_cast<float*>(p.aux) + crow * p.ldaux + n, v);
    else           store4(reinterpret_cast<T*>(p.aux) + crow * p.ldaux + n, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
  } else if (ep == XP_EPI_BIAS_RESID) {
    v += load4(reinterpret_cast<const T*>(p.resid) + crow * p.ldr + n);
  } else if (ep == XP_EPI_GELU_BWD) {
    const f32x4 pre = load4(reinterpret_cast<const T*>(p.resid) + crow * p.ldr + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= quick_gelu_grad_f(pre[e]);
  } else if (ep == XP_EPI_PATCH) {
    const int64_t w = p.cmap.grp ? (m % p.cmap.grp) : m;
    v += load4(p.tab1 + (w / p.tab_L) * p.N + n);
    v += load4(p.tab2 + (w % p.tab_L) * p.N + n);
  }
  if (p.out_f32) store4(Cf + crow * p.ldc + n, v);
  else           store4(Ct + crow * p.ldc + n, v);
}

// 8-column variant: half as many (16-byte) store instructions -- the epilogue is store-ISSUE bound.
struct EpiLane8 {
  f32x8 bias; float cs_lo, cs_hi;
  __device__ __forceinline__ EpiLane8(const KParams& p, int64_t n) {
    const EpiLane a(p, n), b(p, n + 4);
    bias = f32x8{a.bias, b.bias}; cs_lo = a.colscale; cs_hi = b.colscale;
  }
};
template <typename T>
__device__ __forceinline__ void epi_row8(const KParams& p, const EpiLane8& el, f32x8 v, int64_t m, int64_t n, float* Cf, T* Ct) {
  const int ep = p.epilogue;
  v.lo = (v.lo + el.bias.lo) * el.cs_lo; v.hi = (v.hi + el.bias.hi) * el.cs_hi;
  const int64_t crow = p.cmap(m);
  if (ep == XP_EPI_BIAS_GELU) {
    if (p.out_f32) store8(reinterpret_cast<float*>(p.aux) + crow * p.ldaux + n, v);
    else           store8(reinterpret_cast<T*>(p.aux) + crow * p.ldaux + n, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v.lo[e] = quick_gelu_f(v.lo[e]); v.hi[e] = quick_gelu_f(v.hi[e]); }
  } else if (ep == XP_EPI_BIAS_RESID) {
    const f32x8 r = load8(reinterpret_cast<const T*>(p.resid) + crow * p.ldr + n);
    v.lo += r.lo; v.hi += r.hi;
  } else if (ep == XP_EPI_GELU_BWD) {
    const f32x8 pre = load8(reinterpret_cast<const T*>(p.resid) + crow * p.ldr + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v.lo[e] *= quick_gelu_grad_f(pre.lo[e]); v.hi[e] *= quick_gelu_grad_f(pre.hi[e]); }
  } else if (ep == XP_EPI_PATCH) {
    const int64_t w = p.cmap.grp ? (m % p.cmap.grp) : m;
    const f32x8 a = load8(p.tab1 + (w / p.tab_L) * p.N + n), b = load8(p.tab2 + (w % p.tab_L) * p.N + n);
    v.lo += a.lo + b.lo; v.hi += a.hi + b.hi;
  }
  if (p.out_f32) store8(Cf + crow * p.ldc + n, v);
  else           store8(Ct + crow * p.ldc + n, v);
}

__device__ __forceinline__ int ks_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// XCD-aware bijective remap of the linear workgroup id (block b runs on XCD b % 8): consecutive tile ids -- which
// share an activation row panel -- land on one XCD / one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg, int on = 1) {
  if (!on) return bid;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Tile order inside an XCD's contiguous id range: column GROUPS of `group_n` tiles, rows fastest within a group's
// column sweep.  The ~64 workgroups resident on an XCD then cover ~8 x 8 tiles: every weight block and every
// activation row panel is shared by ~8 concurrent tiles and the k-sweep working set (~3 MB at K=768) fits the 4 MiB L2.
// With plain column-fastest order the N=3072 problems touch all 4.7 MB of W per sweep and thrash (PMC: 22 % L2 misses,
// 300 MB fetched from the fabric for 34 MB of unique input).
__device__ __forceinline__ void tile_of(int id, int tiles_m, int tiles_n, int group_n, int& tm, int& tn) {
  const int per_group = group_n * tiles_m;
  const int grp = id / per_group, within = id - grp * per_group;
  const int n0 = grp * group_n;
  const int gw = tiles_n - n0 < group_n ? tiles_n - n0 : group_n;
  tm = within / gw;
  tn = n0 + within - tm * gw;
}

}  // namespace xpgemm

// launcher of the 256x256 family (gemm256.hip); returns false if the problem does not fit its preconditions
bool xp_gemm256_try(const XpGemmDesc* d, const xpgemm::KParams& kp_base, hipStream_t st);
