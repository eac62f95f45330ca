// Shared between the GEMM kernel families (gemm.hip: 128x128 tiles; gemm256.hip: 256x256 deep-pipelined tiles).
#pragma once
#include "common.h"
#include <type_traits>

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
  int fast_epi;              // 1: wide, identity cmap, C / resid / aux each < 4 GiB -> branch-free buffer-addressed epilogue
  float* colsum;             // optional [2 * tiles_m][N] column sums of the finished outputs per wave row block (256 family)
  const float* rside; float* oside; unsigned side_S, side_M;   // BIAS_RESID: fp32 side rows of the residual stream (XpGemmDesc)
  unsigned long long* dbg;   // optional cycle-stamp trace buffer (xp_debug_set_gemm_trace), else null
  int flat_split;            // gemm256s split-K launches: > 0 = the grid is 1-D over (k-chunk, tile), chunk-major per XCD (see kernel)
  // A gathered from a frame tensor by the loader (XpGemmDesc::a_frames): frames [BT,3,H,W] fp32 (im_u8 == 0) or uint8 (im_u8 == 1)
  const void* im_src; int im_u8, im_H, im_W, im_P, im_gw, im_L; float im_mean[3], im_std[3];
};


// fp32 side rows of the residual stream (XpGemmDesc::resid_side / out_side): output row m is a side row iff m % S < M; its side
// index is (m / S) * M + m % S.  The video tower's M proxy tokens of every sample are kept in fp32 beside the bf16 stream: the
// pooled feature is token 0 of the last layer, and rounding ITS residual stream to bf16 at every add is half of the feature error of
// the whole bf16 path (tools/residual_precision_experiment.py) -- 4 rows of 2356.
struct SideRows {
  const float* rin; float* out; unsigned S, M; int64_t N, rows; float inv_S;
  __device__ __forceinline__ SideRows(const KParams& p)
      : rin(p.rside), out(p.oside), S(p.side_S), M(p.side_M), N(p.N), rows(p.M), inv_S(1.0f / (float)p.side_S) {}
  __device__ __forceinline__ bool on() const { return rin != nullptr; }
  // m / S by a float reciprocal + one correction step (m < 2^24: exact in fp32; the launcher checks) -- a handful of instructions
  // per row instead of an integer division in every epilogue pass
  __device__ __forceinline__ bool hit(unsigned m, int64_t& base) const {
    unsigned q = (unsigned)((float)m * inv_S);
    int rem = (int)m - (int)(q * S);
    if (rem < 0) { --q; rem += (int)S; } else if (rem >= (int)S) { ++q; rem -= (int)S; }
    base = ((int64_t)q * M + rem) * N;
    return (unsigned)rem < M && (int64_t)m < rows;
  }
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
    if (!p.aux) {}                                   // forward-only: the pre-activation is not kept
    else if (p.out_f32) store4(reinterpret_cast<float*>(p.aux) + crow * p.ldaux + n, v);
    else                store4(reinterpret_cast<T*>(p.aux) + crow * p.ldaux + n, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
  } else if (ep == XP_EPI_BIAS_RESID) {
    const SideRows sr(p);
    int64_t sb;
    if (sr.on() && sr.hit((unsigned)m, sb)) { v += load4(sr.rin + sb + n); store4(sr.out + sb + n, v); }
    else v += load4(reinterpret_cast<const T*>(p.resid) + crow * p.ldr + n);
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
    if (!p.aux) {}                                   // forward-only: the pre-activation is not kept
    else if (p.out_f32) store8(reinterpret_cast<float*>(p.aux) + crow * p.ldaux + n, v);
    else                store8(reinterpret_cast<T*>(p.aux) + crow * p.ldaux + n, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v.lo[e] = quick_gelu_f(v.lo[e]); v.hi[e] = quick_gelu_f(v.hi[e]); }
  } else if (ep == XP_EPI_BIAS_RESID) {
    const SideRows sr(p);
    int64_t sb;
    if (sr.on() && sr.hit((unsigned)m, sb)) {
      const f32x8 r = load8(sr.rin + sb + n);
      v.lo += r.lo; v.hi += r.hi;
      store8(sr.out + sb + n, v);
    } else {
      const f32x8 r = load8(reinterpret_cast<const T*>(p.resid) + crow * p.ldr + n);
      v.lo += r.lo; v.hi += r.hi;
    }
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

// ---- fast epilogue -----------------------------------------------------------------------------------------------------
// The generic epi_row/epi_row8 above branch on the run-time epilogue kind and on row predicates inside every unrolled
// pass; hipcc's waitcnt insertion then falls back to `s_waitcnt vmcnt(0)` at the block joins, i.e. every pass waits for
// the previous pass's STORES to be acknowledged (measured: 18k cycles for a 256x256 tile, ~1.1k per pass).  The fast
// path is straight-line code per epilogue kind: buffer-addressed accesses whose hardware bounds check drops rows >= M
// (and lanes past N, whose offset is forced out of range), residual / pre-activation loads issued one round ahead of
// the stores that would otherwise sit in front of them in the in-order vmcnt queue.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr unsigned EPI_OOB = 0xFFFFFF00u;     // + 16 (second half of a 32-byte access) must not wrap

template <int EPI> struct EpiTraits {
  static constexpr bool bias = EPI == XP_EPI_BIAS || EPI == XP_EPI_BIAS_QSCALE || EPI == XP_EPI_BIAS_GELU || EPI == XP_EPI_BIAS_RESID;
  static constexpr bool scale = EPI == XP_EPI_BIAS_QSCALE || EPI == XP_EPI_SCALE;
  static constexpr bool pre = EPI == XP_EPI_BIAS_RESID || EPI == XP_EPI_GELU_BWD;     // needs an [M, N] side input
};

template <typename T, bool F32>
__device__ __forceinline__ void bstore8(__amdgpu_buffer_rsrc_t r, unsigned off, const f32x8& v) {
  if constexpr (F32 || sizeof(T) == 4) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v.lo), r, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v.hi), r, off + 16, 0, 0);
  } else {
    bf16x8 o = {(bf16_t)v.lo[0], (bf16_t)v.lo[1], (bf16_t)v.lo[2], (bf16_t)v.lo[3],
                (bf16_t)v.hi[0], (bf16_t)v.hi[1], (bf16_t)v.hi[2], (bf16_t)v.hi[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r, off, 0, 0);
  }
}
// raw 8-element side input of type T (converted at use, so the load can stay in flight)
template <typename T> struct Raw8 { u32x4 a, b; };
template <typename T>
__device__ __forceinline__ Raw8<T> bload8(__amdgpu_buffer_rsrc_t r, unsigned off) {
  Raw8<T> x;
  x.a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  if constexpr (sizeof(T) == 4) x.b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0);
  return x;
}
template <typename T>
__device__ __forceinline__ f32x8 raw8_f32(const Raw8<T>& x) {
  if constexpr (sizeof(T) == 4) return f32x8{__builtin_bit_cast(f32x4, x.a), __builtin_bit_cast(f32x4, x.b)};
  else {
    const bf16x8 v = __builtin_bit_cast(bf16x8, x.a);
    return f32x8{f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]}};
  }
}

// Per-wave state of the fast epilogue; the lane owns columns n .. n+7.
template <typename T, int EPI, bool F32>
struct FastEpi {
  using Tr = EpiTraits<EPI>;
  static constexpr unsigned OSZ = F32 ? 4 : sizeof(T);
  __amdgpu_buffer_rsrc_t rc, rx;
  f32x8 bias; float cs_lo, cs_hi;
  unsigned ld_c, ld_x;      // row pitch in bytes
  unsigned col_c, col_x;    // byte offset of column n
  bool ok, keep_aux;
  SideRows side; int64_t ncol;
  __device__ __forceinline__ FastEpi(const KParams& p, void* Cbase, int64_t n) : side(p) {
    ok = n < p.N; ncol = ok ? n : 0;
    keep_aux = p.aux != nullptr;             // BIAS_GELU without aux: forward-only, the pre-activation is not stored
    const int64_t nn = ok ? n : 0;
    rc = __builtin_amdgcn_make_buffer_rsrc(Cbase, 0, (unsigned)(p.M * p.ldc * OSZ), 0x00020000);
    ld_c = (unsigned)(p.ldc * OSZ); col_c = (unsigned)(nn * OSZ);
    if constexpr (EPI == XP_EPI_BIAS_GELU) {
      rx = __builtin_amdgcn_make_buffer_rsrc(p.aux, 0, (unsigned)(p.M * p.ldaux * OSZ), 0x00020000);
      ld_x = (unsigned)(p.ldaux * OSZ); col_x = (unsigned)(nn * OSZ);
    } else if constexpr (Tr::pre) {
      rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.resid), 0, (unsigned)(p.M * p.ldr * sizeof(T)), 0x00020000);
      ld_x = (unsigned)(p.ldr * sizeof(T)); col_x = (unsigned)(nn * sizeof(T));
    } else { rx = rc; ld_x = 0; col_x = 0; }
    if constexpr (Tr::bias) bias = f32x8{load4(p.bias + nn), load4(p.bias + nn + 4)};
    if constexpr (EPI == XP_EPI_SCALE) cs_lo = cs_hi = p.scale;
    if constexpr (EPI == XP_EPI_BIAS_QSCALE) { cs_lo = nn < p.scale_cols ? p.scale : 1.f; cs_hi = nn + 4 < p.scale_cols ? p.scale : 1.f; }
  }
  __device__ __forceinline__ unsigned off_c(unsigned m) const { return ok ? m * ld_c + col_c : EPI_OOB; }
  __device__ __forceinline__ unsigned off_x(unsigned m) const { return ok ? m * ld_x + col_x : EPI_OOB; }
  __device__ __forceinline__ Raw8<T> load_pre(unsigned m) const { return bload8<T>(rx, off_x(m)); }
  __device__ __forceinline__ f32x8 finish(f32x8 v, const Raw8<T>& pre, unsigned m) const {
    if constexpr (Tr::bias) { v.lo += bias.lo; v.hi += bias.hi; }
    if constexpr (Tr::scale) { v.lo *= cs_lo; v.hi *= cs_hi; }
    if constexpr (EPI == XP_EPI_BIAS_GELU) {
      if (keep_aux) bstore8<T, F32>(rx, off_x(m), v);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v.lo[e] = quick_gelu_f(v.lo[e]); v.hi[e] = quick_gelu_f(v.hi[e]); }
    } else if constexpr (EPI == XP_EPI_BIAS_RESID) {
      int64_t sb;
      if (side.on() && ok && side.hit(m, sb)) {       // fp32 side row: residual operand and result in fp32 (rare, divergent)
        const f32x8 r = load8(side.rin + sb + ncol);
        v.lo += r.lo; v.hi += r.hi;
        store8(side.out + sb + ncol, v);
      } else {
        const f32x8 r = raw8_f32<T>(pre);
        v.lo += r.lo; v.hi += r.hi;
      }
    } else if constexpr (EPI == XP_EPI_GELU_BWD) {
      const f32x8 r = raw8_f32<T>(pre);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v.lo[e] *= quick_gelu_grad_f(r.lo[e]); v.hi[e] *= quick_gelu_grad_f(r.hi[e]); }
    }
    bstore8<T, F32>(rc, off_c(m), v);
    return v;
  }
};

// Calls f(integral_constant<int, EPI>, bool_constant<F32>, bool_constant<COLSUM>) for the (epilogue, output type) pairs the
// fast path specialises; returns false for any other pair (the caller then runs the generic epilogue).  COLSUM (column
// sums of the finished outputs, i.e. the bias gradient of the layer that produced this GEMM's input gradient) exists for
// the two epilogues the backward uses it with; the host rejects every other combination.
template <typename F>
__device__ __forceinline__ bool fast_epi_dispatch(const KParams& p, F&& f) {
  using std::integral_constant; using std::bool_constant;
  if (!p.fast_epi) return false;
  if (p.colsum) {
    if (p.epilogue == XP_EPI_GELU_BWD) f(integral_constant<int, XP_EPI_GELU_BWD>{}, bool_constant<false>{}, bool_constant<true>{});
    else                               f(integral_constant<int, XP_EPI_NONE>{}, bool_constant<false>{}, bool_constant<true>{});
    return true;
  }
  if (p.out_f32) {
    if (p.epilogue == XP_EPI_NONE) { f(integral_constant<int, XP_EPI_NONE>{}, bool_constant<true>{}, bool_constant<false>{}); return true; }
    return false;
  }
  switch (p.epilogue) {
    case XP_EPI_NONE:        f(integral_constant<int, XP_EPI_NONE>{}, bool_constant<false>{}, bool_constant<false>{}); return true;
    case XP_EPI_BIAS:        f(integral_constant<int, XP_EPI_BIAS>{}, bool_constant<false>{}, bool_constant<false>{}); return true;
    case XP_EPI_BIAS_QSCALE: f(integral_constant<int, XP_EPI_BIAS_QSCALE>{}, bool_constant<false>{}, bool_constant<false>{}); return true;
    case XP_EPI_BIAS_GELU:   f(integral_constant<int, XP_EPI_BIAS_GELU>{}, bool_constant<false>{}, bool_constant<false>{}); return true;
    case XP_EPI_BIAS_RESID:  f(integral_constant<int, XP_EPI_BIAS_RESID>{}, bool_constant<false>{}, bool_constant<false>{}); return true;
    case XP_EPI_GELU_BWD:    f(integral_constant<int, XP_EPI_GELU_BWD>{}, bool_constant<false>{}, bool_constant<false>{}); return true;
    default: return false;
  }
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
int xp_gemm256_group_n(const XpGemmDesc* d, int tiles_n);
bool xp_gemm256_legal(const XpGemmDesc* d);
bool xp_gemm256_wanted(const XpGemmDesc* d, int split);
int64_t xp_gemm256_colsum_rows(const XpGemmDesc* d);
bool xp_gemm_fast_epi_ok(const XpGemmDesc* d);
