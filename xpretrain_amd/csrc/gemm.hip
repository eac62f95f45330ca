// MFMA GEMM family for gfx950:  C[M,N] = epilogue( sum_k A(m,k) B(n,k) ).
//
// One kernel template covers every dense contraction on the CLIP-ViP path (forward linears, dX, dW, the
// patch-embedding conv-as-GEMM, projections, the loss logits) -- see include/xpretrain_hip.h.
//
// Tiling (wave64, 16x16 MFMA tiles): 128(M) x 128(N) block, 128 BYTES of k per stage (64 bf16 / 32 f32),
// 4 waves in 2x2, each wave a 64x64 sub-tile = 4x4 accumulators of f32x4.  Operands are staged
// global -> registers -> LDS (double-buffered, loads of stage s+1 in flight during the MFMAs of stage s).
// The MFMA A operand is the N side (weights) and the B operand the M side (activations): the
// accumulator lane then owns 4 CONSECUTIVE n for one m, so the epilogue is a 4-wide vector op
// (bias / residual / activation) and an 8-byte (bf16) or 16-byte (f32) store.
//
// k-strided operands (dX: W[n][k] with contraction over n; dW: dY[m][n], X[m][k] with contraction over m)
// are NOT transposed in memory: they are staged as [k][row] tiles and read with ds_read_b64_tr_b16
// (bf16) / ds_read_b32 (f32), which deliver exactly the MFMA fragment.
#include "common.h"
#include "gemm_common.h"
#include <stdlib.h>
#include <mutex>
#include <vector>

namespace {

using namespace xpgemm;
constexpr int BM = 128, BN = 128, BKB = 128;   // BKB: bytes of k per stage
constexpr int NT = 256;
constexpr int TILE_BYTES = 128 * 128;          // one operand tile per stage (16 KiB)

// ---- staging: global -> registers (4 x 16 B per thread per operand) -------------------------------
// k-contiguous operand: tile = [128 rows][128 B of k].
template <typename T>
__device__ __forceinline__ void gload_kc(u32x4 (&r)[4], const T* base, int64_t ld, int64_t row0, int64_t rows,
                                         int64_t k0, int64_t kend, const Remap& map, int tid) {
  constexpr int EPC = 16 / sizeof(T);           // elements per 16-byte chunk
  const int c = tid & 7, rr = tid >> 3;
  const int64_t k = k0 + c * EPC;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t row = row0 + rr + 32 * j;
    if (row < rows && k < kend)
      r[j] = *reinterpret_cast<const u32x4*>(base + map(row) * ld + k);
    else
      r[j] = u32x4{0, 0, 0, 0};
  }
}
// The patch matrix of a frame tensor as the k-contiguous A operand, gathered on the fly (the conv-as-GEMM of
// CLIPVisionViPEmbeddings.patch_embedding, CLIP_ViP.py:157-159,178): row m = patch (bt, gy, gx), k = (c, dy, dx); a thread's 8
// consecutive k are 8 consecutive pixels of one 16-pixel strip -- 32 contiguous bytes of fp32 frames (8 of uint8 frames) -- read
// straight from [BT,3,H,W], converted (uint8: the collate arithmetic (x / 255 - mean[c]) / std[c], same operation order as
// xp_im2col_u8) and rounded to bf16 exactly like xp_im2col / xp_im2col_u8 do: the staged tile is bit-identical to a tile of the
// materialised matrix.  The four rows of a thread are fixed over the k loop: their frame offsets are computed once (im2col_rows).
struct Im2colRows { int64_t base[4]; bool ok[4]; };
__device__ __forceinline__ Im2colRows im2col_rows(const KParams& p, int64_t row0, int tid) {
  Im2colRows r;
  const int rr = tid >> 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t m = row0 + rr + 32 * j;
    r.ok[j] = m < p.M;
    const int64_t mm = r.ok[j] ? m : 0;
    const int64_t bt = mm / p.im_L;
    const int l = (int)(mm - bt * p.im_L), gy = l / p.im_gw, gx = l - gy * p.im_gw;
    r.base[j] = (bt * 3 * p.im_H + (int64_t)gy * p.im_P) * p.im_W + (int64_t)gx * p.im_P;
  }
  return r;
}
template <bool U8>
__device__ __forceinline__ void gload_im2col(u32x4 (&r)[4], const KParams& p, const Im2colRows& rows, int64_t k0, int64_t kend, int tid) {
  const int c8 = tid & 7;
  const int64_t k = k0 + c8 * 8;
  const int pp = p.im_P * p.im_P;
  const int ch = (int)(k / pp), rem = (int)(k - (int64_t)ch * pp), dy = rem / p.im_P, dx = rem - dy * p.im_P;
  const int64_t koff = ((int64_t)ch * p.im_H + dy) * p.im_W + dx;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (rows.ok[j] && k < kend) {
      f32x4 a, b;
      if constexpr (!U8) {
        const float* src = reinterpret_cast<const float*>(p.im_src) + rows.base[j] + koff;
        a = load4(src); b = load4(src + 4);
      } else {
        const u32x2 raw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(p.im_src) + rows.base[j] + koff);
        const float mean = p.im_mean[ch], sd = p.im_std[ch];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = ((float)((raw[0] >> (8 * e)) & 0xFF) / 255.0f - mean) / sd;
          b[e] = ((float)((raw[1] >> (8 * e)) & 0xFF) / 255.0f - mean) / sd;
        }
      }
      const bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
      r[j] = __builtin_bit_cast(u32x4, o);
    } else {
      r[j] = u32x4{0, 0, 0, 0};
    }
  }
}

template <typename T>
__device__ __forceinline__ void lstore_kc(char* tile, const u32x4 (&r)[4], int tid) {
  const int c = tid & 7, rr = tid >> 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = rr + 32 * j;
    *reinterpret_cast<u32x4*>(tile + tile128_off(row, c)) = r[j];
  }
}

// k-strided operand: storage [k][row]; tile = [KE k-rows][128 rows]  (KE = 64 bf16 / 32 f32).
//   bf16: 256-byte k-rows, 32-byte blocks (16 rows) XOR-swizzled by f(k) = (k&3) | ((k>>3)&1)<<2
//   f32 : 512-byte k-rows, column XOR ((k>>2)&1)<<4

template <typename T>
__device__ __forceinline__ void gload_ks(u32x4 (&r)[4], const T* base, int64_t ld, int64_t row0, int64_t rows,
                                         int64_t k0, int64_t kend, const Remap& map, int tid) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int CPR = 128 / EPC;                // chunks per k-row: 16 (bf16) / 32 (f32)
  constexpr int KSTEP = NT / CPR;               // k-rows covered per pass: 16 / 8
  const int c = tid % CPR, kr = tid / CPR;
  const int64_t row = row0 + c * EPC;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t k = k0 + kr + KSTEP * j;
    if (k < kend && row < rows)
      r[j] = *reinterpret_cast<const u32x4*>(base + map(k) * ld + row);
    else
      r[j] = u32x4{0, 0, 0, 0};
  }
}
template <typename T>
__device__ __forceinline__ void lstore_ks(char* tile, const u32x4 (&r)[4], int tid) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int CPR = 128 / EPC;
  constexpr int KSTEP = NT / CPR;
  const int c = tid % CPR, kr0 = tid / CPR;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kr = kr0 + KSTEP * j;
    int off;
    if (sizeof(T) == 2) off = kr * 256 + ((((c >> 1) ^ ks_f(kr)) << 5) | ((c & 1) << 4));
    else                off = kr * 512 + (((c * 4) ^ (((kr >> 2) & 1) << 4)) << 2);
    *reinterpret_cast<u32x4*>(tile + off) = r[j];
  }
}

// ---- fast staging: global -> LDS directly (buffer_load ... lds, 16 B per lane, no VGPR round trip, no ds_write) --
// One wave instruction fills 1 KiB of LDS lane-linearly (base + lane*16), so the tile image is the same as
// the register-staged one only if the XOR swizzle is applied to the per-lane GLOBAL source address.  The buffer
// descriptor's bounds check zero-fills rows / k-rows beyond the operand (no branches for the M and split-K tails).
// Preconditions (checked on the host): dense operand, no row remap, k-contiguous: K % (128/sizeof T) == 0;
// k-strided: rows % 128 == 0; operand bytes < 4 GiB.
typedef __attribute__((address_space(3))) char lds_char;

template <typename T, bool KS, int NWV>
struct GldsStager {
  static constexpr int NPW = 16 / NWV;   // 1 KiB DMA pieces per wave per operand tile (16 KiB)
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[NPW];   // per-lane byte offset of this wave's piece j at k-tile 0
  unsigned step;        // byte advance per k-tile
  unsigned lds_wave;    // wave-uniform byte offset of this wave's first 1 KiB slot

  __device__ __forceinline__ void init(const T* base, int64_t ld, int64_t row0, int64_t rows, int64_t K, int64_t kbeg,
                                       int lane, int wave) {
    constexpr int ES = sizeof(T);
    const int64_t bytes = (KS ? K : rows) * ld * ES;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (unsigned)bytes, 0x00020000);
    lds_wave = wave * 1024;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int pass = j * NWV + wave;
      int64_t off;
      if constexpr (!KS) {
        const int row = pass * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz128(row);
        off = ((row0 + row) * ld + kbeg) * ES + c * 16;
      } else if constexpr (sizeof(T) == 2) {
        const int kr = pass * 4 + (lane >> 4), c16 = lane & 15;
        const int src = (((c16 >> 1) ^ ks_f(kr)) << 1) | (c16 & 1);
        off = ((kbeg + kr) * ld + row0) * ES + src * 16;
      } else {
        const int kr = pass * 2 + (lane >> 5), c32 = lane & 31;
        const int col = (c32 * 4) ^ (((kr >> 2) & 1) << 4);
        off = ((kbeg + kr) * ld + row0 + col) * ES;
      }
      // anything past the end must stay past the end after the 32-bit cast
      voff[j] = off >= bytes ? 0xFFFFFFF0u : (unsigned)off;
    }
    constexpr int KE = BKB / ES;
    step = (unsigned)((KS ? (int64_t)KE * ld : (int64_t)KE) * ES);
  }
  // issue the 4 DMA passes of k-tile kt into `tile` (16 KiB)
  __device__ __forceinline__ void issue(char* tile, int kt) const {
    lds_char* t3 = (lds_char*)tile;
    const unsigned adv = (unsigned)kt * step;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      unsigned o = voff[j] + adv;
      if (o < voff[j]) o = 0xFFFFFFF0u;       // wrapped: was (and stays) out of bounds
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, t3 + (j * NWV * 1024 + lds_wave), 16, o, 0, 0, 0);
    }
  }
};

// ---- fragments: LDS -> registers ------------------------------------------------------------------
// ot: 16-row sub-tile index (0..7) inside the 128-row tile; ks: 64-byte k super-step (0..1).
template <typename T, bool KS>
__device__ __forceinline__ typename Frag<T>::type lfrag(const char* tile, int ot, int ks, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if constexpr (!KS) {
    const int row = ot * 16 + i;
    return *reinterpret_cast<const typename Frag<T>::type*>(tile + tile128_off(row, ks * 4 + g));
  } else if constexpr (sizeof(T) == 2) {
    // two transpose reads: k = ks*32 + 8g + {0..3}, {4..7}; lane supplies k-row (i>>2), 4 rows at (i&3)*4
    const int f = (i >> 2) | ((g & 1) << 2);
    const int kr = ks * 32 + g * 8 + (i >> 2);
    const char* p = tile + kr * 256 + ((ot ^ f) << 5) + ((i & 3) << 3);
    i16x4 lo = lds_read_tr16(p);
    i16x4 hi = lds_read_tr16(p + 4 * 256);
    typedef __attribute__((ext_vector_type(8))) short i16x8;
    i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  } else {
    f32x4 v;
    const int col = (ot * 16 + i) ^ ((g & 1) << 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kr = ks * 16 + 4 * g + e;
      v[e] = *reinterpret_cast<const float*>(tile + kr * 512 + col * 4);
    }
    return v;
  }
}

// ---- the kernel -----------------------------------------------------------------------------------
// AIM: 0 = A is a matrix; 1 / 2 = A is gathered from fp32 / uint8 frames (register-staged bf16 NT kernel only)
template <typename T, bool AKS, bool BKS, bool GLDS, int AIM = 0>
__global__ __launch_bounds__(GLDS ? 2 * NT : NT) void gemm_kernel(KParams p) {
  static_assert(AIM == 0 || (!AKS && !GLDS && sizeof(T) == 2), "on-the-fly patch gather: bf16, k-contiguous A, register staging");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto sA = [&](int s) -> char* { return smem + (2 * s) * TILE_BYTES; };
  auto sB = [&](int s) -> char* { return smem + (2 * s + 1) * TILE_BYTES; };

  // register-staged family: 4 waves as 2x2, wave tile 64x64.  direct-to-LDS family: 8 waves as 2(M) x 4(N), wave tile
  // 64x32 -- four waves per SIMD (two workgroups per CU) give the latency-bound stage loop twice the wave-level
  // parallelism, and each wave issues only 4 DMA pieces per stage.
  constexpr int NWV = GLDS ? 8 : 4, WN = GLDS ? 4 : 2, NTW = 8 / WN;     // NTW: 16-col sub-tiles per wave (2 / 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware bijective remap: consecutive tile ids land on the same XCD (block b runs on XCD b % 8),
  // so tiles sharing an activation row-panel share one L2.
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg, p.xcd_remap);
  int tm, tn;
  tile_of(bid, p.tiles_m, p.tiles_n, p.group_n, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  constexpr int KE = BKB / sizeof(T);
  const int64_t kbeg = (int64_t)blockIdx.z * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nk = (int)((kend - kbeg + KE - 1) / KE);

  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);
  const Remap ident{0, 0, 0};

  f32x4 acc[NTW][4];
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  Im2colRows imr;
  if constexpr (AIM != 0) imr = im2col_rows(p, m0, tid);
  auto gload = [&](int64_t k0) {
    if constexpr (AIM != 0) gload_im2col<AIM == 2>(ra, p, imr, k0, kend, tid);
    else if constexpr (AKS) gload_ks<T>(ra, A, p.lda, m0, p.M, k0, kend, p.amap, tid);
    else               gload_kc<T>(ra, A, p.lda, m0, p.M, k0, kend, p.amap, tid);
    if constexpr (BKS) gload_ks<T>(rb, B, p.ldb, n0, p.N, k0, kend, ident, tid);
    else               gload_kc<T>(rb, B, p.ldb, n0, p.N, k0, kend, ident, tid);
  };
  auto lstore = [&](int s) {
    if constexpr (AKS) lstore_ks<T>(sA(s), ra, tid); else lstore_kc<T>(sA(s), ra, tid);
    if constexpr (BKS) lstore_ks<T>(sB(s), rb, tid); else lstore_kc<T>(sB(s), rb, tid);
  };

  GldsStager<T, AKS, NWV> ga;
  GldsStager<T, BKS, NWV> gb;
  if constexpr (GLDS) {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    ga.init(A, p.lda, m0, p.M, kend, kbeg, lane, wv);
    gb.init(B, p.ldb, n0, p.N, kend, kbeg, lane, wv);
    if (nk > 0) { ga.issue(sA(0), 0); gb.issue(sB(0), 0); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (nk > 0) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();

  // optional trace: wave 0 of the middle workgroup stamps s_memtime at 5 points of each of its first 24 k-iterations
  const bool trace = p.dbg != nullptr && (int)blockIdx.x == nwg / 2 && blockIdx.z == 0 && wave == 0;
  unsigned long long* tr = p.dbg;
#define XP_STAMP(i) do { if (trace && kt < 24) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) tr[8 + kt * 5 + (i)] = t_; } } while (0)
  if (trace && lane == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[1] = nk; }
  for (int kt = 0; kt < nk; ++kt) {
    const int s = kt & 1;
    XP_STAMP(0);
    if constexpr (GLDS) {
      if (kt + 1 < nk) { ga.issue(sA(s ^ 1), kt + 1); gb.issue(sB(s ^ 1), kt + 1); }
    } else {
      if (kt + 1 < nk) gload(kbeg + (int64_t)(kt + 1) * KE);
    }
    XP_STAMP(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Frag<T>::type fw[NTW], fx[4];
#pragma unroll
      for (int t = 0; t < NTW; ++t) fw[t] = lfrag<T, BKS>(sB(s), wn * NTW + t, ks, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) fx[t] = lfrag<T, AKS>(sA(s), wm * 4 + t, ks, lane);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mma16(fw[nt], fx[mt], acc[nt][mt]);
    }
    XP_STAMP(2);
    if constexpr (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (kt + 1 < nk) lstore(s ^ 1);
    }
    XP_STAMP(3);
    __syncthreads();
    XP_STAMP(4);
  }
  if (trace && lane == 0) tr[2] = __builtin_amdgcn_s_memtime();
#undef XP_STAMP

  // ---- epilogue ---------------------------------------------------------------------------------------
  // The accumulator lane owns 4 consecutive n of ONE row, i.e. a wave store would touch 16 rows x 32 B.  Instead
  // the raw fp32 tile is staged through the (now idle) 64 KiB of LDS -- [128][128] fp32, 16-byte chunks XOR-
  // swizzled by (row & 7) -- and read back row-major: 32 lanes cover one 512-byte fp32 row, so bias / residual /
  // table loads and the bf16 stores are full-line, fully coalesced accesses.  All epilogue math is fp32.
  {
    const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = wm * 64 + mt * 16 + i16;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int chunk = wn * (NTW * 4) + nt * 4 + g;
        *reinterpret_cast<f32x4*>(smem + row * 512 + ((chunk ^ (row & 7)) << 4)) = acc[nt][mt];
      }
    }
  }
  __syncthreads();

  float* Cf = reinterpret_cast<float*>(p.C);
  T* Ct = reinterpret_cast<T*>(p.C);
  if (gridDim.z > 1) Cf += (int64_t)blockIdx.z * p.M * p.N;
  const bool fast = fast_epi_dispatch(p, [&](auto epi_c, auto f32_c, auto) {      // (no fused column sums in this family)
    constexpr int EPI = decltype(epi_c)::value;
    constexpr bool F32 = decltype(f32_c)::value;
    constexpr int RPP = NWV * 4, NP = 128 / RPP;    // 16 lanes per row, RPP rows per pass
    const int c8 = tid & 15, r16 = tid >> 4;
    const FastEpi<T, EPI, F32> fe(p, F32 ? (void*)Cf : (void*)Ct, n0 + c8 * 8);
    const unsigned mrow = (unsigned)m0 + r16;
    Raw8<T> pre[NP];
    if constexpr (EpiTraits<EPI>::pre) {
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) pre[pass] = fe.load_pre(mrow + pass * RPP);
    }
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int row = pass * RPP + r16;
      const char* rp = smem + row * 512;
      const f32x8 v = {*reinterpret_cast<const f32x4*>(rp + (((2 * c8) ^ (row & 7)) << 4)),
                       *reinterpret_cast<const f32x4*>(rp + (((2 * c8 + 1) ^ (row & 7)) << 4))};
      fe.finish(v, pre[pass], mrow + pass * RPP);
    }
  });
  if (fast) {
    if (trace && lane == 0) tr[3] = __builtin_amdgcn_s_memtime();
    return;
  }
  if (p.wide) {          // 8 columns per lane: 16 lanes cover a row, 16 rows per pass, 16-byte bf16 stores
    constexpr int RPP = NWV * 4;                    // rows per pass: 16 lanes per row
    const int c8 = tid & 15, r16 = tid >> 4;
    const int64_t n8 = n0 + c8 * 8;
    if (n8 < p.N) {
      const EpiLane8 el8(p, n8);
#pragma unroll 4
      for (int pass = 0; pass < 128 / RPP; ++pass) {
        const int row = pass * RPP + r16;
        const int64_t m = m0 + row;
        if (m >= p.M) break;
        const char* rp = smem + row * 512;
        const f32x8 v = {*reinterpret_cast<const f32x4*>(rp + (((2 * c8) ^ (row & 7)) << 4)),
                         *reinterpret_cast<const f32x4*>(rp + (((2 * c8 + 1) ^ (row & 7)) << 4))};
        epi_row8<T>(p, el8, v, m, n8, Cf, Ct);
      }
    }
    if (trace && lane == 0) tr[3] = __builtin_amdgcn_s_memtime();
    return;
  }
  const int c = tid & 31, r8 = tid >> 5;
  const int64_t n = n0 + c * 4;
  if (n >= p.N) return;
  const EpiLane el(p, n);

#pragma unroll 4
  for (int pass = 0; pass < 128 / (NWV * 2); ++pass) {
    const int row = pass * (NWV * 2) + r8;
    const int64_t m = m0 + row;
    if (m >= p.M) break;
    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * 512 + ((c ^ (row & 7)) << 4));
    epi_row<T>(p, el, v, m, n, Cf, Ct);
  }
  if (trace && lane == 0) tr[3] = __builtin_amdgcn_s_memtime();
}

template <typename T, bool GLDS>
void launch2(const XpGemmDesc* d, const KParams& kp, dim3 grid, hipStream_t st) {
  const size_t lds = 4 * TILE_BYTES;
  if (!d->a_kstrided && !d->b_kstrided)      gemm_kernel<T, false, false, GLDS><<<grid, GLDS ? 2 * NT : NT, lds, st>>>(kp);
  else if (!d->a_kstrided && d->b_kstrided)  gemm_kernel<T, false, true, GLDS><<<grid, GLDS ? 2 * NT : NT, lds, st>>>(kp);
  else if (d->a_kstrided && d->b_kstrided)   gemm_kernel<T, true, true, GLDS><<<grid, GLDS ? 2 * NT : NT, lds, st>>>(kp);
  else                                       gemm_kernel<T, true, false, GLDS><<<grid, GLDS ? 2 * NT : NT, lds, st>>>(kp);
}

// The direct-to-LDS path needs dense, un-remapped operands whose tails the buffer bounds check can zero-fill.
bool glds_ok(const XpGemmDesc* d, int esz) {
  if (xp_debug_flag("gemm_no_glds")) return false;
  if (d->a_grp != 0) return false;
  const int64_t ke = BKB / esz;
  const int64_t a_rows = d->a_kstrided ? d->K : d->M, b_rows = d->b_kstrided ? d->K : d->N;
  if (!d->a_kstrided && (d->K % ke != 0 || d->lda != d->K)) return false;
  if (!d->b_kstrided && (d->K % ke != 0 || d->ldb != d->K)) return false;
  if (d->a_kstrided && (d->M % BM != 0 || d->lda != d->M)) return false;
  if (d->b_kstrided && (d->N % BN != 0 || d->ldb != d->N)) return false;
  const int64_t lim = (int64_t)0xFFFFFFF0u - 256 * 1024 * 1024;
  if ((a_rows + BM) * d->lda * esz >= lim || (b_rows + BN) * d->ldb * esz >= lim) return false;
  return true;
}

template <typename T>
int launch(const XpGemmDesc* d, const KParams& kp, dim3 grid, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (d->a_frames) {                  // A gathered from the frame tensor by the loader
      const size_t lds = 4 * TILE_BYTES;
      if (d->a_frames_u8) gemm_kernel<T, false, false, false, 2><<<grid, NT, lds, st>>>(kp);
      else                gemm_kernel<T, false, false, false, 1><<<grid, NT, lds, st>>>(kp);
      return 0;
    }
  }
  if (glds_ok(d, sizeof(T))) launch2<T, true>(d, kp, grid, st);
  else                       launch2<T, false>(d, kp, grid, st);
  return 0;
}

// ---- split-K slab reduce, column sums ---------------------------------------------------------------
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n4,
                                     int splits, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const f32x4* sl = reinterpret_cast<const f32x4*>(slabs);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 s = accumulate ? reinterpret_cast<const f32x4*>(out)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 4 <= splits; z += 4) {         // 4 independent 16-byte loads in flight per lane
      const f32x4 a = sl[(int64_t)z * n4 + i], b = sl[(int64_t)(z + 1) * n4 + i];
      const f32x4 c = sl[(int64_t)(z + 2) * n4 + i], d = sl[(int64_t)(z + 3) * n4 + i];
      s += (a + b) + (c + d);
    }
    for (; z < splits; ++z) s += sl[(int64_t)z * n4 + i];
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
}

// Column sums (bias gradients): block = 4 waves; wave w sums rows r0+w, r0+w+4, ... of a CS_ROWS-row chunk for 256
// columns (4 per lane, 8/16-byte loads, 512 B contiguous per wave instruction), 4-deep independent accumulators;
// LDS combine of the 4 waves; one partial row per chunk -> splitk_reduce.  Grid: (cols/256, rows/CS_ROWS).
// rows per chunk: narrow matrices need many chunks to fill the chip, wide ones can take longer chunks (fewer partial rows
// for the second level): aim at >= ~2048 workgroups, 32..128 rows each
static inline int cs_rows(int64_t rows, int64_t cols) {
  const int64_t per = rows * cdiv(cols, 256) / 2048;
  return per >= 128 ? 128 : (per >= 64 ? 64 : 32);
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ X, int64_t rows, int64_t cols, int64_t ldx,
                                                             float* __restrict__ part, int CS_ROWS) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t c = ((int64_t)blockIdx.x * 64 + lane) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
  const int64_t r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (c < cols) {
    int64_t r = r0 + w;
    for (; r + 4 < r1; r += 8) { s0 += load4(X + r * ldx + c); s1 += load4(X + (r + 4) * ldx + c); }
    if (r < r1) s0 += load4(X + r * ldx + c);
  }
  store4(&red[w][lane * 4], s0 + s1);
  __syncthreads();
  if (w == 0 && c < cols) {
    const f32x4 t = load4(&red[0][lane * 4]) + load4(&red[1][lane * 4]) + load4(&red[2][lane * 4]) + load4(&red[3][lane * 4]);
    store4(part + (int64_t)blockIdx.y * cols + c, t);
  }
}

// out[y][c] (+)= sum of `nsum` consecutive rows of in[.][width] starting at y*nsum; 64 columns per block, 4 waves
// take every 4th row, LDS combine (deterministic order).
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ in, float* __restrict__ out, int nrows,
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

// ---- batched two-level row reduction (bias / LayerNorm-parameter gradients of one encoder layer in two launches) ----
struct BatchArgs {
  XpReduceSeg seg[XP_REDUCE_MAX_SEGS];
  float* part2[XP_REDUCE_MAX_SEGS];        // level-1 output [<=32][width] per segment
  int cb0[XP_REDUCE_MAX_SEGS + 1];         // prefix sum of 64-column blocks
  int n;
};
constexpr int RB_DIRECT = 64;              // segments with <= this many rows skip level 1

__device__ __forceinline__ int batch_find(const BatchArgs& a, int bx) {
  int s = 0;
  while (s + 1 < a.n && bx >= a.cb0[s + 1]) ++s;
  return s;
}

// sum rows [r0, r1) of `in` (pitch `stride`) for column c; 4 waves take every 4th row, LDS combine (fixed order)
__device__ __forceinline__ float batch_colsum(const float* __restrict__ in, int64_t stride, int r0, int r1, int c, bool ok,
                                              float (*red)[64]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (ok) {
    int r = r0 + w;
    for (; r + 4 < r1; r += 8) { s0 += in[(int64_t)r * stride + c]; s1 += in[(int64_t)(r + 4) * stride + c]; }
    if (r < r1) s0 += in[(int64_t)r * stride + c];
  }
  red[w][lane] = s0 + s1;
  __syncthreads();
  return red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

__global__ __launch_bounds__(256) void reduce_batch_l1_kernel(BatchArgs a) {
  __shared__ float red[4][64];
  const int s = batch_find(a, blockIdx.x);
  const XpReduceSeg sg = a.seg[s];
  if (sg.nrows <= RB_DIRECT) return;
  const int nsum = (sg.nrows + 31) / 32;
  const int r0 = blockIdx.y * nsum;
  if (r0 >= sg.nrows) return;
  const int r1 = r0 + nsum < sg.nrows ? r0 + nsum : sg.nrows;
  const int c = (blockIdx.x - a.cb0[s]) * 64 + (threadIdx.x & 63);
  const float t = batch_colsum(sg.in, sg.stride, r0, r1, c, c < sg.width, red);
  if (threadIdx.x < 64 && c < sg.width) a.part2[s][(int64_t)blockIdx.y * sg.width + c] = t;
}

__global__ __launch_bounds__(256) void reduce_batch_l2_kernel(BatchArgs a) {
  __shared__ float red[4][64];
  const int s = batch_find(a, blockIdx.x);
  const XpReduceSeg sg = a.seg[s];
  const bool direct = sg.nrows <= RB_DIRECT;
  const int nsum = (sg.nrows + 31) / 32;
  const int n2 = direct ? sg.nrows : (sg.nrows + nsum - 1) / nsum;
  const int c = (blockIdx.x - a.cb0[s]) * 64 + (threadIdx.x & 63);
  const float t = batch_colsum(direct ? sg.in : a.part2[s], direct ? sg.stride : (int64_t)sg.width, 0, n2, c, c < sg.width, red);
  if (threadIdx.x < 64 && c < sg.width) sg.out[c] = sg.accumulate ? sg.out[c] + t : t;
}

// The same two levels with FOUR adjacent columns per lane (16-byte loads, 256 columns per workgroup): a quarter of the workgroups --
// the scalar level 1 of a ViT-B layer is 4992 workgroups of ~4 KB each, 20 us of workgroup dispatch on the backward's critical
// stream for 14 MB.  Per column the rows are summed in exactly the order of batch_colsum (wave w takes rows r0 + w, + 8, ... into s0 and
// r0 + w + 4, ... into s1; four wave sums combined in order): results are bit-identical to the scalar kernels.
__device__ __forceinline__ f32x4 batch_colsum4(const float* __restrict__ in, int64_t stride, int r0, int r1, int c, bool ok,
                                               f32x4 (*red)[64]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    int r = r0 + w;
    for (; r + 4 < r1; r += 8) {
      s0 += *reinterpret_cast<const f32x4*>(in + (int64_t)r * stride + c);
      s1 += *reinterpret_cast<const f32x4*>(in + (int64_t)(r + 4) * stride + c);
    }
    if (r < r1) s0 += *reinterpret_cast<const f32x4*>(in + (int64_t)r * stride + c);
  }
  red[w][lane] = s0 + s1;
  __syncthreads();
  return ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

__global__ __launch_bounds__(256) void reduce_batch4_l1_kernel(BatchArgs a) {
  __shared__ f32x4 red[4][64];
  const int s = batch_find(a, blockIdx.x);
  const XpReduceSeg sg = a.seg[s];
  if (sg.nrows <= RB_DIRECT) return;
  const int nsum = (sg.nrows + 31) / 32;
  const int r0 = blockIdx.y * nsum;
  if (r0 >= sg.nrows) return;
  const int r1 = r0 + nsum < sg.nrows ? r0 + nsum : sg.nrows;
  const int c = (blockIdx.x - a.cb0[s]) * 256 + (threadIdx.x & 63) * 4;
  const f32x4 t = batch_colsum4(sg.in, sg.stride, r0, r1, c, c < sg.width, red);
  if (threadIdx.x < 64 && c < sg.width) *reinterpret_cast<f32x4*>(a.part2[s] + (int64_t)blockIdx.y * sg.width + c) = t;
}

__global__ __launch_bounds__(256) void reduce_batch4_l2_kernel(BatchArgs a) {
  __shared__ f32x4 red[4][64];
  const int s = batch_find(a, blockIdx.x);
  const XpReduceSeg sg = a.seg[s];
  const bool direct = sg.nrows <= RB_DIRECT;
  const int nsum = (sg.nrows + 31) / 32;
  const int n2 = direct ? sg.nrows : (sg.nrows + nsum - 1) / nsum;
  const int c = (blockIdx.x - a.cb0[s]) * 256 + (threadIdx.x & 63) * 4;
  const f32x4 t = batch_colsum4(direct ? sg.in : a.part2[s], direct ? sg.stride : (int64_t)sg.width, 0, n2, c, c < sg.width, red);
  if (threadIdx.x < 64 && c < sg.width) {
    f32x4* o = reinterpret_cast<f32x4*>(sg.out + c);
    *o = sg.accumulate ? *o + t : t;
  }
}

}  // namespace

// occupancy query for the default bf16 NT direct-to-LDS kernel: resident workgroups per CU at `lds_bytes` dynamic LDS
extern "C" int xp_debug_gemm_occupancy(int lds_bytes) {
  int nb = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<bf16_t, false, false, true>, 2 * NT, (size_t)lds_bytes);
  if (e != hipSuccess) return -(int)e;
  return nb;
}

static unsigned long long* g_gemm_trace = nullptr;
extern "C" int xp_debug_set_gemm_trace(void* device_buffer) { g_gemm_trace = (unsigned long long*)device_buffer; return XP_OK; }

// ---- in-step timing of ONE GEMM shape (bench.py's roofline: the dominant kernel timed where it runs, inside training steps) ----
// While armed, every xp_gemm call whose (M, N, K, epilogue, operand layout, split) match is bracketed by a pair of HIP events on
// the stream it is launched on; xp_debug_gemm_timer_read synchronises them and returns the elapsed times.  Host-side only (two
// hipEventRecord per matching launch); not armed, it costs one pointer test per xp_gemm call.
namespace {
struct GemmTimer {
  std::mutex mu;
  bool armed = false;
  int64_t M = 0, N = 0, K = 0; int epi = 0, aks = 0, bks = 0, split = 1;
  std::vector<hipEvent_t> ev;      // pairs: start, stop
  size_t used = 0;
} g_timer;
static volatile bool g_timer_on = false;
}  // namespace

extern "C" int xp_debug_gemm_timer_arm(int64_t M, int64_t N, int64_t K, int32_t epilogue, int32_t a_kstrided, int32_t b_kstrided,
                                       int32_t split_k, int32_t max_launches) {
  XP_REQUIRE(max_launches > 0 && max_launches <= 4096, "xp_debug_gemm_timer_arm: max_launches must be 1..4096");
  std::lock_guard<std::mutex> lock(g_timer.mu);
  while (g_timer.ev.size() < (size_t)max_launches * 2) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { xp_set_error("xp_debug_gemm_timer_arm: hipEventCreate failed"); return XP_ERR_LAUNCH; }
    g_timer.ev.push_back(e);
  }
  g_timer.M = M; g_timer.N = N; g_timer.K = K; g_timer.epi = epilogue; g_timer.aks = a_kstrided; g_timer.bks = b_kstrided;
  g_timer.split = split_k > 1 ? split_k : 1;
  g_timer.used = 0; g_timer.armed = true; g_timer_on = true;
  return XP_OK;
}

// disarms; writes up to `cap` elapsed times (ms) of the bracketed launches, returns how many were bracketed (-1 on error)
extern "C" int32_t xp_debug_gemm_timer_read(float* ms, int32_t cap) {
  std::lock_guard<std::mutex> lock(g_timer.mu);
  g_timer_on = false; g_timer.armed = false;
  const int n = (int)(g_timer.used / 2);
  for (int i = 0; i < n && i < cap; ++i) {
    if (hipEventSynchronize(g_timer.ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms[i], g_timer.ev[2 * i], g_timer.ev[2 * i + 1]) != hipSuccess) {
      xp_set_error("xp_debug_gemm_timer_read: event query failed");
      return -1;
    }
  }
  g_timer.used = 0;
  return n;
}

// returns the index of the event pair to record around this launch, or -1
static int gemm_timer_slot(const XpGemmDesc* d, int split) {
  if (!g_timer_on) return -1;
  std::lock_guard<std::mutex> lock(g_timer.mu);
  if (!g_timer.armed || d->M != g_timer.M || d->N != g_timer.N || d->K != g_timer.K || d->epilogue != g_timer.epi ||
      (d->a_kstrided != 0) != (g_timer.aks != 0) || (d->b_kstrided != 0) != (g_timer.bks != 0) || split != g_timer.split ||
      g_timer.used + 2 > g_timer.ev.size())
    return -1;
  const int slot = (int)g_timer.used;
  g_timer.used += 2;
  return slot;
}

extern "C" int xp_gemm(const XpGemmDesc* d, void* stream) {
  XP_REQUIRE(d && (d->A || d->a_frames) && d->B && d->C, "xp_gemm: null operand");
  XP_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "xp_gemm: empty problem M=%lld N=%lld K=%lld",
             (long long)d->M, (long long)d->N, (long long)d->K);
  XP_REQUIRE(d->in_dtype == XP_BF16 || d->in_dtype == XP_F32, "xp_gemm: bad in_dtype %d", d->in_dtype);
  XP_REQUIRE(d->out_dtype == d->in_dtype || d->out_dtype == XP_F32, "xp_gemm: bad out_dtype %d", d->out_dtype);
  const int esz = d->in_dtype == XP_BF16 ? 2 : 4;
  const int epc = 16 / esz;
  // 16-byte chunks along the contiguous dimension of each operand
  const int64_t a_contig = d->a_kstrided ? d->M : d->K, b_contig = d->b_kstrided ? d->N : d->K;
  XP_REQUIRE(a_contig % epc == 0 && b_contig % epc == 0 && d->lda % epc == 0 && d->ldb % epc == 0,
             "xp_gemm: contiguous extents / leading dims must be multiples of %d elements", epc);
  XP_REQUIRE(d->N % 4 == 0 && d->ldc % 4 == 0, "xp_gemm: N and ldc must be multiples of 4");
  XP_REQUIRE(((uintptr_t)d->A | (uintptr_t)d->B | (uintptr_t)d->C) % 16 == 0, "xp_gemm: operands must be 16-byte aligned");
  if (d->a_frames) {
    const int64_t P = d->fr_P, gh = P > 0 ? d->fr_H / P : 0, gw = P > 0 ? d->fr_W / P : 0;
    XP_REQUIRE(d->in_dtype == XP_BF16 && !d->a_kstrided && !d->b_kstrided && d->a_grp == 0 && (d->split_k <= 1),
               "xp_gemm: a_frames needs bf16 compute, k-contiguous operands, no row remap of A, no split-K");
    XP_REQUIRE(P > 0 && P % 8 == 0 && d->fr_H % P == 0 && d->fr_W % P == 0 && d->K == 3 * P * P && gh * gw > 0 && d->M % (gh * gw) == 0,
               "xp_gemm: a_frames needs P %% 8 == 0, H and W multiples of P, K == 3*P*P and M a multiple of the patches per frame");
    XP_REQUIRE(((uintptr_t)d->a_frames & (d->a_frames_u8 ? 7 : 15)) == 0 && (d->a_frames_u8 ? d->fr_W % 8 == 0 : d->fr_W % 4 == 0),
               "xp_gemm: a_frames must be 16-byte (fp32) / 8-byte (uint8) aligned with a row pitch that keeps 8-pixel strips aligned");
    if (d->a_frames_u8) for (int c = 0; c < 3; ++c) XP_REQUIRE(d->fr_std[c] > 0.f, "xp_gemm: fr_std[%d] must be positive", c);
  }
  const int ep = d->epilogue;
  XP_REQUIRE(ep >= XP_EPI_NONE && ep <= XP_EPI_SCALE, "xp_gemm: bad epilogue %d", ep);
  if (ep == XP_EPI_BIAS || ep == XP_EPI_BIAS_QSCALE || ep == XP_EPI_BIAS_GELU || ep == XP_EPI_BIAS_RESID)
    XP_REQUIRE(d->bias, "xp_gemm: epilogue %d needs bias", ep);
  if (ep == XP_EPI_BIAS_RESID || ep == XP_EPI_GELU_BWD) XP_REQUIRE(d->resid && d->ldr % 4 == 0, "xp_gemm: epilogue %d needs resid", ep);
  if (ep == XP_EPI_BIAS_GELU) XP_REQUIRE(!d->aux || d->ldaux % 4 == 0, "xp_gemm: epilogue BIAS_GELU: bad ldaux");
  if (ep == XP_EPI_PATCH) XP_REQUIRE(d->tab1 && d->tab2 && d->tab_L > 0, "xp_gemm: epilogue PATCH needs tab1/tab2/tab_L");
  const int split = d->split_k > 1 ? d->split_k : 1;
  if (split > 1) XP_REQUIRE(ep == XP_EPI_NONE && d->out_dtype == XP_F32 && d->c_grp == 0 && d->ldc == d->N,
                            "xp_gemm: split_k needs EPI_NONE, f32 output, dense C");

  KParams kp;
  kp.A = d->A; kp.B = d->B; kp.C = d->C;
  kp.M = d->M; kp.N = d->N; kp.K = d->K; kp.lda = d->lda; kp.ldb = d->ldb; kp.ldc = d->ldc;
  kp.amap = Remap{d->a_grp, d->a_grp_stride, d->a_off};
  kp.cmap = Remap{d->c_grp, d->c_grp_stride, d->c_off};
  kp.epilogue = ep; kp.out_f32 = d->out_dtype == XP_F32;
  const int ke = BKB / esz;
  kp.k_per_split = cdiv(cdiv(d->K, split), ke) * ke;
  kp.bias = d->bias; kp.scale = d->scale; kp.scale_cols = d->scale_cols;
  kp.resid = d->resid; kp.ldr = d->ldr; kp.aux = d->aux; kp.ldaux = d->ldaux;
  kp.tab1 = d->tab1; kp.tab2 = d->tab2; kp.tab_L = d->tab_L;
  kp.dbg = g_gemm_trace; kp.flat_split = 0;
  kp.im_src = d->a_frames; kp.im_u8 = d->a_frames_u8; kp.im_H = d->fr_H; kp.im_W = d->fr_W; kp.im_P = d->fr_P;
  kp.im_gw = d->fr_P > 0 ? d->fr_W / d->fr_P : 1; kp.im_L = d->fr_P > 0 ? (d->fr_H / d->fr_P) * kp.im_gw : 1;
  for (int c = 0; c < 3; ++c) { kp.im_mean[c] = d->fr_mean[c]; kp.im_std[c] = d->fr_std[c]; }
  kp.wide = (d->N % 8 == 0 && d->ldc % 8 == 0 && (!d->resid || d->ldr % 8 == 0) && (!d->aux || d->ldaux % 8 == 0)) ? 1 : 0;
  kp.fast_epi = (kp.wide && xp_gemm_fast_epi_ok(d)) ? 1 : 0;
  kp.colsum = d->colsum_partials;
  kp.rside = d->resid_side; kp.oside = d->out_side; kp.side_S = (unsigned)d->side_S; kp.side_M = (unsigned)d->side_M;
  if (d->resid_side || d->out_side)
    XP_REQUIRE(d->resid_side && d->out_side && ep == XP_EPI_BIAS_RESID && d->out_dtype == d->in_dtype && d->c_grp == 0 &&
               d->side_S > 0 && d->side_M > 0 && d->side_M <= d->side_S && d->M + 512 < ((int64_t)1 << 24) && d->N % 8 == 0,
               "xp_gemm: resid_side / out_side need both pointers, EPI_BIAS_RESID, an unmapped output, 0 < side_M <= side_S, M < 2^24");
  if (d->colsum_partials)
    XP_REQUIRE(xp_gemm_colsum_rows(d) > 0, "xp_gemm: fused column sums are not available for this problem "
               "(xp_gemm_colsum_rows() == 0): use xp_colsum / xp_colsum_partials");
  kp.tiles_m = (int)cdiv(d->M, BM); kp.tiles_n = (int)cdiv(d->N, BN);
  {
    const int gmax = 4;                   // L2 super-tile groups of <= 4 tile columns (A/B in round 2: profiles/r02_gemm256_ab_groupn_storepolicy.txt)
    const int ngroups = (int)cdiv(kp.tiles_n, gmax);
    kp.group_n = (int)cdiv(kp.tiles_n, ngroups);
    kp.xcd_remap = 1;
  }
  const int zsplits = (int)cdiv(d->K, kp.k_per_split);
  XP_REQUIRE(split == 1 || zsplits == split, "xp_gemm: split_k=%d leaves empty slabs for K=%lld (use <= %d)",
             split, (long long)d->K, zsplits);
  dim3 grid(kp.tiles_m * kp.tiles_n, 1, split);
  hipStream_t st = (hipStream_t)stream;
  const int tslot = gemm_timer_slot(d, split);                       // (debug: in-step timing of one shape)
  if (tslot >= 0) (void)hipEventRecord(g_timer.ev[tslot], st);
  struct Stop { int slot; hipStream_t st; ~Stop() { if (slot >= 0) (void)hipEventRecord(g_timer.ev[slot + 1], st); } } stop{tslot, st};
  if (xp_gemm256_try(d, kp, st)) {     // large dense problems: 256x256 ping-pong family
    XP_CHECK_LAUNCH("xp_gemm(256)");
    return XP_OK;
  }
  if (d->in_dtype == XP_BF16) launch<bf16_t>(d, kp, grid, st);
  else                        launch<float>(d, kp, grid, st);
  XP_CHECK_LAUNCH("xp_gemm");
  return XP_OK;
}

// fast epilogue (gemm_common.h): identity row map, 32-bit buffer offsets that cannot wrap for any row of the last tile
bool xp_gemm_fast_epi_ok(const XpGemmDesc* d) {
  if (xp_debug_flag("gemm_slow_epi")) return false;
  const int64_t esz = d->in_dtype == XP_BF16 ? 2 : 4, osz = d->out_dtype == XP_F32 ? 4 : esz, lim = (int64_t)EPI_OOB - 64;
  const int64_t rows = d->M + 256;
  const bool wide = d->N % 8 == 0 && d->ldc % 8 == 0 && (!d->resid || d->ldr % 8 == 0) && (!d->aux || d->ldaux % 8 == 0);
  return wide && d->c_grp == 0 && rows * d->ldc * osz < lim && (!d->resid || rows * d->ldr * esz < lim) &&
         (!d->aux || rows * d->ldaux * osz < lim);
}

extern "C" int64_t xp_gemm_colsum_rows(const XpGemmDesc* d) {
  if (!d || d->split_k > 1 || d->in_dtype != XP_BF16 || d->out_dtype != XP_BF16) return 0;
  if (d->epilogue != XP_EPI_NONE && d->epilogue != XP_EPI_GELU_BWD) return 0;
  if (!xp_gemm_fast_epi_ok(d) || !xp_gemm256_wanted(d, 1) || cdiv(d->K, 64) < 2) return 0;
  return xp_gemm256_colsum_rows(d);          // one partial row per wave row block of the tile height the launcher will pick
}

extern "C" int32_t xp_gemm_tile_rows(const XpGemmDesc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  const int split = d->split_k > 1 ? d->split_k : 1;
  const int64_t k_per = cdiv(cdiv(d->K, split), 64) * 64;
  if (xp_gemm256_wanted(d, split) && cdiv(d->K - (int64_t)(split - 1) * k_per, 64) >= 2 && (split == 1 || cdiv(d->K, k_per) == split))
    return 256;
  return BM;
}

// split s0 rounded to one xp_gemm accepts (whole k-steps per slab, no empty slab)
static int valid_split(int64_t K, int64_t s0, int64_t ke) {
  if (s0 <= 1) return 1;
  const int64_t kps = cdiv(cdiv(K, s0), ke) * ke;
  return (int)cdiv(K, kps);
}

// CUs the GEMM planning may count on (default: all 256).  A data-parallel run reserves the CUs its collective kernels occupy:
// RCCL's gfx950 kernels need one wave per SIMD with > 256 registers per lane, so each of their workgroups owns a CU for as long as
// a bucket all-reduce lasts (tools/contention_probe.py); a 252-workgroup dW launch would then need a second round for 12 tiles.
static int g_cu_budget = [] {
  const char* e = getenv("XPRETRAIN_CU_BUDGET");
  const int v = e ? atoi(e) : 256;
  return v >= 64 && v <= 256 ? v : 256;
}();
extern "C" int xp_set_cu_budget(int32_t cus) {
  XP_REQUIRE(cus >= 64 && cus <= 256, "xp_set_cu_budget: %d not in 64..256", cus);
  g_cu_budget = cus;
  return XP_OK;
}
extern "C" int32_t xp_get_cu_budget(void) { return g_cu_budget; }

// Split-K launches (the weight gradients) do not fill the chip: they run on the weight-gradient stream BESIDE the dX chain, which takes
// whatever CUs they leave, and every split costs an fp32 slab written and read again -- at a fill of 256 (splits 7 / 9 / 27 for the
// 36- / 27- / 9-tile outputs of ViT-B) the slabs of one layer are 260 MB each way, 6.2 GB per step, more than the optimizer moves.
//   XP_SPLITK_FILL        144 CUs (176 in round 5): the general answer (xp_gemm_auto_split): 4 / 5 / 16
//   XP_SPLITK_FILL_SLACK  112 CUs: launches nothing waits for soon (xp_gemm_auto_split_slack: fc2, fc1, out_proj of a layer): 3 / - / 12
// Whole step, interleaved: fill 256 -> 176 for all four: 15.55 -> 15.14 ms (208: 15.35, 144: 15.31, 128: 15.29;
// profiles/r05s_in_step_ab_splitk_fill.txt); per-GEMM search around it (profiles/r05x_in_step_ab_splitk_per_gemm.txt): the q/k/v
// gradient -- the last of the layer, the one the join waits for -- has a sharp optimum at 6 (4: +0.27 ms, 9: +0.24), fc1 / fc2 want 3
// (4: +0.10, 5: +0.20), out_proj is flat between 12 and 19 (27: +0.08): 4/6/19 -> 3/6/12 is another -0.16 ms.
// Round 6 (the fused attention backward shortened the dX chain, the balance of the two streams moved): larger fills lose (208 / 240:
// +0.02 ... +0.15 ms, profiles/r06g_*), the q/k/v gradient at 5 slabs instead of 6 (fill 144) is -0.10 ms, 4 (fill 112) level again
// (profiles/r06r_in_step_ab_splitk_fill_smaller.txt): 3 / 5 / 12.
constexpr int64_t XP_SPLITK_FILL = 144, XP_SPLITK_FILL_SLACK = 112;
// XPRETRAIN_SPLITK_FILL=general[xslack] (e.g. 208x144): the A/B switch of the two fills (tools/instep_ab.py)
static int64_t g_fill[2] = {XP_SPLITK_FILL, XP_SPLITK_FILL_SLACK};
static const bool g_fill_env = [] {
  const char* e = getenv("XPRETRAIN_SPLITK_FILL");
  if (!e) return false;
  int a = 0, b = 0;
  const int n = sscanf(e, "%d%*[,x]%d", &a, &b);
  if (n >= 1 && a >= 9 && a <= 256) g_fill[0] = a;
  if (n >= 2 && b >= 9 && b <= 256) g_fill[1] = b;
  return true;
}();
static int32_t auto_split_fill(const XpGemmDesc* d, int64_t fill);
extern "C" int32_t xp_gemm_auto_split(const XpGemmDesc* d) { return auto_split_fill(d, g_fill[0]); }
extern "C" int32_t xp_gemm_auto_split_slack(const XpGemmDesc* d) {
  // never MORE slabs than the general plan: when the 256-wide family refuses the smaller slack split, auto_split_fill falls into the
  // 128-family branch, whose answer does not know the fill
  const int32_t s = auto_split_fill(d, g_fill[1]), g = auto_split_fill(d, g_fill[0]);
  return s < g ? s : g;
}

static int32_t auto_split_fill(const XpGemmDesc* d, int64_t fill) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 1;
  const int esz = d->in_dtype == XP_BF16 ? 2 : 4;
  const int64_t ke = BKB / esz;
  const int64_t t256 = cdiv(d->M, 256) * cdiv(d->N, 256);
  const int64_t cus = g_cu_budget < fill ? g_cu_budget : fill;
  int64_t s0 = cus / t256 < d->K / 512 ? cus / t256 : d->K / 512;
  const int s256 = valid_split(d->K, s0, 64);
  if (xp_gemm256_wanted(d, s256)) return s256;
  const int64_t t128 = cdiv(d->M, BM) * cdiv(d->N, BN);
  if (t128 >= 256) return 1;
  s0 = 512 / t128 < d->K / 512 ? 512 / t128 : d->K / 512;
  return valid_split(d->K, s0, ke);
}

extern "C" int xp_splitk_reduce(const float* slabs, float* out, int64_t n, int32_t splits, int32_t accumulate,
                                void* stream) {
  XP_REQUIRE(slabs && out && n > 0 && n % 4 == 0 && splits >= 1, "xp_splitk_reduce: bad arguments");
  const int64_t n4 = n / 4;
  int blocks = (int)(cdiv(n4, 256) < 8192 ? cdiv(n4, 256) : 8192);
  splitk_reduce_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(slabs, out, n4, splits, accumulate);
  XP_CHECK_LAUNCH("xp_splitk_reduce");
  return XP_OK;
}

extern "C" int64_t xp_colsum_partial_rows(int64_t rows, int64_t cols) { return cdiv(rows, cs_rows(rows, cols)); }

extern "C" int xp_colsum_partials(const void* X, int64_t rows, int64_t cols, int64_t ldx, int32_t dtype, float* partials,
                                  size_t partials_bytes, void* stream) {
  XP_REQUIRE(X && partials && rows > 0 && cols > 0 && cols % 4 == 0 && ldx % 4 == 0, "xp_colsum_partials: bad arguments");
  const int csr = cs_rows(rows, cols), chunks = (int)cdiv(rows, csr);
  XP_REQUIRE(partials_bytes >= (size_t)chunks * cols * sizeof(float), "xp_colsum_partials: partials buffer too small");
  dim3 grid((unsigned)cdiv(cols, 256), chunks);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == XP_BF16) colsum_partial_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)X, rows, cols, ldx, partials, csr);
  else if (dtype == XP_F32) colsum_partial_kernel<float><<<grid, 256, 0, st>>>((const float*)X, rows, cols, ldx, partials, csr);
  else XP_REQUIRE(false, "xp_colsum_partials: bad dtype %d", dtype);
  XP_CHECK_LAUNCH("xp_colsum_partials");
  return XP_OK;
}

extern "C" size_t xp_reduce_rows_batch_workspace_bytes(const XpReduceSeg* segs_host, int32_t n) {
  size_t b = 0;
  for (int i = 0; segs_host && i < n; ++i) b += (size_t)32 * (size_t)(segs_host[i].width > 0 ? segs_host[i].width : 0) * sizeof(float);
  return b + 16;
}

extern "C" int xp_reduce_rows_batch(const XpReduceSeg* segs_host, int32_t n, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  XP_REQUIRE(segs_host && n > 0 && n <= XP_REDUCE_MAX_SEGS, "xp_reduce_rows_batch: n=%d not in 1..%d", n, XP_REDUCE_MAX_SEGS);
  XP_REQUIRE(workspace && workspace_bytes >= xp_reduce_rows_batch_workspace_bytes(segs_host, n), "xp_reduce_rows_batch: workspace too small");
  BatchArgs a;
  a.n = n;
  float* ws = (float*)workspace;
  // four columns per lane when every segment allows 16-byte accesses (widths, pitches and addresses multiples of 4 floats: every
  // segment the encoder layers pass); XPRETRAIN_DEBUG=rows_reduce_scalar keeps the one-column kernels (bit-identity test)
  bool vec = ((uintptr_t)workspace & 15) == 0 && !xp_debug_flag("rows_reduce_scalar");
  for (int i = 0; i < n && vec; ++i) {
    const XpReduceSeg& sg = segs_host[i];
    vec = sg.width % 4 == 0 && sg.stride % 4 == 0 && ((uintptr_t)sg.in & 15) == 0 && ((uintptr_t)sg.out & 15) == 0;
  }
  const int cw = vec ? 256 : 64;
  int cb = 0;
  bool any_l1 = false;
  for (int i = 0; i < n; ++i) {
    const XpReduceSeg& sg = segs_host[i];
    XP_REQUIRE(sg.in && sg.out && sg.nrows > 0 && sg.width > 0 && sg.stride >= sg.width, "xp_reduce_rows_batch: bad segment %d", i);
    a.seg[i] = sg; a.part2[i] = ws; a.cb0[i] = cb;
    ws += (size_t)32 * sg.width;
    cb += (int)cdiv(sg.width, cw);
    any_l1 = any_l1 || sg.nrows > RB_DIRECT;
  }
  for (int i = n; i <= XP_REDUCE_MAX_SEGS; ++i) a.cb0[i] = cb;
  hipStream_t st = (hipStream_t)stream;
  if (any_l1) {
    if (vec) reduce_batch4_l1_kernel<<<dim3((unsigned)cb, 32), 256, 0, st>>>(a);
    else     reduce_batch_l1_kernel<<<dim3((unsigned)cb, 32), 256, 0, st>>>(a);
    XP_CHECK_LAUNCH("xp_reduce_rows_batch(level 1)");
  }
  if (vec) reduce_batch4_l2_kernel<<<(unsigned)cb, 256, 0, st>>>(a);
  else     reduce_batch_l2_kernel<<<(unsigned)cb, 256, 0, st>>>(a);
  XP_CHECK_LAUNCH("xp_reduce_rows_batch(level 2)");
  return XP_OK;
}

extern "C" size_t xp_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  return (size_t)((cdiv(rows, 32) + 32) * cols * sizeof(float));
}

extern "C" int xp_colsum(const void* X, int64_t rows, int64_t cols, int64_t ldx, int32_t dtype, float* out,
                         int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  XP_REQUIRE(X && out && rows > 0 && cols > 0 && cols % 4 == 0 && ldx % 4 == 0, "xp_colsum: bad arguments");
  XP_REQUIRE(workspace && workspace_bytes >= xp_colsum_workspace_bytes(rows, cols), "xp_colsum: workspace too small");
  const int csr = cs_rows(rows, cols), chunks = (int)cdiv(rows, csr);
  dim3 grid((unsigned)cdiv(cols, 256), chunks);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == XP_BF16) colsum_partial_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)X, rows, cols, ldx, part, csr);
  else                  colsum_partial_kernel<float><<<grid, 256, 0, st>>>((const float*)X, rows, cols, ldx, part, csr);
  XP_CHECK_LAUNCH("xp_colsum(partial)");
  // two-level deterministic reduce of the chunk partials (chunks -> <=32 -> 1): no thread walks hundreds of rows
  const int lvl = (int)cdiv(chunks, 32), n2 = (int)cdiv(chunks, lvl);
  float* part2 = part + (int64_t)chunks * cols;
  rows_reduce_kernel<<<dim3((unsigned)cdiv(cols, 64), (unsigned)n2), 256, 0, st>>>(part, part2, chunks, lvl, (int)cols, 0);
  XP_CHECK_LAUNCH("xp_colsum(reduce1)");
  rows_reduce_kernel<<<dim3((unsigned)cdiv(cols, 64), 1), 256, 0, st>>>(part2, out, n2, n2, (int)cols, accumulate);
  XP_CHECK_LAUNCH("xp_colsum(reduce2)");
  return XP_OK;
}
