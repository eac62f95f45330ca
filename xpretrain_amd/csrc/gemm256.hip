// 256x256-tile MFMA GEMM, 8-wave ping-pong main loop: the kernel family of every video-tower GEMM (XPRETRAIN_GEMM256=0|1|2, see
// xp_gemm256_wanted).  ONE family since round 4: round 3's second set (direct LDS-free epilogue through an N-side row permutation,
// 224-row tiles, persistent tile loop) won isolated launches by 2-6 % and lost 0.5 ms per training step on one box
// (profiles/r03v..r03x) -- it was removed; the in-step calibration against the vendor library (profiles/r04a_vendor_library_in_step_calibration.txt)
// has this kernel within 10 % of the vendor's hand-written 256x256x64 stream-K kernel on the forward shapes, level on dX, 2-4x ahead on dW.
//
// Why 256x256 (measured with tools/gemm_trace.py, s_memtime stamps inside the 128x128 kernel): a 128x128x64 stage moves 32 KiB
// through the CU's texture-address path (64 B/clk) per 512 cycles of MFMA work per SIMD, i.e. the LDS-DMA path is as busy as the matrix
// pipe; a 256x256 tile halves the bytes staged per FLOP (64 KiB per 2048 MFMA cycles) and doubles the MFMAs per LDS fragment read.
//
//   workgroup  512 threads = 8 waves as 2 (M) x 4 (N); wave tile 128 x 64 = 8 x 4 accumulators (128 registers)
//   k-tile     64 bf16 of k = four 16 KiB HALF-TILES.  Half h of the M-side tile holds rows {wm*128 + h*64 + r} (r < 64) of both wave
//              rows, half h of the N-side tile the 32 columns {wn*64 + h*32 + ..} of all four wave columns.  One k-tile is two
//              PHASES of 32 MFMAs: A0 x (B0, B1), then A1 x (B0, B1); each phase reads its A fragments + 4 B fragments (B1, resp. the
//              NEXT k-tile's B0, which is kept in registers) and issues the DMA of two half-tiles.
//   ring       8 half-tile slots = 128 KiB LDS.  Consumption order j = 4*kt + w, w = A0(kt), B1(kt), A1(kt), B0(kt+1); slot j % 8.
//              Phase p reads half-tiles 2p, 2p+1 and issues 2p+4, 2p+5 (one k-tile ahead), then waits vmcnt(4): everything phase
//              p+1 reads has landed in every wave before the barrier that separates the phases.  The slots written in phase p were
//              last read in phase p-2 -- two barriers back even for the lagging wave group.
//   ping-pong  each phase is  [reads, DMA issue, vmcnt] barrier [MFMAs at raised priority] barrier ; the wm==1 waves run one
//              barrier behind the wm==0 waves, so on every SIMD one wave issues MFMAs while the other reads fragments and issues DMA.
//   epilogue   wave-private LDS staging (rounds of 32 rows x 64 cols fp32), row-major read-back: full 512-byte rows per store
//              instruction group, shared fused epilogue (gemm_common.h::FastEpi, incl. the fp32 side rows and the fused column sums).
//   split-K    (dW: k = the token dimension) 1-D grid over (k-chunk, tile), chunk-major per XCD: see the kernel.
//
// Tried and not kept (all correct; numbers under profiles/): v_mfma_f32_32x32x16_bf16 in the same loop (8 % fewer cycles per k-tile,
// 6-10 % slower in wall time: the chip is power-limited under MFMA load, tools/clock_probe.hip); 64-byte-row 4-stage ring; 256x128
// 3-stage ring; a coarse half-sub-step stagger; a dedicated DMA producer wave; the round-3 set named above.
//
// LDS images (XOR swizzles are applied to the per-lane GLOBAL source address of the lane-linear DMA):
//   k-contiguous half [128 rows][128 B]:  chunk' = chunk ^ (((row>>1)&3)<<1)
//   k-strided    half [64 k][256 B]:      32-byte block' = block ^ ((k&3) | ((k>>3)&1)<<2)   (ds_read_b64_tr_b16)
#include "common.h"
#include "gemm_common.h"
#include <stdlib.h>
#include <mutex>

namespace {

using namespace xpgemm;

typedef bf16_t T;
constexpr int TM = 256, TN = 256, SKB = 128, KE = 64;   // SKB: bytes of k per k-tile
constexpr int NTH = 512, NWAVES = 8;
constexpr int WAVES_N = 4, MT = 8, NT = 4;               // wave tile 128 x 64
constexpr int HALF_ROWS = 128, HALF_BYTES = HALF_ROWS * SKB, NSLOT = 8, LDS_BYTES = NSLOT * HALF_BYTES;

typedef __attribute__((address_space(3))) char lds_char;

// DMA of one operand's half-tiles.  SUB = rows of one wave in a half (64 on the M side, 32 on the N side):
// local row r of half h is tile row (r / SUB) * 2 * SUB + h * SUB + r % SUB.
template <bool KS, int SUB>
struct HalfStager {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[2][2];     // [half][pass]
  unsigned step;
  unsigned lds_off[2];

  static __device__ __forceinline__ int tile_row(int r, int h) { return (r / SUB) * (2 * SUB) + h * SUB + (r % SUB); }

  __device__ __forceinline__ void init(const T* base, int64_t ld, int64_t row0, int64_t rows, int64_t kend, int64_t kbeg,
                                       int lane, int wave) {
    const int64_t bytes = (KS ? kend : rows) * ld * 2;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (unsigned)bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pass = j * NWAVES + wave;        // 16 passes of 1 KiB per half-tile
      lds_off[j] = pass * 1024;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t off;
        if constexpr (!KS) {
          const int row = pass * 8 + (lane >> 3);
          const int c = (lane & 7) ^ swz128(row);
          off = ((row0 + tile_row(row, h)) * ld + kbeg) * 2 + c * 16;
        } else {
          const int kr = pass * 4 + (lane >> 4), c16 = lane & 15;
          const int src = (((c16 >> 1) ^ ks_f(kr)) << 1) | (c16 & 1);
          off = ((kbeg + kr) * ld + row0 + tile_row(src * 8, h)) * 2;
        }
        voff[h][j] = off >= bytes ? 0xFFFFFFF0u : (unsigned)off;
      }
    }
    step = (unsigned)((KS ? (int64_t)KE * ld : (int64_t)KE) * 2);
  }
  __device__ __forceinline__ void issue(char* slot, int h, int kt) const {
    lds_char* t3 = (lds_char*)slot;
    const unsigned adv = (unsigned)kt * step;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned o = voff[h][j] + adv;
      if (o < voff[h][j]) o = 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, t3 + lds_off[j], 16, o, 0, 0, 0);
    }
  }
};

// fragment of local 16-row sub-tile `ot` (0..7) of a half-tile for the 32-element k sub-step `ks` (0..1)
template <bool KS>
__device__ __forceinline__ bf16x8 frag(const char* tile, int ot, int ks, int lane) {
  constexpr int RB = HALF_ROWS * 2;
  const int i = lane & 15, g = lane >> 4;
  if constexpr (!KS) {
    return *reinterpret_cast<const bf16x8*>(tile + tile128_off(ot * 16 + i, ks * 4 + g));
  } else {
    const int f = (i >> 2) | ((g & 1) << 2);
    const int kr = ks * 32 + g * 8 + (i >> 2);
    const char* p = tile + kr * RB + ((ot ^ f) << 5) + ((i & 3) << 3);
    // asm reads (no compiler-inserted vmcnt(0) beside the DMA ring); consumed after XP_PHASE_MMA's s_waitcnt lgkmcnt(0)
    i16x4 lo = lds_read_tr16_async<0>(p);
    i16x4 hi = lds_read_tr16_async<4 * RB>(p);
    typedef __attribute__((ext_vector_type(8))) short i16x8;
    i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool AKS, bool BKS>
__global__ __launch_bounds__(NTH, 2) void gemm256_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int nwg = p.tiles_m * p.tiles_n;
  // Split-K launches (dW: k = the token dimension) with flat_split > 0 use a 1-D grid over (k-chunk, tile) pairs, CHUNK-MAJOR in
  // the XCD-contiguous order: an XCD's ~32 concurrent workgroups are (almost) all the tiles of ONE k-chunk, so its L2 fetches every
  // operand panel of that chunk once.  With the (tile, z) grid each XCD held ~4.5 tiles of EVERY chunk and each of the chunk's
  // panels was fetched by every XCD that touched it: 386 MB from the fabric for 145 MB of operands at dW1 (PMC, profiles/r03y).
  int bid, zs, nsplit;
  if (p.flat_split > 0) {
    const int l = xcd_remap(blockIdx.x, nwg * p.flat_split, 1);
    zs = l / nwg; bid = l - zs * nwg; nsplit = p.flat_split;
  } else {
    bid = xcd_remap(blockIdx.x, nwg, p.xcd_remap); zs = blockIdx.z; nsplit = gridDim.z;
  }
  int tm, tn;
  tile_of(bid, p.tiles_m, p.tiles_n, p.group_n, tm, tn);
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;

  const int64_t kbeg = (int64_t)zs * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nk = (int)((kend - kbeg + KE - 1) / KE);       // >= 2 (launcher)

  HalfStager<AKS, 64> ga;
  HalfStager<BKS, 32> gb;
  ga.init(reinterpret_cast<const T*>(p.A), p.lda, m0, p.M, kend, kbeg, lane, wave);
  gb.init(reinterpret_cast<const T*>(p.B), p.ldb, n0, p.N, kend, kbeg, lane, wave);

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool trace = p.dbg != nullptr && bid == nwg / 2 && zs == 0 && wave == 0;
  unsigned long long* tr = p.dbg;
  if (trace && lane == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[1] = nk; }
  // per-tile timeline (tools/gemm_timeline.py sets tr[4] = 0x7ace and provides 64 + 4 * tiles words): constant-rate 100 MHz stamps
  // at the start / the end of the main loop / the end of the epilogue (stores acknowledged) from wave 0 of EVERY workgroup
  const bool tline = p.dbg != nullptr && wave == 0 && lane == 0 && zs == 0 && tr[4] == 0x7aceull;
  if (tline) { tr[64 + bid * 4 + 0] = __builtin_amdgcn_s_memrealtime(); tr[64 + bid * 4 + 3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }

  // Half-tiles in consumption order: j = 4*kt + w with w 0: A0(kt), 1: B1(kt), 2: A1(kt), 3: B0(kt+1); slot j % 8.
  // B0 of k-tile 0 ("j = -1") uses slot 7.
  auto slot = [&](int kt, int w) -> char* { return smem + (((kt & 1) << 2) + w) * HALF_BYTES; };

  // prologue: B0(0) and half-tiles 0..3 in flight; B0(0), A0(0), B1(0) landed everywhere before the first phase
  gb.issue(slot(1, 3), 0, 0);
  ga.issue(slot(0, 0), 0, 0); gb.issue(slot(0, 1), 1, 0); ga.issue(slot(0, 2), 1, 0); gb.issue(slot(0, 3), 0, 1);
  wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();       // the wm==1 group runs one barrier behind

  bf16x8 fa[2][4], fb[2][2][2], fbn[2][2];          // fa[ks][mt] (current A half), fb[hB][ks][nt], fbn: next k-tile's B0

#define XP_PHASE_MMA(HA)                                                                              \
  do {                                                                                                \
    __builtin_amdgcn_s_barrier();                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int hb = 0; hb < 2; ++hb)                                                  \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                              \
          _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                            \
            acc[hb * 2 + nt][(HA) * 4 + mt] = mma16(fb[hb][ks][nt], fa[ks][mt], acc[hb * 2 + nt][(HA) * 4 + mt]); \
    __builtin_amdgcn_s_setprio(0);                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    __builtin_amdgcn_s_barrier();                                                                     \
  } while (0)
#define XP_READ_A(TILE)                                                                               \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) fa[ks][mt] = frag<AKS>(TILE, wm * 4 + mt, ks, lane)
#define XP_READ_B(DST, TILE)                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) DST[ks][nt] = frag<BKS>(TILE, wn * 2 + nt, ks, lane)

  XP_READ_B(fbn, slot(1, 3));                       // B0 of k-tile 0

  // One k-tile = 2 phases of 32 MFMAs: A0 x (B0, B1), then A1 x (B0, B1); 12 fragment reads and 2 half-tile DMAs each.
  // TAIL 0: steady state, 1: k-tile nk-2, 2: k-tile nk-1 (fewer half-tiles left to issue / await).
  auto ktile = [&](int t, auto tail_c) {
    constexpr int TAIL = decltype(tail_c)::value;
    // ---- phase 0: A0 x (B0, B1); issues A0, B1 of k-tile t+1; afterwards A1(t) and B0(t+1) have landed ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) fb[0][ks][nt] = fbn[ks][nt];
    XP_READ_B(fb[1], slot(t, 1));
    __builtin_amdgcn_sched_barrier(0);
    XP_READ_A(slot(t, 0));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAIL <= 1) { ga.issue(slot(t + 1, 0), 0, t + 1); gb.issue(slot(t + 1, 1), 1, t + 1); }
    wait_vmcnt<(TAIL <= 1 ? 4 : 0)>();
    XP_PHASE_MMA(0);
    // ---- phase 1: A1 x (B0, B1); issues A1(t+1), B0(t+2); afterwards A0, B1 of k-tile t+1 have landed ----
    if constexpr (TAIL <= 1) { XP_READ_B(fbn, slot(t, 3)); }
    __builtin_amdgcn_sched_barrier(0);
    XP_READ_A(slot(t, 2));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAIL <= 1) ga.issue(slot(t + 1, 2), 1, t + 1);
    if constexpr (TAIL == 0) gb.issue(slot(t + 1, 3), 0, t + 2);
    wait_vmcnt<(TAIL == 0 ? 4 : (TAIL == 1 ? 2 : 0))>();
    XP_PHASE_MMA(1);
  };
  for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
  ktile(nk - 2, std::integral_constant<int, 1>{});
  ktile(nk - 1, std::integral_constant<int, 2>{});
#undef XP_PHASE_MMA
#undef XP_READ_A
#undef XP_READ_B
  if (wm == 0) __builtin_amdgcn_s_barrier();       // re-align the two wave groups
  __builtin_amdgcn_s_barrier();                    // every wave is done reading the ring -> LDS is free for the epilogue
  if (trace && lane == 0) tr[2] = __builtin_amdgcn_s_memtime();
  if (tline) tr[64 + bid * 4 + 1] = __builtin_amdgcn_s_memrealtime();

  // ---- epilogue: wave-private staging, MT/2 rounds of 32 rows x 64 columns fp32 ------------------------------------
  constexpr int CW = NT * 16;
  char* stg = smem + wave * (32 * CW * 4);
  const int i16 = lane & 15, g = lane >> 4;
  float* Cf = reinterpret_cast<float*>(p.C);
  T* Ct = reinterpret_cast<T*>(p.C);
  if (nsplit > 1) Cf += (int64_t)zs * p.M * p.N;
  auto stage_round = [&](int q) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = h * 16 + i16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(stg + row * (CW * 4) + (((nt * 4 + g) ^ (row & 7)) << 4)) = acc[nt][q * 2 + h];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  const bool fast = fast_epi_dispatch(p, [&](auto epi_c, auto f32_c, auto cs_c) {
    constexpr int EPI = decltype(epi_c)::value;
    constexpr bool F32 = decltype(f32_c)::value;
    constexpr bool COLSUM = decltype(cs_c)::value;
    f32x8 cs = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int c = lane & 7, r8 = lane >> 3;           // 8 columns per lane: 8 lanes per row, 8 rows per pass
    const FastEpi<T, EPI, F32> fe(p, F32 ? (void*)Cf : (void*)Ct, n0 + wn * CW + c * 8);
    const unsigned mrow = (unsigned)(m0 + wm * (MT * 16)) + r8;
    Raw8<T> pre[2][4];
    if constexpr (EpiTraits<EPI>::pre) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) pre[0][pass] = fe.load_pre(mrow + pass * 8);
    }
#pragma unroll
    for (int q = 0; q < MT / 2; ++q) {
      if constexpr (EpiTraits<EPI>::pre) {
        if (q + 1 < MT / 2) {
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) pre[(q + 1) & 1][pass] = fe.load_pre(mrow + (q + 1) * 32 + pass * 8);
        }
      }
      stage_round(q);
      f32x8 v[4];
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + r8;
        v[pass].lo = *reinterpret_cast<const f32x4*>(stg + row * (CW * 4) + (((2 * c) ^ (row & 7)) << 4));
        v[pass].hi = *reinterpret_cast<const f32x4*>(stg + row * (CW * 4) + (((2 * c + 1) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const f32x8 o = fe.finish(v[pass], pre[q & 1][pass], mrow + q * 32 + pass * 8);
        if constexpr (COLSUM) {
          if (mrow + q * 32 + pass * 8 < (unsigned)p.M) { cs.lo += o.lo; cs.hi += o.hi; }     // rows >= M are not outputs
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (COLSUM) {
      // the 8 lanes r8 = 0..7 of a column octet hold different rows: butterfly over lane bits 3..5, then lane r8 == 0
      // writes this wave's 128-row column sums (one partial row per (tile row, wm))
#pragma unroll
      for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs.lo[e] += __shfl_xor(cs.lo[e], o, 64); cs.hi[e] += __shfl_xor(cs.hi[e], o, 64); }
      const int64_t n = n0 + wn * CW + c * 8;
      if (r8 == 0 && n < p.N) {
        float* dst = p.colsum + ((int64_t)tm * 2 + wm) * p.N + n;
        store4(dst, cs.lo); store4(dst + 4, cs.hi);
      }
    }
  });
  (void)fast;          // the launcher admits only (epilogue, output type) pairs the fast path specialises (epi_supported)
  if (trace && lane == 0) tr[3] = __builtin_amdgcn_s_memtime();
  if (tline) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[64 + bid * 4 + 2] = __builtin_amdgcn_s_memrealtime(); }
}

template <bool AKS, bool BKS>
bool launch_one(const KParams& kp, dim3 grid, hipStream_t st) {
  auto kern = gemm256_kernel<AKS, BKS>;
  // the 128 KiB dynamic-LDS opt-in is per device: configured once for every device this process launches on
  // (forward thread and autograd thread may both arrive first)
  static std::mutex mu;
  static int configured[64] = {0};                  // 0: not yet, 1: ok, -1: refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!configured[dev])
      configured[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            LDS_BYTES) == hipSuccess ? 1 : -1;
    if (configured[dev] < 0) return false;          // (a device without 128 KiB of LDS per workgroup: the 128x128 family serves it)
  }
  kern<<<grid, NTH, LDS_BYTES, st>>>(kp);
  return true;
}

// explicit instantiations (hipcc otherwise drops the host stubs of the k-strided variants)
template __global__ void gemm256_kernel<false, false>(KParams);
template __global__ void gemm256_kernel<false, true>(KParams);
template __global__ void gemm256_kernel<true, true>(KParams);
template __global__ void gemm256_kernel<true, false>(KParams);

}  // namespace

// the (epilogue, output type, column sums) combinations the fast epilogue is specialised for (fast_epi_dispatch); everything
// else goes to the 128x128 family
static bool epi_supported(const XpGemmDesc* d) {
  if (!xp_gemm_fast_epi_ok(d)) return false;
  const bool f32 = d->out_dtype == XP_F32;
  if (d->colsum_partials) return !f32 && (d->epilogue == XP_EPI_NONE || d->epilogue == XP_EPI_GELU_BWD);
  if (f32) return d->epilogue == XP_EPI_NONE;
  switch (d->epilogue) {
    case XP_EPI_NONE: case XP_EPI_BIAS: case XP_EPI_BIAS_QSCALE: case XP_EPI_BIAS_GELU: case XP_EPI_BIAS_RESID: case XP_EPI_GELU_BWD:
      return true;
    default: return false;
  }
}

// preconditions of the family that do not depend on split_k
bool xp_gemm256_legal(const XpGemmDesc* d) {
  if (d->in_dtype != XP_BF16 || d->a_grp != 0 || d->a_frames) return false;      // (a_frames: the register-staged 128x128 loader gathers)
  const int64_t a_rows = d->a_kstrided ? d->K : d->M, b_rows = d->b_kstrided ? d->K : d->N;
  if (!d->a_kstrided && (d->K % KE != 0 || d->lda != d->K)) return false;
  if (!d->b_kstrided && (d->K % KE != 0 || d->ldb != d->K)) return false;
  if (d->a_kstrided && (d->M % TM != 0 || d->lda != d->M)) return false;
  if (d->b_kstrided && (d->N % TN != 0 || d->ldb != d->N)) return false;
  const int64_t lim = (int64_t)0xFFFFFFF0u - 512 * 1024 * 1024;
  if ((a_rows + TM) * d->lda * 2 >= lim || (b_rows + TN) * d->ldb * 2 >= lim) return false;
  return epi_supported(d);
}

// XPRETRAIN_GEMM256: 0 = never, 1 = from 96 workgroups (default: a half-batch N = 768 GEMM of the forward's two chains is 111 tiles and
// runs beside its twin), 2 = whenever legal.
bool xp_gemm256_wanted(const XpGemmDesc* d, int split) {
  const char* env = getenv("XPRETRAIN_GEMM256");
  const int mode = env ? atoi(env) : 1;
  if (mode == 0 || !xp_gemm256_legal(d)) return false;
  if (mode == 1 && cdiv(d->M, TM) * cdiv(d->N, TN) * split < 96) return false;
  return true;
}

// Tile columns per L2 group (gemm_common.h::tile_of walks column groups, columns fastest inside a group; the XCD-contiguous id
// ranges then make an XCD own (a row range of) ONE column group).  The weight-side panels an XCD's ~32 resident workgroups sweep in
// one k pass must stay in its 4 MiB L2 beside the streaming activation panels: with all 12 tile columns of the N = 3072 problems
// (4.7 MB of weights) every XCD re-streamed the whole weight matrix once per round -- 194 MB fetched for 34 MB of unique operands
// (PMC, profiles/r04s_pmc_gemm256.json).  Rule: the fewest column groups (1, 2 or 4 -- they must tile the 8 XCDs) whose weight
// share is <= 3.6 MB, groups of at least 4 columns (the K = 3072 problems with 3 tile columns lose with a 2 + 1 split: their
// concurrent tiles move through k together, the live weight window is small): two groups of 6 columns at N = 3072 (2.4 MB resident
// per XCD, activations fetched twice), one group everywhere else.  In the step: -0.07 ms against one group, level with four
// (profiles/r05k_in_step_ab_l2_column_groups.txt); fabric fetch of fc1: profiles/r05l_pmc_gemm256.json.
int xp_gemm256_group_n(const XpGemmDesc* d, int tiles_n) {
  int groups = 1;
  const int64_t wbytes = (int64_t)tiles_n * TN * d->K * (d->in_dtype == XP_BF16 ? 2 : 4);      // (element size of the operands: ADVICE r5)
  while (groups < 4 && wbytes / groups > (int64_t)3600 * 1024 && tiles_n / (groups * 2) >= 4) groups *= 2;
  return (int)cdiv(tiles_n, groups);
}

// fused column sums: one partial row per wave row block (128 rows)
int64_t xp_gemm256_colsum_rows(const XpGemmDesc* d) { return 2 * cdiv(d->M, TM); }

bool xp_gemm256_try(const XpGemmDesc* d, const xpgemm::KParams& kp_base, hipStream_t st) {
  const int split = d->split_k > 1 ? d->split_k : 1;
  if (!xp_gemm256_wanted(d, split)) return false;
  xpgemm::KParams kp = kp_base;
  kp.k_per_split = cdiv(cdiv(d->K, split), KE) * KE;
  if (split > 1 && cdiv(d->K, kp.k_per_split) != split) return false;
  const int64_t k_last = d->K - (int64_t)(split - 1) * kp.k_per_split;
  if (cdiv(k_last, KE) < 2) return false;                                         // the pipeline needs >= 2 k-tiles
  kp.tiles_m = (int)cdiv(d->M, TM); kp.tiles_n = (int)cdiv(d->N, TN);
  kp.group_n = split > 1 ? kp.tiles_n : xp_gemm256_group_n(d, kp.tiles_n);
  const bool chunk_major = !xp_debug_flag("dw_tile_major");      // (test facility: the (tile, z) grid of rounds 1-3, bit-identical slabs)
  kp.flat_split = (split > 1 && chunk_major) ? split : 0;
  dim3 grid(kp.tiles_m * kp.tiles_n * (kp.flat_split ? split : 1), 1, kp.flat_split ? 1 : split);
  if (!d->a_kstrided && !d->b_kstrided)      return launch_one<false, false>(kp, grid, st);
  else if (!d->a_kstrided && d->b_kstrided)  return launch_one<false, true>(kp, grid, st);
  else if (d->a_kstrided && d->b_kstrided)   return launch_one<true, true>(kp, grid, st);
  return launch_one<true, false>(kp, grid, st);
}
