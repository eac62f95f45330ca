// 256-wide-tile MFMA GEMM, 8-wave ping-pong main loop (second kernel family; XPRETRAIN_GEMM256=0|1|2, see xp_gemm256_wanted).
//
// Why a second family (measured on MI355X with tools/gemm_trace.py, s_memtime stamps inside the 128x128 kernel): a
// 128x128x64 stage moves 32 KiB through the CU's texture-address path (64 B/clk) per 512 cycles of MFMA work per SIMD,
// i.e. the LDS-DMA path is as busy as the matrix pipe and one k-iteration costs 2200-3000 cycles.  A 256x256 tile halves
// the bytes staged per FLOP (64 KiB per 2048 MFMA cycles) and doubles the MFMAs per LDS fragment read.
//
//   workgroup  512 threads = 8 waves as 2 (M) x 4 (N); wave tile (64 + 16*MT1) x 64 = (4 + MT1) x 4 accumulators
//   tile       TM x 256 with TM = 2 * (64 + 16 * MT1): 256 (MT1 = 4), 224 (MT1 = 3), 192 (MT1 = 2).  18848 rows are 73.6 tiles of
//              256 -- 74 x {3, 9, 12} = 222 / 666 / 888 workgroups fill 0.87 of 1 / 3 / 4 rounds of the 256 CUs -- but 85 tiles of
//              224: 255 / 765 / 1020 = 0.996 of 1 / 3 / 4 rounds.  Which one is used is a policy (xp_gemm256_mt1 below): 224 for
//              latency-first forward passes, 256 where energy per step counts.
//   k-tile     64 bf16 of k = four 16 KiB HALF-TILES.  Half 0 of the M-side tile holds rows {wm*WR + r} (r < 64), half 1 rows
//              {wm*WR + 64 + r} (r < 16*MT1; the remaining LDS rows are zero-filled by the DMA's bounds check and never
//              multiplied) of both wave rows (WR = 64 + 16*MT1); half h of the N-side tile the 32 columns {wn*64 + h*32 + ..}
//              of all four wave columns.  One k-tile is two PHASES: A0 x (B0, B1) = 32 MFMAs, then A1 x (B0, B1) = 8*MT1
//              MFMAs; each phase reads its A fragments + 4 B fragments (B1, resp. the NEXT k-tile's B0, which is kept in
//              registers) and issues the DMA of two half-tiles.
//   ring       8 half-tile slots = 128 KiB LDS.  Consumption order j = 4*kt + w, w = A0(kt), B1(kt), A1(kt), B0(kt+1);
//              slot j % 8.  Phase p reads half-tiles 2p, 2p+1 and issues 2p+4, 2p+5 (one k-tile ahead), then waits
//              vmcnt(4): everything phase p+1 reads has landed in every wave before the barrier that separates the
//              phases.  The slots written in phase p were last read in phase p-2 -- two barriers back even for the
//              lagging wave group, which is what the staggered schedule needs (a deeper look-ahead would not be).
//   ping-pong  each phase is  [reads, DMA issue, vmcnt] barrier [MFMAs at raised priority] barrier ; the wm==1 waves
//              run one barrier behind the wm==0 waves (one extra s_barrier up front, one extra for wm==0 at the end),
//              so on every SIMD one wave issues MFMAs while the other reads fragments and issues DMA.
//   epilogue   DIRECT, no LDS: the N-side rows are fed to the MFMAs in a permuted order -- MFMA row i of 16-column block nt'
//              of half hb is tile column wn*64 + hb*32 + (i>>2)*8 + nt'*4 + (i&3) -- so that an accumulator lane (i16, g) holds,
//              for each of its rows, the 8 CONSECUTIVE columns hb*32 + g*8 .. +7 in {acc[2hb][mt], acc[2hb+1][mt]}: one 16-byte
//              bf16 store (two for fp32) per (mt, hb), the four g-lanes of a row covering 64 contiguous bytes.  The permutation
//              costs nothing in the loop: k-contiguous N-side operands are permuted by the DMA's per-lane SOURCE row (LDS image
//              rows in MFMA order, reads unchanged), k-strided ones by the per-lane address of the transpose read (2-way bank
//              conflict on 8 of the 24 fragment reads of a k-tile -- the LDS array is < 50 % busy).  Round 2 staged the raw
//              fp32 tile through LDS (256 KiB per tile at the 64-85 B/clk LDS WRITE rate = most of its 4.5k-cycle epilogue).
//
// Round 3, tried and NOT kept (profiles/r03c_gemm256_ab_mfma_shape_tile_height.txt): the same loop on v_mfma_f32_32x32x16_bf16
// (16 MFMAs of 32 cycles per phase instead of 32 of >= 17, rows permuted so that a lane owns 2 x 8 consecutive columns per block).
// Correct, 8 % FEWER shader cycles per k-tile (2553 vs 2772 in the stamped builds) and 6-10 % SLOWER in wall time: the kernel runs
// against the chip's power limit (tools/clock_probe.hip: 1.81 GHz with every SIMD issuing MFMAs on random operands, 2.38 GHz on
// zeros), so cycles saved come back as clock (1.4-1.6 GHz under this variant vs 1.6-1.8), and the 32x32 form moves twice the
// accumulator registers per FLOP.
//
// History of this file (all variants were correct; numbers at BASELINE cfg #2 shapes): a plain 2-stage 256x256 loop
// (vmcnt(0) + barrier per 64 KiB stage) reached 903 TFLOP/s at K=3072 but 430-490 at K=768; 64-byte-row 4-stage ring,
// 256x128 3-stage ring with counted vmcnt, and a coarse half-sub-step stagger were all <= the 128x128 family.
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
constexpr int TN = 256, SKB = 128, KE = 64;              // SKB: bytes of k per k-tile
constexpr int NTH = 512, NWAVES = 8;
constexpr int WAVES_N = 4, NT = 4;                       // wave tile (64 + 16*MT1) x 64
constexpr int HALF_ROWS = 128, HALF_BYTES = HALF_ROWS * SKB, NSLOT = 8, LDS_BYTES = NSLOT * HALF_BYTES;

typedef __attribute__((address_space(3))) char lds_char;

// column of the N-side half (0..31 within the wave's 32) that sits at MFMA position (16-block ntp, row i)
__device__ __forceinline__ int perm_n(int ntp, int i) { return ((i >> 2) << 3) + (ntp << 2) + (i & 3); }

// DMA of one operand's half-tiles.  SUB = rows of one wave in half 0 (64 on the M side, 32 on the N side), SUB1 = rows of
// one wave in half 1 (<= SUB; the rest of the half is zero-filled), PERM: LDS row order = MFMA order of the N side.
// local row r of half h is tile row (r / SUB) * (SUB + SUB1) + h * SUB + r % SUB.
template <bool KS, int SUB, int SUB1, bool PERM>
struct HalfStager {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[2][2];     // [half][pass]
  unsigned step;
  unsigned lds_off[2];

  static __device__ __forceinline__ int tile_row(int r, int h) {
    int q = r % SUB;
    if constexpr (PERM && !KS) q = perm_n(q >> 4, q & 15);      // (SUB == 32: 16-block q >> 4, MFMA row q & 15)
    return (r / SUB) * (SUB + SUB1) + h * SUB + q;
  }
  static __device__ __forceinline__ bool valid(int r, int h) { return SUB1 == SUB || h == 0 || (r % SUB) < SUB1; }

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
        bool ok;
        if constexpr (!KS) {
          const int row = pass * 8 + (lane >> 3);
          const int c = (lane & 7) ^ swz128(row);
          off = ((row0 + tile_row(row, h)) * ld + kbeg) * 2 + c * 16;
          ok = valid(row, h);
        } else {
          const int kr = pass * 4 + (lane >> 4), c16 = lane & 15;
          const int src = (((c16 >> 1) ^ ks_f(kr)) << 1) | (c16 & 1);
          off = ((kbeg + kr) * ld + row0 + tile_row(src * 8, h)) * 2;
          ok = valid(src * 8, h);
        }
        voff[h][j] = (off >= bytes || !ok) ? 0xFFFFFFF0u : (unsigned)off;
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

// M side: fragment of local 16-row sub-tile `ot` (0..7) of a half-tile for the 32-element k sub-step `ks` (0..1)
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

// N side: fragment of MFMA 16-block ntp (0..1) of wave column wn.  k-contiguous: the LDS image is already in MFMA row order
// (HalfStager PERM).  k-strided: the image is in natural column order and the transpose read gathers the permuted columns:
// supplying lane (j = i>>2, q = i&3) of a 16-lane group hands k-row j the 4 columns q*8 + ntp*4 .. +3 of the wave's 32.
template <bool KS>
__device__ __forceinline__ bf16x8 frag_n(const char* tile, int wn, int ntp, int ks, int lane) {
  if constexpr (!KS) {
    return frag<false>(tile, wn * 2 + ntp, ks, lane);
  } else {
    constexpr int RB = HALF_ROWS * 2;
    const int i = lane & 15, g = lane >> 4, q = i & 3;
    const int f = (i >> 2) | ((g & 1) << 2);
    const int kr = ks * 32 + g * 8 + (i >> 2);
    const char* p = tile + kr * RB + (((wn * 2 + (q >> 1)) ^ f) << 5) + ((q & 1) << 4) + (ntp << 3);
    i16x4 lo = lds_read_tr16_async<0>(p);
    i16x4 hi = lds_read_tr16_async<4 * RB>(p);
    typedef __attribute__((ext_vector_type(8))) short i16x8;
    i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- direct epilogue ---------------------------------------------------------------------------------------------------
// Per-wave state; the lane owns, of every row it stores, the columns col[hb] .. col[hb]+7 (hb = 0, 1).  Buffer-addressed
// accesses: the hardware bounds check drops rows >= M, lanes past N get an out-of-range offset.  Straight-line per epilogue
// kind (a run-time switch inside the unrolled passes makes hipcc wait vmcnt(0) for the previous pass's STORES, gemm_common.h).
template <int EPI, bool F32>
struct DirectEpi {
  static constexpr int NG = 2, GSTRIDE = 32;      // column groups of 8 per lane, their distance
  using Tr = EpiTraits<EPI>;
  static constexpr unsigned OSZ = F32 ? 4 : sizeof(T);
  __amdgpu_buffer_rsrc_t rc, rx;
  f32x8 bias[NG]; float cs[NG][2];
  unsigned ld_c, ld_x;          // row pitch in bytes
  unsigned col_c[NG], col_x[NG];  // byte offset of the first column of the lane's column group, or EPI_OOB
  bool keep_aux;
  SideRows side; int64_t ncol;    // fp32 side rows of the residual stream (gemm_common.h), first column of group 0
  __device__ __forceinline__ DirectEpi(const KParams& p, void* Cbase, int64_t n) : side(p), ncol(n) {
    keep_aux = p.aux != nullptr;             // BIAS_GELU without aux: forward-only, the pre-activation is not stored
    rc = __builtin_amdgcn_make_buffer_rsrc(Cbase, 0, (unsigned)(p.M * p.ldc * OSZ), 0x00020000);
    ld_c = (unsigned)(p.ldc * OSZ);
    if constexpr (EPI == XP_EPI_BIAS_GELU) {
      rx = __builtin_amdgcn_make_buffer_rsrc(p.aux, 0, (unsigned)(p.M * p.ldaux * OSZ), 0x00020000);
      ld_x = (unsigned)(p.ldaux * OSZ);
    } else if constexpr (Tr::pre) {
      rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.resid), 0, (unsigned)(p.M * p.ldr * sizeof(T)), 0x00020000);
      ld_x = (unsigned)(p.ldr * sizeof(T));
    } else { rx = rc; ld_x = 0; }
#pragma unroll
    for (int hb = 0; hb < NG; ++hb) {
      const int64_t nn = n + hb * GSTRIDE;
      const bool ok = nn < p.N;
      const int64_t ns = ok ? nn : 0;
      col_c[hb] = ok ? (unsigned)(ns * OSZ) : EPI_OOB;
      col_x[hb] = ok ? (unsigned)(ns * (EPI == XP_EPI_BIAS_GELU ? OSZ : sizeof(T))) : EPI_OOB;
      if constexpr (Tr::bias) bias[hb] = f32x8{load4(p.bias + ns), load4(p.bias + ns + 4)};
      if constexpr (EPI == XP_EPI_SCALE) cs[hb][0] = cs[hb][1] = p.scale;
      if constexpr (EPI == XP_EPI_BIAS_QSCALE) { cs[hb][0] = ns < p.scale_cols ? p.scale : 1.f; cs[hb][1] = ns + 4 < p.scale_cols ? p.scale : 1.f; }
    }
  }
  // rows m < M have m * pitch < 4 GiB (xp_gemm_fast_epi_ok); EPI_OOB + m * pitch may wrap for the out-of-range lanes, so
  // their offset is pinned instead of added to
  __device__ __forceinline__ unsigned off_c(unsigned m, int hb) const { return col_c[hb] == EPI_OOB ? EPI_OOB : m * ld_c + col_c[hb]; }
  __device__ __forceinline__ unsigned off_x(unsigned m, int hb) const { return col_x[hb] == EPI_OOB ? EPI_OOB : m * ld_x + col_x[hb]; }
  __device__ __forceinline__ Raw8<T> load_pre(unsigned m, int hb) const { return bload8<T>(rx, off_x(m, hb)); }
  __device__ __forceinline__ f32x8 finish(f32x8 v, const Raw8<T>& pre, unsigned m, int hb) const {
    if constexpr (Tr::bias) { v.lo += bias[hb].lo; v.hi += bias[hb].hi; }
    if constexpr (Tr::scale) { v.lo *= cs[hb][0]; v.hi *= cs[hb][1]; }
    if constexpr (EPI == XP_EPI_BIAS_GELU) {
      if (keep_aux) bstore8<T, F32>(rx, off_x(m, hb), v);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v.lo[e] = quick_gelu_f(v.lo[e]); v.hi[e] = quick_gelu_f(v.hi[e]); }
    } else if constexpr (EPI == XP_EPI_BIAS_RESID) {
      int64_t sb;
      if (side.on() && col_c[hb] != EPI_OOB && side.hit(m, sb)) {   // fp32 side row (rare, divergent): fp32 residual operand + result
        const int64_t n = ncol + hb * GSTRIDE;
        const f32x8 r = load8(side.rin + sb + n);
        v.lo += r.lo; v.hi += r.hi;
        store8(side.out + sb + n, v);
      } else {
        const f32x8 r = raw8_f32<T>(pre);
        v.lo += r.lo; v.hi += r.hi;
      }
    } else if constexpr (EPI == XP_EPI_GELU_BWD) {
      const f32x8 r = raw8_f32<T>(pre);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v.lo[e] *= quick_gelu_grad_f(r.lo[e]); v.hi[e] *= quick_gelu_grad_f(r.hi[e]); }
    }
    bstore8<T, F32>(rc, off_c(m, hb), v);
    return v;
  }
};

// The k-loop of one output tile: on entry B0(0) and half-tiles 0..3 are in flight with B0(0), A0(0), B1(0) landed in every wave
// (vmcnt + barrier), and the wm == 1 waves one barrier behind; on exit every DMA has landed and every fragment has been read.
template <bool AKS, bool BKS, int MT1, bool TRACE, typename GA, typename GB, int MTn>
__device__ __forceinline__ void main_loop(const GA& ga, const GB& gb, f32x4 (&acc)[NT][MTn], char* smem, int nk, int wm, int wn,
                                          int lane, unsigned long long (&stamp)[16], int tk) {
  auto slot = [&](int kt, int w) -> char* { return smem + (((kt & 1) << 2) + w) * HALF_BYTES; };
  bf16x8 fa[2][4], fb[2][2][2], fbn[2][2];          // fa[ks][mt] (current A half), fb[hB][ks][nt], fbn: next k-tile's B0

#define XP_STAMP(S, T_)                                                                               \
  do { if constexpr (TRACE) { if ((T_) == tk) stamp[S] = __builtin_amdgcn_s_memtime(); } } while (0)
#define XP_PHASE_MMA(HA, NMT, T_)                                                                     \
  do {                                                                                                \
    XP_STAMP((HA) * 4 + 0, T_);                                                                       \
    __builtin_amdgcn_s_barrier();                                                                     \
    XP_STAMP((HA) * 4 + 1, T_);                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int hb = 0; hb < 2; ++hb)                                                  \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
        _Pragma("unroll") for (int mt = 0; mt < (NMT); ++mt)                                          \
          _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                            \
            acc[hb * 2 + nt][(HA) * 4 + mt] = mma16(fb[hb][ks][nt], fa[ks][mt], acc[hb * 2 + nt][(HA) * 4 + mt]); \
    __builtin_amdgcn_s_setprio(0);                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    XP_STAMP((HA) * 4 + 2, T_);                                                                       \
    __builtin_amdgcn_s_barrier();                                                                     \
    XP_STAMP((HA) * 4 + 3, T_);                                                                       \
  } while (0)
#define XP_READ_A(TILE, NMT)                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
    _Pragma("unroll") for (int mt = 0; mt < (NMT); ++mt) fa[ks][mt] = frag<AKS>(TILE, wm * 4 + mt, ks, lane)
#define XP_READ_B(DST, TILE)                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) DST[ks][nt] = frag_n<BKS>(TILE, wn, nt, ks, lane)

  XP_READ_B(fbn, slot(1, 3));                       // B0 of k-tile 0

  // One k-tile = 2 phases: A0 x (B0, B1) (32 MFMAs), then A1 x (B0, B1) (8*MT1); 8 + 4 / 2*MT1 + 4 fragment reads and 2
  // half-tile DMAs each.  TAIL 0: steady state, 1: k-tile nk-2, 2: k-tile nk-1 (fewer half-tiles left to issue / await).
  auto ktile = [&](int t, auto tail_c) {
    constexpr int TAIL = decltype(tail_c)::value;
    // ---- phase 0: A0 x (B0, B1); issues A0, B1 of k-tile t+1; afterwards A1(t) and B0(t+1) have landed ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) fb[0][ks][nt] = fbn[ks][nt];
    XP_READ_B(fb[1], slot(t, 1));
    __builtin_amdgcn_sched_barrier(0);
    XP_READ_A(slot(t, 0), 4);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAIL <= 1) { ga.issue(slot(t + 1, 0), 0, t + 1); gb.issue(slot(t + 1, 1), 1, t + 1); }
    wait_vmcnt<(TAIL <= 1 ? 4 : 0)>();
    XP_PHASE_MMA(0, 4, t);
    // ---- phase 1: A1 x (B0, B1); issues A1(t+1), B0(t+2); afterwards A0, B1 of k-tile t+1 have landed ----
    if constexpr (TAIL <= 1) { XP_READ_B(fbn, slot(t, 3)); }
    __builtin_amdgcn_sched_barrier(0);
    XP_READ_A(slot(t, 2), MT1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAIL <= 1) ga.issue(slot(t + 1, 2), 1, t + 1);
    if constexpr (TAIL == 0) gb.issue(slot(t + 1, 3), 0, t + 2);
    wait_vmcnt<(TAIL == 0 ? 4 : (TAIL == 1 ? 2 : 0))>();
    XP_PHASE_MMA(1, MT1, t);
  };
  for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
  ktile(nk - 2, std::integral_constant<int, 1>{});
  ktile(nk - 1, std::integral_constant<int, 2>{});
#undef XP_PHASE_MMA
#undef XP_READ_A
#undef XP_READ_B
#undef XP_STAMP
}

template <bool AKS, bool BKS, int MT1, bool TRACE>
__global__ __launch_bounds__(NTH, 2) void gemm256_kernel(KParams p) {
  constexpr int MT = 4 + MT1, WR = MT * 16, TMv = 2 * WR;     // sub-tiles per wave, rows per wave, rows per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg, p.xcd_remap);
  int tm, tn;
  tile_of(bid, p.tiles_m, p.tiles_n, p.group_n, tm, tn);
  const int64_t m0 = (int64_t)tm * TMv, n0 = (int64_t)tn * TN;

  const int64_t kbeg = (int64_t)blockIdx.z * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nk = (int)((kend - kbeg + KE - 1) / KE);       // >= 2 (launcher)

  HalfStager<AKS, 64, 16 * MT1, false> ga;
  HalfStager<BKS, 32, 32, true> gb;
  ga.init(reinterpret_cast<const T*>(p.A), p.lda, m0, p.M, kend, kbeg, lane, wave);
  gb.init(reinterpret_cast<const T*>(p.B), p.ldb, n0, p.N, kend, kbeg, lane, wave);

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool trace = p.dbg != nullptr && (int)blockIdx.x == nwg / 2 && blockIdx.z == 0 && (wave & 3) == 0;
  unsigned long long* tr = p.dbg;
  unsigned long long stamp[16];
  const int tk = nk / 2;                                   // the k-tile whose barriers the TRACE build stamps
  if constexpr (TRACE) {
#pragma unroll
    for (int s = 0; s < 16; ++s) stamp[s] = 0;
  }
  if (trace && wave == 0 && lane == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[1] = nk; }

  // Half-tiles in consumption order: j = 4*kt + w with w 0: A0(kt), 1: B1(kt), 2: A1(kt), 3: B0(kt+1); slot j % 8.
  // B0 of k-tile 0 ("j = -1") uses slot 7.
  auto slot = [&](int kt, int w) -> char* { return smem + (((kt & 1) << 2) + w) * HALF_BYTES; };

  // prologue: B0(0) and half-tiles 0..3 in flight; B0(0), A0(0), B1(0) landed everywhere before the first phase
  gb.issue(slot(1, 3), 0, 0);
  ga.issue(slot(0, 0), 0, 0); gb.issue(slot(0, 1), 1, 0); ga.issue(slot(0, 2), 1, 0); gb.issue(slot(0, 3), 0, 1);
  wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();       // the wm==1 group runs one barrier behind

  main_loop<AKS, BKS, MT1, TRACE>(ga, gb, acc, smem, nk, wm, wn, lane, stamp, tk);
  if (wm == 0) __builtin_amdgcn_s_barrier();       // both wave groups execute the same number of barriers
  if (trace && wave == 0 && lane == 0) tr[2] = __builtin_amdgcn_s_memtime();

  // ---- epilogue: straight from the accumulators (no LDS, no barrier) ----------------------------------------------------
  const int i16 = lane & 15, g = lane >> 4;
  float* Cf = reinterpret_cast<float*>(p.C);
  T* Ct = reinterpret_cast<T*>(p.C);
  if (gridDim.z > 1) Cf += (int64_t)blockIdx.z * p.M * p.N;
  fast_epi_dispatch(p, [&](auto epi_c, auto f32_c, auto cs_c) {
    constexpr int EPI = decltype(epi_c)::value;
    constexpr bool F32 = decltype(f32_c)::value;
    constexpr bool COLSUM = decltype(cs_c)::value;
    const DirectEpi<EPI, F32> de(p, F32 ? (void*)Cf : (void*)Ct, n0 + wn * 64 + g * 8);
    const unsigned mrow = (unsigned)(m0 + wm * WR) + i16;
    f32x8 cs[2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
    Raw8<T> pre[2][2];                                // side input, one sub-tile ahead of the stores
    if constexpr (EpiTraits<EPI>::pre) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) pre[0][hb] = de.load_pre(mrow, hb);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if constexpr (EpiTraits<EPI>::pre) {
        if (mt + 1 < MT) {
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) pre[(mt + 1) & 1][hb] = de.load_pre(mrow + (mt + 1) * 16, hb);
        }
      }
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const f32x8 o = de.finish(f32x8{acc[hb * 2][mt], acc[hb * 2 + 1][mt]}, pre[mt & 1][hb], mrow + mt * 16, hb);
        if constexpr (COLSUM) {
          if (mrow + mt * 16 < (unsigned)p.M) { cs[hb].lo += o.lo; cs[hb].hi += o.hi; }     // rows >= M are not outputs
        }
      }
    }
    if constexpr (COLSUM) {
      // the 16 lanes i16 = 0..15 of a column octet hold different rows: butterfly over lane bits 0..3, then lane i16 == 0
      // writes this wave's WR-row column sums (one partial row per (tile row, wm))
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) { cs[hb].lo[e] += __shfl_xor(cs[hb].lo[e], o, 64); cs[hb].hi[e] += __shfl_xor(cs[hb].hi[e], o, 64); }
        const int64_t n = n0 + wn * 64 + hb * 32 + g * 8;
        if (i16 == 0 && n < p.N) {
          float* dst = p.colsum + ((int64_t)tm * 2 + wm) * p.N + n;
          store4(dst, cs[hb].lo); store4(dst + 4, cs[hb].hi);
        }
      }
    }
  });
  if (trace && lane == 0) {
    if (wave == 0) tr[3] = __builtin_amdgcn_s_memtime();
    if constexpr (TRACE) {                          // wave 0 (leading group) -> tr[16..31], wave 4 (lagging group) -> tr[32..47]
#pragma unroll
      for (int s = 0; s < 16; ++s) tr[16 + (wave >> 2) * 16 + s] = stamp[s];
    }
  }
}

// ---- persistent variant (forward orientation, epilogues without a side input, no split-K) ---------------------------------
// One workgroup per CU walks the tile ids b, b + gridDim.x, ...  (the ids concurrently in flight are consecutive, so the XCD /
// L2 grouping of xcd_remap + tile_of is that of the one-tile-per-workgroup launch).  What it buys: a K = 768 tile is 12 k-tiles of
// ~2.6k cycles between ~5k cycles of prologue (first half-tiles from L2 / HBM) and ~5k of epilogue (128 KiB of stores the memory
// system accepts at its own pace) -- tools/gemm_trace256.py.  The epilogue needs no LDS (DirectEpi), so the five half-tile DMAs of
// the NEXT tile are issued right after the last fragment read and land while the accumulators are converted and stored; the
// stores themselves are fire-and-forget.  In-order vmcnt accounting: the DMAs are older than every store of the epilogue, so
// `vmcnt(4 + number of stores issued after them)` means "the first three half-tiles have landed" (the count is the minimum the
// epilogue kind issues; extra stores -- the kept pre-activation -- only make the wait conservative).  Epilogues with a side input
// (residual, quick_gelu') stay one-tile-per-workgroup: hipcc drains vmcnt to 0 at the first use of an ordinary load while an
// LDS-DMA is in flight (cdna_hip_programming.md section 5), which would serialise the prefetch behind the epilogue.
template <int MT1>
__global__ __launch_bounds__(NTH, 2) void gemm256_persist_kernel(KParams p) {
  constexpr int MT = 4 + MT1, WR = MT * 16, TMv = 2 * WR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int nwg = p.tiles_m * p.tiles_n;
  const int nk = (int)((p.K + KE - 1) / KE);
  const int i16 = lane & 15, g = lane >> 4;
  T* Ct = reinterpret_cast<T*>(p.C);
  auto slot = [&](int kt, int w) -> char* { return smem + (((kt & 1) << 2) + w) * HALF_BYTES; };

  HalfStager<false, 64, 16 * MT1, false> ga;
  HalfStager<false, 32, 32, true> gb;
  int vid = blockIdx.x, tm, tn;
  tile_of(xcd_remap(vid, nwg, p.xcd_remap), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
  auto prologue = [&]() {
    ga.init(reinterpret_cast<const T*>(p.A), p.lda, (int64_t)tm * TMv, p.M, p.K, 0, lane, wave);
    gb.init(reinterpret_cast<const T*>(p.B), p.ldb, (int64_t)tn * TN, p.N, p.K, 0, lane, wave);
    gb.issue(slot(1, 3), 0, 0);
    ga.issue(slot(0, 0), 0, 0); gb.issue(slot(0, 1), 1, 0); ga.issue(slot(0, 2), 1, 0); gb.issue(slot(0, 3), 0, 1);
  };
  prologue();
  wait_vmcnt<4>();
  unsigned long long stamp[16];
  for (;;) {
    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();       // the wm==1 group runs one barrier behind
    main_loop<false, false, MT1, false>(ga, gb, acc, smem, nk, wm, wn, lane, stamp, -1);
    if (wm == 0) __builtin_amdgcn_s_barrier();       // groups re-aligned: every fragment of this tile has been read
    const int64_t m0 = (int64_t)tm * TMv, n0 = (int64_t)tn * TN;
    const bool has_next = vid + (int)gridDim.x < nwg;
    if (has_next) {
      vid += gridDim.x;
      tile_of(xcd_remap(vid, nwg, p.xcd_remap), p.tiles_m, p.tiles_n, p.group_n, tm, tn);
      prologue();                                    // next tile's first half-tiles fly under this tile's epilogue
    }
    fast_epi_dispatch(p, [&](auto epi_c, auto f32_c, auto cs_c) {
      constexpr int EPI = decltype(epi_c)::value;
      constexpr bool F32 = decltype(f32_c)::value;
      if constexpr (!EpiTraits<EPI>::pre && !F32 && !decltype(cs_c)::value) {      // (the launcher sends nothing else here)
        const DirectEpi<EPI, false> de(p, (void*)Ct, n0 + wn * 64 + g * 8);
        const unsigned mrow = (unsigned)(m0 + wm * WR) + i16;
        const Raw8<T> none{};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) de.finish(f32x8{acc[hb * 2][mt], acc[hb * 2 + 1][mt]}, none, mrow + mt * 16, hb);
        if (has_next) wait_vmcnt<4 + 2 * MT>();      // 2*MT stores (at least) are younger than the 10 DMA pieces
      }
    });
    if (!has_next) break;
  }
}

template <int MT1>
void launch_persist(const KParams& kp, int grid, hipStream_t st) {
  auto kern = gemm256_persist_kernel<MT1>;
  static std::once_flag configured;
  std::call_once(configured, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  kern<<<grid, NTH, LDS_BYTES, st>>>(kp);
}
template __global__ void gemm256_persist_kernel<4>(KParams);
template __global__ void gemm256_persist_kernel<3>(KParams);

template <bool AKS, bool BKS, int MT1, bool TRACE>
void launch_one(const KParams& kp, dim3 grid, hipStream_t st) {
  auto kern = gemm256_kernel<AKS, BKS, MT1, TRACE>;
  static std::once_flag configured;            // (forward thread and autograd thread may both arrive first)
  std::call_once(configured, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  kern<<<grid, NTH, LDS_BYTES, st>>>(kp);
}

// explicit instantiations (hipcc otherwise drops the host stubs of the k-strided variants)
template __global__ void gemm256_kernel<false, false, 4, false>(KParams);
template __global__ void gemm256_kernel<false, false, 3, false>(KParams);
template __global__ void gemm256_kernel<false, true, 4, false>(KParams);
template __global__ void gemm256_kernel<false, true, 3, false>(KParams);
template __global__ void gemm256_kernel<true, true, 4, false>(KParams);
template __global__ void gemm256_kernel<true, false, 4, false>(KParams);
template __global__ void gemm256_kernel<false, false, 4, true>(KParams);
template __global__ void gemm256_kernel<false, false, 3, true>(KParams);

// the (epilogue, output type, column sums) combinations the direct epilogue is specialised for (fast_epi_dispatch)
bool epi_supported(const XpGemmDesc* d) {
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

}  // namespace

// Tile height: MT1 = sub-tiles of the second M half (4: 256 rows, 3: 224, 2: 192 -- 192 is not instantiated).
//   default (desc->tile_rows_hint == 0): 256 rows.  Measured inside the training step at cfg #2 (three interleaved rounds on one
//     box, profiles/r03n_bench_ab_tile_height_in_step.txt): 224-row tiles for forward + dX GEMMs 17.14 ms/step, forward only 17.08,
//     dX only 17.09, 256 everywhere 17.02 -- the step is energy-bound (tools/clock_probe.hip) and the 224-row tile moves 14 % more
//     N-side operand bytes per FLOP, although in isolation it is 1-7 % faster per GEMM (idle CUs of the last round lend power);
//   hint 224: the height that minimises rounds x (sub-tiles + height-independent part) -- a forward-only pass is 3.8 % faster with
//     it (4.59 vs 4.77 ms): the inference forward asks for it (csrc/layer.hip, pre == NULL).
//   XPRETRAIN_GEMM256_MT1 / _MT1_NS = 3|4 force a height for all / for the dX orientation (A/B experiments).
int xp_gemm256_mt1(const XpGemmDesc* d, int split) {
  if (d->a_kstrided) return 4;                       // weight-gradient GEMMs: M = n_out is a multiple of 256
  static const int forced = getenv("XPRETRAIN_GEMM256_MT1") ? atoi(getenv("XPRETRAIN_GEMM256_MT1")) : 0;
  static const int forced_ns = getenv("XPRETRAIN_GEMM256_MT1_NS") ? atoi(getenv("XPRETRAIN_GEMM256_MT1_NS")) : 0;   // dX orientation only
  if (d->b_kstrided && (forced_ns == 3 || forced_ns == 4)) return forced_ns;
  if (forced == 3 || forced == 4) return forced;
  if (d->tile_rows_hint != 224) return 4;
  int best = 4;
  double best_cost = 0;
  for (int mt1 = 4; mt1 >= 3; --mt1) {
    const int64_t tiles = cdiv(d->M, 32 * (4 + mt1)) * cdiv(d->N, TN) * split;
    const double cost = (double)cdiv(tiles, 256) * ((4 + mt1) + 1.5);
    if (mt1 == 4 || cost < best_cost * 0.97) { best = mt1; best_cost = cost; }
  }
  return best;
}

// preconditions of the family that do not depend on split_k
bool xp_gemm256_legal(const XpGemmDesc* d) {
  if (d->in_dtype != XP_BF16 || d->a_grp != 0) return false;
  const int64_t a_rows = d->a_kstrided ? d->K : d->M, b_rows = d->b_kstrided ? d->K : d->N;
  if (!d->a_kstrided && (d->K % KE != 0 || d->lda != d->K)) return false;
  if (!d->b_kstrided && (d->K % KE != 0 || d->ldb != d->K)) return false;
  if (d->a_kstrided && (d->M % 256 != 0 || d->lda != d->M)) return false;
  if (d->b_kstrided && (d->N % TN != 0 || d->ldb != d->N)) return false;
  const int64_t lim = (int64_t)0xFFFFFFF0u - 512 * 1024 * 1024;
  if ((a_rows + 256) * d->lda * 2 >= lim || (b_rows + TN) * d->ldb * 2 >= lim) return false;
  return epi_supported(d);                           // the family has no generic epilogue: everything else -> 128x128 family
}

// XPRETRAIN_GEMM256: 0 = never, 1 = when it fills at least half the CUs (default), 2 = whenever legal.
bool xp_gemm256_wanted(const XpGemmDesc* d, int split) {
  const char* env = getenv("XPRETRAIN_GEMM256");
  const int mode = env ? atoi(env) : 1;
  if (mode == 0 || !xp_gemm256_legal(d)) return false;
  if (mode == 1 && cdiv(d->M, 256) * cdiv(d->N, TN) * split < 128) return false;
  return true;
}

int64_t xp_gemm256_colsum_rows(const XpGemmDesc* d) { return 2 * cdiv(d->M, 32 * (4 + xp_gemm256_mt1(d, 1))); }

bool xp_gemm256_try(const XpGemmDesc* d, const xpgemm::KParams& kp_base, hipStream_t st) {
  const int split = d->split_k > 1 ? d->split_k : 1;
  if (!xp_gemm256_wanted(d, split)) return false;
  xpgemm::KParams kp = kp_base;
  kp.k_per_split = cdiv(cdiv(d->K, split), KE) * KE;
  if (split > 1 && cdiv(d->K, kp.k_per_split) != split) return false;
  const int64_t k_last = d->K - (int64_t)(split - 1) * kp.k_per_split;
  if (cdiv(k_last, KE) < 2) return false;                                         // the pipeline needs >= 2 k-tiles
  const int mt1 = xp_gemm256_mt1(d, split);
  kp.tiles_m = (int)cdiv(d->M, 32 * (4 + mt1)); kp.tiles_n = (int)cdiv(d->N, TN);
  kp.group_n = kp.tiles_n;
  dim3 grid(kp.tiles_m * kp.tiles_n, 1, split);
  const bool tr = kp.dbg != nullptr;                 // xp_debug_set_gemm_trace: the barrier-stamping build of the NT kernel
  // persistent tile loop (XPRETRAIN_GEMM256_PERSIST=0 switches it off): more than one round of tiles, forward orientation,
  // bf16 output, an epilogue without side input
  static const int persist = getenv("XPRETRAIN_GEMM256_PERSIST") ? atoi(getenv("XPRETRAIN_GEMM256_PERSIST")) : 1;
  const int ntiles = kp.tiles_m * kp.tiles_n;
  if (persist && !tr && split == 1 && ntiles > 256 && !d->a_kstrided && !d->b_kstrided && !d->colsum_partials &&
      d->out_dtype == XP_BF16 && (d->epilogue == XP_EPI_NONE || d->epilogue == XP_EPI_BIAS || d->epilogue == XP_EPI_BIAS_QSCALE ||
                                  d->epilogue == XP_EPI_BIAS_GELU)) {
    if (mt1 == 4) launch_persist<4>(kp, 256, st); else launch_persist<3>(kp, 256, st);
    return true;
  }
  if (!d->a_kstrided && !d->b_kstrided) {
    if (mt1 == 4) { if (tr) launch_one<false, false, 4, true>(kp, grid, st); else launch_one<false, false, 4, false>(kp, grid, st); }
    else          { if (tr) launch_one<false, false, 3, true>(kp, grid, st); else launch_one<false, false, 3, false>(kp, grid, st); }
  } else if (!d->a_kstrided && d->b_kstrided) {
    if (mt1 == 4) launch_one<false, true, 4, false>(kp, grid, st); else launch_one<false, true, 3, false>(kp, grid, st);
  } else if (d->a_kstrided && d->b_kstrided) launch_one<true, true, 4, false>(kp, grid, st);
  else                                       launch_one<true, false, 4, false>(kp, grid, st);
  return true;
}
