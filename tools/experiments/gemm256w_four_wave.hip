// 256x256-tile MFMA GEMM, FOUR waves with 128x128 wave tiles, one wave per SIMD: the forward (both operands k-contiguous, "NT")
// member of the 256-wide family since round 5 (XPRETRAIN_GEMM256W=0 puts the forward back on gemm256.hip's 8-wave ping-pong loop).
//
// Why a second main loop (round-4 verdict): the 8-wave loop reads 12 fragments per 32 MFMAs (wave tile 128 x 64) and
// runs 8 barriers per k-tile; under MFMA load the chip is POWER-limited (tools/clock_probe.hip: 1.7 GHz with all CUs issuing MFMAs on
// random operands), so what a kernel spends besides MFMAs -- LDS fragment traffic first -- comes straight out of the clock.  A
// 128 x 128 wave tile reads 16 fragments per 64 MFMAs (a third fewer LDS bytes per FLOP), the whole accumulator lives in the
// wave's 512 registers, and with one wave per SIMD there is no partner to hand the matrix pipe to: the wave's own instruction
// stream interleaves next-quadrant fragment reads and LDS-DMA pieces between its MFMAs, 4 barriers per k-tile.
//
//   workgroup  256 threads = 4 waves as 2 (M) x 2 (N); wave tile 128 x 128 = 8 x 8 accumulators (256 registers)
//   k-tile     64 bf16 of k = four 16 KiB HALF-TILES (gemm256.hip's image: [128 rows][128 B], chunk' = chunk ^ (((row>>1)&3)<<1));
//              half h of an operand holds rows {w*128 + h*64 + r} (r < 64) of both waves w of that side.
//   quadrants  a k-tile is four QUADRANTS of 32 MFMAs in serpentine order  Q0 = A0 x B0, Q1 = A0 x B1, Q2 = A1 x B1, Q3 = A1 x B0.
//              While a quadrant computes, the wave reads the 8 fragments of the half the NEXT quadrant brings in
//              (Q0: B1(t), Q1: A1(t), Q2: A0(t+1), Q3: B0(t+1); five fragment sets of 32 registers, B0 alternates between two) and
//              issues its 4 DMA pieces of one half-tile of k-tile t+2 (Q0: A0, Q1: B0, Q2: B1, Q3: A1).
//   ring       8 half-tile slots = 128 KiB; half X of k-tile t lives in slot (t & 1) * 4 + {A0: 0, B0: 1, B1: 2, A1: 3}.  The
//              DMA of quadrant Qi(t) overwrites the slot whose half was read two quadrants earlier (two barriers back); the half
//              read in the next quadrant was issued five half-tiles before the newest one: `s_waitcnt vmcnt(20)` (4 pieces per
//              half-tile and wave), never 0 inside the loop.  k-tiles past the end are issued with an out-of-range offset (the
//              buffer unit zero-fills), so the counts are the same in the tail.
//   epilogue   gemm256.hip's: wave-private LDS staging (rounds of 32 rows x 128 columns fp32), row-major read-back, the shared
//              straight-line fused epilogue (gemm_common.h::FastEpi).
#include "common.h"
#include "gemm_common.h"
#include <stdlib.h>
#include <mutex>

namespace {

using namespace xpgemm;

typedef bf16_t T;
constexpr int TM = 256, TN = 256, KE = 64;
constexpr int NTH = 256, NWAVES = 4;
constexpr int HALF_BYTES = 128 * 128, NSLOT = 8, LDS_BYTES = NSLOT * HALF_BYTES;
constexpr unsigned OOB = 0xFFFFFFF0u;

typedef __attribute__((address_space(3))) char lds_char;

// DMA of one k-contiguous operand's half-tiles by four waves: 16 passes of 1 KiB (8 rows x 128 B) per half-tile, wave w issues
// passes {j * 4 + w}.  Local row r of half h is tile row (r / 64) * 128 + h * 64 + r % 64.
struct Stager4 {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[2][4];     // [half][piece]

  __device__ __forceinline__ void init(const T* base, int64_t ld, int64_t row0, int64_t rows, int64_t kbeg, int lane, int wave) {
    const int64_t bytes = rows * ld * 2;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (unsigned)bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (j * NWAVES + wave) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz128(row);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t trow = (row >> 6) * 128 + h * 64 + (row & 63);
        const int64_t off = ((row0 + trow) * ld + kbeg) * 2 + c * 16;
        voff[h][j] = off >= bytes ? OOB : (unsigned)off;
      }
    }
  }
  // piece j of half h of k-tile kt into the slot at `slot`; !valid (k-tile past the end): zero-fill
  __device__ __forceinline__ void piece(char* slot, int h, int j, int kt, bool valid, int wave) const {
    unsigned o = voff[h][j] + (unsigned)kt * (unsigned)(KE * 2);
    if (o < voff[h][j] || !valid) o = OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_char*)slot + (j * NWAVES + wave) * 1024, 16, o, 0, 0, 0);
  }
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// In-place MFMA on an accumulator pinned to the AGPR file.  The 64 accumulators fill all 256 AGPRs of the wave: through the
// builtin, hipcc allocates destination and source C separately and, with not one free tuple, shuffles accumulators through
// v_accvgpr_read / _write / scratch around every MFMA (600 moves per 256 MFMAs in the first build of this loop).
__device__ __forceinline__ void mma_acc(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// BAR2: one barrier per TWO quadrants (after Q1 and Q3; the wait then covers the two halves the next two quadrants read).
// ABL (timing experiments only, results are wrong): 1 = no DMA pieces inside the loop, 2 = no fragment reads inside the loop.
template <int ABL, bool BAR2>
__global__ __launch_bounds__(NTH, 1) void gemm256w_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg, p.xcd_remap);
  int tm, tn;
  tile_of(bid, p.tiles_m, p.tiles_n, p.group_n, tm, tn);
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;
  const int nk = (int)(p.K / KE);

  Stager4 ga, gb;
  ga.init(reinterpret_cast<const T*>(p.A), p.lda, m0, p.M, 0, lane, wave);
  gb.init(reinterpret_cast<const T*>(p.B), p.ldb, n0, p.N, 0, lane, wave);

  f32x4 acc[8][8];                                   // [nt][mt]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool trace = p.dbg != nullptr && bid == nwg / 2 && wave == 0;
  unsigned long long* tr = p.dbg;
  if (trace && lane == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[1] = nk; }

  // per-lane fragment offsets inside a half-tile: sub-tile (w * 4 + s) of 16 rows, k sub-step ks
  const int i16 = lane & 15, g = lane >> 4;
  unsigned offA[2], offB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offA[ks] = (unsigned)tile128_off(wm * 64 + i16, ks * 4 + g);
    offB[ks] = (unsigned)tile128_off(wn * 64 + i16, ks * 4 + g);
  }
  constexpr int SA0 = 0, SB0 = 1, SB1 = 2, SA1 = 3;   // slot of a half inside its k-tile's group of four

  bf16x8 fa0[2][4], fa1[2][4], fb0x[2][4], fb0y[2][4], fb1[2][4];     // [ks][sub]

  // prologue: k-tiles 0 and 1 in flight (8 half-tiles, 32 pieces per wave)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ga.piece(smem + (t * 4 + SA0) * HALF_BYTES, 0, j, t, t < nk, wave);
#pragma unroll
    for (int j = 0; j < 4; ++j) gb.piece(smem + (t * 4 + SB0) * HALF_BYTES, 0, j, t, t < nk, wave);
#pragma unroll
    for (int j = 0; j < 4; ++j) gb.piece(smem + (t * 4 + SB1) * HALF_BYTES, 1, j, t, t < nk, wave);
#pragma unroll
    for (int j = 0; j < 4; ++j) ga.piece(smem + (t * 4 + SA1) * HALF_BYTES, 1, j, t, t < nk, wave);
  }
  wait_vmcnt<24>();                                  // A0(0), B0(0) landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    fa0[r >> 2][r & 3]  = *reinterpret_cast<const bf16x8*>(smem + SA0 * HALF_BYTES + (r & 3) * 2048 + offA[r >> 2]);
    fb0x[r >> 2][r & 3] = *reinterpret_cast<const bf16x8*>(smem + SB0 * HALF_BYTES + (r & 3) * 2048 + offB[r >> 2]);
  }
  wait_vmcnt<(BAR2 ? 16 : 20)>();                    // B1(0) (and A1(0)) landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // One quadrant: 8 chunks of 4 MFMAs; chunks 0..3 each issue two of the next half's fragment reads, chunks 4..7 one DMA piece.
  //   FA / FB      fragment sets the MFMAs use;  HA / HB: which accumulator quarter
  //   RD / RSLOT / ROFF   destination set, ring slot and per-lane offsets of the half read for the next quadrant
  //   ST / SSLOT / SH / SKT   stager, ring slot, half and k-tile of the half-tile issued
#define XP_QUAD(QI, HA, HB, FA, FB, RD, RSLOT, ROFF, ST, SSLOT, SH, SKT)                                                   \
  do {                                                                                                                     \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                        \
      if (c < 4) {                                                                                                         \
        if constexpr (ABL != 2) {                                                                                          \
          _Pragma("unroll") for (int r = 2 * c; r < 2 * c + 2; ++r)                                                        \
            RD[r >> 2][r & 3] = *reinterpret_cast<const bf16x8*>(smem + (RSLOT) * HALF_BYTES + (r & 3) * 2048 + ROFF[r >> 2]); \
        }                                                                                                                  \
      } else {                                                                                                             \
        if constexpr (ABL != 1) ST.piece(smem + (SSLOT) * HALF_BYTES, SH, c - 4, SKT, (SKT) < nk, wave);                   \
      }                                                                                                                    \
      _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                        \
        mma_acc(acc[(HB) * 4 + n][(HA) * 4 + (c & 3)], FB[c >> 2][n], FA[c >> 2][c & 3]);                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }                                                                                                                      \
    if constexpr (!BAR2 || ((QI) & 1)) {                                                                                   \
      wait_vmcnt<(BAR2 ? 16 : 20)>();                                                                                      \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
      __builtin_amdgcn_s_barrier();                                                                                        \
    }                                                                                                                      \
  } while (0)

  // k-tile t of parity P = t & 1: its halves sit in slots P*4 + {0..3}, the next k-tile's in (1-P)*4 + {0..3}; the DMA of k-tile t+2
  // goes into this k-tile's own slots.  B0 of k-tile t is in fb0x (P == 0) / fb0y (P == 1).
#define XP_KTILE(P, FB0, FB0N, t)                                                                                          \
  do {                                                                                                                     \
    XP_QUAD(0, 0, 0, fa0, FB0, fb1,  (P) * 4 + SB1,       offB, ga, (P) * 4 + SA0, 0, (t) + 2);                            \
    XP_QUAD(1, 0, 1, fa0, fb1, fa1,  (P) * 4 + SA1,       offA, gb, (P) * 4 + SB0, 0, (t) + 2);                            \
    XP_QUAD(2, 1, 1, fa1, fb1, fa0,  (1 - (P)) * 4 + SA0, offA, gb, (P) * 4 + SB1, 1, (t) + 2);                            \
    XP_QUAD(3, 1, 0, fa1, FB0, FB0N, (1 - (P)) * 4 + SB0, offB, ga, (P) * 4 + SA1, 1, (t) + 2);                            \
  } while (0)

  int t = 0;
  for (; t + 1 < nk; t += 2) {
    XP_KTILE(0, fb0x, fb0y, t);
    XP_KTILE(1, fb0y, fb0x, t + 1);
  }
  if (t < nk) XP_KTILE(0, fb0x, fb0y, t);
#undef XP_KTILE
#undef XP_QUAD
  wait_vmcnt<0>();                                   // the zero-fill pieces of the k-tiles past the end
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the asm MFMAs are invisible to hipcc's MFMA -> accvgpr_read hazard pass)
  __builtin_amdgcn_s_barrier();                      // every wave is done with the ring -> LDS is free for the epilogue
  if (trace && lane == 0) tr[2] = __builtin_amdgcn_s_memtime();

  // ---- epilogue: wave-private staging, 4 rounds of 32 rows x 128 columns fp32 (16 KiB per wave) ----------------------------
  constexpr int CW = 128, MT = 8, NT = 8;
  char* stg = smem + wave * (32 * CW * 4);
  float* Cf = reinterpret_cast<float*>(p.C);
  T* Ct = reinterpret_cast<T*>(p.C);
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
    static_assert(true, "");
    if constexpr (!COLSUM) {                           // (the launcher never sends column-sum problems here: they are dX kernels)
      const int c = lane & 15, r4 = lane >> 4;         // 8 columns per lane: 16 lanes per row, 4 rows per pass
      const FastEpi<T, EPI, F32> fe(p, F32 ? (void*)Cf : (void*)Ct, n0 + wn * CW + c * 8);
      const unsigned mrow = (unsigned)(m0 + wm * (MT * 16)) + r4;
      Raw8<T> pre[2][8];
      if constexpr (EpiTraits<EPI>::pre) {
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) pre[0][pass] = fe.load_pre(mrow + pass * 4);
      }
#pragma unroll
      for (int q = 0; q < MT / 2; ++q) {
        if constexpr (EpiTraits<EPI>::pre) {
          if (q + 1 < MT / 2) {
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) pre[(q + 1) & 1][pass] = fe.load_pre(mrow + (q + 1) * 32 + pass * 4);
          }
        }
        stage_round(q);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x8 v[4];
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) {
            const int row = (half * 4 + pp) * 4 + r4;
            v[pp].lo = *reinterpret_cast<const f32x4*>(stg + row * (CW * 4) + (((2 * c) ^ (row & 7)) << 4));
            v[pp].hi = *reinterpret_cast<const f32x4*>(stg + row * (CW * 4) + (((2 * c + 1) ^ (row & 7)) << 4));
          }
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) {
            const int pass = half * 4 + pp;
            (void)fe.finish(v[pp], pre[q & 1][pass], mrow + q * 32 + pass * 4);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  });
  (void)fast;
  if (trace && lane == 0) tr[3] = __builtin_amdgcn_s_memtime();
}

template <int ABL, bool BAR2>
bool launch_w(const KParams& kp, dim3 grid, hipStream_t st) {
  auto kern = gemm256w_kernel<ABL, BAR2>;
  // the 128 KiB dynamic-LDS opt-in is per device (forward thread and autograd thread may both arrive first)
  static std::mutex mu;
  static int configured[64] = {0};                  // 0: not yet, 1: ok, -1: refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!configured[dev])
      configured[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            LDS_BYTES) == hipSuccess ? 1 : -1;
    if (configured[dev] < 0) return false;
  }
  kern<<<grid, NTH, LDS_BYTES, st>>>(kp);
  return true;
}

}  // namespace

// The four-wave loop serves the k-contiguous (forward) problems of the 256-wide family without split-K or fused column sums.
bool xp_gemm256w_try(const XpGemmDesc* d, const xpgemm::KParams& kp_base, hipStream_t st) {
  static const int mode = getenv("XPRETRAIN_GEMM256W") ? atoi(getenv("XPRETRAIN_GEMM256W")) : 1;
  if (mode == 0 || d->a_kstrided || d->b_kstrided || d->split_k > 1 || d->colsum_partials) return false;
  if (!xp_gemm256_wanted(d, 1) || d->K / KE < 2) return false;
  xpgemm::KParams kp = kp_base;
  kp.tiles_m = (int)cdiv(d->M, TM); kp.tiles_n = (int)cdiv(d->N, TN);
  kp.group_n = xp_gemm256_group_n(d, kp.tiles_n);
  kp.flat_split = 0;
  dim3 grid(kp.tiles_m * kp.tiles_n, 1, 1);
  switch (mode) {
    case 2:  return launch_w<0, true>(kp, grid, st);
    case 11: return launch_w<1, false>(kp, grid, st);     // (timing experiments: wrong results)
    case 12: return launch_w<2, false>(kp, grid, st);
    default: return launch_w<0, false>(kp, grid, st);
  }
}
