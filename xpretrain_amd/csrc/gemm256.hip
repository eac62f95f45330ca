// 256x256-tile MFMA GEMM (experimental second family; OFF by default, XPRETRAIN_GEMM256=1|2 enables it).
//
// Motivation (measured on MI355X with tools/gemm_trace.py, s_memtime stamps inside the 128x128 kernel): one k-iteration
// of the production kernel costs 2200-3000 cycles per wave for 544 cycles of MFMA work -- ~700-1100 cycles ISSUING its
// eight 1-KiB buffer_load...lds pieces (the CU's texture-address path moves 64 B/clk, and a 128x128x64 stage needs 32 KiB
// per 544 MFMA-cycles, i.e. the TA is ~94 % as busy as the matrix pipe), ~320-600 cycles waiting for the DMA and ~330 at
// the barrier.  A 256x256 tile halves the bytes that cross the TA per FLOP.
//
//   workgroup  512 threads = 8 waves as 2 (M) x 4 (N); wave tile 128 x 64 = 8 x 4 accumulators (128 VGPRs)
//   stage      128 BYTES of k (64 bf16 / 32 f32), 256 + 256 rows = 64 KiB; 2 stages = 128 KiB LDS, 1 workgroup / CU
//              (64-byte-row stages measured 0.7x: half-line L1 fills double the L2->L1 traffic)
//   per stage  s_waitcnt vmcnt(0) ; raw s_barrier ; issue stage t+1 ; 2 x (12 fragment reads + 32 MFMA) per wave
//   epilogue   wave-private LDS staging (rounds of 32 rows x 64 cols fp32), row-major read-back, shared fused epilogue
//
// Results at BASELINE cfg #2 shapes (gpurun_out/call18, call19): best steady state of all variants (K=3072 forward
// 903 TFLOP/s vs 871 for 128x128) but WORSE on the K=768 problems (430-490 vs 490-680 TFLOP/s): only 222-888 tiles for
// 256 CUs at one workgroup per CU, so the prologue latency, the 12-stage loop and the large epilogue are fully
// exposed.  Variants tried on the way, all correct and all <= the 128x128 family here: 64-byte-row 4-stage ring;
// 256x128 3-stage ring with counted vmcnt; the same with two wave groups staggered by half a sub-step.
// Next step (not built): persistent tile loop (prefetch the next tile's first stage under the epilogue) + producer
// wave so the MFMA waves never pay the DMA issue cost.
//
// Tile images are the 128x128 family's (gemm.hip) with more rows; XOR swizzles are applied to the per-lane GLOBAL
// source address of the lane-linear DMA:
//   k-contiguous [R rows][128 B]:     chunk' = chunk ^ (((row>>1)&3)<<1)
//   k-strided bf16 [64 k][R*2 B]:     32-byte block' = block ^ ((k&3) | ((k>>3)&1)<<2)   (ds_read_b64_tr_b16)
//   k-strided f32  [32 k][R*4 B]:     col' = col ^ (((k>>2)&1)<<4)                       (ds_read_b32)
#include "common.h"
#include "gemm_common.h"
#include <stdlib.h>

namespace {

using namespace xpgemm;

constexpr int TM = 256, TN = 256, SKB = 128;     // SKB: bytes of k per stage
constexpr int NTH = 512, NWAVES = 8, NS = 2;
constexpr int WAVES_N = 4, MT = TM / (NWAVES / WAVES_N) / 16, NT = TN / WAVES_N / 16;   // wave tile 128 x 64: 8 x 4
constexpr int A_BYTES = TM * SKB, B_BYTES = TN * SKB;
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;   // 48 KiB
constexpr int LPS = (A_BYTES + B_BYTES) / (NTH * 16);   // DMA instructions per thread per stage = 6

typedef __attribute__((address_space(3))) char lds_char;

// R = rows of the operand tile (256 for the M side, 128 for the N side)
template <typename T, bool KS, int R>
struct Stager {
  static constexpr int ES = sizeof(T);
  static constexpr int KE = SKB / ES;
  static constexpr int NPASS = R * SKB / 1024 / NWAVES;      // 1 KiB DMA passes per wave per stage: 4 (R=256) / 2
  static constexpr int RB = R * ES;                          // bytes of one k-row of the k-strided image
  static constexpr int LPR = RB / 16;                        // lanes per k-row
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[NPASS];
  unsigned step;
  unsigned lds_off[NPASS];

  __device__ __forceinline__ void init(const T* base, int64_t ld, int64_t row0, int64_t rows, int64_t kend, int64_t kbeg,
                                       int lane, int wave) {
    const int64_t bytes = (KS ? kend : rows) * ld * ES;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (unsigned)bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const int pass = j * NWAVES + wave;
      lds_off[j] = pass * 1024;
      int64_t off;
      if constexpr (!KS) {
        const int row = pass * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz128(row);
        off = ((row0 + row) * ld + kbeg) * ES + c * 16;
      } else if constexpr (sizeof(T) == 2) {
        const int kr = pass * (64 / LPR) + lane / LPR, c16 = lane % LPR;
        const int src = (((c16 >> 1) ^ ks_f(kr)) << 1) | (c16 & 1);
        off = ((kbeg + kr) * ld + row0) * ES + src * 16;
      } else {
        const int kr = pass * (64 / LPR) + lane / LPR;
        const int col = ((lane % LPR) * 4) ^ (((kr >> 2) & 1) << 4);
        off = ((kbeg + kr) * ld + row0 + col) * ES;
      }
      voff[j] = off >= bytes ? 0xFFFFFFF0u : (unsigned)off;
    }
    step = (unsigned)((KS ? (int64_t)KE * ld : (int64_t)KE) * ES);
  }
  __device__ __forceinline__ void issue(char* tile, int kt) const {
    lds_char* t3 = (lds_char*)tile;
    const unsigned adv = (unsigned)kt * step;
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      unsigned o = voff[j] + adv;
      if (o < voff[j]) o = 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, t3 + lds_off[j], 16, o, 0, 0, 0);
    }
  }
};

// fragment of 16-row sub-tile `ot` for the 64-byte k super-step `ks` (0..1) of a stage
template <typename T, bool KS, int R>
__device__ __forceinline__ typename Frag<T>::type frag(const char* tile, int ot, int ks, int lane) {
  constexpr int RB = R * (int)sizeof(T);
  const int i = lane & 15, g = lane >> 4;
  if constexpr (!KS) {
    return *reinterpret_cast<const typename Frag<T>::type*>(tile + tile128_off(ot * 16 + i, ks * 4 + g));
  } else if constexpr (sizeof(T) == 2) {
    const int f = (i >> 2) | ((g & 1) << 2);
    const int kr = ks * 32 + g * 8 + (i >> 2);
    const char* p = tile + kr * RB + ((ot ^ f) << 5) + ((i & 3) << 3);
    i16x4 lo = lds_read_tr16(p);
    i16x4 hi = lds_read_tr16(p + 4 * RB);
    typedef __attribute__((ext_vector_type(8))) short i16x8;
    i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  } else {
    f32x4 v;
    const int col = (ot * 16 + i) ^ ((g & 1) << 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const float*>(tile + (ks * 16 + 4 * g + e) * RB + col * 4);
    return v;
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(NTH, 2) void gemm256_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  int tm, tn;
  tile_of(bid, p.tiles_m, p.tiles_n, p.group_n, tm, tn);
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;

  constexpr int KE = SKB / sizeof(T);
  const int64_t kbeg = (int64_t)blockIdx.z * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nk = (int)((kend - kbeg + KE - 1) / KE);

  Stager<T, AKS, TM> ga;
  Stager<T, BKS, TN> gb;
  ga.init(reinterpret_cast<const T*>(p.A), p.lda, m0, p.M, kend, kbeg, lane, wave);
  gb.init(reinterpret_cast<const T*>(p.B), p.ldb, n0, p.N, kend, kbeg, lane, wave);
  auto slotA = [&](int s) -> char* { return smem + s * STAGE_BYTES; };
  auto slotB = [&](int s) -> char* { return smem + s * STAGE_BYTES + A_BYTES; };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool trace = p.dbg != nullptr && (int)blockIdx.x == nwg / 2 && blockIdx.z == 0 && wave == 0;
  unsigned long long* tr = p.dbg;
#define XP_STAMP(i) do { if (trace && t < 24) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) tr[8 + t * 5 + (i)] = t_; } } while (0)
  if (trace && lane == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[1] = nk; }
  // prologue: NS-1 stages in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) { ga.issue(slotA(s), s); gb.issue(slotB(s), s); }

  int slot = 0;
  for (int t = 0; t < nk; ++t) {
    XP_STAMP(0);
    // stages issued so far = min(nk, t+NS-1); stage t must have landed -> the younger one may stay in flight
    if (NS > 2 && t + 1 < nk) wait_vmcnt<(NS - 2) * LPS>(); else wait_vmcnt<0>();
    XP_STAMP(1);
    __builtin_amdgcn_s_barrier();
    XP_STAMP(2);
    if (t + NS - 1 < nk) {
      const int s2 = slot == 0 ? NS - 1 : slot - 1;          // (t + NS - 1) % NS == (t - 1) % NS
      ga.issue(slotA(s2), t + NS - 1);
      gb.issue(slotB(s2), t + NS - 1);
    }
    XP_STAMP(3);
    const char* tA = slotA(slot);
    const char* tB = slotB(slot);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Frag<T>::type fw[NT], fx[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i) fw[i] = frag<T, BKS, TN>(tB, wn * NT + i, ks, lane);
#pragma unroll
      for (int i = 0; i < MT; ++i) fx[i] = frag<T, AKS, TM>(tA, wm * MT + i, ks, lane);
      __builtin_amdgcn_sched_barrier(0);      // all fragment reads in flight before the first MFMA (no read/wait/MFMA ping-pong)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = mma16(fw[nt], fx[mt], acc[nt][mt]);
    }
    XP_STAMP(4);
    slot = slot == NS - 1 ? 0 : slot + 1;
  }
#undef XP_STAMP
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();        // every wave is done reading the ring -> LDS is free for the epilogue
  if (trace && lane == 0) tr[2] = __builtin_amdgcn_s_memtime();

  // ---- epilogue: wave-private staging, MT/2 rounds of 32 rows x (NT*16) columns --------------------------------
  constexpr int CW = NT * 16, CCH = CW / 4;                 // staged columns per wave, 16-byte chunks per row
  char* stg = smem + wave * (32 * CW * 4);
  const int i16 = lane & 15, g = lane >> 4;
  const int c = lane % CCH, r4 = lane / CCH;                // read-back: CCH lanes per row, 64/CCH rows per pass
  const int64_t n = n0 + wn * CW + c * 4;
  float* Cf = reinterpret_cast<float*>(p.C);
  T* Ct = reinterpret_cast<T*>(p.C);
  if (gridDim.z > 1) Cf += (int64_t)blockIdx.z * p.M * p.N;
  const bool ncol_ok = n < p.N;
  const EpiLane el(p, ncol_ok ? n : 0);
#pragma unroll
  for (int q = 0; q < MT / 2; ++q) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = h * 16 + i16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(stg + row * (CW * 4) + (((nt * 4 + g) ^ (row & 7)) << 4)) = acc[nt][q * 2 + h];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int pass = 0; pass < 32 / (64 / CCH); ++pass) {
      const int row = pass * (64 / CCH) + r4;
      const int64_t m = m0 + wm * (MT * 16) + q * 32 + row;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * (CW * 4) + ((c ^ (row & 7)) << 4));
      if (ncol_ok && m < p.M) epi_row<T>(p, el, v, m, n, Cf, Ct);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (trace && lane == 0) tr[3] = __builtin_amdgcn_s_memtime();
}

template <typename T, bool AKS, bool BKS>
void launch_one(const KParams& kp, dim3 grid, hipStream_t st) {
  static bool configured = false;
  auto kern = gemm256_kernel<T, AKS, BKS>;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE_BYTES);
    configured = true;
  }
  kern<<<grid, NTH, NS * STAGE_BYTES, st>>>(kp);
}

template <typename T>
void launch_t(const XpGemmDesc* d, const KParams& kp, dim3 grid, hipStream_t st) {
  if (!d->a_kstrided && !d->b_kstrided)      launch_one<T, false, false>(kp, grid, st);
  else if (!d->a_kstrided && d->b_kstrided)  launch_one<T, false, true>(kp, grid, st);
  else if (d->a_kstrided && d->b_kstrided)   launch_one<T, true, true>(kp, grid, st);
  else                                       launch_one<T, true, false>(kp, grid, st);
}

// explicit instantiations (hipcc otherwise drops the host stubs of the k-strided variants)
#define XP_INST(T) \
  template __global__ void gemm256_kernel<T, false, false>(KParams); \
  template __global__ void gemm256_kernel<T, false, true>(KParams);  \
  template __global__ void gemm256_kernel<T, true, true>(KParams);   \
  template __global__ void gemm256_kernel<T, true, false>(KParams);
XP_INST(bf16_t)
XP_INST(float)
#undef XP_INST

}  // namespace

bool xp_gemm256_try(const XpGemmDesc* d, const xpgemm::KParams& kp_base, hipStream_t st) {
  // 0 = off (default), 1 = size heuristic, 2 = whenever legal (see the header comment for the measurements).
  const char* env = getenv("XPRETRAIN_GEMM256");
  const int mode = env ? atoi(env) : 0;
  if (mode == 0) return false;
  const int esz = d->in_dtype == XP_BF16 ? 2 : 4;
  const int64_t ke = SKB / esz;
  if (d->a_grp != 0) return false;
  if (mode == 1 && (d->M < 1024 || d->N < 256 || d->K < 4 * ke)) return false;   // small problems: 128x128 family
  const int64_t a_rows = d->a_kstrided ? d->K : d->M, b_rows = d->b_kstrided ? d->K : d->N;
  if (!d->a_kstrided && (d->K % ke != 0 || d->lda != d->K)) return false;
  if (!d->b_kstrided && (d->K % ke != 0 || d->ldb != d->K)) return false;
  if (d->a_kstrided && (d->M % TM != 0 || d->lda != d->M)) return false;
  if (d->b_kstrided && (d->N % TN != 0 || d->ldb != d->N)) return false;
  const int64_t lim = (int64_t)0xFFFFFFF0u - 512 * 1024 * 1024;
  if ((a_rows + TM) * d->lda * esz >= lim || (b_rows + TN) * d->ldb * esz >= lim) return false;
  const int split = d->split_k > 1 ? d->split_k : 1;
  xpgemm::KParams kp = kp_base;
  kp.k_per_split = cdiv(cdiv(d->K, split), ke) * ke;
  if (split > 1 && cdiv(d->K, kp.k_per_split) != split) return false;
  kp.tiles_m = (int)cdiv(d->M, TM); kp.tiles_n = (int)cdiv(d->N, TN);
  kp.group_n = kp.tiles_n; kp.xcd_remap = 1;
  dim3 grid(kp.tiles_m * kp.tiles_n, 1, split);
  if (d->in_dtype == XP_BF16) launch_t<bf16_t>(d, kp, grid, st);
  else                        launch_t<float>(d, kp, grid, st);
  return true;
}
