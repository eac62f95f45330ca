// Fused attention for the CLIP-ViP path on gfx950 (wave64, 16x16x32 bf16 MFMA, head_dim = 64).
//
// Both attention flavours of the reference are one block-sparse problem family:
//   PROXY  (CLIPAttention.forward2, modeling/CLIP_ViP.py:332-381): for every (batch b, head h, frame n) the
//          "problem" has rows/cols R = [M proxy tokens | L tokens of frame n].  Frame rows see all R cols and
//          are complete inside the problem.  Proxy rows see every frame's tokens: each problem contributes a
//          partial (max, sum, unnormalised O) that a tiny merge kernel combines over n; the proxy x proxy
//          block is counted in problem n == 0 only.  No repeat()/cat() copies of K/V are ever materialised.
//   CAUSAL (CLIPAttention.forward text path :266-330): one problem per (b, h), R = S, key <= query, padded
//          keys get the reference's additive finfo.min (so an all-padded row degenerates to uniform, as there).
//
// Data layout: qkv[B,S,3,H,64] exactly as the fused QKV GEMM writes it (128 contiguous bytes per
// (token, head)); out/dout [B,S,H*64]; stats[B,H,S,2] = (row max, log row sum).
//
// Kernel structure (flash style, no S x S matrix in memory): every kernel owns 7 waves x 16 OWN rows of one problem
// and walks the OTHER dimension, staged in LDS (XOR-swizzled 128-byte rows) in groups of up to 208 rows.
//   fwd : own = queries.  Scores are computed TRANSPOSED (S^T = K Q^T) so that each lane owns one query
//         column: the online-softmax statistics are per-lane scalars and the bf16 P values are already the
//         B-operand fragment of the P.V MFMA -- no cross-lane shuffles or LDS round trip for P.  V^T
//         fragments come straight out of the row-major V tile through ds_read_b64_tr_b16.
//   bwd : two passes, no atomics (deterministic).  dQ: own = queries (transposed orientation; also computes
//         delta = rowsum(dO * O)), then dK/dV: own = keys (S = Q K^T orientation so P/dS are again B-operand
//         fragments).  Proxy-token partials are reduced over frames by a small kernel.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

constexpr int DH = 64;
constexpr float F32_MIN = -3.4028234663852886e38f;   // torch.finfo(float32).min, _expand_mask (:50-61)

struct AP {
  const bf16_t* qkv; int64_t ldqkv;
  bf16_t* out; const bf16_t* dout; int64_t ldo;
  bf16_t* dqkv;
  float* stats;
  const int64_t* pad;
  int mode, B, H, S, M, N, L, R;
  int nq, nprob;                        // forward: query blocks per problem, problems
  float q_scale;
  float* ws0; float* ws1; float* ws2;   // fwd: partials | bwd: delta, dq partials, dkv partials
  unsigned long long* tr;               // xp_debug_set_attn_trace (fused backward): cycle stamps of one workgroup
  float* cs; int cs_main;               // bwd, optional: column-sum partial rows of dqkv [cs_main + B*M][3*H*64] (see xp_attn_bwd2)
};

struct Prob {
  int b, h, n;
  __device__ __forceinline__ Prob(const AP& p, int idx) {
    if (p.mode == XP_ATTN_PROXY) { n = idx % p.N; int bh = idx / p.N; h = bh % p.H; b = bh / p.H; }
    else { n = 0; h = idx % p.H; b = idx / p.H; }
  }
};
// problem row -> token index inside the sample
__device__ __forceinline__ int tok_of(const AP& p, int n, int r) {
  return (p.mode == XP_ATTN_PROXY && r >= p.M) ? p.M + n * p.L + (r - p.M) : r;
}
// may query row rq attend key row rk (both < R)?  padding is handled separately (finfo.min, still "attended")
__device__ __forceinline__ bool allowed(const AP& p, int n, int rq, int rk) {
  if (p.mode == XP_ATTN_PROXY) return !(rq < p.M && rk < p.M && n != 0);
  return rk <= rq;
}

// per-lane register fragment of one row (as MFMA B operand: lane (j = row, g) holds d = 32kk + 8g .. +8)
__device__ __forceinline__ void load_row_frag(bf16x8 (&f)[2], const bf16_t* rowptr, bool valid, int g) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    if (valid) f[kk] = *reinterpret_cast<const bf16x8*>(rowptr + kk * 32 + g * 8);
    else f[kk] = __builtin_bit_cast(bf16x8, u32x4{0, 0, 0, 0});
  }
}
// A-operand fragment, rows = tile rows (16-row sub-tile t), k = d  (ds_read_b128)
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int t, int kk, int lane) {
  const int row = t * 16 + (lane & 15);
  return *reinterpret_cast<const bf16x8*>(tile + tile128_off(row, kk * 4 + (lane >> 4)));
}
// A-operand fragment of the TRANSPOSED tile: rows = d (16-col sub-tile dt), k = tile rows
//   k-slot e < 4 -> row (2c)*16 + 4g + e ; e >= 4 -> row (2c+1)*16 + 4g + e-4   (matches pack_p below)
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int dt, int c, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int r0 = (2 * c) * 16 + 4 * g + (i >> 2);
  const int ch = dt * 2 + ((i & 3) >> 1), sub = (i & 1) << 3;
  i16x4 lo = lds_read_tr16(tile + tile128_off(r0, ch) + sub);
  i16x4 hi = lds_read_tr16(tile + tile128_off(r0 + 16, ch) + sub);
  typedef __attribute__((ext_vector_type(8))) short i16x8;
  i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack_p(f32x4 a, f32x4 b) {
  return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
}
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

constexpr int PART = 2 + DH;   // forward proxy partial: m, l, O[64]

// ============================================================================================ forward
// Forward workgroup: 7 waves x 16 query rows = 112 rows of one problem against ALL its keys.  Keys/values are staged in
// LDS in groups of up to 208 rows (13 sixteen-row sub-tiles = a whole ViT-B/16 frame problem, 52 KiB), so three
// workgroups fit a CU and one workgroup's global loads / dispatch overlap the others' softmax (the loop is VALU-bound:
// ~16 cycles per exp per wave).  The tail is processed at 16-key granularity (a 16x16x16 MFMA takes an odd sub-tile):
// R = 200 costs 13 sub-tiles, not 16.  The query blocks of one problem get workgroup ids 8 apart: same XCD (shared
// L2 lines for the second K/V fetch), adjacent in dispatch order.
constexpr int FW = 7, FTHR = FW * 64, FQ = FW * 16;
constexpr int FG = 208;                                  // key rows per LDS group
constexpr int F_LDS = 2 * FG * 128 + FG;

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ s16x4 pack_p4(f32x4 a) {
  const bf16x4 v = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
  return __builtin_bit_cast(s16x4, v);
}
// A fragment of the transposed tile for ONE 16-row sub-tile (16x16x16 MFMA): rows = d, k-slot e -> row sub*16 + 4g + e
__device__ __forceinline__ s16x4 frag_cols16(const char* tile, int dt, int sub, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int r0 = sub * 16 + 4 * g + (i >> 2);
  return __builtin_bit_cast(s16x4, lds_read_tr16(tile + tile128_off(r0, dt * 2 + ((i & 3) >> 1)) + ((i & 1) << 3)));
}

// cooperative load of key rows [row0, row0 + nrows) of K AND V (same rows, V = K + voff_v bytes) into the linear swizzled
// images.  Branch-free: buffer loads relative to the sample's first token, rows >= R (or >= nrows) get an out-of-range
// offset and read as zero.
__device__ __forceinline__ void fwd_load_kv(char* gK, char* gV, __amdgpu_buffer_rsrc_t rs, unsigned ld_bytes, unsigned voff_v,
                                            const AP& p, const Prob& pr, int row0, int nrows, int tid) {
  constexpr int NJ = (FG * 8 + FTHR - 1) / FTHR;
  u32x4 vk[NJ], vv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int e = tid + j * FTHR, row = e >> 3, c = e & 7, r = row0 + row;
    const unsigned off = (row < nrows && r < p.R) ? (unsigned)tok_of(p, pr.n, r) * ld_bytes + c * 16 : 0xFFFFFF00u;
    vk[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    vv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, voff_v, 0);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int e = tid + j * FTHR, row = e >> 3, c = e & 7;
    if (row < FG) {
      *reinterpret_cast<u32x4*>(gK + tile128_off(row, c)) = vk[j];
      *reinterpret_cast<u32x4*>(gV + tile128_off(row, c)) = vv[j];
    }
  }
}

struct FwdState { f32x4 o[4]; float m, l; };

// The masked score path as a real call: inlined into the step loops its sixteen (sub-tile, slot) predicates are hoisted
// out of the key loop as loop invariants and spill; it runs on the tail step / causal band / proxy corner only.
struct S16 { f32x4 s[4]; };
__device__ __attribute__((noinline)) S16 fwd_mask_scores(S16 v, int ns, const unsigned char* pad, int R, int M, int mode, int n,
                                                         int rq, int qvalid, int kb, int g) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kl = t * 16 + 4 * g + r, rk = kb + kl;
      float x = v.s[t][r];
      if (t < ns) {
        if (pad[kl]) x = F32_MIN;
        const bool ok = mode == XP_ATTN_PROXY ? !(rq < M && rk < M && n != 0) : rk <= rq;
        if (!(qvalid && rk < R && ok)) x = -INFINITY;
      }
      v.s[t][r] = x;
    }
  return v;
}

// one step over NS (1..4) sixteen-key sub-tiles starting at LDS row t0*16 (global key row kb)
template <int NS>
__device__ __forceinline__ void fwd_step(FwdState& st, const AP& p, const Prob& pr, const char* gK, const char* gV,
                                         const unsigned char* gPad, const bf16x8 (&qf)[2], int t0, int kb, int rq,
                                         bool qvalid, int wrow0, int lane) {
  const int g = lane >> 4;
  f32x4 s[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    s[t] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) s[t] = mma16(frag_rows(gK, t0 + t, kk, lane), qf[kk], s[t]);
  }
  float tmax = -INFINITY;
  // masking is only needed (wave-uniformly) where the step touches rows >= R, with a padding mask, on the causal
  // diagonal band, or where proxy rows meet proxy keys in a frame n != 0
  const bool need_mask = kb + NS * 16 > p.R || p.pad != nullptr ||
      (p.mode == XP_ATTN_CAUSAL ? kb + NS * 16 - 1 > wrow0 : (pr.n != 0 && kb < p.M && wrow0 < p.M));
  if (need_mask) {
    S16 v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v.s[t] = t < NS ? s[t < NS ? t : 0] : f32x4{0, 0, 0, 0};
    v = fwd_mask_scores(v, NS, gPad + t0 * 16, p.R, p.M, p.mode, pr.n, rq, qvalid ? 1 : 0, kb, g);
#pragma unroll
    for (int t = 0; t < NS; ++t) s[t] = v.s[t];
  }
#pragma unroll
  for (int t = 0; t < NS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[t][r]);
  tmax = group_max(tmax);
  const float mnew = fmaxf(st.m, tmax);
  const float msafe = mnew == -INFINITY ? 0.f : mnew;
  const float alpha = __expf(st.m - msafe);          // m = -inf -> 0
  float psum = 0.f;
#pragma unroll
  for (int t = 0; t < NS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = __expf(s[t][r] - msafe); s[t][r] = e; psum += e; }
  st.l = st.l * alpha + psum;
  st.m = mnew;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) st.o[dt] *= alpha;
  const char* sV = gV + t0 * 16 * 128;               // row shift by a multiple of 16 keeps the swizzle phase
#pragma unroll
  for (int c = 0; c < NS / 2; ++c) {
    const bf16x8 pf = pack_p(s[2 * c], s[2 * c + 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st.o[dt] = mma16(frag_cols(sV, dt, c, lane), pf, st.o[dt]);
  }
  if constexpr (NS & 1) {
    const s16x4 pf = pack_p4(s[NS - 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      st.o[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(frag_cols16(sV, dt, NS - 1, lane), pf, st.o[dt], 0, 0, 0);
  }
}

__global__ __launch_bounds__(FTHR, 4) void attn_fwd_kernel(AP p) {
  __shared__ __attribute__((aligned(16))) char smem[F_LDS];
  char* gK = smem; char* gV = smem + FG * 128; unsigned char* gPad = reinterpret_cast<unsigned char*>(smem + 2 * FG * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
  const int slot = blockIdx.x >> 3;
  const int prob = (slot / p.nq) * 8 + (blockIdx.x & 7);
  if (prob >= p.nprob) return;
  const Prob pr(p, prob);
  const int qb = (slot % p.nq) * FQ;
  unsigned long long* tr = (p.mode == XP_ATTN_PROXY && p.ws2 && prob == p.nprob / 2 && qb == 0 && tid == 0)
                               ? reinterpret_cast<unsigned long long*>(p.ws2) : nullptr;      // xp_debug_set_attn_trace
  if (tr) tr[0] = __builtin_amdgcn_s_memtime();
  const int rq = qb + wave * 16 + i16;            // this lane's query row (column of S^T)
  const bool qvalid = rq < p.R;
  const int64_t qtok = (int64_t)pr.b * p.S + tok_of(p, pr.n, qvalid ? rq : 0);

  bf16x8 qf[2];
  load_row_frag(qf, p.qkv + qtok * p.ldqkv + pr.h * DH, qvalid, g);

  FwdState st;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) st.o[dt] = f32x4{0, 0, 0, 0};
  st.m = -INFINITY; st.l = 0.f;

  int nsub = (p.R + 15) / 16;                     // sixteen-key sub-tiles this workgroup visits
  if (p.mode == XP_ATTN_CAUSAL) { const int lim = (qb + FQ - 1) / 16 + 1; nsub = nsub < lim ? nsub : lim; }

  // K slice of head h of this sample's first token; V is H*DH elements further
  const bf16_t* kbase = p.qkv + (int64_t)pr.b * p.S * p.ldqkv + (int64_t)p.H * DH + pr.h * DH;
  const unsigned ld_bytes = (unsigned)(p.ldqkv * 2);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(kbase), 0, (unsigned)((int64_t)p.S * p.ldqkv * 2 - ((int64_t)p.H * DH + pr.h * DH) * 2), 0x00020000);
  const unsigned voff_v = (unsigned)(p.H * DH * 2);
  const int wrow0 = qb + wave * 16;
  const bool wave_active = wrow0 < p.R;           // wave-uniform: waves past the last row only keep the barriers
  for (int g0 = 0; g0 < nsub; g0 += FG / 16) {
    const int ng = nsub - g0 < FG / 16 ? nsub - g0 : FG / 16;
    if (g0) __syncthreads();
    fwd_load_kv(gK, gV, rs, ld_bytes, voff_v, p, pr, g0 * 16, ng * 16, tid);
    if (tid < FG) { const int r = g0 * 16 + tid; gPad[tid] = (p.pad && r < p.R) ? (p.pad[(int64_t)pr.b * p.S + r] == 0) : 0; }
    if (tr && g0 == 0) tr[1] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tr && g0 == 0) tr[2] = __builtin_amdgcn_s_memtime();
    if (!wave_active) continue;
    for (int t0 = 0; t0 < ng; t0 += 4) {
      const int ns = ng - t0 < 4 ? ng - t0 : 4;
      const int kb = (g0 + t0) * 16;
      // two instantiations only (register pressure): a 2- or 3-sub-tile tail runs the 4-wide step with its surplus keys
      // masked (rows < FG of the images are always written -- zeros past the group -- and t0 <= 8 here)
      if (ns == 1) fwd_step<1>(st, p, pr, gK, gV, gPad, qf, t0, kb, rq, qvalid, wrow0, lane);
      else         fwd_step<4>(st, p, pr, gK, gV, gPad, qf, t0, kb, rq, qvalid, wrow0, lane);
    }
  }
  const float m = st.m;
  const float l = group_sum(st.l);
  if (tr) tr[3] = __builtin_amdgcn_s_memtime();
  if (!qvalid) return;
  if (p.mode == XP_ATTN_PROXY && rq < p.M) {
    float* part = p.ws0 + ((int64_t)prob * p.M + rq) * PART;
    if (g == 0) { part[0] = m; part[1] = l; }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(part + 2 + dt * 16 + 4 * g, st.o[dt]);
    return;
  }
  const float inv = 1.0f / l;
  bf16_t* orow = p.out + qtok * p.ldo + pr.h * DH;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) store4(orow + dt * 16 + 4 * g, st.o[dt] * inv);
  if (g == 0) {
    float* sp = p.stats + (((int64_t)pr.b * p.H + pr.h) * p.S + tok_of(p, pr.n, rq)) * 2;
    sp[0] = m; sp[1] = __logf(l);
  }
}

// ============================================================================================ forward, persistent + prefetch
// Round 3.  The kernel above is launched as 2304 independent workgroups, each of which loads the whole K/V of its problem
// (51 KiB) and then computes; on the hardware that is lock-step: every CU's workgroups load at the same time (an HBM-bound burst,
// ~11 B/clk/CU: ~4.7k cycles per 51 KiB) and then all compute while HBM idles -- 49 us + 8 us merge for 115.8 MB of q/k/v/o =
// 0.26 of HBM peak, SQ counters 37 % of wave cycles parked at waits / barriers (profiles/r02c_pmc_sq_attention.csv).  A first
// rewrite (one 4-wave workgroup per problem, 4 query tiles per wave sharing every fragment read, DMA staging) kept the lock-step
// and gained 6 us.  This kernel makes the overlap explicit: ONE persistent 8-wave workgroup per CU walks the problems b,
// b + gridDim.x, ...; K/V are staged by buffer_load...lds DMA into one of two 52 KiB LDS buffers, and the DMA (and the Q row
// loads) of the NEXT problem are issued before the current one is computed, one barrier per problem.  Wave w owns the query tiles
// w and w + 8 (2 / 1 of 13), every K / V^T fragment it reads feeds both.  V^T fragments are read by inline-asm transpose reads
// (common.h::lds_read_tr16_async): hipcc would drain the DMA queue (vmcnt(0)) in front of every ds_read_tr builtin.
// Measured (tools/attn_trace.py, profiles/r03i_*): 52 us + merge vs 56: the arithmetic of a problem is ~5k cycles per wave and IS
// hidden, but a problem still takes 15-17k cycles: issuing the next problem's 18 load instructions stalls for ~2.6k cycles and
// the output stores wait behind them -- the CU moves its 102 KB per problem (Q, K, V in, O out) at 6.4 B/clk, 2.6 TB/s chip-wide,
// whatever the row pitch of qkv is (tools/attn_layout_probe.py: 384 B vs 4608 B pitch, same time).  The remaining factor to the
// ~24 us HBM floor is memory-level parallelism per CU, not arithmetic, LDS, or the layout.
// Serves PROXY problems that fit one LDS group (R <= 208: 224^2 frames at patch 16, any frame count); everything else
// (448^2: R = 788, the causal text tower) stays on the kernel above.
constexpr int F3W = 8, F3THR = F3W * 64, F3T = 2;           // waves, threads, query tiles per wave
constexpr int F3_BUF = 2 * FG * 128;                         // K + V image of one problem
constexpr int F3_LDS = 2 * F3_BUF;

// K and V rows [0, R) of one problem -> linear swizzled LDS images by LDS-DMA: one wave instruction = 8 rows x 128 B; the
// XOR swizzle of tile128_off is applied to the per-lane SOURCE chunk; rows >= R read as zero (out-of-range offset)
__device__ __forceinline__ void fwd3_stage_kv(char* gK, char* gV, const AP& p, const Prob& pr, int lane, int wave, int row0 = 0) {
  typedef __attribute__((address_space(3))) char lds_c;
  const bf16_t* kbase = p.qkv + (int64_t)pr.b * p.S * p.ldqkv + (int64_t)p.H * DH + pr.h * DH;
  const unsigned ld_bytes = (unsigned)(p.ldqkv * 2), voff_v = (unsigned)(p.H * DH * 2);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(kbase), 0, (unsigned)((int64_t)p.S * p.ldqkv * 2 - ((int64_t)p.H * DH + pr.h * DH) * 2), 0x00020000);
  constexpr int NPASS = FG / 8;                               // 26 passes of 8 rows per operand
#pragma unroll
  for (int j = 0; j < (NPASS + F3W - 1) / F3W; ++j) {
    const int pass = j * F3W + wave;
    if (pass < NPASS) {
      const int row = pass * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz128(row);
      const unsigned off = row0 + row < p.R ? (unsigned)tok_of(p, pr.n, row0 + row) * ld_bytes + c * 16 : 0xFFFFFF00u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_c*)(gK + pass * 1024), 16, off, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_c*)(gV + pass * 1024), 16, off, voff_v, 0, 0);
    }
  }
}
// the wave's Q rows as B-operand fragments
__device__ __forceinline__ void fwd3_load_q(bf16x8 (&qf)[F3T][2], const AP& p, const Prob& pr, int wave, int lane, int tile0 = 0) {
  const bf16_t* qbase = p.qkv + (int64_t)pr.b * p.S * p.ldqkv + pr.h * DH;
#pragma unroll
  for (int i = 0; i < F3T; ++i) {
    const int rq = (tile0 + wave + F3W * i) * 16 + (lane & 15);
    load_row_frag(qf[i], qbase + (int64_t)tok_of(p, pr.n, rq < p.R ? rq : 0) * p.ldqkv, rq < p.R, lane >> 4);
  }
}
// transposed fragments by asm reads: the caller waits lgkmcnt(0) before the first use (tools/check_isa.py verifies)
__device__ __forceinline__ bf16x8 frag_cols_async(const char* tile, int dt, int c, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int r0 = (2 * c) * 16 + 4 * g + (i >> 2);
  const char* q = tile + tile128_off(r0, dt * 2 + ((i & 3) >> 1)) + ((i & 1) << 3);
  i16x4 lo = lds_read_tr16_async<0>(q);
  i16x4 hi = lds_read_tr16_async<16 * 128>(q);               // row + 16: same swizzle phase
  typedef __attribute__((ext_vector_type(8))) short i16x8;
  i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ s16x4 frag_cols16_async(const char* tile, int dt, int sub, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int r0 = sub * 16 + 4 * g + (i >> 2);
  return __builtin_bit_cast(s16x4, lds_read_tr16_async<0>(tile + tile128_off(r0, dt * 2 + ((i & 3) >> 1)) + ((i & 1) << 3)));
}

// one step over NS (1 or 4) sixteen-key sub-tiles for the wave's NTL (1 or 2) query tiles; TAIL: the step reaches past key R
// (masking code exists in that instantiation only -- as a run-time condition hipcc if-converts it into 32 compares + 32 selects
// per tile in EVERY step).  The loop is VALU-issue bound (stamped: ~2.2k cycles per 64-key step of two tiles for 512 cycles of
// MFMA), so the softmax is written for instruction count: running max kept finite (-1e30: no -inf guards), exp2 of one fma per
// score (log2 e folded in), two independent partial sums.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float M_INIT = -1e30f;
// kb0: global key row of the staged group's first row (0 for single-group problems); corner: tile 0 of this wave holds the proxy
// query rows AND the group holds the proxy keys (query block 0, wave 0, key group 0)
template <int NS, int NTL, bool TAIL>
__device__ __forceinline__ void fwd3_step(FwdState (&st)[F3T], const AP& p, int frame, const char* gK, const char* gV,
                                          const bf16x8 (&qf)[F3T][2], int t0, bool corner, int lane, int kb0 = 0) {
  const int g = lane >> 4, i16 = lane & 15, kb = kb0 + t0 * 16;
  f32x4 s[NTL][NS];
  {
    bf16x8 kf[NS][2];
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) kf[t][kk] = frag_rows(gK, t0 + t, kk, lane);
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        s[i][t] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) s[i][t] = mma16(kf[t][kk], qf[i][kk], s[i][t]);
      }
  }
  // V^T fragments of the step: issued now, consumed after the softmax arithmetic
  const char* sV = gV + t0 * 16 * 128;               // row shift by a multiple of 16 keeps the swizzle phase
  bf16x8 vf[NS / 2 > 0 ? NS / 2 : 1][4];
  s16x4 vt[4];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < NS / 2; ++c)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vf[c][dt] = frag_cols_async(sV, dt, c, lane);
  if constexpr (NS & 1) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vt[dt] = frag_cols16_async(sV, dt, NS - 1, lane);
  }
  bf16x8 pf[NTL][NS / 2 > 0 ? NS / 2 : 1];
  s16x4 pt[NTL];
#pragma unroll
  for (int i = 0; i < NTL; ++i) {
    if constexpr (TAIL) {                             // the lane's key rows are kb + 16t + 4g + r: valid iff 16t + r < klim
      const int klim = p.R - kb - 4 * g;
#pragma unroll
      for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[i][t][r] = (16 * t + r < klim) ? s[i][t][r] : -INFINITY;
    }
    // proxy query rows meet proxy keys: counted in frame 0 only (CLIP_ViP.py:366-375 attends them once over all S keys)
    if (i == 0 && t0 == 0 && corner && frame != 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[0][0][r] = (i16 < p.M && 4 * g + r < p.M) ? -INFINITY : s[0][0][r];
    }
    float tmax = s[i][0][0];
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[i][t][r]);
    tmax = group_max(tmax);
    const float mnew = fmaxf(st[i].m, tmax);          // >= M_INIT: finite
    const float alpha = __builtin_amdgcn_exp2f((st[i].m - mnew) * LOG2E);
    const float nmc = -mnew * LOG2E;
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[i][t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i][t][r], LOG2E, nmc));
      ps0 += s[i][t][0] + s[i][t][1];
      ps1 += s[i][t][2] + s[i][t][3];
    }
    st[i].l = st[i].l * alpha + (ps0 + ps1);
    st[i].m = mnew;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st[i].o[dt] *= alpha;
#pragma unroll
    for (int c = 0; c < NS / 2; ++c) pf[i][c] = pack_p(s[i][2 * c], s[i][2 * c + 1]);
    if constexpr (NS & 1) pt[i] = pack_p4(s[i][NS - 1]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < NS / 2; ++c)
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) st[i].o[dt] = mma16(vf[c][dt], pf[i][c], st[i].o[dt]);
  if constexpr (NS & 1) {
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) st[i].o[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt[dt], pt[i], st[i].o[dt], 0, 0, 0);
  }
}

// all key steps of one problem for the wave's NTL tiles, then normalise and store
template <int NTL>
__device__ __forceinline__ void fwd3_problem(const AP& p, const Prob& pr, int prob, const char* gK, const char* gV,
                                             const bf16x8 (&qf)[F3T][2], int nsub, int wave, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  FwdState st[F3T];
#pragma unroll
  for (int i = 0; i < F3T; ++i) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st[i].o[dt] = f32x4{0, 0, 0, 0};
    st[i].m = M_INIT; st[i].l = 0.f;
  }
  int t0 = 0;
  for (; t0 + 4 <= nsub && (t0 + 4) * 16 <= p.R; t0 += 4) fwd3_step<4, NTL, false>(st, p, pr.n, gK, gV, qf, t0, wave == 0, lane);
  // the rest: 1..4 sub-tiles, the last of which may reach past R.  A 2- or 3-sub-tile rest runs the 4-wide step with its
  // surplus keys masked (rows < FG of the images are always written -- zeros past R -- and t0 <= 8 here)
  if (t0 < nsub) {
    if (nsub - t0 == 1) {
      if (nsub * 16 > p.R) fwd3_step<1, NTL, true>(st, p, pr.n, gK, gV, qf, t0, wave == 0, lane);
      else                 fwd3_step<1, NTL, false>(st, p, pr.n, gK, gV, qf, t0, wave == 0, lane);
    } else {
      fwd3_step<4, NTL, true>(st, p, pr.n, gK, gV, qf, t0, wave == 0, lane);
    }
  }
#pragma unroll
  for (int i = 0; i < NTL; ++i) {
    const int rq = (wave + F3W * i) * 16 + i16;
    const float m = st[i].m;
    const float l = group_sum(st[i].l);
    if (rq >= p.R) continue;
    if (rq < p.M) {
      float* part = p.ws0 + ((int64_t)prob * p.M + rq) * PART;
      if (g == 0) { part[0] = m; part[1] = l; }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(part + 2 + dt * 16 + 4 * g, st[i].o[dt]);
      continue;
    }
    const int tok = tok_of(p, pr.n, rq);
    const float inv = 1.0f / l;
    bf16_t* orow = p.out + ((int64_t)pr.b * p.S + tok) * p.ldo + pr.h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(orow + dt * 16 + 4 * g, st[i].o[dt] * inv);
    if (g == 0) {
      float* sp = p.stats + (((int64_t)pr.b * p.H + pr.h) * p.S + tok) * 2;
      sp[0] = m; sp[1] = __logf(l);
    }
  }
}

__global__ __launch_bounds__(F3THR, 2) void attn_fwd3_kernel(AP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsub = (p.R + 15) / 16;                       // sixteen-row tiles of the problem (<= 13)
  const int nt = (nsub - wave + F3W - 1) / F3W;           // this wave's query tiles: wave, wave + 8
  int prob = blockIdx.x;
  if (prob >= p.nprob) return;
  int cur = 0;
  bf16x8 qf[F3T][2], qn[F3T][2];
  {
    const Prob pr(p, prob);
    fwd3_stage_kv(smem, smem + FG * 128, p, pr, lane, wave);
    fwd3_load_q(qf, p, pr, wave, lane);
  }
  unsigned long long* tr = (p.ws2 && (int)blockIdx.x == (int)gridDim.x / 2 && (wave == 0 || wave == 7) && lane == 0)
                               ? reinterpret_cast<unsigned long long*>(p.ws2) + (wave ? 64 : 0) : nullptr;   // xp_debug_set_attn_trace
  int it = 0;
  for (;;) {
    if (tr && it < 8) tr[it * 4 + 0] = __builtin_amdgcn_s_memtime();
    __syncthreads();                                      // (vmcnt(0) + barrier) this problem's K/V have landed everywhere, and
    if (tr && it < 8) tr[it * 4 + 1] = __builtin_amdgcn_s_memtime();
    const Prob pr(p, prob);                               // every wave is done with the other buffer
    const int next = prob + (int)gridDim.x;
    char* gK = smem + cur * F3_BUF;
    if (next < p.nprob) {
      const Prob pn(p, next);
      char* nK = smem + (cur ^ 1) * F3_BUF;
      fwd3_stage_kv(nK, nK + FG * 128, p, pn, lane, wave);
      fwd3_load_q(qn, p, pn, wave, lane);
    }
    if (tr && it < 8) tr[it * 4 + 2] = __builtin_amdgcn_s_memtime();
    if (nt == 2)      fwd3_problem<2>(p, pr, prob, gK, gK + FG * 128, qf, nsub, wave, lane);
    else if (nt == 1) fwd3_problem<1>(p, pr, prob, gK, gK + FG * 128, qf, nsub, wave, lane);
    if (tr && it < 8) tr[it * 4 + 3] = __builtin_amdgcn_s_memtime();
    ++it;
    if (next >= p.nprob) break;
#pragma unroll
    for (int i = 0; i < F3T; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) qf[i][kk] = qn[i][kk];
    prob = next; cur ^= 1;
  }
}

// ============================================================================================ forward, persistent, multi-group
// Round 4: problems whose key window does not fit one LDS group (448^2 frames: R = 4 + 784 = 788 rows, BASELINE configs[3]).  The
// 7-wave kernel gives every 112-row query block its own workgroup and each of them stages the whole K/V of the problem (8 x 202 KB per
// problem, no overlap of staging and arithmetic inside a workgroup: 242 us per layer at configs[3], a quarter of its step).  Here the
// structure of attn_fwd3_kernel is kept -- one persistent 8-wave workgroup per CU, K/V groups of up to 208 rows staged by LDS-DMA into
// one of two 52 KiB buffers while the previous group is computed, two query tiles per wave sharing every K / V^T fragment read -- with
// a work ITEM = (problem, block of 16 query tiles = 256 rows) and an inner walk over the problem's key groups with the online-softmax
// state carried across them: K/V are staged ceil(tiles / 16) = 4 x per problem instead of 8 x, under the arithmetic.  The four blocks of
// a problem run at the same step on four workgroups of ONE XCD (ids 8 apart), so three of the four K/V fetches are L2 hits; the
// block roles rotate with the step so that the short last block (2 of 16 tiles at 788 rows) visits every workgroup equally.
struct F4Stage { int prob, qb, kg; bool valid; };
__device__ __forceinline__ F4Stage f4_stage(const AP& p, int s, int nqb, int ngrp) {
  // stream position s of this workgroup: item = s / ngrp (one per step), key group = s % ngrp
  F4Stage st;
  const int step = s / ngrp;
  st.kg = s - step * ngrp;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;              // block b runs on XCD b % 8 (speed only)
  const int quads_per_xcd = ((int)gridDim.x >> 3) / nqb;             // workgroups of an XCD in groups of nqb
  const int quad = j / nqb, role = j - quad * nqb;
  st.qb = (role + step) % nqb;
  st.prob = (step * quads_per_xcd + quad) * 8 + xcd;
  st.valid = quad < quads_per_xcd && st.prob < p.nprob;
  return st;
}

__global__ __launch_bounds__(F3THR, 2) void attn_fwd4_kernel(AP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (p.R + 15) / 16;                     // sixteen-row tiles of a problem
  const int nqb = (ntiles + 2 * F3W - 1) / (2 * F3W);     // query blocks of 16 tiles
  const int ngrp = (ntiles + FG / 16 - 1) / (FG / 16);    // key groups of 13 sub-tiles
  int s = 0, cur = 0;
  F4Stage sg = f4_stage(p, 0, nqb, ngrp);
  if (!sg.valid) return;                                  // (workgroups beyond the last whole quad of their XCD, or no problem left)
  bf16x8 qf[F3T][2], qn[F3T][2];
  {
    const Prob pr(p, sg.prob);
    fwd3_stage_kv(smem, smem + FG * 128, p, pr, lane, wave, 0);
    fwd3_load_q(qf, p, pr, wave, lane, sg.qb * 2 * F3W);
  }
  FwdState st[F3T];
  for (;;) {
    __syncthreads();                                      // (vmcnt(0) + barrier) this stage's K/V landed; the other buffer is free
    const Prob pr(p, sg.prob);
    const F4Stage nx = f4_stage(p, s + 1, nqb, ngrp);
    char* gK = smem + cur * F3_BUF;
    if (nx.valid) {
      const Prob pn(p, nx.prob);
      char* nK = smem + (cur ^ 1) * F3_BUF;
      fwd3_stage_kv(nK, nK + FG * 128, p, pn, lane, wave, nx.kg * FG);
      if (nx.kg == 0) fwd3_load_q(qn, p, pn, wave, lane, nx.qb * 2 * F3W);
    }
    const int tile0 = sg.qb * 2 * F3W;
    const int nt = tile0 + wave >= ntiles ? 0 : (tile0 + wave + F3W >= ntiles ? 1 : 2);       // this wave's query tiles in the item
    if (sg.kg == 0) {
#pragma unroll
      for (int i = 0; i < F3T; ++i) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) st[i].o[dt] = f32x4{0, 0, 0, 0};
        st[i].m = M_INIT; st[i].l = 0.f;
      }
    }
    // the key steps of this group: ng sub-tiles starting at global key row kb0
    const int kb0 = sg.kg * FG;
    const int ng = ntiles - sg.kg * (FG / 16) < FG / 16 ? ntiles - sg.kg * (FG / 16) : FG / 16;
    const bool corner = sg.qb == 0 && wave == 0 && sg.kg == 0;
    auto run = [&](auto ntl_c) {
      constexpr int NTL = decltype(ntl_c)::value;
      int t0 = 0;
      for (; t0 + 4 <= ng && kb0 + (t0 + 4) * 16 <= p.R; t0 += 4) fwd3_step<4, NTL, false>(st, p, pr.n, gK, gK + FG * 128, qf, t0, corner, lane, kb0);
      if (t0 < ng) {
        if (ng - t0 == 1) {
          if (kb0 + ng * 16 > p.R) fwd3_step<1, NTL, true>(st, p, pr.n, gK, gK + FG * 128, qf, t0, corner, lane, kb0);
          else                     fwd3_step<1, NTL, false>(st, p, pr.n, gK, gK + FG * 128, qf, t0, corner, lane, kb0);
        } else {
          fwd3_step<4, NTL, true>(st, p, pr.n, gK, gK + FG * 128, qf, t0, corner, lane, kb0);
        }
      }
    };
    if (nt == 2)      run(std::integral_constant<int, 2>{});
    else if (nt == 1) run(std::integral_constant<int, 1>{});
    if (sg.kg == ngrp - 1) {                               // last key group of the item: normalise and store
#pragma unroll
      for (int i = 0; i < F3T; ++i) {
        if (i >= nt) continue;
        const int rq = (tile0 + wave + F3W * i) * 16 + i16;
        const float m = st[i].m;
        const float l = group_sum(st[i].l);
        if (rq >= p.R) continue;
        if (rq < p.M) {
          float* part = p.ws0 + ((int64_t)sg.prob * p.M + rq) * PART;
          if (g == 0) { part[0] = m; part[1] = l; }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) store4(part + 2 + dt * 16 + 4 * g, st[i].o[dt]);
          continue;
        }
        const int tok = tok_of(p, pr.n, rq);
        const float inv = 1.0f / l;
        bf16_t* orow = p.out + ((int64_t)pr.b * p.S + tok) * p.ldo + pr.h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(orow + dt * 16 + 4 * g, st[i].o[dt] * inv);
        if (g == 0) {
          float* sp = p.stats + (((int64_t)pr.b * p.H + pr.h) * p.S + tok) * 2;
          sp[0] = m; sp[1] = __logf(l);
        }
      }
    }
    if (!nx.valid) break;
    if (nx.kg == 0) {
#pragma unroll
      for (int i = 0; i < F3T; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[i][kk] = qn[i][kk];
    }
    sg = nx; ++s; cur ^= 1;
  }
}

// merge the per-frame partials of the proxy query rows: grid = B*H*M, 4 waves x 64 lanes (lane = d).  Every wave finds the global
// row maximum itself (lane n loads frame n's maximum: one load round), then wave w combines frames n = w, w + 4, ... (independent
// loads, no serial chain over the frames) and the four partial sums are added in wave order through LDS (fixed order).
__global__ __launch_bounds__(256) void attn_fwd_merge_kernel(AP p) {
  __shared__ float red[4][DH + 1];
  const int d = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int mrow = blockIdx.x % p.M, bh = blockIdx.x / p.M, h = bh % p.H, b = bh / p.H;
  const float* base = p.ws0 + ((int64_t)bh * p.N * p.M + mrow) * PART;      // frame n: base + n * M * PART
  const int64_t fstride = (int64_t)p.M * PART;
  float mx = -INFINITY;
  for (int n0 = 0; n0 < p.N; n0 += 64) mx = fmaxf(mx, (n0 + d < p.N) ? base[(n0 + d) * fstride] : -INFINITY);
  mx = wave_max(mx);
  float l = 0.f, acc = 0.f;
  for (int n = w; n < p.N; n += 4) {
    const float* part = base + n * fstride;
    const float wgt = __expf(part[0] - mx);
    l += part[1] * wgt; acc += part[2 + d] * wgt;
  }
  red[w][d] = acc;
  if (d == 0) red[w][DH] = l;
  __syncthreads();
  if (w != 0) return;
  acc = ((red[0][d] + red[1][d]) + red[2][d]) + red[3][d];
  l = ((red[0][DH] + red[1][DH]) + red[2][DH]) + red[3][DH];
  const int64_t tok = (int64_t)b * p.S + mrow;
  p.out[tok * p.ldo + h * DH + d] = (bf16_t)(acc / l);
  if (d == 0) { float* st = p.stats + (((int64_t)b * p.H + h) * p.S + mrow) * 2; st[0] = mx; st[1] = __logf(l); }
}

// ============================================================================================ backward
// Both backward kernels use the forward's decomposition: 7 waves x 16 OWN rows per workgroup, the OTHER dimension staged
// in LDS in groups of up to 208 rows by branch-free buffer loads, 64-row steps with a 16-row tail (16x16x16 MFMA), query
// blocks of one problem 8 workgroup ids apart.  LDS 53-56 KiB: two workgroups per CU.

// cooperative load of rows [row0, row0 + nrows) of TWO row-major operands (own resource / pitch each) into linear swizzled
// images; rows >= R or >= nrows read as zero.
__device__ __forceinline__ void stage_rows2(char* gA, char* gB, __amdgpu_buffer_rsrc_t ra, unsigned lda_bytes,
                                            __amdgpu_buffer_rsrc_t rb, unsigned ldb_bytes, const AP& p, const Prob& pr,
                                            int row0, int nrows, int tid) {
  constexpr int NJ = (FG * 8 + FTHR - 1) / FTHR;
  u32x4 va[NJ], vb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int e = tid + j * FTHR, row = e >> 3, c = e & 7, r = row0 + row;
    const bool ok = row < nrows && r < p.R;
    const unsigned tok = ok ? (unsigned)tok_of(p, pr.n, r) : 0u;
    va[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? tok * lda_bytes + c * 16 : 0xFFFFFF00u, 0, 0);
    vb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? tok * ldb_bytes + c * 16 : 0xFFFFFF00u, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int e = tid + j * FTHR, row = e >> 3, c = e & 7;
    if (row < FG) {
      *reinterpret_cast<u32x4*>(gA + tile128_off(row, c)) = va[j];
      *reinterpret_cast<u32x4*>(gB + tile128_off(row, c)) = vb[j];
    }
  }
}

// workgroup id -> (problem, own-row block); false if the id is padding
__device__ __forceinline__ bool wg_problem(const AP& p, int& prob, int& blk) {
  const int slot = blockIdx.x >> 3;
  prob = (slot / p.nq) * 8 + (blockIdx.x & 7);
  blk = slot % p.nq;
  return prob < p.nprob;
}

// Column sums over the workgroup's FINISHED rows (one row per lane column i16, d = 16*dt + 4*g + r) of values already rounded to
// the storage type: butterfly over the 16 row lanes, one LDS slot per wave, thread d sums the 7 waves in fixed order and writes
// dst[d].  Every thread of the workgroup must call it (two barriers); `red` may alias the staging tiles (no longer read).
__device__ __forceinline__ void wg_colsum64(f32x4 (&v)[4], float* red, float* dst, int wave, int lane, int tid) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[dt][r] += __shfl_xor(v[dt][r], o, 64);
  __syncthreads();
  if ((lane & 15) == 0) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(red + wave * DH + dt * 16 + 4 * (lane >> 4), v[dt]);
  }
  __syncthreads();
  if (tid < DH) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < FW; ++w) a += red[w * DH + tid];
    dst[tid] = a;
  }
}
__device__ __forceinline__ f32x4 round_bf16(f32x4 x) {
  return f32x4{(float)(bf16_t)x[0], (float)(bf16_t)x[1], (float)(bf16_t)x[2], (float)(bf16_t)x[3]};
}

// ---- dK, dV: the workgroup owns 112 key rows; loops over the problem's query rows (Q, dO, m, log l, delta staged) ----
struct DkvState { f32x4 dk[4], dv[4]; };

// masked scores for the (own key, staged queries) orientation -- a real call for the same reason as fwd_mask_scores
__device__ __attribute__((noinline)) S16 dkv_mask_scores(S16 v, int ns, int R, int M, int mode, int n, int rk, int kvalid,
                                                         int kpad, int qb, int g) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rqq = qb + t * 16 + 4 * g + r;
      float x = v.s[t][r];
      if (t < ns) {
        if (kpad) x = F32_MIN;
        const bool ok = mode == XP_ATTN_PROXY ? !(rqq < M && rk < M && n != 0) : rk <= rqq;
        if (!(kvalid && rqq < R && ok)) x = -INFINITY;          // exp(-inf - m - log l) = 0
      }
      v.s[t][r] = x;
    }
  return v;
}

template <int NS>
__device__ __forceinline__ void dkv_step(DkvState& st, const AP& p, const Prob& pr, const char* gQ, const char* gDO,
                                         const float* gM, const float* gLg, const float* gDl, const bf16x8 (&kf)[2],
                                         const bf16x8 (&vf)[2], int t0, int qb, int rk, bool kvalid, bool kpad, int wkey0,
                                         int lane) {
  const int g = lane >> 4;
  f32x4 pp[NS], ds[NS];
  // mask-free path unless a padding mask exists, the causal band crosses this (query step, key wave) pair, or proxy
  // queries meet proxy keys in a frame n != 0.  Query rows >= R have zero-filled Q/dO rows and (m, log l) = 0, so
  // their P is finite and multiplies zeros; key lanes >= R only produce their own (discarded) columns.
  const bool need_mask = p.pad != nullptr ||
      (p.mode == XP_ATTN_CAUSAL ? qb < wkey0 + 15 : (pr.n != 0 && wkey0 < p.M && qb < p.M));
  f32x4 sc[NS], dp[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    sc[t] = f32x4{0, 0, 0, 0}; dp[t] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      sc[t] = mma16(frag_rows(gQ, t0 + t, kk, lane), kf[kk], sc[t]);
      dp[t] = mma16(frag_rows(gDO, t0 + t, kk, lane), vf[kk], dp[t]);
    }
  }
  if (need_mask) {
    S16 v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v.s[t] = t < NS ? sc[t < NS ? t : 0] : f32x4{0, 0, 0, 0};
    v = dkv_mask_scores(v, NS, p.R, p.M, p.mode, pr.n, rk, kvalid ? 1 : 0, kpad ? 1 : 0, qb, g);
#pragma unroll
    for (int t = 0; t < NS; ++t) sc[t] = v.s[t];
  }
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    // the lane's 4 query rows are consecutive: one 16-byte LDS read per statistic
    const f32x4 m4 = *reinterpret_cast<const f32x4*>(gM + (t0 + t) * 16 + 4 * g);
    const f32x4 lg4 = *reinterpret_cast<const f32x4*>(gLg + (t0 + t) * 16 + 4 * g);
    const f32x4 dl4 = *reinterpret_cast<const f32x4*>(gDl + (t0 + t) * 16 + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pv = __expf((sc[t][r] - m4[r]) - lg4[r]);
      pp[t][r] = pv;
      ds[t][r] = pv * (dp[t][r] - dl4[r]);
    }
  }
  const char* sQ = gQ + t0 * 16 * 128;
  const char* sDO = gDO + t0 * 16 * 128;
#pragma unroll
  for (int c = 0; c < NS / 2; ++c) {
    const bf16x8 pf = pack_p(pp[2 * c], pp[2 * c + 1]), sf = pack_p(ds[2 * c], ds[2 * c + 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      st.dv[dt] = mma16(frag_cols(sDO, dt, c, lane), pf, st.dv[dt]);
      st.dk[dt] = mma16(frag_cols(sQ, dt, c, lane), sf, st.dk[dt]);
    }
  }
  if constexpr (NS & 1) {
    const s16x4 pf = pack_p4(pp[NS - 1]), sf = pack_p4(ds[NS - 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      st.dv[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(frag_cols16(sDO, dt, NS - 1, lane), pf, st.dv[dt], 0, 0, 0);
      st.dk[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(frag_cols16(sQ, dt, NS - 1, lane), sf, st.dk[dt], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(FTHR, 4) void attn_bwd_dkv_kernel(AP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * FG * 128 + 3 * FG * 4];
  char* gQ = smem; char* gDO = smem + FG * 128;
  float* gM = reinterpret_cast<float*>(smem + 2 * FG * 128); float* gLg = gM + FG; float* gDl = gLg + FG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
  int prob, blk;
  if (!wg_problem(p, prob, blk)) return;
  const Prob pr(p, prob);
  const int kb = blk * FQ;
  const int rk = kb + wave * 16 + i16;            // this lane's key row (column of S)
  const bool kvalid = rk < p.R;
  const int64_t ktok = (int64_t)pr.b * p.S + tok_of(p, pr.n, kvalid ? rk : 0);
  const bool kpad = kvalid && p.pad && p.pad[ktok] == 0;

  bf16x8 kf[2], vf[2];
  load_row_frag(kf, p.qkv + ktok * p.ldqkv + (int64_t)p.H * DH + pr.h * DH, kvalid, g);
  load_row_frag(vf, p.qkv + ktok * p.ldqkv + (int64_t)2 * p.H * DH + pr.h * DH, kvalid, g);

  DkvState st;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { st.dk[dt] = f32x4{0, 0, 0, 0}; st.dv[dt] = f32x4{0, 0, 0, 0}; }

  const int nsub = (p.R + 15) / 16;
  const int s0 = p.mode == XP_ATTN_CAUSAL ? (kb / 64) * 4 : 0;     // queries before the first key never see it
  const bf16_t* qbase = p.qkv + (int64_t)pr.b * p.S * p.ldqkv + pr.h * DH;
  const bf16_t* dobase = p.dout + (int64_t)pr.b * p.S * p.ldo + pr.h * DH;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(qbase), 0, (unsigned)(((int64_t)p.S * p.ldqkv - pr.h * DH) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(dobase), 0, (unsigned)(((int64_t)p.S * p.ldo - pr.h * DH) * 2), 0x00020000);
  const int wkey0 = kb + wave * 16;
  const bool wave_active = wkey0 < p.R;
  for (int g0 = s0; g0 < nsub; g0 += FG / 16) {
    const int ng = nsub - g0 < FG / 16 ? nsub - g0 : FG / 16;
    if (g0 != s0) __syncthreads();
    stage_rows2(gQ, gDO, rq, (unsigned)(p.ldqkv * 2), rdo, (unsigned)(p.ldo * 2), p, pr, g0 * 16, ng * 16, tid);
    if (tid < FG) {
      const int r = g0 * 16 + tid;
      float mm = 0.f, lg = 0.f, dl = 0.f;
      if (tid < ng * 16 && r < p.R) {
        const int64_t si = ((int64_t)pr.b * p.H + pr.h) * p.S + tok_of(p, pr.n, r);
        mm = p.stats[si * 2]; lg = p.stats[si * 2 + 1]; dl = p.ws0[si];
      }
      gM[tid] = mm; gLg[tid] = lg; gDl[tid] = dl;
    }
    __syncthreads();
    if (!wave_active) continue;
    for (int t0 = 0; t0 < ng; t0 += 2) {                // 32 query rows per step: four accumulator sets + S/dP/P/dS of
      const int qb = (g0 + t0) * 16;                    // 64 rows would not fit 128 registers
      if (ng - t0 == 1) dkv_step<1>(st, p, pr, gQ, gDO, gM, gLg, gDl, kf, vf, t0, qb, rk, kvalid, kpad, wkey0, lane);
      else              dkv_step<2>(st, p, pr, gQ, gDO, gM, gLg, gDl, kf, vf, t0, qb, rk, kvalid, kpad, wkey0, lane);
    }
  }
  const bool partial_row = kvalid && p.mode == XP_ATTN_PROXY && rk < p.M;      // proxy keys: per-frame partials, reduced later
  const bool final_row = kvalid && !partial_row;
  if (partial_row) {
    float* part = p.ws2 + ((int64_t)prob * p.M + rk) * (2 * DH);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { store4(part + dt * 16 + 4 * g, st.dk[dt]); store4(part + DH + dt * 16 + 4 * g, st.dv[dt]); }
  }
  if (final_row) {
    bf16_t* base = p.dqkv + ktok * p.ldqkv + pr.h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store4(base + (int64_t)p.H * DH + dt * 16 + 4 * g, st.dk[dt]);
      store4(base + (int64_t)2 * p.H * DH + dt * 16 + 4 * g, st.dv[dt]);
    }
  }
  if (p.cs) {          // bias gradients of k_proj / v_proj: column sums of the rows just stored (as stored: rounded)
    float* row = p.cs + ((int64_t)(pr.b * p.N + pr.n) * p.nq + blk) * (3 * p.H * DH) + pr.h * DH;
    f32x4 v[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) v[dt] = final_row ? round_bf16(st.dk[dt]) : f32x4{0, 0, 0, 0};
    wg_colsum64(v, reinterpret_cast<float*>(smem), row + p.H * DH, wave, lane, tid);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) v[dt] = final_row ? round_bf16(st.dv[dt]) : f32x4{0, 0, 0, 0};
    wg_colsum64(v, reinterpret_cast<float*>(smem), row + 2 * p.H * DH, wave, lane, tid);
  }
}

// ---- dQ: the workgroup owns 112 query rows; loops over the key rows (K, V staged; transposed orientation) -----------
template <int NS>
__device__ __forceinline__ void dq_step(f32x4 (&dq)[4], const AP& p, const Prob& pr, const char* gK, const char* gV,
                                        const unsigned char* gPad, const bf16x8 (&qf)[2], const bf16x8 (&dof)[2], float mq,
                                        float lgq, float dlq, int t0, int kb, int rq, bool qvalid, int wrow0, int lane) {
  const int g = lane >> 4;
  f32x4 ds[NS];
  const bool need_mask = kb + NS * 16 > p.R || p.pad != nullptr ||
      (p.mode == XP_ATTN_CAUSAL ? kb + NS * 16 - 1 > wrow0 : (pr.n != 0 && kb < p.M && wrow0 < p.M));
  f32x4 sc[NS], dp[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    sc[t] = f32x4{0, 0, 0, 0}; dp[t] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      sc[t] = mma16(frag_rows(gK, t0 + t, kk, lane), qf[kk], sc[t]);
      dp[t] = mma16(frag_rows(gV, t0 + t, kk, lane), dof[kk], dp[t]);
    }
  }
  if (need_mask) {
    S16 v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v.s[t] = t < NS ? sc[t < NS ? t : 0] : f32x4{0, 0, 0, 0};
    v = fwd_mask_scores(v, NS, gPad + t0 * 16, p.R, p.M, p.mode, pr.n, rq, qvalid ? 1 : 0, kb, g);
#pragma unroll
    for (int t = 0; t < NS; ++t) sc[t] = v.s[t];
  }
#pragma unroll
  for (int t = 0; t < NS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[t][r] = __expf((sc[t][r] - mq) - lgq) * (dp[t][r] - dlq);
  const char* sK = gK + t0 * 16 * 128;
#pragma unroll
  for (int c = 0; c < NS / 2; ++c) {
    const bf16x8 sf = pack_p(ds[2 * c], ds[2 * c + 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = mma16(frag_cols(sK, dt, c, lane), sf, dq[dt]);
  }
  if constexpr (NS & 1) {
    const s16x4 sf = pack_p4(ds[NS - 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      dq[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(frag_cols16(sK, dt, NS - 1, lane), sf, dq[dt], 0, 0, 0);
  }
}

__global__ __launch_bounds__(FTHR, 4) void attn_bwd_dq_kernel(AP p) {
  __shared__ __attribute__((aligned(16))) char smem[F_LDS];
  char* gK = smem; char* gV = smem + FG * 128; unsigned char* gPad = reinterpret_cast<unsigned char*>(smem + 2 * FG * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
  int prob, blk;
  if (!wg_problem(p, prob, blk)) return;
  const Prob pr(p, prob);
  const int qb = blk * FQ;
  const int rq = qb + wave * 16 + i16;
  const bool qvalid = rq < p.R;
  const int64_t qtok = (int64_t)pr.b * p.S + tok_of(p, pr.n, qvalid ? rq : 0);

  bf16x8 qf[2], dof[2];
  load_row_frag(qf, p.qkv + qtok * p.ldqkv + pr.h * DH, qvalid, g);
  load_row_frag(dof, p.dout + qtok * p.ldo + pr.h * DH, qvalid, g);
  // delta = sum_d dO * O of the own row, from the fragments already in registers (each of the 4 lanes of a row holds 16
  // of the 64 d); published in ws0 for the dK/dV kernel, which runs after this one
  bf16x8 of[2];
  load_row_frag(of, p.out + qtok * p.ldo + pr.h * DH, qvalid, g);
  float dlq = 0.f;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int e = 0; e < 8; ++e) dlq += (float)dof[kk][e] * (float)of[kk][e];
  dlq = group_sum(dlq);
  float mq = 0.f, lgq = 0.f;
  if (qvalid) {
    const int64_t si = ((int64_t)pr.b * p.H + pr.h) * p.S + tok_of(p, pr.n, rq);
    mq = p.stats[si * 2]; lgq = p.stats[si * 2 + 1];
    if (g == 0 && (p.mode != XP_ATTN_PROXY || rq >= p.M || pr.n == 0)) p.ws0[si] = dlq;   // proxy rows: frame 0 writes
  }
  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0, 0, 0, 0};

  int nsub = (p.R + 15) / 16;
  if (p.mode == XP_ATTN_CAUSAL) { const int lim = (qb + FQ - 1) / 16 + 1; nsub = nsub < lim ? nsub : lim; }
  const bf16_t* kbase = p.qkv + (int64_t)pr.b * p.S * p.ldqkv + (int64_t)p.H * DH + pr.h * DH;
  const unsigned ld_bytes = (unsigned)(p.ldqkv * 2);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(kbase), 0, (unsigned)((int64_t)p.S * p.ldqkv * 2 - ((int64_t)p.H * DH + pr.h * DH) * 2), 0x00020000);
  const unsigned voff_v = (unsigned)(p.H * DH * 2);
  const int wrow0 = qb + wave * 16;
  const bool wave_active = wrow0 < p.R;
  for (int g0 = 0; g0 < nsub; g0 += FG / 16) {
    const int ng = nsub - g0 < FG / 16 ? nsub - g0 : FG / 16;
    if (g0) __syncthreads();
    fwd_load_kv(gK, gV, rs, ld_bytes, voff_v, p, pr, g0 * 16, ng * 16, tid);
    if (tid < FG) { const int r = g0 * 16 + tid; gPad[tid] = (p.pad && r < p.R) ? (p.pad[(int64_t)pr.b * p.S + r] == 0) : 0; }
    __syncthreads();
    if (!wave_active) continue;
    for (int t0 = 0; t0 < ng; t0 += 4) {
      const int ns = ng - t0 < 4 ? ng - t0 : 4;
      const int kb = (g0 + t0) * 16;
      if (ns == 1) dq_step<1>(dq, p, pr, gK, gV, gPad, qf, dof, mq, lgq, dlq, t0, kb, rq, qvalid, wrow0, lane);
      else         dq_step<4>(dq, p, pr, gK, gV, gPad, qf, dof, mq, lgq, dlq, t0, kb, rq, qvalid, wrow0, lane);
    }
  }
  const bool partial_row = qvalid && p.mode == XP_ATTN_PROXY && rq < p.M;
  const bool final_row = qvalid && !partial_row;
  if (partial_row) {
    float* part = p.ws1 + ((int64_t)prob * p.M + rq) * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(part + dt * 16 + 4 * g, dq[dt]);
  }
  if (final_row) {
    bf16_t* base = p.dqkv + qtok * p.ldqkv + pr.h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(base + dt * 16 + 4 * g, dq[dt] * p.q_scale);
  }
  if (p.cs) {          // bias gradient of q_proj
    float* row = p.cs + ((int64_t)(pr.b * p.N + pr.n) * p.nq + blk) * (3 * p.H * DH) + pr.h * DH;
    f32x4 v[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) v[dt] = final_row ? round_bf16(dq[dt] * p.q_scale) : f32x4{0, 0, 0, 0};
    wg_colsum64(v, reinterpret_cast<float*>(smem), row, wave, lane, tid);
  }
}

// proxy tokens: sum the per-frame partials.  grid = B*H*M, 4 waves x 64 lanes (lane = d); wave w sums frames n = w, w + 4, ...,
// the four partial sums are added in wave order through LDS (fixed order)
__global__ __launch_bounds__(256) void attn_bwd_proxy_reduce_kernel(AP p) {
  __shared__ float red[4][3][DH];
  const int d = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int mrow = blockIdx.x % p.M, bh = blockIdx.x / p.M, h = bh % p.H, b = bh / p.H;
  float q = 0.f, k = 0.f, v = 0.f;
  for (int n = w; n < p.N; n += 4) {
    const int64_t pi = ((int64_t)bh * p.N + n) * p.M + mrow;
    q += p.ws1[pi * DH + d];
    k += p.ws2[pi * 2 * DH + d];
    v += p.ws2[pi * 2 * DH + DH + d];
  }
  red[w][0][d] = q; red[w][1][d] = k; red[w][2][d] = v;
  __syncthreads();
  if (w != 0) return;
  q = ((red[0][0][d] + red[1][0][d]) + red[2][0][d]) + red[3][0][d];
  k = ((red[0][1][d] + red[1][1][d]) + red[2][1][d]) + red[3][1][d];
  v = ((red[0][2][d] + red[1][2][d]) + red[2][2][d]) + red[3][2][d];
  bf16_t* base = p.dqkv + ((int64_t)b * p.S + mrow) * p.ldqkv + h * DH + d;
  const bf16_t qb = (bf16_t)(q * p.q_scale), kb = (bf16_t)k, vb = (bf16_t)v;
  base[0] = qb;
  base[(int64_t)p.H * DH] = kb;
  base[(int64_t)2 * p.H * DH] = vb;
  if (p.cs) {          // the proxy rows' contribution to the bias column sums: one partial row per (sample, proxy token)
    float* row = p.cs + ((int64_t)p.cs_main + b * p.M + mrow) * (3 * p.H * DH) + h * DH + d;
    row[0] = (float)qb; row[(int64_t)p.H * DH] = (float)kb; row[(int64_t)2 * p.H * DH] = (float)vb;
  }
}

// ============================================================================================ backward, fused (round 6)
// The two kernels above are issue-bound, not HBM-bound (same time with their operands in the Infinity Cache): per SIMD, 3.5 one-tile
// waves each read every staged fragment for themselves, the softmax arithmetic is 6.5 VALU per score, and every problem is staged
// twice per kernel (two 112-row blocks) -- 434 KB through the CU per problem for 205 KB of operands and results.  attn_bwd5_kernel is
// ONE launch for dQ, dK and dV of the proxy problems that fit one LDS group (R <= 208, M <= 16, no padding mask -- the shapes of
// attn_fwd3_kernel):
//   * one persistent 8-wave workgroup per CU, problems handed out by a device counter (late workgroups -- the CUs a weight-gradient
//     GEMM of the other stream held at launch -- simply take fewer), every operand staged ONCE per problem by LDS-DMA:
//     X = {K, V} image, Y = {Q, dO} image, 26 KiB each;
//   * phase A (own = 2 query tiles per wave, S^T orientation, reads X) -> dQ; phase B (own = 2 key tiles per wave, reads Y, own K / V
//     fragments taken from X) -> dK, dV.  A true single pass (5 matmuls, one exp) needs the cross-wave dQ sum staged through LDS
//     beside a prefetched second problem: 163.7 of the 160 KiB; what the recompute costs here is MFMA time the HBM
//     stream hides, what it saves is every barrier inside a problem.  X of the NEXT problem lands during phase B, Y during phase A:
//     two barriers per problem, both of them the points where a DMA must have landed anyway;
//   * both tiles of a wave share every fragment read (half the LDS traffic per score of the one-tile kernels);
//   * softmax arithmetic 3.5-4 VALU per score: the row constants ride in as the MFMA's C operand (S starts from -(m + log l), dP from
//     -delta), P = exp2(S * log2 e), dS = P * dP;
//   * delta = rowsum(dO * O) stays in the workgroup (LDS) -- no global publication, no ordering between two launches;
//   * the q/k/v bias column sums ride on the two barriers a problem has anyway.
// Deterministic: fixed summation order everywhere, the counter only decides WHICH workgroup computes a problem.
constexpr int B5W = 8, B5THR = B5W * 64;
constexpr int B5_ROWS = FG + 16;                               // 13 staged tiles + one pad tile of zeros: every step is two sub-tiles
constexpr int B5_IMG = B5_ROWS * 128;                          // one operand image
constexpr int B5_OFF_STATS = 4 * B5_IMG;                       // c[B5_ROWS] = -(m + log l), nd[B5_ROWS] = -delta
constexpr int B5_OFF_RED = B5_OFF_STATS + 2 * B5_ROWS * 4;     // column-sum partials [3][B5W][DH]
constexpr int B5_OFF_NEXT = B5_OFF_RED + 3 * B5W * DH * 4;
constexpr int B5_STG_ROW = 128 + 16;                           // output staging: 128-byte rows, skewed by 16 bytes against bank conflicts
constexpr int B5_STG_WAVE = 32 * B5_STG_ROW;                   // one wave's two finished 16 x 64 tiles
constexpr int B5_OFF_STG = B5_OFF_NEXT + 16;
constexpr int B5_LDS = B5_OFF_STG + B5W * B5_STG_WAVE;

// rows [0, FG) of one row-major operand (the 64 elements of a (token, head) slice; rows >= R read as zero) -> linear swizzled LDS
// image by LDS-DMA, one wave instruction = 8 rows x 128 B.  The per-lane part of the source offset does not depend on the problem
// (vrow: byte offset of the lane's row inside a 64-row block + its swizzled 16-byte chunk); the frame and the 64-row block ride in
// the scalar offset, so a problem costs the loader no vector registers beyond the proxy rows of block 0 (voff0).
// `j`: ONE of the four 64-row blocks (a wave instruction each): the caller spreads the pieces over its step loop -- issued in one
// burst, the 8 instructions of a wave cost it ~3k cycles of queueing in front of the first MFMA (tools/attn_bwd_trace.py).
__device__ __forceinline__ void b5_stage_piece(char* img, __amdgpu_buffer_rsrc_t rs, unsigned ld_bytes, unsigned soff, unsigned vrow,
                                               unsigned voff0, int R, unsigned frame_off, int lane, int wave, int j) {
  typedef __attribute__((address_space(3))) char lds_c;
  constexpr int NPASS = FG / 8;
  const int pass = j * B5W + wave;
  if (pass < NPASS) {
    const int row = pass * 8 + (lane >> 3);
    const unsigned v = row < R ? (j == 0 ? voff0 : vrow) : 0xFFFFFF00u;
    const unsigned so = j == 0 ? soff : soff + frame_off + (unsigned)(j * 64) * ld_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_c*)(img + pass * 1024), 16, v, so, 0, 0);
  }
}
constexpr int B5_PIECES = (FG / 8 + B5W - 1) / B5W;            // 4
struct B5Lane { unsigned vq, vo, v0q, v0o; };                 // per-lane source offsets for pitch ldqkv / ldo (block >= 1 / block 0)
__device__ __forceinline__ void b5_lane_offsets(B5Lane& L, const AP& p, int n, int lane, int wave) {
  const int row = wave * 8 + (lane >> 3);                       // the lane's row inside a 64-row block
  const unsigned c16 = (unsigned)(((lane & 7) ^ swz128(row)) * 16);
  L.vq = (unsigned)row * (unsigned)(p.ldqkv * 2) + c16;
  L.vo = (unsigned)row * (unsigned)(p.ldo * 2) + c16;
  const unsigned tok0 = (unsigned)tok_of(p, n, row);            // block 0 holds the proxy rows: token index by the general rule
  L.v0q = tok0 * (unsigned)(p.ldqkv * 2) + c16;
  L.v0o = tok0 * (unsigned)(p.ldo * 2) + c16;
}
// pieces [j0, j1) of the {K, V} image of problem pr / of its {Q, dO} image
__device__ __forceinline__ void b5_stage_kv(char* X, const AP& p, const Prob& pr, int lane, int wave, int j0 = 0, int j1 = B5_PIECES) {
  const bf16_t* kbase = p.qkv + (int64_t)pr.b * p.S * p.ldqkv + (int64_t)p.H * DH + pr.h * DH;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(kbase), 0, (unsigned)((int64_t)p.S * p.ldqkv * 2 - ((int64_t)p.H * DH + pr.h * DH) * 2), 0x00020000);
  B5Lane L; b5_lane_offsets(L, p, pr.n, lane, wave);
  const unsigned ld = (unsigned)(p.ldqkv * 2), fo = (unsigned)(pr.n * p.L) * ld;
  for (int j = j0; j < j1; ++j) {
    b5_stage_piece(X, rs, ld, 0, L.vq, L.v0q, p.R, fo, lane, wave, j);
    b5_stage_piece(X + B5_IMG, rs, ld, (unsigned)(p.H * DH * 2), L.vq, L.v0q, p.R, fo, lane, wave, j);
  }
}
__device__ __forceinline__ void b5_stage_qdo(char* Y, const AP& p, const Prob& pr, int lane, int wave, int j0 = 0, int j1 = B5_PIECES) {
  const bf16_t* qbase = p.qkv + (int64_t)pr.b * p.S * p.ldqkv + pr.h * DH;
  const bf16_t* dobase = p.dout + (int64_t)pr.b * p.S * p.ldo + pr.h * DH;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(qbase), 0, (unsigned)(((int64_t)p.S * p.ldqkv - pr.h * DH) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(dobase), 0, (unsigned)(((int64_t)p.S * p.ldo - pr.h * DH) * 2), 0x00020000);
  B5Lane L; b5_lane_offsets(L, p, pr.n, lane, wave);
  const unsigned ldq = (unsigned)(p.ldqkv * 2), ldo = (unsigned)(p.ldo * 2);
  for (int j = j0; j < j1; ++j) {
    b5_stage_piece(Y, rq, ldq, 0, L.vq, L.v0q, p.R, (unsigned)(pr.n * p.L) * ldq, lane, wave, j);
    b5_stage_piece(Y + B5_IMG, rdo, ldo, 0, L.vo, L.v0o, p.R, (unsigned)(pr.n * p.L) * ldo, lane, wave, j);
  }
}

// the wave's own query rows (tiles wave, wave + 8; rows >= R: zeros) straight from global memory: Q, dO, O fragments and (m, log l)
struct B5Own { bf16x8 q[2][2], d[2][2], o[2][2]; float m[2], lg[2]; };
__device__ __forceinline__ void b5_load_own(B5Own& w, const AP& p, const Prob& pr, int wave, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const bf16_t* qbase = p.qkv + (int64_t)pr.b * p.S * p.ldqkv + pr.h * DH;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(qbase), 0, (unsigned)(((int64_t)p.S * p.ldqkv - pr.h * DH) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.dout + (int64_t)pr.b * p.S * p.ldo + pr.h * DH), 0, (unsigned)(((int64_t)p.S * p.ldo - pr.h * DH) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.out + (int64_t)pr.b * p.S * p.ldo + pr.h * DH), 0, (unsigned)(((int64_t)p.S * p.ldo - pr.h * DH) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(
      p.stats + ((int64_t)pr.b * p.H + pr.h) * p.S * 2, 0, (unsigned)(p.S * 8), 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rq_ = (wave + B5W * i) * 16 + i16;
    const bool ok = rq_ < p.R;
    const unsigned tok = (unsigned)tok_of(p, pr.n, ok ? rq_ : 0);
    const unsigned oq = ok ? tok * (unsigned)(p.ldqkv * 2) + g * 16 : 0xFFFFFF00u;
    const unsigned oo = ok ? tok * (unsigned)(p.ldo * 2) + g * 16 : 0xFFFFFF00u;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      w.q[i][kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, oq + kk * 64, 0, 0));
      w.d[i][kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rdo, oo + kk * 64, 0, 0));
      w.o[i][kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ro, oo + kk * 64, 0, 0));
    }
    const unsigned os = ok ? tok * 8u : 0xFFFFFF00u;          // rows >= R: (0, 0).  (two dword loads: hipcc lowers the b64 builtin
    w.m[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rst, os, 0, 0));          //  to ONE dword, splat)
    w.lg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rst, os + 4, 0, 0));
  }
}

struct B5TFrag { bf16x8 f[4]; };                             // transposed A fragments of a 32-row step for the four d tiles
__device__ __forceinline__ void b5_tfrag(B5TFrag& t, const char* tile, int lane) {      // asm reads: wait lgkmcnt(0) before use
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) t.f[dt] = frag_cols_async(tile, dt, 0, lane);
}

// phase A step: two sixteen-key sub-tiles at LDS row t0 * 16 against the wave's two query tiles (S^T orientation: the lane owns
// query column i16 of each tile; rows 4g + r are keys).  tail: the step reaches past key R (those P are forced to zero: their K rows
// are zero-filled, but P * 0 must not see an overflowed P).  corner: the wave's first tile holds the proxy query rows and this is a
// frame n != 0 -- proxy x proxy scores are counted in frame 0 only.
template <bool TAIL>
__device__ __forceinline__ void b5_dq_step(f32x4 (&dq)[2][4], const AP& p, const char* tK, const B5Own& w,
                                           const f32x4 (&c4)[2], const f32x4 (&nd4)[2], int t0, bool corner, int lane) {
  // tK: the K image advanced to the step's first row (a multiple of 16 rows keeps the swizzle phase); V is B5_IMG further
  const int g = lane >> 4, i16 = lane & 15;
  const char* tV = tK + B5_IMG;
  f32x4 s[2][2], dp[2][2];
  {
    bf16x8 kf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) kf[t][kk] = frag_rows(tK, t, kk, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) s[i][t] = mma16(kf[t][1], w.q[i][1], mma16(kf[t][0], w.q[i][0], c4[i]));
  }
  {
    bf16x8 vf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) vf[t][kk] = frag_rows(tV, t, kk, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) dp[i][t] = mma16(vf[t][1], w.d[i][1], mma16(vf[t][0], w.d[i][0], nd4[i]));
  }
  B5TFrag kt;
  __builtin_amdgcn_sched_barrier(0);
  b5_tfrag(kt, tK, lane);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[i][t][r] = __builtin_amdgcn_exp2f(s[i][t][r] * LOG2E);
  if constexpr (TAIL) {                                    // (its own instantiation: as a run-time flag hipcc if-converts the selects
    const int klim = p.R - t0 * 16 - 4 * g;                //  into EVERY step.)  The lane's key rows are t0*16 + 16t + 4g + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[i][t][r] = (16 * t + r < klim) ? s[i][t][r] : 0.f;
  }
  if (corner && t0 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[0][0][r] = (i16 < p.M && 4 * g + r < p.M) ? 0.f : s[0][0][r];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) dp[i][t] *= s[i][t];        // dS^T = P^T * (dP^T - delta)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bf16x8 sf = pack_p(dp[i][0], dp[i][1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[i][dt] = mma16(kt.f[dt], sf, dq[i][dt]);
  }
}

// phase B step: two sixteen-query sub-tiles at LDS row t0 * 16 against the wave's two key tiles (S orientation: the lane owns key
// column i16 of each tile; rows 4g + r are queries, whose constants come from LDS as ready-made C operands).  Query rows >= R have
// zero-filled Q / dO rows and zero constants (P = 1 times zeros); key lanes >= R only produce their own, discarded, columns.
__device__ __forceinline__ void b5_dkv_step(f32x4 (&dk)[2][4], f32x4 (&dv)[2][4], const AP& p, const char* tQ, const float* tC,
                                            const bf16x8 (&kf)[2][2], const bf16x8 (&vf)[2][2], int t0, bool corner, int lane) {
  // tQ: the Q image advanced to the step's first row (dO is B5_IMG further); tC: the row constants -(m + log l) advanced likewise
  // (-delta is B5_ROWS floats further) -- every read of the step is one of a few per-lane bases plus an immediate
  const int g = lane >> 4, i16 = lane & 15;
  const char* tD = tQ + B5_IMG;
  f32x4 s[2][2], dp[2][2];
  {
    bf16x8 qr[2][2];
    f32x4 c4[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      c4[t] = *reinterpret_cast<const f32x4*>(tC + t * 16 + 4 * g);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) qr[t][kk] = frag_rows(tQ, t, kk, lane);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) s[i][t] = mma16(qr[t][1], kf[i][1], mma16(qr[t][0], kf[i][0], c4[t]));
  }
  {
    bf16x8 dr[2][2];
    f32x4 n4[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      n4[t] = *reinterpret_cast<const f32x4*>(tC + B5_ROWS + t * 16 + 4 * g);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) dr[t][kk] = frag_rows(tD, t, kk, lane);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) dp[i][t] = mma16(dr[t][1], vf[i][1], mma16(dr[t][0], vf[i][0], n4[t]));
  }
  B5TFrag qt, dot;
  __builtin_amdgcn_sched_barrier(0);
  b5_tfrag(dot, tD, lane);
  b5_tfrag(qt, tQ, lane);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[i][t][r] = __builtin_amdgcn_exp2f(s[i][t][r] * LOG2E);
  if (corner && t0 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[0][0][r] = (i16 < p.M && 4 * g + r < p.M) ? 0.f : s[0][0][r];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) dp[i][t] *= s[i][t];        // dS = P * (dP - delta)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bf16x8 pf = pack_p(s[i][0], s[i][1]), sf = pack_p(dp[i][0], dp[i][1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dv[i][dt] = mma16(dot.f[dt], pf, dv[i][dt]);
      dk[i][dt] = mma16(qt.f[dt], sf, dk[i][dt]);
    }
  }
}

// The wave's two finished 16 x 64 tiles (C layout: the lane owns row i16 of each tile, d = 16 dt + 4g + r) leave through the wave's
// LDS scratch: written as bf16 rows, read back as 16-byte chunks, stored as FULL 128-byte lines -- one store instruction covers 8
// rows (8 lines) where the row-per-lane form touched 16 lines with 8 bytes per lane and the CU's 8 waves queued behind each other's
// 16 store instructions (8-11k cycles per problem, tools/attn_bwd_trace.py).  Rows < M (proxy rows: fp32 partials, written by the
// caller) and rows >= R are skipped.  cs[e]: the column sums of the rows stored (values as stored), columns 8 (lane & 7) + e,
// summed over the lane's rows; b5_colsum_wave finishes them over the 8 row lanes.
__device__ __forceinline__ void b5_store_tiles(const f32x4 (&acc)[2][4], float scale, char* stg, bf16_t* colbase, const AP& p, int n,
                                               int wave, int lane, float (&cs)[8]) {
  const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 v = acc[i][dt] * scale;
      const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(stg + (i * 16 + i16) * B5_STG_ROW + dt * 32 + g * 8) = o;
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int rr = (wave + B5W * (ps >> 1)) * 16 + (ps & 1) * 8 + (lane >> 3);      // problem row of staging row 8 ps + (lane >> 3)
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + (ps * 8 + (lane >> 3)) * B5_STG_ROW + (lane & 7) * 16);
    if (rr >= p.M && rr < p.R) {
      *reinterpret_cast<bf16x8*>(colbase + (int64_t)tok_of(p, n, rr) * p.ldqkv + (lane & 7) * 8) = v;
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] += (float)v[e];
    }
  }
}
// column sums of one wave: over the 8 row lanes (lane >> 3), then one LDS row per wave; the workgroup's fixed-order sum over the
// waves happens after the next barrier (b5_colsum_finish)
__device__ __forceinline__ void b5_colsum_wave(float (&cs)[8], float* red_row, int lane) {
#pragma unroll
  for (int o = 8; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] += __shfl_xor(cs[e], o, 64);
  if (lane < 8) {
    store4(red_row + lane * 8, f32x4{cs[0], cs[1], cs[2], cs[3]});
    store4(red_row + lane * 8 + 4, f32x4{cs[4], cs[5], cs[6], cs[7]});
  }
}
__device__ __forceinline__ void b5_colsum_finish(const float* red, float* dst, int d) {
  float a = 0.f;
#pragma unroll
  for (int w = 0; w < B5W; ++w) a += red[w * DH + d];
  dst[d] = a;
}

// the three bias column-sum rows of one problem (block 0 of its p.nq rows; the other blocks of the two-kernel layout are zero)
__device__ __forceinline__ float* b5_cs_row(const AP& p, const Prob& pr) {
  return p.cs + ((int64_t)(pr.b * p.N + pr.n) * p.nq) * (3 * p.H * DH) + pr.h * DH;
}

__global__ __launch_bounds__(B5THR, 2) void attn_bwd5_kernel(AP p, int* counter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* X = smem; char* Y = smem + 2 * B5_IMG;
  float* sC = reinterpret_cast<float*>(smem + B5_OFF_STATS); float* sNd = sC + B5_ROWS;
  float* red = reinterpret_cast<float*>(smem + B5_OFF_RED);
  int* sNext = reinterpret_cast<int*>(smem + B5_OFF_NEXT);
  char* stg = smem + B5_OFF_STG + (threadIdx.x >> 6) * B5_STG_WAVE;
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int nt = (p.R + 15) / 16;                          // sixteen-row tiles of a problem (<= 13)
  const int nsteps = (nt + 1) / 2;                         // 32-row steps; an odd last tile is paired with the pad tile
  const bool active = wave < nt;                           // owns at least one real tile (tiles wave, wave + 8)
  int prob = blockIdx.x, prev = -1;
  if (prob >= p.nprob) return;
  // the pad tile of the four images: never staged, zero for the whole launch
  if (tid0 < 4 * 16 * 8) *reinterpret_cast<u32x4*>(smem + (tid0 >> 7) * B5_IMG + FG * 128 + (tid0 & 127) * 16) = u32x4{0, 0, 0, 0};
  B5Own w;
  {
    const Prob pr(p, prob);
    b5_stage_kv(X, p, pr, lane0, wave);
    b5_load_own(w, p, pr, wave, lane0);
  }
  unsigned long long* tr = (p.tr && (int)blockIdx.x == (int)gridDim.x / 2 && (wave == 0 || wave == 7) && lane0 == 0)
                               ? p.tr + (wave ? 64 : 0) : nullptr;
  int it = 0;
#define B5_STAMP(k) do { if (tr && it < 8) tr[it * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  for (;;) {
    B5_STAMP(0);
    // Every per-lane constant (LDS fragment addresses, loader offsets, row predicates) is re-derived per phase from a lane id the
    // compiler cannot see through: hoisted out of the persistent loop they do not fit 256 registers, and a spill RELOAD is a vector
    // memory operation -- its s_waitcnt vmcnt(0) drains the LDS-DMA queue, i.e. exposes the whole prefetch (measured: 208 us).
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    int tid = wave * 64 + lane, i16 = lane & 15, g = lane >> 4;
    if (tid == 0) *sNext = counter ? atomicAdd(counter, 1) + (int)gridDim.x : prob + (int)gridDim.x;
    __syncthreads();                                       // X (K, V of prob) has landed; every wave is done with Y and `red`
    B5_STAMP(1);
    const Prob pr(p, prob);
    if (p.cs && prev >= 0 && tid < 2 * DH) {               // previous problem: dK / dV column sums
      const Prob pv(p, prev);
      float* row = b5_cs_row(p, pv);
      if (tid < DH) b5_colsum_finish(red + B5W * DH, row + p.H * DH, tid);
      else          b5_colsum_finish(red + 2 * B5W * DH, row + 2 * p.H * DH, tid - DH);
    }
    const bool corner = wave == 0 && pr.n != 0;
    bf16x8 kf[2][2], vf[2][2];
    {
      // ---- phase A: dQ of the wave's query tiles; delta = sum_d dO * O from the fragments in registers (each of the 4 lanes of a
      // row holds 16 of the 64 d); the row constants also go to LDS for phase B (rows >= R: zeros)
      f32x4 c4[2], nd4[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int e = 0; e < 8; ++e) dl += (float)w.d[i][kk][e] * (float)w.o[i][kk][e];
        dl = group_sum(dl);
        const float c = -(w.m[i] + w.lg[i]);
        c4[i] = f32x4{c, c, c, c}; nd4[i] = f32x4{-dl, -dl, -dl, -dl};
        const int rq = (wave + B5W * i) * 16 + i16;
        if (g == 0 && rq < B5_ROWS) { sC[rq] = c; sNd[rq] = -dl; }
      }
      f32x4 dq[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[i][dt] = f32x4{0, 0, 0, 0};
      B5_STAMP(2);
      // Y = {Q, dO} of this problem lands during phase A: one piece pair per step (every wave issues its pieces, active or not).
      // The one step that reaches past key R is peeled (its selects in its own instantiation, outside the loop).
      const int nfull = p.R / 32 < nsteps ? p.R / 32 : nsteps;
      int st = 0;
      for (; st < nfull; ++st) {
        if (st < B5_PIECES) b5_stage_qdo(Y, p, pr, lane, wave, st, st + 1);
        if (active) b5_dq_step<false>(dq, p, X + st * (32 * 128), w, c4, nd4, 2 * st, corner, lane);
      }
      if (st < nsteps) {
        if (st < B5_PIECES) b5_stage_qdo(Y, p, pr, lane, wave, st, st + 1);
        if (active) b5_dq_step<true>(dq, p, X + st * (32 * 128), w, c4, nd4, 2 * st, corner, lane);
        ++st;
      }
      if (st < B5_PIECES) b5_stage_qdo(Y, p, pr, lane, wave, st, B5_PIECES);
      B5_STAMP(3);
      if (wave == 0 && i16 < p.M && i16 < p.R) {              // proxy query rows: per-frame partials, reduced later
        float* part = p.ws1 + ((int64_t)prob * p.M + i16) * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(part + dt * 16 + 4 * g, dq[0][dt]);
      }
      float cs[8];
      b5_store_tiles(dq, p.q_scale, stg, p.dqkv + (int64_t)pr.b * p.S * p.ldqkv + pr.h * DH, p, pr.n, wave, lane, cs);
      if (p.cs) b5_colsum_wave(cs, red + wave * DH, lane);
      // the wave's own key rows for phase B, out of the image (X is overwritten after the barrier)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int t = wave + B5W * i < B5_ROWS / 16 ? wave + B5W * i : B5_ROWS / 16 - 1;      // (tiles past the image: the pad tile)
          kf[i][kk] = frag_rows(X, t, kk, lane);
          vf[i][kk] = frag_rows(X + B5_IMG, t, kk, lane);
        }
    }
    B5_STAMP(4);
    __syncthreads();                                       // Y has landed; constants and dQ column sums are in LDS; X is free
    B5_STAMP(5);
    lane = lane0;
    asm volatile("" : "+v"(lane));
    tid = wave * 64 + lane; i16 = lane & 15; g = lane >> 4;
    const int next = *sNext;
    if (p.cs && tid < DH) {
      float* row = b5_cs_row(p, pr);
      b5_colsum_finish(red, row, tid);
      for (int blk = 1; blk < p.nq; ++blk)                 // rows of the two-kernel layout this launch does not use
        for (int j = 0; j < 3; ++j) row[(int64_t)blk * 3 * p.H * DH + j * p.H * DH + tid] = 0.f;
    }
    {
      // ---- phase B: dK, dV of the wave's key tiles
      f32x4 dk[2][4], dv[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[i][dt] = f32x4{0, 0, 0, 0}; dv[i][dt] = f32x4{0, 0, 0, 0}; }
      // X = {K, V} of the NEXT problem lands during phase B, one piece pair per step.  The step's LDS offsets go through an opaque
      // scalar: the Y images and the row constants live above the 64 KiB an LDS instruction's immediate reaches, and hipcc otherwise
      // re-associates every read into (lane part + step) + image constant -- one v_add per read, 25 per step.
      for (int st = 0; st < nsteps || st < B5_PIECES; ++st) {
        if (st < B5_PIECES && next < p.nprob) { const Prob pn(p, next); b5_stage_kv(X, p, pn, lane, wave, st, st + 1); }
        if (active && st < nsteps) {
          unsigned yo = 2u * B5_IMG + (unsigned)st * (32 * 128), co = (unsigned)B5_OFF_STATS + (unsigned)st * (32 * 4);
          asm volatile("" : "+s"(yo), "+s"(co));
          b5_dkv_step(dk, dv, p, smem + yo, reinterpret_cast<const float*>(smem + co), kf, vf, 2 * st, corner, lane);
        }
      }
      B5_STAMP(6);
      // the next problem's own query rows: issued here, behind the step loop (48 registers the loop does not have), in front of the
      // stores and the column sums that cover most of their latency
      if (next < p.nprob) { const Prob pn(p, next); b5_load_own(w, p, pn, wave, lane); }
      if (wave == 0 && i16 < p.M && i16 < p.R) {              // proxy keys: per-frame partials, reduced later
        float* part = p.ws2 + ((int64_t)prob * p.M + i16) * (2 * DH);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { store4(part + dt * 16 + 4 * g, dk[0][dt]); store4(part + DH + dt * 16 + 4 * g, dv[0][dt]); }
      }
      bf16_t* colbase = p.dqkv + (int64_t)pr.b * p.S * p.ldqkv + pr.h * DH;
      float cs[8];
      b5_store_tiles(dk, 1.0f, stg, colbase + (int64_t)p.H * DH, p, pr.n, wave, lane, cs);
      if (p.cs) b5_colsum_wave(cs, red + (B5W + wave) * DH, lane);
      b5_store_tiles(dv, 1.0f, stg, colbase + (int64_t)2 * p.H * DH, p, pr.n, wave, lane, cs);
      if (p.cs) b5_colsum_wave(cs, red + (2 * B5W + wave) * DH, lane);
    }
    B5_STAMP(7);
    ++it;
    prev = prob;
    if (next >= p.nprob) break;
    prob = next;
  }
#undef B5_STAMP
  if (p.cs) {
    __syncthreads();
    const int tid = tid0;
    if (tid < 2 * DH) {
      const Prob pv(p, prev);
      float* row = b5_cs_row(p, pv);
      if (tid < DH) b5_colsum_finish(red + B5W * DH, row + p.H * DH, tid);
      else          b5_colsum_finish(red + 2 * B5W * DH, row + 2 * p.H * DH, tid - DH);
    }
  }
}

int check_common(const char* name, int32_t mode, int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L,
                 int64_t ldqkv, int64_t ldo, int32_t dtype) {
  XP_REQUIRE(dtype == XP_BF16 || dtype == XP_F32, "%s: bad dtype %d", name, dtype);
  XP_REQUIRE(mode == XP_ATTN_PROXY || mode == XP_ATTN_CAUSAL, "%s: bad mode %d", name, mode);
  XP_REQUIRE(B > 0 && H > 0 && S > 0, "%s: empty problem", name);
  if (mode == XP_ATTN_PROXY) XP_REQUIRE(M >= 1 && N >= 1 && L >= 1 && S == M + N * L, "%s: S=%lld != M+N*L (%lld,%lld,%lld)",
                                        name, (long long)S, (long long)M, (long long)N, (long long)L);
  XP_REQUIRE(ldqkv >= 3 * H * DH && ldqkv % 8 == 0 && ldo >= H * DH && ldo % 8 == 0, "%s: bad leading dimensions", name);
  XP_REQUIRE(B * H * (mode == XP_ATTN_PROXY ? N : 1) <= 65535 * 32, "%s: too many problems", name);
  return XP_OK;
}

}  // namespace

// fp32 compute mode (attention_f32.hip): exact-arithmetic kernels, same layout / statistics / workspace contract
int xp_attn_f32_fwd(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, float* stats, const int64_t* pad, int32_t mode,
                    int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L, hipStream_t st);
int xp_attn_f32_bwd(const void* qkv, int64_t ldqkv, const void* out, const void* dout, int64_t ldo, const float* stats,
                    const int64_t* pad, void* dqkv, float q_scale, int32_t mode, int64_t B, int64_t H, int64_t S,
                    int64_t M, int64_t N, int64_t L, float* delta_ws, hipStream_t st);

static void* g_attn_trace = nullptr;
extern "C" int xp_debug_set_attn_trace(void* device_buffer) { g_attn_trace = device_buffer; return XP_OK; }

extern "C" size_t xp_attn_workspace_bytes(int32_t mode, int64_t B, int64_t H, int64_t M, int64_t N, int64_t L) {
  // S is M + N*L for PROXY; for CAUSAL callers pass M=0, N=1, L=S
  const int64_t S = M + N * L;
  const int64_t delta = B * H * S;
  if (mode != XP_ATTN_PROXY) return (size_t)delta * sizeof(float);
  const int64_t P = B * H * N;
  const int64_t fwd = P * M * PART, bwd = delta + P * M * DH + P * M * 2 * DH + 64;      // (+ the fused backward's problem counter)
  return (size_t)(fwd > bwd ? fwd : bwd) * sizeof(float);
}

extern "C" int xp_attn_fwd(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, float* stats,
                           const int64_t* pad_mask, int32_t mode, int64_t B, int64_t H, int64_t S,
                           int64_t M, int64_t N, int64_t L, int32_t dtype,
                           void* workspace, size_t workspace_bytes, void* stream) {
  XP_REQUIRE(qkv && out && stats, "xp_attn_fwd: null pointer");
  if (mode == XP_ATTN_CAUSAL) { M = 0; N = 1; L = S; }
  int rc = check_common("xp_attn_fwd", mode, B, H, S, M, N, L, ldqkv, ldo, dtype);
  if (rc) return rc;
  if (dtype == XP_F32) return xp_attn_f32_fwd(qkv, ldqkv, out, ldo, stats, pad_mask, mode, B, H, S, M, N, L, (hipStream_t)stream);
  XP_REQUIRE(mode == XP_ATTN_CAUSAL || (workspace && workspace_bytes >= xp_attn_workspace_bytes(mode, B, H, M, N, L)),
             "xp_attn_fwd: workspace too small");
  AP p{};
  p.qkv = (const bf16_t*)qkv; p.ldqkv = ldqkv; p.out = (bf16_t*)out; p.ldo = ldo; p.stats = stats; p.pad = pad_mask;
  p.mode = mode; p.B = (int)B; p.H = (int)H; p.S = (int)S; p.M = (int)M; p.N = (int)N; p.L = (int)L;
  p.R = mode == XP_ATTN_PROXY ? (int)(M + L) : (int)S;
  p.ws0 = (float*)workspace;
  p.ws2 = (float*)g_attn_trace;
  hipStream_t st = (hipStream_t)stream;
  p.nq = (int)cdiv(p.R, FQ); p.nprob = (int)(B * H * N);
  // the persistent kernel: proxy problems that fit one LDS group, no padding mask, at most 16 proxy rows (its proxy x proxy mask
  // lives in query tile 0 / key sub-tile 0 only), and a device that grants the 104 KiB dynamic-LDS opt-in (configured per device)
  bool use3 = mode == XP_ATTN_PROXY && p.R <= FG && p.M <= 16 && !pad_mask;
  // the multi-group persistent kernel: proxy problems wider than one LDS group (448^2 frames), same conditions otherwise
  bool use4 = mode == XP_ATTN_PROXY && p.R > FG && p.M <= 16 && !pad_mask;
  int ncu = 256;
  if (use3 || use4) {
    static std::mutex mu;
    static int configured[64] = {0};                  // per device -- 0: not yet, > 0: CU count, -1: refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) use3 = use4 = false;
    else {
      std::lock_guard<std::mutex> lock(mu);
      if (!configured[dev]) {
        int n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            F3_LDS) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            F3_LDS) == hipSuccess;
        configured[dev] = ok ? (n > 0 ? n : 256) : -1;
      }
      if (configured[dev] < 0) use3 = use4 = false; else ncu = configured[dev];
      if (use4 && (ncu % 8 != 0 || ncu / 8 < (int)cdiv(cdiv(p.R, 16), 2 * F3W))) use4 = false;      // needs whole quads per XCD
    }
  }
  if (use3) {
    attn_fwd3_kernel<<<(unsigned)(p.nprob < ncu ? p.nprob : ncu), F3THR, F3_LDS, st>>>(p);
  } else if (use4) {
    attn_fwd4_kernel<<<(unsigned)ncu, F3THR, F3_LDS, st>>>(p);      // persistent: one workgroup per CU (ncu is a multiple of 8 here)
  } else {
    attn_fwd_kernel<<<(unsigned)(cdiv(p.nprob, 8) * 8 * p.nq), FTHR, 0, st>>>(p);
  }
  XP_CHECK_LAUNCH("xp_attn_fwd");
  if (mode == XP_ATTN_PROXY) {
    attn_fwd_merge_kernel<<<(unsigned)(B * H * M), 256, 0, st>>>(p);
    XP_CHECK_LAUNCH("xp_attn_fwd(merge)");
  }
  return XP_OK;
}

extern "C" int64_t xp_attn_bwd_colsum_rows(int32_t mode, int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L, int32_t dtype) {
  if (dtype != XP_BF16) return 0;                    // the fp32 kernels do not produce them: callers run xp_colsum_partials
  if (mode == XP_ATTN_CAUSAL) { M = 0; N = 1; L = S; }
  else if (mode != XP_ATTN_PROXY) return 0;
  return B * N * cdiv(M + L, FQ) + B * M;
}

extern "C" int xp_attn_bwd(const void* qkv, int64_t ldqkv, const void* out, const void* dout, int64_t ldo,
                           const float* stats, const int64_t* pad_mask, void* dqkv, float q_scale,
                           int32_t mode, int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L, int32_t dtype,
                           void* workspace, size_t workspace_bytes, void* stream) {
  return xp_attn_bwd2(qkv, ldqkv, out, dout, ldo, stats, pad_mask, dqkv, q_scale, mode, B, H, S, M, N, L, dtype, workspace,
                      workspace_bytes, nullptr, stream);
}

extern "C" int xp_attn_bwd2(const void* qkv, int64_t ldqkv, const void* out, const void* dout, int64_t ldo,
                            const float* stats, const int64_t* pad_mask, void* dqkv, float q_scale,
                            int32_t mode, int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L, int32_t dtype,
                            void* workspace, size_t workspace_bytes, float* dqkv_colsum_partials, void* stream) {
  XP_REQUIRE(qkv && out && dout && stats && dqkv, "xp_attn_bwd: null pointer");
  XP_REQUIRE(!dqkv_colsum_partials || xp_attn_bwd_colsum_rows(mode, B, H, S, M, N, L, dtype) > 0,
             "xp_attn_bwd2: fused column sums are not available for this problem (xp_attn_bwd_colsum_rows() == 0)");
  if (mode == XP_ATTN_CAUSAL) { M = 0; N = 1; L = S; }
  int rc = check_common("xp_attn_bwd", mode, B, H, S, M, N, L, ldqkv, ldo, dtype);
  if (rc) return rc;
  XP_REQUIRE(workspace && workspace_bytes >= xp_attn_workspace_bytes(mode, B, H, M, N, L), "xp_attn_bwd: workspace too small");
  if (dtype == XP_F32)
    return xp_attn_f32_bwd(qkv, ldqkv, out, dout, ldo, stats, pad_mask, dqkv, q_scale, mode, B, H, S, M, N, L, (float*)workspace,
                           (hipStream_t)stream);
  AP p{};
  p.qkv = (const bf16_t*)qkv; p.ldqkv = ldqkv; p.out = (bf16_t*)out; p.dout = (const bf16_t*)dout; p.ldo = ldo;
  p.dqkv = (bf16_t*)dqkv; p.stats = const_cast<float*>(stats); p.pad = pad_mask; p.q_scale = q_scale;
  p.mode = mode; p.B = (int)B; p.H = (int)H; p.S = (int)S; p.M = (int)M; p.N = (int)N; p.L = (int)L;
  p.R = mode == XP_ATTN_PROXY ? (int)(M + L) : (int)S;
  const int64_t P = B * H * N;
  p.ws0 = (float*)workspace; p.ws1 = p.ws0 + B * H * S; p.ws2 = p.ws1 + P * M * DH;
  hipStream_t st = (hipStream_t)stream;
  p.nq = (int)cdiv(p.R, FQ); p.nprob = (int)P;
  p.cs = dqkv_colsum_partials; p.cs_main = (int)(B * N * p.nq);
  p.tr = reinterpret_cast<unsigned long long*>(g_attn_trace);
  // the fused kernel: the problems of attn_fwd3_kernel (one LDS group, at most 16 proxy rows, no padding mask) on a device that
  // grants the dynamic-LDS opt-in; XPRETRAIN_DEBUG=attn_bwd_split keeps the two-kernel path (A/B and the cross-check test)
  bool use5 = mode == XP_ATTN_PROXY && p.R <= FG && p.M <= 16 && !pad_mask && !xp_debug_flag("attn_bwd_split");
  int ncu = 256;
  if (use5) {
    static std::mutex mu;
    static int configured[64] = {0};                    // per device -- 0: not yet, > 0: CU count, -1: refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) use5 = false;
    else {
      std::lock_guard<std::mutex> lock(mu);
      if (!configured[dev]) {
        int n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            B5_LDS) == hipSuccess;
        configured[dev] = ok ? (n > 0 ? n : 256) : -1;
      }
      if (configured[dev] < 0) use5 = false; else ncu = configured[dev];
    }
  }
  if (use5) {
    int* counter = reinterpret_cast<int*>(p.ws2 + P * M * 2 * DH);
    if (xp_debug_flag("attn_bwd_static")) counter = nullptr;          // (A/B: problems b, b + grid, ... per workgroup, no counter)
    else if (hipMemsetAsync(counter, 0, sizeof(int), st) != hipSuccess) { xp_set_error("xp_attn_bwd: counter reset failed"); return XP_ERR_LAUNCH; }
    attn_bwd5_kernel<<<(unsigned)(p.nprob < ncu ? p.nprob : ncu), B5THR, B5_LDS, st>>>(p, counter);
    XP_CHECK_LAUNCH("xp_attn_bwd(fused)");
  } else {
    const unsigned grid = (unsigned)(cdiv(p.nprob, 8) * 8 * p.nq);
    attn_bwd_dq_kernel<<<grid, FTHR, 0, st>>>(p);          // also computes delta = rowsum(dO * O) into ws0
    XP_CHECK_LAUNCH("xp_attn_bwd(dq)");
    attn_bwd_dkv_kernel<<<grid, FTHR, 0, st>>>(p);
    XP_CHECK_LAUNCH("xp_attn_bwd(dkv)");
  }
  if (mode == XP_ATTN_PROXY) {
    attn_bwd_proxy_reduce_kernel<<<(unsigned)(B * H * M), 256, 0, st>>>(p);
    XP_CHECK_LAUNCH("xp_attn_bwd(proxy reduce)");
  }
  return XP_OK;
}
