// fp32 attention for the fp32 compute mode (north_star's "1e-3 fp32" leg; the reference computes fp32 unless fp16 is set,
// pretrain/run_pretrain.py:234-236).  Same problem family, data layout, statistics and workspace contract as attention.hip
// (PROXY: CLIPAttention.forward2, modeling/CLIP_ViP.py:332-381; CAUSAL: CLIPAttention.forward :266-330), but written for
// exactness, not speed: plain fp32 VALU arithmetic, one wave per OWN row, lane j owns one row of the other side per
// 64-row chunk, online softmax in fp32, fixed-order sums (no atomics).  The bf16 path never calls this file.
//
//   fwd    wave = one query row: scores of 64 keys at a time (lane j: dot(q, K[k_j]) over the 64 head dims), online
//          softmax, O[d] (lane d) += sum_j p_j V[k_j][d] with p_j broadcast by readlane order.
//   bwd-q  wave = one query row: delta = dO.O, p_j recomputed from the saved (max, log-sum), dS_j = p_j (dO.V_j - delta),
//          dQ[d] += sum_j dS_j K[k_j][d]; publishes delta in ws0.
//   bwd-kv wave = one key row: loops over the queries that see it (frame keys: the M proxies + the L tokens of its frame;
//          proxy keys: all S queries; causal: queries >= key), dV[d] += sum_j p_j dO_j[d], dK[d] += sum_j dS_j Q_j[d].
#include "common.h"
#include <math.h>

namespace {

constexpr int DH = 64;
constexpr float F32_MIN = -3.4028234663852886e38f;   // torch.finfo(float32).min, _expand_mask (:50-61)

struct AF {
  const float* qkv; int64_t ldqkv;
  float* out; const float* dout; int64_t ldo;
  float* dqkv; float* stats; const int64_t* pad;
  int mode, B, H, S, M, N, L;
  float q_scale; float* delta;
};

// the "other side" rows an own row interacts with: up to two token ranges [a0, a1) and [b0, b1)
struct Ranges { int a0, a1, b0, b1; };
__device__ __forceinline__ Ranges keys_of_query(const AF& p, int s) {
  if (p.mode == XP_ATTN_CAUSAL) return Ranges{0, s + 1, 0, 0};
  if (s < p.M) return Ranges{0, p.S, 0, 0};                                   // proxy query: every key
  const int n = (s - p.M) / p.L;
  return Ranges{0, p.M, p.M + n * p.L, p.M + (n + 1) * p.L};                  // frame query: proxies + own frame
}
__device__ __forceinline__ Ranges queries_of_key(const AF& p, int s) {
  if (p.mode == XP_ATTN_CAUSAL) return Ranges{s, p.S, 0, 0};
  if (s < p.M) return Ranges{0, p.S, 0, 0};                                   // proxy key: seen by every query
  const int n = (s - p.M) / p.L;
  return Ranges{0, p.M, p.M + n * p.L, p.M + (n + 1) * p.L};                  // frame key: proxies + own frame
}
__device__ __forceinline__ int range_len(const Ranges& r) { return (r.a1 - r.a0) + (r.b1 - r.b0); }
__device__ __forceinline__ int range_at(const Ranges& r, int i) { const int na = r.a1 - r.a0; return i < na ? r.a0 + i : r.b0 + (i - na); }

__device__ __forceinline__ float dot64(const float* a_regs, const float* row) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    const f32x4 v = load4(row + d);
    s = fmaf(a_regs[d], v[0], s); s = fmaf(a_regs[d + 1], v[1], s); s = fmaf(a_regs[d + 2], v[2], s); s = fmaf(a_regs[d + 3], v[3], s);
  }
  return s;
}
__device__ __forceinline__ void load_row(float* regs, const float* row) {
#pragma unroll
  for (int d = 0; d < DH; d += 4) { const f32x4 v = load4(row + d); regs[d] = v[0]; regs[d + 1] = v[1]; regs[d + 2] = v[2]; regs[d + 3] = v[3]; }
}
// the reference's masking of one (query, key) score: padded keys get finfo.min (added to the score; finfo.min absorbs it)
__device__ __forceinline__ float mask_score(const AF& p, int b, int key, float s) {
  return (p.pad && p.pad[(int64_t)b * p.S + key] == 0) ? F32_MIN : s;
}

__global__ __launch_bounds__(256) XP_NO_PK_F32 void attn_f32_fwd_kernel(AF p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)p.B * p.H * p.S) return;
  const int s = (int)(row % p.S), h = (int)((row / p.S) % p.H), b = (int)(row / ((int64_t)p.S * p.H));
  const float* base = p.qkv + (int64_t)b * p.S * p.ldqkv + h * DH;
  float q[DH];
  load_row(q, base + (int64_t)s * p.ldqkv);
  const Ranges r = keys_of_query(p, s);
  const int n = range_len(r);
  float m = -INFINITY, l = 0.f, o = 0.f;                        // o: output dim `lane`
  for (int c = 0; c < n; c += 64) {
    const int i = c + lane;
    const bool ok = i < n;
    const int key = ok ? range_at(r, i) : 0;
    float sc = ok ? mask_score(p, b, key, dot64(q, base + (int64_t)key * p.ldqkv + p.H * DH)) : -INFINITY;
    const float mnew = fmaxf(m, wave_max(sc));
    const float alpha = __expf(m - mnew);                        // m = -inf at the first chunk -> 0 (mnew is finite: >= 1 key)
    const float pj = ok ? __expf(sc - mnew) : 0.f;
    l = l * alpha + wave_sum(pj);
    o *= alpha;
    const int cnt = n - c < 64 ? n - c : 64;
    for (int j = 0; j < cnt; ++j) {
      const float pv = __shfl(pj, j, 64);
      o = fmaf(pv, base[(int64_t)range_at(r, c + j) * p.ldqkv + 2 * p.H * DH + lane], o);
    }
    m = mnew;
  }
  p.out[((int64_t)b * p.S + s) * p.ldo + h * DH + lane] = o / l;
  if (lane == 0) { float* st = p.stats + (((int64_t)b * p.H + h) * p.S + s) * 2; st[0] = m; st[1] = __logf(l); }
}

__global__ __launch_bounds__(256) XP_NO_PK_F32 void attn_f32_bwd_q_kernel(AF p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)p.B * p.H * p.S) return;
  const int s = (int)(row % p.S), h = (int)((row / p.S) % p.H), b = (int)(row / ((int64_t)p.S * p.H));
  const float* base = p.qkv + (int64_t)b * p.S * p.ldqkv + h * DH;
  const int64_t orow = ((int64_t)b * p.S + s) * p.ldo + h * DH;
  float q[DH], go[DH];
  load_row(q, base + (int64_t)s * p.ldqkv);
  load_row(go, p.dout + orow);
  const float delta = wave_sum(p.dout[orow + lane] * p.out[orow + lane]);
  const int64_t si = ((int64_t)b * p.H + h) * p.S + s;
  const float m = p.stats[si * 2], lg = p.stats[si * 2 + 1];
  if (lane == 0) p.delta[si] = delta;
  const Ranges r = keys_of_query(p, s);
  const int n = range_len(r);
  float dq = 0.f;
  for (int c = 0; c < n; c += 64) {
    const int i = c + lane;
    const bool ok = i < n;
    const int key = ok ? range_at(r, i) : 0;
    float ds = 0.f;
    if (ok) {
      const float sc = mask_score(p, b, key, dot64(q, base + (int64_t)key * p.ldqkv + p.H * DH));
      const float pj = __expf((sc - m) - lg);
      ds = pj * (dot64(go, base + (int64_t)key * p.ldqkv + 2 * p.H * DH) - delta);
    }
    const int cnt = n - c < 64 ? n - c : 64;
    for (int j = 0; j < cnt; ++j)
      dq = fmaf(__shfl(ds, j, 64), base[(int64_t)range_at(r, c + j) * p.ldqkv + p.H * DH + lane], dq);
  }
  p.dqkv[((int64_t)b * p.S + s) * p.ldqkv + h * DH + lane] = dq * p.q_scale;
}

__global__ __launch_bounds__(256) XP_NO_PK_F32 void attn_f32_bwd_kv_kernel(AF p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)p.B * p.H * p.S) return;
  const int s = (int)(row % p.S), h = (int)((row / p.S) % p.H), b = (int)(row / ((int64_t)p.S * p.H));
  const float* base = p.qkv + (int64_t)b * p.S * p.ldqkv + h * DH;
  const float* dobase = p.dout + (int64_t)b * p.S * p.ldo + h * DH;
  float k[DH], v[DH];
  load_row(k, base + (int64_t)s * p.ldqkv + p.H * DH);
  load_row(v, base + (int64_t)s * p.ldqkv + 2 * p.H * DH);
  const bool kpad = p.pad && p.pad[(int64_t)b * p.S + s] == 0;
  const Ranges r = queries_of_key(p, s);
  const int n = range_len(r);
  float dk = 0.f, dv = 0.f;
  for (int c = 0; c < n; c += 64) {
    const int i = c + lane;
    const bool ok = i < n;
    const int qi = ok ? range_at(r, i) : 0;
    float pj = 0.f, ds = 0.f;
    if (ok) {
      const int64_t si = ((int64_t)b * p.H + h) * p.S + qi;
      float sc = dot64(k, base + (int64_t)qi * p.ldqkv);
      if (kpad) sc = F32_MIN;
      pj = __expf((sc - p.stats[si * 2]) - p.stats[si * 2 + 1]);
      ds = pj * (dot64(v, dobase + (int64_t)qi * p.ldo) - p.delta[si]);
    }
    const int cnt = n - c < 64 ? n - c : 64;
    for (int j = 0; j < cnt; ++j) {
      const int qj = range_at(r, c + j);
      dv = fmaf(__shfl(pj, j, 64), dobase[(int64_t)qj * p.ldo + lane], dv);
      dk = fmaf(__shfl(ds, j, 64), base[(int64_t)qj * p.ldqkv + lane], dk);
    }
  }
  float* o = p.dqkv + ((int64_t)b * p.S + s) * p.ldqkv + h * DH + lane;
  o[(int64_t)p.H * DH] = dk;
  o[(int64_t)2 * p.H * DH] = dv;
}

}  // namespace

// called by xp_attn_fwd / xp_attn_bwd (attention.hip) for dtype == XP_F32; arguments already validated there
int xp_attn_f32_fwd(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, float* stats, const int64_t* pad, int32_t mode,
                    int64_t B, int64_t H, int64_t S, int64_t M, int64_t N, int64_t L, hipStream_t st) {
  AF p{};
  p.qkv = (const float*)qkv; p.ldqkv = ldqkv; p.out = (float*)out; p.ldo = ldo; p.stats = stats; p.pad = pad;
  p.mode = mode; p.B = (int)B; p.H = (int)H; p.S = (int)S; p.M = (int)M; p.N = (int)N; p.L = (int)L;
  attn_f32_fwd_kernel<<<(unsigned)cdiv(B * H * S, 4), 256, 0, st>>>(p);
  XP_CHECK_LAUNCH("xp_attn_fwd(f32)");
  return XP_OK;
}

int xp_attn_f32_bwd(const void* qkv, int64_t ldqkv, const void* out, const void* dout, int64_t ldo, const float* stats,
                    const int64_t* pad, void* dqkv, float q_scale, int32_t mode, int64_t B, int64_t H, int64_t S,
                    int64_t M, int64_t N, int64_t L, float* delta_ws, hipStream_t st) {
  AF p{};
  p.qkv = (const float*)qkv; p.ldqkv = ldqkv; p.out = (float*)const_cast<void*>(out); p.dout = (const float*)dout; p.ldo = ldo;
  p.dqkv = (float*)dqkv; p.stats = const_cast<float*>(stats); p.pad = pad; p.q_scale = q_scale; p.delta = delta_ws;
  p.mode = mode; p.B = (int)B; p.H = (int)H; p.S = (int)S; p.M = (int)M; p.N = (int)N; p.L = (int)L;
  const unsigned grid = (unsigned)cdiv(B * H * S, 4);
  attn_f32_bwd_q_kernel<<<grid, 256, 0, st>>>(p);
  XP_CHECK_LAUNCH("xp_attn_bwd(f32 dq)");
  attn_f32_bwd_kv_kernel<<<grid, 256, 0, st>>>(p);
  XP_CHECK_LAUNCH("xp_attn_bwd(f32 dkv)");
  return XP_OK;
}
