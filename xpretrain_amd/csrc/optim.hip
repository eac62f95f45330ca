// On-device optimizer step: global gradient-norm clip + HF-style AdamW over ALL parameter tensors
// in one multi-tensor launch, fp32 master weights, optional compute-dtype shadow copies written in the same pass.
//
// Restates  src/optimization/adamw.py:40-103  (AdamW.step: m/v update, denom = sqrt(v) + eps, bias-corrected step size,
// decoupled weight decay AFTER the Adam update with the group's lr) and torch.nn.utils.clip_grad_norm_ as called by the
// training loops (run_video_retrieval.py:390-392): coef = min(1, max_norm / (||g||_2 + 1e-6)).
//
// HBM-bound: per element 16 B read (p, g, m, v) + 12 B written (p, m, v) + 2 B shadow; the norm pass reads g once more.
// Work decomposition: fixed 64 Ki-element CHUNKS of one tensor per workgroup (chunk map built once on the host side of
// the ABI); the gradient pointers, which change every step, travel in the kernel argument block, everything static
// lives in a device table.  All reductions are fixed-order (deterministic): per-chunk partial sums of squares, and
// every workgroup of the update kernel re-derives the same total from the partials array.
#include "common.h"

namespace {

constexpr int NT = 256;                      // threads per workgroup
constexpr int CHUNK = XP_OPT_CHUNK;          // elements per workgroup
constexpr int MAXT = XP_OPT_MAX_TENSORS, MAXG = XP_OPT_MAX_GROUPS;

struct StepArgs {
  const float* g[MAXT];
  unsigned char grp[MAXT];
  XpAdamGroup groups[MAXG];
};
struct NormArgs { const float* g[MAXT]; };

__device__ __forceinline__ float block_sum(float v) {
  __shared__ float red[NT / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) s += red[i];
  return s;
}

__global__ __launch_bounds__(NT) void sqnorm_partials_kernel(const XpAdamTensor* __restrict__ table,
                                                             const int32_t* __restrict__ chunk_map, NormArgs a,
                                                             float* __restrict__ partials) {
  const int t = chunk_map[2 * blockIdx.x], c = chunk_map[2 * blockIdx.x + 1];
  const int64_t n = table[t].numel, beg = (int64_t)c * CHUNK;
  const int64_t len = n - beg < CHUNK ? n - beg : CHUNK;
  const float* g = a.g[t] + beg;
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const int64_t n4 = len >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += NT) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < len; i += NT) s += g[i] * g[i];
  } else {
    for (int64_t i = threadIdx.x; i < len; i += NT) s += g[i] * g[i];
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// One element.  Every rounding point is written out (no compiler-chosen contraction): left to itself hipcc fused  m*b1 + (1-b1)*g  as
// fma(b1, m, [(1-b1)*g]) in this kernel and as fma(1-b1, g, [b1*m]) in a second update kernel built from the same source
// (tools/experiments/adamw_confined.diff).  The form is the one adamw_kernel has compiled to since round 2 (tests/golden/optim.pt: 2e-5).
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const XpAdamGroup& h, float coef) {
#pragma clang fp contract(off)
  g = g * coef;
  m = __builtin_fmaf(h.beta1, m, (1.0f - h.beta1) * g);
  v = __builtin_fmaf(h.beta2, v, g * ((1.0f - h.beta2) * g));
  const float denom = sqrtf(v) + h.eps;
  p = __builtin_fmaf(-h.step_size, m / denom, p);
  if (h.weight_decay > 0.f) p = __builtin_fmaf(-(h.lr * h.weight_decay), p, p);
}

__global__ __launch_bounds__(NT) void adamw_kernel(const XpAdamTensor* __restrict__ table,
                                                   const int32_t* __restrict__ chunk_map, StepArgs a,
                                                   const float* __restrict__ partials, int n_partials, float max_norm,
                                                   float* __restrict__ norm_out) {
  float coef = 1.0f;
  if (partials != nullptr) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += NT) s += partials[i];
    const float norm = sqrtf(block_sum(s));
    if (max_norm > 0.f) { const float c = max_norm / (norm + 1e-6f); coef = c < 1.0f ? c : 1.0f; }
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  }
  const int t = chunk_map[2 * blockIdx.x], c = chunk_map[2 * blockIdx.x + 1];
  const XpAdamTensor tt = table[t];
  const XpAdamGroup h = a.groups[a.grp[t]];
  const int64_t beg = (int64_t)c * CHUNK;
  const int64_t len = tt.numel - beg < CHUNK ? tt.numel - beg : CHUNK;
  float* p = reinterpret_cast<float*>(tt.p) + beg;
  float* m = reinterpret_cast<float*>(tt.m) + beg;
  float* v = reinterpret_cast<float*>(tt.v) + beg;
  const float* g = a.g[t] + beg;
  bf16_t* sb = tt.shadow && tt.shadow_dtype == XP_BF16 ? reinterpret_cast<bf16_t*>(tt.shadow) + beg : nullptr;
  float* sf = tt.shadow && tt.shadow_dtype == XP_F32 ? reinterpret_cast<float*>(tt.shadow) + beg : nullptr;
  const uintptr_t align = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                          reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(sf) |
                          (reinterpret_cast<uintptr_t>(sb) << 1);
  int64_t done = 0;
  if ((align & 15) == 0) {
    const int64_t n4 = len >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += NT) {
      f32x4 pp = *reinterpret_cast<f32x4*>(p + 4 * i), mm = *reinterpret_cast<f32x4*>(m + 4 * i),
            vv = *reinterpret_cast<f32x4*>(v + 4 * i);
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pp[e], me = mm[e], ve = vv[e];
        adam1(pe, gg[e], me, ve, h, coef);
        pp[e] = pe; mm[e] = me; vv[e] = ve;
      }
      *reinterpret_cast<f32x4*>(p + 4 * i) = pp;
      *reinterpret_cast<f32x4*>(m + 4 * i) = mm;
      *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
      if (sb) *reinterpret_cast<bf16x4*>(sb + 4 * i) = bf16x4{(bf16_t)pp[0], (bf16_t)pp[1], (bf16_t)pp[2], (bf16_t)pp[3]};
      if (sf) *reinterpret_cast<f32x4*>(sf + 4 * i) = pp;
    }
    done = n4 << 2;
  }
  for (int64_t i = done + threadIdx.x; i < len; i += NT) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam1(pp, g[i], mm, vv, h, coef);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (sb) sb[i] = (bf16_t)pp;
    if (sf) sf[i] = pp;
  }
}

}  // namespace

extern "C" int xp_grad_sqnorm_partials(const XpAdamTensor* table, const int32_t* chunk_map, int32_t n_chunks,
                                       const void* const* grads_host, int32_t n_tensors, float* partials, void* stream) {
  XP_REQUIRE(table && chunk_map && grads_host && partials && n_chunks > 0, "xp_grad_sqnorm_partials: null argument");
  XP_REQUIRE(n_tensors > 0 && n_tensors <= MAXT, "xp_grad_sqnorm_partials: n_tensors=%d not in 1..%d", n_tensors, MAXT);
  NormArgs a;
  for (int i = 0; i < n_tensors; ++i) {
    XP_REQUIRE(grads_host[i], "xp_grad_sqnorm_partials: gradient %d is null", i);
    a.g[i] = reinterpret_cast<const float*>(grads_host[i]);
  }
  for (int i = n_tensors; i < MAXT; ++i) a.g[i] = nullptr;
  sqnorm_partials_kernel<<<n_chunks, NT, 0, (hipStream_t)stream>>>(table, chunk_map, a, partials);
  XP_CHECK_LAUNCH("xp_grad_sqnorm_partials");
  return XP_OK;
}

extern "C" int xp_adamw_step(const XpAdamTensor* table, const int32_t* chunk_map, int32_t n_chunks,
                             const void* const* grads_host, const uint8_t* group_of_host, int32_t n_tensors,
                             const XpAdamGroup* groups_host, int32_t n_groups, const float* norm_partials,
                             int32_t n_norm_partials, float max_norm, float* grad_norm_out, void* stream) {
  XP_REQUIRE(table && chunk_map && grads_host && group_of_host && groups_host && n_chunks > 0, "xp_adamw_step: null argument");
  XP_REQUIRE(n_tensors > 0 && n_tensors <= MAXT, "xp_adamw_step: n_tensors=%d not in 1..%d", n_tensors, MAXT);
  XP_REQUIRE(n_groups > 0 && n_groups <= MAXG, "xp_adamw_step: n_groups=%d not in 1..%d", n_groups, MAXG);
  XP_REQUIRE(norm_partials || max_norm <= 0.f, "xp_adamw_step: max_norm > 0 needs the gradient-norm partials");
  XP_REQUIRE(!norm_partials || n_norm_partials > 0, "xp_adamw_step: n_norm_partials must be positive");
  StepArgs a;
  for (int i = 0; i < MAXT; ++i) { a.g[i] = nullptr; a.grp[i] = 0; }
  for (int i = 0; i < n_tensors; ++i) {
    XP_REQUIRE(grads_host[i], "xp_adamw_step: gradient %d is null", i);
    XP_REQUIRE(group_of_host[i] < n_groups, "xp_adamw_step: tensor %d has group %d >= %d", i, group_of_host[i], n_groups);
    a.g[i] = reinterpret_cast<const float*>(grads_host[i]);
    a.grp[i] = group_of_host[i];
  }
  for (int i = 0; i < n_groups; ++i) a.groups[i] = groups_host[i];
  for (int i = n_groups; i < MAXG; ++i) a.groups[i] = groups_host[0];
  adamw_kernel<<<n_chunks, NT, 0, (hipStream_t)stream>>>(table, chunk_map, a, norm_partials, n_norm_partials, max_norm,
                                                         grad_norm_out);
  XP_CHECK_LAUNCH("xp_adamw_step");
  return XP_OK;
}
