// Hardware-layout probes: single-wave kernels that expose the raw lane maps of the MFMA instructions and
// of ds_read_b64_tr_b16, so tests/test_probe_gpu.py can pin the assumptions documented in common.h
// against the real gfx950 instead of trusting a table.
#include "common.h"

namespace {

// a, b: [64 lanes][8] bf16 fragments exactly as handed to the instruction; c: [64][4] accumulator out.
__global__ void probe_mfma_bf16_kernel(const bf16x8* a, const bf16x8* b, f32x4* c) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[l], b[l], acc, 0, 0, 0);
  c[l] = acc;
}
// a, b: [64] floats (one per lane); c: [64][4]
__global__ void probe_mfma_f32_kernel(const float* a, const float* b, f32x4* c) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
  c[l] = acc;
}
// in: 4096 u16 copied to LDS verbatim; lane l reads ds_read_b64_tr_b16 at byte offset off[l]; out [64][4] u16
__global__ void probe_tr16_kernel(const unsigned short* in, const int* off, i16x4* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  out[l] = lds_read_tr16(reinterpret_cast<const char*>(lds) + off[l]);
}

// Packed-fp32 probe (determinism hunt): every wave repeats one packed-fp32 instruction form and compares
// both halves with the same arithmetic done by unpacked VALU instructions; err[(variant * 64 + lane) * 2 + half] counts
// the mismatches.  Run beside an MFMA kernel on another stream it tells a hardware interaction from a compiler problem.
//   0: v_pk_add_f32 x, m                      1: ... op_sel_hi:[1,0]            2: ... neg_lo:[0,1] neg_hi:[0,1]
//   3: ... op_sel_hi:[1,0] neg (x - m.lo)     4: ... op_sel:[0,1] neg (x - m.hi) 5: v_pk_mul_f32 op_sel_hi:[1,0]
//   6: v_pk_fma_f32 x, m, x                   7: v_pk_mul_f32 x, m
//   8: v_pk_add_f32 op_sel:[0,1] (no neg)     9: v_pk_mul_f32 op_sel:[0,1]      10: v_pk_fma_f32 x, m, x op_sel:[1,0,0]
//  11: v_pk_add_f32 op_sel:[1,0]             12: v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (halves of m swapped)
//  13: v_pk_mul_f32 op_sel_hi:[0,1]              14: v_pk_mov_b32 x, m op_sel:[1,0]  (the form in the GEMM epilogues)
constexpr int PK_VARIANTS = 15;
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int V>
__device__ __forceinline__ void pk_variant(f32x2 x, f32x2 m, f32x2& r, float& e0, float& e1) {
  if constexpr (V == 0) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %5" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 1) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %4" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 2) { asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %5" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 3) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %4" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 4) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_sub_f32 %0, %2, %5\n\tv_sub_f32 %1, %3, %5" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 5) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 6) { asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_fma_f32 %0, %2, %4, %2\n\tv_fma_f32 %1, %3, %5, %3" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 7) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 8) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_add_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %5" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 9) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_mul_f32 %0, %2, %5\n\tv_mul_f32 %1, %3, %5" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 10) { asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_fma_f32 %0, %3, %4, %2\n\tv_fma_f32 %1, %3, %5, %3" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 11) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_add_f32 %0, %3, %4\n\tv_add_f32 %1, %3, %5" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 12) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_add_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %4" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
  if constexpr (V == 14) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(e0), "=&v"(e1) : "v"(x[1]), "v"(m[0])); }   // D.hi = op_sel[1] ? S1.hi : S1.lo
  if constexpr (V == 13) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=&v"(r) : "v"(x), "v"(m));
    asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %2, %5" : "=&v"(e0), "=&v"(e1) : "v"(x[0]), "v"(x[1]), "v"(m[0]), "v"(m[1])); }
}
template <int V>
__device__ __forceinline__ void pk_run(unsigned* err, int iters, unsigned s, int lane) {
  unsigned e_lo = 0, e_hi = 0;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    const f32x2 x = {__uint_as_float(0x3f000000u | (s >> 9)), __uint_as_float(0x3f000000u | ((s * 7u) >> 9))};
    const f32x2 m = {__uint_as_float(0x3d000000u | ((s * 13u) >> 9)), __uint_as_float(0x3d000000u | ((s * 29u) >> 9))};
    f32x2 r; float e0, e1;
    pk_variant<V>(x, m, r, e0, e1);
    e_lo += r[0] != e0; e_hi += r[1] != e1;
  }
  if (e_lo) atomicAdd(err + (V * 64 + lane) * 2, e_lo);
  if (e_hi) atomicAdd(err + (V * 64 + lane) * 2 + 1, e_hi);
}
__global__ __launch_bounds__(256) void probe_pk_kernel(unsigned* err, int iters, unsigned seed) {
  const int lane = threadIdx.x & 63;
  const unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
  pk_run<0>(err, iters, s, lane); pk_run<1>(err, iters, s + 1, lane); pk_run<2>(err, iters, s + 2, lane); pk_run<3>(err, iters, s + 3, lane);
  pk_run<4>(err, iters, s + 4, lane); pk_run<5>(err, iters, s + 5, lane); pk_run<6>(err, iters, s + 6, lane); pk_run<7>(err, iters, s + 7, lane);
  pk_run<8>(err, iters, s + 8, lane); pk_run<9>(err, iters, s + 9, lane); pk_run<10>(err, iters, s + 10, lane); pk_run<11>(err, iters, s + 11, lane);
  pk_run<12>(err, iters, s + 12, lane); pk_run<13>(err, iters, s + 13, lane); pk_run<14>(err, iters, s + 14, lane);
}

}  // namespace

// A memory-bound copy confined to `blocks` workgroups that loops `iters` times over its buffer: a stand-in for a ring
// collective's kernel (RCCL runs a few dozen channel workgroups that stream the gradient buckets) on a second stream, to measure on
// ONE GPU what such a neighbour costs the backward pass (tools/contention_probe.py).
__global__ __launch_bounds__(256) void probe_stream_copy_kernel(u32x4* dst, const u32x4* src, long n16, int iters) {
  for (int it = 0; it < iters; ++it)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int xp_probe_stream_copy(void* dst, const void* src, int64_t nbytes, int32_t blocks, int32_t iters, void* stream) {
  XP_REQUIRE(dst && src && nbytes >= 16 && blocks > 0 && iters > 0, "xp_probe_stream_copy: bad arguments");
  probe_stream_copy_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((u32x4*)dst, (const u32x4*)src, nbytes / 16, iters);
  XP_CHECK_LAUNCH("xp_probe_stream_copy");
  return XP_OK;
}

// The same copy with the resource footprint of RCCL's gfx950 collective kernel (rcclGenericKernel<...> in the librccl.so this image
// ships, read with llvm-readelf --notes from its unbundled gfx950 code object: 256 threads, 261-280 VGPRs + 17-32 AGPRs, 19,744 B of
// LDS, 352 B of scratch).  More than 256 registers per lane means ONE wave per SIMD: such a workgroup needs four SIMDs with their
// whole register file free, i.e. it cannot share a CU with a 256x256-GEMM workgroup (2 waves x ~244 registers per SIMD) -- it only
// runs on CUs no GEMM workgroup occupies, and holds them for as long as the collective lasts.
__global__ __launch_bounds__(256) void probe_stream_copy_fat_kernel(u32x4* dst, const u32x4* src, long n16, int iters) {
  __shared__ unsigned pad[19744 / 4];
  asm volatile("v_mov_b32 v255, 0" ::: "v255");                      // vgpr_count 256
  asm volatile("v_accvgpr_write_b32 a31, 0" ::: "a31");              // + 32 AGPRs -> 288 registers per lane: one wave per SIMD
  if (threadIdx.x == 0) pad[blockIdx.x % (19744 / 4)] = 0;           // (keeps the LDS allocation)
  for (int it = 0; it < iters; ++it)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int xp_probe_stream_copy_fat(void* dst, const void* src, int64_t nbytes, int32_t blocks, int32_t iters, void* stream) {
  XP_REQUIRE(dst && src && nbytes >= 16 && blocks > 0 && iters > 0, "xp_probe_stream_copy_fat: bad arguments");
  probe_stream_copy_fat_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((u32x4*)dst, (const u32x4*)src, nbytes / 16, iters);
  XP_CHECK_LAUNCH("xp_probe_stream_copy_fat");
  return XP_OK;
}

extern "C" int xp_probe_pk_f32(void* err, int32_t iters, int32_t blocks, uint32_t seed, void* stream) {
  probe_pk_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((unsigned*)err, iters, seed);
  XP_CHECK_LAUNCH("xp_probe_pk_f32");
  return XP_OK;
}

extern "C" int xp_probe_mfma_bf16(const void* a, const void* b, float* c, void* stream) {
  probe_mfma_bf16_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const bf16x8*)a, (const bf16x8*)b, (f32x4*)c);
  XP_CHECK_LAUNCH("xp_probe_mfma_bf16");
  return XP_OK;
}
extern "C" int xp_probe_mfma_f32(const float* a, const float* b, float* c, void* stream) {
  probe_mfma_f32_kernel<<<1, 64, 0, (hipStream_t)stream>>>(a, b, (f32x4*)c);
  XP_CHECK_LAUNCH("xp_probe_mfma_f32");
  return XP_OK;
}
extern "C" int xp_probe_tr16(const void* in, const int32_t* lane_byte_off, void* out, void* stream) {
  probe_tr16_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const unsigned short*)in, lane_byte_off, (i16x4*)out);
  XP_CHECK_LAUNCH("xp_probe_tr16");
  return XP_OK;
}
