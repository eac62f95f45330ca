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

}  // namespace

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
