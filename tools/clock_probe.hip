// What is one s_memtime tick, and what shader clock does an MFMA-dense kernel really run at on this box?
//   * a wave issuing independent v_mfma_f32_32x32x16_bf16 back to back occupies its SIMD's matrix pipe for exactly 32 shader
//     cycles per instruction (MI355X_MICROARCH.md, per-instruction constants), so N of them are a ruler of 32 N shader cycles;
//   * wall_clock64() is the constant 100 MHz device clock.
// The probe runs the ruler on 1 workgroup (idle chip) and on every SIMD of the chip (power-limited, like the GEMMs) with
// random and with zero operands, and prints shader cycles per s_memtime tick and the shader clock in GHz for each case.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/build/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void ruler(unsigned long long* out, const float* seed, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)seed[(lane * 8 + e) & 1023]; b[e] = (__bf16)seed[(lane * 8 + e + 512) & 1023]; }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {           // 4 independent accumulators: issue-bound, 32 cycles each on this wave's SIMD
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { out[0] = t1 - t0; out[1] = w1 - w0; }
  if (s == 12345.678f) out[2] = 1;             // keep the accumulators live
}

int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  float *seed, h[1024];
  hipMalloc(&seed, sizeof(h));
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  const int iters = 100000;                   // 4e5 MFMAs = 12.8e6 shader cycles (~6 ms)
  for (int zero = 0; zero < 2; ++zero) {
    for (int i = 0; i < 1024; ++i) h[i] = zero ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    for (int blocks : {1, 256, 512}) {        // 256-thread workgroups = one wave per SIMD of a CU; 512 blocks = 2 waves per SIMD
      for (int rep = 0; rep < 2; ++rep) {
        ruler<<<blocks, 256>>>(d, seed, iters); hipDeviceSynchronize();
      }
      unsigned long long r[2]; hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
      const double cycles = 32.0 * 4 * iters * (blocks > 256 ? 2 : 1), sec = (double)r[1] / (rate * 1e3);
      printf("%s operands, %3d workgroups: %llu s_memtime ticks, %.3f ms wall (100 MHz clock) for %.3g matrix-pipe cycles -> "
             "%.3f shader cycles per s_memtime tick, shader clock %.3f GHz, s_memtime rate %.3f GHz\n",
             zero ? "zero  " : "random", blocks, r[0], sec * 1e3, cycles, cycles / r[0], cycles / sec / 1e9, r[0] / sec / 1e9);
    }
  }
  return 0;
}
