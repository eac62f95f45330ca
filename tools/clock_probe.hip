// s_memtime ticks vs the 100 MHz wall clock: what does one s_memtime tick mean on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;      // dependent FMA chain
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (unsigned long long)x; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  for (int blocks : {1, 256, 4096}) {
    k<<<blocks, 256>>>(d, 2000000); hipDeviceSynchronize();
    unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    printf("blocks=%d memtime ticks=%llu wall ticks=%llu (wall clock rate %d kHz) -> %.1f MHz per memtime tick; %.2f ticks per FMA iter\n",
           blocks, h[0], h[1], rate, (double)h[0] / ((double)h[1] / rate / 1e3) / 1e6, (double)h[0] / 2000000);
  }
  return 0;
}
