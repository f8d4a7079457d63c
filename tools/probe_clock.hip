// Probe: effective shader clock under a sustained VALU/memory load (clock64 vs wall_clock64).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(float* out, long long* t, int iters) {
  long long c0 = clock64(), w0 = wall_clock64();
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
int main() {
  float* o; long long* t; hipMalloc(&o, 2048 * 256 * 4); hipMalloc(&t, 16);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  for (int rep = 0; rep < 6; ++rep) {
    spin<<<2048, 256>>>(o, t, 2000000);
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("rep %d: shader cycles %lld, wall ticks %lld (wall rate %d kHz) -> shader clock %.1f MHz\n", rep, h[0], h[1], rate, (double)h[0] / ((double)h[1] / rate) / 1e3);
  }
  return 0;
}
