// Probe: cost of first-touch (cold) instruction fetch vs warm re-execution of a long straight-line code region.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R8(x) x x x x x x x x
#define BODY R8(R8(R8(asm volatile("v_add_u32 %0, %0, 1\n\tv_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));)))   // 512 * 2 instrs = 1024 instrs = 4-8 KB
__global__ void k(unsigned* out, long long* t) {
  unsigned a = threadIdx.x, b = blockIdx.x;
  long long c[4];
  for (int it = 0; it < 3; ++it) {
    long long c0 = clock64();
    BODY
    long long c1 = clock64();
    c[it] = c1 - c0;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c[0]; t[1] = c[1]; t[2] = c[2]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
  unsigned* o; long long* t; hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&t, 64);
  for (int rep = 0; rep < 3; ++rep) {
    for (int blocks : {1, 512}) {
      k<<<blocks, 256>>>(o, t); long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
      printf("rep %d, %3d blocks: 1024-instr region: first pass %lld cycles, second %lld, third %lld\n", rep, blocks, h[0], h[1], h[2]);
    }
  }
  return 0;
}
