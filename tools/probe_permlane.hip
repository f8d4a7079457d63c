// Probe: gfx950 v_permlane16_swap / v_permlane32_swap semantics (cross-row sums without LDS).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  v2u a = __builtin_amdgcn_permlane16_swap(lane, lane + 100, false, false);
  v2u b = __builtin_amdgcn_permlane32_swap(lane, lane + 100, false, false);
  out[lane * 4 + 0] = a[0]; out[lane * 4 + 1] = a[1]; out[lane * 4 + 2] = b[0]; out[lane * 4 + 3] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 16); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) if (l % 8 == 0 || l % 16 == 15) printf("lane %2d: swap16 -> (%3u, %3u)   swap32 -> (%3u, %3u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
