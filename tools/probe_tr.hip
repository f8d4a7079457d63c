// Probe: empirical semantics of gfx950 LDS transpose reads (ds_read_b64_tr_b16 / _tr_b8). Prints per-lane results.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(uint16_t* out16, uint8_t* out8, int stride16, int stride8) {
  __shared__ __attribute__((aligned(16))) uint16_t l16[4096];
  __shared__ __attribute__((aligned(16))) uint8_t l8[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) { l16[i] = (uint16_t)i; l8[i] = (uint8_t)(i & 255); }
  __syncthreads();
  int lane = threadIdx.x;
  // each lane supplies the address of row (lane&15)... try: addr = (lane%16)*stride + (lane/16)*8 bytes
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(l16 + (lane & 15) * stride16 + (lane >> 4) * 4));
  for (int e = 0; e < 4; ++e) out16[lane * 4 + e] = (uint16_t)r[e];
  v2i q = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(l8 + (lane & 15) * stride8 + (lane >> 4) * 8));
  for (int e = 0; e < 8; ++e) out8[lane * 8 + e] = (uint8_t)(((e < 4 ? (uint32_t)q[0] : (uint32_t)q[1]) >> (8 * (e & 3))) & 255);
}
int main() {
  uint16_t* d16; uint8_t* d8; hipMalloc(&d16, 64 * 4 * 2); hipMalloc(&d8, 64 * 8);
  int s16 = 64, s8 = 64;   // row strides in elements
  probe<<<1, 64>>>(d16, d8, s16, s8);
  uint16_t h16[256]; uint8_t h8[512];
  hipMemcpy(h16, d16, sizeof(h16), hipMemcpyDeviceToHost); hipMemcpy(h8, d8, sizeof(h8), hipMemcpyDeviceToHost);
  printf("tr16_b64: lane supplies &l16[(lane&15)*%d + (lane>>4)*4]; value v = row*%d + col\n", s16, s16);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h16[l * 4 + e] / s16, h16[l * 4 + e] % s16); printf("\n"); }
  printf("tr8_b64: lane supplies &l8[(lane&15)*%d + (lane>>4)*8]; value = (row*%d+col)&255\n", s8, s8);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 8; ++e) printf(" %3d", h8[l * 8 + e]); printf("\n"); }
  return 0;
}
