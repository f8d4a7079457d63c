// FETCH_SIZE / WRITE_SIZE calibration on known byte counts (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass):
//   k_read16  : every lane one 16-byte load, fully coalesced stream of N bytes            (the guide's calibrated case: FETCH_SIZE*1024*2 = N)
//   k_read8   : every lane one  8-byte load, fully coalesced stream of N bytes            (pointwise wgrad staging, depthwise halo staging)
//   k_read8_c96: the depthwise staging pattern: 8-byte units of a 64-channel block out of 96-byte pixels (C = 96, CBW = 64): 2/3 of every
//               pixel is touched, lines are shared with the neighbouring channel block's workgroup
//   k_write16 : 16-byte stores of N bytes
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_fetch.hip -o /tmp/probe_fetch ; prints the byte counts each kernel touches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k_read16(const uint4* __restrict__ p, int64_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read8(const uint2* __restrict__ p, int64_t n8, uint32_t* sink) {
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) { uint2 v = p[i]; acc ^= v.x ^ v.y; }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read8_c96(const uint8_t* __restrict__ p, int64_t npix, uint32_t* sink) {     // unit u: pixel u / 8, 8-byte chunk u % 8 of channels [0, 64)
  uint32_t acc = 0;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < npix * 8; u += (int64_t)gridDim.x * blockDim.x) {
    const uint2 v = *(const uint2*)(p + (u >> 3) * 96 + (u & 7) * 8); acc ^= v.x ^ v.y;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_write16(uint4* __restrict__ p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
int main() {
  const int64_t N = (int64_t)1 << 30;            // 1 GiB, far beyond L2 + MALL
  uint8_t* buf; uint32_t* sink;
  hipMalloc(&buf, N + 4096); hipMalloc(&sink, 4); hipMemset(buf, 1, N);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, N / 16, sink);
    hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, (const uint2*)buf, N / 8, sink);
    hipLaunchKernelGGL(k_read8_c96, dim3(4096), dim3(256), 0, 0, buf, N / 96, sink);
    hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, (uint4*)buf, N / 16);
  }
  hipDeviceSynchronize();
  printf("bytes touched per launch: read16 %lld  read8 %lld  read8_c96 %lld of %lld spanned  write16 %lld\n", (long long)N, (long long)N,
         (long long)(N / 96 * 64), (long long)(N / 96 * 96), (long long)N);
  return 0;
}
