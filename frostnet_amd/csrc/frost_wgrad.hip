// Pointwise weight gradient: dWq[co][ci] += s_x * sum_p dc[p][co] * (q[p][ci] - zp)    (bf16 MFMA, K = pixels)
// Both operands are pixel-major in HBM, so a 128-pixel block of each is staged to LDS in its natural layout
// with coalesced loads and the K(pixel)-contiguous MFMA fragments are gathered from LDS with strided 16/8-bit
// reads.  A workgroup owns a 64x64 (co x ci) output tile; its 4 waves split the staged pixels (one 32-pixel
// K-step each) and are summed through LDS at the end; the pixel range is split across workgroups (split-M)
// and combined with fp32 atomics.
#include "frost_common.h"
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define WT 64
#define KPIX 128
#define RSD 136   // dc LDS row stride (bytes): 64 bf16 + 8
#define RSX 68    // x  LDS row stride (bytes): 64 int8 + 4

__global__ __launch_bounds__(256) void k_pw_wgrad(const uint16_t* __restrict__ dc, const int8_t* __restrict__ x, const float* qx,
                                                  int64_t npix, int cin, int cout, float* __restrict__ dwq, int nsplit) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * WT * WT * 4];   // 64 KB: staging (26 KB) then reduction
  uint8_t* dcs = lds; uint8_t* xs = lds + KPIX * RSD;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nci = (cin + WT - 1) / WT;
  const int tile = blockIdx.x % (((cout + WT - 1) / WT) * nci), split = blockIdx.x / (((cout + WT - 1) / WT) * nci);
  const int co0 = (tile / nci) * WT, ci0 = (tile % nci) * WT;
  const int zpo = __float_as_int(qx[FROST_Q_ZP]) - 128;      // zero point in the stored (offset-binary) domain
  const uint32_t zfill = (uint32_t)(zpo & 255) * 0x01010101u;

  v4f acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};

  const int64_t nblk = (npix + KPIX - 1) / KPIX;
  for (int64_t blk = split; blk < nblk; blk += nsplit) {
    const int64_t p0 = blk * KPIX;
    __syncthreads();
    for (int u = tid; u < KPIX * 8; u += 256) {          // dc: 8 x 16B per pixel row
      const int pix = u >> 3, c8 = u & 7; const int64_t gp = p0 + pix; const int co = co0 + c8 * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gp < npix && co < cout) v = *(const uint4*)(dc + gp * cout + co);
      *(uint2*)(dcs + pix * RSD + c8 * 16) = make_uint2(v.x, v.y);
      *(uint2*)(dcs + pix * RSD + c8 * 16 + 8) = make_uint2(v.z, v.w);
    }
    for (int u = tid; u < KPIX * 8; u += 256) {          // x: 8 x 8B per pixel row
      const int pix = u >> 3, c8 = u & 7; const int64_t gp = p0 + pix; const int ci = ci0 + c8 * 8;
      uint2 v = make_uint2(zfill, zfill);
      if (gp < npix && ci < cin) v = *(const uint2*)(x + gp * cin + ci);
      *(uint32_t*)(xs + pix * RSX + c8 * 8) = v.x; *(uint32_t*)(xs + pix * RSX + c8 * 8 + 4) = v.y;
    }
    __syncthreads();
    const int pb = w * 32 + g * 8;                        // this lane's 8 pixels (K slots)
    v4i afr[4], bfr[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {                         // A: dc^T rows = co
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint16_t lo = *(const uint16_t*)(dcs + (pb + 2 * e) * RSD + (a * 16 + i16) * 2);
        const uint16_t hi = *(const uint16_t*)(dcs + (pb + 2 * e + 1) * RSD + (a * 16 + i16) * 2);
        pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
      }
      afr[a] = (v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {                         // B: x^T cols = ci, values (q' - zp') exact in bf16
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lo = (int)*(const int8_t*)(xs + (pb + 2 * e) * RSX + b * 16 + i16) - zpo;
        const int hi = (int)*(const int8_t*)(xs + (pb + 2 * e + 1) * RSX + b * 16 + i16) - zpo;
        pk[e] = (uint32_t)(__float_as_uint((float)lo) >> 16) | (__float_as_uint((float)hi) & 0xffff0000u);
      }
      bfr[b] = (v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[a]), __builtin_bit_cast(v8bf, bfr[b]), acc[a][b], 0, 0, 0);
  }
  // cross-wave reduction: red[w][co_local][ci_local]
  __syncthreads();
  float* red = (float*)lds;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(w * WT + a * 16 + 4 * g + r) * WT + b * 16 + i16] = acc[a][b][r];
  __syncthreads();
  const float sx = qx[FROST_Q_SCALE];
  for (int i = tid; i < WT * WT; i += 256) {
    const int co = co0 + i / WT, ci = ci0 + i % WT;
    if (co < cout && ci < cin) {
      const float v = red[i] + red[WT * WT + i] + red[2 * WT * WT + i] + red[3 * WT * WT + i];
      atomicAdd(dwq + (int64_t)co * cin + ci, v * sx);
    }
  }
}
extern "C" int frost_pw_wgrad(const uint16_t* dc, const int8_t* x, const float* qrec_x, int64_t npix, int cin, int cout,
                              float* dwq, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 8 == 0, "pw_wgrad: channels must be multiples of 8");
  const int ntile = ((cout + WT - 1) / WT) * ((cin + WT - 1) / WT);
  const int64_t nblk = (npix + KPIX - 1) / KPIX;
  int nsplit = (1024 + ntile - 1) / ntile; if (nsplit > nblk) nsplit = (int)nblk; if (nsplit < 1) nsplit = 1;
  hipLaunchKernelGGL(k_pw_wgrad, dim3(ntile * nsplit), dim3(256), 0, as_stream(stream), dc, x, qrec_x, npix, cin, cout, dwq, nsplit);
  return frost_check_launch("pw_wgrad");
}
