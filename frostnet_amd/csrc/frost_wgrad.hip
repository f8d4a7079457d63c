// Pointwise weight gradient: dWq[co][ci] += s_x * sum_p dc[p][co] * (q[p][ci] - zp)    (bf16 MFMA, K = pixels)
// Both operands are pixel-major in HBM, so a 128-pixel block of each is staged to LDS in its natural layout with
// coalesced loads; the K(pixel)-contiguous MFMA fragments are produced by the gfx950 LDS transpose reads
// (ds_read_b64_tr_b16 for the bf16 dc operand, ds_read_b64_tr_b8 for the int8 activation operand -- lane/pointer
// semantics measured with tools/probe_tr.hip).  A workgroup owns a 64x64 (co x ci) output tile; its 4 waves split
// the staged pixels (one 32-pixel K-step each) and are summed through LDS at the end; the pixel range is split
// across workgroups (split-M) and combined with fp32 atomics.
#include "frost_common.h"
#include <stdlib.h>
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

#define WT 64
#define KPIX 128
#ifndef RSD
#define RSD 136   // dc LDS row stride (bytes): 64 bf16 + 8
#endif
#ifndef RSX
#define RSX 72    // x  LDS row stride (bytes): 64 int8 + 8  (8-byte aligned rows for the 8-byte transpose reads)
#endif

__device__ __forceinline__ uint32_t pack_trunc_bf16(float lo, float hi) {   // exact for |integers| <= 256
  return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}

__global__ __launch_bounds__(256, 3) void k_pw_wgrad(const uint16_t* __restrict__ dc, const int8_t* __restrict__ x, const float* qx,
                                                  int64_t npix, int cin, int cout, float* __restrict__ dwq, int nsplit, int xmap) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[KPIX * RSD + KPIX * RSX];   // 26.6 KB: staging, then the 16 KB reduction tile
  uint8_t* dcs = lds; uint8_t* xs = lds + KPIX * RSD;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nci = (cin + WT - 1) / WT;
  const int ntile = ((cout + WT - 1) / WT) * nci;
  // XCD-aware block -> (tile, split) map (speed only): block b runs on XCD b % 8 and every output tile of one pixel split reads the SAME
  // dc / x pixel blocks, so the tiles of a split are placed on one XCD and share that XCD's L2 (the plain map spread them over all eight
  // L2s: each re-read of a pixel block by another tile was an HBM fetch -- counters showed 2.1x the algorithmic bytes)
  int tile, split;
  if (xmap) { const int b = blockIdx.x, xcd = b & 7, jb = b >> 3; split = xcd + 8 * (jb / ntile); tile = jb % ntile; }
  else { tile = blockIdx.x % ntile; split = blockIdx.x / ntile; }
  const int co0 = (tile / nci) * WT, ci0 = (tile % nci) * WT;
  const int zpu = __float_as_int(qx[FROST_Q_ZP]);                  // zero point in the unsigned index domain
  const float zpf = (float)zpu;
  const uint32_t zfill = (uint32_t)((zpu - 128) & 255) * 0x01010101u;

  v4f acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};

  // transpose-read source addresses (fixed per lane): this wave's 32 pixels = rows w*32 + 8g + e
  const int pb = w * 32 + g * 8;
  const uint8_t* a_src = dcs + (pb + (i16 >> 2)) * RSD + (i16 & 3) * 8;          // + a*32 bytes, second read + 4 rows
  const uint8_t* b_src = xs + (pb + (i16 >> 1)) * RSX + (i16 & 1) * 8;           // + b*16 bytes

  int na = (cout - co0 + 15) / 16; if (na > 4) na = 4;     // live 16-channel tiles of this workgroup's 64x64 output tile
  int nb = (cin - ci0 + 15) / 16; if (nb > 4) nb = 4;
  const int64_t nblk = (npix + KPIX - 1) / KPIX;
  // register prefetch of the next 128-pixel block (24 VGPRs) so its HBM latency overlaps the MFMAs of the current one
  uint4 pd[4]; uint2 px[4];
#define WG_PREFETCH(BLK_)                                                                                             \
  {                                                                                                                   \
    const int64_t q0_ = (BLK_) * KPIX;                                                                                \
    _Pragma("unroll") for (int jn = 0; jn < 4; ++jn) {                                                                \
      const int u_ = tid + jn * 256; const int pix_ = u_ >> 3, c8_ = u_ & 7; const int64_t gp_ = q0_ + pix_;          \
      pd[jn] = make_uint4(0, 0, 0, 0); px[jn] = make_uint2(zfill, zfill);                                             \
      if (gp_ < npix && (co0 + c8_ * 8) < cout) pd[jn] = *(const uint4*)(dc + gp_ * cout + co0 + c8_ * 8);            \
      if (gp_ < npix && (ci0 + c8_ * 8) < cin) px[jn] = *(const uint2*)(x + gp_ * cin + ci0 + c8_ * 8);               \
    }                                                                                                                 \
  }
  if (split < nblk) WG_PREFETCH((int64_t)split)
  for (int64_t blk = split; blk < nblk; blk += nsplit) {
    __syncthreads();
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u = tid + jn * 256; const int pix = u >> 3, c8 = u & 7;
      *(uint2*)(dcs + pix * RSD + c8 * 16) = make_uint2(pd[jn].x, pd[jn].y);
      *(uint2*)(dcs + pix * RSD + c8 * 16 + 8) = make_uint2(pd[jn].z, pd[jn].w);
      *(uint2*)(xs + pix * RSX + c8 * 8) = px[jn];
    }
    __syncthreads();
    if (blk + nsplit < nblk) WG_PREFETCH(blk + nsplit)
    v4i afr[4], bfr[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {                         // A: dc^T, rows = co; lane i16 gets channel a*16+i16, 8 pixels
      if (a < na) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a_src + a * 32));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a_src + a * 32 + 4 * RSD));
        afr[a] = (v4i){(int)((uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16)), (int)((uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16)),
                       (int)((uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16)), (int)((uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16))};
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {                         // B: x^T, cols = ci; lane i16 gets channel b*16+i16, 8 pixels
      if (b < nb) {
        const v2i raw = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(b_src + b * 16));
        const uint32_t u0 = (uint32_t)raw[0] ^ 0x80808080u, u1 = (uint32_t)raw[1] ^ 0x80808080u;   // offset-binary -> unsigned index
        bfr[b] = (v4i){(int)pack_trunc_bf16((float)(u0 & 255u) - zpf, (float)((u0 >> 8) & 255u) - zpf),
                       (int)pack_trunc_bf16((float)((u0 >> 16) & 255u) - zpf, (float)(u0 >> 24) - zpf),
                       (int)pack_trunc_bf16((float)(u1 & 255u) - zpf, (float)((u1 >> 8) & 255u) - zpf),
                       (int)pack_trunc_bf16((float)((u1 >> 16) & 255u) - zpf, (float)(u1 >> 24) - zpf)};
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (a < na && b < nb)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[a]), __builtin_bit_cast(v8bf, bfr[b]), acc[a][b], 0, 0, 0);
  }
  // cross-wave reduction through LDS float atomics into one 64x64 tile
  __syncthreads();
  float* red = (float*)lds;
  for (int i = tid; i < WT * WT; i += 256) red[i] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (a < na && b < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&red[(a * 16 + 4 * g + r) * WT + b * 16 + i16], acc[a][b][r]);
      }
  __syncthreads();
  const float sx = qx[FROST_Q_SCALE];
  for (int i = tid; i < WT * WT; i += 256) {
    const int co = co0 + i / WT, ci = ci0 + i % WT;
    if (co < cout && ci < cin) atomicAdd(dwq + (int64_t)co * cin + ci, red[i] * sx);
  }
}
// Large-channel variant: 128x128 (co x ci) output tile over ALL 128 staged pixels (4 K-steps), 8 waves: wave w owns the
// 64 x 32 block (co half w>>2, ci quarter w&3): 2x the arithmetic intensity per staged byte of the small kernel, no cross-wave
// reduction, and 32 accumulator + 24 prefetch registers per lane, so two 8-wave workgroups are resident per CU (the 4-wave
// version needed 224 VGPRs: 8 waves per CU, latency-bound).
#define BT 128
#ifndef RSD2
#define RSD2 264   // 128 bf16 + 8 bytes
#endif
#ifndef RSX2
#define RSX2 136   // 128 int8 + 8 bytes
#endif
__global__ __launch_bounds__(512, 2) void k_pw_wgrad_big(const uint16_t* __restrict__ dc, const int8_t* __restrict__ x, const float* qx,
                                                         int64_t npix, int cin, int cout, float* __restrict__ dwq, int nsplit, int xmap) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[KPIX * RSD2 + KPIX * RSX2];   // 51 KB
  uint8_t* dcs = lds; uint8_t* xs = lds + KPIX * RSD2;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qa = w >> 2, qb = w & 3;                                   // co half, ci quarter of this wave
  const int nci = (cin + BT - 1) / BT;
  const int ntile = ((cout + BT - 1) / BT) * nci;
  int tile, split;                   // XCD-aware map, see k_pw_wgrad
  if (xmap) { const int b = blockIdx.x, xcd = b & 7, jb = b >> 3; split = xcd + 8 * (jb / ntile); tile = jb % ntile; }
  else { tile = blockIdx.x % ntile; split = blockIdx.x / ntile; }
  const int co0 = (tile / nci) * BT, ci0 = (tile % nci) * BT;
  const int zpu = __float_as_int(qx[FROST_Q_ZP]);
  const float zpf = (float)zpu;
  const uint32_t zfill = (uint32_t)((zpu - 128) & 255) * 0x01010101u;
  int na = (cout - co0 - qa * 64 + 15) / 16; na = na < 0 ? 0 : (na > 4 ? 4 : na);
  int nb = (cin - ci0 - qb * 32 + 15) / 16; nb = nb < 0 ? 0 : (nb > 2 ? 2 : nb);
  v4f acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
  const uint8_t* a_src = dcs + (g * 8 + (i16 >> 2)) * RSD2 + qa * 128 + (i16 & 3) * 8;
  const uint8_t* b_src = xs + (g * 8 + (i16 >> 1)) * RSX2 + qb * 32 + (i16 & 1) * 8;
  const int64_t nblk = (npix + KPIX - 1) / KPIX;
  uint4 pd[4]; uint2 px[4];         // register prefetch of the next block: HBM/L2 latency overlaps the MFMAs
#define WGB_PREFETCH(BLK_)                                                                                            \
  {                                                                                                                   \
    const int64_t q0_ = (BLK_) * KPIX;                                                                                \
    _Pragma("unroll") for (int jn = 0; jn < 4; ++jn) {                                                                \
      const int u_ = tid + jn * 512; const int pix_ = u_ >> 4, c8_ = u_ & 15; const int64_t gp_ = q0_ + pix_;         \
      pd[jn] = make_uint4(0, 0, 0, 0); px[jn] = make_uint2(zfill, zfill);                                             \
      if (gp_ < npix && (co0 + c8_ * 8) < cout) pd[jn] = *(const uint4*)(dc + gp_ * cout + co0 + c8_ * 8);            \
      if (gp_ < npix && (ci0 + c8_ * 8) < cin) px[jn] = *(const uint2*)(x + gp_ * cin + ci0 + c8_ * 8);               \
    }                                                                                                                 \
  }
  if (split < nblk) WGB_PREFETCH((int64_t)split)
  for (int64_t blk = split; blk < nblk; blk += nsplit) {
    __syncthreads();
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u = tid + jn * 512; const int pix = u >> 4, c8 = u & 15;
      *(uint2*)(dcs + pix * RSD2 + c8 * 16) = make_uint2(pd[jn].x, pd[jn].y);
      *(uint2*)(dcs + pix * RSD2 + c8 * 16 + 8) = make_uint2(pd[jn].z, pd[jn].w);
      *(uint2*)(xs + pix * RSX2 + c8 * 8) = px[jn];
    }
    __syncthreads();
    if (blk + nsplit < nblk) WGB_PREFETCH(blk + nsplit)
    if (na > 0 && nb > 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        v4i afr[4], bfr[2];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (a < na) {
            const v2i lo = __builtin_bit_cast(v2i, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a_src + ks * 32 * RSD2 + a * 32)));
            const v2i hi = __builtin_bit_cast(v2i, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a_src + ks * 32 * RSD2 + a * 32 + 4 * RSD2)));
            afr[a] = (v4i){lo[0], lo[1], hi[0], hi[1]};
          }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (b < nb) {
            const v2i raw = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(b_src + ks * 32 * RSX2 + b * 16));
            const uint32_t u0 = (uint32_t)raw[0] ^ 0x80808080u, u1 = (uint32_t)raw[1] ^ 0x80808080u;
            bfr[b] = (v4i){(int)pack_trunc_bf16((float)(u0 & 255u) - zpf, (float)((u0 >> 8) & 255u) - zpf),
                           (int)pack_trunc_bf16((float)((u0 >> 16) & 255u) - zpf, (float)(u0 >> 24) - zpf),
                           (int)pack_trunc_bf16((float)(u1 & 255u) - zpf, (float)((u1 >> 8) & 255u) - zpf),
                           (int)pack_trunc_bf16((float)((u1 >> 16) & 255u) - zpf, (float)(u1 >> 24) - zpf)};
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            if (a < na && b < nb)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[a]), __builtin_bit_cast(v8bf, bfr[b]), acc[a][b], 0, 0, 0);
      }
    }
  }
  const float sx = qx[FROST_Q_SCALE];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if (a < na && b < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + qa * 64 + a * 16 + 4 * g + r, ci = ci0 + qb * 32 + b * 16 + i16;
          if (co < cout && ci < cin) atomicAdd(dwq + (int64_t)co * cin + ci, acc[a][b][r] * sx);
        }
      }
}
// split counts >= 8 become multiples of 8 so that the XCD-aware map above is a bijection (FROST_WG_XCD=0: the plain map, for A/B runs)
static int xcd_round(int nsplit, int64_t nblk, int* xmap) {
  static const int on = getenv("FROST_WG_XCD") ? atoi(getenv("FROST_WG_XCD")) : 1;
  *xmap = 0;
  if (nsplit < 8) return nsplit;
  int r = (nsplit + 7) & ~7; if (r > nblk) r = nsplit & ~7;
  if (r < 8) return nsplit;
  *xmap = on;
  return r;
}
extern "C" int frost_pw_wgrad(const uint16_t* dc, const int8_t* x, const float* qrec_x, int64_t npix, int cin, int cout,
                              float* dwq, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 8 == 0, "pw_wgrad: channels must be multiples of 8");
  const int64_t nblk = (npix + KPIX - 1) / KPIX;
  if (cin > 64 && cout > 64) {      // wide layers: 128x128 tiles, quadrant-per-wave
    const int ntile = ((cout + BT - 1) / BT) * ((cin + BT - 1) / BT);
    static int target = getenv("FROST_WG_TARGET") ? atoi(getenv("FROST_WG_TARGET")) : 256;     // one 8-wave workgroup per CU: the kernel runs beside the main stream (512 = full residency measured 0.6 % slower end to end, 128 also slower)
    int nsplit = (target + ntile - 1) / ntile; if (nsplit > nblk) nsplit = (int)nblk; if (nsplit < 1) nsplit = 1;
    int xmap; nsplit = xcd_round(nsplit, nblk, &xmap);
    hipLaunchKernelGGL(k_pw_wgrad_big, dim3(ntile * nsplit), dim3(512), 0, as_stream(stream), dc, x, qrec_x, npix, cin, cout, dwq, nsplit, xmap);
    return frost_check_launch("pw_wgrad_big");
  }
  const int ntile = ((cout + WT - 1) / WT) * ((cin + WT - 1) / WT);
  // ~512 workgroups: every split ends with one fp32 atomic per output element, and more splits than that cost more in atomics than they
  // gain in parallelism (168->40 @28x28, B=512: 160 / 120 / 91 / 82 us at 2048 / 1024 / 512 / 256 workgroups; 96->24 @56x56: 151 / 125 / 95 / 131)
  int nsplit = (512 + ntile - 1) / ntile;
  // every split ends with one fp32 atomic per output element: on low-resolution layers hundreds of splits hammering the same few
  // thousand addresses cost more than the GEMM (measured 56 us for a 7 MB layer), so a split keeps at least 4 pixel blocks
  if (nsplit > nblk / 4) nsplit = (int)(nblk / 4); if (nsplit < 1) nsplit = 1;
  int xmap; nsplit = xcd_round(nsplit, nblk, &xmap);
  hipLaunchKernelGGL(k_pw_wgrad, dim3(ntile * nsplit), dim3(256), 0, as_stream(stream), dc, x, qrec_x, npix, cin, cout, dwq, nsplit, xmap);
  return frost_check_launch("pw_wgrad");
}

// ------------------------------------------------------------------------------------------------ data gradient of wide pointwise layers
// dx[p][ci] (+)= s_w * sum_co dc[p][co] * wq[co][ci] as a plain bf16 GEMM with M = pixels, N = Cin, K = Cout (K = 312 .. 1728: dc rows too long
// for k_pw's DMA-staged tile, whose chunked fallback is a chain of synchronised stages).  Workgroup = 256 pixels x up to 8 input-channel tiles;
// wave w owns 64 pixels (4 MFMA column blocks) x all those tiles: 32 accumulators, every weight fragment feeds 4 MFMAs and every dc fragment 8.
//   dc fragments come straight from global memory (a wave owns its pixel rows: nothing to share, 16 bytes per lane = 64-byte row pieces),
//   requested one K step ahead; the weight fragments of a K stage (2 steps of 32) go through LDS once per workgroup, double-buffered.
// wt_pack: [ci tile][K step of 32][lane][8] bf16 -- the A operand of mfma_f32_16x16x32_bf16 as frost_weight_prep lays it out.
typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));
#define DGW_NT 8        // input-channel tiles per workgroup
#ifndef DGW_KS
#define DGW_KS 2        // K steps per LDS stage
#endif
__global__ __launch_bounds__(256, 2) void k_dgrad_wide(const uint16_t* __restrict__ dc, const uint16_t* __restrict__ wt, const float* qw, int64_t npix,
                                                       int cout, int cin, int KB, int CIT, int per, uint16_t* __restrict__ dx, int accumulate,
                                                       const float* __restrict__ bias, int relu) {
  __shared__ __attribute__((aligned(16))) uint8_t wl[2][DGW_KS * DGW_NT * 1024];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t p0 = (int64_t)blockIdx.x * 256 + w * 64;
  const int ct0 = blockIdx.y * per;                        // the tiles are dealt out evenly over gridDim.y (9 tiles -> 5 + 4, not 8 + 1)
  int nct = CIT - ct0; if (nct > per) nct = per;
  v4f acc[DGW_NT][4];
#pragma unroll
  for (int m = 0; m < DGW_NT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[m][t] = (v4f){0.f, 0.f, 0.f, 0.f};
  // weight stage s: K steps [s*KS, s*KS+KS) of tiles ct0 .. ct0+nct-1 -> wl[buf][(ks * NT + m) * 1024 + lane * 16]; 256 threads x 4 x 16 bytes
  const int nst = (KB + DGW_KS - 1) / DGW_KS;
  uint4 wr[DGW_KS * DGW_NT / 4];
  auto wfetch = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < DGW_KS * DGW_NT / 4; ++q) {
      const int u = tid + 256 * q;                       // 16-byte unit: (ks, m, lane)
      const int ln = u & 63, m = (u >> 6) % DGW_NT, ks = u / (64 * DGW_NT);
      const int kb = st * DGW_KS + ks;
      wr[q] = make_uint4(0, 0, 0, 0);
      if (m < nct && kb < KB) wr[q] = *(const uint4*)(wt + ((((int64_t)(ct0 + m) * KB + kb) * 64 + ln) << 3));
    }
  };
  auto wstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < DGW_KS * DGW_NT / 4; ++q) *(uint4*)(wl[buf] + (size_t)(tid + 256 * q) * 16) = wr[q];
  };
  // dc fragments of K step kb: lane (j, g) -> pixel p0 + 16 t + j, channels kb*32 + 8 g .. + 7
  const int rowe = cout;
  auto bfetch = [&](int kb, v4i* b) __attribute__((always_inline)) {
    const int co = kb * 32 + 8 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t p = p0 + 16 * t + j;
      b[t] = (v4i){0, 0, 0, 0};
      if (p < npix && co < cout && kb < KB) b[t] = *(const v4i*)(dc + p * rowe + co);
    }
  };
  v4i bcur[4], bnxt[4];
  wfetch(0); wstore(0);
  bfetch(0, bcur);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) wfetch(st + 1);
#pragma unroll
    for (int ks = 0; ks < DGW_KS; ++ks) {
      const int kb = st * DGW_KS + ks;
      bfetch(kb + 1, bnxt);                               // one K step ahead (zeros past the end)
      if (kb < KB) {
#pragma unroll
        for (int m = 0; m < DGW_NT; ++m) {
          if (m < nct) {
            const v4i a = *(const v4i*)(wl[buf] + (size_t)((ks * DGW_NT + m) * 64 + lane) * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf16, a), __builtin_bit_cast(v8bf16, bcur[t]), acc[m][t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) bcur[t] = bnxt[t];
    }
    if (st + 1 < nst) wstore(buf ^ 1);
    __syncthreads();
  }
  const float sw = qw ? qw[FROST_Q_SCALE] : 1.0f;
#pragma unroll
  for (int m = 0; m < DGW_NT; ++m) {
    if (m >= nct) continue;
    const int ci = (ct0 + m) * 16 + 4 * g;
    if (ci >= cin) continue;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t p = p0 + 16 * t + j;
      if (p >= npix) continue;
      uint16_t* dst = dx + p * cin + ci;
      float v[4] = {acc[m][t][0] * sw, acc[m][t][1] * sw, acc[m][t][2] * sw, acc[m][t][3] * sw};
      if (bias) { const float4 b4 = *(const float4*)(bias + ci); v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }      // the bf16 inference layers: + b'[co], ReLU
      if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if (accumulate) { const uint2 o = *(const uint2*)dst; v[0] += bf2f(o.x & 0xffff); v[1] += bf2f(o.x >> 16); v[2] += bf2f(o.y & 0xffff); v[3] += bf2f(o.y >> 16); }
      uint2 o; o.x = cvt_pk_bf16(v[0], v[1]); o.y = cvt_pk_bf16(v[2], v[3]);
      *(uint2*)dst = o;
    }
  }
}
// 1 if the wide-layer data-gradient kernel takes this shape (FROST_DGRAD_WIDE = smallest Cout, 0 turns it off)
extern "C" int frost_pw_dgrad_wide_ok(int64_t npix, int cin, int cout) {
  static const int minc = getenv("FROST_DGRAD_WIDE") ? atoi(getenv("FROST_DGRAD_WIDE")) : 1;      // smallest Cout it takes (measured: every non-fused layer gains); 0 = off
  return minc > 0 && cout >= minc && (cout & 7) == 0 && (cin & 7) == 0 && npix >= 256;
}
extern "C" int frost_pw_dgrad_wide(const uint16_t* dc, const uint16_t* wt_pack, const float* qrec_w, int64_t npix, int cin, int cout,
                                   uint16_t* dx, int accumulate, void* stream) {
  FROST_REQUIRE((cin & 7) == 0 && (cout & 7) == 0, "pw_dgrad_wide: channels must be multiples of 8");
  const int cpad = round_up(cout, 16); const int KB = cpad / 32 + ((cpad % 32) ? 1 : 0);
  const int CIT = round_up(cin, 16) / 16;
  const int nch = (CIT + DGW_NT - 1) / DGW_NT, per = (CIT + nch - 1) / nch;
  dim3 grid((unsigned)((npix + 255) / 256), (unsigned)nch);
  hipLaunchKernelGGL(k_dgrad_wide, grid, dim3(256), 0, as_stream(stream), dc, wt_pack, qrec_w, npix, cout, cin, KB, CIT, per, dx, accumulate,
                     (const float*)nullptr, 0);
  return frost_check_launch("pw_dgrad_wide");
}
// the same GEMM as a bf16 inference layer: y[p][n] = act(sum_k x[p][k] * W'[n][k] + b'[n]) (frost_infer_pw routes its long-row layers here)
int frost_gemm_bf16_rows(const uint16_t* x, const uint16_t* pack, const float* bias, int64_t npix, int k, int n, int relu, uint16_t* y, hipStream_t s) {
  const int kp = round_up(k, 16); const int KB = kp / 32 + ((kp % 32) ? 1 : 0);
  const int CIT = round_up(n, 16) / 16;
  const int nch = (CIT + DGW_NT - 1) / DGW_NT, per = (CIT + nch - 1) / nch;
  dim3 grid((unsigned)((npix + 255) / 256), (unsigned)nch);
  hipLaunchKernelGGL(k_dgrad_wide, grid, dim3(256), 0, s, x, pack, (const float*)nullptr, npix, k, n, KB, CIT, per, y, 0, bias, relu);
  return frost_check_launch("gemm_bf16_rows");
}
