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
#ifndef WG_PFD
#define WG_PFD 1      // prefetch depth (128-pixel blocks) of the 64 x 64-tile kernel (deeper: measured no faster -- the loop was LDS-bank bound, not latency bound)
#endif
#ifndef RSD
#define RSD 160   // dc LDS row stride (bytes): 64 bf16 + 32 -- 40 dwords = 8 x odd: the 4 (8) rows a 16-lane group (half wave) reads land on disjoint 8-bank ranges
#endif
#ifndef RSX
#define RSX 72    // x  LDS row stride (bytes): 64 int8 + 8  (8-byte aligned rows for the 8-byte transpose reads)
#endif

__device__ __forceinline__ uint32_t pack_trunc_bf16(float lo, float hi) {   // exact for |integers| <= 256
  return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}

__global__ __launch_bounds__(256, (WG_PFD > 1) ? 2 : 3) void k_pw_wgrad(const uint16_t* __restrict__ dc, const int8_t* __restrict__ x, const float* qx,
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

  // transpose-read source addresses (fixed per lane): this wave's 32 pixels = rows w*32 + ..; lane group g holds K slots {4g .. 4g+3} (first read) and
  // {16+4g .. 16+4g+3} (second read) of the 32 -- any K order serves as long as both operands use it, and with this one a half wave reads 8 CONSECUTIVE rows,
  // which the row strides above spread over all 64 banks (rows 8g .. 8g+3 per group: 4-way conflicts on the dc reads, LDS_BANK_CONFLICT ~ the kernel's duration)
  const int pb = w * 32 + g * 4;
  const uint8_t* a_src = dcs + (pb + (i16 >> 2)) * RSD + (i16 & 3) * 8;          // + a*32 bytes, second read + 16 rows
  const int r8 = i16 >> 1;
  const uint8_t* b_src = xs + (pb + r8 + ((r8 >= 4) ? 12 : 0)) * RSX + (i16 & 1) * 8;           // + b*16 bytes

  int na = (cout - co0 + 15) / 16; if (na > 4) na = 4;     // live 16-channel tiles of this workgroup's 64x64 output tile
  int nb = (cin - ci0 + 15) / 16; if (nb > 4) nb = 4;
  const int64_t nblk = (npix + KPIX - 1) / KPIX;
  // register prefetch, WG_PFD 128-pixel blocks deep (24 VGPRs each): the kernel runs on few workgroups beside the main stream, so what hides the HBM round
  // trip is the bytes each workgroup keeps in flight, not residency (one block ahead: ~4 us per block and workgroup, i.e. one round trip per block)
  uint4 pd[WG_PFD][4]; uint2 px[WG_PFD][4];
#define WG_PREFETCH(S_, BLK_)                                                                                         \
  {                                                                                                                   \
    const int64_t q0_ = (BLK_) * KPIX;                                                                                \
    _Pragma("unroll") for (int jn = 0; jn < 4; ++jn) {                                                                \
      const int u_ = tid + jn * 256; const int pix_ = u_ >> 3, c8_ = u_ & 7; const int64_t gp_ = q0_ + pix_;          \
      pd[S_][jn] = make_uint4(0, 0, 0, 0); px[S_][jn] = make_uint2(zfill, zfill);                                     \
      if (gp_ < npix && (co0 + c8_ * 8) < cout) pd[S_][jn] = *(const uint4*)(dc + gp_ * cout + co0 + c8_ * 8);        \
      if (gp_ < npix && (ci0 + c8_ * 8) < cin) px[S_][jn] = *(const uint2*)(x + gp_ * cin + ci0 + c8_ * 8);           \
    }                                                                                                                 \
  }
#pragma unroll
  for (int s = 0; s < WG_PFD; ++s) { const int64_t b_ = (int64_t)split + (int64_t)s * nsplit; if (b_ < nblk) WG_PREFETCH(s, b_) }
  for (int64_t blk0 = split; blk0 < nblk; blk0 += (int64_t)nsplit * WG_PFD) {
#pragma unroll
   for (int s = 0; s < WG_PFD; ++s) {
    const int64_t blk = blk0 + (int64_t)s * nsplit;
    if (blk >= nblk) break;                               // workgroup-uniform
    __syncthreads();
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u = tid + jn * 256; const int pix = u >> 3, c8 = u & 7;
      *(uint4*)(dcs + pix * RSD + c8 * 16) = pd[s][jn];
      *(uint2*)(xs + pix * RSX + c8 * 8) = px[s][jn];
    }
    __syncthreads();
    if (blk + (int64_t)nsplit * WG_PFD < nblk) WG_PREFETCH(s, blk + (int64_t)nsplit * WG_PFD)
    v4i afr[4], bfr[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {                         // A: dc^T, rows = co; lane i16 gets channel a*16+i16, 8 pixels
      if (a < na) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a_src + a * 32));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a_src + a * 32 + 16 * RSD));
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
    if (co < cout && ci < cin) atomicAdd(dwq_dst_k(dwq, (int64_t)cout * cin, split) + (int64_t)co * cin + ci, red[i] * sx);          // small layers: the splits of a tile spread over the dwq copies (frost_common.h)
  }
}
// Large-channel variant: 128x128 (co x ci) output tile over ALL 128 staged pixels (4 K-steps), 8 waves: wave w owns the
// 64 x 32 block (co half w>>2, ci quarter w&3): 2x the arithmetic intensity per staged byte of the small kernel, no cross-wave
// reduction, and 32 accumulator + 24 prefetch registers per lane, so two 8-wave workgroups are resident per CU (the 4-wave
// version needed 224 VGPRs: 8 waves per CU, latency-bound).
#define BT 128
#define RSD2 288   // dc rows: 128 bf16 + 32 bytes (72 dwords = 8 x 9, see RSD)
// NB = 16-channel input tiles per wave: 2 -> a 128 x 128 (co x ci) workgroup tile, 4 -> 128 x 256.  The x tile lives in LDS as bf16 (q - zp, exact), converted ONCE per
// element while staging -- as int8 every wave converted its own B fragments (the two co-halves twice over): 280 of the loop's ~330 VALU instructions per block, on 1 - 2
// waves per SIMD with no MFMA overlap.  The 256-wide tile reads the dc tensor ONCE for layers with 128 < Cin <= 256 (the 7 x 7 expand convs: dc is 6 x the size of x)
// and halves the re-reads of the narrow dc of the reduce convs (counters: 1.7x the algorithmic bytes with 128 x 128 tiles).
template <int NB>
__global__ __launch_bounds__(512, (NB == 2) ? 2 : 1) void k_pw_wgrad_big(const uint16_t* __restrict__ dc, const int8_t* __restrict__ x, const float* qx,
                                                                           int64_t npix, int cin, int cout, float* __restrict__ dwq, int nsplit, int xmap) {
  constexpr int BTI = NB * 64;                   // input channels of the workgroup tile
  constexpr int RSXB = BTI * 2 + 32;              // x rows: bf16 + 32 bytes (8 x odd dwords, like the dc rows)
  constexpr int XU = BTI / 32;                   // 8-byte x units per thread and block
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [128][RSD2] + [128][RSXB]: 72 KB / 105 KB
  uint8_t* dcs = lds; uint8_t* xs = lds + KPIX * RSD2;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qa = w >> 2, qb = w & 3;                                   // co half, ci quarter of this wave
  const int nci = (cin + BTI - 1) / BTI;
  const int ntile = ((cout + BT - 1) / BT) * nci;
  int tile, split;                   // XCD-aware map, see k_pw_wgrad
  if (xmap) { const int b = blockIdx.x, xcd = b & 7, jb = b >> 3; split = xcd + 8 * (jb / ntile); tile = jb % ntile; }
  else { tile = blockIdx.x % ntile; split = blockIdx.x / ntile; }
  const int co0 = (tile / nci) * BT, ci0 = (tile % nci) * BTI;
  const int zpu = __float_as_int(qx[FROST_Q_ZP]);
  const float zpf = (float)zpu;
  const uint32_t zfill = (uint32_t)((zpu - 128) & 255) * 0x01010101u;
  int na = (cout - co0 - qa * 64 + 15) / 16; na = na < 0 ? 0 : (na > 4 ? 4 : na);
  int nb = (cin - ci0 - qb * (NB * 16) + 15) / 16; nb = nb < 0 ? 0 : (nb > NB ? NB : nb);
  v4f acc[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
  const uint8_t* a_src = dcs + (g * 4 + (i16 >> 2)) * RSD2 + qa * 128 + (i16 & 3) * 8;         // K slot order: see k_pw_wgrad
  const uint8_t* b_src = xs + (g * 4 + (i16 >> 2)) * RSXB + qb * (NB * 32) + (i16 & 3) * 8;     // bf16 like the dc tile: + b*32 bytes, second read + 16 rows
  const int64_t nblk = (npix + KPIX - 1) / KPIX;
  uint4 pd[4]; uint2 px[XU];         // register prefetch of the next block: HBM / L2 latency overlaps the MFMAs (deeper: measured no faster -- the loop is issue-bound)
  auto prefetch = [&](int64_t blk_) __attribute__((always_inline)) {
    const int64_t q0_ = blk_ * KPIX;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u_ = tid + jn * 512; const int pix_ = u_ >> 4, c8_ = u_ & 15; const int64_t gp_ = q0_ + pix_;
      pd[jn] = make_uint4(0, 0, 0, 0);
      if (gp_ < npix && (co0 + c8_ * 8) < cout) pd[jn] = *(const uint4*)(dc + gp_ * cout + co0 + c8_ * 8);
    }
#pragma unroll
    for (int jn = 0; jn < XU; ++jn) {
      const int u_ = tid + jn * 512; const int pix_ = u_ / (BTI / 8), c8_ = u_ % (BTI / 8); const int64_t gp_ = q0_ + pix_;
      px[jn] = make_uint2(zfill, zfill);
      if (gp_ < npix && (ci0 + c8_ * 8) < cin) px[jn] = *(const uint2*)(x + gp_ * cin + ci0 + c8_ * 8);
    }
  };
  if (split < nblk) prefetch((int64_t)split);
  for (int64_t blk = split; blk < nblk; blk += nsplit) {
    __syncthreads();
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u = tid + jn * 512; const int pix = u >> 4, c8 = u & 15;
      *(uint4*)(dcs + pix * RSD2 + c8 * 16) = pd[jn];
    }
#pragma unroll
    for (int jn = 0; jn < XU; ++jn) {
      const int u = tid + jn * 512; const int pix = u / (BTI / 8), c8 = u % (BTI / 8);
      const uint32_t u0 = px[jn].x ^ 0x80808080u, u1 = px[jn].y ^ 0x80808080u;          // offset-binary -> unsigned index
      *(uint4*)(xs + pix * RSXB + c8 * 16) = make_uint4(pack_trunc_bf16((float)(u0 & 255u) - zpf, (float)((u0 >> 8) & 255u) - zpf),
                                                        pack_trunc_bf16((float)((u0 >> 16) & 255u) - zpf, (float)(u0 >> 24) - zpf),
                                                        pack_trunc_bf16((float)(u1 & 255u) - zpf, (float)((u1 >> 8) & 255u) - zpf),
                                                        pack_trunc_bf16((float)((u1 >> 16) & 255u) - zpf, (float)(u1 >> 24) - zpf));
    }
    __syncthreads();
    if (blk + nsplit < nblk) prefetch(blk + nsplit);
    if (na > 0 && nb > 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        v4i afr[4], bfr[NB];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (a < na) {
            const v2i lo = __builtin_bit_cast(v2i, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a_src + ks * 32 * RSD2 + a * 32)));
            const v2i hi = __builtin_bit_cast(v2i, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a_src + ks * 32 * RSD2 + a * 32 + 16 * RSD2)));
            afr[a] = (v4i){lo[0], lo[1], hi[0], hi[1]};
          }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b < nb) {
            const v2i lo = __builtin_bit_cast(v2i, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(b_src + ks * 32 * RSXB + b * 32)));
            const v2i hi = __builtin_bit_cast(v2i, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(b_src + ks * 32 * RSXB + b * 32 + 16 * RSXB)));
            bfr[b] = (v4i){lo[0], lo[1], hi[0], hi[1]};
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if (a < na && b < nb)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[a]), __builtin_bit_cast(v8bf, bfr[b]), acc[a][b], 0, 0, 0);
      }
    }
  }
  const float sx = qx[FROST_Q_SCALE];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (a < na && b < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + qa * 64 + a * 16 + 4 * g + r, ci = ci0 + qb * (NB * 16) + b * 16 + i16;
          if (co < cout && ci < cin) atomicAdd(dwq_dst_k(dwq, (int64_t)cout * cin, split) + (int64_t)co * cin + ci, acc[a][b][r] * sx);
        }
      }
}
static int xcd_round(int nsplit, int64_t nblk, int* xmap);
template <int NB>
static void launch_wgb(const uint16_t* dc, const int8_t* x, const float* qx, int64_t npix, int cin, int cout, float* dwq, int64_t nblk, hipStream_t s) {
  constexpr int BTI = NB * 64;
  const size_t lds = (size_t)KPIX * RSD2 + (size_t)KPIX * (BTI * 2 + 32);
  static bool set = false;
  if (!set) { (void)hipFuncSetAttribute((const void*)k_pw_wgrad_big<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
  const int ntile = ((cout + BT - 1) / BT) * ((cin + BTI - 1) / BTI);
  static int target = getenv("FROST_WG_TARGET") ? atoi(getenv("FROST_WG_TARGET")) : 128;     // the kernel runs beside the main stream and competes with it for bandwidth: step 22.05 / 21.99 / 21.91 / 21.94 / 21.87 / 22.1 / 22.18 ms at 256 / 224 / 192 / 160 / 128 / 96 / 64 (end of round 3, interleaved runs; 512 = full residency: 22.3)
  int nsplit = (target + ntile - 1) / ntile; if (nsplit > nblk) nsplit = (int)nblk; if (nsplit < 1) nsplit = 1;
  int xmap; nsplit = xcd_round(nsplit, nblk, &xmap);
  hipLaunchKernelGGL(k_pw_wgrad_big<NB>, dim3(ntile * nsplit), dim3(512), lds, s, dc, x, qx, npix, cin, cout, dwq, nsplit, xmap);
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
  if (cin > 64 && cout > 64) {      // wide layers: 128 x 128 tiles, or 128 x 256 where that saves a pass over dc / halves the passes over a narrow dc (Cin > 128)
    // OFF by default: alone the wider tile is faster on those layers and reads dc once, but inside the step it is SLOWER (21.22 -> 21.34 ms/step, interleaved): one 512-thread
    // workgroup with 105 KB of LDS and ~190 registers takes a whole CU away from the main stream for as long as it runs, the 128 x 128 workgroups share theirs
    static const int wide = getenv("FROST_WG_CI256") ? atoi(getenv("FROST_WG_CI256")) : 0;
    // only where the wider tile adds no padded channels (Cin = 240, 1440, 1728: measured 75 -> 62, 71 -> 55, 103 -> 106 us; Cin = 288 / 312 / 624 would pad 128 more: 103 -> 140, 41 -> 53, 56 -> 66)
    if (wide && cin > 128 && round_up(cin, 256) == round_up(cin, 128)) launch_wgb<4>(dc, x, qrec_x, npix, cin, cout, dwq, nblk, as_stream(stream));
    else launch_wgb<2>(dc, x, qrec_x, npix, cin, cout, dwq, nblk, as_stream(stream));
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
// GM: 0 = bf16 operands (data gradient / inference layer); 1 = int8 operands (64 k per step), epilogue = the integer conv output c[p][n] = acc - zpx * wsum[n]
// as int32; 2 = the same plus the forward statistics of c (sum, sum of squares, min, max per channel) and the conv finalize in the last workgroup
template <int GM, int NTP = 4>      // NTP: 16-pixel column blocks per wave (4: 256-pixel workgroup tile; 2: 128, for launches that would not fill the chip)
__global__ __launch_bounds__(256, 2) void k_dgrad_wide(const uint16_t* __restrict__ dc, const uint16_t* __restrict__ wt, const float* qw, int64_t npix,
                                                       int cout, int cin, int KB, int CIT, int per, uint16_t* __restrict__ dx, int accumulate,
                                                       const float* __restrict__ bias, int relu, const int32_t* __restrict__ wsum, const float* qx, int32_t* __restrict__ cint,
                                                       uint8_t* __restrict__ stats, FrostFinDesc fin, int xmap, int nchy) {
  constexpr bool I8 = GM != 0;
  __shared__ __attribute__((aligned(16))) uint8_t wl[2][DGW_KS * DGW_NT * 1024 + 1024];          // (+ the slack the staged output tile needs, see the epilogue)
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware map (speed only, 1-D grid): the `nchy` input-channel groups of one pixel tile re-read the same dc rows, so they are dealt to ONE XCD back to back
  // (workgroup b runs on XCD b % 8) and the re-reads hit that XCD's L2; pixel tiles past the tensor (grid padding) find every access predicated off
  int bx = blockIdx.x, by = blockIdx.y;
  if (xmap) { const int b = blockIdx.x, xcd = b & 7, jb = b >> 3; bx = (jb / nchy) * 8 + xcd; by = jb % nchy; }
  const int64_t p0 = (int64_t)bx * (64 * NTP) + w * (16 * NTP);
  const int ct0 = by * per;                        // the tiles are dealt out evenly over the channel groups (9 tiles -> 5 + 4, not 8 + 1)
  int nct = CIT - ct0; if (nct > per) nct = per;
  v4f acc[DGW_NT][NTP];
#pragma unroll
  for (int m = 0; m < DGW_NT; ++m)
#pragma unroll
    for (int t = 0; t < NTP; ++t) acc[m][t] = (v4f){0.f, 0.f, 0.f, 0.f};
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
  const int rowb = I8 ? cout : cout * 2;                   // bytes per operand row (K = cout elements); a K step is 64 bytes of it either way
  auto bfetch = [&](int kb, v4i* b) __attribute__((always_inline)) {
    const int kbyte = kb * 64 + 16 * g;
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      const int64_t p = p0 + 16 * t + j;
      b[t] = (v4i){0, 0, 0, 0};
      if (p < npix && kbyte < rowb && kb < KB) b[t] = *(const v4i*)((const uint8_t*)dc + p * rowb + kbyte);
    }
  };
  v4i bcur[NTP], bnxt[NTP];
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
            for (int t = 0; t < NTP; ++t) {       // I8: the same registers hold the int32 accumulators (bit casts at the MFMA)
              if constexpr (I8) acc[m][t] = __builtin_bit_cast(v4f, __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bcur[t], __builtin_bit_cast(v4i, acc[m][t]), 0, 0, 0));
              else acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf16, a), __builtin_bit_cast(v8bf16, bcur[t]), acc[m][t], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NTP; ++t) bcur[t] = bnxt[t];
    }
    if (st + 1 < nst) wstore(buf ^ 1);
    __syncthreads();
  }
  if constexpr (I8) {
    const int zpx = __float_as_int(qx[FROST_Q_ZP]) - 128;
#pragma unroll
    for (int m = 0; m < DGW_NT; ++m) {
      if (m >= nct) continue;
      const int ci = (ct0 + m) * 16 + 4 * g;
      if (ci >= cin) continue;
      const v4i ws = *(const v4i*)(wsum + ci);
#pragma unroll
      for (int t = 0; t < NTP; ++t) {
        const int64_t p = p0 + 16 * t + j;
        if (p >= npix) continue;
        const v4i a = __builtin_bit_cast(v4i, acc[m][t]);
        const v4i c = (v4i){a[0] - zpx * ws[0], a[1] - zpx * ws[1], a[2] - zpx * ws[2], a[3] - zpx * ws[3]};
        *(v4i*)(cint + p * cin + ci) = c;
        acc[m][t] = __builtin_bit_cast(v4f, c);          // the statistics below read the finished values
      }
    }
    if constexpr (GM == 2) {
      // statistics of c over this workgroup's pixels: exact integer sums (c*c in 64 bits), folded over the 16 pixel lanes of a row group, then LDS
      // (the four waves), then one set of global atomics per workgroup -- the table format of k_pw / conv_finalize_dev
      __syncthreads();                                                     // the weight stages are done with: LDS is reused for the partial table
      long long* l_s1 = (long long*)wl; unsigned long long* l_s2 = (unsigned long long*)(l_s1 + DGW_NT * 16);
      int* l_mn = (int*)(l_s2 + DGW_NT * 16); int* l_mx = l_mn + DGW_NT * 16;
      for (int i = tid; i < DGW_NT * 16; i += 256) { l_s1[i] = 0; l_s2[i] = 0; l_mn[i] = INT32_MAX; l_mx[i] = INT32_MIN; }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < DGW_NT; ++m) {
        if (m >= nct) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          long long a1 = 0, a2 = 0; int mn = INT32_MAX, mx = INT32_MIN;
#pragma unroll
          for (int t = 0; t < NTP; ++t) {
            if (p0 + 16 * t + j < npix) {
              const int v = __builtin_bit_cast(v4i, acc[m][t])[r];
              a1 += v; a2 += (long long)v * v; mn = min(mn, v); mx = max(mx, v);
            }
          }
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {                               // the 16 lanes j of this g
            a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); mn = min(mn, __shfl_xor(mn, o)); mx = max(mx, __shfl_xor(mx, o));
          }
          const int cl = m * 16 + 4 * g + r;
          if (j == 0 && (ct0 + m) * 16 + 4 * g + r < cin && mn <= mx) {
            atomicAdd((unsigned long long*)&l_s1[cl], (unsigned long long)a1); atomicAdd(&l_s2[cl], (unsigned long long)a2);
            atomicMin(&l_mn[cl], mn); atomicMax(&l_mx[cl], mx);
          }
        }
      }
      __syncthreads();
      const int cpadn = CIT * 16;
      long long* g_s1 = (long long*)stats_copy(stats, cpadn); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + cpadn);
      int* g_mn = (int*)(g_s2 + cpadn); int* g_mx = g_mn + cpadn;
      for (int i = tid; i < nct * 16; i += 256) {
        const int c = ct0 * 16 + i;
        if (c < cin && l_mn[i] <= l_mx[i]) {
          atomicAdd((unsigned long long*)&g_s1[c], (unsigned long long)l_s1[i]); atomicAdd(&g_s2[c], l_s2[i]);
          atomicMin(&g_mn[c], l_mn[i]); atomicMax(&g_mx[c], l_mx[i]);
        }
      }
      __shared__ int sflag;
      if (last_block_done2(fin.counter, gridDim.x * gridDim.y, &sflag)) {
        float* sh = (float*)wl;
        __syncthreads();
        conv_finalize_dev(stats, npix, cin, cpadn, qx, fin.qrec_w, fin.wscale, fin.gamma, fin.beta, fin.rmean, fin.rvar, fin.nbt, fin.training, fin.relu, fin.observe, 1,
                          fin.coef, fin.qrec_y, tid, 256, sh);
      }
    }
    return;
  }
  const float sw = qw ? qw[FROST_Q_SCALE] : 1.0f;
  if (!bias && !relu && !accumulate && (cin & 7) == 0) {
    // data gradient of a layer with long output rows (reduce_conv: Cin = 312 .. 1728): the lane's results are 8-byte pieces at a 2*Cin-byte stride; staged
    // through a wave-private LDS tile [32 pixels][128 channels] (pitch 264 B, two pixel blocks at a time) they leave as 256 contiguous bytes per pixel, four
    // pixel rows per store instruction
    constexpr int OP = DGW_NT * 32 + 8;
    static_assert(4 * 32 * OP <= (int)sizeof(wl), "output staging tile does not fit the weight buffers");
    __syncthreads();                                       // every wave is done with the weight stages
    uint8_t* const ot = &wl[0][0] + (size_t)w * (32 * OP);
    const int c8 = lane & 15, rsub = lane >> 4;             // 16-byte piece (8 channels) of the row, row within a group of four
    const int cch = ct0 * 16 + c8 * 8;
#pragma unroll
    for (int t0 = 0; t0 < NTP; t0 += 2) {
#pragma unroll
      for (int m = 0; m < DGW_NT; ++m) {
        if (m >= nct) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          uint2 o; o.x = cvt_pk_bf16(acc[m][t0 + t][0] * sw, acc[m][t0 + t][1] * sw); o.y = cvt_pk_bf16(acc[m][t0 + t][2] * sw, acc[m][t0 + t][3] * sw);
          *(uint2*)(ot + (16 * t + j) * OP + (m * 16 + 4 * g) * 2) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private tile: no barrier, the wave's own writes have landed
      if (c8 < nct * 2 && cch < cin) {
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 4) {
          const int64_t p = p0 + 16 * t0 + r0 + rsub;
          if (p < npix) *(uint4*)(dx + p * cin + cch) = *(const uint4*)(ot + (r0 + rsub) * OP + c8 * 16);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the tile is read out before the next pair of pixel blocks overwrites it
    }
    return;
  }
  // accumulate (the gradient slot already holds a gradient: squeeze_conv behind the skip connection): every old value is requested BEFORE the first one is
  // used -- loaded where it is consumed, each load sits in its own block with its wait behind it (up to 16 memory latencies in a row per tile, seen in the ISA)
  uint2 prev[DGW_NT][NTP];
#pragma unroll
  for (int m = 0; m < DGW_NT; ++m)
#pragma unroll
    for (int t = 0; t < NTP; ++t) prev[m][t] = make_uint2(0u, 0u);
  if (accumulate) {
#pragma unroll
    for (int m = 0; m < DGW_NT; ++m) {
      const int ci = (ct0 + m) * 16 + 4 * g;
#pragma unroll
      for (int t = 0; t < NTP; ++t) {
        const int64_t p = p0 + 16 * t + j;
        const bool ok = m < nct && ci < cin && p < npix;
        const uint2 o = *(const uint2*)(dx + (ok ? p * cin + ci : 0));          // unconditional from a clamped address
        prev[m][t] = ok ? o : make_uint2(0u, 0u);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < DGW_NT; ++m) {
    if (m >= nct) continue;
    const int ci = (ct0 + m) * 16 + 4 * g;
    if (ci >= cin) continue;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b4 = *(const float4*)(bias + ci);                                 // the bf16 inference layers: + b'[co], ReLU
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      const int64_t p = p0 + 16 * t + j;
      if (p >= npix) continue;
      uint16_t* dst = dx + p * cin + ci;
      float v[4] = {acc[m][t][0] * sw, acc[m][t][1] * sw, acc[m][t][2] * sw, acc[m][t][3] * sw};
      if (bias) { v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
      if (relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      else if (relu == 2) { v[0] = hswish_f(v[0]); v[1] = hswish_f(v[1]); v[2] = hswish_f(v[2]); v[3] = hswish_f(v[3]); }
      if (accumulate) { const uint2 o = prev[m][t]; v[0] += bf2f(o.x & 0xffff); v[1] += bf2f(o.x >> 16); v[2] += bf2f(o.y & 0xffff); v[3] += bf2f(o.y >> 16); }
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
// 128-pixel tiles when 256-pixel ones would leave CUs without a workgroup (the 7x7 layers at B <= 512)
template <int GM, typename... Args>
static void launch_dgw(hipStream_t s, int64_t npix, int nch, Args... args) {
  static const int xon = getenv("FROST_DGW_XCD") ? atoi(getenv("FROST_DGW_XCD")) : 1;
  const bool small = ((npix + 255) / 256) * nch < 320;
  const int64_t nx = small ? (npix + 127) / 128 : (npix + 255) / 256;
  const bool xm = xon && nch > 1;
  const dim3 grid = xm ? dim3((unsigned)(((nx + 7) / 8) * 8 * nch)) : dim3((unsigned)nx, (unsigned)nch);
  if (small) hipLaunchKernelGGL((k_dgrad_wide<GM, 2>), grid, dim3(256), 0, s, args..., xm ? 1 : 0, nch);
  else hipLaunchKernelGGL((k_dgrad_wide<GM, 4>), grid, dim3(256), 0, s, args..., xm ? 1 : 0, nch);
}
extern "C" int frost_pw_dgrad_wide(const uint16_t* dc, const uint16_t* wt_pack, const float* qrec_w, int64_t npix, int cin, int cout,
                                   uint16_t* dx, int accumulate, void* stream) {
  FROST_REQUIRE((cin & 7) == 0 && (cout & 7) == 0, "pw_dgrad_wide: channels must be multiples of 8");
  const int cpad = round_up(cout, 16); const int KB = cpad / 32 + ((cpad % 32) ? 1 : 0);
  const int CIT = round_up(cin, 16) / 16;
  const int nch = (CIT + DGW_NT - 1) / DGW_NT, per = (CIT + nch - 1) / nch;
  launch_dgw<0>(as_stream(stream), npix, nch, dc, wt_pack, qrec_w, npix, cout, cin, KB, CIT, per, dx, accumulate,
                     (const float*)nullptr, 0, (const int32_t*)nullptr, (const float*)nullptr, (int32_t*)nullptr, (uint8_t*)nullptr, FrostFinDesc{});
  return frost_check_launch("pw_dgrad_wide");
}
// the same GEMM as a bf16 inference layer: y[p][n] = act(sum_k x[p][k] * W'[n][k] + b'[n]) (frost_infer_pw routes its long-row layers here)
int frost_gemm_bf16_rows(const uint16_t* x, const uint16_t* pack, const float* bias, int64_t npix, int k, int n, int relu, uint16_t* y, hipStream_t s) {
  const int kp = round_up(k, 16); const int KB = kp / 32 + ((kp % 32) ? 1 : 0);
  const int CIT = round_up(n, 16) / 16;
  const int nch = (CIT + DGW_NT - 1) / DGW_NT, per = (CIT + nch - 1) / nch;
  launch_dgw<0>(s, npix, nch, x, pack, (const float*)nullptr, npix, k, n, KB, CIT, per, y, 0, bias, relu, (const int32_t*)nullptr,
                     (const float*)nullptr, (int32_t*)nullptr, (uint8_t*)nullptr, FrostFinDesc{});
  return frost_check_launch("gemm_bf16_rows");
}

// ---- the integer conv output of a pointwise layer, kept for element-wise passes: c[p][co] = sum_k xq[p][k] * wq[co][k] - (zp_x - 128) * wsum[co]
// (x holds q - 128; wq_pack / wsum as frost_weight_prep lays them out).  The GEMM above on int8 operands: M = pixels, N = Cout, K = Cin.
extern "C" int frost_pw_conv_int(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout,
                                 int32_t* conv_out, void* stream) {
  FROST_REQUIRE((cin & 7) == 0 && (cout & 3) == 0, "pw_conv_int: cin must be a multiple of 8, cout of 4");
  const int KS = round_up(cin, 64) / 64;
  const int CT = round_up(cout, 16) / 16;
  const int nch = (CT + DGW_NT - 1) / DGW_NT, per = (CT + nch - 1) / nch;
  // kernel arguments in its own (data-gradient) naming: K = "cout" = cin here, N = "cin" = cout here
  launch_dgw<1>(as_stream(stream), npix, nch, (const uint16_t*)x, (const uint16_t*)wq_pack, (const float*)nullptr, npix, cin, cout, KS, CT, per,
                     (uint16_t*)nullptr, 0, (const float*)nullptr, 0, wsum, qrec_x, conv_out, (uint8_t*)nullptr, FrostFinDesc{});
  return frost_check_launch("pw_conv_int");
}
// the forward statistics pass of such a layer on the same kernel: stores the integer conv output, accumulates the statistics table and runs the conv
// finalize in its last workgroup (what frost_pw_conv_fwd_fin does with k_pw); frost_pw_ew mode 2 then emits y from conv_out
extern "C" int frost_pw_conv_fwd_keep(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout,
                                      void* stats, const FrostFinDesc* fin, int32_t* conv_out, void* stream) {
  FROST_REQUIRE((cin & 7) == 0 && (cout & 3) == 0, "pw_conv_fwd_keep: cin must be a multiple of 8, cout of 4");
  FROST_REQUIRE(fin && fin->counter && fin->coef && fin->qrec_y && conv_out && stats, "pw_conv_fwd_keep: incomplete arguments");
  const int KS = round_up(cin, 64) / 64;
  const int CT = round_up(cout, 16) / 16;
  const int nch = (CT + DGW_NT - 1) / DGW_NT, per = (CT + nch - 1) / nch;
  launch_dgw<2>(as_stream(stream), npix, nch, (const uint16_t*)x, (const uint16_t*)wq_pack, (const float*)nullptr, npix, cin, cout, KS, CT, per,
                     (uint16_t*)nullptr, 0, (const float*)nullptr, 0, wsum, qrec_x, conv_out, (uint8_t*)stats, *fin);
  return frost_check_launch("pw_conv_fwd_keep");
}

// ---- element-wise backward passes over a kept integer conv output (frost_pw_conv_int): the epilogues of k_pw's reduce / dc passes, same formulas
//   t = fma(A, c, B) / s_y;  gy = g if t_lo < t <= t_hi else 0 (the STE window of the activation fake-quantise, ReLU included)
//   mode 0: S1 += gy, S2 += gy * xhat (xhat = fma(c, R, -M*R)) -> coefficient rows S1 / S2        mode 1: dc = fma(gy, K1, fma(c, E, F)) -> bf16
//   mode 2 (forward): y = clamp(rint(fma(A, c, B) / s_y) + zp_y, 0, qmax) -> int8 (q - 128), the emit epilogue of k_pw
// thread = 4 channels (16 bytes of c) x strided pixels: coefficients and partial sums stay in registers.
template <int MODE>
__global__ __launch_bounds__(256) void k_pw_ew(const int32_t* __restrict__ cint, int64_t npix, int cout, int cpad, float* __restrict__ coef, const float* qy,
                                               int relu, float inv_count, const uint16_t* __restrict__ gout, uint16_t* __restrict__ dc, int sr, int8_t* __restrict__ yq) {
  extern __shared__ float part[];                   // mode 0: [2][cout]
  const int tid = threadIdx.x;
  if (MODE == 0) { for (int i = tid; i < 2 * cout; i += 256) part[i] = 0.0f; __syncthreads(); }
  const int c4n = cout >> 2;
  const int64_t PP = ((int64_t)gridDim.x * 256) / c4n;
  const int64_t tt = (int64_t)blockIdx.x * 256 + tid;
  const int c4 = (int)(tt % c4n); const int64_t slot = tt / c4n;
  const int ch = c4 * 4;
  const float y_inv = 1.0f / qy[FROST_Q_SCALE];
  const int zpy = __float_as_int(qy[FROST_Q_ZP]), qhi = q_hi(qy);
  const float hi0 = (float)qhi + 0.5f - (float)zpy;                             // rint(hi0) ties to the even neighbour (as in k_pw)
  const float t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
  float t_lo = 0.0f;
  if (!relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  float A[4], B[4], R[4], MR[4], K1[4], E[4], F[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    A[r] = coef[FROST_COEF_A * cpad + ch + r]; B[r] = coef[FROST_COEF_B * cpad + ch + r];
    const float Mv = coef[FROST_COEF_M * cpad + ch + r]; R[r] = coef[FROST_COEF_R * cpad + ch + r]; MR[r] = -Mv * R[r];
    K1[r] = 0.f; E[r] = 0.f; F[r] = 0.f;
    if (MODE == 1) {
      K1[r] = coef[FROST_COEF_K1 * cpad + ch + r];
      const float s1 = s12_sum(coef, cpad, 0, ch + r), s2 = s12_sum(coef, cpad, 1, ch + r);
      E[r] = -K1[r] * (s2 * inv_count) * R[r]; F[r] = -K1[r] * (s1 * inv_count) - E[r] * Mv;
    }
  }
  float r1[4] = {0.f, 0.f, 0.f, 0.f}, r2[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t rng = sr_seed(blockIdx.x, threadIdx.x);
  if constexpr (MODE == 2) {
    const float y_zpf = (float)zpy, qcap = (float)qhi; const bool lowq = qcap < 255.0f;
    if (slot < PP) {
      for (int64_t p0 = slot; p0 < npix; p0 += 4 * PP) {
        v4i cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t p = p0 + u * PP; cv[u] = (v4i){0, 0, 0, 0}; if (p < npix) cv[u] = *(const v4i*)(cint + p * cout + ch); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t p = p0 + u * PP;
          if (p >= npix) continue;
          uint32_t packed = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float yv = fmaf(A[r], (float)cv[u][r], B[r]);
            float qv = rintf(yv * y_inv) + y_zpf;
            if (lowq) qv = fminf(qv, qcap);
            packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);       // saturates at 0 (ReLU layers have zp == 0) and 255
          }
          *(uint32_t*)(yq + p * cout + ch) = packed ^ 0x80808080u;
        }
      }
    }
    return;
  }
  if (slot < PP) {
    for (int64_t p0 = slot; p0 < npix; p0 += 2 * PP) {          // two pixels per trip: their loads are in flight together
      v4i cv[2]; uint2 gv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t p = p0 + u * PP;
        cv[u] = (v4i){0, 0, 0, 0}; gv[u] = make_uint2(0, 0);
        if (p < npix) { cv[u] = *(const v4i*)(cint + p * cout + ch); gv[u] = *(const uint2*)(gout + p * cout + ch); }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t p = p0 + u * PP;
        if (p >= npix) continue;
        const float gq[4] = {__uint_as_float(gv[u].x << 16), __uint_as_float(gv[u].x & 0xffff0000u), __uint_as_float(gv[u].y << 16), __uint_as_float(gv[u].y & 0xffff0000u)};
        float dcv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float af = (float)cv[u][r];
          const float tq = fmaf(A[r], af, B[r]) * y_inv;
          const float gy = (tq > t_lo && tq <= t_hi) ? gq[r] : 0.0f;
          if (MODE == 0) { r1[r] += gy; r2[r] = fmaf(gy, fmaf(af, R[r], MR[r]), r2[r]); }
          else dcv[r] = fmaf(gy, K1[r], fmaf(af, E[r], F[r]));
        }
        if (MODE == 1) {
          uint2 o;
          if (sr) { o.x = sr_pk_bf16(dcv[0], dcv[1], rng); o.y = sr_pk_bf16(dcv[2], dcv[3], rng); }
          else { o.x = cvt_pk_bf16(dcv[0], dcv[1]); o.y = cvt_pk_bf16(dcv[2], dcv[3]); }
          *(uint2*)(dc + p * cout + ch) = o;
        }
      }
    }
  }
  if (MODE == 0) {
    if (slot < PP) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { atomicAdd(&part[ch + r], r1[r]); atomicAdd(&part[cout + ch + r], r2[r]); }
    }
    __syncthreads();
    for (int i = tid; i < cout; i += 256) {
      atomicAdd(s12_dst(coef, cpad, 0) + i, part[i]);
      atomicAdd(s12_dst(coef, cpad, 1) + i, part[cout + i]);
    }
  }
}
extern "C" int frost_pw_ew(const int32_t* conv_out, int64_t npix, int cout, float* coef, const float* qrec_y, int relu, int mode, const uint16_t* gout,
                           void* out, void* stream) {
  uint16_t* dc = (uint16_t*)out;
  FROST_REQUIRE((cout & 3) == 0, "pw_ew: cout must be a multiple of 4");
  FROST_REQUIRE(mode == 0 || ((mode == 1 || mode == 2) && out), "pw_ew: mode 0 (reduce), 1 (dc -> out) or 2 (emit -> out)");
  const int cpad = round_up(cout, 16); const int c4n = cout >> 2;
  const int64_t tot = npix * c4n;
  static int rcap = -1;
  if (rcap < 0) { const char* e = getenv("FROST_PWEW_CAP"); rcap = e ? atoi(e) : 256; if (rcap < 64) rcap = 64; }      // every workgroup ends with 2*cout float atomics into the coefficient rows: 256 measured best (512: +5 us, 2048: +40 us per launch)
  int64_t grid = (tot + 255) / 256; const int64_t cap = mode == 0 ? rcap : 2048; if (grid > cap) grid = cap;
  const int64_t gmin = (c4n + 255) / 256; if (grid < gmin) grid = gmin;
  const float inv_count = 1.0f / (float)npix;
  if (mode == 0) hipLaunchKernelGGL(k_pw_ew<0>, dim3((unsigned)grid), dim3(256), (size_t)2 * cout * 4, as_stream(stream), conv_out, npix, cout, cpad, coef, qrec_y, relu, inv_count, gout, dc, 0, (int8_t*)nullptr);
  else if (mode == 1) hipLaunchKernelGGL(k_pw_ew<1>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), conv_out, npix, cout, cpad, coef, qrec_y, relu, inv_count, gout, dc, frost_sr_enabled(), (int8_t*)nullptr);
  else hipLaunchKernelGGL(k_pw_ew<2>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), conv_out, npix, cout, cpad, coef, qrec_y, relu, inv_count, gout, (uint16_t*)nullptr, 0, (int8_t*)out);
  return frost_check_launch("pw_ew");
}


// ---- block-boundary fusion, backward: skip_add's backward (frostnet.py:142; k_add_bwd8) folded into the element-wise reduce / dc passes of the reduce_conv that
// produced the add's second operand.  g = gsum if the add's FakeQuantize passed the element (STE window of q_sum on fake(a) + fake(yb)) else 0 -- the expression of
// k_add_bwd8, term for term; the residual branch's gradient ga (+)= g leaves in the reduce pass (mode 0); the reduce_conv's own gout (= g) is never written: both
// passes re-derive it from gsum / a / yb (4 bytes per element instead of writing and twice re-reading a 2-byte tensor, and one launch less per residual block).
// Bit-identical to frost_add_bwd + frost_pw_ew (g is a bf16 value passed through or zeroed).  Thread = 4 channels x strided pixels, as k_pw_ew.
__device__ __forceinline__ void ew_acc_store4(uint16_t* dst, const float* v, int accumulate) {      // = acc_store4 of frost_head.hip (k_add_bwd8's store)
  float o[4] = {v[0], v[1], v[2], v[3]};
  if (accumulate) { uint2 t = *(const uint2*)dst; o[0] += bf2f(t.x & 0xffff); o[1] += bf2f(t.x >> 16); o[2] += bf2f(t.y & 0xffff); o[3] += bf2f(t.y >> 16); }
  uint2 w; w.x = cvt_pk_bf16(o[0], o[1]); w.y = cvt_pk_bf16(o[2], o[3]);
  *(uint2*)dst = w;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_pw_ew_add(const int32_t* __restrict__ cint, int64_t npix, int cout, int cpad, float* __restrict__ coef, const float* qy,
                                                   int relu, float inv_count, const uint16_t* __restrict__ gsum, const int8_t* __restrict__ a, const float* qa,
                                                   const int8_t* __restrict__ yb, const float* qsum, uint16_t* __restrict__ ga, int acc_a, uint16_t* __restrict__ dc, int sr) {
  extern __shared__ float part[];                   // mode 0: [2][cout]
  const int tid = threadIdx.x;
  if (MODE == 0) { for (int i = tid; i < 2 * cout; i += 256) part[i] = 0.0f; __syncthreads(); }
  const int c4n = cout >> 2;
  const int64_t PP = ((int64_t)gridDim.x * 256) / c4n;
  const int64_t tt = (int64_t)blockIdx.x * 256 + tid;
  const int c4 = (int)(tt % c4n); const int64_t slot = tt / c4n;
  const int ch = c4 * 4;
  const QP QA = load_qp(qa), QB = load_qp(qy), QS = load_qp(qsum);
  const float y_inv = 1.0f / qy[FROST_Q_SCALE];
  const int zpy = __float_as_int(qy[FROST_Q_ZP]), qhi = q_hi(qy);
  const float hi0 = (float)qhi + 0.5f - (float)zpy;
  const float t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
  float t_lo = 0.0f;
  if (!relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  float A[4], B[4], R[4], MR[4], K1[4], E[4], F[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    A[r] = coef[FROST_COEF_A * cpad + ch + r]; B[r] = coef[FROST_COEF_B * cpad + ch + r];
    const float Mv = coef[FROST_COEF_M * cpad + ch + r]; R[r] = coef[FROST_COEF_R * cpad + ch + r]; MR[r] = -Mv * R[r];
    K1[r] = 0.f; E[r] = 0.f; F[r] = 0.f;
    if (MODE == 1) {
      K1[r] = coef[FROST_COEF_K1 * cpad + ch + r];
      const float s1 = s12_sum(coef, cpad, 0, ch + r), s2 = s12_sum(coef, cpad, 1, ch + r);
      E[r] = -K1[r] * (s2 * inv_count) * R[r]; F[r] = -K1[r] * (s1 * inv_count) - E[r] * Mv;
    }
  }
  float r1[4] = {0.f, 0.f, 0.f, 0.f}, r2[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t rng = sr_seed(blockIdx.x, threadIdx.x);
  if (slot < PP) {
    for (int64_t p0 = slot; p0 < npix; p0 += 2 * PP) {          // two pixels per trip: their loads are in flight together
      v4i cv[2]; uint2 gv[2]; uint32_t av[2], bv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t p = p0 + u * PP;
        cv[u] = (v4i){0, 0, 0, 0}; gv[u] = make_uint2(0, 0); av[u] = 0; bv[u] = 0;
        if (p < npix) {
          cv[u] = *(const v4i*)(cint + p * cout + ch); gv[u] = *(const uint2*)(gsum + p * cout + ch);
          av[u] = *(const uint32_t*)(a + p * cout + ch); bv[u] = *(const uint32_t*)(yb + p * cout + ch);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t p = p0 + u * PP;
        if (p >= npix) continue;
        float gq[4] = {__uint_as_float(gv[u].x << 16), __uint_as_float(gv[u].x & 0xffff0000u), __uint_as_float(gv[u].y << 16), __uint_as_float(gv[u].y & 0xffff0000u)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {          // skip_add's FakeQuantize: the sum of the two fake-quantised operands, window test of q_sum (k_add_bwd8)
          const float v = (float)((int)(int8_t)(av[u] >> (8 * r)) + 128 - QA.zp) * QA.scale + (float)((int)(int8_t)(bv[u] >> (8 * r)) + 128 - QB.zp) * QB.scale;
          bool inr; fq_index(v, QS.inv, QS.zp, 0, QS.hi, &inr);
          if (!inr) gq[r] = 0.0f;
        }
        if (MODE == 0) ew_acc_store4(ga + p * cout + ch, gq, acc_a);
        float dcv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float af = (float)cv[u][r];
          const float tq = fmaf(A[r], af, B[r]) * y_inv;
          const float gy = (tq > t_lo && tq <= t_hi) ? gq[r] : 0.0f;
          if (MODE == 0) { r1[r] += gy; r2[r] = fmaf(gy, fmaf(af, R[r], MR[r]), r2[r]); }
          else dcv[r] = fmaf(gy, K1[r], fmaf(af, E[r], F[r]));
        }
        if (MODE == 1) {
          uint2 o;
          if (sr) { o.x = sr_pk_bf16(dcv[0], dcv[1], rng); o.y = sr_pk_bf16(dcv[2], dcv[3], rng); }
          else { o.x = cvt_pk_bf16(dcv[0], dcv[1]); o.y = cvt_pk_bf16(dcv[2], dcv[3]); }
          *(uint2*)(dc + p * cout + ch) = o;
        }
      }
    }
  }
  if (MODE == 0) {
    if (slot < PP) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { atomicAdd(&part[ch + r], r1[r]); atomicAdd(&part[cout + ch + r], r2[r]); }
    }
    __syncthreads();
    for (int i = tid; i < cout; i += 256) {
      atomicAdd(s12_dst(coef, cpad, 0) + i, part[i]);
      atomicAdd(s12_dst(coef, cpad, 1) + i, part[cout + i]);
    }
  }
}
extern "C" int frost_pw_ew_add_bwd(const int32_t* conv_out, int64_t npix, int cout, float* coef, const float* qrec_y, int relu, int mode, const uint16_t* gsum,
                                   const int8_t* a, const float* qrec_a, const int8_t* yb, const float* qrec_sum, uint16_t* ga, int acc_a, uint16_t* dc, void* stream) {
  FROST_REQUIRE((cout & 3) == 0, "pw_ew_add_bwd: cout must be a multiple of 4");
  FROST_REQUIRE(conv_out && gsum && a && yb && qrec_a && qrec_sum && ((mode == 0 && ga) || (mode == 1 && dc)), "pw_ew_add_bwd: mode 0 (reduce, writes ga) or 1 (dc -> out)");
  const int cpad = round_up(cout, 16); const int c4n = cout >> 2;
  const int64_t tot = npix * c4n;
  static int rcap = -1;
  if (rcap < 0) { const char* e = getenv("FROST_PWEW_CAP"); rcap = e ? atoi(e) : 256; if (rcap < 64) rcap = 64; }
  int64_t grid = (tot + 255) / 256; const int64_t cap = mode == 0 ? rcap : 2048; if (grid > cap) grid = cap;
  const int64_t gmin = (c4n + 255) / 256; if (grid < gmin) grid = gmin;
  const float inv_count = 1.0f / (float)npix;
  if (mode == 0) hipLaunchKernelGGL(k_pw_ew_add<0>, dim3((unsigned)grid), dim3(256), (size_t)2 * cout * 4, as_stream(stream), conv_out, npix, cout, cpad, coef, qrec_y, relu, inv_count,
                                    gsum, a, qrec_a, yb, qrec_sum, ga, acc_a, (uint16_t*)nullptr, 0);
  else hipLaunchKernelGGL(k_pw_ew_add<1>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), conv_out, npix, cout, cpad, coef, qrec_y, relu, inv_count,
                          gsum, a, qrec_a, yb, qrec_sum, (uint16_t*)nullptr, 0, dc, frost_sr_enabled());
  return frost_check_launch("pw_ew_add_bwd");
}

// ---- block-boundary fusion (first piece): the emit pass of a kept-conv-output reduce layer TOGETHER with the range pass of the residual add
// that consumes it (frostnet.py:138-142: out = reduce_conv(out); out = skip_add.add(x, out)).  One sweep over the integer conv output writes the layer's
// int8 output y AND accumulates min / max of (x + y) in the reference's fp32 arithmetic (k_add_minmax's expression, same operand order); the last
// workgroup runs the MovingAverageMinMax update of the add's FakeQuantize.  Replaces frost_pw_ew(mode 2) + frost_add_minmax_observe: one launch and one
// read of y fewer per residual block.  state3 = {lo, hi, ticket}: (+inf, -inf, 0) on entry and again on exit.
__global__ __launch_bounds__(256) void k_pw_ew_emit_add(const int32_t* __restrict__ cint, int64_t npix, int cout, int cpad, const float* __restrict__ coef, const float* qy, int relu,
                                                        const int8_t* __restrict__ a, const float* qa, int8_t* __restrict__ yq, float* state3, float* qsum, int observe) {
  const int tid = threadIdx.x;
  const int c4n = cout >> 2;
  const int64_t PP = ((int64_t)gridDim.x * 256) / c4n;
  const int64_t tt = (int64_t)blockIdx.x * 256 + tid;
  const int c4 = (int)(tt % c4n); const int64_t slot = tt / c4n;
  const int ch = c4 * 4;
  const QP A = load_qp(qa), Y = load_qp(qy);
  const float y_inv = 1.0f / qy[FROST_Q_SCALE];
  const float y_zpf = (float)Y.zp, qcap = (float)q_hi(qy); const bool lowq = qcap < 255.0f;
  float cA[4], cB[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { cA[r] = coef[FROST_COEF_A * cpad + ch + r]; cB[r] = coef[FROST_COEF_B * cpad + ch + r]; }
  float lo = INFINITY, hi = -INFINITY;
  if (slot < PP) {
    for (int64_t p0 = slot; p0 < npix; p0 += 4 * PP) {
      v4i cv[4]; uint32_t av[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + u * PP; cv[u] = (v4i){0, 0, 0, 0}; av[u] = 0;
        if (p < npix) { cv[u] = *(const v4i*)(cint + p * cout + ch); av[u] = *(const uint32_t*)(a + p * cout + ch); }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + u * PP;
        if (p >= npix) continue;
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yv = fmaf(cA[r], (float)cv[u][r], cB[r]);
          float qv = rintf(yv * y_inv) + y_zpf;
          if (lowq) qv = fminf(qv, qcap);
          packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);       // saturates at 0 and 255: the index the emit pass stores
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = (int)((packed >> (8 * r)) & 255u);
          const float v = (float)((int)(int8_t)(av[u] >> (8 * r)) + 128 - A.zp) * A.scale + (float)(qi - Y.zp) * Y.scale;
          lo = fminf(lo, v); hi = fmaxf(hi, v);
        }
        *(uint32_t*)(yq + p * cout + ch) = packed ^ 0x80808080u;
      }
    }
  }
  __shared__ int sflag; __shared__ float shr[8];
  if (range_fold_last(lo, hi, state3 + 2 + FROST_TICKET_WORDS, (uint32_t*)(state3 + 2), shr, &sflag)) observer_update_dev(qsum, lo, hi, 0, 0, observe);
}
extern "C" int frost_pw_ew_emit_add(const int32_t* conv_out, int64_t npix, int cout, const float* coef, const float* qrec_y, int relu, const int8_t* a,
                                    const float* qrec_a, int8_t* y, float* state3, float* qrec_sum, int observe, void* stream) {
  FROST_REQUIRE((cout & 3) == 0 && conv_out && a && y && state3 && qrec_sum, "pw_ew_emit_add: cout must be a multiple of 4, all buffers given");
  const int cpad = round_up(cout, 16); const int c4n = cout >> 2;
  const int64_t tot = npix * c4n;
  int64_t grid = (tot + 1023) / 1024; if (grid > FROST_MM_SLOTS) grid = FROST_MM_SLOTS;           // one range slot per workgroup (range_fold_last)
  const int64_t gmin = (c4n + 255) / 256; if (grid < gmin) grid = gmin;
  hipLaunchKernelGGL(k_pw_ew_emit_add, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), conv_out, npix, cout, cpad, coef, qrec_y, relu, a, qrec_a, y, state3,
                     qrec_sum, observe);
  return frost_check_launch("pw_ew_emit_add");
}
