// Whole-bottleneck fused bf16 INFERENCE kernel for the HIGH-RESOLUTION bottlenecks without a squeeze conv (FrostNet-Large layer1.0 .. layer2.0:
// /root/reference/frostnet.py:81-122, eval mode, BatchNorm folded as in Classification/evaluate.py:131-143):
//   [conv1 1x1 expand + ReLU ->] conv2 depthwise k x k (stride 1 / 2) + ReLU -> reduce_conv 1x1 (linear) [-> + x]
// The same arithmetic, rounding points and summation order as frost_infer_block (frost_iblock.hip) with 32-channel chunks -- logits bit-identical to it and to the
// layer-by-layer launches -- but a different work decomposition: one WAVE owns one small output tile (4 x 4 or 4 x 8 pixels) and walks its chunks alone.
//
// Why: on these maps (112 x 112 ... 28 x 28, Cin = 16 ... 32) frost_infer_block spends its time in workgroup barriers: three barrier-separated phases per chunk,
// each a few hundred cycles of work per wave, on 1 - 3 workgroups per CU (measured: ~20 k cycles per 32-pixel tile and workgroup for ~1 k cycles of issue per
// wave, profiles/r04_infer_blocks.txt).  A wave that owns its tile needs NO barrier -- its LDS operations execute in program order -- so the only waits are its own
// LDS / L2 round trips, hidden by the other 15 - 20 resident waves of the CU (8 KB of LDS and < 96 registers per wave).  conv1's B operand (K = Cin <= 32: one
// MFMA K step) is read from HBM straight into the fragment layout (lane = pixel, 8 consecutive channels = 16 bytes) once per tile and kept in registers for all
// chunks: no staging copy of the input at all.
#include "frost_common.h"
#include <stdlib.h>

typedef __bf16 v8bf_w __attribute__((ext_vector_type(8)));

struct IBwP {
  const uint16_t* x; uint16_t* y;
  const uint16_t* w1; const float* b1;          // conv1 A-fragment pack [ct][kb = 1][64][8] + folded bias (NULL: the depthwise conv reads the block input)
  const float* wdw; const float* bdw;           // conv2: fp32 taps [k*k][cpad_dw], folded bias
  const uint16_t* w3; const float* b3;          // reduce_conv pack [ct3][kb3][64][8], folded bias
  int n, h, w, ho, wo, cin, cexp, cout;
  int ct1, kb3, ct3, cpad_dw, nchunk, tiles_x, tiles_y, residual;
  long long ntiles;
};

#define IBW_PLS 36      // plane row stride (bf16 elements): 32 channels + 4 -> 72 bytes, the 16 pixel rows of an MFMA tile on distinct 8-byte bank slots
#define IBW_Y2S 40      // y2 row stride: 32 + 8 -> 80 bytes (16-byte aligned rows for the B fragment reads)

// a wave's own ordering point: its LDS writes are complete (and the compiler keeps the phases apart)
__device__ __forceinline__ void ibw_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

template <int K, int S, int TH, int TW, bool HASC1>
__global__ __launch_bounds__(256, 4) void k_iblock_w(const IBwP p) {
  constexpr int PAD = (K - 1) / 2, RH = (TH - 1) * S + K, RW = (TW - 1) * S + K, RP = RH * RW, RPT = (RP + 15) / 16;
  constexpr int TP = TH * TW, TPT = TP / 16, SPAN = 3 * S + K, UPR = TW / 4, UNITS = TH * UPR, CT3M = 3;
  static_assert(TP % 16 == 0 && TW % 4 == 0, "tile = whole MFMA pixel tiles, output columns in groups of 4");
  constexpr int WAVE_LDS = (RPT * 16 * IBW_PLS + TPT * 16 * IBW_Y2S) * 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  uint16_t* const pl = (uint16_t*)(smem + (size_t)wv * WAVE_LDS);            // [RPT*16][PLS]: conv1's output (or the block input) of the current chunk over the region
  uint16_t* const y2 = pl + RPT * 16 * IBW_PLS;                              // [TP][Y2S]: conv2's output of the current chunk = the reduce GEMM's B operand
  const int cp = lane & 15, pg = lane >> 4;                                   // depthwise: channel pair of the chunk, output unit group

  for (long long tile = (long long)blockIdx.x * 4 + wv; tile < p.ntiles; tile += (long long)gridDim.x * 4) {
    long long b = tile;
    const int tx = (int)(b % p.tiles_x); b /= p.tiles_x; const int ty = (int)(b % p.tiles_y); const int img = (int)(b / p.tiles_y);
    const int iy0 = ty * TH * S - PAD, ix0 = tx * TW * S - PAD;              // image coordinates of region pixel (0, 0)
    const uint16_t* ximg = p.x + (long long)img * p.h * p.w * p.cin;
    // region pixel of this lane in MFMA pixel tile pt: rp = pt * 16 + j; in-image mask (conv2 zero-pads conv1's OUTPUT: the plane is zero outside the image)
    unsigned vmask = 0;
    int poff[RPT];
#pragma unroll
    for (int pt = 0; pt < RPT; ++pt) {
      const int rp = pt * 16 + j, ry = rp / RW, rx = rp - ry * RW;
      const int iy = iy0 + ry, ix = ix0 + rx;
      const bool ok = rp < RP && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
      vmask |= (ok ? 1u : 0u) << pt;
      poff[pt] = ok ? (iy * p.w + ix) * p.cin : 0;
    }
    uint4 xf[HASC1 ? RPT : 1];
    if (HASC1) {
#pragma unroll
      for (int pt = 0; pt < RPT; ++pt) {
        xf[pt] = make_uint4(0, 0, 0, 0);
        if (((vmask >> pt) & 1u) && g * 8 < p.cin) xf[pt] = *(const uint4*)(ximg + poff[pt] + g * 8);
      }
    }
    v4f acc3[CT3M][TPT];
#pragma unroll
    for (int ct = 0; ct < CT3M; ++ct)
#pragma unroll
      for (int pt = 0; pt < TPT; ++pt) acc3[ct][pt] = (v4f){0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < p.nchunk; ++c) {
      // ---- this chunk's weights: conv1 fragments (2 channel tiles), reduce fragments (K step c), both from L2 -- consumed after the first LDS round trip
      uint4 a1[2]; float4 bb1[2];
      if (HASC1) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ct = min(c * 2 + q, p.ct1 - 1);
          a1[q] = *(const uint4*)(p.w1 + ((size_t)ct * 64 + lane) * 8);
          bb1[q] = *(const float4*)(p.b1 + ct * 16 + 4 * g);
        }
      }
      uint4 a3[CT3M];
#pragma unroll
      for (int ct = 0; ct < CT3M; ++ct) a3[ct] = (ct < p.ct3 && c < p.kb3) ? *(const uint4*)(p.w3 + (((size_t)ct * p.kb3 + c) * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
      const int chd = c * 32 + 2 * cp;                                        // this lane's depthwise channel pair
      const bool chok = chd < p.cpad_dw;
      // ---- conv1 -> plane (bias, ReLU, bf16; zero outside the image), or the block input's channels [32 c, 32 c + 32)
      if (HASC1) {
#pragma unroll
        for (int pt = 0; pt < RPT; ++pt) {
          const bool ok = ((vmask >> pt) & 1u) != 0;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_w, a1[q]), __builtin_bit_cast(v8bf_w, xf[pt]), acc, 0, 0, 0);
            uint2 o = make_uint2(0, 0);
            if (ok && (c * 2 + q) < p.ct1) { o.x = cvt_pk_bf16(fmaxf(acc[0] + bb1[q].x, 0.f), fmaxf(acc[1] + bb1[q].y, 0.f)); o.y = cvt_pk_bf16(fmaxf(acc[2] + bb1[q].z, 0.f), fmaxf(acc[3] + bb1[q].w, 0.f)); }
            *(uint2*)(pl + (pt * 16 + j) * IBW_PLS + q * 16 + 4 * g) = o;
          }
        }
      } else {
#pragma unroll
        for (int pt = 0; pt < RPT; ++pt) {
          // lane (j, g): pixel pt*16 + j, channels 8 g .. 8 g + 7 of the chunk
          uint4 v = make_uint4(0, 0, 0, 0);
          if (((vmask >> pt) & 1u) && (c * 32 + g * 8) < p.cin) v = *(const uint4*)(ximg + poff[pt] + c * 32 + g * 8);
          *(uint2*)(pl + (pt * 16 + j) * IBW_PLS + g * 8) = make_uint2(v.x, v.y);
          *(uint2*)(pl + (pt * 16 + j) * IBW_PLS + g * 8 + 4) = make_uint2(v.z, v.w);
        }
      }
      ibw_sync();
      // ---- conv2: depthwise k x k, stride S, lane = channel pair; bias first, taps in (ky, kx) order (k_inf_dw's / k_iblock's order); ReLU, bf16 -> y2
      {
        float bd[2] = {0.f, 0.f};
        if (chok) { const float2 b2 = *(const float2*)(p.bdw + chd); bd[0] = b2.x; bd[1] = b2.y; }
#pragma unroll
        for (int ui = 0; ui < UNITS / 4; ++ui) {
          const int u = pg + 4 * ui;
          const int oy = u / UPR, ox0 = (u - oy * UPR) * 4;
          float a[4][2];
#pragma unroll
          for (int o = 0; o < 4; ++o) { a[o][0] = bd[0]; a[o][1] = bd[1]; }
#pragma unroll 1
          for (int ky = 0; ky < K; ++ky) {                 // (not unrolled: unrolled, every tap and column of all k rows is hoisted and the wave count per SIMD halves)
            float wt[K][2];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              float2 w2 = make_float2(0.f, 0.f);
              if (chok) w2 = *(const float2*)(p.wdw + (size_t)(ky * K + kx) * p.cpad_dw + chd);
              wt[kx][0] = w2.x; wt[kx][1] = w2.y;
            }
            const uint16_t* rp_ = pl + ((oy * S + ky) * RW + ox0 * S) * IBW_PLS + 2 * cp;
            float col[SPAN][2];
#pragma unroll
            for (int q = 0; q < SPAN; ++q) {
              uint32_t v = 0;
              if (ox0 * S + q < RW) v = *(const uint32_t*)(rp_ + q * IBW_PLS);
              col[q][0] = __uint_as_float(v << 16); col[q][1] = __uint_as_float(v & 0xffff0000u);
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
              for (int o = 0; o < 4; ++o) {
                a[o][0] = fmaf(col[o * S + kx][0], wt[kx][0], a[o][0]);
                a[o][1] = fmaf(col[o * S + kx][1], wt[kx][1], a[o][1]);
              }
          }
#pragma unroll
          for (int o = 0; o < 4; ++o) *(uint32_t*)(y2 + (oy * TW + ox0 + o) * IBW_Y2S + 2 * cp) = cvt_pk_bf16(fmaxf(a[o][0], 0.f), fmaxf(a[o][1], 0.f));
        }
      }
      ibw_sync();
      // ---- reduce_conv: this chunk's K step into the accumulators of the block output (cout x tile pixels), which stay in registers across the chunks
#pragma unroll
      for (int pt = 0; pt < TPT; ++pt) {
        const uint4 bf = *(const uint4*)(y2 + (pt * 16 + j) * IBW_Y2S + g * 8);
#pragma unroll
        for (int ct = 0; ct < CT3M; ++ct)
          if (ct < p.ct3) acc3[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_w, a3[ct]), __builtin_bit_cast(v8bf_w, bf), acc3[ct][pt], 0, 0, 0);
      }
      ibw_sync();          // the next chunk overwrites the plane / y2 this chunk's reads came from
    }

    // ---- epilogue: + bias, the layer's bf16 rounding, + residual (bf16 add, as frost_infer_add), 4 channels per lane
    uint16_t* dst = p.y + (long long)img * p.ho * p.wo * p.cout;
#pragma unroll
    for (int ct = 0; ct < CT3M; ++ct) {
      if (ct >= p.ct3) continue;
#pragma unroll
      for (int pt = 0; pt < TPT; ++pt) {
        const int ch = ct * 16 + 4 * g, px = pt * 16 + j;
        const int oy = px / TW, ox = px - oy * TW;
        const int gy = ty * TH + oy, gx = tx * TW + ox;
        if (gy < p.ho && gx < p.wo && ch < p.cout) {
          const float4 bb = *(const float4*)(p.b3 + ch);
          uint2 o; o.x = cvt_pk_bf16(acc3[ct][pt][0] + bb.x, acc3[ct][pt][1] + bb.y); o.y = cvt_pk_bf16(acc3[ct][pt][2] + bb.z, acc3[ct][pt][3] + bb.w);
          if (p.residual) {
            const uint2 xr = *(const uint2*)(ximg + ((long long)gy * p.w + gx) * p.cin + ch);          // stride 1, cin == cout
            o.x = cvt_pk_bf16(bf2f(xr.x & 0xffff) + bf2f(o.x & 0xffff), bf2f(xr.x >> 16) + bf2f(o.x >> 16));
            o.y = cvt_pk_bf16(bf2f(xr.y & 0xffff) + bf2f(o.y & 0xffff), bf2f(xr.y >> 16) + bf2f(o.y >> 16));
          }
          *(uint2*)(dst + ((long long)gy * p.wo + gx) * p.cout + ch) = o;
        }
      }
    }
  }
}

template <int K, int S, int TH, int TW, bool HASC1>
static int launch_ibw(const IBwP& p, hipStream_t s) {
  constexpr int RP = ((TH - 1) * S + K) * ((TW - 1) * S + K), RPT = (RP + 15) / 16, TPT = TH * TW / 16;
  const size_t lds = (size_t)4 * (RPT * 16 * IBW_PLS + TPT * 16 * IBW_Y2S) * 2;
  static bool set = false; static int occ = 1;
  if (!set) {
    (void)hipFuncSetAttribute((const void*)k_iblock_w<K, S, TH, TW, HASC1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_iblock_w<K, S, TH, TW, HASC1>, 256, lds) != hipSuccess || occ < 1) occ = 1;
    set = true;
  }
  static const int rounds = getenv("FROST_IBW_ROUNDS") ? atoi(getenv("FROST_IBW_ROUNDS")) : 4;      // persistent waves: the grid is `rounds` resident rounds at most
  long long grid = (p.ntiles + 3) / 4;
  const long long cap = (long long)256 * occ * rounds;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((k_iblock_w<K, S, TH, TW, HASC1>), dim3((unsigned)grid), dim3(256), lds, s, p);
  return frost_check_launch("infer_block_w");
}

// 1 if frost_infer_block_w takes this bottleneck with the wave tile (th, tw): no squeeze conv, conv1 K <= 32 (or no conv1: the depthwise conv reads the block input),
// Cout <= 48, tiles 4 x 4 / 4 x 8, the (k, stride, conv1) combinations of FrostNet's high-resolution bottlenecks
extern "C" int frost_infer_block_w_ok(int cin, int r, int cexp, int cout, int k, int stride, int has_conv1, int th, int tw) {
  if (r != 0 || (cin & 7) || (cexp & 7) || (cout & 7) || cout > 48) return 0;
  if (!(th == 4 && (tw == 4 || tw == 8))) return 0;
  if (has_conv1) { if (cin > 32) return 0; return (k == 3 && (stride == 1 || stride == 2)) || (k == 5 && stride == 2); }
  return (cexp == cin && k == 3 && stride == 1) ? 1 : 0;
}

/* One Frost bottleneck WITHOUT a squeeze conv, bf16 inference, one launch, one wave per (th x tw) output tile.  Arguments as frost_infer_block (packs / folded
 * biases of frost_infer_weight_prep; w1 == NULL: no expansion conv).  Results bit-identical to frost_infer_block with chunk = 32 and to the layer launches. */
extern "C" int frost_infer_block_w(const uint16_t* x, const uint16_t* w1, const float* b1, const float* wdw, const float* bdw, const uint16_t* w3, const float* b3,
                                   int n, int h, int w, int cin, int cexp, int cout, int k, int stride, int residual, int th, int tw, uint16_t* y, void* stream) {
  FROST_REQUIRE(frost_infer_block_w_ok(cin, 0, cexp, cout, k, stride, w1 != nullptr, th, tw), "infer_block_w: unsupported geometry / tile");
  FROST_REQUIRE(!residual || (stride == 1 && cin == cout), "infer_block_w: residual needs stride 1 and cin == cout");
  IBwP p = {};
  p.x = x; p.y = y; p.w1 = w1; p.b1 = b1; p.wdw = wdw; p.bdw = bdw; p.w3 = w3; p.b3 = b3;
  const int pad = (k - 1) / 2;
  p.n = n; p.h = h; p.w = w; p.ho = (h + 2 * pad - k) / stride + 1; p.wo = (w + 2 * pad - k) / stride + 1;
  p.cin = cin; p.cexp = cexp; p.cout = cout;
  p.ct1 = round_up(cexp, 16) / 16; p.kb3 = round_up(cexp, 32) / 32; p.ct3 = round_up(cout, 16) / 16; p.cpad_dw = round_up(cexp, 16); p.nchunk = (cexp + 31) / 32;
  p.tiles_x = (p.wo + tw - 1) / tw; p.tiles_y = (p.ho + th - 1) / th; p.residual = residual;
  p.ntiles = (long long)n * p.tiles_x * p.tiles_y;
  hipStream_t s = as_stream(stream);
  const bool c1 = w1 != nullptr;
  if (!c1) return (tw == 4) ? launch_ibw<3, 1, 4, 4, false>(p, s) : launch_ibw<3, 1, 4, 8, false>(p, s);
  if (k == 3 && stride == 1) return (tw == 4) ? launch_ibw<3, 1, 4, 4, true>(p, s) : launch_ibw<3, 1, 4, 8, true>(p, s);
  if (k == 3) return (tw == 4) ? launch_ibw<3, 2, 4, 4, true>(p, s) : launch_ibw<3, 2, 4, 8, true>(p, s);
  return (tw == 4) ? launch_ibw<5, 2, 4, 4, true>(p, s) : launch_ibw<5, 2, 4, 8, true>(p, s);
}
