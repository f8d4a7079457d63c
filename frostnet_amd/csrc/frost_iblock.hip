// Whole-bottleneck fused bf16 INFERENCE kernel (BASELINE.json config c2): one launch computes
//   CascadePreExBottleneck.forward (/root/reference/frostnet.py:124-145) in eval mode with BatchNorm folded (Classification/evaluate.py:131-143 style):
//   [squeeze 1x1 + ReLU -> cat] -> conv1 1x1 expand + ReLU -> conv2 depthwise k x k (stride 1 / 2) + ReLU -> reduce_conv 1x1 (linear) [-> + x]
// Inference has no batch-statistics barrier between the layers, so the EXPANDED tensors (conv1 / conv2 outputs, 3-6 x the block input) never leave the chip:
// only the block input and the block output touch HBM (2.28 M instead of 13.07 M elements per image over the network).
//
// Work decomposition.  A workgroup (4 or 8 waves) owns one spatial OUTPUT tile th x tw of one image.  The in-image part of the input region the tile's depthwise
// windows cover ((th-1)*s + k) x ((tw-1)*s + k) pixels) is staged in LDS once as rows [pixel][r + cin] bf16 -- the squeeze conv writes its r
// channels in front of the copied input, which IS the cat.  Then, per 64-channel chunk of the expanded width:
//   conv1   bf16 MFMA 16x16x32, D[channel][pixel]; wave = one 16-channel tile, all region pixels; + bias, ReLU, bf16 -> plane [pixel][64] (zero outside the image:
//           the depthwise conv zero-pads conv1's OUTPUT)
//   conv2   depthwise from the plane, lane = channel pair, fp32 FMA in the layer kernel's (ky, kx) order, 4 outputs along x share their input columns;
//           + bias, ReLU, bf16 -> y2 [out pixel][64] = the B operand of the next GEMM as it lies
//   reduce  bf16 MFMA, K-split over the chunks: the fp32 accumulators of the block output (cout x th*tw) stay in registers across all chunks
// and at the end: + bias, bf16 (the layer's own rounding), + residual x (bf16 add, as frost_infer_add), store.
// Rounding points are those of the layer-by-layer kernels (every layer output is rounded to bf16 once), so the two paths agree to summation order.
#include "frost_common.h"

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

struct IBlkP {
  const uint16_t* x; uint16_t* y;
  const uint16_t* wsq; const float* bsq;        // squeeze_conv: A-fragment pack [ct][kb][64][8], folded bias (NULL: no squeeze / cat)
  const uint16_t* w1; const float* b1;          // conv1 (NULL: the block's depthwise conv reads the block input, frostnet.py:105-108)
  const float* wdw; const float* bdw;           // conv2: fp32 taps [k*k][cpad_dw], folded bias
  const uint16_t* w3; const float* b3;          // reduce_conv
  int n, h, w, ho, wo, cin, r, cexp, cout;
  int kb_sq, ct_sq, kb1, ct1, kb3, ct3, cpad_dw, nchunk;
  int th, tw, tiles_x, tiles_y, rh, rw, rp, rpt, tp, tpt, xs, residual, tw4, waves, chunk;
};

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would put the L2 round trip of every weight / tap
// prefetch behind the barrier it was issued in front of (the prefetched registers are consumed later, behind the compiler's own waits)
__device__ __forceinline__ void ib_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// plane row stride (bf16 elements) CH + 4: 136 B (72 B), the 16 pixel rows of an MFMA tile fall into distinct 8-byte bank slots; y2 row stride CH + 8
#define IB_PLS(CH) ((CH) + 4)
#define IB_Y2S(CH) ((CH) + 8)

// CH = channels of the expanded width per chunk: 64, or 32 for the narrow high-resolution blocks (cexp = 32 / 72 / 96 / 144: a 64-wide chunk leaves 25-50 % of the
// depthwise lanes on padding channels there, and the smaller planes let more workgroups share a CU)
template <int K, int S, int MAXT, int KB1M, int NW, int CH>
__global__ __launch_bounds__(NW * 64) void k_iblock(const IBlkP p) {
  constexpr int PAD = (K - 1) / 2, SPAN = 3 * S + K, TAPW = (K * K + 1) * CH, NT = NW * 64, TPT = (TAPW + NT - 1) / NT;
  constexpr int PLS = IB_PLS(CH), Y2S = IB_Y2S(CH), LCH = (CH == 64) ? 6 : 5, TPC = CH / 16, KSC = CH / 32, CPN = CH / 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint16_t* const xc = (uint16_t*)smem;                                    // [rpt*16][xs]
  uint16_t* const pl = xc + (size_t)p.rpt * 16 * p.xs;                     // [rpt*16][PLS]
  uint16_t* const y2 = pl + (size_t)p.rpt * 16 * PLS;                      // [tpt*16][Y2S]
  float* const tapl = (float*)(y2 + (size_t)p.tpt * 16 * Y2S);             // [k*k + 1][CH]: the chunk's depthwise taps and (last row) folded bias
  uint16_t* const ptab = (uint16_t*)(tapl + TAPW);                         // [rpt*16]: staged (in-image) pixel -> plane row, 0xffff past the last one
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x; const int ty = b % p.tiles_y; const int img = b / p.tiles_y;
  const int iy0 = ty * p.th * S - PAD, ix0 = tx * p.tw * S - PAD;          // image coordinates of region pixel (0, 0)
  // Only the IN-IMAGE part of the region is staged and convolved (a sub-rectangle [ry0, ry1) x [rx0, rx1) of it, compact pixel index sp): conv2 zero-pads
  // conv1's OUTPUT, so the plane is simply zero elsewhere -- on a 7 x 7 map with a 5 x 5 depthwise conv that is 49 of the region's 121 pixels.
  const int ry0 = max(0, -iy0), rx0 = max(0, -ix0);
  const int srh = min(p.rh, p.h - iy0) - ry0, srw = min(p.rw, p.w - ix0) - rx0;
  const int sp = srh * srw, spt = (sp + 15) >> 4;

  // ---- phase 0: zero the staging rows (K padding of the MFMA operands, pad rows), the plane (conv2's zero padding) and y2's pad rows; pixel table
  {
    uint4* z = (uint4*)xc; const int nx = (spt * 16 * p.xs) >> 3;
    for (int i = tid; i < nx; i += NT) z[i] = make_uint4(0, 0, 0, 0);
    uint4* zp_ = (uint4*)pl; const int np_ = (p.rpt * 16 * PLS) >> 3;
    for (int i = tid; i < np_; i += NT) zp_[i] = make_uint4(0, 0, 0, 0);
    uint4* z2 = (uint4*)y2; const int ny = (p.tpt * 16 * Y2S) >> 3;
    for (int i = tid; i < ny; i += NT) z2[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < spt * 16; i += NT) {
      const int sy = i / srw, sx = i - sy * srw;
      ptab[i] = (i < sp) ? (uint16_t)((ry0 + sy) * p.rw + rx0 + sx) : (uint16_t)0xffff;
    }
  }
  __syncthreads();
  {
    const int upr = p.cin >> 3, total = sp * upr;
    const uint16_t* src = p.x + (int64_t)img * p.h * p.w * p.cin;
    for (int u = tid; u < total; u += NT) {
      const int px = u / upr, part = u - px * upr;
      const int sy = px / srw, sx = px - sy * srw;
      const int iy = iy0 + ry0 + sy, ix = ix0 + rx0 + sx;
      *(uint4*)(xc + (size_t)px * p.xs + p.r + part * 8) = *(const uint4*)(src + ((int64_t)iy * p.w + ix) * p.cin + part * 8);
    }
  }
  // the depthwise taps / bias of a chunk travel global -> registers -> LDS one chunk ahead of their use
  float tpre[TPT];
  auto load_taps = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int i = tid + q * NT; const int t = i >> LCH, ch = c * CH + (i & (CH - 1));
      float v = 0.f;
      if (i < TAPW && ch < p.cpad_dw) v = (t < K * K) ? p.wdw[(size_t)t * p.cpad_dw + ch] : p.bdw[ch];
      tpre[q] = v;
    }
  };
  auto store_taps = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < TPT; ++q) { const int i = tid + q * NT; if (i < TAPW) tapl[i] = tpre[q]; }
  };
  load_taps(0);
  store_taps();
  __syncthreads();

  // ---- phase 1: squeeze_conv (1x1, ReLU) over every region pixel, written in front of the input = cat([squeezed, x], 1)
  if (p.wsq) {
    const int ntile = spt * p.ct_sq;
    for (int idx = wv; idx < ntile; idx += NW) {
      const int ct = idx % p.ct_sq, pt = idx / p.ct_sq;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      for (int kb = 0; kb < p.kb_sq; ++kb) {
        const uint4 af = *(const uint4*)(p.wsq + (((size_t)ct * p.kb_sq + kb) * 64 + lane) * 8);
        const uint4 bf = *(const uint4*)(xc + (size_t)(pt * 16 + j) * p.xs + p.r + kb * 32 + g * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af), __builtin_bit_cast(v8bf, bf), acc, 0, 0, 0);
      }
      const int ch = ct * 16 + 4 * g;
      if (ch < p.r) {
        const float4 bb = *(const float4*)(p.bsq + ch);
        uint2 o; o.x = cvt_pk_bf16(fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f)); o.y = cvt_pk_bf16(fmaxf(acc[2] + bb.z, 0.f), fmaxf(acc[3] + bb.w, 0.f));
        *(uint2*)(xc + (size_t)(pt * 16 + j) * p.xs + ch) = o;
      }
    }
    __syncthreads();
  }

  // reduce_conv accumulators: tile t of this wave = linear tile index wv + NW t over (ct3, tpt)
  v4f acc3[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) acc3[t] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int ntile3 = p.ct3 * p.tpt;
  const int cp = tid & (CPN - 1), pg = tid / CPN;                            // depthwise: channel pair, pixel group

  uint4 af[KB1M];
  const int wq = wv % TPC, wph = wv / TPC;                                    // conv1: channel tile of the chunk, pixel-tile phase (NW / TPC waves share a channel tile)
  auto load_w1 = [&](int c) __attribute__((always_inline)) {
    const int ct = min(c * TPC + wq, p.ct1 - 1);
#pragma unroll
    for (int kb = 0; kb < KB1M; ++kb) if (kb < p.kb1) af[kb] = *(const uint4*)(p.w1 + (((size_t)ct * p.kb1 + kb) * 64 + lane) * 8);
  };
  if (p.w1) load_w1(0);

  for (int c = 0; c < p.nchunk; ++c) {
    // ---- conv1 -> plane (or the block input itself when the block has no expansion conv)
    if (p.w1) {
      const int ct = c * TPC + wq;
      if (ct < p.ct1) {
        const int ch = ct * 16 + 4 * g;
        const float4 bb = *(const float4*)(p.b1 + ch);
        for (int pt = wph; pt < spt; pt += NW / TPC) {
          v4f acc = {0.f, 0.f, 0.f, 0.f};
          const uint16_t* brow = xc + (size_t)(pt * 16 + j) * p.xs + g * 8;
#pragma unroll
          for (int kb = 0; kb < KB1M; ++kb) if (kb < p.kb1) {
            const uint4 bf = *(const uint4*)(brow + kb * 32);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af[kb]), __builtin_bit_cast(v8bf, bf), acc, 0, 0, 0);
          }
          const unsigned prow = ptab[pt * 16 + j];
          if (prow != 0xffffu) {
            uint2 o; o.x = cvt_pk_bf16(fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f)); o.y = cvt_pk_bf16(fmaxf(acc[2] + bb.z, 0.f), fmaxf(acc[3] + bb.w, 0.f));
            *(uint2*)(pl + (size_t)prow * PLS + wq * 16 + 4 * g) = o;
          }
        }
      } else {
        for (int pt = wph; pt < spt; pt += NW / TPC) {
          const unsigned prow = ptab[pt * 16 + j];
          if (prow != 0xffffu) *(uint2*)(pl + (size_t)prow * PLS + wq * 16 + 4 * g) = make_uint2(0, 0);
        }
      }
      if (c + 1 < p.nchunk) load_w1(c + 1);                                  // the next chunk's fragments travel under the depthwise and reduce phases
    } else {
      for (int u = tid; u < sp * (CH / 8); u += NT) {                         // plane = channels [CH c, CH c + CH) of the staged input
        const int px = u / (CH / 8), part = u % (CH / 8); const int ch = c * CH + part * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ch < p.cin) v = *(const uint4*)(xc + (size_t)px * p.xs + p.r + ch);
        *(uint2*)(pl + (size_t)ptab[px] * PLS + part * 8) = make_uint2(v.x, v.y);          // (rows are 8-byte aligned: stride CH + 4 elements)
        *(uint2*)(pl + (size_t)ptab[px] * PLS + part * 8 + 4) = make_uint2(v.z, v.w);
      }
    }
    ib_barrier();
    // ---- conv2: depthwise k x k, stride S, lane = channel pair; bias first, taps in (ky, kx) order (k_inf_dw's order)
    {
      if (c + 1 < p.nchunk) load_taps(c + 1);                                // in flight under this phase; stored after it
      float bd[2];
      { const float2 b2 = *(const float2*)(tapl + K * K * CH + 2 * cp); bd[0] = b2.x; bd[1] = b2.y; }
      const int units = p.th * p.tw4;
#pragma unroll 1
      for (int u = pg; u < units; u += NT / CPN) {
        const int oy = u / p.tw4, ox0 = (u - oy * p.tw4) * 4;
        float a[4][2];
#pragma unroll
        for (int o = 0; o < 4; ++o) { a[o][0] = bd[0]; a[o][1] = bd[1]; }
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {                                     // (not unrolled: unrolled, all k*k taps are hoisted into 2 k*k registers and the kernel drops to one wave per SIMD)
          const int row = oy * S + ky;
          const uint16_t* rp_ = pl + ((size_t)row * p.rw + ox0 * S) * PLS + 2 * cp;
          float wt[K][2];                                                     // this kernel row's taps (LDS: 8-byte reads, the lanes of a pixel group share none)
#pragma unroll
          for (int kx = 0; kx < K; ++kx) { const float2 w2 = *(const float2*)(tapl + (ky * K + kx) * CH + 2 * cp); wt[kx][0] = w2.x; wt[kx][1] = w2.y; }
          float col[SPAN][2];
#pragma unroll
          for (int q = 0; q < SPAN; ++q) {
            uint32_t v = 0;
            if (ox0 * S + q < p.rw) v = *(const uint32_t*)(rp_ + (size_t)q * PLS);
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
        for (int o = 0; o < 4; ++o)
          if (ox0 + o < p.tw) *(uint32_t*)(y2 + (size_t)(oy * p.tw + ox0 + o) * Y2S + 2 * cp) = cvt_pk_bf16(fmaxf(a[o][0], 0.f), fmaxf(a[o][1], 0.f));
      }
    }
    ib_barrier();
    if (c + 1 < p.nchunk) store_taps();                                      // (read again only after the barrier that ends this chunk)
    // ---- reduce_conv: this chunk's 64 K values into the persistent accumulators
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int idx = wv + NW * t;
      if (idx < ntile3) {
        const int ct = idx % p.ct3, pt = idx / p.ct3;
#pragma unroll
        for (int k2 = 0; k2 < KSC; ++k2) {
          const int kb = KSC * c + k2;
          if (kb < p.kb3) {
            const uint4 af = *(const uint4*)(p.w3 + (((size_t)ct * p.kb3 + kb) * 64 + lane) * 8);
            const uint4 bf = *(const uint4*)(y2 + (size_t)(pt * 16 + j) * Y2S + k2 * 32 + g * 8);
            acc3[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af), __builtin_bit_cast(v8bf, bf), acc3[t], 0, 0, 0);
          }
        }
      }
    }
    ib_barrier();
  }

  // ---- epilogue: + bias, the layer's bf16 rounding, + residual (bf16 add), store 4 channels per lane
  uint16_t* dst = p.y + (int64_t)img * p.ho * p.wo * p.cout;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int idx = wv + NW * t;
    if (idx < ntile3) {
      const int ct = idx % p.ct3, pt = idx / p.ct3;
      const int ch = ct * 16 + 4 * g, px = pt * 16 + j;
      const int oy = px / p.tw, ox = px - oy * p.tw;
      const int gy = ty * p.th + oy, gx = tx * p.tw + ox;
      if (px < p.tp && gy < p.ho && gx < p.wo && ch < p.cout) {
        const float4 bb = *(const float4*)(p.b3 + ch);
        uint2 o; o.x = cvt_pk_bf16(acc3[t][0] + bb.x, acc3[t][1] + bb.y); o.y = cvt_pk_bf16(acc3[t][2] + bb.z, acc3[t][3] + bb.w);
        if (p.residual) {
          const uint2 xr = *(const uint2*)(xc + (size_t)((oy + PAD - ry0) * srw + ox + PAD - rx0) * p.xs + p.r + ch);   // stride 1: region pixel (oy + pad, ox + pad), staged compactly
          o.x = cvt_pk_bf16(bf2f(xr.x & 0xffff) + bf2f(o.x & 0xffff), bf2f(xr.x >> 16) + bf2f(o.x >> 16));
          o.y = cvt_pk_bf16(bf2f(xr.y & 0xffff) + bf2f(o.y & 0xffff), bf2f(xr.y >> 16) + bf2f(o.y >> 16));
        }
        *(uint2*)(dst + ((int64_t)gy * p.wo + gx) * p.cout + ch) = o;
      }
    }
  }
}

static size_t iblock_lds(int rpt, int tpt, int xs, int k, int ch) {
  return ((size_t)rpt * 16 * xs + (size_t)rpt * 16 * IB_PLS(ch) + (size_t)tpt * 16 * IB_Y2S(ch)) * 2 + (size_t)(k * k + 1) * ch * 4 + (size_t)rpt * 16 * 2;
}

static int iblock_waves(const IBlkP& p) {
  static const int forced = getenv("FROST_IB_NW") ? atoi(getenv("FROST_IB_NW")) : 0;
  if (forced == 4 || forced == 8 || forced == 16) return forced;
  if (p.waves == 4 || p.waves == 8 || p.waves == 16) return p.waves;   // the caller measured (frostnet_amd/infer.py "auto"); 16: a 7 x 7 map is ONE workgroup per image -- with 256 images that is one workgroup per CU, and only more waves shorten its serial phases
  // 8 waves where a workgroup carries a whole (small) map and a wide expansion: its three phases per 64-channel chunk are serial, more waves shorten each
  return (p.tp >= 49 && p.cexp >= 256) ? 8 : 4;
}

template <int K, int S>
static int launch_iblock(const IBlkP& p, size_t lds, hipStream_t s) {
  const int nw = iblock_waves(p);
  const int per_wave = (p.ct3 * p.tpt + nw - 1) / nw;
  const dim3 grid((unsigned)(p.n * p.tiles_x * p.tiles_y));
#define IB_GO(MT, KB, NW_, CH_) do { \
    static bool set = false; \
    if (!set) { (void)hipFuncSetAttribute((const void*)k_iblock<K, S, MT, KB, NW_, CH_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((k_iblock<K, S, MT, KB, NW_, CH_>), grid, dim3(NW_ * 64), lds, s, p); } while (0)
#define IB_GO2(MT, NW_) do { if (p.chunk == 32) IB_GO(MT, 2, NW_, 32); else if (p.kb1 <= 2) IB_GO(MT, 2, NW_, 64); else IB_GO(MT, 10, NW_, 64); } while (0)
  if (nw == 16) {
    if (p.chunk == 32) { frost_set_error("infer_block: 16 waves take 64-channel chunks only"); return 1; }
    if (per_wave <= 4) { if (p.kb1 <= 2) IB_GO(4, 2, 16, 64); else IB_GO(4, 10, 16, 64); }
    else { frost_set_error("infer_block: too many output tiles per wave"); return 1; }
  } else if (nw == 8) {
    if (per_wave <= 4) IB_GO2(4, 8); else if (per_wave <= 12) IB_GO2(12, 8);
    else { frost_set_error("infer_block: too many output tiles per wave"); return 1; }
  } else {
    if (per_wave <= 4) IB_GO2(4, 4); else if (per_wave <= 8) IB_GO2(8, 4); else if (per_wave <= 12) IB_GO2(12, 4); else if (per_wave <= 20) IB_GO2(20, 4);
    else { frost_set_error("infer_block: too many output tiles per wave"); return 1; }
  }
#undef IB_GO2
#undef IB_GO
  return frost_check_launch("infer_block");
}

// > 0 (= LDS bytes per workgroup) if frost_infer_block can run this bottleneck geometry with the tile (th, tw); the host picks the tile (frostnet_amd/infer.py)
extern "C" int frost_infer_block_ok(int h, int w, int cin, int r, int cexp, int cout, int k, int stride, int th, int tw) {
  if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return 0;
  if ((cin & 7) || (r & 7) || (cexp & 7) || (cout & 7) || th < 1 || tw < 1) return 0;
  const int rh = (th - 1) * stride + k, rw = (tw - 1) * stride + k;
  const int rpt = (rh * rw + 15) / 16, tpt = (th * tw + 15) / 16;
  const int kc = r + round_up(cin, 32) > round_up(r + cin, 32) ? r + round_up(cin, 32) : round_up(r + cin, 32);
  const size_t lds = iblock_lds(rpt, tpt, kc + 8, k, 64);
  if (lds > 160 * 1024 || rh * rw >= 0xffff) return 0;
  if (round_up(r + cin, 32) / 32 > 10) return 0;                       // conv1 K steps held in registers
  if ((round_up(cout, 16) / 16 * tpt + 3) / 4 > 20) return 0;          // reduce accumulators per wave
  (void)h; (void)w;
  return (int)lds;                                                     // > 0: the workgroup's LDS bytes (the host weighs tiles by the residency they allow)
}

/* One Frost bottleneck, bf16 inference, as ONE launch.  x: [n][h][w][cin] bf16 NHWC; y: [n][ho][wo][cout].  Packs / folded biases are those of
 * frost_infer_weight_prep (FrostIDesc.pack / .biasf): wsq / w1 / w3 = bf16 A-fragment packs with kpad = round_up(K, 32) (NULL = layer absent), wdw = fp32 taps
 * [k*k][round_up(cexp,16)].  r = squeeze width (0 without squeeze), cexp = depthwise width (= r + cin ... when conv1 is absent: = cin).  residual: + x.
 * waves: 4 or 8 waves per workgroup, 0 = the kernel's own rule (8 for a whole small map with a wide expansion).  chunk: channels of the expanded width per
 * pass, 64 (= 0) or 32 (the narrow high-resolution blocks, conv1 K <= 64). */
extern "C" int frost_infer_block(const uint16_t* x, const uint16_t* wsq, const float* bsq, const uint16_t* w1, const float* b1, const float* wdw,
                                 const float* bdw, const uint16_t* w3, const float* b3, int n, int h, int w, int cin, int r, int cexp, int cout, int k,
                                 int stride, int residual, int th, int tw, int waves, int chunk, uint16_t* y, void* stream) {
  FROST_REQUIRE(frost_infer_block_ok(h, w, cin, r, cexp, cout, k, stride, th, tw), "infer_block: unsupported geometry / tile");
  if (chunk == 0) chunk = 64;
  FROST_REQUIRE(chunk == 64 || (chunk == 32 && round_up(r + cin, 32) <= 64), "infer_block: chunk = 64, or 32 for blocks whose conv1 has K <= 64");
  FROST_REQUIRE(!residual || (stride == 1 && cin == cout), "infer_block: residual needs stride 1 and cin == cout");
  FROST_REQUIRE((wsq != nullptr) == (r > 0) && (w1 != nullptr || cexp == cin), "infer_block: inconsistent layer set");
  IBlkP p = {};
  p.x = x; p.y = y; p.wsq = wsq; p.bsq = bsq; p.w1 = w1; p.b1 = b1; p.wdw = wdw; p.bdw = bdw; p.w3 = w3; p.b3 = b3;
  const int pad = (k - 1) / 2;
  p.n = n; p.h = h; p.w = w; p.ho = (h + 2 * pad - k) / stride + 1; p.wo = (w + 2 * pad - k) / stride + 1;
  p.cin = cin; p.r = r; p.cexp = cexp; p.cout = cout;
  p.kb_sq = round_up(cin, 32) / 32; p.ct_sq = round_up(r, 16) / 16;
  p.kb1 = round_up(r + cin, 32) / 32; p.ct1 = round_up(cexp, 16) / 16;
  p.kb3 = round_up(cexp, 32) / 32; p.ct3 = round_up(cout, 16) / 16; p.cpad_dw = round_up(cexp, 16); p.nchunk = (cexp + chunk - 1) / chunk; p.chunk = chunk;
  p.th = th; p.tw = tw; p.tiles_x = (p.wo + tw - 1) / tw; p.tiles_y = (p.ho + th - 1) / th;
  p.rh = (th - 1) * stride + k; p.rw = (tw - 1) * stride + k; p.rp = p.rh * p.rw; p.rpt = (p.rp + 15) / 16;
  p.tp = th * tw; p.tpt = (p.tp + 15) / 16; p.tw4 = (tw + 3) / 4;
  const int kc = (r + round_up(cin, 32) > round_up(r + cin, 32)) ? r + round_up(cin, 32) : round_up(r + cin, 32);
  p.xs = kc + 8; p.residual = residual; p.waves = waves;
  const size_t lds = iblock_lds(p.rpt, p.tpt, p.xs, k, chunk);
  hipStream_t s = as_stream(stream);
  if (k == 3 && stride == 1) return launch_iblock<3, 1>(p, lds, s);
  if (k == 3) return launch_iblock<3, 2>(p, lds, s);
  if (stride == 1) return launch_iblock<5, 1>(p, lds, s);
  return launch_iblock<5, 2>(p, lds, s);
}
