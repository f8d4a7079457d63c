// bf16 inference of the FLOAT (not fake-quantised) FrostNet graph -- BASELINE.json config c2 (Large, B = 256, bf16).
// Eval-mode BatchNorm is folded into the conv (W' = W * gamma/sqrt(rv+eps), b' = beta - rm * gamma/sqrt(rv+eps)) once per call
// by frost_infer_weight_prep; activations are NHWC bf16; every conv accumulates in fp32 (bf16 MFMA 16x16x32 for the 1x1s and the
// im2col'd stem, fp32 FMA for the depthwise convs), adds the folded bias, applies ReLU and rounds to bf16 once.
// replaces (eval mode, float model): frostnet.py:14-60 ConvBNReLU / ConvBN, :108-121 block wiring, :295-299 head.
#include "frost_common.h"

typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ weight preparation
__device__ __forceinline__ float inf_sf(const FrostIDesc& d, int co) { return d.gamma ? d.gamma[co] / sqrtf(d.rvar[co] + FROST_BN_EPS) : 1.0f; }
__global__ __launch_bounds__(256) void k_inf_prep(const FrostIDesc* descs) {
  const FrostIDesc d = descs[blockIdx.y];
  for (int c = blockIdx.x * 256 + threadIdx.x; c < d.cpad; c += gridDim.x * 256) {
    float b = 0.0f;
    if (c < d.cout) b = d.gamma ? d.beta[c] - d.rmean[c] * inf_sf(d, c) : (d.beta ? d.beta[c] : 0.0f);
    d.biasf[c] = b;
  }
  if (d.kind == 0 || d.kind == 2) {       // MFMA A-fragments: [ct][kb][lane][8]: W'[ct*16 + (lane&15)][kb*32 + (lane>>4)*8 + e]
    const int CT = d.cpad / 16, KB = d.kpad / 32;
    const int64_t nel = (int64_t)CT * KB * 64 * 8;
    uint16_t* pk = (uint16_t*)d.pack;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < nel; i += (int64_t)gridDim.x * 256) {
      const int e = (int)(i & 7); const int lane = (int)((i >> 3) & 63); const int64_t t = i >> 9; const int kb = (int)(t % KB), ct = (int)(t / KB);
      const int co = ct * 16 + (lane & 15); const int k = kb * 32 + (lane >> 4) * 8 + e;
      float v = 0.0f;
      if (co < d.cout) {
        if (d.kind == 0) { if (k < d.cin_g) v = d.w[(int64_t)co * d.cin_g + k] * inf_sf(d, co); }
        else { const int tap = k >> 2, c = k & 3; if (tap < d.kk && c < d.cin_g) v = d.w[((int64_t)co * d.cin_g + c) * d.kk + tap] * inf_sf(d, co); }   // stem: k = tap*4 + c
      }
      pk[i] = f2bf(v);
    }
  } else if (d.kind == 1) {               // depthwise: fp32 [tap][cpad]
    float* pk = (float*)d.pack;
    const int64_t nel = (int64_t)d.kk * d.cpad;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < nel; i += (int64_t)gridDim.x * 256) {
      const int c = (int)(i % d.cpad), tap = (int)(i / d.cpad);
      pk[i] = (c < d.cout) ? d.w[(int64_t)c * d.kk + tap] * inf_sf(d, c) : 0.0f;
    }
  }
}
extern "C" int frost_infer_weight_prep(const FrostIDesc* descs, int nlayers, void* stream) {
  if (nlayers <= 0) return 0;
  hipLaunchKernelGGL(k_inf_prep, dim3(64, nlayers), dim3(256), 0, as_stream(stream), descs);
  return frost_check_launch("infer_weight_prep");
}

// ------------------------------------------------------------------------------------------------ stem im2col (fp32 image -> bf16)
// 3x3 stride-2 pad-1 patches of the logical (N,3,H,W) fp32 image -> [npix_out][64] bf16, K index = tap*4 + c (c == 3 and k >= 36: 0)
__global__ __launch_bounds__(256) void k_inf_stem_im2col(const float* __restrict__ x, int n, int h, int w, int ho, int wo, int64_t sn, int64_t sc,
                                                         int64_t sh, int64_t sw, uint16_t* __restrict__ out) {
  const int64_t tot = (int64_t)n * ho * wo * 16;           // 16 units of 4 k-values (one tap) per output pixel
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int tap = (int)(i & 15); int64_t p = i >> 4; const int ox = (int)(p % wo); p /= wo; const int oy = (int)(p % ho); const int in = (int)(p / ho);
    float v[3] = {0.f, 0.f, 0.f};
    if (tap < 9) {
      const int iy = oy * 2 - 1 + tap / 3, ix = ox * 2 - 1 + tap % 3;
      if (iy >= 0 && iy < h && ix >= 0 && ix < w)
        for (int c = 0; c < 3; ++c) v[c] = x[in * sn + c * sc + iy * sh + ix * sw];
    }
    uint2 o; o.x = cvt_pk_bf16(v[0], v[1]); o.y = cvt_pk_bf16(v[2], 0.0f);
    *(uint2*)(out + (i << 2)) = o;
  }
}
extern "C" int frost_infer_stem_im2col(const float* x, int n, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, uint16_t* out,
                                       void* stream) {
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t tot = (int64_t)n * ho * wo * 16; int64_t grid = (tot + 255) / 256; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_inf_stem_im2col, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, n, h, w, ho, wo, sn, sc, sh, sw, out);
  return frost_check_launch("infer_stem_im2col");
}

// the 1x1 layers (and the im2col'd stem) run on the pointwise skeleton of frost_pw.hip (frost_infer_pw is defined there)

// ------------------------------------------------------------------------------------------------ depthwise (fp32 FMA)
// one thread = 8 channels x 4 consecutive output pixels of a row: per kernel row the (3*stride + k) input columns are loaded once
// (NHWC: 8 channels = one 16-byte load) and every tap's weights once for the four outputs -- 2-2.5x fewer loads than one output per thread
#define IDW_WO 4
template <int K, int S>
__global__ __launch_bounds__(256) void k_inf_dw(const uint16_t* __restrict__ x, const float* __restrict__ wf, const float* __restrict__ biasf, int n,
                                                int h, int w, int c, int cpad, int ho, int wo, int relu, uint16_t* __restrict__ y) {
  constexpr int PAD = (K - 1) / 2, SPAN = (IDW_WO - 1) * S + K;
  const int c8n = c >> 3; const int wo4 = (wo + IDW_WO - 1) / IDW_WO;
  const int64_t tot = (int64_t)n * ho * wo4 * c8n;
  const float lo = relu ? 0.0f : -INFINITY;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % c8n); int64_t p = i / c8n; const int oxg = (int)(p % wo4); p /= wo4; const int oy = (int)(p % ho); const int in = (int)(p / ho);
    const int ch = c8 * 8, ox0 = oxg * IDW_WO, ix0 = ox0 * S - PAD;
    float acc[IDW_WO][8];
    { const float4 b0 = *(const float4*)(biasf + ch), b1 = *(const float4*)(biasf + ch + 4);
#pragma unroll
      for (int o = 0; o < IDW_WO; ++o) { acc[o][0] = b0.x; acc[o][1] = b0.y; acc[o][2] = b0.z; acc[o][3] = b0.w; acc[o][4] = b1.x; acc[o][5] = b1.y; acc[o][6] = b1.z; acc[o][7] = b1.w; } }
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * S - PAD + ky; if (iy < 0 || iy >= h) continue;
      const uint16_t* rowp = x + ((int64_t)in * h + iy) * w * c + ch;
      float col[SPAN][8];
#pragma unroll
      for (int q = 0; q < SPAN; ++q) {
        const int ix = ix0 + q;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ix >= 0 && ix < w) v = *(const uint4*)(rowp + (int64_t)ix * c);
        col[q][0] = bf2f(v.x & 0xffff); col[q][1] = bf2f(v.x >> 16); col[q][2] = bf2f(v.y & 0xffff); col[q][3] = bf2f(v.y >> 16);
        col[q][4] = bf2f(v.z & 0xffff); col[q][5] = bf2f(v.z >> 16); col[q][6] = bf2f(v.w & 0xffff); col[q][7] = bf2f(v.w >> 16);
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float* wp = wf + (ky * K + kx) * cpad + ch;
        const float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int o = 0; o < IDW_WO; ++o)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(col[o * S + kx][e], wv[e], acc[o][e]);
      }
    }
#pragma unroll
    for (int o = 0; o < IDW_WO; ++o) {
      if (ox0 + o < wo) {
        uint4 ov;
        ov.x = cvt_pk_bf16(fmaxf(acc[o][0], lo), fmaxf(acc[o][1], lo)); ov.y = cvt_pk_bf16(fmaxf(acc[o][2], lo), fmaxf(acc[o][3], lo));
        ov.z = cvt_pk_bf16(fmaxf(acc[o][4], lo), fmaxf(acc[o][5], lo)); ov.w = cvt_pk_bf16(fmaxf(acc[o][6], lo), fmaxf(acc[o][7], lo));
        *(uint4*)(y + (((int64_t)in * ho + oy) * wo + ox0 + o) * c + ch) = ov;
      }
    }
  }
}
extern "C" int frost_infer_dw(const uint16_t* x, const float* wf, const float* biasf, int n, int h, int w, int c, int k, int stride, int relu,
                              uint16_t* y, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "infer_dw: channels must be a multiple of 8");
  FROST_REQUIRE((k == 3 || k == 5) && (stride == 1 || stride == 2), "infer_dw: 3x3 / 5x5, stride 1 / 2");
  const int pad = (k - 1) / 2; const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
  const int64_t tot = (int64_t)n * ho * ((wo + IDW_WO - 1) / IDW_WO) * (c >> 3); int64_t grid = (tot + 255) / 256; if (grid > 16384) grid = 16384;
  hipStream_t s = as_stream(stream);
#define IDW_LAUNCH(K_, S_) hipLaunchKernelGGL((k_inf_dw<K_, S_>), dim3((unsigned)grid), dim3(256), 0, s, x, wf, biasf, n, h, w, c, round_up(c, 16), ho, wo, relu, y)
  if (k == 3 && stride == 1) IDW_LAUNCH(3, 1); else if (k == 3) IDW_LAUNCH(3, 2); else if (stride == 1) IDW_LAUNCH(5, 1); else IDW_LAUNCH(5, 2);
#undef IDW_LAUNCH
  return frost_check_launch("infer_dw");
}

// ------------------------------------------------------------------------------------------------ cat / add / pool
__global__ __launch_bounds__(256) void k_inf_cat(const uint16_t* __restrict__ a, int ca, const uint16_t* __restrict__ b, int cb, int64_t npix,
                                                 uint16_t* __restrict__ y) {
  const int cu = (ca + cb) >> 3; const int64_t tot = npix * cu;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int u = (int)(i % cu); const int64_t p = i / cu; const int ch = u * 8;
    const uint4 v = (ch < ca) ? *(const uint4*)(a + p * ca + ch) : *(const uint4*)(b + p * cb + (ch - ca));
    *(uint4*)(y + p * (ca + cb) + ch) = v;
  }
}
extern "C" int frost_infer_cat(const uint16_t* a, int ca, const uint16_t* b, int cb, int64_t npix, uint16_t* y, void* stream) {
  FROST_REQUIRE(ca % 8 == 0 && cb % 8 == 0, "infer_cat: channel counts must be multiples of 8");
  const int64_t tot = npix * ((ca + cb) >> 3); int64_t grid = (tot + 255) / 256; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_inf_cat, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a, ca, b, cb, npix, y);
  return frost_check_launch("infer_cat");
}
__global__ __launch_bounds__(256) void k_inf_add(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, int64_t n8, uint16_t* __restrict__ y) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const uint4 u = ((const uint4*)a)[i], v = ((const uint4*)b)[i]; uint4 o;
    o.x = cvt_pk_bf16(bf2f(u.x & 0xffff) + bf2f(v.x & 0xffff), bf2f(u.x >> 16) + bf2f(v.x >> 16));
    o.y = cvt_pk_bf16(bf2f(u.y & 0xffff) + bf2f(v.y & 0xffff), bf2f(u.y >> 16) + bf2f(v.y >> 16));
    o.z = cvt_pk_bf16(bf2f(u.z & 0xffff) + bf2f(v.z & 0xffff), bf2f(u.z >> 16) + bf2f(v.z >> 16));
    o.w = cvt_pk_bf16(bf2f(u.w & 0xffff) + bf2f(v.w & 0xffff), bf2f(u.w >> 16) + bf2f(v.w >> 16));
    ((uint4*)y)[i] = o;
  }
}
extern "C" int frost_infer_add(const uint16_t* a, const uint16_t* b, int64_t n, uint16_t* y, void* stream) {
  FROST_REQUIRE(n % 8 == 0, "infer_add: n must be a multiple of 8");
  int64_t grid = ((n >> 3) + 255) / 256; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_inf_add, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a, b, n >> 3, y);
  return frost_check_launch("infer_add");
}
// global average pool: x bf16 [n][hw][c] -> fp32 [n][c]
__global__ __launch_bounds__(256) void k_inf_avgpool(const uint16_t* __restrict__ x, int n, int hw, int c, float* __restrict__ y) {
  const int64_t tot = (int64_t)n * c;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c); const int in = (int)(i / c);
    float s = 0.0f;
    for (int p = 0; p < hw; ++p) s += bf2f(x[((int64_t)in * hw + p) * c + ch]);
    y[i] = s / (float)hw;
  }
}
extern "C" int frost_infer_avgpool(const uint16_t* x, int n, int hw, int c, float* y, void* stream) {
  const int64_t tot = (int64_t)n * c; int64_t grid = (tot + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_inf_avgpool, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, n, hw, c, y);
  return frost_check_launch("infer_avgpool");
}
