// bf16 inference of the FLOAT (not fake-quantised) FrostNet graph -- BASELINE.json config c2 (Large, B = 256, bf16).
// Eval-mode BatchNorm is folded into the conv (W' = W * gamma/sqrt(rv+eps), b' = beta - rm * gamma/sqrt(rv+eps)) once per call
// by frost_infer_weight_prep; activations are NHWC bf16; every conv accumulates in fp32 (bf16 MFMA 16x16x32 for the 1x1s and the
// im2col'd stem, fp32 FMA for the depthwise convs), adds the folded bias, applies the activation (ReLU, or hard-swish for act='hswish' networks) and rounds to bf16 once.
// replaces (eval mode, float model): frostnet.py:14-60 ConvBNReLU / ConvBN, :108-121 block wiring, :295-299 head.
#include "frost_common.h"

typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));

// activation of an inference epilogue: the `relu` argument is a code, 0 = none, 1 = ReLU, 2 = hard-swish (hswish_f, frost_common.h)
__device__ __forceinline__ float i_act(float z, float lo, bool hs) { return hs ? hswish_f(z) : fmaxf(z, lo); }

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

// ------------------------------------------------------------------------------------------------ stem without the im2col round trip
// conv1 of the network (frostnet.py:250: ConvBNReLU(3, 32, 3, 2)) straight from the fp32 image: the im2col path above writes and re-reads 128 bytes per output
// pixel (411 MB at B = 256) around a GEMM whose real input is 12 bytes per INPUT pixel.  Here a workgroup stages the input rows of a 4-row x 128-column output
// tile in LDS as bf16 [row][column][c0 c1 c2 0] (8 bytes per input pixel: one tap of the im2col row), a lane builds its MFMA B operand with two 8-byte LDS
// reads per K block, and the same two bf16 MFMAs per 16 x 16 tile as the GEMM path run on it (K index = tap * 4 + c, K blocks 0 and 1 in that order, then
// + bias, ReLU, one bf16 rounding) -- bit-identical to im2col + frost_infer_pw.  HBM traffic: the image once (+ 1/8 halo rows) and the output once.
#define IS_TH 4
#define IS_XT 8                                   // 16-pixel MFMA tiles along x per workgroup
#define IS_LW (IS_XT * 32 + 2)                    // staged columns: 2 * 128 + 1, rounded up to even
template <int CT>
__global__ __launch_bounds__(256) void k_inf_stem(const float* __restrict__ x, int h, int w, int ho, int wo, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                                  const uint16_t* __restrict__ pack, const float* __restrict__ biasf, int cout, int relu, int tiles_x,
                                                  int tiles_y, uint16_t* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) uint2 img[(2 * IS_TH + 1) * IS_LW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x; const int ty = b % tiles_y; const int in = b / tiles_y;
  const int oy0 = ty * IS_TH, ox0 = tx * IS_XT * 16;
  const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;                       // image coordinates of staged (row 0, column 0)
  const int ncol = min(IS_LW, (wo - ox0) * 2 + 1);                      // staged columns that a valid output pixel can touch
  const float* src = x + (int64_t)in * sn;
  // staging: a thread converts 2 adjacent columns x 3 channels per step (consecutive threads walk a row: coalesced for unit sw)
  const int cpairs = (ncol + 1) >> 1;
  for (int u = tid; u < (2 * IS_TH + 1) * cpairs; u += 256) {
    const int r = u / cpairs, cp = u - r * cpairs;
    const int iy = iy0 + r;
    float v[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (iy >= 0 && iy < h) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ix = ix0 + 2 * cp + q;
        if (ix >= 0 && ix < w) {
#pragma unroll
          for (int c = 0; c < 3; ++c) v[q][c] = src[c * sc + iy * sh + ix * sw];
        }
      }
    }
    uint4 o; o.x = cvt_pk_bf16(v[0][0], v[0][1]); o.y = cvt_pk_bf16(v[0][2], 0.0f); o.z = cvt_pk_bf16(v[1][0], v[1][1]); o.w = cvt_pk_bf16(v[1][2], 0.0f);
    *(uint4*)(img + r * IS_LW + 2 * cp) = o;
  }
  // A fragments (K blocks 0 and 1 of every channel tile) and this lane's biases
  uint4 af[CT][2]; float4 bb[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) af[ct][kb] = *(const uint4*)(pack + (((size_t)ct * 2 + kb) * 64 + lane) * 8);
    bb[ct] = *(const float4*)(biasf + ct * 16 + 4 * g);
  }
  // K block 0: k = 8 g .. 8 g + 7 = taps 2 g and 2 g + 1; K block 1: tap 8 in group 0, zero elsewhere
  const int t0 = 2 * g, t1 = 2 * g + 1;
  const int off0 = (t0 / 3) * IS_LW + t0 % 3, off1 = (t1 / 3) * IS_LW + t1 % 3, off8 = 2 * IS_LW + 2;
  const float flo = relu ? 0.0f : -INFINITY; const bool hs = relu == 2;
  __syncthreads();
  const int nxt = min(IS_XT, (wo - ox0 + 15) >> 4);
  uint16_t* dst = y + (int64_t)in * ho * wo * cout;
  for (int t = wv; t < IS_TH * nxt; t += 4) {
    const int r = t / nxt, xt = t - r * nxt;
    const int oy = oy0 + r, ox = ox0 + xt * 16 + j;
    const uint2* bp = img + (2 * r) * IS_LW + (xt * 16 + j) * 2;
    const uint2 a0 = bp[off0], a1 = bp[off1];
    uint2 a8 = make_uint2(0, 0);
    if (g == 0) a8 = bp[off8];
    const uint4 b0 = make_uint4(a0.x, a0.y, a1.x, a1.y), b1 = make_uint4(a8.x, a8.y, 0, 0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf16, af[ct][0]), __builtin_bit_cast(v8bf16, b0), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf16, af[ct][1]), __builtin_bit_cast(v8bf16, b1), acc, 0, 0, 0);
      const int ch = ct * 16 + 4 * g;
      if (oy < ho && ox < wo && ch < cout) {
        uint2 o;
        o.x = cvt_pk_bf16(i_act(acc[0] + bb[ct].x, flo, hs), i_act(acc[1] + bb[ct].y, flo, hs));
        o.y = cvt_pk_bf16(i_act(acc[2] + bb[ct].z, flo, hs), i_act(acc[3] + bb[ct].w, flo, hs));
        *(uint2*)(dst + ((int64_t)oy * wo + ox) * cout + ch) = o;
      }
    }
  }
}
extern "C" int frost_infer_stem_ok(int cout) { return (cout % 8 == 0 && cout >= 8 && cout <= 64) ? 1 : 0; }
/* The stem conv (3 x 3, stride 2, pad 1, 3 input channels) of the bf16 inference graph in one launch.  x: logical (N,3,H,W) fp32 with element strides
 * (sn, sc, sh, sw); pack / biasf: the stem's FrostIDesc.pack (kind 2: K index = tap * 4 + c, kpad = 64) and folded bias; y: [N][ho][wo][cout] bf16. */
extern "C" int frost_infer_stem(const float* x, int n, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const uint16_t* pack,
                                const float* biasf, int cout, int relu, uint16_t* y, void* stream) {
  FROST_REQUIRE(frost_infer_stem_ok(cout), "infer_stem: cout must be a multiple of 8 in 8..64");
  FROST_REQUIRE(x && pack && biasf && y && n > 0 && h > 0 && w > 0, "infer_stem: incomplete arguments");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int tiles_x = (wo + IS_XT * 16 - 1) / (IS_XT * 16), tiles_y = (ho + IS_TH - 1) / IS_TH;
  const dim3 grid((unsigned)((int64_t)n * tiles_x * tiles_y));
  hipStream_t s = as_stream(stream);
  const int ct = (cout + 15) / 16;
#define IS_GO(C) hipLaunchKernelGGL((k_inf_stem<C>), grid, dim3(256), 0, s, x, h, w, ho, wo, sn, sc, sh, sw, pack, biasf, cout, relu, tiles_x, tiles_y, y)
  if (ct == 1) IS_GO(1); else if (ct == 2) IS_GO(2); else if (ct == 3) IS_GO(3); else IS_GO(4);
#undef IS_GO
  return frost_check_launch("infer_stem");
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
  const float lo = relu ? 0.0f : -INFINITY; const bool hs = relu == 2;
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
        ov.x = cvt_pk_bf16(i_act(acc[o][0], lo, hs), i_act(acc[o][1], lo, hs)); ov.y = cvt_pk_bf16(i_act(acc[o][2], lo, hs), i_act(acc[o][3], lo, hs));
        ov.z = cvt_pk_bf16(i_act(acc[o][4], lo, hs), i_act(acc[o][5], lo, hs)); ov.w = cvt_pk_bf16(i_act(acc[o][6], lo, hs), i_act(acc[o][7], lo, hs));
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
