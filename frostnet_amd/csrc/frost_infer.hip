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

// ------------------------------------------------------------------------------------------------ pointwise (bf16 MFMA GEMM)
// y[p][co] = act( sum_k T[p][k] * W'[co][k] + b'[co] ).  64-pixel tile per workgroup staged once in LDS (coalesced 16-byte loads), each of
// the 4 waves owns 16 pixels and walks the channel tiles four at a time; weight fragments are 1 KiB wave-loads from the packed (L2-resident) matrix.
// WPX = waves along pixels (4: 64-pixel tile, every wave all channels; 1: 16-pixel tile shared by the 4 waves, which split the channel
// tiles -- rows too long for a 64-row LDS tile, i.e. the 7x7 layers with 720..1728 input channels)
template <int WPX>
__global__ __launch_bounds__(256) void k_inf_pw(const uint16_t* __restrict__ T, const uint16_t* __restrict__ pack, const float* __restrict__ biasf,
                                                int64_t npix, int cin, int cout, int cpad, int KB, int kstr, int relu, uint16_t* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int IPX = 16 * WPX, WCH = 4 / WPX;
  const int wpx = w % WPX, wch = w / WPX;
  const int64_t p0 = (int64_t)blockIdx.x * IPX;
  const int rowb = cin * 2; const int U = (KB * 64) >> 4;          // 16-byte units per (K-padded) row
  for (int u = tid; u < IPX * U; u += 256) {
    const int row = u / U, col = (u - row * U) << 4;
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((p0 + row) < npix && col < rowb) v = *(const uint4*)((const uint8_t*)T + (p0 + row) * rowb + col);
    *(uint4*)(smem + row * kstr + col) = v;
  }
  __syncthreads();
  const int CT = cpad >> 4;
  const int64_t prow = p0 + wpx * 16 + j;
  const float lo = relu ? 0.0f : -INFINITY;
  for (int ct0 = wch * 4; ct0 < CT; ct0 += 4 * WCH) {
    v4f acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < KB; ++kb) {
      const v4i bfr = *(const v4i*)(smem + (wpx * 16 + j) * kstr + kb * 64 + g * 16);
      v4i afr[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) if (ct0 + m < CT) afr[m] = *(const v4i*)(pack + ((((int64_t)(ct0 + m) * KB + kb) * 64 + lane) << 3));
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (ct0 + m < CT) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf16, afr[m]), __builtin_bit_cast(v8bf16, bfr), acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int ch0 = (ct0 + m) * 16 + 4 * g;
      if (ct0 + m < CT && ch0 < cout && prow < npix) {
        const float4 b4 = *(const float4*)(biasf + ch0);
        uint2 o; o.x = cvt_pk_bf16(fmaxf(acc[m][0] + b4.x, lo), fmaxf(acc[m][1] + b4.y, lo)); o.y = cvt_pk_bf16(fmaxf(acc[m][2] + b4.z, lo), fmaxf(acc[m][3] + b4.w, lo));
        *(uint2*)(y + prow * cout + ch0) = o;
      }
    }
  }
}
extern "C" int frost_infer_pw(const uint16_t* x, const uint16_t* pack, const float* biasf, int64_t npix, int cin, int cout, int relu,
                              uint16_t* y, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 4 == 0, "infer_pw: cin must be a multiple of 8, cout of 4");
  const int KB = (cin + 31) / 32; const int kstr = KB * 64 + 16; const int cpad = round_up(cout, 16);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)k_inf_pw<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_inf_pw<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if ((size_t)64 * kstr <= 64 * 1024)
    hipLaunchKernelGGL(k_inf_pw<4>, dim3((unsigned)((npix + 63) / 64)), dim3(256), (size_t)64 * kstr, as_stream(stream), x, pack, biasf, npix, cin, cout,
                       cpad, KB, kstr, relu, y);
  else {
    FROST_REQUIRE((size_t)16 * kstr <= 160 * 1024, "infer_pw: row too long for the LDS tile");
    hipLaunchKernelGGL(k_inf_pw<1>, dim3((unsigned)((npix + 15) / 16)), dim3(256), (size_t)16 * kstr, as_stream(stream), x, pack, biasf, npix, cin, cout,
                       cpad, KB, kstr, relu, y);
  }
  return frost_check_launch("infer_pw");
}

// ------------------------------------------------------------------------------------------------ depthwise (fp32 FMA)
// one thread = one output pixel x 8 channels: k*k 16-byte loads (NHWC: the 8 channels are contiguous), fp32 accumulate
__global__ __launch_bounds__(256) void k_inf_dw(const uint16_t* __restrict__ x, const float* __restrict__ wf, const float* __restrict__ biasf, int n,
                                                int h, int w, int c, int cpad, int k, int stride, int ho, int wo, int relu, uint16_t* __restrict__ y) {
  const int c8n = c >> 3; const int pad = (k - 1) / 2;
  const int64_t tot = (int64_t)n * ho * wo * c8n;
  const float lo = relu ? 0.0f : -INFINITY;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % c8n); int64_t p = i / c8n; const int ox = (int)(p % wo); p /= wo; const int oy = (int)(p % ho); const int in = (int)(p / ho);
    const int ch = c8 * 8;
    float acc[8];
    { const float4 b0 = *(const float4*)(biasf + ch), b1 = *(const float4*)(biasf + ch + 4); acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w; }
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * stride - pad + ky; if (iy < 0 || iy >= h) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * stride - pad + kx; if (ix < 0 || ix >= w) continue;
        const uint4 v = *(const uint4*)(x + (((int64_t)in * h + iy) * w + ix) * c + ch);
        const float* wp = wf + (ky * k + kx) * cpad + ch;
        const float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
        acc[0] = fmaf(bf2f(v.x & 0xffff), w0.x, acc[0]); acc[1] = fmaf(bf2f(v.x >> 16), w0.y, acc[1]);
        acc[2] = fmaf(bf2f(v.y & 0xffff), w0.z, acc[2]); acc[3] = fmaf(bf2f(v.y >> 16), w0.w, acc[3]);
        acc[4] = fmaf(bf2f(v.z & 0xffff), w1.x, acc[4]); acc[5] = fmaf(bf2f(v.z >> 16), w1.y, acc[5]);
        acc[6] = fmaf(bf2f(v.w & 0xffff), w1.z, acc[6]); acc[7] = fmaf(bf2f(v.w >> 16), w1.w, acc[7]);
      }
    }
    uint4 o;
    o.x = cvt_pk_bf16(fmaxf(acc[0], lo), fmaxf(acc[1], lo)); o.y = cvt_pk_bf16(fmaxf(acc[2], lo), fmaxf(acc[3], lo));
    o.z = cvt_pk_bf16(fmaxf(acc[4], lo), fmaxf(acc[5], lo)); o.w = cvt_pk_bf16(fmaxf(acc[6], lo), fmaxf(acc[7], lo));
    *(uint4*)(y + (((int64_t)in * ho + oy) * wo + ox) * c + ch) = o;
  }
}
extern "C" int frost_infer_dw(const uint16_t* x, const float* wf, const float* biasf, int n, int h, int w, int c, int k, int stride, int relu,
                              uint16_t* y, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "infer_dw: channels must be a multiple of 8");
  const int pad = (k - 1) / 2; const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
  const int64_t tot = (int64_t)n * ho * wo * (c >> 3); int64_t grid = (tot + 255) / 256; if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(k_inf_dw, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, wf, biasf, n, h, w, c, round_up(c, 16), k, stride, ho, wo,
                     relu, y);
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
