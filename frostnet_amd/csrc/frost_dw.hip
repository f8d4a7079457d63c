// Depthwise k x k (k in {3,5}, stride in {1,2}) and the dense 3x3/s2 stem on fake-quantised operands.
// LDS-tiled direct convolutions on NHWC bytes: a workgroup stages an input halo tile for a 64-channel block
// with coalesced 8-byte loads, every thread owns 4 consecutive channels (one dword) x one output column x 8
// output rows and slides down the tile converting each input dword once (v_cvt_f32_ubyte) and FMA-ing it into
// every output row it touches.  All sums are exact integers in fp32 (|sum| <= 25*255*128 < 2^24).
// Zero padding = zero-point fill, so acc_true = sum(w*q) - zp*sum(w) holds at borders too.
#include "frost_common.h"

enum { D_STATS = 0, D_EMIT = 1, D_BRED = 2, D_BDC = 3 };
#define TH 8
#define TW 16
#define CB 64

struct DwP {
  const int8_t* x; const float* qx; const int8_t* wq; const int32_t* wsum;
  int n, h, w, c, cpad, ho, wo, pad;
  uint8_t* stats; float* coef; const float* qy; int relu; int8_t* y;
  const uint16_t* gout; uint16_t* dc;
  int tiles_x, tiles_y, ncb, ngroups; int64_t ntiles; float inv_count;
};

__device__ __forceinline__ void unpack4(uint32_t v, float* f) {   // offset-binary bytes -> unsigned index floats
  v ^= 0x80808080u;
  f[0] = (float)(v & 255u); f[1] = (float)((v >> 8) & 255u); f[2] = (float)((v >> 16) & 255u); f[3] = (float)(v >> 24);
}
__device__ __forceinline__ void unpack4s(uint32_t v, float* f) {  // signed int8 -> floats
  f[0] = (float)(int8_t)(v & 255u); f[1] = (float)(int8_t)((v >> 8) & 255u); f[2] = (float)(int8_t)((v >> 16) & 255u); f[3] = (float)(int8_t)(v >> 24);
}

template <int K, int S, int MODE>
__global__ __launch_bounds__(256) void k_dw(const DwP p) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ double red_d[4][16][8]; __shared__ float red_f[4][16][8];
  const int tid = threadIdx.x;
  const int cq = tid & 15, ox = tid >> 4;
  const int cb = blockIdx.x % p.ncb, grp = blockIdx.x / p.ncb;
  const int c0 = cb * CB + cq * 4;
  const bool chok = c0 < p.c;
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]);
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;

  // hoisted weights (floats) and per-channel constants
  float wf[K * K][4];
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    uint32_t v = chok ? *(const uint32_t*)(p.wq + t * p.cpad + c0) : 0u;
    unpack4s(v, wf[t]);
  }
  float corr[4] = {0, 0, 0, 0};
  if (chok) { int4 ws = *(const int4*)(p.wsum + c0); corr[0] = (float)(zp * ws.x); corr[1] = (float)(zp * ws.y); corr[2] = (float)(zp * ws.z); corr[3] = (float)(zp * ws.w); }
  float A[4] = {0, 0, 0, 0}, B[4] = {0, 0, 0, 0}, Mv[4] = {0, 0, 0, 0}, Rv[4] = {0, 0, 0, 0}, K1[4] = {0, 0, 0, 0}, S1[4] = {0, 0, 0, 0}, S2[4] = {0, 0, 0, 0};
  float y_inv = 1.0f; int y_zp = 0;
  if (MODE != D_STATS) {
    y_inv = 1.0f / p.qy[FROST_Q_SCALE]; y_zp = __float_as_int(p.qy[FROST_Q_ZP]);
    if (chok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        A[r] = p.coef[FROST_COEF_A * p.cpad + c0 + r]; B[r] = p.coef[FROST_COEF_B * p.cpad + c0 + r];
        Mv[r] = p.coef[FROST_COEF_M * p.cpad + c0 + r]; Rv[r] = p.coef[FROST_COEF_R * p.cpad + c0 + r];
        if (MODE == D_BDC) {
          K1[r] = p.coef[FROST_COEF_K1 * p.cpad + c0 + r];
          S1[r] = p.coef[FROST_COEF_S1 * p.cpad + c0 + r] * p.inv_count; S2[r] = p.coef[FROST_COEF_S2 * p.cpad + c0 + r] * p.inv_count;
        }
      }
    }
  }
  double st1[4] = {0, 0, 0, 0}, st2[4] = {0, 0, 0, 0}; float smn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, smx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  float r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};

  const int tiles_per_img = p.tiles_x * p.tiles_y;
  for (int64_t tile = grp; tile < p.ntiles; tile += p.ngroups) {
    const int img = (int)(tile / tiles_per_img); const int tr = (int)(tile - (int64_t)img * tiles_per_img);
    const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    __syncthreads();
    // stage halo tile: [IH][IW][64B], 8-byte units
    for (int u = tid; u < IH * IW * 8; u += 256) {
      const int c8 = u & 7; const int pix = u >> 3; const int iy = pix / IW, ix = pix - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix; const int cc = cb * CB + c8 * 8;
      uint2 v = make_uint2(zfill, zfill);
      if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w && cc < p.c)
        v = *(const uint2*)(p.x + (((int64_t)img * p.h + gy) * p.w + gx) * p.c + cc);
      *(uint2*)(smem + pix * CB + c8 * 8) = v;
    }
    __syncthreads();
    float acc[TH][4];
#pragma unroll
    for (int o = 0; o < TH; ++o) { acc[o][0] = 0; acc[o][1] = 0; acc[o][2] = 0; acc[o][3] = 0; }
#pragma unroll
    for (int iy = 0; iy < IH; ++iy) {
      float xf[K][4];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) unpack4(*(const uint32_t*)(smem + (iy * IW + ox * S + kx) * CB + cq * 4), xf[kx]);
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        if ((iy - ky) >= 0 && ((iy - ky) % S) == 0 && (iy - ky) / S < TH) {
          const int o = (iy - ky) / S;
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[o][r] = fmaf(xf[kx][r], wf[ky * K + kx][r], acc[o][r]);
        }
      }
    }
    // epilogue
#pragma unroll
    for (int o = 0; o < TH; ++o) {
      const int oy = oy0 + o, oxx = ox0 + ox;
      const bool valid = chok && oy < p.ho && oxx < p.wo;
      const int64_t opix = ((int64_t)img * p.ho + oy) * p.wo + oxx;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[o][r] - corr[r];
      if (MODE == D_STATS) {
        if (valid) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { st1[r] += (double)v[r]; st2[r] += (double)v[r] * (double)v[r]; smn[r] = fminf(smn[r], v[r]); smx[r] = fmaxf(smx[r], v[r]); }
        }
      } else if (MODE == D_EMIT) {
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float yv = fmaf(A[r], v[r], B[r]);
          if (p.relu) yv = fmaxf(yv, 0.0f);
          packed |= ((uint32_t)((fq_index(yv, y_inv, y_zp, 0, 255) - 128) & 255)) << (8 * r);
        }
        if (valid) *(uint32_t*)(p.y + opix * p.c + c0) = packed;
      } else {
        uint2 gv = make_uint2(0, 0);
        if (valid) gv = *(const uint2*)(p.gout + opix * p.c + c0);
        const float gq[4] = {bf2f(gv.x & 0xffff), bf2f(gv.x >> 16), bf2f(gv.y & 0xffff), bf2f(gv.y >> 16)};
        float dcv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float yv = fmaf(A[r], v[r], B[r]);
          bool alive = true;
          if (p.relu) { alive = yv > 0.0f; yv = fmaxf(yv, 0.0f); }
          bool inr; fq_index(yv, y_inv, y_zp, 0, 255, &inr);
          const float gyv = (alive && inr && valid) ? gq[r] : 0.0f;
          const float xhat = (v[r] - Mv[r]) * Rv[r];
          if (MODE == D_BRED) { r1[r] += gyv; r2[r] += gyv * xhat; }
          else dcv[r] = K1[r] * (gyv - S1[r] - xhat * S2[r]);
        }
        if (MODE == D_BDC && valid) {
          uint2 ov; ov.x = (uint32_t)f2bf(dcv[0]) | ((uint32_t)f2bf(dcv[1]) << 16); ov.y = (uint32_t)f2bf(dcv[2]) | ((uint32_t)f2bf(dcv[3]) << 16);
          *(uint2*)(p.dc + opix * p.c + c0) = ov;
        }
      }
    }
  }

  // cross-thread reduction over the 16 threads (ox) that share a channel quad: lanes g=(ox&3), waves (ox>>2)
  if (MODE == D_STATS || MODE == D_BRED) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == D_STATS) {
        double a = st1[r], b = st2[r]; float c = smn[r], d = smx[r];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c = fminf(c, __shfl_xor(c, o)); d = fmaxf(d, __shfl_xor(d, o)); }
        if (lane < 16) { red_d[wv][lane][r] = a; red_d[wv][lane][4 + r] = b; red_f[wv][lane][r] = c; red_f[wv][lane][4 + r] = d; }
      } else {
        float a = r1[r], b = r2[r];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (lane < 16) { red_f[wv][lane][r] = a; red_f[wv][lane][4 + r] = b; }
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int q = tid >> 2, r = tid & 3; const int ch = cb * CB + q * 4 + r;
      if (ch < p.c) {
        if (MODE == D_STATS) {
          double a = 0, b = 0; float c = INFINITY, d = -INFINITY;
          for (int wv2 = 0; wv2 < 4; ++wv2) { a += red_d[wv2][q][r]; b += red_d[wv2][q][4 + r]; c = fminf(c, red_f[wv2][q][r]); d = fmaxf(d, red_f[wv2][q][4 + r]); }
          int64_t* g_s1 = (int64_t*)p.stats; unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
          int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
          if (c <= d) {
            atomicAdd((unsigned long long*)&g_s1[ch], (unsigned long long)(long long)a);
            atomicAdd(&g_s2[ch], (unsigned long long)b);
            atomicMin(&g_mn[ch], (int)c); atomicMax(&g_mx[ch], (int)d);
          }
        } else {
          float a = 0, b = 0;
          for (int wv2 = 0; wv2 < 4; ++wv2) { a += red_f[wv2][q][r]; b += red_f[wv2][q][4 + r]; }
          atomicAdd(p.coef + FROST_COEF_S1 * p.cpad + ch, a); atomicAdd(p.coef + FROST_COEF_S2 * p.cpad + ch, b);
        }
      }
    }
  }
}

template <int K, int S, int MODE>
static int launch_dw(DwP& p, hipStream_t s) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  size_t lds = (size_t)IH * IW * CB;
  int64_t want = 1024 / p.ncb; if (want < 1) want = 1;
  p.ngroups = (int)(p.ntiles < want ? p.ntiles : want);
  hipLaunchKernelGGL((k_dw<K, S, MODE>), dim3(p.ncb * p.ngroups), dim3(256), lds, s, p);
  return frost_check_launch("dw");
}
template <int MODE>
static int dispatch_dw(DwP& p, int k, int stride, hipStream_t s) {
  if (k == 3 && stride == 1) return launch_dw<3, 1, MODE>(p, s);
  if (k == 3 && stride == 2) return launch_dw<3, 2, MODE>(p, s);
  if (k == 5 && stride == 1) return launch_dw<5, 1, MODE>(p, s);
  if (k == 5 && stride == 2) return launch_dw<5, 2, MODE>(p, s);
  frost_set_error("dw: unsupported kernel/stride (k in {3,5}, stride in {1,2})");
  return 1;
}
static void fill_dw(DwP& p, const int8_t* x, const float* qx, const int8_t* wq, const int32_t* wsum, int n, int h, int w,
                    int c, int k, int stride) {
  p.x = x; p.qx = qx; p.wq = wq; p.wsum = wsum; p.n = n; p.h = h; p.w = w; p.c = c; p.cpad = round_up(c, 16);
  p.pad = (k - 1) / 2; p.ho = (h + 2 * p.pad - k) / stride + 1; p.wo = (w + 2 * p.pad - k) / stride + 1;
  p.tiles_x = (p.wo + TW - 1) / TW; p.tiles_y = (p.ho + TH - 1) / TH; p.ncb = (c + CB - 1) / CB;
  p.ntiles = (int64_t)n * p.tiles_x * p.tiles_y; p.inv_count = 1.0f / (float)((int64_t)n * p.ho * p.wo);
}
extern "C" int frost_dw_conv_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n,
                                 int h, int w, int c, int k, int stride, int mode, void* stats, const float* coef,
                                 const float* qrec_y, int relu, int8_t* y, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  DwP p = {}; fill_dw(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.stats = (uint8_t*)stats; p.coef = (float*)coef; p.qy = qrec_y; p.relu = relu; p.y = y;
  return mode == 0 ? dispatch_dw<D_STATS>(p, k, stride, as_stream(stream)) : dispatch_dw<D_EMIT>(p, k, stride, as_stream(stream));
}
extern "C" int frost_dw_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                                 const float* qrec_w, int n, int h, int w, int c, int k, int stride, int pass, float* coef,
                                 const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  DwP p = {}; fill_dw(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.coef = coef; p.qy = qrec_y; p.relu = relu; p.gout = gout; p.dc = dc;
  return pass == 0 ? dispatch_dw<D_BRED>(p, k, stride, as_stream(stream)) : dispatch_dw<D_BDC>(p, k, stride, as_stream(stream));
}

// ---- depthwise dgrad: dx[n][iy][ix][c] (+)= s_w * sum_taps dc[n][oy][ox][c] * wq[ky][kx][c]  (gather form)
__global__ __launch_bounds__(256) void k_dw_dgrad(const uint16_t* __restrict__ dc, const int8_t* __restrict__ wq, const float* qw,
                                                  int n, int h, int w, int c, int cpad, int k, int stride, int ho, int wo,
                                                  uint16_t* __restrict__ dx, int accumulate) {
  const int pad = (k - 1) / 2; const float sw = qw[FROST_Q_SCALE];
  const int cq_n = c >> 2; const int64_t tot = (int64_t)n * h * w * cq_n;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % cq_n); int64_t t = i / cq_n; const int ix = (int)(t % w); t /= w; const int iy = (int)(t % h); const int img = (int)(t / h);
    const int c0 = cq * 4;
    float a[4] = {0, 0, 0, 0};
    for (int ky = 0; ky < k; ++ky) {
      const int ty = iy + pad - ky; if (ty < 0 || (ty % stride) != 0) continue; const int oy = ty / stride; if (oy >= ho) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int tx = ix + pad - kx; if (tx < 0 || (tx % stride) != 0) continue; const int ox = tx / stride; if (ox >= wo) continue;
        const uint2 gv = *(const uint2*)(dc + (((int64_t)img * ho + oy) * wo + ox) * c + c0);
        float wv[4]; unpack4s(*(const uint32_t*)(wq + (ky * k + kx) * cpad + c0), wv);
        a[0] = fmaf(bf2f(gv.x & 0xffff), wv[0], a[0]); a[1] = fmaf(bf2f(gv.x >> 16), wv[1], a[1]);
        a[2] = fmaf(bf2f(gv.y & 0xffff), wv[2], a[2]); a[3] = fmaf(bf2f(gv.y >> 16), wv[3], a[3]);
      }
    }
    uint16_t* dst = dx + (((int64_t)img * h + iy) * w + ix) * c + c0;
    float v[4] = {a[0] * sw, a[1] * sw, a[2] * sw, a[3] * sw};
    if (accumulate) { uint2 o = *(const uint2*)dst; v[0] += bf2f(o.x & 0xffff); v[1] += bf2f(o.x >> 16); v[2] += bf2f(o.y & 0xffff); v[3] += bf2f(o.y >> 16); }
    uint2 o; o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *(uint2*)dst = o;
  }
}
extern "C" int frost_dw_dgrad(const uint16_t* dc, const int8_t* wq_pack, const float* qrec_w, int n, int h, int w, int c,
                              int k, int stride, uint16_t* dx, int accumulate, void* stream) {
  const int pad = (k - 1) / 2; const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
  int64_t tot = (int64_t)n * h * w * (c / 4); int64_t grid = (tot + 255) / 256; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_dw_dgrad, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), dc, wq_pack, qrec_w, n, h, w, c,
                     round_up(c, 16), k, stride, ho, wo, dx, accumulate);
  return frost_check_launch("dw_dgrad");
}

// ---- depthwise wgrad: dwq[c][ky][kx] += s_x * sum_{n,oy,ox} dc * (q - zp)     (dwq fp32 [c][k*k], pre-zeroed)
template <int K>
__global__ __launch_bounds__(256) void k_dw_wgrad(const uint16_t* __restrict__ dc, const int8_t* __restrict__ x, const float* qx,
                                                  int n, int h, int w, int c, int stride, int ho, int wo, float* dwq, int ngroups) {
  constexpr int pad = (K - 1) / 2;
  const int cq = threadIdx.x & 15; const int pl = threadIdx.x >> 4;   // 16 pixel lanes
  const int ncb = (c + 63) / 64; const int cb = blockIdx.x % ncb, grp = blockIdx.x / ncb;
  const int c0 = cb * 64 + cq * 4; const bool chok = c0 < c;
  const int zp = __float_as_int(qx[FROST_Q_ZP]); const float sx = qx[FROST_Q_SCALE];
  float acc[K * K][4];
#pragma unroll
  for (int t = 0; t < K * K; ++t) { acc[t][0] = 0; acc[t][1] = 0; acc[t][2] = 0; acc[t][3] = 0; }
  const int64_t npix = (int64_t)n * ho * wo;
  if (chok) {
    for (int64_t pi = (int64_t)grp * 16 + pl; pi < npix; pi += (int64_t)ngroups * 16) {
      const int ox = (int)(pi % wo); int64_t t = pi / wo; const int oy = (int)(t % ho); const int img = (int)(t / ho);
      const uint2 gv = *(const uint2*)(dc + pi * c + c0);
      const float gq[4] = {bf2f(gv.x & 0xffff), bf2f(gv.x >> 16), bf2f(gv.y & 0xffff), bf2f(gv.y >> 16)};
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * stride - pad + ky; if (iy < 0 || iy >= h) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int ix = ox * stride - pad + kx; if (ix < 0 || ix >= w) continue;
          float xf[4]; unpack4(*(const uint32_t*)(x + (((int64_t)img * h + iy) * w + ix) * c + c0), xf);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ky * K + kx][r] = fmaf(gq[r], xf[r] - (float)zp, acc[ky * K + kx][r]);
        }
      }
    }
  }
  __shared__ float red[K * K][64];
  for (int i = threadIdx.x; i < K * K * 64; i += 256) ((float*)red)[i] = 0.0f;
  __syncthreads();
  if (chok) {
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[t][cq * 4 + r], acc[t][r]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K * 64; i += 256) {
    const int t = i / 64, cc = i % 64; const int ch = cb * 64 + cc;
    if (ch < c) atomicAdd(dwq + (int64_t)ch * K * K + t, red[t][cc] * sx);
  }
}
extern "C" int frost_dw_wgrad(const uint16_t* dc, const int8_t* x, const float* qrec_x, int n, int h, int w, int c, int k,
                              int stride, float* dwq, void* stream) {
  const int pad = (k - 1) / 2; const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
  const int ncb = (c + 63) / 64; int ngroups = 1024 / ncb; if (ngroups < 1) ngroups = 1;
  int64_t npix = (int64_t)n * ho * wo; if (ngroups > (npix + 15) / 16) ngroups = (int)((npix + 15) / 16);
  if (k == 3) hipLaunchKernelGGL(k_dw_wgrad<3>, dim3(ncb * ngroups), dim3(256), 0, as_stream(stream), dc, x, qrec_x, n, h, w, c, stride, ho, wo, dwq, ngroups);
  else if (k == 5) hipLaunchKernelGGL(k_dw_wgrad<5>, dim3(ncb * ngroups), dim3(256), 0, as_stream(stream), dc, x, qrec_x, n, h, w, c, stride, ho, wo, dwq, ngroups);
  else { frost_set_error("dw_wgrad: k must be 3 or 5"); return 1; }
  return frost_check_launch("dw_wgrad");
}

// ---------------------------------------------------------------------------------------------------- stem
// dense 3x3 stride-2 pad-1 conv, input NHWC with 4 bytes per pixel (3 channels + pad), cout <= 32.
// thread = (output pixel, 4-channel quad); weights hoisted as floats from the plain [cout][3][3][3]-ordered
// pack written by weight-prep kind 2?  -> here we use a simple [tap][c][cpad] int8 pack (kind 4) for VALU.
struct StemP {
  const int8_t* x; const float* qx; const int8_t* wq; const int32_t* wsum; int n, h, w, cout, cpad, ho, wo;
  uint8_t* stats; float* coef; const float* qy; int relu; int8_t* y; const uint16_t* gout; float* dwq; float inv_count;
};
template <int MODE>
__global__ __launch_bounds__(256) void k_stem(const StemP p) {
  const int nq = p.cout >> 2; const int cq = threadIdx.x % nq; const int pl = threadIdx.x / nq; const int npl = 256 / nq;
  const int c0 = cq * 4; const bool active = pl < npl;
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]);
  float wf[27][4];
#pragma unroll
  for (int t = 0; t < 27; ++t) unpack4s(*(const uint32_t*)(p.wq + t * p.cpad + c0), wf[t]);   // t = tap*3 + c
  float corr[4]; { int4 ws = *(const int4*)(p.wsum + c0); corr[0] = (float)(zp * ws.x); corr[1] = (float)(zp * ws.y); corr[2] = (float)(zp * ws.z); corr[3] = (float)(zp * ws.w); }
  float A[4] = {0, 0, 0, 0}, B[4] = {0, 0, 0, 0}, Mv[4] = {0, 0, 0, 0}, Rv[4] = {0, 0, 0, 0}, K1[4] = {0, 0, 0, 0}, S1[4] = {0, 0, 0, 0}, S2[4] = {0, 0, 0, 0};
  float y_inv = 1.0f; int y_zp = 0;
  if (MODE != D_STATS) {
    y_inv = 1.0f / p.qy[FROST_Q_SCALE]; y_zp = __float_as_int(p.qy[FROST_Q_ZP]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      A[r] = p.coef[FROST_COEF_A * p.cpad + c0 + r]; B[r] = p.coef[FROST_COEF_B * p.cpad + c0 + r];
      Mv[r] = p.coef[FROST_COEF_M * p.cpad + c0 + r]; Rv[r] = p.coef[FROST_COEF_R * p.cpad + c0 + r];
      if (MODE == D_BDC) { K1[r] = p.coef[FROST_COEF_K1 * p.cpad + c0 + r]; S1[r] = p.coef[FROST_COEF_S1 * p.cpad + c0 + r] * p.inv_count; S2[r] = p.coef[FROST_COEF_S2 * p.cpad + c0 + r] * p.inv_count; }
    }
  }
  double st1[4] = {0, 0, 0, 0}, st2[4] = {0, 0, 0, 0}; float smn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, smx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  float r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
  float dw[27][4];
  if (MODE == D_BDC) {
#pragma unroll
    for (int t = 0; t < 27; ++t) { dw[t][0] = 0; dw[t][1] = 0; dw[t][2] = 0; dw[t][3] = 0; }
  }
  const int64_t npix = (int64_t)p.n * p.ho * p.wo;
  if (active) {
    for (int64_t pi = (int64_t)blockIdx.x * npl + pl; pi < npix; pi += (int64_t)gridDim.x * npl) {
      const int ox = (int)(pi % p.wo); int64_t t = pi / p.wo; const int oy = (int)(t % p.ho); const int img = (int)(t / p.ho);
      float acc[4] = {0, 0, 0, 0}; float xin[9][3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
          float xf[4] = {(float)zp, (float)zp, (float)zp, (float)zp};
          if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) unpack4(*(const uint32_t*)(p.x + (((int64_t)img * p.h + iy) * p.w + ix) * 4), xf);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            xin[ky * 3 + kx][c] = xf[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(xf[c], wf[(ky * 3 + kx) * 3 + c][r], acc[r]);
          }
        }
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[r] - corr[r];
      if (MODE == D_STATS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { st1[r] += (double)v[r]; st2[r] += (double)v[r] * (double)v[r]; smn[r] = fminf(smn[r], v[r]); smx[r] = fmaxf(smx[r], v[r]); }
      } else if (MODE == D_EMIT) {
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float yv = fmaf(A[r], v[r], B[r]); if (p.relu) yv = fmaxf(yv, 0.0f);
          packed |= ((uint32_t)((fq_index(yv, y_inv, y_zp, 0, 255) - 128) & 255)) << (8 * r);
        }
        *(uint32_t*)(p.y + pi * p.cout + c0) = packed;
      } else {
        const uint2 gv = *(const uint2*)(p.gout + pi * p.cout + c0);
        const float gq[4] = {bf2f(gv.x & 0xffff), bf2f(gv.x >> 16), bf2f(gv.y & 0xffff), bf2f(gv.y >> 16)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float yv = fmaf(A[r], v[r], B[r]); bool alive = true;
          if (p.relu) { alive = yv > 0.0f; yv = fmaxf(yv, 0.0f); }
          bool inr; fq_index(yv, y_inv, y_zp, 0, 255, &inr);
          const float gyv = (alive && inr) ? gq[r] : 0.0f;
          const float xhat = (v[r] - Mv[r]) * Rv[r];
          if (MODE == D_BRED) { r1[r] += gyv; r2[r] += gyv * xhat; }
          else {
            const float dcv = K1[r] * (gyv - S1[r] - xhat * S2[r]);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
              for (int c = 0; c < 3; ++c) dw[t9 * 3 + c][r] = fmaf(dcv, xin[t9][c] - (float)zp, dw[t9 * 3 + c][r]);
          }
        }
      }
    }
  }
  // block reduction through LDS atomics (few, once per block)
  __shared__ double sd[2][32]; __shared__ float sf[2][32]; __shared__ float sdw[27][32];
  for (int i = threadIdx.x; i < 32; i += 256) { sd[0][i] = 0; sd[1][i] = 0; sf[0][i] = (MODE == D_STATS) ? INFINITY : 0.0f; sf[1][i] = (MODE == D_STATS) ? -INFINITY : 0.0f; }
  if (MODE == D_BDC) for (int i = threadIdx.x; i < 27 * 32; i += 256) ((float*)sdw)[i] = 0.0f;
  __syncthreads();
  if (active) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == D_STATS) {
        atomicAdd(&sd[0][c0 + r], st1[r]); atomicAdd(&sd[1][c0 + r], st2[r]);
        atomic_min_f32(&sf[0][c0 + r], smn[r]); atomic_max_f32(&sf[1][c0 + r], smx[r]);
      } else if (MODE == D_BRED) { atomicAdd(&sf[0][c0 + r], r1[r]); atomicAdd(&sf[1][c0 + r], r2[r]); }
      else if (MODE == D_BDC) {
#pragma unroll
        for (int t = 0; t < 27; ++t) atomicAdd(&sdw[t][c0 + r], dw[t][r]);
      }
    }
  }
  __syncthreads();
  if (MODE == D_STATS) {
    int64_t* g_s1 = (int64_t*)p.stats; unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
    int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
    for (int c = threadIdx.x; c < p.cout; c += 256) {
      if (sf[0][c] <= sf[1][c]) {
        atomicAdd((unsigned long long*)&g_s1[c], (unsigned long long)(long long)sd[0][c]); atomicAdd(&g_s2[c], (unsigned long long)sd[1][c]);
        atomicMin(&g_mn[c], (int)sf[0][c]); atomicMax(&g_mx[c], (int)sf[1][c]);
      }
    }
  } else if (MODE == D_BRED) {
    for (int c = threadIdx.x; c < p.cout; c += 256) { atomicAdd(p.coef + FROST_COEF_S1 * p.cpad + c, sf[0][c]); atomicAdd(p.coef + FROST_COEF_S2 * p.cpad + c, sf[1][c]); }
  } else if (MODE == D_BDC) {
    const float sx = p.qx[FROST_Q_SCALE];
    for (int i = threadIdx.x; i < 27 * p.cout; i += 256) {
      const int co = i / 27, t = i % 27; const int tap = t / 3, c = t % 3;      // dwq layout [cout][3][3][3] (OIHW)
      atomicAdd(p.dwq + (int64_t)co * 27 + c * 9 + tap, sdw[t][co] * sx);
    }
  }
}
static void fill_stem(StemP& p, const int8_t* x, const float* qx, const int8_t* wq, const int32_t* wsum, int n, int h, int w, int cout) {
  p.x = x; p.qx = qx; p.wq = wq; p.wsum = wsum; p.n = n; p.h = h; p.w = w; p.cout = cout; p.cpad = round_up(cout, 16);
  p.ho = (h + 2 - 3) / 2 + 1; p.wo = (w + 2 - 3) / 2 + 1; p.inv_count = 1.0f / (float)((int64_t)n * p.ho * p.wo);
}
template <int MODE> static int launch_stem(StemP& p, hipStream_t s) {
  int npl = 256 / (p.cout / 4); int64_t npix = (int64_t)p.n * p.ho * p.wo; int64_t grid = (npix + npl - 1) / npl;
  int cap = (MODE == D_EMIT) ? 8192 : 1024; if (grid > cap) grid = cap;
  hipLaunchKernelGGL((k_stem<MODE>), dim3((unsigned)grid), dim3(256), 0, s, p);
  return frost_check_launch("stem");
}
extern "C" int frost_stem_conv_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n,
                                   int h, int w, int cout, int mode, void* stats, const float* coef, const float* qrec_y,
                                   int relu, int8_t* y, void* stream) {
  FROST_REQUIRE(cout % 4 == 0 && cout <= 32 && cout >= 4, "stem: cout must be a multiple of 4 in [4,32]");
  StemP p = {}; fill_stem(p, x, qrec_x, wq_pack, wsum, n, h, w, cout);
  p.stats = (uint8_t*)stats; p.coef = (float*)coef; p.qy = qrec_y; p.relu = relu; p.y = y;
  return mode == 0 ? launch_stem<D_STATS>(p, as_stream(stream)) : launch_stem<D_EMIT>(p, as_stream(stream));
}
extern "C" int frost_stem_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n,
                                   int h, int w, int cout, int pass, float* coef, const float* qrec_y, int relu,
                                   const uint16_t* gout, float* dwq, void* stream) {
  FROST_REQUIRE(cout % 4 == 0 && cout <= 32 && cout >= 4, "stem: cout must be a multiple of 4 in [4,32]");
  StemP p = {}; fill_stem(p, x, qrec_x, wq_pack, wsum, n, h, w, cout);
  p.coef = coef; p.qy = qrec_y; p.relu = relu; p.gout = gout; p.dwq = dwq;
  return pass == 0 ? launch_stem<D_BRED>(p, as_stream(stream)) : launch_stem<D_BDC>(p, as_stream(stream));
}
