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

// v2: 512 threads = 32 channel pairs x 16 output columns; each thread owns 2 channels x 1 column x 8 rows.
// ~110 VGPRs -> 2 workgroups (16 waves) per CU, so one workgroup's halo staging overlaps the other's FMAs.
// Staging loads are issued as one batch (fixed trip count) before a single wait.
__device__ __forceinline__ void unpack2(uint32_t v, float* f) {    // two offset-binary bytes -> unsigned index floats
  v ^= 0x8080u;
  f[0] = (float)(v & 255u); f[1] = (float)((v >> 8) & 255u);
}
template <int K, int S>
__device__ __forceinline__ void dw_stage_tile(const int8_t* __restrict__ x, uint8_t* smem, int tid, int img, int iy0, int ix0,
                                              int cb, int h, int w, int c, uint32_t zfill) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int NU = (IH * IW * 8 + 511) / 512;
  uint2 v[NU];
#pragma unroll
  for (int jn = 0; jn < NU; ++jn) {
    const int u = tid + jn * 512;
    const int c8 = u & 7; const int pix = u >> 3; const int iy = pix / IW, ix = pix - iy * IW;
    const int gy = iy0 + iy, gx = ix0 + ix; const int cc = cb * CB + c8 * 8;
    v[jn] = make_uint2(zfill, zfill);
    if (u < IH * IW * 8 && gy >= 0 && gy < h && gx >= 0 && gx < w && cc < c)
      v[jn] = *(const uint2*)(x + (((int64_t)img * h + gy) * w + gx) * c + cc);
  }
#pragma unroll
  for (int jn = 0; jn < NU; ++jn) {
    const int u = tid + jn * 512;
    if (u < IH * IW * 8) *(uint2*)(smem + u * 8) = v[jn];
  }
}

template <int K, int S, int MODE>
__global__ __launch_bounds__(512, ((K == 3 && (S == 1 || MODE < 2)) ? 4 : 2)) void k_dw(const DwP p) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ double red_d[8][32][4]; __shared__ float red_f[8][32][4];
  const int tid = threadIdx.x;
  const int cp = tid & 31, ox = tid >> 5;
  const int cb = blockIdx.x % p.ncb, grp = blockIdx.x / p.ncb;
  const int c0 = cb * CB + cp * 2;
  const bool chok = c0 < p.c;
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]);
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;

  float wf[K * K][2];
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    const uint32_t v = chok ? (uint32_t)*(const uint16_t*)(p.wq + t * p.cpad + c0) : 0u;
    wf[t][0] = (float)(int8_t)(v & 255u); wf[t][1] = (float)(int8_t)(v >> 8);
  }
  float corr[2] = {0, 0};
  if (chok) { corr[0] = (float)(zp * p.wsum[c0]); corr[1] = (float)(zp * p.wsum[c0 + 1]); }
  float y_inv = 1.0f; int y_zp = 0;
  if (MODE != D_STATS) { y_inv = 1.0f / p.qy[FROST_Q_SCALE]; y_zp = __float_as_int(p.qy[FROST_Q_ZP]); }
  double st1[2] = {0, 0}, st2[2] = {0, 0}; float smn[2] = {INFINITY, INFINITY}, smx[2] = {-INFINITY, -INFINITY};
  float r1[2] = {0, 0}, r2[2] = {0, 0};

  const int tiles_per_img = p.tiles_x * p.tiles_y;
  for (int64_t tile = grp; tile < p.ntiles; tile += p.ngroups) {
    const int img = (int)(tile / tiles_per_img); const int tr = (int)(tile - (int64_t)img * tiles_per_img);
    const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
    __syncthreads();
    dw_stage_tile<K, S>(p.x, smem, tid, img, oy0 * S - p.pad, ox0 * S - p.pad, cb, p.h, p.w, p.c, zfill);
    __syncthreads();
    float acc[TH][2];
#pragma unroll
    for (int o = 0; o < TH; ++o) { acc[o][0] = 0; acc[o][1] = 0; }
#pragma unroll
    for (int iy = 0; iy < IH; ++iy) {
      float xf[K][2];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) unpack2(*(const uint16_t*)(smem + (iy * IW + ox * S + kx) * CB + cp * 2), xf[kx]);
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        if ((iy - ky) >= 0 && ((iy - ky) % S) == 0 && (iy - ky) / S < TH) {
          const int o = (iy - ky) / S;
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            acc[o][0] = fmaf(xf[kx][0], wf[ky * K + kx][0], acc[o][0]);
            acc[o][1] = fmaf(xf[kx][1], wf[ky * K + kx][1], acc[o][1]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the LDS look-ahead to one row: register pressure, not latency, binds here
    }
    float A[2] = {0, 0}, B[2] = {0, 0}, Mv[2] = {0, 0}, Rv[2] = {0, 0}, K1[2] = {0, 0}, S1[2] = {0, 0}, S2[2] = {0, 0};
    if (MODE != D_STATS && chok) {      // (re)loaded per tile from L1/L2: keeps them out of the main loop's live set
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        A[r] = p.coef[FROST_COEF_A * p.cpad + c0 + r]; B[r] = p.coef[FROST_COEF_B * p.cpad + c0 + r];
        if (MODE != D_EMIT) { Mv[r] = p.coef[FROST_COEF_M * p.cpad + c0 + r]; Rv[r] = p.coef[FROST_COEF_R * p.cpad + c0 + r]; }
        if (MODE == D_BDC) {
          K1[r] = p.coef[FROST_COEF_K1 * p.cpad + c0 + r];
          S1[r] = p.coef[FROST_COEF_S1 * p.cpad + c0 + r] * p.inv_count; S2[r] = p.coef[FROST_COEF_S2 * p.cpad + c0 + r] * p.inv_count;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < TH; ++o) {
      const int oy = oy0 + o, oxx = ox0 + ox;
      const bool valid = chok && oy < p.ho && oxx < p.wo;
      const int64_t opix = ((int64_t)img * p.ho + oy) * p.wo + oxx;
      const float v[2] = {acc[o][0] - corr[0], acc[o][1] - corr[1]};
      if (MODE == D_STATS) {
        if (valid) {
#pragma unroll
          for (int r = 0; r < 2; ++r) { st1[r] += (double)v[r]; st2[r] += (double)v[r] * (double)v[r]; smn[r] = fminf(smn[r], v[r]); smx[r] = fmaxf(smx[r], v[r]); }
        }
      } else if (MODE == D_EMIT) {
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float yv = fmaf(A[r], v[r], B[r]);
          if (p.relu) yv = fmaxf(yv, 0.0f);
          packed |= ((uint32_t)((fq_index(yv, y_inv, y_zp, 0, 255) - 128) & 255)) << (8 * r);
        }
        if (valid) *(uint16_t*)(p.y + opix * p.c + c0) = (uint16_t)packed;
      } else {
        uint32_t gv = 0;
        if (valid) gv = *(const uint32_t*)(p.gout + opix * p.c + c0);
        const float gq[2] = {bf2f(gv & 0xffff), bf2f(gv >> 16)};
        float dcv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float yv = fmaf(A[r], v[r], B[r]);
          bool alive = true;
          if (p.relu) { alive = yv > 0.0f; yv = fmaxf(yv, 0.0f); }
          bool inr; fq_index(yv, y_inv, y_zp, 0, 255, &inr);
          const float gyv = (alive && inr && valid) ? gq[r] : 0.0f;
          const float xhat = (v[r] - Mv[r]) * Rv[r];
          if (MODE == D_BRED) { r1[r] += gyv; r2[r] += gyv * xhat; }
          else dcv[r] = K1[r] * (gyv - S1[r] - xhat * S2[r]);
        }
        if (MODE == D_BDC && valid) *(uint32_t*)(p.dc + opix * p.c + c0) = cvt_pk_bf16(dcv[0], dcv[1]);
      }
    }
  }

  // reduce over the 16 threads (ox) that share a channel pair: lane bit 5 (ox&1) and the 8 waves (ox>>1)
  if (MODE == D_STATS || MODE == D_BRED) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (MODE == D_STATS) {
        double a = st1[r], b = st2[r]; float c = smn[r], d = smx[r];
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32); c = fminf(c, __shfl_xor(c, 32)); d = fmaxf(d, __shfl_xor(d, 32));
        if (lane < 32) { red_d[wv][lane][r] = a; red_d[wv][lane][2 + r] = b; red_f[wv][lane][r] = c; red_f[wv][lane][2 + r] = d; }
      } else {
        float a = r1[r], b = r2[r];
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
        if (lane < 32) { red_f[wv][lane][r] = a; red_f[wv][lane][2 + r] = b; }
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int q = tid >> 1, r = tid & 1; const int ch = cb * CB + q * 2 + r;
      if (ch < p.c) {
        if (MODE == D_STATS) {
          double a = 0, b = 0; float c = INFINITY, d = -INFINITY;
          for (int w2 = 0; w2 < 8; ++w2) { a += red_d[w2][q][r]; b += red_d[w2][q][2 + r]; c = fminf(c, red_f[w2][q][r]); d = fmaxf(d, red_f[w2][q][2 + r]); }
          int64_t* g_s1 = (int64_t*)p.stats; unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
          int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
          if (c <= d) {
            atomicAdd((unsigned long long*)&g_s1[ch], (unsigned long long)(long long)a);
            atomicAdd(&g_s2[ch], (unsigned long long)b);
            atomicMin(&g_mn[ch], (int)c); atomicMax(&g_mx[ch], (int)d);
          }
        } else {
          float a = 0, b = 0;
          for (int w2 = 0; w2 < 8; ++w2) { a += red_f[w2][q][r]; b += red_f[w2][q][2 + r]; }
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
  int occ = 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_dw<K, S, MODE>, 512, lds) != hipSuccess || occ < 1) occ = 1;
  if (occ > 4) occ = 4;
  int64_t want = (256 * occ) / p.ncb; if (want < 1) want = 1;
  p.ngroups = (int)(p.ntiles < want ? p.ntiles : want);
  hipLaunchKernelGGL((k_dw<K, S, MODE>), dim3(p.ncb * p.ngroups), dim3(512), lds, s, p);
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
extern "C" int frost_dw_conv_fwd_v2(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n,
                                 int h, int w, int c, int k, int stride, int mode, void* stats, const float* coef,
                                 const float* qrec_y, int relu, int8_t* y, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  DwP p = {}; fill_dw(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.stats = (uint8_t*)stats; p.coef = (float*)coef; p.qy = qrec_y; p.relu = relu; p.y = y;
  return mode == 0 ? dispatch_dw<D_STATS>(p, k, stride, as_stream(stream)) : dispatch_dw<D_EMIT>(p, k, stride, as_stream(stream));
}
extern "C" int frost_dw_conv_bwd_v2(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                                 const float* qrec_w, int n, int h, int w, int c, int k, int stride, int pass, float* coef,
                                 const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  DwP p = {}; fill_dw(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.coef = coef; p.qy = qrec_y; p.relu = relu; p.gout = gout; p.dc = dc;
  return pass == 0 ? dispatch_dw<D_BRED>(p, k, stride, as_stream(stream)) : dispatch_dw<D_BDC>(p, k, stride, as_stream(stream));
}

// ---- depthwise dgrad: dx[n][iy][ix][c] (+)= s_w * sum_{ky,kx} dc[n][(iy+pad-ky)/s][(ix+pad-kx)/s][c] * wq[ky][kx][c]
// (taps with a non-integer source index contribute nothing).  LDS-tiled: the bf16 dc halo of an 8x16 dx tile is
// staged once, every thread owns 2 channels x 1 column x 8 rows.  For stride 2 a wave handles columns of ONE
// parity (w and w+8), so which kx taps exist is wave-uniform and every tap test folds at compile time.
__host__ __device__ constexpr int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
template <int K, int S, int PAR>
__device__ __forceinline__ void dw_dgrad_tile(const uint8_t* smem, const float (*wf)[2], int cp, int m, float (*acc)[2]) {
  constexpr int PAD = (K - 1) / 2;
  constexpr int LO = fdiv(-PAD, S);
  constexpr int DW = (TW - 1 + PAD) / S - LO + 1;
#pragma unroll
  for (int r = 0; r < TH; ++r) {
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      if (((r + PAD - ky) % S + S) % S != 0) continue;
      const int rr = fdiv(r + PAD - ky, S) - LO;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        if (((PAR + PAD - kx) % S + S) % S != 0) continue;
        const int cc = fdiv(PAR + PAD - kx, S) - LO;           // + m (runtime)
        const uint32_t v = *(const uint32_t*)(smem + ((rr * DW + cc + m) * CB + cp * 2) * 2);
        acc[r][0] = fmaf(bf2f(v & 0xffff), wf[ky * K + kx][0], acc[r][0]);
        acc[r][1] = fmaf(bf2f(v >> 16), wf[ky * K + kx][1], acc[r][1]);
      }
    }
  }
}
template <int K, int S>
__global__ __launch_bounds__(512, 2) void k_dw_dgrad2(const uint16_t* __restrict__ dc, const int8_t* __restrict__ wq, const float* qw,
                                                      int n, int h, int w, int c, int cpad, int ho, int wo, uint16_t* __restrict__ dx,
                                                      int accumulate, int ncb, int ngroups, int tiles_x, int tiles_y) {
  constexpr int PAD = (K - 1) / 2;
  constexpr int LO = fdiv(-PAD, S);
  constexpr int DH = (TH - 1 + PAD) / S - LO + 1, DW = (TW - 1 + PAD) / S - LO + 1;
  constexpr int NU = (DH * DW * 8 + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int cp = tid & 31;
  const int wv = tid >> 6, hi = (tid >> 5) & 1;
  const int ox = (S == 2) ? (wv + 8 * hi) : (tid >> 5);          // S=2: a wave owns columns {w, w+8} (same parity)
  const int cb = blockIdx.x % ncb, grp = blockIdx.x / ncb;
  const int c0 = cb * CB + cp * 2;
  const bool chok = c0 < c;
  const float sw = qw[FROST_Q_SCALE];
  float wf[K * K][2];
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    const uint32_t v = chok ? (uint32_t)*(const uint16_t*)(wq + t * cpad + c0) : 0u;
    wf[t][0] = (float)(int8_t)(v & 255u); wf[t][1] = (float)(int8_t)(v >> 8);
  }
  const int tiles_per_img = tiles_x * tiles_y; const int64_t ntiles = (int64_t)n * tiles_per_img;
  for (int64_t tile = grp; tile < ntiles; tile += ngroups) {
    const int img = (int)(tile / tiles_per_img); const int tr = (int)(tile - (int64_t)img * tiles_per_img);
    const int iy0 = (tr / tiles_x) * TH, ix0 = (tr % tiles_x) * TW;
    const int oy_lo = iy0 / S + LO, ox_lo = ix0 / S + LO;
    __syncthreads();
    {
      uint4 v[NU];
#pragma unroll
      for (int jn = 0; jn < NU; ++jn) {
        const int u = tid + jn * 512; const int c8 = u & 7; const int pix = u >> 3; const int ry = pix / DW, rx = pix - ry * DW;
        const int oy = oy_lo + ry, oxx = ox_lo + rx; const int cc = cb * CB + c8 * 8;
        v[jn] = make_uint4(0, 0, 0, 0);
        if (u < DH * DW * 8 && oy >= 0 && oy < ho && oxx >= 0 && oxx < wo && cc < c)
          v[jn] = *(const uint4*)(dc + (((int64_t)img * ho + oy) * wo + oxx) * c + cc);
      }
#pragma unroll
      for (int jn = 0; jn < NU; ++jn) { const int u = tid + jn * 512; if (u < DH * DW * 8) *(uint4*)(smem + u * 16) = v[jn]; }
    }
    __syncthreads();
    float acc[TH][2];
#pragma unroll
    for (int r = 0; r < TH; ++r) { acc[r][0] = 0; acc[r][1] = 0; }
    if (S == 1) dw_dgrad_tile<K, S, 0>(smem, wf, cp, ox, acc);
    else if ((wv & 1) == 0) dw_dgrad_tile<K, S, 0>(smem, wf, cp, ox >> 1, acc);
    else dw_dgrad_tile<K, S, 1>(smem, wf, cp, ox >> 1, acc);
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      const int iy = iy0 + r, ix = ix0 + ox;
      if (chok && iy < h && ix < w) {
        uint16_t* dst = dx + (((int64_t)img * h + iy) * w + ix) * c + c0;
        float v0 = acc[r][0] * sw, v1 = acc[r][1] * sw;
        if (accumulate) { const uint32_t o = *(const uint32_t*)dst; v0 += bf2f(o & 0xffff); v1 += bf2f(o >> 16); }
        *(uint32_t*)dst = cvt_pk_bf16(v0, v1);
      }
    }
  }
}
template <int K, int S>
static int launch_dw_dgrad(const uint16_t* dc, const int8_t* wq, const float* qw, int n, int h, int w, int c, int ho, int wo,
                           uint16_t* dx, int accumulate, hipStream_t s) {
  constexpr int PAD = (K - 1) / 2; constexpr int LO = fdiv(-PAD, S);
  constexpr int DH = (TH - 1 + PAD) / S - LO + 1, DW = (TW - 1 + PAD) / S - LO + 1;
  const int ncb = (c + CB - 1) / CB; const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
  const int64_t ntiles = (int64_t)n * tiles_x * tiles_y;
  int64_t want = 1536 / ncb; if (want < 1) want = 1;
  const int ngroups = (int)(ntiles < want ? ntiles : want);
  hipLaunchKernelGGL((k_dw_dgrad2<K, S>), dim3(ncb * ngroups), dim3(512), (size_t)DH * DW * CB * 2, s, dc, wq, qw, n, h, w, c,
                     round_up(c, 16), ho, wo, dx, accumulate, ncb, ngroups, tiles_x, tiles_y);
  return frost_check_launch("dw_dgrad");
}
extern "C" int frost_dw_dgrad_v2(const uint16_t* dc, const int8_t* wq_pack, const float* qrec_w, int n, int h, int w, int c,
                              int k, int stride, uint16_t* dx, int accumulate, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw_dgrad: channels must be a multiple of 8");
  const int pad = (k - 1) / 2; const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
  hipStream_t s = as_stream(stream);
  if (k == 3 && stride == 1) return launch_dw_dgrad<3, 1>(dc, wq_pack, qrec_w, n, h, w, c, ho, wo, dx, accumulate, s);
  if (k == 3 && stride == 2) return launch_dw_dgrad<3, 2>(dc, wq_pack, qrec_w, n, h, w, c, ho, wo, dx, accumulate, s);
  if (k == 5 && stride == 1) return launch_dw_dgrad<5, 1>(dc, wq_pack, qrec_w, n, h, w, c, ho, wo, dx, accumulate, s);
  if (k == 5 && stride == 2) return launch_dw_dgrad<5, 2>(dc, wq_pack, qrec_w, n, h, w, c, ho, wo, dx, accumulate, s);
  frost_set_error("dw_dgrad: unsupported kernel/stride"); return 1;
}

// ---- depthwise wgrad: dwq[c][ky][kx] += s_x * sum_{n,oy,ox} dc * (q - zp)     (dwq fp32 [c][k*k], pre-zeroed)
// Same tiling as the forward kernel: the x halo tile is staged in LDS, each thread holds dc for its 8 outputs x 2
// channels and slides over the tile accumulating K*K x 2 partial sums in registers across ALL its tiles; one block
// reduction + K*K*64 atomics per workgroup at the end.
template <int K, int S>
__global__ __launch_bounds__(512, ((K == 3 && S == 1) ? 4 : 2)) void k_dw_wgrad(const DwP p, float* __restrict__ dwq) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int cp = tid & 31, ox = tid >> 5;
  const int cb = blockIdx.x % p.ncb, grp = blockIdx.x / p.ncb;
  const int c0 = cb * CB + cp * 2;
  const bool chok = c0 < p.c;
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]);
  const float zpf = (float)zp;
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;
  float acc[K * K][2];
#pragma unroll
  for (int t = 0; t < K * K; ++t) { acc[t][0] = 0; acc[t][1] = 0; }
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  for (int64_t tile = grp; tile < p.ntiles; tile += p.ngroups) {
    const int img = (int)(tile / tiles_per_img); const int tr = (int)(tile - (int64_t)img * tiles_per_img);
    const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
    float g[TH][2];
#pragma unroll
    for (int o = 0; o < TH; ++o) {
      const int oy = oy0 + o, oxx = ox0 + ox;
      uint32_t gv = 0;
      if (chok && oy < p.ho && oxx < p.wo) gv = *(const uint32_t*)(p.dc + (((int64_t)img * p.ho + oy) * p.wo + oxx) * p.c + c0);
      g[o][0] = bf2f(gv & 0xffff); g[o][1] = bf2f(gv >> 16);
    }
    __syncthreads();
    dw_stage_tile<K, S>(p.x, smem, tid, img, oy0 * S - p.pad, ox0 * S - p.pad, cb, p.h, p.w, p.c, zfill);
    __syncthreads();
#pragma unroll
    for (int iy = 0; iy < IH; ++iy) {
      float xf[K][2];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) { unpack2(*(const uint16_t*)(smem + (iy * IW + ox * S + kx) * CB + cp * 2), xf[kx]); xf[kx][0] -= zpf; xf[kx][1] -= zpf; }
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        if ((iy - ky) >= 0 && ((iy - ky) % S) == 0 && (iy - ky) / S < TH) {
          const int o = (iy - ky) / S;
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            acc[ky * K + kx][0] = fmaf(g[o][0], xf[kx][0], acc[ky * K + kx][0]);
            acc[ky * K + kx][1] = fmaf(g[o][1], xf[kx][1], acc[ky * K + kx][1]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  float* red = (float*)smem;                      // [K*K][64]
  for (int i = tid; i < K * K * 64; i += 512) red[i] = 0.0f;
  __syncthreads();
  if (chok) {
#pragma unroll
    for (int t = 0; t < K * K; ++t) { atomicAdd(&red[t * 64 + cp * 2], acc[t][0]); atomicAdd(&red[t * 64 + cp * 2 + 1], acc[t][1]); }
  }
  __syncthreads();
  const float sx = p.qx[FROST_Q_SCALE];
  for (int i = tid; i < K * K * 64; i += 512) {
    const int t = i / 64, cc = i % 64; const int ch = cb * 64 + cc;
    if (ch < p.c) atomicAdd(dwq + (int64_t)ch * K * K + t, red[i] * sx);
  }
}
template <int K, int S>
static int launch_dw_wgrad(DwP& p, float* dwq, hipStream_t s) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  size_t lds = (size_t)IH * IW * CB; if (lds < (size_t)K * K * 64 * 4) lds = (size_t)K * K * 64 * 4;
  int64_t want = 1024 / p.ncb; if (want < 1) want = 1;
  p.ngroups = (int)(p.ntiles < want ? p.ntiles : want);
  hipLaunchKernelGGL((k_dw_wgrad<K, S>), dim3(p.ncb * p.ngroups), dim3(512), lds, s, p, dwq);
  return frost_check_launch("dw_wgrad");
}
static void fill_dw(DwP& p, const int8_t* x, const float* qx, const int8_t* wq, const int32_t* wsum, int n, int h, int w, int c, int k, int stride);
extern "C" int frost_dw_wgrad_v2(const uint16_t* dc, const int8_t* x, const float* qrec_x, int n, int h, int w, int c, int k,
                              int stride, float* dwq, void* stream) {
  DwP p = {}; fill_dw(p, x, qrec_x, nullptr, nullptr, n, h, w, c, k, stride); p.dc = (uint16_t*)dc;
  hipStream_t s = as_stream(stream);
  if (k == 3 && stride == 1) return launch_dw_wgrad<3, 1>(p, dwq, s);
  if (k == 3 && stride == 2) return launch_dw_wgrad<3, 2>(p, dwq, s);
  if (k == 5 && stride == 1) return launch_dw_wgrad<5, 1>(p, dwq, s);
  if (k == 5 && stride == 2) return launch_dw_wgrad<5, 2>(p, dwq, s);
  frost_set_error("dw_wgrad: unsupported kernel/stride"); return 1;
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
