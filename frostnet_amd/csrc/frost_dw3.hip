// Depthwise k x k conv (k in {3,5}, stride {1,2}) on fake-quantised NHWC bytes -- "lane = channel" formulation.
//
// A workgroup (4 waves) owns an 8x16 output tile of a 64-channel block; every LANE owns ONE channel and a 4-row x 8-column
// patch of outputs.  The halo tile is staged in LDS in its natural [row][col][64 ch] layout with coalesced loads and each
// lane pulls 8 consecutive x-pixels OF ITS OWN CHANNEL with one gfx950 LDS transpose read (ds_read_b64_tr_b8; semantics
// measured in tools/probe_tr.hip), converts them once and slides down the tile with fp32 FMAs (exact: |sum| < 2^24).
// Consequences: 25 (not 50-100) weight registers per thread, no cross-lane reductions anywhere (per-channel statistics,
// S1/S2 and the 25 weight-gradient sums are lane-local across the whole persistent tile loop), per-channel coefficients
// are scalars per lane.  Results go back to NHWC through an LDS transpose (ds_write_b8/b16 -> coalesced 8/16-byte stores).
// Zero padding = zero-point fill, so acc_true = sum(w*q) - zp*sum(w) also holds at the borders.
#include "frost_common.h"
#include <map>

typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
enum { D_STATS = 0, D_EMIT = 1, D_BRED = 2, D_BDC = 3 };
#define TH 8
#define TW 16
#define CB 64
#define RH 4
#define RW 8

struct Dw3P {
  const int8_t* x; const float* qx; const int8_t* wq; const int32_t* wsum; const float* qw;
  int n, h, w, c, cpad, ho, wo, pad;
  uint8_t* stats; float* coef; const float* qy; int relu; int8_t* y;
  const uint16_t* gout; uint16_t* dc; float* dwq; uint16_t* dx; int accumulate;
  int tiles_x, tiles_y, ncb, ngroups; int64_t ntiles; float inv_count;
};

__host__ __device__ constexpr int fdiv3(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// 8 consecutive x-pixels of this lane's channel (offset-binary bytes) -> 8 unsigned-index floats
__device__ __forceinline__ void tr8_run(const uint8_t* tile_row, int col0, int lane, float* out8) {
  const int jp = lane & 15, G = lane >> 4;
  const v2i raw = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(tile_row + (col0 + (jp >> 1)) * CB + 16 * G + 8 * (jp & 1)));
  const uint32_t u0 = (uint32_t)raw[0] ^ 0x80808080u, u1 = (uint32_t)raw[1] ^ 0x80808080u;
  out8[0] = (float)(u0 & 255u); out8[1] = (float)((u0 >> 8) & 255u); out8[2] = (float)((u0 >> 16) & 255u); out8[3] = (float)(u0 >> 24);
  out8[4] = (float)(u1 & 255u); out8[5] = (float)((u1 >> 8) & 255u); out8[6] = (float)((u1 >> 16) & 255u); out8[7] = (float)(u1 >> 24);
}
// 4 consecutive x-pixels of this lane's channel from a bf16 [row][col][64 ch] tile
__device__ __forceinline__ void tr16_run(const uint8_t* tile_row, int col0, int lane, float* out4) {
  const int jp = lane & 15, G = lane >> 4;
  const v4s raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(tile_row + ((col0 + (jp >> 2)) * CB + 16 * G + 4 * (jp & 3)) * 2));
  out4[0] = bf2f((uint16_t)raw[0]); out4[1] = bf2f((uint16_t)raw[1]); out4[2] = bf2f((uint16_t)raw[2]); out4[3] = bf2f((uint16_t)raw[3]);
}

// stage the int8 halo tile [IH][IW][64 B] (8-byte units: channel counts are multiples of 8, not always of 16)
template <int IH, int IW>
__device__ __forceinline__ void stage_in_tile(const int8_t* __restrict__ x, uint8_t* tile, int tid, int img, int iy0, int ix0, int cb,
                                              int h, int w, int c, uint32_t zfill) {
  constexpr int NUNIT = IH * IW * 8;
#pragma unroll 1
  for (int base = 0; base < NUNIT; base += 256 * 8) {
    uint2 v[8];
#pragma unroll
    for (int jn = 0; jn < 8; ++jn) {
      const int u = base + tid + jn * 256; const int c8 = u & 7; const int pix = u >> 3; const int iy = pix / IW, ix = pix - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix; const int cc = cb * CB + c8 * 8;
      v[jn] = make_uint2(zfill, zfill);
      if (u < NUNIT && gy >= 0 && gy < h && gx >= 0 && gx < w && cc < c) v[jn] = *(const uint2*)(x + (((int64_t)img * h + gy) * w + gx) * c + cc);
    }
#pragma unroll
    for (int jn = 0; jn < 8; ++jn) { const int u = base + tid + jn * 256; if (u < NUNIT) *(uint2*)(tile + u * 8) = v[jn]; }
  }
}
// stage a bf16 [NR][NC][64 ch] tile (16-byte units), rows/cols outside [0,hh)x[0,ww) -> 0
template <int NR, int NC>
__device__ __forceinline__ void stage_bf16_tile(const uint16_t* __restrict__ src, uint8_t* tile, int tid, int img, int r0, int c0, int cb,
                                                int hh, int ww, int c) {
  constexpr int NUNIT = NR * NC * 8;
#pragma unroll 1
  for (int base = 0; base < NUNIT; base += 256 * 4) {
    uint4 v[4];
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u = base + tid + jn * 256; const int c8 = u & 7; const int pix = u >> 3; const int ry = pix / NC, rx = pix - ry * NC;
      const int gy = r0 + ry, gx = c0 + rx; const int cc = cb * CB + c8 * 8;
      v[jn] = make_uint4(0, 0, 0, 0);
      if (u < NUNIT && gy >= 0 && gy < hh && gx >= 0 && gx < ww && cc < c) v[jn] = *(const uint4*)(src + (((int64_t)img * hh + gy) * ww + gx) * c + cc);
    }
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) { const int u = base + tid + jn * 256; if (u < NUNIT) *(uint4*)(tile + u * 16) = v[jn]; }
  }
}

template <int K, int S, int MODE>
__global__ __launch_bounds__(256, 2) void k_dw3(const Dw3P p) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int IN_BYTES = ((IH * IW * CB + 255) / 256) * 256 + 256;
  constexpr int NXR = (RW - 1) * S + K;                 // input pixels a lane needs per row
  constexpr int NBLK = (NXR + 7) / 8;                   // 8-pixel transpose reads per row
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* tin = smem;
  uint8_t* aux = smem + IN_BYTES;                       // gout / dc bf16 tile (16 KB) or int8 out tile (8 KB)
  double* red_d = (double*)(smem + IN_BYTES + 16384);   // [4][64][2]
  float* red_f = (float*)(red_d + 4 * 64 * 2);          // [4][64][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wv >> 1, wx = wv & 1;
  const int cb = blockIdx.x % p.ncb, grp = blockIdx.x / p.ncb;
  const int ch = cb * CB + lane;
  const bool chok = ch < p.c;
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]);
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;

  float wf[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) wf[t] = chok ? (float)p.wq[t * p.cpad + ch] : 0.0f;
  const float corr = chok ? (float)(zp * p.wsum[ch]) : 0.0f;
  float cA = 0, cB = 0, cM = 0, cR = 0, cK1 = 0, cS1 = 0, cS2 = 0, y_inv = 1.0f, y_zpf = 0.0f;
  if (MODE != D_STATS) {
    y_inv = 1.0f / p.qy[FROST_Q_SCALE]; y_zpf = (float)__float_as_int(p.qy[FROST_Q_ZP]);
    if (chok) {
      cA = p.coef[FROST_COEF_A * p.cpad + ch]; cB = p.coef[FROST_COEF_B * p.cpad + ch];
      if (MODE != D_EMIT) { cM = p.coef[FROST_COEF_M * p.cpad + ch]; cR = p.coef[FROST_COEF_R * p.cpad + ch]; }
      if (MODE == D_BDC) { cK1 = p.coef[FROST_COEF_K1 * p.cpad + ch]; cS1 = p.coef[FROST_COEF_S1 * p.cpad + ch] * p.inv_count; cS2 = p.coef[FROST_COEF_S2 * p.cpad + ch] * p.inv_count; }
    }
  }
  const float relu_floor = p.relu ? 0.0f : -INFINITY;
  double st1 = 0.0, st2 = 0.0; float smn = INFINITY, smx = -INFINITY, r1 = 0.0f, r2 = 0.0f;

  const int tiles_per_img = p.tiles_x * p.tiles_y;
  for (int64_t tile = grp; tile < p.ntiles; tile += p.ngroups) {
    const int img = (int)(tile / tiles_per_img); const int tr = (int)(tile - (int64_t)img * tiles_per_img);
    const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
    __syncthreads();
    stage_in_tile<IH, IW>(p.x, tin, tid, img, oy0 * S - p.pad, ox0 * S - p.pad, cb, p.h, p.w, p.c, zfill);
    if (MODE == D_BRED || MODE == D_BDC) stage_bf16_tile<TH, TW>(p.gout, aux, tid, img, oy0, ox0, cb, p.ho, p.wo, p.c);
    __syncthreads();

    float acc[RH][RW];
#pragma unroll
    for (int o = 0; o < RH; ++o)
#pragma unroll
      for (int r = 0; r < RW; ++r) acc[o][r] = 0.0f;
    constexpr int NROW = (RH - 1) * S + K;
#pragma unroll
    for (int jr = 0; jr < NROW; ++jr) {
      float xr[NBLK * 8];
      const uint8_t* rowp = tin + ((wy * RH * S + jr) * IW) * CB;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) tr8_run(rowp, wx * RW * S + b * 8, lane, xr + b * 8);
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        if ((jr - ky) >= 0 && ((jr - ky) % S) == 0 && (jr - ky) / S < RH) {
          const int o = (jr - ky) / S;
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[o][r] = fmaf(xr[r * S + kx], wf[ky * K + kx], acc[o][r]);
        }
      }
    }

    // ---------------------------------------------------------------- epilogue (lane-local, one channel)
#pragma unroll
    for (int o = 0; o < RH; ++o) {
      const int oy = oy0 + wy * RH + o;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int ox = ox0 + wx * RW + r;
        const bool valid = chok && oy < p.ho && ox < p.wo;
        const float v = acc[o][r] - corr;
        const int lp = (wy * RH + o) * TW + wx * RW + r;          // pixel index inside the 8x16 tile
        if (MODE == D_STATS) {
          if (valid) { st1 += (double)v; st2 += (double)v * (double)v; smn = fminf(smn, v); smx = fmaxf(smx, v); }
        } else if (MODE == D_EMIT) {
          const float yv = fmaf(cA, v, cB);
          const float qf = fminf(fmaxf(rintf(fmaxf(yv, relu_floor) * y_inv) + y_zpf, 0.0f), 255.0f);
          aux[lp * CB + lane] = (uint8_t)(((int)qf - 128) & 255);
        } else {
          const float gq = bf2f(*(const uint16_t*)(aux + (lp * CB + lane) * 2));
          const float yv = fmaf(cA, v, cB);
          const float qf = rintf(fmaxf(yv, relu_floor) * y_inv) + y_zpf;
          const bool pass = valid && (yv > relu_floor) && qf >= 0.0f && qf <= 255.0f;
          const float gy = pass ? gq : 0.0f;
          const float xhat = (v - cM) * cR;
          if (MODE == D_BRED) { r1 += gy; r2 += gy * xhat; }
          else *(uint16_t*)(aux + (lp * CB + lane) * 2) = (uint16_t)cvt_pk_bf16(cK1 * (gy - cS1 - xhat * cS2), 0.0f);
        }
      }
    }
    if (MODE == D_EMIT || MODE == D_BDC) {
      __syncthreads();
      if (MODE == D_EMIT) {
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {        // 128 pixels x 8 units of 8 B
          const int u = tid + jn * 256; const int c8 = u & 7, lp = u >> 3; const int oy = oy0 + lp / TW, ox = ox0 + lp % TW; const int cc = cb * CB + c8 * 8;
          if (oy < p.ho && ox < p.wo && cc < p.c) *(uint2*)(p.y + (((int64_t)img * p.ho + oy) * p.wo + ox) * p.c + cc) = *(const uint2*)(aux + lp * CB + c8 * 8);
        }
      } else {
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {        // 128 pixels x 8 units of 16 B
          const int u = tid + jn * 256; const int c8 = u & 7, lp = u >> 3; const int oy = oy0 + lp / TW, ox = ox0 + lp % TW; const int cc = cb * CB + c8 * 8;
          if (oy < p.ho && ox < p.wo && cc < p.c) *(uint4*)(p.dc + (((int64_t)img * p.ho + oy) * p.wo + ox) * p.c + cc) = *(const uint4*)(aux + (lp * CB + c8 * 8) * 2);
        }
      }
    }
  }

  if (MODE == D_STATS || MODE == D_BRED) {      // sum the 4 waves' lane-local partials, one global atomic set per channel
    __syncthreads();
    if (MODE == D_STATS) { red_d[(wv * 64 + lane) * 2] = st1; red_d[(wv * 64 + lane) * 2 + 1] = st2; red_f[(wv * 64 + lane) * 2] = smn; red_f[(wv * 64 + lane) * 2 + 1] = smx; }
    else { red_f[(wv * 64 + lane) * 2] = r1; red_f[(wv * 64 + lane) * 2 + 1] = r2; }
    __syncthreads();
    if (tid < 64 && chok) {
      if (MODE == D_STATS) {
        double a = 0, b = 0; float c = INFINITY, d = -INFINITY;
        for (int w2 = 0; w2 < 4; ++w2) { a += red_d[(w2 * 64 + lane) * 2]; b += red_d[(w2 * 64 + lane) * 2 + 1]; c = fminf(c, red_f[(w2 * 64 + lane) * 2]); d = fmaxf(d, red_f[(w2 * 64 + lane) * 2 + 1]); }
        long long* g_s1 = (long long*)p.stats; unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
        int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
        if (c <= d) {
          atomicAdd((unsigned long long*)&g_s1[ch], (unsigned long long)(long long)a); atomicAdd(&g_s2[ch], (unsigned long long)b);
          atomicMin(&g_mn[ch], (int)c); atomicMax(&g_mx[ch], (int)d);
        }
      } else {
        float a = 0, b = 0;
        for (int w2 = 0; w2 < 4; ++w2) { a += red_f[(w2 * 64 + lane) * 2]; b += red_f[(w2 * 64 + lane) * 2 + 1]; }
        atomicAdd(p.coef + FROST_COEF_S1 * p.cpad + ch, a); atomicAdd(p.coef + FROST_COEF_S2 * p.cpad + ch, b);
      }
    }
  }
}

// ---- wgrad: dwq[c][ky][kx] += s_x * sum dc * (q - zp); lane-local K*K sums across the whole persistent loop
template <int K, int S>
__global__ __launch_bounds__(256, 2) void k_dw3_wgrad(const Dw3P p) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int IN_BYTES = ((IH * IW * CB + 255) / 256) * 256 + 256;
  constexpr int NXR = (RW - 1) * S + K; constexpr int NBLK = (NXR + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* tin = smem; uint8_t* aux = smem + IN_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wv >> 1, wx = wv & 1;
  const int cb = blockIdx.x % p.ncb, grp = blockIdx.x / p.ncb;
  const int ch = cb * CB + lane; const bool chok = ch < p.c;
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]); const float zpf = (float)zp;
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;
  float acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = 0.0f;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  for (int64_t tile = grp; tile < p.ntiles; tile += p.ngroups) {
    const int img = (int)(tile / tiles_per_img); const int tr = (int)(tile - (int64_t)img * tiles_per_img);
    const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
    __syncthreads();
    stage_in_tile<IH, IW>(p.x, tin, tid, img, oy0 * S - p.pad, ox0 * S - p.pad, cb, p.h, p.w, p.c, zfill);
    stage_bf16_tile<TH, TW>(p.dc, aux, tid, img, oy0, ox0, cb, p.ho, p.wo, p.c);
    __syncthreads();
    float g[RH][RW];
#pragma unroll
    for (int o = 0; o < RH; ++o) {
      const uint8_t* rowp = aux + ((wy * RH + o) * TW) * CB * 2;
      tr16_run(rowp, wx * RW, lane, &g[o][0]); tr16_run(rowp, wx * RW + 4, lane, &g[o][4]);
    }
    constexpr int NROW = (RH - 1) * S + K;
#pragma unroll
    for (int jr = 0; jr < NROW; ++jr) {
      float xr[NBLK * 8];
      const uint8_t* rowp = tin + ((wy * RH * S + jr) * IW) * CB;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) tr8_run(rowp, wx * RW * S + b * 8, lane, xr + b * 8);
#pragma unroll
      for (int i = 0; i < NBLK * 8; ++i) xr[i] -= zpf;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        if ((jr - ky) >= 0 && ((jr - ky) % S) == 0 && (jr - ky) / S < RH) {
          const int o = (jr - ky) / S;
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[ky * K + kx] = fmaf(g[o][r], xr[r * S + kx], acc[ky * K + kx]);
        }
      }
    }
  }
  __syncthreads();
  float* red = (float*)smem;                       // [4][K*K][64]
#pragma unroll
  for (int t = 0; t < K * K; ++t) red[(wv * K * K + t) * 64 + lane] = acc[t];
  __syncthreads();
  const float sx = p.qx[FROST_Q_SCALE];
  for (int i = tid; i < K * K * 64; i += 256) {
    const int t = i >> 6, l2 = i & 63; const int c2 = cb * CB + l2;
    if (c2 < p.c) atomicAdd(p.dwq + (int64_t)c2 * K * K + t, (red[i] + red[K * K * 64 + i] + red[2 * K * K * 64 + i] + red[3 * K * K * 64 + i]) * sx);
  }
}

// ---- dgrad: dx[iy][ix][c] (+)= s_w * sum_{ky,kx} dc[(iy+pad-ky)/s][(ix+pad-kx)/s][c] * wq[ky][kx][c]
template <int K, int S>
__global__ __launch_bounds__(256, 2) void k_dw3_dgrad(const Dw3P p) {
  constexpr int PAD = (K - 1) / 2;
  constexpr int LO = fdiv3(-PAD, S);
  constexpr int DH = (TH - 1 + PAD) / S - LO + 1, DW = (TW - 1 + PAD) / S - LO + 1;
  constexpr int D_BYTES = ((DH * DW * CB * 2 + 255) / 256) * 256 + 512;
  constexpr int NJR = (S == 1) ? (RH + K - 1) : ((RH - 1 + PAD) / 2 - LO + 1);
  constexpr int NR4 = (S == 1) ? (RW + K - 1 + 3) / 4 : 2;               // 4-pixel transpose reads per dc row
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* tdc = smem; uint8_t* tout = smem + D_BYTES;                   // dx bf16 out tile [128][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wv >> 1, wx = wv & 1;
  const int cb = blockIdx.x % p.ncb, grp = blockIdx.x / p.ncb;
  const int ch = cb * CB + lane; const bool chok = ch < p.c;
  const float sw = p.qw[FROST_Q_SCALE];
  float wf[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) wf[t] = chok ? (float)p.wq[t * p.cpad + ch] : 0.0f;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  for (int64_t tile = grp; tile < p.ntiles; tile += p.ngroups) {
    const int img = (int)(tile / tiles_per_img); const int tr = (int)(tile - (int64_t)img * tiles_per_img);
    const int iy0 = (tr / p.tiles_x) * TH, ix0 = (tr % p.tiles_x) * TW;
    __syncthreads();
    stage_bf16_tile<DH, DW>(p.dc, tdc, tid, img, iy0 / S + LO, ix0 / S + LO, cb, p.ho, p.wo, p.c);
    __syncthreads();
    float acc[RH][RW];
#pragma unroll
    for (int o = 0; o < RH; ++o)
#pragma unroll
      for (int r = 0; r < RW; ++r) acc[o][r] = 0.0f;
    const int rb = (S == 1) ? wy * RH : wy * (RH / 2), cbase = (S == 1) ? wx * RW : wx * (RW / 2);
#pragma unroll
    for (int jr = 0; jr < NJR; ++jr) {
      float dcr[NR4 * 4];
      const uint8_t* rowp = tdc + ((rb + jr) * DW) * CB * 2;
#pragma unroll
      for (int b = 0; b < NR4; ++b) tr16_run(rowp, cbase + b * 4, lane, dcr + b * 4);
#pragma unroll
      for (int o = 0; o < RH; ++o)
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const int ty = o + PAD - ky;
          if ((S == 1 || ((ty % 2 + 2) % 2) == 0) && (fdiv3(ty, S) - LO) == jr) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
              for (int kx = 0; kx < K; ++kx) {
                const int tx = r + PAD - kx;
                if (S == 1 || ((tx % 2 + 2) % 2) == 0) acc[o][r] = fmaf(dcr[fdiv3(tx, S) - LO], wf[ky * K + kx], acc[o][r]);
              }
          }
        }
    }
#pragma unroll
    for (int o = 0; o < RH; ++o)
#pragma unroll
      for (int r = 0; r < RW; ++r) *(uint16_t*)(tout + (((wy * RH + o) * TW + wx * RW + r) * CB + lane) * 2) = (uint16_t)cvt_pk_bf16(acc[o][r] * sw, 0.0f);
    __syncthreads();
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u = tid + jn * 256; const int c8 = u & 7, lp = u >> 3; const int iy = iy0 + lp / TW, ix = ix0 + lp % TW; const int cc = cb * CB + c8 * 8;
      if (iy < p.h && ix < p.w && cc < p.c) {
        uint16_t* dst = p.dx + (((int64_t)img * p.h + iy) * p.w + ix) * p.c + cc;
        uint4 v = *(const uint4*)(tout + (lp * CB + c8 * 8) * 2);
        if (p.accumulate) {
          const uint4 o = *(const uint4*)dst;
          v.x = cvt_pk_bf16(bf2f(v.x & 0xffff) + bf2f(o.x & 0xffff), bf2f(v.x >> 16) + bf2f(o.x >> 16));
          v.y = cvt_pk_bf16(bf2f(v.y & 0xffff) + bf2f(o.y & 0xffff), bf2f(v.y >> 16) + bf2f(o.y >> 16));
          v.z = cvt_pk_bf16(bf2f(v.z & 0xffff) + bf2f(o.z & 0xffff), bf2f(v.z >> 16) + bf2f(o.z >> 16));
          v.w = cvt_pk_bf16(bf2f(v.w & 0xffff) + bf2f(o.w & 0xffff), bf2f(v.w >> 16) + bf2f(o.w >> 16));
        }
        *(uint4*)dst = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static void fill3(Dw3P& p, const int8_t* x, const float* qx, const int8_t* wq, const int32_t* wsum, int n, int h, int w, int c, int k, int stride) {
  p.x = x; p.qx = qx; p.wq = wq; p.wsum = wsum; p.n = n; p.h = h; p.w = w; p.c = c; p.cpad = round_up(c, 16);
  p.pad = (k - 1) / 2; p.ho = (h + 2 * p.pad - k) / stride + 1; p.wo = (w + 2 * p.pad - k) / stride + 1;
  p.tiles_x = (p.wo + TW - 1) / TW; p.tiles_y = (p.ho + TH - 1) / TH; p.ncb = (c + CB - 1) / CB;
  p.ntiles = (int64_t)n * p.tiles_x * p.tiles_y; p.inv_count = 1.0f / (float)((int64_t)n * p.ho * p.wo);
}
template <typename KF>
static int launch3(KF kern, Dw3P& p, size_t lds, const char* what, hipStream_t s) {
  static std::map<const void*, int> occ_cache;     // keyed by kernel: all instantiations share this launcher's type
  const void* key = (const void*)kern;
  auto it = occ_cache.find(key);
  int occ = 1;
  if (it == occ_cache.end()) {
    hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, key, 256, lds) != hipSuccess || occ < 1) occ = 1;
    occ_cache[key] = occ;
  } else occ = it->second;
  if (occ > 8) occ = 8;
  int64_t want = (256 * occ) / p.ncb; if (want < 1) want = 1;
  p.ngroups = (int)(p.ntiles < want ? p.ntiles : want);
  hipLaunchKernelGGL(kern, dim3(p.ncb * p.ngroups), dim3(256), lds, s, p);
  return frost_check_launch(what);
}
template <int K, int S> static constexpr size_t in_bytes() { return (size_t)((((TH - 1) * S + K) * ((TW - 1) * S + K) * CB + 255) / 256) * 256 + 256; }
template <int K, int S, int MODE>
static int launch_fwd(Dw3P& p, hipStream_t s) { return launch3(k_dw3<K, S, MODE>, p, in_bytes<K, S>() + 16384 + 4 * 64 * 2 * 8 + 4 * 64 * 2 * 4, "dw", s); }
template <int MODE>
static int dispatch3(Dw3P& p, int k, int stride, hipStream_t s) {
  if (k == 3 && stride == 1) return launch_fwd<3, 1, MODE>(p, s);
  if (k == 3 && stride == 2) return launch_fwd<3, 2, MODE>(p, s);
  if (k == 5 && stride == 1) return launch_fwd<5, 1, MODE>(p, s);
  if (k == 5 && stride == 2) return launch_fwd<5, 2, MODE>(p, s);
  frost_set_error("dw: unsupported kernel/stride (k in {3,5}, stride in {1,2})"); return 1;
}
extern "C" int frost_dw_conv_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n,
                                 int h, int w, int c, int k, int stride, int mode, void* stats, const float* coef,
                                 const float* qrec_y, int relu, int8_t* y, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  Dw3P p = {}; fill3(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.stats = (uint8_t*)stats; p.coef = (float*)coef; p.qy = qrec_y; p.relu = relu; p.y = y;
  return mode == 0 ? dispatch3<D_STATS>(p, k, stride, as_stream(stream)) : dispatch3<D_EMIT>(p, k, stride, as_stream(stream));
}
extern "C" int frost_dw_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                                 const float* qrec_w, int n, int h, int w, int c, int k, int stride, int pass, float* coef,
                                 const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  Dw3P p = {}; fill3(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.coef = coef; p.qy = qrec_y; p.relu = relu; p.gout = gout; p.dc = dc; p.qw = qrec_w;
  return pass == 0 ? dispatch3<D_BRED>(p, k, stride, as_stream(stream)) : dispatch3<D_BDC>(p, k, stride, as_stream(stream));
}
extern "C" int frost_dw_wgrad(const uint16_t* dc, const int8_t* x, const float* qrec_x, int n, int h, int w, int c, int k,
                              int stride, float* dwq, void* stream) {
  Dw3P p = {}; fill3(p, x, qrec_x, nullptr, nullptr, n, h, w, c, k, stride); p.dc = (uint16_t*)dc; p.dwq = dwq;
  hipStream_t s = as_stream(stream);
  if (k == 3 && stride == 1) return launch3(k_dw3_wgrad<3, 1>, p, in_bytes<3, 1>() + 16384, "dw_wgrad", s);
  if (k == 3 && stride == 2) return launch3(k_dw3_wgrad<3, 2>, p, in_bytes<3, 2>() + 16384, "dw_wgrad", s);
  if (k == 5 && stride == 1) return launch3(k_dw3_wgrad<5, 1>, p, (in_bytes<5, 1>() + 16384 > 25 * 64 * 16 ? in_bytes<5, 1>() + 16384 : 25 * 64 * 16), "dw_wgrad", s);
  if (k == 5 && stride == 2) return launch3(k_dw3_wgrad<5, 2>, p, in_bytes<5, 2>() + 16384, "dw_wgrad", s);
  frost_set_error("dw_wgrad: unsupported kernel/stride"); return 1;
}
template <int K, int S> static constexpr size_t dgrad_bytes() {
  constexpr int PAD = (K - 1) / 2; constexpr int LO = fdiv3(-PAD, S);
  constexpr int DH = (TH - 1 + PAD) / S - LO + 1, DW = (TW - 1 + PAD) / S - LO + 1;
  return (size_t)((DH * DW * CB * 2 + 255) / 256) * 256 + 512 + 16384;
}
extern "C" int frost_dw_dgrad(const uint16_t* dc, const int8_t* wq_pack, const float* qrec_w, int n, int h, int w, int c,
                              int k, int stride, uint16_t* dx, int accumulate, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw_dgrad: channels must be a multiple of 8");
  Dw3P p = {}; fill3(p, nullptr, nullptr, wq_pack, nullptr, n, h, w, c, k, stride);
  p.dc = (uint16_t*)dc; p.dx = dx; p.accumulate = accumulate; p.qw = qrec_w;
  p.tiles_x = (w + TW - 1) / TW; p.tiles_y = (h + TH - 1) / TH; p.ntiles = (int64_t)n * p.tiles_x * p.tiles_y;   // tiles over dx
  hipStream_t s = as_stream(stream);
  if (k == 3 && stride == 1) return launch3(k_dw3_dgrad<3, 1>, p, dgrad_bytes<3, 1>(), "dw_dgrad", s);
  if (k == 3 && stride == 2) return launch3(k_dw3_dgrad<3, 2>, p, dgrad_bytes<3, 2>(), "dw_dgrad", s);
  if (k == 5 && stride == 1) return launch3(k_dw3_dgrad<5, 1>, p, dgrad_bytes<5, 1>(), "dw_dgrad", s);
  if (k == 5 && stride == 2) return launch3(k_dw3_dgrad<5, 2>, p, dgrad_bytes<5, 2>(), "dw_dgrad", s);
  frost_set_error("dw_dgrad: unsupported kernel/stride"); return 1;
}
