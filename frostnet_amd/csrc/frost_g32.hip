// fp32-GRADIENT parity mode of the fake-quant backward (reference: loss.backward() is fp32 autograd, Classification/utils/helper_functions.py:139-143).
//
// The production backward stores activation gradients and dc in bf16 (stochastically rounded) and feeds bf16 operands to the MFMA data / weight gradient
// GEMMs: measured 2-9e-3 from an fp64 evaluation per layer, 2-3e-2 on ill-conditioned sums.  This file is the SAME backward -- the same masks (STE window of the
// activation fake-quantise incl. ReLU, evaluated on the exact integer conv output with the forward's coefficient rows), the same BatchNorm expression
// dc = K1 (gy - S1/n - xhat S2/n), the same fake-quantised weights / inputs in the data / weight gradient -- with every gradient held in fp32 and every
// long sum accumulated in fp64.  Round 4 wrote it as plain one-thread-per-output kernels (a parity instrument, 30 x slower than the production step); since round 5
// the same entries run tiled / coalesced kernels (section "fast forms" below: the pointwise conv output on the int8 MFMA, pointwise data / weight gradients on
// v_mfma_f32_16x16x4_f32 with fp32 operands, or on the bf16 MFMA with the fp32 operand split three ways (default), per-channel sums as deterministic two-stage reductions with fp64 partials, every element-wise pass with a fixed
// channel quad per thread) so that a user can TRAIN at the reference's gradient precision; the plain kernels stay as the fallback for channel counts that are not
// a multiple of 4.  It shows that the formulas meet the reference's fp32 autograd at <= 1e-3, and what the bf16 storage costs (tests/test_gpu_round4.py).
// Entry points mirror the production passes: frost_g32_conv_acc (integer conv output) -> frost_g32_reduce -> frost_g32_dc -> frost_g32_dgrad / frost_g32_wgrad,
// then the ordinary frost_weight_grad_finalize(_table).
#include "frost_common.h"

// fake-quantised weight INDICES in OIHW order [cout][per] (per = cin_g * k * k): q = clamp(rint(W * sf / s_w), -128, 127), sf = gamma / sigma_r -- the
// expression of frost_weight_prep and of the parameter-gradient finalize (gamma == NULL: the classifier, sf = 1)
__global__ __launch_bounds__(256) void k_g32_wq(const float* __restrict__ w, const float* gamma, const float* sigma, const float* qw, const float* wscale,
                                                int cout, int per, int8_t* __restrict__ out) {
  const int64_t tot = (int64_t)cout * per;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i / per);
    const float sf = gamma ? gamma[co] / sigma[co] : 1.0f;
    const float inv = 1.0f / (wscale ? wscale[co] : qw[FROST_Q_SCALE]);
    out[i] = (int8_t)fq_index(w[i] * sf, inv, 0, -128, 127);
  }
}
extern "C" int frost_g32_wq(const float* w, const float* gamma, const float* sigma, const float* qrec_w, const float* wscale, int cout, int per, int8_t* out,
                            void* stream) {
  int64_t grid = ((int64_t)cout * per + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_g32_wq, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), w, gamma, sigma, qrec_w, wscale, cout, per, out);
  return frost_check_launch("g32_wq");
}

// geometry of one conv layer: kind 0 = pointwise (x: [npix][xc], K = cin), 1 = depthwise k x k (x: [n][h][w][c]), 2 = stem on the im2col'd input
// (x: [npix][40], K index tap*4 + c; weights OIHW c*9 + tap)
struct G32Geo { int kind, n, h, w, ho, wo, xc, cin_g, cout, k, stride, pad; };

__device__ __forceinline__ int g32_x(const int8_t* x, int64_t i) { return (int)x[i] + 128; }

// acc[p][co] = sum (q_x - zp_x) * q_w   (exact int32; the quantity the coefficient rows A / B / M / R are defined on)
__global__ __launch_bounds__(256) void k_g32_conv_acc(const int8_t* __restrict__ x, const float* qx, const int8_t* __restrict__ qw, G32Geo g, int32_t* __restrict__ acc) {
  const int zp = __float_as_int(qx[FROST_Q_ZP]);
  const int64_t npo = (int64_t)g.n * g.ho * g.wo, tot = npo * g.cout;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i % g.cout); const int64_t p = i / g.cout;
    int s = 0;
    if (g.kind == 0) {
      const int8_t* xr = x + p * g.xc; const int8_t* wr = qw + (int64_t)co * g.cin_g;
      for (int k = 0; k < g.cin_g; ++k) s += (g32_x(xr, k) - zp) * (int)wr[k];
    } else if (g.kind == 2) {
      const int8_t* xr = x + p * g.xc; const int8_t* wr = qw + (int64_t)co * g.cin_g * 9;
      for (int t = 0; t < 9; ++t) for (int c = 0; c < g.cin_g; ++c) s += (g32_x(xr, t * 4 + c) - zp) * (int)wr[c * 9 + t];
    } else {
      const int ox = (int)(p % g.wo); const int oy = (int)((p / g.wo) % g.ho); const int in = (int)(p / ((int64_t)g.wo * g.ho));
      for (int ky = 0; ky < g.k; ++ky) { const int iy = oy * g.stride - g.pad + ky; if (iy < 0 || iy >= g.h) continue;
        for (int kx = 0; kx < g.k; ++kx) { const int ix = ox * g.stride - g.pad + kx; if (ix < 0 || ix >= g.w) continue;
          s += (g32_x(x, (((int64_t)in * g.h + iy) * g.w + ix) * g.xc + co) - zp) * (int)qw[(int64_t)co * g.k * g.k + ky * g.k + kx]; } }
    }
    acc[i] = s;
  }
}
// STE window of the activation fake-quantise on t = fma(A, acc, B) / s_y (k_pw_ew's expression, ReLU included)
struct G32Win { float y_inv, t_lo, t_hi; };
__device__ __forceinline__ G32Win g32_win(const float* qy, int relu) {
  G32Win w; w.y_inv = 1.0f / qy[FROST_Q_SCALE];
  const int zpy = __float_as_int(qy[FROST_Q_ZP]), qhi = q_hi(qy);
  const float hi0 = (float)qhi + 0.5f - (float)zpy;
  w.t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
  w.t_lo = 0.0f;
  if (!relu) { const float lo0 = -(float)zpy - 0.5f; w.t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  return w;
}


// ================================================================================================ fast forms (round 5)
// Shared thread map of the channel-parallel passes: a thread owns ONE channel quad (4 consecutive channels: 16-byte loads of fp32 / int32 rows, 4-byte loads of
// int8 rows) for the whole launch and walks pixels; blockIdx.y selects the block of 256 quads for layers wider than 1024 channels.  cq: quad index, pl / PL: the
// thread's pixel lane and the number of pixel lanes per workgroup.
__device__ __forceinline__ int g32_sb(int v, int b) { return (int)(int8_t)(v >> (8 * b)); }      // signed byte b of a dword (v_bfe_i32)
struct G32Map { int cq, pl, PL; bool ok; };
__device__ __forceinline__ G32Map g32_map(int c) {
  const int CQ = c >> 2, CQB = min(CQ, 256);
  G32Map m; m.PL = 256 / CQB; m.pl = (int)threadIdx.x / CQB; m.cq = (int)blockIdx.y * 256 + (int)threadIdx.x % CQB;
  m.ok = m.pl < m.PL && m.cq < CQ;
  return m;
}
static inline dim3 g32_map_grid(int c, int64_t blocks_x) { return dim3((unsigned)blocks_x, (unsigned)(((c >> 2) + 255) / 256)); }
static inline int64_t g32_run_blocks(int c, int64_t np, int target) {      // blockIdx.x extent: runs of ~`target` pixels per thread, at most 4096 blocks
  const int CQB = (c >> 2) < 256 ? (c >> 2) : 256; const int PL = 256 / CQB;
  int64_t bx = (np + (int64_t)PL * target - 1) / ((int64_t)PL * target);
  if (bx > 4096) bx = 4096; if (bx < 1) bx = 1;
  return bx;
}
static inline int g32_map_pl(int c) { const int CQB = (c >> 2) < 256 ? (c >> 2) : 256; return 256 / CQB; }

// ---- pointwise (kind 0) / stem-on-im2col (kind 2) integer conv output on v_mfma_i32_16x16x64_i8: D[co][pixel], one wave = MI x 4 tiles of 16 x 16, operands straight
// from global memory (rows of q_w and of x are contiguous in k); the zero-point term (128 - zp) * sum_k q_w comes from one more MFMA against an all-ones operand
__device__ __forceinline__ v4i g32_row16(const int8_t* row, int k0, int klim) {      // 16 bytes of a row at k0 (k0 % 4 == 0, klim % 4 == 0), zero past klim
  v4i v = {0, 0, 0, 0};
  if (k0 + 16 <= klim) { v[0] = *(const int*)(row + k0); v[1] = *(const int*)(row + k0 + 4); v[2] = *(const int*)(row + k0 + 8); v[3] = *(const int*)(row + k0 + 12); }
  else {
#pragma unroll
    for (int d = 0; d < 4; ++d) if (k0 + 4 * d < klim) v[d] = *(const int*)(row + k0 + 4 * d);
  }
  return v;
}
// EPI 0: store the integer conv output (acc);  EPI 1 / 2: the REDUCE / DC pass of the backward straight from the accumulators -- the int32 tensor is never written or read
// (24 -> 12 bytes per output element over the three passes; the int8 GEMM is recomputed, it costs a fraction of the bytes it saves).  EPI 1: a wave walks pixel blocks
// pb = its index, += the launch's wave count, keeps fp64 sums of its 4 * MI channels per lane and writes ONE partial row part[wave][2][cout] at the end (k_g32_reduce_fin adds
// the rows in a fixed order); EPI 2: dc = K1 (gy - S1 / n - xhat S2 / n) as 16-byte stores.  Coefficient rows of the block's channels sit in LDS.
struct G32Epi { const float* coef; int cpad; const float* qy; int relu; const float* gout; double* part; float* dc; };
template <int MI, int EPI>
__global__ __launch_bounds__(256, 2) void k_g32_pw(const int8_t* __restrict__ x, const float* qx, const int8_t* __restrict__ qw, G32Geo g, int32_t* __restrict__ acc, G32Epi e) {
  constexpr int NJ = 4;
  __shared__ float rows[7][MI * 16];                             // A, B, M, R, K1, S1, S2 of this block's channels
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, gq = lane >> 4;
  const int64_t npix = (int64_t)g.n * g.ho * g.wo;
  const int co0 = (int)blockIdx.y * (MI * 16);
  const int K = (g.kind == 2) ? g.xc : g.cin_g;                  // stem: the im2col columns tap * 4 + c
  const int zp = __float_as_int(qx[FROST_Q_ZP]);
  G32Win win = {1.0f, 0.0f, 0.0f};
  if (EPI != 0) {
    for (int i = threadIdx.x; i < 7 * MI * 16; i += 256) {
      const int r = i / (MI * 16), c = i - r * (MI * 16);
      const int row = (r < 4) ? r : (r == 4 ? FROST_COEF_K1 : (r == 5 ? FROST_COEF_S1 : FROST_COEF_S2));          // rows 0 .. 3 are A, B, M, R
      rows[r][c] = (co0 + c < g.cout) ? e.coef[row * e.cpad + co0 + c] : 0.0f;
    }
    win = g32_win(e.qy, e.relu);
    __syncthreads();
  }
  const double inv_n = 1.0 / (double)npix;
  float s1[EPI == 1 ? MI : 1][4], s2[EPI == 1 ? MI : 1][4];          // fp32 over a lane's ~200 values of a channel, fp64 across lanes / waves / rows
  if (EPI == 1) {
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) { s1[m][i] = 0.0f; s2[m][i] = 0.0f; }
  }
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
  const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  for (int64_t pb = wave; pb * (NJ * 16) < npix; pb += nwaves) {
    const int64_t p0 = pb * (NJ * 16);
    v4i d[MI][NJ], dw[MI];
#pragma unroll
    for (int m = 0; m < MI; ++m) { dw[m] = (v4i){0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < NJ; ++t) d[m][t] = (v4i){0, 0, 0, 0}; }
    const int8_t* xr[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) xr[t] = x + min(p0 + 16 * t + j, npix - 1) * g.xc;
    if (g.kind == 0 && K >= 16) {
      // pointwise: operands by unconditional 16-byte loads at k clamped into the row (no branch around a load), two operand sets in turn so that the next K step's
      // loads are in flight under this step's MFMAs; a clamped load repeats bytes another lane group / step already contributed: those dwords of the A operand are zeroed
      struct Op { v4i a[MI]; v4i b[NJ]; };
      const int8_t* wr[MI]; bool cok[MI];
#pragma unroll
      for (int m = 0; m < MI; ++m) { const int co = co0 + 16 * m + j; cok[m] = co < g.cout; wr[m] = qw + (int64_t)min(co, g.cout - 1) * g.cin_g; }
      auto ld = [&](int k0) __attribute__((always_inline)) {
        Op o; const int kc = min(k0 + 16 * gq, K - 16);
#pragma unroll
        for (int m = 0; m < MI; ++m) { const int8_t* r = wr[m] + kc; o.a[m] = (v4i){*(const int*)r, *(const int*)(r + 4), *(const int*)(r + 8), *(const int*)(r + 12)}; }
#pragma unroll
        for (int t = 0; t < NJ; ++t) { const int8_t* r = xr[t] + kc; o.b[t] = (v4i){*(const int*)r, *(const int*)(r + 4), *(const int*)(r + 8), *(const int*)(r + 12)}; }
        return o;
      };
      auto step = [&](const Op& o, int k0) __attribute__((always_inline)) {
        const int kreq = k0 + 16 * gq, kc = min(kreq, K - 16);
        v4i a[MI];
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) a[m][dd] = (cok[m] && (kc + 4 * dd) >= kreq) ? o.a[m][dd] : 0;
#pragma unroll
        for (int m = 0; m < MI; ++m) dw[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], ones, dw[m], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NJ; ++t)
#pragma unroll
          for (int m = 0; m < MI; ++m) d[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], o.b[t], d[m][t], 0, 0, 0);
      };
      Op o0 = ld(0);
      for (int k0 = 0; k0 < K; k0 += 128) {
        const Op o1 = ld(k0 + 64);
        step(o0, k0);
        o0 = ld(k0 + 128);
        if (k0 + 64 < K) step(o1, k0 + 64);
      }
    } else
    for (int k0 = 0; k0 < K; k0 += 64) {
      v4i a[MI];
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int co = co0 + 16 * m + j;
        a[m] = (v4i){0, 0, 0, 0};
        if (co < g.cout) {
          if (g.kind == 0) a[m] = g32_row16(qw + (int64_t)co * g.cin_g, k0 + 16 * gq, K);
          else {                                                   // stem weights OIHW [co][c * 9 + tap] -> column tap * 4 + c
            const int8_t* wr = qw + (int64_t)co * g.cin_g * 9;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
              uint32_t pk = 0;
#pragma unroll
              for (int b = 0; b < 4; ++b) {
                const int col = k0 + 16 * gq + 4 * dd + b, tap = col >> 2, c = col & 3;
                if (tap < 9 && c < g.cin_g) pk |= (uint32_t)(uint8_t)wr[c * 9 + tap] << (8 * b);
              }
              a[m][dd] = (int)pk;
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MI; ++m) dw[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], ones, dw[m], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NJ; ++t) {
        const v4i b = g32_row16(xr[t], k0 + 16 * gq, K);
#pragma unroll
        for (int m = 0; m < MI; ++m) d[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], b, d[m][t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < MI; ++m) {
      const int cl = 16 * m + 4 * gq, co = co0 + cl;
      if (co >= g.cout) continue;
      float rA[4], rB[4], rM[4], rR[4]; double rK[4], rS1[4], rS2[4];          // this lane's four channels: read from LDS once per channel tile, not once per pixel
      if (EPI != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rA[i] = rows[0][cl + i]; rB[i] = rows[1][cl + i]; rM[i] = rows[2][cl + i]; rR[i] = rows[3][cl + i];
          if (EPI == 2) { rK[i] = (double)rows[4][cl + i]; rS1[i] = (double)rows[5][cl + i] * inv_n; rS2[i] = (double)rows[6][cl + i] * inv_n; }
        }
      }
#pragma unroll
      for (int t = 0; t < NJ; ++t) {
        const int64_t p = p0 + 16 * t + j;
        if (p >= npix) continue;
        v4i o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = d[m][t][i] + (128 - zp) * dw[m][i];
        if (EPI == 0) { *(v4i*)(acc + p * g.cout + co) = o; continue; }
        const v4f gy4 = *(const v4f*)(e.gout + p * g.cout + co);
        v4f dcv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float af = (float)o[i];
          const float tq = fmaf(rA[i], af, rB[i]) * win.y_inv;
          const bool in = tq > win.t_lo && tq <= win.t_hi;
          if (EPI == 1) { if (in) { const float gy = gy4[i]; s1[m][i] += gy; s2[m][i] = fmaf(gy, (af - rM[i]) * rR[i], s2[m][i]); } }
          else {
            const double xhat = ((double)af - (double)rM[i]) * (double)rR[i];
            dcv[i] = (float)(rK[i] * ((in ? (double)gy4[i] : 0.0) - rS1[i] - xhat * rS2[i]));
          }
        }
        if (EPI == 2) *(v4f*)(e.dc + p * g.cout + co) = dcv;
      }
    }
  }
  if (EPI == 1) {                                                  // the 16 pixel lanes of a lane group -> one value; one partial row per wave
    double* row = e.part + wave * 2 * (int64_t)g.cout;
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double a = (double)s1[m][i], b = (double)s2[m][i];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        const int co = co0 + 16 * m + 4 * gq + i;
        if (j == 0 && co < g.cout) { row[co] = a; row[g.cout + co] = b; }
      }
  }
}

// ---- depthwise kernels: a thread owns a channel quad and a CONTIGUOUS run of pixels (row = blockIdx.x * PL + pl of gridDim.x * PL rows): the pixel coordinates are
// divided out once and then stepped, the taps of the quad are unpacked once into registers, K and the stride are compile-time
struct G32Run { int p, pe, x, y, n; };
__device__ __forceinline__ G32Run g32_run(const G32Map& mp, int np, int W, int H) {
  const int rows = (int)gridDim.x * mp.PL, row = (int)blockIdx.x * mp.PL + mp.pl;
  const int R = (np + rows - 1) / rows;
  G32Run r; r.p = min(row * R, np); r.pe = min(r.p + R, np);
  r.x = r.p % W; r.y = (r.p / W) % H; r.n = r.p / (W * H);
  return r;
}
#define G32_RUN_STEP(r, W, H) do { ++(r).p; if (++(r).x == (W)) { (r).x = 0; if (++(r).y == (H)) { (r).y = 0; ++(r).n; } } } while (0)

// the K x K input window of a channel quad as packed bytes in registers, sliding along the output row: a step loads its K * S new columns; an out-of-map entry is stored as
// the zero point (q - zp = 0), so no mask is left in the arithmetic; the start of a run and of a row reload the window
template <int K, int S>
struct G32Window {
  uint32_t win[K][K]; bool fresh = true;
  template <typename F> __device__ __forceinline__ void advance(int x, F&& entry) {
    if (fresh || x == 0) {
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int jc = 0; jc < K; ++jc) win[ky][jc] = entry(ky, jc);
      fresh = false;
    } else {
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
#pragma unroll
        for (int jc = 0; jc < K - S; ++jc) win[ky][jc] = win[ky][jc + S];
#pragma unroll
        for (int jc = (K - S > 0 ? K - S : 0); jc < K; ++jc) win[ky][jc] = entry(ky, jc);
      }
    }
  }
};
template <int K, int S>
__global__ __launch_bounds__(256) void k_g32_dw_acc(const int8_t* __restrict__ x, const float* qx, const int8_t* __restrict__ qw, G32Geo g, int32_t* __restrict__ acc) {
  constexpr int KK = K * K, PAD = (K - 1) / 2;
  const G32Map mp = g32_map(g.cout);
  if (!mp.ok) return;
  const int c0 = mp.cq * 4, zp = __float_as_int(qx[FROST_Q_ZP]);
  const uint32_t zpb = (uint32_t)(zp & 255) * 0x01010101u;
  int wq[KK][4];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int b = 0; b < 4; ++b) wq[t][b] = (int)qw[(int64_t)(c0 + b) * KK + t];
  G32Run r = g32_run(mp, g.n * g.ho * g.wo, g.wo, g.ho);
  G32Window<K, S> W;
  for (; r.p < r.pe;) {
    const int8_t* base = x + ((int64_t)r.n * g.h * g.w) * g.xc + c0;
    const int ix0 = r.x * S - PAD;
    W.advance(r.x, [&](int ky, int jc) __attribute__((always_inline)) {
      const int iy = r.y * S - PAD + ky, ix = ix0 + jc;
      const int iyc = min(max(iy, 0), g.h - 1), ixc = min(max(ix, 0), g.w - 1);
      const uint32_t v = (uint32_t)*(const int*)(base + (int64_t)(iyc * g.w + ixc) * g.xc) ^ 0x80808080u;          // the four indices q as unsigned bytes
      return (iy == iyc && ix == ixc) ? v : zpb;
    });
    v4i s = {0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const uint32_t xq = W.win[ky][kx];
#pragma unroll
        for (int b = 0; b < 4; ++b) s[b] += ((int)((xq >> (8 * b)) & 255u) - zp) * wq[ky * K + kx][b];
      }
    *(v4i*)(acc + (int64_t)r.p * g.cout + c0) = s;
    G32_RUN_STEP(r, g.wo, g.ho);
  }
}

// ---- reduce pass, stage 1: fp64 partial sums of a thread's channel quad over its pixels -> part[row][2][cpad4] (row = blockIdx.x * PL + pl); stage 2 adds the rows in
// a fixed order (deterministic, no atomics -- as the plain kernel)
__global__ __launch_bounds__(256) void k_g32_reduce_part(const int32_t* __restrict__ acc, int npix, int cout, int cpad, const float* __restrict__ coef, const float* qy, int relu,
                                                         const float* __restrict__ gout, double* __restrict__ part) {
  const G32Map mp = g32_map(cout);
  if (!mp.ok) return;
  const int c0 = mp.cq * 4;
  const G32Win w = g32_win(qy, relu);
  const v4f A = *(const v4f*)(coef + FROST_COEF_A * cpad + c0), B = *(const v4f*)(coef + FROST_COEF_B * cpad + c0);
  const v4f M = *(const v4f*)(coef + FROST_COEF_M * cpad + c0), R = *(const v4f*)(coef + FROST_COEF_R * cpad + c0);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int p = (int)blockIdx.x * mp.PL + mp.pl; p < npix; p += (int)gridDim.x * mp.PL) {
    const v4i a = *(const v4i*)(acc + (int64_t)p * cout + c0); const v4f gy4 = *(const v4f*)(gout + (int64_t)p * cout + c0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float af = (float)a[b];
      const float tq = fmaf(A[b], af, B[b]) * w.y_inv;
      if (tq > w.t_lo && tq <= w.t_hi) { const double gy = (double)gy4[b]; s1[b] += gy; s2[b] += gy * (((double)af - (double)M[b]) * (double)R[b]); }
    }
  }
  double* row = part + (int64_t)((int)blockIdx.x * mp.PL + mp.pl) * 2 * cout;
#pragma unroll
  for (int b = 0; b < 4; ++b) { row[c0 + b] = s1[b]; row[cout + c0 + b] = s2[b]; }
}
__global__ __launch_bounds__(256) void k_g32_reduce_fin(const double* __restrict__ part, int rows, int cout, int cpad, float* coef) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int r = tid; r < rows; r += 256) { a += part[(int64_t)r * 2 * cout + c]; b += part[(int64_t)r * 2 * cout + cout + c]; }
  sh[0][tid] = a; sh[1][tid] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; } __syncthreads(); }
  if (tid == 0) { coef[FROST_COEF_S1 * cpad + c] = (float)sh[0][0]; coef[FROST_COEF_S2 * cpad + c] = (float)sh[1][0]; }
}

// ---- dc pass with the same thread map (coefficient rows loaded once per thread)
__global__ __launch_bounds__(256) void k_g32_dc4(const int32_t* __restrict__ acc, int npix, int cout, int cpad, const float* __restrict__ coef, const float* qy, int relu,
                                                 const float* __restrict__ gout, float* __restrict__ dc) {
  const G32Map mp = g32_map(cout);
  if (!mp.ok) return;
  const int c0 = mp.cq * 4;
  const G32Win w = g32_win(qy, relu);
  const double inv_n = 1.0 / (double)npix;
  const v4f A = *(const v4f*)(coef + FROST_COEF_A * cpad + c0), B = *(const v4f*)(coef + FROST_COEF_B * cpad + c0);
  const v4f M = *(const v4f*)(coef + FROST_COEF_M * cpad + c0), R = *(const v4f*)(coef + FROST_COEF_R * cpad + c0);
  const v4f K1 = *(const v4f*)(coef + FROST_COEF_K1 * cpad + c0), S1 = *(const v4f*)(coef + FROST_COEF_S1 * cpad + c0), S2 = *(const v4f*)(coef + FROST_COEF_S2 * cpad + c0);
  for (int p = (int)blockIdx.x * mp.PL + mp.pl; p < npix; p += (int)gridDim.x * mp.PL) {
    const v4i a = *(const v4i*)(acc + (int64_t)p * cout + c0); const v4f gy4 = *(const v4f*)(gout + (int64_t)p * cout + c0);
    v4f o;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float af = (float)a[b];
      const float tq = fmaf(A[b], af, B[b]) * w.y_inv;
      const double gy = (tq > w.t_lo && tq <= w.t_hi) ? (double)gy4[b] : 0.0;
      const double xhat = ((double)af - (double)M[b]) * (double)R[b];
      o[b] = (float)((double)K1[b] * (gy - (double)S1[b] * inv_n - xhat * (double)S2[b] * inv_n));
    }
    *(v4f*)(dc + (int64_t)p * cout + c0) = o;
  }
}

typedef __bf16 g32_v8bf __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void g32_split3(const float (&v)[8], v4i& hi, v4i& mid, v4i& lo) {
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t xb = __float_as_uint(v[i]); h[i] = xb & 0xffff0000u;
    const float r1 = v[i] - __uint_as_float(h[i]); m[i] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m[i]); l[i] = __float_as_uint(r2) & 0xffff0000u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {          // two bf16 per dword: element 2i in the low half
    hi[i] = (int)((h[2 * i] >> 16) | h[2 * i + 1]); mid[i] = (int)((m[2 * i] >> 16) | m[2 * i + 1]); lo[i] = (int)((l[2 * i] >> 16) | l[2 * i + 1]);
  }
}
// ---- pointwise data gradient on v_mfma_f32_16x16x4_f32: D[ci][pixel] = sum_co wf[co][ci] * dc[pixel][co], wf = q_w * s_w (exact in fp32).  The four k slots of
// MFMA q of a 16-channel step hold co = kb + 4 g + q, so that a lane's dc operands of the four MFMAs are ONE 16-byte load of its pixel's row.
template <int MI>
__global__ __launch_bounds__(256) void k_g32_pw_dgrad(const float* __restrict__ dc, const int8_t* __restrict__ qw, const float* qrec_w, const float* wscale, G32Geo g,
                                                      float* __restrict__ gx, int accumulate) {
  constexpr int NJ = 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, gq = lane >> 4;
  const int64_t npix = (int64_t)g.n * g.h * g.w;
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * (NJ * 16);
  const int ci0 = (int)blockIdx.y * (MI * 16);
  if (p0 >= npix) return;
  const float sw0 = qrec_w[FROST_Q_SCALE];
  v4f d[MI][NJ];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int t = 0; t < NJ; ++t) d[m][t] = (v4f){0, 0, 0, 0};
  const float* dr[NJ]; float pm[NJ];
#pragma unroll
  for (int t = 0; t < NJ; ++t) { const int64_t p = p0 + 16 * t + j; pm[t] = (p < npix) ? 1.0f : 0.0f; dr[t] = dc + min(p, npix - 1) * g.cout; }
  // operands: unconditional loads at clamped indices, masked by a factor afterwards (no branch around a load), the next 16-channel step's issued before this step's MFMAs
  int cia[MI]; float cim[MI];
#pragma unroll
  for (int m = 0; m < MI; ++m) { const int ci = ci0 + 16 * m + j; cim[m] = (ci < g.cin_g) ? 1.0f : 0.0f; cia[m] = min(ci, g.cin_g - 1); }
  struct Op { int8_t w[MI][4]; float sc[4]; v4f b[NJ]; };
  auto ld = [&](int kb) __attribute__((always_inline)) {
    Op o; const int cok = min(kb + 4 * gq, g.cout - 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o.sc[q] = wscale ? wscale[cok + q] : sw0;
#pragma unroll
      for (int m = 0; m < MI; ++m) o.w[m][q] = qw[(int64_t)(cok + q) * g.cin_g + cia[m]];
    }
#pragma unroll
    for (int t = 0; t < NJ; ++t) o.b[t] = *(const v4f*)(dr[t] + cok);
    return o;
  };
  auto step = [&](const Op& o, int kb) __attribute__((always_inline)) {
    const float km = ((kb + 4 * gq) < g.cout) ? 1.0f : 0.0f;          // cout % 4 == 0: this lane's four output channels of the step are inside, or none
    float a[MI][4];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) a[m][q] = (float)o.w[m][q] * o.sc[q] * (km * cim[m]);
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
      const v4f b = o.b[t] * pm[t];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < MI; ++m) d[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][q], b[q], d[m][t], 0, 0, 0);
    }
  };
  Op o0 = ld(0);          // two operand sets in turn, no register copy between them
  for (int kb = 0; kb < g.cout; kb += 32) {
    const Op o1 = ld(min(kb + 16, g.cout - 4));
    step(o0, kb);
    o0 = ld(min(kb + 32, g.cout - 4));
    step(o1, kb + 16);          // past cout every lane is masked off (km = 0)
  }
#pragma unroll
  for (int m = 0; m < MI; ++m) {
    const int ci = ci0 + 16 * m + 4 * gq;
    if (ci >= g.xc) continue;
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
      const int64_t p = p0 + 16 * t + j;
      if (p >= npix) continue;
      float* dst = gx + p * g.xc + ci;
      v4f o = d[m][t];
      if (accumulate) { const v4f old = *(const v4f*)dst; o = old + o; }
      *(v4f*)dst = o;
    }
  }
}

// ---- the same data gradient on the bf16 MFMA with fp32 operands SPLIT into three bf16 pieces (hi + mid + lo = the fp32 value exactly: 3 x 8 mantissa bits, by truncation):
// dc * s_w as B operand (three pieces -> three MFMAs), the weight indices as A operand (an int8 is exact in bf16), products exact in fp32, accumulation in fp32 as before --
// the arithmetic of k_g32_pw_dgrad at 3 v_mfma_f32_16x16x32_bf16 per 32 output channels instead of 8 v_mfma_f32_16x16x4_f32 (2.7 x fewer cycles of the matrix pipe).
// A lane's B operand of a step is 8 consecutive channels of its pixel's dc row (two 16-byte loads).  cout % 8 == 0.
template <int MI>
__global__ __launch_bounds__(256, 2) void k_g32_pw_dgrad_b3(const float* __restrict__ dc, const int8_t* __restrict__ qw, const float* qrec_w, const float* wscale, G32Geo g,
                                                            float* __restrict__ gx, int accumulate) {
  constexpr int NJ = 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, gq = lane >> 4;
  const int64_t npix = (int64_t)g.n * g.h * g.w;
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * (NJ * 16);
  const int ci0 = (int)blockIdx.y * (MI * 16);
  if (p0 >= npix) return;
  const float sw0 = qrec_w[FROST_Q_SCALE];
  v4f d[MI][NJ];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int t = 0; t < NJ; ++t) d[m][t] = (v4f){0, 0, 0, 0};
  const float* dr[NJ]; float pm[NJ];
#pragma unroll
  for (int t = 0; t < NJ; ++t) { const int64_t p = p0 + 16 * t + j; pm[t] = (p < npix) ? 1.0f : 0.0f; dr[t] = dc + min(p, npix - 1) * g.cout; }
  int cia[MI]; bool cio[MI];
#pragma unroll
  for (int m = 0; m < MI; ++m) { const int ci = ci0 + 16 * m + j; cio[m] = ci < g.cin_g; cia[m] = min(ci, g.cin_g - 1); }
  struct Op { int8_t w[MI][8]; float sc[8]; v4f b[NJ][2]; };
  auto ld = [&](int kb) __attribute__((always_inline)) {
    Op o; const int cok = min(kb + 8 * gq, g.cout - 8);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      o.sc[q] = wscale ? wscale[cok + q] : sw0;
#pragma unroll
      for (int m = 0; m < MI; ++m) o.w[m][q] = qw[(int64_t)(cok + q) * g.cin_g + cia[m]];
    }
#pragma unroll
    for (int t = 0; t < NJ; ++t) { o.b[t][0] = *(const v4f*)(dr[t] + cok); o.b[t][1] = *(const v4f*)(dr[t] + cok + 4); }
    return o;
  };
  auto step = [&](const Op& o, int kb) __attribute__((always_inline)) {
    const bool kok = (kb + 8 * gq) < g.cout;                     // cout % 8 == 0: this lane's eight output channels of the step are inside, or none
    v4i a[MI];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t e0 = __float_as_uint((float)o.w[m][2 * i]) >> 16, e1 = __float_as_uint((float)o.w[m][2 * i + 1]) & 0xffff0000u;          // an int8 is exact in bf16
        a[m][i] = (kok && cio[m]) ? (int)(e0 | e1) : 0;
      }
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = o.b[t][q >> 2][q & 3] * o.sc[q] * pm[t];
      v4i hi, mid, lo;
      g32_split3(v, hi, mid, lo);
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        d[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g32_v8bf, a[m]), __builtin_bit_cast(g32_v8bf, hi), d[m][t], 0, 0, 0);
        d[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g32_v8bf, a[m]), __builtin_bit_cast(g32_v8bf, mid), d[m][t], 0, 0, 0);
        d[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g32_v8bf, a[m]), __builtin_bit_cast(g32_v8bf, lo), d[m][t], 0, 0, 0);
      }
    }
  };
  Op o0 = ld(0);          // two operand sets in turn, no register copy between them
  for (int kb = 0; kb < g.cout; kb += 64) {
    const Op o1 = ld(min(kb + 32, g.cout - 8));
    step(o0, kb);
    o0 = ld(min(kb + 64, g.cout - 8));
    step(o1, kb + 32);          // past cout every lane is masked off
  }
#pragma unroll
  for (int m = 0; m < MI; ++m) {
    const int ci = ci0 + 16 * m + 4 * gq;
    if (ci >= g.xc) continue;
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
      const int64_t p = p0 + 16 * t + j;
      if (p >= npix) continue;
      float* dst = gx + p * g.xc + ci;
      v4f o = d[m][t];
      if (accumulate) { const v4f old = *(const v4f*)dst; o = old + o; }
      *(v4f*)dst = o;
    }
  }
}

// ---- depthwise data gradient (fp32 sums of <= 25 terms; the weight scale is applied once, after the sum)
template <int K, int S>
__global__ __launch_bounds__(256) void k_g32_dw_dgrad(const float* __restrict__ dc, const int8_t* __restrict__ qw, const float* qrec_w, const float* wscale, G32Geo g,
                                                      float* __restrict__ gx, int accumulate) {
  constexpr int KK = K * K, PAD = (K - 1) / 2;
  const G32Map mp = g32_map(g.xc);
  if (!mp.ok) return;
  const int c0 = mp.cq * 4;
  const float sw0 = qrec_w[FROST_Q_SCALE];
  float sw[4], wf[KK][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) sw[b] = wscale ? wscale[c0 + b] : sw0;
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int b = 0; b < 4; ++b) wf[t][b] = (float)qw[(int64_t)(c0 + b) * KK + t];
  G32Run r = g32_run(mp, g.n * g.h * g.w, g.w, g.h);
  if (S == 1) {
    // stride 1: the K x K window of dc values around the pixel lives in registers and slides along the row (K new 16-byte loads per pixel instead of K * K); an entry
    // outside the map is 0, so no mask is left in the arithmetic; dx = sum win[r][c] * w[K - 1 - r][K - 1 - c]
    v4f win[K][K]; bool fresh = true;
    for (; r.p < r.pe;) {
      const float* base = dc + ((int64_t)r.n * g.ho * g.wo) * g.cout + c0;
      auto entry = [&](int rr, int cc) __attribute__((always_inline)) {
        const int oy = r.y - PAD + rr, ox = r.x - PAD + cc;
        const int oyc = min(max(oy, 0), g.ho - 1), oxc = min(max(ox, 0), g.wo - 1);
        const v4f v = *(const v4f*)(base + (int64_t)(oyc * g.wo + oxc) * g.cout);
        return (oy == oyc && ox == oxc) ? v : (v4f){0, 0, 0, 0};
      };
      if (fresh || r.x == 0) {
#pragma unroll
        for (int rr = 0; rr < K; ++rr)
#pragma unroll
          for (int cc = 0; cc < K; ++cc) win[rr][cc] = entry(rr, cc);
        fresh = false;
      } else {
#pragma unroll
        for (int rr = 0; rr < K; ++rr) {
#pragma unroll
          for (int cc = 0; cc < K - 1; ++cc) win[rr][cc] = win[rr][cc + 1];
          win[rr][K - 1] = entry(rr, K - 1);
        }
      }
      v4f sacc = {0, 0, 0, 0};
#pragma unroll
      for (int rr = 0; rr < K; ++rr)
#pragma unroll
        for (int cc = 0; cc < K; ++cc)
#pragma unroll
          for (int b = 0; b < 4; ++b) sacc[b] = fmaf(win[rr][cc][b], wf[(K - 1 - rr) * K + (K - 1 - cc)][b], sacc[b]);
      float* dst = gx + (int64_t)r.p * g.xc + c0;
      v4f o = {sacc[0] * sw[0], sacc[1] * sw[1], sacc[2] * sw[2], sacc[3] * sw[3]};
      if (accumulate) { const v4f old = *(const v4f*)dst; o = old + o; }
      *(v4f*)dst = o;
      G32_RUN_STEP(r, g.w, g.h);
    }
    return;
  }
  for (; r.p < r.pe;) {
    v4f sacc = {0, 0, 0, 0};
    const float* base = dc + ((int64_t)r.n * g.ho * g.wo) * g.cout + c0;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int ty = r.y + PAD - ky, oy = ty / S, oyc = min(max(oy, 0), g.ho - 1);
      const bool yok = ty >= 0 && (ty % S) == 0 && oy < g.ho;
      if (S == 2 && !yok) continue;                    // stride 2: the row parity is the same for every tap column (half of the rows drop out as a whole)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int tx = r.x + PAD - kx, ox = tx / S, oxc = min(max(ox, 0), g.wo - 1);
        const bool ok = yok && tx >= 0 && (tx % S) == 0 && ox < g.wo;
        const v4f dl = *(const v4f*)(base + (int64_t)(oyc * g.wo + oxc) * g.cout);          // issued for every tap (clamped): no branch between the loads
        const float m = ok ? 1.0f : 0.0f;
#pragma unroll
        for (int b = 0; b < 4; ++b) sacc[b] = fmaf(dl[b] * m, wf[ky * K + kx][b], sacc[b]);
      }
    }
    float* dst = gx + (int64_t)r.p * g.xc + c0;
    v4f o = {sacc[0] * sw[0], sacc[1] * sw[1], sacc[2] * sw[2], sacc[3] * sw[3]};
    if (accumulate) { const v4f old = *(const v4f*)dst; o = old + o; }
    *(v4f*)dst = o;
    G32_RUN_STEP(r, g.w, g.h);
  }
}

// ---- pointwise / stem weight gradient, stage 1, on v_mfma_f32_16x16x4_f32: D[co][col] = sum_pixels dc[p][co] * (q_x[p][col] - zp), K = the pixels of the wave's chunk
// (fp32 accumulation over <= chunk_px pixels; stage 2 adds the chunks in fp64).  One load of NA consecutive floats of a dc row gives a lane the A operands of NA co tiles
// (tile r holds co = co0 + NA m + r in row m), one load of NB consecutive bytes of the x row the B operands of NB column tiles (col = ci0 + NB n + r): NA x NB MFMAs per
// two loads; NA / NB = 1, 2, 4 by the layer's widths (a 16 -> 96 layer at 112 x 112 runs 4 x 1 tiles, not 4 x 4).  Partial tiles -> part[chunk][co][ncp] (fp32).
template <int NA, int NB>
__global__ __launch_bounds__(256) void k_g32_pw_wgrad_part(const float* __restrict__ dc, const int8_t* __restrict__ x, const float* qx, G32Geo g, int ncol, int ncp,
                                                           int chunk_px, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, gq = lane >> 4;
  const int64_t npix = (int64_t)g.n * g.ho * g.wo;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + wv;
  const int64_t lo = chunk * chunk_px, hi = min(lo + chunk_px, npix);
  if (lo >= npix) return;
  const int co0 = (int)blockIdx.y * (16 * NA), ci0 = (int)blockIdx.z * (16 * NB);
  const float zpf = (float)__float_as_int(qx[FROST_Q_ZP]);
  const bool aok = (co0 + NA * j) < g.cout, bok = (ci0 + NB * j) < ncol;          // cout % 4 == 0, ncol % 4 == 0 and NA, NB in {1, 2, 4}: a lane's run is inside or outside as a whole
  v4f d[NA][NB];
#pragma unroll
  for (int ra = 0; ra < NA; ++ra)
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) d[ra][rb] = (v4f){0, 0, 0, 0};
  // operands: every load unconditional at clamped addresses (a branch around a load puts its wait right behind it), the next step's issued before this step's MFMAs
  const float* ap = dc + min(co0 + NA * j, g.cout - NA); const int8_t* bp = x + min(ci0 + NB * j, ncol - NB);
  struct Op { float a[NA]; uint32_t xq; };
  auto ld = [&](int64_t pb) __attribute__((always_inline)) {
    Op o; const int64_t p = min(pb + gq, hi - 1);
    const float* sa = ap + p * g.cout; const int8_t* sb = bp + p * g.xc;
    if (NA == 4) { const v4f v = *(const v4f*)sa; o.a[0] = v[0]; o.a[1 % NA] = v[1]; o.a[2 % NA] = v[2]; o.a[3 % NA] = v[3]; }
    else if (NA == 2) { const float2 v = *(const float2*)sa; o.a[0] = v.x; o.a[1 % NA] = v.y; }
    else o.a[0] = *sa;
    if (NB == 4) o.xq = *(const uint32_t*)sb; else if (NB == 2) o.xq = *(const uint16_t*)sb; else o.xq = *(const uint8_t*)sb;
    return o;
  };
  auto step = [&](const Op& o, int64_t pb) __attribute__((always_inline)) {
    const bool ok = (pb + gq) < hi;
    const float ma = (ok && aok) ? 1.0f : 0.0f, mb = (ok && bok) ? 1.0f : 0.0f;
    float a[NA], b[NB];
    const uint32_t xq = o.xq ^ 0x80808080u;
#pragma unroll
    for (int r = 0; r < NA; ++r) a[r] = o.a[r] * ma;
#pragma unroll
    for (int r = 0; r < NB; ++r) b[r] = ((float)((xq >> (8 * r)) & 255u) - zpf) * mb;
#pragma unroll
    for (int ra = 0; ra < NA; ++ra)
#pragma unroll
      for (int rb = 0; rb < NB; ++rb) d[ra][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ra], b[rb], d[ra][rb], 0, 0, 0);
  };
  // two operand sets in turn (no register copy between them: a copy would wait for the load it copies)
  // a ring of four operand sets, loads three steps ahead (a step is 16 MFMAs = 512 cycles of the matrix pipe; a global load takes ~2000)
  Op o0 = ld(lo), o1 = ld(min(lo + 4, hi - 1)), o2 = ld(min(lo + 8, hi - 1));
  for (int64_t pb = lo; pb < hi; pb += 16) {
    const Op o3 = ld(min(pb + 12, hi - 1));
    step(o0, pb);
    o0 = ld(min(pb + 16, hi - 1));
    step(o1, pb + 4);          // past `hi` every lane is masked off
    o1 = ld(min(pb + 20, hi - 1));
    step(o2, pb + 8);
    o2 = ld(min(pb + 24, hi - 1));
    step(o3, pb + 12);
  }
  float* prt = part + chunk * (int64_t)g.cout * ncp;
#pragma unroll
  for (int ra = 0; ra < NA; ++ra)
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = co0 + NA * (4 * gq + i) + ra, col = ci0 + NB * j + rb;
        if (co < g.cout && col < ncp) prt[(int64_t)co * ncp + col] = d[ra][rb][i];
      }
}
// ---- the same weight gradient on the bf16 MFMA: dc split three ways (hi + mid + lo exact) as A operand, the input indices q - zp (integers of magnitude <= 255: exact in
// bf16) as B operand, K = 32 pixels per MFMA triple; a lane gathers its eight pixels of a channel column row by row (the 16 lanes of a group read 64 consecutive bytes of a dc
// row, 16 consecutive bytes of an x row).  Tile = 16 NA output channels x 16 NB columns, D[co][col]; partial tiles as in k_g32_pw_wgrad_part.
template <int NA, int NB>
__global__ __launch_bounds__(256, 2) void k_g32_pw_wgrad_part_b3(const float* __restrict__ dc, const int8_t* __restrict__ x, const float* qx, G32Geo g, int ncol, int ncp,
                                                                 int chunk_px, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, gq = lane >> 4;
  const int64_t npix = (int64_t)g.n * g.ho * g.wo;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + wv;
  const int64_t lo = chunk * chunk_px, hi = min(lo + chunk_px, npix);
  if (lo >= npix) return;
  const int co0 = (int)blockIdx.y * (16 * NA), ci0 = (int)blockIdx.z * (16 * NB);
  const int zp = __float_as_int(qx[FROST_Q_ZP]);
  v4f d[NA][NB];
#pragma unroll
  for (int ra = 0; ra < NA; ++ra)
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) d[ra][rb] = (v4f){0, 0, 0, 0};
  const float* ap[NA]; bool aok[NA]; const int8_t* bp[NB]; bool bok[NB];
#pragma unroll
  for (int ra = 0; ra < NA; ++ra) { const int co = co0 + 16 * ra + j; aok[ra] = co < g.cout; ap[ra] = dc + min(co, g.cout - 1); }
#pragma unroll
  for (int rb = 0; rb < NB; ++rb) { const int ci = ci0 + 16 * rb + j; bok[rb] = ci < ncol; bp[rb] = x + min(ci, ncol - 1); }
  for (int64_t pb = lo; pb < hi; pb += 32) {
    int64_t pq[8]; float pmk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int64_t p = pb + 8 * gq + q; pmk[q] = (p < hi) ? 1.0f : 0.0f; pq[q] = min(p, hi - 1); }
    v4i ah[NA], am[NA], al[NA], bb[NB];
#pragma unroll
    for (int ra = 0; ra < NA; ++ra) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = ap[ra][pq[q] * g.cout];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = aok[ra] ? v[q] * pmk[q] : 0.0f;
      g32_split3(v, ah[ra], am[ra], al[ra]);
    }
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
      int xb[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) xb[q] = (int)bp[rb][pq[q] * g.xc];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float f0 = bok[rb] ? (float)(xb[2 * i] + 128 - zp) : 0.0f, f1 = bok[rb] ? (float)(xb[2 * i + 1] + 128 - zp) : 0.0f;          // |q - zp| <= 255: exact in bf16
        bb[rb][i] = (int)((__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u));
      }
    }
#pragma unroll
    for (int ra = 0; ra < NA; ++ra)
#pragma unroll
      for (int rb = 0; rb < NB; ++rb) {
        d[ra][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g32_v8bf, ah[ra]), __builtin_bit_cast(g32_v8bf, bb[rb]), d[ra][rb], 0, 0, 0);
        d[ra][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g32_v8bf, am[ra]), __builtin_bit_cast(g32_v8bf, bb[rb]), d[ra][rb], 0, 0, 0);
        d[ra][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g32_v8bf, al[ra]), __builtin_bit_cast(g32_v8bf, bb[rb]), d[ra][rb], 0, 0, 0);
      }
  }
  float* prt = part + chunk * (int64_t)g.cout * ncp;
#pragma unroll
  for (int ra = 0; ra < NA; ++ra)
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = co0 + 16 * ra + 4 * gq + i, col = ci0 + 16 * rb + j;
        if (co < g.cout && col < ncp) prt[(int64_t)co * ncp + col] = d[ra][rb][i];
      }
}
// stage 2 of the weight gradients: out[e] = scale * sum_rows part[row][e'] in fp64, in a fixed order (16 row lanes per element, a fixed tree over them: deterministic).
// kind 2 maps the OIHW index c * 9 + tap to the im2col column tap * 4 + c.
template <int EL>            // EL element lanes x 256 / EL row lanes per workgroup: 16 x 16, or 4 x 64 for a layer with few weights and many partial rows (a narrow depthwise layer at 112 x 112)
__global__ __launch_bounds__(256) void k_g32_sum_part(const float* __restrict__ part, int rows, int cout, int per, int ncp, int kind, const float* qx, float* __restrict__ out) {
  constexpr int RL = 256 / EL;
  __shared__ double sh[RL][EL + 1];
  const int el = threadIdx.x % EL, rl = threadIdx.x / EL;
  const int e = (int)blockIdx.x * EL + el;
  double s = 0.0;
  if (e < cout * per) {
    const int co = e / per, jj = e - co * per;
    const int col = (kind == 2) ? ((jj % 9) * 4 + jj / 9) : jj;
    const int64_t stride = (int64_t)cout * ncp;
    const float* src = part + (int64_t)co * ncp + col;
    for (int r = rl; r < rows; r += RL) s += (double)src[r * stride];
  }
  sh[rl][el] = s;
  __syncthreads();
  if (rl == 0 && e < cout * per) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < RL; ++r) t += sh[r][el];
    out[e] = (float)(t * (double)qx[FROST_Q_SCALE]);
  }
}
static void g32_sum_part(const float* part, int rows, int cout, int per, int ncp, int kind, const float* qx, float* out, hipStream_t s) {
  const int ne = cout * per;
  if (ne <= 8192 && rows >= 1024) hipLaunchKernelGGL((k_g32_sum_part<4>), dim3((unsigned)((ne + 3) / 4)), dim3(256), 0, s, part, rows, cout, per, ncp, kind, qx, out);
  else hipLaunchKernelGGL((k_g32_sum_part<16>), dim3((unsigned)((ne + 15) / 16)), dim3(256), 0, s, part, rows, cout, per, ncp, kind, qx, out);
}
// ---- depthwise weight gradient, stage 1: a thread = one channel quad over its run of pixels, fp32 sums per tap -> part[row][c][kk] (fp32; a run is <= ~512 pixels).
// The K x K input window of the quad lives in registers as packed bytes and SLIDES along the row: a step loads its K * S new columns (5 instead of 25 loads for k = 5,
// stride 1), an out-of-map entry is stored as the zero point (q - zp = 0: no mask in the arithmetic); a new row or the start of the run reloads the window.
template <int K, int S>
__global__ __launch_bounds__(256) void k_g32_dw_wgrad_part(const float* __restrict__ dc, const int8_t* __restrict__ x, const float* qx, G32Geo g, float* __restrict__ part) {
  constexpr int KK = K * K, PAD = (K - 1) / 2;
  const G32Map mp = g32_map(g.cout);
  if (!mp.ok) return;
  const int c0 = mp.cq * 4;
  const int zp = __float_as_int(qx[FROST_Q_ZP]);
  const float zpf = (float)zp;
  const uint32_t zpb = (uint32_t)(zp & 255) * 0x01010101u;
  float sf[KK][4];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int b = 0; b < 4; ++b) sf[t][b] = 0.0f;
  G32Run r = g32_run(mp, g.n * g.ho * g.wo, g.wo, g.ho);
  const int row = (int)blockIdx.x * mp.PL + mp.pl;
  uint32_t win[K][K];
  bool fresh = true;
  for (; r.p < r.pe;) {
    const int8_t* base = x + ((int64_t)r.n * g.h * g.w) * g.xc + c0;
    const int ix0 = r.x * S - PAD;
    auto entry = [&](int ky, int jc) __attribute__((always_inline)) {          // window entry (ky, column jc) of the current pixel: the four indices q as unsigned bytes
      const int iy = r.y * S - PAD + ky, ix = ix0 + jc;
      const int iyc = min(max(iy, 0), g.h - 1), ixc = min(max(ix, 0), g.w - 1);
      const uint32_t v = (uint32_t)*(const int*)(base + (int64_t)(iyc * g.w + ixc) * g.xc) ^ 0x80808080u;
      return (iy == iyc && ix == ixc) ? v : zpb;
    };
    if (fresh || r.x == 0) {
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int jc = 0; jc < K; ++jc) win[ky][jc] = entry(ky, jc);
      fresh = false;
    } else {
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
#pragma unroll
        for (int jc = 0; jc < K - S; ++jc) win[ky][jc] = win[ky][jc + S];
#pragma unroll
        for (int jc = (K - S > 0 ? K - S : 0); jc < K; ++jc) win[ky][jc] = entry(ky, jc);
      }
    }
    const v4f dv = *(const v4f*)(dc + (int64_t)r.p * g.cout + c0);
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const uint32_t xq = win[ky][kx];
#pragma unroll
        for (int b = 0; b < 4; ++b) sf[ky * K + kx][b] = fmaf(dv[b], (float)((xq >> (8 * b)) & 255u) - zpf, sf[ky * K + kx][b]);
      }
    G32_RUN_STEP(r, g.wo, g.ho);
  }
  float* dst = part + ((int64_t)row * g.cout + c0) * KK;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int t = 0; t < KK; ++t) dst[b * KK + t] = sf[t][b];
}

static G32Geo g32_geo(int kind, int n, int h, int w, int xc, int cin_g, int cout, int k, int stride) {
  G32Geo g; g.kind = kind; g.n = n; g.h = h; g.w = w; g.xc = xc; g.cin_g = cin_g; g.cout = cout; g.k = k; g.stride = stride; g.pad = (k - 1) / 2;
  if (kind == 1) { g.ho = (h + 2 * g.pad - k) / stride + 1; g.wo = (w + 2 * g.pad - k) / stride + 1; } else { g.ho = h; g.wo = w; }
  return g;
}
/* kind 0 / 2: (n, h, w) = the OUTPUT map (x is already per output pixel); kind 1: (n, h, w) = the input map */
// FROST_G32_PLAIN=1 / frost_g32_set_plain(1): the round-4 one-thread-per-output kernels (A/B, debugging, the yardstick of tests/test_gpu_round5.py)
static int g_g32_plain = -1;
static int g32_fast() { if (g_g32_plain < 0) g_g32_plain = (getenv("FROST_G32_PLAIN") && atoi(getenv("FROST_G32_PLAIN"))) ? 1 : 0; return !g_g32_plain; }
extern "C" int frost_g32_set_plain(int on) { g_g32_plain = on ? 1 : 0; return 0; }
extern "C" int frost_g32_conv_acc(const int8_t* x, const float* qrec_x, const int8_t* qw, int kind, int n, int h, int w, int xc, int cin_g, int cout, int k,
                                  int stride, int32_t* acc, void* stream) {
  const G32Geo g = g32_geo(kind, n, h, w, xc, cin_g, cout, k, stride);
  hipStream_t s = as_stream(stream);
  const int64_t npo = (int64_t)g.n * g.ho * g.wo;
  if (g32_fast() && (cout % 4) == 0 && (xc % 4) == 0 && npo < (1ll << 31)) {
    if ((kind == 0 && (cin_g % 4) == 0) || (kind == 2 && xc <= 64 && cin_g <= 4)) {
      const unsigned gx = (unsigned)((npo + 255) / 256);
      if (cout <= 16) hipLaunchKernelGGL((k_g32_pw<1, 0>), dim3(gx, (unsigned)((cout + 15) / 16)), dim3(256), 0, s, x, qrec_x, qw, g, acc, G32Epi{});
      else if (cout <= 32) hipLaunchKernelGGL((k_g32_pw<2, 0>), dim3(gx, (unsigned)((cout + 31) / 32)), dim3(256), 0, s, x, qrec_x, qw, g, acc, G32Epi{});
      else hipLaunchKernelGGL((k_g32_pw<4, 0>), dim3(gx, (unsigned)((cout + 63) / 64)), dim3(256), 0, s, x, qrec_x, qw, g, acc, G32Epi{});
      return frost_check_launch("g32_conv_acc");
    }
    if (kind == 1 && (k == 3 || k == 5) && (stride == 1 || stride == 2)) {
      const int64_t bx = g32_run_blocks(cout, npo, 16);
#define G32_ACC(KK_, SS_) hipLaunchKernelGGL((k_g32_dw_acc<KK_, SS_>), g32_map_grid(cout, bx), dim3(256), 0, s, x, qrec_x, qw, g, acc)
      if (k == 3) { if (stride == 1) G32_ACC(3, 1); else G32_ACC(3, 2); } else { if (stride == 1) G32_ACC(5, 1); else G32_ACC(5, 2); }
#undef G32_ACC
      return frost_check_launch("g32_conv_acc");
    }
  }
  int64_t grid = (npo * cout + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_conv_acc, dim3((unsigned)grid), dim3(256), 0, s, x, qrec_x, qw, g, acc);
  return frost_check_launch("g32_conv_acc");
}

// reduce pass: S1[c] = sum gy, S2[c] = sum gy * xhat -- one workgroup per channel, fp64 partial sums, no atomics (deterministic)
__global__ __launch_bounds__(256) void k_g32_reduce(const int32_t* __restrict__ acc, int64_t npix, int cout, int cpad, float* coef, const float* qy, int relu,
                                                    const float* __restrict__ gout) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const G32Win w = g32_win(qy, relu);
  const float A = coef[FROST_COEF_A * cpad + c], B = coef[FROST_COEF_B * cpad + c], M = coef[FROST_COEF_M * cpad + c], R = coef[FROST_COEF_R * cpad + c];
  double s1 = 0.0, s2 = 0.0;
  for (int64_t p = tid; p < npix; p += 256) {
    const float af = (float)acc[p * cout + c];
    const float tq = fmaf(A, af, B) * w.y_inv;
    if (tq > w.t_lo && tq <= w.t_hi) { const double gy = (double)gout[p * cout + c]; s1 += gy; s2 += gy * (((double)af - (double)M) * (double)R); }
  }
  sh[0][tid] = s1; sh[1][tid] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; } __syncthreads(); }
  if (tid == 0) { coef[FROST_COEF_S1 * cpad + c] = (float)sh[0][0]; coef[FROST_COEF_S2 * cpad + c] = (float)sh[1][0]; }
}
/* scratch: fp64 partial sums of the two-stage form, >= frost_g32_scratch_bytes(); NULL selects the plain one-workgroup-per-channel kernel */
extern "C" int64_t frost_g32_scratch_bytes(void) { return (int64_t)160 << 20; }
extern "C" int frost_g32_reduce(const int32_t* acc, int64_t npix, int cout, float* coef, const float* qrec_y, int relu, const float* gout, void* scratch, void* stream) {
  hipStream_t s = as_stream(stream);
  const int cpad = round_up(cout, 16);
  if (g32_fast() && scratch && (cout % 4) == 0 && npix < (1ll << 31)) {
    const int PL = g32_map_pl(cout);
    int64_t bx = (npix + (int64_t)PL * 64 - 1) / ((int64_t)PL * 64);                   // ~64 pixels per thread
    const int64_t cap = frost_g32_scratch_bytes() / ((int64_t)PL * 2 * cout * 8);
    if (bx > cap) bx = cap; if (bx > 2048) bx = 2048; if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_g32_reduce_part, g32_map_grid(cout, bx), dim3(256), 0, s, acc, (int)npix, cout, cpad, coef, qrec_y, relu, gout, (double*)scratch);
    hipLaunchKernelGGL(k_g32_reduce_fin, dim3((unsigned)cout), dim3(256), 0, s, (const double*)scratch, (int)(bx * PL), cout, cpad, coef);
    return frost_check_launch("g32_reduce");
  }
  hipLaunchKernelGGL(k_g32_reduce, dim3((unsigned)cout), dim3(256), 0, s, acc, npix, cout, cpad, coef, qrec_y, relu, gout);
  return frost_check_launch("g32_reduce");
}

// dc = K1 * (gy - S1/n - xhat * S2/n), fp32 (S1 = S2 = 0 in the rows: the frozen-BatchNorm form dc = K1 * gy)
__global__ __launch_bounds__(256) void k_g32_dc(const int32_t* __restrict__ acc, int64_t npix, int cout, int cpad, const float* __restrict__ coef, const float* qy,
                                                int relu, const float* __restrict__ gout, float* __restrict__ dc) {
  const G32Win w = g32_win(qy, relu);
  const double inv_n = 1.0 / (double)npix;
  const int64_t tot = npix * cout;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cout);
    const float A = coef[FROST_COEF_A * cpad + c], B = coef[FROST_COEF_B * cpad + c], M = coef[FROST_COEF_M * cpad + c], R = coef[FROST_COEF_R * cpad + c];
    const double K1 = coef[FROST_COEF_K1 * cpad + c], S1 = coef[FROST_COEF_S1 * cpad + c], S2 = coef[FROST_COEF_S2 * cpad + c];
    const float af = (float)acc[i];
    const float tq = fmaf(A, af, B) * w.y_inv;
    const double gy = (tq > w.t_lo && tq <= w.t_hi) ? (double)gout[i] : 0.0;
    const double xhat = ((double)af - (double)M) * (double)R;
    dc[i] = (float)(K1 * (gy - S1 * inv_n - xhat * S2 * inv_n));
  }
}
extern "C" int frost_g32_dc(const int32_t* acc, int64_t npix, int cout, const float* coef, const float* qrec_y, int relu, const float* gout, float* dc,
                            void* stream) {
  hipStream_t s = as_stream(stream);
  if (g32_fast() && (cout % 4) == 0 && npix < (1ll << 31)) {
    const int PL = g32_map_pl(cout);
    int64_t bx = (npix + (int64_t)PL * 16 - 1) / ((int64_t)PL * 16); if (bx > 16384) bx = 16384;
    hipLaunchKernelGGL(k_g32_dc4, g32_map_grid(cout, bx), dim3(256), 0, s, acc, (int)npix, cout, round_up(cout, 16), coef, qrec_y, relu, gout, dc);
    return frost_check_launch("g32_dc");
  }
  int64_t grid = (npix * cout + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_dc, dim3((unsigned)grid), dim3(256), 0, s, acc, npix, cout, round_up(cout, 16), coef, qrec_y, relu, gout, dc);
  return frost_check_launch("g32_dc");
}

// ---- reduce / dc passes of a pointwise (kind 0) or stem (kind 2) layer that RECOMPUTE the integer conv output on the int8 MFMA instead of reading a stored one
// (frost_g32_conv_acc + frost_g32_reduce + frost_g32_dc without the int32 tensor): frost_g32_x_ok says whether the shape has the form
extern "C" int frost_g32_x_ok(int kind, int64_t npix, int xc, int cin_g, int cout) {
  return (g32_fast() && (cout % 4) == 0 && (xc % 4) == 0 && npix < (1ll << 31) && ((kind == 0 && (cin_g % 4) == 0) || (kind == 2 && xc <= 64 && cin_g <= 4))) ? 1 : 0;
}
template <int EPI>
static int g32_launch_x(const int8_t* x, const float* qrec_x, const int8_t* qw, const G32Geo& g, G32Epi e, unsigned gx, hipStream_t s) {
  const int cout = g.cout;
  if (cout <= 16) hipLaunchKernelGGL((k_g32_pw<1, EPI>), dim3(gx, (unsigned)((cout + 15) / 16)), dim3(256), 0, s, x, qrec_x, qw, g, (int32_t*)nullptr, e);
  else if (cout <= 32) hipLaunchKernelGGL((k_g32_pw<2, EPI>), dim3(gx, (unsigned)((cout + 31) / 32)), dim3(256), 0, s, x, qrec_x, qw, g, (int32_t*)nullptr, e);
  else hipLaunchKernelGGL((k_g32_pw<4, EPI>), dim3(gx, (unsigned)((cout + 63) / 64)), dim3(256), 0, s, x, qrec_x, qw, g, (int32_t*)nullptr, e);
  return 0;
}
extern "C" int frost_g32_reduce_x(const int8_t* x, const float* qrec_x, const int8_t* qw, int kind, int n, int h, int w, int xc, int cin_g, int cout, float* coef,
                                  const float* qrec_y, int relu, const float* gout, void* scratch, void* stream) {
  const G32Geo g = g32_geo(kind, n, h, w, xc, cin_g, cout, 1, 1);
  const int64_t npix = (int64_t)n * h * w;
  FROST_REQUIRE(frost_g32_x_ok(kind, npix, xc, cin_g, cout) && scratch, "g32_reduce_x: no recomputing form for this shape (frost_g32_x_ok), or no scratch");
  hipStream_t s = as_stream(stream);
  int64_t gx = (npix + 255) / 256;                                  // one wave per 64-pixel block at most; <= 512 workgroups (2048 partial rows) walk the rest
  const int64_t cap = frost_g32_scratch_bytes() / ((int64_t)4 * 2 * cout * 8);
  if (gx > 512) gx = 512; if (gx > cap) gx = cap; if (gx < 1) gx = 1;
  G32Epi e = {coef, round_up(cout, 16), qrec_y, relu, gout, (double*)scratch, nullptr};
  g32_launch_x<1>(x, qrec_x, qw, g, e, (unsigned)gx, s);
  hipLaunchKernelGGL(k_g32_reduce_fin, dim3((unsigned)cout), dim3(256), 0, s, (const double*)scratch, (int)(gx * 4), cout, round_up(cout, 16), coef);
  return frost_check_launch("g32_reduce_x");
}
extern "C" int frost_g32_dc_x(const int8_t* x, const float* qrec_x, const int8_t* qw, int kind, int n, int h, int w, int xc, int cin_g, int cout, const float* coef,
                              const float* qrec_y, int relu, const float* gout, float* dc, void* stream) {
  const G32Geo g = g32_geo(kind, n, h, w, xc, cin_g, cout, 1, 1);
  const int64_t npix = (int64_t)n * h * w;
  FROST_REQUIRE(frost_g32_x_ok(kind, npix, xc, cin_g, cout), "g32_dc_x: no recomputing form for this shape (frost_g32_x_ok)");
  G32Epi e = {coef, round_up(cout, 16), qrec_y, relu, gout, nullptr, dc};
  g32_launch_x<2>(x, qrec_x, qw, g, e, (unsigned)((npix + 255) / 256), as_stream(stream));
  return frost_check_launch("g32_dc_x");
}

// data gradient: gx[pixel][ci] (+)= sum_co dc[.][co] * q_w * s_w[co]     (kind 0: pointwise, 1: depthwise; the stem needs none)
__global__ __launch_bounds__(256) void k_g32_dgrad(const float* __restrict__ dc, const int8_t* __restrict__ qw, const float* qrec_w, const float* wscale, G32Geo g,
                                                   float* __restrict__ gx, int accumulate) {
  const int64_t npi = (int64_t)g.n * g.h * g.w, tot = npi * g.xc;
  const float sw0 = qrec_w[FROST_Q_SCALE];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int ci = (int)(i % g.xc); const int64_t p = i / g.xc;
    double s = 0.0;
    if (g.kind == 0) {
      if (ci < g.cin_g) for (int co = 0; co < g.cout; ++co) s += (double)dc[p * g.cout + co] * (double)((float)qw[(int64_t)co * g.cin_g + ci] * (wscale ? wscale[co] : sw0));
    } else {
      const int ix = (int)(p % g.w); const int iy = (int)((p / g.w) % g.h); const int in = (int)(p / ((int64_t)g.w * g.h));
      const float sw = wscale ? wscale[ci] : sw0;
      for (int ky = 0; ky < g.k; ++ky) { const int ty = iy + g.pad - ky; if (ty < 0 || ty % g.stride) continue; const int oy = ty / g.stride; if (oy >= g.ho) continue;
        for (int kx = 0; kx < g.k; ++kx) { const int tx = ix + g.pad - kx; if (tx < 0 || tx % g.stride) continue; const int ox = tx / g.stride; if (ox >= g.wo) continue;
          s += (double)dc[(((int64_t)in * g.ho + oy) * g.wo + ox) * g.cout + ci] * (double)((float)qw[(int64_t)ci * g.k * g.k + ky * g.k + kx] * sw); } }
    }
    gx[i] = accumulate ? gx[i] + (float)s : (float)s;
  }
}
extern "C" int frost_g32_dgrad(const float* dc, const int8_t* qw, const float* qrec_w, const float* wscale, int kind, int n, int h, int w, int xc, int cin_g,
                               int cout, int k, int stride, float* gx, int accumulate, void* stream) {
  FROST_REQUIRE(kind == 0 || kind == 1, "g32_dgrad: pointwise or depthwise");
  G32Geo g = g32_geo(kind, n, h, w, xc, cin_g, cout, k, stride);
  hipStream_t s = as_stream(stream);
  const int64_t npi = (int64_t)n * h * w;
  if (g32_fast() && (cout % 4) == 0 && (xc % 4) == 0 && npi < (1ll << 31)) {
    static const int b3 = getenv("FROST_G32_B3") ? atoi(getenv("FROST_G32_B3")) : 3;          // bit 0: data gradient on the bf16 MFMA with three-way split fp32 operands (0: v_mfma_f32_16x16x4_f32)
    if (kind == 0 && (b3 & 1) && (cout % 8) == 0 && cout >= 8) {
      const unsigned bx = (unsigned)((npi + 255) / 256);
      if (xc <= 16) hipLaunchKernelGGL((k_g32_pw_dgrad_b3<1>), dim3(bx, (unsigned)((xc + 15) / 16)), dim3(256), 0, s, dc, qw, qrec_w, wscale, g, gx, accumulate);
      else if (xc <= 32) hipLaunchKernelGGL((k_g32_pw_dgrad_b3<2>), dim3(bx, (unsigned)((xc + 31) / 32)), dim3(256), 0, s, dc, qw, qrec_w, wscale, g, gx, accumulate);
      else hipLaunchKernelGGL((k_g32_pw_dgrad_b3<4>), dim3(bx, (unsigned)((xc + 63) / 64)), dim3(256), 0, s, dc, qw, qrec_w, wscale, g, gx, accumulate);
      return frost_check_launch("g32_dgrad");
    }
    if (kind == 0) {
      const unsigned bx = (unsigned)((npi + 255) / 256);
      if (xc <= 16) hipLaunchKernelGGL((k_g32_pw_dgrad<1>), dim3(bx, (unsigned)((xc + 15) / 16)), dim3(256), 0, s, dc, qw, qrec_w, wscale, g, gx, accumulate);
      else if (xc <= 32) hipLaunchKernelGGL((k_g32_pw_dgrad<2>), dim3(bx, (unsigned)((xc + 31) / 32)), dim3(256), 0, s, dc, qw, qrec_w, wscale, g, gx, accumulate);
      else hipLaunchKernelGGL((k_g32_pw_dgrad<4>), dim3(bx, (unsigned)((xc + 63) / 64)), dim3(256), 0, s, dc, qw, qrec_w, wscale, g, gx, accumulate);
      return frost_check_launch("g32_dgrad");
    }
    if (xc == cout && (k == 3 || k == 5) && (stride == 1 || stride == 2)) {
      const int64_t bx = g32_run_blocks(xc, npi, 16);
#define G32_DG(KK_, SS_) hipLaunchKernelGGL((k_g32_dw_dgrad<KK_, SS_>), g32_map_grid(xc, bx), dim3(256), 0, s, dc, qw, qrec_w, wscale, g, gx, accumulate)
      if (k == 3) { if (stride == 1) G32_DG(3, 1); else G32_DG(3, 2); } else { if (stride == 1) G32_DG(5, 1); else G32_DG(5, 2); }
#undef G32_DG
      return frost_check_launch("g32_dgrad");
    }
  }
  int64_t grid = (npi * xc + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_dgrad, dim3((unsigned)grid), dim3(256), 0, s, dc, qw, qrec_w, wscale, g, gx, accumulate);
  return frost_check_launch("g32_dgrad");
}

// weight gradient w.r.t. the fake-quantised scaled weight: dwq[co][j] = s_x * sum_pixels dc[.][co] * (q_x - zp_x); one workgroup per weight element, fp64 sums
__global__ __launch_bounds__(256) void k_g32_wgrad(const float* __restrict__ dc, const int8_t* __restrict__ x, const float* qx, G32Geo g, float* __restrict__ dwq) {
  __shared__ double sh[256];
  const int per = (g.kind == 1) ? g.k * g.k : (g.kind == 2 ? g.cin_g * 9 : g.cin_g);
  const int co = blockIdx.x / per, jj = blockIdx.x % per, tid = threadIdx.x;
  const int zp = __float_as_int(qx[FROST_Q_ZP]);
  const int64_t npo = (int64_t)g.n * g.ho * g.wo;
  double s = 0.0;
  if (g.kind != 1) {
    const int col = (g.kind == 2) ? ((jj % 9) * 4 + jj / 9) : jj;            // stem: OIHW index c*9 + tap -> im2col column tap*4 + c
    for (int64_t p = tid; p < npo; p += 256) s += (double)dc[p * g.cout + co] * (double)(g32_x(x, p * g.xc + col) - zp);
  } else {
    const int ky = jj / g.k, kx = jj % g.k;
    for (int64_t p = tid; p < npo; p += 256) {
      const int ox = (int)(p % g.wo); const int oy = (int)((p / g.wo) % g.ho); const int in = (int)(p / ((int64_t)g.wo * g.ho));
      const int iy = oy * g.stride - g.pad + ky, ix = ox * g.stride - g.pad + kx;
      if (iy < 0 || iy >= g.h || ix < 0 || ix >= g.w) continue;
      s += (double)dc[p * g.cout + co] * (double)(g32_x(x, (((int64_t)in * g.h + iy) * g.w + ix) * g.xc + co) - zp);
    }
  }
  sh[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) sh[tid] += sh[tid + o]; __syncthreads(); }
  if (tid == 0) dwq[(int64_t)co * per + jj] = (float)(sh[0] * (double)qx[FROST_Q_SCALE]);
}
/* scratch: partial tiles of the two-stage form, >= frost_g32_scratch_bytes(); NULL selects the plain one-workgroup-per-weight kernel */
extern "C" int frost_g32_wgrad(const float* dc, const int8_t* x, const float* qrec_x, int kind, int n, int h, int w, int xc, int cin_g, int cout, int k, int stride,
                               float* dwq, void* scratch, void* stream) {
  G32Geo g = g32_geo(kind, n, h, w, xc, cin_g, cout, k, stride);
  hipStream_t s = as_stream(stream);
  const int per = (kind == 1) ? k * k : (kind == 2 ? cin_g * 9 : cin_g);
  const int64_t npo = (int64_t)g.n * g.ho * g.wo;
  if (g32_fast() && scratch && (cout % 4) == 0 && (xc % 4) == 0 && npo < (1ll << 31)) {
    if ((kind == 0 && (cin_g % 4) == 0) || (kind == 2 && cin_g <= 4)) {
      const int ncol = (kind == 2) ? xc : cin_g, ncp = ncol;
      int64_t chunk_px = 1024;
      const int64_t cap = frost_g32_scratch_bytes() / ((int64_t)cout * ncp * 4);            // chunks the scratch holds
      while ((npo + chunk_px - 1) / chunk_px > cap || (npo + chunk_px - 1) / chunk_px > 4096) chunk_px *= 2;
      const int64_t nchunk = (npo + chunk_px - 1) / chunk_px;
      const int na = cout <= 16 ? 1 : (cout <= 32 ? 2 : 4), nb = ncol <= 16 ? 1 : (ncol <= 32 ? 2 : 4);
      const dim3 grid((unsigned)((nchunk + 3) / 4), (unsigned)((cout + 16 * na - 1) / (16 * na)), (unsigned)((ncol + 16 * nb - 1) / (16 * nb)));
      static const int b3 = getenv("FROST_G32_B3") ? atoi(getenv("FROST_G32_B3")) : 3;          // bit 1: weight gradient on the bf16 MFMA with three-way split dc (0: v_mfma_f32_16x16x4_f32)
#define G32_WG(A_, B_) do { if (b3 & 2) hipLaunchKernelGGL((k_g32_pw_wgrad_part_b3<A_, B_>), grid, dim3(256), 0, s, dc, x, qrec_x, g, ncol, ncp, (int)chunk_px, (float*)scratch); \
                           else hipLaunchKernelGGL((k_g32_pw_wgrad_part<A_, B_>), grid, dim3(256), 0, s, dc, x, qrec_x, g, ncol, ncp, (int)chunk_px, (float*)scratch); } while (0)
      if (na == 1) { if (nb == 1) G32_WG(1, 1); else if (nb == 2) G32_WG(1, 2); else G32_WG(1, 4); }
      else if (na == 2) { if (nb == 1) G32_WG(2, 1); else if (nb == 2) G32_WG(2, 2); else G32_WG(2, 4); }
      else { if (nb == 1) G32_WG(4, 1); else if (nb == 2) G32_WG(4, 2); else G32_WG(4, 4); }
#undef G32_WG
      g32_sum_part((const float*)scratch, (int)nchunk, cout, per, ncp, kind, qrec_x, dwq, s);
      return frost_check_launch("g32_wgrad");
    }
    if (kind == 1 && (k == 3 || k == 5) && (stride == 1 || stride == 2)) {
      const int PL = g32_map_pl(cout);
      int64_t bx = g32_run_blocks(cout, npo, 64);                                           // runs of ~64 pixels per thread (fp32 sums per run, fp64 across the runs): runs of 256 left 392 workgroups for a 14 x 14 layer
      const int64_t cap = frost_g32_scratch_bytes() / ((int64_t)PL * cout * per * 4);
      if (bx > cap) bx = cap;
      if (bx < 1) bx = 1;
#define G32_DWG(KK_, SS_) hipLaunchKernelGGL((k_g32_dw_wgrad_part<KK_, SS_>), g32_map_grid(cout, bx), dim3(256), 0, s, dc, x, qrec_x, g, (float*)scratch)
      if (k == 3) { if (stride == 1) G32_DWG(3, 1); else G32_DWG(3, 2); } else { if (stride == 1) G32_DWG(5, 1); else G32_DWG(5, 2); }
#undef G32_DWG
      g32_sum_part((const float*)scratch, (int)(bx * PL), cout, per, per, 1, qrec_x, dwq, s);
      return frost_check_launch("g32_wgrad");
    }
  }
  hipLaunchKernelGGL(k_g32_wgrad, dim3((unsigned)(cout * per)), dim3(256), 0, s, dc, x, qrec_x, g, dwq);
  return frost_check_launch("g32_wgrad");
}

// ---- block wiring: FloatFunctional.cat / .add backward (the STE masks of their FakeQuantize), fp32 gradients
__global__ __launch_bounds__(256) void k_g32_cat_bwd(const float* __restrict__ gy, const int8_t* __restrict__ a, const float* qa, int ca, const int8_t* __restrict__ b,
                                                     const float* qb, int cb, int64_t npix, const float* qy, float* __restrict__ ga, int acc_a, float* __restrict__ gb, int acc_b) {
  const QP A = load_qp(qa), B = load_qp(qb), Y = load_qp(qy);
  const int cy = ca + cb; const int64_t tot = npix * cy;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cy); const int64_t p = i / cy;
    bool inr; float g = gy[i];
    if (c < ca) {
      fq_index((float)(g32_x(a, p * ca + c) - A.zp) * A.scale, Y.inv, Y.zp, 0, Y.hi, &inr);
      if (!inr) g = 0.0f;
      float* d = ga + p * ca + c; *d = acc_a ? *d + g : g;
    } else {
      fq_index((float)(g32_x(b, p * cb + c - ca) - B.zp) * B.scale, Y.inv, Y.zp, 0, Y.hi, &inr);
      if (!inr) g = 0.0f;
      float* d = gb + p * cb + c - ca; *d = acc_b ? *d + g : g;
    }
  }
}
extern "C" int frost_g32_cat_bwd(const float* gy, const int8_t* a, const float* qrec_a, int ca, const int8_t* b, const float* qrec_b, int cb, int64_t npix,
                                 const float* qrec_y, float* ga, int acc_a, float* gb, int acc_b, void* stream) {
  int64_t grid = (npix * (ca + cb) + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_cat_bwd, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), gy, a, qrec_a, ca, b, qrec_b, cb, npix, qrec_y, ga, acc_a, gb, acc_b);
  return frost_check_launch("g32_cat_bwd");
}
__global__ __launch_bounds__(256) void k_g32_add_bwd(const float* __restrict__ gy, const int8_t* __restrict__ a, const float* qa, const int8_t* __restrict__ b, const float* qb,
                                                     int64_t n, const float* qy, float* __restrict__ ga, int acc_a, float* __restrict__ gb, int acc_b) {
  const QP A = load_qp(qa), B = load_qp(qb), Y = load_qp(qy);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = (float)(g32_x(a, i) - A.zp) * A.scale + (float)(g32_x(b, i) - B.zp) * B.scale;        // k_add_bwd's operand order
    bool inr; fq_index(v, Y.inv, Y.zp, 0, Y.hi, &inr);
    const float g = inr ? gy[i] : 0.0f;
    ga[i] = acc_a ? ga[i] + g : g;
    gb[i] = acc_b ? gb[i] + g : g;
  }
}
extern "C" int frost_g32_add_bwd(const float* gy, const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n, const float* qrec_y,
                                 float* ga, int acc_a, float* gb, int acc_b, void* stream) {
  int64_t grid = (n + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_add_bwd, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), gy, a, qrec_a, b, qrec_b, n, qrec_y, ga, acc_a, gb, acc_b);
  return frost_check_launch("g32_add_bwd");
}
// gx[n][hw][c] = dpool[n][c] * drop[n][c] / hw in fp32 (the head's frost_head_bwd writes the bf16 form; dpool is its scratch output)
__global__ __launch_bounds__(256) void k_g32_pool_bwd(const float* __restrict__ dpool, const float* __restrict__ drop, int n, int hw, int c, float* __restrict__ gx) {
  const int64_t tot = (int64_t)n * hw * c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c); const int in = (int)(i / ((int64_t)hw * c));
    float v = dpool[(int64_t)in * c + ch];
    if (drop) v *= drop[(int64_t)in * c + ch];
    gx[i] = v / (float)hw;
  }
}
extern "C" int frost_g32_pool_bwd(const float* dpool, const float* drop, int n, int hw, int c, float* gx, void* stream) {
  int64_t grid = ((int64_t)n * hw * c + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_pool_bwd, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), dpool, drop, n, hw, c, gx);
  return frost_check_launch("g32_pool_bwd");
}
