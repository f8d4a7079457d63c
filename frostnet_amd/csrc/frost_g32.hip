// fp32-GRADIENT parity mode of the fake-quant backward (reference: loss.backward() is fp32 autograd, Classification/utils/helper_functions.py:139-143).
//
// The production backward stores activation gradients and dc in bf16 (stochastically rounded) and feeds bf16 operands to the MFMA data / weight gradient
// GEMMs: measured 2-9e-3 from an fp64 evaluation per layer, 2-3e-2 on ill-conditioned sums.  This file is the SAME backward -- the same masks (STE window of the
// activation fake-quantise incl. ReLU, evaluated on the exact integer conv output with the forward's coefficient rows), the same BatchNorm expression
// dc = K1 (gy - S1/n - xhat S2/n), the same fake-quantised weights / inputs in the data / weight gradient -- with every gradient held in fp32 and every
// long sum accumulated in fp64, as plain one-thread-per-output kernels.  Not for speed (it is 10-30 x slower than the production kernels): it exists to show
// that the formulas meet the reference's fp32 autograd at <= 1e-3, and what the bf16 storage costs (tests/test_gpu_round4.py, DESIGN "fp32-gradient mode").
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
static G32Geo g32_geo(int kind, int n, int h, int w, int xc, int cin_g, int cout, int k, int stride) {
  G32Geo g; g.kind = kind; g.n = n; g.h = h; g.w = w; g.xc = xc; g.cin_g = cin_g; g.cout = cout; g.k = k; g.stride = stride; g.pad = (k - 1) / 2;
  if (kind == 1) { g.ho = (h + 2 * g.pad - k) / stride + 1; g.wo = (w + 2 * g.pad - k) / stride + 1; } else { g.ho = h; g.wo = w; }
  return g;
}
/* kind 0 / 2: (n, h, w) = the OUTPUT map (x is already per output pixel); kind 1: (n, h, w) = the input map */
extern "C" int frost_g32_conv_acc(const int8_t* x, const float* qrec_x, const int8_t* qw, int kind, int n, int h, int w, int xc, int cin_g, int cout, int k,
                                  int stride, int32_t* acc, void* stream) {
  const G32Geo g = g32_geo(kind, n, h, w, xc, cin_g, cout, k, stride);
  int64_t grid = ((int64_t)g.n * g.ho * g.wo * cout + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_conv_acc, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, qrec_x, qw, g, acc);
  return frost_check_launch("g32_conv_acc");
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
extern "C" int frost_g32_reduce(const int32_t* acc, int64_t npix, int cout, float* coef, const float* qrec_y, int relu, const float* gout, void* stream) {
  hipLaunchKernelGGL(k_g32_reduce, dim3((unsigned)cout), dim3(256), 0, as_stream(stream), acc, npix, cout, round_up(cout, 16), coef, qrec_y, relu, gout);
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
  int64_t grid = (npix * cout + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_dc, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), acc, npix, cout, round_up(cout, 16), coef, qrec_y, relu, gout, dc);
  return frost_check_launch("g32_dc");
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
  int64_t grid = ((int64_t)n * h * w * xc + 255) / 256; if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(k_g32_dgrad, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), dc, qw, qrec_w, wscale, g, gx, accumulate);
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
extern "C" int frost_g32_wgrad(const float* dc, const int8_t* x, const float* qrec_x, int kind, int n, int h, int w, int xc, int cin_g, int cout, int k, int stride,
                               float* dwq, void* stream) {
  G32Geo g = g32_geo(kind, n, h, w, xc, cin_g, cout, k, stride);
  const int per = (kind == 1) ? k * k : (kind == 2 ? cin_g * 9 : cin_g);
  hipLaunchKernelGGL(k_g32_wgrad, dim3((unsigned)(cout * per)), dim3(256), 0, as_stream(stream), dc, x, qrec_x, g, dwq);
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
