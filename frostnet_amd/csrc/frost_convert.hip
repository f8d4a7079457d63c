// Converted int8 inference: what `torch.quantization.convert(model.eval())` + the QNNPACK engine compute
// (Classification/evaluate.py:130-134), restated in oracle/frost_oracle.py::converted_forward and pinned bit-exactly to the reference by
// tests/golden/g9_convert_*.npz.  The convolutions reuse the int8-MFMA / LDS depthwise EMIT kernels (mode 2: integer bias add + fp32
// requantisation in the epilogue); this file holds the pieces that differ from the fake-quant eval graph:
//   * the per-layer requantisation coefficients (BN folded with the RUNNING statistics, bias quantised to int32 at scale s_x*s_w),
//   * QNNPACK's integer fixed-point add (FloatFunctional.add -> quantized::add),
//   * the quantised adaptive average pool (mean index rounded half-to-even, input qparams kept),
//   * the classifier as an exact int32 GEMV with requantisation.
#include "frost_common.h"
typedef int v4i_c __attribute__((ext_vector_type(4)));

// replaces: nniqat.ConvBn(ReLU)2d.to_float -> fuse_conv_bn_weights (bias) + nnq.Conv2d.from_float + aten qconv.cpp (QNNPACK path):
//   bias_q = nearbyint(b_fold / (s_x*s_w)),  requant scale = s_w*s_x/s_y  (all fp32, this op order).
// coef row A <- requantisation scale, row B <- bias_q (int32 bits).  gamma == NULL: plain conv with bias `beta` (the classifier).
__global__ __launch_bounds__(256) void k_finalize_converted(const float* qx, const float* qw, const float* gamma, const float* beta,
                                                            const float* rmean, const float* rvar, int cout, int cpad, float* coef, const float* qy) {
  const float sx = qx[FROST_Q_SCALE], sw = qw[FROST_Q_SCALE], sy = qy[FROST_Q_SCALE];
  const float rs = (sw * sx) / sy;
  const float bscale = sx * sw;
  for (int c = threadIdx.x; c < cpad; c += 256) {
    float A = 0.0f; int bq = 0;
    if (c < cout) {
      float b;
      if (gamma) { const float rstd = 1.0f / sqrtf(rvar[c] + FROST_BN_EPS); b = (0.0f - rmean[c]) * rstd * gamma[c] + beta[c]; }
      else b = beta ? beta[c] : 0.0f;
      bq = (int)rintf(b / bscale);
      A = rs;
    }
    coef[FROST_COEF_A * cpad + c] = A;
    coef[FROST_COEF_B * cpad + c] = __int_as_float(bq);
  }
}
extern "C" int frost_conv_finalize_converted(const float* qrec_x, const float* qrec_w, const float* gamma, const float* beta,
                                             const float* rmean, const float* rvar, int cout, float* coef, const float* qrec_y, void* stream) {
  const int cpad = round_up(cout, 16);
  hipLaunchKernelGGL(k_finalize_converted, dim3(1), dim3(256), 0, as_stream(stream), qrec_x, qrec_w, gamma, beta, rmean, rvar, cout, cpad, coef, qrec_y);
  return frost_check_launch("conv_finalize_converted");
}

// The same coefficients for a model prepared with the per-channel 'fbgemm' qconfig and converted on the FBGEMM engine (Classification/latency_check.py:221-226;
// aten qconv.cpp -> fbgemm::ReQuantizeOutput with a float bias): row A <- (s_x * s_w[c]) / s_y, row B <- b_fold[c] / (s_x * s_w[c]) as a FLOAT -- the emit
// passes in mode 3 compute q = cvtps2dq((float(acc) + B) * A) + zp (oracle.fbgemm_conv, pinned by tests/golden/g13_convert_fbgemm_*).
__global__ __launch_bounds__(256) void k_finalize_converted_fb(const float* qx, const float* wscale, const float* gamma, const float* beta, const float* rmean,
                                                               const float* rvar, int cout, int cpad, float* coef, const float* qy) {
  const float sx = qx[FROST_Q_SCALE], sy = qy[FROST_Q_SCALE];
  for (int c = threadIdx.x; c < cpad; c += 256) {
    float A = 0.0f, B = 0.0f;
    if (c < cout) {
      float b;
      if (gamma) { const float rstd = 1.0f / sqrtf(rvar[c] + FROST_BN_EPS); b = (0.0f - rmean[c]) * rstd * gamma[c] + beta[c]; }
      else b = beta ? beta[c] : 0.0f;
      const float bs = sx * wscale[c];
      A = bs / sy; B = b / bs;
    }
    coef[FROST_COEF_A * cpad + c] = A;
    coef[FROST_COEF_B * cpad + c] = B;
  }
}
extern "C" int frost_conv_finalize_converted_fb(const float* qrec_x, const float* wscale, const float* gamma, const float* beta, const float* rmean,
                                                const float* rvar, int cout, float* coef, const float* qrec_y, void* stream) {
  FROST_REQUIRE(wscale != nullptr, "conv_finalize_converted_fb: per-channel weight scales required");
  const int cpad = round_up(cout, 16);
  hipLaunchKernelGGL(k_finalize_converted_fb, dim3(1), dim3(256), 0, as_stream(stream), qrec_x, wscale, gamma, beta, rmean, rvar, cout, cpad, coef, qrec_y);
  return frost_check_launch("conv_finalize_converted_fb");
}

// replaces: quantized::add on the QNNPACK engine (pytorch_qnnp_create_add_nc_q8 + q8vadd micro-kernel): integer fixed point.
//   a_mul = lrint(s_a/s_y * 2^shift), shift = 21 - exponent(max(s_a/s_y, s_b/s_y));
//   acc = a*a_mul + b*b_mul - (a_mul*zp_a + b_mul*zp_b);  y = clamp((acc >> shift) + (rem > thr) + zp_y, 0, 255), rem = (acc & mask) - (acc < 0)
__global__ __launch_bounds__(256) void k_add_qnnpack(const int8_t* __restrict__ a, const float* qa, const int8_t* __restrict__ b, const float* qb,
                                                     int64_t n, const float* qy, int8_t* __restrict__ y) {
  const float sa = qa[FROST_Q_SCALE], sb = qb[FROST_Q_SCALE], sy = qy[FROST_Q_SCALE];
  const int zpa = __float_as_int(qa[FROST_Q_ZP]), zpb = __float_as_int(qb[FROST_Q_ZP]), zpy = __float_as_int(qy[FROST_Q_ZP]);
  const float aos = sa / sy, bos = sb / sy;
  const float mx = fmaxf(aos, bos);
  const int shift = 21 - (int)(((uint32_t)__float_as_int(mx) >> 23) - 127u);
  const float two = __int_as_float((127 + shift) << 23);
  const int amul = (int)rintf(aos * two), bmul = (int)rintf(bos * two);
  const int zpp = -(amul * zpa + bmul * zpb);
  const int mask = (1 << shift) - 1, thr = mask >> 1;
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint32_t va = ((const uint32_t*)a)[i] ^ 0x80808080u, vb = ((const uint32_t*)b)[i] ^ 0x80808080u;   // offset-binary -> unsigned index
    uint32_t o = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int acc = zpp + (int)((va >> (8 * e)) & 255u) * amul + (int)((vb >> (8 * e)) & 255u) * bmul;
      const int rem = (acc & mask) - (acc < 0 ? 1 : 0);
      acc = (acc >> shift) + (rem > thr ? 1 : 0) + zpy;
      acc = min(max(acc, 0), 255);
      o |= (uint32_t)acc << (8 * e);
    }
    ((uint32_t*)y)[i] = o ^ 0x80808080u;
  }
}
extern "C" int frost_add_qnnpack(const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n, const float* qrec_y,
                                 int8_t* y, void* stream) {
  FROST_REQUIRE(n % 4 == 0, "add_qnnpack: n must be a multiple of 4");
  int64_t g = (n / 4 + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
  hipLaunchKernelGGL(k_add_qnnpack, dim3((unsigned)g), dim3(256), 0, as_stream(stream), a, qrec_a, b, qrec_b, n, qrec_y, y);
  return frost_check_launch("add_qnnpack");
}

// replaces: quantized adaptive_avg_pool2d(1): out index = rint_half_even(sum q / hw), qparams of the input.  pooled: int32 [n][c]
__global__ __launch_bounds__(256) void k_avgpool_q(const int8_t* __restrict__ x, int n, int hw, int c, int32_t* __restrict__ pooled) {
  const int64_t tot = (int64_t)n * c;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c); const int64_t in = i / c;
    const int8_t* p = x + in * hw * c + ch;
    int s = 0;
    for (int t = 0; t < hw; ++t) s += (int)p[(int64_t)t * c] + 128;
    int q = s / hw; const int r2 = 2 * (s - q * hw);             // exact round-half-even of s / hw
    if (r2 > hw || (r2 == hw && (q & 1))) ++q;
    pooled[i] = q;
  }
}
extern "C" int frost_avgpool_q(const int8_t* x, int n, int hw, int c, int32_t* pooled, void* stream) {
  int64_t g = ((int64_t)n * c + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_avgpool_q, dim3((unsigned)g), dim3(256), 0, as_stream(stream), x, n, hw, c, pooled);
  return frost_check_launch("avgpool_q");
}

// replaces: quantized::conv2d of the 1x1 classifier (nnq.Conv2d.from_float of nnqat.Conv2d, frostnet.py:298): one wave per output,
// exact int32 accumulation, integer bias, fp32 requantisation; writes the dequantised logits (DeQuantStub) and, optionally, the indices.
__global__ __launch_bounds__(256) void k_classifier_q(const int32_t* __restrict__ pooled, const float* qx, const int8_t* __restrict__ wq, const float* coef,
                                                      int cpad, int n, int c, int cout, const float* qy, float* __restrict__ logits, uint8_t* __restrict__ idx, int fb) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n * cout) return;
  const int img = wave / cout, co = wave - img * cout;
  const int zpx = __float_as_int(qx[FROST_Q_ZP]);
  int acc = 0;
  for (int k = lane; k < c; k += 64) acc += (pooled[(int64_t)img * c + k] - zpx) * (int)wq[(int64_t)co * c + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    const float rs = coef[FROST_COEF_A * cpad + co]; const int bq = __float_as_int(coef[FROST_COEF_B * cpad + co]);
    const int zpy = __float_as_int(qy[FROST_Q_ZP]);
    int q = fb ? (int)rintf(((float)acc + coef[FROST_COEF_B * cpad + co]) * rs) + zpy        // FBGEMM form: float bias row, per-channel multiplier
               : (int)rintf((float)(acc + bq) * rs) + zpy;
    q = min(max(q, 0), 255);
    logits[(int64_t)img * cout + co] = (float)(q - zpy) * qy[FROST_Q_SCALE];
    if (idx) idx[(int64_t)img * cout + co] = (uint8_t)q;
  }
}
// The same integer GEMM on the int8 MFMA: a wave owns 16 output channels x 16 images and walks K in steps of 64 (one v_mfma_i32_16x16x64_i8 each); both
// operands come straight from global memory in fragment layout (A = 16 bytes of a weight row, B = 16 pooled indices re-packed as offset-binary bytes), the
// zero-point term -(zp_x - 128) * sum_k w[co][k] from dot4 sums of the A bytes the wave loads anyway.  Integer accumulation is exact in any order, the
// epilogue is the one above: bit-identical logits.  (One wave per OUTPUT, as above, is 256 000 waves of 20 dependent byte loads: 134 us at B = 256.)
__global__ __launch_bounds__(256) void k_classifier_q_mfma(const int32_t* __restrict__ pooled, const float* qx, const int8_t* __restrict__ wq, const float* coef,
                                                           int cpad, int n, int c, int cout, const float* qy, float* __restrict__ logits, uint8_t* __restrict__ idx,
                                                           int fb, int tiles_co) {
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tco = wave % tiles_co, timg = wave / tiles_co;
  if (timg * 16 >= n) return;
  const int i = lane & 15, g = lane >> 4;
  const int8_t* wrow = wq + (int64_t)min(tco * 16 + i, cout - 1) * c + g * 16;
  const int32_t* prow = pooled + (int64_t)min(timg * 16 + i, n - 1) * c + g * 16;
  v4i_c acc = {0, 0, 0, 0};
  int ws = 0;
  for (int k0 = 0; k0 < c; k0 += 64) {
    const v4i_c a = *(const v4i_c*)(wrow + k0);
    v4i_c b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 v = *(const int4*)(prow + k0 + 4 * q);
      b[q] = (int)((uint32_t)((v.x - 128) & 255) | ((uint32_t)((v.y - 128) & 255) << 8) | ((uint32_t)((v.z - 128) & 255) << 16) | ((uint32_t)((v.w - 128) & 255) << 24));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) ws = __builtin_amdgcn_sdot4(a[q], 0x01010101, ws, false);
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
  }
  ws += __shfl_xor(ws, 16); ws += __shfl_xor(ws, 32);                        // every lane: sum_k w of channel tco * 16 + (lane & 15)
  const int zpx = __float_as_int(qx[FROST_Q_ZP]), zpy = __float_as_int(qy[FROST_Q_ZP]);
  const float sy = qy[FROST_Q_SCALE];
  const int img = timg * 16 + i;                                             // D: lane (j = image, g) holds channels 4 g + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = tco * 16 + 4 * g + r;
    const int wsr = __shfl(ws, 4 * g + r);
    if (co < cout && img < n) {
      const int a = acc[r] + (128 - zpx) * wsr;
      const float rs = coef[FROST_COEF_A * cpad + co];
      int q = fb ? (int)rintf(((float)a + coef[FROST_COEF_B * cpad + co]) * rs) + zpy : (int)rintf((float)(a + __float_as_int(coef[FROST_COEF_B * cpad + co])) * rs) + zpy;
      q = min(max(q, 0), 255);
      logits[(int64_t)img * cout + co] = (float)(q - zpy) * sy;
      if (idx) idx[(int64_t)img * cout + co] = (uint8_t)q;
    }
  }
}
static int classifier_q(const int32_t* pooled, const float* qrec_x, const int8_t* wq, const float* coef, int n, int c, int cout,
                        const float* qrec_y, float* logits, uint8_t* idx, int fb, void* stream) {
  static const int mfma_on = getenv("FROST_CLS_MFMA") ? atoi(getenv("FROST_CLS_MFMA")) : 1;
  if (mfma_on && (c & 63) == 0 && ((((uintptr_t)pooled) | ((uintptr_t)wq)) & 15) == 0) {
    const int tiles_co = (cout + 15) / 16, tiles_img = (n + 15) / 16;
    hipLaunchKernelGGL(k_classifier_q_mfma, dim3((unsigned)((tiles_co * tiles_img + 3) / 4)), dim3(256), 0, as_stream(stream), pooled, qrec_x, wq, coef,
                       round_up(cout, 16), n, c, cout, qrec_y, logits, idx, fb, tiles_co);
    return frost_check_launch("classifier_q");
  }
  const int64_t waves = (int64_t)n * cout;
  hipLaunchKernelGGL(k_classifier_q, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), pooled, qrec_x, wq, coef, round_up(cout, 16),
                     n, c, cout, qrec_y, logits, idx, fb);
  return frost_check_launch("classifier_q");
}
extern "C" int frost_classifier_q(const int32_t* pooled, const float* qrec_x, const int8_t* wq, const float* coef, int n, int c, int cout,
                                  const float* qrec_y, float* logits, uint8_t* idx, void* stream) {
  return classifier_q(pooled, qrec_x, wq, coef, n, c, cout, qrec_y, logits, idx, 0, stream);
}
extern "C" int frost_classifier_q_fb(const int32_t* pooled, const float* qrec_x, const int8_t* wq, const float* coef, int n, int c, int cout,
                                     const float* qrec_y, float* logits, uint8_t* idx, void* stream) {
  return classifier_q(pooled, qrec_x, wq, coef, n, c, cout, qrec_y, logits, idx, 1, stream);
}

// ------------------------------------------------------------------------------------------------ converted stem in one launch
// QuantStub + quantized::conv2d_relu of conv1 (frostnet.py:250, 319-320 after convert) straight from the fp32 image: today three launches (frost_quantize_input
// -> 4 bytes per input pixel, frost_stem_im2col -> 40 bytes per OUTPUT pixel, the int8 GEMM on those rows: 173 us at B = 256).  Here a workgroup quantises the
// input rows of a 4-row x 128-column output tile into LDS ([row][column] dwords c0 c1 c2 zp, offset-binary, padding = the zero point), a lane builds the
// K = 64 MFMA operand (k = tap * 4 + c, the pack's order) from four of those dwords, and the emit epilogue is k_pw's converted one.  The integer accumulators are
// exact and the epilogue expressions identical, so the output indices are those of the three-launch path bit for bit.
#define CS_TH 4
#define CS_XT 8
#define CS_LW (CS_XT * 32 + 2)
template <int CT>
__global__ __launch_bounds__(256) void k_stem_converted(const float* __restrict__ x, int h, int w, int ho, int wo, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                                        const float* qx, const int8_t* __restrict__ wpack, const int32_t* __restrict__ wsum,
                                                        const float* __restrict__ coef, int cpad, const float* qy, int cout, int cvt, int tiles_x, int tiles_y,
                                                        int8_t* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) uint32_t img[(2 * CS_TH + 1) * CS_LW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x; const int ty = b % tiles_y; const int in = b / tiles_y;
  const int oy0 = ty * CS_TH, ox0 = tx * CS_XT * 16;
  const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;
  const int ncol = min(CS_LW, (wo - ox0) * 2 + 1);
  const QP X = load_qp(qx);
  const uint32_t zb = (uint32_t)((X.zp - 128) & 255), zfill = zb * 0x01010101u;
  const float* src = x + (int64_t)in * sn;
  const int cpairs = (ncol + 1) >> 1;
  for (int u = tid; u < (2 * CS_TH + 1) * cpairs; u += 256) {
    const int r = u / cpairs, cp = u - r * cpairs;
    const int iy = iy0 + r;
    uint32_t o[2] = {zfill, zfill};
    if (iy >= 0 && iy < h) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ix = ix0 + 2 * cp + q;
        if (ix >= 0 && ix < w) {
          uint32_t packed = zb << 24;
#pragma unroll
          for (int c = 0; c < 3; ++c) packed |= ((uint32_t)((fq_index(src[c * sc + iy * sh + ix * sw], X.inv, X.zp, 0, X.hi) - 128) & 255)) << (8 * c);
          o[q] = packed;
        }
      }
    }
    *(uint2*)(img + r * CS_LW + 2 * cp) = make_uint2(o[0], o[1]);
  }
  // A fragments, weight sums, requantisation rows of this lane's 4 channels per channel tile
  v4i_c af[CT]; int4 ws[CT]; float A[CT][4], B[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    af[ct] = *(const v4i_c*)(wpack + (((int64_t)ct * 64 + lane) << 4));
    const int ch = ct * 16 + 4 * g;
    ws[ct] = *(const int4*)(wsum + ch);
    const float4 a4 = *(const float4*)(coef + FROST_COEF_A * cpad + ch), b4 = *(const float4*)(coef + FROST_COEF_B * cpad + ch);
    A[ct][0] = a4.x; A[ct][1] = a4.y; A[ct][2] = a4.z; A[ct][3] = a4.w; B[ct][0] = b4.x; B[ct][1] = b4.y; B[ct][2] = b4.z; B[ct][3] = b4.w;
  }
  // k = 16 g .. 16 g + 15 = taps 4 g .. 4 g + 3; taps past the ninth carry zero weights
  int off[4]; bool tv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int t = 4 * g + q; tv[q] = t < 9; off[q] = tv[q] ? (t / 3) * CS_LW + t % 3 : 0; }
  const int zpx = X.zp - 128;
  const float y_zpf = (float)__float_as_int(qy[FROST_Q_ZP]);
  const float y_inv = 1.0f / qy[FROST_Q_SCALE];
  const float qcap = (float)q_hi(qy); const bool lowq = qcap < 255.0f;
  __syncthreads();
  const int nxt = min(CS_XT, (wo - ox0 + 15) >> 4);
  int8_t* dst = y + (int64_t)in * ho * wo * cout;
  for (int t = wv; t < CS_TH * nxt; t += 4) {
    const int r = t / nxt, xt = t - r * nxt;
    const int oy = oy0 + r, ox = ox0 + xt * 16 + j;
    const uint32_t* bp = img + (2 * r) * CS_LW + (xt * 16 + j) * 2;
    v4i_c bf;
#pragma unroll
    for (int q = 0; q < 4; ++q) bf[q] = tv[q] ? (int)bp[off[q]] : 0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      v4i_c acc = {-zpx * ws[ct].x, -zpx * ws[ct].y, -zpx * ws[ct].z, -zpx * ws[ct].w};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[ct], bf, acc, 0, 0, 0);
      uint32_t packed = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float qv;
        if (cvt == 2) qv = rintf(((float)acc[e] + B[ct][e]) * A[ct][e]) + y_zpf;
        else {
          const float yv = (cvt == 1) ? fmaf(A[ct][e], (float)(acc[e] + __float_as_int(B[ct][e])), 0.0f) : fmaf(A[ct][e], (float)acc[e], B[ct][e]);
          qv = rintf(yv * (cvt == 1 ? 1.0f : y_inv)) + y_zpf;
        }
        if (lowq) qv = fminf(qv, qcap);
        packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, e, packed);
      }
      const int ch = ct * 16 + 4 * g;
      if (oy < ho && ox < wo && ch < cout) *(uint32_t*)(dst + ((int64_t)oy * wo + ox) * cout + ch) = packed ^ 0x80808080u;
    }
  }
}
extern "C" int frost_stem_converted_ok(int cout) { return (cout % 4 == 0 && cout >= 4 && cout <= 64) ? 1 : 0; }
/* x: logical (N,3,H,W) fp32 with element strides; qrec_x: the QuantStub's (frozen) record; wq_pack / wsum / coef: the stem layer's pack (kind 2), weight sums
 * and coefficient rows after frost_conv_finalize_converted(_fb); mode 2 / 3 = QNNPACK / FBGEMM requantisation (as frost_pw_conv_fwd), 1 = fake-quant emit;
 * y: [N][ho][wo][cout] offset-binary indices under qrec_y. */
extern "C" int frost_stem_converted(const float* x, int n, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* qrec_x,
                                    const int8_t* wq_pack, const int32_t* wsum, const float* coef, const float* qrec_y, int cout, int mode, int8_t* y,
                                    void* stream) {
  FROST_REQUIRE(frost_stem_converted_ok(cout), "stem_converted: cout must be a multiple of 4 in 4..64");
  FROST_REQUIRE(mode >= 1 && mode <= 3, "stem_converted: mode 1 (fake-quant emit), 2 / 3 (converted QNNPACK / FBGEMM form)");
  FROST_REQUIRE(x && qrec_x && wq_pack && wsum && coef && qrec_y && y && n > 0, "stem_converted: incomplete arguments");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int tiles_x = (wo + CS_XT * 16 - 1) / (CS_XT * 16), tiles_y = (ho + CS_TH - 1) / CS_TH;
  const dim3 grid((unsigned)((int64_t)n * tiles_x * tiles_y));
  hipStream_t s = as_stream(stream);
  const int ct = (cout + 15) / 16, cpad = round_up(cout, 16);
#define CS_GO(C) hipLaunchKernelGGL((k_stem_converted<C>), grid, dim3(256), 0, s, x, h, w, ho, wo, sn, sc, sh, sw, qrec_x, wq_pack, wsum, coef, cpad, qrec_y, cout, \
                                    mode - 1, tiles_x, tiles_y, y)
  if (ct == 1) CS_GO(1); else if (ct == 2) CS_GO(2); else if (ct == 3) CS_GO(3); else CS_GO(4);
#undef CS_GO
  return frost_check_launch("stem_converted");
}

// ------------------------------------------------------------------------------------------------ quantizable h-swish
// replaces: the reference's quantizable hard-swish (Classification/models/imagenet/mobilenetv3.py:43-56, `_Hswish`):
//   t   = relu6(add_scalar(x, 3))                            -> observed: prepare_qat hangs a FakeQuantize on nn.ReLU6
//   out = FloatFunctional.mul(x, t)                           -> observed: activation_post_process of quant_mul1
//   y   = mul_scalar(out, 1/6)                                -> unobserved: same indices, scale * (1/6)
// x is a fake-quantised activation, i.e. at most 256 distinct values: both observers' min/max are taken over the indices that
// are PRESENT, the forward is a 256-entry byte table, the backward a 256-entry float table (STE mask x f'(x) x 1/6).
__global__ __launch_bounds__(256) void k_hsw_presence(const int8_t* __restrict__ x, int64_t n, uint32_t* __restrict__ present) {
  __shared__ uint32_t bits[8];
  if (threadIdx.x < 8) bits[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint32_t v = ((const uint32_t*)x)[i] ^ 0x80808080u;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const uint32_t b = (v >> (8 * e)) & 255u; loc[b >> 5] |= 1u << (b & 31u); }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) if (loc[k]) atomicOr(&bits[k], loc[k]);
  __syncthreads();
  if (threadIdx.x < 8 && bits[threadIdx.x]) atomicOr(&present[threadIdx.x], bits[threadIdx.x]);
}
// one block of 256 threads: thread i = input index i.  lut layout: [0,256) forward bytes (offset-binary), then 256 floats (backward factor)
__device__ __forceinline__ void hsw_block_minmax(float v, bool here, float* slo, float* shi, float& lo, float& hi) {
  const int i = threadIdx.x;
  lo = wave_min(here ? v : INFINITY); hi = wave_max(here ? v : -INFINITY);
  __syncthreads();
  if ((i & 63) == 0) { slo[i >> 6] = lo; shi[i >> 6] = hi; }
  __syncthreads();
  lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3])); hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
}
__global__ __launch_bounds__(256) void k_hsw_finalize(const float* qx, uint32_t* present, float* qr6, float* qsite, float* qout, int observe, uint8_t* lut) {
  const int i = threadIdx.x;
  __shared__ float slo[4], shi[4];
  const float sx = qx[FROST_Q_SCALE]; const int zpx = __float_as_int(qx[FROST_Q_ZP]);
  const float xv = (float)(i - zpx) * sx;
  const bool here = (present[i >> 5] >> (i & 31)) & 1u;
  // site 1: the FakeQuantize prepare_qat hangs on nn.ReLU6 (observed: min/max of relu6(x + 3) over the values present)
  const float t0 = xv + 3.0f;
  const float t = fminf(fmaxf(t0, 0.0f), 6.0f);
  float lo, hi;
  hsw_block_minmax(t, here, slo, shi, lo, hi);
  if (i == 0) observer_update_dev(qr6, lo, hi, 0, 0, observe);
  __syncthreads();
  bool in6; const float s6r = qr6[FROST_Q_SCALE]; const int zp6 = __float_as_int(qr6[FROST_Q_ZP]);
  const int q6 = fq_index(t, 1.0f / s6r, zp6, 0, q_hi(qr6), &in6);
  const float tq = (float)(q6 - zp6) * s6r;
  // site 2: quant_mul1.mul(x, .) with its FakeQuantize
  const float f = xv * tq;
  hsw_block_minmax(f, here, slo, shi, lo, hi);
  if (i == 0) {
    observer_update_dev(qsite, lo, hi, 0, 0, observe);
    for (int k = 0; k < FROST_Q_STRIDE; ++k) qout[k] = qsite[k];                 // mul_scalar(1/6): same indices, scale * (1/6)
    const float s6 = qsite[FROST_Q_SCALE] * (1.0f / 6.0f);
    qout[FROST_Q_SCALE] = s6; qout[FROST_Q_INV] = 1.0f / s6;
    qout[FROST_Q_FQMIN] = qsite[FROST_Q_FQMIN] * (1.0f / 6.0f); qout[FROST_Q_FQMAX] = qsite[FROST_Q_FQMAX] * (1.0f / 6.0f);
  }
  __syncthreads();
  if (i < 8) present[i] = 0u;                                                      // re-armed for the next call
  const float inv = 1.0f / qsite[FROST_Q_SCALE]; const int zp = __float_as_int(qsite[FROST_Q_ZP]);
  bool inr; const int q = fq_index(f, inv, zp, 0, q_hi(qsite), &inr);
  lut[i] = (uint8_t)((q - 128) & 255);
  // d/dx [x * FQ(relu6(x + 3))] = FQ(.) + x * [FQ in range] * [0 < x + 3 < 6]   (hardtanh backward: strict inequalities)
  const float dfdx = tq + ((in6 && t0 > 0.0f && t0 < 6.0f) ? xv : 0.0f);
  ((float*)(lut + 256))[i] = inr ? dfdx * (1.0f / 6.0f) : 0.0f;
}
__global__ __launch_bounds__(256) void k_hsw_apply(const int8_t* __restrict__ x, int64_t n, const uint8_t* __restrict__ lut, int8_t* __restrict__ y) {
  __shared__ uint8_t l[256];
  l[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint32_t v = ((const uint32_t*)x)[i] ^ 0x80808080u;
    ((uint32_t*)y)[i] = (uint32_t)l[v & 255u] | ((uint32_t)l[(v >> 8) & 255u] << 8) | ((uint32_t)l[(v >> 16) & 255u] << 16) | ((uint32_t)l[v >> 24] << 24);
  }
}
__global__ __launch_bounds__(256) void k_hsw_bwd(const uint16_t* __restrict__ gout, const int8_t* __restrict__ x, int64_t n, const uint8_t* __restrict__ lut,
                                                 uint16_t* __restrict__ dx, int accumulate) {
  __shared__ float l[256];
  l[threadIdx.x] = ((const float*)(lut + 256))[threadIdx.x];
  __syncthreads();
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = bf2f(gout[i]) * l[((int)x[i] + 128) & 255];
    if (accumulate) v += bf2f(dx[i]);
    dx[i] = f2bf(v);
  }
}
extern "C" int frost_hswish_fwd(const int8_t* x, const float* qrec_x, int64_t n, uint32_t* present8, float* qrec_relu6, float* qrec_site, float* qrec_out,
                                int observe, uint8_t* lut, int8_t* y, void* stream) {
  FROST_REQUIRE(n % 4 == 0, "hswish: n must be a multiple of 4");
  hipStream_t s = as_stream(stream);
  int64_t g = (n / 4 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
  hipLaunchKernelGGL(k_hsw_presence, dim3((unsigned)g), dim3(256), 0, s, x, n, present8);
  hipLaunchKernelGGL(k_hsw_finalize, dim3(1), dim3(256), 0, s, qrec_x, present8, qrec_relu6, qrec_site, qrec_out, observe, lut);
  hipLaunchKernelGGL(k_hsw_apply, dim3((unsigned)g), dim3(256), 0, s, x, n, lut, y);
  return frost_check_launch("hswish_fwd");
}
// The CONVERTED model's hard-swish (torch.quantization.convert of `_Hswish`: QFunctional.add_scalar -> nnq.ReLU6 -> QFunctional.mul -> QFunctional.mul_scalar on
// quint8 tensors), again a 256-entry table of the input index -- but integer arithmetic, not the fake-quant graph's:
//   add_scalar (ATen quantized BinaryOps _add_scalar_out): c_q = nearbyint(3 / s_x);  z_x - c_q in [0, 255]: same indices, zero point z_x - c_q;  below 0:
//     scale s' = (255 - (z_x - c_q)) / 255 * s_x, zero point 0, indices requantised: 0 + lrintf((q - z_x + c_q) * (float(s_x) * (1.0f / float(s'))))
//   relu6 (qrelu6): clamp(index, z', z' + nearbyint(6.0f * (1.0f / float(s'))))      -- no record of its own: nn.ReLU6's observer is NOT used after convert
//   mul (qmul): z_m + lrintf(((q - z_x) * (t - z')) * (float(s_x) * float(s') * (1.0f / float(s_m)))), saturated to [0, 255]; (s_m, z_m) = quant_mul1's record
//   mul_scalar(1/6): same indices, scale double(s_m) * (1/6)
// tests: oracle.frost_oracle.converted_hswish_table (pinned against stock torch on the CPU, tests/golden g14) and the whole converted network against stock torch.
__global__ __launch_bounds__(256) void k_hsw_cvt_lut(const float* qx, const float* qsite, float* qout, uint8_t* lut) {
  const int q = threadIdx.x;
  const float sxf = qx[FROST_Q_SCALE]; const int zx = __float_as_int(qx[FROST_Q_ZP]);
  const double sx = (double)sxf;
  const int c_q = (int)nearbyint(3.0 / sx);
  int z1 = zx - c_q, a = q; float s1f = sxf;
  if (z1 < 0) {
    const double s1 = (255.0 - (double)z1) / 255.0 * sx;
    s1f = (float)s1;
    const float mult = sxf * (1.0f / s1f);
    const int r = (int)lrintf((float)(q - zx + c_q) * mult);
    a = min(max(r, 0), 255); z1 = 0;
  }
  const int six = min(max(z1 + (int)nearbyintf(6.0f * (1.0f / s1f)), 0), 255);
  const int t = min(max(a, z1), six);
  const float smf = qsite[FROST_Q_SCALE]; const int zm = __float_as_int(qsite[FROST_Q_ZP]);
  const float mult2 = sxf * s1f * (1.0f / smf);
  const int c = zm + (int)lrintf((float)((q - zx) * (t - z1)) * mult2);
  lut[q] = (uint8_t)((min(max(c, 0), 255) - 128) & 255);
  if (q == 0) {
    for (int k = 0; k < FROST_Q_STRIDE; ++k) qout[k] = qsite[k];
    const float s6 = (float)((double)smf * (1.0 / 6.0));
    qout[FROST_Q_SCALE] = s6; qout[FROST_Q_INV] = 1.0f / s6;
    qout[FROST_Q_FQMIN] = qsite[FROST_Q_FQMIN] * (1.0f / 6.0f); qout[FROST_Q_FQMAX] = qsite[FROST_Q_FQMAX] * (1.0f / 6.0f);
  }
}
extern "C" int frost_hswish_converted(const int8_t* x, const float* qrec_x, int64_t n, const float* qrec_site, float* qrec_out, uint8_t* lut, int8_t* y, void* stream) {
  FROST_REQUIRE(n % 4 == 0, "hswish_converted: n must be a multiple of 4");
  hipStream_t s = as_stream(stream);
  int64_t g = (n / 4 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
  if (qrec_site) hipLaunchKernelGGL(k_hsw_cvt_lut, dim3(1), dim3(256), 0, s, qrec_x, qrec_site, qrec_out, lut);      // NULL: `lut` (and qrec_out) were built by an earlier call
  hipLaunchKernelGGL(k_hsw_apply, dim3((unsigned)g), dim3(256), 0, s, x, n, lut, y);
  return frost_check_launch("hswish_converted");
}
extern "C" int frost_hswish_bwd(const uint16_t* gout, const int8_t* x, int64_t n, const uint8_t* lut, uint16_t* dx, int accumulate, void* stream) {
  int64_t g = (n + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
  hipLaunchKernelGGL(k_hsw_bwd, dim3((unsigned)g), dim3(256), 0, as_stream(stream), gout, x, n, lut, dx, accumulate);
  return frost_check_launch("hswish_bwd");
}
