// Float (not fake-quantised) FrostNet TRAINING on the device: the StatAssist warm-up phase (Classification/train.py:149-165 runs the
// float model for FP_epoch epochs with the same GradBoost optimizer before prepare_qat) and eval-mode forward of the float model.
// replaces: frostnet.py:14-60 ConvBNReLU / ConvBN in train mode = Conv2d(bias=False) -> BatchNorm2d(batch statistics, running-stat
// update, momentum 0.1) -> ReLU, and their autograd backward.
// Storage: activations and activation gradients NHWC, either bf16 (the fast default) or fp32 (the reference's own precision: every
// kernel below is a template on the element type, the *_f32 entries instantiate it with float and the pointwise products run on the
// fp32 MFMA, so the FP32-train end-to-end gate of the reference applies); parameters / statistics / accumulators fp32 (sums in double
// across workgroups).  Same recompute structure as the fake-quant path: forward = statistics pass -> frost_float_bn_finalize -> emit pass;
// backward = reduce pass (S1 = sum g, S2 = sum g*xhat) -> frost_float_bwd_finalize -> dc pass -> dgrad, wgrad.  Every pass recomputes
// the convolution with identical arithmetic, so the ReLU mask z > 0 of the backward is exactly the forward's.
// Pointwise convs (and the im2col'd stem) run on the bf16 MFMA (16x16x32), depthwise convs on fp32 FMAs.
#include "frost_common.h"
#include <type_traits>

typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));
typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef int v2i32 __attribute__((ext_vector_type(2)));

enum { F_STATS = 0, F_EMIT = 1, F_BRED = 2, F_BDC = 3, F_PLAIN = 4 };      // F_PLAIN: out = conv (the data-gradient GEMM of the fp32 mode)
// The `relu` argument of every entry is an activation code: 0 = none, 1 = ReLU, 2 = hard-swish z * relu6(z + 3) / 6 (the reference's _Hswish,
// Classification/models/imagenet/mobilenetv3.py:43-56, in the order torch evaluates it: add_scalar, relu6, mul, mul_scalar(1/6)).  The backward passes
// recompute z from the kept conv output and scale the incoming gradient by the derivative: ReLU's mask, or for hard-swish
// (relu6(z + 3) + z * [0 < z + 3 < 6]) / 6 -- hardtanh's open-interval gradient, as autograd forms it.
enum { F_ACT_NONE = 0, F_ACT_RELU = 1, F_ACT_HSWISH = 2 };
__device__ __forceinline__ float f_act(float z, float lo, bool hs) {
  return hs ? hswish_f(z) : fmaxf(z, lo);
}
__device__ __forceinline__ float f_dhswish(float z) {
  const float t = z + 3.0f;
  return (fminf(fmaxf(t, 0.0f), 6.0f) + ((t > 0.0f && t < 6.0f) ? z : 0.0f)) * (1.0f / 6.0f);
}
// element access for the two storage types
template <typename T> struct FEl;
template <> struct FEl<uint16_t> {
  static __device__ __forceinline__ void ld8(const uint16_t* p, float* o) {
    const uint4 v = *(const uint4*)p;
    o[0] = bf2f(v.x & 0xffff); o[1] = bf2f(v.x >> 16); o[2] = bf2f(v.y & 0xffff); o[3] = bf2f(v.y >> 16);
    o[4] = bf2f(v.z & 0xffff); o[5] = bf2f(v.z >> 16); o[6] = bf2f(v.w & 0xffff); o[7] = bf2f(v.w >> 16);
  }
  static __device__ __forceinline__ void st8(uint16_t* p, const float* v) {
    uint4 o; o.x = cvt_pk_bf16(v[0], v[1]); o.y = cvt_pk_bf16(v[2], v[3]); o.z = cvt_pk_bf16(v[4], v[5]); o.w = cvt_pk_bf16(v[6], v[7]);
    *(uint4*)p = o;
  }
  static __device__ __forceinline__ void ld4(const uint16_t* p, float* o) {
    const uint2 v = *(const uint2*)p; o[0] = bf2f(v.x & 0xffff); o[1] = bf2f(v.x >> 16); o[2] = bf2f(v.y & 0xffff); o[3] = bf2f(v.y >> 16);
  }
  static __device__ __forceinline__ void st4(uint16_t* p, const float* v) { uint2 o; o.x = cvt_pk_bf16(v[0], v[1]); o.y = cvt_pk_bf16(v[2], v[3]); *(uint2*)p = o; }
  using R8 = uint4;
  static __device__ __forceinline__ R8 ldr8(const uint16_t* p) { return *(const uint4*)p; }
  static __device__ __forceinline__ R8 zero8() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ void cv8(R8 v, float* o) {
    o[0] = bf2f(v.x & 0xffff); o[1] = bf2f(v.x >> 16); o[2] = bf2f(v.y & 0xffff); o[3] = bf2f(v.y >> 16);
    o[4] = bf2f(v.z & 0xffff); o[5] = bf2f(v.z >> 16); o[6] = bf2f(v.w & 0xffff); o[7] = bf2f(v.w >> 16);
  }
  using R4 = uint2;           // four elements as loaded (conversion deferred: the load is issued long before its use)
  static __device__ __forceinline__ R4 ldr4(const uint16_t* p) { return *(const uint2*)p; }
  static __device__ __forceinline__ R4 zero4() { return make_uint2(0, 0); }
  static __device__ __forceinline__ void cv4(R4 v, float* o) { o[0] = bf2f(v.x & 0xffff); o[1] = bf2f(v.x >> 16); o[2] = bf2f(v.y & 0xffff); o[3] = bf2f(v.y >> 16); }
  static __device__ __forceinline__ float ld1(const uint16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st1(uint16_t* p, float v) { *p = f2bf(v); }
};
template <> struct FEl<float> {
  static __device__ __forceinline__ void ld8(const float* p, float* o) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
  static __device__ __forceinline__ void st8(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]); }
  static __device__ __forceinline__ void ld4(const float* p, float* o) { const float4 a = *(const float4*)p; o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; }
  static __device__ __forceinline__ void st4(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
  struct R8 { float4 a, b; };
  static __device__ __forceinline__ R8 ldr8(const float* p) { R8 r; r.a = *(const float4*)p; r.b = *(const float4*)(p + 4); return r; }
  static __device__ __forceinline__ R8 zero8() { R8 r; r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; return r; }
  static __device__ __forceinline__ void cv8(R8 v, float* o) { o[0] = v.a.x; o[1] = v.a.y; o[2] = v.a.z; o[3] = v.a.w; o[4] = v.b.x; o[5] = v.b.y; o[6] = v.b.z; o[7] = v.b.w; }
  using R4 = float4;
  static __device__ __forceinline__ R4 ldr4(const float* p) { return *(const float4*)p; }
  static __device__ __forceinline__ R4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ void cv4(R4 v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
// inclusive scan over a 16-lane row with row_shr 1,2,4,8: lane 15 of every row holds the row total (pure VALU, no LDS crossbar)
template <int CTRL> __device__ __forceinline__ float f_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float f_row_sum(float v) { v += f_dpp<0x111>(v); v += f_dpp<0x112>(v); v += f_dpp<0x114>(v); v += f_dpp<0x118>(v); return v; }
__device__ __forceinline__ double f_row_sum(double v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; }
#define FST_SLOTS 8       // statistic partials are spread over 8 copies of the table (workgroup index & 7): 8x fewer same-address atomics for the L2 to serialise; the finalize kernels add the copies up
enum { FC_SCALE = 0, FC_BIAS = 1, FC_MEAN = 2, FC_INV = 3, FC_K1 = 4, FC_E = 5, FC_F = 6, FC_VAR = 7 };

// ------------------------------------------------------------------------------------------------ weight preparation (per step)
// packs the CURRENT fp32 master weights (no BN folding: batch statistics are not known yet) and zeroes the statistics accumulators
__global__ __launch_bounds__(256) void k_f_prep(const FrostFDesc* descs) {
  const FrostFDesc d = descs[blockIdx.y];
  const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
  for (int i = gtid; i < FST_SLOTS * 4 * d.cpad; i += gsz) d.stat[i] = 0.0;
  if ((d.kind == 0 || d.kind == 2) && d.fp32) {       // fp32 mode: A-fragments [ct][kb][lane][4]: W[ct*16 + (lane&15)][kb*16 + (lane>>4)*4 + e]
    const int CT = d.cpad / 16, KB = d.kpad / 16;
    const int64_t nel = (int64_t)CT * KB * 256;
    float* pk = (float*)d.pack;
    for (int64_t i = gtid; i < nel; i += gsz) {
      const int e = (int)(i & 3), lane = (int)((i >> 2) & 63); const int64_t t = i >> 8; const int kb = (int)(t % KB), ct = (int)(t / KB);
      const int co = ct * 16 + (lane & 15), k = kb * 16 + (lane >> 4) * 4 + e;
      float v = 0.0f;
      if (co < d.cout) {
        if (d.kind == 0) { if (k < d.cin_g) v = d.w[(int64_t)co * d.cin_g + k]; }
        else { const int tap = k >> 2, c = k & 3; if (tap < d.kk && c < d.cin_g) v = d.w[((int64_t)co * d.cin_g + c) * d.kk + tap]; }
      }
      pk[i] = v;
    }
    if (d.pack_t) {
      const int CTt = round_up(d.cin_g, 16) / 16, KBt = d.kpad_t / 16;
      const int64_t nt = (int64_t)CTt * KBt * 256;
      float* pt = (float*)d.pack_t;
      for (int64_t i = gtid; i < nt; i += gsz) {
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63); const int64_t t = i >> 8; const int kb = (int)(t % KBt), ct = (int)(t / KBt);
        const int ci = ct * 16 + (lane & 15), co = kb * 16 + (lane >> 4) * 4 + e;
        pt[i] = (ci < d.cin_g && co < d.cout) ? d.w[(int64_t)co * d.cin_g + ci] : 0.0f;
      }
    }
  } else if (d.kind == 0 || d.kind == 2) {       // A-fragments [ct][kb][lane][8]: W[ct*16 + (lane&15)][kb*32 + (lane>>4)*8 + e]
    const int CT = d.cpad / 16, KB = d.kpad / 32;
    const int64_t nel = (int64_t)CT * KB * 512;
    uint16_t* pk = (uint16_t*)d.pack;
    for (int64_t i = gtid; i < nel; i += gsz) {
      const int e = (int)(i & 7), lane = (int)((i >> 3) & 63); const int64_t t = i >> 9; const int kb = (int)(t % KB), ct = (int)(t / KB);
      const int co = ct * 16 + (lane & 15), k = kb * 32 + (lane >> 4) * 8 + e;
      float v = 0.0f;
      if (co < d.cout) {
        if (d.kind == 0) { if (k < d.cin_g) v = d.w[(int64_t)co * d.cin_g + k]; }
        else { const int tap = k >> 2, c = k & 3; if (tap < d.kk && c < d.cin_g) v = d.w[((int64_t)co * d.cin_g + c) * d.kk + tap]; }   // stem: k = tap*4 + c
      }
      pk[i] = f2bf(v);
    }
    if (d.pack_t) {                       // dgrad operand: rows = input channel, K = output channel
      const int CTt = round_up(d.cin_g, 16) / 16, KBt = d.kpad_t / 32;
      const int64_t nt = (int64_t)CTt * KBt * 512;
      uint16_t* pt = (uint16_t*)d.pack_t;
      for (int64_t i = gtid; i < nt; i += gsz) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63); const int64_t t = i >> 9; const int kb = (int)(t % KBt), ct = (int)(t / KBt);
        const int ci = ct * 16 + (lane & 15), co = kb * 32 + (lane >> 4) * 8 + e;
        pt[i] = f2bf((ci < d.cin_g && co < d.cout) ? d.w[(int64_t)co * d.cin_g + ci] : 0.0f);
      }
    }
  } else {                                // depthwise: fp32 [tap][cpad]
    float* pk = (float*)d.pack;
    const int64_t nel = (int64_t)d.kk * d.cpad;
    for (int64_t i = gtid; i < nel; i += gsz) {
      const int c = (int)(i % d.cpad), tap = (int)(i / d.cpad);
      pk[i] = (c < d.cout) ? d.w[(int64_t)c * d.kk + tap] : 0.0f;
    }
  }
}
extern "C" int frost_float_weight_prep(const FrostFDesc* descs, int nlayers, void* stream) {
  if (nlayers <= 0) return 0;
  hipLaunchKernelGGL(k_f_prep, dim3(64, nlayers), dim3(256), 0, as_stream(stream), descs);
  return frost_check_launch("float_weight_prep");
}

// ------------------------------------------------------------------------------------------------ BatchNorm coefficient kernels
// train: batch mean / biased variance from the statistics pass -> scale = gamma*invsigma, bias = beta - mean*scale; running statistics
// updated with momentum 0.1 and the unbiased variance, num_batches_tracked += 1 (torch BatchNorm2d training semantics)
__global__ __launch_bounds__(256) void k_f_bn_finalize(const FrostFDesc* dp, double count) {
  const FrostFDesc d = *dp;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && d.nbt) *d.nbt += 1;
  if (c >= d.cout) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < FST_SLOTS; ++k) { s1 += d.stat[k * 4 * d.cpad + c]; s2 += d.stat[k * 4 * d.cpad + d.cpad + c]; }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean; if (var < 0.0) var = 0.0;
  const float inv = (float)(1.0 / sqrt(var + (double)FROST_BN_EPS));
  const float sc = d.gamma[c] * inv;
  d.coef[FC_SCALE * d.cpad + c] = sc; d.coef[FC_BIAS * d.cpad + c] = d.beta[c] - (float)mean * sc;
  d.coef[FC_MEAN * d.cpad + c] = (float)mean; d.coef[FC_INV * d.cpad + c] = inv; d.coef[FC_VAR * d.cpad + c] = (float)var;
  const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
  d.rmean[c] = (1.0f - FROST_BN_MOM) * d.rmean[c] + FROST_BN_MOM * (float)mean;
  d.rvar[c] = (1.0f - FROST_BN_MOM) * d.rvar[c] + FROST_BN_MOM * (float)unb;
}
extern "C" int frost_float_bn_finalize(const FrostFDesc* desc, int cout, int64_t count, void* stream) {
  hipLaunchKernelGGL(k_f_bn_finalize, dim3((cout + 255) / 256), dim3(256), 0, as_stream(stream), desc, (double)count);
  return frost_check_launch("float_bn_finalize");
}
// eval: running statistics, one launch for the whole table
__global__ __launch_bounds__(256) void k_f_bn_eval(const FrostFDesc* descs) {
  const FrostFDesc d = descs[blockIdx.y];
  for (int c = blockIdx.x * 256 + threadIdx.x; c < d.cout; c += gridDim.x * 256) {
    const float inv = 1.0f / sqrtf(d.rvar[c] + FROST_BN_EPS), sc = d.gamma[c] * inv;
    d.coef[FC_SCALE * d.cpad + c] = sc; d.coef[FC_BIAS * d.cpad + c] = d.beta[c] - d.rmean[c] * sc;
    d.coef[FC_MEAN * d.cpad + c] = d.rmean[c]; d.coef[FC_INV * d.cpad + c] = inv;
  }
}
extern "C" int frost_float_bn_eval(const FrostFDesc* descs, int nlayers, void* stream) {
  if (nlayers <= 0) return 0;
  hipLaunchKernelGGL(k_f_bn_eval, dim3(8, nlayers), dim3(256), 0, as_stream(stream), descs);
  return frost_check_launch("float_bn_eval");
}
// backward coefficients: dc = K1*(g - S1/N - xhat*S2/N) = g*K1 + conv*E + F;  dgamma += S2, dbeta += S1
__global__ __launch_bounds__(256) void k_f_bwd_finalize(const FrostFDesc* dp, double count) {
  const FrostFDesc d = *dp;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d.cout) return;
  double S1 = 0.0, S2 = 0.0;
  for (int k = 0; k < FST_SLOTS; ++k) { S1 += d.stat[k * 4 * d.cpad + 2 * d.cpad + c]; S2 += d.stat[k * 4 * d.cpad + 3 * d.cpad + c]; }
  const float inv = d.coef[FC_INV * d.cpad + c], mean = d.coef[FC_MEAN * d.cpad + c];
  const float K1 = d.gamma[c] * inv;
  const float a = (float)(S1 / count), b = (float)(S2 / count);
  d.coef[FC_K1 * d.cpad + c] = K1;
  if (d.fp32) {        // fp32 mode evaluates dc = K1 * ((g - a) - xhat * b) with xhat = (conv - mean) * inv, the reference's own order: rows E / F carry b / a
    d.coef[FC_E * d.cpad + c] = b; d.coef[FC_F * d.cpad + c] = a;
  } else {             // bf16 mode: folded, dc = g*K1 + conv*E + F (two fmas; the cancellation it carries is below the bf16 storage error)
    d.coef[FC_E * d.cpad + c] = -K1 * b * inv;
    d.coef[FC_F * d.cpad + c] = -K1 * a + K1 * b * mean * inv;
  }
  d.dgamma[c] += (float)S2; d.dbeta[c] += (float)S1;
}
extern "C" int frost_float_bwd_finalize(const FrostFDesc* desc, int cout, int64_t count, void* stream) {
  hipLaunchKernelGGL(k_f_bwd_finalize, dim3((cout + 255) / 256), dim3(256), 0, as_stream(stream), desc, (double)count);
  return frost_check_launch("float_bwd_finalize");
}

// ------------------------------------------------------------------------------------------------ pointwise passes (bf16 MFMA)
// conv[p][co] = sum_k T[p][k] * W[co][k].  64-pixel tile per iteration staged in LDS with coalesced 16-byte loads; wave w owns 16 pixels
// and walks the channel tiles four at a time (WPX = 4), or the four waves split the channel tiles of a 16-pixel tile (WPX = 1: rows too
// long for 64 LDS rows).  Workgroups stride over the tiles; the statistics modes keep per-channel partial sums in LDS across all their
// tiles and flush them once (double atomics), so the number of global atomics is grid x channels, not tiles x channels.
//   F_STATS: sum / sum of squares of conv            F_EMIT: y = [relu](conv*scale + bias) -> bf16
//   F_BRED : S1 += g*m, S2 += g*m*xhat               F_BDC : dc = g*m*K1 + conv*E + F -> bf16        (m = z > 0 for ReLU layers)
// (the data gradient dx = dc . W^T is a plain bf16 GEMM: frost_infer_pw on the transposed pack, frost_pw.hip)
template <int MODE, int WPX, typename ET, int NT = 1, bool CL = false>
__global__ __launch_bounds__(256) void k_f_pw(const FrostFDesc* dp, const ET* __restrict__ T, const void* __restrict__ packv, int64_t npix,
                                              int cin, int cout, int cpad, int KB, int kstr, int relu, const ET* __restrict__ gy, int ldg,
                                              ET* __restrict__ y, int ldy, int64_t ntiles) {
  constexpr bool F32 = sizeof(ET) == 4;          // K step = 64 bytes of a row either way: 32 bf16 (one 16x16x32 MFMA) or 16 floats (four 16x16x4 MFMAs)
  static_assert(NT == 1 || WPX == 4, "several pixel sub-tiles per wave only in the pixel-split layout");
  const uint8_t* pack = (const uint8_t*)packv;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int IPX = 16 * WPX * NT, WCH = 4 / WPX;     // NT 16-pixel sub-tiles per wave: every weight fragment fetched from L2 feeds NT MFMAs
  constexpr bool RED = (MODE == F_STATS || MODE == F_BRED);
  const int wpx = w % WPX, wch = w / WPX;
  using SA = typename std::conditional<F32, double, float>::type;     // fp32 mode: statistics partials in double (the reference's CPU BatchNorm accumulates in double)
  SA* sacc = (SA*)(smem + IPX * kstr);
  const float* coef = (MODE == F_PLAIN) ? nullptr : dp->coef;
  if (RED) for (int i = tid; i < 2 * cpad; i += 256) sacc[i] = (SA)0;
  // CL: the layer's coefficient rows live in LDS for the kernel's life (narrow layers): the epilogue then has no dependent global load left but
  // gy, and gy is requested before the K loop -- the passes of the 112x112 / 56x56 layers were chains of five memory latencies per tile
  float* cl = (float*)(smem + IPX * kstr + (RED ? 2 * cpad * (int)sizeof(SA) : 0));
  if constexpr (CL) for (int i = tid; i < 7 * cpad; i += 256) cl[i] = coef[i];
  auto C4 = [&](int row, int cc) __attribute__((always_inline)) -> float4 {
    if constexpr (CL) return *(const float4*)(cl + row * cpad + cc);
    else return *(const float4*)(coef + row * cpad + cc);
  };
  const int rowb = cin * (int)sizeof(ET); const int U = (KB * 64) >> 4;
  const int CT = cpad >> 4;
  const float lo = relu ? 0.0f : -INFINITY; const bool hs = relu == F_ACT_HSWISH;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t p0 = tile * IPX;
    __syncthreads();
    for (int u = tid; u < IPX * U; u += 256) {
      const int row = u / U, col = (u - row * U) << 4;
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((p0 + row) < npix && col < rowb) v = *(const uint4*)((const uint8_t*)T + (p0 + row) * rowb + col);
      *(uint4*)(smem + row * kstr + col) = v;
    }
    __syncthreads();
    const int lrow0 = wpx * 16 * NT + j;                 // sub-tile t of this wave: LDS rows lrow0 + 16 t
    for (int ct0 = (blockIdx.y * WCH + wch) * 4; ct0 < CT; ct0 += 4 * WCH * gridDim.y) {      // gridDim.y > 1: low-resolution layers split their channel tiles over workgroups
      v4f acc[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = (v4f){0.f, 0.f, 0.f, 0.f};
      typename FEl<ET>::R4 gpre[NT][4];
      if constexpr (MODE == F_BRED || MODE == F_BDC) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int64_t prow = p0 + lrow0 + 16 * t; const int ch0 = (ct0 + m) * 16 + 4 * g;
            gpre[t][m] = FEl<ET>::zero4();
            if (ct0 + m < CT && ch0 < cout && prow < npix) gpre[t][m] = FEl<ET>::ldr4(gy + prow * ldg + ch0);
          }
      }
      for (int kb = 0; kb < KB; ++kb) {
        v4i bfr[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bfr[t] = *(const v4i*)(smem + (lrow0 + 16 * t) * kstr + kb * 64 + g * 16);
        v4i afr[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) if (ct0 + m < CT) afr[m] = *(const v4i*)(pack + ((((int64_t)(ct0 + m) * KB + kb) * 64 + lane) << 4));
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (ct0 + m < CT) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              if (F32) {         // lane (i, g) holds k = 4g .. 4g+3 of this 16-wide K block in both operands: instruction e contracts {4g + e}
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(__int_as_float(afr[m][e]), __int_as_float(bfr[t][e]), acc[t][m], 0, 0, 0);
              } else acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf16, afr[m]), __builtin_bit_cast(v8bf16, bfr[t]), acc[t][m], 0, 0, 0);
            }
          }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (ct0 + m >= CT) continue;                 // wave-uniform
        const int ch0 = (ct0 + m) * 16 + 4 * g;
        const bool cv = ch0 < cout;
        const int cc = cv ? ch0 : 0;
        float sc[4] = {0.f, 0.f, 0.f, 0.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE != F_PLAIN && MODE != F_STATS) {
          const float4 sc4 = C4(FC_SCALE, cc), bi4 = C4(FC_BIAS, cc);
          sc[0] = sc4.x; sc[1] = sc4.y; sc[2] = sc4.z; sc[3] = sc4.w; bi[0] = bi4.x; bi[1] = bi4.y; bi[2] = bi4.z; bi[3] = bi4.w;
        }
        SA rs[4] = {(SA)0, (SA)0, (SA)0, (SA)0}, rq[4] = {(SA)0, (SA)0, (SA)0, (SA)0};      // statistics partials of this lane over its NT sub-tiles
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int64_t prow = p0 + lrow0 + 16 * t;
          const bool ok = cv && prow < npix;
          const v4f a = acc[t][m];
          if constexpr (MODE == F_PLAIN) {
            if (ok) { const float o4[4] = {a[0], a[1], a[2], a[3]}; FEl<ET>::st4(y + prow * ldy + ch0, o4); }
          } else if constexpr (MODE == F_STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const SA v = ok ? (SA)a[r] : (SA)0; rs[r] += v; rq[r] += v * v; }
            if (y && ok) { const float o4[4] = {a[0], a[1], a[2], a[3]}; FEl<ET>::st4(y + prow * ldy + ch0, o4); }       // the conv output itself (training: kept for the element-wise passes)
          } else if constexpr (MODE == F_EMIT) {
            if (ok) {
              const float o4[4] = {f_act(fmaf(a[0], sc[0], bi[0]), lo, hs), f_act(fmaf(a[1], sc[1], bi[1]), lo, hs),
                                   f_act(fmaf(a[2], sc[2], bi[2]), lo, hs), f_act(fmaf(a[3], sc[3], bi[3]), lo, hs)};
              FEl<ET>::st4(y + prow * ldy + ch0, o4);
            }
          } else {
            float gm[4];
            FEl<ET>::cv4(gpre[t][m], gm);               // requested before the K loop (zeros outside the tensor)
            if (hs) {
#pragma unroll
              for (int r = 0; r < 4; ++r) gm[r] *= f_dhswish(fmaf(a[r], sc[r], bi[r]));
            } else if (relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r) if (!(fmaf(a[r], sc[r], bi[r]) > 0.0f)) gm[r] = 0.0f;
            }
            if constexpr (MODE == F_BRED) {
              const float4 iv4 = C4(FC_INV, cc), mu4 = C4(FC_MEAN, cc);
              const float iv[4] = {iv4.x, iv4.y, iv4.z, iv4.w}, mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) { const float xh = (a[r] - mu[r]) * iv[r]; if (ok) { rs[r] += (SA)gm[r]; rq[r] += (SA)gm[r] * (SA)xh; } }
            } else {   // F_BDC
              const float4 k4 = C4(FC_K1, cc), e4 = C4(FC_E, cc), f4 = C4(FC_F, cc);
              const float k1[4] = {k4.x, k4.y, k4.z, k4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w}, ff[4] = {f4.x, f4.y, f4.z, f4.w};
              if (ok) {
                float dcv[4];
                if (F32) {
                  const float4 iv4 = C4(FC_INV, cc), mu4 = C4(FC_MEAN, cc);
                  const float iv[4] = {iv4.x, iv4.y, iv4.z, iv4.w}, mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
#pragma unroll
                  for (int r = 0; r < 4; ++r) dcv[r] = k1[r] * ((gm[r] - ff[r]) - ((a[r] - mu[r]) * iv[r]) * ee[r]);
                } else {
#pragma unroll
                  for (int r = 0; r < 4; ++r) dcv[r] = fmaf(gm[r], k1[r], fmaf(a[r], ee[r], ff[r]));
                }
                FEl<ET>::st4(y + prow * ldy + ch0, dcv);
              }
            }
          }
        }
        if constexpr (RED) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { rs[r] = f_row_sum(rs[r]); rq[r] = f_row_sum(rq[r]); }
          if (j == 15 && cv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { atomicAdd(&sacc[ch0 + r], rs[r]); atomicAdd(&sacc[cpad + ch0 + r], rq[r]); }
          }
        }
      }
    }
  }
  if (RED) {
    __syncthreads();
    double* st = dp->stat + (size_t)((blockIdx.x + blockIdx.y) & (FST_SLOTS - 1)) * 4 * cpad + (MODE == F_BRED ? 2 * cpad : 0);
    for (int i = tid; i < 2 * cpad; i += 256) { const SA v = sacc[i]; if (v != (SA)0) atomicAdd(st + i, (double)v); }
  }
}
static int f_pw_nt() {        // FROST_FPW_NT=1/2/4 caps the pixel sub-tiles per wave (A/B knob; default 1: more resident workgroups beat the fragment reuse)
  static int v = -1;
  if (v < 0) { const char* e = getenv("FROST_FPW_NT"); v = e ? atoi(e) : 1; if (v != 1 && v != 2 && v != 4) v = 1; }
  return v;
}
static int f_red_cap() {      // workgroups of a statistics launch (FROST_FRED_CAP): every one ends with an LDS fold + 2*cpad global atomics -- fewer, fatter workgroups win (2048 -> 1024: -0.6 ms per step)
  static int v = -1;
  if (v < 0) { const char* e = getenv("FROST_FRED_CAP"); v = e ? atoi(e) : 1024; if (v < 64) v = 64; }
  return v;
}
static int f_pw_csplit_cap() {      // FROST_FPW_CSPLIT=1 turns the channel split off (A/B knob)
  static int v = -1;
  if (v < 0) { const char* e = getenv("FROST_FPW_CSPLIT"); v = e ? atoi(e) : 16; if (v < 1) v = 1; }
  return v;
}
template <int MODE, typename ET, int WPX, int NT, bool CL>
static void launch_f_pw_cl(const FrostFDesc* dp, const ET* T, const void* pack, int64_t npix, int cin, int cout, int cpad, int KB, int kstr, int relu,
                           const ET* gy, int ldg, ET* y, int ldy, size_t extra, hipStream_t s) {
  constexpr bool RED = (MODE == F_STATS || MODE == F_BRED);
  constexpr int IPX = 16 * WPX * NT;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)k_f_pw<MODE, WPX, ET, NT, CL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  const int64_t nt = (npix + IPX - 1) / IPX; int64_t grid = nt; if (RED && grid > f_red_cap()) grid = f_red_cap();
  // few pixel tiles and many channel tiles (the 14x14 / 7x7 expand layers): the groups of 4 channel tiles a workgroup walks are dealt out over
  // gridDim.y workgroups, each staging the (small) x tile again, until the launch has ~4 workgroups per CU
  const int groups = ((cpad >> 4) + 4 * (4 / WPX) - 1) / (4 * (4 / WPX));
  int csplit = 1;
  while (grid * csplit < 1024 && csplit * 2 <= groups && csplit < f_pw_csplit_cap()) csplit *= 2;
  hipLaunchKernelGGL((k_f_pw<MODE, WPX, ET, NT, CL>), dim3((unsigned)grid, (unsigned)csplit), dim3(256), (size_t)IPX * kstr + extra + (CL ? (size_t)7 * cpad * 4 : 0), s,
                     dp, T, pack, npix, cin, cout, cpad, KB, kstr, relu, gy, ldg, y, ldy, nt);
}
static size_t f_pw_cl_bytes(int mode, int cpad) {      // coefficient rows in LDS: the epilogue modes of narrow layers (FROST_FPW_CL=0 turns it off)
  static int on = -1;
  if (on < 0) { const char* e = getenv("FROST_FPW_CL"); on = e ? atoi(e) : 1; }
  const size_t b = (size_t)7 * cpad * 4;
  return (on && (mode == F_EMIT || mode == F_BRED || mode == F_BDC) && b <= 16 * 1024) ? b : 0;
}
template <int MODE, typename ET, int WPX, int NT>
static void launch_f_pw_nt(const FrostFDesc* dp, const ET* T, const void* pack, int64_t npix, int cin, int cout, int cpad, int KB, int kstr, int relu,
                           const ET* gy, int ldg, ET* y, int ldy, size_t extra, hipStream_t s) {
  if constexpr (MODE == F_EMIT || MODE == F_BRED || MODE == F_BDC) {
    if (f_pw_cl_bytes(MODE, cpad)) return launch_f_pw_cl<MODE, ET, WPX, NT, true>(dp, T, pack, npix, cin, cout, cpad, KB, kstr, relu, gy, ldg, y, ldy, extra, s);
  }
  launch_f_pw_cl<MODE, ET, WPX, NT, false>(dp, T, pack, npix, cin, cout, cpad, KB, kstr, relu, gy, ldg, y, ldy, extra, s);
}
template <int MODE, typename ET>
static int launch_f_pw(const FrostFDesc* dp, const ET* T, const void* pack, int64_t npix, int cin, int cout, int relu, const ET* gy,
                       int ldg, ET* y, int ldy, hipStream_t s) {
  const int KB = (cin * (int)sizeof(ET) + 63) / 64; const int kstr = KB * 64 + 16; const int cpad = round_up(cout, 16);
  constexpr bool RED = (MODE == F_STATS || MODE == F_BRED);
  const size_t extra = RED ? (size_t)2 * cpad * sizeof(typename std::conditional<sizeof(ET) == 4, double, float>::type) : 0;
  const size_t fix = extra + f_pw_cl_bytes(MODE, cpad);
  const int cap = f_pw_nt();
  // the x tile (64 NT rows) stays under 64 KB of LDS so that two workgroups share a CU; a tile needs >= 2 x the CU count to be worth widening
  if (cap >= 4 && (size_t)256 * kstr + fix <= 64 * 1024 && npix >= 256 * 1024)
    launch_f_pw_nt<MODE, ET, 4, 4>(dp, T, pack, npix, cin, cout, cpad, KB, kstr, relu, gy, ldg, y, ldy, extra, s);
  else if (cap >= 2 && (size_t)128 * kstr + fix <= 64 * 1024 && npix >= 64 * 1024)
    launch_f_pw_nt<MODE, ET, 4, 2>(dp, T, pack, npix, cin, cout, cpad, KB, kstr, relu, gy, ldg, y, ldy, extra, s);
  else if ((size_t)64 * kstr <= 48 * 1024)
    launch_f_pw_nt<MODE, ET, 4, 1>(dp, T, pack, npix, cin, cout, cpad, KB, kstr, relu, gy, ldg, y, ldy, extra, s);
  else {
    if ((size_t)16 * kstr + fix > 160 * 1024) { frost_set_error("float_pw: row too long for the LDS tile"); return 1; }
    launch_f_pw_nt<MODE, ET, 1, 1>(dp, T, pack, npix, cin, cout, cpad, KB, kstr, relu, gy, ldg, y, ldy, extra, s);
  }
  return frost_check_launch("float_pw");
}
template <typename ET>
static int float_pw_any(const FrostFDesc* desc, const ET* x, const void* pack, int64_t npix, int cin, int cout, int relu, int mode, const ET* gy, int ldg,
                        ET* out, int ldy, hipStream_t s) {
  switch (mode) {
    case F_STATS: return launch_f_pw<F_STATS, ET>(desc, x, pack, npix, cin, cout, relu, gy, ldg, out, ldy, s);
    case F_EMIT: return launch_f_pw<F_EMIT, ET>(desc, x, pack, npix, cin, cout, relu, gy, ldg, out, ldy, s);
    case F_BRED: return launch_f_pw<F_BRED, ET>(desc, x, pack, npix, cin, cout, relu, gy, ldg, out, ldy, s);
    case F_BDC: return launch_f_pw<F_BDC, ET>(desc, x, pack, npix, cin, cout, relu, gy, ldg, out, ldy, s);
    default: return launch_f_pw<F_PLAIN, ET>(desc, x, pack, npix, cin, cout, relu, gy, ldg, out, ldy, s);
  }
}
// x: [npix][cin] (cin = the row length: 64 for the im2col'd stem); pack: the layer's forward pack; gy / out rows may be slices of
// wider tensors (ldg / ldy = row length in elements)
extern "C" int frost_float_pw(const FrostFDesc* desc, const uint16_t* x, const uint16_t* pack, int64_t npix, int cin, int cout, int relu, int mode,
                              const uint16_t* gy, int ldg, uint16_t* out, int ldy, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 4 == 0, "float_pw: cin must be a multiple of 8, cout of 4");
  FROST_REQUIRE(mode >= 0 && mode <= 3, "float_pw: mode 0..3");
  return float_pw_any<uint16_t>(desc, x, pack, npix, cin, cout, relu, mode, gy, ldg, out, ldy, as_stream(stream));
}
// fp32 activations; mode 4 = plain GEMM out = x . pack^T (the data gradient: x = dc, pack = the layer's transposed pack, desc unused)
extern "C" int frost_float_pw_f32(const FrostFDesc* desc, const float* x, const float* pack, int64_t npix, int cin, int cout, int relu, int mode,
                                  const float* gy, int ldg, float* out, int ldy, void* stream) {
  FROST_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "float_pw_f32: channel counts must be multiples of 4");
  FROST_REQUIRE(mode >= 0 && mode <= 4, "float_pw_f32: mode 0..4");
  return float_pw_any<float>(desc, x, pack, npix, cin, cout, relu, mode, gy, ldg, out, ldy, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------ depthwise passes (fp32 FMA)
typedef float v2f __attribute__((ext_vector_type(2)));
template <typename ET> struct FdwRaw;          // two adjacent channels as loaded (conversion deferred to the arithmetic phase)
template <> struct FdwRaw<uint16_t> {
  using T = uint32_t;
  static __device__ __forceinline__ T ld(const uint16_t* p) { return *(const uint32_t*)p; }
  static __device__ __forceinline__ v2f cv(T v) { return (v2f){__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)}; }
};
template <> struct FdwRaw<float> {
  using T = v2f;
  static __device__ __forceinline__ T ld(const float* p) { return *(const v2f*)p; }
  static __device__ __forceinline__ v2f cv(T v) { return v; }
};
__device__ __forceinline__ v2f fdw_ld2(const uint16_t* p) { const uint32_t v = *(const uint32_t*)p; return (v2f){__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)}; }
__device__ __forceinline__ v2f fdw_ld2(const float* p) { return *(const v2f*)p; }
template <typename ET> struct FdwOut;
template <> struct FdwOut<uint16_t> { static __device__ __forceinline__ void st(uint16_t* p, v2f v) { *(uint32_t*)p = cvt_pk_bf16(v[0], v[1]); } };
template <> struct FdwOut<float> { static __device__ __forceinline__ void st(float* p, v2f v) { *(v2f*)p = v; } };
// one thread = 8 channels (fixed for the thread's whole life, so statistics stay in registers) x a strided set of output-pixel PAIRS
// (two horizontally adjacent outputs share the S + K input columns of a kernel row and every tap's weights: 1.5-1.7x fewer loads)
#define FDW_WO 2
template <int MODE, int K, int S, typename ET>
__global__ __launch_bounds__(256) void k_f_dw(const FrostFDesc* dp, const ET* __restrict__ x, int n, int h, int w, int c, int cpad, int ho, int wo,
                                              int relu, const ET* __restrict__ gy, ET* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr bool RED = (MODE == F_STATS || MODE == F_BRED);
  constexpr int PAD = (K - 1) / 2, SPAN = (FDW_WO - 1) * S + K;
  using SA = typename std::conditional<sizeof(ET) == 4, double, float>::type;     // fp32 mode: statistics partials in double
  SA* sacc = (SA*)smem;
  const int tid = threadIdx.x;
  if (RED) { for (int i = tid; i < 2 * cpad; i += 256) sacc[i] = (SA)0; __syncthreads(); }
  const int c8n = c >> 3; const int wog = (wo + FDW_WO - 1) / FDW_WO;
  // XCD-aware map (workgroups go to the 8 XCDs round-robin): XCD x owns the x-th contiguous eighth of the output units, its workgroups stride inside it,
  // so the K input rows neighbouring units share are fetched into ONE L2 instead of eight (gridDim.x is a multiple of 8)
  const int xcd = blockIdx.x & 7; const int64_t nthreads = (int64_t)(gridDim.x >> 3) * 256; const int64_t PP = nthreads / c8n;
  const int64_t t = (int64_t)(blockIdx.x >> 3) * 256 + tid;
  const int c8 = (int)(t % c8n); const int64_t slot = t / c8n;
  const int ch = c8 * 8;
  const int64_t nunits_all = (int64_t)n * ho * wog; const int64_t upx = (nunits_all + 7) >> 3;
  const int64_t u_lo = xcd * upx; const int64_t nunits = (u_lo + upx < nunits_all) ? u_lo + upx : nunits_all;
  const float* wf = (const float*)dp->pack; const float* coef = dp->coef;
  float sc[8], bi[8], c2[8], c3[8], c4[8], c5[8], c6[8];
  if (MODE != F_STATS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = coef[FC_SCALE * cpad + ch + e]; bi[e] = coef[FC_BIAS * cpad + ch + e];
      if (MODE == F_BRED) { c2[e] = coef[FC_MEAN * cpad + ch + e]; c3[e] = coef[FC_INV * cpad + ch + e]; }
      if (MODE == F_BDC) { c2[e] = coef[FC_K1 * cpad + ch + e]; c3[e] = coef[FC_E * cpad + ch + e]; c4[e] = coef[FC_F * cpad + ch + e];
                           if (sizeof(ET) == 4) { c5[e] = coef[FC_MEAN * cpad + ch + e]; c6[e] = coef[FC_INV * cpad + ch + e]; } }
    }
  }
  SA s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = (SA)0; q[e] = (SA)0; }
  const float lo = relu ? 0.0f : -INFINITY; const bool hs = relu == F_ACT_HSWISH;
  if (slot < PP) {
    for (int64_t u = u_lo + slot; u < nunits; u += PP) {
      int64_t pp = u; const int oxg = (int)(pp % wog); pp /= wog; const int oy = (int)(pp % ho); const int in = (int)(pp / ho);
      const int ox0 = oxg * FDW_WO, ix0 = ox0 * S - PAD;
      float acc[FDW_WO][8];
#pragma unroll
      for (int o = 0; o < FDW_WO; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = 0.0f;
      typename FEl<ET>::R8 graw[FDW_WO];          // the output gradient is requested before the taps: its latency runs under theirs
      if constexpr (MODE == F_BRED || MODE == F_BDC) {
#pragma unroll
        for (int o = 0; o < FDW_WO; ++o) {
          graw[o] = FEl<ET>::zero8();
          if (ox0 + o < wo) graw[o] = FEl<ET>::ldr8(gy + (((int64_t)in * ho + oy) * wo + ox0 + o) * c + ch);
        }
      }
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S - PAD + ky; if (iy < 0 || iy >= h) continue;
        const ET* rowp = x + ((int64_t)in * h + iy) * w * c + ch;
        float col[SPAN][8];
#pragma unroll
        for (int qx = 0; qx < SPAN; ++qx) {
          const int ix = ix0 + qx;
#pragma unroll
          for (int e = 0; e < 8; ++e) col[qx][e] = 0.0f;
          if (ix >= 0 && ix < w) FEl<ET>::ld8(rowp + (int64_t)ix * c, col[qx]);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float* wp = wf + (ky * K + kx) * cpad + ch;
          const float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int o = 0; o < FDW_WO; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(col[o * S + kx][e], wv[e], acc[o][e]);
        }
      }
#pragma unroll
      for (int o = 0; o < FDW_WO; ++o) {
        if (ox0 + o >= wo) continue;
        const int64_t p = ((int64_t)in * ho + oy) * wo + ox0 + o;
        if constexpr (MODE == F_STATS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { s[e] += (SA)acc[o][e]; q[e] += (SA)acc[o][e] * (SA)acc[o][e]; }
          if (y) FEl<ET>::st8(y + p * c + ch, acc[o]);
        } else if constexpr (MODE == F_EMIT) {
          float o8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = f_act(fmaf(acc[o][e], sc[e], bi[e]), lo, hs);
          FEl<ET>::st8(y + p * c + ch, o8);
        } else {
          float gm[8];
          FEl<ET>::cv8(graw[o], gm);
          if (hs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gm[e] *= f_dhswish(fmaf(acc[o][e], sc[e], bi[e]));
          } else if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (!(fmaf(acc[o][e], sc[e], bi[e]) > 0.0f)) gm[e] = 0.0f;
          }
          if constexpr (MODE == F_BRED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += (SA)gm[e]; q[e] += (SA)gm[e] * (SA)((acc[o][e] - c2[e]) * c3[e]); }
          } else {
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              o8[e] = (sizeof(ET) == 4) ? c2[e] * ((gm[e] - c4[e]) - ((acc[o][e] - c5[e]) * c6[e]) * c3[e]) : fmaf(gm[e], c2[e], fmaf(acc[o][e], c3[e], c4[e]));
            FEl<ET>::st8(y + p * c + ch, o8);
          }
        }
      }
    }
  }
  if (RED) {
    if (slot < PP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { atomicAdd(&sacc[ch + e], s[e]); atomicAdd(&sacc[cpad + ch + e], q[e]); }
    }
    __syncthreads();
    double* st = dp->stat + (size_t)((blockIdx.x + blockIdx.y) & (FST_SLOTS - 1)) * 4 * cpad + (MODE == F_BRED ? 2 * cpad : 0);
    for (int i = tid; i < 2 * cpad; i += 256) { const SA v = sacc[i]; if (v != (SA)0) atomicAdd(st + i, (double)v); }
  }
}
template <int K, int S, typename ET>
static void launch_f_dw(int mode, int64_t grid, size_t lds, hipStream_t s, const FrostFDesc* desc, const ET* x, int n, int h, int w, int c, int cpad, int ho,
                        int wo, int relu, const ET* gy, ET* out) {
#define FDW_LAUNCH(M) hipLaunchKernelGGL((k_f_dw<M, K, S, ET>), dim3((unsigned)grid), dim3(256), lds, s, desc, x, n, h, w, c, cpad, ho, wo, relu, gy, out)
  switch (mode) { case F_STATS: FDW_LAUNCH(F_STATS); break; case F_EMIT: FDW_LAUNCH(F_EMIT); break; case F_BRED: FDW_LAUNCH(F_BRED); break; default: FDW_LAUNCH(F_BDC); }
#undef FDW_LAUNCH
}
// Row-walking form of the forward passes (F_STATS: conv output + statistics, F_EMIT: y = [relu](c*scale + bias)) -- the default: a wave owns 2 * GW channels of
// 64 / GW output rows and walks along them with the k x k input window and its channels' weights in registers (see k_f_dw_wgrad_row for the rotation of the
// window slots); per output pixel S*k loads of 4 bytes, k*k packed FMAs in the same (ky, kx) order as k_f_dw -- the conv output is bit-identical --, one store.
// SRC: x holds the KEPT CONV OUTPUT of the layer in front (conv1 of the bottleneck) instead of its activation: y1 = [relu](c1 * scale + bias) is applied as the
// window is loaded (padding stays zero), so conv1's element-wise emit pass -- one read of c1 and one write of y1 -- never runs in training.
template <int MODE, int K, int S, int GW, int SRC, typename ET>
__global__ __launch_bounds__(256) void k_f_dw_row(const FrostFDesc* dp, const ET* __restrict__ x, int n, int h, int w, int c, int cpad, int ho, int wo, int relu,
                                                  ET* __restrict__ y, const FrostFDesc* dsrc, int relu_src) {
  constexpr int PAD = (K - 1) / 2, RPW = 64 / GW;
  using SA = typename std::conditional<sizeof(ET) == 4, double, float>::type;     // fp32 mode: statistics partials in double
  __shared__ SA sred[2][2 * GW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cl = (lane % GW) * 2, rsub = lane / GW;
  const int ch = blockIdx.y * (2 * GW) + cl; const bool live = ch < c;
  if (MODE == F_STATS) { for (int i = tid; i < 4 * GW; i += 256) (&sred[0][0])[i] = (SA)0; __syncthreads(); }
  SA s0 = 0, s1 = 0, q0 = 0, q1 = 0;
  if (live) {
    const float* wf = (const float*)dp->pack;
    v2f wt[K][K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) wt[ky][kx] = *(const v2f*)(wf + (ky * K + kx) * cpad + ch);
    v2f sc = (v2f){1.0f, 1.0f}, bi = (v2f){0.0f, 0.0f};
    if (MODE == F_EMIT) { sc = *(const v2f*)(dp->coef + FC_SCALE * cpad + ch); bi = *(const v2f*)(dp->coef + FC_BIAS * cpad + ch); }
    v2f ssc = (v2f){1.0f, 1.0f}, sbi = (v2f){0.0f, 0.0f}; float slo = -INFINITY; constexpr bool shs = SRC == 2;          // (the source's activation is a template value: a per-element select costs the stencil kernels 20 - 60 %)
    if (SRC) { ssc = *(const v2f*)(dsrc->coef + FC_SCALE * cpad + ch); sbi = *(const v2f*)(dsrc->coef + FC_BIAS * cpad + ch); slo = relu_src ? 0.0f : -INFINITY; }
    auto xf = [&](v2f v) __attribute__((always_inline)) { return SRC ? (v2f){f_act(fmaf(v[0], ssc[0], sbi[0]), slo, shs), f_act(fmaf(v[1], ssc[1], sbi[1]), slo, shs)} : v; };
    const float lo = relu ? 0.0f : -INFINITY; const bool hs = relu == F_ACT_HSWISH;
    const int rows = n * ho;
    for (int r = (blockIdx.x * 4 + wv) * RPW + rsub; r < rows; r += gridDim.x * 4 * RPW) {
      const int in = r / ho, oy = r - in * ho;
      const ET* xr[K]; bool rv[K];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S - PAD + ky; rv[ky] = (unsigned)iy < (unsigned)h;
        xr[ky] = x + ((int64_t)(in * h + (rv[ky] ? iy : 0)) * w) * c + ch;
      }
      ET* orow = y ? y + (int64_t)r * wo * c + ch : nullptr;
      v2f win[K][K];
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) win[ky][kx] = (v2f){0.0f, 0.0f};
#pragma unroll
      for (int kx = 0; kx < K - S; ++kx) {
        const int ix = kx - PAD;
        const bool cv = (unsigned)ix < (unsigned)w; const int64_t off = (int64_t)(cv ? ix : 0) * c;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) { const v2f v = xf(fdw_ld2(xr[ky] + off)); win[ky][kx] = (cv && rv[ky]) ? v : (v2f){0.0f, 0.0f}; }
      }
      for (int ox0 = 0; ox0 < wo; ox0 += K) {
        typename FdwRaw<ET>::T raw[K][S][K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
          const int ox = ox0 + u;
#pragma unroll
          for (int j = 0; j < S; ++j) {
            const int ix = ox * S - PAD + (K - S + j);
            const int64_t off = (int64_t)(((unsigned)ix < (unsigned)w) ? ix : 0) * c;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) raw[u][j][ky] = FdwRaw<ET>::ld(xr[ky] + off);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < K; ++u) {
          const int ox = ox0 + u;
#pragma unroll
          for (int j = 0; j < S; ++j) {
            const int kx = K - S + j, ix = ox * S - PAD + kx, slot = (u * S + kx) % K;
            const bool cv = (unsigned)ix < (unsigned)w;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) win[ky][slot] = (cv && rv[ky]) ? xf(FdwRaw<ET>::cv(raw[u][j][ky])) : (v2f){0.0f, 0.0f};
          }
          v2f acc = (v2f){0.0f, 0.0f};
#pragma unroll
          for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc = __builtin_elementwise_fma(win[ky][(u * S + kx) % K], wt[ky][kx], acc);
          if (ox < wo) {
            if (MODE == F_STATS) {
              s0 += (SA)acc[0]; s1 += (SA)acc[1]; q0 += (SA)acc[0] * (SA)acc[0]; q1 += (SA)acc[1] * (SA)acc[1];
              if (orow) FdwOut<ET>::st(orow + (int64_t)ox * c, acc);
            } else {
              const v2f o = (v2f){f_act(fmaf(acc[0], sc[0], bi[0]), lo, hs), f_act(fmaf(acc[1], sc[1], bi[1]), lo, hs)};
              FdwOut<ET>::st(orow + (int64_t)ox * c, o);
            }
          }
        }
      }
    }
  }
  if (MODE == F_STATS) {
    if (live) { atomicAdd(&sred[0][cl], s0); atomicAdd(&sred[0][cl + 1], s1); atomicAdd(&sred[1][cl], q0); atomicAdd(&sred[1][cl + 1], q1); }
    __syncthreads();
    double* st = dp->stat + (size_t)((blockIdx.x + blockIdx.y) & (FST_SLOTS - 1)) * 4 * cpad;
    for (int i = tid; i < 4 * GW; i += 256) {
      const int which = i / (2 * GW), cc = blockIdx.y * (2 * GW) + i % (2 * GW);
      const SA v = sred[which][i % (2 * GW)];
      if (cc < c && v != (SA)0) atomicAdd(st + which * cpad + cc, (double)v);
    }
  }
}
template <int MODE, int K, int S, int GW, typename ET>
static void launch_f_dw_row2(const FrostFDesc* desc, const ET* x, int n, int h, int w, int c, int cpad, int ho, int wo, int relu, ET* out, hipStream_t s,
                             const FrostFDesc* dsrc, int relu_src) {
  static int wgs = -1;
  if (wgs < 0) { const char* e = getenv("FROST_FDW_ROW_WGS"); wgs = e ? atoi(e) : (K == 3 ? 2048 : 1024); }
  constexpr int RPW = 64 / GW;
  const int groups = (c + 2 * GW - 1) / (2 * GW); const int rows = n * ho;
  int gx = (rows + 4 * RPW - 1) / (4 * RPW); int cap = wgs / groups; if (cap < 1) cap = 1; if (gx > cap) gx = cap;
  if (dsrc && relu_src == F_ACT_HSWISH) hipLaunchKernelGGL((k_f_dw_row<MODE, K, S, GW, 2, ET>), dim3((unsigned)gx, groups), dim3(256), 0, s, desc, x, n, h, w, c, cpad, ho, wo, relu, out, dsrc, relu_src);
  else if (dsrc) hipLaunchKernelGGL((k_f_dw_row<MODE, K, S, GW, 1, ET>), dim3((unsigned)gx, groups), dim3(256), 0, s, desc, x, n, h, w, c, cpad, ho, wo, relu, out, dsrc, relu_src);
  else hipLaunchKernelGGL((k_f_dw_row<MODE, K, S, GW, 0, ET>), dim3((unsigned)gx, groups), dim3(256), 0, s, desc, x, n, h, w, c, cpad, ho, wo, relu, out, dsrc, relu_src);
}
template <int MODE, int K, int S, typename ET>
static void launch_f_dw_row(const FrostFDesc* desc, const ET* x, int n, int h, int w, int c, int cpad, int ho, int wo, int relu, ET* out, hipStream_t s,
                            const FrostFDesc* dsrc, int relu_src) {
  if (c > 64) launch_f_dw_row2<MODE, K, S, 64, ET>(desc, x, n, h, w, c, cpad, ho, wo, relu, out, s, dsrc, relu_src);
  else if (c > 32) launch_f_dw_row2<MODE, K, S, 32, ET>(desc, x, n, h, w, c, cpad, ho, wo, relu, out, s, dsrc, relu_src);
  else launch_f_dw_row2<MODE, K, S, 16, ET>(desc, x, n, h, w, c, cpad, ho, wo, relu, out, s, dsrc, relu_src);
}
template <int MODE, typename ET>
static void launch_f_dw_row_ks(const FrostFDesc* desc, const ET* x, int n, int h, int w, int c, int cpad, int k, int stride, int ho, int wo, int relu, ET* out, hipStream_t s,
                               const FrostFDesc* dsrc, int relu_src) {
  if (k == 3 && stride == 1) launch_f_dw_row<MODE, 3, 1, ET>(desc, x, n, h, w, c, cpad, ho, wo, relu, out, s, dsrc, relu_src);
  else if (k == 3) launch_f_dw_row<MODE, 3, 2, ET>(desc, x, n, h, w, c, cpad, ho, wo, relu, out, s, dsrc, relu_src);
  else if (stride == 1) launch_f_dw_row<MODE, 5, 1, ET>(desc, x, n, h, w, c, cpad, ho, wo, relu, out, s, dsrc, relu_src);
  else launch_f_dw_row<MODE, 5, 2, ET>(desc, x, n, h, w, c, cpad, ho, wo, relu, out, s, dsrc, relu_src);
}
template <typename ET>
static int float_dw_any(const FrostFDesc* desc, const ET* x, int n, int h, int w, int c, int k, int stride, int relu, int mode, const ET* gy, ET* out, hipStream_t s,
                        const FrostFDesc* dsrc = nullptr, int relu_src = 0) {
  FROST_REQUIRE(c % 8 == 0, "float_dw: channels must be a multiple of 8");
  FROST_REQUIRE(mode >= 0 && mode <= 3, "float_dw: mode 0..3");
  FROST_REQUIRE((k == 3 || k == 5) && (stride == 1 || stride == 2), "float_dw: 3x3 / 5x5, stride 1 / 2");
  const int pad = (k - 1) / 2; const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
  const int cpad = round_up(c, 16); const int c8n = c >> 3;
  static int row_form = -1;
  if (row_form < 0) { const char* e = getenv("FROST_FDW_ROW"); row_form = e ? atoi(e) : 1; }
  FROST_REQUIRE(!dsrc || mode == F_STATS || mode == F_EMIT, "float_dw_src: forward modes only (0: statistics + conv output, 1: emit)");
  if ((row_form || dsrc) && (mode == F_STATS || mode == F_EMIT)) {
    if (mode == F_STATS) launch_f_dw_row_ks<F_STATS, ET>(desc, x, n, h, w, c, cpad, k, stride, ho, wo, relu, out, s, dsrc, relu_src);
    else launch_f_dw_row_ks<F_EMIT, ET>(desc, x, n, h, w, c, cpad, k, stride, ho, wo, relu, out, s, dsrc, relu_src);
    return frost_check_launch("float_dw");
  }
  const int64_t tot = (int64_t)n * ho * ((wo + FDW_WO - 1) / FDW_WO) * c8n;
  const bool red = (mode == F_STATS || mode == F_BRED);
  int64_t grid = (tot + 255) / 256; const int64_t cap = red ? f_red_cap() : 8192; if (grid > cap) grid = cap;
  const int64_t gmin = (c8n + 255) / 256; if (grid < 8 * gmin) grid = 8 * gmin;      // every channel group needs at least one thread in every XCD's share
  grid = (grid + 7) & ~(int64_t)7;
  const size_t lds = red ? (size_t)2 * cpad * (sizeof(ET) == 4 ? 8 : 4) : 0;
  if (k == 3 && stride == 1) launch_f_dw<3, 1, ET>(mode, grid, lds, s, desc, x, n, h, w, c, cpad, ho, wo, relu, gy, out);
  else if (k == 3) launch_f_dw<3, 2, ET>(mode, grid, lds, s, desc, x, n, h, w, c, cpad, ho, wo, relu, gy, out);
  else if (stride == 1) launch_f_dw<5, 1, ET>(mode, grid, lds, s, desc, x, n, h, w, c, cpad, ho, wo, relu, gy, out);
  else launch_f_dw<5, 2, ET>(mode, grid, lds, s, desc, x, n, h, w, c, cpad, ho, wo, relu, gy, out);
  return frost_check_launch("float_dw");
}
extern "C" int frost_float_dw(const FrostFDesc* desc, const uint16_t* x, int n, int h, int w, int c, int k, int stride, int relu, int mode,
                              const uint16_t* gy, uint16_t* out, void* stream) {
  return float_dw_any<uint16_t>(desc, x, n, h, w, c, k, stride, relu, mode, gy, out, as_stream(stream));
}
extern "C" int frost_float_dw_f32(const FrostFDesc* desc, const float* x, int n, int h, int w, int c, int k, int stride, int relu, int mode,
                                  const float* gy, float* out, void* stream) {
  return float_dw_any<float>(desc, x, n, h, w, c, k, stride, relu, mode, gy, out, as_stream(stream));
}
// forward modes with the input given as the kept conv output of the layer in front (descriptor desc_src, ReLU flag relu_src): see k_f_dw_row<SRC>
extern "C" int frost_float_dw_src(const FrostFDesc* desc, const uint16_t* conv_src, const FrostFDesc* desc_src, int relu_src, int n, int h, int w, int c, int k, int stride,
                                  int relu, int mode, uint16_t* out, void* stream) {
  FROST_REQUIRE(desc_src && conv_src, "float_dw_src: descriptor and kept conv output of the layer in front");
  return float_dw_any<uint16_t>(desc, conv_src, n, h, w, c, k, stride, relu, mode, nullptr, out, as_stream(stream), desc_src, relu_src);
}
extern "C" int frost_float_dw_src_f32(const FrostFDesc* desc, const float* conv_src, const FrostFDesc* desc_src, int relu_src, int n, int h, int w, int c, int k, int stride,
                                      int relu, int mode, float* out, void* stream) {
  FROST_REQUIRE(desc_src && conv_src, "float_dw_src: descriptor and kept conv output of the layer in front");
  return float_dw_any<float>(desc, conv_src, n, h, w, c, k, stride, relu, mode, nullptr, out, as_stream(stream), desc_src, relu_src);
}

// ------------------------------------------------------------------------------------------------ element-wise passes over a kept conv output
// A training forward keeps every layer's convolution output c (the statistics pass stores it), so the passes that only need c per element do not
// recompute it:  F_EMIT: y = [relu](c*scale + bias)     F_BRED: S1 += g*m, S2 += g*m*xhat     F_BDC: dc = g*m*K1 + c*E + F     (m = z > 0 for ReLU layers)
// c: dense [npix][ch]; gy / out rows of ldg / ldy elements.  Thread = 8 channels (fixed: coefficients and partial sums in registers) x strided pixels.
template <int MODE, typename ET, bool HS = false>          // HS: a hard-swish layer (its own instances: a run-time test costs the ReLU instances 7 - 15 %)
__global__ __launch_bounds__(256) void k_f_ew(const FrostFDesc* dp, const ET* __restrict__ cv, int64_t npix, int c, int cpad, int relu,
                                              const ET* __restrict__ gy, int ldg, ET* __restrict__ y, int ldy) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  using SA = typename std::conditional<sizeof(ET) == 4, double, float>::type;
  constexpr bool F32 = sizeof(ET) == 4;
  SA* sacc = (SA*)smem;
  const int tid = threadIdx.x;
  if (MODE == F_BRED) { for (int i = tid; i < 2 * cpad; i += 256) sacc[i] = (SA)0; __syncthreads(); }
  const int c8n = c >> 3;
  const int64_t PP = ((int64_t)gridDim.x * 256) / c8n;
  const int64_t t = (int64_t)blockIdx.x * 256 + tid;
  const int c8 = (int)(t % c8n); const int64_t slot = t / c8n;
  const int ch = c8 * 8;
  const float* coef = dp->coef;
  float sc[8], bi[8], c2[8], c3[8], c4[8], c5[8], c6[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = coef[FC_SCALE * cpad + ch + e]; bi[e] = coef[FC_BIAS * cpad + ch + e];
    if (MODE == F_BRED) { c2[e] = coef[FC_MEAN * cpad + ch + e]; c3[e] = coef[FC_INV * cpad + ch + e]; }
    if (MODE == F_BDC) { c2[e] = coef[FC_K1 * cpad + ch + e]; c3[e] = coef[FC_E * cpad + ch + e]; c4[e] = coef[FC_F * cpad + ch + e];
                         if (F32) { c5[e] = coef[FC_MEAN * cpad + ch + e]; c6[e] = coef[FC_INV * cpad + ch + e]; } }
  }
  SA s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = (SA)0; q[e] = (SA)0; }
  const float lo = relu ? 0.0f : -INFINITY; constexpr bool hs = HS;
  if (slot < PP) {
    for (int64_t p0 = slot; p0 < npix; p0 += 4 * PP) {          // four pixels per trip: their loads are in flight together
      typename FEl<ET>::R8 ra[4], rg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + u * PP;
        ra[u] = FEl<ET>::zero8(); rg[u] = FEl<ET>::zero8();
        if (p < npix) { ra[u] = FEl<ET>::ldr8(cv + p * c + ch); if (MODE != F_EMIT) rg[u] = FEl<ET>::ldr8(gy + p * ldg + ch); }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + u * PP;
        if (p >= npix) continue;
        float a[8];
        FEl<ET>::cv8(ra[u], a);
        if constexpr (MODE == F_EMIT) {
          float o8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = f_act(fmaf(a[e], sc[e], bi[e]), lo, hs);
          FEl<ET>::st8(y + p * ldy + ch, o8);
        } else {
          float gm[8];
          FEl<ET>::cv8(rg[u], gm);
          if (hs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gm[e] *= f_dhswish(fmaf(a[e], sc[e], bi[e]));
          } else if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (!(fmaf(a[e], sc[e], bi[e]) > 0.0f)) gm[e] = 0.0f;
          }
          if constexpr (MODE == F_BRED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += (SA)gm[e]; q[e] += (SA)gm[e] * (SA)((a[e] - c2[e]) * c3[e]); }
          } else {
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              o8[e] = F32 ? c2[e] * ((gm[e] - c4[e]) - ((a[e] - c5[e]) * c6[e]) * c3[e]) : fmaf(gm[e], c2[e], fmaf(a[e], c3[e], c4[e]));
            FEl<ET>::st8(y + p * ldy + ch, o8);
          }
        }
      }
    }
  }
  if (MODE == F_BRED) {
    if (slot < PP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { atomicAdd(&sacc[ch + e], s[e]); atomicAdd(&sacc[cpad + ch + e], q[e]); }
    }
    __syncthreads();
    double* st = dp->stat + (size_t)(blockIdx.x & (FST_SLOTS - 1)) * 4 * cpad + 2 * cpad;
    for (int i = tid; i < 2 * cpad; i += 256) { const SA v = sacc[i]; if (v != (SA)0) atomicAdd(st + i, (double)v); }
  }
}
template <typename ET>
static int float_ew_any(const FrostFDesc* desc, const ET* cv, int64_t npix, int c, int relu, int mode, const ET* gy, int ldg, ET* out, int ldy, hipStream_t s) {
  FROST_REQUIRE(c % 8 == 0 && ldg % 8 == 0 && ldy % 8 == 0, "float_ew: channels and row strides must be multiples of 8");
  FROST_REQUIRE(mode >= 1 && mode <= 3, "float_ew: mode 1 (emit), 2 (reduce), 3 (dc)");
  const int cpad = round_up(c, 16); const int c8n = c >> 3;
  const int64_t tot = npix * c8n;
  static int rcap = -1;
  if (rcap < 0) { const char* e = getenv("FROST_FEW_CAP"); rcap = e ? atoi(e) : 256; }      // (512 -> 256: -0.23 ms per step; 128: +0.2)
  int64_t grid = (tot + 255) / 256; const int64_t cap = (mode == F_BRED) ? rcap : 8192; if (grid > cap) grid = cap;
  const int64_t gmin = (c8n + 255) / 256; if (grid < gmin) grid = gmin;
  const size_t lds = (mode == F_BRED) ? (size_t)2 * cpad * (sizeof(ET) == 4 ? 8 : 4) : 0;
#define FEW_GO(M) { if (relu == F_ACT_HSWISH) hipLaunchKernelGGL((k_f_ew<M, ET, true>), dim3((unsigned)grid), dim3(256), lds, s, desc, cv, npix, c, cpad, relu, gy, ldg, out, ldy); \
                    else hipLaunchKernelGGL((k_f_ew<M, ET, false>), dim3((unsigned)grid), dim3(256), lds, s, desc, cv, npix, c, cpad, relu, gy, ldg, out, ldy); }
  if (mode == F_EMIT) FEW_GO(F_EMIT) else if (mode == F_BRED) FEW_GO(F_BRED) else FEW_GO(F_BDC)
#undef FEW_GO
  return frost_check_launch("float_ew");
}
extern "C" int frost_float_ew(const FrostFDesc* desc, const uint16_t* conv, int64_t npix, int c, int relu, int mode, const uint16_t* gy, int ldg,
                              uint16_t* out, int ldy, void* stream) {
  return float_ew_any<uint16_t>(desc, conv, npix, c, relu, mode, gy, ldg, out, ldy, as_stream(stream));
}
extern "C" int frost_float_ew_f32(const FrostFDesc* desc, const float* conv, int64_t npix, int c, int relu, int mode, const float* gy, int ldg,
                                  float* out, int ldy, void* stream) {
  return float_ew_any<float>(desc, conv, npix, c, relu, mode, gy, ldg, out, ldy, as_stream(stream));
}

// depthwise data gradient: dx[n][iy][ix][c] = sum_{ky,kx} dc[n][(iy+pad-ky)/s][(ix+pad-kx)/s][c] * w[c][ky][kx]   (where divisible / in range)
template <typename ET>
__global__ __launch_bounds__(256) void k_f_dw_dgrad(const FrostFDesc* dp, const ET* __restrict__ dc, int n, int h, int w, int c, int cpad, int k, int stride,
                                                    int ho, int wo, ET* __restrict__ dx) {
  const int c8n = c >> 3; const int pad = (k - 1) / 2;
  const int64_t tot = (int64_t)n * h * w * c8n;
  const float* wf = (const float*)dp->pack;
  const int64_t ipx = (((tot + 7) >> 3) + c8n - 1) / c8n * c8n;        // XCD x owns the x-th contiguous eighth of the input pixels (see k_f_dw)
  const int64_t i_lo = (blockIdx.x & 7) * ipx; const int64_t i_hi = (i_lo + ipx < tot) ? i_lo + ipx : tot;
  for (int64_t i = i_lo + (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x; i < i_hi; i += (int64_t)(gridDim.x >> 3) * 256) {
    const int c8 = (int)(i % c8n); int64_t p = i / c8n; const int ix = (int)(p % w); p /= w; const int iy = (int)(p % h); const int in = (int)(p / h);
    const int ch = c8 * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    for (int ky = 0; ky < k; ++ky) {
      const int ty = iy + pad - ky; if (ty < 0 || ty % stride) continue; const int oy = ty / stride; if (oy >= ho) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int tx = ix + pad - kx; if (tx < 0 || tx % stride) continue; const int ox = tx / stride; if (ox >= wo) continue;
        float v[8];
        FEl<ET>::ld8(dc + (((int64_t)in * ho + oy) * wo + ox) * c + ch, v);
        const float* wp = wf + (ky * k + kx) * cpad + ch;
        const float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
        acc[0] = fmaf(v[0], w0.x, acc[0]); acc[1] = fmaf(v[1], w0.y, acc[1]); acc[2] = fmaf(v[2], w0.z, acc[2]); acc[3] = fmaf(v[3], w0.w, acc[3]);
        acc[4] = fmaf(v[4], w1.x, acc[4]); acc[5] = fmaf(v[5], w1.y, acc[5]); acc[6] = fmaf(v[6], w1.z, acc[6]); acc[7] = fmaf(v[7], w1.w, acc[7]);
      }
    }
    FEl<ET>::st8(dx + (((int64_t)in * h + iy) * w + ix) * c + ch, acc);
  }
}
// Row-walking form (the default for k = 3 / 5, stride 1 / 2), the data-gradient sibling of k_f_dw_wgrad_row: a wave owns 2 * GW channels of 64 / GW input rows and
// walks along them with a k x k window of dc in registers and the (flipped) weights of its channels in registers -- per input pixel k new loads, k*k packed
// FMAs, one store.  Stride 2 runs the same walk over the zero-inserted gradient (dcu[2 oy][2 ox] = dc[oy][ox]): odd columns load nothing (the walk is unrolled
// 2k times so that the column parity is a compile-time constant), odd rows contribute zeros.
template <int K, int S, int GW, typename ET>
__global__ __launch_bounds__(256) void k_f_dw_dgrad_row(const FrostFDesc* dp, const ET* __restrict__ dc, int n, int h, int w, int c, int cpad, int ho, int wo,
                                                        ET* __restrict__ dx) {
  constexpr int PAD = (K - 1) / 2, RPW = 64 / GW, UNR = (S == 2) ? 2 * K : K;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cl = (lane % GW) * 2, rsub = lane / GW;
  const int ch = blockIdx.y * (2 * GW) + cl;
  if (ch >= c) return;
  const float* wf = (const float*)dp->pack;
  v2f wt[K][K];                                   // wt[a][b] = w[K-1-a][K-1-b]: out[iy][ix] = sum_ab dcu[iy - PAD + a][ix - PAD + b] * wt[a][b]
#pragma unroll
  for (int a = 0; a < K; ++a)
#pragma unroll
    for (int b = 0; b < K; ++b) wt[a][b] = *(const v2f*)(wf + ((K - 1 - a) * K + (K - 1 - b)) * cpad + ch);
  const int rows = n * h;
  for (int r = (blockIdx.x * 4 + wv) * RPW + rsub; r < rows; r += gridDim.x * 4 * RPW) {
    const int in = r / h, iy = r - in * h;
    const ET* dr[K]; bool rv[K];
#pragma unroll
    for (int a = 0; a < K; ++a) {
      const int ty = iy - PAD + a;                  // row of the (zero-inserted) gradient
      const int oy = (S == 2) ? (ty >> 1) : ty;
      rv[a] = ty >= 0 && oy < ho && (S == 1 || (ty & 1) == 0);
      dr[a] = dc + ((int64_t)(in * ho + (rv[a] ? oy : 0)) * wo) * c + ch;
    }
    ET* orow = dx + (int64_t)r * w * c + ch;
    v2f win[K][K];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
      for (int b = 0; b < K; ++b) win[a][b] = (v2f){0.0f, 0.0f};
#pragma unroll
    for (int b = PAD; b < K - 1; ++b) {             // columns tx = 0 .. PAD - 1 of the first pixel's window (tx < 0: zeros)
      const int tx = b - PAD;
      if (S == 2 && (tx & 1)) continue;
      const int ox = (S == 2) ? (tx >> 1) : tx;
      const bool cv = ox < wo; const int64_t off = (int64_t)(cv ? ox : 0) * c;
#pragma unroll
      for (int a = 0; a < K; ++a) { const v2f v = fdw_ld2(dr[a] + off); win[a][b] = (cv && rv[a]) ? v : (v2f){0.0f, 0.0f}; }
    }
    for (int ix0 = 0; ix0 < w; ix0 += UNR) {
      typename FdwRaw<ET>::T raw[UNR][K];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int tx = ix0 + u + PAD;                // the window column that enters at pixel ix0 + u (parity known at compile time: ix0 is even for S = 2)
        if (S == 2 && ((u + PAD) & 1)) continue;
        const int ox = (S == 2) ? (tx >> 1) : tx;
        const int64_t off = (int64_t)((ox < wo) ? ox : 0) * c;
#pragma unroll
        for (int a = 0; a < K; ++a) raw[u][a] = FdwRaw<ET>::ld(dr[a] + off);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int ix = ix0 + u, tx = ix + PAD, slot = (u + K - 1) % K;
        if (S == 2 && ((u + PAD) & 1)) {
#pragma unroll
          for (int a = 0; a < K; ++a) win[a][slot] = (v2f){0.0f, 0.0f};
        } else {
          const int ox = (S == 2) ? (tx >> 1) : tx;
          const bool cv = ox < wo;
#pragma unroll
          for (int a = 0; a < K; ++a) win[a][slot] = (cv && rv[a]) ? FdwRaw<ET>::cv(raw[u][a]) : (v2f){0.0f, 0.0f};
        }
        v2f acc = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
          for (int b = 0; b < K; ++b) acc = __builtin_elementwise_fma(win[a][(u + b) % K], wt[a][b], acc);
        if (ix < w) FdwOut<ET>::st(orow + (int64_t)ix * c, acc);
      }
    }
  }
}
template <int K, int S, int GW, typename ET>
static void launch_f_dw_dgrad_row2(const FrostFDesc* desc, const ET* dc, int n, int h, int w, int c, int ho, int wo, ET* dx, hipStream_t s) {
  static int wgs = -1;
  if (wgs < 0) { const char* e = getenv("FROST_FDW_DGRAD_WGS"); wgs = e ? atoi(e) : 4096; }
  constexpr int RPW = 64 / GW;
  const int groups = (c + 2 * GW - 1) / (2 * GW); const int rows = n * h;
  int gx = (rows + 4 * RPW - 1) / (4 * RPW); int cap = wgs / groups; if (cap < 1) cap = 1; if (gx > cap) gx = cap;
  hipLaunchKernelGGL((k_f_dw_dgrad_row<K, S, GW, ET>), dim3((unsigned)gx, groups), dim3(256), 0, s, desc, dc, n, h, w, c, round_up(c, 16), ho, wo, dx);
}
template <int K, int S, typename ET>
static void launch_f_dw_dgrad_row(const FrostFDesc* desc, const ET* dc, int n, int h, int w, int c, int ho, int wo, ET* dx, hipStream_t s) {
  if (c > 64) launch_f_dw_dgrad_row2<K, S, 64, ET>(desc, dc, n, h, w, c, ho, wo, dx, s);
  else if (c > 32) launch_f_dw_dgrad_row2<K, S, 32, ET>(desc, dc, n, h, w, c, ho, wo, dx, s);
  else launch_f_dw_dgrad_row2<K, S, 16, ET>(desc, dc, n, h, w, c, ho, wo, dx, s);
}
template <typename ET>
static int float_dw_dgrad_any(const FrostFDesc* desc, const ET* dc, int n, int h, int w, int c, int k, int stride, ET* dx, hipStream_t s) {
  FROST_REQUIRE(c % 8 == 0, "float_dw_dgrad: channels must be a multiple of 8");
  const int pad = (k - 1) / 2; const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
  static int row_form = -1;
  if (row_form < 0) { const char* e = getenv("FROST_FDW_DGRAD_ROW"); row_form = e ? atoi(e) : 1; }
  if (row_form && (k == 3 || k == 5) && (stride == 1 || stride == 2)) {
    if (k == 3 && stride == 1) launch_f_dw_dgrad_row<3, 1, ET>(desc, dc, n, h, w, c, ho, wo, dx, s);
    else if (k == 3) launch_f_dw_dgrad_row<3, 2, ET>(desc, dc, n, h, w, c, ho, wo, dx, s);
    else if (stride == 1) launch_f_dw_dgrad_row<5, 1, ET>(desc, dc, n, h, w, c, ho, wo, dx, s);
    else launch_f_dw_dgrad_row<5, 2, ET>(desc, dc, n, h, w, c, ho, wo, dx, s);
    return frost_check_launch("float_dw_dgrad");
  }
  const int64_t tot = (int64_t)n * h * w * (c >> 3); int64_t grid = (tot + 255) / 256; if (grid > 16384) grid = 16384;
  grid = (grid + 7) & ~(int64_t)7;
  hipLaunchKernelGGL(k_f_dw_dgrad<ET>, dim3((unsigned)grid), dim3(256), 0, s, desc, dc, n, h, w, c, round_up(c, 16), k, stride, ho, wo, dx);
  return frost_check_launch("float_dw_dgrad");
}
extern "C" int frost_float_dw_dgrad(const FrostFDesc* desc, const uint16_t* dc, int n, int h, int w, int c, int k, int stride, uint16_t* dx, void* stream) {
  return float_dw_dgrad_any<uint16_t>(desc, dc, n, h, w, c, k, stride, dx, as_stream(stream));
}
extern "C" int frost_float_dw_dgrad_f32(const FrostFDesc* desc, const float* dc, int n, int h, int w, int c, int k, int stride, float* dx, void* stream) {
  return float_dw_dgrad_any<float>(desc, dc, n, h, w, c, k, stride, dx, as_stream(stream));
}

// depthwise weight gradient: dW[c][ky][kx] += sum_p dc[p][c] * x[n][oy*s-pad+ky][ox*s-pad+kx][c]
// workgroup = (32-channel group, kernel row ky) x a strided set of output pixels: thread = 8 channels (4 adjacent threads read one 64-byte
// segment) x one of 64 pixel lanes, k*8 accumulators; lanes are summed through LDS and leave with one atomic per (channel, tap).
template <typename ET>
__global__ __launch_bounds__(256) void k_f_dw_wgrad(const ET* __restrict__ dc, const ET* __restrict__ x, int n, int h, int w, int c, int k, int stride,
                                                    int ho, int wo, float* __restrict__ dw) {
  __shared__ float red[32 * 5];
  const int tid = threadIdx.x, c8l = tid & 3, pl = tid >> 2;
  const int unit = blockIdx.y; const int ky = unit % k, cg = unit / k;
  const int ch = cg * 32 + c8l * 8; const bool live = ch < c;
  const int pad = (k - 1) / 2;
  const int64_t npix = (int64_t)n * ho * wo;
  float a[5][8];
#pragma unroll
  for (int kx = 0; kx < 5; ++kx)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[kx][e] = 0.0f;
  if (live) {
    const int64_t ppx = (npix + 7) >> 3; const int64_t p_lo = (blockIdx.x & 7) * ppx; const int64_t p_hi = (p_lo + ppx < npix) ? p_lo + ppx : npix;   // XCD-aware, see k_f_dw
    for (int64_t p = p_lo + (int64_t)(blockIdx.x >> 3) * 64 + pl; p < p_hi; p += (int64_t)(gridDim.x >> 3) * 64) {
      int64_t pp = p; const int ox = (int)(pp % wo); pp /= wo; const int oy = (int)(pp % ho); const int in = (int)(pp / ho);
      const int iy = oy * stride - pad + ky; if (iy < 0 || iy >= h) continue;
      float g8[8];
      FEl<ET>::ld8(dc + p * c + ch, g8);
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        if (kx >= k) break;
        const int ix = ox * stride - pad + kx; if (ix < 0 || ix >= w) continue;
        float v[8];
        FEl<ET>::ld8(x + (((int64_t)in * h + iy) * w + ix) * c + ch, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[kx][e] = fmaf(g8[e], v[e], a[kx][e]);
      }
    }
  }
  for (int i = tid; i < 32 * 5; i += 256) red[i] = 0.0f;
  __syncthreads();
  // lanes tid, tid^4, ... share (c8l): fold the 16 pixel lanes of a wave first (lane bits 2..5), then LDS atomics across the 4 waves
#pragma unroll
  for (int kx = 0; kx < 5; ++kx)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = a[kx][e];
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if ((tid & 63) < 4 && kx < k) atomicAdd(&red[(c8l * 8 + e) * 5 + kx], v);
    }
  __syncthreads();
  for (int i = tid; i < 32 * k; i += 256) {
    const int cl = i / k, kx = i % k; const int cc = cg * 32 + cl;
    if (cc < c) atomicAdd(dw + (int64_t)cc * k * k + ky * k + kx, red[cl * 5 + kx]);
  }
}
// Row-walking form (the default for k = 3 / 5): a wave owns 128 channels (lane = 2 adjacent channels: one 256-byte segment per pixel and wave) and walks
// along output rows with the k x k input window in registers -- per output pixel it loads the S new window columns (k loads of 4 bytes) and one dc value and
// runs k*k packed FMAs, so dc and x are read ONCE (the form above reads both once per kernel row, in 64-byte segments).  The window's column slots rotate
// with the output column ((ox*S + kx) mod k; the ox loop is unrolled k times so that every slot index is a compile-time constant: no register moves).
template <int K, int S, int GW, int SRC, typename ET>
__global__ __launch_bounds__(256) void k_f_dw_wgrad_row(const ET* __restrict__ dc, const ET* __restrict__ x, int n, int h, int w, int c, int ho, int wo,
                                                        float* __restrict__ dw, const FrostFDesc* dsrc, int relu_src) {
  // GW lanes (2 * GW channels) per output row, 64 / GW rows per wave: narrow layers (the biggest maps) keep every lane busy
  constexpr int PAD = (K - 1) / 2, KK = K * K, RPW = 64 / GW;
  __shared__ float red[2 * GW * KK];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cl = (lane % GW) * 2, rsub = lane / GW;
  const int ch = blockIdx.y * (2 * GW) + cl; const bool live = ch < c;
  v2f acc[K][K];
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) acc[ky][kx] = (v2f){0.0f, 0.0f};
  v2f ssc = (v2f){1.0f, 1.0f}, sbi = (v2f){0.0f, 0.0f}; float slo = -INFINITY; constexpr bool shs = SRC == 2;          // SRC: x is the kept conv output of the layer in front (see k_f_dw_row<SRC>)
  if (SRC && live) { const int scp = dsrc->cpad; ssc = *(const v2f*)(dsrc->coef + FC_SCALE * scp + ch); sbi = *(const v2f*)(dsrc->coef + FC_BIAS * scp + ch); slo = relu_src ? 0.0f : -INFINITY; }
  auto xf = [&](v2f v) __attribute__((always_inline)) { return SRC ? (v2f){f_act(fmaf(v[0], ssc[0], sbi[0]), slo, shs), f_act(fmaf(v[1], ssc[1], sbi[1]), slo, shs)} : v; };
  const int rows = n * ho;
  if (live) {
    for (int r = (blockIdx.x * 4 + wv) * RPW + rsub; r < rows; r += gridDim.x * 4 * RPW) {
      const int in = r / ho, oy = r - in * ho;
      const ET* xr[K]; bool rv[K];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S - PAD + ky; rv[ky] = (unsigned)iy < (unsigned)h;
        xr[ky] = x + ((int64_t)(in * h + (rv[ky] ? iy : 0)) * w) * c + ch;
      }
      const ET* dr = dc + (int64_t)r * wo * c + ch;
      v2f win[K][K];
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) win[ky][kx] = (v2f){0.0f, 0.0f};
#pragma unroll
      for (int kx = 0; kx < K - S; ++kx) {              // the columns of pixel 0 that no later step loads
        const int ix = kx - PAD;                        // (every load unconditional from a clamped address, zeroed afterwards: a conditional load ends in a wait at its join)
        const bool cv = (unsigned)ix < (unsigned)w; const int64_t off = (int64_t)(cv ? ix : 0) * c;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) { const v2f v = xf(fdw_ld2(xr[ky] + off)); win[ky][kx] = (cv && rv[ky]) ? v : (v2f){0.0f, 0.0f}; }
      }
      for (int ox0 = 0; ox0 < wo; ox0 += K) {
        // phase 1: every load of the next K output pixels (no branch on ox: past the row end dc counts as zero), phase 2: the arithmetic
        typename FdwRaw<ET>::T raw[K][S][K], graw[K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
          const int ox = ox0 + u;
#pragma unroll
          for (int j = 0; j < S; ++j) {
            const int ix = ox * S - PAD + (K - S + j);
            const int64_t off = (int64_t)(((unsigned)ix < (unsigned)w) ? ix : 0) * c;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) raw[u][j][ky] = FdwRaw<ET>::ld(xr[ky] + off);
          }
          graw[u] = FdwRaw<ET>::ld(dr + (int64_t)(ox < wo ? ox : wo - 1) * c);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < K; ++u) {
          const int ox = ox0 + u;
#pragma unroll
          for (int j = 0; j < S; ++j) {
            const int kx = K - S + j, ix = ox * S - PAD + kx, slot = (u * S + kx) % K;
            const bool cv = (unsigned)ix < (unsigned)w;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) win[ky][slot] = (cv && rv[ky]) ? xf(FdwRaw<ET>::cv(raw[u][j][ky])) : (v2f){0.0f, 0.0f};
          }
          const v2f g = (ox < wo) ? FdwRaw<ET>::cv(graw[u]) : (v2f){0.0f, 0.0f};
#pragma unroll
          for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[ky][kx] = __builtin_elementwise_fma(g, win[ky][(u * S + kx) % K], acc[ky][kx]);
        }
      }
    }
  }
  for (int i = tid; i < 2 * GW * KK; i += 256) red[i] = 0.0f;
  __syncthreads();
  if (live) {
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) { atomicAdd(&red[cl * KK + ky * K + kx], acc[ky][kx][0]); atomicAdd(&red[(cl + 1) * KK + ky * K + kx], acc[ky][kx][1]); }
  }
  __syncthreads();
  for (int i = tid; i < 2 * GW * KK; i += 256) {
    const int cc = blockIdx.y * (2 * GW) + i / KK;
    if (cc < c && red[i] != 0.0f) atomicAdd(dw + (int64_t)cc * KK + (i % KK), red[i]);
  }
}
template <int K, int S, int GW, typename ET>
static void launch_f_dw_wgrad_row2(const ET* dc, const ET* x, int n, int h, int w, int c, int ho, int wo, float* dw, hipStream_t s, const FrostFDesc* dsrc, int relu_src) {
  static int wgs = -1;
  if (wgs < 0) { const char* e = getenv("FROST_FDW_WGRAD_WGS"); wgs = e ? atoi(e) : (K == 3 ? 2048 : 1024); }
  constexpr int RPW = 64 / GW;
  const int groups = (c + 2 * GW - 1) / (2 * GW); const int rows = n * ho;
  int gx = (rows + 4 * RPW - 1) / (4 * RPW); int cap = wgs / groups; if (cap < 1) cap = 1; if (gx > cap) gx = cap;
  if (dsrc && relu_src == F_ACT_HSWISH) hipLaunchKernelGGL((k_f_dw_wgrad_row<K, S, GW, 2, ET>), dim3((unsigned)gx, groups), dim3(256), 0, s, dc, x, n, h, w, c, ho, wo, dw, dsrc, relu_src);
  else if (dsrc) hipLaunchKernelGGL((k_f_dw_wgrad_row<K, S, GW, 1, ET>), dim3((unsigned)gx, groups), dim3(256), 0, s, dc, x, n, h, w, c, ho, wo, dw, dsrc, relu_src);
  else hipLaunchKernelGGL((k_f_dw_wgrad_row<K, S, GW, 0, ET>), dim3((unsigned)gx, groups), dim3(256), 0, s, dc, x, n, h, w, c, ho, wo, dw, dsrc, relu_src);
}
template <int K, int S, typename ET>
static void launch_f_dw_wgrad_row(const ET* dc, const ET* x, int n, int h, int w, int c, int ho, int wo, float* dw, hipStream_t s, const FrostFDesc* dsrc, int relu_src) {
  // lanes per row: 64 (256-byte segments) unless the layer is so narrow that most of them would idle -- measured: 96 channels on 64 lanes (25 % idle) beat
  // three 16-lane groups of 64-byte segments (547 vs 717 us, layer1.1), 32 channels on 16 lanes beat 64 (202 vs 462 us, layer1.0)
  if (c > 64) launch_f_dw_wgrad_row2<K, S, 64, ET>(dc, x, n, h, w, c, ho, wo, dw, s, dsrc, relu_src);
  else if (c > 32) launch_f_dw_wgrad_row2<K, S, 32, ET>(dc, x, n, h, w, c, ho, wo, dw, s, dsrc, relu_src);
  else launch_f_dw_wgrad_row2<K, S, 16, ET>(dc, x, n, h, w, c, ho, wo, dw, s, dsrc, relu_src);
}
template <typename ET>
static int float_dw_wgrad_any(const ET* dc, const ET* x, int n, int h, int w, int c, int k, int stride, float* dw, hipStream_t s, const FrostFDesc* dsrc = nullptr, int relu_src = 0) {
  FROST_REQUIRE(c % 8 == 0 && k <= 5, "float_dw_wgrad: channels must be a multiple of 8, k <= 5");
  const int pad = (k - 1) / 2; const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
  static int row_form = -1;
  if (row_form < 0) { const char* e = getenv("FROST_FDW_WGRAD_ROW"); row_form = e ? atoi(e) : 1; }
  FROST_REQUIRE(!dsrc || ((k == 3 || k == 5) && (stride == 1 || stride == 2)), "float_dw_wgrad_src: 3x3 / 5x5, stride 1 / 2");
  if ((row_form || dsrc) && (k == 3 || k == 5) && (stride == 1 || stride == 2)) {
    if (k == 3 && stride == 1) launch_f_dw_wgrad_row<3, 1, ET>(dc, x, n, h, w, c, ho, wo, dw, s, dsrc, relu_src);
    else if (k == 3) launch_f_dw_wgrad_row<3, 2, ET>(dc, x, n, h, w, c, ho, wo, dw, s, dsrc, relu_src);
    else if (stride == 1) launch_f_dw_wgrad_row<5, 1, ET>(dc, x, n, h, w, c, ho, wo, dw, s, dsrc, relu_src);
    else launch_f_dw_wgrad_row<5, 2, ET>(dc, x, n, h, w, c, ho, wo, dw, s, dsrc, relu_src);
    return frost_check_launch("float_dw_wgrad");
  }
  const int64_t npix = (int64_t)n * ho * wo;
  const int units = ((c + 31) / 32) * k;
  int64_t gx = (npix + 63) / 64; int64_t cap = 4096 / units; if (cap < 1) cap = 1; if (gx > cap) gx = cap;
  gx = (gx + 7) & ~(int64_t)7;
  hipLaunchKernelGGL(k_f_dw_wgrad<ET>, dim3((unsigned)gx, units), dim3(256), 0, s, dc, x, n, h, w, c, k, stride, ho, wo, dw);
  return frost_check_launch("float_dw_wgrad");
}
extern "C" int frost_float_dw_wgrad(const uint16_t* dc, const uint16_t* x, int n, int h, int w, int c, int k, int stride, float* dw, void* stream) {
  return float_dw_wgrad_any<uint16_t>(dc, x, n, h, w, c, k, stride, dw, as_stream(stream));
}
extern "C" int frost_float_dw_wgrad_f32(const float* dc, const float* x, int n, int h, int w, int c, int k, int stride, float* dw, void* stream) {
  return float_dw_wgrad_any<float>(dc, x, n, h, w, c, k, stride, dw, as_stream(stream));
}
// weight gradient with the layer's input given as the kept conv output of the layer in front (see frost_float_dw_src)
extern "C" int frost_float_dw_wgrad_src(const uint16_t* dc, const uint16_t* conv_src, const FrostFDesc* desc_src, int relu_src, int n, int h, int w, int c, int k, int stride,
                                        float* dw, void* stream) {
  FROST_REQUIRE(desc_src && conv_src, "float_dw_wgrad_src: descriptor and kept conv output of the layer in front");
  return float_dw_wgrad_any<uint16_t>(dc, conv_src, n, h, w, c, k, stride, dw, as_stream(stream), desc_src, relu_src);
}
extern "C" int frost_float_dw_wgrad_src_f32(const float* dc, const float* conv_src, const FrostFDesc* desc_src, int relu_src, int n, int h, int w, int c, int k, int stride,
                                            float* dw, void* stream) {
  FROST_REQUIRE(desc_src && conv_src, "float_dw_wgrad_src: descriptor and kept conv output of the layer in front");
  return float_dw_wgrad_any<float>(dc, conv_src, n, h, w, c, k, stride, dw, as_stream(stream), desc_src, relu_src);
}

// ------------------------------------------------------------------------------------------------ pointwise weight gradient (bf16 MFMA, K = pixels)
// dW[co][ci] += sum_p dc[p][co] * x[p][ci].  Both operands pixel-major bf16: 128-pixel blocks staged in their natural layout, the
// K(pixel)-contiguous fragments come from the gfx950 LDS transpose read (ds_read_b64_tr_b16).  64x64 output tile per workgroup, the 4
// waves split the staged pixels, pixel range split across workgroups, fp32 atomics at the end.
#define FW_T 64
#define FW_KP 128
#define FW_RS 136
__global__ __launch_bounds__(256, 2) void k_f_pw_wgrad(const uint16_t* __restrict__ dc, const uint16_t* __restrict__ x, int64_t npix, int cin, int ldx, int cout,
                                                       float* __restrict__ dw, int ldw, int nsplit) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * FW_KP * FW_RS];
  uint8_t* dcs = lds; uint8_t* xs = lds + FW_KP * FW_RS;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nci = (cin + FW_T - 1) / FW_T;
  const int ntile = ((cout + FW_T - 1) / FW_T) * nci;
  const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
  const int co0 = (tile / nci) * FW_T, ci0 = (tile % nci) * FW_T;
  v4f acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int pb = w * 32 + g * 8;
  const uint8_t* a_src = dcs + (pb + (i16 >> 2)) * FW_RS + (i16 & 3) * 8;
  const uint8_t* b_src = xs + (pb + (i16 >> 2)) * FW_RS + (i16 & 3) * 8;
  int na = (cout - co0 + 15) / 16; if (na > 4) na = 4;
  int nb = (cin - ci0 + 15) / 16; if (nb > 4) nb = 4;
  const int64_t nblk = (npix + FW_KP - 1) / FW_KP;
  uint4 pd[4], px[4];         // register prefetch of the next 128-pixel block: its HBM latency overlaps the MFMAs of the current one
  auto fetch = [&](int64_t blk) __attribute__((always_inline)) {
    const int64_t q0 = blk * FW_KP;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u = tid + jn * 256; const int pix = u >> 3, c8 = u & 7; const int64_t gp = q0 + pix;
      pd[jn] = make_uint4(0, 0, 0, 0); px[jn] = make_uint4(0, 0, 0, 0);
      if (gp < npix && (co0 + c8 * 8) < cout) pd[jn] = *(const uint4*)(dc + gp * cout + co0 + c8 * 8);
      if (gp < npix && (ci0 + c8 * 8) < cin) px[jn] = *(const uint4*)(x + gp * ldx + ci0 + c8 * 8);
    }
  };
  if (split < nblk) fetch(split);
  for (int64_t blk = split; blk < nblk; blk += nsplit) {
    __syncthreads();
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int u = tid + jn * 256; const int pix = u >> 3, c8 = u & 7;
      *(uint2*)(dcs + pix * FW_RS + c8 * 16) = make_uint2(pd[jn].x, pd[jn].y); *(uint2*)(dcs + pix * FW_RS + c8 * 16 + 8) = make_uint2(pd[jn].z, pd[jn].w);
      *(uint2*)(xs + pix * FW_RS + c8 * 16) = make_uint2(px[jn].x, px[jn].y); *(uint2*)(xs + pix * FW_RS + c8 * 16 + 8) = make_uint2(px[jn].z, px[jn].w);
    }
    __syncthreads();
    if (blk + nsplit < nblk) fetch(blk + nsplit);
    v4i afr[4], bfr[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a < na) {
        const v2i32 l2 = __builtin_bit_cast(v2i32, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(a_src + a * 32)));
        const v2i32 h2 = __builtin_bit_cast(v2i32, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(a_src + a * 32 + 4 * FW_RS)));
        afr[a] = (v4i){l2[0], l2[1], h2[0], h2[1]};
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nb) {
        const v2i32 l2 = __builtin_bit_cast(v2i32, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(b_src + b * 32)));
        const v2i32 h2 = __builtin_bit_cast(v2i32, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(b_src + b * 32 + 4 * FW_RS)));
        bfr[b] = (v4i){l2[0], l2[1], h2[0], h2[1]};
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (a < na && b < nb)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf16, afr[a]), __builtin_bit_cast(v8bf16, bfr[b]), acc[a][b], 0, 0, 0);
  }
  __syncthreads();
  float* red = (float*)lds;
  for (int i = tid; i < FW_T * FW_T; i += 256) red[i] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (a < na && b < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&red[(a * 16 + 4 * g + r) * FW_T + b * 16 + i16], acc[a][b][r]);
      }
  __syncthreads();
  for (int i = tid; i < FW_T * FW_T; i += 256) {
    const int co = co0 + i / FW_T, ci = ci0 + i % FW_T;
    if (co < cout && ci < cin) atomicAdd(dw + (int64_t)co * ldw + ci, red[i]);
  }
}
// x: bf16 [npix][ldx] (first cin columns used); dw: fp32 [cout][ldw], accumulated into (must hold the running gradient or zeros)
extern "C" int frost_float_pw_wgrad(const uint16_t* dc, const uint16_t* x, int64_t npix, int cin, int ldx, int cout, float* dw, int ldw, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 8 == 0 && ldx % 8 == 0, "float_pw_wgrad: channels must be multiples of 8");
  const int64_t nblk = (npix + FW_KP - 1) / FW_KP;
  const int ntile = ((cout + FW_T - 1) / FW_T) * ((cin + FW_T - 1) / FW_T);
  // pixel split: every workgroup ends with 4096 fp32 atomics into dW, so a split must own enough 128-pixel blocks to pay for them
  static int target = -1, minblk = 32;
  if (target < 0) {
    const char* e = getenv("FROST_FWG_TARGET"); target = e ? atoi(e) : 1024; if (target < 64) target = 64;
    e = getenv("FROST_FWG_MINBLK"); if (e) minblk = atoi(e); if (minblk < 1) minblk = 1;
  }
  int nsplit = (target + ntile - 1) / ntile;
  if (nsplit > nblk / minblk) nsplit = (int)(nblk / minblk); if (nsplit < 1) nsplit = 1;
  hipLaunchKernelGGL(k_f_pw_wgrad, dim3(ntile * nsplit), dim3(256), 0, as_stream(stream), dc, x, npix, cin, ldx, cout, dw, ldw, nsplit);
  return frost_check_launch("float_pw_wgrad");
}
// fp32 storage: the same 64x64 output tile per workgroup on the fp32 MFMA (16x16x4).  64-pixel blocks staged in their natural layout
// ([pixel][64 channels] floats); wave w contracts pixels 16w .. 16w+15 of the block: lane (i, g) supplies, for instruction e, pixel
// 16w + 4g + e of channel i in both operands (the contraction order is free as long as both operands agree).
#define FW32_KP 64
#define FW32_RS 272
__global__ __launch_bounds__(256, 2) void k_f_pw_wgrad_f32(const float* __restrict__ dc, const float* __restrict__ x, int64_t npix, int cin, int ldx, int cout,
                                                           float* __restrict__ dw, int ldw, int nsplit) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * FW32_KP * FW32_RS];
  uint8_t* dcs = lds; uint8_t* xs = lds + FW32_KP * FW32_RS;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nci = (cin + FW_T - 1) / FW_T;
  const int ntile = ((cout + FW_T - 1) / FW_T) * nci;
  const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
  const int co0 = (tile / nci) * FW_T, ci0 = (tile % nci) * FW_T;
  v4f acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
  int na = (cout - co0 + 15) / 16; if (na > 4) na = 4;
  int nb = (cin - ci0 + 15) / 16; if (nb > 4) nb = 4;
  const int64_t nblk = (npix + FW32_KP - 1) / FW32_KP;
  for (int64_t blk = split; blk < nblk; blk += nsplit) {
    const int64_t q0 = blk * FW32_KP;
    __syncthreads();
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {          // 64 pixels x 16 four-channel units per operand
      const int u = tid + jn * 256; const int pix = u >> 4, c4 = u & 15; const int64_t gp = q0 + pix;
      float4 vd = make_float4(0.f, 0.f, 0.f, 0.f), vx = vd;
      if (gp < npix && (co0 + c4 * 4) < cout) vd = *(const float4*)(dc + gp * cout + co0 + c4 * 4);
      if (gp < npix && (ci0 + c4 * 4) < cin) vx = *(const float4*)(x + gp * ldx + ci0 + c4 * 4);
      *(float4*)(dcs + pix * FW32_RS + c4 * 16) = vd; *(float4*)(xs + pix * FW32_RS + c4 * 16) = vx;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pix = w * 16 + g * 4 + e;
      float afr[4], bfr[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) afr[a] = (a < na) ? *(const float*)(dcs + pix * FW32_RS + (a * 16 + i16) * 4) : 0.0f;
#pragma unroll
      for (int b = 0; b < 4; ++b) bfr[b] = (b < nb) ? *(const float*)(xs + pix * FW32_RS + (b * 16 + i16) * 4) : 0.0f;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (a < na && b < nb) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  }
  __syncthreads();
  float* red = (float*)lds;
  for (int i = tid; i < FW_T * FW_T; i += 256) red[i] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (a < na && b < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&red[(a * 16 + 4 * g + r) * FW_T + b * 16 + i16], acc[a][b][r]);
      }
  __syncthreads();
  for (int i = tid; i < FW_T * FW_T; i += 256) {
    const int co = co0 + i / FW_T, ci = ci0 + i % FW_T;
    if (co < cout && ci < cin) atomicAdd(dw + (int64_t)co * ldw + ci, red[i]);
  }
}
extern "C" int frost_float_pw_wgrad_f32(const float* dc, const float* x, int64_t npix, int cin, int ldx, int cout, float* dw, int ldw, void* stream) {
  FROST_REQUIRE(cin % 4 == 0 && cout % 4 == 0 && ldx % 4 == 0, "float_pw_wgrad_f32: channels must be multiples of 4");
  const int64_t nblk = (npix + FW32_KP - 1) / FW32_KP;
  const int ntile = ((cout + FW_T - 1) / FW_T) * ((cin + FW_T - 1) / FW_T);
  int nsplit = (1024 + ntile - 1) / ntile;
  if (nsplit > nblk / 4) nsplit = (int)(nblk / 4); if (nsplit < 1) nsplit = 1;
  hipLaunchKernelGGL(k_f_pw_wgrad_f32, dim3(ntile * nsplit), dim3(256), 0, as_stream(stream), dc, x, npix, cin, ldx, cout, dw, ldw, nsplit);
  return frost_check_launch("float_pw_wgrad_f32");
}
// stem: the im2col'd weight gradient [cout][64] (K = tap*4 + c) back to the OIHW parameter layout [cout][3][9]
__global__ __launch_bounds__(256) void k_f_stem_wscatter(const float* __restrict__ tmp, int cout, float* __restrict__ dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cout * 27) return;
  const int co = i / 27, r = i % 27, c = r / 9, tap = r % 9;
  dw[i] += tmp[co * 64 + tap * 4 + c];
}
extern "C" int frost_float_stem_wscatter(const float* tmp, int cout, float* dw, void* stream) {
  hipLaunchKernelGGL(k_f_stem_wscatter, dim3((cout * 27 + 255) / 256), dim3(256), 0, as_stream(stream), tmp, cout, dw);
  return frost_check_launch("float_stem_wscatter");
}

// ------------------------------------------------------------------------------------------------ block wiring, backward
// gradient of a bottleneck's input = [residual branch: the block output's gradient] + [cat branch: columns cs.. of the cat gradient]
// + [squeeze conv's data gradient]; any of the three may be absent (NULL).
template <typename ET>
__global__ __launch_bounds__(256) void k_f_grad_merge(const ET* __restrict__ res, const ET* __restrict__ cat, int cs, int ccat,
                                                      const ET* __restrict__ sq, int64_t npix, int c, ET* __restrict__ out) {
  const int cu = c >> 3; const int64_t tot = npix * cu;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int u = (int)(i % cu); const int64_t p = i / cu; const int ch = u * 8;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](const ET* src) {
      float v[8];
      FEl<ET>::ld8(src, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[e];
    };
    if (res) add(res + p * c + ch);
    if (cat) add(cat + p * ccat + cs + ch);
    if (sq) add(sq + p * c + ch);
    FEl<ET>::st8(out + p * c + ch, a);
  }
}
template <typename ET>
static int float_grad_merge_any(const ET* res, const ET* cat, int cs, int ccat, const ET* sq, int64_t npix, int c, ET* out, hipStream_t s) {
  FROST_REQUIRE(c % 8 == 0 && cs % 8 == 0 && ccat % 8 == 0, "float_grad_merge: channel counts must be multiples of 8");
  const int64_t tot = npix * (c >> 3); int64_t grid = (tot + 255) / 256; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_f_grad_merge<ET>, dim3((unsigned)grid), dim3(256), 0, s, res, cat, cs, ccat, sq, npix, c, out);
  return frost_check_launch("float_grad_merge");
}
extern "C" int frost_float_grad_merge(const uint16_t* res, const uint16_t* cat, int cs, int ccat, const uint16_t* sq, int64_t npix, int c, uint16_t* out,
                                      void* stream) {
  return float_grad_merge_any<uint16_t>(res, cat, cs, ccat, sq, npix, c, out, as_stream(stream));
}
extern "C" int frost_float_grad_merge_f32(const float* res, const float* cat, int cs, int ccat, const float* sq, int64_t npix, int c, float* out, void* stream) {
  return float_grad_merge_any<float>(res, cat, cs, ccat, sq, npix, c, out, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------ head
// global average pool with the dropout mask applied (mask holds 0 or 1/keep; NULL = no dropout): y fp32 [n][c]
template <typename ET>
__global__ __launch_bounds__(256) void k_f_avgpool(const ET* __restrict__ x, int n, int hw, int c, const float* __restrict__ drop, float* __restrict__ y) {
  const int64_t tot = (int64_t)n * c;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c); const int in = (int)(i / c);
    float s = 0.0f;
    for (int p = 0; p < hw; ++p) s += FEl<ET>::ld1(x + ((int64_t)in * hw + p) * c + ch);
    s /= (float)hw;
    y[i] = drop ? s * drop[i] : s;
  }
}
extern "C" int frost_float_avgpool(const uint16_t* x, int n, int hw, int c, const float* drop, float* y, void* stream) {
  const int64_t tot = (int64_t)n * c; int64_t grid = (tot + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_f_avgpool<uint16_t>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, n, hw, c, drop, y);
  return frost_check_launch("float_avgpool");
}
extern "C" int frost_float_avgpool_f32(const float* x, int n, int hw, int c, const float* drop, float* y, void* stream) {
  const int64_t tot = (int64_t)n * c; int64_t grid = (tot + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_f_avgpool<float>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, n, hw, c, drop, y);
  return frost_check_launch("float_avgpool_f32");
}

// ------------------------------------------------------------------------------------------------ fp32 mode: block wiring and the stem's im2col
// (the bf16 mode uses frost_infer_cat / frost_infer_add / frost_infer_stem_im2col of the inference path)
__global__ __launch_bounds__(256) void k_f32_cat(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb, int64_t npix, float* __restrict__ y) {
  const int cu = (ca + cb) >> 2; const int64_t tot = npix * cu;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int u = (int)(i % cu); const int64_t p = i / cu; const int ch = u * 4;
    *(float4*)(y + p * (ca + cb) + ch) = (ch < ca) ? *(const float4*)(a + p * ca + ch) : *(const float4*)(b + p * cb + (ch - ca));
  }
}
extern "C" int frost_float_cat_f32(const float* a, int ca, const float* b, int cb, int64_t npix, float* y, void* stream) {
  FROST_REQUIRE(ca % 4 == 0 && cb % 4 == 0, "float_cat_f32: channel counts must be multiples of 4");
  const int64_t tot = npix * ((ca + cb) >> 2); int64_t grid = (tot + 255) / 256; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_f32_cat, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a, ca, b, cb, npix, y);
  return frost_check_launch("float_cat_f32");
}
__global__ __launch_bounds__(256) void k_f32_add(const float* __restrict__ a, const float* __restrict__ b, int64_t n4, float* __restrict__ y) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 u = ((const float4*)a)[i], v = ((const float4*)b)[i];
    ((float4*)y)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}
extern "C" int frost_float_add_f32(const float* a, const float* b, int64_t n, float* y, void* stream) {
  FROST_REQUIRE(n % 4 == 0, "float_add_f32: n must be a multiple of 4");
  int64_t grid = ((n >> 2) + 255) / 256; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_f32_add, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a, b, n >> 2, y);
  return frost_check_launch("float_add_f32");
}
// 3x3 stride-2 pad-1 patches of the logical (N,3,H,W) fp32 image -> [npix_out][64] fp32, K index = tap*4 + c (c == 3 and k >= 36: 0)
__global__ __launch_bounds__(256) void k_f32_stem_im2col(const float* __restrict__ x, int n, int h, int w, int ho, int wo, int64_t sn, int64_t sc,
                                                         int64_t sh, int64_t sw, float* __restrict__ out) {
  const int64_t tot = (int64_t)n * ho * wo * 16;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int tap = (int)(i & 15); int64_t p = i >> 4; const int ox = (int)(p % wo); p /= wo; const int oy = (int)(p % ho); const int in = (int)(p / ho);
    float v[3] = {0.f, 0.f, 0.f};
    if (tap < 9) {
      const int iy = oy * 2 - 1 + tap / 3, ix = ox * 2 - 1 + tap % 3;
      if (iy >= 0 && iy < h && ix >= 0 && ix < w)
        for (int c = 0; c < 3; ++c) v[c] = x[in * sn + c * sc + iy * sh + ix * sw];
    }
    *(float4*)(out + (i << 2)) = make_float4(v[0], v[1], v[2], 0.0f);
  }
}
extern "C" int frost_float_stem_im2col_f32(const float* x, int n, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* out, void* stream) {
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t tot = (int64_t)n * ho * wo * 16; int64_t grid = (tot + 255) / 256; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_f32_stem_im2col, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, n, h, w, ho, wo, sn, sc, sh, sw, out);
  return frost_check_launch("float_stem_im2col_f32");
}
