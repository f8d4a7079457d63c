// Element-wise, observer, weight-prep, finalize, cat/add and head kernels (HBM-bound byte/float work).
#include "frost_common.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[256] = "";
extern "C" void frost_set_error(const char* msg) { strncpy(g_err, msg, sizeof(g_err) - 1); }
extern "C" const char* frost_last_error(void) { return g_err; }
extern "C" int frost_abi_version(void) { return FROST_ABI_VERSION; }
extern "C" int frost_ticket_words(void) { return FROST_TICKET_WORDS; }
extern "C" int frost_fin_desc_bytes(void) { return (int)sizeof(FrostFinDesc); }
int frost_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -(int)e;
  }
  return 0;
}

static inline int grid_for(int64_t n, int per_block, int cap = 4096) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

// ------------------------------------------------------------------------------------------------ min/max
__global__ void k_fill_minmax(float* out2, int count) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) { out2[2 * i] = INFINITY; out2[2 * i + 1] = -INFINITY; }
}
extern "C" int frost_fill_minmax(float* out2, int count, void* stream) {
  hipLaunchKernelGGL(k_fill_minmax, dim3((count + 255) / 256), dim3(256), 0, as_stream(stream), out2, count);
  return frost_check_launch("fill_minmax");
}

__device__ __forceinline__ void block_minmax_commit(float lo, float hi, float* out2) {
  __shared__ float slo[4], shi[4];
  lo = wave_min(lo); hi = wave_max(hi);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { slo[w] = lo; shi[w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) { lo = fminf(lo, slo[i]); hi = fmaxf(hi, shi[i]); }
    // hundreds of workgroups hitting ONE address serialise in the L2 (~45 ns each: 46 us for the 500 workgroups of the logits' range): look first --
    // the running extremes only move one way, so a workgroup whose candidate cannot win skips its atomic (a stale look only costs a redundant atomic)
    if (lo < __hip_atomic_load(out2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_min_f32(out2, lo);
    if (hi > __hip_atomic_load(out2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_f32(out2 + 1, hi);
  }
}

__global__ __launch_bounds__(256) void k_minmax_f32(const float* __restrict__ x, int64_t n, float* out2) {
  float lo = INFINITY, hi = -INFINITY;
  int64_t n4 = n >> 2;
  const float4* x4 = (const float4*)x;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 v = x4[i];
    lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
    hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = x[(n4 << 2) + threadIdx.x]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  block_minmax_commit(lo, hi, out2);
}
extern "C" int frost_minmax_f32(const float* x, int64_t n, float* out2, void* stream) {
  FROST_REQUIRE(((uintptr_t)x & 15) == 0, "minmax: x must be 16B aligned");
  hipLaunchKernelGGL(k_minmax_f32, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, as_stream(stream), x, n, out2);
  return frost_check_launch("minmax_f32");
}

__global__ void k_observer_update(float* q, const float* cur2, int symmetric, int rule127, int observe) {
  if (threadIdx.x == 0) observer_update_dev(q, cur2[0], cur2[1], symmetric, rule127, observe);
}
extern "C" int frost_observer_update(float* qrec, const float* cur2, int symmetric, int rule127, int observe,
                                     void* stream) {
  hipLaunchKernelGGL(k_observer_update, dim3(1), dim3(64), 0, as_stream(stream), qrec, cur2, symmetric, rule127, observe);
  return frost_check_launch("observer_update");
}

// ------------------------------------------------------------------------------------------------ input quant
__global__ __launch_bounds__(256) void k_quantize_input(const float* __restrict__ x, int n, int c, int h, int w,
                                                        int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                                        const float* __restrict__ qrec, int8_t* __restrict__ out, int cpad) {
  QP q = load_qp(qrec);
  int64_t npix = (int64_t)n * h * w;
  for (int64_t p = blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    int xw = (int)(p % w); int64_t t = p / w; int yh = (int)(t % h); int in = (int)(t / h);
    const float* px = x + in * sn + yh * sh + xw * sw;
    uint32_t packed = 0;
    for (int ch = 0; ch < cpad; ++ch) {
      int idx = q.zp;
      if (ch < c) idx = fq_index(px[ch * sc], q.inv, q.zp, 0, q.hi);
      packed |= ((uint32_t)((idx - 128) & 255)) << (8 * (ch & 3));
      if ((ch & 3) == 3) { *(uint32_t*)(out + p * cpad + ch - 3) = packed; packed = 0; }
    }
  }
}
// channels_last RGB input (the reference's loaders after .to(memory_format=channels_last); bench.py): the tensor is one dense [pixel][3] fp32 array -- four pixels =
// three 16-byte loads and one 16-byte store per thread (the generic kernel: three 4-byte loads, one 4-byte store and 64-bit divisions per pixel)
__global__ __launch_bounds__(256) void k_quantize_input_nhwc3(const float* __restrict__ x, int64_t npix4, const float* __restrict__ qrec, int8_t* __restrict__ out) {
  const QP q = load_qp(qrec);
  const uint32_t padb = (uint32_t)((q.zp - 128) & 255) << 24;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < npix4; i += (int64_t)gridDim.x * 256) {
    const float4 a = ((const float4*)x)[3 * i], b = ((const float4*)x)[3 * i + 1], c = ((const float4*)x)[3 * i + 2];
    const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    uint32_t o[4];
#pragma unroll
    for (int pxl = 0; pxl < 4; ++pxl) {
      uint32_t packed = padb;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) packed |= ((uint32_t)((fq_index(v[3 * pxl + ch], q.inv, q.zp, 0, q.hi) - 128) & 255)) << (8 * ch);
      o[pxl] = packed;
    }
    ((uint4*)out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
extern "C" int frost_quantize_input(const float* x, int n, int c, int h, int w, int64_t sn, int64_t sc, int64_t sh,
                                    int64_t sw, const float* qrec, int8_t* out, int cpad, void* stream) {
  FROST_REQUIRE(cpad % 4 == 0 && cpad >= c, "quantize_input: cpad must be a multiple of 4 and >= c");
  const int64_t npix = (int64_t)n * h * w;
  if (c == 3 && cpad == 4 && sc == 1 && sw == 3 && sh == (int64_t)w * 3 && sn == (int64_t)h * w * 3 && (npix & 3) == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
    hipLaunchKernelGGL(k_quantize_input_nhwc3, dim3(grid_for(npix / 4, 256, 8192)), dim3(256), 0, as_stream(stream), x, npix / 4, qrec, out);
    return frost_check_launch("quantize_input");
  }
  hipLaunchKernelGGL(k_quantize_input, dim3(grid_for((int64_t)n * h * w, 256)), dim3(256), 0, as_stream(stream), x, n,
                     c, h, w, sn, sc, sh, sw, qrec, out, cpad);
  return frost_check_launch("quantize_input");
}

// strided min/max of the logical (N,C,H,W) fp32 input (the QuantStub observer)
__global__ __launch_bounds__(256) void k_minmax_strided(const float* __restrict__ x, int n, int c, int h, int w,
                                                        int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* out2) {
  float lo = INFINITY, hi = -INFINITY;
  int64_t tot = (int64_t)n * c * h * w;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    // iterate in memory-friendly order for both NCHW-contiguous and channels_last: decode as (n, h, w, c) if sc==1
    int64_t t = i; float v;
    if (sc == 1) { int ch = (int)(t % c); t /= c; int xw = (int)(t % w); t /= w; int yh = (int)(t % h); int in = (int)(t / h);
      v = x[in * sn + ch * sc + yh * sh + xw * sw]; }
    else { int xw = (int)(t % w); t /= w; int yh = (int)(t % h); t /= h; int ch = (int)(t % c); int in = (int)(t / c);
      v = x[in * sn + ch * sc + yh * sh + xw * sw]; }
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  block_minmax_commit(lo, hi, out2);
}
extern "C" int frost_minmax_input(const float* x, int n, int c, int h, int w, int64_t sn, int64_t sc, int64_t sh,
                                  int64_t sw, float* out2, void* stream) {
  // min/max is order-free: a dense tensor (NCHW-contiguous or channels_last) is scanned linearly with 16-byte loads
  const int64_t tot = (int64_t)n * c * h * w;
  const bool nchw = (sw == 1 && sh == w && sc == (int64_t)h * w && sn == (int64_t)c * h * w);
  const bool nhwc = (sc == 1 && sw == c && sh == (int64_t)w * c && sn == (int64_t)h * w * c);
  if ((nchw || nhwc) && ((uintptr_t)x & 15) == 0) {
    hipLaunchKernelGGL(k_minmax_f32, dim3(grid_for(tot, 16384, 1024)), dim3(256), 0, as_stream(stream), x, tot, out2);
    return frost_check_launch("minmax_input");
  }
  hipLaunchKernelGGL(k_minmax_strided, dim3(grid_for(tot, 1024, 2048)), dim3(256), 0,
                     as_stream(stream), x, n, c, h, w, sn, sc, sh, sw, out2);
  return frost_check_launch("minmax_input");
}

// ------------------------------------------------------------------------------------------------ generic FQ
__global__ __launch_bounds__(256) void k_fake_quant_f32(const float* __restrict__ x, int64_t n, const float* qrec, int qmin,
                                                        int qmax, float* __restrict__ y, uint8_t* __restrict__ mask) {
  QP q = load_qp(qrec);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    bool in; int idx = fq_index(x[i], q.inv, q.zp, qmin, qmax, &in);
    y[i] = (float)(idx - q.zp) * q.scale;
    if (mask) mask[i] = in ? 1 : 0;
  }
}
extern "C" int frost_fake_quant_f32(const float* x, int64_t n, const float* qrec, int qmin, int qmax, float* y,
                                    uint8_t* mask, void* stream) {
  hipLaunchKernelGGL(k_fake_quant_f32, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), x, n, qrec, qmin, qmax, y, mask);
  return frost_check_launch("fake_quant_f32");
}
__global__ __launch_bounds__(256) void k_fake_quant_bwd_f32(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                            int64_t n, float* __restrict__ dx) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] = mask[i] ? dy[i] : 0.0f;
}
extern "C" int frost_fake_quant_bwd_f32(const float* dy, const uint8_t* mask, int64_t n, float* dx, void* stream) {
  hipLaunchKernelGGL(k_fake_quant_bwd_f32, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), dy, mask, n, dx);
  return frost_check_launch("fake_quant_bwd_f32");
}
__global__ __launch_bounds__(256) void k_dequant_act(const int8_t* __restrict__ qv, int64_t n, const float* qrec, float* __restrict__ y) {
  QP q = load_qp(qrec);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = (float)((int)qv[i] + 128 - q.zp) * q.scale;
}
extern "C" int frost_dequant_act(const int8_t* qv, int64_t n, const float* qrec, float* y, void* stream) {
  hipLaunchKernelGGL(k_dequant_act, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), qv, n, qrec, y);
  return frost_check_launch("dequant_act");
}

// ------------------------------------------------------------------------------------------------ weight prep
__device__ __forceinline__ float w_scaled(const FrostWDesc& d, int co, int rest) {
  float wv = d.w[(int64_t)co * d.cin_g * d.kk + rest];
  if (d.gamma) {
    // QAT forward: scale_factor = gamma / sqrt(running_var + eps) (conv_fused.py:133-135); convert: gamma * rsqrt(running_var + eps)
    // (torch.nn.utils.fusion.fuse_conv_bn_weights) -- the same number up to one rounding, kept apart so that both modes are bit-faithful
    const float sf = d.reserved0 ? d.gamma[co] * (1.0f / sqrtf(d.rvar[co] + FROST_BN_EPS)) : d.gamma[co] / sqrtf(d.rvar[co] + FROST_BN_EPS);
    wv = wv * sf;
  }
  return wv;
}
__global__ __launch_bounds__(256) void k_wprep_minmax(const FrostWDesc* descs) {
  const FrostWDesc d = descs[blockIdx.y];
  int per = d.cin_g * d.kk; int64_t tot = (int64_t)d.cout * per;
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int co = (int)(i / per);
    float v = w_scaled(d, co, (int)(i - (int64_t)co * per));
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  block_minmax_commit(lo, hi, d.minmax2);
}
__global__ void k_wprep_observe(const FrostWDesc* descs, int nlayers, int rule127, int observe) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlayers) return;
  const FrostWDesc d = descs[i];
  observer_update_dev(d.qrec, d.minmax2[0], d.minmax2[1], 1, rule127, observe);
  d.minmax2[0] = INFINITY; d.minmax2[1] = -INFINITY;
}
// Per-output-channel weight scales (always produced: the downstream kernels read d.wscale[c]).
//   per-tensor mode (qnnpack qconfig): wscale[c] = qrec.scale for every channel.
//   per-channel mode (FrostWDesc.reserved1; torch MovingAveragePerChannelMinMaxObserver + per_channel_symmetric qint8, ch_axis 0):
//     one wave per output channel: min/max of the BN-scaled weights of that channel, EMA (c = 0.01, first call takes the values),
//     s_c = max(-min-, max+) / 127.5 (v0 rule; rule127: max(-min-/128, max+/127)), eps clamp; qrec.scale <- max_c s_c.
__global__ __launch_bounds__(256) void k_wprep_scales(const FrostWDesc* descs, int rule127, int observe) {
  const FrostWDesc d = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (!d.reserved1) {
    const float s = d.qrec[FROST_Q_SCALE];
    for (int c = tid; c < d.cpad; c += 256) d.wscale[c] = s;
    return;
  }
  const bool obs = observe && (__float_as_int(d.qrec[FROST_Q_OBS_EN]) != 0);
  const int per = d.cin_g * d.kk;
  __shared__ float smax[4];
  float best = 0.0f;
  for (int co = wv; co < d.cpad; co += 4) {
    float sc = 1.0f;
    if (co < d.cout) {
      if (obs) {
        float lo = INFINITY, hi = -INFINITY;
        for (int r = lane; r < per; r += 64) { const float v = w_scaled(d, co, r); lo = fminf(lo, v); hi = fmaxf(hi, v); }
        lo = wave_min(lo); hi = wave_max(hi);
        float mn = d.wmin[co], mx = d.wmax[co];
        if (isinf(mn) && isinf(mx) && mn > 0.0f && mx < 0.0f) { mn = lo; mx = hi; }
        else { mn = mn + FROST_OBS_C * (lo - mn); mx = mx + FROST_OBS_C * (hi - mx); }
        if (lane == 0) { d.wmin[co] = mn; d.wmax[co] = mx; }
        const float mn_neg = fminf(mn, 0.0f), mx_pos = fmaxf(mx, 0.0f);
        sc = rule127 ? fmaxf(-mn_neg / 128.0f, mx_pos / 127.0f) : fmaxf(-mn_neg, mx_pos) / 127.5f;
        sc = fmaxf(sc, FROST_F32_EPS);
      } else sc = d.wscale[co];
      best = fmaxf(best, sc);
    }
    if (lane == 0) d.wscale[co] = sc;
  }
  if (lane == 0) smax[wv] = best;
  __syncthreads();
  if (tid == 0) { const float m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])); d.qrec[FROST_Q_SCALE] = m; d.qrec[FROST_Q_INV] = 1.0f / m; }
}
__device__ __forceinline__ int wq_at(const FrostWDesc& d, float inv, int co, int rest) {
  return fq_index(w_scaled(d, co, rest), d.reserved1 ? 1.0f / d.wscale[co] : inv, 0, -128, 127);
}
// pack kernel: one thread per packed dword
__global__ __launch_bounds__(256) void k_wprep_pack(const FrostWDesc* descs) {
  const FrostWDesc d = descs[blockIdx.y];
  float inv = 1.0f / d.qrec[FROST_Q_SCALE];
  if (d.kind == 0) {            // pointwise: [ct][ks][lane][16B]; also bf16 transposed pack [cit][kb][lane][8]
    int CT = d.cpad / 16, KS = d.kpad / 64;
    int64_t ndw = (int64_t)CT * KS * 64 * 4;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < ndw; i += (int64_t)gridDim.x * 256) {
      int dwi = (int)(i & 3); int lane = (int)((i >> 2) & 63); int64_t t = i >> 8; int ks = (int)(t % KS); int ct = (int)(t / KS);
      int co = ct * 16 + (lane & 15); int k0 = ks * 64 + (lane >> 4) * 16 + dwi * 4;
      uint32_t packed = 0;
      for (int e = 0; e < 4; ++e) {
        int k = k0 + e; int v = 0;
        if (co < d.cout && k < d.cin_g) v = wq_at(d, inv, co, k);
        packed |= ((uint32_t)(v & 255)) << (8 * e);
      }
      ((uint32_t*)d.wq_pack)[i] = packed;
    }
    if (d.wt_pack) {
      int cinp = round_up(d.cin_g, 16); int CIT = cinp / 16, KB = d.cpad / 32 + ((d.cpad % 32) ? 1 : 0);
      int64_t nel = (int64_t)CIT * KB * 64 * 8;
      for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < nel; i += (int64_t)gridDim.x * 256) {
        int e = (int)(i & 7); int lane = (int)((i >> 3) & 63); int64_t t = i >> 9; int kb = (int)(t % KB); int cit = (int)(t / KB);
        int ci = cit * 16 + (lane & 15); int g = lane >> 4;
        int co = kb * 32 + 8 * g + e;
        float v = 0.0f;      // per-channel mode: the data-gradient kernels apply the scalar qrec.scale, so the pack carries s_c / scale (1 in per-tensor mode)
        if (co < d.cout && ci < d.cin_g) v = (float)wq_at(d, inv, co, ci) * (d.reserved1 ? d.wscale[co] * inv : 1.0f);
        d.wt_pack[i] = f2bf(v);
      }
    }
  } else if (d.kind == 1) {     // depthwise: [tap][cpad]
    int64_t nb = (int64_t)d.kk * d.cpad;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) {
      int c = (int)(i % d.cpad), tap = (int)(i / d.cpad);
      d.wq_pack[i] = (int8_t)((c < d.cout) ? wq_at(d, inv, c, tap) : 0);
    }
  } else if (d.kind == 2) {     // stem as im2col + pointwise: K index k = tap*4 + c (c == 3 is the zero pad channel), K = 4*kk
    int CT = d.cpad / 16, KS = d.kpad / 64;
    int64_t ndw = (int64_t)CT * KS * 64 * 4;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < ndw; i += (int64_t)gridDim.x * 256) {
      int dwi = (int)(i & 3); int lane = (int)((i >> 2) & 63); int64_t t = i >> 8; int ks = (int)(t % KS); int ct = (int)(t / KS);
      int co = ct * 16 + (lane & 15); int k0 = ks * 64 + (lane >> 4) * 16 + dwi * 4;
      uint32_t packed = 0;
      for (int e = 0; e < 4; ++e) {
        int k = k0 + e; int tap = k >> 2, c = k & 3; int v = 0;
        if (co < d.cout && tap < d.kk && c < d.cin_g) v = wq_at(d, inv, co, c * d.kk + tap);
        packed |= ((uint32_t)(v & 255)) << (8 * e);
      }
      ((uint32_t*)d.wq_pack)[i] = packed;
    }
  } else {                      // classifier: plain [cout][cin]
    int64_t nb = (int64_t)d.cout * d.cin_g;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) {
      int co = (int)(i / d.cin_g);
      d.wq_pack[i] = (int8_t)wq_at(d, inv, co, (int)(i - (int64_t)co * d.cin_g));
    }
  }
}
// wsum: one wave per output channel
__global__ __launch_bounds__(256) void k_wprep_wsum(const FrostWDesc* descs) {
  const FrostWDesc d = descs[blockIdx.y];
  if (!d.wsum) return;
  float inv = 1.0f / d.qrec[FROST_Q_SCALE];
  int per = d.cin_g * d.kk;
  for (int co = blockIdx.x * 4 + (threadIdx.x >> 6); co < d.cpad; co += gridDim.x * 4) {
    int s = 0;
    if (co < d.cout) for (int r = threadIdx.x & 63; r < per; r += 64) s += wq_at(d, inv, co, r);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) d.wsum[co] = s;
  }
}
extern "C" int frost_weight_prep(const FrostWDesc* descs, int nlayers, int max_elems, int rule127, int observe,
                                 void* stream) {
  hipStream_t s = as_stream(stream);
  // grid.x = workgroups per layer: the launch is as long as its largest layers (classifier 1.28 M, last_layer 0.41 M, layer5.0.conv1 0.5 M weights), which at 64 / 32 workgroups
  // each kept the three big launches at 38 + 53 + 63 us; the small layers' extra workgroups exit at once
  static const int wcap = getenv("FROST_WPREP_WGS") ? atoi(getenv("FROST_WPREP_WGS")) : 256;
  int gx = grid_for(max_elems, 1024, wcap);
  if (observe) hipLaunchKernelGGL(k_wprep_minmax, dim3(gx, nlayers), dim3(256), 0, s, descs);
  hipLaunchKernelGGL(k_wprep_observe, dim3((nlayers + 63) / 64), dim3(64), 0, s, descs, nlayers, rule127, observe);
  hipLaunchKernelGGL(k_wprep_scales, dim3(nlayers), dim3(256), 0, s, descs, rule127, observe);
  hipLaunchKernelGGL(k_wprep_pack, dim3(gx, nlayers), dim3(256), 0, s, descs);
  static const int wsg = getenv("FROST_WSUM_WGS") ? atoi(getenv("FROST_WSUM_WGS")) : (wcap >= 128 ? 128 : 32);      // A/B: a wave per output channel, channels strided over the grid
  hipLaunchKernelGGL(k_wprep_wsum, dim3(wsg, nlayers), dim3(256), 0, s, descs);
  return frost_check_launch("weight_prep");
}

// ------------------------------------------------------------------------------------------------ step prologue in three launches (round 6)
// frost_save_sigma + frost_weight_prep (5 launches) + frost_stats_init_table = 7 launches of `nlayers` x 8 ... 256 workgroups each, most of which exit at once, every one as
// long as its largest layer: 0.28 ms at the head of every captured step (FROST_ABL_SKIP pricing, profiles/r06_pricing.txt) whether or not it ran on a second stream.  Here the
// same arithmetic -- w_scaled(), observer_update_dev(), wq_at(): bit-identical packs -- over a FLAT workgroup map built by the host (a layer gets workgroups in proportion to
// its weights: wgmap[b] = {layer, slot, slots of the layer, -}):
//   A  range of the BN-scaled weights (look-before-atomic commit) + sigma_r snapshot + reset of the layer's integer statistics rows
//   B  one workgroup per layer: weight observer / qparams, per-channel scales (k_wprep_observe + k_wprep_scales)
//   C  packs (int8 MFMA fragments, transposed bf16, depthwise taps, classifier rows) + per-channel weight sums
__global__ __launch_bounds__(256) void k_wprep_a(const FrostWDesc* descs, const int4* __restrict__ wgmap, float* const* sigma_outs, uint8_t* stats_base,
                                                 const int32_t* cpads, const int64_t* offs, int observe) {
  const int4 wm = wgmap[blockIdx.x];
  const int l = wm.x, slot = wm.y, nsl = wm.z;
  const FrostWDesc d = descs[l];
  {   // integer statistics rows of the layer: identity values
    const int cp = cpads[l];
    int64_t* s1 = (int64_t*)(stats_base + offs[l]); uint64_t* s2 = (uint64_t*)(s1 + cp);
    int32_t* mn = (int32_t*)(s2 + cp); int32_t* mx = mn + cp;
    for (int i = slot * 256 + threadIdx.x; i < cp * FROST_STATS_NC; i += nsl * 256) {          // every replicated table (frost_common.h, stats_copy)
      const int k = i / cp, c = i - k * cp; const size_t o8 = (size_t)k * cp * 3, o4 = (size_t)k * cp * 6;
      s1[o8 + c] = 0; s2[o8 + c] = 0; mn[o4 + c] = INT32_MAX; mx[o4 + c] = INT32_MIN;
    }
  }
  if (d.rvar) {   // sigma_r = sqrt(running_var + eps) BEFORE this step's forward updates running_var
    float* o = sigma_outs[l];
    for (int c = slot * 256 + threadIdx.x; c < d.cout; c += nsl * 256) o[c] = sqrtf(d.rvar[c] + FROST_BN_EPS);
  }
  if (!observe) return;
  const int per = d.cin_g * d.kk; const int64_t tot = (int64_t)d.cout * per;
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = (int64_t)slot * 256 + threadIdx.x; i < tot; i += (int64_t)nsl * 256) {
    const int co = (int)(i / per);
    const float v = w_scaled(d, co, (int)(i - (int64_t)co * per));
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  block_minmax_commit(lo, hi, d.minmax2);
}
__global__ __launch_bounds__(256) void k_wprep_b(const FrostWDesc* descs, int rule127, int observe) {
  const FrostWDesc d = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    observer_update_dev(d.qrec, d.minmax2[0], d.minmax2[1], 1, rule127, observe);
    d.minmax2[0] = INFINITY; d.minmax2[1] = -INFINITY;
  }
  __syncthreads();          // (one workgroup: the record written by thread 0 is visible to it after the barrier)
  if (!d.reserved1) {
    const float s = d.qrec[FROST_Q_SCALE];
    for (int c = tid; c < d.cpad; c += 256) d.wscale[c] = s;
    return;
  }
  const bool obs = observe && (__float_as_int(d.qrec[FROST_Q_OBS_EN]) != 0);
  const int per = d.cin_g * d.kk;
  __shared__ float smax[4];
  float best = 0.0f;
  for (int co = wv; co < d.cpad; co += 4) {
    float sc = 1.0f;
    if (co < d.cout) {
      if (obs) {
        float lo = INFINITY, hi = -INFINITY;
        for (int r = lane; r < per; r += 64) { const float v = w_scaled(d, co, r); lo = fminf(lo, v); hi = fmaxf(hi, v); }
        lo = wave_min(lo); hi = wave_max(hi);
        float mn = d.wmin[co], mx = d.wmax[co];
        if (isinf(mn) && isinf(mx) && mn > 0.0f && mx < 0.0f) { mn = lo; mx = hi; }
        else { mn = mn + FROST_OBS_C * (lo - mn); mx = mx + FROST_OBS_C * (hi - mx); }
        if (lane == 0) { d.wmin[co] = mn; d.wmax[co] = mx; }
        const float mn_neg = fminf(mn, 0.0f), mx_pos = fmaxf(mx, 0.0f);
        sc = rule127 ? fmaxf(-mn_neg / 128.0f, mx_pos / 127.0f) : fmaxf(-mn_neg, mx_pos) / 127.5f;
        sc = fmaxf(sc, FROST_F32_EPS);
      } else sc = d.wscale[co];
      best = fmaxf(best, sc);
    }
    if (lane == 0) d.wscale[co] = sc;
  }
  if (lane == 0) smax[wv] = best;
  __syncthreads();
  if (tid == 0) { const float m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])); d.qrec[FROST_Q_SCALE] = m; d.qrec[FROST_Q_INV] = 1.0f / m; }
}
__global__ __launch_bounds__(256) void k_wprep_c(const FrostWDesc* descs, const int4* __restrict__ wgmap) {
  const int4 wm = wgmap[blockIdx.x];
  const int slot = wm.y, nsl = wm.z;
  const FrostWDesc d = descs[wm.x];
  const float inv = 1.0f / d.qrec[FROST_Q_SCALE];
  const int64_t i0 = (int64_t)slot * 256 + threadIdx.x, istep = (int64_t)nsl * 256;
  if (d.kind == 0) {            // pointwise: [ct][ks][lane][16B]; also bf16 transposed pack [cit][kb][lane][8]   (k_wprep_pack's index arithmetic)
    const int CT = d.cpad / 16, KS = d.kpad / 64;
    const int64_t ndw = (int64_t)CT * KS * 64 * 4;
    for (int64_t i = i0; i < ndw; i += istep) {
      const int dwi = (int)(i & 3), lane = (int)((i >> 2) & 63); const int64_t t = i >> 8; const int ks = (int)(t % KS), ct = (int)(t / KS);
      const int co = ct * 16 + (lane & 15), k0 = ks * 64 + (lane >> 4) * 16 + dwi * 4;
      uint32_t packed = 0;
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + e; int v = 0;
        if (co < d.cout && k < d.cin_g) v = wq_at(d, inv, co, k);
        packed |= ((uint32_t)(v & 255)) << (8 * e);
      }
      ((uint32_t*)d.wq_pack)[i] = packed;
    }
    if (d.wt_pack) {
      const int cinp = round_up(d.cin_g, 16); const int CIT = cinp / 16, KB = d.cpad / 32 + ((d.cpad % 32) ? 1 : 0);
      const int64_t nel = (int64_t)CIT * KB * 64 * 8;
      for (int64_t i = i0; i < nel; i += istep) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63); const int64_t t = i >> 9; const int kb = (int)(t % KB), cit = (int)(t / KB);
        const int ci = cit * 16 + (lane & 15), g = lane >> 4;
        const int co = kb * 32 + 8 * g + e;
        float v = 0.0f;
        if (co < d.cout && ci < d.cin_g) v = (float)wq_at(d, inv, co, ci) * (d.reserved1 ? d.wscale[co] * inv : 1.0f);
        d.wt_pack[i] = f2bf(v);
      }
    }
  } else if (d.kind == 1) {     // depthwise: [tap][cpad]
    const int64_t nb = (int64_t)d.kk * d.cpad;
    for (int64_t i = i0; i < nb; i += istep) {
      const int c = (int)(i % d.cpad), tap = (int)(i / d.cpad);
      d.wq_pack[i] = (int8_t)((c < d.cout) ? wq_at(d, inv, c, tap) : 0);
    }
  } else if (d.kind == 2) {     // stem as im2col + pointwise: K index k = tap*4 + c
    const int CT = d.cpad / 16, KS = d.kpad / 64;
    const int64_t ndw = (int64_t)CT * KS * 64 * 4;
    for (int64_t i = i0; i < ndw; i += istep) {
      const int dwi = (int)(i & 3), lane = (int)((i >> 2) & 63); const int64_t t = i >> 8; const int ks = (int)(t % KS), ct = (int)(t / KS);
      const int co = ct * 16 + (lane & 15), k0 = ks * 64 + (lane >> 4) * 16 + dwi * 4;
      uint32_t packed = 0;
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + e; const int tap = k >> 2, c = k & 3; int v = 0;
        if (co < d.cout && tap < d.kk && c < d.cin_g) v = wq_at(d, inv, co, c * d.kk + tap);
        packed |= ((uint32_t)(v & 255)) << (8 * e);
      }
      ((uint32_t*)d.wq_pack)[i] = packed;
    }
  } else {                      // classifier: plain [cout][cin]
    const int64_t nb = (int64_t)d.cout * d.cin_g;
    for (int64_t i = i0; i < nb; i += istep) {
      const int co = (int)(i / d.cin_g);
      d.wq_pack[i] = (int8_t)wq_at(d, inv, co, (int)(i - (int64_t)co * d.cin_g));
    }
  }
  if (d.wsum) {                 // per-channel sums of the quantised weights: one wave per output channel (k_wprep_wsum)
    const int per = d.cin_g * d.kk;
    for (int co = slot * 4 + (threadIdx.x >> 6); co < d.cpad; co += nsl * 4) {
      int sm = 0;
      if (co < d.cout) for (int r = threadIdx.x & 63; r < per; r += 64) sm += wq_at(d, inv, co, r);
      for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
      if ((threadIdx.x & 63) == 0) d.wsum[co] = sm;
    }
  }
}
// descs / sigma_outs / cpads / offs: the tables of frost_save_sigma / frost_weight_prep / frost_stats_init_table for `nlayers` layers; wgmap: `nwg` entries
// {layer index into those tables, slot, slots of that layer, unused}, every layer with >= 1 slot, slots of a layer 0 .. n-1 each exactly once.
extern "C" int frost_step_prologue(const FrostWDesc* descs, int nlayers, const int32_t* wgmap, int nwg, float* const* sigma_outs, void* stats,
                                   const int32_t* cpads, const int64_t* offs, int rule127, int observe, void* stream) {
  FROST_REQUIRE(descs && wgmap && sigma_outs && stats && cpads && offs && nlayers >= 1 && nwg >= nlayers, "step_prologue: incomplete arguments");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(k_wprep_a, dim3(nwg), dim3(256), 0, s, descs, (const int4*)wgmap, sigma_outs, (uint8_t*)stats, cpads, offs, observe);
  hipLaunchKernelGGL(k_wprep_b, dim3(nlayers), dim3(256), 0, s, descs, rule127, observe);
  hipLaunchKernelGGL(k_wprep_c, dim3(nwg), dim3(256), 0, s, descs, (const int4*)wgmap);
  return frost_check_launch("step_prologue");
}

// replaces: the `weight` entry of a converted module's state_dict (nnq.Conv2d._weight_bias(), Classification/evaluate.py:140-143 saves it): the int8 values the
// packs of frost_weight_prep hold, in the module's own [cout][cin/g][kh][kw] order -- the same wq_at() the pack kernels call, so the exported tensor IS the
// weight the device convolves with
__global__ __launch_bounds__(256) void k_export_wq(const FrostWDesc* descs, int layer, int8_t* __restrict__ out) {
  const FrostWDesc d = descs[layer];
  const float inv = 1.0f / d.qrec[FROST_Q_SCALE];
  const int per = d.cin_g * d.kk;
  const int64_t n = (int64_t)d.cout * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i / per);
    out[i] = (int8_t)wq_at(d, inv, co, (int)(i - (int64_t)co * per));
  }
}
extern "C" int frost_export_wq(const FrostWDesc* descs, int layer, int64_t nelem, int8_t* out, void* stream) {
  FROST_REQUIRE(descs && out && layer >= 0 && nelem > 0, "export_wq: bad arguments");
  hipLaunchKernelGGL(k_export_wq, dim3(grid_for(nelem, 256, 1024)), dim3(256), 0, as_stream(stream), descs, layer, out);
  return frost_check_launch("export_wq");
}

// ------------------------------------------------------------------------------------------------ stats init
__global__ __launch_bounds__(256) void k_stats_init(uint8_t* base, const int32_t* cpads, const int64_t* offs) {
  int l = blockIdx.y; int cp = cpads[l];
  int64_t* s1 = (int64_t*)(base + offs[l]); uint64_t* s2 = (uint64_t*)(s1 + cp);
  int32_t* mn = (int32_t*)(s2 + cp); int32_t* mx = mn + cp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < cp * FROST_STATS_NC; i += gridDim.x * 256) {          // every replicated table (frost_common.h, stats_copy)
    const int k = i / cp, c = i - k * cp; const size_t o8 = (size_t)k * cp * 3, o4 = (size_t)k * cp * 6;
    s1[o8 + c] = 0; s2[o8 + c] = 0; mn[o4 + c] = INT32_MAX; mx[o4 + c] = INT32_MIN;
  }
}
extern "C" int frost_stats_init_table(void* stats, const int32_t* cpads, const int64_t* offs, int nlayers, void* stream) {
  hipLaunchKernelGGL(k_stats_init, dim3(8, nlayers), dim3(256), 0, as_stream(stream), (uint8_t*)stats, cpads, offs);
  return frost_check_launch("stats_init");
}

// ------------------------------------------------------------------------------------------------ finalize
// One block. Turns integer stats into BN coefficients, running-stat updates, and the activation qrecord.
__global__ __launch_bounds__(256) void k_conv_finalize(const uint8_t* stats, int64_t count, int cout, int cpad,
                                                       const float* qx, const float* qw, const float* gamma,
                                                       const float* beta, float* rmean, float* rvar, int64_t* nbt,
                                                       int training, int relu, int observe, int have_stats, float* coef, float* qy, const float* wscale) {
  __shared__ float sh[8];
  conv_finalize_dev(stats, count, cout, cpad, qx, qw, wscale, gamma, beta, rmean, rvar, nbt, training, relu, observe, have_stats, coef, qy,
                    threadIdx.x, 256, sh);
}
extern "C" int frost_conv_finalize(const void* stats, int64_t count, int cout, const float* qrec_x, const float* qrec_w,
                                   const float* gamma, const float* beta, float* rmean, float* rvar, int64_t* nbt,
                                   int training, int relu, int observe, float* coef, float* qrec_y, const float* wscale, void* stream) {
  int cpad = round_up(cout, 16);
  hipLaunchKernelGGL(k_conv_finalize, dim3(1), dim3(256), 0, as_stream(stream), (const uint8_t*)stats, count, cout, cpad,
                     qrec_x, qrec_w, gamma, beta, rmean, rvar, nbt, training, relu, observe, stats ? 1 : 0, coef, qrec_y, wscale);
  return frost_check_launch("conv_finalize");
}

// ------------------------------------------------------------------------------------------------ cat
__global__ void k_cat_observe(const float* qa, const float* qb, float* qy, int observe) {
  if (threadIdx.x == 0)
    observer_update_dev(qy, fminf(qa[FROST_Q_FQMIN], qb[FROST_Q_FQMIN]), fmaxf(qa[FROST_Q_FQMAX], qb[FROST_Q_FQMAX]), 0, 0, observe);
}
extern "C" int frost_cat_observe(const float* qrec_a, const float* qrec_b, float* qrec_y, int observe, void* stream) {
  hipLaunchKernelGGL(k_cat_observe, dim3(1), dim3(64), 0, as_stream(stream), qrec_a, qrec_b, qrec_y, observe);
  return frost_check_launch("cat_observe");
}
// requantise through two 256-entry LUTs held in LDS; y[p] = [lutA[a[p][:ca]], lutB[b[p][:cb]]]
__global__ __launch_bounds__(256) void k_cat_requant(const int8_t* __restrict__ a, const float* qa, int ca,
                                                     const int8_t* __restrict__ b, const float* qb, int cb, int64_t npix,
                                                     const float* qy, int8_t* __restrict__ y) {
  __shared__ uint8_t lut[2][256];
  {
    QP A = load_qp(qa), B = load_qp(qb), Y = load_qp(qy);
    int i = threadIdx.x;   // i = stored byte as unsigned; offset-binary index q = (int8)i + 128
    int q = (int)(int8_t)i + 128;
    float va = (float)(q - A.zp) * A.scale, vb = (float)(q - B.zp) * B.scale;
    lut[0][i] = (uint8_t)((fq_index(va, Y.inv, Y.zp, 0, Y.hi) - 128) & 255);
    lut[1][i] = (uint8_t)((fq_index(vb, Y.inv, Y.zp, 0, Y.hi) - 128) & 255);
  }
  __syncthreads();
  int cy = ca + cb; int dpp = cy >> 2;           // dwords per pixel (channels multiple of 4)
  int64_t ndw = npix * dpp;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < ndw; i += (int64_t)gridDim.x * 256) {
    int64_t p = i / dpp; int c0 = (int)(i - p * dpp) * 4;
    uint32_t src; const uint8_t* l;
    if (c0 < ca) { src = *(const uint32_t*)(a + p * ca + c0); l = lut[0]; }
    else { src = *(const uint32_t*)(b + p * cb + (c0 - ca)); l = lut[1]; }
    uint32_t o = (uint32_t)l[src & 255] | ((uint32_t)l[(src >> 8) & 255] << 8) | ((uint32_t)l[(src >> 16) & 255] << 16) |
                 ((uint32_t)l[src >> 24] << 24);
    *(uint32_t*)(y + i * 4) = o;
  }
}
extern "C" int frost_cat_requant(const int8_t* a, const float* qrec_a, int ca, const int8_t* b, const float* qrec_b,
                                 int cb, int64_t npix, const float* qrec_y, int8_t* y, void* stream) {
  FROST_REQUIRE(ca % 4 == 0 && cb % 4 == 0, "cat: channel counts must be multiples of 4");
  hipLaunchKernelGGL(k_cat_requant, dim3(grid_for(npix * ((ca + cb) / 4), 1024, 4096)), dim3(256), 0, as_stream(stream),
                     a, qrec_a, ca, b, qrec_b, cb, npix, qrec_y, y);
  return frost_check_launch("cat_requant");
}

// ------------------------------------------------------------------------------------------------ add
__device__ __forceinline__ float add_val(int ba, int bb, const QP& A, const QP& B) {
  return (float)(ba + 128 - A.zp) * A.scale + (float)(bb + 128 - B.zp) * B.scale;
}
__global__ __launch_bounds__(256) void k_add_minmax(const int8_t* __restrict__ a, const float* qa, const int8_t* __restrict__ b,
                                                    const float* qb, int64_t n, float* out2, float* qy, uint32_t* ticket, int observe) {
  // 16 B per operand per lane and two independent loads in flight; few, fat workgroups: the final float atomics on the two
  // result words serialise, so their count (one pair per workgroup) is part of the critical path
  QP A = load_qp(qa), B = load_qp(qb);
  float lo = INFINITY, hi = -INFINITY;
  const int64_t n16 = n >> 4; const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += 2 * stride) {
    const uint4 va0 = ((const uint4*)a)[i], vb0 = ((const uint4*)b)[i];
    const bool two = (i + stride) < n16;
    uint4 va1 = va0, vb1 = vb0;
    if (two) { va1 = ((const uint4*)a)[i + stride]; vb1 = ((const uint4*)b)[i + stride]; }
    const uint32_t wa[8] = {va0.x, va0.y, va0.z, va0.w, va1.x, va1.y, va1.z, va1.w}, wb[8] = {vb0.x, vb0.y, vb0.z, vb0.w, vb1.x, vb1.y, vb1.z, vb1.w};
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = add_val((int)(int8_t)(wa[q] >> (8 * e)), (int)(int8_t)(wb[q] >> (8 * e)), A, B);
        lo = fminf(lo, v); hi = fmaxf(hi, v);
      }
  }
  for (int64_t i = (n16 << 4) + blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {     // tail (n % 16)
    const float v = add_val((int)a[i], (int)b[i], A, B); lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  if (ticket) {        // fused form: per-workgroup slots, the last workgroup folds them and runs the observer update (no launch of its own, no same-address float atomics)
    __shared__ int sflag; __shared__ float shr[8];
    if (range_fold_last(lo, hi, out2 + 2 + FROST_TICKET_WORDS, ticket, shr, &sflag)) observer_update_dev(qy, lo, hi, 0, 0, observe);
    return;
  }
  block_minmax_commit(lo, hi, out2);
}
// range pass + MovingAverageMinMax update of the sum's FakeQuantize in one launch.  state: frost_add_state_floats() floats = {2 unused, arrival ticket
// [FROST_TICKET_WORDS] (zeroed once, left zeroed), 2 * FROST_MM_SLOTS per-workgroup range slots}.
extern "C" int frost_add_state_floats(void) { return 2 + FROST_TICKET_WORDS + 2 * FROST_MM_SLOTS; }
extern "C" int frost_add_minmax_observe(const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n,
                                        float* state3, float* qrec_y, int observe, void* stream) {
  FROST_REQUIRE(n % 4 == 0, "add: n must be a multiple of 4");
  // (with per-workgroup slots the tail no longer grows with the grid: more, thinner workgroups than the atomics form could afford)
  hipLaunchKernelGGL(k_add_minmax, dim3(grid_for(n, 8192, FROST_MM_SLOTS)), dim3(256), 0, as_stream(stream), a, qrec_a, b, qrec_b, n, state3, qrec_y,
                     (uint32_t*)(state3 + 2), observe);
  return frost_check_launch("add_minmax_observe");
}
extern "C" int frost_add_minmax(const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n,
                                float* minmax2, void* stream) {
  FROST_REQUIRE(n % 4 == 0, "add: n must be a multiple of 4");
  hipLaunchKernelGGL(k_add_minmax, dim3(grid_for(n, 32768, 512)), dim3(256), 0, as_stream(stream), a, qrec_a, b, qrec_b, n, minmax2, (float*)nullptr,
                     (uint32_t*)nullptr, 0);
  return frost_check_launch("add_minmax");
}
__global__ __launch_bounds__(256) void k_add_requant(const int8_t* __restrict__ a, const float* qa, const int8_t* __restrict__ b,
                                                     const float* qb, int64_t n, const float* qy, int8_t* __restrict__ y) {
  QP A = load_qp(qa), B = load_qp(qb), Y = load_qp(qy);
  int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    uint32_t va = ((const uint32_t*)a)[i], vb = ((const uint32_t*)b)[i], o = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = add_val((int)(int8_t)(va >> (8 * e)), (int)(int8_t)(vb >> (8 * e)), A, B);
      o |= ((uint32_t)((fq_index(v, Y.inv, Y.zp, 0, Y.hi) - 128) & 255)) << (8 * e);
    }
    ((uint32_t*)y)[i] = o;
  }
}
extern "C" int frost_add_requant(const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n,
                                 const float* qrec_y, int8_t* y, void* stream) {
  FROST_REQUIRE(n % 4 == 0, "add: n must be a multiple of 4");
  hipLaunchKernelGGL(k_add_requant, dim3(grid_for(n, 4096, 4096)), dim3(256), 0, as_stream(stream), a, qrec_a, b, qrec_b, n, qrec_y, y);
  return frost_check_launch("add_requant");
}

// ------------------------------------------------------------------------------------------------ head
// avg-pool over hw positions of an offset-binary activation -> fp32 [n][c]  (x mask if given)
__global__ __launch_bounds__(256) void k_avgpool(const int8_t* __restrict__ x, const float* qx, int n, int hw, int c,
                                                 const float* __restrict__ drop, float* __restrict__ y) {
  QP X = load_qp(qx);
  int64_t tot = (int64_t)n * c;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int ch = (int)(i % c); int64_t in = i / c;
    const int8_t* p = x + in * hw * c + ch;
    float s = 0.0f;   // torch mean(): fp32 sum of the dequantised values, then / hw
    for (int t = 0; t < hw; ++t) s += (float)((int)p[(int64_t)t * c] + 128 - X.zp) * X.scale;
    float v = s / (float)hw;
    if (drop) v *= drop[i];
    y[i] = v;
  }
}
extern "C" int frost_avgpool(const int8_t* x, const float* qrec_x, int n, int hw, int c, const float* drop_mask, float* y,
                             void* stream) {
  hipLaunchKernelGGL(k_avgpool, dim3(grid_for((int64_t)n * c, 256)), dim3(256), 0, as_stream(stream), x, qrec_x, n, hw, c, drop_mask, y);
  return frost_check_launch("avgpool");
}
// ------------------------------------------------------------------------------------------------ stem im2col
// 3x3 stride-2 pad-1 patches of the 4-byte-per-pixel quantised image -> [npix_out][40] (36 patch bytes + 4 zero-point
// bytes), so the stem runs on the pointwise int8-MFMA kernels (K = 40) forward AND backward (wgrad = plain pw wgrad).
__global__ __launch_bounds__(256) void k_stem_im2col(const int8_t* __restrict__ x, const float* qx, int n, int h, int w, int ho, int wo,
                                                     int8_t* __restrict__ out) {
  const int zp = __float_as_int(qx[FROST_Q_ZP]);
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;
  const int64_t npix = (int64_t)n * ho * wo;
  for (int64_t p = blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const int ox = (int)(p % wo); int64_t t = p / wo; const int oy = (int)(t % ho); const int img = (int)(t / ho);
    uint32_t v[10];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
        v[ky * 3 + kx] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? *(const uint32_t*)(x + (((int64_t)img * h + iy) * w + ix) * 4) : zfill;
      }
    v[9] = zfill;
    uint32_t* dst = (uint32_t*)(out + p * 40);
#pragma unroll
    for (int i = 0; i < 10; i += 2) *(uint2*)(dst + i) = make_uint2(v[i], v[i + 1]);
  }
}
extern "C" int frost_stem_im2col(const int8_t* x, const float* qrec_x, int n, int h, int w, int8_t* out, void* stream) {
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(k_stem_im2col, dim3(grid_for((int64_t)n * ho * wo, 256, 8192)), dim3(256), 0, as_stream(stream), x, qrec_x, n, h, w, ho, wo, out);
  return frost_check_launch("stem_im2col");
}
// dwq_col[cout][40] (k = tap*4 + c) -> dwq[cout][cin_g][3][3] (OIHW)
__global__ void k_stem_wgrad_remap(const float* __restrict__ src, int cout, int cin_g, float* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout * cin_g * 9) return;
  int co = i / (cin_g * 9), r = i % (cin_g * 9); int c = r / 9, tap = r % 9;
  dst[i] = dwq_sum(src, (int64_t)cout * 40, co * 40 + tap * 4 + c);          // the fused stem backward spreads its sums over the copies of dwq_col (frost_common.h)
}
extern "C" int frost_stem_wgrad_remap(const float* dwq_col, int cout, int cin_g, float* dwq, void* stream) {
  int tot = cout * cin_g * 9;
  hipLaunchKernelGGL(k_stem_wgrad_remap, dim3((tot + 255) / 256), dim3(256), 0, as_stream(stream), dwq_col, cout, cin_g, dwq);
  return frost_check_launch("stem_wgrad_remap");
}
