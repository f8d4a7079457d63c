// Shared device/host helpers for libfrost_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include "../../include/frost_hip.h"

#define FROST_BN_EPS 1e-5f
#define FROST_BN_MOM 0.1f
#define FROST_OBS_C 0.01f
#define FROST_F32_EPS 1.1920928955078125e-07f

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

extern "C" void frost_set_error(const char* msg);
int frost_check_launch(const char* what);

#define FROST_REQUIRE(cond, msg) \
  do { if (!(cond)) { frost_set_error(msg); return 1; } } while (0)

__host__ __device__ static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---- qrecord access ---------------------------------------------------------------------------------------
struct QP { float scale, inv; int zp, hi; };
__device__ __forceinline__ int q_hi(const float* q) { const float m = q[FROST_Q_QMAX]; return (m > 0.0f) ? (int)m : 255; }   // activation index range 0..hi
__device__ __forceinline__ QP load_qp(const float* q) {
  QP r; r.scale = q[FROST_Q_SCALE]; r.zp = __float_as_int(q[FROST_Q_ZP]); r.inv = 1.0f / r.scale; r.hi = q_hi(q); return r;
}
// fake-quantise to the uint8 index (torch fake_quantize_per_tensor_affine): q = clamp(rint(x*inv)+zp, lo, hi)
__device__ __forceinline__ int fq_index(float x, float inv, int zp, int lo, int hi, bool* inrange = nullptr) {
  float qf = rintf(x * inv) + (float)zp;   // same fp32 op order as aten: zp + nearbyint(x*inv)
  if (inrange) *inrange = (qf >= (float)lo) && (qf <= (float)hi);
  qf = fminf(fmaxf(qf, (float)lo), (float)hi);
  return (int)qf;
}

// ---- bf16 ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// hard-swish z * relu6(z + 3) / 6 in the order torch evaluates the reference's _Hswish (mobilenetv3.py:43-56): add_scalar, relu6, mul, mul_scalar(1/6).
// The float and bf16-inference entries take it as activation code 2 of their `relu` argument (0 = none, 1 = ReLU).
__device__ __forceinline__ float hswish_f(float z) { return (z * fminf(fmaxf(z + 3.0f, 0.0f), 6.0f)) * (1.0f / 6.0f); }
__device__ __forceinline__ uint16_t f2bf(float f) {   // round-to-nearest-even
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// two fp32 -> packed bf16x2 (lo in bits 0..15), round-to-nearest-even, ONE instruction on gfx950 (no builtin: inline asm)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// ---- stochastic rounding fp32 -> bf16 for the dc tensor (gradient w.r.t. the conv output) ----------------------------------------
// Why: dc[p][c] = K1[c] * g[p][c] + small terms, with g itself on the bf16 lattice and K1 constant per channel -- round-to-nearest then has a
// deterministic per-channel bias (~1e-4 relative, measured: tests/devtools/dbg_wgrad.py).  The weight gradient sums dc * x over all pixels of a
// batch (up to 6.4 M) while training-mode BatchNorm makes sum_p dc = 0, so that bias is amplified by sqrt(pixels): 1.6e-2 (dW) / 4.5e-2 (dgamma)
// at 263 k pixels against the fp64 evaluation, and growing with the batch.  Stochastic rounding (add 16 pseudo-random bits below the kept
// mantissa, truncate) is unbiased whatever the lattice: the error stays at the 2^-9/sqrt(3) level independent of the pixel count.
// The generator is a 24-bit LCG per lane (v_mad_u32_u24, full rate), seeded from the workgroup / thread index: launches are reproducible.
__device__ __forceinline__ uint32_t sr_seed(uint32_t a, uint32_t b) { uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu; return (h ^ (h >> 15)) | 1u; }
// One LCG step = two full-rate VALU instructions, kept opaque: left to itself the compiler rewrites four consecutive steps as four independent
// closed forms x*a^k + c_k with 32-bit constants, i.e. four quarter-rate v_mul_lo_u32 per four elements (seen in the ISA of the dc epilogue).
__device__ __forceinline__ uint32_t sr_step(uint32_t st) {
  uint32_t r;
  asm("v_mul_u32_u24_e32 %0, 0x5bd1e5, %1\n\tv_add_u32_e32 %0, 0x9e3779, %0" : "=v"(r) : "v"(st));
  return r;
}
__device__ __forceinline__ uint32_t sr_next16(uint32_t& st) { st = sr_step(st); return (st >> 8) & 0xffffu; }    // the top 16 of the low 24 bits
__device__ __forceinline__ uint32_t sr_bf16(float v, uint32_t& st) { return (__float_as_uint(v) + sr_next16(st)) >> 16; }
// two values per step: 12 random bits each (bits 12..23 and 0..11 of the state: both halves run through all 4096 values once per 4096 steps),
// placed in bits 4..15 below the kept mantissa -- the rounding threshold is quantised to 1/4096 ulp, the expectation error is <= 2^-13 ulp
__device__ __forceinline__ uint32_t sr_pk_bf16(float lo, float hi, uint32_t& st) {
  st = sr_step(st);
  const uint32_t a = (__builtin_amdgcn_ubfe(st, 12, 12) << 4) + __float_as_uint(lo), b = ((st & 0xfffu) << 4) + __float_as_uint(hi);
  return __builtin_amdgcn_perm(b, a, 0x07060302u);              // {b[31:16], a[31:16]}
}
static inline int frost_sr_enabled() { static const int on = getenv("FROST_SR") ? atoi(getenv("FROST_SR")) : 1; return on; }

// The k x k int8 taps of channel `ch` (row pitch cpad): every load unconditional from a clamped channel, the lanes past the last channel zeroed afterwards.
// (`chok ? w[..] : 0` per tap puts every load into its own exec-masked block with its wait behind it: 25 memory latencies in a row -- ~17 us -- at the head of
// every depthwise workgroup, seen in the ISA as global_load_sbyte / s_waitcnt vmcnt(0) pairs.)
template <int KK> __device__ __forceinline__ void load_taps_i8(const int8_t* __restrict__ wq, int cpad, int ch, bool chok, int8_t (&t)[KK]) {
  const int chc = chok ? ch : 0;
#pragma unroll
  for (int i = 0; i < KK; ++i) t[i] = wq[i * cpad + chc];
#pragma unroll
  for (int i = 0; i < KK; ++i) t[i] = chok ? t[i] : (int8_t)0;
}
// ---- ordered float atomics ---------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  if (v >= 0.0f) atomicMin((int*)addr, __float_as_int(v));
  else atomicMax((unsigned int*)addr, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.0f) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

// ---- wave / block reductions -------------------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// observer EMA + qparams, shared by every finalize-type kernel (Appendix B of SURVEY.md).
// cur_lo/cur_hi: min/max of the current tensor. Writes all qrecord fields. One thread.
// the record fields the update reads, fetched ahead of the dependent chain that produces cur_lo / cur_hi (a finalize tail is a chain of global round trips:
// ticket -> statistics -> this record; loaded early the record travels with the statistics)
struct ObsPre { float mn, mx, en, qmax; };
__device__ __forceinline__ ObsPre observer_prefetch(const float* q) { ObsPre p; p.mn = q[FROST_Q_MIN]; p.mx = q[FROST_Q_MAX]; p.en = q[FROST_Q_OBS_EN]; p.qmax = q[FROST_Q_QMAX]; return p; }
// AG = true: every record / coefficient word is written with an agent-scope store and read back with an agent-scope load, so that OTHER workgroups of the SAME launch
// may consume them after a device-wide barrier (k_sq_fwd, csrc/frost_block.hip) -- per-location coherence at agent scope, no L2 write-back / invalidate on the path
template <bool AG> __device__ __forceinline__ void fin_st(float* p, float v) {
  if (AG) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool AG> __device__ __forceinline__ float fin_ld(const float* p) { return AG ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
template <bool AG = false>
__device__ inline void observer_update_dev(float* q, float cur_lo, float cur_hi, int symmetric, int rule127,
                                           int observe, const ObsPre* pre = nullptr) {
  float mn = pre ? pre->mn : fin_ld<AG>(q + FROST_Q_MIN), mx = pre ? pre->mx : fin_ld<AG>(q + FROST_Q_MAX);
  observe = observe && (__float_as_int(pre ? pre->en : fin_ld<AG>(q + FROST_Q_OBS_EN)) != 0);       // the site's own observer_enabled buffer (device resident)
  float fscale = 1.0f; int fzp = 0; bool have = false;
  const int qhi = pre ? ((pre->qmax > 0.0f) ? (int)pre->qmax : 255) : q_hi(q);
  if (observe) {
    if (isinf(mn) && isinf(mx) && mn > 0.0f && mx < 0.0f) { mn = cur_lo; mx = cur_hi; }
    else { mn = mn + FROST_OBS_C * (cur_lo - mn); mx = mx + FROST_OBS_C * (cur_hi - mx); }
    fin_st<AG>(q + FROST_Q_MIN, mn); fin_st<AG>(q + FROST_Q_MAX, mx);
    float scale; int zp = 0;
    if (mx < mn) { scale = 1.0f; zp = 0; }
    else {
      float mn_neg = fminf(mn, 0.0f), mx_pos = fmaxf(mx, 0.0f);
      if (symmetric) {
        if (rule127) scale = fmaxf(-mn_neg / 128.0f, mx_pos / 127.0f);
        else scale = fmaxf(-mn_neg, mx_pos) / 127.5f;
        scale = fmaxf(scale, FROST_F32_EPS);
      } else {
        scale = (mx_pos - mn_neg) / (float)qhi;          // (qmax - qmin): 255, or 127 with reduce_range
        scale = fmaxf(scale, FROST_F32_EPS);
        zp = 0 - (int)rintf(mn_neg / scale);
        zp = min(max(zp, 0), qhi);
      }
    }
    fin_st<AG>(q + FROST_Q_SCALE, scale); fin_st<AG>(q + FROST_Q_ZP, __int_as_float(zp));
    fscale = scale; fzp = zp; have = true;
  }
  const float scale = have ? fscale : fin_ld<AG>(q + FROST_Q_SCALE); const int zp = have ? fzp : __float_as_int(fin_ld<AG>(q + FROST_Q_ZP));
  float inv = 1.0f / scale;
  fin_st<AG>(q + FROST_Q_INV, inv);
  int lo = symmetric ? -128 : 0, hi = symmetric ? 127 : qhi;
  int ilo = fq_index(cur_lo, inv, zp, lo, hi), ihi = fq_index(cur_hi, inv, zp, lo, hi);
  fin_st<AG>(q + FROST_Q_FQMIN, (float)(ilo - zp) * scale);
  fin_st<AG>(q + FROST_Q_FQMAX, (float)(ihi - zp) * scale);
}

// ---- replicated statistics tables (round 6) -------------------------------------------------------------------------------
// A layer's integer statistics live in FROST_STATS_NC identical tables ([s1 | s2 | min | max] x cpad, 24 bytes per channel each): a workgroup flushes into table
// stats_copy() picks from its index, the finalize adds the tables up (sums are exact integers, min / max exact: the result does not depend on who wrote where).
// Hundreds of workgroups finishing together hit ONE address per channel and value before: same-address device atomics are serialised at the memory side
// (~45 ns each; the step with k_pw's flush atomics compiled out was 0.40 ms shorter, profiles/r05_fusion_ab.txt).
#define FROST_STATS_NC FROST_STATS_TABLES
__device__ __forceinline__ uint8_t* stats_copy(uint8_t* base, int cpad) {
  const unsigned b = blockIdx.x;
  return base + (size_t)(((b >> 3) ^ b) & (FROST_STATS_NC - 1)) * (size_t)cpad * 24;          // (index bits mixed: launches that deal channel ranges out by index % 2 / 4 still spread)
}

// ---- spread S1 / S2 rows of the backward reduce passes (round 6) -------------------------------------------------------------
// The reduce passes add their per-workgroup sums (S1 = sum gy, S2 = sum gy * xhat) into the layer's coefficient rows with float atomics: hundreds of workgroups, ONE
// address per channel and sum.  The rows now exist FROST_S12_NC times -- copy 0 = rows FROST_COEF_S1 / _S2, copy k > 0 = rows FROST_COEF_ROWS + 2 (k - 1) + {0, 1} of the
// same table (a coefficient table is FROST_COEF_ROWS_ALLOC rows) -- a workgroup adds into the copy s12_dst() picks from its index, every reader (the dc kernels' prologues,
// the parameter-gradient finalize) takes s12_sum().  The forward finalize zeroes every copy.  Kernels that WRITE totals (the fp32-gradient mode) use copy 0, the rest stays 0.
#define FROST_S12_NC ((FROST_COEF_ROWS_ALLOC - FROST_COEF_ROWS) / 2 + 1)
__host__ __device__ __forceinline__ int s12_row(int which, int k) { return k == 0 ? (which ? FROST_COEF_S2 : FROST_COEF_S1) : FROST_COEF_ROWS + 2 * (k - 1) + which; }
__device__ __forceinline__ float* s12_dst(float* coef, int cpad, int which) {
  const unsigned b = blockIdx.x;
  return coef + (size_t)s12_row(which, (int)(((b >> 3) ^ b) & (FROST_S12_NC - 1))) * cpad;
}
__device__ __forceinline__ float s12_sum(const float* coef, int cpad, int which, int c) {
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < FROST_S12_NC; ++k) s += coef[(size_t)s12_row(which, k) * cpad + c];
  return s;
}
__device__ __forceinline__ float4 s12_sum4(const float* coef, int cpad, int which, int c) {          // c a multiple of 4
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < FROST_S12_NC; ++k) { const float4 v = *(const float4*)(coef + (size_t)s12_row(which, k) * cpad + c); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  return s;
}

// ---- spread raw weight-gradient sums (round 6) ------------------------------------------------------------------------------
// dL/d(fake-quantised weight) is accumulated with float atomics into the layer's `dwq` buffer [cout][cin/g * k * k]; the fused pointwise backward and the depthwise backward
// kernels are persistent launches whose 500 - 1000 workgroups all flush the SAME few thousand addresses at the end.  Layers of at most FROST_DWQ_SPREAD_MAX weights -- every
// layer those kernels serve -- keep FROST_DWQ_NC copies of the buffer (copy k at dwq + k * dwq_stride(numel); the caller sizes and zeroes all of them), a workgroup adds into
// the copy picked from its index, the parameter-gradient finalize (its only reader) adds the copies up.  Producers that do not spread (split-K GEMMs of the wide layers, the
// fp32-gradient mode) write copy 0; larger layers have ONE copy.  The rule is a function of the layer's weight count alone, so both sides evaluate it from their own arguments.
__host__ __device__ __forceinline__ int64_t dwq_stride(int64_t numel) { return (numel + 63) & ~(int64_t)63; }
__device__ __forceinline__ float* dwq_dst(float* dwq, int64_t numel) {
  const unsigned b = blockIdx.x;
  return (numel <= FROST_DWQ_SPREAD_MAX) ? dwq + (int64_t)(((b >> 3) ^ b) & (FROST_DWQ_NC - 1)) * dwq_stride(numel) : dwq;
}
__device__ __forceinline__ float* dwq_dst_k(float* dwq, int64_t numel, int k) {          // ... with the copy chosen by the caller (split-K kernels: the pixel split)
  return (numel <= FROST_DWQ_SPREAD_MAX) ? dwq + (int64_t)(k & (FROST_DWQ_NC - 1)) * dwq_stride(numel) : dwq;
}
__device__ __forceinline__ float dwq_sum(const float* dwq, int64_t numel, int64_t idx) {
  float s = dwq[idx];
  if (numel <= FROST_DWQ_SPREAD_MAX) {
    const int64_t st = dwq_stride(numel);
#pragma unroll
    for (int k = 1; k < FROST_DWQ_NC; ++k) s += dwq[idx + k * st];
  }
  return s;
}

// ---- conv finalize (shared by k_conv_finalize and the statistics kernels' last-workgroup tail) ------------------------------
// Turns the integer statistics of one layer into BN coefficients, running-stat updates and the activation qrecord.  Runs in ONE
// workgroup of `nthr` threads.  The statistics were produced by device-scope atomics of (possibly) other workgroups: they are read
// with agent-scope loads.  sh: >= 2 * (nthr / 64) floats of shared memory.
template <bool AG = false>
__device__ inline void conv_finalize_dev(const uint8_t* stats, int64_t count, int cout, int cpad, const float* qx, const float* qw, const float* wscale,
                                         const float* gamma, const float* beta, float* rmean, float* rvar, int64_t* nbt, int training,
                                         int relu, int observe, int have_stats, float* coef, float* qy, int tid, int nthr, float* sh,
                                         const float* cat_qb = nullptr, float* cat_qy = nullptr) {
  const int64_t* s1 = (const int64_t*)stats; const uint64_t* s2 = (const uint64_t*)(s1 + cpad);
  const int32_t* mnp = (const int32_t*)(s2 + cpad); const int32_t* mxp = mnp + cpad;
  const size_t cst8 = (size_t)cpad * 3, cst4 = (size_t)cpad * 6;          // stride between the FROST_STATS_NC replicated tables, in 8- / 4-byte words
  const float sx = qx[FROST_Q_SCALE], sw0 = qw[FROST_Q_SCALE];
  ObsPre pre_y = {}, pre_cat = {}; float cat_b_lo = 0.0f, cat_b_hi = 0.0f;
  if (tid == 0 && have_stats) {      // the records thread 0 updates at the end: requested now, they arrive with the statistics
    pre_y = observer_prefetch(qy);
    if (cat_qy) { pre_cat = observer_prefetch(cat_qy); cat_b_lo = cat_qb[FROST_Q_FQMIN]; cat_b_hi = cat_qb[FROST_Q_FQMAX]; }
  }
  float lo = INFINITY, hi = -INFINITY;
  // FIN_U channels per thread and round: every load of a round (the other workgroups' statistics through agent-scope loads, the BatchNorm parameters and running
  // statistics) is issued before the first fp64 chain starts, at clamped addresses and without per-channel branches -- one memory round trip per round instead of
  // one per channel (1728 channels on 256 threads were seven dependent round trips of ~2 us on the critical path of the layer; now two)
  constexpr int FIN_U = 2;          // (round 6: 4 -> 2 with the four replicated tables -- 2 channels x 4 tables x 4 values in flight per thread; four channels went to scratch memory)
  for (int c0 = tid; c0 < cpad; c0 += FIN_U * nthr) {
    float f_sw[FIN_U], f_rv[FIN_U], f_rm[FIN_U], f_g[FIN_U], f_b[FIN_U]; int64_t f_v1[FIN_U]; uint64_t f_v2[FIN_U]; int32_t f_mn[FIN_U], f_mx[FIN_U];
#pragma unroll
    for (int u = 0; u < FIN_U; ++u) {
      const int cc = min(c0 + u * nthr, cout - 1);
      f_rv[u] = rvar[cc]; f_rm[u] = rmean[cc]; f_g[u] = gamma[cc]; f_b[u] = beta[cc];
    }
    if (wscale) {
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) f_sw[u] = wscale[min(c0 + u * nthr, cout - 1)];
    } else {
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) f_sw[u] = sw0;
    }
    if (training) {
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        const int cc = min(c0 + u * nthr, cout - 1);
        int64_t a = 0; uint64_t b = 0;
#pragma unroll
        for (int k = 0; k < FROST_STATS_NC; ++k) {          // the producers spread their flushes over FROST_STATS_NC tables (stats_copy): every load of the round in flight together
          a += __hip_atomic_load(&s1[cc + k * cst8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b += __hip_atomic_load(&s2[cc + k * cst8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        f_v1[u] = a; f_v2[u] = b;
      }
    } else {
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) { f_v1[u] = 0; f_v2[u] = 0; }
    }
    if (have_stats) {
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        const int cc = min(c0 + u * nthr, cout - 1);
        int32_t a = INT32_MAX, b = INT32_MIN;
#pragma unroll
        for (int k = 0; k < FROST_STATS_NC; ++k) {
          a = min(a, __hip_atomic_load(&mnp[cc + k * cst4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); b = max(b, __hip_atomic_load(&mxp[cc + k * cst4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        f_mn[u] = a; f_mx[u] = b;
      }
    } else {
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) { f_mn[u] = 0; f_mx[u] = 0; }
    }
#pragma unroll
    for (int u = 0; u < FIN_U; ++u) {
      const int c = c0 + u * nthr;
      if (c >= cpad) break;
      float A = 0, B = 0, M = 0, R = 0, K1 = 0, VF = 0;
      if (c < cout) {
        const float sw = f_sw[u];                              // per-output-channel weight scale (per-tensor mode: all equal)
        const float sigr = sqrtf(f_rv[u] + FROST_BN_EPS);
        const float sf = f_g[u] / sigr;
        const double alpha = (double)sx * (double)sw / (double)sf;       // c0 = acc * alpha
        double mean_acc, mu, v;
        if (training) {
          mean_acc = (double)f_v1[u] / (double)count;
          double var_acc = (double)f_v2[u] / (double)count - mean_acc * mean_acc;
          if (var_acc < 0) var_acc = 0;
          mu = mean_acc * alpha; v = var_acc * alpha * alpha;
          const double unb = (count > 1) ? v * (double)count / (double)(count - 1) : v;
          rmean[c] = (float)((1.0 - (double)FROST_BN_MOM) * (double)f_rm[u] + (double)FROST_BN_MOM * mu);
          rvar[c] = (float)((1.0 - (double)FROST_BN_MOM) * (double)f_rv[u] + (double)FROST_BN_MOM * unb);
        } else {
          mu = (double)f_rm[u]; v = (double)f_rv[u]; mean_acc = mu / alpha;
        }
        const double invstd = 1.0 / sqrt(v + (double)FROST_BN_EPS);
        const double a = (double)f_g[u] * invstd * alpha;
        A = (float)a; B = (float)((double)f_b[u] - a * mean_acc);
        M = (float)mean_acc; R = (float)(alpha * invstd);
        K1 = (float)(invstd * (double)sigr);                         // gamma*invstd/sf
        VF = (float)(v / (v + (double)FROST_BN_EPS));
        if (have_stats) {
          float ya = fmaf(A, (float)f_mn[u], B), yb = fmaf(A, (float)f_mx[u], B);
          if (relu) { ya = fmaxf(ya, 0.0f); yb = fmaxf(yb, 0.0f); }
          lo = fminf(lo, fminf(ya, yb)); hi = fmaxf(hi, fmaxf(ya, yb));
        }
      }
      fin_st<AG>(coef + FROST_COEF_A * cpad + c, A); fin_st<AG>(coef + FROST_COEF_B * cpad + c, B); fin_st<AG>(coef + FROST_COEF_M * cpad + c, M);
      fin_st<AG>(coef + FROST_COEF_R * cpad + c, R); fin_st<AG>(coef + FROST_COEF_K1 * cpad + c, K1); fin_st<AG>(coef + FROST_COEF_VFRAC * cpad + c, VF);
#pragma unroll
      for (int k = 0; k < FROST_S12_NC; ++k) { fin_st<AG>(coef + (size_t)s12_row(0, k) * cpad + c, 0.0f); fin_st<AG>(coef + (size_t)s12_row(1, k) * cpad + c, 0.0f); }
    }
  }
  const int nw = nthr >> 6;
  lo = wave_min(lo); hi = wave_max(hi);
  __syncthreads();
  if ((tid & 63) == 0) { sh[tid >> 6] = lo; sh[nw + (tid >> 6)] = hi; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < nw; ++i) { lo = fminf(lo, sh[i]); hi = fmaxf(hi, sh[nw + i]); }
    if (training && nbt) *nbt += 1;
    if (have_stats) observer_update_dev<AG>(qy, lo, hi, 0, 0, observe, &pre_y);
    else fin_st<AG>(qy + FROST_Q_INV, 1.0f / fin_ld<AG>(qy + FROST_Q_SCALE));
    // squeeze_conv of a Frost bottleneck: the cat's FakeQuantize sees min / max of the fake-quantised halves (k_cat_observe's expression)
    if (cat_qy) {
      if (have_stats) observer_update_dev<AG>(cat_qy, fminf(fin_ld<AG>(qy + FROST_Q_FQMIN), cat_b_lo), fmaxf(fin_ld<AG>(qy + FROST_Q_FQMAX), cat_b_hi), 0, 0, observe, &pre_cat);
      else observer_update_dev<AG>(cat_qy, fminf(fin_ld<AG>(qy + FROST_Q_FQMIN), cat_qb[FROST_Q_FQMIN]), fmaxf(fin_ld<AG>(qy + FROST_Q_FQMAX), cat_qb[FROST_Q_FQMAX]), 0, 0, observe);
    }
  }
}

// Memory-model note.  The hand-off below (statistics written ONLY by agent-scope atomics, read back ONLY by agent-scope loads, the ticket ordered after
// them by s_waitcnt vmcnt(0)) is what gfx942 / gfx950 guarantee for device-scope atomics, which are performed at the memory side; it is not a
// release / acquire pair in the HIP / LLVM memory model.  This library is built for gfx950 only (__graft_entry__.FLAGS); on any other target the ticket is
// bracketed by agent-scope fences (1.7 us each on the critical path of every layer on MI355X, measured -- hence not there).  tests/test_gpu_round3.py
// runs the finalize stress (tests/devtools/stress_finalize.py) so that a compiler or architecture change that breaks the hand-off fails a test.
#if defined(__gfx950__) || defined(__gfx942__) || !defined(__HIP_DEVICE_COMPILE__)
#define FROST_TICKET_FENCE() ((void)0)
#else
#define FROST_TICKET_FENCE() __threadfence()
#endif
// Last-workgroup-done tail of a statistics kernel: every thread has issued its atomics; returns true in ALL threads of the one workgroup
// that arrives last (its view of the other workgroups' atomics is then complete).  sflag: one int of shared memory.
__device__ __forceinline__ bool last_block_done(uint32_t* counter, unsigned total, int* sflag) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's atomics have been performed
  __syncthreads();
  if (threadIdx.x == 0) {
    // No release / acquire fence (1.7 us each on the critical path of every layer): the statistics are written ONLY by agent-scope atomics and
    // read back ONLY by agent-scope (sc1) loads -- "agent atomics on both sides" needs no cache maintenance (MI355X guide, cross-XCD hand-off
    // forms); the vmcnt(0) above orders this workgroup's atomics before its ticket.
    FROST_TICKET_FENCE();
    const unsigned t = atomicAdd(counter, 1u);
    *sflag = (t == total - 1u) ? 1 : 0;
    if (t == total - 1u) { FROST_TICKET_FENCE(); __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }    // re-armed for the next step
  }
  __syncthreads();
  return *sflag != 0;
}

// Two-level ticket for launches of many workgroups that finish together.  Same-address device-scope atomics are serialised at the memory side
// (~45 ns each, measured: 46 us for the 1024 equal workgroups of an element-wise pass), so the tail of a last-workgroup-done kernel grows with its
// grid.  Here workgroup b takes a ticket of sub-counter 1 + (b mod 32) (32 addresses: 32 chains run in parallel), and the workgroup that completes a
// sub-counter takes one of the main counter: 2 x ~total/32 serialised atomics instead of `total`.  counter: FROST_TICKET_WORDS zeroed uint32, left
// zeroed.  Visibility argument as for last_block_done: every atomic is performed at the coherence point before its workgroup's ticket.
__device__ __forceinline__ bool last_block_done2(uint32_t* counter, unsigned total, int* sflag) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int last = 0;
#ifdef FROST_TICKET_SINGLE
    if (true) {
#else
    if (total <= 64u) {
#endif
      FROST_TICKET_FENCE();
      const unsigned t = atomicAdd(counter, 1u);
      if (t == total - 1u) { FROST_TICKET_FENCE(); __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = 1; }
    } else {
      const unsigned b = blockIdx.x + blockIdx.y * gridDim.x, s = b & 31u;
      const unsigned mine = (total - s + 31u) >> 5;                   // workgroups with linear index = s (mod 32)
      FROST_TICKET_FENCE();
      const unsigned t = atomicAdd(counter + 1 + s, 1u);
      if (t == mine - 1u) {
        __hip_atomic_store(counter + 1 + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t2 = atomicAdd(counter, 1u);
        if (t2 == 31u) { FROST_TICKET_FENCE(); __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = 1; }
      }
    }
    *sflag = last;
  }
  __syncthreads();
  return *sflag != 0;
}

// Range (min / max) of a launch WITHOUT same-address float atomics: every workgroup stores its pair into its own slot of `part` (agent-scope stores: written
// through, like the statistics atomics), the last workgroup to arrive (ticket) folds the gridDim.x pairs.  The same-address atomics this replaces serialise at
// the memory side (~45 ns each): a range pass of 150-500 workgroups spent 15-40 us in them for 2-5 us of work.  Returns true in thread 0 of the last workgroup
// with the full range in lo / hi.  part: 2 * gridDim.x floats; sh: 8 floats + one int of shared memory.
#define FROST_MM_SLOTS 1024
__device__ __forceinline__ bool range_fold_last(float& lo, float& hi, float* part, uint32_t* ticket, float* sh, int* sflag) {
  lo = wave_min(lo); hi = wave_max(hi);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[w] = lo; sh[4 + w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < nw; ++i) { lo = fminf(lo, sh[i]); hi = fmaxf(hi, sh[4 + i]); }
    __hip_atomic_store(part + 2 * blockIdx.x, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + 2 * blockIdx.x + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!last_block_done2(ticket, gridDim.x, sflag)) return false;
  lo = INFINITY; hi = -INFINITY;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) {
    lo = fminf(lo, __hip_atomic_load(part + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    hi = fmaxf(hi, __hip_atomic_load(part + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  lo = wave_min(lo); hi = wave_max(hi);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sh[w] = lo; sh[4 + w] = hi; }
  __syncthreads();
  if (threadIdx.x != 0) return false;
  for (int i = 1; i < nw; ++i) { lo = fminf(lo, sh[i]); hi = fmaxf(hi, sh[4 + i]); }
  return true;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
