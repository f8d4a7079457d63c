// Pointwise (1x1) convolution on fake-quantised operands: int8 MFMA forward (stats / emit passes),
// recompute-based backward (reduce / dc passes) and the bf16 MFMA dgrad, all on ONE skeleton:
//
//   D[chan][pix] = sum_K Wpack[chan][K] * T[pix][K]
//
// T is a pixel-major tensor (NHWC activation bytes, or the bf16 dc tensor for dgrad) whose 128-pixel tile is
// staged ONCE through LDS with fully coalesced loads and stays resident while the workgroup walks every
// output-channel group (activation-stationary: the big operand is read from HBM exactly once per pass);
// the small operand (packed weights, L2 resident) is fetched straight into MFMA A-fragments with 1 KiB
// wave-loads.  Both element types use a 64-byte K-step (16x16x64 i8 / 16x16x32 bf16), so staging, LDS layout
// and fragment addressing are shared.  D's lane layout (lane = pixel column, 4 consecutive channels per
// lane) makes every epilogue access (coefficients, gradients, packed int8 / bf16 stores) a 4-channel vector.
#include "frost_common.h"

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

enum { M_STATS = 0, M_EMIT = 1, M_BRED = 2, M_BDC = 3, M_DGRAD = 4 };

struct PwP {
  const uint8_t* T; int64_t npix; int rowbytes;
  int cout, cpad;                 // output channels of this GEMM
  const uint8_t* wpack; int KS;   // total 64-byte K steps
  int kstr, kc_bytes, nchunks;    // LDS row stride, chunk width (bytes), chunks per row
  const int32_t* wsum; const float* qx; const float* qy; const float* qw; float* coef;
  uint8_t* stats; int relu;
  int8_t* y; const uint16_t* gout; uint16_t* dc; uint16_t* dx; int accumulate;
  int ngroups, mi_eff; int64_t ntiles; float inv_count;
};

#define BP 128
#define MI 4

template <int MODE> struct ModeTraits { static constexpr bool bf16 = (MODE == M_DGRAD); };

// DPP row reductions (16-lane rows = the 16 pixel columns of an MFMA tile): inclusive scan with row_shr 1,2,4,8;
// lane 15 of every row ends up with the row total.  Pure VALU -- no LDS traffic (unlike __shfl_xor -> ds_bpermute).
template <int CTRL> __device__ __forceinline__ int dpp_i(int v, int identity) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ int row_sum_i(int v) {
  v += dpp_i<0x111>(v, 0); v += dpp_i<0x112>(v, 0); v += dpp_i<0x114>(v, 0); v += dpp_i<0x118>(v, 0); return v;
}
__device__ __forceinline__ int row_min_i(int v) {
  v = min(v, dpp_i<0x111>(v, INT32_MAX)); v = min(v, dpp_i<0x112>(v, INT32_MAX)); v = min(v, dpp_i<0x114>(v, INT32_MAX)); v = min(v, dpp_i<0x118>(v, INT32_MAX)); return v;
}
__device__ __forceinline__ int row_max_i(int v) {
  v = max(v, dpp_i<0x111>(v, INT32_MIN)); v = max(v, dpp_i<0x112>(v, INT32_MIN)); v = max(v, dpp_i<0x114>(v, INT32_MIN)); v = max(v, dpp_i<0x118>(v, INT32_MIN)); return v;
}
__device__ __forceinline__ float row_sum_f(float v) {
  v += __int_as_float(dpp_i<0x111>(__float_as_int(v), 0)); v += __int_as_float(dpp_i<0x112>(__float_as_int(v), 0));
  v += __int_as_float(dpp_i<0x114>(__float_as_int(v), 0)); v += __int_as_float(dpp_i<0x118>(__float_as_int(v), 0)); return v;
}
__device__ __forceinline__ long long row_sum_ll(long long v) {
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const int lo = (int)(unsigned)(v & 0xffffffffll), hi = (int)(v >> 32);
    int slo, shi;
    if (step == 0) { slo = dpp_i<0x111>(lo, 0); shi = dpp_i<0x111>(hi, 0); }
    else if (step == 1) { slo = dpp_i<0x112>(lo, 0); shi = dpp_i<0x112>(hi, 0); }
    else if (step == 2) { slo = dpp_i<0x114>(lo, 0); shi = dpp_i<0x114>(hi, 0); }
    else { slo = dpp_i<0x118>(lo, 0); shi = dpp_i<0x118>(hi, 0); }
    v += (long long)(((unsigned long long)(unsigned)shi << 32) | (unsigned)slo);
  }
  return v;
}

template <int MODE, int WP>
__global__ __launch_bounds__(512, 4) void k_pw(const PwP p) {
  constexpr int WC = 8 / WP;          // waves along channels
  constexpr int NT = 8 / WP;          // 16-pixel tiles per wave
  constexpr bool BF = ModeTraits<MODE>::bf16;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* xs = smem;
  const int xs_bytes = BP * p.kstr + 64;
  // per-channel LDS accumulators (stats / backward-reduce modes)
  int64_t* l_s1 = (int64_t*)(smem + xs_bytes);
  unsigned long long* l_s2 = (unsigned long long*)(l_s1 + p.cpad);
  int* l_mn = (int*)(l_s2 + p.cpad);
  int* l_mx = l_mn + p.cpad;
  float* l_f1 = (float*)(smem + xs_bytes);
  float* l_f2 = l_f1 + p.cpad;

  const int tid = threadIdx.x;
  const int lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = w / WC, wc = w % WC;
  const int CT = p.cpad >> 4;

  if (MODE == M_STATS) {
    for (int c = tid; c < p.cpad; c += 512) { l_s1[c] = 0; l_s2[c] = 0; l_mn[c] = INT32_MAX; l_mx[c] = INT32_MIN; }
  } else if (MODE == M_BRED) {
    for (int c = tid; c < p.cpad; c += 512) { l_f1[c] = 0.0f; l_f2[c] = 0.0f; }
  }

  int zpx = 0; float sw = 1.0f; float y_inv = 1.0f; int y_zp = 0; float y_zpf = 0.0f;
  if (!BF) zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  if (MODE == M_DGRAD) sw = p.qw[FROST_Q_SCALE];
  if (MODE == M_EMIT || MODE == M_BRED || MODE == M_BDC) { y_inv = 1.0f / p.qy[FROST_Q_SCALE]; y_zp = __float_as_int(p.qy[FROST_Q_ZP]); y_zpf = (float)y_zp; }

  for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int64_t p0 = tile * BP;
    for (int cg = 0; cg < p.ngroups; ++cg) {
      const int ct0 = (cg * WC + wc) * p.mi_eff;
      int mi_n = CT - ct0; mi_n = mi_n < 0 ? 0 : (mi_n > p.mi_eff ? p.mi_eff : mi_n);
      v4i acci[MI][NT]; v4f accf[MI][NT];
#pragma unroll
      for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) { acci[m][t] = (v4i){0, 0, 0, 0}; accf[m][t] = (v4f){0.f, 0.f, 0.f, 0.f}; }

      for (int ch = 0; ch < p.nchunks; ++ch) {
        const int kc0 = ch * p.kc_bytes;
        int kcw = p.rowbytes - kc0; if (kcw > p.kc_bytes) kcw = p.kc_bytes;
        const int kcw_pad = (kcw + 63) & ~63;
        if (p.nchunks > 1 || cg == 0) {
          __syncthreads();
          if (((p.rowbytes | kc0) & 15) == 0) {
            const int U = kcw_pad >> 4; const int total = BP * U;
            for (int u = tid; u < total; u += 512) {
              int row = u / U; int col = (u - row * U) << 4; int64_t pix = p0 + row;
              uint4 v = make_uint4(0, 0, 0, 0);
              if (pix < p.npix && col < kcw) v = *(const uint4*)(p.T + pix * p.rowbytes + kc0 + col);
              *(uint4*)(xs + row * p.kstr + col) = v;
            }
          } else {
            const int U = kcw_pad >> 3; const int total = BP * U;
            for (int u = tid; u < total; u += 512) {
              int row = u / U; int col = (u - row * U) << 3; int64_t pix = p0 + row;
              uint2 v = make_uint2(0, 0);
              if (pix < p.npix && col < kcw) v = *(const uint2*)(p.T + pix * p.rowbytes + kc0 + col);
              *(uint2*)(xs + row * p.kstr + col) = v;
            }
          }
          __syncthreads();
        }
        if (mi_n > 0) {
          const int ks_n = kcw_pad >> 6; const int ks0 = kc0 >> 6;
          for (int ks = 0; ks < ks_n; ++ks) {
            v4i bfr[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
              bfr[t] = *(const v4i*)(xs + ((wp * NT + t) * 16 + j) * p.kstr + ks * 64 + g * 16);
            v4i afr[MI];
#pragma unroll
            for (int m = 0; m < MI; ++m)
              if (m < mi_n) afr[m] = *(const v4i*)(p.wpack + ((((int64_t)(ct0 + m) * p.KS + ks0 + ks) * 64 + lane) << 4));
#pragma unroll
            for (int m = 0; m < MI; ++m) {
              if (m < mi_n) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                  if (BF) accf[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[m]), __builtin_bit_cast(v8bf, bfr[t]), accf[m][t], 0, 0, 0);
                  else acci[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[m], bfr[t], acci[m][t], 0, 0, 0);
                }
              }
            }
          }
        }
      }

      // ------------------------------------------------------------------------------- epilogue
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        if (m >= mi_n) continue;
        const int ch0 = (ct0 + m) * 16 + 4 * g;          // 4 consecutive channels of this lane
        const bool chok = ch0 < p.cout;
        if (MODE == M_DGRAD) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int64_t pix = p0 + (wp * NT + t) * 16 + j;
            if (pix < p.npix && chok) {
              uint16_t* dst = p.dx + pix * p.cout + ch0;
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = accf[m][t][r] * sw;
              if (p.accumulate) { uint2 o = *(const uint2*)dst; v[0] += bf2f(o.x & 0xffff); v[1] += bf2f(o.x >> 16); v[2] += bf2f(o.y & 0xffff); v[3] += bf2f(o.y >> 16); }
              uint2 o; o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
              *(uint2*)dst = o;
            }
          }
          continue;
        }
        const int4 ws4 = *(const int4*)(p.wsum + ch0);
        const int corr[4] = {zpx * ws4.x, zpx * ws4.y, zpx * ws4.z, zpx * ws4.w};
        if (MODE == M_STATS) {
          int s1[4] = {0, 0, 0, 0}; long long s2[4] = {0, 0, 0, 0};
          int mn[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX}, mx[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const bool valid = (p0 + (wp * NT + t) * 16 + j) < p.npix;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int v = acci[m][t][r] - corr[r];
              if (valid) { s1[r] += v; s2[r] += (long long)v * v; mn[r] = min(mn[r], v); mx[r] = max(mx[r], v); }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = row_sum_i(s1[r]); const long long b = row_sum_ll(s2[r]);
            const int c = row_min_i(mn[r]), d = row_max_i(mx[r]);
            if (j == 15 && chok) {
              atomicAdd((unsigned long long*)&l_s1[ch0 + r], (unsigned long long)(long long)a);
              atomicAdd(&l_s2[ch0 + r], (unsigned long long)b);
              atomicMin(&l_mn[ch0 + r], c); atomicMax(&l_mx[ch0 + r], d);
            }
          }
          continue;
        }
        const float4 A4 = *(const float4*)(p.coef + FROST_COEF_A * p.cpad + ch0);
        const float4 B4 = *(const float4*)(p.coef + FROST_COEF_B * p.cpad + ch0);
        const float A[4] = {A4.x, A4.y, A4.z, A4.w}, B[4] = {B4.x, B4.y, B4.z, B4.w};
        if (MODE == M_EMIT) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int64_t pix = p0 + (wp * NT + t) * 16 + j;
            uint32_t packed = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float yv = fmaf(A[r], (float)(acci[m][t][r] - corr[r]), B[r]);
              if (p.relu) yv = fmaxf(yv, 0.0f);
              // q = clamp(rint(y*inv)+zp, 0, 255): v_cvt_pk_u8_f32 saturates to [0,255] and inserts the byte in one op
              packed = __builtin_amdgcn_cvt_pk_u8_f32(rintf(yv * y_inv) + y_zpf, r, packed);
            }
            if (pix < p.npix && chok) *(uint32_t*)(p.y + pix * p.cout + ch0) = packed ^ 0x80808080u;
          }
          continue;
        }
        // backward modes
        const float4 M4 = *(const float4*)(p.coef + FROST_COEF_M * p.cpad + ch0);
        const float4 R4 = *(const float4*)(p.coef + FROST_COEF_R * p.cpad + ch0);
        const float Mv[4] = {M4.x, M4.y, M4.z, M4.w}, Rv[4] = {R4.x, R4.y, R4.z, R4.w};
        float K1[4] = {0, 0, 0, 0}, S1[4] = {0, 0, 0, 0}, S2[4] = {0, 0, 0, 0};
        if (MODE == M_BDC) {
          const float4 k4 = *(const float4*)(p.coef + FROST_COEF_K1 * p.cpad + ch0);
          const float4 a4 = *(const float4*)(p.coef + FROST_COEF_S1 * p.cpad + ch0);
          const float4 b4 = *(const float4*)(p.coef + FROST_COEF_S2 * p.cpad + ch0);
          K1[0] = k4.x; K1[1] = k4.y; K1[2] = k4.z; K1[3] = k4.w;
          S1[0] = a4.x * p.inv_count; S1[1] = a4.y * p.inv_count; S1[2] = a4.z * p.inv_count; S1[3] = a4.w * p.inv_count;
          S2[0] = b4.x * p.inv_count; S2[1] = b4.y * p.inv_count; S2[2] = b4.z * p.inv_count; S2[3] = b4.w * p.inv_count;
        }
        float r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int64_t pix = p0 + (wp * NT + t) * 16 + j;
          const bool valid = pix < p.npix && chok;
          uint2 gv = make_uint2(0, 0);
          if (valid) gv = *(const uint2*)(p.gout + pix * p.cout + ch0);
          const float gq[4] = {bf2f(gv.x & 0xffff), bf2f(gv.x >> 16), bf2f(gv.y & 0xffff), bf2f(gv.y >> 16)};
          float dcv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float af = (float)(acci[m][t][r] - corr[r]);
            float yv = fmaf(A[r], af, B[r]);
            bool alive = true;
            if (p.relu) { alive = yv > 0.0f; yv = fmaxf(yv, 0.0f); }
            bool inr; fq_index(yv, y_inv, y_zp, 0, 255, &inr);
            const float gy = (alive && inr && valid) ? gq[r] : 0.0f;
            const float xhat = (af - Mv[r]) * Rv[r];
            if (MODE == M_BRED) { r1[r] += gy; r2[r] += gy * xhat; }
            else dcv[r] = K1[r] * (gy - S1[r] - xhat * S2[r]);
          }
          if (MODE == M_BDC && valid) {
            uint2 o; o.x = (uint32_t)f2bf(dcv[0]) | ((uint32_t)f2bf(dcv[1]) << 16); o.y = (uint32_t)f2bf(dcv[2]) | ((uint32_t)f2bf(dcv[3]) << 16);
            *(uint2*)(p.dc + pix * p.cout + ch0) = o;
          }
        }
        if (MODE == M_BRED) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = row_sum_f(r1[r]), b = row_sum_f(r2[r]);
            if (j == 15 && chok) { atomicAdd(&l_f1[ch0 + r], a); atomicAdd(&l_f2[ch0 + r], b); }
          }
        }
      }
    }
  }

  if (MODE == M_STATS) {
    __syncthreads();
    int64_t* g_s1 = (int64_t*)p.stats; unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
    int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
    for (int c = tid; c < p.cout; c += 512) {
      atomicAdd((unsigned long long*)&g_s1[c], (unsigned long long)l_s1[c]);
      atomicAdd(&g_s2[c], l_s2[c]);
      atomicMin(&g_mn[c], l_mn[c]); atomicMax(&g_mx[c], l_mx[c]);
    }
  } else if (MODE == M_BRED) {
    __syncthreads();
    for (int c = tid; c < p.cout; c += 512) {
      atomicAdd(p.coef + FROST_COEF_S1 * p.cpad + c, l_f1[c]);
      atomicAdd(p.coef + FROST_COEF_S2 * p.cpad + c, l_f2[c]);
    }
  }
}

template <int MODE, int WP>
static int launch_pw(PwP& p, hipStream_t s) {
  size_t lds = (size_t)BP * p.kstr + 64;
  if (MODE == M_STATS) lds += (size_t)p.cpad * 24;
  if (MODE == M_BRED) lds += (size_t)p.cpad * 8;
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)k_pw<MODE, WP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  FROST_REQUIRE(lds <= 160 * 1024, "pw: LDS budget exceeded");
  int occ = lds <= 78 * 1024 ? 2 : 1;
  int64_t grid = p.ntiles < 256 * occ ? p.ntiles : 256 * occ;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_pw<MODE, WP>), dim3((unsigned)grid), dim3(512), lds, s, p);
  return frost_check_launch("pw");
}

template <int MODE>
static int dispatch_pw(PwP& p, hipStream_t s) {
  const int CT = p.cpad / 16;
  int WPsel = CT <= 4 ? 8 : (CT <= 8 ? 4 : 2);
  int WC = 8 / WPsel;
  p.ngroups = (CT + WC * MI - 1) / (WC * MI);
  p.mi_eff = (CT + p.ngroups * WC - 1) / (p.ngroups * WC);
  if (WPsel == 8) return launch_pw<MODE, 8>(p, s);
  if (WPsel == 4) return launch_pw<MODE, 4>(p, s);
  return launch_pw<MODE, 2>(p, s);
}

static void set_tiling(PwP& p, int64_t npix, int rowbytes) {
  p.npix = npix; p.rowbytes = rowbytes;
  p.ntiles = (npix + BP - 1) / BP;
  p.KS = (rowbytes + 63) / 64;
  if (rowbytes <= 512) { p.nchunks = 1; p.kc_bytes = rowbytes; }
  else { p.kc_bytes = 512; p.nchunks = (rowbytes + 511) / 512; }
  int kpad = ((p.kc_bytes + 63) / 64) * 64;
  p.kstr = kpad + 16;
  p.inv_count = 1.0f / (float)npix;
}

extern "C" int frost_pw_conv_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                                 int64_t npix, int cin, int cout, int mode, void* stats, const float* coef,
                                 const float* qrec_y, int relu, int8_t* y, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 4 == 0, "pw_fwd: cin must be a multiple of 8, cout of 4");
  FROST_REQUIRE(((uintptr_t)x & 15) == 0, "pw_fwd: x must be 16B aligned");
  PwP p = {};
  p.T = (const uint8_t*)x; p.cout = cout; p.cpad = round_up(cout, 16); p.wpack = (const uint8_t*)wq_pack;
  p.wsum = wsum; p.qx = qrec_x; p.qy = qrec_y; p.coef = (float*)coef; p.stats = (uint8_t*)stats; p.relu = relu; p.y = y;
  set_tiling(p, npix, cin);
  if (mode == 0) return dispatch_pw<M_STATS>(p, as_stream(stream));
  return dispatch_pw<M_EMIT>(p, as_stream(stream));
}

extern "C" int frost_pw_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                                 const uint16_t* wt_pack, const float* qrec_w, int64_t npix, int cin, int cout, int pass,
                                 float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc,
                                 uint16_t* dx, int accumulate, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 8 == 0, "pw_bwd: channels must be multiples of 8");
  PwP p = {};
  p.T = (const uint8_t*)x; p.cout = cout; p.cpad = round_up(cout, 16); p.wpack = (const uint8_t*)wq_pack;
  p.wsum = wsum; p.qx = qrec_x; p.qy = qrec_y; p.qw = qrec_w; p.coef = coef; p.relu = relu; p.gout = gout; p.dc = dc;
  set_tiling(p, npix, cin);
  if (pass == 0) return dispatch_pw<M_BRED>(p, as_stream(stream));
  if (pass == 1) return dispatch_pw<M_BDC>(p, as_stream(stream));
  // pass 2: dgrad   dx[pix][cin] (+)= s_w * sum_co dc[pix][co] * wq[co][cin]
  PwP d = {};
  d.T = (const uint8_t*)dc; d.cout = cin; d.cpad = round_up(cin, 16); d.wpack = (const uint8_t*)wt_pack;
  d.qw = qrec_w; d.dx = dx; d.accumulate = accumulate;
  set_tiling(d, npix, cout * 2);
  return dispatch_pw<M_DGRAD>(d, as_stream(stream));
}
