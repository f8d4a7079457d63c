// Reduce and dc passes of the backward of WIDE pointwise ConvBN(ReLU) layers (the expand convs of the 14 x 14 / 7 x 7 bottlenecks, last_layer): the same
// arithmetic as k_pw's BRED / BDC passes (frost_pw.hip; replaces the torch stages frost_pw_conv_bwd pass 0 / 1 cite) in the chunked layout of the block
// kernels (frost_block.hip).
//
// Why a second kernel: k_pw keeps a 128-pixel tile and walks ALL output channels of it; with Cout = 624 .. 1728 the gout / dc rows of a tile (2 * Cout bytes per
// pixel) do not fit LDS, so its epilogue reads gout and writes dc from registers in 8-byte pieces at a 2 * Cout-byte stride (16 + 30 us of the 90 us dc pass of
// 240 -> 1440 @ 7 x 7, profiles/r02_pw_ablation.txt).  Here a workgroup owns a 64-pixel tile and walks 64-channel CHUNKS: the chunk's gout window is 64 x 128
// contiguous bytes per pixel row -> staged through LDS with full-line loads (register prefetch one chunk ahead), dc is written in place over it and leaves
// with full-line stores; the x tile is staged once per workgroup, the chunk's weight fragments come from L2 one chunk ahead, the per-channel coefficient
// rows of the workgroup's chunk range sit in LDS (E / F folded once).
#include "frost_common.h"

struct PwcP {
  const int8_t* x; const float* qx; const int8_t* w; const int32_t* wsum;
  float* coef; const float* qy; const uint16_t* gout; uint16_t* dc; int8_t* y;
  int64_t npix; int cin, c, cpad, kstr, nchunk, csplit, relu, sr; float inv_count;
  uint8_t* stats; FrostFinDesc fin; unsigned fin_total; int ptpw; int64_t ntiles;      // k_pwc_stats: the layer's statistics tables, the finalize folded into the last workgroup, pixel tiles per workgroup
};

__device__ __forceinline__ void pwc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int CTRL>
__device__ __forceinline__ float pwc_dpp_add(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
// sum over the 16 lanes of a row (lanes j = 0..15 of one g): the total lands in lane 15 (row_shr 1, 2, 4, 8 -- k_pw's row_sum_f)
__device__ __forceinline__ float pwc_row_sum(float v) { v = pwc_dpp_add<0x111>(v); v = pwc_dpp_add<0x112>(v); v = pwc_dpp_add<0x114>(v); v = pwc_dpp_add<0x118>(v); return v; }

#define PWC_PX 64
// MODE 0: reduce pass (S1 += gy, S2 += gy * xhat into the coefficient rows); MODE 1: dc pass; MODE 2: forward emit.  KSM: K steps of 64 input channels (exact).
template <int MODE, int KSM>
__global__ __launch_bounds__(256, 4) void k_pwc(const PwcP p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xs = smem;                                   // [64][kstr]
  uint8_t* const gwin = smem + PWC_PX * p.kstr;               // [2][64 px][64 ch] bf16: gout window, dc in place
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int cs = (int)blockIdx.x % p.csplit; const int64_t tile = (int64_t)blockIdx.x / p.csplit;
  const int chunk_lo = (cs * p.nchunk) / p.csplit, chunk_hi = ((cs + 1) * p.nchunk) / p.csplit;
  const int cw = (chunk_hi - chunk_lo) * 64;
  // per-channel rows of the chunk range: A, B, R, MR (= -M R) [, K1, E, F] [, the two sums]
  float* const tA = (float*)(gwin + 2 * PWC_PX * 64 * 2); float* const tB = tA + cw; float* const tR = tB + cw; float* const tMR = tR + cw;
  float* const tK1 = tMR + cw; float* const tE = tK1 + cw; float* const tF = tE + cw;
  int* const tW = (int*)(tF + cw);
  float* const l_f1 = (float*)(tW + cw); float* const l_f2 = l_f1 + cw;
  const int64_t p0 = tile * PWC_PX;
  const int CT = p.cpad >> 4;
  for (int i = tid; i < cw; i += 256) {
    const int c2 = chunk_lo * 64 + i; const bool ok = c2 < p.c;
    const int cc = ok ? c2 : 0;                  // every row unconditional from a clamped channel (a load under `ok ?` waits at its own join), zeroed afterwards
    float A = p.coef[FROST_COEF_A * p.cpad + cc], B = p.coef[FROST_COEF_B * p.cpad + cc], M = p.coef[FROST_COEF_M * p.cpad + cc], R = p.coef[FROST_COEF_R * p.cpad + cc];
    int ws = p.wsum[cc];
    float K1 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    if (MODE == 1) { K1 = p.coef[FROST_COEF_K1 * p.cpad + cc]; s1 = s12_sum(p.coef, p.cpad, 0, cc); s2 = s12_sum(p.coef, p.cpad, 1, cc); }
    if (!ok) { A = 0.0f; B = 0.0f; M = 0.0f; R = 0.0f; ws = 0; K1 = 0.0f; s1 = 0.0f; s2 = 0.0f; }
    tA[i] = A; tB[i] = B; tR[i] = R; tMR[i] = -M * R; tW[i] = ws;
    if (MODE == 1) {
      const float E = -K1 * (s2 * p.inv_count) * R;
      tK1[i] = K1; tE[i] = E; tF[i] = -K1 * (s1 * p.inv_count) - E * M;
    } else if (MODE == 0) { l_f1[i] = 0.0f; l_f2[i] = 0.0f; }
  }
  {   // the tile's input rows: contiguous in HBM; rows past the tensor stay zero
    const int upr = p.cin >> 3; const int64_t rows = p.npix - p0 < PWC_PX ? p.npix - p0 : PWC_PX; const int total = (int)rows * upr;
    constexpr int XB = (PWC_PX * KSM * 8 + 255) / 256;
    const int8_t* src = p.x + p0 * p.cin;
    uint2 xv[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) { const int u = tid + i * 256; xv[i] = (u < total) ? *(const uint2*)(src + (int64_t)u * 8) : make_uint2(0, 0); }
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int u = tid + i * 256; const int row = u / upr, col = u - row * upr;
      if (u < PWC_PX * upr) *(uint2*)(xs + row * p.kstr + col * 8) = xv[i];
    }
  }
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  const float y_inv = 1.0f / p.qy[FROST_Q_SCALE];
  const float y_zpf = (float)__float_as_int(p.qy[FROST_Q_ZP]), qcap = (float)q_hi(p.qy); const bool lowq = qcap < 255.0f;
  float t_lo = 0.0f, t_hi;
  {
    const int zpy = __float_as_int(p.qy[FROST_Q_ZP]), qhi = q_hi(p.qy);
    const float hi0 = (float)qhi + 0.5f - (float)zpy;
    t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
    if (!p.relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  }
  uint32_t rng = sr_seed(blockIdx.x, threadIdx.x);
  // gout window units: pixel u >> 3, 16-byte part u & 7 (two per thread)
  const int part = tid & 7, upx0 = tid >> 3, upx1 = (tid + 256) >> 3;
  const bool pv0 = (p0 + upx0) < p.npix, pv1 = (p0 + upx1) < p.npix;
  auto load_g = [&](int chunk, uint4 (&dst)[2]) __attribute__((always_inline)) {
    const bool cok = (chunk * 64 + part * 8) < p.c;
    const uint16_t* gs = p.gout + p0 * p.c + chunk * 64 + part * 8;
    dst[0] = (pv0 && cok) ? *(const uint4*)(gs + (int64_t)upx0 * p.c) : make_uint4(0, 0, 0, 0);
    dst[1] = (pv1 && cok) ? *(const uint4*)(gs + (int64_t)upx1 * p.c) : make_uint4(0, 0, 0, 0);
  };
  auto load_w = [&](int chunk, v4i (&dst)[KSM]) __attribute__((always_inline)) {
    const int ct = min(chunk * 4 + w, CT - 1);
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) dst[ks] = *(const v4i*)(p.w + ((((int64_t)ct * KSM + ks) * 64 + lane) << 4));
  };
  uint4 gv[2]; v4i afr[KSM];
  if (chunk_lo < chunk_hi) { if (MODE != 2) load_g(chunk_lo, gv); load_w(chunk_lo, afr); }
  __syncthreads();

  for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
    uint8_t* const gw = gwin + ((chunk - chunk_lo) & 1) * (PWC_PX * 64 * 2);
    if (MODE != 2) {
      *(uint4*)(gw + (upx0 * 64 + part * 8) * 2) = gv[0];
      *(uint4*)(gw + (upx1 * 64 + part * 8) * 2) = gv[1];
    }
    // conv recomputation: wave = channel tile w of the chunk, 4 pixel tiles
    const int ti = (chunk - chunk_lo) * 64 + w * 16 + 4 * g;
    v4i acc[4];
    {
      const int4 ws = *(const int4*)(tW + ti); const v4i init = (v4i){-zpx * ws.x, -zpx * ws.y, -zpx * ws.z, -zpx * ws.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = init;
    }
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const v4i bfr = *(const v4i*)(xs + (t * 16 + j) * p.kstr + ks * 64 + g * 16);
        acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[ks], bfr, acc[t], 0, 0, 0);          // D[chan][pix]
      }
    if (chunk + 1 < chunk_hi) { if (MODE != 2) load_g(chunk + 1, gv); load_w(chunk + 1, afr); }
    if (MODE != 2) pwc_barrier();                              // the gout window is complete
    const float4 A4 = *(const float4*)(tA + ti), B4 = *(const float4*)(tB + ti);
    const float A[4] = {A4.x, A4.y, A4.z, A4.w}, B[4] = {B4.x, B4.y, B4.z, B4.w};
    if (MODE == 2) {
      // forward emit (k_pw's expression): q = clamp(rint(fma(A, acc, B) / s) + zp, 0, hi) -> int8 window [64 px][64 ch] -> 64-byte rows out
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yv = fmaf(A[r], (float)acc[t][r], B[r]);
          float qv = rintf(yv * y_inv) + y_zpf;
          if (lowq) qv = fminf(qv, qcap);
          packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);
        }
        *(uint32_t*)(gw + (t * 16 + j) * 64 + w * 16 + 4 * g) = packed ^ 0x80808080u;
      }
      pwc_barrier();
      const int px = tid >> 2, q16 = tid & 3;                  // 64 pixels x 4 pieces of 16 bytes
      if ((p0 + px) < p.npix && (chunk * 64 + q16 * 16) < p.c) {
        int8_t* dst = p.y + (p0 + px) * p.c + chunk * 64 + q16 * 16;
        if ((p.c & 15) == 0) *(uint4*)dst = *(const uint4*)(gw + px * 64 + q16 * 16);
        else { *(uint2*)dst = *(const uint2*)(gw + px * 64 + q16 * 16); if ((chunk * 64 + q16 * 16 + 8) < p.c) *(uint2*)(dst + 8) = *(const uint2*)(gw + px * 64 + q16 * 16 + 8); }
      }
      continue;
    }
    if (MODE == 0) {
      const float4 R4 = *(const float4*)(tR + ti), M4 = *(const float4*)(tMR + ti);
      const float R[4] = {R4.x, R4.y, R4.z, R4.w}, MR[4] = {M4.x, M4.y, M4.z, M4.w};
      float r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint2 gq2 = *(const uint2*)(gw + ((t * 16 + j) * 64 + w * 16 + 4 * g) * 2);
        const float gq[4] = {__uint_as_float(gq2.x << 16), __uint_as_float(gq2.x & 0xffff0000u), __uint_as_float(gq2.y << 16), __uint_as_float(gq2.y & 0xffff0000u)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float af = (float)acc[t][r];
          const float tq = fmaf(A[r], af, B[r]) * y_inv;
          const float gy = (tq > t_lo && tq <= t_hi) ? gq[r] : 0.0f;        // rows past the tensor carry gout = 0
          r1[r] += gy; r2[r] = fmaf(gy, fmaf(af, R[r], MR[r]), r2[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = pwc_row_sum(r1[r]), b = pwc_row_sum(r2[r]);
        if (j == 15) { atomicAdd(&l_f1[ti + r], a); atomicAdd(&l_f2[ti + r], b); }
      }
    } else {
      const float4 K4 = *(const float4*)(tK1 + ti), E4 = *(const float4*)(tE + ti), F4 = *(const float4*)(tF + ti);
      const float K1[4] = {K4.x, K4.y, K4.z, K4.w}, E[4] = {E4.x, E4.y, E4.z, E4.w}, F[4] = {F4.x, F4.y, F4.z, F4.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint8_t* const gp = gw + ((t * 16 + j) * 64 + w * 16 + 4 * g) * 2;
        const uint2 gq2 = *(const uint2*)gp;
        const float gq[4] = {__uint_as_float(gq2.x << 16), __uint_as_float(gq2.x & 0xffff0000u), __uint_as_float(gq2.y << 16), __uint_as_float(gq2.y & 0xffff0000u)};
        float dcv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float af = (float)acc[t][r];
          const float tq = fmaf(A[r], af, B[r]) * y_inv;
          const float gy = (tq > t_lo && tq <= t_hi) ? gq[r] : 0.0f;
          dcv[r] = fmaf(gy, K1[r], fmaf(af, E[r], F[r]));
        }
        uint2 o;
        if (p.sr) { o.x = sr_pk_bf16(dcv[0], dcv[1], rng); o.y = sr_pk_bf16(dcv[2], dcv[3], rng); }
        else { o.x = cvt_pk_bf16(dcv[0], dcv[1]); o.y = cvt_pk_bf16(dcv[2], dcv[3]); }
        *(uint2*)gp = o;                                        // dc in place over the gout window
      }
      pwc_barrier();
      // the dc window out: full 128-byte lines per pixel
      const bool cok = (chunk * 64 + part * 8) < p.c;
      uint16_t* dd = p.dc + p0 * p.c + chunk * 64 + part * 8;
      if (pv0 && cok) *(uint4*)(dd + (int64_t)upx0 * p.c) = *(const uint4*)(gw + (upx0 * 64 + part * 8) * 2);
      if (pv1 && cok) *(uint4*)(dd + (int64_t)upx1 * p.c) = *(const uint4*)(gw + (upx1 * 64 + part * 8) * 2);
    }
  }
  if (MODE == 0) {
    __syncthreads();
    for (int i = tid; i < cw; i += 256) {
      const int c2 = chunk_lo * 64 + i;
      if (c2 < p.c) { atomicAdd(s12_dst(p.coef, p.cpad, 0) + c2, l_f1[i]); atomicAdd(s12_dst(p.coef, p.cpad, 1) + c2, l_f2[i]); }
    }
  }
}

template <int MODE, int KSM>
static int launch_pwc(PwcP& p, hipStream_t s) {
  const int cwmax = ((p.nchunk + p.csplit - 1) / p.csplit) * 64;
  const size_t lds = (size_t)PWC_PX * p.kstr + 2 * PWC_PX * 64 * 2 + (size_t)cwmax * 4 * 10 + 64;
  FROST_REQUIRE(lds <= 64 * 1024, "pwc: LDS budget exceeded");
  const int64_t tiles = (p.npix + PWC_PX - 1) / PWC_PX;
  hipLaunchKernelGGL((k_pwc<MODE, KSM>), dim3((unsigned)(tiles * p.csplit)), dim3(256), lds, s, p);
  return frost_check_launch("pwc");
}

// 1 if the chunked kernel takes this layer: wide output (the rows k_pw cannot stage), input rows of at most 320 bytes
extern "C" int frost_pwc_bwd_ok(int64_t npix, int cin, int cout) {
  static const int minc = getenv("FROST_PWC") ? atoi(getenv("FROST_PWC")) : 256;        // smallest Cout it takes; 0 = off
  return (minc > 0 && cout >= minc && (cin % 8) == 0 && cin <= 320 && (cout % 8) == 0 && npix >= 1024) ? 1 : 0;
}

// pass 0: reduce, pass 1: dc -- the contract of frost_pw_conv_bwd passes 0 / 1
extern "C" int frost_pwc_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout, int pass,
                                  float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, void* stream) {
  FROST_REQUIRE(frost_pwc_bwd_ok(npix, cin, cout) && (pass == 0 || (pass == 1 && dc)), "pwc_conv_bwd: unsupported layer or pass");
  PwcP p = {};
  p.x = x; p.qx = qrec_x; p.w = wq_pack; p.wsum = wsum; p.coef = coef; p.qy = qrec_y; p.gout = gout; p.dc = dc; p.npix = npix; p.cin = cin; p.c = cout;
  p.cpad = round_up(cout, 16); const int KS = (cin + 63) / 64; p.kstr = KS * 64 + 16; p.nchunk = (cout + 63) / 64; p.relu = relu; p.sr = frost_sr_enabled();
  p.inv_count = 1.0f / (float)npix;
  static const int cpw_env = getenv("FROST_PWC_CPW") ? atoi(getenv("FROST_PWC_CPW")) : 0;
  // chunks per workgroup: the x tile is staged once per workgroup, so more chunks amortise it; fewer keep >= ~2000 workgroups in the launch
  const int64_t tiles = (npix + PWC_PX - 1) / PWC_PX;
  int cpw = cpw_env > 0 ? cpw_env : (int)((tiles * p.nchunk + 2047) / 2048);
  if (cpw < 2) cpw = 2;
  if (cpw > p.nchunk) cpw = p.nchunk;
  if (cpw > 12) cpw = 12;
  p.csplit = (p.nchunk + cpw - 1) / cpw;
  hipStream_t s = as_stream(stream);
#define PWC_GO(KK) if (KS == KK) return pass == 0 ? launch_pwc<0, KK>(p, s) : launch_pwc<1, KK>(p, s);
  PWC_GO(1) PWC_GO(2) PWC_GO(3) PWC_GO(4) PWC_GO(5)
#undef PWC_GO
  FROST_REQUIRE(false, "pwc_conv_bwd: no instance");
  return 1;
}

// the forward emit pass of the same layers (the contract of frost_pw_conv_fwd mode 1: training / eval form, not the converted-inference forms): the quantised
// output leaves through an LDS window with 64-byte row pieces instead of 4-byte pieces at a Cout-byte stride
extern "C" int frost_pwc_conv_fwd_emit(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout,
                                       const float* coef, const float* qrec_y, int8_t* y, void* stream) {
  FROST_REQUIRE(frost_pwc_bwd_ok(npix, cin, cout) && y, "pwc_conv_fwd_emit: unsupported layer");
  PwcP p = {};
  p.x = x; p.qx = qrec_x; p.w = wq_pack; p.wsum = wsum; p.coef = (float*)coef; p.qy = qrec_y; p.y = y; p.npix = npix; p.cin = cin; p.c = cout;
  p.cpad = round_up(cout, 16); const int KS = (cin + 63) / 64; p.kstr = KS * 64 + 16; p.nchunk = (cout + 63) / 64;
  const int64_t tiles = (npix + PWC_PX - 1) / PWC_PX;
  int cpw = (int)((tiles * p.nchunk + 2047) / 2048);
  if (cpw < 2) cpw = 2;
  if (cpw > p.nchunk) cpw = p.nchunk;
  if (cpw > 12) cpw = 12;
  p.csplit = (p.nchunk + cpw - 1) / cpw;
  hipStream_t s = as_stream(stream);
#define PWC_GO(KK) if (KS == KK) return launch_pwc<2, KK>(p, s);
  PWC_GO(1) PWC_GO(2) PWC_GO(3) PWC_GO(4) PWC_GO(5)
#undef PWC_GO
  FROST_REQUIRE(false, "pwc_conv_fwd_emit: no instance");
  return 1;
}

// The forward STATISTICS pass of the same layers with the finalize folded in (the contract of frost_pw_conv_fwd_fin, which routes here): sum, sum of squares, min, max of
// the integer conv output per channel -> the layer's replicated tables (stats_copy), conv_finalize_dev in the last workgroup.  k_pw walks ALL output channels of a 128-pixel
// tile with 8 waves per workgroup -- on the widest layers that is < 1 workgroup per CU, each wave waiting on one weight fetch after the other (46 - 83 us for 4 - 6 us of MFMA +
// VALU work).  Here a workgroup owns `ptpw` 64-pixel tiles x a range of 64-channel chunks.  MFMA operands as k_pw's M_STATS: D'[pix][chan], a lane owns ONE channel and 16
// pixels, the fp32 square sums run in k_pw's order and leave as integers -- tables bit-identical to k_pw's.
template <int KSM>
__global__ __launch_bounds__(256, 4) void k_pwc_stats(const PwcP p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xs = smem;                                   // [64][kstr]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int cs = (int)blockIdx.x % p.csplit; const int64_t grp = (int64_t)blockIdx.x / p.csplit;
  const int chunk_lo = (cs * p.nchunk) / p.csplit, chunk_hi = ((cs + 1) * p.nchunk) / p.csplit;
  const int cw = (chunk_hi - chunk_lo) * 64;
  const int CT = p.cpad >> 4;
  // behind the x tile: [cw] sum (i64) | [cw] sum of squares (u64) | [cw] min | [cw] max | [cw] weight sums -- every entry has ONE owner lane (wave = channel tile of the chunk)
  long long* const q_s1 = (long long*)(smem + PWC_PX * p.kstr); unsigned long long* const q_s2 = (unsigned long long*)(q_s1 + cw);
  int* const q_mn = (int*)(q_s2 + cw); int* const q_mx = q_mn + cw; int* const q_w = q_mx + cw;
  for (int i = tid; i < cw; i += 256) {
    const int c2 = chunk_lo * 64 + i;
    q_w[i] = c2 < p.c ? p.wsum[c2] : 0; q_s1[i] = 0; q_s2[i] = 0; q_mn[i] = INT32_MAX; q_mx[i] = INT32_MIN;
  }
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  auto load_w = [&](int chunk, v4i (&dst)[KSM]) __attribute__((always_inline)) {
    const int ct = min(chunk * 4 + w, CT - 1);
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) dst[ks] = *(const v4i*)(p.w + ((((int64_t)ct * KSM + ks) * 64 + lane) << 4));
  };
  const int upr = p.cin >> 3;
  constexpr int XB = (PWC_PX * KSM * 8 + 255) / 256;
  const int64_t t0 = grp * p.ptpw, t1 = (t0 + p.ptpw < p.ntiles) ? t0 + p.ptpw : p.ntiles;
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t p0 = tile * PWC_PX;
    {   // the tile's input rows: contiguous in HBM; rows past the tensor stay zero
      const int64_t rows = p.npix - p0 < PWC_PX ? p.npix - p0 : PWC_PX; const int total = (int)rows * upr;
      const int8_t* src = p.x + p0 * p.cin;
      uint2 xv[XB];
#pragma unroll
      for (int i = 0; i < XB; ++i) { const int u = tid + i * 256; xv[i] = (u < total) ? *(const uint2*)(src + (int64_t)u * 8) : make_uint2(0, 0); }
      if (tile > t0) __syncthreads();                         // the previous tile's fragment reads are done
#pragma unroll
      for (int i = 0; i < XB; ++i) {
        const int u = tid + i * 256; const int row = u / upr, col = u - row * upr;
        if (u < PWC_PX * upr) *(uint2*)(xs + row * p.kstr + col * 8) = xv[i];
      }
    }
    v4i afr[KSM];
    if (chunk_lo < chunk_hi) load_w(chunk_lo, afr);
    __syncthreads();
    const bool full = p0 + PWC_PX <= p.npix;
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
      // D'[pix][chan]: lane (j, g) holds channel w * 16 + j of the chunk, pixels t * 16 + 4 g + r
      const int ci = (chunk - chunk_lo) * 64 + w * 16 + j;
      v4i acc[4];
      { const int c0 = -zpx * q_w[ci]; const v4i init = (v4i){c0, c0, c0, c0};
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = init; }
#pragma unroll
      for (int ks = 0; ks < KSM; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const v4i bfr = *(const v4i*)(xs + (t * 16 + j) * p.kstr + ks * 64 + g * 16);
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bfr, afr[ks], acc[t], 0, 0, 0);
        }
      if (chunk + 1 < chunk_hi) load_w(chunk + 1, afr);
      int a1 = 0; float a2 = 0.0f; int mn = INT32_MAX, mx = INT32_MIN;
      if (full) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int v0 = acc[t][0], v1 = acc[t][1], v2 = acc[t][2], v3 = acc[t][3];
          a1 += (v0 + v1) + (v2 + v3);
          const float f0 = (float)v0, f1 = (float)v1, f2 = (float)v2, f3 = (float)v3;
          a2 = fmaf(f0, f0, a2); a2 = fmaf(f1, f1, a2); a2 = fmaf(f2, f2, a2); a2 = fmaf(f3, f3, a2);
          mn = min(mn, min(v0, v1)); mn = min(mn, min(v2, v3)); mx = max(mx, max(v0, v1)); mx = max(mx, max(v2, v3));
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = acc[t][r];
            if ((p0 + t * 16 + 4 * g + r) < p.npix) { a1 += v; const float fv = (float)v; a2 = fmaf(fv, fv, a2); mn = min(mn, v); mx = max(mx, v); }
          }
      }
      long long b1 = a1; double b2 = (double)a2;
      b1 += __shfl_xor(b1, 16); b1 += __shfl_xor(b1, 32); b2 += __shfl_xor(b2, 16); b2 += __shfl_xor(b2, 32);
      mn = min(mn, __shfl_xor(mn, 16)); mn = min(mn, __shfl_xor(mn, 32)); mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
      if (g == 0 && mn <= mx) {                              // the owner lane: plain read-modify-write (integers: the order of the tiles does not matter)
        q_s1[ci] += b1; q_s2[ci] += (unsigned long long)__double2ll_rn(b2); q_mn[ci] = min(q_mn[ci], mn); q_mx[ci] = max(q_mx[ci], mx);
      }
    }
  }
  __syncthreads();
  long long* g_s1 = (long long*)stats_copy(p.stats, p.cpad); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
  int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
  for (int i = tid; i < cw; i += 256) {
    const int c2 = chunk_lo * 64 + i;
    if (c2 < p.c && q_mn[i] <= q_mx[i]) {
      atomicAdd((unsigned long long*)&g_s1[c2], (unsigned long long)q_s1[i]); atomicAdd(&g_s2[c2], q_s2[i]);
      atomicMin(&g_mn[c2], q_mn[i]); atomicMax(&g_mx[c2], q_mx[i]);
    }
  }
  int* sflag = (int*)smem;                                    // (the x tile is dead: last_block_done2 starts with a workgroup barrier)
  if (last_block_done2(p.fin.counter, p.fin_total, sflag)) {
    float* sh = (float*)(smem + 16);
    conv_finalize_dev(p.stats, p.npix, p.c, p.cpad, p.qx, p.fin.qrec_w, p.fin.wscale, p.fin.gamma, p.fin.beta, p.fin.rmean, p.fin.rvar, p.fin.nbt,
                      p.fin.training, p.fin.relu, p.fin.observe, 1, p.fin.coef, p.fin.qrec_y, tid, 256, sh, p.fin.cat_qrec_b, p.fin.cat_qrec_y);
  }
}

// Measured (profiles/r06_pwc_stats_ab.txt, B = 512): with ONE tile per workgroup faster only from Cout = 1280 up (82 -> 50 us at 288 -> 1728) and slower on the 14 x 14 layers
// (33 -> 40 us: 3136 small workgroups, each ending in its own flush + ticket round trip); with 4 tiles per workgroup equal or faster everywhere it applies, -0.04 ms in the step.
int frost_pwc_stats_fin_ok(int64_t npix, int cin, int cout) {
  static const int minc = getenv("FROST_PWC_STATS") ? atoi(getenv("FROST_PWC_STATS")) : 1;        // smallest Cout it takes (on top of frost_pwc_bwd_ok's 256); 0 = off
  return minc > 0 && cout >= minc && frost_pwc_bwd_ok(npix, cin, cout);
}
int frost_pwc_stats_fin(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout, void* stats, const FrostFinDesc* fin,
                        hipStream_t s) {
  PwcP p = {};
  p.x = x; p.qx = qrec_x; p.w = wq_pack; p.wsum = wsum; p.npix = npix; p.cin = cin; p.c = cout; p.stats = (uint8_t*)stats; p.fin = *fin;
  p.cpad = round_up(cout, 16); const int KS = (cin + 63) / 64; p.kstr = KS * 64 + 16; p.nchunk = (cout + 63) / 64;
  static const int cpw_env = getenv("FROST_PWC_STATS_CPW") ? atoi(getenv("FROST_PWC_STATS_CPW")) : 0;
  static const int pt_env = getenv("FROST_PWC_STATS_PT") ? atoi(getenv("FROST_PWC_STATS_PT")) : 4;           // 64-pixel tiles per workgroup (1 / 2 / 4 / 8 measured: r06_pwc_stats_ab.txt)
  p.ntiles = (npix + PWC_PX - 1) / PWC_PX;
  p.ptpw = pt_env < 1 ? 1 : pt_env;
  const int64_t groups = (p.ntiles + p.ptpw - 1) / p.ptpw;
  int cpw = cpw_env > 0 ? cpw_env : (int)((groups * p.nchunk + 2047) / 2048);
  if (cpw < 2) cpw = 2;
  if (cpw > p.nchunk) cpw = p.nchunk;
  if (cpw > 12) cpw = 12;
  p.csplit = (p.nchunk + cpw - 1) / cpw;
  p.fin_total = (unsigned)(groups * p.csplit);
  const int cwmax = ((p.nchunk + p.csplit - 1) / p.csplit) * 64;
  const size_t lds = (size_t)PWC_PX * p.kstr + (size_t)cwmax * 28 + 64;
  FROST_REQUIRE(lds <= 64 * 1024, "pwc_stats: LDS budget exceeded");
#define PWC_GO(KK) if (KS == KK) { hipLaunchKernelGGL((k_pwc_stats<KK>), dim3((unsigned)(groups * p.csplit)), dim3(256), lds, s, p); return frost_check_launch("pwc_stats"); }
  PWC_GO(1) PWC_GO(2) PWC_GO(3) PWC_GO(4) PWC_GO(5)
#undef PWC_GO
  FROST_REQUIRE(false, "pwc_stats_fin: no instance");
  return 1;
}
