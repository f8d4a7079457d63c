// Pointwise (1x1) convolution on fake-quantised operands: int8 MFMA forward (stats / emit passes),
// recompute-based backward (reduce / dc passes) and the bf16 MFMA dgrad, all on ONE skeleton:
//
//   D[chan][pix] = sum_K Wpack[chan][K] * T[pix][K]
//
// T is a pixel-major tensor (NHWC activation bytes, or the bf16 dc tensor for dgrad) whose 128-pixel tile is
// staged ONCE through LDS with fully coalesced loads and stays resident while the workgroup walks every
// output-channel group (activation-stationary: the big operand is read from HBM exactly once per pass);
// the small operand (packed weights, L2 resident) is fetched straight into MFMA fragments with 1 KiB
// wave-loads.  Both element types use a 64-byte K-step (16x16x64 i8 / 16x16x32 bf16), so staging, LDS layout
// and fragment addressing are shared.
//
// Orientation: emit / backward / dgrad compute D[chan][pix] (lane = pixel column, 4 consecutive channels per
// lane) so every global access of the epilogue is a 4-channel vector (packed int8 dword, bf16x4).  The STATS pass
// swaps the MFMA operands and gets D'[pix][chan] (lane = ONE channel, 4 pixels per register): per-channel
// sum / sum^2 / min / max then accumulate lane-locally across the whole persistent tile loop and are reduced
// across lanes once per workgroup instead of once per tile.
//
// The zero-point correction -zp'*sum_k(wq) is folded into the accumulator initial value (free).
#include "frost_common.h"
#include <stdlib.h>

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
  int ngroups, mi_eff; int64_t ntiles; float inv_count; int dbg;
};

#define BP 128
#define MI 4

template <int CTRL> __device__ __forceinline__ int dpp_i(int v, int identity) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}
// inclusive scan over a 16-lane row with row_shr 1,2,4,8: lane 15 of every row holds the row total (pure VALU)
__device__ __forceinline__ float row_sum_f(float v) {
  v += __int_as_float(dpp_i<0x111>(__float_as_int(v), 0)); v += __int_as_float(dpp_i<0x112>(__float_as_int(v), 0));
  v += __int_as_float(dpp_i<0x114>(__float_as_int(v), 0)); v += __int_as_float(dpp_i<0x118>(__float_as_int(v), 0)); return v;
}

__device__ __forceinline__ uint32_t pack_bf2(float a, float b) { return cvt_pk_bf16(a, b); }

// Next-tile register prefetch (<= 20 VGPRs) for single-chunk rows of <= 320 B.  `tid` is made opaque so LICM does not
// hoist a dozen per-unit 64-bit addresses out of the persistent tile loop (that spilled every variant).
__device__ __forceinline__ void pw_prefetch(const PwP& p, int64_t tile, int tid, int kpad0, bool al16, uint4 (&pre)[5]) {
  asm volatile("" : "+v"(tid));
  const int64_t q0 = tile * BP; const uint8_t* src = p.T + q0 * p.rowbytes;
  if (al16) {
    const int U = kpad0 >> 4; const int total = BP * U;
#pragma unroll
    for (int jn = 0; jn < 5; ++jn) {
      const int u = tid + jn * 512; const int row = u / U; const int col = (u - row * U) << 4;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (u < total && (q0 + row) < p.npix && col < p.rowbytes) v = *(const uint4*)(src + row * p.rowbytes + col);
      pre[jn] = v;
    }
  } else {
    const int U = kpad0 >> 3; const int total = BP * U;
#pragma unroll
    for (int jh = 0; jh < 5; ++jh) {
      uint2 v0 = make_uint2(0, 0), v1 = make_uint2(0, 0);
      { const int u = tid + (2 * jh) * 512; const int row = u / U; const int col = (u - row * U) << 3;
        if (u < total && (q0 + row) < p.npix && col < p.rowbytes) v0 = *(const uint2*)(src + row * p.rowbytes + col); }
      { const int u = tid + (2 * jh + 1) * 512; const int row = u / U; const int col = (u - row * U) << 3;
        if (u < total && (q0 + row) < p.npix && col < p.rowbytes) v1 = *(const uint2*)(src + row * p.rowbytes + col); }
      pre[jh] = make_uint4(v0.x, v0.y, v1.x, v1.y);
    }
  }
}

// RES = "resident" mode for small layers: the packed weights, wsum and the BN/quant coefficient rows are copied into LDS
// once per workgroup, so the persistent tile loop touches global memory only for the activation stream itself (no
// per-tile L2 round trips on the critical path), and the next tile is register-prefetched.
template <int MODE, int WP, bool RES>
__global__ __launch_bounds__(512, 4) void k_pw(const PwP p) {
  constexpr int WC = 8 / WP;          // waves along channels
  constexpr int NT = 8 / WP;          // 16-pixel tiles per wave
  constexpr bool BF = (MODE == M_DGRAD);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* xs = smem;
  const int xs_bytes = BP * p.kstr + 64;
  long long* l_s1 = (long long*)(smem + xs_bytes);
  unsigned long long* l_s2 = (unsigned long long*)(l_s1 + p.cpad);
  int* l_mn = (int*)(l_s2 + p.cpad);
  int* l_mx = l_mn + p.cpad;
  float* l_f1 = (float*)(smem + xs_bytes);
  float* l_f2 = l_f1 + p.cpad;
  const int red_bytes = (MODE == M_STATS) ? p.cpad * 24 : ((MODE == M_BRED) ? p.cpad * 8 : 0);
  const uint8_t* wl = smem + xs_bytes + red_bytes;                                  // [CT][KS][64][16 B]
  const int wl_bytes = (p.cpad >> 4) * p.KS * 1024;
  const float* cl = (const float*)(wl + wl_bytes);                                  // [FROST_COEF_ROWS][cpad]
  const int* wsl = (const int*)(cl + FROST_COEF_ROWS * p.cpad);                      // [cpad]
  const float* coefp = RES ? cl : p.coef;
  const int* wsump = RES ? wsl : p.wsum;

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
  if (RES) {
    for (int i = tid; i < (wl_bytes >> 4); i += 512) ((uint4*)wl)[i] = ((const uint4*)p.wpack)[i];
    if (MODE != M_STATS && MODE != M_DGRAD) for (int i = tid; i < FROST_COEF_ROWS * p.cpad; i += 512) ((float*)cl)[i] = p.coef[i];
    if (MODE != M_DGRAD) for (int i = tid; i < p.cpad; i += 512) ((int*)wsl)[i] = p.wsum[i];
  }
  if (RES) __syncthreads();

  int zpx = 0; float sw = 1.0f; float y_inv = 1.0f; float y_zpf = 0.0f;
  if (!BF) zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  if (MODE == M_DGRAD) sw = p.qw[FROST_Q_SCALE];
  if (MODE == M_EMIT || MODE == M_BRED || MODE == M_BDC) { y_inv = 1.0f / p.qy[FROST_Q_SCALE]; y_zpf = (float)__float_as_int(p.qy[FROST_Q_ZP]); }

  // lane-local accumulators that live across the persistent tile loop (single channel group only)
  // BRED with WP==2 already holds 64 accumulator registers: deferring would spill (measured 2x slower)
  const bool defer = (p.ngroups == 1) && !(MODE == M_BRED && WP == 2);
  long long st1[MI]; long long st2[MI]; int smn[MI], smx[MI];       // STATS: lane's channel = ct*16 + j
  float br1[MI][4], br2[MI][4];                                     // BRED: lane's channels = ct*16 + 4g + r
#pragma unroll
  for (int m = 0; m < MI; ++m) {
    st1[m] = 0; st2[m] = 0; smn[m] = INT32_MAX; smx[m] = INT32_MIN;
#pragma unroll
    for (int r = 0; r < 4; ++r) { br1[m][r] = 0.0f; br2[m][r] = 0.0f; }
  }

  const int kpad0 = (p.rowbytes + 63) & ~63;
  const bool can_pf = RES && (p.nchunks == 1) && (kpad0 <= 320);
  const bool al16 = (p.rowbytes & 15) == 0;
  uint4 pre[5];
  if (can_pf && (int64_t)blockIdx.x < p.ntiles) pw_prefetch(p, (int64_t)blockIdx.x, tid, kpad0, al16, pre);

  for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int64_t p0 = tile * BP;
    const bool full = (p0 + BP) <= p.npix;
    for (int cg = 0; cg < p.ngroups; ++cg) {
      const int ct0 = (cg * WC + wc) * p.mi_eff;
      int mi_n = CT - ct0; mi_n = mi_n < 0 ? 0 : (mi_n > p.mi_eff ? p.mi_eff : mi_n);
      v4i acci[MI][NT]; v4f accf[MI][NT];
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        v4i init = (v4i){0, 0, 0, 0};
        if (!BF && m < mi_n) {
          if (MODE == M_STATS) { const int c = -zpx * wsump[(ct0 + m) * 16 + j]; init = (v4i){c, c, c, c}; }
          else { const int4 ws = *(const int4*)(wsump + (ct0 + m) * 16 + 4 * g); init = (v4i){-zpx * ws.x, -zpx * ws.y, -zpx * ws.z, -zpx * ws.w}; }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) { acci[m][t] = init; accf[m][t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
      }

      for (int ch = 0; ch < p.nchunks; ++ch) {
        const int kc0 = ch * p.kc_bytes;
        int kcw = p.rowbytes - kc0; if (kcw > p.kc_bytes) kcw = p.kc_bytes;
        const int kcw_pad = (kcw + 63) & ~63;
        if (p.nchunks > 1 || cg == 0) {
          __syncthreads();
          const uint8_t* src = p.T + p0 * p.rowbytes + kc0;
          if (can_pf) {
            int t2 = threadIdx.x; asm volatile("" : "+v"(t2));
            if (al16) {
              const int U = kpad0 >> 4; const int total = BP * U;
#pragma unroll
              for (int jn = 0; jn < 5; ++jn) { const int u = t2 + jn * 512; const int row = u / U; const int col = (u - row * U) << 4; if (u < total) *(uint4*)(xs + row * p.kstr + col) = pre[jn]; }
            } else {
              const int U = kpad0 >> 3; const int total = BP * U;
#pragma unroll
              for (int jh = 0; jh < 5; ++jh) {
                { const int u = t2 + (2 * jh) * 512; const int row = u / U; const int col = (u - row * U) << 3;
                  if (u < total) *(uint2*)(xs + row * p.kstr + col) = make_uint2(pre[jh].x, pre[jh].y); }
                { const int u = t2 + (2 * jh + 1) * 512; const int row = u / U; const int col = (u - row * U) << 3;
                  if (u < total) *(uint2*)(xs + row * p.kstr + col) = make_uint2(pre[jh].z, pre[jh].w); }
              }
            }
          } else if (((p.rowbytes | kc0) & 15) == 0) {
            const int U = kcw_pad >> 4; const int total = BP * U;
            for (int u = tid; u < total; u += 512) {
              const int row = u / U; const int col = (u - row * U) << 4;
              uint4 v = make_uint4(0, 0, 0, 0);
              if ((p0 + row) < p.npix && col < kcw) v = *(const uint4*)(src + row * p.rowbytes + col);
              *(uint4*)(xs + row * p.kstr + col) = v;
            }
          } else {
            const int U = kcw_pad >> 3; const int total = BP * U;
            for (int u = tid; u < total; u += 512) {
              const int row = u / U; const int col = (u - row * U) << 3;
              uint2 v = make_uint2(0, 0);
              if ((p0 + row) < p.npix && col < kcw) v = *(const uint2*)(src + row * p.rowbytes + col);
              *(uint2*)(xs + row * p.kstr + col) = v;
            }
          }
          __syncthreads();
          if (can_pf) { const int64_t nxt = tile + gridDim.x; if (nxt < p.ntiles) pw_prefetch(p, nxt, tid, kpad0, al16, pre); }
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
              if (m < mi_n) afr[m] = *(const v4i*)((RES ? wl : p.wpack) + ((((int64_t)(ct0 + m) * p.KS + ks0 + ks) * 64 + lane) << 4));
#pragma unroll
            for (int m = 0; m < MI; ++m) {
              if (m < mi_n) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                  if (BF) accf[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[m]), __builtin_bit_cast(v8bf, bfr[t]), accf[m][t], 0, 0, 0);
                  else if (MODE == M_STATS) acci[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bfr[t], afr[m], acci[m][t], 0, 0, 0);   // D'[pix][chan]
                  else acci[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[m], bfr[t], acci[m][t], 0, 0, 0);                        // D[chan][pix]
                }
              }
            }
          }
        }
      }

      // ------------------------------------------------------------------------------- epilogue
      if (MODE == M_STATS) {
        // lane: channel ct*16+j; register r of tile t: pixel p0 + (wp*NT+t)*16 + 4g + r
#pragma unroll
        for (int m = 0; m < MI; ++m) {
          if (m >= mi_n) continue;
          long long a1 = 0; long long a2 = 0; int mn = INT32_MAX, mx = INT32_MIN;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int v = acci[m][t][r];
              if (full || (p0 + (wp * NT + t) * 16 + 4 * g + r) < p.npix) { a1 += v; a2 += (long long)v * v; mn = min(mn, v); mx = max(mx, v); }
            }
          }
          if (defer) { st1[m] += a1; st2[m] += a2; smn[m] = min(smn[m], mn); smx[m] = max(smx[m], mx); }
          else {
            a1 += __shfl_xor(a1, 16); a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 16); a2 += __shfl_xor(a2, 32);
            mn = min(mn, __shfl_xor(mn, 16)); mn = min(mn, __shfl_xor(mn, 32)); mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
            const int chn = (ct0 + m) * 16 + j;
            if (g == 0 && chn < p.cout) {
              atomicAdd((unsigned long long*)&l_s1[chn], (unsigned long long)a1); atomicAdd(&l_s2[chn], (unsigned long long)a2);
              atomicMin(&l_mn[chn], mn); atomicMax(&l_mx[chn], mx);
            }
          }
        }
        continue;
      }
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        if (m >= mi_n) continue;
        const int ch0 = (ct0 + m) * 16 + 4 * g;          // 4 consecutive channels of this lane
        const bool chok = ch0 < p.cout;
        if (MODE == M_DGRAD) {
          uint16_t* base = p.dx + p0 * p.cout;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int prow = (wp * NT + t) * 16 + j;
            if ((full || (p0 + prow) < p.npix) && chok) {
              uint16_t* dst = base + prow * p.cout + ch0;
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = accf[m][t][r] * sw;
              if (p.accumulate) { const uint2 o = *(const uint2*)dst; v[0] += bf2f(o.x & 0xffff); v[1] += bf2f(o.x >> 16); v[2] += bf2f(o.y & 0xffff); v[3] += bf2f(o.y >> 16); }
              uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
              *(uint2*)dst = o;
            }
          }
          continue;
        }
        const float4 A4 = *(const float4*)(coefp + FROST_COEF_A * p.cpad + ch0);
        const float4 B4 = *(const float4*)(coefp + FROST_COEF_B * p.cpad + ch0);
        const float A[4] = {A4.x, A4.y, A4.z, A4.w}, B[4] = {B4.x, B4.y, B4.z, B4.w};
        if (MODE == M_EMIT) {
          int8_t* base = p.y + p0 * p.cout;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int prow = (wp * NT + t) * 16 + j;
            uint32_t packed = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // q = clamp(rint(y*inv)+zp, 0, 255); ReLU is implied: ReLU layers have zp == 0 and v_cvt_pk_u8_f32
              // saturates at 0 (and at 255) while inserting the byte -- one op for clamp + convert + pack.
              const float yv = fmaf(A[r], (float)acci[m][t][r], B[r]);
              packed = __builtin_amdgcn_cvt_pk_u8_f32(rintf(yv * y_inv) + y_zpf, r, packed);
            }
            if (p.dbg & 1) { asm volatile("" :: "v"(packed)); }
            else if (p.dbg & 4) { *(uint32_t*)(base + ((((cg * 8 + w) * MI + m) * NT + t) * 64 + lane) * 4) = packed; }
            else if ((full || (p0 + prow) < p.npix) && chok) *(uint32_t*)(base + prow * p.cout + ch0) = packed ^ 0x80808080u;
          }
          continue;
        }
        // backward modes
        const float4 M4 = *(const float4*)(coefp + FROST_COEF_M * p.cpad + ch0);
        const float4 R4 = *(const float4*)(coefp + FROST_COEF_R * p.cpad + ch0);
        const float Mv[4] = {M4.x, M4.y, M4.z, M4.w}, Rv[4] = {R4.x, R4.y, R4.z, R4.w};
        float K1[4] = {0, 0, 0, 0}, S1[4] = {0, 0, 0, 0}, S2[4] = {0, 0, 0, 0};
        if (MODE == M_BDC) {
          const float4 k4 = *(const float4*)(coefp + FROST_COEF_K1 * p.cpad + ch0);
          const float4 a4 = *(const float4*)(coefp + FROST_COEF_S1 * p.cpad + ch0);
          const float4 b4 = *(const float4*)(coefp + FROST_COEF_S2 * p.cpad + ch0);
          K1[0] = k4.x; K1[1] = k4.y; K1[2] = k4.z; K1[3] = k4.w;
          S1[0] = a4.x * p.inv_count; S1[1] = a4.y * p.inv_count; S1[2] = a4.z * p.inv_count; S1[3] = a4.w * p.inv_count;
          S2[0] = b4.x * p.inv_count; S2[1] = b4.y * p.inv_count; S2[2] = b4.z * p.inv_count; S2[3] = b4.w * p.inv_count;
        }
        const uint16_t* gbase = p.gout + p0 * p.cout;
        uint16_t* dbase = p.dc + p0 * p.cout;
        const float relu_floor = p.relu ? 0.0f : -INFINITY;
        float r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int prow = (wp * NT + t) * 16 + j;
          const bool valid = (full || (p0 + prow) < p.npix) && chok;
          uint2 gv = make_uint2(0, 0);
          if (valid) gv = *(const uint2*)(gbase + prow * p.cout + ch0);
          const float gq[4] = {bf2f(gv.x & 0xffff), bf2f(gv.x >> 16), bf2f(gv.y & 0xffff), bf2f(gv.y >> 16)};
          float dcv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float af = (float)acci[m][t][r];
            const float yv = fmaf(A[r], af, B[r]);
            // STE masks: ReLU alive (y > 0) and fake-quant in range: 0 <= rint(relu(y)*inv)+zp <= 255
            const float qf = rintf(fmaxf(yv, relu_floor) * y_inv) + y_zpf;
            const bool pass = (yv > relu_floor) && qf >= 0.0f && qf <= 255.0f;
            const float gy = pass ? gq[r] : 0.0f;
            const float xhat = (af - Mv[r]) * Rv[r];
            if (MODE == M_BRED) { r1[r] += gy; r2[r] += gy * xhat; }
            else dcv[r] = K1[r] * (gy - S1[r] - xhat * S2[r]);
          }
          if (MODE == M_BDC && valid) {
            uint2 o; o.x = pack_bf2(dcv[0], dcv[1]); o.y = pack_bf2(dcv[2], dcv[3]);
            *(uint2*)(dbase + prow * p.cout + ch0) = o;
          }
        }
        if (MODE == M_BRED) {
          if (defer) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { br1[m][r] += r1[r]; br2[m][r] += r2[r]; }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a = row_sum_f(r1[r]), b = row_sum_f(r2[r]);
              if (j == 15 && chok) { atomicAdd(&l_f1[ch0 + r], a); atomicAdd(&l_f2[ch0 + r], b); }
            }
          }
        }
      }
    }
  }

  if (MODE == M_STATS) {
    if (defer) {
      const int ct0 = wc * p.mi_eff;
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int chn = (ct0 + m) * 16 + j;
        long long a1 = st1[m], a2 = st2[m]; int mn = smn[m], mx = smx[m];
        a1 += __shfl_xor(a1, 16); a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 16); a2 += __shfl_xor(a2, 32);
        mn = min(mn, __shfl_xor(mn, 16)); mn = min(mn, __shfl_xor(mn, 32)); mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
        if (m < p.mi_eff && g == 0 && chn < p.cout && mn <= mx) {
          atomicAdd((unsigned long long*)&l_s1[chn], (unsigned long long)a1); atomicAdd(&l_s2[chn], (unsigned long long)a2);
          atomicMin(&l_mn[chn], mn); atomicMax(&l_mx[chn], mx);
        }
      }
    }
    __syncthreads();
    long long* g_s1 = (long long*)p.stats; unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
    int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
    for (int c = tid; c < p.cout; c += 512) {
      if (l_mn[c] <= l_mx[c]) {
        atomicAdd((unsigned long long*)&g_s1[c], (unsigned long long)l_s1[c]);
        atomicAdd(&g_s2[c], l_s2[c]);
        atomicMin(&g_mn[c], l_mn[c]); atomicMax(&g_mx[c], l_mx[c]);
      }
    }
  } else if (MODE == M_BRED) {
    if (defer) {
      const int ct0 = wc * p.mi_eff;
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int ch0 = (ct0 + m) * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row_sum_f(br1[m][r]), b = row_sum_f(br2[m][r]);
          if (m < p.mi_eff && j == 15 && ch0 < p.cout) { atomicAdd(&l_f1[ch0 + r], a); atomicAdd(&l_f2[ch0 + r], b); }
        }
      }
    }
    __syncthreads();
    for (int c = tid; c < p.cout; c += 512) {
      atomicAdd(p.coef + FROST_COEF_S1 * p.cpad + c, l_f1[c]);
      atomicAdd(p.coef + FROST_COEF_S2 * p.cpad + c, l_f2[c]);
    }
  }
}

template <int MODE, int WP, bool RES>
static int launch_pw2(PwP& p, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)k_pw<MODE, WP, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  FROST_REQUIRE(lds <= 160 * 1024, "pw: LDS budget exceeded");
  int occ = 1;   // persistent workgroups: residency = what the register/LDS budget admits (queried, not guessed)
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_pw<MODE, WP, RES>, 512, lds) != hipSuccess || occ < 1) occ = 1;
  if (occ > 4) occ = 4;
  int64_t grid = p.ntiles < 256 * occ ? p.ntiles : 256 * occ;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_pw<MODE, WP, RES>), dim3((unsigned)grid), dim3(512), lds, s, p);
  return frost_check_launch("pw");
}
template <int MODE, int WP>
static int launch_pw(PwP& p, hipStream_t s) {
  size_t lds = (size_t)BP * p.kstr + 64;
  if (MODE == M_STATS) lds += (size_t)p.cpad * 24;
  if (MODE == M_BRED) lds += (size_t)p.cpad * 8;
  const size_t res_bytes = (size_t)(p.cpad >> 4) * p.KS * 1024 + (size_t)p.cpad * (FROST_COEF_ROWS + 1) * 4;
  static const int res_on = getenv("FROST_PW_RES") ? atoi(getenv("FROST_PW_RES")) : 1;
  // measured per (mode, wave split): resident mode pays where the epilogue is latency-chained and registers allow it
  constexpr bool res_ok = (WP != 2) && ((MODE == M_BDC) || (MODE == M_DGRAD) || (MODE == M_EMIT && WP == 4) || (MODE == M_BRED && WP == 8));
  if (res_ok && res_on && res_bytes <= 40 * 1024 && p.ntiles >= 2048) return launch_pw2<MODE, WP, true>(p, lds + res_bytes, s);
  return launch_pw2<MODE, WP, false>(p, lds, s);
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
  const char* e = getenv("FROST_DBG"); p.dbg = e ? atoi(e) : 0;
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
