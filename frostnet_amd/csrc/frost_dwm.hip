// Depthwise k x k conv (k in {3,5}) on the MATRIX cores -- "Toeplitz MFMA" formulation.
//
// Why: the lane-=-channel VALU stencil of frost_dw3.hip spends 20-55 vector instructions per output element (profiles/r02: the depthwise
// kernels are 3.4 G of the step's 6.8 G wave-level VALU instructions and 39 % of its time for 7 % of its MACs) while the MFMA pipe idles.
// For ONE channel a k x k depthwise conv over a 16 x 16 output tile is a sum over the kernel rows ky of small matrix products
//
//     out[i][j] = sum_ky  X_ky[i][a] * T_ky[a][j],      X_ky[i][a] = x[i*S + ky][a]  (a = input column of the halo tile),
//                                                      T_ky[a][j] = w[ky][a - j*S]   (banded Toeplitz matrix of the kernel row),
//
// i.e. exactly the shape of v_mfma_i32_16x16x64_i8 with M = 16 output rows, N = 16 output columns and K = input columns: at stride 1 two
// kernel rows share one instruction (32 K slots each, 16 + k - 1 <= 20 used), so a 5 x 5 conv of 256 outputs is THREE matrix instructions
// (48 cycles) instead of ~4500 lane-cycles of dot products, the integer result is the same exact int32 sum, and what is left on the vector
// ALU is the epilogue.  The price is a layout change: the A operand wants 16 consecutive input columns of one channel per lane, so the NHWC
// halo tile is transposed once in LDS into channel planes ([channel][row][column], gfx950's ds_read_b64_tr_b8 does the 8 x 8 byte transposes)
// and the B operand (the Toeplitz band) is assembled per channel from its k packed taps with one v_perm_b32 per dword and per-lane
// selectors that never change.
//
// Orientation of the result: lane (j = lane & 15, g = lane >> 4) holds out[4g + r][j], r = 0..3, of one channel -- per-channel statistics
// are lane-local sums across the whole persistent tile loop, the emit epilogue packs its 4 rows and 4 consecutive channels into dwords of
// an NHWC out tile in LDS that leaves with coalesced stores.
//
// Entry points are frost_dw3.hip's (frost_dw_conv_fwd / _fin / _bwd ...): those route here (FROST_DW_MFMA, default on) where this file has an instance.
#include "frost_common.h"
#include <map>

typedef int dv2i __attribute__((ext_vector_type(2)));
enum { DM_STATS = 0, DM_EMIT = 1, DM_BRED = 2, DM_BDC = 3 };

struct DwmP {
  const int8_t* x; const float* qx; const int8_t* wq; const int32_t* wsum;
  int n, h, w, c, cpad, ho, wo, pad;
  uint8_t* stats; float* coef; const float* qy; int relu; int8_t* y;
  const uint16_t* gout; uint16_t* dc;
  int tiles_x, tiles_y, ncb, ngrp; int nsp;      // spatial tiles (all images), channel blocks, workgroup groups per XCD
  FrostFinDesc fin; int fin_on; int sr; int cvt; float inv_count;
  int abl;       // timing ablation (FROST_DWM_ABL; results are wrong with any bit set): 1 no global fetch, 2 no LDS staging, 4 no transpose, 8 no matrix phase, 16 no copy-out
};

// Geometry of one workgroup tile: CB channels x NSUB spatial tiles of 16 x 16 outputs (stride 1), CB * NSUB = 64.  The NSUB tiles of a
// channel share its B operand (the Toeplitz band is a property of the channel), so 32-channel blocks halve the operand assembly per output and
// waste fewer lanes on the narrow high-resolution layers (C = 32, 72, 96, 144, 168).
template <int K, int CB>
struct DwmGeo {
  static constexpr int NSUB = 64 / CB;
  static constexpr int IH = 15 + K;                 // halo rows = halo columns (20 / 18)
  static constexpr int IWP = 24;                    // staged columns: a multiple of 8 (the transposing read takes 8 pixels)
  static constexpr int NM = (K + 1) / 2;            // matrix instructions per channel tile: two kernel rows each
  static constexpr int SUB_BYTES = IH * IWP * CB;   // NHWC halo tile of one sub-tile
  static constexpr int NHWC_BYTES = NSUB * SUB_BYTES + 512;     // + slack: the last transposing read of a 32-channel tile runs 8 pixels past its last row
  static constexpr int PROW = 32;                   // plane row pitch (bytes): K slots of one kernel row
  static constexpr int PSUB = (IH + 1) * PROW;      // plane of one (channel, sub-tile): one spare row for the unused kernel row K
  static constexpr int PCH = NSUB * PSUB + 16;      // plane pitch per channel: odd multiple of 16 bytes (LDS banks)
  static constexpr int PLANAR_BYTES = CB * PCH;
  static constexpr int OUT_PITCH = CB + 16;         // bytes per pixel of the NHWC out tile (pad: spreads the 4-channel dword writes over the banks, 16-byte aligned rows)
  static constexpr int OUT_BYTES = NSUB * 256 * OUT_PITCH;
  static constexpr int WTAB_BYTES = CB * (K + 1) * 8;
  static constexpr int STAGE_BYTES = NHWC_BYTES > OUT_BYTES ? NHWC_BYTES : OUT_BYTES;
  static constexpr int CTAB_BYTES = CB * 16;
  static constexpr int LDS_BYTES = STAGE_BYTES + PLANAR_BYTES + WTAB_BYTES + CTAB_BYTES + 64;
  static constexpr int NCH = CB / 8;                // channels per wave
};

__device__ __forceinline__ dv2i dwm_tr8(const uint8_t* p) {
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((dv2i __attribute__((address_space(3)))*)p);
}

// One workgroup = 8 waves; wave w owns NCH = CB / 8 channels of the block in groups of 4 consecutive channels (the out tile takes dwords of
// 4 channels): channel (q >> 2) * 32 + 4 w + (q & 3), q = 0 .. NCH - 1.
// Latency plan (the first version of this kernel, with loads and waits in program order, ran at 6-10 us per tile against ~1 us of arithmetic):
//   * the next tile's global loads are issued into registers right after this tile's bytes went to LDS and land under its arithmetic;
//   * every per-channel constant (taps, accumulator start value, A / B coefficient) sits in LDS tables filled once per workgroup;
//   * all LDS operands of a channel are requested together, and no branch sits inside the matrix phase.
// MODE: statistics / emit / emit in the converted-inference form (integer bias add + requantisation scale, see frost_convert.hip).
enum { DM_EMIT_CVT = 4 };
template <int K, int CB, int MODE>
__global__ __launch_bounds__(512, 4) void k_dwm(const DwmP p) {
  typedef DwmGeo<K, CB> G;
  constexpr int NSUB = G::NSUB, IH = G::IH, IWP = G::IWP, NM = G::NM, PCH = G::PCH, PROW = G::PROW, PSUB = G::PSUB, NCH = G::NCH;
  constexpr bool EMIT = (MODE == DM_EMIT || MODE == DM_EMIT_CVT), CVT = (MODE == DM_EMIT_CVT);
  constexpr int UPP = CB / 16;                                   // 16-byte staging units per pixel
  constexpr int NUNIT = NSUB * IH * IWP * UPP;                   // = IH * IWP * 4
  constexpr int NU = (NUNIT + 511) / 512;                        // staging units per thread
  constexpr int TPR = (CB == 64) ? 3 : 2;                        // transposing reads per halo row: 8 pixels x 64 channels, or 16 pixels x 32 channels
  constexpr int NITEM = NSUB * IH * TPR;
  constexpr int NT = (NITEM + 7) / 8;                            // transposing reads per wave
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const nhwc = smem;                                  // halo tiles [NSUB][IH][IWP][CB]; later the out tiles [NSUB][256][OUT_PITCH]
  uint8_t* const planar = smem + G::STAGE_BYTES;               // [CB][NSUB][IH + 1][32]
  uint8_t* const wtab = planar + G::PLANAR_BYTES;              // [CB][K + 1][8]: taps of a kernel row in bytes 0..K-1, row K = zeros
  uint8_t* const ctab = wtab + G::WTAB_BYTES;                  // [CB][16]: {accumulator start value, A, B, -}

  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- workgroup -> (XCD, channel block, tile range): the channel blocks of a spatial tile and neighbouring tiles (shared halos, shared 128-byte
  // lines) stay on one XCD's L2 (the lesson of frost_dw3.hip's dw_block_map)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cb = slot % p.ncb, grp = slot / p.ncb;
  const int nwt = (p.nsp + NSUB - 1) / NSUB;                                   // workgroup tiles = groups of NSUB consecutive spatial tiles
  const int wt_lo = (int)(((int64_t)nwt * xcd) >> 3), wt_hi = (int)(((int64_t)nwt * (xcd + 1)) >> 3);
  const int cvalid = min(CB, p.c - cb * CB);
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]);
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;

  // kernel taps and per-channel constants of this channel block -> LDS
  for (int i = tid; i < CB * (K + 1); i += 512) {
    const int lc = i / (K + 1), ky = i - lc * (K + 1);
    uint32_t lo = 0, hi = 0;
    if (ky < K && lc < cvalid) {
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const uint32_t b = (uint32_t)(uint8_t)p.wq[(ky * K + kx) * p.cpad + cb * CB + lc];
        if (kx < 4) lo |= b << (8 * kx); else hi |= b;
      }
    }
    *(uint2*)(wtab + i * 8) = make_uint2(lo, hi);
  }
  if (tid < CB) {
    uint4 cv = make_uint4(0, 0, 0, 0);
    if (tid < cvalid) {
      const int ch = cb * CB + tid;
      cv.x = (uint32_t)((128 - zp) * p.wsum[ch]);
      if (EMIT) { cv.y = __float_as_uint(p.coef[FROST_COEF_A * p.cpad + ch]); cv.z = __float_as_uint(p.coef[FROST_COEF_B * p.cpad + ch]); }
    }
    *(uint4*)(ctab + tid * 16) = cv;
  }
  // Toeplitz selectors: byte e of dword d of this lane's B operand is tap t = 16 (g & 1) + 4 d + e - j of kernel row 2 m + (g >> 1)
  // (v_perm_b32 pool: bytes 0..3 = taps 0..3, bytes 4..7 = taps 4..7 (taps >= K are zero bytes); selector 0x0c = constant 0)
  uint32_t sel[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = 16 * (g & 1) + 4 * d + e - j;
      s |= (uint32_t)((t >= 0 && t <= 7) ? t : 0x0c) << (8 * e);
    }
    sel[d] = s;
  }
  const int a_off = j * PROW + (g >> 1) * PROW + 16 * (g & 1);      // + lc * PCH + sub * PSUB + 2 m * PROW
  const int w_off = (g >> 1) * 8;                                    // + (lc * (K + 1) + 2 m) * 8

  // epilogue constants
  float y_inv = 1.0f, y_zpf = 0.0f, qcap = 255.0f;
  if (EMIT) {
    y_inv = CVT ? 1.0f : 1.0f / p.qy[FROST_Q_SCALE]; y_zpf = (float)__float_as_int(p.qy[FROST_Q_ZP]);
    qcap = (float)q_hi(p.qy);                     // 255, or 127 with 7-bit activations; the u8 conversion saturates at 255 by itself, so the min is a no-op there
  }
  const float relu_floor = p.relu ? 0.0f : -INFINITY;

  // staging: thread t moves the 16-byte units t, t + 512, ... of the halo tiles [NSUB][IH][IWP][UPP]; the decode is redone per tile (registers
  // are the scarce resource of this kernel, a few integer operations per tile are not)
  const int tpi = p.tiles_x * p.tiles_y;
  uint2 pre[NU][2]; uint32_t pre_ok = 0;
  auto fetch = [&](int wt) __attribute__((always_inline)) {
    pre_ok = 0;
#pragma unroll
    for (int q = 0; q < NU; ++q) {
      const int u = tid + 512 * q; const int cu = u % UPP, pix = u / UPP; const int sub = pix / (IH * IWP), pr = pix - sub * (IH * IWP);
      const int iy = pr / IWP, ix = pr - iy * IWP;
      const int sp = wt * NSUB + sub;                          // uniform per unit index q for NSUB <= 2 ... per lane in general: plain integer math
      const int img = sp / tpi; const int tr = sp - img * tpi; const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
      const int gy = ty * 16 - p.pad + iy, gx = tx * 16 - p.pad + ix;
      const bool ok = (u < NUNIT) && sp < p.nsp && ix < IH && cu * 16 < cvalid && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w;
      const uint2* src = (const uint2*)(ok ? p.x + (((int64_t)img * p.h + gy) * p.w + gx) * p.c + cb * CB + cu * 16 : p.x);   // always readable: the loads are unconditional (they batch)
      pre[q][0] = src[0]; pre[q][1] = src[1]; pre_ok |= ok ? (1u << q) : 0u;
    }
  };

  // lane-local statistics of the wave's channels (summed over its sub-tiles), alive across the tile loop
  int s1[NCH]; double s2[NCH]; int smn[NCH], smx[NCH];      // int32 sums: the host bounds the tiles per workgroup
  if (MODE == DM_STATS) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) { s1[q] = 0; s2[q] = 0.0; smn[q] = INT32_MAX; smx[q] = INT32_MIN; }
  }
  int wt = wt_lo + grp;
  if (wt < wt_hi && !(p.abl & 1)) fetch(wt);
  __syncthreads();

  for (; wt < wt_hi; wt += p.ngrp) {
    // ---- the halo tiles -> LDS, zero padding = zero-point fill
    if (!(p.abl & 2))
#pragma unroll
    for (int q = 0; q < NU; ++q) {
      const int u = tid + 512 * q;
      if (u < NUNIT) {
        uint4 v = make_uint4(zfill, zfill, zfill, zfill);
        if ((pre_ok >> q) & 1u) v = make_uint4(pre[q][0].x, pre[q][0].y, pre[q][1].x, pre[q][1].y);
        *(uint4*)(nhwc + u * 16) = v;
      }
    }
    __syncthreads();
    if (wt + p.ngrp < wt_hi && !(p.abl & 1)) fetch(wt + p.ngrp);              // lands under this tile's arithmetic
    // ---- transpose into channel planes.  CB = 64: one transposing read = 8 pixels x 64 channels (lane = channel gets its 8 pixels);
    // CB = 32: 16 pixels x 32 channels (lane & 31 = channel, lane >> 5 = which 8 of the 16 pixels)
    if (!(p.abl & 4)) {
      dv2i raw[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int it = min(w + 8 * i, NITEM - 1);              // a spare item repeats the last one (same bytes again): no branch
        const int sub = it / (IH * TPR), r2 = it - sub * (IH * TPR); const int iy = r2 / TPR, cg = r2 - iy * TPR;
        const uint8_t* row = nhwc + sub * G::SUB_BYTES + iy * IWP * CB;
        if (CB == 64) raw[i] = dwm_tr8(row + (cg * 8 + (j >> 1)) * 64 + 16 * g + 8 * (j & 1));
        else raw[i] = dwm_tr8(row + (cg * 16 + 8 * (g >> 1) + (j >> 1)) * 32 + 16 * (g & 1) + 8 * (j & 1));
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int it = min(w + 8 * i, NITEM - 1);
        const int sub = it / (IH * TPR), r2 = it - sub * (IH * TPR); const int iy = r2 / TPR, cg = r2 - iy * TPR;
        if (CB == 64) *(dv2i*)(planar + lane * PCH + sub * PSUB + iy * PROW + cg * 8) = raw[i];
        else *(dv2i*)(planar + (lane & 31) * PCH + sub * PSUB + iy * PROW + cg * 16 + 8 * (lane >> 5)) = raw[i];
      }
    }
    __syncthreads();

    // ---- matrix phase: per channel one B operand set, then per sub-tile 3 A operands, NM matrix instructions and the epilogue
    uint32_t pk[NSUB][4];
    if (!(p.abl & 8))
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int lc0 = (q >> 2) * 32 + 4 * w, lc = lc0 + (q & 3);         // channels past the end of a partial block run too (zero taps, zero start value; nothing of theirs is flushed or stored)
      uint4 ct;
      if (EMIT) ct = *(const uint4*)(ctab + lc * 16); else ct.x = *(const uint32_t*)(ctab + lc * 16);
      uint2 wl[NM]; v4i af[NSUB][NM];
#pragma unroll
      for (int m = 0; m < NM; ++m) wl[m] = *(const uint2*)(wtab + (lc * (K + 1) + 2 * m) * 8 + w_off);
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int m = 0; m < NM; ++m) af[sub][m] = *(const v4i*)(planar + lc * PCH + sub * PSUB + 2 * m * PROW + a_off);
      v4i bf[NM];
#pragma unroll
      for (int m = 0; m < NM; ++m)
        bf[m] = (v4i){(int)__builtin_amdgcn_perm(wl[m].y, wl[m].x, sel[0]), (int)__builtin_amdgcn_perm(wl[m].y, wl[m].x, sel[1]),
                      (int)__builtin_amdgcn_perm(wl[m].y, wl[m].x, sel[2]), (int)__builtin_amdgcn_perm(wl[m].y, wl[m].x, sel[3])};
      const int init = (int)ct.x;
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        v4i acc = (v4i){init, init, init, init};
#pragma unroll
        for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[sub][m], bf[m], acc, 0, 0, 0);
        if (MODE == DM_STATS) {
          // exact integer sum and min / max; squares in fp32 per element (as frost_pw.hip), their sums in double across tiles
          const int sp = wt * NSUB + sub;
          const int img = sp / tpi; const int tr = sp - img * tpi; const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
          const int hv = (sp < p.nsp) ? min(16, p.ho - ty * 16) : 0, wv = min(16, p.wo - tx * 16);
          const bool colok = j < wv;
          int t1 = 0; float t2 = 0.0f; int mn = INT32_MAX, mx = INT32_MIN;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = colok && (4 * g + r) < hv;
            const int v = acc[r]; const float f = (float)v;
            t1 += ok ? v : 0; t2 = ok ? fmaf(f, f, t2) : t2;
            mn = ok ? min(mn, v) : mn; mx = ok ? max(mx, v) : mx;
          }
          s1[q] += t1; s2[q] += (double)t2; smn[q] = min(smn[q], mn); smx[q] = max(smx[q], mx);
        } else {
          const float cA = __uint_as_float(ct.y), cB = __uint_as_float(ct.z);
          uint32_t packed = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float yv = CVT ? cA * (float)(acc[r] + (int)ct.z) : fmaf(cA, (float)acc[r], cB);
            const float qv = fminf(rintf(fmaxf(yv, relu_floor) * y_inv) + y_zpf, qcap);
            packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);
          }
          pk[sub][q & 3] = packed ^ 0x80808080u;
        }
      }
      if (EMIT && (q & 3) == 3) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
          // 4 channels x 4 rows of bytes -> 4 rows x (4 consecutive channels): dword writes into the NHWC out tile
          const uint32_t t0 = __builtin_amdgcn_perm(pk[sub][1], pk[sub][0], 0x05010400u), t1 = __builtin_amdgcn_perm(pk[sub][1], pk[sub][0], 0x07030602u);
          const uint32_t u0 = __builtin_amdgcn_perm(pk[sub][3], pk[sub][2], 0x05010400u), u1 = __builtin_amdgcn_perm(pk[sub][3], pk[sub][2], 0x07030602u);
          const uint32_t o[4] = {__builtin_amdgcn_perm(u0, t0, 0x05040100u), __builtin_amdgcn_perm(u0, t0, 0x07060302u),
                                 __builtin_amdgcn_perm(u1, t1, 0x05040100u), __builtin_amdgcn_perm(u1, t1, 0x07060302u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) *(uint32_t*)(nhwc + (sub * 256 + (4 * g + r) * 16 + j) * G::OUT_PITCH + lc0) = o[r];
        }
      }
    }
    if (EMIT && !(p.abl & 16)) {
      __syncthreads();
      // copy-out: 8-byte pieces (8 channels of one pixel), coalesced along the channels
      constexpr int C8 = CB / 8;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u = tid + 512 * q; const int c8 = u % C8, px2 = u / C8; const int sub = px2 >> 8, px = px2 & 255; const int row = px >> 4, col = px & 15;
        const int sp = wt * NSUB + sub;
        const int img = sp / tpi; const int tr = sp - img * tpi; const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int oy = ty * 16 + row, ox = tx * 16 + col;
        if (sp < p.nsp && oy < p.ho && ox < p.wo && c8 * 8 < cvalid)
          *(uint2*)(p.y + (((int64_t)img * p.ho + oy) * p.wo + ox) * p.c + cb * CB + c8 * 8) = *(const uint2*)(nhwc + px2 * G::OUT_PITCH + c8 * 8);
      }
    }
    __syncthreads();                  // the stage region and the planes are free for the next tile
  }

  if (MODE == DM_STATS) {
    // fold the 64 lanes of every channel, lane q keeps channel q's totals, one set of global atomics per wave
    long long m1 = 0; double m2 = 0.0; int mmn = INT32_MAX, mmx = INT32_MIN;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      long long a1 = s1[q]; double a2 = s2[q]; int mn = smn[q], mx = smx[q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); mn = min(mn, __shfl_xor(mn, o)); mx = max(mx, __shfl_xor(mx, o));
      }
      if (lane == q) { m1 = a1; m2 = a2; mmn = mn; mmx = mx; }
    }
    if (lane < NCH) {
      const int lc = (lane >> 2) * 32 + 4 * w + (lane & 3);
      if (lc < cvalid && mmn <= mmx) {
        const int ch = cb * CB + lc;
        long long* g_s1 = (long long*)stats_copy(p.stats, p.cpad); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
        int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
        atomicAdd((unsigned long long*)&g_s1[ch], (unsigned long long)m1); atomicAdd(&g_s2[ch], (unsigned long long)__double2ll_rn(m2));
        atomicMin(&g_mn[ch], mmn); atomicMax(&g_mx[ch], mmx);
      }
    }
    if (p.fin_on) {             // last workgroup done -> conv finalize in this launch (see frost_common.h)
      int* sflag = (int*)smem;
      if (last_block_done2(p.fin.counter, gridDim.x, sflag)) {
        float* sh = (float*)(smem + 16);
        conv_finalize_dev(p.stats, (int64_t)p.n * p.ho * p.wo, p.c, p.cpad, p.qx, p.fin.qrec_w, p.fin.wscale, p.fin.gamma, p.fin.beta, p.fin.rmean, p.fin.rvar,
                          p.fin.nbt, p.fin.training, p.fin.relu, p.fin.observe, 1, p.fin.coef, p.fin.qrec_y, tid, 512, sh);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
template <int K, int CB, int MODE>
static int launch_dwm(DwmP& p, hipStream_t s) {
  typedef DwmGeo<K, CB> G;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)k_dwm<K, CB, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  p.tiles_x = (p.wo + 15) / 16; p.tiles_y = (p.ho + 15) / 16; p.ncb = (p.c + CB - 1) / CB;
  const int64_t nsp = (int64_t)p.n * p.tiles_x * p.tiles_y;
  FROST_REQUIRE(nsp < ((int64_t)1 << 30), "dwm: too many tiles");
  p.nsp = (int)nsp;
  const int64_t nwt = (nsp + G::NSUB - 1) / G::NSUB;
  // workgroups per XCD: ~64 (two per CU), a multiple of the channel-block count so that a workgroup keeps its channel block
  int ngrp = 64 / p.ncb; if (ngrp < 1) ngrp = 1;
  const int per_xcd = (int)((nwt + 7) / 8); if (ngrp > per_xcd) ngrp = per_xcd < 1 ? 1 : per_xcd;
  while ((per_xcd + ngrp - 1) / ngrp > 256) ++ngrp;          // the statistics pass keeps int32 lane sums: at most 256 tiles x 2 sub-tiles x 4 outputs x 2^20 per lane
  p.ngrp = ngrp;
  hipLaunchKernelGGL((k_dwm<K, CB, MODE>), dim3(8 * ngrp * p.ncb), dim3(512), G::LDS_BYTES, s, p);
  return frost_check_launch("dwm");
}
// channel-block width: 32 where 64 would waste lanes (C = 32, 72, 96, 144, 168 ...) -- FROST_DWM_CB forces one
static int dwm_cb(int c) {
  static const int force = getenv("FROST_DWM_CB") ? atoi(getenv("FROST_DWM_CB")) : 0;
  if (force == 32 || force == 64) return force;
  return (round_up(c, 32) < round_up(c, 64) || c <= 192) ? 32 : 64;
}
template <int K, int MODE>
static int launch_dwm_cb(DwmP& p, hipStream_t s) { return dwm_cb(p.c) == 32 ? launch_dwm<K, 32, MODE>(p, s) : launch_dwm<K, 64, MODE>(p, s); }

// 1 if the MFMA path is switched on (FROST_DW_MFMA=1) and has an instance for this layer (stride 1).  Off by default: parity-equal to the stencil
// kernels (tests/test_gpu_paths.py) but, stand-alone, not faster than them -- see DESIGN.md (d): the NHWC -> plane transposition, the operand
// assembly and the tile-level synchronisation cost what the dot products saved; the formulation pays where the producer already holds the
// planes (the block-level kernels).
int frost_dwm_ok(int k, int stride, int c) {
  static const int on = getenv("FROST_DW_MFMA") ? atoi(getenv("FROST_DW_MFMA")) : 0;
  return on && stride == 1 && (k == 3 || k == 5) && (c % 8) == 0;
}
// mode: 0 statistics (+ folded finalize when fin != NULL), 1 emit, 2 emit in converted-inference form
int frost_dwm_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h, int w, int c, int k, int mode,
                  void* stats, const float* coef, const float* qrec_y, int relu, int8_t* y, const FrostFinDesc* fin, hipStream_t s) {
  DwmP p = {};
  p.x = x; p.qx = qrec_x; p.wq = wq_pack; p.wsum = wsum; p.n = n; p.h = h; p.w = w; p.c = c; p.cpad = round_up(c, 16);
  p.pad = (k - 1) / 2; p.ho = h; p.wo = w;
  p.inv_count = 1.0f / (float)((int64_t)n * p.ho * p.wo);
  p.stats = (uint8_t*)stats; p.coef = (float*)coef; p.qy = qrec_y; p.relu = relu; p.y = y; p.cvt = (mode == 2);
  static const int abl = getenv("FROST_DWM_ABL") ? atoi(getenv("FROST_DWM_ABL")) : 0; p.abl = abl;
  if (fin) { p.fin = *fin; p.fin_on = 1; p.coef = fin->coef; p.qy = fin->qrec_y; p.relu = fin->relu; }
  if (mode == 0) return k == 3 ? launch_dwm_cb<3, DM_STATS>(p, s) : launch_dwm_cb<5, DM_STATS>(p, s);
  if (mode == 2) return k == 3 ? launch_dwm_cb<3, DM_EMIT_CVT>(p, s) : launch_dwm_cb<5, DM_EMIT_CVT>(p, s);
  return k == 3 ? launch_dwm_cb<3, DM_EMIT>(p, s) : launch_dwm_cb<5, DM_EMIT>(p, s);
}
