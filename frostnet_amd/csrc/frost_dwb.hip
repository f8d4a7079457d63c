// Depthwise 3 x 3 stride-1 backward of the high-resolution layers in ONE sweep: dc pass + weight gradient + data gradient (the backward of nniqat.ConvBnReLU2d
// with groups = channels after its reduce pass: /root/reference/frostnet.py:96-101 conv2 of a bottleneck; formulas: k_dw3's, frost_dw3.hip).
//
// Today three launches (frost_dw_conv_bwd_dc_wgrad + frost_dw_dgrad: 9 bytes per element through HBM -- x 1, gy 2, dc written 2 and read 2, dx 2): dc exists only to
// be read back by the data gradient.  Here it never leaves registers: 5 bytes per element (x, gy in; dx out).
//
// Decomposition: "lane = channel, strip-streaming".  A WAVE owns (image, channel block, column strip of SW = 8 * (64 / CBW) columns, row chunk) and walks the rows top to
// bottom alone -- no workgroup barrier anywhere; the four waves of a workgroup only share the launch and the final weight-gradient fold.  A lane owns ONE channel and 8
// columns.  Per step (one row d) it
//   1. takes the next x row (d + 1) and the gy row d from the wave's LDS rings (filled by direct-to-LDS loads PD rows ahead; transposed LDS reads give the lane the 12 /
//      10 consecutive pixels of its own channel),
//   2. recomputes the integer conv of row d on its 8 columns AND the two neighbouring ones (v_dot4 over packed byte windows), applies the STE window and
//      dc = fma(gy, K1, fma(acc, E, F)), rounds to bf16 (k_dw3's expressions) -- the 3 x 10 dc window lives in registers, rolling,
//   3. weight gradient: wacc[ky][kx] += dc[d][c] * q[d-1+ky][c-1+kx] on its own 8 columns / own rows (lane-local sums across the whole persistent loop),
//   4. data gradient of row d - 1 from the dc rows d-2, d-1, d (k_dw3_dgrad's summation order: bit-identical dx when dc's rounding is round-to-nearest),
//   5. dx row -> [pixel][channel] in LDS -> one 16-byte store per lane.
// Cost of not sharing dc between lanes: the 2 halo columns per 8 (conv + epilogue x 1.25) and one halo row at each end of a row chunk; nothing else is recomputed.
// Out-of-image handling without fill passes or exec-masked copies: every copy reads a clamped (valid) address; x columns outside the image are replaced by the zero-point
// byte with one v_bfi per dword (lane-constant masks), x rows outside by a select; gy / dc outside are masked by `valid`.  Every step issues exactly 3 copies (+ 1 store),
// so the wait for the rows of step s is a counted s_waitcnt vmcnt (in-order retirement) that leaves the PD - 1 younger row bundles in flight.
// With stochastic rounding on, a dc element that two lanes need (strip / chunk borders) is drawn independently by each: both draws are unbiased (see sr_bf16).
#include "frost_common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef DWB_PD
#define DWB_PD 1          // rows of prefetch distance of the direct-to-LDS rings (measured: profiles/r05_dw_onesweep_ab.txt)
#endif
#ifndef DWB_LB1
#define DWB_LB1 3         // waves per SIMD the register allocator leaves room for: stride-1 kernel / stride-2 kernels
#endif
#ifndef DWB_LB2
#define DWB_LB2 2
#endif
typedef int v2i_b __attribute__((ext_vector_type(2)));
typedef short v4s_b __attribute__((ext_vector_type(4)));

struct DwbP {
  const int8_t* x; const float* qx; const int8_t* wq; const int32_t* wsum; const float* qw; const float* wscale;
  const float* coef; const float* qy; const uint16_t* gout; uint16_t* dx; float* dwq;
  int n, h, w, c, cpad, relu, sr; float inv_count;
  int ncb, nstrips, nchunks, rc;          // channel blocks, column strips per row, row chunks per image, rows per chunk (stride 2: output rows / columns)
  int ho, wo;
  // conv1 fold (stride-2 kernels, C1D > 0): the reduce pass of the pointwise layer that PRODUCED x (its S1 / S2) rides on the dx rows as they are formed
  const int8_t* c1_x; const float* c1_qx; const int8_t* c1_wq; const int32_t* c1_wsum; float* c1_coef; int c1_relu;
};

__device__ __forceinline__ void dwb_glds16(const void* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
#define DWB_WAIT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void dwb_wait_vm(int n) {          // n: wave-uniform LOWER bound of the VMEM instructions issued after the ones waited for
  switch (n) {
    DWB_WAIT_CASE(1) DWB_WAIT_CASE(2) DWB_WAIT_CASE(3) DWB_WAIT_CASE(4) DWB_WAIT_CASE(5) DWB_WAIT_CASE(6) DWB_WAIT_CASE(7) DWB_WAIT_CASE(8)
    DWB_WAIT_CASE(9) DWB_WAIT_CASE(10) DWB_WAIT_CASE(11) DWB_WAIT_CASE(12) DWB_WAIT_CASE(13) DWB_WAIT_CASE(14) DWB_WAIT_CASE(15) DWB_WAIT_CASE(16)
    DWB_WAIT_CASE(17) DWB_WAIT_CASE(18) DWB_WAIT_CASE(19) DWB_WAIT_CASE(20) DWB_WAIT_CASE(21) DWB_WAIT_CASE(22) DWB_WAIT_CASE(23) DWB_WAIT_CASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
// 8 consecutive pixels of this lane's channel from an int8 [pixel][CBW] row (ds_read_b64_tr_b8; tr8_raw of frost_dw3.hip)
template <int CBW>
__device__ __forceinline__ v2i_b dwb_tr8(const uint8_t* row, int col0, int lane) {
  const int jp = lane & 15, G = (lane >> 4) & (CBW / 16 - 1);
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i_b __attribute__((address_space(3)))*)(row + (col0 + (jp >> 1)) * CBW + 16 * G + 8 * (jp & 1)));
}
// 4 consecutive pixels of this lane's channel from a bf16 [pixel][CBW] row
template <int CBW>
__device__ __forceinline__ void dwb_tr16(const uint8_t* row, int col0, int lane, float* out4) {
  const int jp = lane & 15, G = (lane >> 4) & (CBW / 16 - 1);
  const v4s_b raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_b __attribute__((address_space(3)))*)(row + ((col0 + (jp >> 2)) * CBW + 16 * G + 4 * (jp & 3)) * 2));
  out4[0] = bf2f((uint16_t)raw[0]); out4[1] = bf2f((uint16_t)raw[1]); out4[2] = bf2f((uint16_t)raw[2]); out4[3] = bf2f((uint16_t)raw[3]);
}

template <int CBW, int PD>
struct DwbGeo {
  static constexpr int HALF = 64 / CBW, SW = 8 * HALF, XPX = SW + 4, GPX = SW + 2;
  static constexpr int XR = XPX * CBW, GR = GPX * CBW * 2, NXS = PD + 2, NGS = PD + 1;      // bytes per x / gy row record, ring slots
  static constexpr int XU = XR / 16, GU = GR / 16;                                            // 16-byte units per row record
  static constexpr int XSL = 4 * CBW, GSL = 4 * CBW;                                          // slack behind the rings: the last transposed read of a row runs past its record
  static constexpr int X_OFF = 0, G_OFF = NXS * XR + XSL, O_OFF = G_OFF + NGS * GR + GSL, WAVE_LDS = O_OFF + 1024;
  static_assert(XU <= 64 && GU > 64 && GU <= 128, "one copy instruction per x row, two per gy row");
};

template <int CBW, int PD>
__global__ __launch_bounds__(256, DWB_LB1) void k_dwb_s1(const DwbP p) {
  using G = DwbGeo<CBW, PD>;
  constexpr int SW = G::SW, XR = G::XR, GR = G::GR, NXS = G::NXS, NGS = G::NGS;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* const wl = smem + (size_t)wv * G::WAVE_LDS;
  uint8_t* const xring = wl + G::X_OFF; uint8_t* const gring = wl + G::G_OFF; uint8_t* const orow = wl + G::O_OFF;
  const uint32_t xring_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)xring);
  const uint32_t gring_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)gring);
  const int lc = lane & (CBW - 1), hf = lane / CBW;

  // ---- work map: workgroup b runs on XCD b % 8; an XCD owns a contiguous eighth of the images (strips / chunks of an image share their halos through ONE L2);
  //      a workgroup keeps ONE channel block (its lanes' weight-gradient sums live across the whole loop), its 4 waves take neighbouring tasks
  const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3, nl = (int)gridDim.x >> 3;
  const int cb = li % p.ncb, lwg = li / p.ncb, nlc = nl / p.ncb;
  const int img_lo = (int)(((int64_t)p.n * xcd) >> 3), img_hi = (int)(((int64_t)p.n * (xcd + 1)) >> 3);
  const int pit = p.nstrips * p.nchunks, ntask = (img_hi - img_lo) * pit;
  const int ch = cb * CBW + lc; const bool chok = ch < p.c;

  // ---- per-lane (= per-channel) constants
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]);
  const uint32_t zp4 = (uint32_t)((zpx - 128) & 255) * 0x01010101u;
  int wpk[3]; float wf[9];
  {
    int8_t taps[9];
    load_taps_i8<9>(p.wq, p.cpad, ch, chok, taps);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      uint32_t pk = 0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) { pk |= (uint32_t)(uint8_t)taps[ky * 3 + kx] << (8 * kx); wf[ky * 3 + kx] = (float)taps[ky * 3 + kx]; }
      wpk[ky] = (int)pk;
    }
  }
  const int chc = chok ? ch : 0;
  const int acc0 = chok ? (128 - zpx) * p.wsum[chc] : 0;
  const float sw = (p.wscale && chok) ? p.wscale[chc] : p.qw[FROST_Q_SCALE];
  float cA = 0, cB = 0, cK1 = 0, cE = 0, cF = 0;
  if (chok) {
    cA = p.coef[FROST_COEF_A * p.cpad + ch]; cB = p.coef[FROST_COEF_B * p.cpad + ch];
    const float m = p.coef[FROST_COEF_M * p.cpad + ch], cR = p.coef[FROST_COEF_R * p.cpad + ch];
    cK1 = p.coef[FROST_COEF_K1 * p.cpad + ch];
    cE = -cK1 * (s12_sum(p.coef, p.cpad, 1, ch) * p.inv_count) * cR;
    cF = -cK1 * (s12_sum(p.coef, p.cpad, 0, ch) * p.inv_count) - cE * m;
  }
  const float y_inv = 1.0f / p.qy[FROST_Q_SCALE];
  float t_lo = 0.0f, t_hi;
  {
    const int zpy = __float_as_int(p.qy[FROST_Q_ZP]), qhi = q_hi(p.qy);
    const float hi0 = (float)qhi + 0.5f - (float)zpy;
    t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
    if (!p.relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  }
  uint32_t rng = sr_seed(blockIdx.x, threadIdx.x);
  const bool sr_on = p.sr != 0;
  float wacc[9]; float sdc = 0.0f;
#pragma unroll
  for (int t = 0; t < 9; ++t) wacc[t] = 0.0f;

  const int64_t rowpitch = (int64_t)p.w * p.c;
  for (int tt = lwg * 4 + wv; tt < ntask; tt += nlc * 4) {
    const int img = img_lo + tt / pit; const int rem = tt - (tt / pit) * pit;
    const int chunk = rem / p.nstrips, strip = rem - chunk * p.nstrips;
    const int r0 = chunk * p.rc, r1 = min(r0 + p.rc, p.h), c0 = strip * SW;
    const int NS = (r1 - r0) + 2;
    const int8_t* const ximg = p.x + (int64_t)img * p.h * rowpitch;
    const uint16_t* const gimg = p.gout + (int64_t)img * p.h * rowpitch;
    uint16_t* const dimg = p.dx + (int64_t)img * p.h * rowpitch;
    // copy plans of this strip (clamped: always a valid address): x unit `lane`, gy units `lane` and 64 + `lane`, dx unit `lane`
    int xoff, goff0, goff1, ooff; bool ook;
    {
      constexpr int UPX = CBW / 16, UPG = CBW / 8;
      int px = lane / UPX, cu = lane % UPX, col = min(max(c0 - 2 + px, 0), p.w - 1), cn = cb * CBW + cu * 16; if (cn >= p.c) cn = 0;
      xoff = col * p.c + cn;
      px = lane / UPG; cu = lane % UPG; col = min(max(c0 - 1 + px, 0), p.w - 1); cn = cb * CBW + cu * 8; if (cn >= p.c) cn = 0;
      goff0 = col * p.c + cn;
      px = (64 + lane) / UPG; cu = (64 + lane) % UPG; col = min(max(c0 - 1 + px, 0), p.w - 1); cn = cb * CBW + cu * 8; if (cn >= p.c) cn = 0;
      goff1 = col * p.c + cn;
      px = lane / UPG; cu = lane % UPG; col = c0 + px; cn = cb * CBW + cu * 8;
      ook = col < p.w && cn < p.c; ooff = col * p.c + cn;
    }
    // lane-constant masks: bytes of the 12-pixel x window inside the image (window pixel t = column lc0 - 2 + t), dc columns j = 0 .. 9 (column lc0 - 1 + j) inside the image
    const int lc0 = c0 + 8 * hf;
    uint32_t xm[3]; uint32_t cmask = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) { const int col = lc0 - 2 + 4 * i + b; if (col >= 0 && col < p.w) m |= 0xffu << (8 * b); }
      xm[i] = m;
    }
#pragma unroll
    for (int j = 0; j < 10; ++j) { const int col = lc0 - 1 + j; if (col >= 0 && col < p.w && chok) cmask |= 1u << j; }

    auto issue_x = [&](int row, int slot) __attribute__((always_inline)) {        // x row `row` -> ring slot (1 copy instruction)
      const int rr = min(max(row, 0), p.h - 1);
      if (lane < G::XU) dwb_glds16(ximg + (int64_t)rr * rowpitch + xoff, xring_a + (uint32_t)(slot * XR));
    };
    auto issue_g = [&](int row, int slot) __attribute__((always_inline)) {        // gy row -> ring slot (2 copy instructions)
      const int rr = min(max(row, 0), p.h - 1);
      const uint16_t* base = gimg + (int64_t)rr * rowpitch;
      dwb_glds16(base + goff0, gring_a + (uint32_t)(slot * GR));
      if (lane < G::GU - 64) dwb_glds16(base + goff1, gring_a + (uint32_t)(slot * GR + 1024));
    };
    auto load_x = [&](int row, int slot, uint32_t* d3) __attribute__((always_inline)) {     // the lane's 12-pixel window of an x row, zero-point outside the image
      const uint8_t* rp = xring + slot * XR;
      const v2i_b a = dwb_tr8<CBW>(rp, 8 * hf, lane), b = dwb_tr8<CBW>(rp, 8 * hf + 8, lane);
      const bool rok = row >= 0 && row < p.h;
      d3[0] = rok ? (((uint32_t)a[0] & xm[0]) | (zp4 & ~xm[0])) : zp4;
      d3[1] = rok ? (((uint32_t)a[1] & xm[1]) | (zp4 & ~xm[1])) : zp4;
      d3[2] = rok ? (((uint32_t)b[0] & xm[2]) | (zp4 & ~xm[2])) : zp4;
    };

    // ---- prologue: x rows r0-2, r0-1 and the bundles of steps 0 .. PD-1 (bundle s = x row r0 + s, gy row r0 - 1 + s); x slot of row r = (r - (r0 - 2)) % NXS
    issue_x(r0 - 2, 0); issue_x(r0 - 1, 1);
#pragma unroll
    for (int b = 0; b < PD; ++b) { issue_x(r0 + b, (b + 2) % NXS); issue_g(r0 - 1 + b, b % NGS); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t xd[3][3]; float dcw[3][10];
    load_x(r0 - 2, 0, xd[0]); load_x(r0 - 1, 1, xd[1]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 10; ++j) dcw[i][j] = 0.0f;

    // one step; PH = s % 3 (compile time): xd[PH] = x row d-1, xd[PH+1] = row d, xd[PH+2] <- row d+1; dcw[PH] = dc row d-2, dcw[PH+1] = d-1, dcw[PH+2] <- d
    auto step = [&](int s, auto phc) __attribute__((always_inline)) {
      constexpr int PH = decltype(phc)::value, I0 = PH % 3, I1 = (PH + 1) % 3, I2 = (PH + 2) % 3;
      const int d = r0 - 1 + s;
      if (s > 0) dwb_wait_vm(3 * (PD - 1) + min(max(s - 2, 0), PD));
      issue_x(r0 + s + PD, (s + PD + 2) % NXS); issue_g(d + PD, (s + PD) % NGS);
      load_x(d + 1, (s + 2) % NXS, xd[I2]);
      const bool drow = d >= 0 && d < p.h;
      if (drow) {
        float gq[12];
        const uint8_t* gp = gring + (s % NGS) * GR;
        dwb_tr16<CBW>(gp, 8 * hf, lane, gq); dwb_tr16<CBW>(gp, 8 * hf + 4, lane, gq + 4); dwb_tr16<CBW>(gp, 8 * hf + 8, lane, gq + 8);
        int acc[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[j] = acc0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const uint32_t* r = xd[(PH + ky) % 3];
#pragma unroll
          for (int j = 0; j < 10; ++j) {
            const uint32_t hi = (j / 4 + 1 < 3) ? r[(j / 4 + 1) % 3] : 0u;
            const int win = (j % 4 == 0) ? (int)r[j / 4] : (int)__builtin_amdgcn_alignbyte(hi, r[j / 4], j % 4);
            acc[j] = __builtin_amdgcn_sdot4(win, wpk[ky], acc[j], false);
          }
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const bool valid = (cmask >> j) & 1u;
          const float v = (float)acc[j];
          const float tq = fmaf(cA, v, cB) * y_inv;
          const float gy = (valid && tq > t_lo && tq <= t_hi) ? gq[j] : 0.0f;
          const float dcf = fmaf(gy, cK1, fmaf(v, cE, cF));
          const uint32_t db = __float_as_uint(dcf), dr = sr_next16(rng);
          const uint32_t hb = (db + (sr_on ? dr : 0x7fffu + ((db >> 16) & 1u))) >> 16;
          dcw[I2][j] = valid ? __uint_as_float(hb << 16) : 0.0f;
        }
        if (d >= r0 && d < r1) {       // weight gradient: own rows, own columns (j = 1 .. 8); q = unsigned index, the zero point comes off through sdc at the end
#pragma unroll
          for (int j = 1; j <= 8; ++j) sdc += dcw[I2][j];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const uint32_t* r = xd[(PH + ky) % 3];
            float xf[12];
#pragma unroll
            for (int t = 1; t <= 10; ++t) xf[t] = (float)(((r[t >> 2] ^ 0x80808080u) >> (8 * (t & 3))) & 255u);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int j = 1; j <= 8; ++j) wacc[ky * 3 + kx] = fmaf(dcw[I2][j], xf[j + kx], wacc[ky * 3 + kx]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 10; ++j) dcw[I2][j] = 0.0f;
      }
      if (s >= 2) {      // data gradient of row d - 1: dx[c0 + i] = s_w * sum dc[d - ky][i + 2 - kx] * wq[ky][kx], dc rows ascending, kx ascending (k_dw3_dgrad's order)
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = 0.0f;
#pragma unroll
        for (int ky = 2; ky >= 0; --ky) {
          const float* row = dcw[(PH + 2 - ky) % 3];
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a[i] = fmaf(row[i + 2 - kx], wf[ky * 3 + kx], a[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *(uint16_t*)(orow + ((8 * hf + i) * CBW + lc) * 2) = (uint16_t)cvt_pk_bf16(a[i] * sw, 0.0f);
        const uint4 v = *(const uint4*)(orow + lane * 16);
        if (ook) *(uint4*)(dimg + (int64_t)(d - 1) * rowpitch + ooff) = v;
      }
    };
    for (int s3 = 0; s3 < NS; s3 += 3) {
      step(s3, std::integral_constant<int, 0>());
      if (s3 + 1 < NS) step(s3 + 1, std::integral_constant<int, 1>());
      if (s3 + 2 < NS) step(s3 + 2, std::integral_constant<int, 2>());
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing (unused) copies have landed before the next task re-uses the rings
  }

  // ---- dW[c][tap] += s_x * (sum dc * q - zp * sum dc): the 4 waves' (and, for 32-channel blocks, both halves') partials through LDS, one atomic per (channel, tap) and workgroup
  __syncthreads();
  float* red = (float*)smem;                       // [4][9][64]
  const float zpf = (float)zpx;
#pragma unroll
  for (int t = 0; t < 9; ++t) red[(wv * 9 + t) * 64 + lane] = wacc[t] - zpf * sdc;
  __syncthreads();
  const float sx = p.qx[FROST_Q_SCALE];
  for (int i = tid; i < 9 * CBW; i += 256) {
    const int t = i / CBW, l2 = i % CBW; const int c2 = cb * CBW + l2;
    float sum = 0.0f;
    for (int w2 = 0; w2 < 4; ++w2)
      for (int l3 = l2; l3 < 64; l3 += CBW) sum += red[(w2 * 9 + t) * 64 + l3];
    if (c2 < p.c) atomicAdd(dwq_dst(p.dwq, (int64_t)p.c * 9) + (int64_t)c2 * 9 + t, sum * sx);
  }
}

// ================================================================================================ stride 2 (k = 3, 5)
// The same strip-streaming scheme for the stride-2 layers (112 -> 56 k3, 56 -> 28 / 28 -> 14 / 14 -> 7 k5).  Today four launches (dc pass, weight gradient, data gradient:
// x read twice, dc written once and read twice, dx written: 2 + 1 + 1 + 2 ... = 6 bytes per INPUT element); here 3.5 (x 1, gy 0.5, dx 2).  A lane owns 8 OUTPUT columns = 16
// input columns; a step is one dc row m: two new x rows (the other k - 2 are carried in registers), one gy row, NDC = 9 (k3) / 10 (k5) dc columns (one halo column on the
// right, for k5 one on the left too), and the two dx rows that became final -- rows 2m-1, 2m (k3: dc rows m-1, m) or 2m-2, 2m-1 (k5: dc rows m-2 .. m).  Only the taps whose
// parity matches contribute to a dx element (2.25 of 9, 6.25 of 25 on average), in k_dw3_dgrad's order (dc rows ascending, kx ascending).
template <int K, int CBW, int PD, int C1D = 0>
struct DwbGeo2 {
  static constexpr int PAD = (K - 1) / 2, LH = (K == 5) ? 1 : 0, NDC = 9 + LH, XL = (K == 3) ? 1 : 4, NDR = (K == 3) ? 2 : 3;
  static constexpr int NXW = 2 * (NDC - 1) + K, NXD = 6;                                     // x window of a lane: 19 / 23 pixels; 3 transposed reads = 24 pixels = 6 dwords
  static constexpr int HALF = 64 / CBW, SWO = 8 * HALF, SWI = 16 * HALF;
  static constexpr int XPX = SWI - 16 + NXW, GPX = SWO + 1 + LH;
  static constexpr int XR = XPX * CBW, GR = GPX * CBW * 2, OR = SWI * CBW * 2;               // bytes per x / gy row record, dx row (2 KB)
  static constexpr int NXS = 2 * PD + ((K == 3) ? 2 : 3), NGS = PD + 1;
  static constexpr int XU = XR / 16, GU = GR / 16;
  static constexpr int XSL = 8 * CBW, GSL = 4 * CBW;                                          // slack: the last transposed read of a row runs past its record
  static constexpr int C1R = SWI * C1D * 4, C1U = C1R / 16, N1S = 2 * (PD + 1);                // conv1 fold: bytes of an x0 row segment [SWI pixels][Cin], its 16-byte units, ring slots
  static constexpr int X_OFF = 0, G_OFF = NXS * XR + XSL, O_OFF = G_OFF + NGS * GR + GSL, C1_OFF = O_OFF + OR, WAVE_LDS = C1_OFF + N1S * C1R;
  static constexpr int NDMA = 6 + ((C1D > 0) ? 2 : 0);
  static_assert(XU > 64 && XU <= 128 && GU > 64 && GU <= 128 && OR == 2048 && C1U <= 64, "two copy instructions per x row, per gy row and per dx row, one per x0 row");
};

template <int K, int CBW, int PD, int C1D>
__global__ __launch_bounds__(256, DWB_LB2) void k_dwb_s2(const DwbP p) {
  using G = DwbGeo2<K, CBW, PD, C1D>;
  constexpr bool C1 = C1D > 0; constexpr int CIN1 = C1D * 4;
  constexpr int PAD = G::PAD, LH = G::LH, NDC = G::NDC, XL = G::XL, NDR = G::NDR, NXD = G::NXD, SWO = G::SWO, SWI = G::SWI;
  constexpr int XR = G::XR, GR = G::GR, NXS = G::NXS, NGS = G::NGS, KK = K * K, NPK = (K == 3) ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* const wl = smem + (size_t)wv * G::WAVE_LDS;
  uint8_t* const xring = wl + G::X_OFF; uint8_t* const gring = wl + G::G_OFF; uint8_t* const orow = wl + G::O_OFF;
  const uint32_t xring_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)xring);
  const uint32_t gring_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)gring);
  const int lc = lane & (CBW - 1), hf = lane / CBW;
  const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3, nl = (int)gridDim.x >> 3;
  const int cb = li % p.ncb, lwg = li / p.ncb, nlc = nl / p.ncb;
  const int img_lo = (int)(((int64_t)p.n * xcd) >> 3), img_hi = (int)(((int64_t)p.n * (xcd + 1)) >> 3);
  const int pit = p.nstrips * p.nchunks, ntask = (img_hi - img_lo) * pit;
  const int ch = cb * CBW + lc; const bool chok = ch < p.c;

  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]);
  const uint32_t zp4 = (uint32_t)((zpx - 128) & 255) * 0x01010101u;
  int wpk[K][NPK]; float wf[KK];
  {
    int8_t taps[KK];
    load_taps_i8<KK>(p.wq, p.cpad, ch, chok, taps);
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const uint32_t b = (uint32_t)(uint8_t)taps[ky * K + kx]; wf[ky * K + kx] = (float)taps[ky * K + kx];
        if (kx < 4) lo |= b << (8 * kx); else hi |= b;
      }
      wpk[ky][0] = (int)lo; if (NPK > 1) wpk[ky][NPK - 1] = (int)hi;
    }
  }
  const int chc = chok ? ch : 0;
  const int acc0 = chok ? (128 - zpx) * p.wsum[chc] : 0;
  const float sw = (p.wscale && chok) ? p.wscale[chc] : p.qw[FROST_Q_SCALE];
  float cA = 0, cB = 0, cK1 = 0, cE = 0, cF = 0;
  if (chok) {
    cA = p.coef[FROST_COEF_A * p.cpad + ch]; cB = p.coef[FROST_COEF_B * p.cpad + ch];
    const float m = p.coef[FROST_COEF_M * p.cpad + ch], cR = p.coef[FROST_COEF_R * p.cpad + ch];
    cK1 = p.coef[FROST_COEF_K1 * p.cpad + ch];
    cE = -cK1 * (s12_sum(p.coef, p.cpad, 1, ch) * p.inv_count) * cR;
    cF = -cK1 * (s12_sum(p.coef, p.cpad, 0, ch) * p.inv_count) - cE * m;
  }
  const float y_inv = 1.0f / p.qy[FROST_Q_SCALE];
  float t_lo = 0.0f, t_hi;
  {
    const int zpy = __float_as_int(p.qy[FROST_Q_ZP]), qhi = q_hi(p.qy);
    const float hi0 = (float)qhi + 0.5f - (float)zpy;
    t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
    if (!p.relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  }
  uint32_t rng = sr_seed(blockIdx.x, threadIdx.x);
  const bool sr_on = p.sr != 0;
  float wacc[KK]; float sdc = 0.0f;
#pragma unroll
  for (int t = 0; t < KK; ++t) wacc[t] = 0.0f;

  // conv1 fold: this lane's row of conv1's fake-quantised weights (A-fragment pack [ct][ks = 0][kb * 16 + m][16 B]: channel ct * 16 + m, k = kb * 16 ..), its coefficient rows and
  // STE window -- k_pw's reduce pass (frost_pw.hip, M_BRED) evaluated per dx element: S1 += g, S2 += g * xhat with g = the dx value BEFORE its bf16 rounding
  uint8_t* const c1ring = wl + G::C1_OFF;
  const uint32_t c1ring_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)c1ring);
  int w1[C1 ? C1D : 1]; int acc01 = 0; float a1A = 0, a1B = 0, a1R = 0, a1MR = 0, y_inv1 = 1.0f, t_lo1 = 0.0f, t_hi1 = 0.0f, c1s1 = 0.0f, c1s2 = 0.0f;
  if (C1) {
    const uint4* wp = (const uint4*)p.c1_wq + (size_t)(chc >> 4) * 64 + (chc & 15);
    const uint4 q0 = wp[0]; w1[0] = (int)q0.x; w1[1] = (int)q0.y; w1[2] = (int)q0.z; w1[C1D > 3 ? 3 : 0] = (int)q0.w;
    if (C1D > 4) { const uint4 q1 = wp[16]; w1[C1D > 4 ? 4 : 0] = (int)q1.x; w1[C1D > 5 ? 5 : 0] = (int)q1.y; }
    if (!chok) { for (int i = 0; i < C1D; ++i) w1[i] = 0; }
    const int zp0 = __float_as_int(p.c1_qx[FROST_Q_ZP]);
    acc01 = chok ? (128 - zp0) * p.c1_wsum[chc] : 0;
    if (chok) {
      a1A = p.c1_coef[FROST_COEF_A * p.cpad + ch]; a1B = p.c1_coef[FROST_COEF_B * p.cpad + ch];
      const float m = p.c1_coef[FROST_COEF_M * p.cpad + ch]; a1R = p.c1_coef[FROST_COEF_R * p.cpad + ch]; a1MR = -m * a1R;
    }
    y_inv1 = 1.0f / p.qx[FROST_Q_SCALE];                      // conv1's output record IS this layer's input record
    const int qhi = q_hi(p.qx);
    const float hi0 = (float)qhi + 0.5f - (float)zpx;
    t_hi1 = ((qhi - zpx) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
    if (!p.c1_relu) { const float lo0 = -(float)zpx - 0.5f; t_lo1 = (zpx & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  }

  const int64_t xpitch = (int64_t)p.w * p.c, gpitch = (int64_t)p.wo * p.c;
  for (int tt = lwg * 4 + wv; tt < ntask; tt += nlc * 4) {
    const int img = img_lo + tt / pit; const int rem = tt - (tt / pit) * pit;
    const int chunk = rem / p.nstrips, strip = rem - chunk * p.nstrips;
    const int m0 = chunk * p.rc, m1 = min(m0 + p.rc, p.ho);           // own dc rows [m0, m1); own dx rows [2 m0, 2 m1)
    const int oc0s = strip * SWO, ic0s = 2 * oc0s;
    const int NS = (m1 - m0) + 1 + LH;                                 // dc rows m0 - LH .. m1
    const int rbase = 2 * (m0 - LH) - PAD;                             // first x row of step 0
    const int8_t* const ximg = p.x + (int64_t)img * p.h * xpitch;
    const uint16_t* const gimg = p.gout + (int64_t)img * p.ho * gpitch;
    uint16_t* const dimg = p.dx + (int64_t)img * p.h * xpitch;
    int xoff0, xoff1, goff0, goff1, ooff0, ooff1; bool ook0, ook1;
    {
      constexpr int UPX = CBW / 16, UPG = CBW / 8;
      auto xo = [&](int u) { const int px = u / UPX, cu = u % UPX, col = min(max(ic0s - XL + px, 0), p.w - 1); int cn = cb * CBW + cu * 16; if (cn >= p.c) cn = 0; return col * p.c + cn; };
      auto go = [&](int u) { const int px = u / UPG, cu = u % UPG, col = min(max(oc0s - LH + px, 0), p.wo - 1); int cn = cb * CBW + cu * 8; if (cn >= p.c) cn = 0; return col * p.c + cn; };
      xoff0 = xo(lane); xoff1 = xo(64 + lane); goff0 = go(lane); goff1 = go(64 + lane);
      int px = lane / UPG, cu = lane % UPG, col = ic0s + px, cn = cb * CBW + cu * 8; ook0 = col < p.w && cn < p.c; ooff0 = col * p.c + cn;
      px = (64 + lane) / UPG; cu = (64 + lane) % UPG; col = ic0s + px; cn = cb * CBW + cu * 8; ook1 = col < p.w && cn < p.c; ooff1 = col * p.c + cn;
    }
    const int nst_row = 1 + (__builtin_amdgcn_ballot_w64(ook1) != 0ull ? 1 : 0);        // store instructions a dx row certainly issues (the counted wait needs a lower bound)
    const int ic0 = ic0s + 16 * hf, oc0 = oc0s + 8 * hf;
    int c1off = 0; uint32_t pmask = 0;                 // conv1 fold: byte offset of this lane's 16-byte unit inside the strip's x0 row segment (clamped into the image row), own dx pixels inside the map
    const int8_t* c1img = nullptr;
    if (C1) {
      c1off = min(lane * 16, (p.w - ic0s) * CIN1 - 16);
      c1img = p.c1_x + ((int64_t)img * p.h * p.w + ic0s) * CIN1;
#pragma unroll
      for (int i = 0; i < 16; ++i) if (ic0 + i < p.w && chok) pmask |= 1u << i;
    }
    uint32_t xm[NXD]; uint32_t cmask = 0;
#pragma unroll
    for (int i = 0; i < NXD; ++i) {
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) { const int col = ic0 - XL + 4 * i + b; if (col >= 0 && col < p.w) m |= 0xffu << (8 * b); }
      xm[i] = m;
    }
#pragma unroll
    for (int j = 0; j < NDC; ++j) { const int col = oc0 - LH + j; if (col >= 0 && col < p.wo && chok) cmask |= 1u << j; }

    auto issue_x = [&](int row, int slot) __attribute__((always_inline)) {
      const int rr = min(max(row, 0), p.h - 1);
      const int8_t* base = ximg + (int64_t)rr * xpitch;
      dwb_glds16(base + xoff0, xring_a + (uint32_t)(slot * XR));
      if (lane < G::XU - 64) dwb_glds16(base + xoff1, xring_a + (uint32_t)(slot * XR + 1024));
    };
    auto issue_g = [&](int row, int slot) __attribute__((always_inline)) {
      const int rr = min(max(row, 0), p.ho - 1);
      const uint16_t* base = gimg + (int64_t)rr * gpitch;
      dwb_glds16(base + goff0, gring_a + (uint32_t)(slot * GR));
      if (lane < G::GU - 64) dwb_glds16(base + goff1, gring_a + (uint32_t)(slot * GR + 1024));
    };
    auto issue_c1 = [&](int step) __attribute__((always_inline)) {      // the x0 rows of the two dx rows that step `step` finalises
      if (C1) {
        constexpr int IO0 = (K == 3) ? -1 : -2;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int iy = min(max(2 * (m0 - LH + step) + IO0 + rr, 0), p.h - 1);
          if (lane < G::C1U) dwb_glds16(c1img + (int64_t)iy * p.w * CIN1 + c1off, c1ring_a + (uint32_t)(((step % (PD + 1)) * 2 + rr) * G::C1R));
        }
      }
    };
    auto load_x = [&](int row, uint32_t* d6) __attribute__((always_inline)) {
      const int slot = (row - rbase) % NXS;
      const uint8_t* rp = xring + slot * XR;
      const v2i_b a = dwb_tr8<CBW>(rp, 16 * hf, lane), b = dwb_tr8<CBW>(rp, 16 * hf + 8, lane), c = dwb_tr8<CBW>(rp, 16 * hf + 16, lane);
      const bool rok = row >= 0 && row < p.h;
      const uint32_t raw[6] = {(uint32_t)a[0], (uint32_t)a[1], (uint32_t)b[0], (uint32_t)b[1], (uint32_t)c[0], (uint32_t)c[1]};
#pragma unroll
      for (int i = 0; i < NXD; ++i) d6[i] = rok ? ((raw[i] & xm[i]) | (zp4 & ~xm[i])) : zp4;
    };
    // ---- prologue: the k - 2 carried rows of step 0 and the bundles of steps 0 .. PD-1 (bundle s = x rows rbase + (K-2) + 2s, +1 and gy row m0 - LH + s)
#pragma unroll
    for (int i = 0; i < K - 2; ++i) issue_x(rbase + i, i % NXS);
#pragma unroll
    for (int b = 0; b < PD; ++b) {
      issue_x(rbase + (K - 2) + 2 * b, (K - 2 + 2 * b) % NXS); issue_x(rbase + (K - 2) + 2 * b + 1, (K - 2 + 2 * b + 1) % NXS);
      issue_g(m0 - LH + b, b % NGS); issue_c1(b);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t xd[K][NXD]; float dcw[NDR][NDC];
#pragma unroll
    for (int i = 0; i < K - 2; ++i) load_x(rbase + i, xd[i + 2]);          // the step shifts them down by two before it loads its own two rows
#pragma unroll
    for (int i = 0; i < NDR; ++i)
#pragma unroll
      for (int j = 0; j < NDC; ++j) dcw[i][j] = 0.0f;
    int sth[PD];                 // stores issued by the last PD steps (for the counted wait)
#pragma unroll
    for (int i = 0; i < PD; ++i) sth[i] = 0;

#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      const int m = m0 - LH + s;
      if (s > 0) { int ny = G::NDMA * (PD - 1); for (int i = 0; i < PD; ++i) ny += sth[i]; dwb_wait_vm(ny); }
      { const int r = rbase + (K - 2) + 2 * (s + PD); issue_x(r, (r - rbase) % NXS); issue_x(r + 1, (r + 1 - rbase) % NXS); issue_g(m + PD, (s + PD) % NGS); issue_c1(s + PD); }
#pragma unroll
      for (int ky = 0; ky < K - 2; ++ky)
#pragma unroll
        for (int i = 0; i < NXD; ++i) xd[ky][i] = xd[ky + 2][i];
      load_x(2 * m - PAD + K - 2, xd[K - 2]); load_x(2 * m - PAD + K - 1, xd[K - 1]);
#pragma unroll
      for (int i = 0; i + 1 < NDR; ++i)
#pragma unroll
        for (int j = 0; j < NDC; ++j) dcw[i][j] = dcw[i + 1][j];
      const bool drow = m >= 0 && m < p.ho;
      if (drow) {
        float gq[12];
        const uint8_t* gp = gring + (s % NGS) * GR;
        dwb_tr16<CBW>(gp, 8 * hf, lane, gq); dwb_tr16<CBW>(gp, 8 * hf + 4, lane, gq + 4); dwb_tr16<CBW>(gp, 8 * hf + 8, lane, gq + 8);
        int acc[NDC];
#pragma unroll
        for (int j = 0; j < NDC; ++j) acc[j] = acc0;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll
          for (int j = 0; j < NDC; ++j) {
#pragma unroll
            for (int q = 0; q < NPK; ++q) {
              const int o = 2 * j + 4 * q;                       // byte offset of the 4-byte window in the lane's row
              const uint32_t lo = xd[ky][o / 4], hi = (o / 4 + 1 < NXD) ? xd[ky][(o / 4 + 1) % NXD] : 0u;
              const int win = (o % 4 == 0) ? (int)lo : (int)__builtin_amdgcn_alignbyte(hi, lo, 2);
              acc[j] = __builtin_amdgcn_sdot4(win, wpk[ky][q], acc[j], false);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < NDC; ++j) {
          const bool valid = (cmask >> j) & 1u;
          const float v = (float)acc[j];
          const float tq = fmaf(cA, v, cB) * y_inv;
          const float gy = (valid && tq > t_lo && tq <= t_hi) ? gq[j] : 0.0f;
          const float dcf = fmaf(gy, cK1, fmaf(v, cE, cF));
          const uint32_t db = __float_as_uint(dcf), dr = sr_next16(rng);
          const uint32_t hb = (db + (sr_on ? dr : 0x7fffu + ((db >> 16) & 1u))) >> 16;
          dcw[NDR - 1][j] = valid ? __uint_as_float(hb << 16) : 0.0f;
        }
        if (m >= m0 && m < m1) {          // weight gradient: own rows, own columns j = LH .. LH + 7; x pixel of (j, kx) = 2 j + kx
#pragma unroll
          for (int j = LH; j < LH + 8; ++j) sdc += dcw[NDR - 1][j];
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            float xf[NXD * 4];
#pragma unroll
            for (int t = 2 * LH; t < 2 * LH + 14 + K; ++t) xf[t] = (float)(((xd[ky][t >> 2] ^ 0x80808080u) >> (8 * (t & 3))) & 255u);
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
              for (int j = LH; j < LH + 8; ++j) wacc[ky * K + kx] = fmaf(dcw[NDR - 1][j], xf[2 * j + kx], wacc[ky * K + kx]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NDC; ++j) dcw[NDR - 1][j] = 0.0f;
      }
      // ---- the two dx rows that are final now: iy = 2 m + IO, IO = -1, 0 (k3) / -2, -1 (k5)
      int nst = 0;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        constexpr int IO0 = (K == 3) ? -1 : -2;
        const int io = IO0 + rr;                                   // compile time after unrolling
        const int iy = 2 * m + io;
        if (iy >= 2 * m0 && iy < 2 * m1 && iy < p.h) {
          float a[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) a[i] = 0.0f;
#pragma unroll
          for (int ky = K - 1; ky >= 0; --ky) {
            const int ty = io + PAD - ky;                          // dc row = m + ty / 2 when ty is even
            if (((ty % 2) + 2) % 2 == 0) {
              const int dr = NDR - 1 + ty / 2;                     // ty <= 0 and even: exact
              if (dr >= 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                  for (int kx = 0; kx < K; ++kx) {
                    const int tx = i + PAD - kx;
                    if (((tx % 2) + 2) % 2 == 0) a[i] = fmaf(dcw[dr < 0 ? 0 : dr][tx / 2 + LH], wf[ky * K + kx], a[i]);
                  }
              }
            }
          }
          if (C1) {
            // two pixels at a time, the next group's LDS reads in flight under this group's arithmetic; the volatile asm ties the running sums to program order (left alone, the
            // compiler reads all sixteen pixels first and keeps 64 - 96 registers of x0 bytes alive: 229 - 256 VGPRs, spills)
            const uint8_t* x0r = c1ring + ((s % (PD + 1)) * 2 + rr) * G::C1R + 16 * hf * CIN1;
            constexpr int GS = 2, NG = 16 / GS;
            uint2 xb[2][GS][C1D / 2];
#pragma unroll
            for (int i = 0; i < GS; ++i)
#pragma unroll
              for (int q = 0; q < C1D / 2; ++q) xb[0][i][q] = *(const uint2*)(x0r + i * CIN1 + 8 * q);
#pragma unroll
            for (int g4 = 0; g4 < NG; ++g4) {
              if (g4 + 1 < NG) {
#pragma unroll
                for (int i = 0; i < GS; ++i)
#pragma unroll
                  for (int q = 0; q < C1D / 2; ++q) xb[(g4 + 1) & 1][i][q] = *(const uint2*)(x0r + (GS * (g4 + 1) + i) * CIN1 + 8 * q);
              }
#pragma unroll
              for (int i = 0; i < GS; ++i) {
                int acc1 = acc01;
#pragma unroll
                for (int q = 0; q < C1D / 2; ++q) {
                  acc1 = __builtin_amdgcn_sdot4((int)xb[g4 & 1][i][q].x, w1[2 * q], acc1, false); acc1 = __builtin_amdgcn_sdot4((int)xb[g4 & 1][i][q].y, w1[2 * q + 1], acc1, false);
                }
                const float v = (float)acc1;
                const float tq = fmaf(a1A, v, a1B) * y_inv1;
                const float gg = (((pmask >> (GS * g4 + i)) & 1u) && tq > t_lo1 && tq <= t_hi1) ? a[GS * g4 + i] * sw : 0.0f;
                c1s1 += gg; c1s2 = fmaf(gg, v, c1s2);          // sum g and sum g * acc: xhat = acc * R - M * R is applied once, at the end
              }
              asm volatile("" : "+v"(c1s1), "+v"(c1s2) :: "memory");
            }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) *(uint16_t*)(orow + ((16 * hf + i) * CBW + lc) * 2) = (uint16_t)cvt_pk_bf16(a[i] * sw, 0.0f);
          const uint4 v0 = *(const uint4*)(orow + lane * 16), v1 = *(const uint4*)(orow + 1024 + lane * 16);
          uint16_t* drow_p = dimg + (int64_t)iy * xpitch;
          if (ook0) *(uint4*)(drow_p + ooff0) = v0;
          if (ook1) *(uint4*)(drow_p + ooff1) = v1;
          nst += nst_row;
        }
      }
#pragma unroll
      for (int i = 0; i + 1 < PD; ++i) sth[i] = sth[i + 1];
      sth[PD - 1] = nst;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  __syncthreads();
  float* red = (float*)smem;                       // [4][k*k][64]
  const float zpf = (float)zpx;
#pragma unroll
  for (int t = 0; t < KK; ++t) red[(wv * KK + t) * 64 + lane] = wacc[t] - zpf * sdc;
  __syncthreads();
  const float sx = p.qx[FROST_Q_SCALE];
  for (int i = tid; i < KK * CBW; i += 256) {
    const int t = i / CBW, l2 = i % CBW; const int c2 = cb * CBW + l2;
    float sum = 0.0f;
    for (int w2 = 0; w2 < 4; ++w2)
      for (int l3 = l2; l3 < 64; l3 += CBW) sum += red[(w2 * KK + t) * 64 + l3];
    if (c2 < p.c) atomicAdd(dwq_dst(p.dwq, (int64_t)p.c * KK) + (int64_t)c2 * KK + t, sum * sx);
  }
  if (C1) {           // conv1's S1 / S2: the waves' lane-local sums, one pair of float atomics per channel and workgroup (k_pw's reduce-pass tail)
    __syncthreads();
    red[(wv * 64 + lane) * 2] = c1s1; red[(wv * 64 + lane) * 2 + 1] = fmaf(c1s2, a1R, a1MR * c1s1);
    __syncthreads();
    if (tid < CBW && (cb * CBW + tid) < p.c) {
      float a = 0, b = 0;
      for (int w2 = 0; w2 < 4; ++w2)
        for (int l2 = tid; l2 < 64; l2 += CBW) { a += red[(w2 * 64 + l2) * 2]; b += red[(w2 * 64 + l2) * 2 + 1]; }
      atomicAdd(s12_dst(p.c1_coef, p.cpad, 0) + cb * CBW + tid, a); atomicAdd(s12_dst(p.c1_coef, p.cpad, 1) + cb * CBW + tid, b);
    }
  }
}

// ================================================================================================ the single-sweep passes on the same skeleton
// Statistics pass, emit pass and backward reduce pass of the tiled depthwise layers (k_dw3 modes 0 / 1 / 2, frost_dw3.hip: same integer conv, same epilogue expressions)
// as strip-streaming waves: a lane owns one channel and 8 OUTPUT columns, a step is one output row (S new x rows by direct-to-LDS copy, the other k - S carried in
// registers; the backward reduce pass also takes the gy row).  No halo columns are recomputed here (nothing consumes a neighbour's result), no workgroup barrier, no tile
// halo re-read in y.  Statistics are exact integers (sum, sum of squares in fp64 < 2^53, min, max): bit-identical to k_dw3 whatever the summation order; the emit pass
// is a pure function of the accumulator; S1 / S2 of the reduce pass are fp32 sums grouped per lane here and per tile there (1e-7).
enum { W_STATS = 0, W_EMIT = 1, W_BRED = 2 };
struct DwsP {
  const int8_t* x; const float* qx; const int8_t* wq; const int32_t* wsum;
  uint8_t* stats; float* coef; const float* qy; int8_t* y; const uint16_t* gout;
  int n, h, w, c, cpad, relu, ho, wo; float inv_count;
  int ncb, nstrips, nchunks, rc;
  FrostFinDesc fin; int fin_on;
};
template <int K, int S, int CBW, int PD, int MODE>
struct DwsGeo {
  static constexpr int PAD = (K - 1) / 2, HALF = 64 / CBW, SW = 8 * HALF;
  static constexpr int NXW = 7 * S + K, NTR = (NXW + 7) / 8, NXD = 2 * NTR;                   // x window of a lane (pixels), transposed reads, dwords
  static constexpr int XPX = S * 8 * (HALF - 1) + NXW;
  static constexpr int XR = ((XPX * CBW + 15) / 16) * 16, GR = SW * CBW * 2, OR = SW * CBW;   // x row record, gy row record (1 KB), int8 out row (512 B)
  static constexpr int XU = XR / 16, NXI = (XU + 63) / 64;                                    // 16-byte units / copy instructions per x row
  static constexpr int NXS = S * PD + ((K - S > S) ? K - S : S), NGS = PD + 1;
  static constexpr int XSL = 8 * CBW;
  static constexpr int X_OFF = 0, G_OFF = NXS * XR + XSL, O_OFF = G_OFF + ((MODE == W_BRED) ? NGS * GR : 0), WAVE_RAW = O_OFF + ((MODE == W_EMIT) ? OR : 0);
  static constexpr int WAVE_LDS = (WAVE_RAW < 2048) ? 2048 : WAVE_RAW;                        // (the final fold of the statistics needs 6 KB + 16 B per workgroup)
  static constexpr int NDMA = S * NXI + ((MODE == W_BRED) ? 1 : 0), NST = (MODE == W_EMIT) ? 1 : 0;
  static_assert(NXI <= 2 && GR == 1024, "one or two copy instructions per x row, one per gy row");
};

template <int K, int S, int CBW, int PD, int MODE>
__global__ __launch_bounds__(256, 3) void k_dws(const DwsP p) {
  using G = DwsGeo<K, S, CBW, PD, MODE>;
  constexpr int PAD = G::PAD, NXD = G::NXD, NTR = G::NTR, SW = G::SW, XR = G::XR, GR = G::GR, NXS = G::NXS, NGS = G::NGS, KK = K * K, NPK = (K == 3) ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* const wl = smem + (size_t)wv * G::WAVE_LDS;
  uint8_t* const xring = wl + G::X_OFF; uint8_t* const gring = wl + G::G_OFF; uint8_t* const orow = wl + G::O_OFF;
  const uint32_t xring_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)xring);
  const uint32_t gring_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)gring);
  const int lc = lane & (CBW - 1), hf = lane / CBW;
  const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3, nl = (int)gridDim.x >> 3;
  const int cb = li % p.ncb, lwg = li / p.ncb, nlc = nl / p.ncb;
  const int img_lo = (int)(((int64_t)p.n * xcd) >> 3), img_hi = (int)(((int64_t)p.n * (xcd + 1)) >> 3);
  const int pit = p.nstrips * p.nchunks, ntask = (img_hi - img_lo) * pit;
  const int ch = cb * CBW + lc; const bool chok = ch < p.c;

  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]);
  const uint32_t zp4 = (uint32_t)((zpx - 128) & 255) * 0x01010101u;
  int wpk[K][NPK];
  {
    int8_t taps[KK];
    load_taps_i8<KK>(p.wq, p.cpad, ch, chok, taps);
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) { const uint32_t b = (uint32_t)(uint8_t)taps[ky * K + kx]; if (kx < 4) lo |= b << (8 * kx); else hi |= b; }
      wpk[ky][0] = (int)lo; if (NPK > 1) wpk[ky][NPK - 1] = (int)hi;
    }
  }
  const int chc = chok ? ch : 0;
  const int acc0 = chok ? (128 - zpx) * p.wsum[chc] : 0;
  float cA = 0, cB = 0, cR = 0, cMR = 0, y_inv = 1.0f, y_zpf = 0.0f, t_lo = 0.0f, t_hi = 0.0f, qcap = 255.0f; bool lowq = false;
  if (MODE != W_STATS) {
    y_inv = 1.0f / p.qy[FROST_Q_SCALE]; const int zpy = __float_as_int(p.qy[FROST_Q_ZP]); y_zpf = (float)zpy;
    if (chok) {
      cA = p.coef[FROST_COEF_A * p.cpad + ch]; cB = p.coef[FROST_COEF_B * p.cpad + ch];
      if (MODE == W_BRED) { const float m = p.coef[FROST_COEF_M * p.cpad + ch]; cR = p.coef[FROST_COEF_R * p.cpad + ch]; cMR = -m * cR; }
    }
    if (MODE == W_EMIT) { qcap = (float)q_hi(p.qy); lowq = qcap < 255.0f; }
    if (MODE == W_BRED) {
      const int qhi = q_hi(p.qy);
      const float hi0 = (float)qhi + 0.5f - (float)zpy;
      t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
      if (!p.relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
    }
  }
  const float relu_floor = p.relu ? 0.0f : -INFINITY;
  double st1 = 0.0, st2 = 0.0; float smn = INFINITY, smx = -INFINITY, r1 = 0.0f, r2 = 0.0f;

  const int64_t xpitch = (int64_t)p.w * p.c, opitch = (int64_t)p.wo * p.c;
  for (int tt = lwg * 4 + wv; tt < ntask; tt += nlc * 4) {
    const int img = img_lo + tt / pit; const int rem = tt - (tt / pit) * pit;
    const int chunk = rem / p.nstrips, strip = rem - chunk * p.nstrips;
    const int m0 = chunk * p.rc, m1 = min(m0 + p.rc, p.ho);
    const int oc0s = strip * SW, NS = m1 - m0, rbase = S * m0 - PAD;
    const int8_t* const ximg = p.x + (int64_t)img * p.h * xpitch;
    const uint16_t* const gimg = (MODE == W_BRED) ? p.gout + (int64_t)img * p.ho * opitch : nullptr;
    int8_t* const yimg = (MODE == W_EMIT) ? p.y + (int64_t)img * p.ho * opitch : nullptr;
    int xoff0, xoff1 = 0, goff = 0, ooff = 0; bool ook = false;
    {
      constexpr int UPX = CBW / 16, UPG = CBW / 8;
      auto xo = [&](int u) { const int px = u / UPX, cu = u % UPX, col = min(max(S * oc0s - PAD + px, 0), p.w - 1); int cn = cb * CBW + cu * 16; if (cn >= p.c) cn = 0; return col * p.c + cn; };
      xoff0 = xo(lane); if (G::NXI > 1) xoff1 = xo(64 + lane);
      const int px = lane / UPG, cu = lane % UPG, col = oc0s + px, cn = cb * CBW + cu * 8;
      ook = col < p.wo && cn < p.c; ooff = col * p.c + cn;                      // 8-channel unit of the out row (emit) ...
      goff = min(col, p.wo - 1) * p.c + (cn < p.c ? cn : 0);                   // ... and of the gy row (reduce pass), clamped
    }
    const int oc0 = oc0s + 8 * hf;
    uint32_t xm[NXD]; uint32_t cmask = 0;
#pragma unroll
    for (int i = 0; i < NXD; ++i) {
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) { const int col = S * oc0 - PAD + 4 * i + b; if (col >= 0 && col < p.w) m |= 0xffu << (8 * b); }
      xm[i] = m;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (oc0 + j < p.wo && chok) cmask |= 1u << j;

    auto issue_x = [&](int row) __attribute__((always_inline)) {
      const int slot = (row - rbase) % NXS;
      const int8_t* base = ximg + (int64_t)min(max(row, 0), p.h - 1) * xpitch;
      if (G::XU >= 64 || lane < G::XU) dwb_glds16(base + xoff0, xring_a + (uint32_t)(slot * XR));
      if (G::NXI > 1) { if (lane < G::XU - 64) dwb_glds16(base + xoff1, xring_a + (uint32_t)(slot * XR + 1024)); }
    };
    auto issue_g = [&](int row, int slot) __attribute__((always_inline)) {
      if (MODE == W_BRED) dwb_glds16(gimg + (int64_t)min(max(row, 0), p.ho - 1) * opitch + goff, gring_a + (uint32_t)(slot * GR));
    };
    auto load_x = [&](int row, uint32_t* d) __attribute__((always_inline)) {
      const uint8_t* rp = xring + ((row - rbase) % NXS) * XR;
      const bool rok = row >= 0 && row < p.h;
#pragma unroll
      for (int t = 0; t < NTR; ++t) {
        const v2i_b a = dwb_tr8<CBW>(rp, 8 * S * hf + 8 * t, lane);
        d[2 * t] = rok ? (((uint32_t)a[0] & xm[2 * t]) | (zp4 & ~xm[2 * t])) : zp4;
        d[2 * t + 1] = rok ? (((uint32_t)a[1] & xm[2 * t + 1]) | (zp4 & ~xm[2 * t + 1])) : zp4;
      }
    };
#pragma unroll
    for (int i = 0; i < K - S; ++i) issue_x(rbase + i);
#pragma unroll
    for (int b = 0; b < PD; ++b) {
#pragma unroll
      for (int i = 0; i < S; ++i) issue_x(rbase + (K - S) + S * b + i);
      issue_g(m0 + b, b % NGS);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t xd[K][NXD];
#pragma unroll
    for (int i = 0; i < K - S; ++i) load_x(rbase + i, xd[i + S]);

#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      const int m = m0 + s;
      if (s > 0) dwb_wait_vm((PD - 1) * G::NDMA + min(s, PD) * G::NST);
#pragma unroll
      for (int i = 0; i < S; ++i) issue_x(rbase + (K - S) + S * (s + PD) + i);
      issue_g(m + PD, (s + PD) % NGS);
#pragma unroll
      for (int ky = 0; ky < K - S; ++ky)
#pragma unroll
        for (int i = 0; i < NXD; ++i) xd[ky][i] = xd[ky + S][i];
#pragma unroll
      for (int i = 0; i < S; ++i) load_x(S * m - PAD + (K - S) + i, xd[K - S + i]);
      int acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = acc0;
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int q = 0; q < NPK; ++q) {
            const int o = S * j + 4 * q;
            const uint32_t lo = xd[ky][o / 4], hi = (o / 4 + 1 < NXD) ? xd[ky][(o / 4 + 1) % NXD] : 0u;
            const int win = (o % 4 == 0) ? (int)lo : (int)__builtin_amdgcn_alignbyte(hi, lo, o % 4);
            acc[j] = __builtin_amdgcn_sdot4(win, wpk[ky][q], acc[j], false);
          }
      if (MODE == W_STATS) {
        int t1 = 0, tmn = INT32_MAX, tmx = INT32_MIN; double t2 = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool valid = (cmask >> j) & 1u;
          const int vi = acc[j]; const double v = (double)vi;
          t1 += valid ? vi : 0; t2 = valid ? fma(v, v, t2) : t2; tmn = valid ? min(tmn, vi) : tmn; tmx = valid ? max(tmx, vi) : tmx;
        }
        st1 += (double)t1; st2 += t2; smn = fminf(smn, (float)tmn); smx = fmaxf(smx, (float)tmx);
      } else if (MODE == W_EMIT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yv = fmaf(cA, (float)acc[j], cB);
          float qv = rintf(fmaxf(yv, relu_floor) * y_inv) + y_zpf;
          if (lowq) qv = fminf(qv, qcap);
          orow[(8 * hf + j) * CBW + lc] = (uint8_t)((__builtin_amdgcn_cvt_pk_u8_f32(qv, 0, 0u) ^ 0x80u) & 255u);
        }
        const uint2 v = *(const uint2*)(orow + lane * 8);
        if (ook) *(uint2*)(yimg + (int64_t)m * opitch + ooff) = v;
      } else {
        float gq[8];
        const uint8_t* gp = gring + (s % NGS) * GR;
        dwb_tr16<CBW>(gp, 8 * hf, lane, gq); dwb_tr16<CBW>(gp, 8 * hf + 4, lane, gq + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool valid = (cmask >> j) & 1u;
          const float v = (float)acc[j];
          const float tq = fmaf(cA, v, cB) * y_inv;
          const float gy = (valid && tq > t_lo && tq <= t_hi) ? gq[j] : 0.0f;
          r1 += gy; r2 = fmaf(gy, fmaf(v, cR, cMR), r2);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  if (MODE == W_STATS || MODE == W_BRED) {      // the waves' lane-local partials, one global atomic set per channel and workgroup (k_dw3's tail)
    __syncthreads();
    double* red_d = (double*)smem; float* red_f = (float*)(red_d + 4 * 64 * 2);
    if (MODE == W_STATS) { red_d[(wv * 64 + lane) * 2] = st1; red_d[(wv * 64 + lane) * 2 + 1] = st2; red_f[(wv * 64 + lane) * 2] = smn; red_f[(wv * 64 + lane) * 2 + 1] = smx; }
    else { red_f[(wv * 64 + lane) * 2] = r1; red_f[(wv * 64 + lane) * 2 + 1] = r2; }
    __syncthreads();
    if (tid < CBW && (cb * CBW + tid) < p.c) {
      const int ch2 = cb * CBW + tid;
      if (MODE == W_STATS) {
        double a = 0, b = 0; float c = INFINITY, d = -INFINITY;
        for (int w2 = 0; w2 < 4; ++w2)
          for (int l2 = tid; l2 < 64; l2 += CBW) { a += red_d[(w2 * 64 + l2) * 2]; b += red_d[(w2 * 64 + l2) * 2 + 1]; c = fminf(c, red_f[(w2 * 64 + l2) * 2]); d = fmaxf(d, red_f[(w2 * 64 + l2) * 2 + 1]); }
        long long* g_s1 = (long long*)stats_copy(p.stats, p.cpad); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
        int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
        if (c <= d) {
          atomicAdd((unsigned long long*)&g_s1[ch2], (unsigned long long)(long long)a); atomicAdd(&g_s2[ch2], (unsigned long long)b);
          atomicMin(&g_mn[ch2], (int)c); atomicMax(&g_mx[ch2], (int)d);
        }
      } else {
        float a = 0, b = 0;
        for (int w2 = 0; w2 < 4; ++w2)
          for (int l2 = tid; l2 < 64; l2 += CBW) { a += red_f[(w2 * 64 + l2) * 2]; b += red_f[(w2 * 64 + l2) * 2 + 1]; }
        atomicAdd(s12_dst(p.coef, p.cpad, 0) + ch2, a); atomicAdd(s12_dst(p.coef, p.cpad, 1) + ch2, b);
      }
    }
    if (MODE == W_STATS && p.fin_on) {       // last workgroup done -> conv finalize in this launch (see frost_common.h)
      int* sflag = (int*)smem;
      if (last_block_done2(p.fin.counter, gridDim.x, sflag)) {
        float* sh = (float*)(smem + 16);
        conv_finalize_dev(p.stats, (int64_t)p.n * p.ho * p.wo, p.c, p.cpad, p.qx, p.fin.qrec_w, p.fin.wscale, p.fin.gamma, p.fin.beta, p.fin.rmean, p.fin.rvar,
                          p.fin.nbt, p.fin.training, p.fin.relu, p.fin.observe, 1, p.fin.coef, p.fin.qrec_y, tid, 256, sh);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int dwb_env(const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; }
// grid and row chunks: one round of resident workgroups (a multiple of 8 XCDs x channel blocks); enough wave tasks per (XCD, channel block) for >= 4 per resident wave,
// chunks no shorter than `min_rows` rows.  rows / strips are counted in the domain the waves stream over (stride 1: input = output rows; stride 2: output rows)
template <typename PT>
static unsigned dwb_plan(PT& p, int occ, int rows, int strips, int cbw, int min_rows) {
  static const int occ_env = dwb_env("FROST_DWB_OCC"), chunks_env = dwb_env("FROST_DWB_CHUNKS");
  const int o = occ_env > 0 ? occ_env : occ;
  p.ncb = (p.c + cbw - 1) / cbw; p.nstrips = strips;
  int per = (256 * o) / (8 * p.ncb); if (per < 1) per = 1;
  int nch = 1;
  if (chunks_env > 0) nch = chunks_env;
  else while (nch < 8 && (int64_t)(p.n / 8 > 0 ? p.n / 8 : 1) * strips * nch < (int64_t)16 * per && (rows + 2 * nch - 1) / (2 * nch) >= min_rows) nch *= 2;
  p.rc = (rows + nch - 1) / nch; p.nchunks = (rows + p.rc - 1) / p.rc;
  const int64_t tasks_x = (int64_t)((p.n + 7) / 8) * strips * p.nchunks;       // wave tasks per (XCD, channel block)
  if ((int64_t)per * 4 > tasks_x) per = (int)((tasks_x + 3) / 4);
  return (unsigned)(8 * p.ncb * per);
}
template <int CBW, int PD>
static int launch_dwb(DwbP& p, hipStream_t s) {
  using G = DwbGeo<CBW, PD>;
  const size_t lds = (size_t)4 * G::WAVE_LDS;
  static int occ = 0;
  if (!occ) {
    (void)hipFuncSetAttribute((const void*)k_dwb_s1<CBW, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_dwb_s1<CBW, PD>, 256, lds) != hipSuccess || occ < 1) occ = 1;
    if (occ > 8) occ = 8;
  }
  const unsigned grid = dwb_plan(p, occ, p.h, (p.w + G::SW - 1) / G::SW, CBW, 14);
  hipLaunchKernelGGL((k_dwb_s1<CBW, PD>), dim3(grid), dim3(256), lds, s, p);
  return frost_check_launch("dw_bwd_fused");
}
template <int K, int CBW, int PD, int C1D = 0>
static int launch_dwb2(DwbP& p, hipStream_t s) {
  using G = DwbGeo2<K, CBW, PD, C1D>;
  const size_t lds = (size_t)4 * G::WAVE_LDS;
  static int occ = 0;
  if (!occ) {
    (void)hipFuncSetAttribute((const void*)k_dwb_s2<K, CBW, PD, C1D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_dwb_s2<K, CBW, PD, C1D>, 256, lds) != hipSuccess || occ < 1) occ = 1;
    if (occ > 8) occ = 8;
  }
  const unsigned grid = dwb_plan(p, occ, p.ho, (p.wo + G::SWO - 1) / G::SWO, CBW, 7);
  hipLaunchKernelGGL((k_dwb_s2<K, CBW, PD, C1D>), dim3(grid), dim3(256), lds, s, p);
  return frost_check_launch("dw_bwd_fused");
}

extern "C" int frost_dw_bwd_fused_ok(int h, int w, int c, int k, int stride) {
  static const int s2_on = getenv("FROST_DWB_S2") ? atoi(getenv("FROST_DWB_S2")) : 1, minw = getenv("FROST_DWB_MINW") ? atoi(getenv("FROST_DWB_MINW")) : 28;     // measured: the 14 -> 7 layer is no faster here (9 steps per task: the prologue dominates)
  if ((c % 8) != 0 || w < minw || h < 8) return 0;
  if (k == 3 && stride == 1) return 1;
  if (stride == 2 && (k == 3 || k == 5) && (h % 2) == 0 && (w % 2) == 0) return s2_on ? 1 : 0;
  return 0;
}

extern "C" int frost_dw_bwd_fused_c1_ok(int h, int w, int c, int k, int stride, int c1_cin);
static int dwb_fused_impl(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                          int n, int h, int w, int c, int k, int stride, const float* coef, const float* qrec_y, int relu, const uint16_t* gout,
                          uint16_t* dx, float* dwq, const int8_t* c1_x, const float* c1_qrec_x, const int8_t* c1_wq_pack, const int32_t* c1_wsum, float* c1_coef,
                          int c1_cin, int c1_relu, void* stream) {
  FROST_REQUIRE(frost_dw_bwd_fused_ok(h, w, c, k, stride), "dw_bwd_fused: unsupported shape (k = 3 stride 1, or k in {3, 5} stride 2 on even maps; channels a multiple of 8)");
  FROST_REQUIRE(x && wq_pack && wsum && coef && qrec_y && gout && dx && dwq, "dw_bwd_fused: incomplete arguments");
  DwbP p = {};
  p.x = x; p.qx = qrec_x; p.wq = wq_pack; p.wsum = wsum; p.qw = qrec_w; p.wscale = wscale; p.coef = coef; p.qy = qrec_y; p.gout = gout; p.dx = dx; p.dwq = dwq;
  p.n = n; p.h = h; p.w = w; p.c = c; p.cpad = round_up(c, 16); p.relu = relu; p.sr = frost_sr_enabled();
  const int pad = (k - 1) / 2; p.ho = (h + 2 * pad - k) / stride + 1; p.wo = (w + 2 * pad - k) / stride + 1;
  p.inv_count = 1.0f / (float)((int64_t)n * p.ho * p.wo);
  hipStream_t s = as_stream(stream);
  // 32-channel blocks when they waste fewer lanes than 64-channel blocks (32, 72, 96, 144 channels: the high-resolution layers), as pick_geo of frost_dw3.hip
  static const int cbw_env = dwb_env("FROST_DWB_CBW");
  const bool c32 = cbw_env ? (cbw_env == 32) : (round_up(c, 32) < round_up(c, 64));
  if (c1_x) {
    FROST_REQUIRE(frost_dw_bwd_fused_c1_ok(h, w, c, k, stride, c1_cin) && c32, "dw_bwd_fused_c1: the conv1 fold has no instance for this shape");
    FROST_REQUIRE(c1_qrec_x && c1_wq_pack && c1_wsum && c1_coef, "dw_bwd_fused_c1: incomplete conv1 arguments");
    p.c1_x = c1_x; p.c1_qx = c1_qrec_x; p.c1_wq = c1_wq_pack; p.c1_wsum = c1_wsum; p.c1_coef = c1_coef; p.c1_relu = c1_relu;
    if (k == 3) return (c1_cin == 16) ? launch_dwb2<3, 32, DWB_PD, 4>(p, s) : launch_dwb2<3, 32, DWB_PD, 6>(p, s);
    return (c1_cin == 16) ? launch_dwb2<5, 32, DWB_PD, 4>(p, s) : launch_dwb2<5, 32, DWB_PD, 6>(p, s);
  }
  if (stride == 1) return c32 ? launch_dwb<32, DWB_PD>(p, s) : launch_dwb<64, DWB_PD>(p, s);
  if (k == 3) return c32 ? launch_dwb2<3, 32, DWB_PD>(p, s) : launch_dwb2<3, 64, DWB_PD>(p, s);
  return c32 ? launch_dwb2<5, 32, DWB_PD>(p, s) : launch_dwb2<5, 64, DWB_PD>(p, s);
}
extern "C" int frost_dw_bwd_fused(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                                  int n, int h, int w, int c, int k, int stride, const float* coef, const float* qrec_y, int relu, const uint16_t* gout,
                                  uint16_t* dx, float* dwq, void* stream) {
  return dwb_fused_impl(x, qrec_x, wq_pack, wsum, qrec_w, wscale, n, h, w, c, k, stride, coef, qrec_y, relu, gout, dx, dwq, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
}
// the same sweep carrying the REDUCE PASS of the pointwise layer that produced x (conv1 of the bottleneck: frost_pw_conv_bwd pass 0 on (c1_x -> x)): its S1 / S2 accumulate into
// c1_coef from the dx values as they are formed (before their bf16 rounding), so that layer's backward starts at its dc pass and never re-reads dx for the sums
extern "C" int frost_dw_bwd_fused_c1_ok(int h, int w, int c, int k, int stride, int c1_cin) {
  static const int on = getenv("FROST_DWB_C1") ? atoi(getenv("FROST_DWB_C1")) : 3;          // bit 0: k = 3 layers, bit 1: k = 5 layers
  return (((on >> (k == 3 ? 0 : 1)) & 1) && stride == 2 && (c1_cin == 16 || c1_cin == 24) && round_up(c, 32) < round_up(c, 64) && frost_dw_bwd_fused_ok(h, w, c, k, stride)) ? 1 : 0;
}
extern "C" int frost_dw_bwd_fused_c1(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                                     int n, int h, int w, int c, int k, int stride, const float* coef, const float* qrec_y, int relu, const uint16_t* gout,
                                     uint16_t* dx, float* dwq, const int8_t* c1_x, const float* c1_qrec_x, const int8_t* c1_wq_pack, const int32_t* c1_wsum, float* c1_coef,
                                     int c1_cin, int c1_relu, void* stream) {
  FROST_REQUIRE(c1_x, "dw_bwd_fused_c1: conv1's input is required");
  return dwb_fused_impl(x, qrec_x, wq_pack, wsum, qrec_w, wscale, n, h, w, c, k, stride, coef, qrec_y, relu, gout, dx, dwq, c1_x, c1_qrec_x, c1_wq_pack, c1_wsum, c1_coef, c1_cin,
                        c1_relu, stream);
}

// ---- the single-sweep passes: called by the frost_dw_conv_fwd / _fwd_fin / _bwd entries of frost_dw3.hip when the shape qualifies (FROST_DW_STREAM: bit 0 statistics, bit 1 emit,
// bit 2 backward reduce pass; default 0 = off: bit-identical results, but measured no faster than k_dw3's tile kernels inside the step, profiles/r05_dw_onesweep_ab.txt).  mode: 0 statistics (fin != NULL: with the folded finalize), 1 emit, 2 reduce pass
int frost_dws_ok(int h, int w, int c, int k, int stride, int mode) {
  static const int mask = getenv("FROST_DW_STREAM") ? atoi(getenv("FROST_DW_STREAM")) : 0, minw = getenv("FROST_DWS_MINW") ? atoi(getenv("FROST_DWS_MINW")) : 28;
  if (!((mask >> mode) & 1) || (c % 8) != 0 || w < minw || h < 8) return 0;
  if (k == 3 && stride == 1) return 1;
  if (stride == 2 && (k == 3 || k == 5) && (h % 2) == 0 && (w % 2) == 0) return 1;
  return 0;
}
template <int K, int S, int CBW, int MODE>
static int launch_dws(DwsP& p, hipStream_t s) {
  constexpr int PD = 2;
  using G = DwsGeo<K, S, CBW, PD, MODE>;
  const size_t lds = (size_t)4 * G::WAVE_LDS;
  static int occ = 0;
  if (!occ) {
    (void)hipFuncSetAttribute((const void*)k_dws<K, S, CBW, PD, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_dws<K, S, CBW, PD, MODE>, 256, lds) != hipSuccess || occ < 1) occ = 1;
    if (occ > 8) occ = 8;
  }
  const unsigned grid = dwb_plan(p, occ, p.ho, (p.wo + G::SW - 1) / G::SW, CBW, 7);
  hipLaunchKernelGGL((k_dws<K, S, CBW, PD, MODE>), dim3(grid), dim3(256), lds, s, p);
  return frost_check_launch("dw_stream");
}
template <int MODE>
static int dispatch_dws(DwsP& p, int k, int stride, hipStream_t s) {
  static const int cbw_env = dwb_env("FROST_DWB_CBW");
  const bool c32 = cbw_env ? (cbw_env == 32) : (round_up(p.c, 32) < round_up(p.c, 64));
  if (stride == 1) return c32 ? launch_dws<3, 1, 32, MODE>(p, s) : launch_dws<3, 1, 64, MODE>(p, s);
  if (k == 3) return c32 ? launch_dws<3, 2, 32, MODE>(p, s) : launch_dws<3, 2, 64, MODE>(p, s);
  return c32 ? launch_dws<5, 2, 32, MODE>(p, s) : launch_dws<5, 2, 64, MODE>(p, s);
}
int frost_dws_launch(int mode, const int8_t* x, const float* qx, const int8_t* wq, const int32_t* wsum, int n, int h, int w, int c, int k, int stride,
                     void* stats, float* coef, const float* qy, int relu, int8_t* y, const uint16_t* gout, const FrostFinDesc* fin, hipStream_t s) {
  DwsP p = {};
  p.x = x; p.qx = qx; p.wq = wq; p.wsum = wsum; p.stats = (uint8_t*)stats; p.coef = coef; p.qy = qy; p.relu = relu; p.y = y; p.gout = gout;
  p.n = n; p.h = h; p.w = w; p.c = c; p.cpad = round_up(c, 16);
  const int pad = (k - 1) / 2; p.ho = (h + 2 * pad - k) / stride + 1; p.wo = (w + 2 * pad - k) / stride + 1;
  p.inv_count = 1.0f / (float)((int64_t)n * p.ho * p.wo);
  if (fin) { p.fin = *fin; p.fin_on = 1; p.coef = fin->coef; p.qy = fin->qrec_y; p.relu = fin->relu; }
  if (mode == 0) return dispatch_dws<W_STATS>(p, k, stride, s);
  if (mode == 1) return dispatch_dws<W_EMIT>(p, k, stride, s);
  return dispatch_dws<W_BRED>(p, k, stride, s);
}
