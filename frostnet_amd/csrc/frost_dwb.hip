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

typedef int v2i_b __attribute__((ext_vector_type(2)));
typedef short v4s_b __attribute__((ext_vector_type(4)));

struct DwbP {
  const int8_t* x; const float* qx; const int8_t* wq; const int32_t* wsum; const float* qw; const float* wscale;
  const float* coef; const float* qy; const uint16_t* gout; uint16_t* dx; float* dwq;
  int n, h, w, c, cpad, relu, sr; float inv_count;
  int ncb, nstrips, nchunks, rc;          // channel blocks, column strips per row, row chunks per image, rows per chunk
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
__global__ __launch_bounds__(256, 3) void k_dwb_s1(const DwbP p) {
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
    cE = -cK1 * (p.coef[FROST_COEF_S2 * p.cpad + ch] * p.inv_count) * cR;
    cF = -cK1 * (p.coef[FROST_COEF_S1 * p.cpad + ch] * p.inv_count) - cE * m;
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
    if (c2 < p.c) atomicAdd(p.dwq + (int64_t)c2 * 9 + t, sum * sx);
  }
}

// ------------------------------------------------------------------------------------------------ host side
template <int CBW, int PD>
static int launch_dwb(DwbP& p, hipStream_t s) {
  using G = DwbGeo<CBW, PD>;
  const size_t lds = (size_t)4 * G::WAVE_LDS;
  static int occ = 0;
  if (!occ) {
    hipFuncSetAttribute((const void*)k_dwb_s1<CBW, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_dwb_s1<CBW, PD>, 256, lds) != hipSuccess || occ < 1) occ = 1;
    if (occ > 8) occ = 8;
  }
  p.ncb = (p.c + CBW - 1) / CBW; p.nstrips = (p.w + G::SW - 1) / G::SW;
  static const int occ_env = getenv("FROST_DWB_OCC") ? atoi(getenv("FROST_DWB_OCC")) : 0;
  static const int chunks_env = getenv("FROST_DWB_CHUNKS") ? atoi(getenv("FROST_DWB_CHUNKS")) : 0;
  const int o = occ_env > 0 ? occ_env : occ;
  // workgroups: one round of resident ones, a multiple of 8 XCDs x channel blocks
  int per = (256 * o) / (8 * p.ncb); if (per < 1) per = 1;
  // row chunks: enough wave tasks per (XCD, channel block) for >= 4 per resident wave, chunks no shorter than 14 rows
  int nch = 1;
  if (chunks_env > 0) nch = chunks_env;
  else while (nch < 8 && (int64_t)(p.n / 8 > 0 ? p.n / 8 : 1) * p.nstrips * nch < (int64_t)16 * per && (p.h + 2 * nch - 1) / (2 * nch) >= 14) nch *= 2;
  p.rc = (p.h + nch - 1) / nch; p.nchunks = (p.h + p.rc - 1) / p.rc;
  const int64_t tasks_x = (int64_t)((p.n + 7) / 8) * p.nstrips * p.nchunks;       // wave tasks per (XCD, channel block)
  if ((int64_t)per * 4 > tasks_x) per = (int)((tasks_x + 3) / 4);
  hipLaunchKernelGGL((k_dwb_s1<CBW, PD>), dim3((unsigned)(8 * p.ncb * per)), dim3(256), lds, s, p);
  return frost_check_launch("dw_bwd_fused");
}

extern "C" int frost_dw_bwd_fused_ok(int h, int w, int c, int k, int stride) {
  return (k == 3 && stride == 1 && (c % 8) == 0 && w >= 16 && h >= 8) ? 1 : 0;
}

extern "C" int frost_dw_bwd_fused(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                                  int n, int h, int w, int c, int k, int stride, const float* coef, const float* qrec_y, int relu, const uint16_t* gout,
                                  uint16_t* dx, float* dwq, void* stream) {
  FROST_REQUIRE(frost_dw_bwd_fused_ok(h, w, c, k, stride), "dw_bwd_fused: unsupported shape (k = 3, stride 1, maps >= 8 x 16, channels a multiple of 8)");
  FROST_REQUIRE(x && wq_pack && wsum && coef && qrec_y && gout && dx && dwq, "dw_bwd_fused: incomplete arguments");
  DwbP p = {};
  p.x = x; p.qx = qrec_x; p.wq = wq_pack; p.wsum = wsum; p.qw = qrec_w; p.wscale = wscale; p.coef = coef; p.qy = qrec_y; p.gout = gout; p.dx = dx; p.dwq = dwq;
  p.n = n; p.h = h; p.w = w; p.c = c; p.cpad = round_up(c, 16); p.relu = relu; p.sr = frost_sr_enabled();
  p.inv_count = 1.0f / (float)((int64_t)n * h * w);
  hipStream_t s = as_stream(stream);
  // 32-channel blocks when they waste fewer lanes than 64-channel blocks (32, 72, 96 channels: the high-resolution layers), as pick_geo of frost_dw3.hip
  static const int cbw_env = getenv("FROST_DWB_CBW") ? atoi(getenv("FROST_DWB_CBW")) : 0;
  const bool c32 = cbw_env ? (cbw_env == 32) : (round_up(c, 32) < round_up(c, 64));
  return c32 ? launch_dwb<32, 2>(p, s) : launch_dwb<64, 2>(p, s);
}
