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
#include <type_traits>

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
  int nbuf;                       // LDS tile buffers of the DMA path: 2, or 1 when no workgroup gets a second tile
  int csplit, nbt;                // channel-group split across workgroups (few-tile layers), workgroups per split
  int64_t tile0;                  // first tile of this launch (the partial last tile of a tensor gets its own launch)
  int cres;                       // BN/quant coefficient rows (folded) live in LDS even when the weights do not (RES)
  int io, io_bytes, g_bytes, o_bytes;   // io bit0: gout tile arrives by DMA (reduce/dc passes); bit1: outputs leave through an LDS tile
  int gl, tile_bytes;             // gl: linear double-buffered LDS tile image filled by direct-to-LDS loads
  const float* bias; int act_relu;   // bf16 GEMM mode used as an inference layer: + bias[ch], activation code 0 none / 1 ReLU / 2 hard-swish (qw == NULL: no weight scale)
  // fused backward (k_pw<M_BDC, .., FTW > 0>): the dc tile never leaves LDS -- data gradient and weight gradient are taken from it
  const uint8_t* wtp; int KSd;       // bf16 transposed weight pack [cin tile][co step][lane][16 B], K steps of 32 output channels
  float* dwq;                        // fp32 dL/dWq accumulator [cout][cin] (atomics, one flush per workgroup)
  int dxo_off, dx_bytes;             // LDS offset / size of the dx output tile [128][cin] bf16
  int wtl_off, wtl_bytes;            // LDS-resident copy of the transposed weight pack
  int xb_off;                        // fused backward: the x tile as bf16 (q - zp) [128][cin], converted once per tile for the weight gradient's B operand
  FrostFinDesc fin; int fin_on; unsigned fin_total;   // statistics pass: finalize folded into the last workgroup's tail
  int sr;                            // dc is rounded to bf16 stochastically (unbiased; see sr_pk_bf16 in frost_common.h)
  int cvt;                           // emit pass in converted-inference form: q = rint(float(acc + b_q) * rs) + zp (QNNPACK requantisation)
};

#define BP 128
#ifndef PW_APF
#define PW_APF(MODE, WP) ((WP) == 2 || ((MODE) == M_BDC && (WP) == 8) || (((MODE) == M_STATS || (MODE) == M_EMIT || (MODE) == M_DGRAD) && (WP) != 2))
#endif
// channel tiles (16 channels each) a wave carries per channel group: the 4-wave-wide channel split (WP == 2) holds 4 pixel
// subtiles per channel tile, so its backward passes carry 2 (4 would spill the epilogue state: measured 30% slower)
#ifndef PW_MI_DGRAD2
#define PW_MI_DGRAD2 2
#endif
#ifndef PW_MI2
#define PW_MI2 2
#endif
#ifndef PW_MI2_FWD
#define PW_MI2_FWD 2      // ... the forward passes (statistics / emit) of the same split; 3 = one channel group for 9 .. 12 tiles (144 / 168 channels): measured, see DESIGN "Round 6"
#endif
#define PW_MI(MODE, WP) (((WP) == 2) ? (((MODE) == M_DGRAD) ? PW_MI_DGRAD2 : (((MODE) == M_STATS || (MODE) == M_EMIT) ? PW_MI2_FWD : PW_MI2)) : 4)

template <int CTRL> __device__ __forceinline__ int dpp_i(int v, int identity) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}
// inclusive scan over a 16-lane row with row_shr 1,2,4,8: lane 15 of every row holds the row total (pure VALU)
__device__ __forceinline__ float row_sum_f(float v) {
  v += __int_as_float(dpp_i<0x111>(__float_as_int(v), 0)); v += __int_as_float(dpp_i<0x112>(__float_as_int(v), 0));
  v += __int_as_float(dpp_i<0x114>(__float_as_int(v), 0)); v += __int_as_float(dpp_i<0x118>(__float_as_int(v), 0)); return v;
}

__device__ __forceinline__ uint32_t pack_bf2(float a, float b) { return cvt_pk_bf16(a, b); }

// Direct-to-LDS staging of one 128-pixel tile (gl mode).  A tile is 128*rowbytes CONTIGUOUS bytes of the pixel-major
// tensor (always a multiple of 1 KiB), so its LDS image is a linear copy: every wave-level global_load_lds_dwordx4 moves one
// 1 KiB unit (M0 = wave-uniform LDS base, + lane*16) and costs one VALU op -- no staging registers, no index arithmetic, and
// the copy of tile t+1 runs under the work of tile t (two LDS buffers).  The DMA is issued from inline asm on purpose: when
// the compiler sees an LDS-DMA in flight it puts s_waitcnt vmcnt(0) in front of EVERY later LDS read (it cannot disprove
// aliasing), which also drains the epilogue's stores once per channel tile.  Ordering is therefore explicit: vmcnt retires
// in issue order, so `s_waitcnt vmcnt(N)` with N <= (VMEM instructions this wave issued after its DMA) guarantees the DMA
// has landed while leaving the younger stores in flight; pw_wait_barrier() does that wait and the workgroup barrier.
// Only the last, partial tile of a tensor takes the bounds-checked register path (zero fill).
__device__ __forceinline__ void pw_glds16(const uint8_t* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void pw_stage_linear(const PwP& p, int64_t tile, uint8_t* dst, int tid) {
  const int64_t p0 = tile * BP;
  const uint8_t* src = p.T + p0 * p.rowbytes;
  if ((p0 + BP) <= p.npix) {
    const int lane = tid & 63; const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nunits = p.tile_bytes >> 10;
    const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)dst);
    for (int u = w; u < nunits; u += 8) pw_glds16(src + u * 1024 + lane * 16, lbase + u * 1024);
  } else {
    const int valid = (int)(p.npix - p0) * p.rowbytes;
    for (int o = tid * 8; o < p.tile_bytes; o += 512 * 8) {
      uint2 v = make_uint2(0, 0);
      if (o < valid) v = *(const uint2*)(src + o);
      *(uint2*)(dst + o) = v;
    }
  }
}
// full-tile DMA of `nbytes` (multiple of 1 KiB) contiguous bytes; wave w takes units w, w+8, ...
__device__ __forceinline__ void pw_dma_tile(const uint8_t* src, uint8_t* dst, int nbytes, int tid) {
  const int lane = tid & 63; const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nunits = nbytes >> 10;
  const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)dst);
  for (int u = w; u < nunits; u += 8) pw_glds16(src + u * 1024 + lane * 16, lbase + u * 1024);
}
// DMA of `nbytes` (multiple of 16) contiguous bytes, 16-byte aligned on both sides; the last unit may be partial (lanes past the end are masked off)
__device__ __forceinline__ void pw_dma_bytes(const uint8_t* src, uint8_t* dst, int nbytes, int tid) {
  const int lane = tid & 63; const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nunits = (nbytes + 1023) >> 10;
  const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)dst);
  for (int u = w; u < nunits; u += 8)
    if (u * 1024 + lane * 16 < nbytes) pw_glds16(src + u * 1024 + lane * 16, lbase + u * 1024);
}
#define PW_WB_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
__device__ __forceinline__ void pw_wait_barrier(int n_younger) {      // n_younger: wave-uniform lower bound, see above
  switch (n_younger) {
    PW_WB_CASE(1) PW_WB_CASE(2) PW_WB_CASE(3) PW_WB_CASE(4) PW_WB_CASE(5) PW_WB_CASE(6) PW_WB_CASE(7) PW_WB_CASE(8)
    PW_WB_CASE(9) PW_WB_CASE(10) PW_WB_CASE(11) PW_WB_CASE(12) PW_WB_CASE(13) PW_WB_CASE(14) PW_WB_CASE(15) PW_WB_CASE(16)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
}

// RES = "resident" mode for small layers: the packed weights, wsum and the BN/quant coefficient rows are copied into LDS
// once per workgroup, so the persistent tile loop touches global memory only for the activation stream itself (no
// per-tile L2 round trips on the critical path); the next tile arrives by DMA (pw_stage_linear).
// waves per SIMD the register allocator must leave room for (measured: 2 -> no spills anywhere but half the residency: slower)
#ifndef PW_MINW
#define PW_MINW(MODE, WP) 4
#endif
typedef short v4s_ __attribute__((ext_vector_type(4)));
typedef int v2i_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pw_trunc_bf2(float lo, float hi) {   // exact for |integers| <= 256
  return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}

// FTW > 0 (M_BDC, RES, FULLT only) = the fused backward: after the dc tile has been assembled in LDS (over the DMA'd gout tile) the same
// workgroup computes  dx[pix][ci] (+)= s_w * sum_co dc[pix][co] * wq[co][ci]   (wave w: pixel sub-tile w, all input-channel tiles) and
// dWq[co][ci] += s_x * sum_pix dc[pix][co] * (q[pix][ci] - zp)   (16x16 output tiles dealt round-robin to the 8 waves, FTW per wave, persistent
// accumulators flushed once per workgroup) -- so dc is neither written to nor re-read from HBM twice (-6 B per output element), and the
// weight gradient does not read x again.  Replaces the dc pass + frost_pw dgrad + frost_pw_wgrad for layers with Cout*Cin <= ~19 k.
#ifndef PW_FUSE_MINW
#define PW_FUSE_MINW 4
#endif
// SP > 0 = shape-specialised instance for the large narrow layers (stem, 32->16, 16->96: 3.6 ms of the step): ONE channel group whose channel
// tiles are all full (cout == WC * MIE * 16), no channel-group split, standard LDS-staged I/O.  SP = MIE + 8 * (rows 16-byte aligned).  The
// generic instance decides all of that at run time, per channel tile and per MFMA: the counters showed 130-260 SALU instructions per wave and
// tile (s_cbranch / exec masking around every guarded block, SGPRs spilled to VGPR lanes) beside ~190-320 VALU -- the scalar unit is shared
// by the CU's four SIMDs.  With the shape known the guards fold away.  SP & 16: the last channel tile is partial (cout = 24, 40: the channel
// bound stays a run-time test, so does the row alignment).
template <int MODE, int WP, bool RES, bool FULLT, int FTW = 0, int SP = 0>
__global__ __launch_bounds__(512, (FTW > 0) ? PW_FUSE_MINW : PW_MINW(MODE, WP)) void k_pw(const PwP p) {
  constexpr bool FUSE = FTW > 0;
  // XB: the fused backward's weight gradient takes its B operand from a bf16 copy of the x tile made once per tile (in the dx tile's LDS, weight gradient BEFORE the data
  // gradient).  Measured per shape: it wins where a wave owns <= 2 weight-gradient tiles (16 -> 96 @112: 655 -> 612 us, 24 -> 72 @56: 192 -> 179) and loses where it owns 6 - 11
  // (the per-fragment conversions hide under the MFMA / LDS latencies of the long unrolled tile loop; the extra phase and barrier do not): 24 -> 144 418 -> 451, 168 -> 40 166 -> 194
  constexpr bool XB = FTW == 2;
  constexpr bool SPC = SP > 0;
  constexpr bool SPF = SPC && !(SP & 16);        // every channel tile full
  constexpr int MIE = SP & 7;
  constexpr int WC = 8 / WP;          // waves along channels
  constexpr int NT = 8 / WP;          // 16-pixel tiles per wave
  constexpr int MI = PW_MI(MODE, WP);
  constexpr bool APF = !RES && PW_APF(MODE, WP);          // next-K-step weight-fragment prefetch
  constexpr bool BF = (MODE == M_DGRAD);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const bool gl = p.gl != 0;
  uint8_t* xs = smem;
  const int xs_bytes = gl ? p.nbuf * p.tile_bytes + 64 : BP * p.kstr + 64;
  // io region (full-tile kernels of small layers): every global access of the pass is a contiguous 16 B/lane stream --
  //   reduce/dc: the gout tile [128][cout] bf16 arrives by DMA (two buffers); dc is written IN PLACE over it and leaves as a
  //   linear copy;  emit/dgrad: the y / dx tile is assembled in LDS and leaves as a linear copy.
  // (4- and 8-byte accesses at a cout-byte stride kept the texture-address unit ~75 % busy: 16 requests per wave instruction.)
  // the fused and the specialised instances always stage their tile I/O through LDS: compile-time there
  const bool g_lds = FUSE || (SPC ? (MODE == M_BRED || MODE == M_BDC) : (p.io & 1) != 0), o_lds = FUSE || (SPC ? (MODE == M_EMIT || MODE == M_BDC || MODE == M_DGRAD) : (p.io & 2) != 0);
  uint8_t* const io_base = smem + xs_bytes;
  long long* l_s1 = (long long*)(smem + xs_bytes + p.io_bytes);
  unsigned long long* l_s2 = (unsigned long long*)(l_s1 + p.cpad);
  int* l_mn = (int*)(l_s2 + p.cpad);
  int* l_mx = l_mn + p.cpad;
  float* l_f1 = (float*)(smem + xs_bytes + p.io_bytes);
  float* l_f2 = l_f1 + p.cpad;
  const int red_bytes = (MODE == M_STATS) ? p.cpad * 24 : ((MODE == M_BRED) ? p.cpad * 8 : 0);
  const uint8_t* wl = smem + xs_bytes + p.io_bytes + red_bytes;                                  // [CT][KS][64][16 B]
  const int wl_bytes = RES ? (p.cpad >> 4) * p.KS * 1024 : 0;
  const bool cres = RES || p.cres;
  const float* cl = (const float*)(wl + wl_bytes);                                  // [FROST_COEF_ROWS][cpad]
  const int* wsl = (const int*)(cl + FROST_COEF_ROWS * p.cpad);                      // [cpad]
  const float* coefp = cres ? cl : p.coef;
  const int* wsump = RES ? wsl : p.wsum;

  const int tid = threadIdx.x;
  const int lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = w / WC, wc = w % WC;
  const int CT = SPC ? (8 / WP) * MIE : (p.cpad >> 4);
  const int mi_eff = SPC ? MIE : p.mi_eff;

  if (MODE == M_STATS) {
    for (int c = tid; c < p.cpad; c += 512) { l_s1[c] = 0; l_s2[c] = 0; l_mn[c] = INT32_MAX; l_mx[c] = INT32_MIN; }
  } else if (MODE == M_BRED) {
    for (int c = tid; c < p.cpad; c += 512) { l_f1[c] = 0.0f; l_f2[c] = 0.0f; }
  }
  // resident weights / weight sums / (emit) coefficient rows go to LDS by DMA: issued from inline asm they put no wait into the code that follows, so every
  // prologue copy travels in ONE memory round trip (as register copies each loop ended in a wait for its own loads: 3 - 4 round trips at the head of every launch);
  // the __syncthreads() below (vmcnt(0) + barrier) is their completion point
  if (RES) pw_dma_tile(p.wpack, (uint8_t*)wl, wl_bytes, tid);
  if (RES && MODE != M_DGRAD) pw_dma_bytes((const uint8_t*)p.wsum, (uint8_t*)wsl, p.cpad * 4, tid);
  if (cres) {
    if (MODE == M_EMIT) pw_dma_bytes((const uint8_t*)p.coef, (uint8_t*)cl, 2 * p.cpad * 4, tid);        // rows A, B
    if (MODE == M_BRED || MODE == M_BDC) {     // folded rows (see the backward epilogue): M row <- -M*R, S1 row <- E, S2 row <- F
      float* c2 = (float*)cl;
      for (int c = tid; c < p.cpad; c += 512) {
        const float Mv = p.coef[FROST_COEF_M * p.cpad + c], Rv = p.coef[FROST_COEF_R * p.cpad + c], Kv = p.coef[FROST_COEF_K1 * p.cpad + c];
        const float Ev = -Kv * (s12_sum(p.coef, p.cpad, 1, c) * p.inv_count) * Rv;
        c2[FROST_COEF_A * p.cpad + c] = p.coef[FROST_COEF_A * p.cpad + c]; c2[FROST_COEF_B * p.cpad + c] = p.coef[FROST_COEF_B * p.cpad + c];
        c2[FROST_COEF_M * p.cpad + c] = -Mv * Rv; c2[FROST_COEF_R * p.cpad + c] = Rv; c2[FROST_COEF_K1 * p.cpad + c] = Kv;
        c2[FROST_COEF_S1 * p.cpad + c] = Ev; c2[FROST_COEF_S2 * p.cpad + c] = -Kv * (s12_sum(p.coef, p.cpad, 0, c) * p.inv_count) - Ev * Mv;
      }
    }
  }
  // the data gradient's K tail reads past a dc row (next row / other buffer / the 64-byte pad): stale LDS bytes x zero weights must not be NaN
  if (FUSE) {
    for (int i = tid; i < (p.io_bytes >> 4); i += 512) ((uint4*)(smem + xs_bytes))[i] = make_uint4(0, 0, 0, 0);
    if (p.dx) for (int i = tid; i < (p.wtl_bytes >> 4); i += 512) ((uint4*)(smem + p.wtl_off))[i] = ((const uint4*)p.wtp)[i];
  }
  if (cres) __syncthreads();
  v4f wacc[FUSE ? FTW : 1];
#pragma unroll
  for (int i = 0; i < (FUSE ? FTW : 1); ++i) wacc[i] = (v4f){0.f, 0.f, 0.f, 0.f};

  uint32_t rng = sr_seed(blockIdx.x, threadIdx.x);
  int zpx = 0; float sw = 1.0f; float y_inv = 1.0f; float y_zpf = 0.0f;
  if (!BF) zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  if (MODE == M_DGRAD && p.qw) sw = p.qw[FROST_Q_SCALE];
  if (MODE == M_EMIT || MODE == M_BRED || MODE == M_BDC) { y_inv = 1.0f / p.qy[FROST_Q_SCALE]; y_zpf = (float)__float_as_int(p.qy[FROST_Q_ZP]); }
  const float qcap = (MODE == M_EMIT) ? (float)q_hi(p.qy) : 255.0f; const bool lowq = qcap < 255.0f;      // 7-bit activations (reduce_range): cvt_pk_u8 saturates at 255 only
  if (MODE == M_EMIT && p.cvt) y_inv = 1.0f;           // row A already is the requantisation scale s_x*s_w/s_y, row B carries the int32 bias bits
  float t_lo = 0.0f, t_hi = 0.0f;            // STE pass window in t = y/scale:  t_lo < t <= t_hi
  if (MODE == M_BRED || MODE == M_BDC) {
    const int zpy = __float_as_int(p.qy[FROST_Q_ZP]), qhi = q_hi(p.qy);
    const float hi0 = (float)qhi + 0.5f - (float)zpy;                             // rint(hi0) ties to the even neighbour
    t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
    if (!p.relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  }

  // lane-local accumulators that live across the persistent tile loop (single channel group only)
  // BRED with WP==2 already holds 64 accumulator registers: deferring would spill (measured 2x slower)
  // few-tile (7x7) layers: the channel groups are divided among `csplit` sets of workgroups so the launch still fills the chip
  const int bsplit = (int)blockIdx.x / p.nbt, bslot = (int)blockIdx.x - bsplit * p.nbt;
  const int cg_lo = SPC ? 0 : (bsplit * p.ngroups) / p.csplit, cg_hi = SPC ? 1 : ((bsplit + 1) * p.ngroups) / p.csplit;
  const bool defer = (WP != 2) && (cg_hi - cg_lo == 1);      // WP == 2 holds 64 accumulator registers already: deferring would spill
  long long st1[MI]; double st2[MI]; int smn[MI], smx[MI];       // STATS: lane's channel = ct*16 + j
  float br1[MI][4], br2[MI][4];                                     // BRED: lane's channels = ct*16 + 4g + r
#pragma unroll
  for (int m = 0; m < MI; ++m) {
    st1[m] = 0; st2[m] = 0.0; smn[m] = INT32_MAX; smx[m] = INT32_MIN;
#pragma unroll
    for (int r = 0; r < 4; ++r) { br1[m][r] = 0.0f; br2[m][r] = 0.0f; }
  }

  const bool al16 = SPF ? ((SP >> 3) & 1) != 0 : (p.kstr & 15) == 0;       // gl mode keeps the natural row stride: rows of 8 (mod 16) bytes -> 2 x b64 fragment reads
  int buf = 0; bool full_prev = false;
  int n_younger = 0;     // VMEM instructions this wave is certain to issue after its DMA within one tile (last channel group)
  if (gl && MODE != M_STATS) {
    if (o_lds) {                                   // the copy-out stores (exact count)
      const int nu = p.o_bytes >> 10;
      n_younger = (w < nu) ? (nu - w + 7) / 8 : 0;
      if (MODE == M_DGRAD && p.accumulate) n_younger *= 2;
    } else if (!g_lds) {
      const int ct0l = ((cg_hi - 1) * WC + wc) * mi_eff;
      int nfull = 0;
      for (int m = 0; m < mi_eff; ++m) if ((ct0l + m) * 16 + 16 <= p.cout) ++nfull;
      const int per = (MODE == M_BDC) ? 2 : ((MODE == M_DGRAD && p.accumulate) ? 2 : 1);
      n_younger = nfull * NT * per;
    }
    if (n_younger > 16) n_younger = 16;
    if (FUSE) {                     // after its DMA a wave issues only the dx copy-out (loads of the accumulate path are consumed, hence retired)
      const int nu = p.dx ? (p.dx_bytes >> 10) : 0;
      n_younger = (w < nu) ? (nu - w + 7) / 8 : 0;
      if (n_younger > 16) n_younger = 16;
    }
  }
  if (gl) {
    // K padding reads past a row's end (next row / next buffer / the 64-byte tail): harmless for int8 (zero weights), but a
    // bf16 NaN pattern times zero is NaN, so the dgrad buffers start out zeroed
    if (BF) { for (int i = tid; i < (xs_bytes >> 4); i += 512) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0); __syncthreads(); }
    if (p.tile0 + (int64_t)bslot < p.ntiles) {
      pw_stage_linear(p, p.tile0 + (int64_t)bslot, smem, tid);
      if (g_lds) pw_dma_tile((const uint8_t*)p.gout + (p.tile0 + (int64_t)bslot) * p.g_bytes, io_base, p.g_bytes, tid);
    }
  }

  for (int64_t tile = p.tile0 + bslot; tile < p.ntiles; tile += p.nbt) {
    const int64_t p0 = tile * BP;
    const bool full = FULLT;
    if (gl) {
      pw_wait_barrier(full_prev ? n_younger : 0);   // tile t has landed (each wave retired its own DMA); buffer buf^1 is free
      xs = smem + buf * p.tile_bytes;
      if (RES) {
        const int64_t nxt = tile + p.nbt;
        if (nxt < p.ntiles) {
          pw_stage_linear(p, nxt, smem + (buf ^ 1) * p.tile_bytes, tid);
          if (g_lds) pw_dma_tile((const uint8_t*)p.gout + nxt * p.g_bytes, io_base + (buf ^ 1) * p.g_bytes, p.g_bytes, tid);
        }
      }
      full_prev = full;
    }
    for (int cg = cg_lo; cg < cg_hi; ++cg) {
      const int ct0 = (cg * WC + wc) * mi_eff;
      int mi_n = CT - ct0; mi_n = mi_n < 0 ? 0 : (mi_n > mi_eff ? mi_eff : mi_n);
      if (SPC) mi_n = MIE;
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
        if (!gl && (p.nchunks > 1 || cg == cg_lo)) {
          __syncthreads();
          const uint8_t* src = p.T + p0 * p.rowbytes + kc0;
          if (((p.rowbytes | kc0) & 15) == 0) {
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
        }
        if (mi_n > 0) {
          const int ks_n = kcw_pad >> 6; const int ks0 = kc0 >> 6;
          // non-resident weights come from L2 (~400+ cycles): the fragments of K-step ks+1 are requested before the MFMAs of ks
          auto load_a = [&](int ks, v4i (&dst)[MI]) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < MI; ++m)
              if (m < mi_n) dst[m] = *(const v4i*)((RES ? wl : p.wpack) + ((((int64_t)(ct0 + m) * p.KS + ks0 + ks) * 64 + lane) << 4));
          };
          v4i afr[MI], afn[MI];
          if (APF) load_a(0, afr);
          for (int ks = 0; ks < ks_n; ++ks) {
            v4i bfr[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const uint8_t* fp = xs + ((wp * NT + t) * 16 + j) * p.kstr + ks * 64 + g * 16;
              if (al16) bfr[t] = *(const v4i*)fp;
              else { const int2 lo = *(const int2*)fp, hi = *(const int2*)(fp + 8); bfr[t] = (v4i){lo.x, lo.y, hi.x, hi.y}; }
            }
            if (APF) { if (ks + 1 < ks_n) load_a(ks + 1, afn); } else load_a(ks, afr);
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
            if (APF) {
#pragma unroll
              for (int m = 0; m < MI; ++m) afr[m] = afn[m];
            }
          }
        }
      }

      if (gl && !RES && cg == cg_hi - 1) {   // non-resident weights are global loads younger than the DMA would be: issue it after them
        const int64_t nxt = tile + p.nbt;
        if (nxt < p.ntiles) {
          pw_stage_linear(p, nxt, smem + (buf ^ 1) * p.tile_bytes, tid);
          if (g_lds) pw_dma_tile((const uint8_t*)p.gout + nxt * p.g_bytes, io_base + (buf ^ 1) * p.g_bytes, p.g_bytes, tid);
        }
      }
      uint8_t* const gcur = io_base + (g_lds ? buf * p.g_bytes : 0);      // this tile's gout / dc tile, or the y / dx out tile
      // ------------------------------------------------------------------------------- epilogue
      // FULL tiles (all but the last tile of a tensor; their own kernel instance) carry no per-pixel validity logic at all.
      auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        if (MODE == M_STATS) {
#ifdef PW_ABL_NOSTATEPI
          return;
#endif
          // lane: channel ct*16+j; register r of tile t: pixel p0 + (wp*NT+t)*16 + 4g + r.  Per element: one int add (sum),
          // cvt + fma (sum of squares, fp32 within the tile -> double across tiles), half a min3 and half a max3.
#pragma unroll
          for (int m = 0; m < MI; ++m) {
            if (m >= mi_n) continue;
            int a1 = 0; float a2 = 0.0f; int mn = INT32_MAX, mx = INT32_MIN;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              if (FULL) {
                const int v0 = acci[m][t][0], v1 = acci[m][t][1], v2 = acci[m][t][2], v3 = acci[m][t][3];
                a1 += (v0 + v1) + (v2 + v3);
                const float f0 = (float)v0, f1 = (float)v1, f2 = (float)v2, f3 = (float)v3;
                a2 = fmaf(f0, f0, a2); a2 = fmaf(f1, f1, a2); a2 = fmaf(f2, f2, a2); a2 = fmaf(f3, f3, a2);
                mn = min(mn, min(v0, v1)); mn = min(mn, min(v2, v3)); mx = max(mx, max(v0, v1)); mx = max(mx, max(v2, v3));
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int v = acci[m][t][r];
                  if ((p0 + (wp * NT + t) * 16 + 4 * g + r) < p.npix) { a1 += v; const float fv = (float)v; a2 = fmaf(fv, fv, a2); mn = min(mn, v); mx = max(mx, v); }
                }
              }
            }
            if (defer) { st1[m] += a1; st2[m] += (double)a2; smn[m] = min(smn[m], mn); smx[m] = max(smx[m], mx); }
            else {
              long long b1 = a1; double b2 = (double)a2;
              b1 += __shfl_xor(b1, 16); b1 += __shfl_xor(b1, 32); b2 += __shfl_xor(b2, 16); b2 += __shfl_xor(b2, 32);
              mn = min(mn, __shfl_xor(mn, 16)); mn = min(mn, __shfl_xor(mn, 32)); mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
              const int chn = (ct0 + m) * 16 + j;
              if (g == 0 && chn < p.cout && mn <= mx) {
                atomicAdd((unsigned long long*)&l_s1[chn], (unsigned long long)b1); atomicAdd(&l_s2[chn], (unsigned long long)__double2ll_rn(b2));
                atomicMin(&l_mn[chn], mn); atomicMax(&l_mx[chn], mx);
              }
            }
          }
          return;
        }
#pragma unroll
        for (int m = 0; m < MI; ++m) {
          if (m >= mi_n) continue;
          const int ch0 = (ct0 + m) * 16 + 4 * g;          // 4 consecutive channels of this lane
          const bool chok = SPF || ch0 < p.cout;
          if (MODE == M_DGRAD) {
            uint16_t* base = p.dx + p0 * p.cout;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && chok) { const float4 b4 = *(const float4*)(p.bias + ch0); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
            const float flo = p.act_relu == 1 ? 0.0f : -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int prow = (wp * NT + t) * 16 + j;
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(accf[m][t][r], sw, bv[r]), flo);
              if (p.act_relu == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = hswish_f(v[r]);
              }
              if (o_lds) {
                if (chok) { uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); *(uint2*)(gcur + (prow * p.cout + ch0) * 2) = o; }
              } else if ((FULL || (p0 + prow) < p.npix) && chok) {
                uint16_t* dst = base + prow * p.cout + ch0;
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
            // converted inference: y = float(acc + bias_q) * requant scale (row B carries the int32 bias bits); training / eval: y = fma(A, acc, B)
            const bool cvq = !SPC && p.cvt == 1;
            int addq[4]; float Bq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { addq[r] = cvq ? __float_as_int(B[r]) : 0; Bq[r] = cvq ? 0.0f : B[r]; }
            int8_t* base = p.y + p0 * p.cout;
            if (SPC) {
              // specialised instances: the requantisation flavour is a uniform branch around the tile loop, so the training emit pays nothing for the converted
              // forms (expressions = the generic path's below, term for term)
              auto emit_tiles = [&](auto cv_tag) __attribute__((always_inline)) {
                constexpr int CV = decltype(cv_tag)::value;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                  const int prow = (wp * NT + t) * 16 + j;
                  uint32_t packed = 0;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    float qv;
                    if (CV == 2) qv = rintf(((float)acci[m][t][r] + B[r]) * A[r]) + y_zpf;
                    else if (CV == 1) qv = rintf(fmaf(A[r], (float)(acci[m][t][r] + __float_as_int(B[r])), 0.0f) * y_inv) + y_zpf;
                    else qv = rintf(fmaf(A[r], (float)acci[m][t][r], B[r]) * y_inv) + y_zpf;
                    if (lowq) qv = fminf(qv, qcap);
                    packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);
                  }
                  if (o_lds) { if (chok) *(uint32_t*)(gcur + prow * p.cout + ch0) = packed ^ 0x80808080u; }
                  else if ((FULL || (p0 + prow) < p.npix) && chok) *(uint32_t*)(base + prow * p.cout + ch0) = packed ^ 0x80808080u;
                }
              };
              if (p.cvt == 1) emit_tiles(std::integral_constant<int, 1>{});
              else if (p.cvt == 2) emit_tiles(std::integral_constant<int, 2>{});
              else emit_tiles(std::integral_constant<int, 0>{});
              continue;
            }
            if (!SPC && p.cvt == 2) {
              // converted inference on the FBGEMM engine (per-channel 'fbgemm' qconfig): float bias, per-channel multiplier --
              // q = cvtps2dq((float(acc) + b[c] / (s_x s_w[c])) * (s_x s_w[c] / s_y)) + zp, rows B / A (oracle.fbgemm_conv).  Uniform branch per channel tile.
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const int prow = (wp * NT + t) * 16 + j;
                uint32_t packed = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float qv = rintf(((float)acci[m][t][r] + B[r]) * A[r]) + y_zpf;
                  if (lowq) qv = fminf(qv, qcap);
                  packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);
                }
                if (o_lds) { if (chok) *(uint32_t*)(gcur + prow * p.cout + ch0) = packed ^ 0x80808080u; }
                else if ((FULL || (p0 + prow) < p.npix) && chok) *(uint32_t*)(base + prow * p.cout + ch0) = packed ^ 0x80808080u;
              }
              continue;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int prow = (wp * NT + t) * 16 + j;
              uint32_t packed = 0;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                // q = clamp(rint(y*inv)+zp, 0, 255); ReLU is implied: ReLU layers have zp == 0 and v_cvt_pk_u8_f32
                // saturates at 0 (and at 255) while inserting the byte -- one op for clamp + convert + pack.
                const float yv = fmaf(A[r], (float)(acci[m][t][r] + addq[r]), Bq[r]);     // one form for both emit flavours: no per-element branch
                float qv = rintf(yv * y_inv) + y_zpf;
                if (lowq) qv = fminf(qv, qcap);
                packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);
              }
              if (o_lds) { if (chok) *(uint32_t*)(gcur + prow * p.cout + ch0) = packed ^ 0x80808080u; }
              else if ((FULL || (p0 + prow) < p.npix) && chok) *(uint32_t*)(base + prow * p.cout + ch0) = packed ^ 0x80808080u;
            }
            continue;
          }
          // backward modes.  Folded per-channel coefficients (resident mode keeps them folded in LDS):
          //   xhat = fma(acc, R, MR)                      MR = -M*R
          //   dc   = fma(gy, K1, fma(acc, E, F))          E = -K1*(S2/n)*R,  F = -K1*(S1/n) - E*M
          // STE mask in the t = y/scale domain: pass <=> t_lo < t <= t_hi (thresholds derived from the zero point so that the
          // test equals 0 <= rint(relu(y)/scale)+zp <= 255 and y > relu floor exactly, ties-to-even included).
          float R[4], MR[4], K1[4] = {0, 0, 0, 0}, E[4] = {0, 0, 0, 0}, F[4] = {0, 0, 0, 0};
          if (cres) {
            const float4 r4 = *(const float4*)(coefp + FROST_COEF_R * p.cpad + ch0), m4 = *(const float4*)(coefp + FROST_COEF_M * p.cpad + ch0);
            R[0] = r4.x; R[1] = r4.y; R[2] = r4.z; R[3] = r4.w; MR[0] = m4.x; MR[1] = m4.y; MR[2] = m4.z; MR[3] = m4.w;
            if (MODE == M_BDC) {
              const float4 k4 = *(const float4*)(coefp + FROST_COEF_K1 * p.cpad + ch0), e4 = *(const float4*)(coefp + FROST_COEF_S1 * p.cpad + ch0), f4 = *(const float4*)(coefp + FROST_COEF_S2 * p.cpad + ch0);
              K1[0] = k4.x; K1[1] = k4.y; K1[2] = k4.z; K1[3] = k4.w; E[0] = e4.x; E[1] = e4.y; E[2] = e4.z; E[3] = e4.w; F[0] = f4.x; F[1] = f4.y; F[2] = f4.z; F[3] = f4.w;
            }
          } else {
            const float4 m4 = *(const float4*)(coefp + FROST_COEF_M * p.cpad + ch0), r4 = *(const float4*)(coefp + FROST_COEF_R * p.cpad + ch0);
            const float Mv[4] = {m4.x, m4.y, m4.z, m4.w};
            R[0] = r4.x; R[1] = r4.y; R[2] = r4.z; R[3] = r4.w;
#pragma unroll
            for (int r = 0; r < 4; ++r) MR[r] = -Mv[r] * R[r];
            if (MODE == M_BDC) {
              const float4 k4 = *(const float4*)(coefp + FROST_COEF_K1 * p.cpad + ch0), a4 = s12_sum4(p.coef, p.cpad, 0, ch0), b4 = s12_sum4(p.coef, p.cpad, 1, ch0);          // (not resident: coefp == p.coef)
              const float kk[4] = {k4.x, k4.y, k4.z, k4.w}, s1[4] = {a4.x, a4.y, a4.z, a4.w}, s2[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) { K1[r] = kk[r]; E[r] = -kk[r] * (s2[r] * p.inv_count) * R[r]; F[r] = -kk[r] * (s1[r] * p.inv_count) - E[r] * Mv[r]; }
            }
          }
          const uint16_t* gbase = p.gout + p0 * p.cout;
          uint16_t* dbase = p.dc + p0 * p.cout;
          float r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int prow = (wp * NT + t) * 16 + j;
            const bool valid = (FULL || (p0 + prow) < p.npix) && chok;
            uint2 gv = make_uint2(0, 0);
            if (g_lds) { if (chok) gv = *(const uint2*)(gcur + (prow * p.cout + ch0) * 2); }
            else if (valid) gv = *(const uint2*)(gbase + prow * p.cout + ch0);
            const float gq[4] = {__uint_as_float(gv.x << 16), __uint_as_float(gv.x & 0xffff0000u), __uint_as_float(gv.y << 16), __uint_as_float(gv.y & 0xffff0000u)};
            float dcv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float af = (float)acci[m][t][r];
              const float tq = fmaf(A[r], af, B[r]) * y_inv;
              const float gy = (tq > t_lo && tq <= t_hi) ? gq[r] : 0.0f;
              if (MODE == M_BRED) { r1[r] += gy; r2[r] = fmaf(gy, fmaf(af, R[r], MR[r]), r2[r]); }
              else dcv[r] = fmaf(gy, K1[r], fmaf(af, E[r], F[r]));
            }
            if (MODE == M_BDC && valid) {
              uint2 o;
              if (p.sr) { o.x = sr_pk_bf16(dcv[0], dcv[1], rng); o.y = sr_pk_bf16(dcv[2], dcv[3], rng); }
              else { o.x = pack_bf2(dcv[0], dcv[1]); o.y = pack_bf2(dcv[2], dcv[3]); }
              if (o_lds) *(uint2*)(gcur + (prow * p.cout + ch0) * 2) = o;
              else *(uint2*)(dbase + prow * p.cout + ch0) = o;
            }
          }
          if (MODE == M_BRED) {
            if (defer) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { br1[m][r] += r1[r]; br2[m][r] += r2[r]; }
              __builtin_amdgcn_sched_barrier(0);      // keep the next channel tile's coefficient / gout loads from being hoisted (spills)
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float a = row_sum_f(r1[r]), b = row_sum_f(r2[r]);
                if (j == 15 && chok) { atomicAdd(&l_f1[ch0 + r], a); atomicAdd(&l_f2[ch0 + r], b); }
              }
            }
          }
        }
      };
      epilogue(std::integral_constant<bool, FULLT>{});
    }
    if (FUSE) {
      if (XB) {                                                                         // x tile (linear, [128][cin] offset-binary bytes) -> xb [128][cin] bf16 of (q - zp)
        const float zpf = (float)(zpx + 128);
        uint8_t* xbw = smem + p.xb_off;
        for (int u = tid; u < 16 * p.rowbytes; u += 512) {
          const uint2 rv = *(const uint2*)(xs + u * 8);
          const uint32_t u0 = rv.x ^ 0x80808080u, u1 = rv.y ^ 0x80808080u;      // offset-binary -> unsigned index
          *(uint4*)(xbw + u * 16) = make_uint4(pw_trunc_bf2((float)(u0 & 255u) - zpf, (float)((u0 >> 8) & 255u) - zpf), pw_trunc_bf2((float)((u0 >> 16) & 255u) - zpf, (float)(u0 >> 24) - zpf),
                                               pw_trunc_bf2((float)(u1 & 255u) - zpf, (float)((u1 >> 8) & 255u) - zpf), pw_trunc_bf2((float)((u1 >> 16) & 255u) - zpf, (float)(u1 >> 24) - zpf));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // the dc tile [128][cout] bf16 (and xb) is complete in LDS
      const uint8_t* dct = io_base + buf * p.g_bytes;
      const int rbd = p.cout * 2, cin = p.rowbytes;
      uint8_t* dxo = smem + p.dxo_off;
      if (XB) {                                                                   // ---- weight gradient: 16x16 (co x ci) tiles, K = the 128 pixels
        // B operand = the x tile as bf16, converted ONCE per tile by the whole workgroup (xb, filled before the barrier above): as transposed int8 reads every
        // wave converted the fragments of each of its tiles itself -- ~35 VALU per K step and tile, up to 40 % of the kernel's VALU on the 24 -> 144 / 56 -> 168 layers
        const int nbw = (cin + 15) >> 4, ntw = CT * nbw;
        const uint8_t* xb = smem + p.xb_off; const int rbx = cin * 2;
#pragma unroll
        for (int i = 0; i < FTW; ++i) {
          const int tt = w + 8 * i;
          if (tt < ntw) {
            const int a = tt / nbw, b = tt - a * nbw;
            const uint8_t* a_src = dct + (g * 8 + (j >> 2)) * rbd + (j & 3) * 8 + a * 32;
            const uint8_t* b_src = xb + (g * 8 + (j >> 2)) * rbx + (j & 3) * 8 + b * 32;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const v2i_ lo = __builtin_bit_cast(v2i_, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_ __attribute__((address_space(3)))*)(a_src + ks * 32 * rbd)));
              const v2i_ hi = __builtin_bit_cast(v2i_, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_ __attribute__((address_space(3)))*)(a_src + ks * 32 * rbd + 4 * rbd)));
              const v4i af = (v4i){lo[0], lo[1], hi[0], hi[1]};
              const v2i_ blo = __builtin_bit_cast(v2i_, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_ __attribute__((address_space(3)))*)(b_src + ks * 32 * rbx)));
              const v2i_ bhi = __builtin_bit_cast(v2i_, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_ __attribute__((address_space(3)))*)(b_src + ks * 32 * rbx + 4 * rbx)));
              const v4i bf = (v4i){blo[0], blo[1], bhi[0], bhi[1]};
              wacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af), __builtin_bit_cast(v8bf, bf), wacc[i], 0, 0, 0);
            }
          }
        }
      }
      if (XB && p.dx) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // xb shares the dx tile's LDS (same size): every wave is done reading it before the data gradient writes there
      if (p.dx) {                                                                 // ---- data gradient: D[ci][pix], wave w = pixel sub-tile w
        const float swq = p.qw[FROST_Q_SCALE];
        const int CTd = (cin + 15) >> 4;
        const uint8_t* brow = dct + (w * 16 + j) * rbd + g * 16;
        for (int c0 = 0; c0 < CTd; c0 += 4) {
          v4f dacc[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) dacc[m] = (v4f){0.f, 0.f, 0.f, 0.f};
          for (int ks = 0; ks < p.KSd; ++ks) {
            const v4i bf = *(const v4i*)(brow + ks * 64);
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (c0 + m < CTd) {
                const v4i af = *(const v4i*)(smem + p.wtl_off + ((((c0 + m) * p.KSd + ks) * 64 + lane) << 4));
                dacc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af), __builtin_bit_cast(v8bf, bf), dacc[m], 0, 0, 0);
              }
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int ch0 = (c0 + m) * 16 + 4 * g;
            if (c0 + m < CTd && ch0 < cin) {
              uint2 o; o.x = pack_bf2(dacc[m][0] * swq, dacc[m][1] * swq); o.y = pack_bf2(dacc[m][2] * swq, dacc[m][3] * swq);
              *(uint2*)(dxo + ((w * 16 + j) * cin + ch0) * 2) = o;
            }
          }
        }
      }
      if (!XB) {                                                                         // ---- weight gradient (many tiles per wave): B fragments straight from the int8 x tile
        const int nbw = (cin + 15) >> 4, ntw = CT * nbw;
        const float zpf = (float)(zpx + 128);
#pragma unroll
        for (int i = 0; i < FTW; ++i) {
          const int tt = w + 8 * i;
          if (tt < ntw) {
            const int a = tt / nbw, b = tt - a * nbw;
            const uint8_t* a_src = dct + (g * 8 + (j >> 2)) * rbd + (j & 3) * 8 + a * 32;
            const uint8_t* b_src = xs + (g * 8 + (j >> 1)) * cin + (j & 1) * 8 + b * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const v2i_ lo = __builtin_bit_cast(v2i_, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_ __attribute__((address_space(3)))*)(a_src + ks * 32 * rbd)));
              const v2i_ hi = __builtin_bit_cast(v2i_, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_ __attribute__((address_space(3)))*)(a_src + ks * 32 * rbd + 4 * rbd)));
              const v4i af = (v4i){lo[0], lo[1], hi[0], hi[1]};
              const v2i_ raw = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i_ __attribute__((address_space(3)))*)(b_src + ks * 32 * cin));
              const uint32_t u0 = (uint32_t)raw[0] ^ 0x80808080u, u1 = (uint32_t)raw[1] ^ 0x80808080u;   // offset-binary -> unsigned index
              const v4i bf = (v4i){(int)pw_trunc_bf2((float)(u0 & 255u) - zpf, (float)((u0 >> 8) & 255u) - zpf),
                                   (int)pw_trunc_bf2((float)((u0 >> 16) & 255u) - zpf, (float)(u0 >> 24) - zpf),
                                   (int)pw_trunc_bf2((float)(u1 & 255u) - zpf, (float)((u1 >> 8) & 255u) - zpf),
                                   (int)pw_trunc_bf2((float)((u1 >> 16) & 255u) - zpf, (float)(u1 >> 24) - zpf)};
              wacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af), __builtin_bit_cast(v8bf, bf), wacc[i], 0, 0, 0);
            }
          }
        }
      }
      if (p.dx) {                                                                 // linear copy-out of the dx tile
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        uint8_t* gdst = (uint8_t*)p.dx + tile * p.dx_bytes;
        const int nu = p.dx_bytes >> 10;
        for (int u = w; u < nu; u += 8) {
          uint4 v = *(const uint4*)(dxo + u * 1024 + lane * 16);
          if (p.accumulate) {
            const uint4 o = *(const uint4*)(gdst + u * 1024 + lane * 16);
            v.x = pack_bf2(bf2f(v.x & 0xffff) + bf2f(o.x & 0xffff), bf2f(v.x >> 16) + bf2f(o.x >> 16));
            v.y = pack_bf2(bf2f(v.y & 0xffff) + bf2f(o.y & 0xffff), bf2f(v.y >> 16) + bf2f(o.y >> 16));
            v.z = pack_bf2(bf2f(v.z & 0xffff) + bf2f(o.z & 0xffff), bf2f(v.z >> 16) + bf2f(o.z >> 16));
            v.w = pack_bf2(bf2f(v.w & 0xffff) + bf2f(o.w & 0xffff), bf2f(v.w >> 16) + bf2f(o.w >> 16));
          }
          *(uint4*)(gdst + u * 1024 + lane * 16) = v;
        }
      }
    } else if (o_lds) {             // linear copy-out of the assembled tile: 1 KiB per wave instruction
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const uint8_t* osrc = io_base + ((MODE == M_BDC) ? buf * p.g_bytes : 0);
      uint8_t* gdst = (MODE == M_EMIT) ? (uint8_t*)p.y + tile * p.o_bytes : ((MODE == M_BDC) ? (uint8_t*)p.dc + tile * p.o_bytes : (uint8_t*)p.dx + tile * p.o_bytes);
      const int nu = p.o_bytes >> 10;
      for (int u = w; u < nu; u += 8) {
        uint4 v = *(const uint4*)(osrc + u * 1024 + lane * 16);
        if (MODE == M_DGRAD && p.accumulate) {
          const uint4 o = *(const uint4*)(gdst + u * 1024 + lane * 16);
          v.x = pack_bf2(bf2f(v.x & 0xffff) + bf2f(o.x & 0xffff), bf2f(v.x >> 16) + bf2f(o.x >> 16));
          v.y = pack_bf2(bf2f(v.y & 0xffff) + bf2f(o.y & 0xffff), bf2f(v.y >> 16) + bf2f(o.y >> 16));
          v.z = pack_bf2(bf2f(v.z & 0xffff) + bf2f(o.z & 0xffff), bf2f(v.z >> 16) + bf2f(o.z >> 16));
          v.w = pack_bf2(bf2f(v.w & 0xffff) + bf2f(o.w & 0xffff), bf2f(v.w >> 16) + bf2f(o.w >> 16));
        }
        *(uint4*)(gdst + u * 1024 + lane * 16) = v;
      }
    }
    buf ^= 1;
  }

  if (FUSE) {                          // one flush of this workgroup's weight-gradient partials
    const int cin = p.rowbytes, nbw = (cin + 15) >> 4, ntw = CT * nbw;
    const float sx = p.qx[FROST_Q_SCALE];
    float* const dwq_w = dwq_dst(p.dwq, (int64_t)p.cout * cin);          // this workgroup's copy of the raw weight-gradient sums (frost_common.h)
#pragma unroll
    for (int i = 0; i < FTW; ++i) {
      const int tt = w + 8 * i;
      if (tt < ntw) {
        const int a = tt / nbw, b = tt - a * nbw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = a * 16 + 4 * g + r, ci = b * 16 + j;
          if (co < p.cout && ci < cin) atomicAdd(dwq_w + (int64_t)co * cin + ci, wacc[i][r] * sx);
        }
      }
    }
  }
  if (MODE == M_STATS) {
    if (defer) {
      const int ct0 = (cg_lo * WC + wc) * mi_eff;
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int chn = (ct0 + m) * 16 + j;
        long long a1 = st1[m]; double a2 = st2[m]; int mn = smn[m], mx = smx[m];
        a1 += __shfl_xor(a1, 16); a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 16); a2 += __shfl_xor(a2, 32);
        mn = min(mn, __shfl_xor(mn, 16)); mn = min(mn, __shfl_xor(mn, 32)); mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
        if (m < mi_eff && g == 0 && chn < p.cout && mn <= mx) {
          atomicAdd((unsigned long long*)&l_s1[chn], (unsigned long long)a1); atomicAdd(&l_s2[chn], (unsigned long long)__double2ll_rn(a2));
          atomicMin(&l_mn[chn], mn); atomicMax(&l_mx[chn], mx);
        }
      }
    }
    __syncthreads();
    long long* g_s1 = (long long*)stats_copy(p.stats, p.cpad); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
    int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
#ifndef PW_ABL_NOGATOM
    for (int c = cg_lo * WC * mi_eff * 16 + tid; c < p.cout && c < cg_hi * WC * mi_eff * 16; c += 512) {
      if (l_mn[c] <= l_mx[c]) {
        atomicAdd((unsigned long long*)&g_s1[c], (unsigned long long)l_s1[c]);
        atomicAdd(&g_s2[c], l_s2[c]);
        atomicMin(&g_mn[c], l_mn[c]); atomicMax(&g_mx[c], l_mx[c]);
      }
    }
#endif
#ifndef PW_ABL_NOFIN
    if (p.fin_on) {            // last workgroup done: the layer's statistics are complete -> conv finalize here instead of in its own launch
      int* sflag = (int*)smem;
      if (last_block_done2(p.fin.counter, p.fin_total, sflag)) {
        float* sh = (float*)(smem + 16);
        conv_finalize_dev(p.stats, p.npix, p.cout, p.cpad, p.qx, p.fin.qrec_w, p.fin.wscale, p.fin.gamma, p.fin.beta, p.fin.rmean, p.fin.rvar, p.fin.nbt,
                          p.fin.training, p.fin.relu, p.fin.observe, 1, p.fin.coef, p.fin.qrec_y, tid, 512, sh, p.fin.cat_qrec_b, p.fin.cat_qrec_y);
      }
    }
#endif
  } else if (MODE == M_BRED) {
    if (defer) {
      const int ct0 = (cg_lo * WC + wc) * mi_eff;
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int ch0 = (ct0 + m) * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row_sum_f(br1[m][r]), b = row_sum_f(br2[m][r]);
          if (m < mi_eff && j == 15 && ch0 < p.cout) { atomicAdd(&l_f1[ch0 + r], a); atomicAdd(&l_f2[ch0 + r], b); }
        }
      }
    }
    __syncthreads();
#ifndef PW_ABL_NOGATOM
    for (int c = cg_lo * WC * mi_eff * 16 + tid; c < p.cout && c < cg_hi * WC * mi_eff * 16; c += 512) {
      atomicAdd(s12_dst(p.coef, p.cpad, 0) + c, l_f1[c]);
      atomicAdd(s12_dst(p.coef, p.cpad, 1) + c, l_f2[c]);
    }
#endif
  }
}

template <int MODE, int WP, bool RES, bool FULLT, int FTW = 0, int SP = 0>
static int launch_pw3(PwP& p, size_t lds, int64_t tile0, int64_t tile_end, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)k_pw<MODE, WP, RES, FULLT, FTW, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  FROST_REQUIRE(lds <= 160 * 1024, "pw: LDS budget exceeded");
  auto occ_for = [](size_t bytes) {     // persistent workgroups: residency = what the register/LDS budget admits (queried, not guessed)
    static size_t key[2] = {(size_t)-1, (size_t)-1}; static int val[2] = {0, 0};
    for (int i = 0; i < 2; ++i) if (key[i] == bytes) return val[i];
    int occ = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_pw<MODE, WP, RES, FULLT, FTW, SP>, 512, bytes) != hipSuccess || occ < 1) occ = 1;
    if (occ > 4) occ = 4;
    key[1] = key[0]; val[1] = val[0]; key[0] = bytes; val[0] = occ;
    return occ;
  };
  int occ_cache = occ_for(lds);
  p.nbuf = 2;
  if (p.gl && !p.io) {      // a launch in which every workgroup gets one tile has no use for the second buffer: trade it for residency
    const size_t lds1 = lds - (size_t)p.tile_bytes;
    const int occ1 = occ_for(lds1);
    if (tile_end - tile0 <= (int64_t)256 * occ1) { p.nbuf = 1; lds = lds1; occ_cache = occ1; }
  }
  PwP q = p; q.tile0 = tile0; q.ntiles = tile_end;
  const int64_t n = tile_end - tile0;
  int64_t grid = n < 256 * occ_cache ? n : 256 * occ_cache;
  if (grid < 1) return 0;
  // statistics passes of wide layers: every workgroup ends with 4 global atomics per channel it touched, so the atomics per channel grow with the number of
  // pixel-tile workgroups; FROST_PW_STATS_CAP (dev) caps those and lets the channel-group split fill the chip instead
  static const int stats_cap = getenv("FROST_PW_STATS_CAP") ? atoi(getenv("FROST_PW_STATS_CAP")) : 0;
  if (MODE == M_STATS && stats_cap > 0 && !q.io && grid > stats_cap && q.ngroups > 1) grid = stats_cap;
  static const int cs_on = getenv("FROST_PW_CSPLIT") ? atoi(getenv("FROST_PW_CSPLIT")) : 1;
  int cs = 1;
  static const int cs_mul = getenv("FROST_PW_CSMUL") ? atoi(getenv("FROST_PW_CSMUL")) : 1;
  if (cs_on && !q.io && grid * 2 <= 256 * occ_cache * cs_mul) { cs = (int)((256 * occ_cache * cs_mul) / grid); if (cs > q.ngroups) cs = q.ngroups; if (cs < 1) cs = 1; }
  q.csplit = cs; q.nbt = (int)grid;
  q.fin_total = (unsigned)(grid * cs);
  if (SP > 0) FROST_REQUIRE(cs == 1, "pw: the specialised instance takes no channel-group split");
  hipLaunchKernelGGL((k_pw<MODE, WP, RES, FULLT, FTW, SP>), dim3((unsigned)(grid * cs)), dim3(512), lds, s, q);
  return frost_check_launch("pw");
}
// which specialised instance (k_pw's SP) fits this launch, 0 = none.  Instantiated: the stem (32 channels, 40-byte rows), 32->16, 16->96.
template <int WP>
static int pw_spec(const PwP& p, bool io_std) {
  static const int on = getenv("FROST_PW_SPEC") ? atoi(getenv("FROST_PW_SPEC")) : 1;
  constexpr int WC = 8 / WP;
  if (!on || WP == 2 || !io_std || p.ngroups != 1 || (p.cpad >> 4) != WC * p.mi_eff || !p.gl) return 0;
  const bool al = (p.kstr & 15) == 0;
  if (p.cout != p.cpad) {                  // partial last channel tile: 24 and 40 output channels (96->24, 72->24, 144->40, 168->40)
    if (WP == 8 && p.mi_eff == 2) return 18;
    if (WP == 8 && p.mi_eff == 3) return 19;
    return 0;
  }
  if (WP == 8 && p.mi_eff == 2 && !al) return 2;
  if (WP == 8 && p.mi_eff == 1) return al ? 9 : 1;
  if (WP == 4 && p.mi_eff == 3 && al) return 11;
  return 0;
}
template <int MODE, int WP>
static int launch_pw(PwP& p, hipStream_t s) {
  FROST_REQUIRE((((uintptr_t)p.wpack | (uintptr_t)p.wsum | (uintptr_t)p.coef) & 15) == 0, "pw: weight pack, weight sums and coefficient rows must be 16-byte aligned (LDS DMA)");
  size_t lds = p.gl ? (size_t)2 * p.tile_bytes + 64 : (size_t)BP * p.kstr + 64;
  if (MODE == M_STATS) lds += (size_t)p.cpad * 24;
  if (MODE == M_BRED) lds += (size_t)p.cpad * 8;
  const size_t res_bytes = (size_t)(p.cpad >> 4) * p.KS * 1024 + (size_t)p.cpad * (FROST_COEF_ROWS + 1) * 4;
  static const int res_on = getenv("FROST_PW_RES") ? atoi(getenv("FROST_PW_RES")) : 1;
  // measured per (mode, wave split): resident mode pays where the epilogue is latency-chained and registers allow it
  // resident weights: measured faster (or equal) for every pass and wave split once the tile I/O is staged; the dgrad of a
  // wide layer (dc rows too long for the DMA path) is the exception
  constexpr int res_bit = MODE * 3 + (WP == 8 ? 0 : (WP == 4 ? 1 : 2));
  static const long res_mask = getenv("FROST_PW_RESMASK") ? strtol(getenv("FROST_PW_RESMASK"), nullptr, 0) : 0x7fff;
  const bool res_ok = (((res_mask >> res_bit) & 1) != 0) && (MODE != M_DGRAD || p.gl);
  static const int io_on = getenv("FROST_PW_IO") ? atoi(getenv("FROST_PW_IO")) : 1;
  PwP pf = p;                                   // the full-tile launch may use the LDS-staged I/O path
  if (io_on && p.gl && MODE != M_STATS && (p.cout & 7) == 0) {
    pf.g_bytes = 256 * p.cout; pf.o_bytes = (MODE == M_EMIT) ? 128 * p.cout : 256 * p.cout;
    pf.io = (MODE == M_BRED) ? 1 : ((MODE == M_BDC) ? 3 : 2);
    pf.io_bytes = (pf.io & 1) ? 2 * pf.g_bytes : pf.o_bytes;
    if (lds + pf.io_bytes > 80 * 1024) { pf.io = 0; pf.io_bytes = 0; }
  }
  const size_t lds_f = lds + pf.io_bytes;
  const int64_t nfull = p.npix / BP;           // full tiles: validity-free kernel instance; the ragged tail: one extra tiny launch
  if (nfull < p.ntiles) pf.fin_on = 0;          // the folded finalize belongs to the LAST launch of the pass (stream order makes the first one's atomics visible)
  int rc = 0;
  if (nfull > 0) {
    const size_t cres_bytes = (size_t)p.cpad * FROST_COEF_ROWS * 4;
    static const int res_mint = getenv("FROST_PW_RES_MINTILES") ? atoi(getenv("FROST_PW_RES_MINTILES")) : 64;      // (was 2048: the squeeze convs of the 14 x 14 stage run 2 - 5 us faster per pass resident, measured per layer)
    static const int res_maxb = getenv("FROST_PW_RES_MAXKB") ? atoi(getenv("FROST_PW_RES_MAXKB")) : 40;
    if (res_ok && res_on && res_bytes <= (size_t)res_maxb * 1024 && lds_f + res_bytes <= (size_t)(res_maxb + 40) * 1024 && nfull >= res_mint) {
      const int sp = (MODE == M_STATS || MODE == M_EMIT || MODE == M_BRED) ? pw_spec<WP>(pf, MODE == M_STATS || pf.io != 0) : 0;
      if constexpr (WP == 8 && (MODE == M_STATS || MODE == M_EMIT || MODE == M_BRED)) {
        if (sp == 2) rc = launch_pw3<MODE, 8, true, true, 0, 2>(pf, lds_f + res_bytes, 0, nfull, s);
        else if (sp == 9) rc = launch_pw3<MODE, 8, true, true, 0, 9>(pf, lds_f + res_bytes, 0, nfull, s);
        else if (sp == 1) rc = launch_pw3<MODE, 8, true, true, 0, 1>(pf, lds_f + res_bytes, 0, nfull, s);
        else if (sp == 18) rc = launch_pw3<MODE, 8, true, true, 0, 18>(pf, lds_f + res_bytes, 0, nfull, s);
        else if (sp == 19) rc = launch_pw3<MODE, 8, true, true, 0, 19>(pf, lds_f + res_bytes, 0, nfull, s);
        else rc = launch_pw3<MODE, WP, true, true>(pf, lds_f + res_bytes, 0, nfull, s);
      } else if constexpr (WP == 4 && (MODE == M_STATS || MODE == M_EMIT || MODE == M_BRED)) {
        if (sp == 11) rc = launch_pw3<MODE, 4, true, true, 0, 11>(pf, lds_f + res_bytes, 0, nfull, s);
        else rc = launch_pw3<MODE, WP, true, true>(pf, lds_f + res_bytes, 0, nfull, s);
      } else rc = launch_pw3<MODE, WP, true, true>(pf, lds_f + res_bytes, 0, nfull, s);
    }
    else if ((MODE == M_EMIT || MODE == M_BRED || MODE == M_BDC) && lds_f + cres_bytes <= 80 * 1024) { pf.cres = 1; rc = launch_pw3<MODE, WP, false, true>(pf, lds_f + cres_bytes, 0, nfull, s); }
    else rc = launch_pw3<MODE, WP, false, true>(pf, lds_f, 0, nfull, s);
  }
  if (rc == 0 && nfull < p.ntiles) rc = launch_pw3<MODE, WP, false, false>(p, lds, nfull, p.ntiles, s);
  return rc;
}

static int pw_wave_split(int CT);
template <int MODE>
static int dispatch_pw(PwP& p, hipStream_t s) {
  const int CT = p.cpad / 16;
  int WPsel = pw_wave_split(CT);
  int WC = 8 / WPsel;
  const int MI = PW_MI(MODE, WPsel);
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
  static const int gl_on = getenv("FROST_PW_GL") ? atoi(getenv("FROST_PW_GL")) : 1;
  p.gl = 0; p.tile_bytes = BP * rowbytes;
  if (gl_on && p.nchunks == 1 && p.tile_bytes <= 32 * 1024 && (rowbytes & 7) == 0) { p.gl = 1; p.kstr = rowbytes; }
  p.inv_count = 1.0f / (float)npix;
}

// pixel-wave split of the 8 waves for a layer of CT channel tiles: 8 pixel waves x 1 channel wave up to 4 tiles, 4 x 2 up to 8, 2 x 4 above.
// FROST_PW_WP8_CT (dev): bit mask of CT values that take 8 x 1 instead (a tile count the channel waves cannot share evenly, e.g. 9 = 3 groups of 3 for one channel wave)
static int pw_wave_split(int CT) {
  static const long wp8 = getenv("FROST_PW_WP8_CT") ? strtol(getenv("FROST_PW_WP8_CT"), nullptr, 0) : 0;
  if (CT < 63 && ((wp8 >> CT) & 1)) return 8;
  return CT <= 4 ? 8 : (CT <= 8 ? 4 : 2);
}

// ---- fused backward (dc + dgrad + wgrad in one kernel), host side
struct FusePlan { int ok, wp, ftw; size_t lds; int dxo_off, wtl_off, wtl_bytes, xb_off; };
static FusePlan fuse_plan(int64_t npix, int cin, int cout, bool want_dx) {
  FusePlan f = {};
  // the fused kernel is built from the resident-weight, DMA-staged, LDS-I/O instance: any of those switched off (A/B runs) switches it off too
  static const int on = (getenv("FROST_PW_FUSE") ? atoi(getenv("FROST_PW_FUSE")) : 1) && (getenv("FROST_PW_GL") ? atoi(getenv("FROST_PW_GL")) : 1) &&
                        (getenv("FROST_PW_IO") ? atoi(getenv("FROST_PW_IO")) : 1) && (getenv("FROST_PW_RESMASK") ? strtol(getenv("FROST_PW_RESMASK"), nullptr, 0) != 0 : 1);
  if (!on || (cin & 7) || (cout & 7) || npix < BP || (npix % BP) != 0) return f;   // full 128-pixel tiles only (ragged tensors keep the three-kernel path)
  const int cpad = round_up(cout, 16), CT = cpad / 16, KS = (cin + 63) / 64;
  const int tile_bytes = BP * cin;
  if (cin > 512 || tile_bytes > 32 * 1024) return f;                         // the DMA'd linear x tile (set_tiling's gl rule)
  const int ntw = CT * ((cin + 15) / 16);
  if (ntw > 88) return f;
  f.wp = pw_wave_split(CT);
  f.ftw = ntw <= 16 ? 2 : (ntw <= 48 ? 6 : 11);
  if ((f.wp == 4 && f.ftw != 2) || (f.wp == 2 && f.ftw == 2)) return f;     // instantiated pairs only
  const size_t io_bytes = (size_t)2 * 256 * cout + 64;
  const size_t res_bytes = (size_t)CT * KS * 1024 + (size_t)cpad * (FROST_COEF_ROWS + 1) * 4;
  size_t lds = (size_t)2 * tile_bytes + 64 + io_bytes + res_bytes;
  f.dxo_off = (int)((lds + 15) & ~(size_t)15);
  f.wtl_off = f.dxo_off + 256 * cin;
  f.wtl_bytes = want_dx ? ((cin + 15) / 16) * ((cout + 31) / 32) * 1024 : 0;
  f.xb_off = f.dxo_off;                                        // the bf16 x tile of the weight gradient lives where the dx tile is assembled afterwards (256 * cin bytes either way)
  f.lds = (size_t)f.wtl_off + f.wtl_bytes + 64;               // (+ slack: the last K group's second read of a partial channel tile runs past the tile)
  if (f.lds > 160 * 1024) return f;
  f.ok = 1;
  return f;
}
int frost_gemm_bf16_rows(const uint16_t* x, const uint16_t* pack, const float* bias, int64_t npix, int k, int n, int relu, uint16_t* y, hipStream_t s);      // frost_wgrad.hip
int frost_pwc_stats_fin_ok(int64_t npix, int cin, int cout);                                                                                                    // frost_pwc.hip
int frost_pwc_stats_fin(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout, void* stats, const FrostFinDesc* fin,
                        hipStream_t s);
extern "C" int frost_pw_bwd_fused_ok(int64_t npix, int cin, int cout) { return fuse_plan(npix, cin, cout, true).ok; }

extern "C" int frost_pw_conv_bwd_fused(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                                       const uint16_t* wt_pack, const float* qrec_w, int64_t npix, int cin, int cout, float* coef,
                                       const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc_scratch, uint16_t* dx,
                                       int accumulate, float* dwq, void* stream) {
  const FusePlan f = fuse_plan(npix, cin, cout, dx != nullptr);
  FROST_REQUIRE(f.ok, "pw_bwd_fused: shape not supported by the fused kernel (ask frost_pw_bwd_fused_ok first)");
  FROST_REQUIRE(dx == nullptr || wt_pack != nullptr, "pw_bwd_fused: the data gradient needs the transposed weight pack");
  PwP p = {};
  p.T = (const uint8_t*)x; p.cout = cout; p.cpad = round_up(cout, 16); p.wpack = (const uint8_t*)wq_pack;
  p.wsum = wsum; p.qx = qrec_x; p.qy = qrec_y; p.qw = qrec_w; p.coef = coef; p.relu = relu; p.gout = gout; p.dc = dc_scratch;
  set_tiling(p, npix, cin);
  p.sr = frost_sr_enabled();
  FROST_REQUIRE(p.gl, "pw_bwd_fused: linear tile staging unavailable");
  const int CT = p.cpad / 16, WC = 8 / f.wp, MI = PW_MI(M_BDC, f.wp);
  p.ngroups = (CT + WC * MI - 1) / (WC * MI);
  p.mi_eff = (CT + p.ngroups * WC - 1) / (p.ngroups * WC);
  const int64_t nfull = npix / BP;
  PwP pf = p;
  pf.g_bytes = 256 * cout; pf.o_bytes = 256 * cout; pf.io = 3; pf.io_bytes = 2 * pf.g_bytes + 64;
  pf.wtp = (const uint8_t*)wt_pack; pf.KSd = (cout + 31) / 32; pf.dwq = dwq; pf.dx = dx; pf.accumulate = accumulate;
  pf.dxo_off = f.dxo_off; pf.dx_bytes = 256 * cin; pf.wtl_off = f.wtl_off; pf.wtl_bytes = f.wtl_bytes; pf.xb_off = f.xb_off;
  hipStream_t s = as_stream(stream);
  int rc;
  const int sp = (f.ftw == 2 && nfull >= 2048) ? (f.wp == 8 ? pw_spec<8>(pf, true) : (f.wp == 4 ? pw_spec<4>(pf, true) : 0)) : 0;
  if (sp == 2) rc = launch_pw3<M_BDC, 8, true, true, 2, 2>(pf, f.lds, 0, nfull, s);
  else if (sp == 9) rc = launch_pw3<M_BDC, 8, true, true, 2, 9>(pf, f.lds, 0, nfull, s);
  else if (sp == 1) rc = launch_pw3<M_BDC, 8, true, true, 2, 1>(pf, f.lds, 0, nfull, s);
  else if (sp == 18) rc = launch_pw3<M_BDC, 8, true, true, 2, 18>(pf, f.lds, 0, nfull, s);
  else if (sp == 19) rc = launch_pw3<M_BDC, 8, true, true, 2, 19>(pf, f.lds, 0, nfull, s);
  else if (sp == 11) rc = launch_pw3<M_BDC, 4, true, true, 2, 11>(pf, f.lds, 0, nfull, s);
  else if (f.wp == 8) rc = (f.ftw == 2) ? launch_pw3<M_BDC, 8, true, true, 2>(pf, f.lds, 0, nfull, s) : launch_pw3<M_BDC, 8, true, true, 6>(pf, f.lds, 0, nfull, s);
  else if (f.wp == 4) rc = launch_pw3<M_BDC, 4, true, true, 2>(pf, f.lds, 0, nfull, s);
  else rc = (f.ftw == 6) ? launch_pw3<M_BDC, 2, true, true, 6>(pf, f.lds, 0, nfull, s) : launch_pw3<M_BDC, 2, true, true, 11>(pf, f.lds, 0, nfull, s);
  if (rc) return rc;
  return 0;
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
  p.cvt = (mode == 2) ? 1 : ((mode == 3) ? 2 : 0);          // 2: QNNPACK-form converted inference, 3: FBGEMM-form (float bias, per-channel multipliers)
  return dispatch_pw<M_EMIT>(p, as_stream(stream));
}

extern "C" int frost_pw_conv_fwd_fin(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout,
                                     void* stats, const FrostFinDesc* fin, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 4 == 0, "pw_fwd: cin must be a multiple of 8, cout of 4");
  FROST_REQUIRE(((uintptr_t)x & 15) == 0, "pw_fwd: x must be 16B aligned");
  FROST_REQUIRE(fin && fin->counter && fin->coef && fin->qrec_y, "pw_fwd_fin: incomplete finalize descriptor");
  if (frost_pwc_stats_fin_ok(npix, cin, cout)) return frost_pwc_stats_fin(x, qrec_x, wq_pack, wsum, npix, cin, cout, stats, fin, as_stream(stream));      // wide layers: chunked form
  PwP p = {};
  p.T = (const uint8_t*)x; p.cout = cout; p.cpad = round_up(cout, 16); p.wpack = (const uint8_t*)wq_pack;
  p.wsum = wsum; p.qx = qrec_x; p.qy = fin->qrec_y; p.coef = fin->coef; p.stats = (uint8_t*)stats; p.relu = fin->relu;
  p.fin = *fin; p.fin_on = 1;
  set_tiling(p, npix, cin);
  return dispatch_pw<M_STATS>(p, as_stream(stream));
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
  p.sr = frost_sr_enabled();
  if (pass == 0) return dispatch_pw<M_BRED>(p, as_stream(stream));
  if (pass == 1) return dispatch_pw<M_BDC>(p, as_stream(stream));
  // pass 2: dgrad   dx[pix][cin] (+)= s_w * sum_co dc[pix][co] * wq[co][cin]
  PwP d = {};
  d.T = (const uint8_t*)dc; d.cout = cin; d.cpad = round_up(cin, 16); d.wpack = (const uint8_t*)wt_pack;
  d.qw = qrec_w; d.dx = dx; d.accumulate = accumulate;
  set_tiling(d, npix, cout * 2);
  return dispatch_pw<M_DGRAD>(d, as_stream(stream));
}

// bf16 inference layer of the float model (BASELINE.json config c2) on the same skeleton as the data gradient: y[p][co] = act(sum_k
// x[p][k] * W'[co][k] + b'[co]) -- DMA double-buffered pixel tiles, LDS-resident packed weights on small layers, LDS-staged output.
// replaces (eval mode): frostnet.py:14-60 Conv2d + BatchNorm2d(running stats, folded by frost_infer_weight_prep) + ReLU for 1x1 convs.
extern "C" int frost_infer_pw(const uint16_t* x, const uint16_t* pack, const float* biasf, int64_t npix, int cin, int cout, int relu,
                              uint16_t* y, void* stream) {
  FROST_REQUIRE(cin % 8 == 0 && cout % 8 == 0, "infer_pw: channels must be multiples of 8");
  static const int wide_min = getenv("FROST_INFER_WIDE") ? atoi(getenv("FROST_INFER_WIDE")) : 129;     // rows of more than 256 bytes: the stand-alone GEMM (0 = off)
  if (wide_min > 0 && cin >= wide_min && npix >= 256) return frost_gemm_bf16_rows(x, pack, biasf, npix, cin, cout, relu, y, as_stream(stream));
  PwP d = {};
  d.T = (const uint8_t*)x; d.cout = cout; d.cpad = round_up(cout, 16); d.wpack = (const uint8_t*)pack;
  d.qw = nullptr; d.dx = y; d.accumulate = 0; d.bias = biasf; d.act_relu = relu;
  set_tiling(d, npix, cin * 2);
  return dispatch_pw<M_DGRAD>(d, as_stream(stream));
}
