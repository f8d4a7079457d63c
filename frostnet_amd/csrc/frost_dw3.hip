// Depthwise k x k conv (k in {3,5}, stride {1,2}) on fake-quantised NHWC bytes -- "lane = channel" formulation.
//
// A workgroup (4 waves) owns an 8x16 output tile of a 64-channel block; every LANE owns ONE channel and a 4-row x 8-column
// patch of outputs.  The halo tile is staged in LDS in its natural [row][col][64 ch] layout with coalesced loads and each
// lane pulls 8 consecutive x-pixels OF ITS OWN CHANNEL with one gfx950 LDS transpose read (ds_read_b64_tr_b8; semantics
// measured in tools/probe_tr.hip), converts them once and slides down the tile with fp32 FMAs (exact: |sum| < 2^24).
// Consequences: 25 (not 50-100) weight registers per thread, no cross-lane reductions anywhere (per-channel statistics,
// S1/S2 and the 25 weight-gradient sums are lane-local across the whole persistent tile loop), per-channel coefficients
// are scalars per lane.  Results go back to NHWC through an LDS transpose (ds_write_b8/b16 -> coalesced 8/16-byte stores).
// Zero padding = zero-point fill, so acc_true = sum(w*q) - zp*sum(w) also holds at the borders.
#include "frost_common.h"
#include <map>
#include <type_traits>

typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
enum { D_STATS = 0, D_EMIT = 1, D_BRED = 2, D_BDC = 3, D_BDCW = 4 };   // D_BDCW: dc pass + the weight gradient in the same sweep
#define TH 8
#define RH 4
#define RW 8
// Tile geometry.  A tile is TH x TWT outputs of a CBW-channel block, made of NSUB horizontally adjacent sub-tiles of SUBW
// columns; every sub-tile is a different (image, row-tile, column-tile) unit with its own halo, so small feature maps pack
// several images into one tile (7x7 maps: 38% -> 77% of the lanes do useful work) instead of padding one image to 8x16.
//   CBW = 64: lane = channel, wave (wy, wx) owns the 4 x 8 patch at rows 4wy, columns 8wx           (TWT = 16)
//   CBW = 32: lane = (channel = lane & 31, half = lane >> 5), the wave owns two patches, columns 16wx + 8half   (TWT = 32)
//             -- for channel counts just above a multiple of 64 (32, 72, 96, 144: the high-resolution layers).
template <int K_, int S_, int CBW_, int SUBW_>
struct DwGeo {
  static constexpr int K = K_, S = S_, CBW = CBW_, SUBW = SUBW_;
  static constexpr int TWT = (CBW == 64) ? 16 : 32;
  static constexpr int NSUB = TWT / SUBW;
  static constexpr int UPP = CBW / 8;                          // 8-byte units per pixel
  static constexpr int IH = (TH - 1) * S + K, IWS = (SUBW - 1) * S + K, IWT = NSUB * IWS;
  static constexpr int IN_BYTES = ((IH * IWT * CBW + 255) / 256) * 256 + 256;
  static constexpr int NXR = (RW - 1) * S + K, NBLK = (NXR + 7) / 8, NROW = (RH - 1) * S + K;
  static constexpr int AUX_BYTES = TH * TWT * CBW * 2;         // bf16 gout / dc tile (16 KB); int8 out tile uses half
};

struct Dw3P {
  const int8_t* x; const float* qx; const int8_t* wq; const int32_t* wsum; const float* qw;
  int n, h, w, c, cpad, ho, wo, pad;
  uint8_t* stats; float* coef; const float* qy; int relu; int8_t* y;
  const uint16_t* gout; uint16_t* dc; float* dwq; uint16_t* dx; int accumulate;
  int tiles_x, tiles_y, ncb, ngroups; int64_t nunits, ntiles; float inv_count;
  FrostFinDesc fin; int fin_on;       // statistics pass: finalize folded into the last workgroup's tail
  const float* wscale;   // [cpad] per-output-channel weight scale (NULL: the scalar of qw)
  int sr;        // dc is rounded to bf16 stochastically (unbiased; see sr_bf16 in frost_common.h)
  int cvt;       // emit pass in converted-inference form: q = rint(float(acc + b_q) * rs) + zp (QNNPACK requantisation), see frost_convert.hip
  int nb;        // LDS input-tile buffers (1 or 2), see dw_nbuf
  int xmap;      // XCD-aware block -> (channel block, tile range) map, see dw_block_map
};

// Which channel block and which tiles a workgroup takes.  Plain map: cb = b % ncb, tiles grp, grp + ngroups, ...  With it, the channel blocks of
// one spatial tile (which split the same 128-byte lines when C is not a multiple of 128) and neighbouring tiles (which share their halos) land
// on DIFFERENT XCDs (block b runs on XCD b % 8), so every XCD's L2 fetched the same lines from HBM: counters showed 2.5-3.3x the algorithmic
// bytes.  XCD-aware map (speed only): XCD x owns the contiguous tile range [x*T/8, (x+1)*T/8); inside it, local index li = b / 8 gives
// cb = li % ncb and the group gl = li / ncb, tiles t0 + gl, t0 + gl + gpx, ...  (ngroups = 8 * gpx).
struct DwMap { int cb; int64_t t0, t1; int step; };
__device__ __forceinline__ DwMap dw_block_map(const Dw3P& p) {
  DwMap m;
  if (p.xmap) {
    const int b = blockIdx.x, xcd = b & 7, li = b >> 3, gpx = p.ngroups >> 3;
    m.cb = li % p.ncb; const int gl = li / p.ncb;
    const int64_t lo = (p.ntiles * xcd) >> 3;
    m.t0 = lo + gl; m.t1 = (p.ntiles * (xcd + 1)) >> 3; m.step = gpx;
  } else { m.cb = blockIdx.x % p.ncb; m.t0 = blockIdx.x / p.ncb; m.t1 = p.ntiles; m.step = p.ngroups; }
  return m;
}

__host__ __device__ constexpr int fdiv3(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// per-sub-tile unit decode (wave-uniform): image and output-domain origin of sub-tile s of block tile `tile`.  The first tile of a workgroup is
// decoded with divisions; every later one is reached by dw_it_next with adds and compares only (the persistent loop visits tiles at a constant
// stride, so the (image, tile row, tile column) increment is a constant plus carries) -- the 64-bit division alone was ~160 scalar
// instructions per tile, a quarter of the SALU stream of the stride-1 kernels.
template <int NSUB, int SUBW>
struct DwSub { int img[NSUB], r0[NSUB], c0[NSUB]; bool ok[NSUB]; };
template <int NSUB>
struct DwIt { int u[NSUB], img[NSUB], ty[NSUB], tx[NSUB]; };          // iterator state: unit index, image, tile row, tile column per sub-tile
struct DwStep { int du, dimg, dy, dx; };
template <int NSUB>
__device__ __forceinline__ void dw_it_init(const Dw3P& p, int64_t tile, DwIt<NSUB>& it) {
  const int tpi = p.tiles_x * p.tiles_y;
#pragma unroll
  for (int s2 = 0; s2 < NSUB; ++s2) {
    const int u = (int)tile * NSUB + s2;
    const int img = u / tpi; const int tr = u - img * tpi;
    it.u[s2] = u; it.img[s2] = img; it.ty[s2] = tr / p.tiles_x; it.tx[s2] = tr - it.ty[s2] * p.tiles_x;
  }
}
template <int NSUB>
__device__ __forceinline__ DwStep dw_step(const Dw3P& p, int step) {
  const int tpi = p.tiles_x * p.tiles_y;
  DwStep d; d.du = step * NSUB; d.dimg = d.du / tpi;
  const int rem = d.du - d.dimg * tpi; d.dy = rem / p.tiles_x; d.dx = rem - d.dy * p.tiles_x;
  return d;
}
template <int NSUB>
__device__ __forceinline__ void dw_it_next(const Dw3P& p, const DwStep& d, DwIt<NSUB>& it) {
#pragma unroll
  for (int s2 = 0; s2 < NSUB; ++s2) {
    it.u[s2] += d.du;
    int tx = it.tx[s2] + d.dx; const int cx = tx >= p.tiles_x ? 1 : 0; tx -= cx ? p.tiles_x : 0;
    int ty = it.ty[s2] + d.dy + cx; const int cy = ty >= p.tiles_y ? 1 : 0; ty -= cy ? p.tiles_y : 0;
    it.img[s2] += d.dimg + cy; it.tx[s2] = tx; it.ty[s2] = ty;
  }
}
template <int NSUB, int SUBW>
__device__ __forceinline__ void dw_fill(const Dw3P& p, const DwIt<NSUB>& it, DwSub<NSUB, SUBW>& su) {
#pragma unroll
  for (int s2 = 0; s2 < NSUB; ++s2) {
    su.ok[s2] = it.u[s2] < (int)p.nunits; su.img[s2] = it.img[s2]; su.r0[s2] = it.ty[s2] * TH; su.c0[s2] = it.tx[s2] * SUBW;
  }
}
#define DW_SEL(ARR, SUBIDX) ((NSUB > 1 && (SUBIDX) == 1) ? ARR[NSUB > 1 ? 1 : 0] : ((NSUB > 2 && (SUBIDX) == 2) ? ARR[NSUB > 2 ? 2 : 0] : ((NSUB > 3 && (SUBIDX) == 3) ? ARR[NSUB > 3 ? 3 : 0] : ARR[0])))

// 8 consecutive x-pixels of this lane's channel (offset-binary bytes) -> 8 unsigned-index floats
template <int CBW>
__device__ __forceinline__ void tr8_run(const uint8_t* tile_row, int col0, int lane, float* out8) {
  const int jp = lane & 15, G = (lane >> 4) & (CBW / 16 - 1);
  const v2i raw = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(tile_row + (col0 + (jp >> 1)) * CBW + 16 * G + 8 * (jp & 1)));
  const uint32_t u0 = (uint32_t)raw[0] ^ 0x80808080u, u1 = (uint32_t)raw[1] ^ 0x80808080u;
  out8[0] = (float)(u0 & 255u); out8[1] = (float)((u0 >> 8) & 255u); out8[2] = (float)((u0 >> 16) & 255u); out8[3] = (float)(u0 >> 24);
  out8[4] = (float)(u1 & 255u); out8[5] = (float)((u1 >> 8) & 255u); out8[6] = (float)((u1 >> 16) & 255u); out8[7] = (float)(u1 >> 24);
}
// the same 8 pixels as packed offset-binary bytes (two dwords) -- operands of the integer dot-product path
template <int CBW>
__device__ __forceinline__ v2i tr8_raw(const uint8_t* tile_row, int col0, int lane) {
  const int jp = lane & 15, G = (lane >> 4) & (CBW / 16 - 1);
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(tile_row + (col0 + (jp >> 1)) * CBW + 16 * G + 8 * (jp & 1)));
}
// 4 consecutive x-pixels of this lane's channel from a bf16 [row][col][CBW ch] tile
template <int CBW>
__device__ __forceinline__ void tr16_run(const uint8_t* tile_row, int col0, int lane, float* out4) {
  const int jp = lane & 15, G = (lane >> 4) & (CBW / 16 - 1);
  const v4s raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(tile_row + ((col0 + (jp >> 2)) * CBW + 16 * G + 4 * (jp & 3)) * 2));
  out4[0] = bf2f((uint16_t)raw[0]); out4[1] = bf2f((uint16_t)raw[1]); out4[2] = bf2f((uint16_t)raw[2]); out4[3] = bf2f((uint16_t)raw[3]);
}

// Tile staging: direct-to-LDS loads.  A halo tile is [row][col][CBW channels] -- a linear array of 16-byte units (16 int8 or 8 bf16
// channels of one pixel), 64 consecutive units = 1 KiB per wave instruction.  Every lane of a global_load_lds_dwordx4 supplies its own
// global address (row / column / channel-block strided) while the LDS side is linear (M0 base + lane * 16), so the natural tile layout IS
// the DMA layout: no staging registers, no ds_write, and the copy of tile t+1 runs under the arithmetic of tile t (two buffers where the
// LDS budget keeps two workgroups per CU, see dw_nbuf).  Units that fall outside the image (zero padding = zero-point fill) or belong to an
// absent sub-tile are written by the same lane with an ordinary 16-byte LDS store instead (exec-masked DMA lanes write nothing).
// The DMA is issued from inline asm (the compiler would otherwise drain vmcnt in front of every later LDS read); dw_wait_barrier() is the
// one place where a wave waits for its own copies before the workgroup barrier.  When C % 16 == 8 the last unit of a pixel reads 8 bytes
// of its neighbour (dead lanes: their weights are zero and they never reach a statistic or a store; buffers carry SLACK bytes).
__device__ __forceinline__ void dw_glds16(const uint8_t* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dw_wait_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Which (row, column, channel unit) of the tile a thread moves is the same for every tile of the persistent loop: decoded once per
// thread, one packed register per unit.  EB = bytes per element (1: int8 activations, 2: bf16 gradients).
// LEAN: nothing is kept, the decode is redone per unit (the k = 5, stride 2 weight-gradient kernel has no registers to spare).
template <int NR, int NCS, int NSUB, int CBW, int EB, bool LEAN = false>
struct DwPlan {
  static constexpr int UPP = CBW * EB / 16, CPU = 16 / EB, NC = NSUB * NCS, NUNIT = NR * NC * UPP, NU = (NUNIT + 255) / 256;
  int pos_[LEAN ? 1 : NU];        // iy | ix << 8 | sub << 16 | live << 20
  int cc0, ww_, c_, tid_, cb_;    // this thread's channel offset (the same for all its units: 256 is a multiple of UPP), row pitch, channels
  __device__ __forceinline__ static int decode(int tid, int jn, int cb, int c) {
    const int u = tid + jn * 256; const int cu = u % UPP; const int pix = u / UPP; const int iy = pix / NC, ixx = pix - iy * NC;
    const int sb = ixx / NCS; const int ix = ixx - sb * NCS; const int cc = cb * CBW + cu * CPU;
    const bool live = (u < NUNIT) && (cc < c);
    return iy | (ix << 8) | (sb << 16) | ((live ? 1 : 0) << 20);
  }
  __device__ __forceinline__ int pos(int jn) const { return LEAN ? decode(tid_, jn, cb_, c_) : pos_[LEAN ? 0 : jn]; }
  __device__ __forceinline__ int rel(int ps) const { return ((ps & 255) * ww_ + ((ps >> 8) & 255)) * c_ + cc0; }
  __device__ __forceinline__ void init(int tid, int cb, int ww, int c) {
    if (!LEAN) {
#pragma unroll
      for (int jn = 0; jn < NU; ++jn) pos_[LEAN ? 0 : jn] = decode(tid, jn, cb, c);
    }
    cc0 = cb * CBW + (tid % UPP) * CPU; ww_ = ww; c_ = c; tid_ = tid; cb_ = cb;
  }
};
// sub-tile s has its origin at row su.r0[s] * mul / dv + roff (columns alike) of image su.img[s] in the [hh][ww][c] tensor
template <int NR, int NCS, int NSUB, int SUBW, int CBW, int EB, bool LEAN>
__device__ __forceinline__ void stage_tile(const void* __restrict__ src, uint8_t* tile, int tid, const DwPlan<NR, NCS, NSUB, CBW, EB, LEAN>& pl,
                                           const DwSub<NSUB, SUBW>& su, int mul, int dv, int roff, int hh, int ww, int c, uint32_t fill) {
  typedef DwPlan<NR, NCS, NSUB, CBW, EB, LEAN> P;
  int gy0[NSUB], gx0[NSUB]; const uint8_t* bp[NSUB];
#pragma unroll
  for (int s2 = 0; s2 < NSUB; ++s2) {
    gy0[s2] = su.r0[s2] * mul / dv + roff; gx0[s2] = su.c0[s2] * mul / dv + roff;
    bp[s2] = (const uint8_t*)src + (((int64_t)su.img[s2] * hh + gy0[s2]) * ww + gx0[s2]) * c * EB;
  }
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)tile);
  const uint4 f4 = make_uint4(fill, fill, fill, fill);
  auto unit = [&](int jn) __attribute__((always_inline)) {
    const int ps = pl.pos(jn); const int sb = (ps >> 16) & 3;
    const int gy = DW_SEL(gy0, sb) + (ps & 255), gx = DW_SEL(gx0, sb) + ((ps >> 8) & 255);
    if (ps >> 20) {
      if (DW_SEL(su.ok, sb) && (unsigned)gy < (unsigned)hh && (unsigned)gx < (unsigned)ww)
        dw_glds16(DW_SEL(bp, sb) + (int64_t)pl.rel(ps) * EB, lbase + (uint32_t)(jn * 4 + wv) * 1024u);
      else *(uint4*)(tile + (tid + jn * 256) * 16) = f4;
    }
  };
  if (LEAN) {
#pragma unroll 1
    for (int jn = 0; jn < P::NU; ++jn) unit(jn);
  } else {
#pragma unroll
    for (int jn = 0; jn < P::NU; ++jn) unit(jn);
  }
}
// copy an assembled [TH][TWT][CBW] tile (EB bytes per element) out to the NHWC tensor dst[img][hh][ww][c]
template <int EB, int NSUB, int SUBW, int CBW, bool ACC>
__device__ __forceinline__ void copy_out_tile(const uint8_t* tile, uint8_t* dst, int tid, const DwSub<NSUB, SUBW>& su, int cb, int hh, int ww, int c) {
  constexpr int TWT = NSUB * SUBW, UPP = CBW / 8, NUNIT = TH * TWT * UPP;     // units of 8 channels
#pragma unroll
  for (int jn = 0; jn < NUNIT / 256; ++jn) {
    const int u = tid + jn * 256; const int c8 = u % UPP, lp = u / UPP; const int row = lp / TWT, col = lp - row * TWT;
    const int sb = col / SUBW; const int oy = DW_SEL(su.r0, sb) + row, ox = DW_SEL(su.c0, sb) + (col - sb * SUBW); const int cc = cb * CBW + c8 * 8;
    if (DW_SEL(su.ok, sb) && oy < hh && ox < ww && cc < c) {
      uint8_t* d = dst + ((((int64_t)DW_SEL(su.img, sb) * hh + oy) * ww + ox) * c + cc) * EB;
      if (EB == 1) *(uint2*)d = *(const uint2*)(tile + lp * CBW + c8 * 8);
      else {
        uint4 v = *(const uint4*)(tile + (lp * CBW + c8 * 8) * 2);
        if (ACC) {
          const uint4 o = *(const uint4*)d;
          v.x = cvt_pk_bf16(bf2f(v.x & 0xffff) + bf2f(o.x & 0xffff), bf2f(v.x >> 16) + bf2f(o.x >> 16));
          v.y = cvt_pk_bf16(bf2f(v.y & 0xffff) + bf2f(o.y & 0xffff), bf2f(v.y >> 16) + bf2f(o.y >> 16));
          v.z = cvt_pk_bf16(bf2f(v.z & 0xffff) + bf2f(o.z & 0xffff), bf2f(v.z >> 16) + bf2f(o.z >> 16));
          v.w = cvt_pk_bf16(bf2f(v.w & 0xffff) + bf2f(o.w & 0xffff), bf2f(v.w >> 16) + bf2f(o.w >> 16));
        }
        *(uint4*)d = v;
      }
    }
  }
}

// Shape specialisation.  The three big high-resolution depthwise layers (and 144 @ 56) spend a third of their instruction stream on index
// arithmetic whose operands never change: with the map size and the channel count known at compile time the bounds, pitches and tile counts
// fold into immediates (strength-reduced multiplies, no 64-bit pitch products, constant carries in the tile iterator).  DwAny = run-time shape.
struct DwAny { static constexpr int H = 0, W = 0, C = 0; };
template <int H_, int W_, int C_> struct DwShape { static constexpr int H = H_, W = W_, C = C_; };
template <typename SH, typename G, bool DGRAD>
__device__ __forceinline__ Dw3P dw_fix(const Dw3P& pin) {
  Dw3P q = pin;
  if constexpr (SH::H > 0) {
    constexpr int PAD = (G::K - 1) / 2, HO = (SH::H + 2 * PAD - G::K) / G::S + 1, WO = (SH::W + 2 * PAD - G::K) / G::S + 1;
    constexpr int DH_ = DGRAD ? SH::H : HO, DW_ = DGRAD ? SH::W : WO;
    q.h = SH::H; q.w = SH::W; q.c = SH::C; q.cpad = (SH::C + 15) / 16 * 16; q.pad = PAD; q.ho = HO; q.wo = WO;
    q.tiles_x = (DW_ + G::SUBW - 1) / G::SUBW; q.tiles_y = (DH_ + TH - 1) / TH; q.ncb = (SH::C + G::CBW - 1) / G::CBW;
    q.nunits = (int64_t)q.n * q.tiles_x * q.tiles_y; q.ntiles = (q.nunits + G::NSUB - 1) / G::NSUB;
  }
  return q;
}

// LDS bytes of the auxiliary tile per conv mode: int8 out tile (emit), bf16 gout / dc tile (backward).  The statistics pass has no such
// tile but keeps the allocation: the residency it would gain (more workgroups = more atomics in front of the finalize) measured slower.
template <typename G>
__host__ __device__ constexpr int dw_aux_bytes(int mode) { return mode == D_EMIT ? G::AUX_BYTES / 2 : G::AUX_BYTES; }

// lane geometry shared by the three kernels
template <typename G>
struct DwLane {
  int lane, wv, wy, lc, pc, sb, colo, xcol0;     // channel lane, patch column index, sub-tile, out col in tile, input col of the patch
  __device__ __forceinline__ DwLane(int tid) {
    lane = tid & 63; wv = __builtin_amdgcn_readfirstlane(tid >> 6); wy = wv >> 1;
    const int wx = wv & 1;
    lc = lane & (G::CBW - 1);
    pc = (G::CBW == 64) ? wx : wx * 2 + (lane >> 5);
    colo = pc * RW; sb = colo / G::SUBW;
    xcol0 = sb * G::IWS + (colo - sb * G::SUBW) * G::S;
  }
};

template <typename G, int MODE_, typename SH = DwAny>
__global__ __launch_bounds__(256, 2) void k_dw3(const Dw3P pin) {
  const Dw3P p = dw_fix<SH, G, false>(pin);
  constexpr int MODE = (MODE_ == D_BDCW) ? D_BDC : MODE_;
  constexpr bool WG = (MODE_ == D_BDCW);
  constexpr int K = G::K, S = G::S, CBW = G::CBW, SUBW = G::SUBW, NSUB = G::NSUB, TWT = G::TWT, IWT = G::IWT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int nb = p.nb;                                   // LDS tile buffers: 2 = the next tile's copy runs under this tile's arithmetic
  constexpr int AUXB = dw_aux_bytes<G>(MODE_);
  uint8_t* const tin0 = smem;
  uint8_t* const aux0 = smem + nb * G::IN_BYTES;        // gout / dc bf16 tile (16 KB) or int8 out tile (8 KB)
  double* red_d = (double*)(smem + nb * (G::IN_BYTES + AUXB));   // [4][64][2]
  float* red_f = (float*)(red_d + 4 * 64 * 2);                     // [4][64][2]

  const int tid = threadIdx.x;
  const DwLane<G> L(tid);
  const int lane = L.lane, wv = L.wv, wy = L.wy;
  const DwMap bm = dw_block_map(p); const int cb = bm.cb;
  const int ch = cb * CBW + L.lc;
  const bool chok = ch < p.c;
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]);
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;

  // Integer conv core: the lane's 8-pixel runs stay packed (4 offset-binary bytes per dword), every output takes its 4-byte
  // window with one v_alignbyte and one (k = 3) or two (k = 5) v_dot4_i32_i8 against the packed taps of a kernel row.
  // acc starts at (128 - zp) * sum(w), so acc = sum w * (q - zp) exactly (zero padding = zero-point fill).
  constexpr int NPK = (K == 3) ? 1 : 2;
  int wpk[K][NPK];
  int8_t taps[K * K];
  load_taps_i8<K * K>(p.wq, p.cpad, ch, chok, taps);
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const uint32_t b = (uint32_t)(uint8_t)taps[ky * K + kx];
      if (kx < 4) lo |= b << (8 * kx); else hi |= b;
    }
    wpk[ky][0] = (int)lo; if (NPK > 1) wpk[ky][NPK - 1] = (int)hi;
  }
  const int acc0 = chok ? (128 - zp) * p.wsum[ch] : 0;
  float qcap = 255.0f; bool lowq = false;        // 7-bit activations (reduce_range): the u8 conversion saturates at 255 only
  float cA = 0, cB = 0, cMR = 0, cR = 0, cK1 = 0, cE = 0, cF = 0, y_inv = 1.0f, y_zpf = 0.0f, t_lo = 0.0f, t_hi = 0.0f;
  int emit_add = 0; float emit_b = 0.0f;
  if (MODE != D_STATS) {
    y_inv = 1.0f / p.qy[FROST_Q_SCALE]; const int zpy = __float_as_int(p.qy[FROST_Q_ZP]); y_zpf = (float)zpy;
    if (chok) {
      cA = p.coef[FROST_COEF_A * p.cpad + ch]; cB = p.coef[FROST_COEF_B * p.cpad + ch];
      if (MODE != D_EMIT) { const float m = p.coef[FROST_COEF_M * p.cpad + ch]; cR = p.coef[FROST_COEF_R * p.cpad + ch]; cMR = -m * cR;
        if (MODE == D_BDC) {     // dc = fma(gy, K1, fma(acc, E, F)): same folding as the pointwise backward
          cK1 = p.coef[FROST_COEF_K1 * p.cpad + ch];
          cE = -cK1 * (s12_sum(p.coef, p.cpad, 1, ch) * p.inv_count) * cR;
          cF = -cK1 * (s12_sum(p.coef, p.cpad, 0, ch) * p.inv_count) - cE * m;
        }
      }
    }
    if (MODE == D_EMIT && p.cvt) y_inv = 1.0f;
    // converted inference: y = float(acc + bias_q) * requant scale (row B carries the int32 bias bits); training / eval: y = fma(A, acc, B)
    if (MODE == D_EMIT) { emit_add = (p.cvt == 1) ? __float_as_int(cB) : 0; emit_b = (p.cvt == 1) ? 0.0f : cB; }
    if (MODE == D_EMIT) { qcap = (float)q_hi(p.qy); lowq = qcap < 255.0f; }        // row A already is the requantisation scale s_x*s_w/s_y
    if (MODE == D_BRED || MODE == D_BDC) {   // STE pass window in t = y/scale: t_lo < t <= t_hi (see frost_pw.hip)
      const int qhi = q_hi(p.qy);
      const float hi0 = (float)qhi + 0.5f - (float)zpy;
      t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
      if (!p.relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
    }
  }
  const float relu_floor = p.relu ? 0.0f : -INFINITY;
  uint32_t rng = sr_seed(blockIdx.x, threadIdx.x);
  const bool sr_on = p.sr != 0;
  double st1 = 0.0, st2 = 0.0; float smn = INFINITY, smx = -INFINITY, r1 = 0.0f, r2 = 0.0f;
  DwPlan<G::IH, G::IWS, NSUB, CBW, 1, WG> plx; plx.init(tid, cb, p.w, p.c);          // the fused weight-gradient variant has no registers to spare: lean plans
  DwPlan<TH, SUBW, NSUB, CBW, 2, WG> plg; if (MODE == D_BRED || MODE == D_BDC) plg.init(tid, cb, p.wo, p.c);
  float wacc[WG ? K * K : 1]; float sdc = 0.0f;      // fused weight gradient: sum dc * q (unsigned index) per tap, and sum dc
#pragma unroll
  for (int t = 0; t < (WG ? K * K : 1); ++t) wacc[t] = 0.0f;
  auto stage = [&](const DwSub<NSUB, SUBW>& s_, int b) __attribute__((always_inline)) {
    stage_tile<G::IH, G::IWS, NSUB, SUBW, CBW, 1>(p.x, tin0 + b * G::IN_BYTES, tid, plx, s_, S, 1, -p.pad, p.h, p.w, p.c, zfill);
    if (MODE == D_BRED || MODE == D_BDC) stage_tile<TH, SUBW, NSUB, SUBW, CBW, 2>(p.gout, aux0 + b * AUXB, tid, plg, s_, 1, 1, 0, p.ho, p.wo, p.c, 0u);
  };
  DwSub<NSUB, SUBW> su, sun;
  int buf = 0;
  const DwStep dstep = dw_step<NSUB>(p, bm.step);
  DwIt<NSUB> it;
  if (bm.t0 < bm.t1) { dw_it_init<NSUB>(p, bm.t0, it); dw_fill<NSUB, SUBW>(p, it, su); stage(su, 0); }

  for (int64_t tile = bm.t0; tile < bm.t1; tile += bm.step) {
    dw_wait_barrier();                // this tile has landed (every wave waited for its own copies); the other buffer is free
    const bool more = (tile + bm.step) < bm.t1;
    if (more) { dw_it_next<NSUB>(p, dstep, it); dw_fill<NSUB, SUBW>(p, it, sun); if (nb == 2) stage(sun, buf ^ 1); }
    uint8_t* const tin = tin0 + buf * G::IN_BYTES;
    uint8_t* const aux = aux0 + buf * AUXB;

    int acc[RH][RW];
#pragma unroll
    for (int o = 0; o < RH; ++o)
#pragma unroll
      for (int r = 0; r < RW; ++r) acc[o][r] = acc0;
    constexpr int NWIN = (K == 3) ? RW : RW + 4 / S;            // window i starts at byte i*S of the lane's row
#pragma unroll
    for (int jr = 0; jr < G::NROW; ++jr) {
      int d[G::NBLK * 2 + 1];
      const uint8_t* rowp = tin + ((wy * RH * S + jr) * IWT) * CBW;
#pragma unroll
      for (int b = 0; b < G::NBLK; ++b) { const v2i raw = tr8_raw<CBW>(rowp, L.xcol0 + b * 8, lane); d[2 * b] = raw[0]; d[2 * b + 1] = raw[1]; }
      d[G::NBLK * 2] = 0;
      int win[NWIN];
#pragma unroll
      for (int i = 0; i < NWIN; ++i) {
        const int off = i * S;
        win[i] = (off % 4 == 0) ? d[off / 4] : __builtin_amdgcn_alignbyte(d[off / 4 + 1], d[off / 4], off % 4);
      }
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        if ((jr - ky) >= 0 && ((jr - ky) % S) == 0 && (jr - ky) / S < RH) {
          const int o = (jr - ky) / S;
#pragma unroll
          for (int r = 0; r < RW; ++r) {
            acc[o][r] = __builtin_amdgcn_sdot4(win[r], wpk[ky][0], acc[o][r], false);
            if (K == 5) acc[o][r] = __builtin_amdgcn_sdot4(win[r + 4 / S], wpk[ky][NPK - 1], acc[o][r], false);
          }
        }
      }
    }

    // ---------------------------------------------------------------- epilogue (lane-local, one channel)
    const int oyb = DW_SEL(su.r0, L.sb) + wy * RH, oxb = DW_SEL(su.c0, L.sb) + (L.colo - L.sb * SUBW);
    const bool subok = chok && DW_SEL(su.ok, L.sb);
    int t1 = 0; double t2 = 0.0; int tmn = INT32_MAX, tmx = INT32_MIN;
    const bool fbq = (MODE == D_EMIT) && p.cvt == 2;
    if (fbq) {     // converted inference, FBGEMM form: q = cvtps2dq((float(acc) + b / (s_x s_w[c])) * (s_x s_w[c] / s_y)) + zp (rows B / A; oracle.fbgemm_conv)
#pragma unroll
      for (int o = 0; o < RH; ++o)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int lp = (wy * RH + o) * TWT + L.colo + r;
          float qv = rintf(fmaxf(((float)acc[o][r] + cB) * cA, relu_floor)) + y_zpf;
          if (lowq) qv = fminf(qv, qcap);
          aux[lp * CBW + L.lc] = (uint8_t)((__builtin_amdgcn_cvt_pk_u8_f32(qv, 0, 0u) ^ 0x80u) & 255u);
        }
    }
    if (!fbq)
#pragma unroll
    for (int o = 0; o < RH; ++o) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const bool valid = subok && (oyb + o) < p.ho && (oxb + r) < p.wo;
        const float v = (float)acc[o][r];
        const int lp = (wy * RH + o) * TWT + L.colo + r;          // pixel index inside the tile
        if (MODE == D_STATS) {     // exact int sum and min/max; sum of squares in double (v*v needs 40 bits)
          const int vi = acc[o][r];
          if (valid) { t1 += vi; t2 = fma((double)v, (double)v, t2); tmn = min(tmn, vi); tmx = max(tmx, vi); }
        } else if (MODE == D_EMIT) {
          // q = clamp(rint(relu(y)/s) + zp, 0, 255): v_cvt_pk_u8_f32 saturates at both ends while converting
          const float yv = fmaf(cA, (float)(acc[o][r] + emit_add), emit_b);       // one form for both emit flavours (see emit_add): no per-output branch
          float qv = rintf(fmaxf(yv, relu_floor) * y_inv) + y_zpf;
          if (lowq) qv = fminf(qv, qcap);
          aux[lp * CBW + L.lc] = (uint8_t)((__builtin_amdgcn_cvt_pk_u8_f32(qv, 0, 0u) ^ 0x80u) & 255u);
        } else {
          const float gq = bf2f(*(const uint16_t*)(aux + (lp * CBW + L.lc) * 2));
          const float tq = fmaf(cA, v, cB) * y_inv;
          const float gy = (valid && tq > t_lo && tq <= t_hi) ? gq : 0.0f;
          if (MODE == D_BRED) { r1 += gy; r2 = fmaf(gy, fmaf(v, cR, cMR), r2); }
          else {
            const float dcf = fmaf(gy, cK1, fmaf(v, cE, cF));
            // stochastic (default) or nearest-even rounding from ONE instruction stream: the added 16 bits are the draw or 0x7fff + lsb
            const uint32_t db = __float_as_uint(dcf), dr = sr_next16(rng);
            const uint32_t hb = (db + (sr_on ? dr : 0x7fffu + ((db >> 16) & 1u))) >> 16;
            *(uint16_t*)(aux + (lp * CBW + L.lc) * 2) = (uint16_t)hb;
            if (WG) acc[o][r] = valid ? (int)(hb << 16) : 0;          // the bf16-rounded dc (float bits) replaces the dead accumulator
          }
        }
      }
    }
    if (WG) {     // second sweep over the halo rows still in LDS: wacc[ky][kx] += dc[o][r] * q[o*S+ky][r*S+kx]
#pragma unroll
      for (int o = 0; o < RH; ++o)
#pragma unroll
        for (int r = 0; r < RW; ++r) sdc += __int_as_float(acc[o][r]);
#pragma unroll
      for (int jr = 0; jr < G::NROW; ++jr) {
        float xr[G::NBLK * 8];
        const uint8_t* rowp = tin + ((wy * RH * S + jr) * IWT) * CBW;
#pragma unroll
        for (int b = 0; b < G::NBLK; ++b) tr8_run<CBW>(rowp, L.xcol0 + b * 8, lane, xr + b * 8);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          if ((jr - ky) >= 0 && ((jr - ky) % S) == 0 && (jr - ky) / S < RH) {
            const int o = (jr - ky) / S;
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
              for (int r = 0; r < RW; ++r) wacc[ky * K + kx] = fmaf(__int_as_float(acc[o][r]), xr[r * S + kx], wacc[ky * K + kx]);
          }
        }
      }
    }
    if (MODE == D_STATS) { st1 += (double)t1; st2 += t2; smn = fminf(smn, (float)tmn); smx = fmaxf(smx, (float)tmx); }
    if (MODE == D_EMIT || MODE == D_BDC) {
      __syncthreads();
      if (MODE == D_EMIT) copy_out_tile<1, NSUB, SUBW, CBW, false>(aux, (uint8_t*)p.y, tid, su, cb, p.ho, p.wo, p.c);
      else copy_out_tile<2, NSUB, SUBW, CBW, false>(aux, (uint8_t*)p.dc, tid, su, cb, p.ho, p.wo, p.c);
    }
    if (more) {
      if (nb == 1) { __syncthreads(); stage(sun, 0); }     // single buffer: every wave is done with the tile before it is overwritten
      else buf ^= 1;
      su = sun;
    }
  }

  if (WG) {       // dW[c][tap] += s_x * (sum dc*q - zp * sum dc): the 4 waves' (and, for 32-channel blocks, both halves') partials through LDS
    __syncthreads();
    float* red = (float*)smem;                       // [4][K*K][64]
    const float zpf = (float)zp;
#pragma unroll
    for (int t = 0; t < K * K; ++t) red[(wv * K * K + t) * 64 + lane] = wacc[t] - zpf * sdc;
    __syncthreads();
    const float sx = p.qx[FROST_Q_SCALE];
    for (int i = tid; i < K * K * CBW; i += 256) {
      const int t = i / CBW, l2 = i % CBW; const int c2 = cb * CBW + l2;
      float sum = 0.0f;
      for (int w2 = 0; w2 < 4; ++w2)
        for (int l3 = l2; l3 < 64; l3 += CBW) sum += red[(w2 * K * K + t) * 64 + l3];
      if (c2 < p.c) atomicAdd(dwq_dst(p.dwq, (int64_t)p.c * K * K) + (int64_t)c2 * K * K + t, sum * sx);
    }
  }
  if (MODE == D_STATS || MODE == D_BRED) {      // sum the waves' lane-local partials, one global atomic set per channel
    __syncthreads();
    if (MODE == D_STATS) { red_d[(wv * 64 + lane) * 2] = st1; red_d[(wv * 64 + lane) * 2 + 1] = st2; red_f[(wv * 64 + lane) * 2] = smn; red_f[(wv * 64 + lane) * 2 + 1] = smx; }
    else { red_f[(wv * 64 + lane) * 2] = r1; red_f[(wv * 64 + lane) * 2 + 1] = r2; }
    __syncthreads();
    if (tid < CBW && (cb * CBW + tid) < p.c) {
      const int ch2 = cb * CBW + tid;
      if (MODE == D_STATS) {
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
    if (MODE == D_STATS && p.fin_on) {       // last workgroup done -> conv finalize in this launch (see frost_common.h)
      int* sflag = (int*)smem;
      if (last_block_done2(p.fin.counter, gridDim.x, sflag)) {
        float* sh = (float*)(smem + 16);
        conv_finalize_dev(p.stats, (int64_t)p.n * p.ho * p.wo, p.c, p.cpad, p.qx, p.fin.qrec_w, p.fin.wscale, p.fin.gamma, p.fin.beta, p.fin.rmean, p.fin.rvar,
                          p.fin.nbt, p.fin.training, p.fin.relu, p.fin.observe, 1, p.fin.coef, p.fin.qrec_y, tid, 256, sh);
      }
    }
  }
}

// ---- wgrad: dwq[c][ky][kx] += s_x * sum dc * (q - zp); lane-local K*K sums across the whole persistent loop
template <typename G, typename SH = DwAny>
__global__ __launch_bounds__(256, 2) void k_dw3_wgrad(const Dw3P pin) {
  const Dw3P p = dw_fix<SH, G, false>(pin);
  constexpr int K = G::K, S = G::S, CBW = G::CBW, SUBW = G::SUBW, NSUB = G::NSUB, TWT = G::TWT, IWT = G::IWT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int nb = p.nb;
  uint8_t* const tin0 = smem; uint8_t* const aux0 = smem + nb * G::IN_BYTES;
  const int tid = threadIdx.x;
  const DwLane<G> L(tid);
  const int lane = L.lane, wv = L.wv, wy = L.wy;
  const DwMap bm = dw_block_map(p); const int cb = bm.cb;
  const int zp = __float_as_int(p.qx[FROST_Q_ZP]); const float zpf = (float)zp;
  const uint32_t zfill = (uint32_t)((zp - 128) & 255) * 0x01010101u;
  float acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = 0.0f;
  constexpr bool LEAN = (K == 5 && S == 2);
  DwPlan<G::IH, G::IWS, NSUB, CBW, 1, LEAN> plx; DwPlan<TH, SUBW, NSUB, CBW, 2, LEAN> plg;
  plx.init(tid, cb, p.w, p.c); plg.init(tid, cb, p.wo, p.c);
  auto stage = [&](const DwSub<NSUB, SUBW>& s_, int b) __attribute__((always_inline)) {
    stage_tile<G::IH, G::IWS, NSUB, SUBW, CBW, 1>(p.x, tin0 + b * G::IN_BYTES, tid, plx, s_, S, 1, -p.pad, p.h, p.w, p.c, zfill);
    stage_tile<TH, SUBW, NSUB, SUBW, CBW, 2>(p.dc, aux0 + b * G::AUX_BYTES, tid, plg, s_, 1, 1, 0, p.ho, p.wo, p.c, 0u);     // rows/cols outside the map -> 0
  };
  DwSub<NSUB, SUBW> su, sun;
  int buf = 0;
  const DwStep dstep = dw_step<NSUB>(p, bm.step);
  DwIt<NSUB> it;
  if (bm.t0 < bm.t1) { dw_it_init<NSUB>(p, bm.t0, it); dw_fill<NSUB, SUBW>(p, it, su); stage(su, 0); }
  for (int64_t tile = bm.t0; tile < bm.t1; tile += bm.step) {
    dw_wait_barrier();
    const bool more = (tile + bm.step) < bm.t1;
    if (more) { dw_it_next<NSUB>(p, dstep, it); dw_fill<NSUB, SUBW>(p, it, sun); if (nb == 2) stage(sun, buf ^ 1); }
    const uint8_t* const tin = tin0 + buf * G::IN_BYTES;
    const uint8_t* const aux = aux0 + buf * G::AUX_BYTES;
    float g[RH][RW];
#pragma unroll
    for (int o = 0; o < RH; ++o) {
      const uint8_t* rowp = aux + ((wy * RH + o) * TWT) * CBW * 2;
      tr16_run<CBW>(rowp, L.colo, lane, &g[o][0]); tr16_run<CBW>(rowp, L.colo + 4, lane, &g[o][4]);
    }
#pragma unroll
    for (int jr = 0; jr < G::NROW; ++jr) {
      float xr[G::NBLK * 8];
      const uint8_t* rowp = tin + ((wy * RH * S + jr) * IWT) * CBW;
#pragma unroll
      for (int b = 0; b < G::NBLK; ++b) tr8_run<CBW>(rowp, L.xcol0 + b * 8, lane, xr + b * 8);
#pragma unroll
      for (int i = 0; i < G::NBLK * 8; ++i) xr[i] -= zpf;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        if ((jr - ky) >= 0 && ((jr - ky) % S) == 0 && (jr - ky) / S < RH) {
          const int o = (jr - ky) / S;
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[ky * K + kx] = fmaf(g[o][r], xr[r * S + kx], acc[ky * K + kx]);
        }
      }
    }
    if (more) {
      if (nb == 1) { __syncthreads(); stage(sun, 0); }
      else buf ^= 1;
      su = sun;
    }
  }
  __syncthreads();
  float* red = (float*)smem;                       // [4][K*K][64]
#pragma unroll
  for (int t = 0; t < K * K; ++t) red[(wv * K * K + t) * 64 + lane] = acc[t];
  __syncthreads();
  const float sx = p.qx[FROST_Q_SCALE];
  for (int i = tid; i < K * K * CBW; i += 256) {
    const int t = i / CBW, l2 = i % CBW; const int c2 = cb * CBW + l2;
    float sum = 0.0f;
    for (int w2 = 0; w2 < 4; ++w2)
      for (int l3 = l2; l3 < 64; l3 += CBW) sum += red[(w2 * K * K + t) * 64 + l3];
    if (c2 < p.c) atomicAdd(dwq_dst(p.dwq, (int64_t)p.c * K * K) + (int64_t)c2 * K * K + t, sum * sx);
  }
}

// ---- dgrad: dx[iy][ix][c] (+)= s_w * sum_{ky,kx} dc[(iy+pad-ky)/s][(ix+pad-kx)/s][c] * wq[ky][kx][c]   (tiles over dx)
template <typename G, typename SH = DwAny>
__global__ __launch_bounds__(256, 2) void k_dw3_dgrad(const Dw3P pin) {
  const Dw3P p = dw_fix<SH, G, true>(pin);
  constexpr int K = G::K, S = G::S, CBW = G::CBW, SUBW = G::SUBW, NSUB = G::NSUB, TWT = G::TWT;
  constexpr int PAD = (K - 1) / 2;
  constexpr int LO = fdiv3(-PAD, S);
  constexpr int DH = (TH - 1 + PAD) / S - LO + 1, DWS = (SUBW - 1 + PAD) / S - LO + 1, DWT = NSUB * DWS;
  constexpr int D_BYTES = ((DH * DWT * CBW * 2 + 255) / 256) * 256 + 512;
  constexpr int NJR = (S == 1) ? (RH + K - 1) : ((RH - 1 + PAD) / 2 - LO + 1);
  constexpr int NR4 = (S == 1) ? (RW + K - 1 + 3) / 4 : 2;               // 4-pixel transpose reads per dc row
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int nb = p.nb;
  uint8_t* const tdc0 = smem; uint8_t* const tout = smem + nb * D_BYTES;     // dx bf16 out tile [TH*TWT][CBW]
  const int tid = threadIdx.x;
  const DwLane<G> L(tid);
  const int lane = L.lane, wy = L.wy;
  const DwMap bm = dw_block_map(p); const int cb = bm.cb;
  const int ch = cb * CBW + L.lc; const bool chok = ch < p.c;
  const float sw = (p.wscale && chok) ? p.wscale[ch] : p.qw[FROST_Q_SCALE];     // lane = channel: the per-channel weight scale is a per-lane scalar
  float wf[K * K];
  {
    int8_t taps[K * K];
    load_taps_i8<K * K>(p.wq, p.cpad, ch, chok, taps);
#pragma unroll
    for (int t = 0; t < K * K; ++t) wf[t] = (float)taps[t];
  }
  const int rb = (S == 1) ? wy * RH : wy * (RH / 2);
  const int cin_sub = L.colo - L.sb * SUBW;                                // patch column inside its sub-tile (multiple of 8)
  const int cbase = L.sb * DWS + ((S == 1) ? cin_sub : cin_sub / 2);
  DwPlan<DH, DWS, NSUB, CBW, 2> pld; pld.init(tid, cb, p.wo, p.c);
  DwSub<NSUB, SUBW> su, sun;                                             // units over the dx (input) domain
  int buf = 0;
  const DwStep dstep = dw_step<NSUB>(p, bm.step);
  DwIt<NSUB> it;
  if (bm.t0 < bm.t1) { dw_it_init<NSUB>(p, bm.t0, it); dw_fill<NSUB, SUBW>(p, it, su); stage_tile<DH, DWS, NSUB, SUBW, CBW, 2>(p.dc, tdc0, tid, pld, su, 1, S, LO, p.ho, p.wo, p.c, 0u); }
  for (int64_t tile = bm.t0; tile < bm.t1; tile += bm.step) {
    dw_wait_barrier();               // dc tile landed; the previous tile's copy-out has read tout
    const bool more = (tile + bm.step) < bm.t1;
    if (more) { dw_it_next<NSUB>(p, dstep, it); dw_fill<NSUB, SUBW>(p, it, sun); if (nb == 2) stage_tile<DH, DWS, NSUB, SUBW, CBW, 2>(p.dc, tdc0 + (buf ^ 1) * D_BYTES, tid, pld, sun, 1, S, LO, p.ho, p.wo, p.c, 0u); }
    const uint8_t* const tdc = tdc0 + buf * D_BYTES;
    float acc[RH][RW];
#pragma unroll
    for (int o = 0; o < RH; ++o)
#pragma unroll
      for (int r = 0; r < RW; ++r) acc[o][r] = 0.0f;
#pragma unroll
    for (int jr = 0; jr < NJR; ++jr) {
      float dcr[NR4 * 4];
      const uint8_t* rowp = tdc + ((rb + jr) * DWT) * CBW * 2;
#pragma unroll
      for (int b = 0; b < NR4; ++b) tr16_run<CBW>(rowp, cbase + b * 4, lane, dcr + b * 4);
#pragma unroll
      for (int o = 0; o < RH; ++o)
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const int ty = o + PAD - ky;
          if ((S == 1 || ((ty % 2 + 2) % 2) == 0) && (fdiv3(ty, S) - LO) == jr) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
              for (int kx = 0; kx < K; ++kx) {
                const int tx = r + PAD - kx;
                if (S == 1 || ((tx % 2 + 2) % 2) == 0) acc[o][r] = fmaf(dcr[fdiv3(tx, S) - LO], wf[ky * K + kx], acc[o][r]);
              }
          }
        }
    }
#pragma unroll
    for (int o = 0; o < RH; ++o)
#pragma unroll
      for (int r = 0; r < RW; ++r) *(uint16_t*)(tout + (((wy * RH + o) * TWT + L.colo + r) * CBW + L.lc) * 2) = (uint16_t)cvt_pk_bf16(acc[o][r] * sw, 0.0f);
    __syncthreads();
    if (p.accumulate) copy_out_tile<2, NSUB, SUBW, CBW, true>(tout, (uint8_t*)p.dx, tid, su, cb, p.h, p.w, p.c);
    else copy_out_tile<2, NSUB, SUBW, CBW, false>(tout, (uint8_t*)p.dx, tid, su, cb, p.h, p.w, p.c);
    if (more) {
      if (nb == 1) { __syncthreads(); stage_tile<DH, DWS, NSUB, SUBW, CBW, 2>(p.dc, tdc0, tid, pld, sun, 1, S, LO, p.ho, p.wo, p.c, 0u); }
      else buf ^= 1;
      su = sun;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static void fill3(Dw3P& p, const int8_t* x, const float* qx, const int8_t* wq, const int32_t* wsum, int n, int h, int w, int c, int k, int stride) {
  p.x = x; p.qx = qx; p.wq = wq; p.wsum = wsum; p.n = n; p.h = h; p.w = w; p.c = c; p.cpad = round_up(c, 16);
  p.pad = (k - 1) / 2; p.ho = (h + 2 * p.pad - k) / stride + 1; p.wo = (w + 2 * p.pad - k) / stride + 1;
  p.inv_count = 1.0f / (float)((int64_t)n * p.ho * p.wo);
}
// geometry choice: 32-channel blocks when they waste fewer lanes than 64-channel blocks; 8-column sub-tiles (two images per
// tile) for maps no wider than 8.  dom_h/dom_w: the domain the tiles run over (outputs; dx for the dgrad)
enum { GEO_A = 0, GEO_B = 1, GEO_C = 2 };     // (64,16)  (64,8)  (32,32)
static int pick_geo(int c, int dom_w, int k, int stride, bool dgrad) {
  if (dom_w <= 8 && k == 5) return GEO_B;                                   // instantiated for k = 5 only (the 7x7 layers)
  const bool c_inst = (k == 3) || (k == 5 && stride == 2);                  // GEO_C instantiations
  if (dgrad && k == 5) return GEO_A;                                       // measured: the k = 5 dgrad prefers 64-channel blocks
  if (c_inst && round_up(c, 32) < round_up(c, 64) && dom_w > 16) return GEO_C;
  return GEO_A;
}
template <typename G>
static void set_tiles(Dw3P& p, int dom_h, int dom_w) {
  p.tiles_x = (dom_w + G::SUBW - 1) / G::SUBW; p.tiles_y = (dom_h + TH - 1) / TH; p.ncb = (p.c + G::CBW - 1) / G::CBW;
  p.nunits = (int64_t)p.n * p.tiles_x * p.tiles_y; p.ntiles = (p.nunits + G::NSUB - 1) / G::NSUB;
}
template <typename KF>
static int launch3(KF kern, Dw3P& p, size_t lds, const char* what, hipStream_t s) {
  static std::map<const void*, int> occ_cache;     // keyed by kernel: all instantiations share this launcher's type
  const void* key = (const void*)kern;
  auto it = occ_cache.find(key);
  int occ = 1;
  if (it == occ_cache.end()) {
    hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, key, 256, lds) != hipSuccess || occ < 1) occ = 1;
    occ_cache[key] = occ;
  } else occ = it->second;
  if (occ > 8) occ = 8;
  FROST_REQUIRE(p.nunits < ((int64_t)1 << 30), "dw: too many tiles for the 32-bit tile iterator");
  int64_t want = (256 * occ) / p.ncb; if (want < 1) want = 1;
  p.ngroups = (int)(p.ntiles < want ? p.ntiles : want);
  static const int xon = getenv("FROST_DW_XCD") ? atoi(getenv("FROST_DW_XCD")) : 1;
  p.xmap = 0;
  if (xon && p.ngroups >= 8 && p.ntiles >= 64) { p.ngroups &= ~7; p.xmap = 1; }
  hipLaunchKernelGGL(kern, dim3(p.ncb * p.ngroups), dim3(256), lds, s, p);
  return frost_check_launch(what);
}
// Input buffers per workgroup.  Measured (tests/devtools/pw_micro.py, every FrostNet-Large depthwise shape): ONE buffer and the residency it
// leaves (3-6 workgroups per CU cover each other's copies) beats two buffers (copy of tile t+1 under the arithmetic of tile t) by 8-30 %;
// FROST_DW_NB=2 keeps the double-buffered loop reachable for experiments.
static int dw_nbuf(size_t per_buffer, size_t fixed) {
  static const int force = getenv("FROST_DW_NB") ? atoi(getenv("FROST_DW_NB")) : 1;
  return (force == 2 && 2 * per_buffer + fixed <= 158 * 1024) ? 2 : 1;
}
template <typename G, int MODE, typename SH = DwAny>
static int launch_fwd(Dw3P& p, hipStream_t s) {
  set_tiles<G>(p, p.ho, p.wo);
  const size_t per = (size_t)G::IN_BYTES + dw_aux_bytes<G>(MODE), fixed = 4 * 64 * 2 * 8 + 4 * 64 * 2 * 4, red = (size_t)4 * G::K * G::K * 64 * 4;
  p.nb = dw_nbuf(per, fixed);
  const size_t lds = p.nb * per + fixed;
  return launch3(k_dw3<G, MODE, SH>, p, lds > red ? lds : red, "dw", s);
}
template <typename G, typename SH = DwAny>
static int launch_wgrad(Dw3P& p, hipStream_t s) {
  set_tiles<G>(p, p.ho, p.wo);
  const size_t per = (size_t)G::IN_BYTES + G::AUX_BYTES, red = (size_t)4 * G::K * G::K * 64 * 4;
  p.nb = dw_nbuf(per, 0);
  const size_t lds = p.nb * per;
  return launch3(k_dw3_wgrad<G, SH>, p, lds > red ? lds : red, "dw_wgrad", s);
}
template <typename G, typename SH = DwAny>
static int launch_dgrad(Dw3P& p, hipStream_t s) {
  constexpr int PAD = (G::K - 1) / 2; constexpr int LO = fdiv3(-PAD, G::S);
  constexpr int DH = (TH - 1 + PAD) / G::S - LO + 1, DWS = (G::SUBW - 1 + PAD) / G::S - LO + 1;
  set_tiles<G>(p, p.h, p.w);
  const size_t per = (size_t)((DH * G::NSUB * DWS * G::CBW * 2 + 255) / 256) * 256 + 512;
  p.nb = dw_nbuf(per, G::AUX_BYTES);
  return launch3(k_dw3_dgrad<G, SH>, p, p.nb * per + G::AUX_BYTES, "dw_dgrad", s);
}
// dc pass + weight gradient: fused where the combined register state fits (measured: no spills), else two launches
template <typename G, typename SH = DwAny>
static int launch_bdc_wgrad(Dw3P& p, hipStream_t s) {
  constexpr bool FUSE = (G::K == 3 && G::S == 1) || (G::K == 5 && G::S == 1 && G::SUBW == 8);
  if constexpr (FUSE) return launch_fwd<G, D_BDCW, SH>(p, s);
  else { const int rc = launch_fwd<G, D_BDC, SH>(p, s); return rc ? rc : launch_wgrad<G, SH>(p, s); }
}
// one switch over (k, stride, geometry); OP: 0..3 = conv modes, 4 = wgrad, 5 = dgrad, 6 = dc pass with the fused weight gradient
#define DW_CASE(KK, SS, CB_, SW_)                                                                             \
  { typedef DwGeo<KK, SS, CB_, SW_> G_;                                                                       \
    switch (op) { case 0: return launch_fwd<G_, D_STATS>(p, s); case 1: return launch_fwd<G_, D_EMIT>(p, s);  \
                  case 2: return launch_fwd<G_, D_BRED>(p, s); case 3: return launch_fwd<G_, D_BDC>(p, s);    \
                  case 4: return launch_wgrad<G_>(p, s); case 6: return launch_bdc_wgrad<G_>(p, s); default: return launch_dgrad<G_>(p, s); } }
// FW: the fused dc + weight-gradient kernel too (not for 32 @ 112: its specialised instance needs 21 more VGPRs and loses a wave: +10 %)
#define DW_SHAPE_CASE(KK, SS, CB_, SW_, H_, W_, C_, FW)                                                                           \
  if (spec_on && k == KK && stride == SS && p.h == H_ && p.w == W_ && p.c == C_) { typedef DwGeo<KK, SS, CB_, SW_> G_; typedef DwShape<H_, W_, C_> S_;   \
    typedef typename std::conditional<FW, S_, DwAny>::type SW2_;                                                                  \
    switch (op) { case 0: return launch_fwd<G_, D_STATS, S_>(p, s); case 1: return launch_fwd<G_, D_EMIT, S_>(p, s);                \
                  case 2: return launch_fwd<G_, D_BRED, S_>(p, s); case 3: return launch_fwd<G_, D_BDC, S_>(p, s);                  \
                  case 4: return launch_wgrad<G_, S_>(p, s); case 6: return launch_bdc_wgrad<G_, SW2_>(p, s); default: return launch_dgrad<G_, S_>(p, s); } }
static int dispatch3(Dw3P& p, int k, int stride, int geo, int op, hipStream_t s) {
  static const int spec_on = getenv("FROST_DW_SPEC") ? atoi(getenv("FROST_DW_SPEC")) : 1;
  if (geo == GEO_C) {       // FrostNet-Large @ 224: layer1.0 / 1.1 / 1.2 / 2.0 depthwise convs
    DW_SHAPE_CASE(3, 1, 32, 32, 112, 112, 32, false) DW_SHAPE_CASE(3, 2, 32, 32, 112, 112, 96, true) DW_SHAPE_CASE(3, 1, 32, 32, 56, 56, 72, true)
    DW_SHAPE_CASE(5, 2, 32, 32, 56, 56, 144, true)
  }
  if (geo == GEO_B) { if (k == 5 && stride == 1) DW_CASE(5, 1, 64, 8) if (k == 5 && stride == 2) DW_CASE(5, 2, 64, 8) }
  if (geo == GEO_C) { if (k == 3 && stride == 1) DW_CASE(3, 1, 32, 32) if (k == 3 && stride == 2) DW_CASE(3, 2, 32, 32) if (k == 5 && stride == 2) DW_CASE(5, 2, 32, 32) }
  if (k == 3 && stride == 1) DW_CASE(3, 1, 64, 16)
  if (k == 3 && stride == 2) DW_CASE(3, 2, 64, 16)
  if (k == 5 && stride == 1) DW_CASE(5, 1, 64, 16)
  if (k == 5 && stride == 2) DW_CASE(5, 2, 64, 16)
  frost_set_error("dw: unsupported kernel/stride (k in {3,5}, stride in {1,2})"); return 1;
}
static int geo_env(int geo) { static const int force = getenv("FROST_DW_GEO") ? atoi(getenv("FROST_DW_GEO")) : -1; return force == 0 ? GEO_A : geo; }

// the matrix-core formulation (frost_dwm.hip) takes the layers it has an instance for
int frost_dwm_ok(int k, int stride, int c);
int frost_dwm_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h, int w, int c, int k, int mode,
                  void* stats, const float* coef, const float* qrec_y, int relu, int8_t* y, const FrostFinDesc* fin, hipStream_t s);

// the strip-streaming forms of the statistics / emit / reduce passes (csrc/frost_dwb.hip) take the tiled layers they have an instance for
int frost_dws_ok(int h, int w, int c, int k, int stride, int mode);
int frost_dws_launch(int mode, const int8_t* x, const float* qx, const int8_t* wq, const int32_t* wsum, int n, int h, int w, int c, int k, int stride,
                     void* stats, float* coef, const float* qy, int relu, int8_t* y, const uint16_t* gout, const FrostFinDesc* fin, hipStream_t s);

extern "C" int frost_dw_conv_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n,
                                 int h, int w, int c, int k, int stride, int mode, void* stats, const float* coef,
                                 const float* qrec_y, int relu, int8_t* y, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  if (mode <= 1 && frost_dws_ok(h, w, c, k, stride, mode))
    return frost_dws_launch(mode, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride, stats, (float*)coef, qrec_y, relu, y, nullptr, nullptr, as_stream(stream));
  if (mode != 3 && frost_dwm_ok(k, stride, c)) return frost_dwm_fwd(x, qrec_x, wq_pack, wsum, n, h, w, c, k, mode, stats, coef, qrec_y, relu, y, nullptr, as_stream(stream));
  Dw3P p = {}; fill3(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.stats = (uint8_t*)stats; p.coef = (float*)coef; p.qy = qrec_y; p.relu = relu; p.y = y; p.cvt = (mode == 2) ? 1 : ((mode == 3) ? 2 : 0);
  return dispatch3(p, k, stride, geo_env(pick_geo(c, p.wo, k, stride, false)), mode == 0 ? 0 : 1, as_stream(stream));
}
extern "C" int frost_dw_conv_fwd_fin(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h, int w, int c, int k,
                                     int stride, void* stats, const FrostFinDesc* fin, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  FROST_REQUIRE(fin && fin->counter && fin->coef && fin->qrec_y, "dw_fwd_fin: incomplete finalize descriptor");
  if (frost_dws_ok(h, w, c, k, stride, 0))
    return frost_dws_launch(0, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride, stats, nullptr, nullptr, 0, nullptr, nullptr, fin, as_stream(stream));
  if (frost_dwm_ok(k, stride, c)) return frost_dwm_fwd(x, qrec_x, wq_pack, wsum, n, h, w, c, k, 0, stats, nullptr, nullptr, 0, nullptr, fin, as_stream(stream));
  Dw3P p = {}; fill3(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.stats = (uint8_t*)stats; p.coef = fin->coef; p.qy = fin->qrec_y; p.relu = fin->relu; p.fin = *fin; p.fin_on = 1;
  return dispatch3(p, k, stride, geo_env(pick_geo(c, p.wo, k, stride, false)), 0, as_stream(stream));
}
extern "C" int frost_dw_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                                 const float* qrec_w, int n, int h, int w, int c, int k, int stride, int pass, float* coef,
                                 const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  if (pass == 0 && frost_dws_ok(h, w, c, k, stride, 2))
    return frost_dws_launch(2, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride, nullptr, coef, qrec_y, relu, nullptr, gout, nullptr, as_stream(stream));
  Dw3P p = {}; fill3(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.coef = coef; p.qy = qrec_y; p.relu = relu; p.gout = gout; p.dc = dc; p.qw = qrec_w; p.sr = frost_sr_enabled();
  return dispatch3(p, k, stride, geo_env(pick_geo(c, p.wo, k, stride, false)), pass == 0 ? 2 : 3, as_stream(stream));
}
extern "C" int frost_dw_conv_bwd_dc_wgrad(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                                          const float* qrec_w, int n, int h, int w, int c, int k, int stride, float* coef,
                                          const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, float* dwq, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw: channels must be a multiple of 8");
  Dw3P p = {}; fill3(p, x, qrec_x, wq_pack, wsum, n, h, w, c, k, stride);
  p.coef = coef; p.qy = qrec_y; p.relu = relu; p.gout = gout; p.dc = dc; p.qw = qrec_w; p.dwq = dwq; p.sr = frost_sr_enabled();
  return dispatch3(p, k, stride, geo_env(pick_geo(c, p.wo, k, stride, false)), 6, as_stream(stream));
}
extern "C" int frost_dw_wgrad(const uint16_t* dc, const int8_t* x, const float* qrec_x, int n, int h, int w, int c, int k,
                              int stride, float* dwq, void* stream) {
  Dw3P p = {}; fill3(p, x, qrec_x, nullptr, nullptr, n, h, w, c, k, stride); p.dc = (uint16_t*)dc; p.dwq = dwq;
  return dispatch3(p, k, stride, geo_env(pick_geo(c, p.wo, k, stride, false)), 4, as_stream(stream));
}
extern "C" int frost_dw_dgrad(const uint16_t* dc, const int8_t* wq_pack, const float* qrec_w, int n, int h, int w, int c,
                              int k, int stride, uint16_t* dx, int accumulate, const float* wscale, void* stream) {
  FROST_REQUIRE(c % 8 == 0, "dw_dgrad: channels must be a multiple of 8");
  Dw3P p = {}; fill3(p, nullptr, nullptr, wq_pack, nullptr, n, h, w, c, k, stride);
  p.dc = (uint16_t*)dc; p.dx = dx; p.accumulate = accumulate; p.qw = qrec_w; p.wscale = wscale;
  return dispatch3(p, k, stride, geo_env(pick_geo(c, w, k, stride, true)), 5, as_stream(stream));
}
