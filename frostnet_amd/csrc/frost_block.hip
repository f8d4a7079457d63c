// Block-level fused forward kernels of a Frost bottleneck at the 14 x 14 and 7 x 7 stages (CascadePreExBottleneck.forward,
// /root/reference/frostnet.py:124-145: squeeze -> cat -> conv1 (1x1 expand) -> conv2 (depthwise) -> reduce_conv (1x1) -> skip add).
//
// Every ConvBN(ReLU) of the block needs two global reductions before its output exists (BatchNorm batch statistics + the activation observer's
// min / max), so a block forward keeps its synchronisation points whatever is fused.  What fusion removes at these stages is a full read / write of
// the EXPANDED tensor (6 x the block input) and one latency-bound launch per boundary:
//
//   A  k_blk_expand_dw      conv1's emit pass  +  conv2's statistics pass (+ its folded finalize)
//        a workgroup owns ONE image (49 / 196 pixels): the block-internal input x (cat) is staged in LDS once; per 64-channel chunk the int8 MFMA
//        GEMM produces conv1's accumulators, the epilogue quantises them (same expression as k_pw's emit) straight into a zero-point-haloed
//        [row][col][64 ch] LDS plane, and the depthwise stencil runs on that plane with lane = channel (ds_read_b64_tr_b8 + v_dot4_i32_i8, the core of
//        k_dw3) accumulating conv2's exact integer statistics.  y1 leaves for HBM once (the backward and conv2's emit pass read it), never re-read here.
//   B  k_blk_dw_reduce      conv2's emit pass  +  reduce_conv's int8 GEMM / statistics pass (+ folded finalize), K-split over the same chunks
//        y1 chunk -> plane -> depthwise -> quantise y2 (LDS, and out to HBM for the backward) -> MFMA partial sums of the reduce conv stay in
//        registers across the chunks; the integer conv output is stored once (the element-wise emit / backward of the kept-output path use it).
//
// Results are bit-identical to the layer-by-layer kernels: same integer accumulators, same quantisation expression, exact integer statistics.
#include "frost_common.h"

typedef int v2i __attribute__((ext_vector_type(2)));
#include <stdlib.h>

// XCD-aware block index (speed only): workgroup b runs on XCD b % 8, and the (image group, chunk) kernels below read 64-byte (int8) / 128-byte (bf16) pieces of
// pixel rows whose length is no multiple of a cache line (c = 312 ... 1728 channels), so a neighbouring chunk's workgroup on ANOTHER XCD fetches the shared lines
// again (counters: 1.7 - 2.9x the algorithmic bytes).  Runs of 4 consecutive work items (= neighbouring chunks of one image group) are therefore dealt to ONE XCD,
// resident together.  Bijective on [0, total): the last total % 32 items keep their index.  FROST_BLK_XCD=0: the plain map (A/B runs).
__device__ __forceinline__ int blk_xcd_item(int b, int total, int on) {
  if (!on || b >= (total & ~31)) return b;
  const int xcd = b & 7, j = b >> 3;
  return (j >> 2) * 32 + xcd * 4 + (j & 3);
}
// the same with runs of `run` consecutive items per XCD (k_blk_expand_dw: the `csplit` workgroups of one image group read the same input rows)
__device__ __forceinline__ int blk_xcd_run(int b, int total, int run) {
  if (run < 2 || b >= total - total % (8 * run)) return b;
  const int xcd = b & 7, j = b >> 3;
  return (j / run) * (8 * run) + xcd * run + (j % run);
}
static int blk_xcd_on() { static const int on = getenv("FROST_BLK_XCD") ? atoi(getenv("FROST_BLK_XCD")) : 1; return on; }

struct BlkAP {
  const int8_t* x; const float* qx;                 // conv1's input (offset-binary bytes, [n*map][cin]) and its record
  const int8_t* w1; const int32_t* wsum1;           // conv1: MFMA-fragment weight pack [CT][KS][64][16 B], per-channel weight sums
  const float* coef1; const float* qy1; int8_t* y1; // conv1: coefficient rows (finalized), output record, output tensor [n*map][c]
  const int8_t* wq2; const int32_t* wsum2;          // conv2 (depthwise): taps [k*k][cpad], weight sums
  uint8_t* stats2; FrostFinDesc fin;                // conv2: statistics table and the finalize descriptor
  int n, cin, c, cpad, KS, kstr, nchunk, csplit, imgs, rounds, xrun;
};

template <int K, int HW, int NW>
struct BlkGeo {
  static constexpr int MAP = HW * HW, NPT = (MAP + 15) / 16;      // pixels of one image, 16-pixel MFMA tiles
  static constexpr int NPH = NW / 4, NPTW = (NPT + NPH - 1) / NPH;  // pixel-tile halves over the waves, tiles per wave
  static constexpr int PAD = (K - 1) / 2, NRG = (HW + 1) / 2;     // depthwise: 2 output rows per wave unit
  static constexpr int PHA = 2 * NRG + K - 1;                      // plane rows (incl. the rows an odd map's last unit reads past the halo)
  static constexpr int NSEG = (HW + 7) / 8, PITCH = NSEG * 8 + 8;  // 8-column output segments; plane columns (16 read per segment)
  static constexpr int PLANE = PHA * PITCH * 64;
  static constexpr int NT = NW * 64;
  // input rows + plane + ticket flag + per channel of the chunk range: statistics 24 B, conv1 rows 12 B, conv2 weight sum 4 B, taps k*k B
  __host__ __device__ static constexpr int lds(int kstr, int nch) { return NPT * 16 * kstr + PLANE + 64 + nch * 64 * (24 + 12 + 4 + K * K); }
};

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would serialise every barrier behind the
// weight-fragment prefetch and the y stores in flight (an L2 / HBM round trip per chunk)
__device__ __forceinline__ void blk_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ v2i blk_tr8(const uint8_t* row, int col0, int lane) {     // 8 consecutive pixels of channel `lane` of a [col][64 ch] row
  const int jp = lane & 15, G = lane >> 4;
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(row + (col0 + (jp >> 1)) * 64 + 16 * G + 8 * (jp & 1)));
}

// depthwise accumulators of one wave unit: output rows r0, r0 + 1, columns seg*8 .. +8 of channel `lane`, from the plane (signed bytes q - 128)
// RV: columns of the segment that exist (a 7 x 7 map has 7: the eighth column of every stencil, epilogue and store was computed and thrown away -- 1/8 of the work)
template <int K, int PITCH, int RV = 8>
__device__ __forceinline__ void blk_dw_unit(const uint8_t* pl, int r0, int seg, int lane, const int (&wpk)[K][2], int acc0, int (&acc)[2][8]) {
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[o][r] = acc0;
  constexpr int NWIN = (K == 3) ? 8 : 12;
#pragma unroll
  for (int jr = 0; jr < K + 1; ++jr) {
    const uint8_t* rowp = pl + (r0 + jr) * PITCH * 64;
    int d[5];
    const v2i a = blk_tr8(rowp, seg * 8, lane), b = blk_tr8(rowp, seg * 8 + 8, lane);
    d[0] = a[0]; d[1] = a[1]; d[2] = b[0]; d[3] = b[1]; d[4] = 0;
    int win[NWIN];
#pragma unroll
    for (int i = 0; i < NWIN; ++i) win[i] = (i % 4 == 0) ? d[i / 4] : __builtin_amdgcn_alignbyte(d[i / 4 + 1], d[i / 4], i % 4);
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      if ((jr - ky) >= 0 && (jr - ky) < 2) {
        const int o = jr - ky;
#pragma unroll
        for (int r = 0; r < RV; ++r) {
          acc[o][r] = __builtin_amdgcn_sdot4(win[r], wpk[ky][0], acc[o][r], false);
          if (K == 5) acc[o][r] = __builtin_amdgcn_sdot4(win[r + 4], wpk[ky][1], acc[o][r], false);
        }
      }
    }
  }
}

template <int K, int HW, int NW, int KSM>
__global__ __launch_bounds__(NW * 64, (NW == 4) ? 3 : 4) void k_blk_expand_dw(const BlkAP p) {
  using G = BlkGeo<K, HW, NW>;
  constexpr int MAP = G::MAP, NPT = G::NPT, NPTW = G::NPTW, PAD = G::PAD, PITCH = G::PITCH, NT = G::NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xs = smem;                                     // [NPT*16][kstr]
  uint8_t* const pl = smem + NPT * 16 * p.kstr;                 // [PHA][PITCH][64]
  int* const sflag = (int*)(pl + G::PLANE);
  // per-channel rows of the workgroup's chunk range (cw channels), resident for all its images:
  //   conv2 statistics s1, s2 (8 B each), min, max | conv1 A, B, weight sum | conv2 weight sum | conv2 taps [chunk][k*k][64]
  const int bi = blk_xcd_run((int)blockIdx.x, (int)gridDim.x, p.xrun);
  const int cs = bi % p.csplit, ig = bi / p.csplit;
  const int chunk_lo = (cs * p.nchunk) / p.csplit, chunk_hi = ((cs + 1) * p.nchunk) / p.csplit;
  const int cw = (chunk_hi - chunk_lo) * 64;
  unsigned long long* const l_s1 = (unsigned long long*)(sflag + 16);
  unsigned long long* const l_s2 = l_s1 + cw;
  int* const l_mn = (int*)(l_s2 + cw); int* const l_mx = l_mn + cw;
  float* const tabA = (float*)(l_mx + cw); float* const tabB = tabA + cw;
  int* const tabW = (int*)(tabB + cw); int* const tabW2 = tabW + cw;
  uint8_t* const taps = (uint8_t*)(tabW2 + cw);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int CT = p.cpad >> 4;
  const int img_lo = ig * p.imgs, img_hi = min(img_lo + p.imgs, p.n);

  // ---- prologue: the plane's zero-point fill, identity statistics, the chunk range's per-channel rows and taps
  const int zp1 = __float_as_int(p.qy1[FROST_Q_ZP]);
  {
    const uint32_t zf = (uint32_t)((zp1 - 128) & 255) * 0x01010101u;
    for (int i = tid; i < (G::PLANE >> 4); i += NT) ((uint4*)pl)[i] = make_uint4(zf, zf, zf, zf);
    for (int i = tid; i < cw; i += NT) {
      const int c2 = chunk_lo * 64 + i; const bool ok = c2 < p.c;
      l_s1[i] = 0; l_s2[i] = 0; l_mn[i] = INT32_MAX; l_mx[i] = INT32_MIN;
      const int cc = ok ? c2 : 0;          // unconditional from a clamped channel, zeroed afterwards (a load under `ok ?` waits at its own join)
      const float rA = p.coef1[FROST_COEF_A * p.cpad + cc], rB = p.coef1[FROST_COEF_B * p.cpad + cc]; const int rW = p.wsum1[cc], rW2 = p.wsum2[cc];
      tabA[i] = ok ? rA : 0.0f; tabB[i] = ok ? rB : 0.0f; tabW[i] = ok ? rW : 0; tabW2[i] = ok ? rW2 : 0;
    }
    const int ntap = (chunk_hi - chunk_lo) * K * K * 16;                          // 4 channels of one tap per thread, in batches of 4 loads
    for (int i0 = tid; i0 < ntap; i0 += 4 * NT) {
      uint32_t tv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q * NT; const int cl = i / (K * K * 16), rem = i - cl * (K * K * 16), t = rem >> 4, c4 = (rem & 15) * 4;
        const int c2 = (chunk_lo + cl) * 64 + c4;
        tv[q] = (i < ntap && c2 < p.c) ? *(const uint32_t*)(p.wq2 + t * p.cpad + c2) : 0u;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q * NT; const int cl = i / (K * K * 16), rem = i - cl * (K * K * 16), t = rem >> 4, c4 = (rem & 15) * 4;
        if (i < ntap) *(uint32_t*)(taps + (cl * K * K + t) * 64 + c4) = tv[q];
      }
    }
  }
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  const float y_inv = 1.0f / p.qy1[FROST_Q_SCALE], y_zpf = (float)zp1;
  const float qcap = (float)q_hi(p.qy1); const bool lowq = qcap < 255.0f;
  // this lane's pixel of each of the wave's pixel tiles -> byte offset of the pixel inside the plane (-1: padding pixel)
  const int ph = (NW == 8) ? (w >> 2) : 0, wct = w & 3;
  int off[NPTW];
#pragma unroll
  for (int t = 0; t < NPTW; ++t) {
    const int pi = (ph * NPTW + t) * 16 + j;
    const int r = pi / HW, cc = pi - r * HW;
    off[t] = (pi < MAP) ? ((r + PAD) * PITCH + cc + PAD) * 64 : -1;
  }
  auto load_w1 = [&](int chunk, v4i (&dst)[KSM]) __attribute__((always_inline)) {
    const int ct = min(chunk * 4 + wct, CT - 1);          // a partial last chunk recomputes the last tile (its bytes are never used)
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks)
      dst[ks] = *(const v4i*)(p.w1 + ((((int64_t)ct * KSM + ks) * 64 + lane) << 4));
  };
  v4i afr[KSM];
  if (chunk_lo < chunk_hi) load_w1(chunk_lo, afr);

  for (int img = img_lo; img < img_hi; ++img) {
    {   // the image's input rows (the previous image's last barrier freed the buffer)
      const int8_t* src = p.x + (int64_t)img * MAP * p.cin;
      const int upr = p.cin >> 3, total = MAP * upr;
      constexpr int XB = (MAP * KSM * 8 + NT - 1) / NT;           // 8-byte units per thread: all loads in flight before the first LDS store
      uint2 xv[XB];
#pragma unroll
      for (int i = 0; i < XB; ++i) { const int u = tid + i * NT; xv[i] = (u < total) ? *(const uint2*)(src + (int64_t)u * 8) : make_uint2(0, 0); }
#pragma unroll
      for (int i = 0; i < XB; ++i) {
        const int u = tid + i * NT; const int row = u / upr, col = u - row * upr;
        if (u < total) *(uint2*)(xs + row * p.kstr + col * 8) = xv[i];
      }
    }
    blk_barrier();
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
      const int cl = chunk - chunk_lo;
      const int ti = cl * 64 + wct * 16 + 4 * g;       // GEMM epilogue: 4 consecutive channels of this lane (index into the resident rows)
      const bool chok = (chunk * 64 + lane) < p.c;     // depthwise: this lane's channel
      // ---- conv1: int8 MFMA GEMM of the wave's channel tile over its pixel tiles
      v4i acc[NPTW];
      float A[4], B[4];
      {
        const int4 ws = *(const int4*)(tabW + ti); const v4i init = (v4i){-zpx * ws.x, -zpx * ws.y, -zpx * ws.z, -zpx * ws.w};
        const float4 A4 = *(const float4*)(tabA + ti), B4 = *(const float4*)(tabB + ti);
        A[0] = A4.x; A[1] = A4.y; A[2] = A4.z; A[3] = A4.w; B[0] = B4.x; B[1] = B4.y; B[2] = B4.z; B[3] = B4.w;
#pragma unroll
        for (int t = 0; t < NPTW; ++t) acc[t] = init;
      }
      {
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
#pragma unroll
          for (int t = 0; t < NPTW; ++t) {       // (the second half's last tile may lie past the image: it reads on into the plane and is dropped)
            const v4i bfr = *(const v4i*)(xs + ((ph * NPTW + t) * 16 + j) * p.kstr + ks * 64 + g * 16);
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[ks], bfr, acc[t], 0, 0, 0);          // D[chan][pix]
          }
        }
        // emit: q = clamp(rint(fma(A, acc, B) / s) + zp, 0, hi) -- k_pw's expression -- into the plane
#pragma unroll
        for (int t = 0; t < NPTW; ++t) {
          uint32_t packed = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float yv = fmaf(A[r], (float)acc[t][r], B[r]);
            float qv = rintf(yv * y_inv) + y_zpf;
            if (lowq) qv = fminf(qv, qcap);
            packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);
          }
          if (off[t] >= 0) *(uint32_t*)(pl + off[t] + wct * 16 + 4 * g) = packed ^ 0x80808080u;
        }
      }
      blk_barrier();                                           // the chunk's y1 plane is complete
      {   // the next (image, chunk)'s weight fragments travel under the depthwise phase
        const int nxt = (chunk + 1 < chunk_hi) ? chunk + 1 : chunk_lo;
        if (chunk + 1 < chunk_hi || img + 1 < img_hi) load_w1(nxt, afr);
      }
      // ---- y1 chunk out to HBM (8-byte pieces, 64 contiguous bytes per pixel)
      {
        int8_t* dst = p.y1 + (int64_t)img * MAP * p.c + chunk * 64;
        for (int u = tid; u < MAP * 8; u += NT) {
          const int px = u >> 3, part = u & 7;
          const int r = px / HW, cc = px - r * HW;
          if (chunk * 64 + part * 8 < p.c) *(uint2*)(dst + (int64_t)px * p.c + part * 8) = *(const uint2*)(pl + ((r + PAD) * PITCH + cc + PAD) * 64 + part * 8);
        }
      }
      // ---- conv2 statistics: wave w = output rows 2w, 2w + 1 of the map, lane = channel
      if (w < G::NRG) {
        const uint8_t* tapb = taps + cl * K * K * 64;
        int wpk[K][2];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          uint32_t lo = 0, hi = 0;
#pragma unroll
          for (int kx = 0; kx < K; ++kx) { const uint32_t b = tapb[(ky * K + kx) * 64 + lane]; if (kx < 4) lo |= b << (8 * kx); else hi |= b; }
          wpk[ky][0] = (int)lo; wpk[ky][1] = (int)hi;
        }
        const int acc0 = (128 - zp1) * tabW2[cl * 64 + lane];
        int t1 = 0; double t2 = 0.0; int tmn = INT32_MAX, tmx = INT32_MIN;
#pragma unroll 1
        for (int seg = 0; seg < G::NSEG; ++seg) {
          int a[2][8];
          blk_dw_unit<K, PITCH, (HW == 7) ? 7 : 8>(pl, 2 * w, seg, lane, wpk, acc0, a);
#pragma unroll
          for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < ((HW == 7) ? 7 : 8); ++r) {
              if ((2 * w + o) < HW && (seg * 8 + r) < HW) {
                const int vi = a[o][r]; const float v = (float)vi;
                t1 += vi; t2 = fma((double)v, (double)v, t2); tmn = min(tmn, vi); tmx = max(tmx, vi);
              }
            }
        }
        if (chok) {
          const int li = cl * 64 + lane;
          atomicAdd(&l_s1[li], (unsigned long long)(long long)t1); atomicAdd(&l_s2[li], (unsigned long long)t2);
          atomicMin(&l_mn[li], tmn); atomicMax(&l_mx[li], tmx);
        }
      }
      blk_barrier();                                           // plane (and, after the last chunk, the input rows) free again
    }
  }
  // ---- one set of global atomics per channel of the range, then: last workgroup done -> conv2's finalize in this launch
  {
    long long* g_s1 = (long long*)stats_copy(p.stats2, p.cpad); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
    int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
    for (int i = tid; i < cw; i += NT) {
      const int c2 = chunk_lo * 64 + i;
      if (c2 < p.c && l_mn[i] <= l_mx[i]) {
        atomicAdd((unsigned long long*)&g_s1[c2], l_s1[i]); atomicAdd(&g_s2[c2], l_s2[i]);
        atomicMin(&g_mn[c2], l_mn[i]); atomicMax(&g_mx[c2], l_mx[i]);
      }
    }
  }
  if (last_block_done2(p.fin.counter, gridDim.x, sflag)) {
    float* sh = (float*)smem;
    conv_finalize_dev(p.stats2, (int64_t)p.n * MAP, p.c, p.cpad, p.qy1, p.fin.qrec_w, p.fin.wscale, p.fin.gamma, p.fin.beta, p.fin.rmean, p.fin.rvar, p.fin.nbt,
                      p.fin.training, p.fin.relu, p.fin.observe, 1, p.fin.coef, p.fin.qrec_y, tid, NT, sh);
  }
}

template <int K, int HW, int NW, int KSM>
static int launch_blk_a(BlkAP& p, hipStream_t s) {
  using G = BlkGeo<K, HW, NW>;
  const size_t lds = (size_t)G::lds(p.kstr, (p.nchunk + p.csplit - 1) / p.csplit);
  FROST_REQUIRE(lds <= 160 * 1024, "block_expand_dw: LDS budget exceeded");
  static bool attr_set = false; static int occ = 0; static size_t occ_lds = 0;
  if (!attr_set || occ_lds != lds) {
    hipFuncSetAttribute((const void*)k_blk_expand_dw<K, HW, NW, KSM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; occ_lds = lds;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_blk_expand_dw<K, HW, NW, KSM>, NW * 64, lds) != hipSuccess || occ < 1) occ = 1;
  }
  if (p.imgs <= 0) {       // whole rounds of resident workgroups: image groups = floor(rounds * slots / chunk ranges)
    const int rounds = p.rounds > 0 ? p.rounds : 1;
    int groups = (256 * occ * rounds) / p.csplit; if (groups < 1) groups = 1; if (groups > p.n) groups = p.n;
    p.imgs = (p.n + groups - 1) / groups;
  }
  p.xrun = (blk_xcd_on() && p.csplit >= 2 && p.csplit <= 8) ? p.csplit : 0;
  hipLaunchKernelGGL((k_blk_expand_dw<K, HW, NW, KSM>), dim3((unsigned)(((p.n + p.imgs - 1) / p.imgs) * p.csplit)), dim3(NW * 64), lds, s, p);
  return frost_check_launch("block_expand_dw");
}

// ================================================================================================ kernel B: conv2 emit + reduce_conv GEMM / statistics
struct BlkBP {
  const int8_t* y1; const float* qy1;               // conv2's input (conv1's output) and its record
  const int8_t* wq2; const int32_t* wsum2;          // conv2 (depthwise): taps [k*k][cpad], weight sums
  const float* coef2; const float* qy2; int8_t* y2; // conv2: coefficient rows (finalized), output record, output tensor [n*map][c]
  const int8_t* w3; const int32_t* wsum3;           // reduce_conv: MFMA-fragment weight pack [CT3][nchunk][64][16 B], weight sums
  int32_t* cint; uint8_t* stats3; FrostFinDesc fin; // reduce_conv: integer conv output [n*map][cout], statistics table, finalize descriptor
  int n, c, cpad, cout, cpad3, nchunk, imgs, relu2;
};

template <int K, int HW, int NW>
struct BlkGeoB : BlkGeo<K, HW, NW> {
  using G = BlkGeo<K, HW, NW>;
  static constexpr int Y2 = G::NPH * G::NPTW * 16 * 64;                  // conv2's quantised output of one chunk, [pixel][64 ch]: the GEMM's B operand as it lies
  __host__ __device__ static constexpr int lds(int cpad3) { return G::PLANE + Y2 + 2 * K * K * 64 + 64 + cpad3 * 24; }
};

// NCTW: reduce_conv output-channel tiles per wave (tiles wct, wct + 4, ...)
template <int K, int HW, int NW, int NCTW>
__global__ __launch_bounds__(NW * 64, 2) void k_blk_dw_reduce(const BlkBP p) {
  using G = BlkGeoB<K, HW, NW>;
  constexpr int MAP = G::MAP, NPTW = G::NPTW, PAD = G::PAD, PITCH = G::PITCH, NT = G::NT;
  constexpr int YU = (MAP * 8 + NT - 1) / NT;                    // 8-byte units of a y chunk per thread
  constexpr int NTAPW = (K * K * 16 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const pl = smem;                                      // [PHA][PITCH][64]  y1 chunk with its zero-point halo
  uint8_t* const y2t = smem + G::PLANE;                          // [NPT*16][64]      y2 chunk
  uint8_t* const taps = y2t + G::Y2;                             // [2][k*k][64]
  int* const sflag = (int*)(taps + 2 * K * K * 64);
  unsigned long long* const l_s1 = (unsigned long long*)(sflag + 16);
  unsigned long long* const l_s2 = l_s1 + p.cpad3;
  int* const l_mn = (int*)(l_s2 + p.cpad3); int* const l_mx = l_mn + p.cpad3;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int ph = (NW == 8) ? (w >> 2) : 0, wct = w & 3;
  const int CT3 = p.cpad3 >> 4;
  const int img_lo = (int)blockIdx.x * p.imgs, img_hi = min(img_lo + p.imgs, p.n);
  const int nit = (img_hi - img_lo) * p.nchunk;

  const int zp1 = __float_as_int(p.qy1[FROST_Q_ZP]), zp2 = __float_as_int(p.qy2[FROST_Q_ZP]);
  {
    const uint32_t zf = (uint32_t)((zp1 - 128) & 255) * 0x01010101u;
    for (int i = tid; i < (G::PLANE >> 4); i += NT) ((uint4*)pl)[i] = make_uint4(zf, zf, zf, zf);
    for (int i = tid; i < (G::Y2 >> 4); i += NT) ((uint4*)y2t)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < p.cpad3; i += NT) { l_s1[i] = 0; l_s2[i] = 0; l_mn[i] = INT32_MAX; l_mx[i] = INT32_MIN; }
  }
  const float y_inv = 1.0f / p.qy2[FROST_Q_SCALE], y_zpf = (float)zp2;
  const float qcap = (float)q_hi(p.qy2); const bool lowq = qcap < 255.0f;
  const float relu_floor = p.relu2 ? 0.0f : -INFINITY;
  const int zpx3 = zp2 - 128;
  // the plane / tensor position of the y units this thread moves (the same for every chunk)
  int uoff[YU]; int upx[YU];
#pragma unroll
  for (int i = 0; i < YU; ++i) {
    const int u = tid + i * NT, px = u >> 3, part = u & 7;
    const int r = px / HW, cc = px - r * HW;
    upx[i] = (u < MAP * 8) ? px : -1;
    uoff[i] = ((r + PAD) * PITCH + cc + PAD) * 64 + part * 8;
  }
  const int upart = (tid & 7) * 8;                                // NT is a multiple of 8: every unit of a thread has the same channel part
  const int tap_c = (tid & 15) * 4;
  auto load_y1 = [&](int img, int chunk, uint2 (&dst)[YU]) __attribute__((always_inline)) {
    const int8_t* src = p.y1 + (int64_t)img * MAP * p.c + chunk * 64 + upart;
    const bool cok = (chunk * 64 + upart) < p.c;
#pragma unroll
    for (int i = 0; i < YU; ++i) dst[i] = (upx[i] >= 0 && cok) ? *(const uint2*)(src + (int64_t)upx[i] * p.c) : make_uint2(0, 0);
  };
  auto load_taps = [&](int chunk, uint32_t (&dst)[NTAPW]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NTAPW; ++i) {
      const int t = (tid + i * NT) >> 4;
      dst[i] = (t < K * K && (chunk * 64 + tap_c) < p.c) ? *(const uint32_t*)(p.wq2 + t * p.cpad + chunk * 64 + tap_c) : 0u;
    }
  };
  struct ChRow { float A, B; int ws; };
  // the row of the NEXT chunk is requested with the other prefetches and must stay un-touched until it is consumed: zeroing the lanes past the last channel right here
  // (`ok ? rA : 0`) is a use of the loaded value, i.e. an s_waitcnt vmcnt(0) in front of the depthwise phase that drains the whole prefetch (y1 rows, taps, weight
  // fragments) every iteration -- seen in the ISA; the select now happens where the row is taken over, one iteration later (row_take)
  auto load_row = [&](int chunk) __attribute__((always_inline)) {
    ChRow r; const int c2 = chunk * 64 + lane; const int cc = (c2 < p.c) ? c2 : 0;
    r.A = p.coef2[FROST_COEF_A * p.cpad + cc]; r.B = p.coef2[FROST_COEF_B * p.cpad + cc]; r.ws = p.wsum2[cc];
    return r;
  };
  auto row_take = [&](const ChRow& raw, int chunk) __attribute__((always_inline)) {
    ChRow r; const bool ok = (chunk * 64 + lane) < p.c;
    r.A = ok ? raw.A : 0.0f; r.B = ok ? raw.B : 0.0f; r.ws = ok ? raw.ws : 0;
    return r;
  };
  uint2 yv[YU]; uint32_t tv[NTAPW]; ChRow row_n;
  if (nit > 0) { load_y1(img_lo, 0, yv); load_taps(0, tv); row_n = load_row(0); }
  v4i acc[NCTW][NPTW];
  __syncthreads();

  for (int it = 0; it < nit; ++it) {
    const int img = img_lo + it / p.nchunk, chunk = it - (it / p.nchunk) * p.nchunk;
    uint8_t* const tapb = taps + (it & 1) * K * K * 64;
    if (chunk == 0) {
#pragma unroll
      for (int m = 0; m < NCTW; ++m)
#pragma unroll
        for (int t = 0; t < NPTW; ++t) acc[m][t] = (v4i){0, 0, 0, 0};
    }
    // ---- 1. this chunk's y1 rows into the plane, its taps into LDS; 2. the next chunk's come into registers under the arithmetic
#pragma unroll
    for (int i = 0; i < YU; ++i) if (upx[i] >= 0) *(uint2*)(pl + uoff[i]) = yv[i];
#pragma unroll
    for (int i = 0; i < NTAPW; ++i) { const int t = (tid + i * NT) >> 4; if (t < K * K) *(uint32_t*)(tapb + t * 64 + tap_c) = tv[i]; }
    const ChRow row = row_take(row_n, chunk);
    v4i afr[NCTW];
#pragma unroll
    for (int m = 0; m < NCTW; ++m) {       // reduce_conv weight fragments of this K step (one chunk = 64 input channels), consumed after the depthwise phase
      const int ct = min(wct + 4 * m, CT3 - 1);
      afr[m] = *(const v4i*)(p.w3 + ((((int64_t)ct * p.nchunk + chunk) * 64 + lane) << 4));
    }
    if (it + 1 < nit) {
      const int it2 = it + 1; const int img2 = img_lo + it2 / p.nchunk, chunk2 = it2 - (it2 / p.nchunk) * p.nchunk;
      load_y1(img2, chunk2, yv); load_taps(chunk2, tv); row_n = load_row(chunk2);
    }
    blk_barrier();
    // ---- conv2: depthwise conv + emit (k_dw3's expression) of output rows 2w, 2w + 1, lane = channel -> y2 chunk [pixel][64]
    if (w < G::NRG) {
      int wpk[K][2];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) { const uint32_t b = tapb[(ky * K + kx) * 64 + lane]; if (kx < 4) lo |= b << (8 * kx); else hi |= b; }
        wpk[ky][0] = (int)lo; wpk[ky][1] = (int)hi;
      }
      const int acc0 = (128 - zp1) * row.ws;
#pragma unroll 1
      for (int seg = 0; seg < G::NSEG; ++seg) {
        int a[2][8];
        blk_dw_unit<K, PITCH, (HW == 7) ? 7 : 8>(pl, 2 * w, seg, lane, wpk, acc0, a);
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int r = 0; r < ((HW == 7) ? 7 : 8); ++r) {
            if ((2 * w + o) < HW && (seg * 8 + r) < HW) {
              const float yf = fmaf(row.A, (float)a[o][r], row.B);
              float qv = rintf(fmaxf(yf, relu_floor) * y_inv) + y_zpf;
              if (lowq) qv = fminf(qv, qcap);
              y2t[((2 * w + o) * HW + seg * 8 + r) * 64 + lane] = (uint8_t)((__builtin_amdgcn_cvt_pk_u8_f32(qv, 0, 0u) ^ 0x80u) & 255u);
            }
          }
      }
    }
    blk_barrier();
    // ---- y2 chunk out to HBM; reduce_conv: acc[cout tile][pixel tile] += W3[:, chunk] * y2 chunk
    {
      int8_t* dst = p.y2 + (int64_t)img * MAP * p.c + chunk * 64 + upart;
      if ((chunk * 64 + upart) < p.c) {
#pragma unroll
        for (int i = 0; i < YU; ++i) if (upx[i] >= 0) *(uint2*)(dst + (int64_t)upx[i] * p.c) = *(const uint2*)(y2t + upx[i] * 64 + upart);
      }
    }
#pragma unroll
    for (int t = 0; t < NPTW; ++t) {
      const v4i bfr = *(const v4i*)(y2t + ((ph * NPTW + t) * 16 + j) * 64 + g * 16);
#pragma unroll
      for (int m = 0; m < NCTW; ++m) acc[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[m], bfr, acc[m][t], 0, 0, 0);        // D[chan][pix]
    }
    if (chunk == p.nchunk - 1) {
      // ---- the image's integer conv output c = acc - (zp - 128) * wsum and its exact statistics (table format of k_pw / conv_finalize_dev)
#pragma unroll
      for (int m = 0; m < NCTW; ++m) {
        const int ct = wct + 4 * m;
        if (ct >= CT3) continue;
        const int ch0 = ct * 16 + 4 * g;
        const v4i ws = *(const v4i*)(p.wsum3 + ch0);
        long long a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}; int mn[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX}, mx[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
#pragma unroll
        for (int t = 0; t < NPTW; ++t) {
          const int px = (ph * NPTW + t) * 16 + j;
          if (px < MAP && ch0 < p.cout) {
            const v4i cv = (v4i){acc[m][t][0] - zpx3 * ws[0], acc[m][t][1] - zpx3 * ws[1], acc[m][t][2] - zpx3 * ws[2], acc[m][t][3] - zpx3 * ws[3]};
            *(v4i*)(p.cint + ((int64_t)img * MAP + px) * p.cout + ch0) = cv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int v = cv[r]; a1[r] += v; a2[r] += (long long)v * v; mn[r] = min(mn[r], v); mx[r] = max(mx[r], v); }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {                               // the 16 pixel lanes j of this g
            a1[r] += __shfl_xor(a1[r], o); a2[r] += __shfl_xor(a2[r], o); mn[r] = min(mn[r], __shfl_xor(mn[r], o)); mx[r] = max(mx[r], __shfl_xor(mx[r], o));
          }
          if (j == 0 && (ch0 + r) < p.cout && mn[r] <= mx[r]) {
            atomicAdd(&l_s1[ch0 + r], (unsigned long long)a1[r]); atomicAdd(&l_s2[ch0 + r], (unsigned long long)a2[r]);
            atomicMin(&l_mn[ch0 + r], mn[r]); atomicMax(&l_mx[ch0 + r], mx[r]);
          }
        }
      }
    }
  }
  __syncthreads();
  {
    long long* g_s1 = (long long*)stats_copy(p.stats3, p.cpad3); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad3);
    int* g_mn = (int*)(g_s2 + p.cpad3); int* g_mx = g_mn + p.cpad3;
    for (int i = tid; i < p.cout; i += NT) {
      if (l_mn[i] <= l_mx[i]) {
        atomicAdd((unsigned long long*)&g_s1[i], l_s1[i]); atomicAdd(&g_s2[i], l_s2[i]);
        atomicMin(&g_mn[i], l_mn[i]); atomicMax(&g_mx[i], l_mx[i]);
      }
    }
  }
  if (last_block_done2(p.fin.counter, gridDim.x, sflag)) {
    float* sh = (float*)smem;
    conv_finalize_dev(p.stats3, (int64_t)p.n * MAP, p.cout, p.cpad3, p.qy2, p.fin.qrec_w, p.fin.wscale, p.fin.gamma, p.fin.beta, p.fin.rmean, p.fin.rvar, p.fin.nbt,
                      p.fin.training, p.fin.relu, p.fin.observe, 1, p.fin.coef, p.fin.qrec_y, tid, NT, sh, p.fin.cat_qrec_b, p.fin.cat_qrec_y);
  }
}

// ---- the same kernel for 7 x 7 maps with TWO 64-channel chunks per iteration on eight waves (waves 0-3: the depthwise stencil of the even chunk, waves 4-7: of the odd
// chunk; the reduce GEMM's pixel tiles split over the two wave groups, each wave taking both chunks' K steps).  One image per workgroup is one serial chain of chunk
// iterations with ONE wave per SIMD and workgroup; this form halves the chain and puts two waves of the workgroup on every SIMD.  Integer results: bit-identical to
// k_blk_dw_reduce<K, 7, 4, NCTW>.  MEASURED SLOWER at B = 512 (see frost_block_dw_reduce below): its accumulators + both chunks' fragments need > 128 registers, so only one
// workgroup fits a CU (or it spills), and 256 workgroups at a time of half the chain each is no shorter than 512 at a time of the whole chain.  Kept as an A/B switch.
template <int K, int NCTW>
__global__ __launch_bounds__(512, 2) void k_blk_dw_reduce2(const BlkBP p) {
  using G = BlkGeoB<K, 7, 4>;                                     // plane / stencil-unit geometry of ONE chunk
  constexpr int HW = 7, MAP = 49, PAD = G::PAD, PITCH = G::PITCH, NT = 512, NPTW = 2, KK = K * K;
  constexpr int UCH = MAP * 8;                                    // 8-byte units of one y chunk
  constexpr int YU = (2 * UCH + NT - 1) / NT;                     // ... of both chunks, per thread
  constexpr int TCH = KK * 16;                                    // dword units of one chunk's taps
  constexpr int NTAPW = (2 * TCH + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const pl = smem;                                       // [2][PHA][PITCH][64]
  uint8_t* const y2t = smem + 2 * G::PLANE;                       // [2][64 px][64]
  uint8_t* const taps = y2t + 2 * G::Y2;                         // [2 (iteration parity)][2][k*k][64]
  int* const sflag = (int*)(taps + 4 * KK * 64);
  unsigned long long* const l_s1 = (unsigned long long*)(sflag + 16);
  unsigned long long* const l_s2 = l_s1 + p.cpad3;
  int* const l_mn = (int*)(l_s2 + p.cpad3); int* const l_mx = l_mn + p.cpad3;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int slot_w = w >> 2, unit = w & 3;                        // depthwise phase: this wave's chunk slot and its pair of output rows
  const int ph = w >> 2, wct = w & 3;                             // GEMM phase: pixel-tile half, first channel tile
  const int CT3 = p.cpad3 >> 4;
  const int img_lo = (int)blockIdx.x * p.imgs, img_hi = min(img_lo + p.imgs, p.n);
  const int nit_img = (p.nchunk + 1) >> 1;
  const int nit = (img_hi - img_lo) * nit_img;

  const int zp1 = __float_as_int(p.qy1[FROST_Q_ZP]), zp2 = __float_as_int(p.qy2[FROST_Q_ZP]);
  {
    const uint32_t zf = (uint32_t)((zp1 - 128) & 255) * 0x01010101u;
    for (int i = tid; i < (2 * G::PLANE >> 4); i += NT) ((uint4*)pl)[i] = make_uint4(zf, zf, zf, zf);
    for (int i = tid; i < (2 * G::Y2 >> 4); i += NT) ((uint4*)y2t)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < p.cpad3; i += NT) { l_s1[i] = 0; l_s2[i] = 0; l_mn[i] = INT32_MAX; l_mx[i] = INT32_MIN; }
  }
  const float y_inv = 1.0f / p.qy2[FROST_Q_SCALE], y_zpf = (float)zp2;
  const float qcap = (float)q_hi(p.qy2); const bool lowq = qcap < 255.0f;
  const float relu_floor = p.relu2 ? 0.0f : -INFINITY;
  const int zpx3 = zp2 - 128;
  // the chunk slot / plane position / tensor pixel of the y units this thread moves (the same for every iteration)
  int uoff[YU], upx[YU], uslot[YU];
#pragma unroll
  for (int i = 0; i < YU; ++i) {
    const int u = tid + i * NT, sl = (u >= UCH) ? 1 : 0, v = u - sl * UCH, px = v >> 3, part = v & 7;
    const int r = px / HW, cc = px - r * HW;
    upx[i] = (u < 2 * UCH) ? px : -1; uslot[i] = sl;
    uoff[i] = sl * G::PLANE + ((r + PAD) * PITCH + cc + PAD) * 64 + part * 8;
  }
  const int upart = (tid & 7) * 8;                                // NT and UCH are multiples of 8: every unit of a thread has the same channel part
  const int tap_c = (tid & 15) * 4;
  auto load_y1 = [&](int img, int itc, uint2 (&dst)[YU]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < YU; ++i) {
      const int chunk = 2 * itc + uslot[i];
      const bool ok = upx[i] >= 0 && chunk < p.nchunk && (chunk * 64 + upart) < p.c;
      const int8_t* src = p.y1 + ((int64_t)img * MAP + (ok ? upx[i] : 0)) * p.c + (ok ? chunk * 64 + upart : 0);
      const uint2 v = *(const uint2*)src;
      dst[i] = ok ? v : make_uint2(0, 0);
    }
  };
  auto load_taps = [&](int itc, uint32_t (&dst)[NTAPW]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NTAPW; ++i) {
      const int v = tid + i * NT, sl = (v >= TCH) ? 1 : 0, t = (v - sl * TCH) >> 4, chunk = 2 * itc + sl;
      const bool ok = v < 2 * TCH && chunk < p.nchunk && (chunk * 64 + tap_c) < p.c;
      const uint32_t q = *(const uint32_t*)(p.wq2 + (ok ? t * p.cpad + chunk * 64 + tap_c : 0));
      dst[i] = ok ? q : 0u;
    }
  };
  struct ChRow { float A, B; int ws; };
  auto load_row = [&](int itc) __attribute__((always_inline)) {          // this wave's chunk slot, lane = channel; raw values: the zeroing waits until the row is taken over
    ChRow r; const int chunk = 2 * itc + slot_w; const int c2 = chunk * 64 + lane; const bool ok = chunk < p.nchunk && c2 < p.c;
    const int cc = ok ? c2 : 0;
    r.A = p.coef2[FROST_COEF_A * p.cpad + cc]; r.B = p.coef2[FROST_COEF_B * p.cpad + cc]; r.ws = p.wsum2[cc];
    return r;
  };
  auto row_take = [&](const ChRow& raw, int itc) __attribute__((always_inline)) {
    ChRow r; const int chunk = 2 * itc + slot_w; const bool ok = chunk < p.nchunk && (chunk * 64 + lane) < p.c;
    r.A = ok ? raw.A : 0.0f; r.B = ok ? raw.B : 0.0f; r.ws = ok ? raw.ws : 0;
    return r;
  };
  uint2 yv[YU]; uint32_t tv[NTAPW]; ChRow row_n;
  if (nit > 0) { load_y1(img_lo, 0, yv); load_taps(0, tv); row_n = load_row(0); }
  v4i acc[NCTW][NPTW];
  __syncthreads();

  for (int it = 0; it < nit; ++it) {
    const int img = img_lo + it / nit_img, itc = it - (it / nit_img) * nit_img;
    const bool have1 = (2 * itc + 1) < p.nchunk;                    // the odd chunk of this iteration exists
    uint8_t* const tapb = taps + (it & 1) * 2 * KK * 64;
    if (itc == 0) {
#pragma unroll
      for (int m = 0; m < NCTW; ++m)
#pragma unroll
        for (int t = 0; t < NPTW; ++t) acc[m][t] = (v4i){0, 0, 0, 0};
    }
    // ---- 1. both chunks' y1 rows into their planes, their taps into LDS; 2. the next iteration's come into registers under the arithmetic
#pragma unroll
    for (int i = 0; i < YU; ++i) if (upx[i] >= 0) *(uint2*)(pl + uoff[i]) = yv[i];
#pragma unroll
    for (int i = 0; i < NTAPW; ++i) {
      const int v = tid + i * NT, sl = (v >= TCH) ? 1 : 0, t = (v - sl * TCH) >> 4;
      if (v < 2 * TCH) *(uint32_t*)(tapb + (sl * KK + t) * 64 + tap_c) = tv[i];
    }
    const ChRow row = row_take(row_n, itc);
    if (it + 1 < nit) {
      const int it2 = it + 1; const int img2 = img_lo + it2 / nit_img, itc2 = it2 - (it2 / nit_img) * nit_img;
      load_y1(img2, itc2, yv); load_taps(itc2, tv); row_n = load_row(itc2);
    }
    blk_barrier();
    // ---- conv2: depthwise conv + emit of output rows 2 unit, 2 unit + 1 of this wave group's chunk, lane = channel -> its y2 chunk [pixel][64]
    if (slot_w == 0 || have1) {
      const uint8_t* const plw = pl + slot_w * G::PLANE; uint8_t* const y2w = y2t + slot_w * G::Y2; const uint8_t* const tapw = tapb + slot_w * KK * 64;
      int wpk[K][2];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) { const uint32_t b = tapw[(ky * K + kx) * 64 + lane]; if (kx < 4) lo |= b << (8 * kx); else hi |= b; }
        wpk[ky][0] = (int)lo; wpk[ky][1] = (int)hi;
      }
      const int acc0 = (128 - zp1) * row.ws;
      int a[2][8];
      blk_dw_unit<K, PITCH, 7>(plw, 2 * unit, 0, lane, wpk, acc0, a);
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 7; ++r) {
          if ((2 * unit + o) < HW) {
            const float yf = fmaf(row.A, (float)a[o][r], row.B);
            float qv = rintf(fmaxf(yf, relu_floor) * y_inv) + y_zpf;
            if (lowq) qv = fminf(qv, qcap);
            y2w[((2 * unit + o) * HW + r) * 64 + lane] = (uint8_t)((__builtin_amdgcn_cvt_pk_u8_f32(qv, 0, 0u) ^ 0x80u) & 255u);
          }
        }
    }
    // reduce_conv weight fragments of the two K steps: requested only now (the stencil's registers are free again -- held across the depthwise phase they cost 40 of
    // the 128 registers a wave may have with two workgroups per CU), they travel under the barrier and the y2 stores
    v4i afr[2][NCTW];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int m = 0; m < NCTW; ++m) {
        const int ct = min(wct + 4 * m, CT3 - 1), chunk = min(2 * itc + sl, p.nchunk - 1);
        const v4i f = *(const v4i*)(p.w3 + ((((int64_t)ct * p.nchunk + chunk) * 64 + lane) << 4));
        afr[sl][m] = (sl == 0 || have1) ? f : (v4i){0, 0, 0, 0};
      }
    blk_barrier();
    // ---- both y2 chunks out to HBM; reduce_conv: acc[cout tile][pixel tile] += W3[:, chunk] * y2 chunk, two K steps
#pragma unroll
    for (int i = 0; i < YU; ++i) {
      const int chunk = 2 * itc + uslot[i];
      if (upx[i] >= 0 && chunk < p.nchunk && (chunk * 64 + upart) < p.c)
        *(uint2*)(p.y2 + ((int64_t)img * MAP + upx[i]) * p.c + chunk * 64 + upart) = *(const uint2*)(y2t + uslot[i] * G::Y2 + upx[i] * 64 + upart);
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int t = 0; t < NPTW; ++t) {
        const v4i bfr = *(const v4i*)(y2t + sl * G::Y2 + ((ph * NPTW + t) * 16 + j) * 64 + g * 16);
#pragma unroll
        for (int m = 0; m < NCTW; ++m) acc[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[sl][m], bfr, acc[m][t], 0, 0, 0);        // D[chan][pix]
      }
    if (itc == nit_img - 1) {
      // ---- the image's integer conv output c = acc - (zp - 128) * wsum and its exact statistics (table format of k_pw / conv_finalize_dev)
#pragma unroll
      for (int m = 0; m < NCTW; ++m) {
        const int ct = wct + 4 * m;
        if (ct >= CT3) continue;
        const int ch0 = ct * 16 + 4 * g;
        const v4i ws = *(const v4i*)(p.wsum3 + ch0);
        long long a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}; int mn[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX}, mx[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
#pragma unroll
        for (int t = 0; t < NPTW; ++t) {
          const int px = (ph * NPTW + t) * 16 + j;
          if (px < MAP && ch0 < p.cout) {
            const v4i cv = (v4i){acc[m][t][0] - zpx3 * ws[0], acc[m][t][1] - zpx3 * ws[1], acc[m][t][2] - zpx3 * ws[2], acc[m][t][3] - zpx3 * ws[3]};
            *(v4i*)(p.cint + ((int64_t)img * MAP + px) * p.cout + ch0) = cv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int v = cv[r]; a1[r] += v; a2[r] += (long long)v * v; mn[r] = min(mn[r], v); mx[r] = max(mx[r], v); }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {                               // the 16 pixel lanes j of this g
            a1[r] += __shfl_xor(a1[r], o); a2[r] += __shfl_xor(a2[r], o); mn[r] = min(mn[r], __shfl_xor(mn[r], o)); mx[r] = max(mx[r], __shfl_xor(mx[r], o));
          }
          if (j == 0 && (ch0 + r) < p.cout && mn[r] <= mx[r]) {
            atomicAdd(&l_s1[ch0 + r], (unsigned long long)a1[r]); atomicAdd(&l_s2[ch0 + r], (unsigned long long)a2[r]);
            atomicMin(&l_mn[ch0 + r], mn[r]); atomicMax(&l_mx[ch0 + r], mx[r]);
          }
        }
      }
    }
  }
  __syncthreads();
  {
    long long* g_s1 = (long long*)stats_copy(p.stats3, p.cpad3); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad3);
    int* g_mn = (int*)(g_s2 + p.cpad3); int* g_mx = g_mn + p.cpad3;
    for (int i = tid; i < p.cout; i += NT) {
      if (l_mn[i] <= l_mx[i]) {
        atomicAdd((unsigned long long*)&g_s1[i], l_s1[i]); atomicAdd(&g_s2[i], l_s2[i]);
        atomicMin(&g_mn[i], l_mn[i]); atomicMax(&g_mx[i], l_mx[i]);
      }
    }
  }
  if (last_block_done2(p.fin.counter, gridDim.x, sflag)) {
    float* sh = (float*)smem;
    conv_finalize_dev(p.stats3, (int64_t)p.n * MAP, p.cout, p.cpad3, p.qy2, p.fin.qrec_w, p.fin.wscale, p.fin.gamma, p.fin.beta, p.fin.rmean, p.fin.rvar, p.fin.nbt,
                      p.fin.training, p.fin.relu, p.fin.observe, 1, p.fin.coef, p.fin.qrec_y, tid, NT, sh, p.fin.cat_qrec_b, p.fin.cat_qrec_y);
  }
}
template <int K, int NCTW>
static int launch_blk_b2(BlkBP& p, hipStream_t s) {
  using G = BlkGeoB<K, 7, 4>;
  const size_t lds = (size_t)2 * G::PLANE + 2 * G::Y2 + 4 * K * K * 64 + 64 + (size_t)p.cpad3 * 24;
  FROST_REQUIRE(lds <= 160 * 1024, "block_dw_reduce: LDS budget exceeded");
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)k_blk_dw_reduce2<K, NCTW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  hipLaunchKernelGGL((k_blk_dw_reduce2<K, NCTW>), dim3((unsigned)((p.n + p.imgs - 1) / p.imgs)), dim3(512), lds, s, p);
  return frost_check_launch("block_dw_reduce");
}

template <int K, int HW, int NW, int NCTW>
static int launch_blk_b(BlkBP& p, hipStream_t s) {
  using G = BlkGeoB<K, HW, NW>;
  const size_t lds = (size_t)G::lds(p.cpad3);
  FROST_REQUIRE(lds <= 160 * 1024, "block_dw_reduce: LDS budget exceeded");
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)k_blk_dw_reduce<K, HW, NW, NCTW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  hipLaunchKernelGGL((k_blk_dw_reduce<K, HW, NW, NCTW>), dim3((unsigned)((p.n + p.imgs - 1) / p.imgs)), dim3(NW * 64), lds, s, p);
  return frost_check_launch("block_dw_reduce");
}

extern "C" int frost_block_dw_reduce_supported(int h, int w, int k, int stride, int c, int cout) {
  const int ct3 = (cout + 15) / 16;
  return (h == w && (h == 7 || h == 14) && (k == 3 || k == 5) && stride == 1 && (c % 8) == 0 && (cout % 4) == 0 && ct3 <= (h == 7 ? 20 : 8)) ? 1 : 0;
}

extern "C" int frost_block_dw_reduce(const int8_t* y1, const float* qrec_y1, const int8_t* wq2, const int32_t* wsum2, const float* coef2, const float* qrec_y2,
                                     int relu2, int8_t* y2, int n, int h, int w, int c, int k, const int8_t* w3_pack, const int32_t* wsum3, int cout,
                                     int32_t* conv_out, void* stats3, const FrostFinDesc* fin3, void* stream) {
  FROST_REQUIRE(frost_block_dw_reduce_supported(h, w, k, 1, c, cout), "block_dw_reduce: unsupported shape (7x7 / 14x14 maps, k in {3,5}, stride 1, cout <= 320 / 128)");
  FROST_REQUIRE(fin3 && fin3->counter && fin3->coef && fin3->qrec_y && conv_out && stats3, "block_dw_reduce: incomplete arguments");
  BlkBP p = {};
  p.y1 = y1; p.qy1 = qrec_y1; p.wq2 = wq2; p.wsum2 = wsum2; p.coef2 = coef2; p.qy2 = qrec_y2; p.y2 = y2; p.w3 = w3_pack; p.wsum3 = wsum3; p.cint = conv_out;
  p.stats3 = (uint8_t*)stats3; p.fin = *fin3; p.n = n; p.c = c; p.cpad = round_up(c, 16); p.cout = cout; p.cpad3 = round_up(cout, 16); p.nchunk = (c + 63) / 64;
  p.relu2 = relu2;
  static const int imgs_env = getenv("FROST_BLK_IMGS_B") ? atoi(getenv("FROST_BLK_IMGS_B")) : 0;
  int imgs = imgs_env > 0 ? imgs_env : (n + 1023) / 1024;          // the K split keeps an image in one workgroup: one image each up to 1024 workgroups
  if (imgs < 1) imgs = 1;
  p.imgs = imgs;
  hipStream_t s = as_stream(stream);
  const int ct3 = p.cpad3 >> 4;
  if (h == 7) {
    const int nctw = (ct3 + 3) / 4;
    // two chunks per iteration on eight waves (k_blk_dw_reduce2): bit-identical, measured SLOWER (profiles/r05_fusion_ab.txt: 118 / 77 / 157 us against 93 / 63 / 117 with
    // 256 registers and one workgroup per CU, worse with 128 registers and spills) -- off; the four-wave kernel stays
    static const int two = getenv("FROST_BLK_B2") ? atoi(getenv("FROST_BLK_B2")) : 0;
    if (two) {
#define BLK_B2(KK, NN) if (k == KK && nctw <= NN) return launch_blk_b2<KK, NN>(p, s);
      BLK_B2(3, 3) BLK_B2(3, 5) BLK_B2(5, 3) BLK_B2(5, 5)
#undef BLK_B2
    }
#define BLK_B(KK, NN) if (k == KK && nctw <= NN) return launch_blk_b<KK, 7, 4, NN>(p, s);
    BLK_B(3, 3) BLK_B(3, 5) BLK_B(5, 3) BLK_B(5, 5)
#undef BLK_B
  } else {
    if (k == 3) return launch_blk_b<3, 14, 8, 2>(p, s);
    return launch_blk_b<5, 14, 8, 2>(p, s);
  }
  FROST_REQUIRE(false, "block_dw_reduce: no instance");
  return 1;
}

// ================================================================================================ kernel C: depthwise backward -- dc + weight gradient + data gradient
// conv2's backward after its reduce pass (S1 / S2 in the coefficient rows): today three launches (dc pass, weight gradient, data gradient: 12 bytes per
// element through HBM, the dc tensor written once and read twice).  Here a workgroup holds one image's 64-channel chunk: x plane (int8, zero-point halo)
// and gout tile in LDS -> recomputed conv, STE window, dc = fma(gy, K1, fma(acc, E, F)) rounded to bf16 (k_dw3's expressions) -> dc plane in LDS (zero
// halo) -> weight-gradient sums (lane-local, 25 accumulators across the workgroup's images) and the data gradient straight from the plane: the whole map
// is resident, so the data gradient needs no halo recomputation and dc never reaches HBM.  5 bytes per element.
struct BlkCP {
  const int8_t* x; const float* qx;                 // conv2's input (y1) and its record
  const int8_t* wq; const int32_t* wsum; const float* qw; const float* wscale;   // taps [k*k][cpad], weight sums, weight record, per-channel scales (or NULL)
  float* coef; const float* qy;                     // coefficient rows (the reduce pass accumulates S1 / S2 into them; the dc pass reads them), output record
  const uint16_t* gout; uint16_t* dx; float* dwq;   // gradient w.r.t. conv2's output (bf16), w.r.t. its input (bf16, or NULL), raw weight-gradient sums [c][k*k]
  int n, c, cpad, nchunk, imgs, rounds, relu, sr; float inv_count; int xmap;
  // conv1 fold (k_blk_dw_bwd<.., KS1 > 0>): the backward REDUCE pass of the pointwise layer that produced x (conv1 of the bottleneck) rides on the dx values of this
  // kernel -- c1_x = conv1's input (the block-internal cat, [n*map][c1_cin] offset-binary bytes) and its record, conv1's MFMA weight pack / weight sums, and its coefficient
  // rows (A, B, M, R read; S1 / S2 accumulated with float atomics).  conv1's output record is this layer's input record (qx).
  const int8_t* c1_x; const float* c1_qx; const int8_t* c1_w; const int32_t* c1_wsum; float* c1_coef; int c1_cin, c1_kstr, c1_relu;
};

template <int K, int HW, int NW>
struct BlkGeoC : BlkGeo<K, HW, NW> {
  using G = BlkGeo<K, HW, NW>;
  static constexpr int DPL = G::PLANE * 2;                                // dc plane, bf16, same [row][col][64] geometry as the x plane
  static constexpr int GT = (G::MAP + 8) * 64 * 2;                        // gout tile bf16 [pixel][64] (+ slack: the last row's 4-pixel reads run past it)
  static constexpr int RED = NW * K * K * 64 * 4;                         // weight-gradient partials of the waves (reuses the planes after the loop)
  __host__ __device__ static constexpr int lds() { return (G::PLANE + DPL + GT > RED ? G::PLANE + DPL + GT : RED) + 64; }
  // conv1 fold: the fp32 dx values of the image's chunk [pixel][64] live over the x plane + the gout tile (both dead once the dc phase is done; layout x plane | gout tile |
  // dc plane), behind the dc plane: conv1's input rows of the image [NPT*16][kstr] and the chunk's 64 x K weight fragments + per-channel rows (A, B, weight sum)
  static constexpr int GB = G::MAP * 64 * 4;
  static_assert(GB <= G::PLANE + GT, "conv1 fold: the fp32 dx tile must fit over the x plane and the gout tile");
  __host__ __device__ static constexpr int lds_c1(int kstr, int ks1) { return lds() + G::NPT * 16 * kstr + ks1 * 4 * 1024 + 64 * 12; }
};

__device__ __forceinline__ void blk_tr16(const uint8_t* row, int col0, int lane, float* out4) {     // 4 consecutive pixels of channel `lane` of a bf16 [col][64 ch] row
  typedef short v4s_ __attribute__((ext_vector_type(4)));
  const int jp = lane & 15, G = lane >> 4;
  const v4s_ raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_ __attribute__((address_space(3)))*)(row + ((col0 + (jp >> 2)) * 64 + 16 * G + 4 * (jp & 3)) * 2));
  out4[0] = bf2f((uint16_t)raw[0]); out4[1] = bf2f((uint16_t)raw[1]); out4[2] = bf2f((uint16_t)raw[2]); out4[3] = bf2f((uint16_t)raw[3]);
}


// v_dot2c_f32_bf16: acc += a.lo * b.lo + a.hi * b.hi (bf16 pairs, fp32 accumulate): the depthwise backward's stencils at two multiply-adds per VALU instruction
typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float blk_dot2(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf_, a), __builtin_bit_cast(v2bf_, b), acc, false);
}
__device__ __forceinline__ void blk_tr16_raw(const uint8_t* row, int col0, int lane, uint32_t* out2) {     // 4 consecutive pixels of channel `lane` of a bf16 [col][64 ch] row, packed
  typedef short v4s_ __attribute__((ext_vector_type(4)));
  typedef int v2i__ __attribute__((ext_vector_type(2)));
  const int jp = lane & 15, G = lane >> 4;
  const v2i__ raw = __builtin_bit_cast(v2i__, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_ __attribute__((address_space(3)))*)(row + ((col0 + (jp >> 2)) * 64 + 16 * G + 4 * (jp & 3)) * 2)));
  out2[0] = (uint32_t)raw[0]; out2[1] = (uint32_t)raw[1];
}
template <int K, int HW, int NW, int KS1 = 0>
__global__ __launch_bounds__(NW * 64, 2) void k_blk_dw_bwd(const BlkCP p) {
  using G = BlkGeoC<K, HW, NW>;
  constexpr int MAP = G::MAP, PAD = G::PAD, PITCH = G::PITCH, NT = G::NT;
  constexpr bool C1 = KS1 > 0;
  constexpr bool X1_EARLY = (HW == 7);       // conv1 fold: the image's conv1 input rows are prefetched one image ahead with the x / gout rows (7 x 7) or requested under the data gradient (14 x 14)
  constexpr int NPTW = G::NPTW, NPT = G::NPT;
  constexpr int XU = (MAP * 8 + NT - 1) / NT, GU = (MAP * 8 + NT - 1) / NT;       // 8-byte units of x, 16-byte units of gout per thread (8 per pixel each)
  constexpr int X1U = C1 ? (MAP * KS1 * 8 + NT - 1) / NT : 1;                      // conv1 fold: 8-byte units of the image's conv1 input rows per thread
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xpl = smem;                                    // [PHA][PITCH][64] int8
  uint8_t* const gt = smem + G::PLANE;                          // [MAP][64] bf16            (x plane | gout tile: the fold's fp32 dx tile lies over both)
  uint8_t* const dpl = gt + G::GT;                              // [PHA][PITCH][64] bf16
  float* const gb = (float*)smem;                               // conv1 fold: [MAP][64] fp32 dx of the chunk BEFORE its bf16 rounding
  uint8_t* const x1s = smem + G::lds();                         // conv1 fold: [NPT*16][kstr] conv1's input rows of the image
  uint8_t* const w1s = x1s + (C1 ? NPT * 16 * p.c1_kstr : 0);   //             [4 channel tiles][KS1][64 lanes][16 B] conv1 weight fragments of the chunk
  float* const tab1 = (float*)(w1s + KS1 * 4 * 1024);           //             A[64], B[64], weight sum[64] of the chunk's conv1 channels
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = blk_xcd_item((int)blockIdx.x, (int)gridDim.x, p.xmap);
  const int chunk = bi % p.nchunk, ig = bi / p.nchunk;
  const int img_lo = ig * p.imgs, img_hi = min(img_lo + p.imgs, p.n);
  const int ch = chunk * 64 + lane; const bool chok = ch < p.c;

  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]);
  {
    const uint32_t zf = (uint32_t)((zpx - 128) & 255) * 0x01010101u;
    for (int i = tid; i < (G::PLANE >> 4); i += NT) ((uint4*)xpl)[i] = make_uint4(zf, zf, zf, zf);
    for (int i = tid; i < ((G::DPL + G::GT) >> 4); i += NT) ((uint4*)gt)[i] = make_uint4(0, 0, 0, 0);
  }
  // conv1 fold: the chunk's weight fragments and per-channel rows (the same for every image of this workgroup), the rows of the input tile past the map zeroed
  int zpx1 = 0; float y_inv1 = 1.0f, t_lo1 = 0.0f, t_hi1 = 0.0f;
  float c1s1[4] = {0.f, 0.f, 0.f, 0.f}, c1s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (C1) {
    const int CT1 = p.cpad >> 4;
    for (int i = tid; i < KS1 * 4 * 64; i += NT) {
      const int ln = i & 63, ks = (i >> 6) % KS1, wt = i / (64 * KS1);
      const int ct = min(chunk * 4 + wt, CT1 - 1);            // (a partial last chunk repeats the last tile: its channels are masked out below)
      ((uint4*)w1s)[(wt * KS1 + ks) * 64 + ln] = *(const uint4*)(p.c1_w + ((((int64_t)ct * KS1 + ks) * 64 + ln) << 4));
    }
    for (int i = tid; i < 64; i += NT) {
      const int c2 = chunk * 64 + i; const bool ok = c2 < p.c; const int cc = ok ? c2 : 0;
      const float rA = p.c1_coef[FROST_COEF_A * p.cpad + cc], rB = p.c1_coef[FROST_COEF_B * p.cpad + cc]; const int rW = p.c1_wsum[cc];
      tab1[i] = ok ? rA : 0.0f; tab1[64 + i] = ok ? rB : 0.0f; ((int*)tab1)[128 + i] = ok ? rW : 0;
    }
    for (int i = tid; i < ((NPT * 16 * p.c1_kstr) >> 4); i += NT) ((uint4*)x1s)[i] = make_uint4(0, 0, 0, 0);
    zpx1 = __float_as_int(p.c1_qx[FROST_Q_ZP]) - 128;
    y_inv1 = 1.0f / p.qx[FROST_Q_SCALE];                       // conv1's output record IS this layer's input record
    const int zpy1 = __float_as_int(p.qx[FROST_Q_ZP]), qhi1 = q_hi(p.qx);
    const float hi0 = (float)qhi1 + 0.5f - (float)zpy1;
    t_hi1 = ((qhi1 - zpy1) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
    if (!p.c1_relu) { const float lo0 = -(float)zpy1 - 0.5f; t_lo1 = (zpy1 & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  }
  // per-lane (= per-channel) constants
  int wpk[K][2]; float wf[K * K];         // (the taps as floats in LDS instead, to fit 3 waves per SIMD: 28 spilled registers, 20 % slower -- measured)
  int8_t taps[K * K];
  load_taps_i8<K * K>(p.wq, p.cpad, ch, chok, taps);
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int8_t wv = taps[ky * K + kx];
      wf[ky * K + kx] = (float)wv;
      const uint32_t b = (uint32_t)(uint8_t)wv; if (kx < 4) lo |= b << (8 * kx); else hi |= b;
    }
    wpk[ky][0] = (int)lo; wpk[ky][1] = (int)hi;
  }
  // data gradient: taps as bf16 pairs (integers <= 127: exact) in the order the dot product meets them -- pair a covers (kx = 2a + 1, kx = 2a) against the ascending
  // pixel pair (ix + K - 2 - 2a, ix + K - 1 - 2a); the odd tap K - 1 stands alone against (ix, ix + 1) with a zero partner
  constexpr int NPW = (K + 1) / 2;
  uint32_t wp2[K][NPW];
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int a = 0; a < NPW; ++a) {
      const uint32_t hi = __float_as_uint(wf[ky * K + 2 * a]) >> 16;
      const uint32_t lo = (2 * a + 1 < K) ? (__float_as_uint(wf[ky * K + 2 * a + 1]) >> 16) : 0u;
      wp2[ky][a] = (2 * a + 1 < K) ? (lo | (hi << 16)) : hi;               // lone tap: (w, 0) against (p_ix, p_ix+1)
    }
  const int acc0 = chok ? (128 - zpx) * p.wsum[ch] : 0;
  const float sw = (p.wscale && chok) ? p.wscale[ch] : p.qw[FROST_Q_SCALE];
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
  const uint32_t zpfill = (uint32_t)((zpx - 128) & 255) * 0x01010101u;
  float wacc[K * K]; float sdc = 0.0f;
#pragma unroll
  for (int t = 0; t < K * K; ++t) wacc[t] = 0.0f;
  // staging plan (the same for every image): unit u -> pixel u >> 3, 8-byte (x) / 16-byte (gout) part u & 7
  int upx[XU], uoff[XU];
#pragma unroll
  for (int i = 0; i < XU; ++i) {
    const int u = tid + i * NT, px = u >> 3;
    const int r = px / HW, cc = px - r * HW;
    upx[i] = (u < MAP * 8) ? px : -1;
    uoff[i] = ((r + PAD) * PITCH + cc + PAD) * 64;
  }
  const int part = tid & 7;
  const bool pok = (chunk * 64 + part * 8) < p.c;
  uint2 xv[XU]; uint4 gv[GU]; uint2 x1v[X1U];
  const int upr1 = C1 ? (p.c1_cin >> 3) : 1, tot1 = MAP * upr1;
  auto prefetch = [&](int img) __attribute__((always_inline)) {
    const int8_t* xs = p.x + (int64_t)img * MAP * p.c + chunk * 64 + part * 8;
    const uint16_t* gs = p.gout + (int64_t)img * MAP * p.c + chunk * 64 + part * 8;
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      const bool ok = upx[i] >= 0 && pok;
      xv[i] = ok ? *(const uint2*)(xs + (int64_t)upx[i] * p.c) : make_uint2(0, 0);
      gv[i] = ok ? *(const uint4*)(gs + (int64_t)upx[i] * p.c) : make_uint4(0, 0, 0, 0);
    }
    if (C1 && X1_EARLY) {      // conv1's input rows of the image (MAP * c1_cin contiguous bytes) travel with the other prefetches
      const int8_t* s1 = p.c1_x + (int64_t)img * MAP * p.c1_cin;
#pragma unroll
      for (int i = 0; i < X1U; ++i) { const int u = min(tid + i * NT, tot1 - 1); x1v[i] = *(const uint2*)(s1 + (int64_t)u * 8); }
    }
  };
  if (img_lo < img_hi) prefetch(img_lo);
  __syncthreads();

  for (int img = img_lo; img < img_hi; ++img) {
    if (C1 && X1_EARLY) {      // (conv1's input tile is read only by the GEMM phase at the end of an iteration: free here)
#pragma unroll
      for (int i = 0; i < X1U; ++i) { const int u = tid + i * NT; const int row = u / upr1, col = u - row * upr1; if (u < tot1) *(uint2*)(x1s + row * p.c1_kstr + col * 8) = x1v[i]; }
    }
    if (C1 && img > img_lo) {
      // the fp32 dx tile of the previous image lay over the x plane: its zero-point halo comes back (cells outside the map; the interior is stored right below, the
      // two sets of addresses are disjoint, so no barrier between them)
      const uint32_t zf = (uint32_t)((zpx - 128) & 255) * 0x01010101u;
      for (int i = tid; i < G::PHA * PITCH * 4; i += NT) {
        const int cell = i >> 2, r = cell / PITCH, cc = cell - r * PITCH;
        if (!(r >= PAD && r < PAD + HW && cc >= PAD && cc < PAD + HW)) ((uint4*)xpl)[i] = make_uint4(zf, zf, zf, zf);
      }
      for (int i = tid; i < ((G::GT - MAP * 128) >> 4); i += NT) ((uint4*)(gt + MAP * 128))[i] = make_uint4(0, 0, 0, 0);      // the slack rows behind the gout tile
    }
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      if (upx[i] >= 0) { if (pok) *(uint2*)(xpl + uoff[i] + part * 8) = xv[i]; else if (C1) *(uint2*)(xpl + uoff[i] + part * 8) = make_uint2(zpfill, zpfill); *(uint4*)(gt + (upx[i] * 64 + part * 8) * 2) = gv[i]; }
    }
    if (img + 1 < img_hi) prefetch(img + 1);
    blk_barrier();
    // ---- dc of output rows 2w, 2w + 1 (lane = channel) -> dc plane; weight-gradient sums over the same rows
    if (w < G::NRG) {
#pragma unroll 1
      for (int seg = 0; seg < G::NSEG; ++seg) {
        int a[2][8];
        blk_dw_unit<K, PITCH, (HW == 7) ? 7 : 8>(xpl, 2 * w, seg, lane, wpk, acc0, a);
        float dcv[2][8];
        if (HW == 7) { dcv[0][7] = 0.0f; dcv[1][7] = 0.0f; }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          float gq[8];
          const uint8_t* grow = gt + ((2 * w + o) * HW) * 64 * 2;
          blk_tr16(grow, seg * 8, lane, gq); blk_tr16(grow, seg * 8 + 4, lane, gq + 4);
#pragma unroll
          for (int r = 0; r < ((HW == 7) ? 7 : 8); ++r) {
            const bool valid = (2 * w + o) < HW && (seg * 8 + r) < HW && chok;
            const float v = (float)a[o][r];
            const float tq = fmaf(cA, v, cB) * y_inv;
            const float gy = (valid && tq > t_lo && tq <= t_hi) ? gq[r] : 0.0f;
            const float dcf = fmaf(gy, cK1, fmaf(v, cE, cF));
            const uint32_t db = __float_as_uint(dcf), dr = sr_next16(rng);
            const uint32_t hb = (db + (sr_on ? dr : 0x7fffu + ((db >> 16) & 1u))) >> 16;
            if (valid) *(uint16_t*)(dpl + (((2 * w + o + PAD) * PITCH + seg * 8 + r + PAD) * 64 + lane) * 2) = (uint16_t)hb;
            dcv[o][r] = valid ? __uint_as_float(hb << 16) : 0.0f;
            sdc += dcv[o][r];
          }
        }
        // wacc[ky][kx] += dc[o][r] * q[o + ky][r + kx]  (q = unsigned index; the zero point comes off through sdc at the end), two pixels per v_dot2c_f32_bf16:
        // dc pairs (r, r + 1) for even r against the index pairs XP[i] = (q_i, q_i+1) as bf16 (0 .. 255: exact)
        uint32_t D2[2][4];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int r2 = 0; r2 < 4; ++r2) D2[o][r2] = __builtin_amdgcn_perm(__float_as_uint(dcv[o][2 * r2 + 1]), __float_as_uint(dcv[o][2 * r2]), 0x07060302u);   // {hi[31:16], lo[31:16]}
#pragma unroll
        for (int jr = 0; jr < K + 1; ++jr) {
          const uint8_t* rowp = xpl + (2 * w + jr) * PITCH * 64;
          const v2i ra = blk_tr8(rowp, seg * 8, lane), rb = blk_tr8(rowp, seg * 8 + 8, lane);
          const uint32_t u[4] = {(uint32_t)ra[0] ^ 0x80808080u, (uint32_t)ra[1] ^ 0x80808080u, (uint32_t)rb[0] ^ 0x80808080u, (uint32_t)rb[1] ^ 0x80808080u};
          uint32_t xf[12];
#pragma unroll
          for (int i = 0; i < 12; ++i) xf[i] = __float_as_uint((float)((u[i >> 2] >> (8 * (i & 3))) & 255u));
          uint32_t XP[11];
#pragma unroll
          for (int i = 0; i < 11; ++i) XP[i] = __builtin_amdgcn_perm(xf[i + 1], xf[i], 0x07060302u);
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            if ((jr - ky) >= 0 && (jr - ky) < 2) {
              const int o = jr - ky;
#pragma unroll
              for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) wacc[ky * K + kx] = blk_dot2(D2[o][r2], XP[2 * r2 + kx], wacc[ky * K + kx]);
            }
          }
        }
      }
    }
    blk_barrier();
    if (C1 && !X1_EARLY) {      // conv1's input rows of THIS image: requested here, they travel under the data gradient (unconditional loads from clamped
                   // addresses: a branch around a load puts its wait right behind it) and go to LDS before the barrier in front of the GEMM phase.  (14 x 14 form: held
                   // across the whole iteration like the other prefetches they spill; the price is that the wait below also drains this image's dx stores)
      const int8_t* s1 = p.c1_x + (int64_t)img * MAP * p.c1_cin;
#pragma unroll
      for (int i = 0; i < X1U; ++i) { const int u = min(tid + i * NT, tot1 - 1); x1v[i] = *(const uint2*)(s1 + (int64_t)u * 8); }
    }
    // ---- data gradient of input rows 2w, 2w + 1 from the dc plane: dx[iy][ix] = s_w * sum dc[iy + pad - ky][ix + pad - kx] * wq[ky][kx]  (k_dw3_dgrad's order)
    if (p.dx && w < G::NRG) {
#pragma unroll 1
      for (int seg = 0; seg < G::NSEG; ++seg) {
        float acc[2][8];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int r = 0; r < 8; ++r) acc[o][r] = 0.0f;
#pragma unroll
        for (int jr = 0; jr < K + 1; ++jr) {
          // 12 pixels of the dc row as bf16 pairs: P[i] = (p_i, p_i+1) -- even i straight from the transposed read, odd i one v_alignbit
          uint32_t d[6];
          const uint8_t* rowp = dpl + ((2 * w + jr) * PITCH) * 64 * 2;
          blk_tr16_raw(rowp, seg * 8, lane, d); blk_tr16_raw(rowp, seg * 8 + 4, lane, d + 2); blk_tr16_raw(rowp, seg * 8 + 8, lane, d + 4);
          uint32_t P[11];
#pragma unroll
          for (int i = 0; i < 11; ++i) P[i] = (i & 1) ? __builtin_amdgcn_alignbit(d[(i >> 1) + 1], d[i >> 1], 16) : d[i >> 1];
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            const int ky = o + K - 1 - jr;
            if (ky >= 0 && ky < K) {
#pragma unroll
              for (int r = 0; r < ((HW == 7) ? 7 : 8); ++r)
#pragma unroll
                for (int a = 0; a < NPW; ++a) acc[o][r] = blk_dot2(P[(2 * a + 1 < K) ? r + K - 2 - 2 * a : r], wp2[ky][a], acc[o][r]);
            }
          }
        }
        uint16_t* dst = p.dx + (int64_t)img * MAP * p.c + ch;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int r = 0; r < ((HW == 7) ? 7 : 8); ++r)
            if ((2 * w + o) < HW && (seg * 8 + r) < HW) {
              const float dxv = acc[o][r] * sw;
              if (chok) dst[(int64_t)((2 * w + o) * HW + seg * 8 + r) * p.c] = (uint16_t)cvt_pk_bf16(dxv, 0.0f);
              // (the x plane / gout tile are dead: every wave passed the barrier above.  Column XOR-swizzled by the pixel's position in its 16-pixel tile: the GEMM phase
              // reads float4s of one channel quad for 16 consecutive pixels -- 16 rows of 64 floats would all sit on the same four banks)
              if (C1) { const int px = (2 * w + o) * HW + seg * 8 + r; gb[px * 64 + (lane ^ ((px & 15) << 2))] = chok ? dxv : 0.0f; }
            }
      }
    }
    if (C1 && !X1_EARLY) {
#pragma unroll
      for (int i = 0; i < X1U; ++i) { const int u = tid + i * NT; const int row = u / upr1, col = u - row * upr1; if (u < tot1) *(uint2*)(x1s + row * p.c1_kstr + col * 8) = x1v[i]; }
    }
    blk_barrier();
    if (C1) {
      // ---- conv1's reduce pass on this (image, chunk): recompute conv1's integer output of the chunk's 64 channels on the matrix cores (k_blk_expand_dw's GEMM: channel
      // tile = wave & 3, the pixel tiles split over the wave halves), STE window of conv1's output FakeQuantize, S1 += g, S2 += g * acc with g = the fp32 dx value
      // (frost_pw.hip M_BRED's expressions; xhat = acc * R - M * R is applied once, at the end)
      const int ph = (NW == 8) ? (w >> 2) : 0, wct = w & 3;
      const int j = lane & 15, g4 = lane >> 4;
      const int ti = wct * 16 + 4 * g4;
      const float4 A4 = *(const float4*)(tab1 + ti), B4 = *(const float4*)(tab1 + 64 + ti);
      const int4 ws = *(const int4*)((const int*)tab1 + 128 + ti);
      const float Ar[4] = {A4.x, A4.y, A4.z, A4.w}, Br[4] = {B4.x, B4.y, B4.z, B4.w};
      constexpr int TG = (NPTW > 4) ? 4 : NPTW;               // pixel tiles in flight (the 14 x 14 form has 7 per wave: two rounds keep the accumulators at 16 registers)
#pragma unroll
      for (int t0 = 0; t0 < NPTW; t0 += TG) {
        v4i acc1[TG];
#pragma unroll
        for (int t = 0; t < TG; ++t) acc1[t] = (v4i){-zpx1 * ws.x, -zpx1 * ws.y, -zpx1 * ws.z, -zpx1 * ws.w};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const v4i afr = *(const v4i*)(w1s + (((wct * KS1 + ks) * 64 + lane) << 4));
#pragma unroll
          for (int t = 0; t < TG; ++t) {
            if (t0 + t < NPTW) {
              const int prow = min((ph * NPTW + t0 + t) * 16 + j, NPT * 16 - 1);
              const v4i bfr = *(const v4i*)(x1s + prow * p.c1_kstr + ks * 64 + g4 * 16);
              acc1[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr, bfr, acc1[t], 0, 0, 0);          // D[chan][pix]
            }
          }
        }
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          const int pix = (ph * NPTW + t0 + t) * 16 + j;
          if (t0 + t < NPTW && pix < MAP) {
            const float4 gq = *(const float4*)(gb + pix * 64 + (ti ^ ((pix & 15) << 2)));
            const float gr[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = (float)acc1[t][r];
              const float tq = fmaf(Ar[r], v, Br[r]) * y_inv1;
              const float gg = (tq > t_lo1 && tq <= t_hi1) ? gr[r] : 0.0f;
              c1s1[r] += gg; c1s2[r] = fmaf(gg, v, c1s2[r]);
            }
          }
        }
      }
      blk_barrier();          // the dx tile is free: the next image's x plane / gout tile may be stored over it
    }
  }
  if (C1) {
    // conv1's S1 / S2 of the chunk: the 16 pixel lanes of every channel quad fold with DPP-free shuffles, then one pair of float atomics per channel and wave
    // (4 or 8 waves per workgroup: k_pw's reduce pass issues one pair per channel and workgroup, this kernel runs ~22 workgroups per chunk)
    const int wct = w & 3, g4 = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = c1s1[r], b = c1s2[r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      const int c2 = chunk * 64 + wct * 16 + 4 * g4 + r;
      if ((lane & 15) == 0 && c2 < p.c) {
        const float Rv = p.c1_coef[FROST_COEF_R * p.cpad + c2], Mv = p.c1_coef[FROST_COEF_M * p.cpad + c2];
        atomicAdd(s12_dst(p.c1_coef, p.cpad, 0) + c2, a);
        atomicAdd(s12_dst(p.c1_coef, p.cpad, 1) + c2, fmaf(b, Rv, -Mv * Rv * a));
      }
    }
  }
  // ---- dW[c][tap] += s_x * (sum dc*q - zp * sum dc): the waves' partials through LDS, one atomic per (channel, tap) and workgroup
  __syncthreads();
  float* red = (float*)smem;                       // [NW][K*K][64]
  const float zpf = (float)zpx;
#pragma unroll
  for (int t = 0; t < K * K; ++t) red[(w * K * K + t) * 64 + lane] = (w < G::NRG) ? wacc[t] - zpf * sdc : 0.0f;
  __syncthreads();
  const float sx = p.qx[FROST_Q_SCALE];
  for (int i = tid; i < K * K * 64; i += NT) {
    const int t = i >> 6, l2 = i & 63; const int c2 = chunk * 64 + l2;
    float sum = 0.0f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) sum += red[(w2 * K * K + t) * 64 + l2];
    if (c2 < p.c) atomicAdd(dwq_dst(p.dwq, (int64_t)p.c * K * K) + (int64_t)c2 * K * K + t, sum * sx);
  }
}

template <int K, int HW, int NW, int KS1 = 0>
static int launch_blk_c(BlkCP& p, hipStream_t s) {
  using G = BlkGeoC<K, HW, NW>;
  if (KS1 > 0) p.c1_kstr = KS1 * 64 + 16;          // conv1's input rows in LDS: 16 bytes of padding (the K tail of a row reads on into the next: zero weights there)
  const size_t lds = (KS1 > 0) ? (size_t)G::lds_c1(p.c1_kstr, KS1) : (size_t)G::lds();
  FROST_REQUIRE(lds <= 160 * 1024, "block_dw_bwd: LDS budget exceeded");
  static bool attr_set = false; static int occ = 0;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_blk_dw_bwd<K, HW, NW, KS1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_blk_dw_bwd<K, HW, NW, KS1>, NW * 64, lds) != hipSuccess || occ < 1) occ = 1;
  }
  if (p.imgs <= 0) {
    // one round of RESIDENT workgroups (queried, not guessed): a launch of 1.5 rounds costs two -- image groups = floor(slots / chunks), images per group to match
    const int rounds = p.rounds > 0 ? p.rounds : 1;
    int groups = (256 * occ * rounds) / p.nchunk; if (groups < 1) groups = 1; if (groups > p.n) groups = p.n;
    p.imgs = (p.n + groups - 1) / groups;
  }
  p.xmap = blk_xcd_on();
  hipLaunchKernelGGL((k_blk_dw_bwd<K, HW, NW, KS1>), dim3((unsigned)(((p.n + p.imgs - 1) / p.imgs) * p.nchunk)), dim3(NW * 64), lds, s, p);
  return frost_check_launch("block_dw_bwd");
}

extern "C" int frost_block_dw_bwd_supported(int h, int w, int k, int stride, int c) {
  return (h == w && (h == 7 || h == 14) && (k == 3 || k == 5) && stride == 1 && (c % 8) == 0) ? 1 : 0;
}

// conv1 fold: 1 if frost_block_dw_bwd_c1 has an instance for a depthwise layer (h x w, c channels, kernel k, stride 1) behind a pointwise conv1 of c1_cin input channels
extern "C" int frost_block_dw_bwd_c1_ok(int h, int w, int k, int stride, int c, int c1_cin) {
  // bit 0: 7 x 7 maps, bit 1: 14 x 14 maps.  Default 2: inside the captured step (B = 512, interleaved, profiles/r06_conv1_fold_ab.txt) the fold is -0.04 ms at 14 x 14 and +0.14 ms
  // at 7 x 7 -- the image-resident kernel is the loaded resource there (23 chunk iterations per image), the reduce pass it replaces a cheaper kernel for the same sums
  static const int on = getenv("FROST_BLK_C1") ? atoi(getenv("FROST_BLK_C1")) : 2;
  const int ks1 = (c1_cin + 63) / 64;
  return (frost_block_dw_bwd_supported(h, w, k, stride, c) && ((on >> (h == 7 ? 0 : 1)) & 1) && (c1_cin % 8) == 0 && ks1 >= 1 && ks1 <= (h == 7 ? 5 : (k == 5 ? 2 : 3))) ? 1 : 0;      // (the 14 x 14 k = 5 form with three K steps spills)
}
static int blk_dw_bwd_impl(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                           int n, int h, int w, int c, int k, const float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dx,
                           float* dwq, const int8_t* c1_x, const float* c1_qrec_x, const int8_t* c1_wq_pack, const int32_t* c1_wsum, float* c1_coef, int c1_cin, int c1_relu,
                           void* stream);
extern "C" int frost_block_dw_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                                  int n, int h, int w, int c, int k, const float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dx,
                                  float* dwq, void* stream) {
  return blk_dw_bwd_impl(x, qrec_x, wq_pack, wsum, qrec_w, wscale, n, h, w, c, k, coef, qrec_y, relu, gout, dx, dwq, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
}
// The same launch carrying the backward REDUCE pass of the pointwise layer that produced x (conv1 of the bottleneck, frost_pw_conv_bwd pass 0 on (c1_x -> x)): conv1's integer
// output of every (image, 64-channel chunk) is recomputed on the matrix cores from conv1's input rows and its S1 / S2 accumulate into c1_coef from the dx values of this kernel
// BEFORE their bf16 rounding -- conv1's backward then starts at its dc pass (one launch and one read of the expanded gradient tensor less per bottleneck; conv1's dgamma / dbeta
// no longer carry the rounding of dx).  The training half of SURVEY 8(f) N1 for the conv2 -> conv1 boundary of the 14 x 14 / 7 x 7 stages.
extern "C" int frost_block_dw_bwd_c1(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                                     int n, int h, int w, int c, int k, const float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dx,
                                     float* dwq, const int8_t* c1_x, const float* c1_qrec_x, const int8_t* c1_wq_pack, const int32_t* c1_wsum, float* c1_coef, int c1_cin,
                                     int c1_relu, void* stream) {
  FROST_REQUIRE(c1_x && c1_qrec_x && c1_wq_pack && c1_wsum && c1_coef && dx, "block_dw_bwd_c1: incomplete conv1 arguments (the data gradient is required)");
  FROST_REQUIRE(frost_block_dw_bwd_c1_ok(h, w, k, 1, c, c1_cin), "block_dw_bwd_c1: the conv1 fold has no instance for this shape");
  return blk_dw_bwd_impl(x, qrec_x, wq_pack, wsum, qrec_w, wscale, n, h, w, c, k, coef, qrec_y, relu, gout, dx, dwq, c1_x, c1_qrec_x, c1_wq_pack, c1_wsum, c1_coef, c1_cin, c1_relu, stream);
}
static int blk_dw_bwd_impl(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                           int n, int h, int w, int c, int k, const float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dx,
                           float* dwq, const int8_t* c1_x, const float* c1_qrec_x, const int8_t* c1_wq_pack, const int32_t* c1_wsum, float* c1_coef, int c1_cin, int c1_relu,
                           void* stream) {
  FROST_REQUIRE(frost_block_dw_bwd_supported(h, w, k, 1, c), "block_dw_bwd: unsupported shape (7x7 / 14x14 maps, k in {3,5}, stride 1)");
  FROST_REQUIRE(x && wq_pack && wsum && coef && qrec_y && gout && dwq, "block_dw_bwd: incomplete arguments");
  BlkCP p = {};
  p.c1_x = c1_x; p.c1_qx = c1_qrec_x; p.c1_w = c1_wq_pack; p.c1_wsum = c1_wsum; p.c1_coef = c1_coef; p.c1_cin = c1_cin; p.c1_relu = c1_relu;
  p.x = x; p.qx = qrec_x; p.wq = wq_pack; p.wsum = wsum; p.qw = qrec_w; p.wscale = wscale; p.coef = (float*)coef /* read-only in this pass */; p.qy = qrec_y; p.gout = gout; p.dx = dx; p.dwq = dwq;
  p.n = n; p.c = c; p.cpad = round_up(c, 16); p.nchunk = (c + 63) / 64; p.relu = relu; p.sr = frost_sr_enabled();
  p.inv_count = 1.0f / (float)((int64_t)n * h * w);
  static const int imgs_env = getenv("FROST_BLK_IMGS_C") ? atoi(getenv("FROST_BLK_IMGS_C")) : 0;
  static const int wgs_env = getenv("FROST_BLK_WGS_C") ? atoi(getenv("FROST_BLK_WGS_C")) : 0;
  // images per workgroup: the k*k weight-gradient sums stay in registers across them and every workgroup ends with 64 * k*k float atomics, so the launch
  // is sized to ONE round of resident workgroups (2 per CU at ~200 VGPRs), not to many small ones
  static const int rounds_env = getenv("FROST_BLK_ROUNDS_C") ? atoi(getenv("FROST_BLK_ROUNDS_C")) : 0;
  int imgs = imgs_env > 0 ? imgs_env : (wgs_env > 0 ? (int)(((int64_t)n * p.nchunk + wgs_env - 1) / wgs_env) : 0);      // 0: sized from the kernel's residency at launch
  if (imgs > n) imgs = n;
  p.imgs = imgs; p.rounds = rounds_env;
  hipStream_t s = as_stream(stream);
  if (c1_x) {
    const int ks1 = (c1_cin + 63) / 64;
#define BLK_C1(KK, HH, NWW, KS) if (k == KK && h == HH && ks1 == KS) return launch_blk_c<KK, HH, NWW, KS>(p, s);
    BLK_C1(5, 7, 4, 1) BLK_C1(5, 7, 4, 2) BLK_C1(5, 7, 4, 3) BLK_C1(5, 7, 4, 4) BLK_C1(5, 7, 4, 5)
    BLK_C1(3, 7, 4, 1) BLK_C1(3, 7, 4, 2) BLK_C1(3, 7, 4, 3) BLK_C1(3, 7, 4, 4) BLK_C1(3, 7, 4, 5)
    BLK_C1(5, 14, 8, 1) BLK_C1(5, 14, 8, 2)
    BLK_C1(3, 14, 8, 1) BLK_C1(3, 14, 8, 2) BLK_C1(3, 14, 8, 3)
#undef BLK_C1
    FROST_REQUIRE(false, "block_dw_bwd_c1: no instance");
  }
  if (h == 7) return (k == 3) ? launch_blk_c<3, 7, 4>(p, s) : launch_blk_c<5, 7, 4>(p, s);
  return (k == 3) ? launch_blk_c<3, 14, 8>(p, s) : launch_blk_c<5, 14, 8>(p, s);
}

// ================================================================================================ kernel R: depthwise backward, reduce pass
// S1 = sum gy, S2 = sum gy * xhat per channel (k_dw3's reduce pass) in the image-resident scheme of k_blk_dw_bwd: lane = channel, the two sums stay in registers
// across the workgroup's images, ONE pair of float atomics per channel and workgroup.
template <int K, int HW, int NW>
__global__ __launch_bounds__(NW * 64, 4) void k_blk_dw_bred(const BlkCP p) {
  using G = BlkGeoC<K, HW, NW>;
  constexpr int MAP = G::MAP, PAD = G::PAD, PITCH = G::PITCH, NT = G::NT;
  constexpr int XU = (MAP * 8 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xpl = smem;                                    // [PHA][PITCH][64] int8
  uint8_t* const gt = smem + G::PLANE;                          // [MAP][64] bf16
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = blk_xcd_item((int)blockIdx.x, (int)gridDim.x, p.xmap);
  const int chunk = bi % p.nchunk, ig = bi / p.nchunk;
  const int img_lo = ig * p.imgs, img_hi = min(img_lo + p.imgs, p.n);
  const int ch = chunk * 64 + lane; const bool chok = ch < p.c;
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]);
  {
    const uint32_t zf = (uint32_t)((zpx - 128) & 255) * 0x01010101u;
    for (int i = tid; i < (G::PLANE >> 4); i += NT) ((uint4*)xpl)[i] = make_uint4(zf, zf, zf, zf);
    for (int i = tid; i < (G::GT >> 4); i += NT) ((uint4*)gt)[i] = make_uint4(0, 0, 0, 0);
  }
  int wpk[K][2];
  int8_t taps[K * K];
  load_taps_i8<K * K>(p.wq, p.cpad, ch, chok, taps);
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) { const uint32_t b = (uint32_t)(uint8_t)taps[ky * K + kx]; if (kx < 4) lo |= b << (8 * kx); else hi |= b; }
    wpk[ky][0] = (int)lo; wpk[ky][1] = (int)hi;
  }
  const int acc0 = chok ? (128 - zpx) * p.wsum[ch] : 0;
  float cA = 0, cB = 0, cR = 0, cMR = 0;
  if (chok) {
    cA = p.coef[FROST_COEF_A * p.cpad + ch]; cB = p.coef[FROST_COEF_B * p.cpad + ch];
    cR = p.coef[FROST_COEF_R * p.cpad + ch]; cMR = -p.coef[FROST_COEF_M * p.cpad + ch] * cR;
  }
  const float y_inv = 1.0f / p.qy[FROST_Q_SCALE];
  float t_lo = 0.0f, t_hi;
  {
    const int zpy = __float_as_int(p.qy[FROST_Q_ZP]), qhi = q_hi(p.qy);
    const float hi0 = (float)qhi + 0.5f - (float)zpy;
    t_hi = ((qhi - zpy) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
    if (!p.relu) { const float lo0 = -(float)zpy - 0.5f; t_lo = (zpy & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  }
  float r1 = 0.0f, r2 = 0.0f;
  int upx[XU], uoff[XU];
#pragma unroll
  for (int i = 0; i < XU; ++i) {
    const int u = tid + i * NT, px = u >> 3;
    const int r = px / HW, cc = px - r * HW;
    upx[i] = (u < MAP * 8) ? px : -1;
    uoff[i] = ((r + PAD) * PITCH + cc + PAD) * 64;
  }
  const int part = tid & 7;
  const bool pok = (chunk * 64 + part * 8) < p.c;
  uint2 xv[XU]; uint4 gv[XU];
  auto prefetch = [&](int img) __attribute__((always_inline)) {
    const int8_t* xs = p.x + (int64_t)img * MAP * p.c + chunk * 64 + part * 8;
    const uint16_t* gs = p.gout + (int64_t)img * MAP * p.c + chunk * 64 + part * 8;
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      const bool ok = upx[i] >= 0 && pok;
      xv[i] = ok ? *(const uint2*)(xs + (int64_t)upx[i] * p.c) : make_uint2(0, 0);
      gv[i] = ok ? *(const uint4*)(gs + (int64_t)upx[i] * p.c) : make_uint4(0, 0, 0, 0);
    }
  };
  if (img_lo < img_hi) prefetch(img_lo);
  __syncthreads();
  for (int img = img_lo; img < img_hi; ++img) {
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      if (upx[i] >= 0) { if (pok) *(uint2*)(xpl + uoff[i] + part * 8) = xv[i]; *(uint4*)(gt + (upx[i] * 64 + part * 8) * 2) = gv[i]; }
    }
    if (img + 1 < img_hi) prefetch(img + 1);
    blk_barrier();
    if (w < G::NRG) {
#pragma unroll 1
      for (int seg = 0; seg < G::NSEG; ++seg) {
        int a[2][8];
        blk_dw_unit<K, PITCH, (HW == 7) ? 7 : 8>(xpl, 2 * w, seg, lane, wpk, acc0, a);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          float gq[8];
          const uint8_t* grow = gt + ((2 * w + o) * HW) * 64 * 2;
          blk_tr16(grow, seg * 8, lane, gq); blk_tr16(grow, seg * 8 + 4, lane, gq + 4);
#pragma unroll
          for (int r = 0; r < ((HW == 7) ? 7 : 8); ++r) {
            const bool valid = (2 * w + o) < HW && (seg * 8 + r) < HW && chok;
            const float v = (float)a[o][r];
            const float tq = fmaf(cA, v, cB) * y_inv;
            const float gy = (valid && tq > t_lo && tq <= t_hi) ? gq[r] : 0.0f;
            r1 += gy; r2 = fmaf(gy, fmaf(v, cR, cMR), r2);
          }
        }
      }
    }
    blk_barrier();
  }
  __syncthreads();
  float* red = (float*)smem;                       // [NW][2][64]
  red[(w * 2) * 64 + lane] = r1; red[(w * 2 + 1) * 64 + lane] = r2;
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6, l2 = tid & 63, c2 = chunk * 64 + l2;
    float sum = 0.0f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) sum += red[(w2 * 2 + which) * 64 + l2];
    if (c2 < p.c) atomicAdd(s12_dst(p.coef, p.cpad, which) + c2, sum);
  }
}

// ================================================================================================ kernel S: depthwise forward, statistics pass
// conv2's exact integer statistics (sum, sum of squares, min, max of the conv output per channel: k_dw3's statistics pass) in the image-resident scheme of the
// backward kernels: lane = channel, the four values stay in registers across the workgroup's images, one set of atomics per channel and workgroup, finalize folded
// into the last workgroup.  With frost_pwc_conv_fwd_emit in front it is the two-launch alternative to k_blk_expand_dw (y1 makes one HBM round trip more, both
// kernels are simpler: no chunk-range re-staging of x, no per-chunk barriers around a GEMM).
template <int K, int HW, int NW>
__global__ __launch_bounds__(NW * 64, 4) void k_blk_dw_stats(const BlkCP p, uint8_t* __restrict__ stats, const FrostFinDesc fin) {
  using G = BlkGeoC<K, HW, NW>;
  constexpr int MAP = G::MAP, PAD = G::PAD, PITCH = G::PITCH, NT = G::NT;
  constexpr int XU = (MAP * 8 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xpl = smem;                                    // [PHA][PITCH][64] int8
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = blk_xcd_item((int)blockIdx.x, (int)gridDim.x, p.xmap);
  const int chunk = bi % p.nchunk, ig = bi / p.nchunk;
  const int img_lo = ig * p.imgs, img_hi = min(img_lo + p.imgs, p.n);
  const int ch = chunk * 64 + lane; const bool chok = ch < p.c;
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]);
  {
    const uint32_t zf = (uint32_t)((zpx - 128) & 255) * 0x01010101u;
    for (int i = tid; i < (G::PLANE >> 4); i += NT) ((uint4*)xpl)[i] = make_uint4(zf, zf, zf, zf);
  }
  int wpk[K][2];
  int8_t taps[K * K];
  load_taps_i8<K * K>(p.wq, p.cpad, ch, chok, taps);
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) { const uint32_t b = (uint32_t)(uint8_t)taps[ky * K + kx]; if (kx < 4) lo |= b << (8 * kx); else hi |= b; }
    wpk[ky][0] = (int)lo; wpk[ky][1] = (int)hi;
  }
  const int acc0 = chok ? (128 - zpx) * p.wsum[ch] : 0;
  long long st1 = 0; double st2 = 0.0; int smn = INT32_MAX, smx = INT32_MIN;
  int upx[XU], uoff[XU];
#pragma unroll
  for (int i = 0; i < XU; ++i) {
    const int u = tid + i * NT, px = u >> 3;
    const int r = px / HW, cc = px - r * HW;
    upx[i] = (u < MAP * 8) ? px : -1;
    uoff[i] = ((r + PAD) * PITCH + cc + PAD) * 64;
  }
  const int part = tid & 7;
  const bool pok = (chunk * 64 + part * 8) < p.c;
  uint2 xv[XU];
  auto prefetch = [&](int img) __attribute__((always_inline)) {
    const int8_t* xs = p.x + (int64_t)img * MAP * p.c + chunk * 64 + part * 8;
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      const bool ok = upx[i] >= 0 && pok;
      xv[i] = ok ? *(const uint2*)(xs + (int64_t)upx[i] * p.c) : make_uint2(0, 0);
    }
  };
  if (img_lo < img_hi) prefetch(img_lo);
  __syncthreads();
  for (int img = img_lo; img < img_hi; ++img) {
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      if (upx[i] >= 0 && pok) *(uint2*)(xpl + uoff[i] + part * 8) = xv[i];
    }
    if (img + 1 < img_hi) prefetch(img + 1);
    blk_barrier();
    if (w < G::NRG) {
#pragma unroll 1
      for (int seg = 0; seg < G::NSEG; ++seg) {
        int a[2][8];
        blk_dw_unit<K, PITCH, (HW == 7) ? 7 : 8>(xpl, 2 * w, seg, lane, wpk, acc0, a);
        int t1 = 0; double t2 = 0.0;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int r = 0; r < ((HW == 7) ? 7 : 8); ++r) {
            if ((2 * w + o) < HW && (seg * 8 + r) < HW) {
              const int vi = a[o][r]; const float v = (float)vi;
              t1 += vi; t2 = fma((double)v, (double)v, t2); smn = min(smn, vi); smx = max(smx, vi);
            }
          }
        st1 += t1; st2 += t2;
      }
    }
    blk_barrier();
  }
  __syncthreads();
  // the waves' lane-local partials through LDS, one set of global atomics per channel and workgroup; then the folded finalize
  double* red_d = (double*)smem;                    // [NW][2][64]
  int* red_i = (int*)(red_d + NW * 2 * 64);         // [NW][2][64]
  red_d[(w * 2) * 64 + lane] = (double)st1; red_d[(w * 2 + 1) * 64 + lane] = st2; red_i[(w * 2) * 64 + lane] = smn; red_i[(w * 2 + 1) * 64 + lane] = smx;
  __syncthreads();
  if (tid < 64) {
    const int c2 = chunk * 64 + tid;
    double a1 = 0, a2 = 0; int mn = INT32_MAX, mx = INT32_MIN;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) { a1 += red_d[(w2 * 2) * 64 + tid]; a2 += red_d[(w2 * 2 + 1) * 64 + tid]; mn = min(mn, red_i[(w2 * 2) * 64 + tid]); mx = max(mx, red_i[(w2 * 2 + 1) * 64 + tid]); }
    if (c2 < p.c && mn <= mx) {
      long long* g_s1 = (long long*)stats_copy(stats, p.cpad); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
      int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
      atomicAdd((unsigned long long*)&g_s1[c2], (unsigned long long)(long long)a1); atomicAdd(&g_s2[c2], (unsigned long long)a2);
      atomicMin(&g_mn[c2], mn); atomicMax(&g_mx[c2], mx);
    }
  }
  int* sflag = (int*)(smem + NW * 2 * 64 * 12);
  if (last_block_done2(fin.counter, gridDim.x, sflag)) {
    float* sh = (float*)smem;
    conv_finalize_dev(stats, (int64_t)p.n * MAP, p.c, p.cpad, p.qx, fin.qrec_w, fin.wscale, fin.gamma, fin.beta, fin.rmean, fin.rvar, fin.nbt, fin.training, fin.relu, fin.observe, 1,
                      fin.coef, fin.qrec_y, tid, NT, sh);
  }
}

template <int K, int HW, int NW>
static int launch_blk_s(BlkCP& p, uint8_t* stats, const FrostFinDesc& fin, hipStream_t s) {
  using G = BlkGeoC<K, HW, NW>;
  const size_t lds = (size_t)(G::PLANE > NW * 2 * 64 * 12 + 64 ? G::PLANE : NW * 2 * 64 * 12 + 64) + 64;
  static bool attr_set = false; static int occ = 0;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_blk_dw_stats<K, HW, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_blk_dw_stats<K, HW, NW>, NW * 64, lds) != hipSuccess || occ < 1) occ = 1;
  }
  int groups = (256 * occ) / p.nchunk; if (groups < 1) groups = 1; if (groups > p.n) groups = p.n;       // one round of resident workgroups
  p.imgs = (p.n + groups - 1) / groups;
  p.xmap = blk_xcd_on();
  hipLaunchKernelGGL((k_blk_dw_stats<K, HW, NW>), dim3((unsigned)(((p.n + p.imgs - 1) / p.imgs) * p.nchunk)), dim3(NW * 64), lds, s, p, stats, fin);
  return frost_check_launch("block_dw_stats");
}

// conv2's statistics pass + folded finalize on 14x14 / 7x7 maps (the contract of frost_dw_conv_fwd_fin)
extern "C" int frost_block_dw_stats(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h, int w, int c, int k, void* stats,
                                    const FrostFinDesc* fin, void* stream) {
  FROST_REQUIRE((h == w && (h == 7 || h == 14) && (k == 3 || k == 5) && (c % 8) == 0), "block_dw_stats: unsupported shape (7x7 / 14x14 maps, k in {3,5}, stride 1)");
  FROST_REQUIRE(fin && fin->counter && fin->coef && fin->qrec_y && stats, "block_dw_stats: incomplete finalize descriptor");
  BlkCP p = {};
  p.x = x; p.qx = qrec_x; p.wq = wq_pack; p.wsum = wsum; p.n = n; p.c = c; p.cpad = round_up(c, 16); p.nchunk = (c + 63) / 64;
  hipStream_t s = as_stream(stream);
  if (h == 7) return (k == 3) ? launch_blk_s<3, 7, 4>(p, (uint8_t*)stats, *fin, s) : launch_blk_s<5, 7, 4>(p, (uint8_t*)stats, *fin, s);
  return (k == 3) ? launch_blk_s<3, 14, 8>(p, (uint8_t*)stats, *fin, s) : launch_blk_s<5, 14, 8>(p, (uint8_t*)stats, *fin, s);
}

template <int K, int HW, int NW>
static int launch_blk_r(BlkCP& p, hipStream_t s) {
  using G = BlkGeoC<K, HW, NW>;
  const size_t lds = (size_t)(G::PLANE + G::GT + 64);
  static bool attr_set = false; static int occ = 0;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_blk_dw_bred<K, HW, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_blk_dw_bred<K, HW, NW>, NW * 64, lds) != hipSuccess || occ < 1) occ = 1;
  }
  if (p.imgs <= 0) {       // one round of resident workgroups (see launch_blk_c)
    const int rounds = p.rounds > 0 ? p.rounds : 1;
    int groups = (256 * occ * rounds) / p.nchunk; if (groups < 1) groups = 1; if (groups > p.n) groups = p.n;
    p.imgs = (p.n + groups - 1) / groups;
  }
  p.xmap = blk_xcd_on();
  hipLaunchKernelGGL((k_blk_dw_bred<K, HW, NW>), dim3((unsigned)(((p.n + p.imgs - 1) / p.imgs) * p.nchunk)), dim3(NW * 64), lds, s, p);
  return frost_check_launch("block_dw_bwd_reduce");
}

extern "C" int frost_block_dw_bwd_reduce(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h, int w, int c, int k,
                                         float* coef, const float* qrec_y, int relu, const uint16_t* gout, void* stream) {
  FROST_REQUIRE((h == w && (h == 7 || h == 14) && (k == 3 || k == 5) && (c % 8) == 0), "block_dw_bwd_reduce: unsupported shape (7x7 / 14x14 maps, k in {3,5}, stride 1)");
  FROST_REQUIRE(x && wq_pack && wsum && coef && qrec_y && gout, "block_dw_bwd_reduce: incomplete arguments");
  BlkCP p = {};
  p.x = x; p.qx = qrec_x; p.wq = wq_pack; p.wsum = wsum; p.coef = coef; p.qy = qrec_y; p.gout = gout;
  p.n = n; p.c = c; p.cpad = round_up(c, 16); p.nchunk = (c + 63) / 64; p.relu = relu;
  static const int wgs_env = getenv("FROST_BLK_WGS_R") ? atoi(getenv("FROST_BLK_WGS_R")) : 0;
  static const int rounds_env = getenv("FROST_BLK_ROUNDS_R") ? atoi(getenv("FROST_BLK_ROUNDS_R")) : 2;      // two rounds of resident workgroups: -0.04 ms in the step (round 6 re-sweep, profiles/r06_sweep.txt; 3: the same)
  int imgs = wgs_env > 0 ? (int)(((int64_t)n * p.nchunk + wgs_env - 1) / wgs_env) : 0;       // 0: sized from the kernel's residency at launch
  if (imgs > n) imgs = n;
  p.imgs = imgs; p.rounds = rounds_env;
  hipStream_t s = as_stream(stream);
  if (h == 7) return (k == 3) ? launch_blk_r<3, 7, 4>(p, s) : launch_blk_r<5, 7, 4>(p, s);
  return (k == 3) ? launch_blk_r<3, 14, 8>(p, s) : launch_blk_r<5, 14, 8>(p, s);
}

extern "C" int frost_block_supported(int h, int w, int k, int stride, int cin, int c) {
  return (h == w && (h == 7 || h == 14) && (k == 3 || k == 5) && stride == 1 && (cin % 8) == 0 && cin > (h == 7 ? 64 : 0) && cin <= (h == 7 ? 320 : 192) && (c % 8) == 0) ? 1 : 0;
}

extern "C" int frost_block_expand_dw_stats(const int8_t* x, const float* qrec_x, const int8_t* w1_pack, const int32_t* wsum1, const float* coef1,
                                           const float* qrec_y1, int8_t* y1, int n, int h, int w, int cin, int c, const int8_t* wq2,
                                           const int32_t* wsum2, int k, void* stats2, const FrostFinDesc* fin2, void* stream) {
  FROST_REQUIRE(frost_block_supported(h, w, k, 1, cin, c), "block_expand_dw: unsupported shape (7x7 / 14x14 maps, k in {3,5}, stride 1, cin <= 320 / 192)");
  FROST_REQUIRE(fin2 && fin2->counter && fin2->coef && fin2->qrec_y, "block_expand_dw: incomplete finalize descriptor");
  BlkAP p = {};
  p.x = x; p.qx = qrec_x; p.w1 = w1_pack; p.wsum1 = wsum1; p.coef1 = coef1; p.qy1 = qrec_y1; p.y1 = y1; p.wq2 = wq2; p.wsum2 = wsum2;
  p.stats2 = (uint8_t*)stats2; p.fin = *fin2; p.n = n; p.cin = cin; p.c = c; p.cpad = round_up(c, 16); p.KS = (cin + 63) / 64; p.kstr = p.KS * 64 + 16;
  p.nchunk = (c + 63) / 64;
  // work split: a workgroup takes `imgs` images x a range of channel chunks.  Per-channel state of the range stays in LDS across the images and is
  // flushed once (4 atomics per channel per workgroup); the split keeps >= ~4 workgroups per CU in flight at the training batch.
  static const int cpw_env = getenv("FROST_BLK_CPW") ? atoi(getenv("FROST_BLK_CPW")) : 0, imgs_env = getenv("FROST_BLK_IMGS") ? atoi(getenv("FROST_BLK_IMGS")) : 0;
  int cpw = cpw_env > 0 ? cpw_env : 2;                 // chunks per workgroup
  if (cpw > p.nchunk) cpw = p.nchunk;
  p.csplit = (p.nchunk + cpw - 1) / cpw;
  static const int rounds_env = getenv("FROST_BLK_ROUNDS_A") ? atoi(getenv("FROST_BLK_ROUNDS_A")) : 0;
  p.imgs = imgs_env > 0 ? imgs_env : 0;          // 0: sized from the kernel's residency at launch (whole rounds of resident workgroups)
  p.rounds = rounds_env > 0 ? rounds_env : (h == 14 ? 2 : 1);      // measured: two rounds at 14 x 14 (72 vs 82 us), one at 7 x 7
  hipStream_t s = as_stream(stream);
#define BLK_A(KK, HH, NWW, KSS) if (k == KK && h == HH && p.KS == KSS) return launch_blk_a<KK, HH, NWW, KSS>(p, s);
  BLK_A(3, 7, 4, 2) BLK_A(3, 7, 4, 3) BLK_A(3, 7, 4, 4) BLK_A(3, 7, 4, 5) BLK_A(5, 7, 4, 2) BLK_A(5, 7, 4, 3) BLK_A(5, 7, 4, 4) BLK_A(5, 7, 4, 5)
  BLK_A(3, 14, 8, 1) BLK_A(3, 14, 8, 2) BLK_A(3, 14, 8, 3) BLK_A(5, 14, 8, 1) BLK_A(5, 14, 8, 2) BLK_A(5, 14, 8, 3)
#undef BLK_A
  FROST_REQUIRE(false, "block_expand_dw: no instance");
  return 1;
}

// ================================================================================================ frost_block_fwd / frost_block_bwd: host composites
extern "C" int frost_pw_conv_fwd_fin(const int8_t*, const float*, const int8_t*, const int32_t*, int64_t, int, int, void*, const FrostFinDesc*, void*);
extern "C" int frost_pw_ew(const int32_t*, int64_t, int, float*, const float*, int, int, const uint16_t*, void*, void*);
extern "C" int frost_pw_dgrad_wide(const uint16_t*, const uint16_t*, const float*, int64_t, int, int, uint16_t*, int, void*);
extern "C" int frost_pw_wgrad(const uint16_t*, const int8_t*, const float*, int64_t, int, int, float*, void*);
extern "C" int frost_pw_conv_bwd(const int8_t*, const float*, const int8_t*, const int32_t*, const uint16_t*, const float*, int64_t, int, int, int, float*, const float*, int,
                                 const uint16_t*, uint16_t*, uint16_t*, int, void*);
extern "C" int frost_pwc_bwd_ok(int64_t, int, int);
extern "C" int frost_pwc_conv_bwd(const int8_t*, const float*, const int8_t*, const int32_t*, int64_t, int, int, int, float*, const float*, int, const uint16_t*, uint16_t*, void*);

static int blk_desc_check(const FrostBlockDesc* d) {
  FROST_REQUIRE(d && d->x && d->y1 && d->y2 && d->y3 && d->conv_out3, "block: incomplete descriptor");
  FROST_REQUIRE(d->conv2.cout == d->conv1.cout && d->conv1.k == 1 && d->reduce.k == 1, "block: conv1 / reduce_conv are 1x1, conv2 is depthwise over conv1's channels");
  FROST_REQUIRE(frost_block_supported(d->h, d->w, d->conv2.k, 1, d->cin, d->conv1.cout) && frost_block_dw_reduce_supported(d->h, d->w, d->conv2.k, 1, d->conv1.cout, d->reduce.cout)
                && d->reduce.cout < d->conv1.cout, "block: unsupported shape (14x14 / 7x7 maps, depthwise stride 1, reduce_conv narrower than the expanded tensor)");
  return 0;
}

extern "C" int frost_block_fwd(const FrostBlockDesc* d, void* stream) {
  if (int rc = blk_desc_check(d)) return rc;
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  const FrostBlockLayer &l1 = d->conv1, &l2 = d->conv2, &l3 = d->reduce;
  int rc = frost_pw_conv_fwd_fin(d->x, d->qrec_x, l1.wq_pack, l1.wsum, npix, d->cin, l1.cout, l1.stats, &l1.fin, stream);
  if (rc) return rc;
  rc = frost_block_expand_dw_stats(d->x, d->qrec_x, l1.wq_pack, l1.wsum, l1.fin.coef, l1.fin.qrec_y, d->y1, d->n, d->h, d->w, d->cin, l1.cout, l2.wq_pack, l2.wsum, l2.k, l2.stats,
                                   &l2.fin, stream);
  if (rc) return rc;
  rc = frost_block_dw_reduce(d->y1, l1.fin.qrec_y, l2.wq_pack, l2.wsum, l2.fin.coef, l2.fin.qrec_y, l2.fin.relu, d->y2, d->n, d->h, d->w, l1.cout, l2.k, l3.wq_pack, l3.wsum, l3.cout,
                             d->conv_out3, l3.stats, &l3.fin, stream);
  if (rc) return rc;
  return frost_pw_ew(d->conv_out3, npix, l3.cout, l3.fin.coef, l3.fin.qrec_y, l3.fin.relu, 2, nullptr, d->y3, stream);
}

extern "C" int frost_block_bwd(const FrostBlockDesc* d, const FrostBlockBwd* b, void* stream) {
  if (int rc = blk_desc_check(d)) return rc;
  FROST_REQUIRE(b && b->gout3 && b->dc3 && b->g2 && b->g1 && b->dc1, "block_bwd: incomplete gradient buffers");
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  const FrostBlockLayer &l1 = d->conv1, &l2 = d->conv2, &l3 = d->reduce;
  hipStream_t ms = as_stream(stream), ss = b->side_stream ? as_stream(b->side_stream) : ms;
  hipEvent_t ev = nullptr;
  auto fork = [&]() { if (ss != ms) { if (!ev) hipEventCreateWithFlags(&ev, hipEventDisableTiming); hipEventRecord(ev, ms); hipStreamWaitEvent(ss, ev, 0); } };
  int rc;
  // reduce_conv (kept integer conv output: element-wise reduce and dc)
  if ((rc = frost_pw_ew(d->conv_out3, npix, l3.cout, l3.fin.coef, l3.fin.qrec_y, l3.fin.relu, 0, b->gout3, nullptr, stream))) return rc;
  if ((rc = frost_pw_ew(d->conv_out3, npix, l3.cout, l3.fin.coef, l3.fin.qrec_y, l3.fin.relu, 1, b->gout3, b->dc3, stream))) return rc;
  fork();
  if ((rc = frost_pw_dgrad_wide(b->dc3, l3.wt_pack, l3.fin.qrec_w, npix, l1.cout, l3.cout, b->g2, 0, stream))) return rc;
  if ((rc = frost_pw_wgrad(b->dc3, d->y2, l2.fin.qrec_y, npix, l1.cout, l3.cout, l3.dwq, (void*)ss))) return rc;
  // conv2 (depthwise): reduce pass, then dc + weight gradient + data gradient in one launch
  if ((rc = frost_block_dw_bwd_reduce(d->y1, l1.fin.qrec_y, l2.wq_pack, l2.wsum, d->n, d->h, d->w, l1.cout, l2.k, l2.fin.coef, l2.fin.qrec_y, l2.fin.relu, b->g2, stream))) return rc;
  if ((rc = frost_block_dw_bwd(d->y1, l1.fin.qrec_y, l2.wq_pack, l2.wsum, l2.fin.qrec_w, l2.fin.wscale, d->n, d->h, d->w, l1.cout, l2.k, l2.fin.coef, l2.fin.qrec_y, l2.fin.relu,
                               b->g2, b->g1, l2.dwq, stream))) return rc;
  // conv1
  const bool pwc = frost_pwc_bwd_ok(npix, d->cin, l1.cout) != 0;
  if (pwc && npix <= 131072) rc = frost_pwc_conv_bwd(d->x, d->qrec_x, l1.wq_pack, l1.wsum, npix, d->cin, l1.cout, 0, l1.fin.coef, l1.fin.qrec_y, l1.fin.relu, b->g1, nullptr, stream);
  else rc = frost_pw_conv_bwd(d->x, d->qrec_x, l1.wq_pack, l1.wsum, l1.wt_pack, l1.fin.qrec_w, npix, d->cin, l1.cout, 0, l1.fin.coef, l1.fin.qrec_y, l1.fin.relu, b->g1, nullptr, nullptr, 0, stream);
  if (rc) return rc;
  if (pwc) rc = frost_pwc_conv_bwd(d->x, d->qrec_x, l1.wq_pack, l1.wsum, npix, d->cin, l1.cout, 1, l1.fin.coef, l1.fin.qrec_y, l1.fin.relu, b->g1, b->dc1, stream);
  else rc = frost_pw_conv_bwd(d->x, d->qrec_x, l1.wq_pack, l1.wsum, l1.wt_pack, l1.fin.qrec_w, npix, d->cin, l1.cout, 1, l1.fin.coef, l1.fin.qrec_y, l1.fin.relu, b->g1, b->dc1, nullptr, 0, stream);
  if (rc) return rc;
  fork();
  if (b->dx && (rc = frost_pw_dgrad_wide(b->dc1, l1.wt_pack, l1.fin.qrec_w, npix, d->cin, l1.cout, b->dx, 0, stream))) return rc;
  if ((rc = frost_pw_wgrad(b->dc1, d->x, d->qrec_x, npix, d->cin, l1.cout, l1.dwq, (void*)ss))) return rc;
  if (ss != ms) {          // join: what follows on the main stream may reuse dc1 / dc3
    hipEventRecord(ev, ss); hipStreamWaitEvent(ms, ev, 0);
  }
  if (ev) hipEventDestroy(ev);
  return frost_check_launch("block_bwd");
}

// ================================================================================================ squeeze_conv emit + cat in one launch
// out = quant_cat.cat([squeeze_conv(x), x], 1) (frostnet.py:127-129): the emit pass of the squeeze conv (k_pw's expression on the int8 MFMA) and the cat's
// requantisation of BOTH halves (k_cat_requant's two 256-entry tables) from one staged x tile -- the squeezed activation and the input are not re-read, one launch
// instead of two per CAS bottleneck.  The cat's FakeQuantize record must already be final (it is updated in the squeeze's statistics / finalize tail).
// Results are bit-identical to frost_pw_conv_fwd(mode 1) + frost_cat_requant.
struct SqCatP {
  const int8_t* x; const float* qx; const int8_t* w; const int32_t* wsum; const float* coef; const float* qsq; const float* qcat;
  int8_t* ysq; int8_t* ycat; int64_t npix; int cin, r, cpad, kstr;
  int cvt;       // 0: fake-quant emit q = rint(fma(A, acc, B) / s_y) + zp;  1: converted QNNPACK form q = rint(float(acc + b_q) * A) + zp (row B = int32 bias bits);
};               // 2: converted FBGEMM form q = rint((float(acc) + B) * A) + zp   (k_pw's emit flavours, csrc/frost_pw.hip)
template <int KSM>
__global__ __launch_bounds__(256) void k_sq_emit_cat(const SqCatP p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xs = smem;                          // [128][kstr]
  uint8_t* const lut = smem + 128 * p.kstr;          // [2][256]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int64_t p0 = (int64_t)blockIdx.x * 128;
  {
    const QP A = load_qp(p.qsq), B = load_qp(p.qx), Y = load_qp(p.qcat);
    const int q = (int)(int8_t)tid + 128;
    lut[tid] = (uint8_t)((fq_index((float)(q - A.zp) * A.scale, Y.inv, Y.zp, 0, Y.hi) - 128) & 255);
    lut[256 + tid] = (uint8_t)((fq_index((float)(q - B.zp) * B.scale, Y.inv, Y.zp, 0, Y.hi) - 128) & 255);
  }
  {
    const int upr = p.cin >> 3; const int total = 128 * upr;
    for (int u = tid; u < total; u += 256) {
      const int row = u / upr, col = u - row * upr;
      const int64_t px = p0 + row;
      uint2 v = make_uint2(0, 0);
      if (px < p.npix) v = *(const uint2*)(p.x + px * p.cin + col * 8);
      *(uint2*)(xs + row * p.kstr + col * 8) = v;
    }
  }
  __syncthreads();
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  const int zps = __float_as_int(p.qsq[FROST_Q_ZP]);
  const float y_inv = 1.0f / p.qsq[FROST_Q_SCALE], y_zpf = (float)zps;
  const float qcap = (float)q_hi(p.qsq); const bool lowq = qcap < 255.0f;
  const int cy = p.r + p.cin;
  const int CT = p.cpad >> 4;
  for (int ct = 0; ct < CT; ++ct) {
    const int ch = ct * 16 + 4 * g;
    v4i afr[KSM];
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) afr[ks] = *(const v4i*)(p.w + ((((int64_t)ct * KSM + ks) * 64 + lane) << 4));
    const int4 ws = *(const int4*)(p.wsum + ch);
    const float4 A4 = *(const float4*)(p.coef + FROST_COEF_A * p.cpad + ch), B4 = *(const float4*)(p.coef + FROST_COEF_B * p.cpad + ch);
    const float A[4] = {A4.x, A4.y, A4.z, A4.w}, B[4] = {B4.x, B4.y, B4.z, B4.w};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = (wv * 2 + t) * 16 + j;
      v4i acc = (v4i){-zpx * ws.x, -zpx * ws.y, -zpx * ws.z, -zpx * ws.w};
#pragma unroll
      for (int ks = 0; ks < KSM; ++ks) {
        const v4i bfr = *(const v4i*)(xs + row * p.kstr + ks * 64 + g * 16);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[ks], bfr, acc, 0, 0, 0);
      }
      uint32_t packed = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float qv;
        if (p.cvt == 2) qv = rintf(((float)acc[r] + B[r]) * A[r]) + y_zpf;
        else {
          const float yv = (p.cvt == 1) ? fmaf(A[r], (float)(acc[r] + __float_as_int(B[r])), 0.0f) : fmaf(A[r], (float)acc[r], B[r]);
          qv = rintf(yv * (p.cvt == 1 ? 1.0f : y_inv)) + y_zpf;
        }
        if (lowq) qv = fminf(qv, qcap);
        packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);
      }
      packed ^= 0x80808080u;
      const int64_t px = p0 + row;
      if (px < p.npix && ch < p.r) {
        *(uint32_t*)(p.ysq + px * p.r + ch) = packed;
        const uint32_t o = (uint32_t)lut[packed & 255] | ((uint32_t)lut[(packed >> 8) & 255] << 8) | ((uint32_t)lut[(packed >> 16) & 255] << 16) | ((uint32_t)lut[packed >> 24] << 24);
        *(uint32_t*)(p.ycat + px * cy + ch) = o;
      }
    }
  }
  {   // the input half of the cat, from the staged tile
    const int dpp = p.cin >> 2; const int total = 128 * dpp;
    const uint8_t* l1 = lut + 256;
    for (int u = tid; u < total; u += 256) {
      const int row = u / dpp, c0 = (u - row * dpp) * 4;
      const int64_t px = p0 + row;
      if (px >= p.npix) continue;
      const uint32_t src = *(const uint32_t*)(xs + row * p.kstr + c0);
      const uint32_t o = (uint32_t)l1[src & 255] | ((uint32_t)l1[(src >> 8) & 255] << 8) | ((uint32_t)l1[(src >> 16) & 255] << 16) | ((uint32_t)l1[src >> 24] << 24);
      *(uint32_t*)(p.ycat + px * cy + p.r + c0) = o;
    }
  }
}
extern "C" int frost_sq_emit_cat_ok(int cin, int r) { return (cin % 8 == 0) && (r % 4 == 0) && cin <= 192 && r <= 128; }
extern "C" int frost_sq_emit_cat(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int r, const float* coef,
                                 const float* qrec_sq, const float* qrec_cat, int8_t* y_sq, int8_t* y_cat, int mode, void* stream) {
  FROST_REQUIRE(frost_sq_emit_cat_ok(cin, r), "sq_emit_cat: cin <= 192 (multiple of 8), r <= 128 (multiple of 4)");
  FROST_REQUIRE(mode >= 1 && mode <= 3, "sq_emit_cat: mode 1 (fake-quant emit), 2 / 3 (converted QNNPACK / FBGEMM form), as frost_pw_conv_fwd");
  SqCatP p = {};
  p.cvt = mode - 1;
  p.x = x; p.qx = qrec_x; p.w = wq_pack; p.wsum = wsum; p.coef = coef; p.qsq = qrec_sq; p.qcat = qrec_cat; p.ysq = y_sq; p.ycat = y_cat;
  p.npix = npix; p.cin = cin; p.r = r; p.cpad = round_up(r, 16);
  const int ksm = round_up(cin, 64) / 64;
  p.kstr = ksm * 64 + 16;
  const size_t lds = (size_t)128 * p.kstr + 512;
  const dim3 grid((unsigned)((npix + 127) / 128));
  hipStream_t s = as_stream(stream);
  if (ksm == 1) hipLaunchKernelGGL(k_sq_emit_cat<1>, grid, dim3(256), lds, s, p);
  else if (ksm == 2) hipLaunchKernelGGL(k_sq_emit_cat<2>, grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL(k_sq_emit_cat<3>, grid, dim3(256), lds, s, p);
  return frost_check_launch("sq_emit_cat");
}

// ================================================================================================ squeeze_conv forward as ONE persistent launch
// statistics -> [device-wide barrier, the finalize inside it] -> emit + cat (frostnet.py:127-129), the first multi-phase kernel of the training path (SURVEY 8(f) N1;
// VERDICT r4 #1).  One workgroup per 128-pixel tile, every workgroup resident at once (frost_sq_fwd_ok checks the grid against the occupancy): the tile's integer conv
// output stays in the MFMA accumulators across the barrier, so x is staged once, the GEMM runs once and the emit pass has no launch, no prologue and no recomputation.
// The barrier is the two-level ticket of last_block_done2 (32 sub-counters + one main counter, relaxed agent-scope atomics: 1.7 - 2.4 us for 256 - 1024 workgroups,
// profiles/r05_gridbar_probe.txt) with the LAST arrival running conv_finalize_dev<true> -- coefficient rows and FakeQuantize records written by agent-scope stores --
// before it flips the generation word (ticket word 36, monotonic) the other workgroups poll; they read rows and records back through agent-scope loads, so no L2
// write-back / invalidate sits on the path.  The statistics are EXACT integers here (sum of squares in int64; k_pw's statistics pass forms it in fp32 within a tile),
// so coefficient rows agree with frost_pw_conv_fwd_fin to ~1e-7 and the emitted indices wherever that does not move a rounding tie.
// A workgroup that polls for ~2^22 rounds gives up, raises ticket word 37 and emits with whatever it reads: wrong results and a test failure instead of a hung device.
#define SQF_GEN 36
#define SQF_ERR 37
struct SqFwdP { SqCatP c; uint8_t* stats; uint8_t* slots; FrostFinDesc fin; int dbg; };      // dbg (FROST_SQF_DBG, timing only, wrong results): 1 = nobody waits, 2 = no fold / finalize either
template <int KSM, int CTM>
__global__ __launch_bounds__(256, 4) void k_sq_fwd(const SqFwdP q) {
  const SqCatP& p = q.c;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xs = smem;                          // [128][kstr]
  uint8_t* const lut = smem + 128 * p.kstr;          // [2][256]
  int* const sflag = (int*)(lut + 512);              // 16 B: flag; then 8 floats for the finalize's reduction
  float* const shf = (float*)(lut + 512 + 16);
  unsigned long long* const l_s1 = (unsigned long long*)(lut + 512 + 64);
  unsigned long long* const l_s2 = l_s1 + p.cpad;
  int* const l_mn = (int*)(l_s2 + p.cpad); int* const l_mx = l_mn + p.cpad;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int64_t p0 = (int64_t)blockIdx.x * 128;
  uint32_t gen0 = 0;
  if (tid == 0) gen0 = __hip_atomic_load(q.fin.counter + SQF_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // before this workgroup can arrive: nobody can have flipped it
  for (int i = tid; i < p.cpad; i += 256) { l_s1[i] = 0; l_s2[i] = 0; l_mn[i] = INT32_MAX; l_mx[i] = INT32_MIN; }
  {
    const int upr = p.cin >> 3; const int total = 128 * upr;
    for (int u = tid; u < total; u += 256) {
      const int row = u / upr, col = u - row * upr;
      const int64_t px = p0 + row;
      uint2 v = make_uint2(0, 0);
      if (px < p.npix) v = *(const uint2*)(p.x + px * p.cin + col * 8);
      *(uint2*)(xs + row * p.kstr + col * 8) = v;
    }
  }
  __syncthreads();
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  const int CT = p.cpad >> 4;
  v4i acc[CTM][2];
  // ---- phase 1: the tile's integer conv output (kept) and its exact statistics
#pragma unroll
  for (int ct = 0; ct < CTM; ++ct) {
    if (ct < CT) {
      const int ch = ct * 16 + 4 * g;
      v4i afr[KSM];
#pragma unroll
      for (int ks = 0; ks < KSM; ++ks) afr[ks] = *(const v4i*)(p.w + ((((int64_t)ct * KSM + ks) * 64 + lane) << 4));
      const int4 ws = *(const int4*)(p.wsum + ch);
      int a1[4] = {0, 0, 0, 0}, mn[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX}, mx[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
      long long a2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = (wv * 2 + t) * 16 + j;
        v4i a = (v4i){-zpx * ws.x, -zpx * ws.y, -zpx * ws.z, -zpx * ws.w};
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
          const v4i bfr = *(const v4i*)(xs + row * p.kstr + ks * 64 + g * 16);
          a = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[ks], bfr, a, 0, 0, 0);
        }
        acc[ct][t] = a;
        if ((p0 + row) < p.npix) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int v = a[r]; a1[r] += v; a2[r] += (long long)v * v; mn[r] = min(mn[r], v); mx[r] = max(mx[r], v); }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {          // the 16 pixel lanes j of this g (a tile's 128 values of one channel sum to < 2^31: 127 * 255 * 192 * 128)
          a1[r] += __shfl_xor(a1[r], o); a2[r] += __shfl_xor(a2[r], o); mn[r] = min(mn[r], __shfl_xor(mn[r], o)); mx[r] = max(mx[r], __shfl_xor(mx[r], o));
        }
        if (j == 0 && (ch + r) < p.r && mn[r] <= mx[r]) {
          atomicAdd(&l_s1[ch + r], (unsigned long long)(long long)a1[r]); atomicAdd(&l_s2[ch + r], (unsigned long long)a2[r]);
          atomicMin(&l_mn[ch + r], mn[r]); atomicMax(&l_mx[ch + r], mx[r]);
        }
      }
    }
  }
  __syncthreads();
  // ---- the tile's statistics -> the layer's table WITHOUT one same-address atomic per workgroup and channel (784 workgroups x 45 ns serialised at the memory side were
  // 35 us of the first version of this kernel): every workgroup stores its rows into its own slot (agent-scope stores), the last arrival of each of the 32 ticket
  // sub-groups adds its group's ~total / 32 slots and issues ONE set of atomics for the group, the last of those finalizes: a reduction tree riding on the barrier's tickets
  const unsigned total = gridDim.x, bidx = blockIdx.x, sub = bidx & 31u, nsub = total < 32u ? total : 32u;
  {
    unsigned long long* sl1 = (unsigned long long*)q.slots + (size_t)bidx * 3 * p.cpad; unsigned long long* sl2 = sl1 + p.cpad; int* slm = (int*)(sl2 + p.cpad);
    for (int c = tid; c < p.r; c += 256) {
      __hip_atomic_store(sl1 + c, l_s1[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(sl2 + c, l_s2[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(slm + 2 * c, l_mn[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(slm + 2 * c + 1, l_mx[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's slot stores have been performed
  __syncthreads();
  if (tid == 0) {
    const unsigned mine = (total - sub + 31u) >> 5;                // workgroups with index = sub (mod 32)
    const unsigned t = atomicAdd(q.fin.counter + 1 + sub, 1u);
    if (t == mine - 1u) __hip_atomic_store(q.fin.counter + 1 + sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *sflag = (t == mine - 1u) ? 1 : 0;
  }
  __syncthreads();
  bool last = false;
  if (*sflag && !(q.dbg & 2)) {                                                    // last of its sub-group: fold the group's slots, one set of atomics, then the main ticket
    long long* g_s1 = (long long*)stats_copy(q.stats, p.cpad); unsigned long long* g_s2 = (unsigned long long*)(g_s1 + p.cpad);
    int* g_mn = (int*)(g_s2 + p.cpad); int* g_mx = g_mn + p.cpad;
    // thread = (channel, member slice): the group's ~total / 32 slots are read by 256 / r... threads side by side (one memory round trip, not one per member), LDS atomics fold the slices
    for (int i = tid; i < p.cpad; i += 256) { l_s1[i] = 0; l_s2[i] = 0; l_mn[i] = INT32_MAX; l_mx[i] = INT32_MIN; }
    __syncthreads();
    {
      const int nsl = 256 / p.cpad > 0 ? 256 / p.cpad : 1;        // member slices (cpad <= 96: 2 .. 16)
      const int c = tid % p.cpad, sl = tid / p.cpad;
      if (sl < nsl && c < p.r) {
        unsigned long long a1 = 0, a2 = 0; int mn = INT32_MAX, mx = INT32_MIN;
        for (unsigned m = sub + 32u * (unsigned)sl; m < total; m += 32u * (unsigned)nsl) {
          const unsigned long long* sl1 = (const unsigned long long*)q.slots + (size_t)m * 3 * p.cpad; const unsigned long long* sl2 = sl1 + p.cpad; const int* slm = (const int*)(sl2 + p.cpad);
          a1 += __hip_atomic_load(sl1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); a2 += __hip_atomic_load(sl2 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          mn = min(mn, __hip_atomic_load(slm + 2 * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); mx = max(mx, __hip_atomic_load(slm + 2 * c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        if (mn <= mx) { atomicAdd(&l_s1[c], a1); atomicAdd(&l_s2[c], a2); atomicMin(&l_mn[c], mn); atomicMax(&l_mx[c], mx); }
      }
    }
    __syncthreads();
    for (int c = tid; c < p.r; c += 256)
      if (l_mn[c] <= l_mx[c]) { atomicAdd((unsigned long long*)&g_s1[c], l_s1[c]); atomicAdd(&g_s2[c], l_s2[c]); atomicMin(&g_mn[c], l_mn[c]); atomicMax(&g_mx[c], l_mx[c]); }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned t2 = atomicAdd(q.fin.counter, 1u);
      if (t2 == nsub - 1u) __hip_atomic_store(q.fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sflag[1] = (t2 == nsub - 1u) ? 1 : 0;
    }
    __syncthreads();
    last = sflag[1] != 0;
  }
  // ---- the barrier: the last workgroup to arrive finalizes (BatchNorm coefficients, running statistics, the conv's and the cat's FakeQuantize records), then releases
  if (last) {
    conv_finalize_dev<true>(q.stats, p.npix, p.r, p.cpad, p.qx, q.fin.qrec_w, q.fin.wscale, q.fin.gamma, q.fin.beta, q.fin.rmean, q.fin.rvar, q.fin.nbt, q.fin.training,
                            q.fin.relu, q.fin.observe, 1, q.fin.coef, q.fin.qrec_y, tid, 256, shf, q.fin.cat_qrec_b, q.fin.cat_qrec_y);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's agent-scope stores have been performed
    __syncthreads();
    if (tid == 0) __hip_atomic_store(q.fin.counter + SQF_GEN, gen0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (!(q.dbg & 1)) {
    if (tid == 0) {
      int it = 0;
      while (__hip_atomic_load(q.fin.counter + SQF_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0) {
        __builtin_amdgcn_s_sleep(2);
        if (++it > (1 << 22)) { __hip_atomic_store(q.fin.counter + SQF_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
  }
  // ---- phase 2: emit from the kept accumulators + both halves of the cat (k_sq_emit_cat's expressions); rows / records through agent-scope loads
  auto ld = [](const float* a) __attribute__((always_inline)) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto ld_qp = [&](const float* a) __attribute__((always_inline)) {
    QP r; r.scale = ld(a + FROST_Q_SCALE); r.zp = __float_as_int(ld(a + FROST_Q_ZP)); r.inv = 1.0f / r.scale;
    const float m = ld(a + FROST_Q_QMAX); r.hi = (m > 0.0f) ? (int)m : 255; return r;
  };
  const QP A = ld_qp(p.qsq);
  {
    const QP B = load_qp(p.qx), Y = ld_qp(p.qcat);
    const int qi = (int)(int8_t)tid + 128;
    lut[tid] = (uint8_t)((fq_index((float)(qi - A.zp) * A.scale, Y.inv, Y.zp, 0, Y.hi) - 128) & 255);
    lut[256 + tid] = (uint8_t)((fq_index((float)(qi - B.zp) * B.scale, Y.inv, Y.zp, 0, Y.hi) - 128) & 255);
  }
  __syncthreads();
  const float y_inv = 1.0f / A.scale, y_zpf = (float)A.zp;
  const float qcap = (float)A.hi; const bool lowq = qcap < 255.0f;
  const int cy = p.r + p.cin;
#pragma unroll
  for (int ct = 0; ct < CTM; ++ct) {
    if (ct < CT) {
      const int ch = ct * 16 + 4 * g;
      float cA[4], cB[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { cA[r] = ld(p.coef + FROST_COEF_A * p.cpad + ch + r); cB[r] = ld(p.coef + FROST_COEF_B * p.cpad + ch + r); }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = (wv * 2 + t) * 16 + j;
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yv = fmaf(cA[r], (float)acc[ct][t][r], cB[r]);
          float qv = rintf(yv * y_inv) + y_zpf;
          if (lowq) qv = fminf(qv, qcap);
          packed = __builtin_amdgcn_cvt_pk_u8_f32(qv, r, packed);
        }
        packed ^= 0x80808080u;
        const int64_t px = p0 + row;
        if (px < p.npix && ch < p.r) {
          *(uint32_t*)(p.ysq + px * p.r + ch) = packed;
          const uint32_t o = (uint32_t)lut[packed & 255] | ((uint32_t)lut[(packed >> 8) & 255] << 8) | ((uint32_t)lut[(packed >> 16) & 255] << 16) | ((uint32_t)lut[packed >> 24] << 24);
          *(uint32_t*)(p.ycat + px * cy + ch) = o;
        }
      }
    }
  }
  {   // the input half of the cat, from the staged tile
    const int dpp = p.cin >> 2; const int total = 128 * dpp;
    const uint8_t* l1 = lut + 256;
    for (int u = tid; u < total; u += 256) {
      const int row = u / dpp, c0 = (u - row * dpp) * 4;
      const int64_t px = p0 + row;
      if (px >= p.npix) continue;
      const uint32_t src = *(const uint32_t*)(xs + row * p.kstr + c0);
      const uint32_t o = (uint32_t)l1[src & 255] | ((uint32_t)l1[(src >> 8) & 255] << 8) | ((uint32_t)l1[(src >> 16) & 255] << 16) | ((uint32_t)l1[src >> 24] << 24);
      *(uint32_t*)(p.ycat + px * cy + p.r + c0) = o;
    }
  }
}
template <int KSM, int CTM>
static int sq_fwd_capacity(size_t lds) {          // workgroups of this instance the device holds at once
  static int cap = -1;
  if (cap < 0) {
    int occ = 0, dev = 0; hipDeviceProp_t pr;
    (void)hipFuncSetAttribute((const void*)k_sq_fwd<KSM, CTM>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_sq_fwd<KSM, CTM>, 256, 40 * 1024) != hipSuccess) occ = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) occ = 0; else occ *= pr.multiProcessorCount;
    cap = occ;
  }
  (void)lds;
  return cap;
}
static size_t sq_fwd_lds(int cin, int r) { const int ksm = round_up(cin, 64) / 64; return (size_t)128 * (ksm * 64 + 16) + 512 + 64 + (size_t)round_up(r, 16) * 24; }
extern "C" int frost_sq_fwd_ok(int64_t npix, int cin, int r) {
  static const int on = getenv("FROST_SQ_PERSIST") ? atoi(getenv("FROST_SQ_PERSIST")) : 1;
  if (!on || !frost_sq_emit_cat_ok(cin, r) || npix <= 0) return 0;
  const int ksm = round_up(cin, 64) / 64, ct = round_up(r, 16) / 16, ctm = ct <= 2 ? 2 : (ct <= 4 ? 4 : 6);
  const size_t lds = sq_fwd_lds(cin, r);
  if (lds > 40 * 1024 || ct > 6) return 0;
  int cap = 0;
#define SQF_CAP(K_, C_) if (ksm == K_ && ctm == C_) cap = sq_fwd_capacity<K_, C_>(lds);
  SQF_CAP(1, 2) SQF_CAP(2, 2) SQF_CAP(3, 2) SQF_CAP(1, 4) SQF_CAP(2, 4) SQF_CAP(3, 4) SQF_CAP(1, 6) SQF_CAP(2, 6) SQF_CAP(3, 6)
#undef SQF_CAP
  const int64_t tiles = (npix + 127) / 128;
  return (tiles <= (int64_t)cap * 9 / 10) ? 1 : 0;          // a margin: the grid must be resident as a whole
}
extern "C" int64_t frost_sq_fwd_slot_bytes(int64_t npix, int r) { return ((npix + 127) / 128) * 3 * (int64_t)round_up(r, 16) * 8; }
extern "C" int frost_sq_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int r, void* stats,
                            const FrostFinDesc* fin, void* slots, int8_t* y_sq, int8_t* y_cat, void* stream) {
  FROST_REQUIRE(frost_sq_fwd_ok(npix, cin, r), "sq_fwd: the shape has no instance, or its grid does not fit the device at once (frost_sq_fwd_ok)");
  FROST_REQUIRE(fin && fin->counter && fin->coef && fin->qrec_y && fin->cat_qrec_b && fin->cat_qrec_y && stats && slots && fin->relu, "sq_fwd: a squeeze_conv (ReLU) with its cat records, statistics table and slot buffer is required");
  SqFwdP q = {};
  SqCatP& p = q.c;
  p.cvt = 0;
  p.x = x; p.qx = qrec_x; p.w = wq_pack; p.wsum = wsum; p.coef = fin->coef; p.qsq = fin->qrec_y; p.qcat = fin->cat_qrec_y; p.ysq = y_sq; p.ycat = y_cat;
  p.npix = npix; p.cin = cin; p.r = r; p.cpad = round_up(r, 16);
  const int ksm = round_up(cin, 64) / 64, ct = p.cpad / 16, ctm = ct <= 2 ? 2 : (ct <= 4 ? 4 : 6);
  p.kstr = ksm * 64 + 16;
  q.stats = (uint8_t*)stats; q.slots = (uint8_t*)slots; q.fin = *fin;
  { static const int dbg = getenv("FROST_SQF_DBG") ? atoi(getenv("FROST_SQF_DBG")) : 0; q.dbg = dbg; }
  const size_t lds = sq_fwd_lds(cin, r);
  const dim3 grid((unsigned)((npix + 127) / 128));
  hipStream_t s = as_stream(stream);
#define SQF_GO(K_, C_) if (ksm == K_ && ctm == C_) hipLaunchKernelGGL((k_sq_fwd<K_, C_>), grid, dim3(256), lds, s, q);
  SQF_GO(1, 2) SQF_GO(2, 2) SQF_GO(3, 2) SQF_GO(1, 4) SQF_GO(2, 4) SQF_GO(3, 4) SQF_GO(1, 6) SQF_GO(2, 6) SQF_GO(3, 6)
#undef SQF_GO
  return frost_check_launch("sq_fwd");
}

// ================================================================================================ cat backward + squeeze_conv reduce pass in one launch
// Backward sibling of k_sq_emit_cat: quant_cat's backward (frostnet.py:129; k_cat_bwd8: the gradient of the concatenated tensor passes where the cat's FakeQuantize
// kept the requantised operand in range, into the squeeze_conv output's gradient ga and, accumulated, into the block input's gradient gb) TOGETHER with the
// squeeze_conv's backward reduce pass (k_pw<M_BRED>: S1 += gy, S2 += gy * xhat over the STE window of its own output FakeQuantize) from ONE staged x tile:
// the squeeze conv is recomputed on the int8 MFMA (as every backward pass of this design does), its output index -- the cat's first operand -- is re-derived from
// the accumulator with the emit expression instead of being re-read.  ga is still written (the dc pass reads it as gout).  One launch less per CAS bottleneck;
// same expressions as frost_cat_bwd + frost_pw_conv_bwd(mode 0), sums in a different order.
struct SqBwdP {
  const int8_t* x; const float* qx; const int8_t* w; const int32_t* wsum; float* coef; const float* qsq; const float* qcat;
  const uint16_t* gcat; uint16_t* ga; uint16_t* gb; int acc_a, acc_b;
  int64_t npix, ntiles; int cin, r, cpad, kstr, relu;
};
__device__ __forceinline__ void sqb_store4(uint16_t* dst, const float* v, int accumulate) {
  float o[4] = {v[0], v[1], v[2], v[3]};
  if (accumulate) { const uint2 t = *(const uint2*)dst; o[0] += bf2f(t.x & 0xffff); o[1] += bf2f(t.x >> 16); o[2] += bf2f(t.y & 0xffff); o[3] += bf2f(t.y >> 16); }
  uint2 wv; wv.x = cvt_pk_bf16(o[0], o[1]); wv.y = cvt_pk_bf16(o[2], o[3]);
  *(uint2*)dst = wv;
}
template <int KSM, int CTM>
__global__ __launch_bounds__(256) void k_sq_bwd_cat(const SqBwdP p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const xs = smem;                          // [128][kstr]
  uint8_t* const ok = smem + 128 * p.kstr;           // [2][256]: does the cat's FakeQuantize pass index i of the squeeze output / of the block input
  float* const red = (float*)(ok + 512);             // [2][cpad]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  {
    const QP A = load_qp(p.qsq), B = load_qp(p.qx), Y = load_qp(p.qcat);
    const int q = (int)(int8_t)tid + 128; bool ia, ib;
    fq_index((float)(q - A.zp) * A.scale, Y.inv, Y.zp, 0, Y.hi, &ia);
    fq_index((float)(q - B.zp) * B.scale, Y.inv, Y.zp, 0, Y.hi, &ib);
    ok[tid] = ia; ok[256 + tid] = ib;
    for (int c = tid; c < 2 * p.cpad; c += 256) red[c] = 0.0f;
  }
  const int zpx = __float_as_int(p.qx[FROST_Q_ZP]) - 128;
  const int zps = __float_as_int(p.qsq[FROST_Q_ZP]), qhi = q_hi(p.qsq);
  const float y_inv = 1.0f / p.qsq[FROST_Q_SCALE], y_zpf = (float)zps;
  const float qcap = (float)qhi; const bool lowq = qcap < 255.0f;
  float t_lo = 0.0f, t_hi;
  {
    const float hi0 = (float)qhi + 0.5f - (float)zps;                             // rint(hi0) ties to the even neighbour (k_pw's window)
    t_hi = ((qhi - zps) & 1) ? __int_as_float(__float_as_int(hi0) - 1) : hi0;
    if (!p.relu) { const float lo0 = -(float)zps - 0.5f; t_lo = (zps & 1) ? lo0 : __int_as_float(__float_as_int(lo0) + 1); }
  }
  const int cy = p.r + p.cin;
  const int CT = p.cpad >> 4;
  float s1[CTM][4], s2[CTM][4];
#pragma unroll
  for (int m = 0; m < CTM; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[m][r] = 0.0f; s2[m][r] = 0.0f; }
  for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int64_t p0 = tile * 128;
    __syncthreads();
    {
      const int upr = p.cin >> 3; const int total = 128 * upr;
      for (int u = tid; u < total; u += 256) {
        const int row = u / upr, col = u - row * upr;
        const int64_t px = p0 + row;
        uint2 v = make_uint2(0, 0);
        if (px < p.npix) v = *(const uint2*)(p.x + px * p.cin + col * 8);
        *(uint2*)(xs + row * p.kstr + col * 8) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < CTM; ++ct) {
      if (ct >= CT) break;
      const int ch = ct * 16 + 4 * g;
      const bool chok = ch < p.r;
      v4i afr[KSM];
#pragma unroll
      for (int ks = 0; ks < KSM; ++ks) afr[ks] = *(const v4i*)(p.w + ((((int64_t)ct * KSM + ks) * 64 + lane) << 4));
      const int4 ws = *(const int4*)(p.wsum + ch);
      const float4 A4 = *(const float4*)(p.coef + FROST_COEF_A * p.cpad + ch), B4 = *(const float4*)(p.coef + FROST_COEF_B * p.cpad + ch);
      const float4 M4 = *(const float4*)(p.coef + FROST_COEF_M * p.cpad + ch), R4 = *(const float4*)(p.coef + FROST_COEF_R * p.cpad + ch);
      const float A[4] = {A4.x, A4.y, A4.z, A4.w}, B[4] = {B4.x, B4.y, B4.z, B4.w}, R[4] = {R4.x, R4.y, R4.z, R4.w};
      const float MR[4] = {-M4.x * R4.x, -M4.y * R4.y, -M4.z * R4.z, -M4.w * R4.w};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = (wv * 2 + t) * 16 + j;
        v4i acc = (v4i){-zpx * ws.x, -zpx * ws.y, -zpx * ws.z, -zpx * ws.w};
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
          const v4i bfr = *(const v4i*)(xs + row * p.kstr + ks * 64 + g * 16);
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[ks], bfr, acc, 0, 0, 0);
        }
        const int64_t px = p0 + row;
        if (px < p.npix && chok) {
          const uint2 gv = *(const uint2*)(p.gcat + px * cy + ch);
          float gq[4] = {__uint_as_float(gv.x << 16), __uint_as_float(gv.x & 0xffff0000u), __uint_as_float(gv.y << 16), __uint_as_float(gv.y & 0xffff0000u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float af = (float)acc[r];
            const float yv = fmaf(A[r], af, B[r]);
            const float tq = yv * y_inv;
            float qv = rintf(tq) + y_zpf;                         // the emitted index (k_sq_emit_cat / k_pw's emit expression): the cat's first operand
            if (lowq) qv = fminf(qv, qcap);
            const int qi = (int)fminf(fmaxf(qv, 0.0f), 255.0f);
            if (!ok[(qi - 128) & 255]) gq[r] = 0.0f;              // quant_cat's STE window on the squeeze half
            const float gy = (tq > t_lo && tq <= t_hi) ? gq[r] : 0.0f;   // the squeeze output's own FakeQuantize (+ ReLU)
            s1[ct][r] += gy; s2[ct][r] = fmaf(gy, fmaf(af, R[r], MR[r]), s2[ct][r]);
          }
          sqb_store4(p.ga + px * p.r + ch, gq, p.acc_a);
        }
      }
    }
    {   // the input half of the cat: gb (+)= gcat[:, r:] where the requantised input stayed in range
      const int upr = p.cin >> 3; const int total = 128 * upr;
      const uint8_t* okb = ok + 256;
      for (int u = tid; u < total; u += 256) {
        const int row = u / upr, c0 = (u - row * upr) * 8;
        const int64_t px = p0 + row;
        if (px >= p.npix) continue;
        const uint2 src = *(const uint2*)(xs + row * p.kstr + c0);
        const uint4 gv = *(const uint4*)(p.gcat + px * cy + p.r + c0);
        float gq[8] = {bf2f(gv.x & 0xffff), bf2f(gv.x >> 16), bf2f(gv.y & 0xffff), bf2f(gv.y >> 16), bf2f(gv.z & 0xffff), bf2f(gv.z >> 16), bf2f(gv.w & 0xffff), bf2f(gv.w >> 16)};
#pragma unroll
        for (int r = 0; r < 4; ++r) { if (!okb[(src.x >> (8 * r)) & 255]) gq[r] = 0.0f; if (!okb[(src.y >> (8 * r)) & 255]) gq[4 + r] = 0.0f; }
        uint16_t* dst = p.gb + px * p.cin + c0;
        sqb_store4(dst, gq, p.acc_b); sqb_store4(dst + 4, gq + 4, p.acc_b);
      }
    }
  }
  // per-channel sums: the 16 pixel lanes of a lane group -> one value, the waves through LDS, one pair of atomics per channel and workgroup
#pragma unroll
  for (int ct = 0; ct < CTM; ++ct) {
    if (ct >= CT) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = s1[ct][r], b = s2[ct][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      const int c = ct * 16 + 4 * g + r;
      if (j == 0 && c < p.r) { atomicAdd(&red[c], a); atomicAdd(&red[p.cpad + c], b); }
    }
  }
  __syncthreads();
  for (int c = tid; c < p.r; c += 256) {
    atomicAdd(s12_dst(p.coef, p.cpad, 0) + c, red[c]);
    atomicAdd(s12_dst(p.coef, p.cpad, 1) + c, red[p.cpad + c]);
  }
}
extern "C" int frost_sq_bwd_cat_ok(int cin, int r) { return (cin % 8 == 0) && (r % 8 == 0) && cin <= 192 && r <= 96; }
extern "C" int frost_sq_bwd_cat(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int r, float* coef,
                                const float* qrec_sq, int relu, const float* qrec_cat, const uint16_t* g_cat, uint16_t* ga, int acc_a, uint16_t* gb, int acc_b, void* stream) {
  FROST_REQUIRE(frost_sq_bwd_cat_ok(cin, r), "sq_bwd_cat: cin <= 192, r <= 96 (multiples of 8)");
  FROST_REQUIRE(x && g_cat && ga && gb && coef, "sq_bwd_cat: incomplete arguments");
  SqBwdP p = {};
  p.x = x; p.qx = qrec_x; p.w = wq_pack; p.wsum = wsum; p.coef = coef; p.qsq = qrec_sq; p.qcat = qrec_cat; p.gcat = g_cat; p.ga = ga; p.gb = gb;
  p.acc_a = acc_a; p.acc_b = acc_b; p.relu = relu;
  p.npix = npix; p.ntiles = (npix + 127) / 128; p.cin = cin; p.r = r; p.cpad = round_up(r, 16);
  const int ksm = round_up(cin, 64) / 64;
  p.kstr = ksm * 64 + 16;
  const size_t lds = (size_t)128 * p.kstr + 512 + (size_t)2 * p.cpad * 4;
  static const int cap = getenv("FROST_SQB_WGS") ? atoi(getenv("FROST_SQB_WGS")) : 512;        // every workgroup ends with 2 * r float atomics into the coefficient rows
  const dim3 grid((unsigned)(p.ntiles < cap ? p.ntiles : cap));
  hipStream_t s = as_stream(stream);
  const int ctm = p.cpad / 16;
#define SQB(KS, CM) hipLaunchKernelGGL((k_sq_bwd_cat<KS, CM>), grid, dim3(256), lds, s, p)
  if (ksm == 1) { if (ctm <= 2) SQB(1, 2); else SQB(1, 6); }
  else if (ksm == 2) { if (ctm <= 2) SQB(2, 2); else SQB(2, 6); }
  else { if (ctm <= 3) SQB(3, 3); else SQB(3, 6); }
#undef SQB
  return frost_check_launch("sq_bwd_cat");
}
