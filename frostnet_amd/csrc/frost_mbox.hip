// SSD MultiBoxLoss on the device: matching + encoding, per-prior classification loss, hard negative mining, both losses and their gradients -- four kernels
// instead of ~25 torch stages (two full sorts over N x 24 528 priors among them).
//
// replaces: Object_Detection/layers/modules/multibox_loss.py:48-117 (MultiBoxLoss.forward) with layers/box_utils.py:71-139 (match, encode, jaccard, point_form,
// log_sum_exp), as restated batched and sync-free by frostnet_amd/ssdlite.py::match_priors / MultiBoxLoss (that torch form stays the CPU definition and the
// parity yardstick: tests/test_gpu_detect.py).  Semantics kept term for term:
//   * every prior takes its best ground truth (jaccard), every valid ground truth keeps its best prior (overlap forced to 2, a LATER ground truth wins a shared prior);
//     best overlap < threshold -> background; loc_t = encode(matched box, prior, variances), conf_t = label + 1;
//   * loss_l = sum over positives of smooth-L1(loc - loc_t); lc = logsumexp(conf) - conf[conf_t] per prior;
//   * mining: per image the num_neg = min(negpos * num_pos, P - 1) largest lc among the non-positives (positives ranked as 0) are selected; loss_c = sum of lc
//     over positives and selected negatives; both losses are divided by the total number of positives.
// Nothing here synchronises with the host: counts, sums and 1 / N_pos stay in device memory (`out`), so the loss captures into the step's hipGraph.
#include "frost_common.h"
#include <math.h>

#define MB_T 1024          // threads of the per-image kernels

__device__ __forceinline__ float mb_block_sum(float v, float* sh) {       // sum over the workgroup; sh: >= 32 floats
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.0f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

// ---- matching: one workgroup per image
__global__ __launch_bounds__(MB_T) void k_mbox_match(const float* __restrict__ priors, const float* __restrict__ boxes, const uint8_t* __restrict__ valid, int P, int K,
                                                     float threshold, float var0, float var1, float* __restrict__ bto, int* __restrict__ bti,
                                                     float* __restrict__ loc_t, int* __restrict__ conf_t, int* __restrict__ num_pos) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float* tb = (float*)smem;                                          // [K][5]
  unsigned long long* best = (unsigned long long*)(tb + ((K * 5 + 1) & ~1));   // [K]: (overlap bits << 32) | ~prior: max = largest overlap, FIRST prior among equals
  int* tv = (int*)(best + K);                                        // [K] valid
  float* shf = (float*)(tv + K);                                     // [32]
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < K * 5; i += MB_T) tb[i] = boxes[(int64_t)n * K * 5 + i];
  for (int j = tid; j < K; j += MB_T) { tv[j] = valid[(int64_t)n * K + j] ? 1 : 0; best[j] = 0ull; }
  __syncthreads();
  float* bto_n = bto + (int64_t)n * P; int* bti_n = bti + (int64_t)n * P;
  for (int p = tid; p < P; p += MB_T) {
    const float4 pr = *(const float4*)(priors + (int64_t)p * 4);
    const float px0 = pr.x - pr.z / 2, py0 = pr.y - pr.w / 2, px1 = pr.x + pr.z / 2, py1 = pr.y + pr.w / 2;      // point_form
    const float area_p = (px1 - px0) * (py1 - py0);
    float bo = -1.0f; int bi = 0;
    for (int j = 0; j < K; ++j) {
      float ov = -1.0f;
      if (tv[j]) {
        const float tx0 = tb[j * 5], ty0 = tb[j * 5 + 1], tx1 = tb[j * 5 + 2], ty1 = tb[j * 5 + 3];
        const float iw = fmaxf(fminf(tx1, px1) - fmaxf(tx0, px0), 0.0f), ih = fmaxf(fminf(ty1, py1) - fmaxf(ty0, py0), 0.0f);
        const float inter = iw * ih;
        ov = inter / ((tx1 - tx0) * (ty1 - ty0) + area_p - inter);
        const unsigned long long key = ((unsigned long long)__float_as_uint(fmaxf(ov, 0.0f)) << 32) | (unsigned long long)(0xffffffffu - (unsigned)p);
        if (key > best[j]) atomicMax(&best[j], key);
      }
      if (ov > bo) { bo = ov; bi = j; }                               // first maximum along the ground truths
    }
    bto_n[p] = bo; bti_n[p] = bi;
  }
  __syncthreads();
  // every valid ground truth keeps its best prior (overlap forced to 2); a LATER ground truth wins a shared prior (box_utils.py:97-100: the sequential loop).  The
  // best priors stay in LDS (tv[j] <- prior index, -1 for padding rows) and every thread checks its own priors against them: no global read-after-write
  for (int j = tid; j < K; j += MB_T) tv[j] = tv[j] ? (int)(0xffffffffu - (unsigned)(best[j] & 0xffffffffull)) : -1;
  __syncthreads();
  float cnt = 0.0f;
  for (int p = tid; p < P; p += MB_T) {
    float bo = bto_n[p]; int bi = bti_n[p];                            // (this thread's own stores)
    for (int j = 0; j < K; ++j) if (tv[j] == p) { bo = 2.0f; bi = j; }
    int ct = 0;
    float4 lt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(bo < threshold)) {
      ct = (int)tb[bi * 5 + 4] + 1;
      const float4 pr = *(const float4*)(priors + (int64_t)p * 4);
      const float tx0 = tb[bi * 5], ty0 = tb[bi * 5 + 1], tx1 = tb[bi * 5 + 2], ty1 = tb[bi * 5 + 3];
      lt.x = ((tx0 + tx1) / 2 - pr.x) / (var0 * pr.z); lt.y = ((ty0 + ty1) / 2 - pr.y) / (var0 * pr.w);      // encode (box_utils.py:122-139)
      lt.z = logf((tx1 - tx0) / pr.z) / var1; lt.w = logf((ty1 - ty0) / pr.w) / var1;
      cnt += 1.0f;
    }
    conf_t[(int64_t)n * P + p] = ct;
    *(float4*)(loc_t + ((int64_t)n * P + p) * 4) = lt;
  }
  cnt = mb_block_sum(cnt, shf);
  if (tid == 0) num_pos[n] = (int)cnt;
}

// ---- per-prior losses: lc = logsumexp(conf) - conf[target]; smooth-L1 of the positives
__global__ __launch_bounds__(256) void k_mbox_loss(const float* __restrict__ loc, const float* __restrict__ conf, const float* __restrict__ loc_t, const int* __restrict__ conf_t,
                                                   int P, int C, float* __restrict__ lc, float* __restrict__ llp) {
  // llp [n][P]: the smooth-L1 term of every prior (0 off the positives).  The sums over it are formed by k_mbox_mine in a FIXED order (per image: one workgroup's
  // tree; across images: slot by slot in the last arrival) -- float atomics here made the two losses differ in the last bits from run to run
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  float ll = 0.0f;
  if (p < P) {
    const int64_t i = (int64_t)n * P + p;
    const float* cr = conf + i * C;
    const int t = conf_t[i];
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, cr[c]);
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(cr[c] - m);
    lc[i] = (m + logf(s)) - cr[t];
    if (t > 0) {
      const float4 a = *(const float4*)(loc + i * 4), b = *(const float4*)(loc_t + i * 4);
      const float d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float ad = fabsf(d[e]); ll += ad < 1.0f ? 0.5f * d[e] * d[e] : ad - 0.5f; }
    }
    llp[i] = ll;
  }
}

// ---- hard negative mining: one workgroup per image; the image's lc (positives as 0) sits in LDS, the num_neg-th largest value is found by a radix select on the
// float bits (all values >= 0: bit order = value order); out[] is finished by the last image to arrive
__global__ __launch_bounds__(MB_T) void k_mbox_mine(const float* __restrict__ lc, const int* __restrict__ conf_t, const int* __restrict__ num_pos, int N, int P, int negpos,
                                                    uint8_t* __restrict__ sel, float* __restrict__ llp, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  unsigned* v = (unsigned*)smem;                                      // [P] bits of lc, positives 0
  unsigned* hist = v + P;                                            // [256]
  unsigned* ctl = hist + 256;                                        // {prefix, remaining k, equal-taken}
  float* shf = (float*)(ctl + 4);                                    // [32]
  const int n = blockIdx.x, tid = threadIdx.x;
  const int64_t base = (int64_t)n * P;
  for (int p = tid; p < P; p += MB_T) v[p] = conf_t[base + p] > 0 ? 0u : __float_as_uint(fmaxf(lc[base + p], 0.0f));
  int k = negpos * num_pos[n]; if (k > P - 1) k = P - 1;
  if (tid == 0) { ctl[0] = 0u; ctl[1] = (unsigned)k; ctl[2] = 0u; }
  __syncthreads();
  unsigned thr = 0u, need_eq = 0u;                                    // select v > thr, plus need_eq of the v == thr
  if (k > 0) {
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = tid; i < 256; i += MB_T) hist[i] = 0u;
      __syncthreads();
      const unsigned prefix = ctl[0];
      const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      for (int p = tid; p < P; p += MB_T) { const unsigned x = v[p]; if ((x & mask) == prefix) atomicAdd(&hist[(x >> shift) & 255u], 1u); }
      __syncthreads();
      if (tid == 0) {
        unsigned rem = ctl[1], b = 255u;
        for (;; --b) { if (hist[b] >= rem) break; rem -= hist[b]; if (b == 0u) break; }
        ctl[0] = prefix | (b << shift); ctl[1] = rem;
      }
      __syncthreads();
    }
    thr = ctl[0]; need_eq = ctl[1];
  }
  float lcs = 0.0f, lls = 0.0f;
  for (int p = tid; p < P; p += MB_T) {
    const bool pos = conf_t[base + p] > 0;
    if (pos) lls += llp[base + p];
    bool s = pos;
    if (!pos && k > 0) {
      const unsigned x = v[p];
      if (x > thr) s = true;
      else if (x == thr) s = atomicAdd(&ctl[2], 1u) < need_eq;       // (equal losses: which of them is taken does not change the loss)
    }
    sel[base + p] = s ? 1 : 0;
    if (s) lcs += lc[base + p];
  }
  lcs = mb_block_sum(lcs, shf);
  lls = mb_block_sum(lls, shf);                                        // (its barriers also end every thread's reads of this image's llp row)
  if (tid == 0) {
    // the image's two sums into the first two words of its own llp row (agent-scope stores), then the arrival ticket; the LAST image adds the slots up in image
    // order: the same bits every run
    __hip_atomic_store(llp + base, lls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(llp + base + 1, lcs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const unsigned t = atomicAdd((unsigned*)(out + 5), 1u);
    if (t == (unsigned)N - 1u) {                                     // last image: totals -> the two losses and 1 / N_pos for the backward; re-arm
      __threadfence();
      float np = 0.0f, sl = 0.0f, sc = 0.0f;
      for (int i = 0; i < N; ++i) {
        np += (float)num_pos[i];
        sl += __hip_atomic_load(llp + (int64_t)i * P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sc += __hip_atomic_load(llp + (int64_t)i * P + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      out[0] = sl / np; out[1] = sc / np; out[2] = 1.0f / np;
      out[3] = 0.0f; out[4] = 0.0f; ((unsigned*)out)[5] = 0u;
    }
  }
}

// ---- gradients: dconf = (softmax - onehot) * g_c / N_pos on the selected priors, dloc = smooth-L1' * g_l / N_pos on the positives
__global__ __launch_bounds__(256) void k_mbox_bwd(const float* __restrict__ loc, const float* __restrict__ conf, const float* __restrict__ loc_t, const int* __restrict__ conf_t,
                                                  const uint8_t* __restrict__ sel, const float* __restrict__ out, const float* __restrict__ g_l, const float* __restrict__ g_c,
                                                  int P, int C, float* __restrict__ dloc, float* __restrict__ dconf) {
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int64_t i = (int64_t)n * P + p;
  const float inv_n = out[2];
  const float gl = (g_l ? g_l[0] : 0.0f) * inv_n, gc = (g_c ? g_c[0] : 0.0f) * inv_n;
  const int t = conf_t[i];
  const float* cr = conf + i * C; float* dr = dconf + i * C;
  if (sel[i]) {
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, cr[c]);
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(cr[c] - m);
    const float is = gc / s;
    for (int c = 0; c < C; ++c) dr[c] = expf(cr[c] - m) * is - (c == t ? gc : 0.0f);
  } else {
    for (int c = 0; c < C; ++c) dr[c] = 0.0f;
  }
  float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t > 0) {
    const float4 a = *(const float4*)(loc + i * 4), b = *(const float4*)(loc_t + i * 4);
    const float e[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = gl * (fabsf(e[q]) < 1.0f ? e[q] : (e[q] > 0.0f ? 1.0f : -1.0f));
    d = make_float4(o[0], o[1], o[2], o[3]);
  }
  *(float4*)(dloc + i * 4) = d;
}

extern "C" int frost_mbox_workspace_floats(void) { return 8; }
// out: 8 floats, zeroed once by the caller: {loss_l, loss_c, 1 / N_pos, running sum_l, running sum_c, arrival ticket, -, -}; every call leaves the running words zeroed
extern "C" int frost_mbox_forward(const float* loc, const float* conf, const float* priors, const float* boxes, const uint8_t* valid, int n, int p, int c, int k,
                                  float threshold, int negpos, float var0, float var1, float* bto, int32_t* bti, float* loc_t, int32_t* conf_t, float* lc, uint8_t* sel,
                                  int32_t* num_pos, float* out, void* stream) {
  FROST_REQUIRE(loc && conf && priors && boxes && valid && bto && bti && loc_t && conf_t && lc && sel && num_pos && out, "mbox_forward: incomplete arguments");
  FROST_REQUIRE(n >= 1 && n <= 65535 && p >= 1 && c >= 2 && k >= 1, "mbox_forward: bad sizes");
  const size_t lds_m = (size_t)((k * 5 + 1) & ~1) * 4 + (size_t)k * 8 + (size_t)k * 4 + 32 * 4 + 16;
  const size_t lds_s = (size_t)p * 4 + 256 * 4 + 16 + 32 * 4;
  FROST_REQUIRE(lds_m <= 64 * 1024 && lds_s <= 160 * 1024, "mbox_forward: too many ground-truth boxes per image or priors per image for the LDS-resident kernels");
  hipStream_t s = as_stream(stream);
  static bool set = false;
  if (!set) { (void)hipFuncSetAttribute((const void*)k_mbox_mine, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
  hipLaunchKernelGGL(k_mbox_match, dim3(n), dim3(MB_T), lds_m, s, priors, boxes, valid, p, k, threshold, var0, var1, bto, bti, loc_t, conf_t, num_pos);
  FROST_REQUIRE(p >= 2, "mbox_forward: at least two priors per image (the per-image sums take the first two words of the scratch row)");
  // bto (best overlap per prior) is dead once the matching kernel has run: its rows carry the per-prior smooth-L1 terms, then the per-image sums
  hipLaunchKernelGGL(k_mbox_loss, dim3((p + 255) / 256, n), dim3(256), 0, s, loc, conf, loc_t, conf_t, p, c, lc, bto);
  hipLaunchKernelGGL(k_mbox_mine, dim3(n), dim3(MB_T), lds_s, s, lc, conf_t, num_pos, n, p, negpos, sel, bto, out);
  return frost_check_launch("mbox_forward");
}
extern "C" int frost_mbox_backward(const float* loc, const float* conf, const float* loc_t, const int32_t* conf_t, const uint8_t* sel, const float* out, const float* g_l,
                                   const float* g_c, int n, int p, int c, float* dloc, float* dconf, void* stream) {
  FROST_REQUIRE(loc && conf && loc_t && conf_t && sel && out && dloc && dconf, "mbox_backward: incomplete arguments");
  hipLaunchKernelGGL(k_mbox_bwd, dim3((p + 255) / 256, n), dim3(256), 0, as_stream(stream), loc, conf, loc_t, conf_t, sel, out, g_l, g_c, p, c, dloc, dconf);
  return frost_check_launch("mbox_backward");
}
