// GradBoost optimizers (QSGD / QRMSprop / QAdam / QAdamW) as ONE multi-tensor launch.
// replaces: the Python loop over 209 single-tensor param groups x ~15 tiny torch kernels + host numpy Laplace
// + H2D copy per tensor (optimizer.py:134-204, :313-338, :469-494, :610-635).  Per element everything is fused;
// Laplace noise and the coin come from an on-device Philox4x32-10 stream (or injected tensors for parity tests).
#include "frost_common.h"

__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void k_gradboost(const FrostOptTensor* table, const FrostOptHyper* hp, const float* __restrict__ noise_in,
                                                   const float* __restrict__ coin_in, const int64_t* prefix) {
  const FrostOptTensor t = table[blockIdx.y];
  const FrostOptHyper h = *hp;
  const int64_t base = prefix ? prefix[blockIdx.y] : 0;
  const float lr = t.lr, wd = t.weight_decay;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < t.n; i += (int64_t)gridDim.x * 256) {
    float p = t.p[i], g = t.g[i];
    if (h.kind == 3) p = p * (1.0f - lr * wd);                               // QAdamW decoupled decay first (:580)
    if (h.kind == 2 && wd != 0.0f) g = g + wd * p;                             // QAdam: L2 before the statistics (:466)
    const float ag = fabsf(g);
    float emn = t.exp_min[i], emx = t.exp_max[i];
    emn = (emn * h.beta + (1.0f - h.beta) * fminf(emn, ag)) / h.bc_beta;      // (:165-168)
    emx = (emx * h.beta + (1.0f - h.beta) * fmaxf(emx, ag)) / h.bc_beta;
    t.exp_min[i] = emn; t.exp_max[i] = emx;
    if (h.boost) {
      float lap, coin = 1.0f;
      if (noise_in) { lap = fabsf(noise_in[base + i]); if (h.toss_coin) coin = coin_in[base + i]; }
      else {
        uint32_t r[4]; const uint64_t idx = (uint64_t)(base + i);
        philox4x32((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)h.offset, (uint32_t)(h.offset >> 32), (uint32_t)h.seed, (uint32_t)(h.seed >> 32), r);
        const float u = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
        lap = -logf(u);                                                       // |Laplace(0,1)| ~ Exp(1)
        coin = (float)(r[1] >> 31);
      }
      float nz = lap * ((emx - emn) * h.noise_scale);
      if (h.toss_coin) { nz = nz * coin; t.coin[i] = coin; }
      const float sgn = (g > 0.0f) ? 1.0f : ((g < 0.0f) ? -1.0f : 0.0f);
      nz = nz * sgn;
      if (h.clip_by > 0.0f) nz = fminf(fmaxf(nz, -h.clip_by), h.clip_by);
      g = g + nz;
    }
    if (h.kind == 0) {                                                         // QSGD (:191-204)
      if (wd != 0.0f) g = g + wd * p;
      t.g[i] = g;
      float d = g;
      if (h.momentum != 0.0f) {
        float buf;
        if (t.first_step) buf = g; else buf = t.buf0[i] * h.momentum + (1.0f - h.dampening) * g;
        t.buf0[i] = buf;
        d = h.nesterov ? (g + h.momentum * buf) : buf;
      }
      p = p - lr * d;
    } else if (h.kind == 1) {                                                  // QRMSprop (:340-357)
      t.g[i] = g;
      const float gg = (wd != 0.0f) ? (g + wd * p) : g;
      float sq = t.buf1[i] * h.alpha + (1.0f - h.alpha) * gg * gg; t.buf1[i] = sq;
      float avg;
      if (h.centered) { float ga = t.buf2[i] * h.alpha + (1.0f - h.alpha) * gg; t.buf2[i] = ga; avg = sqrtf(sq - ga * ga) + h.eps; }
      else avg = sqrtf(sq) + h.eps;
      if (h.momentum > 0.0f) { float buf = t.buf0[i] * h.momentum + gg / avg; t.buf0[i] = buf; p = p - lr * buf; }
      else p = p - lr * (gg / avg);
    } else {                                                                   // QAdam / QAdamW (:496-510)
      t.g[i] = g;
      float m = t.buf0[i] * h.beta1 + (1.0f - h.beta1) * g; t.buf0[i] = m;
      float v = t.buf1[i] * h.beta2 + (1.0f - h.beta2) * g * g; t.buf1[i] = v;
      float den;
      if (h.amsgrad) { float mv = fmaxf(t.buf2[i], v); t.buf2[i] = mv; den = sqrtf(mv) / h.bc2 + h.eps; }
      else den = sqrtf(v) / h.bc2 + h.eps;
      p = p - (lr / h.bc1) * (m / den);
    }
    t.p[i] = p;
  }
}
extern "C" int frost_gradboost_step(const FrostOptTensor* table, int ntensors, int64_t max_n, const FrostOptHyper* hyper,
                                    const float* noise, const float* coin, const int64_t* prefix, void* stream) {
  int64_t gx = (max_n + 1023) / 1024; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_gradboost, dim3((unsigned)gx, ntensors), dim3(256), 0, as_stream(stream), table, hyper, noise, coin, prefix);
  return frost_check_launch("gradboost_step");
}
