// Probe: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the conv epilogues are built from.
// One workgroup of 256 threads per CU-slot, 8 independent chains per lane, s_memtime bracketed; 4 waves on a CU = 1 per SIMD,
// so cycles / (iters * 8 * UNROLL) is the per-instruction issue cost with no co-resident waves to hide it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CH 8
#define OPS(NAME, DECL, BODY, SINK)                                                           \
  __global__ void NAME(int* out, long long* t, int iters) {                                   \
    DECL;                                                                                     \
    long long c0 = clock64();                                                                 \
    for (int i = 0; i < iters; ++i) {                                                         \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                         \
      _Pragma("unroll") for (int c = 0; c < CH; ++c) { BODY; } }                              \
    }                                                                                         \
    long long c1 = clock64();                                                                 \
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;                                  \
    int s = 0; _Pragma("unroll") for (int c = 0; c < CH; ++c) s += SINK;                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                           \
  }

OPS(k_fma, float a[CH]; float b = threadIdx.x * 1e-3f; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = __builtin_fmaf(a[c], b, 0.5f), (int)a[c])
__device__ __forceinline__ v2f mk2(float x, float y) { v2f r; r.x = x; r.y = y; return r; }
OPS(k_pkfma, v2f a[CH]; v2f b = mk2(threadIdx.x * 1e-3f, 1.0f); for (int c = 0; c < CH; ++c) a[c] = mk2((float)c, b.x),
    a[c] = __builtin_elementwise_fma(a[c], b, b), (int)(a[c].x + a[c].y))
OPS(k_addi, int a[CH]; int b = threadIdx.x; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = a[c] + b; asm volatile("" : "+v"(a[c])), a[c])
OPS(k_mad64, long long a[CH]; int b = threadIdx.x + 3; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = a[c] + (long long)b * (int)(a[c] & 0xffff); asm volatile("" : "+v"(a[c])), (int)a[c])
OPS(k_mad64u, unsigned long long a[CH]; unsigned b = threadIdx.x + 3; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = a[c] + (unsigned long long)b * (unsigned)(a[c]); asm volatile("" : "+v"(a[c])), (int)a[c])
OPS(k_dot4, int a[CH]; int b = threadIdx.x * 0x01010101; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = __builtin_amdgcn_sdot4(a[c], b, a[c], false), a[c])
OPS(k_align, int a[CH]; int b = threadIdx.x * 0x01010101; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = __builtin_amdgcn_alignbyte(a[c], b, 1), a[c])
OPS(k_min3, int a[CH]; int b = threadIdx.x; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = min(min(a[c], b + i), c - i); asm volatile("" : "+v"(a[c])), a[c])
OPS(k_cvt, float a[CH]; int b = threadIdx.x; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = (float)(__float_as_int(a[c]) + b); asm volatile("" : "+v"(a[c])), (int)a[c])
OPS(k_fma64, double a[CH]; double b = threadIdx.x * 1e-3; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = __builtin_fma(a[c], b, 0.5), (int)a[c])
OPS(k_rint, float a[CH]; float b = threadIdx.x * 1e-3f; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = __builtin_rintf(a[c]) + b, (int)a[c])
OPS(k_cvtpk, unsigned a[CH]; float b = threadIdx.x * 1e-3f; for (int c = 0; c < CH; ++c) a[c] = c,
    a[c] = __builtin_amdgcn_cvt_pk_u8_f32(b + (float)i, c & 3, a[c]), (int)a[c])
OPS(k_cndmask, float a[CH]; float b = threadIdx.x * 1e-3f; for (int c = 0; c < CH; ++c) a[c] = c + b,
    a[c] = (a[c] > b) ? a[c] - 1.0f : b, (int)a[c])

template <typename K> static void run(const char* name, K kern, int instr_per_body, int waves_per_simd) {
  int* o; long long* t; hipMalloc(&o, 4096 * 1024 * 4); hipMalloc(&t, 16);
  const int iters = 4000;
  const int blocks = 256 * waves_per_simd;     // 256-thread blocks: 4 waves = 1 per SIMD; more blocks/CU -> more waves/SIMD
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, t, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    if (rep == 1) {
      const double n = (double)iters * CH * 8 * instr_per_body;
      printf("%-10s waves/SIMD=%d: wave-local %.2f cyc/instr   SIMD issue %.2f cyc/instr (from wall %.3f ms @2.4GHz)\n", name, waves_per_simd,
             (double)h / n, ms * 1e-3 * 2.4e9 / (n * waves_per_simd), ms);
    }
  }
  hipFree(o); hipFree(t);
}
int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run("fma_f32", k_fma, 1, w); run("pk_fma_f32", k_pkfma, 1, w); run("add_i32", k_addi, 1, w); run("mad_i64_i32", k_mad64, 1, w);
    run("mad_u64_u32", k_mad64u, 1, w); run("sdot4", k_dot4, 1, w); run("alignbyte", k_align, 1, w); run("min3ish", k_min3, 2, w);
    run("cvt_f32_i32", k_cvt, 2, w); run("fma_f64", k_fma64, 1, w); run("rint+add", k_rint, 2, w); run("cvt_pk_u8", k_cvtpk, 2, w);
    run("cmp+cndmask", k_cndmask, 3, w);
  }
  return 0;
}
