// Dev probe: what does a device-wide barrier cost on an MI355X against a kernel boundary?  (DESIGN (f): persistent multi-phase kernels vs one launch per phase)
//   hipcc --offload-arch=gfx950 -O3 -o build/probe_gridbar tools/probes/probe_gridbar.hip && build/probe_gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// sense-reversing barrier: one agent-scope atomic per workgroup, the last arrival flips the generation word the others poll
__device__ __forceinline__ void grid_barrier(unsigned* count, unsigned* gen, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned t = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == nwg - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gen, g + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
// hierarchical form: groups of 16 workgroups arrive on their own sub-counter (one 128-byte line each), the group's last arrival on the top counter, the very last flips
// the generation word; relaxed agent-scope atomics throughout (performed at the memory side, as the library's tickets), FENCE = release before / acquire after for a
// phase whose plain stores the next phase reads.  Same-address atomics serialise at ~45 ns: 16 + nwg / 16 of them on the critical path instead of nwg.
template <bool FENCE>
__device__ __forceinline__ void grid_barrier_h(unsigned* ctl, unsigned nwg, unsigned& gen_local) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned b = blockIdx.x, grp = b >> 4, ngrp = (nwg + 15u) >> 4;
    const unsigned mine = min(16u, nwg - grp * 16u);
    unsigned* gen = ctl; unsigned* top = ctl + 32; unsigned* sub = ctl + 64 + grp * 32;
    const unsigned g = gen_local;
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    bool last = false;
    if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine - 1u) {
      __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1u) {
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gen, g + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = true;
      }
    }
    if (!last) while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    gen_local = g + 1u;
  }
  __syncthreads();
}
template <bool FENCE>
__global__ void k_barriers_h(unsigned* ctl, unsigned gen0, int n, float* sink) {
  float v = threadIdx.x; unsigned gl = gen0;
  for (int i = 0; i < n; ++i) { v = v * 1.0001f + 1.0f; grid_barrier_h<FENCE>(ctl, gridDim.x, gl); }
  if (v == 12345.678f) sink[0] = v;
}
template <bool FENCE>
__global__ void k_phases_h(const float* __restrict__ src, float* sink, unsigned* ctl, unsigned gen0, int n) {
  unsigned gl = gen0;
  for (int ph = 0; ph < n; ++ph) {
    float a = 0.f;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) a += src[(size_t)blockIdx.x * 16384 + i];
    if (a == 12345.678f) atomicAdd(sink, a);
    grid_barrier_h<FENCE>(ctl, gridDim.x, gl);
  }
}
__global__ void k_barriers(unsigned* count, unsigned* gen, int n, float* sink) {
  float v = threadIdx.x;
  for (int i = 0; i < n; ++i) { v = v * 1.0001f + 1.0f; grid_barrier(count, gen, gridDim.x); }
  if (v == 12345.678f) sink[0] = v;
}
__global__ void k_empty(float* sink) { if (threadIdx.x == 9999) sink[0] = 1.0f; }
// a "phase" with a little memory work: every workgroup reads 64 KB and writes one atomic
__global__ void k_phase(const float* __restrict__ src, float* sink) {
  float a = 0.f;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) a += src[(size_t)blockIdx.x * 16384 + i];
  if (a == 12345.678f) atomicAdd(sink, a);
}
__global__ void k_phases(const float* __restrict__ src, float* sink, unsigned* count, unsigned* gen, int n) {
  for (int ph = 0; ph < n; ++ph) {
    float a = 0.f;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) a += src[(size_t)blockIdx.x * 16384 + i];
    if (a == 12345.678f) atomicAdd(sink, a);
    grid_barrier(count, gen, gridDim.x);
  }
}
int main() {
  unsigned* ctl; unsigned* hctl; float* sink; float* src; unsigned hgen = 0;
  CK(hipMalloc(&ctl, 256)); CK(hipMemset(ctl, 0, 256)); CK(hipMalloc(&hctl, 65536)); CK(hipMemset(hctl, 0, 65536)); CK(hipMalloc(&sink, 256)); CK(hipMalloc(&src, (size_t)2048 * 16384 * 4)); CK(hipMemset(src, 0, (size_t)2048 * 16384 * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 200;
  const int grids[4] = {256, 512, 1024, 2048}, thr[2] = {256, 512};
  for (int gi = 0; gi < 4; ++gi) for (int ti = 0; ti < 2; ++ti) {
    const int g = grids[gi], t = thr[ti];
    if ((long)g * t > 256L * 2048) continue;          // must be co-resident: <= 2048 threads per CU
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_barriers, dim3(g), dim3(t), 0, s, ctl, ctl + 16, N, sink);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("grid barrier  %4d workgroups x %3d threads: %.2f us per barrier\n", g, t, best * 1e3f / N);
    for (int fence = 0; fence < 2; ++fence) {
      best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        if (fence) hipLaunchKernelGGL(k_barriers_h<true>, dim3(g), dim3(t), 0, s, hctl, hgen, N, sink);
        else hipLaunchKernelGGL(k_barriers_h<false>, dim3(g), dim3(t), 0, s, hctl, hgen, N, sink);
        hgen += N;
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("  hierarchical (16-way sub-counters, relaxed atomics%s): %.2f us per barrier\n", fence ? ", release / acquire fences" : "", best * 1e3f / N);
      best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        if (fence) hipLaunchKernelGGL(k_phases_h<true>, dim3(g), dim3(t), 0, s, src, sink, hctl, hgen, N);
        else hipLaunchKernelGGL(k_phases_h<false>, dim3(g), dim3(t), 0, s, src, sink, hctl, hgen, N);
        hgen += N;
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("  hierarchical%s, persistent phase + barrier: %.2f us per phase\n", fence ? " + fences" : "", best * 1e3f / N);
    }
    best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_phases, dim3(g), dim3(t), 0, s, src, sink, ctl, ctl + 16, N);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("  persistent: phase (64 KB read / workgroup) + barrier: %.2f us per phase\n", best * 1e3f / N);
    // the same phases as a captured chain of kernels
    hipGraph_t graph; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_phase, dim3(g), dim3(t), 0, s, src, sink);
    CK(hipStreamEndCapture(s, &graph)); CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("  hipGraph chain of the same phases:                     %.2f us per phase\n", best * 1e3f / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(g), dim3(t), 0, s, sink);
    CK(hipStreamEndCapture(s, &graph)); CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("  hipGraph chain of EMPTY kernels:                       %.2f us per node\n", best * 1e3f / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
  }
  return 0;
}
