// Micro-benchmark for DESIGN.md finding 53 ("stage-persistent kernels"): what does a dependent PHASE cost on this
// chip when it is (a) a kernel boundary on one stream, (b) a device-wide barrier inside one persistent launch?
// Both forms run the same tiny phase body: every workgroup reads 256 floats the PREVIOUS phase wrote (another workgroup's slice:
// a real cross-workgroup dependency), adds one, writes its own slice.  256 workgroups x 256 threads (one per CU), N phases.
//   hipcc --offload-arch=gfx950 -O3 barrier_vs_boundary.hip -o barrier_vs_boundary && ./barrier_vs_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int WG = 256, TPB = 256;

__device__ __forceinline__ void phase_body(const float* __restrict__ src, float* __restrict__ dst, int wg, int phase) {
  const int from = (wg + 37 * (phase + 1)) % WG;                 // another workgroup's slice of the previous phase
  const float v = __builtin_nontemporal_load(src + from * TPB + threadIdx.x);
  dst[wg * TPB + threadIdx.x] = v + 1.f;
}

__global__ __launch_bounds__(TPB) void one_phase(const float* src, float* dst, int phase) { phase_body(src, dst, blockIdx.x, phase); }

// sense-free counting barrier: phase p is passed when the counter reaches (p + 1) * WG.  One agent-scope atomic per workgroup,
// one lane polls with sc1 loads; release / acquire fences around it.
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// hierarchical: workgroups of one XCD (blockIdx & 7) meet on their own counter, the last arriver of each XCD adds to the global one
__device__ __forceinline__ void grid_barrier_xcd(unsigned* counters, unsigned phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int xcd = blockIdx.x & 7;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(counters + 64 * (1 + xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (phase + 1) * (WG / 8) - 1) __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (phase + 1) * 8) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// tight-spin forms (no s_sleep between polls)
__device__ __forceinline__ void grid_barrier_spin(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
// hierarchical with a per-XCD release flag: the last arriver of the chip writes 8 flags (one line per XCD), pollers read their own
__device__ __forceinline__ void grid_barrier_xcd_flag(unsigned* counters, unsigned phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int xcd = blockIdx.x & 7;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(counters + 64 * (1 + xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (phase + 1) * (WG / 8) - 1) {
      const unsigned g = __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g == (phase + 1) * 8 - 1)
        for (int x = 0; x < 8; ++x) __hip_atomic_store(counters + 64 * (1 + x) + 32, phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    while (__hip_atomic_load(counters + 64 * (1 + xcd) + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase + 1) {}
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int KIND>
__global__ __launch_bounds__(TPB) void persistent(float* a, float* b, unsigned* counters, int nphases) {
  for (int p = 0; p < nphases; ++p) {
    phase_body((p & 1) ? b : a, (p & 1) ? a : b, blockIdx.x, p);
    if (KIND == 0) grid_barrier(counters, (unsigned)(p + 1) * WG);
    else if (KIND == 1) grid_barrier_xcd(counters, (unsigned)p);
    else if (KIND == 2) grid_barrier_spin(counters, (unsigned)(p + 1) * WG);
    else grid_barrier_xcd_flag(counters, (unsigned)p);
  }
}

int main() {
  const int N = 400;
  float *a, *b;
  unsigned* cnt;
  CK(hipMalloc(&a, WG * TPB * 4)); CK(hipMalloc(&b, WG * TPB * 4)); CK(hipMalloc(&cnt, 64 * 9 * 4));
  CK(hipMemset(a, 0, WG * TPB * 4)); CK(hipMemset(b, 0, WG * TPB * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    // (a) N dependent launches
    CK(hipEventRecord(e0, st));
    for (int p = 0; p < N; ++p) hipLaunchKernelGGL(one_phase, dim3(WG), dim3(TPB), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, p);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rep %d: %d dependent launches        : %.2f us per phase\n", rep, N, ms * 1e3 / N);
    for (int kind = 0; kind < 4; ++kind) {
      CK(hipMemsetAsync(cnt, 0, 64 * 9 * 4, st));
      CK(hipEventRecord(e0, st));
      if (kind == 0) hipLaunchKernelGGL(persistent<0>, dim3(WG), dim3(TPB), 0, st, a, b, cnt, N);
      else if (kind == 1) hipLaunchKernelGGL(persistent<1>, dim3(WG), dim3(TPB), 0, st, a, b, cnt, N);
      else if (kind == 2) hipLaunchKernelGGL(persistent<2>, dim3(WG), dim3(TPB), 0, st, a, b, cnt, N);
      else hipLaunchKernelGGL(persistent<3>, dim3(WG), dim3(TPB), 0, st, a, b, cnt, N);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      { const char* nm[4] = {"single counter, s_sleep polls ", "per-XCD + global, s_sleep     ", "single counter, tight spin    ", "per-XCD + global + XCD flags  "};
        printf("rep %d: one launch, %d barriers (%s): %.2f us per phase\n", rep, N, nm[kind], ms * 1e3 / N); }
    }
  }
  // correctness: after N phases every element is N (each phase adds one to a value of the previous phase)
  std::vector<float> h(WG * TPB);
  CK(hipMemcpy(h.data(), (N & 1) ? b : a, WG * TPB * 4, hipMemcpyDeviceToHost));
  int bad = 0; for (float v : h) bad += (v != (float)(3 * 3 * N) && v != (float)N);
  printf("check: first element %.0f (phases accumulate across the 9 runs), mismatching elements vs first: ", h[0]);
  int diff = 0; for (float v : h) diff += v != h[0];
  printf("%d\n", diff);
  return 0;
}
