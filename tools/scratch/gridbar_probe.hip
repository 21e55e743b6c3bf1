// probe: cost of a software grid barrier (256 workgroups x 512 threads, one per CU) and of streaming a 307-KB weight matrix from L2
// into every workgroup, on gfx950.  build: hipcc --offload-arch=gfx950 -O3 gridbar_probe.hip -o gridbar_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int V>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned& epoch, unsigned nblk) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++epoch;
    const unsigned want = epoch * nblk;
    long spins = 0;
    if (V == 0) {
      __threadfence();
      atomicAdd(ctr, 1u);
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1l << 22)) break;      // never hang the box
      }
    } else if (V == 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1l << 22)) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
      __builtin_amdgcn_s_waitcnt(0);
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1l << 22)) break;
      }
    }
  }
  __syncthreads();
}
template <int V>
__global__ __launch_bounds__(512) void bar_kernel(unsigned* ctr, int nbar, double* stats, int nstat, const uint4* w, int wn, float* sink) {
  unsigned epoch = 0;
  float acc = 0.f;
  for (int i = 0; i < nbar; ++i) {
    if (w) {
      uint4 s = {0, 0, 0, 0};
      for (int j = threadIdx.x; j < wn; j += 512) { const uint4 v = w[j]; s.x ^= v.x; s.y ^= v.y; s.z ^= v.z; s.w ^= v.w; }
      acc += (float)(s.x ^ s.y ^ s.z ^ s.w);
    }
    if (stats) for (int j = threadIdx.x; j < nstat; j += 512) atomicAdd(stats + (size_t)(blockIdx.x % 16) * nstat + j, 1.0);
    grid_barrier<V>(ctr, epoch, gridDim.x);
  }
  if (acc == 123.f) sink[0] = acc;
}
int main() {
  unsigned* ctr; double* stats; uint4* w; float* sink;
  hipMalloc(&ctr, 4); hipMalloc(&stats, 16 * 1920 * 8); hipMalloc(&w, 307200); hipMalloc(&sink, 4);
  hipMemset(stats, 0, 16 * 1920 * 8); hipMemset(w, 1, 307200);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 3; ++v)
  for (int mode = 0; mode < 4; ++mode) {
    const int nbar = 64;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(ctr, 0, 4);
      hipEventRecord(e0);
      hipLaunchKernelGGL((v == 0 ? bar_kernel<0> : v == 1 ? bar_kernel<1> : bar_kernel<2>), dim3(256), dim3(512), 0, 0, ctr, nbar, (mode & 1) ? stats : nullptr, 1920, (mode & 2) ? w : nullptr, 307200 / 16, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("barrier v%d mode %d (stats %d, weights %d): %.2f us per phase\n", v, mode, mode & 1, (mode >> 1) & 1, ms * 1e3 / nbar);
    }
  }
  return 0;
}
