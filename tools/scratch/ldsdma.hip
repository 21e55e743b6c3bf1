// feasibility probe: direct global -> LDS loads (no VGPR destination) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int n) {
  __shared__ unsigned buf[4][256];                      // 4 waves x 64 lanes x 4 slots
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned* g = src + blockIdx.x * 1024 + wave * 256;
  // four 256-B wave loads into this wave's 1 KB slice
#pragma unroll
  for (int s = 0; s < 4; ++s)
    __builtin_amdgcn_global_load_lds(g + s * 64 + lane, (__attribute__((address_space(3))) void*)(&buf[wave][s * 64]), 4, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc.
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) dst[blockIdx.x * 1024 + wave * 256 + s * 64 + lane] = buf[wave][s * 64 + lane] + 1;
}
int main() {
  const int n = 1024 * 64;
  unsigned *a, *b;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  unsigned* h = new unsigned[n];
  for (int i = 0; i < n; ++i) h[i] = i * 7u;
  hipMemcpy(a, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 1024), dim3(256), 0, 0, a, b, n);
  hipMemcpy(h, b, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += h[i] != i * 7u + 1;
  printf("bad %d of %d\n", bad, n);
  return bad != 0;
}
