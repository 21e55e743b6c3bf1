// Store-pattern probe: how fast can 256 CUs WRITE a [M][N] bf16 tensor (N = 96, 144: rows of 192 / 288 B)
//   mode 0: the streaming pointwise kernel's epilogue pattern -- per store instruction 16 pixels x 64 B (lane (lc, lg) writes the
//           16-B piece lg of pair q of pixel lc), NT/2 instructions per 16-pixel group
//   mode 1: the same bytes, each instruction 1 KB contiguous (lane l writes piece 64 k + l of the group's contiguous 16 N * 2 bytes)
//   mode 2: mode 1 with a small read stream beside it (16 px x 32 B per group: the expand conv's input)
//   mode 3: mode 0 with the same read stream
// build: hipcc --offload-arch=gfx950 -O3 -o storepattern storepattern.hip ; run: ./storepattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f4;

template <int MODE>
__global__ __launch_bounds__(512) void k(f4* __restrict__ out, const f4* __restrict__ in, int M, int N, f4 v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane & 15, lg = lane >> 4;
  const int ngroups = M / 16, nw = gridDim.x * 8, w0 = blockIdx.x * 8 + wave;
  const int ppr = N / 8;            // 16-B pieces per pixel row
  const int pairs = N / 32;
  f4 acc = v;
  for (int g = w0; g < ngroups; g += nw) {
    const size_t base = (size_t)g * 16 * ppr;        // in 16-B pieces
    if (MODE == 2 || MODE == 3) {
      const f4 x = in[(size_t)g * 32 + (lane & 31)];
      acc += x;
    }
    if (MODE == 0 || MODE == 3) {
      for (int q = 0; q < pairs; ++q) out[base + (size_t)lc * ppr + q * 4 + lg] = acc;
    } else {
      const int total = 16 * ppr;
      for (int p = lane; p < total; p += 64) out[base + p] = acc;
    }
  }
}

int main() {
  const int M = 3211264;
  for (int N : {96, 144, 192}) {
    const size_t bytes = (size_t)M * N * 2;
    f4 *out, *in;
    hipMalloc(&out, bytes); hipMalloc(&in, (size_t)M * 32 * 2);
    hipMemset(in, 0, (size_t)M * 32 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
      for (int blocks : {512, 1024, 2048}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          f4 v = {1.f, 2.f, 3.f, 4.f};
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, in, M, N, v);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, in, M, N, v);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, in, M, N, v);
          if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, out, in, M, N, v);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("N=%3d mode %d blocks %4d: %7.1f us  %.2f TB/s written\n", N, mode, blocks, best * 1e3, bytes / (best * 1e-3) / 1e12);
      }
    }
    hipFree(out); hipFree(in);
  }
  return 0;
}
