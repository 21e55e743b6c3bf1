// Microbenchmark: throughput of v_fma_f32 vs v_pk_fma_f32 (and v_pk_mul / cvt mixes) on gfx950 at 1, 2, 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o pkfma pkfma.hip && ./pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr int ITERS = 4096, NACC = 16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float acc[NACC];
  f32x2 acc2[NACC / 2];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) acc2[i] = f32x2{acc[2 * i], acc[2 * i + 1]};
  const f32x2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0) {            // NACC scalar FMAs
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {     // NACC/2 packed FMAs = the same flops
#pragma unroll
      for (int i = 0; i < NACC / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
    } else if (MODE == 2) {     // packed mul
#pragma unroll
      for (int i = 0; i < NACC / 2; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc2[i]) : "v"(a2));
    } else if (MODE == 3) {     // v_med3 + shifts (non-fma VALU)
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) s += acc2[i][0] + acc2[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, double flops_per_instr, int ninstr) {
  float* out;
  hipMalloc(&out, 256 * 256 * 8 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)grid * 4 * ITERS * ninstr;          // wave-instructions
  const double cyc_per_instr_per_simd = ms * 1e-3 * 2.4e9 / (winstr / (256.0 * 4));
  printf("%-14s waves/SIMD %d: %7.3f ms  %6.2f cyc per wave-instr per SIMD (at 2.4 GHz)  %6.1f TFLOP/s\n", name, blocks_per_cu,
         ms, cyc_per_instr_per_simd, winstr * 64 * flops_per_instr / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", w, 2, NACC);
    run<1>("v_pk_fma_f32", w, 4, NACC / 2);
    run<2>("v_pk_mul_f32", w, 2, NACC / 2);
    run<3>("v_med3_f32", w, 1, NACC);
  }
  return 0;
}
