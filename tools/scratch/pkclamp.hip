// probe: does gfx950 execute v_pk_fma_f32 with the clamp modifier (result clamped to [0, 1] per half) ?
// build + run on the box:  hipcc --offload-arch=gfx950 -O3 tools/scratch/pkclamp.hip -o /tmp/pkclamp && /tmp/pkclamp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2;
__global__ void k(const f32x2* a, const f32x2* b, f32x2* o, f32x2* o2) {
  int i = threadIdx.x;
  f32x2 x = a[i], y = b[i], d, m;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(x), "v"(y), "v"(f32x2{0.25f, -0.25f}));
  m = f32x2{fmaxf(x[0], y[0]), fmaxf(x[1], y[1])};      // (there is no v_pk_max_f32 on gfx950: the assembler rejects it)
  o[i] = d;
  o2[i] = m;
}
int main() {
  f32x2 ha[64], hb[64], ho[64], ho2[64], *a, *b, *o, *o2;
  for (int i = 0; i < 64; ++i) { ha[i] = f32x2{(i - 32) * 0.1f, (i - 20) * 0.07f}; hb[i] = f32x2{0.5f, -1.5f}; }
  hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&o, sizeof ho); hipMalloc(&o2, sizeof ho2);
  hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, o2);
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(ho2, o2, sizeof ho2, hipMemcpyDeviceToHost);
  int bad = 0, bad2 = 0;
  for (int i = 0; i < 64; ++i) {
    float e0 = fmaf(ha[i][0], hb[i][0], 0.25f), e1 = fmaf(ha[i][1], hb[i][1], -0.25f);
    e0 = e0 < 0 ? 0 : (e0 > 1 ? 1 : e0); e1 = e1 < 0 ? 0 : (e1 > 1 ? 1 : e1);
    if (ho[i][0] != e0 || ho[i][1] != e1) ++bad;
    float m0 = ha[i][0] > hb[i][0] ? ha[i][0] : hb[i][0], m1 = ha[i][1] > hb[i][1] ? ha[i][1] : hb[i][1];
    if (ho2[i][0] != m0 || ho2[i][1] != m1) ++bad2;
  }
  printf("pk_fma clamp: %d mismatches of 64; pk_max: %d mismatches of 64 (sample %g %g -> %g %g)\n", bad, bad2, ha[5][0], ha[5][1], ho[5][0], ho[5][1]);
  return 0;
}
