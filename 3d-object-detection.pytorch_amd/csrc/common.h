// Shared device helpers for the gfx950 kernels (wave64, NHWC activations).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/t3d.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// packed fp32 FMA: one v_pk_fma_f32 does two lanes' worth of fused multiply-adds per issue slot (the fp32 VALU peak
// on gfx950 is only reachable in this form); same rounding as two fmaf() calls.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

#define T3D_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t e_ = hipGetLastError();                      \
    if (e_ != hipSuccess) return T3D_ERR_LAUNCH;            \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Reduction replicas (t3d_set_reduction_replicas, misc.hip): contended atomics into one small array run an order of
// magnitude below the chip's atomic rate, so block b adds its BatchNorm sums / depthwise weight gradient into
// replica b % nrep; the finalize kernels (and the caller, for dw) sum the replicas.
struct T3dReduceCfg { int nrep; long long stats_stride; };
extern T3dReduceCfg g_t3d_reduce;

// Scratch workspace in device memory (t3d_set_workspace, misc.hip): partial results of split reductions (the pointwise
// weight gradient) are written there with plain stores and summed by a second small kernel instead of leaving as atomics.
struct T3dWorkspace { void* ptr; long long bytes; };
extern T3dWorkspace g_t3d_ws;

// ---- 8-channel vector load/store, storage type T, math in fp32 ------------
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  static __device__ __forceinline__ float round(float x) { return x; }
};
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float v[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float v[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
  }
  static __device__ __forceinline__ float round(float x) { return (float)(bf16_t)x; }
};

// ---- activations (reference: torchdet3d/models/mobilenetv3.py:74-89) ------
// h_swish(x) = x * relu6(x + 3) / 6.  The 1/6 is applied as a multiplication (an IEEE division costs ~10 VALU
// instructions per element; the results agree to 1 ulp, far inside the 1e-4 parity bound).
#define T3D_SIXTH 0.16666667163372040f
__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case T3D_ACT_RELU: return fmaxf(x, 0.f);
    case T3D_ACT_RELU6: return __builtin_amdgcn_fmed3f(x, 0.f, 6.f);           // one v_med3_f32
    case T3D_ACT_HSWISH: return x * (__builtin_amdgcn_fmed3f(x + 3.f, 0.f, 6.f) * T3D_SIXTH);
    default: return x;
  }
}
// derivative w.r.t. the pre-activation value x (PyTorch conventions at the kinks:
// relu'(0)=0, hardtanh/relu6 passes gradient only strictly inside (0,6))
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case T3D_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case T3D_ACT_RELU6: return (x > 0.f && x < 6.f) ? 1.f : 0.f;
    case T3D_ACT_HSWISH: {
      // d/dx [x * relu6(x+3)/6] = relu6(x+3)/6 + x * [0<x+3<6]/6
      const float h = __builtin_amdgcn_fmed3f(x + 3.f, 0.f, 6.f) * T3D_SIXTH;
      return h + ((x > -3.f && x < 3.f) ? x * T3D_SIXTH : 0.f);
    }
    default: return 1.f;
  }
}
// Vector forms: the (wave-uniform) switch is taken ONCE per vector, not once per element -- inside an unrolled
// element loop hipcc keeps the per-element scalar compare/branch chain, which dominated the VALU-light kernels.
//   v[i] = act(v[i]*sc[i] + sh[i])
template <int NV>
__device__ __forceinline__ void act_affine_vec(float* v, const float* sc, const float* sh, int act) {
  switch (act) {
    case T3D_ACT_RELU:
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f);
      break;
    case T3D_ACT_RELU6:
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = __builtin_amdgcn_fmed3f(fmaf(v[i], sc[i], sh[i]), 0.f, 6.f);
      break;
    case T3D_ACT_HSWISH:
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float u = fmaf(v[i], sc[i], sh[i]);
        v[i] = u * (__builtin_amdgcn_fmed3f(u + 3.f, 0.f, 6.f) * T3D_SIXTH);
      }
      break;
    default:
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
  }
}
//   g[i] *= act'(y[i]*sc[i] + sh[i])
template <int NV>
__device__ __forceinline__ void act_grad_affine_vec(float* g, const float* y, const float* sc, const float* sh, int act) {
  switch (act) {
    case T3D_ACT_RELU:
#pragma unroll
      for (int i = 0; i < NV; ++i) g[i] = (fmaf(y[i], sc[i], sh[i]) > 0.f) ? g[i] : 0.f;
      break;
    case T3D_ACT_RELU6:
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float u = fmaf(y[i], sc[i], sh[i]);
        g[i] = (u > 0.f && u < 6.f) ? g[i] : 0.f;
      }
      break;
    case T3D_ACT_HSWISH:
#pragma unroll
      for (int i = 0; i < NV; ++i) g[i] *= act_grad(fmaf(y[i], sc[i], sh[i]), T3D_ACT_HSWISH);
      break;
    default:
      break;
  }
}
__device__ __forceinline__ float hsigmoid(float x) { return __builtin_amdgcn_fmed3f(x + 3.f, 0.f, 6.f) * T3D_SIXTH; }

// ---- wave64 helpers --------------------------------------------------------
// sum over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
#define T3D_ROR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
  v += T3D_ROR(8);
  v += T3D_ROR(4);
  v += T3D_ROR(2);
  v += T3D_ROR(1);
#undef T3D_ROR
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
