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
extern T3dWorkspace g_t3d_ws_main;   // scratch of the launches on the caller's MAIN stream (t3d_set_main_workspace)

// BatchNorm finalize folded into the LAST workgroup of the kernel that produced the sums (t3d_fold_request, misc.hip;
// descriptor: t3d_bn_fold in include/t3d.h, in DEVICE memory -- the kernels take one pointer, not the ~130 bytes: the
// streaming kernels sit at the SGPR limit and the by-value form spilled scalars into their row loops, -40 %).
// Every workgroup ends with: its stat atomics -> wait for them -> barrier -> one ticket from a device counter; the
// workgroup that draws the last ticket knows all sums are complete, acquires, and does what t3d_bn_finalize /
// t3d_bn_bwd_finalize would have done in a launch of their own (~5 us each, ~100 of them per training step, all on the
// critical stream).
typedef t3d_bn_fold T3dFold;
struct T3dFoldReq { const T3dFold* desc; const double* stats; };
extern T3dFoldReq g_t3d_fold;
// host side: the pending request if it is for the sums this launch accumulates (consumes it), else null
inline const T3dFold* t3d_take_fold(const double* stats) {
  if (!stats || !g_t3d_fold.desc || g_t3d_fold.stats != stats) return nullptr;
  const T3dFold* d = g_t3d_fold.desc;
  g_t3d_fold.desc = nullptr;
  return d;
}

// called by ALL threads of every workgroup, after the workgroup's stat atomics have been issued
__device__ __forceinline__ void t3d_fold_tail(const T3dFold* __restrict__ fp, int nrep, long long rstride) {
  if (fp == nullptr) return;
  // 16 bytes, 16-aligned: the kernels' dynamic LDS starts behind this static block, and an odd 4-byte shift of that
  // base turned their 8- / 16-byte LDS reads (stencil weights) into misaligned ones: the depthwise backward lost 40 %
  __shared__ __attribute__((aligned(16))) int s_lastv[4];
  int& s_last = s_lastv[0];
  // The sums leave as device-scope atomics, which are performed at the memory side: waiting for them (vmcnt) orders
  // them before the ticket.  NOT __threadfence(): its release half writes back the XCD's dirty L2 lines -- after a
  // kernel that has just stored hundreds of MB that made every workgroup's exit cost more than the finalize launch
  // it replaces (measured: step 9.4 -> 15.9 ms).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
    s_last = atomicAdd(fp->counter, 1u) == nblk - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const T3dFold f = *fp;
  if (threadIdx.x == 0) {
    *f.counter = 0;
    if (f.kind == 1 && f.nbt) *f.nbt += 1;
  }
  // 16 replica lanes per channel (all replica loads of a channel in flight at once), 4 channel groups unrolled
  const int rl = threadIdx.x & 15, cl = threadIdx.x >> 4, cpp = blockDim.x >> 4;     // channels per pass
  for (int c0 = 0; c0 < f.C; c0 += 4 * cpp) {
    double s1[4], s2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * cpp + cl;
      s1[u] = s2[u] = 0.0;
      if (c < f.C) {
        for (int r = rl; r < nrep; r += 16) {
          s1[u] += f.stats[r * rstride + c];
          s2[u] += f.stats[r * rstride + f.C + c];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        s1[u] += __shfl_xor(s1[u], o, 16);
        s2[u] += __shfl_xor(s2[u], o, 16);
      }
      const int c = c0 + u * cpp + cl;
      if (c >= f.C || rl != 0) continue;
      if (f.kind == 1) {
        const double mean = s1[u] / f.count;
        double var = s2[u] / f.count - mean * mean;  // biased
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
        const float sc = g * invstd;
        f.o0[c] = sc;
        f.o1[c] = b - (float)mean * sc;
        if (f.o2) f.o2[c] = (float)mean;
        if (f.o3) f.o3[c] = invstd;
        if (f.rm) f.rm[c] = (1.f - f.momentum) * f.rm[c] + f.momentum * (float)mean;
        if (f.rv) {
          const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
          f.rv[c] = (1.f - f.momentum) * f.rv[c] + f.momentum * (float)unbiased;
        }
      } else {
        const double mu = (double)f.mean[c], is = (double)f.invstd[c];
        const double dg = is * (s2[u] - mu * s1[u]);
        const double al = (double)(f.gamma ? f.gamma[c] : 1.f) * is;
        const double be = -al * is * dg / f.count;
        f.o0[c] = (float)al;
        f.o1[c] = (float)be;
        f.o2[c] = (float)(-al * s1[u] / f.count - be * mu);
        if (f.o3) f.o3[c] = (float)dg;
        if (f.o4) f.o4[c] = (float)s1[u];
      }
    }
  }
}

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
