// Depthwise 5x5 stride-1 convolution on 7x7 feature maps, forward and backward (MobileNetV3: features.14 / .15 of
// mobilenetv3_large -- 960 channels --, the 576-channel tail of mobilenetv3_small; models/mobilenetv3.py:35-36,50-51 rows
// with k = 5 at the 1/32 stage, InvertedResidual :126-166).  Round 6.
//
// The generic streaming kernels (dwconvk_stream.hip, dwconv5_bwd_stream.hip) walk down the rows of a column block with a
// ring of rows in flight.  At 7x7 that walk is all prologue: 11 input rows for 7 output rows, 50 strided weight loads per
// thread for 56 outputs, one pooled-sum atomic per (item, channel) -- 72-76 us forward and 134-139 us backward per launch
// for 48 / 96 MB (0.65 TB/s; rocprofv3 of the step, profiles/r6_a_mnv3_large_*).  Here a THREAD owns the whole 7x7 plane
// of (image, channel pair): 49 independent loads per tensor up front (no ring, no halo, no column masks), every stencil
// bound a compile-time constant (the 49 x 25 tap loop unrolls to the 841 taps that exist), the pooled sum of a sample is
// one add per channel, weights come through LDS once per workgroup.  Forward: 98 activated values + 50 weights in
// registers, two waves per SIMD.  Backward: the 7x7 plane of dy (98) + the raw input (49) + weights and weight-gradient
// accumulators (100): one wave per SIMD with the accumulator file as spill space, like dw5_bwd_s1_kernel.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

constexpr int PH = 7, PW = 7, NP = PH * PW, KK = 5, PADK = 2, SLAB = 128;   // channels per workgroup: 64 lanes x 2

struct P7Args {
  const void *x, *dz, *yraw, *res;
  void *y, *dx;
  const float* w;        // [C][25]
  const float *scale, *shift;
  const float *alpha, *beta, *gamma;
  int per_sample;
  double* stats;
  float *gap, *dw;
  int gapq;
  int B, C;
  int nrep;
  long long rstride;
  T3dQuant quant;
  int dw_slots;
  int* dw_used;
};

template <typename T, int CH> using rawvec = T __attribute__((ext_vector_type(CH)));

// A channel pair as it lies in memory, kept OPAQUE until the point of use: held as a vector of two bf16 the compiler widened
// every loaded pair to two fp32 registers at once (the backward then wanted ~480 registers for three 49-pixel planes).
template <typename T> struct Pair;
template <> struct Pair<bf16_t> {
  typedef unsigned int raw;
  static __device__ __forceinline__ raw load(const bf16_t* p) { return *reinterpret_cast<const unsigned int*>(p); }
  static __device__ __forceinline__ f32x2 widen(raw v) { return f32x2{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)}; }
  static __device__ __forceinline__ void opaque(raw& v) { asm volatile("" : "+v"(v)); }
};
template <> struct Pair<float> {
  typedef f32x2 raw;
  static __device__ __forceinline__ raw load(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
  static __device__ __forceinline__ f32x2 widen(raw v) { return v; }
  static __device__ __forceinline__ void opaque(raw& v) { asm volatile("" : "+v"(v)); }
};

// weights of the workgroup's channel slab -> LDS [25][SLAB] (tap-major: a lane reads its channel pair as one 8-byte word)
__device__ __forceinline__ void stage_weights(float* wl, const float* __restrict__ w, int cbase, int Cb) {
  for (int i = threadIdx.x; i < 25 * Cb; i += 256) {
    const int cl = i / 25, t = i - cl * 25;
    wl[t * SLAB + cl] = w[(size_t)cbase * 25 + i];
  }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void dw5_plane7_fwd_kernel(const P7Args a) {
  constexpr int CH = 2;
  __shared__ __attribute__((aligned(16))) float wl[25 * SLAB];
  __shared__ double lstat[2 * SLAB];
  using RV = rawvec<T, CH>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cbase = blockIdx.y * SLAB, Cb = min(SLAB, a.C - cbase);
  const bool on = 2 * lane < Cb;
  const int c0 = cbase + (on ? 2 * lane : 0);
  stage_weights(wl, a.w, cbase, Cb);
  for (int i = threadIdx.x; i < 2 * SLAB; i += 256) lstat[i] = 0.0;
  __syncthreads();
  f32x2 wt[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) wt[t] = *reinterpret_cast<const f32x2*>(wl + t * SLAB + 2 * lane);
  const float sc[CH] = {a.scale ? a.scale[c0] : 1.f, a.scale ? a.scale[c0 + 1] : 1.f};
  const float sh[CH] = {a.scale ? a.shift[c0] : 0.f, a.scale ? a.shift[c0 + 1] : 0.f};
  float psum[CH] = {0.f, 0.f}, psq[CH] = {0.f, 0.f};

  for (int b = blockIdx.x * 4 + wave; b < a.B && on; b += gridDim.x * 4) {
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x) + (size_t)b * NP * a.C + c0;
    T* __restrict__ yg = reinterpret_cast<T*>(a.y) + (size_t)b * NP * a.C + c0;
    RV r[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) r[p] = *reinterpret_cast<const RV*>(xg + (size_t)p * a.C);
    f32x2 A[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float t[CH] = {(float)r[p][0], (float)r[p][1]};
      act_affine_vec<CH>(t, sc, sh, ACT);
      A[p] = f32x2{t[0], t[1]};
    }
    float gs[CH] = {0.f, 0.f};
#pragma unroll
    for (int oy = 0; oy < PH; ++oy)
#pragma unroll
      for (int ox = 0; ox < PW; ++ox) {
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KK; ++ky)
#pragma unroll
          for (int kx = 0; kx < KK; ++kx) {
            const int iy = oy - PADK + ky, ix = ox - PADK + kx;                   // compile time
            if (iy >= 0 && iy < PH && ix >= 0 && ix < PW) acc = pk_fma(A[iy * PW + ix], wt[ky * KK + kx], acc);
          }
        RV o;
        o[0] = (T)acc[0];
        o[1] = (T)acc[1];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const float v = (float)o[i];
          psum[i] += v;
          psq[i] = fmaf(v, v, psq[i]);
          gs[i] += v;
        }
        *reinterpret_cast<RV*>(yg + (size_t)(oy * PW + ox) * a.C) = o;
      }
    if (a.gap) {        // the sample's pooled sum: this thread holds all of it (one add into the cleared cell, no contention)
#pragma unroll
      for (int i = 0; i < CH; ++i) t3d_pool_add(a.gap, (size_t)b * a.C + c0 + i, gs[i], a.gapq);
    }
  }

  if (a.stats) {
    if (on) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        atomicAdd(lstat + 2 * lane + i, t3d_snap(psum[i], a.quant, false));
        atomicAdd(lstat + Cb + 2 * lane + i, t3d_snap(psq[i], a.quant, true));
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cb; i += 256)
      if (lstat[i] != 0.0)
        atomicAdd(a.stats + (size_t)((blockIdx.x + blockIdx.y) % a.nrep) * a.rstride + (size_t)(i / Cb) * a.C + cbase + i % Cb,
                  lstat[i]);
  }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void dw5_plane7_bwd_kernel(const P7Args a) {
  constexpr int CH = 2;
  extern __shared__ __attribute__((aligned(16))) float lred[];       // weights [25][SLAB] fp32 | [27][Cb] fp64 accumulators
  float* wl = lred;
  double* lacc = reinterpret_cast<double*>(lred + 25 * SLAB);
  using RV = rawvec<T, CH>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cbase = blockIdx.y * SLAB, Cb = min(SLAB, a.C - cbase);
  const bool on = 2 * lane < Cb;
  const int c0 = cbase + (on ? 2 * lane : 0);
  stage_weights(wl, a.w, cbase, Cb);
  for (int i = threadIdx.x; i < 27 * Cb; i += 256) lacc[i] = 0.0;
  __syncthreads();
  f32x2 wt[25], wacc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    wt[t] = *reinterpret_cast<const f32x2*>(wl + t * SLAB + 2 * lane);
    wacc[t] = f32x2{0.f, 0.f};
  }
  const float scf[CH] = {a.scale ? a.scale[c0] : 1.f, a.scale ? a.scale[c0 + 1] : 1.f};
  const float shf[CH] = {a.scale ? a.shift[c0] : 0.f, a.scale ? a.shift[c0 + 1] : 0.f};
  const f32x2 be2 = {a.beta[c0], a.beta[c0 + 1]};
  f32x2 al2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.alpha[c0], a.alpha[c0 + 1]};
  f32x2 ga2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.gamma[c0], a.gamma[c0 + 1]};
  float psum[CH] = {0.f, 0.f}, psq[CH] = {0.f, 0.f};

  for (int b = blockIdx.x * 4 + wave; b < a.B && on; b += gridDim.x * 4) {
    const size_t img = (size_t)b * NP * a.C + c0;
    const T* __restrict__ zg = reinterpret_cast<const T*>(a.dz) + img;
    const T* __restrict__ yg = reinterpret_cast<const T*>(a.yraw) + img;
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x) + img;
    const T* __restrict__ rg = a.res ? reinterpret_cast<const T*>(a.res) + img : nullptr;
    T* __restrict__ dxg = reinterpret_cast<T*>(a.dx) + img;
    if (a.per_sample) {
      const size_t o = (size_t)b * a.C + c0;
      al2 = f32x2{a.alpha[o], a.alpha[o + 1]};
      ga2 = f32x2{a.gamma[o], a.gamma[o + 1]};
    }
    typename Pair<T>::raw rz[NP], ry[NP], rx[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      rz[p] = Pair<T>::load(zg + (size_t)p * a.C);
      ry[p] = Pair<T>::load(yg + (size_t)p * a.C);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) rx[p] = Pair<T>::load(xg + (size_t)p * a.C);
    f32x2 D[NP];          // dy = alpha dz + beta y + gamma: the BatchNorm backward of this conv's output, on load
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      D[p] = pk_fma(al2, Pair<T>::widen(rz[p]), pk_fma(be2, Pair<T>::widen(ry[p]), ga2));
      if (p % PW == PW - 1) __builtin_amdgcn_sched_barrier(0);
    }
    // phase 1: the data gradient (stencil weights live), phase 2: the weight gradient (its accumulators live) -- one after
    // the other, so that the 50 weights and the 50 accumulators are never in registers together
#pragma unroll
    for (int iy = 0; iy < PH; ++iy)
#pragma unroll
      for (int ix = 0; ix < PW; ++ix) {
        const int p = iy * PW + ix;
        const f32x2 xr = Pair<T>::widen(rx[p]);
        f32x2 g = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KK; ++ky)
#pragma unroll
          for (int kx = 0; kx < KK; ++kx) {
            const int oy = iy + PADK - ky, ox = ix + PADK - kx;               // compile time
            if (oy >= 0 && oy < PH && ox >= 0 && ox < PW) g = pk_fma(wt[ky * KK + kx], D[oy * PW + ox], g);
          }
        float gv[CH] = {g[0], g[1]}, xv[CH] = {xr[0], xr[1]};
        act_grad_affine_vec<CH>(gv, xv, scf, shf, ACT);
        if (rg) {
          const RV rr = *reinterpret_cast<const RV*>(rg + (size_t)p * a.C);
#pragma unroll
          for (int i = 0; i < CH; ++i) gv[i] += (float)rr[i];
        }
        RV ov;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          ov[i] = (T)gv[i];
          const float v = (float)ov[i];
          psum[i] += v;
          psq[i] = fmaf(v, xv[i], psq[i]);
        }
        *reinterpret_cast<RV*>(dxg + (size_t)p * a.C) = ov;
        __builtin_amdgcn_sched_barrier(0);
      }
    // (the second phase must not share sub-expressions with the first: common-subexpression elimination kept the widened
    //  input AND its BatchNorm affine of all 49 pixels alive from phase 1 to phase 2 -- ~200 registers, scratch for h-swish)
#pragma unroll
    for (int p = 0; p < NP; ++p) Pair<T>::opaque(rx[p]);
#pragma unroll
    for (int iy = 0; iy < PH; ++iy)
#pragma unroll
      for (int ix = 0; ix < PW; ++ix) {
        const int p = iy * PW + ix;
        const f32x2 xw = Pair<T>::widen(rx[p]);
        float t[CH] = {xw[0], xw[1]};
        act_affine_vec<CH>(t, scf, shf, ACT);
        const f32x2 av = {t[0], t[1]};
#pragma unroll
        for (int ky = 0; ky < KK; ++ky)
#pragma unroll
          for (int kx = 0; kx < KK; ++kx) {
            const int oy = iy + PADK - ky, ox = ix + PADK - kx;
            if (oy >= 0 && oy < PH && ox >= 0 && ox < PW) wacc[ky * KK + kx] = pk_fma(av, D[oy * PW + ox], wacc[ky * KK + kx]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
  }

  if (a.dw || a.stats) {
    if (on) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const int c = 2 * lane + e;
        if (a.dw) {
#pragma unroll
          for (int t = 0; t < 25; ++t) atomicAdd(lacc + t * Cb + c, (double)wacc[t][e]);
        }
        if (a.stats) {
          atomicAdd(lacc + 25 * Cb + c, (double)psum[e]);
          atomicAdd(lacc + 26 * Cb + c, (double)psq[e]);
        }
      }
    }
    __syncthreads();
    t3d_dw_flush<25, 256>(lacc, Cb, cbase, a.C, a.dw, a.stats, a.nrep, a.rstride, a.dw_slots, (int)blockIdx.x, a.dw_used);
  }
}

// grid: (image walkers, channel slabs); 4 waves per workgroup, a wave per image at a time
static dim3 plane_grid(int B, int C) {
  const int ns = cdiv(C, SLAB);
  int gx = 2048 / (4 * ns);                  // ~8 waves per CU over the chip
  if (gx > cdiv(B, 4)) gx = cdiv(B, 4);
  if (gx < 1) gx = 1;
  return dim3(gx, ns);
}

template <typename T>
int launch_fwd(P7Args& a, int act, hipStream_t st) {
  const dim3 grid = plane_grid(a.B, a.C);
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  if (a.stats && a.nrep < 1) { a.nrep = 1; a.rstride = 0; }
  a.quant = (a.stats && std::is_same<T, bf16_t>::value && !T3D_ENV_SET("T3D_NO_SNAP")) ? t3d_quant_for((long long)a.B * NP) : T3dQuant{0.0, 0.0};
  switch (act) {
    case T3D_ACT_RELU: T3D_LAUNCH_TIMED((dw5_plane7_fwd_kernel<T, T3D_ACT_RELU>), grid, dim3(256), 0, st, a); break;
    case T3D_ACT_RELU6: T3D_LAUNCH_TIMED((dw5_plane7_fwd_kernel<T, T3D_ACT_RELU6>), grid, dim3(256), 0, st, a); break;
    case T3D_ACT_HSWISH: T3D_LAUNCH_TIMED((dw5_plane7_fwd_kernel<T, T3D_ACT_HSWISH>), grid, dim3(256), 0, st, a); break;
    default: T3D_LAUNCH_TIMED((dw5_plane7_fwd_kernel<T, T3D_ACT_NONE>), grid, dim3(256), 0, st, a); break;
  }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <typename T>
int launch_bwd(P7Args& a, int act, hipStream_t st) {
  dim3 grid = plane_grid(a.B, a.C);
  if (a.dw && g_t3d_reduce.dw_slots > 0 && (int)grid.x > g_t3d_reduce.dw_slots) grid.x = g_t3d_reduce.dw_slots;
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  if (a.nrep < 1) { a.nrep = 1; a.rstride = 0; }
  a.dw_slots = (a.dw && g_t3d_reduce.dw_slots >= (int)grid.x) ? (int)grid.x : 0;
  a.dw_used = a.dw ? g_t3d_reduce.dw_used : nullptr;
  const size_t lds = (size_t)25 * SLAB * sizeof(float) + (size_t)27 * SLAB * sizeof(double);
  switch (act) {
    case T3D_ACT_RELU: T3D_LAUNCH_TIMED((dw5_plane7_bwd_kernel<T, T3D_ACT_RELU>), grid, dim3(256), lds, st, a); break;
    case T3D_ACT_RELU6: T3D_LAUNCH_TIMED((dw5_plane7_bwd_kernel<T, T3D_ACT_RELU6>), grid, dim3(256), lds, st, a); break;
    case T3D_ACT_HSWISH: T3D_LAUNCH_TIMED((dw5_plane7_bwd_kernel<T, T3D_ACT_HSWISH>), grid, dim3(256), lds, st, a); break;
    default: T3D_LAUNCH_TIMED((dw5_plane7_bwd_kernel<T, T3D_ACT_NONE>), grid, dim3(256), lds, st, a); break;
  }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

// t3d_dwconv_fwd, k = 5, stride 1, 7x7 planes (dwconv_fwd.hip dispatches; T3D_ERR_UNSUPPORTED -> the generic kernels)
int t3d_dw5_plane7_fwd(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats, float* gap_sum,
                       int B, int C, hipStream_t st) {
  if ((C % 2) || (pro && pro->se) || T3D_ENV_SET("T3D_DW5_NO_PLANE")) return T3D_ERR_UNSUPPORTED;
  P7Args a{};
  a.x = x; a.y = y; a.w = w; a.stats = stats; a.gap = gap_sum; a.gapq = g_t3d_reduce.pool_exact;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; }
  a.B = B; a.C = C;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  if (dtype == T3D_F32) return launch_fwd<float>(a, act, st);
  if (dtype == T3D_BF16) return launch_fwd<bf16_t>(a, act, st);
  return T3D_ERR_UNSUPPORTED;
}

// t3d_dwconv_bwd, same shapes (dwconv_bwd.hip dispatches)
int t3d_dw5_plane7_bwd(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                       const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int C,
                       hipStream_t st) {
  if ((C % 2) || (pro && pro->se) || T3D_ENV_SET("T3D_DW5_NO_PLANE")) return T3D_ERR_UNSUPPORTED;
  if (const int rc = t3d_fold_fallback(bb->alpha, st)) return rc;     // finished coefficients (no derive prologue here)
  P7Args a{};
  a.dz = dz; a.yraw = y; a.x = x; a.res = residual; a.dx = dx; a.w = w;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma; a.per_sample = bb->per_sample;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; }
  a.stats = stats; a.dw = dw; a.B = B; a.C = C;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  if (dtype == T3D_F32) return launch_bwd<float>(a, act, st);
  if (dtype == T3D_BF16) return launch_bwd<bf16_t>(a, act, st);
  return T3D_ERR_UNSUPPORTED;
}
