// Depthwise K x K convolution (K = 5; K = 3 on the small planes), stride 1 and 2, forward and backward, as REGISTER TILES (round 6; MobileNetV3's 5x5 layers at 28x28 and
// 14x14: models/mobilenetv3.py:20-38 rows with k = 5, InvertedResidual :126-166) -- the 7x7 plane kernel of
// dwconv5_plane7.hip generalised to any plane size.
//
// A thread owns 2 channels of a TH x TW tile of output pixels of one image.  It loads the (TH + 4) x (TW + 4) input window
// it needs UP FRONT -- every load independent, the whole latency paid once per tile instead of once per row of a ring --
// applies the producer's BatchNorm + activation once per loaded value, zeroes what lies outside the image with a row mask
// x column mask (the padding of the ACTIVATED tensor), and then runs the stencil with every index a compile-time constant.
// The halo is re-read by the neighbouring tiles ((TH + 4)(TW + 4) / (TH TW) = 3.1x at 4 x 7) -- out of L1 / L2, which have
// the bandwidth to spare: the generic row walk (dwconvk_stream.hip, dwconv5_bwd_stream.hip) runs these layers at 1.0-1.7 TB/s
// with one or two waves per SIMD waiting on a 2-5-row ring (28x28x120: 58 us forward, 185 us backward for 96 / 193 MB).
// Backward: the same window of dy = alpha dz + beta y + gamma, the tile's own input pixels for act' and the weight gradient;
// data gradient first, weight gradient second (plane kernel's two phases, for the same register reason).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

constexpr int SLAB = 128;

struct TileArgs {
  const void *x, *dz, *yraw, *res;
  void *y, *dx;
  const float* w;        // [C][25]
  const float *scale, *shift;
  const float *alpha, *beta, *gamma;
  int per_sample;
  double* stats;
  float *gap, *dw;
  int gapq;
  int B, H, W, C;
  int tiles_y, tiles_x, nitems;
  int nrep;
  long long rstride;
  T3dQuant quant;
  int dw_slots;
  int* dw_used;
};

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
template <typename T, int CH> using rawvec = T __attribute__((ext_vector_type(CH)));

template <int K2>
__device__ __forceinline__ void stage_weights(float* wl, const float* __restrict__ w, int cbase, int Cb) {
  for (int i = threadIdx.x; i < K2 * Cb; i += 256) {
    const int cl = i / K2, t = i - cl * K2;
    wl[t * SLAB + cl] = w[(size_t)cbase * K2 + i];
  }
}

template <typename T, int ACT, int K, int TH, int TW>
__global__ __launch_bounds__(256) void dw5_tile_fwd_kernel(const TileArgs a) {
  constexpr int CH = 2, KK = K, PADK = K / 2, K2 = K * K, WH = TH + K - 1, WW = TW + K - 1;
  __shared__ __attribute__((aligned(16))) float wl[K2 * SLAB];
  __shared__ double lstat[2 * SLAB];
  using RV = rawvec<T, CH>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cbase = blockIdx.y * SLAB, Cb = min(SLAB, a.C - cbase);
  const bool on = 2 * lane < Cb;
  const int c0 = cbase + (on ? 2 * lane : 0);
  stage_weights<K2>(wl, a.w, cbase, Cb);
  for (int i = threadIdx.x; i < 2 * SLAB; i += 256) lstat[i] = 0.0;
  __syncthreads();
  f32x2 wt[K2];
#pragma unroll
  for (int t = 0; t < K2; ++t) wt[t] = *reinterpret_cast<const f32x2*>(wl + t * SLAB + 2 * lane);
  const float sc[CH] = {a.scale ? a.scale[c0] : 1.f, a.scale ? a.scale[c0 + 1] : 1.f};
  const float sh[CH] = {a.scale ? a.shift[c0] : 0.f, a.scale ? a.shift[c0 + 1] : 0.f};
  float psum[CH] = {0.f, 0.f}, psq[CH] = {0.f, 0.f};

  for (int q = blockIdx.x * 4 + wave; q < a.nitems && on; q += gridDim.x * 4) {
    const int tx = q % a.tiles_x, r1 = q / a.tiles_x, ty = r1 % a.tiles_y, b = r1 / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.C + c0;
    T* __restrict__ yg = reinterpret_cast<T*>(a.y) + (size_t)b * a.H * a.W * a.C + c0;
    int roff[WH], coff[WW];
    float rm[WH], cm[WW];
#pragma unroll
    for (int r = 0; r < WH; ++r) {
      const int iy = oy0 - PADK + r;
      rm[r] = (iy >= 0 && iy < a.H) ? 1.f : 0.f;
      roff[r] = min(max(iy, 0), a.H - 1) * a.W;
    }
#pragma unroll
    for (int c = 0; c < WW; ++c) {
      const int ix = ox0 - PADK + c;
      cm[c] = (ix >= 0 && ix < a.W) ? 1.f : 0.f;
      coff[c] = min(max(ix, 0), a.W - 1);
    }
    typename Pair<T>::raw rw[WH * WW];
#pragma unroll
    for (int r = 0; r < WH; ++r)
#pragma unroll
      for (int c = 0; c < WW; ++c) rw[r * WW + c] = Pair<T>::load(xg + (size_t)(roff[r] + coff[c]) * a.C);
    f32x2 A[WH * WW];
#pragma unroll
    for (int r = 0; r < WH; ++r)
#pragma unroll
      for (int c = 0; c < WW; ++c) {
        const f32x2 xv = Pair<T>::widen(rw[r * WW + c]);
        float t[CH] = {xv[0], xv[1]};
        act_affine_vec<CH>(t, sc, sh, ACT);
        const float m = rm[r] * cm[c];
        A[r * WW + c] = f32x2{t[0] * m, t[1] * m};
      }
    float gs[CH] = {0.f, 0.f};
#pragma unroll
    for (int oy = 0; oy < TH; ++oy)
#pragma unroll
      for (int ox = 0; ox < TW; ++ox) {
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KK; ++ky)
#pragma unroll
          for (int kx = 0; kx < KK; ++kx) acc = pk_fma(A[(oy + ky) * WW + ox + kx], wt[ky * KK + kx], acc);
        if (oy0 + oy < a.H && ox0 + ox < a.W) {
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
          *reinterpret_cast<RV*>(yg + ((size_t)(oy0 + oy) * a.W + ox0 + ox) * a.C) = o;
        }
      }
    if (a.gap) {
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

template <typename T, int ACT, int K, int TH, int TW>
__global__ __launch_bounds__(256) void dw5_tile_bwd_kernel(const TileArgs a) {
  constexpr int CH = 2, KK = K, PADK = K / 2, K2 = K * K, WH = TH + K - 1, WW = TW + K - 1;
  extern __shared__ __attribute__((aligned(16))) float lred[];       // weights [25][SLAB] fp32 | [27][Cb] fp64 accumulators
  float* wl = lred;
  double* lacc = reinterpret_cast<double*>(lred + K2 * SLAB);
  using RV = rawvec<T, CH>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cbase = blockIdx.y * SLAB, Cb = min(SLAB, a.C - cbase);
  const bool on = 2 * lane < Cb;
  const int c0 = cbase + (on ? 2 * lane : 0);
  stage_weights<K2>(wl, a.w, cbase, Cb);
  for (int i = threadIdx.x; i < (K2 + 2) * Cb; i += 256) lacc[i] = 0.0;
  __syncthreads();
  f32x2 wt[K2], wacc[K2];
#pragma unroll
  for (int t = 0; t < K2; ++t) {
    wt[t] = *reinterpret_cast<const f32x2*>(wl + t * SLAB + 2 * lane);
    wacc[t] = f32x2{0.f, 0.f};
  }
  const float scf[CH] = {a.scale ? a.scale[c0] : 1.f, a.scale ? a.scale[c0 + 1] : 1.f};
  const float shf[CH] = {a.scale ? a.shift[c0] : 0.f, a.scale ? a.shift[c0 + 1] : 0.f};
  const f32x2 be2 = {a.beta[c0], a.beta[c0 + 1]};
  f32x2 al2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.alpha[c0], a.alpha[c0 + 1]};
  f32x2 ga2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.gamma[c0], a.gamma[c0 + 1]};
  float psum[CH] = {0.f, 0.f}, psq[CH] = {0.f, 0.f};

  for (int q = blockIdx.x * 4 + wave; q < a.nitems && on; q += gridDim.x * 4) {
    const int tx = q % a.tiles_x, r1 = q / a.tiles_x, ty = r1 % a.tiles_y, b = r1 / a.tiles_y;
    const int iy0 = ty * TH, ix0 = tx * TW;
    const size_t img = (size_t)b * a.H * a.W * a.C + c0;
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
    int roff[WH], coff[WW];
    float rm[WH], cm[WW];
#pragma unroll
    for (int r = 0; r < WH; ++r) {
      const int oy = iy0 - PADK + r;
      rm[r] = (oy >= 0 && oy < a.H) ? 1.f : 0.f;
      roff[r] = min(max(oy, 0), a.H - 1) * a.W;
    }
#pragma unroll
    for (int c = 0; c < WW; ++c) {
      const int ox = ix0 - PADK + c;
      cm[c] = (ox >= 0 && ox < a.W) ? 1.f : 0.f;
      coff[c] = min(max(ox, 0), a.W - 1);
    }
    typename Pair<T>::raw rz[WH * WW], ry[WH * WW], rx[TH * TW];
#pragma unroll
    for (int r = 0; r < WH; ++r)
#pragma unroll
      for (int c = 0; c < WW; ++c) {
        const size_t o = (size_t)(roff[r] + coff[c]) * a.C;
        rz[r * WW + c] = Pair<T>::load(zg + o);
        ry[r * WW + c] = Pair<T>::load(yg + o);
      }
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int c = 0; c < TW; ++c) rx[r * TW + c] = Pair<T>::load(xg + (size_t)(roff[r + PADK] + coff[c + PADK]) * a.C);
    f32x2 D[WH * WW];        // dy = alpha dz + beta y + gamma inside the image, 0 outside
#pragma unroll
    for (int r = 0; r < WH; ++r) {
#pragma unroll
      for (int c = 0; c < WW; ++c) {
        const float m = rm[r] * cm[c];
        D[r * WW + c] = pk_fma(al2, Pair<T>::widen(rz[r * WW + c]), pk_fma(be2, Pair<T>::widen(ry[r * WW + c]), ga2)) * f32x2{m, m};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // phase 1: data gradient of the tile's own pixels (local (r, c) <-> window (r + 2, c + 2)): dy row r + 2 + 2 - ky
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int c = 0; c < TW; ++c) {
        const f32x2 xr = Pair<T>::widen(rx[r * TW + c]);
        f32x2 g = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KK; ++ky)
#pragma unroll
          for (int kx = 0; kx < KK; ++kx) g = pk_fma(wt[ky * KK + kx], D[(r + K - 1 - ky) * WW + c + K - 1 - kx], g);
        float gv[CH] = {g[0], g[1]}, xv[CH] = {xr[0], xr[1]};
        act_grad_affine_vec<CH>(gv, xv, scf, shf, ACT);
        const bool inside = iy0 + r < a.H && ix0 + c < a.W;
        const size_t off = ((size_t)(iy0 + r) * a.W + ix0 + c) * a.C;
        if (inside) {
          if (rg) {
            const RV rr = *reinterpret_cast<const RV*>(rg + off);
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
          *reinterpret_cast<RV*>(dxg + off) = ov;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    // phase 2: weight gradient (no sub-expression shared with phase 1: see dwconv5_plane7.hip)
#pragma unroll
    for (int p = 0; p < TH * TW; ++p) Pair<T>::opaque(rx[p]);
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int c = 0; c < TW; ++c) {
        const f32x2 xw = Pair<T>::widen(rx[r * TW + c]);
        float t[CH] = {xw[0], xw[1]};
        act_affine_vec<CH>(t, scf, shf, ACT);
        const float m = (iy0 + r < a.H && ix0 + c < a.W) ? 1.f : 0.f;      // (a tile that hangs over the edge: no contribution)
        const f32x2 av = {t[0] * m, t[1] * m};
#pragma unroll
        for (int ky = 0; ky < KK; ++ky)
#pragma unroll
          for (int kx = 0; kx < KK; ++kx) wacc[ky * KK + kx] = pk_fma(av, D[(r + K - 1 - ky) * WW + c + K - 1 - kx], wacc[ky * KK + kx]);
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
          for (int t = 0; t < K2; ++t) atomicAdd(lacc + t * Cb + c, (double)wacc[t][e]);
        }
        if (a.stats) {
          atomicAdd(lacc + K2 * Cb + c, (double)psum[e]);
          atomicAdd(lacc + (K2 + 1) * Cb + c, (double)psq[e]);
        }
      }
    }
    __syncthreads();
    t3d_dw_flush<K2, 256>(lacc, Cb, cbase, a.C, a.dw, a.stats, a.nrep, a.rstride, a.dw_slots, (int)blockIdx.x, a.dw_used);
  }
}

// ---- stride 2 (pad 2): output (oy, ox) reads input (2 oy - 2 + ky, 2 ox - 2 + kx).  Forward: a TH x TW output tile from its
// (2 TH + 3) x (2 TW + 3) input window.  Backward: a thread owns a (2 TH) x (2 TW) tile of INPUT pixels; tap (ky, kx) reaches
// input pixel (iy, ix) from output ((iy + 2 - ky) / 2, (ix + 2 - kx) / 2) when both are whole, i.e. ky has the parity of iy and
// kx that of ix (tile origins are even) -- 25 / 4 taps per pixel on average, from a (TH + 2) x (TW + 2) window of dy.
template <typename T, int ACT, int K, int TH, int TW>
__global__ __launch_bounds__(256) void dw5_tile_fwd_s2_kernel(const TileArgs a) {
  constexpr int CH = 2, KK = K, PADK = K / 2, K2 = K * K, WH = 2 * TH + K - 2, WW = 2 * TW + K - 2;
  __shared__ __attribute__((aligned(16))) float wl[K2 * SLAB];
  __shared__ double lstat[2 * SLAB];
  using RV = rawvec<T, CH>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cbase = blockIdx.y * SLAB, Cb = min(SLAB, a.C - cbase);
  const bool on = 2 * lane < Cb;
  const int c0 = cbase + (on ? 2 * lane : 0);
  const int Ho = (a.H + 2 * PADK - K) / 2 + 1, Wo = (a.W + 2 * PADK - K) / 2 + 1;
  stage_weights<K2>(wl, a.w, cbase, Cb);
  for (int i = threadIdx.x; i < 2 * SLAB; i += 256) lstat[i] = 0.0;
  __syncthreads();
  f32x2 wt[K2];
#pragma unroll
  for (int t = 0; t < K2; ++t) wt[t] = *reinterpret_cast<const f32x2*>(wl + t * SLAB + 2 * lane);
  const float sc[CH] = {a.scale ? a.scale[c0] : 1.f, a.scale ? a.scale[c0 + 1] : 1.f};
  const float sh[CH] = {a.scale ? a.shift[c0] : 0.f, a.scale ? a.shift[c0 + 1] : 0.f};
  float psum[CH] = {0.f, 0.f}, psq[CH] = {0.f, 0.f};

  for (int q = blockIdx.x * 4 + wave; q < a.nitems && on; q += gridDim.x * 4) {
    const int tx = q % a.tiles_x, r1 = q / a.tiles_x, ty = r1 % a.tiles_y, b = r1 / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.C + c0;
    T* __restrict__ yg = reinterpret_cast<T*>(a.y) + (size_t)b * Ho * Wo * a.C + c0;
    int roff[WH], coff[WW];
    float rm[WH], cm[WW];
#pragma unroll
    for (int r = 0; r < WH; ++r) {
      const int iy = 2 * oy0 - PADK + r;
      rm[r] = (iy >= 0 && iy < a.H) ? 1.f : 0.f;
      roff[r] = min(max(iy, 0), a.H - 1) * a.W;
    }
#pragma unroll
    for (int c = 0; c < WW; ++c) {
      const int ix = 2 * ox0 - PADK + c;
      cm[c] = (ix >= 0 && ix < a.W) ? 1.f : 0.f;
      coff[c] = min(max(ix, 0), a.W - 1);
    }
    typename Pair<T>::raw rw[WH * WW];
#pragma unroll
    for (int r = 0; r < WH; ++r)
#pragma unroll
      for (int c = 0; c < WW; ++c) rw[r * WW + c] = Pair<T>::load(xg + (size_t)(roff[r] + coff[c]) * a.C);
    f32x2 A[WH * WW];
#pragma unroll
    for (int r = 0; r < WH; ++r)
#pragma unroll
      for (int c = 0; c < WW; ++c) {
        const f32x2 xv = Pair<T>::widen(rw[r * WW + c]);
        float t[CH] = {xv[0], xv[1]};
        act_affine_vec<CH>(t, sc, sh, ACT);
        const float m = rm[r] * cm[c];
        A[r * WW + c] = f32x2{t[0] * m, t[1] * m};
      }
    float gs[CH] = {0.f, 0.f};
#pragma unroll
    for (int oy = 0; oy < TH; ++oy)
#pragma unroll
      for (int ox = 0; ox < TW; ++ox) {
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KK; ++ky)
#pragma unroll
          for (int kx = 0; kx < KK; ++kx) acc = pk_fma(A[(2 * oy + ky) * WW + 2 * ox + kx], wt[ky * KK + kx], acc);
        if (oy0 + oy < Ho && ox0 + ox < Wo) {
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
          *reinterpret_cast<RV*>(yg + ((size_t)(oy0 + oy) * Wo + ox0 + ox) * a.C) = o;
        }
      }
    if (a.gap) {
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

template <typename T, int ACT, int K, int TH, int TW>
__global__ __launch_bounds__(256) void dw5_tile_bwd_s2_kernel(const TileArgs a) {
  // taps of an input row r (tile origin even): ky = (r + PAD) mod 2, +2, ...; they reach output rows (r + PAD - ky) / 2 relative
  // to iy0 / 2: from LO = (PAD - kymax) / 2 (kymax: the largest tap of PAD's parity) to TH -- K = 5: -1 .. TH, K = 3: 0 .. TH
  constexpr int CH = 2, KK = K, PADK = K / 2, K2 = K * K, IH = 2 * TH, IW = 2 * TW;
  constexpr int KYMAX = (K - 1) - ((K - 1 - PADK) & 1), LO = (PADK - KYMAX) / 2, WH = TH - LO + 1, WW = TW - LO + 1;
  extern __shared__ __attribute__((aligned(16))) float lred[];
  float* wl = lred;
  double* lacc = reinterpret_cast<double*>(lred + K2 * SLAB);
  using RV = rawvec<T, CH>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cbase = blockIdx.y * SLAB, Cb = min(SLAB, a.C - cbase);
  const bool on = 2 * lane < Cb;
  const int c0 = cbase + (on ? 2 * lane : 0);
  const int Ho = (a.H + 2 * PADK - K) / 2 + 1, Wo = (a.W + 2 * PADK - K) / 2 + 1;
  stage_weights<K2>(wl, a.w, cbase, Cb);
  for (int i = threadIdx.x; i < (K2 + 2) * Cb; i += 256) lacc[i] = 0.0;
  __syncthreads();
  f32x2 wt[K2], wacc[K2];
#pragma unroll
  for (int t = 0; t < K2; ++t) {
    wt[t] = *reinterpret_cast<const f32x2*>(wl + t * SLAB + 2 * lane);
    wacc[t] = f32x2{0.f, 0.f};
  }
  const float scf[CH] = {a.scale ? a.scale[c0] : 1.f, a.scale ? a.scale[c0 + 1] : 1.f};
  const float shf[CH] = {a.scale ? a.shift[c0] : 0.f, a.scale ? a.shift[c0 + 1] : 0.f};
  const f32x2 be2 = {a.beta[c0], a.beta[c0 + 1]};
  f32x2 al2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.alpha[c0], a.alpha[c0 + 1]};
  f32x2 ga2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.gamma[c0], a.gamma[c0 + 1]};
  float psum[CH] = {0.f, 0.f}, psq[CH] = {0.f, 0.f};

  for (int q = blockIdx.x * 4 + wave; q < a.nitems && on; q += gridDim.x * 4) {
    const int tx = q % a.tiles_x, r1 = q / a.tiles_x, ty = r1 % a.tiles_y, b = r1 / a.tiles_y;
    const int iy0 = ty * IH, ix0 = tx * IW;                 // even
    const size_t oimg = (size_t)b * Ho * Wo * a.C + c0, iimg = (size_t)b * a.H * a.W * a.C + c0;
    const T* __restrict__ zg = reinterpret_cast<const T*>(a.dz) + oimg;
    const T* __restrict__ yg = reinterpret_cast<const T*>(a.yraw) + oimg;
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x) + iimg;
    const T* __restrict__ rg = a.res ? reinterpret_cast<const T*>(a.res) + iimg : nullptr;
    T* __restrict__ dxg = reinterpret_cast<T*>(a.dx) + iimg;
    if (a.per_sample) {
      const size_t o = (size_t)b * a.C + c0;
      al2 = f32x2{a.alpha[o], a.alpha[o + 1]};
      ga2 = f32x2{a.gamma[o], a.gamma[o + 1]};
    }
    int roff[WH], coff[WW];
    float rm[WH], cm[WW];
#pragma unroll
    for (int r = 0; r < WH; ++r) {
      const int oy = iy0 / 2 + LO + r;
      rm[r] = (oy >= 0 && oy < Ho) ? 1.f : 0.f;
      roff[r] = min(max(oy, 0), Ho - 1) * Wo;
    }
#pragma unroll
    for (int c = 0; c < WW; ++c) {
      const int ox = ix0 / 2 + LO + c;
      cm[c] = (ox >= 0 && ox < Wo) ? 1.f : 0.f;
      coff[c] = min(max(ox, 0), Wo - 1);
    }
    typename Pair<T>::raw rz[WH * WW], ry[WH * WW], rx[IH * IW];
#pragma unroll
    for (int r = 0; r < WH; ++r)
#pragma unroll
      for (int c = 0; c < WW; ++c) {
        const size_t o = (size_t)(roff[r] + coff[c]) * a.C;
        rz[r * WW + c] = Pair<T>::load(zg + o);
        ry[r * WW + c] = Pair<T>::load(yg + o);
      }
#pragma unroll
    for (int r = 0; r < IH; ++r)
#pragma unroll
      for (int c = 0; c < IW; ++c)
        rx[r * IW + c] = Pair<T>::load(xg + ((size_t)min(iy0 + r, a.H - 1) * a.W + min(ix0 + c, a.W - 1)) * a.C);
    f32x2 D[WH * WW];
#pragma unroll
    for (int r = 0; r < WH; ++r)
#pragma unroll
      for (int c = 0; c < WW; ++c) {
        const float m = rm[r] * cm[c];
        D[r * WW + c] = pk_fma(al2, Pair<T>::widen(rz[r * WW + c]), pk_fma(be2, Pair<T>::widen(ry[r * WW + c]), ga2)) * f32x2{m, m};
      }
    // phase 1: data gradient.  Local pixel (r, c), tap (ky, kx) of its parity -> window row / column as in the header comment
#pragma unroll
    for (int r = 0; r < IH; ++r)
#pragma unroll
      for (int c = 0; c < IW; ++c) {
        const f32x2 xr = Pair<T>::widen(rx[r * IW + c]);
        f32x2 g = {0.f, 0.f};
#pragma unroll
        for (int ky = (r + PADK) & 1; ky < KK; ky += 2)
#pragma unroll
          for (int kx = (c + PADK) & 1; kx < KK; kx += 2) g = pk_fma(wt[ky * KK + kx], D[((r + PADK - ky) / 2 - LO) * WW + (c + PADK - kx) / 2 - LO], g);
        float gv[CH] = {g[0], g[1]}, xv[CH] = {xr[0], xr[1]};
        act_grad_affine_vec<CH>(gv, xv, scf, shf, ACT);
        const bool inside = iy0 + r < a.H && ix0 + c < a.W;
        const size_t off = ((size_t)(iy0 + r) * a.W + ix0 + c) * a.C;
        if (inside) {
          if (rg) {
            const RV rr = *reinterpret_cast<const RV*>(rg + off);
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
          *reinterpret_cast<RV*>(dxg + off) = ov;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    // phase 2: weight gradient
#pragma unroll
    for (int p = 0; p < IH * IW; ++p) Pair<T>::opaque(rx[p]);
#pragma unroll
    for (int r = 0; r < IH; ++r)
#pragma unroll
      for (int c = 0; c < IW; ++c) {
        const f32x2 xw = Pair<T>::widen(rx[r * IW + c]);
        float t[CH] = {xw[0], xw[1]};
        act_affine_vec<CH>(t, scf, shf, ACT);
        const float m = (iy0 + r < a.H && ix0 + c < a.W) ? 1.f : 0.f;
        const f32x2 av = {t[0] * m, t[1] * m};
#pragma unroll
        for (int ky = (r + PADK) & 1; ky < KK; ky += 2)
#pragma unroll
          for (int kx = (c + PADK) & 1; kx < KK; kx += 2)
            wacc[ky * KK + kx] = pk_fma(av, D[((r + PADK - ky) / 2 - LO) * WW + (c + PADK - kx) / 2 - LO], wacc[ky * KK + kx]);
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
          for (int t = 0; t < K2; ++t) atomicAdd(lacc + t * Cb + c, (double)wacc[t][e]);
        }
        if (a.stats) {
          atomicAdd(lacc + K2 * Cb + c, (double)psum[e]);
          atomicAdd(lacc + (K2 + 1) * Cb + c, (double)psq[e]);
        }
      }
    }
    __syncthreads();
    t3d_dw_flush<K2, 256>(lacc, Cb, cbase, a.C, a.dw, a.stats, a.nrep, a.rstride, a.dw_slots, (int)blockIdx.x, a.dw_used);
  }
}

template <int TH, int TW>
static dim3 tile_grid(TileArgs& a, int waves_target, int PH = 0, int PW = 0) {
  a.tiles_y = cdiv(PH ? PH : a.H, TH);
  a.tiles_x = cdiv(PW ? PW : a.W, TW);
  a.nitems = a.B * a.tiles_y * a.tiles_x;
  const int ns = cdiv(a.C, SLAB);
  int gx = waves_target / (4 * ns);
  if (gx > cdiv(a.nitems, 4)) gx = cdiv(a.nitems, 4);
  if (gx < 1) gx = 1;
  return dim3(gx, ns);
}

// tile shapes per stencil size: K = 5 -- 4 x 7 outputs forward (8 x 11 window), 2 x 7 pixels backward; stride 2: 2 x 4 outputs forward
// (7 x 11 window), 4 x 8 input pixels backward (4 x 6 window of dy).  K = 3 -- 7 x 7 forward (9 x 9 window), 4 x 7 backward; stride 2:
// 2 x 7 outputs forward (5 x 15 window), 4 x 8 input pixels backward (3 x 5 window).
template <int K> struct TileShape;
template <> struct TileShape<5> { static constexpr int FH = 4, FW = 7, BH = 2, BW = 7, F2H = 2, F2W = 4, B2H = 2, B2W = 4; };
template <> struct TileShape<3> { static constexpr int FH = 7, FW = 7, BH = 4, BW = 7, F2H = 2, F2W = 7, B2H = 2, B2W = 4; };

#define T3D_TILE_ACT(KERNEL, ...)                                                                                     \
  switch (act) {                                                                                                      \
    case T3D_ACT_RELU: T3D_LAUNCH_TIMED((KERNEL<T, T3D_ACT_RELU, __VA_ARGS__>), grid, dim3(256), lds, st, a); break;    \
    case T3D_ACT_RELU6: T3D_LAUNCH_TIMED((KERNEL<T, T3D_ACT_RELU6, __VA_ARGS__>), grid, dim3(256), lds, st, a); break;  \
    case T3D_ACT_HSWISH: T3D_LAUNCH_TIMED((KERNEL<T, T3D_ACT_HSWISH, __VA_ARGS__>), grid, dim3(256), lds, st, a); break; \
    default: T3D_LAUNCH_TIMED((KERNEL<T, T3D_ACT_NONE, __VA_ARGS__>), grid, dim3(256), lds, st, a); break;              \
  }

template <typename T, int K>
int launch_fwd(TileArgs& a, int act, int stride, hipStream_t st) {
  typedef TileShape<K> S;
  const int Ho = stride == 2 ? (a.H + 2 * (K / 2) - K) / 2 + 1 : a.H, Wo = stride == 2 ? (a.W + 2 * (K / 2) - K) / 2 + 1 : a.W;
  const dim3 grid = stride == 2 ? tile_grid<S::F2H, S::F2W>(a, 4096, Ho, Wo) : tile_grid<S::FH, S::FW>(a, 4096);
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  if (a.stats && a.nrep < 1) { a.nrep = 1; a.rstride = 0; }
  a.quant = (a.stats && std::is_same<T, bf16_t>::value && !T3D_ENV_SET("T3D_NO_SNAP")) ? t3d_quant_for((long long)a.B * Ho * Wo) : T3dQuant{0.0, 0.0};
  const size_t lds = 0;
  if (stride == 2) { T3D_TILE_ACT(dw5_tile_fwd_s2_kernel, K, S::F2H, S::F2W) }
  else { T3D_TILE_ACT(dw5_tile_fwd_kernel, K, S::FH, S::FW) }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <typename T, int K>
int launch_bwd(TileArgs& a, int act, int stride, hipStream_t st) {
  typedef TileShape<K> S;
  dim3 grid = stride == 2 ? tile_grid<2 * S::B2H, 2 * S::B2W>(a, 4096) : tile_grid<S::BH, S::BW>(a, K == 5 ? 2048 : 4096);
  // one weight-gradient slot per workgroup column (t3d_set_dw_slots): never more columns than slots -- past that the flush falls
  // back to fp32 atomics and the step is no longer bit-reproducible (56x56x72 stride 2 wanted 1024 against the engine's 512)
  if (a.dw && g_t3d_reduce.dw_slots > 0 && (int)grid.x > g_t3d_reduce.dw_slots) grid.x = g_t3d_reduce.dw_slots;
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  if (a.nrep < 1) { a.nrep = 1; a.rstride = 0; }
  a.dw_slots = (a.dw && g_t3d_reduce.dw_slots >= (int)grid.x) ? (int)grid.x : 0;
  a.dw_used = a.dw ? g_t3d_reduce.dw_used : nullptr;
  const size_t lds = (size_t)K * K * SLAB * sizeof(float) + (size_t)(K * K + 2) * SLAB * sizeof(double);
  if (stride == 2) { T3D_TILE_ACT(dw5_tile_bwd_s2_kernel, K, S::B2H, S::B2W) }
  else { T3D_TILE_ACT(dw5_tile_bwd_kernel, K, S::BH, S::BW) }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
#undef T3D_TILE_ACT

// which planes take the tile kernels (isolated, B = 256, us tile | row walk; profiles/r6_*_dw5_isolated.txt, _dw3_small_planes_*):
//   5x5 s1 28x28x120: forward 48 | 58 (79 before the chunk fix), backward 116 | 185;  5x5 s2 14x14x672: 55 | 74, 117 | 163;
//   5x5 s2 56x56x72: 96 | 90, 171 | 178 -- no gain, stays on the walk;  7x7 planes: dwconv5_plane7.hip (17 | 67, 50 | 116).
//   3x3: the tiles LOSE to the row-walk kernels of dwconv3_stream.hip / dwconv3_bwd_stream.hip on every plane of MobileNetV2
//   (28x28x192: 50 | 36 forward, 127 | 85 backward; 14x14x384: 27 | 24, 66 | 42; 7x7x960: 18 | 19, 47 | 34) -- those kernels
//   are at 3.2-4.4 TB/s alone, their in-step times are contention, not structure.  OPT-IN: T3D_DW3_TILE_MAX = largest plane side.
static int dw3_tile_max() {      // (read per call: the test suite flips it inside one process, like T3D_DW_TILED)
  const char* e = getenv("T3D_DW3_TILE_MAX");
  return e ? atoi(e) : 0;
}
static bool tile_shape_ok(int k, int stride, int H, int W, int C, bool pooled = false) {
  if ((C % 2) || H < 2 || W < 2) return false;
  if (k == 5) return H >= 8 && W >= 8 && H <= (stride == 2 ? 28 : 64) && W <= (stride == 2 ? 28 : 64) && !T3D_ENV_SET("T3D_DW5_NO_TILE");
  // 3x3 forward WITH squeeze-excite pooled sums on the 14x14 stage and below: those launches go to the generic walk of
  // dwconvk_stream.hip (the 3x3 row-walk kernel has no pooled sums), which the tiles beat: 14x14x480 52 -> 36 us, x672 59 -> 51
  if (k == 3 && pooled && stride == 1 && H <= 14 && W <= 14 && !T3D_ENV_SET("T3D_DW5_NO_TILE")) return true;
  if (k == 3) return H <= dw3_tile_max() && W <= dw3_tile_max();
  return false;
}

}  // namespace

int t3d_dw_tile_fwd(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats, float* gap_sum,
                    int B, int H, int W, int C, int k, int stride, hipStream_t st) {
  if (!tile_shape_ok(k, stride, H, W, C, gap_sum != nullptr) || (pro && pro->se) || (stride != 1 && stride != 2)) return T3D_ERR_UNSUPPORTED;
  if (pro)
    if (const int rc = t3d_fold_fallback(pro->scale, st)) return rc;      // finished coefficients (no derive prologue here)
  TileArgs a{};
  a.x = x; a.y = y; a.w = w; a.stats = stats; a.gap = gap_sum; a.gapq = g_t3d_reduce.pool_exact;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; }
  a.B = B; a.H = H; a.W = W; a.C = C;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  if (k == 5) {
    if (dtype == T3D_F32) return launch_fwd<float, 5>(a, act, stride, st);
    if (dtype == T3D_BF16) return launch_fwd<bf16_t, 5>(a, act, stride, st);
  } else {
    if (dtype == T3D_F32) return launch_fwd<float, 3>(a, act, stride, st);
    if (dtype == T3D_BF16) return launch_fwd<bf16_t, 3>(a, act, stride, st);
  }
  return T3D_ERR_UNSUPPORTED;
}

int t3d_dw_tile_bwd(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                    const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int H, int W, int C,
                    int k, int stride, hipStream_t st) {
  if (!tile_shape_ok(k, stride, H, W, C) || (pro && pro->se) || (stride != 1 && stride != 2)) return T3D_ERR_UNSUPPORTED;
  if (const int rc = t3d_fold_fallback(bb->alpha, st)) return rc;
  TileArgs a{};
  a.dz = dz; a.yraw = y; a.x = x; a.res = residual; a.dx = dx; a.w = w;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma; a.per_sample = bb->per_sample;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; }
  a.stats = stats; a.dw = dw; a.B = B; a.H = H; a.W = W; a.C = C;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  if (k == 5) {
    if (dtype == T3D_F32) return launch_bwd<float, 5>(a, act, stride, st);
    if (dtype == T3D_BF16) return launch_bwd<bf16_t, 5>(a, act, stride, st);
  } else {
    if (dtype == T3D_F32) return launch_bwd<float, 3>(a, act, stride, st);
    if (dtype == T3D_BF16) return launch_bwd<bf16_t, 3>(a, act, stride, st);
  }
  return T3D_ERR_UNSUPPORTED;
}
