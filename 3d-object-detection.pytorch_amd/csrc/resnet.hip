// Layers a ResNet-50 backbone needs beside the 1x1 convolutions the regression path already has (BASELINE config 4:
// "ResNet-50 backbone via torchdet3d.builders"; the reference itself has no ResNet -- SURVEY.md section 0 -- so the
// architecture is the standard torchvision one and parity is against oracle/resnet.py, unpinned):
//   * dense k x k convolution (3x3 in the bottlenecks, 7x7 stem) = patch gather + the pointwise GEMM kernels:
//       t3d_im2col       x [B,H,W,C] (raw, the producer's BatchNorm + activation applied on load, zero padding AFTER the
//                        activation) -> col [B*Ho*Wo][Kp], column (ky*k + kx)*C + c, columns >= k*k*C zero
//       t3d_col2im_bwd   gradient of the patch matrix -> gradient at the producer's BatchNorm OUTPUT (times act'), plus
//                        that BatchNorm's backward sums -- the gather form (each input pixel collects its <= k*k taps)
//       t3d_pack_conv_weight / t3d_unpack_conv_grad   [N][C][k][k] fp32 <-> [N][Kp] in patch-column order
//   * 3x3 / stride-2 max-pool over the activated stem output (arg-max kept for the backward)
//   * the bottleneck's tail  z = relu(BN3(y3) + shortcut)  and its backward  g = dz * [z > 0]  with the BatchNorm-backward
//     sums of both branches
//   * stride-2 sub-sampling in front of the shortcut's 1x1 conv, and the zero-filled up-sampling of its gradient.
// All of them are HBM-bound elementwise / gather kernels: one thread per (pixel, 8-channel vector) where C % 8 == 0,
// scalar otherwise (the 3-channel stem input).  fp32 or bf16 storage.
#include <algorithm>

#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

// ---- im2col ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int act, T* __restrict__ col, int B,
                                                     int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int Kp) {
  const long long total = (long long)B * Ho * Wo * Kp;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int kc = (int)(e % Kp);
    const long long m = e / Kp;
    float v = 0.f;
    if (kc < k * k * C) {
      const int c = kc % C, t = kc / C, ky = t / k, kx = t - ky * k;
      const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long long)Wo * Ho));
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        v = ldf(x + (((size_t)b * H + iy) * W + ix) * C + c);
        if (scale) v = fmaf(v, scale[c], shift[c]);
        v = act_apply(v, act);
      }
    }
    col[e] = (T)v;
  }
}

// fp32 NCHW image (the reference's input contract) -> patch matrix of the stem
template <typename T>
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, T* __restrict__ col, int B, int H, int W,
                                                          int C, int k, int stride, int pad, int Ho, int Wo, int Kp) {
  const long long total = (long long)B * Ho * Wo * Kp;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int kc = (int)(e % Kp);
    const long long m = e / Kp;
    float v = 0.f;
    if (kc < k * k * C) {
      const int c = kc % C, t = kc / C, ky = t / k, kx = t - ky * k;
      const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long long)Wo * Ho));
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * C + c) * H + iy) * W + ix];
    }
    col[e] = (T)v;
  }
}

// ---- col2im backward (gather form) --------------------------------------------------------------------------------
// thread = (input pixel, channel); dx = sum over the taps that read this pixel; times act'(scale*x + shift); sums
template <typename T>
__global__ __launch_bounds__(256) void col2im_bwd_kernel(const T* __restrict__ dcol, const T* __restrict__ xraw,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                         T* __restrict__ dx, double* __restrict__ stats, int B, int H, int W, int C,
                                                         int k, int stride, int pad, int Ho, int Wo, int Kp, int nrep, long long rstride) {
  extern __shared__ float lst[];      // [2][C]
  for (int i = threadIdx.x; i < 2 * C; i += 256) lst[i] = 0.f;
  __syncthreads();
  const long long total = (long long)B * H * W * C;
  // a thread keeps one channel (stride of the grid is a multiple of C): its sums stay in registers
  const long long nthr = ((long long)gridDim.x * 256 / C) * C;
  const long long g0 = blockIdx.x * 256LL + threadIdx.x;
  float s1 = 0.f, s2 = 0.f;
  const int c = (int)(g0 % C);
  if (g0 < nthr) {
    for (long long e = g0; e < total; e += nthr) {
      const long long px = e / C;
      const int ix = (int)(px % W), iy = (int)((px / W) % H), b = (int)(px / ((long long)W * H));
      float acc = 0.f;
      for (int ky = 0; ky < k; ++ky) {
        const int ty = iy + pad - ky;
        if (ty < 0 || ty % stride) continue;
        const int oy = ty / stride;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < k; ++kx) {
          const int tx = ix + pad - kx;
          if (tx < 0 || tx % stride) continue;
          const int ox = tx / stride;
          if (ox >= Wo) continue;
          acc += ldf(dcol + (((size_t)b * Ho + oy) * Wo + ox) * Kp + (ky * k + kx) * C + c);
        }
      }
      const float xr = ldf(xraw + e);
      const float u = scale ? fmaf(xr, scale[c], shift[c]) : xr;
      const T o = (T)(acc * act_grad(u, act));
      dx[e] = o;
      const float ov = (float)o;
      s1 += ov;
      s2 = fmaf(ov, xr, s2);
    }
  }
  if (stats) {
    if (g0 < nthr) {
      atomicAdd(lst + c, s1);
      atomicAdd(lst + C + c, s2);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256)
      if (lst[i] != 0.f) atomicAdd(stats + (size_t)(blockIdx.x % nrep) * rstride + i, (double)lst[i]);
  }
}

// ---- conv weight layout ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ w, T* __restrict__ out, int N, int C, int k, int Kp) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * Kp) return;
  const int n = e / Kp, kc = e - n * Kp;
  float v = 0.f;
  if (kc < k * k * C) {
    const int c = kc % C, t = kc / C;
    v = w[((size_t)n * C + c) * k * k + t];
  }
  out[e] = (T)v;
}
__global__ void unpack_conv_grad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int N, int C, int k, int Kp) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * C * k * k) return;
  const int t = e % (k * k), c = (e / (k * k)) % C, n = e / (k * k * C);
  dw[e] = dwp[(size_t)n * Kp + t * C + c];
}

// ---- max-pool 3x3 / stride 2 / pad 1 over act(scale*y + shift) ---------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act, T* __restrict__ out,
                                                          unsigned char* __restrict__ idx, int B, int H, int W, int C, int Ho, int Wo) {
  const long long total = (long long)B * Ho * Wo * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long long px = e / C;
    const int ox = (int)(px % Wo), oy = (int)((px / Wo) % Ho), b = (int)(px / ((long long)Wo * Ho));
    float best = -3.0e38f;
    int bi = 0;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        float v = ldf(y + (((size_t)b * H + iy) * W + ix) * C + c);
        if (scale) v = fmaf(v, scale[c], shift[c]);
        v = act_apply(v, act);
        if (v > best) { best = v; bi = ky * 3 + kx; }       // first maximum in scan order (PyTorch's choice)
      }
    }
    out[e] = (T)best;
    idx[e] = (unsigned char)bi;
  }
}

// gradient at the BatchNorm output of the pooled tensor's producer: each input pixel collects the windows whose maximum it is
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ idx,
                                                          const T* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act, T* __restrict__ dy,
                                                          double* __restrict__ stats, int B, int H, int W, int C, int Ho, int Wo,
                                                          int nrep, long long rstride) {
  extern __shared__ float lst[];
  for (int i = threadIdx.x; i < 2 * C; i += 256) lst[i] = 0.f;
  __syncthreads();
  const long long total = (long long)B * H * W * C;
  const long long nthr = ((long long)gridDim.x * 256 / C) * C;
  const long long g0 = blockIdx.x * 256LL + threadIdx.x;
  const int c = (int)(g0 % C);
  float s1 = 0.f, s2 = 0.f;
  if (g0 < nthr) {
    for (long long e = g0; e < total; e += nthr) {
      const long long px = e / C;
      const int ix = (int)(px % W), iy = (int)((px / W) % H), b = (int)(px / ((long long)W * H));
      float acc = 0.f;
      for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + 1 - ky;
        if (ty < 0 || (ty & 1)) continue;
        const int oy = ty >> 1;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int tx = ix + 1 - kx;
          if (tx < 0 || (tx & 1)) continue;
          const int ox = tx >> 1;
          if (ox >= Wo) continue;
          const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + c;
          if (idx[o] == ky * 3 + kx) acc += ldf(dout + o);
        }
      }
      const float yr = ldf(y + e);
      const float u = scale ? fmaf(yr, scale[c], shift[c]) : yr;
      const T o = (T)(acc * act_grad(u, act));
      dy[e] = o;
      const float ov = (float)o;
      s1 += ov;
      s2 = fmaf(ov, yr, s2);
    }
  }
  if (stats) {
    if (g0 < nthr) {
      atomicAdd(lst + c, s1);
      atomicAdd(lst + C + c, s2);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256)
      if (lst[i] != 0.f) atomicAdd(stats + (size_t)(blockIdx.x % nrep) * rstride + i, (double)lst[i]);
  }
}

// ---- bottleneck tail: z = relu(s3*y3 + t3 + shortcut) ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void res_relu_fwd_kernel(const T* __restrict__ y3, const float* __restrict__ s3,
                                                           const float* __restrict__ t3, const T* __restrict__ sh,
                                                           const float* __restrict__ ss, const float* __restrict__ ts,
                                                           T* __restrict__ z, long long total, int C) {
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % C);
    float v = fmaf(ldf(y3 + e), s3[c], t3[c]);
    float r = ldf(sh + e);
    if (ss) r = fmaf(r, ss[c], ts[c]);
    z[e] = (T)fmaxf(v + r, 0.f);
  }
}

// g = dz * [z > 0];  stats3 += sum g, sum g*y3;  statsd += sum g, sum g*yd (projection shortcut only)
template <typename T>
__global__ __launch_bounds__(256) void res_relu_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ z,
                                                           const T* __restrict__ y3, const T* __restrict__ yd, T* __restrict__ g,
                                                           double* __restrict__ stats3, double* __restrict__ statsd,
                                                           long long total, int C, int nrep, long long rstride) {
  extern __shared__ float lst[];      // [4][C]
  for (int i = threadIdx.x; i < 4 * C; i += 256) lst[i] = 0.f;
  __syncthreads();
  const long long nthr = ((long long)gridDim.x * 256 / C) * C;
  const long long g0 = blockIdx.x * 256LL + threadIdx.x;
  const int c = (int)(g0 % C);
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (g0 < nthr) {
    for (long long e = g0; e < total; e += nthr) {
      const T o = (T)(ldf(z + e) > 0.f ? ldf(dz + e) : 0.f);
      g[e] = o;
      const float ov = (float)o;
      s1 += ov;
      s2 = fmaf(ov, ldf(y3 + e), s2);
      if (yd) s3 = fmaf(ov, ldf(yd + e), s3);
    }
    atomicAdd(lst + c, s1);
    atomicAdd(lst + C + c, s2);
    if (yd) atomicAdd(lst + 2 * C + c, s3);
  }
  __syncthreads();
  const size_t rep = (size_t)(blockIdx.x % nrep) * rstride;
  for (int i = threadIdx.x; i < 2 * C; i += 256)
    if (lst[i] != 0.f) atomicAdd(stats3 + rep + i, (double)lst[i]);
  if (statsd) {
    for (int i = threadIdx.x; i < C; i += 256) {
      if (lst[i] != 0.f) atomicAdd(statsd + rep + i, (double)lst[i]);
      if (lst[2 * C + i] != 0.f) atomicAdd(statsd + rep + C + i, (double)lst[2 * C + i]);
    }
  }
}

// ---- stride-2 sub-sampling / zero-filled up-sampling ------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void subsample_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C,
                                                        int s, int Ho, int Wo, int up) {
  // up == 0: out[b,oy,ox,:] = x[b,s*oy,s*ox,:]   (out is the small tensor)
  // up == 1: out[b,iy,ix,:] = (iy % s == 0 && ix % s == 0) ? x[b,iy/s,ix/s,:] : 0   (out is the large tensor, x the small one)
  const long long total = up ? (long long)B * H * W * C : (long long)B * Ho * Wo * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long long px = e / C;
    if (!up) {
      const int ox = (int)(px % Wo), oy = (int)((px / Wo) % Ho), b = (int)(px / ((long long)Wo * Ho));
      out[e] = x[(((size_t)b * H + oy * s) * W + ox * s) * C + c];
    } else {
      const int ix = (int)(px % W), iy = (int)((px / W) % H), b = (int)(px / ((long long)W * H));
      const bool on = (iy % s == 0) && (ix % s == 0) && iy / s < Ho && ix / s < Wo;
      out[e] = on ? x[(((size_t)b * Ho + iy / s) * Wo + ix / s) * C + c] : (T)0.f;
    }
  }
}

// ==== 8-channel vector forms (C % 8 == 0): one lane moves 16 bytes of bf16 / 32 bytes of fp32 ============================
// Every wave owns a CONTIGUOUS run of rows (output pixels for the gather, input pixels for the backward forms) and walks its
// (b, y, x) coordinates incrementally -- no integer division inside the loops; a lane keeps one 8-channel group for the
// whole launch, so the BatchNorm-backward sums stay in registers until the end.

// tap -> ky for small k:  (tap * (65536 / k + 1)) >> 16  is exact for tap < k * k <= 49 * 49
__device__ __forceinline__ int tap_row(int tap, int kinv) { return (tap * kinv) >> 16; }

struct RowWalk {          // (b, y, x) of row `m` of a [B][Hh][Ww] raster, advanced by a constant step
  int b, y, x;
  __device__ __forceinline__ void init(unsigned m, int Hh, int Ww) {
    const unsigned t = m / (unsigned)Ww;
    x = (int)(m - t * (unsigned)Ww);
    b = (int)(t / (unsigned)Hh);
    y = (int)(t - (unsigned)b * (unsigned)Hh);
  }
  __device__ __forceinline__ void step(int d, int Hh, int Ww) {
    x += d;
    while (x >= Ww) {
      x -= Ww;
      if (++y == Hh) { y = 0; ++b; }
    }
  }
};

template <typename T>
__global__ __launch_bounds__(256) void im2col_v_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int act, T* __restrict__ col, int B,
                                                       int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int Kp,
                                                       int cshift, int kinv, int rows_per_wave) {
  extern __shared__ float lco[];      // [2][C] scale, shift
  if (scale) {
    for (int i = threadIdx.x; i < C; i += 256) { lco[i] = scale[i]; lco[C + i] = shift[i]; }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const unsigned M = (unsigned)B * Ho * Wo;
  const unsigned r0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)rows_per_wave;
  if (r0 >= M) return;
  const unsigned r1 = min(M, r0 + (unsigned)rows_per_wave);
  const int KK = k * k * C, Kp8 = Kp >> 3;
  RowWalk rw;
  rw.init(r0, Ho, Wo);
  for (unsigned m = r0; m < r1; ++m) {
    const T* xb = x + (size_t)rw.b * H * W * C;
    const int iy0 = rw.y * stride - pad, ix0 = rw.x * stride - pad;
    T* crow = col + (size_t)m * Kp;
    for (int kc8 = lane; kc8 < Kp8; kc8 += 64) {
      const int kc = kc8 << 3;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
      if (kc < KK) {
        const int tap = cshift >= 0 ? (kc >> cshift) : kc / C;
        const int c = kc - tap * C, ky = tap_row(tap, kinv), kx = tap - ky * k;
        const int iy = iy0 + ky, ix = ix0 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          Vec8<T>::load(xb + ((size_t)iy * W + ix) * C + c, v);
          if (scale) act_affine_vec<8>(v, lco + c, lco + C + c, act);
          else if (act != T3D_ACT_NONE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = act_apply(v[i], act);
          }
        }
      }
      Vec8<T>::store(crow + kc, v);
    }
    rw.step(1, Ho, Wo);
  }
}

// fp32 NCHW image -> patch matrix; a lane gathers 8 consecutive patch columns; 64 / (Kp/8) rows per wave iteration
template <typename T>
__global__ __launch_bounds__(256) void im2col_nchw_v_kernel(const float* __restrict__ x, T* __restrict__ col, int B, int H, int W,
                                                            int C, int k, int stride, int pad, int Ho, int Wo, int Kp, int cinv,
                                                            int kinv, int rows_per_wave) {
  const int lane = threadIdx.x & 63, Kp8 = Kp >> 3;
  const int sub = lane / Kp8, kc8 = lane - sub * Kp8, RPI = 64 / Kp8;      // host guarantees Kp8 <= 64
  if (sub >= RPI) return;
  const unsigned M = (unsigned)B * Ho * Wo;
  const unsigned r0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)rows_per_wave;
  if (r0 + sub >= M) return;
  const unsigned r1 = min(M, r0 + (unsigned)rows_per_wave);
  const int KK = k * k * C;
  RowWalk rw;
  rw.init(r0 + sub, Ho, Wo);
  for (unsigned m = r0 + sub; m < r1; m += RPI) {
    const float* xb = x + (size_t)rw.b * C * H * W;
    const int iy0 = rw.y * stride - pad, ix0 = rw.x * stride - pad;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kc = (kc8 << 3) + i;
      const int tap = (kc * cinv) >> 16, c = kc - tap * C;
      const int ky = tap_row(tap, kinv), kx = tap - ky * k;
      const int iy = iy0 + ky, ix = ix0 + kx;
      v[i] = (kc < KK && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xb[((size_t)c * H + iy) * W + ix] : 0.f;
    }
    Vec8<T>::store(col + (size_t)m * Kp + (kc8 << 3), v);
    rw.step(RPI, Ho, Wo);
  }
}

// block-level tail of the backward forms: per-lane fp32 register sums -> fp64 LDS accumulators (exact adds in any order: the
// BatchNorm-backward coefficients, and with them every gradient downstream, are bit-reproducible) -> one fp64 atomic per
// channel and workgroup
template <int NS>
__device__ __forceinline__ void flush_sums(double* lst, int C, int c, bool active, const float (*s)[8]) {
  if (active) {
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(lst + j * C + c + i, (double)s[j][i]);
  }
  __syncthreads();
}

template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void col2im_bwd_v_kernel(const T* __restrict__ dcol, const T* __restrict__ xraw,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                           T* __restrict__ dx, double* __restrict__ stats, int B, int H, int W, int C,
                                                           int k, int pad, int Ho, int Wo, int Kp, int nrep, long long rstride,
                                                           int px_per_wave) {
  extern __shared__ double lacc[];      // [2][C] sums
  for (int i = threadIdx.x; i < 2 * C; i += 256) lacc[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, C8 = C >> 3;                    // host guarantees C8 <= 64
  const int sub = lane / C8, c = (lane - sub * C8) << 3, PPI = 64 / C8;
  const unsigned P = (unsigned)B * H * W;
  const unsigned p0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)px_per_wave;
  const bool active = sub < PPI && p0 + sub < P;
  float s[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[0][i] = s[1][i] = 0.f;
  if (active) {
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = scale ? scale[c + i] : 1.f; sh[i] = scale ? shift[c + i] : 0.f; }
    const unsigned p1 = min(P, p0 + (unsigned)px_per_wave);
    RowWalk rw;
    rw.init(p0 + sub, H, W);
    for (unsigned p = p0 + sub; p < p1; p += PPI) {
      float acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      const T* db = dcol + (size_t)rw.b * Ho * Wo * Kp + c;
      for (int ky = 0; ky < k; ++ky) {
        const int ty = rw.y + pad - ky;
        if (ty < 0 || (STRIDE == 2 && (ty & 1))) continue;
        const int oy = STRIDE == 2 ? ty >> 1 : ty;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < k; ++kx) {
          const int tx = rw.x + pad - kx;
          if (tx < 0 || (STRIDE == 2 && (tx & 1))) continue;
          const int ox = STRIDE == 2 ? tx >> 1 : tx;
          if (ox >= Wo) continue;
          float d[8];
          Vec8<T>::load(db + ((size_t)oy * Wo + ox) * Kp + (ky * k + kx) * C, d);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += d[i];
        }
      }
      float xr[8];
      Vec8<T>::load(xraw + (size_t)p * C + c, xr);
      act_grad_affine_vec<8>(acc, xr, sc, sh, act);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = Vec8<T>::round(acc[i]);
      Vec8<T>::store(dx + (size_t)p * C + c, acc);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[0][i] += acc[i]; s[1][i] = fmaf(acc[i], xr[i], s[1][i]); }
      rw.step(PPI, H, W);
    }
  }
  if (stats) {
    flush_sums<2>(lacc, C, c, active, s);
    for (int i = threadIdx.x; i < 2 * C; i += 256)
      if (lacc[i] != 0.f) atomicAdd(stats + (size_t)(blockIdx.x % nrep) * rstride + i, (double)lacc[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_v_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int act, T* __restrict__ out,
                                                            unsigned char* __restrict__ idx, int B, int H, int W, int C, int Ho, int Wo,
                                                            int px_per_wave) {
  const int lane = threadIdx.x & 63, C8 = C >> 3;
  const int sub = lane / C8, c = (lane - sub * C8) << 3, PPI = 64 / C8;
  const unsigned P = (unsigned)B * Ho * Wo;
  const unsigned p0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)px_per_wave;
  if (sub >= PPI || p0 + sub >= P) return;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sc[i] = scale ? scale[c + i] : 1.f; sh[i] = scale ? shift[c + i] : 0.f; }
  const unsigned p1 = min(P, p0 + (unsigned)px_per_wave);
  RowWalk rw;
  rw.init(p0 + sub, Ho, Wo);
  for (unsigned p = p0 + sub; p < p1; p += PPI) {
    float best[8];
    unsigned bi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { best[i] = -3.0e38f; bi[i] = 0; }
    const T* yb = y + (size_t)rw.b * H * W * C + c;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * rw.y - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * rw.x - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        float v[8];
        Vec8<T>::load(yb + ((size_t)iy * W + ix) * C, v);
        act_affine_vec<8>(v, sc, sh, act);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (v[i] > best[i]) { best[i] = v[i]; bi[i] = ky * 3 + kx; }       // first maximum in scan order (PyTorch's choice)
      }
    }
    Vec8<T>::store(out + (size_t)p * C + c, best);
    uint2 pk;
    pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + (size_t)p * C + c) = pk;
    rw.step(PPI, Ho, Wo);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_v_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ idx,
                                                            const T* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int act, T* __restrict__ dy,
                                                            double* __restrict__ stats, int B, int H, int W, int C, int Ho, int Wo,
                                                            int nrep, long long rstride, int px_per_wave) {
  extern __shared__ double lacc[];
  for (int i = threadIdx.x; i < 2 * C; i += 256) lacc[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, C8 = C >> 3;
  const int sub = lane / C8, c = (lane - sub * C8) << 3, PPI = 64 / C8;
  const unsigned P = (unsigned)B * H * W;
  const unsigned p0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)px_per_wave;
  const bool active = sub < PPI && p0 + sub < P;
  float s[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[0][i] = s[1][i] = 0.f;
  if (active) {
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = scale ? scale[c + i] : 1.f; sh[i] = scale ? shift[c + i] : 0.f; }
    const unsigned p1 = min(P, p0 + (unsigned)px_per_wave);
    RowWalk rw;
    rw.init(p0 + sub, H, W);
    for (unsigned p = p0 + sub; p < p1; p += PPI) {
      float acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      for (int ky = 0; ky < 3; ++ky) {
        const int ty = rw.y + 1 - ky;
        if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int tx = rw.x + 1 - kx;
          if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
          const size_t o = (((size_t)rw.b * Ho + (ty >> 1)) * Wo + (tx >> 1)) * C + c;
          const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
          float d[8];
          Vec8<T>::load(dout + o, d);
          const unsigned want = ky * 3 + kx;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const unsigned bi = ((i < 4 ? pk.x : pk.y) >> (8 * (i & 3))) & 255u;
            if (bi == want) acc[i] += d[i];
          }
        }
      }
      float yr[8];
      Vec8<T>::load(y + (size_t)p * C + c, yr);
      act_grad_affine_vec<8>(acc, yr, sc, sh, act);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = Vec8<T>::round(acc[i]);
      Vec8<T>::store(dy + (size_t)p * C + c, acc);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[0][i] += acc[i]; s[1][i] = fmaf(acc[i], yr[i], s[1][i]); }
      rw.step(PPI, H, W);
    }
  }
  if (stats) {
    flush_sums<2>(lacc, C, c, active, s);
    for (int i = threadIdx.x; i < 2 * C; i += 256)
      if (lacc[i] != 0.f) atomicAdd(stats + (size_t)(blockIdx.x % nrep) * rstride + i, (double)lacc[i]);
  }
}

// flat forms: vector index e8 over [M][C/8]; the grid's thread count is a multiple of C/8, so a thread keeps its channels
template <typename T>
__global__ __launch_bounds__(256) void res_relu_fwd_v_kernel(const T* __restrict__ y3, const float* __restrict__ s3,
                                                             const float* __restrict__ t3, const T* __restrict__ sh,
                                                             const float* __restrict__ ss, const float* __restrict__ ts,
                                                             T* __restrict__ z, long long total8, int C8) {
  const long long nthr = ((long long)gridDim.x * 256 / C8) * C8;
  const long long g0 = blockIdx.x * 256LL + threadIdx.x;
  if (g0 >= nthr) return;
  const int c = (int)(g0 % C8) << 3;
  float a3[8], b3[8], as[8], bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a3[i] = s3[c + i]; b3[i] = t3[c + i]; as[i] = ss ? ss[c + i] : 1.f; bs[i] = ss ? ts[c + i] : 0.f; }
  for (long long e = g0; e < total8; e += nthr) {
    float v[8], r[8];
    Vec8<T>::load(y3 + e * 8, v);
    Vec8<T>::load(sh + e * 8, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float rr = ss ? fmaf(r[i], as[i], bs[i]) : r[i];
      v[i] = fmaxf(fmaf(v[i], a3[i], b3[i]) + rr, 0.f);
    }
    Vec8<T>::store(z + e * 8, v);
  }
}

template <typename T, bool PROJ>
__global__ __launch_bounds__(256) void res_relu_bwd_v_kernel(const T* __restrict__ dz, const T* __restrict__ z,
                                                             const T* __restrict__ y3, const T* __restrict__ yd, T* __restrict__ g,
                                                             double* __restrict__ stats3, double* __restrict__ statsd,
                                                             long long total8, int C, int nrep, long long rstride) {
  extern __shared__ double lacc[];      // [3][C]
  for (int i = threadIdx.x; i < 3 * C; i += 256) lacc[i] = 0.0;
  __syncthreads();
  const int C8 = C >> 3;
  const long long nthr = ((long long)gridDim.x * 256 / C8) * C8;
  const long long g0 = blockIdx.x * 256LL + threadIdx.x;
  const bool active = g0 < nthr;
  const int c = (int)(g0 % C8) << 3;
  float s[3][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[0][i] = s[1][i] = s[2][i] = 0.f;
  if (active) {
    for (long long e = g0; e < total8; e += nthr) {
      float zz[8], d[8], a[8], b[8];
      Vec8<T>::load(z + e * 8, zz);
      Vec8<T>::load(dz + e * 8, d);
      Vec8<T>::load(y3 + e * 8, a);
      if (PROJ) Vec8<T>::load(yd + e * 8, b);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        d[i] = zz[i] > 0.f ? d[i] : 0.f;
        s[0][i] += d[i];
        s[1][i] = fmaf(d[i], a[i], s[1][i]);
        if (PROJ) s[2][i] = fmaf(d[i], b[i], s[2][i]);
      }
      Vec8<T>::store(g + e * 8, d);
    }
  }
  flush_sums<PROJ ? 3 : 2>(lacc, C, c, active, s);
  const size_t rep = (size_t)(blockIdx.x % nrep) * rstride;
  for (int i = threadIdx.x; i < 2 * C; i += 256)
    if (lacc[i] != 0.f) atomicAdd(stats3 + rep + i, (double)lacc[i]);
  if (PROJ) {
    for (int i = threadIdx.x; i < C; i += 256) {
      if (lacc[i] != 0.f) atomicAdd(statsd + rep + i, (double)lacc[i]);
      if (lacc[2 * C + i] != 0.f) atomicAdd(statsd + rep + C + i, (double)lacc[2 * C + i]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void subsample_v_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C8,
                                                          int s, int Ho, int Wo, int up, unsigned total8) {
  // one 16/32-byte vector per thread and iteration; coordinates by 32-bit division (3 per vector, the kernel is short)
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total8; e += gridDim.x * 256u) {
    const unsigned px = e / (unsigned)C8, cv = e - px * (unsigned)C8;
    RowWalk rw;
    rw.init(px, up ? H : Ho, up ? W : Wo);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    if (!up) {
      Vec8<T>::load(x + ((((size_t)rw.b * H + rw.y * s) * W + rw.x * s) * C8 + cv) * 8, v);
    } else {
      const int oy = rw.y / s, ox = rw.x / s;
      if (oy * s == rw.y && ox * s == rw.x && oy < Ho && ox < Wo)
        Vec8<T>::load(x + ((((size_t)rw.b * Ho + oy) * Wo + ox) * C8 + cv) * 8, v);
    }
    Vec8<T>::store(out + (size_t)e * 8, v);
  }
}

inline int grid_for(long long total) {
  long long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}
// a grid whose thread count is a multiple of C and covers at least one thread per channel
inline int grid_for_c(long long total, int C) {
  int g = grid_for(total);
  while ((long long)g * 256 < C) ++g;
  return g;
}

// contiguous run of rows per wave so that about `waves` waves cover `rows`; returns the grid (4 waves per workgroup)
inline int runs_for(long long rows, int waves, int* per_wave) {
  long long r = (rows + waves - 1) / waves;
  if (r < 1) r = 1;
  *per_wave = (int)r;
  return (int)((rows + 4 * r - 1) / (4 * r));
}
inline int log2_exact(int v) {
  for (int s = 0; s < 31; ++s)
    if ((1 << s) == v) return s;
  return -1;
}
inline bool fits_u32(long long rows) { return rows < (1LL << 31); }

}  // namespace

#define T3D_DISPATCH(dtype, CALL_F32, CALL_BF16) \
  do {                                            \
    if ((dtype) == T3D_F32) { CALL_F32; }         \
    else if ((dtype) == T3D_BF16) { CALL_BF16; }  \
    else return T3D_ERR_ARG;                      \
  } while (0)

extern "C" int t3d_im2col(int dtype, const void* x, const t3d_prologue* pro, void* col, int B, int H, int W, int C, int k,
                          int stride, int pad, int Kp, void* stream) {
  if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || Kp < k * k * C) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pro)
    if (const int rc = t3d_fold_fallback(pro->scale, st)) return rc;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const float* sc = pro ? pro->scale : nullptr;
  const float* sh = pro ? pro->shift : nullptr;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  const long long M = (long long)B * Ho * Wo;
  if (C % 8 == 0 && Kp % 8 == 0 && k <= 48 && fits_u32(M)) {
    int rpw;
    const int g = runs_for(M, 8192, &rpw), cs = log2_exact(C), kinv = 65536 / k + 1;
    const size_t lds = sc ? (size_t)2 * C * sizeof(float) : 0;
    T3D_DISPATCH(dtype,
                 T3D_LAUNCH(im2col_v_kernel<float>, dim3(g), dim3(256), lds, st, (const float*)x, sc, sh, act, (float*)col, B, H, W, C, k, stride, pad, Ho, Wo, Kp, cs, kinv, rpw),
                 T3D_LAUNCH(im2col_v_kernel<bf16_t>, dim3(g), dim3(256), lds, st, (const bf16_t*)x, sc, sh, act, (bf16_t*)col, B, H, W, C, k, stride, pad, Ho, Wo, Kp, cs, kinv, rpw));
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int g = grid_for((long long)B * Ho * Wo * Kp);
  T3D_DISPATCH(dtype,
               T3D_LAUNCH(im2col_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, sc, sh, act, (float*)col, B, H, W, C, k, stride, pad, Ho, Wo, Kp),
               T3D_LAUNCH(im2col_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)x, sc, sh, act, (bf16_t*)col, B, H, W, C, k, stride, pad, Ho, Wo, Kp));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_im2col_nchw(int dtype, const float* x, void* col, int B, int H, int W, int C, int k, int stride, int pad,
                               int Kp, void* stream) {
  if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || Kp < k * k * C) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const long long M = (long long)B * Ho * Wo;
  if (Kp % 8 == 0 && Kp / 8 <= 64 && k <= 48 && (long long)Kp * C < 65536 && fits_u32(M)) {
    int rpw;
    const int g = runs_for(M, 8192, &rpw), cinv = 65536 / C + 1, kinv = 65536 / k + 1;
    T3D_DISPATCH(dtype,
                 T3D_LAUNCH(im2col_nchw_v_kernel<float>, dim3(g), dim3(256), 0, st, x, (float*)col, B, H, W, C, k, stride, pad, Ho, Wo, Kp, cinv, kinv, rpw),
                 T3D_LAUNCH(im2col_nchw_v_kernel<bf16_t>, dim3(g), dim3(256), 0, st, x, (bf16_t*)col, B, H, W, C, k, stride, pad, Ho, Wo, Kp, cinv, kinv, rpw));
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int g = grid_for((long long)B * Ho * Wo * Kp);
  T3D_DISPATCH(dtype,
               T3D_LAUNCH(im2col_nchw_kernel<float>, dim3(g), dim3(256), 0, st, x, (float*)col, B, H, W, C, k, stride, pad, Ho, Wo, Kp),
               T3D_LAUNCH(im2col_nchw_kernel<bf16_t>, dim3(g), dim3(256), 0, st, x, (bf16_t*)col, B, H, W, C, k, stride, pad, Ho, Wo, Kp));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_col2im_bwd(int dtype, const void* dcol, const void* x_raw, const t3d_prologue* pro, void* dx, double* stats,
                              int B, int H, int W, int C, int k, int stride, int pad, int Kp, void* stream) {
  if (!dcol || !x_raw || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || Kp < k * k * C) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const float* sc = pro ? pro->scale : nullptr;
  const float* sh = pro ? pro->shift : nullptr;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  const size_t lds = (size_t)2 * C * sizeof(double);    // (the scalar forms use the first half as floats)
  if (C % 8 == 0 && C / 8 <= 64 && Kp % 8 == 0 && (stride == 1 || stride == 2) && fits_u32((long long)B * H * W)) {
    int ppw;
    const int g = runs_for((long long)B * H * W, 4096, &ppw), nrep = g_t3d_reduce.nrep;
    const long long rs = g_t3d_reduce.stats_stride;
#define T3D_C2I(TT, SS) T3D_LAUNCH((col2im_bwd_v_kernel<TT, SS>), dim3(g), dim3(256), lds, st, (const TT*)dcol, (const TT*)x_raw, sc, sh, act, (TT*)dx, stats, B, H, W, C, k, pad, Ho, Wo, Kp, nrep, rs, ppw)
    if (stride == 1) T3D_DISPATCH(dtype, T3D_C2I(float, 1), T3D_C2I(bf16_t, 1));
    else T3D_DISPATCH(dtype, T3D_C2I(float, 2), T3D_C2I(bf16_t, 2));
#undef T3D_C2I
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int g = grid_for_c((long long)B * H * W * C, C);
  T3D_DISPATCH(dtype,
               T3D_LAUNCH(col2im_bwd_kernel<float>, dim3(g), dim3(256), lds, st, (const float*)dcol, (const float*)x_raw, sc, sh, act, (float*)dx, stats, B, H, W, C, k, stride, pad, Ho, Wo, Kp, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride),
               T3D_LAUNCH(col2im_bwd_kernel<bf16_t>, dim3(g), dim3(256), lds, st, (const bf16_t*)dcol, (const bf16_t*)x_raw, sc, sh, act, (bf16_t*)dx, stats, B, H, W, C, k, stride, pad, Ho, Wo, Kp, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pack_conv_weight(int dtype, const float* w, void* out, int N, int C, int k, int Kp, void* stream) {
  if (!w || !out || N <= 0 || C <= 0 || k <= 0 || Kp < k * k * C) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = cdiv(N * Kp, 256);
  T3D_DISPATCH(dtype, T3D_LAUNCH(pack_conv_kernel<float>, dim3(g), dim3(256), 0, st, w, (float*)out, N, C, k, Kp),
               T3D_LAUNCH(pack_conv_kernel<bf16_t>, dim3(g), dim3(256), 0, st, w, (bf16_t*)out, N, C, k, Kp));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_unpack_conv_grad(const float* dw_packed, float* dw, int N, int C, int k, int Kp, void* stream) {
  if (!dw_packed || !dw || N <= 0 || C <= 0 || k <= 0 || Kp < k * k * C) return T3D_ERR_ARG;
  T3D_LAUNCH(unpack_conv_grad_kernel, dim3(cdiv(N * C * k * k, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     dw_packed, dw, N, C, k, Kp);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_maxpool_fwd(int dtype, const void* y, const t3d_prologue* pro, void* out, unsigned char* argmax, int B, int H,
                               int W, int C, void* stream) {
  if (!y || !out || !argmax || B <= 0 || H <= 0 || W <= 0 || C <= 0) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pro)
    if (const int rc = t3d_fold_fallback(pro->scale, st)) return rc;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const float* sc = pro ? pro->scale : nullptr;
  const float* sh = pro ? pro->shift : nullptr;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  if (C % 8 == 0 && C / 8 <= 64 && fits_u32((long long)B * Ho * Wo)) {
    int ppw;
    const int g = runs_for((long long)B * Ho * Wo, 8192, &ppw);
    T3D_DISPATCH(dtype,
                 T3D_LAUNCH(maxpool_fwd_v_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)y, sc, sh, act, (float*)out, argmax, B, H, W, C, Ho, Wo, ppw),
                 T3D_LAUNCH(maxpool_fwd_v_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)y, sc, sh, act, (bf16_t*)out, argmax, B, H, W, C, Ho, Wo, ppw));
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int g = grid_for((long long)B * Ho * Wo * C);
  T3D_DISPATCH(dtype,
               T3D_LAUNCH(maxpool_fwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)y, sc, sh, act, (float*)out, argmax, B, H, W, C, Ho, Wo),
               T3D_LAUNCH(maxpool_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)y, sc, sh, act, (bf16_t*)out, argmax, B, H, W, C, Ho, Wo));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_maxpool_bwd(int dtype, const void* dout, const unsigned char* argmax, const void* y, const t3d_prologue* pro,
                               void* dy, double* stats, int B, int H, int W, int C, void* stream) {
  if (!dout || !argmax || !y || !dy || B <= 0 || H <= 0 || W <= 0 || C <= 0) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const float* sc = pro ? pro->scale : nullptr;
  const float* sh = pro ? pro->shift : nullptr;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  const size_t lds = (size_t)2 * C * sizeof(double);    // (the scalar form uses the first half as floats)
  if (C % 8 == 0 && C / 8 <= 64 && fits_u32((long long)B * H * W)) {
    int ppw;
    const int g = runs_for((long long)B * H * W, 4096, &ppw), nrep = g_t3d_reduce.nrep;
    const long long rs = g_t3d_reduce.stats_stride;
    T3D_DISPATCH(dtype,
                 T3D_LAUNCH(maxpool_bwd_v_kernel<float>, dim3(g), dim3(256), lds, st, (const float*)dout, argmax, (const float*)y, sc, sh, act, (float*)dy, stats, B, H, W, C, Ho, Wo, nrep, rs, ppw),
                 T3D_LAUNCH(maxpool_bwd_v_kernel<bf16_t>, dim3(g), dim3(256), lds, st, (const bf16_t*)dout, argmax, (const bf16_t*)y, sc, sh, act, (bf16_t*)dy, stats, B, H, W, C, Ho, Wo, nrep, rs, ppw));
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int g = grid_for_c((long long)B * H * W * C, C);
  T3D_DISPATCH(dtype,
               T3D_LAUNCH(maxpool_bwd_kernel<float>, dim3(g), dim3(256), lds, st, (const float*)dout, argmax, (const float*)y, sc, sh, act, (float*)dy, stats, B, H, W, C, Ho, Wo, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride),
               T3D_LAUNCH(maxpool_bwd_kernel<bf16_t>, dim3(g), dim3(256), lds, st, (const bf16_t*)dout, argmax, (const bf16_t*)y, sc, sh, act, (bf16_t*)dy, stats, B, H, W, C, Ho, Wo, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_res_relu_fwd(int dtype, const void* y3, const t3d_prologue* pro3, const void* shortcut,
                                const t3d_prologue* pro_s, void* z, int M, int C, void* stream) {
  if (!y3 || !pro3 || !pro3->scale || !shortcut || !z || M <= 0 || C <= 0) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (const int rc = t3d_fold_fallback(pro3->scale, st)) return rc;
  if (pro_s)
    if (const int rc = t3d_fold_fallback(pro_s->scale, st)) return rc;
  const float* ss = pro_s ? pro_s->scale : nullptr;
  const float* ts = pro_s ? pro_s->shift : nullptr;
  const long long total = (long long)M * C;
  if (C % 8 == 0 && C / 8 <= 256) {
    const int C8 = C / 8;
    int g = (int)std::min<long long>(2048, (total / 8 + 256 * 4 - 1) / (256 * 4));
    while ((long long)g * 256 < C8) ++g;
    T3D_DISPATCH(dtype,
                 T3D_LAUNCH(res_relu_fwd_v_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)y3, pro3->scale, pro3->shift, (const float*)shortcut, ss, ts, (float*)z, total / 8, C8),
                 T3D_LAUNCH(res_relu_fwd_v_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)y3, pro3->scale, pro3->shift, (const bf16_t*)shortcut, ss, ts, (bf16_t*)z, total / 8, C8));
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int g = grid_for(total);
  T3D_DISPATCH(dtype,
               T3D_LAUNCH(res_relu_fwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)y3, pro3->scale, pro3->shift, (const float*)shortcut, ss, ts, (float*)z, total, C),
               T3D_LAUNCH(res_relu_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)y3, pro3->scale, pro3->shift, (const bf16_t*)shortcut, ss, ts, (bf16_t*)z, total, C));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_res_relu_bwd(int dtype, const void* dz, const void* z, const void* y3, const void* yd, void* g, double* stats3,
                                double* statsd, int M, int C, void* stream) {
  if (!dz || !z || !y3 || !g || !stats3 || M <= 0 || C <= 0 || ((yd == nullptr) != (statsd == nullptr))) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long total = (long long)M * C;
  if (C % 8 == 0 && C / 8 <= 256) {
    // at least 8 vectors per thread, at most 1024 workgroups: the double atomics at the end are 2C..3C per workgroup
    int gv = (int)std::min<long long>(1024, (total / 8 + 256 * 8 - 1) / (256 * 8));
    while ((long long)gv * 256 < C / 8) ++gv;
    const size_t ldsv = (size_t)3 * C * sizeof(double);
    const int nrep = g_t3d_reduce.nrep;
    const long long rs = g_t3d_reduce.stats_stride;
#define T3D_RRB(TT, PP) T3D_LAUNCH((res_relu_bwd_v_kernel<TT, PP>), dim3(gv), dim3(256), ldsv, st, (const TT*)dz, (const TT*)z, (const TT*)y3, (const TT*)yd, (TT*)g, stats3, statsd, total / 8, C, nrep, rs)
    if (yd) T3D_DISPATCH(dtype, T3D_RRB(float, true), T3D_RRB(bf16_t, true));
    else T3D_DISPATCH(dtype, T3D_RRB(float, false), T3D_RRB(bf16_t, false));
#undef T3D_RRB
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int gr = grid_for_c(total, C);
  const size_t lds = (size_t)4 * C * sizeof(float);
  T3D_DISPATCH(dtype,
               T3D_LAUNCH(res_relu_bwd_kernel<float>, dim3(gr), dim3(256), lds, st, (const float*)dz, (const float*)z, (const float*)y3, (const float*)yd, (float*)g, stats3, statsd, total, C, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride),
               T3D_LAUNCH(res_relu_bwd_kernel<bf16_t>, dim3(gr), dim3(256), lds, st, (const bf16_t*)dz, (const bf16_t*)z, (const bf16_t*)y3, (const bf16_t*)yd, (bf16_t*)g, stats3, statsd, total, C, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_subsample(int dtype, const void* x, void* out, int B, int H, int W, int C, int stride, int upsample,
                             void* stream) {
  if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || stride <= 0) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long total = upsample ? (long long)B * H * W * C : (long long)B * Ho * Wo * C;
  if (C % 8 == 0 && total / 8 < (1LL << 31)) {
    const unsigned t8 = (unsigned)(total / 8);
    const int g = (int)std::min<long long>(4096, (t8 + 255) / 256);
    T3D_DISPATCH(dtype,
                 T3D_LAUNCH(subsample_v_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, (float*)out, B, H, W, C / 8, stride, Ho, Wo, upsample, t8),
                 T3D_LAUNCH(subsample_v_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, B, H, W, C / 8, stride, Ho, Wo, upsample, t8));
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int g = grid_for(total);
  T3D_DISPATCH(dtype,
               T3D_LAUNCH(subsample_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, (float*)out, B, H, W, C, stride, Ho, Wo, upsample),
               T3D_LAUNCH(subsample_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, B, H, W, C, stride, Ho, Wo, upsample));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
