// Streaming (one-pass, HBM-bound) kernels around the convolutions, NHWC, 8 channels (16 B bf16 /
// 32 B fp32) per lane:
//   * stem patch gather (NCHW fp32 crops -> [pixels, 32] patch rows for the stem GEMM),
//   * block-output materialisation  z = act(scale*y + shift) + residual  and its activation backward,
//   * global average pool forward / backward.
// Per-channel reductions: every thread keeps ONE fixed 8-channel group across its grid-stride
// loop, so the sums stay in registers, meet in LDS once per block and leave as one fp64 atomic
// per channel per block.
#include "common.h"

namespace {

// ------------------------------------------------------------------ stem im2col
// One workgroup per (image, pair of output rows): the five input rows x 3 channels it needs are read ONCE, fully coalesced,
// into an LDS tile with a zero frame (normalised on the way in for uint8 crops); then a thread per (output pixel, 8 of its 32
// patch columns) gathers from LDS and writes ONE 16-B piece -- the four lanes of a pixel write its 64-byte row, a wave 1 KB
// contiguous.  (A thread per pixel gathering its 27 taps from global memory with stride-2 lanes and four 16-B stores 64 B
// apart: 138 us for 154 + 205 MB at B = 256, 2.6 TB/s; the same with the four-lane store but the gathers still global: 158.)
// U8: the crops are raw uint8 NHWC pixels, normalised here as (u/255 - mean[c]) * inv_std[c]; padding taps stay 0
// (the reference pads the NORMALISED image, mobilenetv3.py:110-115 after dataloaders/objectron_main.py:84-96).
template <typename T, bool U8>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ xin, const float* __restrict__ mean,
                                                     const float* __restrict__ istd, T* __restrict__ col, int B,
                                                     int H, int W, int Ho, int Wo) {
  extern __shared__ float tile[];          // [3][5][W + 2]: column ix + 1, row iy - (2 oy0 - 1)
  const int WP = W + 2;
  const int hp = (Ho + 1) / 2;
  const int b = blockIdx.x / hp, oy0 = (blockIdx.x % hp) * 2;
  const int iy0 = oy0 * 2 - 1;
  const int tid = threadIdx.x;
  for (int i = tid; i < 15 * 2; i += 256) {      // the zero frame: columns 0 and W + 1 of every row
    const int rr = i >> 1;
    tile[rr * WP + ((i & 1) ? W + 1 : 0)] = 0.f;
  }
  if constexpr (U8) {
    const unsigned char* xu = reinterpret_cast<const unsigned char*>(xin);
    const float mu[3] = {mean[0], mean[1], mean[2]}, is[3] = {istd[0], istd[1], istd[2]};
    const int RB = W * 3;                       // bytes per input row
    if ((RB & 3) == 0) {                        // four bytes per lane
      const int R4 = RB >> 2;
      for (int i = tid; i < 5 * R4; i += 256) {
        const int r = i / R4, e0 = (i - r * R4) * 4;
        const int iy = iy0 + r;
        const bool ok = iy >= 0 && iy < H;
        const unsigned int u = ok ? *reinterpret_cast<const unsigned int*>(xu + ((size_t)b * H + iy) * RB + e0) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = e0 + k, ix = e / 3, ci = e - ix * 3;
          const float m = ci == 0 ? mu[0] : (ci == 1 ? mu[1] : mu[2]), sc = ci == 0 ? is[0] : (ci == 1 ? is[1] : is[2]);
          tile[(ci * 5 + r) * WP + ix + 1] = ok ? ((float)((u >> (8 * k)) & 0xffu) * (1.0f / 255.0f) - m) * sc : 0.f;
        }
      }
    } else {
      for (int i = tid; i < 5 * RB; i += 256) {
        const int r = i / RB, e = i - r * RB, ix = e / 3, ci = e - ix * 3;
        const int iy = iy0 + r;
        float v = 0.f;
        if (iy >= 0 && iy < H) v = ((float)xu[((size_t)b * H + iy) * RB + e] * (1.0f / 255.0f) - mu[ci]) * is[ci];
        tile[(ci * 5 + r) * WP + ix + 1] = v;
      }
    }
  } else {
    const float* xf = reinterpret_cast<const float*>(xin);
    if ((W & 3) == 0) {
      const int W4 = W >> 2;
      for (int i = tid; i < 15 * W4; i += 256) {
        const int rr = i / W4, x4 = i - rr * W4, ci = rr / 5, r = rr - ci * 5;
        const int iy = iy0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H) v = *reinterpret_cast<const float4*>(xf + (((size_t)b * 3 + ci) * H + iy) * W + x4 * 4);
        float* d = tile + rr * WP + x4 * 4 + 1;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    } else {
      for (int i = tid; i < 15 * W; i += 256) {
        const int rr = i / W, ix = i - rr * W, ci = rr / 5, r = rr - ci * 5;
        const int iy = iy0 + r;
        tile[rr * WP + ix + 1] = (iy >= 0 && iy < H) ? xf[(((size_t)b * 3 + ci) * H + iy) * W + ix] : 0.f;
      }
    }
  }
  __syncthreads();
  const int nrow = min(2, Ho - oy0);
  for (int it = tid; it < nrow * Wo * 4; it += 256) {
    const int q = it & 3, pp = it >> 2, ro = pp / Wo, ox = pp - ro * Wo;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = q * 8 + i;                       // column (ci * 3 + ky) * 3 + kx; 27 .. 31 are padding
      const int ci = idx / 9, r = idx - ci * 9, ky = r / 3, kx = r - ky * 3;
      v[i] = idx < 27 ? tile[(ci * 5 + ro * 2 + ky) * WP + ox * 2 + kx] : 0.f;
    }
    Vec8<T>::store(col + (((size_t)b * Ho + oy0 + ro) * Wo + ox) * 32 + q * 8, v);
  }
}

// ------------------------------------------------------------------ channel-group bookkeeping
struct EwArgs {
  const void *a, *b;   // primary / secondary input
  const void* res;
  void* out;
  const float *scale, *shift;
  const float* vec;    // per-sample fp32 vector (gap bwd: dpooled; se-after apply: gate s)
  const float* vec2;   // se-after apply: pooled-path gradient g
  float* pooled;
  double* stats;
  int act;
  int M, C, HW;
  float inv_hw;
  const T3dFold* fold;        // requested BatchNorm finalize of (scale, shift), derived by the kernel (bn_apply only)
  int mode;            // global pool: T3D_POOL_AVG / _MAX / _AVGMAX
  int* argmax;         // [B,C] position (hw) of the per-sample maximum, written by the forward, read by the backward
};

__device__ __forceinline__ void load_affine(const EwArgs& a, int c0, float sc[8], float sh[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = a.scale ? a.scale[c0 + i] : 1.f;
    sh[i] = a.scale ? a.shift[c0 + i] : 0.f;
  }
}

// block-level merge of per-thread (sum, sumsq-like) pairs for a fixed channel group per thread: fp64 LDS accumulators (the
// threads' fp32 partials add exactly in any order -- bit-reproducible sums), one fp64 atomic per channel and workgroup
__device__ __forceinline__ void flush_stats(float* lraw, int C, int c0, bool on, const float s1[8],
                                            const float s2[8], double* stats) {
  double* lstat = reinterpret_cast<double*>(lraw);
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lstat[i] = 0.0;
  __syncthreads();
  if (on) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      atomicAdd(lstat + c0 + i, (double)s1[i]);
      atomicAdd(lstat + C + c0 + i, (double)s2[i]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(stats + i, lstat[i]);
}

// z = act(scale*y + shift) + res
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const EwArgs a) {
  const int CG = a.C / 8;
  const size_t nvec = (size_t)a.M * CG;
  const T* __restrict__ y = reinterpret_cast<const T*>(a.a);
  const T* __restrict__ r = reinterpret_cast<const T*>(a.res);
  T* __restrict__ z = reinterpret_cast<T*>(a.out);
  // stride is a multiple of CG -> each thread keeps one channel group
  const size_t nthr = ((size_t)gridDim.x * 256 / CG) * CG;
  const size_t g = blockIdx.x * (size_t)256 + threadIdx.x;
  extern __shared__ float fco[];     // [2][C], only with a requested finalize (common.h): derived here, block 0 publishes
  if (a.fold) t3d_fold_block(a.fold, 0, a.C, fco, a.C, blockIdx.x == 0);
  if (g >= nthr) return;
  const int c0 = (int)(g % CG) * 8;
  float sc[8], sh[8];
  if (a.fold) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = fco[c0 + i]; sh[i] = fco[a.C + c0 + i]; }
  } else {
    load_affine(a, c0, sc, sh);
  }
  for (size_t i = g; i < nvec; i += nthr) {
    float v[8];
    Vec8<T>::load(y + i * 8, v);
    act_affine_vec<8>(v, sc, sh, a.act);
    if (r) {
      float rr[8];
      Vec8<T>::load(r + i * 8, rr);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += rr[j];
    }
    Vec8<T>::store(z + i * 8, v);
  }
}

// dzp = dz * act'(scale*y + shift), stats += sum(dzp), sum(dzp*y)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const EwArgs a) {
  extern __shared__ float lstat[];
  const int CG = a.C / 8;
  const size_t nvec = (size_t)a.M * CG;
  const T* __restrict__ dz = reinterpret_cast<const T*>(a.a);
  const T* __restrict__ y = reinterpret_cast<const T*>(a.b);
  T* __restrict__ o = reinterpret_cast<T*>(a.out);
  const size_t nthr = ((size_t)gridDim.x * 256 / CG) * CG;
  const size_t g = blockIdx.x * (size_t)256 + threadIdx.x;
  const bool on = g < nthr;
  const int c0 = (int)(g % CG) * 8;
  float sc[8], sh[8], s1[8], s2[8];
  load_affine(a, c0, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (on) {
    for (size_t i = g; i < nvec; i += nthr) {
      float d[8], yv[8];
      Vec8<T>::load(dz + i * 8, d);
      Vec8<T>::load(y + i * 8, yv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = Vec8<T>::round(d[j] * act_grad(yv[j] * sc[j] + sh[j], a.act));
        d[j] = v;
        s1[j] += v;
        s2[j] = fmaf(v, yv[j], s2[j]);
      }
      Vec8<T>::store(o + i * 8, d);
    }
  }
  if (a.stats) flush_stats(lstat, a.C, c0, on, s1, s2, a.stats);
}

// Squeeze-excite AFTER the activation (no-expand layout, mobilenetv3.py:138-140; MobileNetV3-small features.1):
// v = s * a, a = act(u), u = scale*y + shift.  Backward, step 2 of 2:
//   du = (s[b] * dv + g[b]) * act'(u),  stats += sum(du), sum(du*y)       (dv: gradient at the gated tensor)
template <typename T>
__global__ __launch_bounds__(256) void se_after_apply_kernel(const EwArgs a) {
  extern __shared__ float lstat[];
  const int CG = a.C / 8;
  const size_t nvec = (size_t)a.M * CG;
  const T* __restrict__ dv = reinterpret_cast<const T*>(a.a);
  const T* __restrict__ y = reinterpret_cast<const T*>(a.b);
  T* __restrict__ o = reinterpret_cast<T*>(a.out);
  const size_t nthr = ((size_t)gridDim.x * 256 / CG) * CG;
  const size_t g = blockIdx.x * (size_t)256 + threadIdx.x;
  const bool on = g < nthr;
  const int c0 = (int)(g % CG) * 8;
  float sc[8], sh[8], s1[8], s2[8];
  load_affine(a, c0, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (on) {
    for (size_t i = g; i < nvec; i += nthr) {
      float d[8], yv[8];
      Vec8<T>::load(dv + i * 8, d);
      Vec8<T>::load(y + i * 8, yv);
      const size_t bo = (i / CG / a.HW) * a.C + c0;     // sample of this pixel row
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float da = fmaf(a.vec[bo + j], d[j], a.vec2[bo + j]);
        const float v = Vec8<T>::round(da * act_grad(yv[j] * sc[j] + sh[j], a.act));
        d[j] = v;
        s1[j] += v;
        s2[j] = fmaf(v, yv[j], s2[j]);
      }
      Vec8<T>::store(o + i * 8, d);
    }
  }
  if (a.stats) flush_stats(lstat, a.C, c0, on, s1, s2, a.stats);
}

// 256 threads = cgs channel groups (8 channels each) x 256 / cgs pixel slots: 32 x 8 for layers of >= 256 channels; narrower layers
// give the idle group lanes to more pixel slots (round 6: MobileNetV3-small's 16-channel gated block at 56x56 ran two groups x 8 slots
// -- 16 live threads per workgroup walking 392 pixels each: 135 us for 26 MB).  The slot sums meet in slot order: deterministic;
// identical to the fixed 32 x 8 split where that applied.
__device__ __forceinline__ int pool_cgs(int CG) {
  int c = 1;
  while (c < CG && c < 32) c <<= 1;
  return c;
}

// step 1 of 2: ps[b][c][0] = sum_hw dv * act(scale*y + shift)  (= d loss / d gate), ps[b][c][1] = 0
// grid (B, ceil(CG/32)), block 256 = 32 groups x 8 hw slots (as gap_fwd_kernel)
template <typename T>
__global__ __launch_bounds__(256) void se_after_sums_kernel(const EwArgs a) {
  __shared__ float red[8 * 32 * 8];
  const int CG = a.C / 8, b = blockIdx.x;
  const int cgs = pool_cgs(CG), nslots = 256 / cgs, rowl = cgs * 8;
  const int cgl = threadIdx.x & (cgs - 1), slot = threadIdx.x / cgs;
  const int cg = blockIdx.y * cgs + cgl;
  const bool on = cg < CG;
  const int c0 = on ? cg * 8 : 0;
  const T* __restrict__ dv = reinterpret_cast<const T*>(a.a);
  const T* __restrict__ y = reinterpret_cast<const T*>(a.b);
  float sc[8], sh[8], acc[8];
  load_affine(a, c0, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (on) {
    for (int hw = slot; hw < a.HW; hw += nslots) {
      float v[8], d[8];
      const size_t off = ((size_t)b * a.HW + hw) * a.C + c0;
      Vec8<T>::load(y + off, v);
      Vec8<T>::load(dv + off, d);
      act_affine_vec<8>(v, sc, sh, a.act);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(d[j], v[j], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[slot * rowl + cgl * 8 + j] = acc[j];
  __syncthreads();
  const int t = threadIdx.x;
  const int c = blockIdx.y * rowl + t;
  if (t < rowl && c < a.C) {
    float s = 0.f;
    for (int q = 0; q < nslots; ++q) s += red[q * rowl + t];
    a.pooled[((size_t)b * a.C + c) * 2] = s;
    a.pooled[((size_t)b * a.C + c) * 2 + 1] = 0.f;
  }
}

// pooled[b][c] = mean_hw a (avg) | max_hw a (max) | both added (avg+max), a = act(scale*y + shift);
// grid (B, ceil(CG/32)), block 256 = 32 groups x 8 hw slots.  The maximum keeps the FIRST position in (h,w) scan order
// among equal values (what F.adaptive_max_pool2d's backward routes the gradient to).
template <typename T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const EwArgs a) {
  __shared__ float red[8 * 32 * 8];
  __shared__ float redm[8 * 32 * 8];
  __shared__ int redi[8 * 32 * 8];
  const int CG = a.C / 8, b = blockIdx.x;
  const int cgs = pool_cgs(CG), nslots = 256 / cgs, rowl = cgs * 8;
  const int cgl = threadIdx.x & (cgs - 1), slot = threadIdx.x / cgs;
  const int cg = blockIdx.y * cgs + cgl;
  const bool on = cg < CG;
  const int c0 = on ? cg * 8 : 0;
  const bool want_max = a.mode != T3D_POOL_AVG;
  const T* __restrict__ y = reinterpret_cast<const T*>(a.a);
  float sc[8], sh[8], acc[8], mx[8];
  int mi[8];
  load_affine(a, c0, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[j] = 0.f; mx[j] = -INFINITY; mi[j] = 0; }
  if (on) {
    for (int hw = slot; hw < a.HW; hw += nslots) {
      float v[8];
      Vec8<T>::load(y + ((size_t)b * a.HW + hw) * a.C + c0, v);
      act_affine_vec<8>(v, sc, sh, a.act);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[j] += v[j];
        if (v[j] > mx[j]) { mx[j] = v[j]; mi[j] = hw; }      // strict: first position wins inside a slot
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[slot * rowl + cgl * 8 + j] = acc[j];
    if (want_max) { redm[slot * rowl + cgl * 8 + j] = mx[j]; redi[slot * rowl + cgl * 8 + j] = mi[j]; }
  }
  __syncthreads();
  const int t = threadIdx.x;  // the first cgs x 8 threads: one channel each
  const int c = blockIdx.y * rowl + t;
  if (t < rowl && c < a.C) {
    float s = 0.f;
    for (int q = 0; q < nslots; ++q) s += red[q * rowl + t];
    float out = s * a.inv_hw;
    if (want_max) {
      float m = redm[t];
      int im = redi[t];
      for (int q = 1; q < nslots; ++q) {
        const float mq = redm[q * rowl + t];
        const int iq = redi[q * rowl + t];
        if (mq > m || (mq == m && iq < im)) { m = mq; im = iq; }
      }
      if (a.argmax) a.argmax[(size_t)b * a.C + c] = im;
      out = (a.mode == T3D_POOL_MAX) ? m : out + m;
    }
    a.pooled[(size_t)b * a.C + c] = out;
  }
}

// dz[b][hw][c] = dpooled[b][c]/HW * act'(scale*y+shift); stats.  grid (nb, ceil(CG/32)): block loops over samples
template <typename T>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const EwArgs a, int B, const int g_nrep, const long long g_rstride) {
  __shared__ float red[2][8 * 32 * 8];
  const int CG = a.C / 8;
  const int cgs = pool_cgs(CG), nslots = 256 / cgs, rowl = cgs * 8;
  const int cgl = threadIdx.x & (cgs - 1), slot = threadIdx.x / cgs;
  const int cg = blockIdx.y * cgs + cgl;
  const bool on = cg < CG;
  const int c0 = on ? cg * 8 : 0;
  const T* __restrict__ y = reinterpret_cast<const T*>(a.a);
  T* __restrict__ dz = reinterpret_cast<T*>(a.out);
  float sc[8], sh[8], s1[8], s2[8];
  load_affine(a, c0, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (on) {
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
      float dp[8], dm[8];
      int am[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = a.vec[(size_t)b * a.C + c0 + j];
        dp[j] = (a.mode == T3D_POOL_MAX) ? 0.f : g * a.inv_hw;       // the mean's share, every position
        dm[j] = (a.mode == T3D_POOL_AVG) ? 0.f : g;                  // the maximum's share, its position only
        am[j] = (a.mode == T3D_POOL_AVG) ? -1 : a.argmax[(size_t)b * a.C + c0 + j];
      }
      for (int hw = slot; hw < a.HW; hw += nslots) {
        const size_t off = ((size_t)b * a.HW + hw) * a.C + c0;
        float v[8], d[8];
        Vec8<T>::load(y + off, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = dp[j] + (hw == am[j] ? dm[j] : 0.f);
        act_grad_affine_vec<8>(d, v, sc, sh, a.act);          // one activation switch per 8 elements
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          d[j] = Vec8<T>::round(d[j]);
          s1[j] += d[j];
          s2[j] = fmaf(d[j], v[j], s2[j]);
        }
        Vec8<T>::store(dz + off, d);
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[0][slot * rowl + cgl * 8 + j] = s1[j];
      red[1][slot * rowl + cgl * 8 + j] = s2[j];
    }
    __syncthreads();
    const int t = threadIdx.x, c = blockIdx.y * rowl + t;
    if (t < rowl && c < a.C) {
      double u1 = 0.0, u2 = 0.0;
      for (int q = 0; q < nslots; ++q) {
        u1 += (double)red[0][q * rowl + t];
        u2 += (double)red[1][q * rowl + t];
      }
      // one add per (block, channel), 256 sample blocks per channel: spread over the reduction replicas
      double* st = a.stats + (size_t)(blockIdx.x % g_nrep) * g_rstride;
      atomicAdd(st + c, u1);
      atomicAdd(st + a.C + c, u2);
    }
  }
}

inline void fill_pro(EwArgs& a, const t3d_prologue* pro) {
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
}
inline int ew_grid(size_t nvec) {
  size_t g = (nvec + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

static int im2col_launch(int dtype, const void* x, bool u8, const float* mean, const float* istd, void* col, int B,
                         int H, int W, void* stream) {
  if (!x || !col || B <= 0 || H <= 0 || W <= 0 || (u8 && (!mean || !istd))) return T3D_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int grid = B * ((Ho + 1) / 2);
  const size_t lds = (size_t)15 * (W + 2) * sizeof(float);
  if (lds > 64 * 1024) return T3D_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define T3D_IM2COL(T, U) \
  T3D_LAUNCH((im2col_kernel<T, U>), dim3(grid), dim3(256), lds, st, x, mean, istd, (T*)col, B, H, W, Ho, Wo)
  if (dtype == T3D_F32) {
    if (u8) T3D_IM2COL(float, true); else T3D_IM2COL(float, false);
  } else if (dtype == T3D_BF16) {
    if (u8) T3D_IM2COL(bf16_t, true); else T3D_IM2COL(bf16_t, false);
  } else if (dtype == T3D_F16) {
    if (u8) T3D_IM2COL(f16_t, true); else T3D_IM2COL(f16_t, false);
  } else {
    return T3D_ERR_ARG;
  }
#undef T3D_IM2COL
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_stem_im2col(int dtype, const float* x, void* col, int B, int H, int W, void* stream) {
  return im2col_launch(dtype, x, false, nullptr, nullptr, col, B, H, W, stream);
}

extern "C" int t3d_stem_im2col_u8(int dtype, const unsigned char* x, const float* mean, const float* inv_std, void* col,
                                  int B, int H, int W, void* stream) {
  return im2col_launch(dtype, x, true, mean, inv_std, col, B, H, W, stream);
}

extern "C" int t3d_bn_apply(int dtype, const void* y, const t3d_prologue* pro, const void* residual, void* z,
                            int M, int C, void* stream) {
  if (!y || !z || M <= 0 || C <= 0 || (C % 8)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  EwArgs a{};
  a.a = y; a.res = residual; a.out = z; a.M = M; a.C = C;
  fill_pro(a, pro);
  const int grid = ew_grid((size_t)M * (C / 8));
  if ((size_t)grid * 256 < (size_t)(C / 8)) return T3D_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  a.fold = t3d_take_fold(a.scale);      // a requested finalize of this BatchNorm is derived inside the kernel
  const size_t lds = a.fold ? (size_t)2 * C * sizeof(float) : 0;
  if (dtype == T3D_F32) T3D_LAUNCH(bn_apply_kernel<float>, dim3(grid), dim3(256), lds, st, a);
  else if (dtype == T3D_BF16) T3D_LAUNCH(bn_apply_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, a);
  else if (dtype == T3D_F16) T3D_LAUNCH(bn_apply_kernel<f16_t>, dim3(grid), dim3(256), lds, st, a);
  else return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_bn_act_bwd(int dtype, const void* dz, const void* y, const t3d_prologue* pro, void* dzp,
                              double* stats, int M, int C, void* stream) {
  if (!dz || !y || !dzp || M <= 0 || C <= 0 || (C % 8)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  EwArgs a{};
  a.a = dz; a.b = y; a.out = dzp; a.stats = stats; a.M = M; a.C = C;
  fill_pro(a, pro);
  int grid = ew_grid((size_t)M * (C / 8));
  if (grid > 1024) grid = 1024;
  if ((size_t)grid * 256 < (size_t)(C / 8)) return T3D_ERR_UNSUPPORTED;
  const size_t lds = (size_t)2 * C * sizeof(double);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == T3D_F32) T3D_LAUNCH(bn_act_bwd_kernel<float>, dim3(grid), dim3(256), lds, st, a);
  else if (dtype == T3D_BF16) T3D_LAUNCH(bn_act_bwd_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, a);
  else return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_se_after_sums(int dtype, const void* dv, const void* y, const t3d_prologue* pro, float* ps, int B, int HW,
                                 int C, void* stream) {
  if (!dv || !y || !ps || B <= 0 || HW <= 0 || C <= 0 || (C % 8)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  EwArgs a{};
  a.a = dv; a.b = y; a.pooled = ps; a.C = C; a.HW = HW;
  fill_pro(a, pro);
  dim3 grid(B, cdiv(C / 8, 32));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (const int rc = t3d_fold_fallback(a.scale, st)) return rc;      // no derive prologue here: finalize as its own launch
  if (dtype == T3D_F32) T3D_LAUNCH(se_after_sums_kernel<float>, grid, dim3(256), 0, st, a);
  else if (dtype == T3D_BF16) T3D_LAUNCH(se_after_sums_kernel<bf16_t>, grid, dim3(256), 0, st, a);
  else return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_se_after_apply(int dtype, const void* dv, const void* y, const t3d_prologue* pro, const float* s,
                                  const float* g, void* du, double* stats, int B, int HW, int C, void* stream) {
  if (!dv || !y || !s || !g || !du || B <= 0 || HW <= 0 || C <= 0 || (C % 8)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  EwArgs a{};
  a.a = dv; a.b = y; a.out = du; a.stats = stats; a.vec = s; a.vec2 = g; a.M = B * HW; a.C = C; a.HW = HW;
  fill_pro(a, pro);
  int grid = ew_grid((size_t)a.M * (C / 8));
  if (grid > 1024) grid = 1024;
  if ((size_t)grid * 256 < (size_t)(C / 8)) return T3D_ERR_UNSUPPORTED;
  const size_t lds = (size_t)2 * C * sizeof(double);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == T3D_F32) T3D_LAUNCH(se_after_apply_kernel<float>, dim3(grid), dim3(256), lds, st, a);
  else if (dtype == T3D_BF16) T3D_LAUNCH(se_after_apply_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, a);
  else return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pool_fwd(int dtype, const void* y, const t3d_prologue* pro, int mode, float* pooled, int* argmax,
                            int B, int HW, int C, void* stream) {
  if (!y || !pooled || B <= 0 || HW <= 0 || C <= 0 || (C % 8)) return T3D_ERR_ARG;
  if (mode != T3D_POOL_AVG && mode != T3D_POOL_MAX && mode != T3D_POOL_AVGMAX) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  EwArgs a{};
  a.a = y; a.pooled = pooled; a.C = C; a.HW = HW; a.inv_hw = 1.f / (float)HW; a.mode = mode; a.argmax = argmax;
  fill_pro(a, pro);
  dim3 grid(B, cdiv(C / 8, 32));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (const int rc = t3d_fold_fallback(a.scale, st)) return rc;      // no derive prologue here: finalize as its own launch
  if (dtype == T3D_F32) T3D_LAUNCH(gap_fwd_kernel<float>, grid, dim3(256), 0, st, a);
  else if (dtype == T3D_BF16) T3D_LAUNCH(gap_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, a);
  else if (dtype == T3D_F16) T3D_LAUNCH(gap_fwd_kernel<f16_t>, grid, dim3(256), 0, st, a);
  else return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_gap_fwd(int dtype, const void* y, const t3d_prologue* pro, float* pooled, int B, int HW, int C,
                           void* stream) {
  return t3d_pool_fwd(dtype, y, pro, T3D_POOL_AVG, pooled, nullptr, B, HW, C, stream);
}

extern "C" int t3d_pool_bwd(int dtype, const float* dpooled, const void* y, const t3d_prologue* pro, int mode,
                            const int* argmax, void* dz, double* stats, int B, int HW, int C, void* stream) {
  if (!dpooled || !y || !dz || B <= 0 || HW <= 0 || C <= 0 || (C % 8)) return T3D_ERR_ARG;
  if (mode != T3D_POOL_AVG && mode != T3D_POOL_MAX && mode != T3D_POOL_AVGMAX) return T3D_ERR_ARG;
  if (mode != T3D_POOL_AVG && !argmax) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  EwArgs a{};
  a.a = y; a.vec = dpooled; a.out = dz; a.stats = stats; a.C = C; a.HW = HW; a.inv_hw = 1.f / (float)HW;
  a.mode = mode; a.argmax = const_cast<int*>(argmax);
  fill_pro(a, pro);
  dim3 grid(B < 256 ? B : 256, cdiv(C / 8, 32));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == T3D_F32) T3D_LAUNCH(gap_bwd_kernel<float>, grid, dim3(256), 0, st, a, B, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride);
  else if (dtype == T3D_BF16) T3D_LAUNCH(gap_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, a, B, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride);
  else return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_gap_bwd(int dtype, const float* dpooled, const void* y, const t3d_prologue* pro, void* dz,
                           double* stats, int B, int HW, int C, void* stream) {
  return t3d_pool_bwd(dtype, dpooled, y, pro, T3D_POOL_AVG, nullptr, dz, stats, B, HW, C, stream);
}
