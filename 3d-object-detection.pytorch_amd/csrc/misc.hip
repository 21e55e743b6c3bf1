// Small utility kernels: weight packing (fp32 master -> storage dtype, optional transpose).
#include "common.h"

namespace {

template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, T* __restrict__ out, int rows, int cols, int transpose) {
  const size_t n = (size_t)rows * cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (!transpose) {
      out[i] = (T)w[i];
    } else {  // out [cols][rows]
      const size_t r = i / cols, c = i % cols;
      out[c * rows + r] = (T)w[i];
    }
  }
}

// Fragment order of a [rows][cols] matrix (include/t3d.h: t3d_pwconv_pack_frag): 16-B piece index of elements
// (r, 8 c8 .. 8 c8 + 7).  Rows are grouped in pairs of 16-row tiles with the streaming kernel's permutation (MFMA row lc of tile
// T  <->  row (T >> 1) * 32 + (lc >> 2) * 8 + (T & 1) * 4 + (lc & 3)), so that a lane ends up with 8 consecutive output
// channels of a 32-channel block.
__device__ __forceinline__ size_t t3d_frag_piece(int r, int c8, int KS) {
  const int rr = r & 31, T = (r >> 5) * 2 + ((rr >> 2) & 1), lc = (rr >> 3) * 4 + (rr & 3);
  return ((size_t)T * KS + (c8 >> 2)) * 64 + (c8 & 3) * 16 + lc;
}

// one launch for every weight matrix of the model: desc[t] = {src, out (or 0), out_t (or 0), rows, cols, frag (or 0), frag_t (or 0)}
// as int64.  frag / frag_t: fragment-order copies of the matrix / its transpose for the bf16 / fp16 pointwise kernels (the
// zero padding is the caller's, the buffers are cleared once).  A thread converts 8 consecutive elements of a row (one 16-B
// piece of `out` and of `frag`), then 8 consecutive elements of a COLUMN (one piece of `out_t` and `frag_t`: strided 4-byte
// reads of the fp32 master, L2 hits) -- every write is 16 B; the element-per-thread form wrote three of the four copies as
// scattered 2-byte stores.
template <typename T>
__global__ void pack_batched_kernel(const long long* __restrict__ desc) {
  const long long* d = desc + (size_t)blockIdx.x * 7;
  const float* __restrict__ w = reinterpret_cast<const float*>(d[0]);
  T* __restrict__ out = reinterpret_cast<T*>(d[1]);
  T* __restrict__ out_t = reinterpret_cast<T*>(d[2]);
  const int rows = (int)d[3], cols = (int)d[4];
  T* __restrict__ frag = reinterpret_cast<T*>(d[5]);
  T* __restrict__ frag_t = reinterpret_cast<T*>(d[6]);
  const int KS = (cols + 31) / 32, KSt = (rows + 31) / 32;
  const size_t n = (size_t)rows * cols;
  if constexpr (sizeof(T) == 2) {
    if (!(rows & 7) && !(cols & 7)) {
      typedef T V8 __attribute__((ext_vector_type(8)));
      const int c8n = cols >> 3, r8n = rows >> 3;
      const size_t n1 = (size_t)rows * c8n, n2 = (out_t || frag_t) ? (size_t)cols * r8n : 0;
      for (size_t i = blockIdx.y * (size_t)blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.y * blockDim.x) {
        V8 v;
        if (i < n1) {
          const int r = (int)(i / c8n), c8 = (int)(i - (size_t)r * c8n);
          const float4 a = *reinterpret_cast<const float4*>(w + (size_t)r * cols + c8 * 8);
          const float4 b = *reinterpret_cast<const float4*>(w + (size_t)r * cols + c8 * 8 + 4);
          v[0] = (T)a.x; v[1] = (T)a.y; v[2] = (T)a.z; v[3] = (T)a.w; v[4] = (T)b.x; v[5] = (T)b.y; v[6] = (T)b.z; v[7] = (T)b.w;
          if (out) *reinterpret_cast<V8*>(out + (size_t)r * cols + c8 * 8) = v;
          if (frag) *reinterpret_cast<V8*>(frag + t3d_frag_piece(r, c8, KS) * 8) = v;
        } else {
          const size_t k = i - n1;
          const int c = (int)(k / r8n), r8 = (int)(k - (size_t)c * r8n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (T)w[(size_t)(r8 * 8 + e) * cols + c];
          if (out_t) *reinterpret_cast<V8*>(out_t + (size_t)c * rows + r8 * 8) = v;
          if (frag_t) *reinterpret_cast<V8*>(frag_t + t3d_frag_piece(c, r8, KSt) * 8) = v;
        }
      }
      return;
    }
  }
  for (size_t i = blockIdx.y * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.y * blockDim.x) {
    const T v = (T)w[i];
    const int r = (int)(i / cols), c = (int)(i % cols);
    if (out) out[i] = v;
    if (out_t) out_t[(size_t)c * rows + r] = v;
    if constexpr (sizeof(T) == 2) {
      if (frag) frag[t3d_frag_piece(r, c >> 3, KS) * 8 + (c & 7)] = v;
      if (frag_t) frag_t[t3d_frag_piece(c, r >> 3, KSt) * 8 + (r & 7)] = v;
    }
  }
}

// desc[t] = {src [nrep][count] fp32, dst [count] fp32, count}: dst = sum over the replicas (overwrites)
__global__ void sum_replicas_batched_kernel(const long long* __restrict__ desc, int nrep) {
  const long long* d = desc + (size_t)blockIdx.x * 3;
  const float* __restrict__ src = reinterpret_cast<const float*>(d[0]);
  float* __restrict__ dst = reinterpret_cast<float*>(d[1]);
  const int n = (int)d[2];
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < nrep; ++r) s += src[(size_t)r * n + i];
    dst[i] = s;
  }
}

// desc[t] = {src [slots][count] fp32, dst [count] fp32, count, used (device int*)}: dst = sum over the first *used slots, in
// index order (overwrites) -- the deterministic end of the depthwise weight gradient (t3d_set_dw_slots)
__global__ __launch_bounds__(256) void sum_slots_batched_kernel(const long long* __restrict__ desc) {
  // 64 elements x 4 slot groups per workgroup: a wave's lanes run along the elements (256 contiguous bytes per slot row),
  // group g adds slots g, g + 4, ... in that order with eight loads in flight, the four group sums meet in LDS in index order.
  // (Round 3's 16 x 16 shape read 64-byte pieces: 65 us at the tail of the backward for ~100 MB of slots; a thread per
  // element over up to 512 slots was a 128-round latency chain before that.)
  __shared__ float part[4][64];
  const long long* d = desc + (size_t)blockIdx.x * 4;
  const float* __restrict__ src = reinterpret_cast<const float*>(d[0]);
  float* __restrict__ dst = reinterpret_cast<float*>(d[1]);
  const int n = (int)d[2], nslots = *reinterpret_cast<const int*>(d[3]);
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int i0 = blockIdx.y * 64; i0 < n; i0 += gridDim.y * 64) {
    const int i = min(i0 + el, n - 1);          // (clamped: branch-free loads, lanes past n are not written)
    const float* col = src + i;
    float s = 0.f;
    int r = g;
    for (; r + 28 < nslots; r += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(r + 4 * u) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nslots; r += 4) s += col[(size_t)r * n];
    part[g][el] = s;
    __syncthreads();
    if (g == 0 && i0 + el < n) dst[i0 + el] = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
    __syncthreads();
  }
}

// AdamW over one flat fp32 buffer (torch.optim.AdamW semantics, decoupled weight decay, no amsgrad):
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// 16 B per lane per array; the bias corrections are computed on the host in fp64.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n4,
                                                    float decay, float omb1, float b2, float omb2, float step_size,
                                                    float inv_sqrt_bc2, float eps, float gscale, long long* watch,
                                                    long long step) {
  bool bad = false;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg4 = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pe = reinterpret_cast<float*>(&pp);
    const float* ge = reinterpret_cast<const float*>(&gg4);
    float* me = reinterpret_cast<float*>(&mm);
    float* ve = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = ge[j] * gscale;
      bad |= !(fabsf(gj) <= 3.4028235e38f);                         // inf or NaN
      const float pj = pe[j] * decay;
      const float mj = me[j] + (gj - me[j]) * omb1;                 // lerp(m, g, 1 - beta1)
      const float vj = ve[j] * b2 + omb2 * gj * gj;                 // 1 - beta taken in fp64 on the host, like PyTorch
      const float denom = sqrtf(vj) * inv_sqrt_bc2 + eps;
      pe[j] = pj - step_size * (mj / denom);
      me[j] = mj;
      ve[j] = vj;
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // t3d_set_grad_watch: the first optimizer step that met a non-finite gradient (one atomic per wave that saw one)
  if (watch && __any(bad) && (threadIdx.x & 63) == 0) atomicMin(reinterpret_cast<unsigned long long*>(watch), (unsigned long long)step);
}

// dst [rows][cd] <- src [rows][cs]: the leading min(cs, cd) columns are copied, further dst columns zeroed
__global__ void copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cs, int cd) {
  const int n = rows * cd;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int r = i / cd, c = i % cd;
    dst[i] = c < cs ? src[(size_t)r * cs + c] : 0.f;
  }
}

// Linear bias gradient under a train-mode BatchNorm: sum_b dy = alpha*sum(dz) + beta*sum(y) + count*gamma
__global__ void bn_bias_grad_kernel(const double* __restrict__ fstats, const double* __restrict__ bstats, int nrep,
                                    long long rstride, int C, double count, const float* __restrict__ alpha,
                                    const float* __restrict__ beta, const float* __restrict__ gammac, float* dbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sy = 0.0, sdz = 0.0;
  for (int r = 0; r < nrep; ++r) {
    sy += fstats[r * rstride + c];
    sdz += bstats[r * rstride + c];
  }
  dbias[c] = (float)((double)alpha[c] * sdz + (double)beta[c] * sy + count * (double)gammac[c]);
}

// squeeze-excite gate between a BatchNorm and its consumer: per-sample backward affine
//   aps[b][c] = s[b][c]*alpha[c],  gps[b][c] = gammac[c] + g[b][c]*alpha[c]
__global__ void se_bwd_affine_kernel(const float* __restrict__ s, const float* __restrict__ g,
                                     const float* __restrict__ alpha, const float* __restrict__ gammac,
                                     float* __restrict__ aps, float* __restrict__ gps, int B, int C) {
  const int n = B * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i % C;
    aps[i] = s[i] * alpha[c];
    gps[i] = gammac[c] + g[i] * alpha[c];
  }
}

// ... with the BatchNorm-backward finalize of (alpha, beta, gamma) derived HERE from the replica sums (common.h: a pending request
// for `alpha`; round 6): a workgroup owns 32 channels of every sample, derives and PUBLISHES them (every channel has exactly one
// owner), then forms its [B][32] slice -- the standalone 5-us finalize launch + its dispatch gap ahead of this 3-us kernel go
// (8 gated blocks per MobileNetV3-large step)
__global__ __launch_bounds__(256) void se_bwd_affine_fold_kernel(const float* __restrict__ s, const float* __restrict__ g,
                                                                 const T3dFold* __restrict__ fold, float* __restrict__ aps,
                                                                 float* __restrict__ gps, int B, int C) {
  __shared__ float co[3][32];
  const int c0 = blockIdx.x * 32, cb = min(32, C - c0);
  // blockIdx.y: a chunk of samples (the derive is repeated per chunk -- 32 channels x the replicas, a few hundred bytes; chunk 0
  // publishes).  One workgroup per channel block walking all samples was a 17-us latency chain on 30 CUs.
  t3d_fold_block(fold, c0, cb, &co[0][0], 32, blockIdx.y == 0);      // (ends with a barrier)
  const int per = (B + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  for (int i = b0 * 32 + threadIdx.x; i < b1 * 32; i += 256) {
    const int b = i >> 5, cl = i & 31;
    if (cl < cb) {
      const size_t o = (size_t)b * C + c0 + cl;
      aps[o] = s[o] * co[0][cl];
      gps[o] = co[2][cl] + g[o] * co[0][cl];
    }
  }
}

// Dropout(0.5) keep/scale factors {0, 2} from a counter-based generator (Philox-4x32-10 keyed by `seed`, counter =
// (element index / 4, offset)): one launch, no state on the device, reproducible for a given (seed, offset).
__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__global__ void dropout_mask_kernel(float* __restrict__ mask, long long n, unsigned long long seed,
                                    unsigned long long offset, float keep, float scale) {
  const long long n4 = (n + 3) / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = i * 4 + j;
      if (e < n) mask[e] = ((float)(c[j] >> 8) * (1.f / 16777216.f) < keep) ? scale : 0.f;
    }
  }
}

// desc[t] = {ptr, bytes (multiple of 16)}: zero fill of several buffers in one launch
__global__ void zero_batched_kernel(const long long* __restrict__ desc) {
  const long long* d = desc + (size_t)blockIdx.x * 2;
  float4* __restrict__ dst = reinterpret_cast<float4*>(d[0]);
  const long long n = d[1] / 16;
  for (long long i = blockIdx.y * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.y * blockDim.x)
    dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace

static long long* g_grad_watch = nullptr;
extern "C" int t3d_set_grad_watch(long long* first_bad_step) {
  g_grad_watch = first_bad_step;
  return T3D_OK;
}

extern "C" int t3d_adamw_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1,
                              double beta2, double eps, double weight_decay, long long step, double grad_scale,
                              void* stream) {
  if (!p || !g || !m || !v || n <= 0 || (n % 4) || step <= 0) return T3D_ERR_ARG;
  // hyper-parameters in fp64 like the Python floats PyTorch holds: 1 - beta and beta^step are formed in fp64 and only
  // then rounded to fp32 (1 - 0.999 -> 1.0000000e-3f, where 1.f - 0.999f would be 0.99998713e-3f)
  const double b1d = beta1, b2d = beta2;
  const double bc1 = 1.0 - pow(b1d, (double)step), bc2 = 1.0 - pow(b2d, (double)step);
  const long long n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  T3D_LAUNCH(adamw_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n4,
                     (float)(1.0 - lr * weight_decay), (float)(1.0 - b1d), (float)beta2, (float)(1.0 - b2d),
                     (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)eps, (float)grad_scale, g_grad_watch, step);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_copy_cols(const float* src, float* dst, int rows, int cols_src, int cols_dst, void* stream) {
  if (!src || !dst || rows <= 0 || cols_src <= 0 || cols_dst <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(copy_cols_kernel, dim3(cdiv(rows * cols_dst, 256) < 256 ? cdiv(rows * cols_dst, 256) : 256),
                     dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, rows, cols_src, cols_dst);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_bn_bias_grad(const double* fwd_stats, const double* bwd_stats, int C, double count,
                                const float* alpha, const float* beta, const float* gammac, float* dbias,
                                void* stream) {
  if (!fwd_stats || !bwd_stats || !alpha || !beta || !gammac || !dbias || C <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(bn_bias_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     fwd_stats, bwd_stats, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride, C, count, alpha, beta, gammac,
                     dbias);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_se_bwd_affine(const float* s, const float* g, const float* alpha, const float* gammac, float* aps,
                                 float* gps, int B, int C, void* stream) {
  if (!s || !g || !alpha || !gammac || !aps || !gps || B <= 0 || C <= 0) return T3D_ERR_ARG;
  if (const T3dFold* fold = t3d_take_fold(alpha)) {
    T3D_LAUNCH(se_bwd_affine_fold_kernel, dim3(cdiv(C, 32), B >= 64 ? 8 : 1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), s, g, fold, aps, gps, B, C);
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  T3D_LAUNCH(se_bwd_affine_kernel, dim3(cdiv(B * C, 256) < 512 ? cdiv(B * C, 256) : 512), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), s, g, alpha, gammac, aps, gps, B, C);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_dropout_mask(float* mask, long long n, unsigned long long seed, unsigned long long offset, float p,
                                void* stream) {
  if (!mask || n <= 0 || !(p >= 0.f && p < 1.f)) return T3D_ERR_ARG;
  const long long n4 = (n + 3) / 4;
  T3D_LAUNCH(dropout_mask_kernel, dim3((int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), mask, n, seed, offset, 1.f - p, 1.f / (1.f - p));
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_zero_batched(const long long* desc, int n, void* stream) {
  if (!desc || n <= 0) return T3D_ERR_ARG;
  // (64 workgroups per buffer when there are many; a single large buffer -- a 9-MB weight-gradient matrix -- gets the chip)
  T3D_LAUNCH(zero_batched_kernel, dim3(n, n >= 16 ? 64 : 1024 / n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_sum_slots_batched(const long long* desc, int n, void* stream) {
  if (!desc || n <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(sum_slots_batched_kernel, dim3(n, 48), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_sum_replicas_batched(const long long* desc, int n, int nrep, void* stream) {
  if (!desc || n <= 0 || nrep < 1) return T3D_ERR_ARG;
  T3D_LAUNCH(sum_replicas_batched_kernel, dim3(n, 8), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc, nrep);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pack_weights_batched(int dtype, const long long* desc, int n, void* stream) {
  if (!desc || n <= 0) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == T3D_F32) T3D_LAUNCH(pack_batched_kernel<float>, dim3(n, 48), dim3(256), 0, st, desc);
  else if (dtype == T3D_BF16) T3D_LAUNCH(pack_batched_kernel<bf16_t>, dim3(n, 48), dim3(256), 0, st, desc);
  else if (dtype == T3D_F16) T3D_LAUNCH(pack_batched_kernel<f16_t>, dim3(n, 48), dim3(256), 0, st, desc);
  else return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pack_weight(int dtype, const float* w, void* out, int rows, int cols, int transpose,
                               void* stream) {
  if (!w || !out || rows <= 0 || cols <= 0) return T3D_ERR_ARG;
  const size_t n = (size_t)rows * cols;
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == T3D_F32)
    T3D_LAUNCH(pack_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)out, rows, cols, transpose);
  else if (dtype == T3D_BF16)
    T3D_LAUNCH(pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w, (bf16_t*)out, rows, cols, transpose);
  else if (dtype == T3D_F16)
    T3D_LAUNCH(pack_kernel<f16_t>, dim3(grid), dim3(256), 0, st, w, (f16_t*)out, rows, cols, transpose);
  else
    return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

T3dReduceCfg g_t3d_reduce = {1, 0, 0, nullptr, 0};

extern "C" int t3d_set_exact_pool(int on) {
  g_t3d_reduce.pool_exact = on ? 1 : 0;
  return T3D_OK;
}

T3dLaunchEvents g_t3d_time = {nullptr, nullptr};
T3dSignal g_t3d_signal = {nullptr, nullptr};
unsigned long long g_t3d_launches = 0;
hipStream_t g_t3d_last_stream = nullptr;

extern "C" int t3d_launch_count(unsigned long long* count, void** last_stream) {
  if (count) *count = g_t3d_launches;
  if (last_stream) *last_stream = g_t3d_last_stream;
  return T3D_OK;
}

extern "C" int t3d_set_launch_events(void* start_event, void* stop_event) {
  if ((start_event == nullptr) != (stop_event == nullptr)) return T3D_ERR_ARG;
  g_t3d_time.start = reinterpret_cast<hipEvent_t>(start_event);
  g_t3d_time.stop = reinterpret_cast<hipEvent_t>(stop_event);
  return T3D_OK;
}

extern "C" int t3d_set_dw_slots(int capacity, int* used_out) {
  if (capacity < 0 || (capacity > 0 && !used_out)) return T3D_ERR_ARG;
  g_t3d_reduce.dw_slots = capacity;
  g_t3d_reduce.dw_used = capacity > 0 ? used_out : nullptr;
  return T3D_OK;
}

extern "C" int t3d_set_reduction_replicas(int nrep, long long stats_stride) {
  if (nrep < 1 || nrep > 16 || (nrep > 1 && stats_stride <= 0)) return T3D_ERR_ARG;
  g_t3d_reduce.nrep = nrep;
  g_t3d_reduce.stats_stride = nrep > 1 ? stats_stride : 0;
  return T3D_OK;
}

T3dFoldReq g_t3d_fold = {nullptr, nullptr};

extern "C" int t3d_fold_request(const t3d_bn_fold* desc_device, const void* key) {
  if (!desc_device || !key) return T3D_ERR_ARG;
  g_t3d_fold.desc = desc_device;
  g_t3d_fold.key = key;
  return T3D_OK;
}

/* 1: a fold request is still pending (the last launch did not derive the coefficients); clears it either way */
extern "C" int t3d_fold_pending(void) {
  const int p = g_t3d_fold.desc != nullptr;
  g_t3d_fold.desc = nullptr;
  return p;
}

T3dWorkspace g_t3d_ws = {nullptr, 0};
T3dWorkspace g_t3d_ws_main = {nullptr, 0};

extern "C" int t3d_set_main_workspace(void* ptr, long long bytes) {
  if ((ptr == nullptr) != (bytes <= 0)) return T3D_ERR_ARG;
  g_t3d_ws_main.ptr = ptr;
  g_t3d_ws_main.bytes = ptr ? bytes : 0;
  return T3D_OK;
}

extern "C" int t3d_set_workspace(void* ptr, long long bytes) {
  if ((ptr == nullptr) != (bytes <= 0)) return T3D_ERR_ARG;
  g_t3d_ws.ptr = ptr;
  g_t3d_ws.bytes = ptr ? bytes : 0;
  return T3D_OK;
}

// see common.h: the attribute only ever has to grow, so remember the largest value set per kernel
#include <mutex>
#include <unordered_map>
hipError_t t3d_max_lds(const void* fn, int bytes) {
  // keyed by (device, kernel): a model moved to / built on a second GPU of the same process sets the attribute there too
  static std::mutex mu;
  static std::unordered_map<unsigned long long, int> seen;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  int& cur = seen[(unsigned long long)reinterpret_cast<uintptr_t>(fn) * 64ull + (unsigned)dev];
  if (bytes <= cur) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) cur = bytes;
  return e;
}

