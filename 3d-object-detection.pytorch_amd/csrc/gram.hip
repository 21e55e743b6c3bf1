// BatchNorm statistics of an EXPANSION conv from the Gram matrix of its narrow input (round 6; DESIGN.md findings 35 / 40,
// VERDICT r5 #3): y1 = W1 z with z [M][K], K <= 16, W1 [C][K] (models/mobilenetv3.py:146-148: Conv2d(inp, hidden, 1) +
// BatchNorm2d(hidden) of an InvertedResidual).  Train-mode BatchNorm needs mean / variance of y1 over the M pixels BEFORE
// anything may consume BN(y1) -- which is why the fused expand + depthwise forward (csrc/expdw_fwd.hip) could not be used in
// training: a statistics-only pass of the 1x1 conv over the 6x wider output cost 113 us at 112x112.  But y1 is linear in z:
//     sum_m y1[m][c]   = w_c . (sum_m z[m])                       (K numbers)
//     sum_m y1[m][c]^2 = w_c^T (sum_m z[m] z[m]^T) w_c            (K x K numbers)
// so ONE pass over the narrow tensor gives both, for every one of the C expanded channels -- the same algebra as the y-free
// backward (pwconv_yfree.hip), where the Gram matrix is a by-product of the weight gradient.  That pass is the one that
// materialises the block input anyway (z = BN(y3) + skip of the previous block; t3d_bn_apply's job), so it costs nothing
// extra in bytes.
//   t3d_bn_apply_gram      z = storage(act(scale * y + shift) + residual) (optional), gram += [z^T z | 1^T z] in fp64
//   t3d_gram_bn_finalize   mean / biased variance of W1 z per expanded channel from the Gram sums -> scale, shift, mean, invstd,
//                          running statistics: what t3d_bn_finalize computes from sum(y1), sum(y1^2)
// The statistics are those of the EXACT products (fp32 MFMA accumulators); the stored expansion is their bf16 rounding, whose
// statistics the two-launch path measures: the two differ by the rounding noise's moments (~1e-6 relative on the variance).
#include <cstdlib>
#include "common.h"

namespace {

constexpr int GRAM_NREP = 16;

template <int K>
__global__ __launch_bounds__(256) void bn_apply_gram_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int act,
                                                            const bf16_t* __restrict__ res, bf16_t* __restrict__ z,
                                                            double* __restrict__ gram, int M, T3dQuant quant) {
  // gram: GRAM_NREP replicas of [NG + K] (workgroup b adds into replica b % GRAM_NREP): 1024 workgroups x 152 fp64 atomics on 152
  // addresses made the first version 136 us for a 40-us pass (the round-1 lesson, DESIGN.md finding 1)
  constexpr int NG = K * (K + 1) / 2;          // upper triangle, row-major: (i, j >= i)
  __shared__ double lacc[NG + K];
  for (int i = threadIdx.x; i < NG + K; i += 256) lacc[i] = 0.0;
  __syncthreads();
  float sc[K], sh[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    sc[k] = scale ? scale[k] : 1.f;
    sh[k] = scale ? shift[k] : 0.f;
  }
  float g[NG], s[K];
#pragma unroll
  for (int i = 0; i < NG; ++i) g[i] = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s[k] = 0.f;
  const bool affine = scale != nullptr || act != T3D_ACT_NONE;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    float v[K];
#pragma unroll
    for (int q = 0; q < K / 8; ++q) {
      const bf16x8 r = *reinterpret_cast<const bf16x8*>(y + (size_t)m * K + 8 * q);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[8 * q + j] = (float)r[j];
    }
    if (affine) act_affine_vec<K>(v, sc, sh, act);
    if (res) {
#pragma unroll
      for (int q = 0; q < K / 8; ++q) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(res + (size_t)m * K + 8 * q);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[8 * q + j] += (float)r[j];
      }
    }
    if (z) {
#pragma unroll
      for (int q = 0; q < K / 8; ++q) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o[j] = (bf16_t)v[8 * q + j];
          v[8 * q + j] = (float)o[j];             // the statistics are those of what the conv will read
        }
        *reinterpret_cast<bf16x8*>(z + (size_t)m * K + 8 * q) = o;
      }
    }
    int t = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      s[i] += v[i];
#pragma unroll
      for (int j = i; j < K; ++j, ++t) g[t] = fmaf(v[i], v[j], g[t]);
    }
  }
  // lanes of a wave meet by DPP (fixed order), waves and workgroups as snapped fp64 adds (exact in any order: common.h)
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const float w = wave_sum(g[i]);
    if (lane == 0) atomicAdd(lacc + i, t3d_snap(w, quant, true));
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float w = wave_sum(s[k]);
    if (lane == 0) atomicAdd(lacc + NG + k, t3d_snap(w, quant, false));
  }
  __syncthreads();
  double* dst = gram + (size_t)(blockIdx.x % GRAM_NREP) * (NG + K);
  for (int i = threadIdx.x; i < NG + K; i += 256)
    if (lacc[i] != 0.0) atomicAdd(dst + i, lacc[i]);
}

// one thread per expanded channel: the K x K form in fp64, the replica sums met in LDS first
__global__ __launch_bounds__(128) void gram_bn_finalize_kernel(const double* __restrict__ gram, const bf16_t* __restrict__ w, int C, int K,
                                                               double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                                               float* scale, float* shift, float* mean_out, float* invstd_out) {
  __shared__ double gs[16 * 17 / 2 + 16];
  const int NG = K * (K + 1) / 2, n = NG + K;
  for (int i = threadIdx.x; i < n; i += 128) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < GRAM_NREP; ++r) v += gram[(size_t)r * n + i];
    gs[i] = v;
  }
  __syncthreads();
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double wk[16];
  for (int k = 0; k < K; ++k) wk[k] = (double)(float)w[(size_t)c * K + k];
  double s1 = 0.0, s2 = 0.0;
  int t = 0;
  for (int i = 0; i < K; ++i) {
    s1 += wk[i] * gs[NG + i];
    for (int j = i; j < K; ++j, ++t) s2 += (i == j ? 1.0 : 2.0) * wk[i] * wk[j] * gs[t];
  }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;  // biased
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  if (mean_out) mean_out[c] = (float)mean;
  if (invstd_out) invstd_out[c] = invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

}  // namespace

// include/t3d.h
extern "C" int t3d_bn_apply_gram(int dtype, const void* y, const t3d_prologue* pro, const void* residual, void* z, double* gram,
                                 int M, int K, void* stream) {
  if (!y || !gram || M <= 0 || K <= 0) return T3D_ERR_ARG;
  if (dtype != T3D_BF16 || (K != 8 && K != 16) || (pro && pro->se)) return T3D_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pro)
    if (const int rc = t3d_fold_fallback(pro->scale, st)) return rc;      // finished coefficients (no derive prologue here)
  const T3dQuant quant = T3D_ENV_SET("T3D_NO_SNAP") ? T3dQuant{0.0, 0.0} : t3d_quant_for(M);
  const int grid = cdiv(M, 256) < 512 ? cdiv(M, 256) : 512;
  const bf16_t* yb = reinterpret_cast<const bf16_t*>(y);
  const bf16_t* rb = reinterpret_cast<const bf16_t*>(residual);
  bf16_t* zb = reinterpret_cast<bf16_t*>(z);
  const float* sc = pro ? pro->scale : nullptr;
  const float* sh = pro ? pro->shift : nullptr;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  if (K == 8) T3D_LAUNCH(bn_apply_gram_kernel<8>, dim3(grid), dim3(256), 0, st, yb, sc, sh, act, rb, zb, gram, M, quant);
  else T3D_LAUNCH(bn_apply_gram_kernel<16>, dim3(grid), dim3(256), 0, st, yb, sc, sh, act, rb, zb, gram, M, quant);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_gram_bn_finalize(const double* gram, const void* w, int C, int K, double count, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                    float momentum, float eps, float* scale, float* shift, float* mean_out, float* invstd_out,
                                    void* stream) {
  if (!gram || !w || !scale || !shift || C <= 0 || K <= 0 || K > 16 || count <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(gram_bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), gram,
             reinterpret_cast<const bf16_t*>(w), C, K, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
             scale, shift, mean_out, invstd_out);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
