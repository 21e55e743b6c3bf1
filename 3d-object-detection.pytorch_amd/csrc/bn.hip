// BatchNorm bookkeeping kernels (tiny, one thread per channel).
// The heavy part of BatchNorm -- the reductions and the normalisation -- lives in the
// producing / consuming convolution kernels; these turn the reduced sums into the
// per-channel affine the consumers apply on load.
#include "common.h"

namespace {

// Replica-parallel load of the two per-channel sums: a block is 16 channels x 16 "replica lanes"; lane rl loads
// replicas rl, rl+16, ... (all loads of a channel are in flight at once) and the 16 partial sums meet through DPP-free
// shuffles inside the 16-lane row.
__device__ __forceinline__ void load_sums(const double* __restrict__ stats, int nrep, long long rstride, int C, int c, int rl,
                                          double& s1, double& s2) {
  s1 = 0.0;
  s2 = 0.0;
  if (c < C) {
    for (int r = rl; r < nrep; r += 16) {
      s1 += stats[r * rstride + c];
      s2 += stats[r * rstride + C + c];
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 16);
    s2 += __shfl_xor(s2, o, 16);
  }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int nrep, long long rstride, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, int64_t* nbt, float momentum,
                                   float eps, float* scale, float* shift, float* mean_out, float* invstd_out) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4), rl = threadIdx.x & 15;
  if (c == 0 && rl == 0 && nbt) *nbt += 1;
  double s1, s2;
  load_sums(stats, nrep, rstride, C, c, rl, s1, s2);
  if (c >= C || rl != 0) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;  // biased
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float s = g * invstd;
  scale[c] = s;
  shift[c] = b - (float)mean * s;
  if (mean_out) mean_out[c] = (float)mean;
  if (invstd_out) invstd_out[c] = invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_affine_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float s = (gamma ? gamma[c] : 1.f) * invstd;
  scale[c] = s;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * s;
}

// every BatchNorm of the model in one launch: desc[i] = {gamma, beta, running_mean, running_var, scale, shift, C} (int64)
__global__ void bn_eval_affine_batched_kernel(const long long* __restrict__ desc, float eps) {
  const long long* d = desc + (size_t)blockIdx.x * 7;
  const float* gamma = reinterpret_cast<const float*>(d[0]);
  const float* beta = reinterpret_cast<const float*>(d[1]);
  const float* rm = reinterpret_cast<const float*>(d[2]);
  const float* rv = reinterpret_cast<const float*>(d[3]);
  float* scale = reinterpret_cast<float*>(d[4]);
  float* shift = reinterpret_cast<float*>(d[5]);
  const int C = (int)d[6];
  for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < C; c += gridDim.y * blockDim.x) {
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float s = (gamma ? gamma[c] : 1.f) * invstd;
    scale[c] = s;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * s;
  }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ stats, int nrep, long long rstride, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* alpha, float* beta, float* gammac,
                                       float* dgamma, float* dbeta) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4), rl = threadIdx.x & 15;
  double s1, s2;
  load_sums(stats, nrep, rstride, C, c, rl, s1, s2);
  if (c >= C || rl != 0) return;
  const double mu = (double)mean[c], is = (double)invstd[c];
  const double dg = is * (s2 - mu * s1);
  const double a = (double)(gamma ? gamma[c] : 1.f) * is;
  const double b = -a * is * dg / count;
  alpha[c] = (float)a;
  beta[c] = (float)b;
  gammac[c] = (float)(-a * s1 / count - b * mu);
  if (dgamma) dgamma[c] = (float)dg;
  if (dbeta) dbeta[c] = (float)s1;
}

// a requested finalize whose consumer has no derive prologue: the same arithmetic as a launch of its own, driven by the
// device-resident descriptor (the host does not know its contents); 64 channels per workgroup, C <= 64 * gridDim.x
__global__ __launch_bounds__(256) void bn_fold_kernel(const T3dFold* __restrict__ fp) {
  __shared__ float sink[3 * 64];
  const int C = fp->C, cbase = blockIdx.x * 64;
  if (cbase >= C) return;
  t3d_fold_block(fp, cbase, min(64, C - cbase), sink, 64, true);
}

}  // namespace

int t3d_fold_fallback(const void* key, hipStream_t st) {
  const T3dFold* d = t3d_take_fold(key);
  if (!d) return T3D_OK;
  T3D_LAUNCH(bn_fold_kernel, dim3(32), dim3(256), 0, st, d);      // up to 2048 channels
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_bn_finalize(const double* stats, int C, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                               void* stream) {
  if (!stats || !scale || !shift || C <= 0 || count <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     stats, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride, C, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                     scale, shift, mean, invstd);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, void* stream) {
  if (!running_mean || !running_var || !scale || !shift || C <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), C, gamma, beta, running_mean, running_var, eps, scale,
                     shift);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_bn_eval_affine_batched(const long long* desc, int n, float eps, void* stream) {
  if (!desc || n <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(bn_eval_affine_batched_kernel, dim3(n, 2), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc, eps);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_version(void) { return 2; }

extern "C" int t3d_bn_bwd_finalize(const double* stats, int C, double count, const float* gamma, const float* mean,
                                   const float* invstd, float* alpha, float* beta, float* gammac, float* dgamma,
                                   float* dbeta, void* stream) {
  if (!stats || !mean || !invstd || !alpha || !beta || !gammac || C <= 0 || count <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), stats, g_t3d_reduce.nrep, g_t3d_reduce.stats_stride, C, count, gamma, mean, invstd, alpha, beta,
                     gammac, dgamma, dbeta);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
