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

// one launch for every weight matrix of the model: desc[t] = {src, out (or 0), out_t (or 0), rows, cols} as int64
template <typename T>
__global__ void pack_batched_kernel(const long long* __restrict__ desc) {
  const long long* d = desc + (size_t)blockIdx.x * 5;
  const float* __restrict__ w = reinterpret_cast<const float*>(d[0]);
  T* __restrict__ out = reinterpret_cast<T*>(d[1]);
  T* __restrict__ out_t = reinterpret_cast<T*>(d[2]);
  const int rows = (int)d[3], cols = (int)d[4];
  const size_t n = (size_t)rows * cols;
  for (size_t i = blockIdx.y * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.y * blockDim.x) {
    const T v = (T)w[i];
    if (out) out[i] = v;
    if (out_t) out_t[(i % cols) * rows + i / cols] = v;
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

}  // namespace

extern "C" int t3d_sum_replicas_batched(const long long* desc, int n, int nrep, void* stream) {
  if (!desc || n <= 0 || nrep < 1) return T3D_ERR_ARG;
  hipLaunchKernelGGL(sum_replicas_batched_kernel, dim3(n, 8), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc, nrep);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pack_weights_batched(int dtype, const long long* desc, int n, void* stream) {
  if (!desc || n <= 0) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == T3D_F32) hipLaunchKernelGGL(pack_batched_kernel<float>, dim3(n, 48), dim3(256), 0, st, desc);
  else if (dtype == T3D_BF16) hipLaunchKernelGGL(pack_batched_kernel<bf16_t>, dim3(n, 48), dim3(256), 0, st, desc);
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
    hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)out, rows, cols, transpose);
  else if (dtype == T3D_BF16)
    hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w, (bf16_t*)out, rows, cols, transpose);
  else
    return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

T3dReduceCfg g_t3d_reduce = {1, 0};

extern "C" int t3d_set_reduction_replicas(int nrep, long long stats_stride) {
  if (nrep < 1 || nrep > 64 || (nrep > 1 && stats_stride <= 0)) return T3D_ERR_ARG;
  g_t3d_reduce.nrep = nrep;
  g_t3d_reduce.stats_stride = nrep > 1 ? stats_stride : 0;
  return T3D_OK;
}

T3dWorkspace g_t3d_ws = {nullptr, 0};

extern "C" int t3d_set_workspace(void* ptr, long long bytes) {
  if ((ptr == nullptr) != (bytes <= 0)) return T3D_ERR_ARG;
  g_t3d_ws.ptr = ptr;
  g_t3d_ws.bytes = ptr ? bytes : 0;
  return T3D_OK;
}
