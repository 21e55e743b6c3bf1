// Squeeze-excite gate of MobileNetV3 (SELayer, torchdet3d/models/mobilenetv3.py:92-107), fp32:
//   m = mean_hw BN(y)  ->  h = relu(W1 m + b1)  ->  q = W2 h + b2  ->  s = h_sigmoid(q),   x * s
// The spatial mean never touches the feature map again: the depthwise kernel already emitted
// sum_hw(y) per (sample, channel), and BN is affine, so m = scale * sum/HW + shift.
// One workgroup per sample; every FC row is a 64-lane dot product + wave reduction (the two
// weight matrices, <= 960x240 floats each, stay L2-resident across the 256 workgroups).
// Backward (same shape of work, reversed) consumes the per-sample sums  P1 = sum_hw dv,
// P2 = sum_hw dv*y  that the projection conv's data-gradient kernel emitted (dv = gradient at the
// gated tensor), and produces
//   g[b][c]     = dL/dm / HW                      (the pooled path's contribution to every pixel)
//   bn sums     sum(du), sum(du*y) of the depthwise conv's BatchNorm  (du = s*dv + g)
//   dq, dp      pre-activation gradients of the two FCs, for the weight-gradient kernel.
#include "common.h"

namespace {

struct SeArgs {
  const float *gap, *scale, *shift;
  const float *w1, *b1, *w2, *b2;
  float *m, *h, *q, *s;
  const float* ps;   // [B][C][2]
  float *g, *dq, *dp;
  double* stats;     // [2][C]
  float *dw1, *db1, *dw2, *db2;
  int B, C, R, HW;
};

__global__ __launch_bounds__(256) void se_fwd_kernel(const SeArgs a) {
  extern __shared__ float sm[];  // m[C], h[R]
  float* mh = sm;
  float* hh = sm + a.C;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv = 1.f / (float)a.HW;
  for (int c = tid; c < a.C; c += 256) {
    const float v = a.scale[c] * (a.gap[(size_t)b * a.C + c] * inv) + a.shift[c];
    mh[c] = v;
    a.m[(size_t)b * a.C + c] = v;
  }
  __syncthreads();
  for (int r = wave; r < a.R; r += 4) {
    const float* w = a.w1 + (size_t)r * a.C;
    float acc = 0.f;
    for (int c = lane; c < a.C; c += 64) acc = fmaf(w[c], mh[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = fmaxf(acc + a.b1[r], 0.f);
      hh[r] = v;
      a.h[(size_t)b * a.R + r] = v;
    }
  }
  __syncthreads();
  for (int c = wave; c < a.C; c += 4) {
    const float* w = a.w2 + (size_t)c * a.R;
    float acc = 0.f;
    for (int r = lane; r < a.R; r += 64) acc = fmaf(w[r], hh[r], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = acc + a.b2[c];
      a.q[(size_t)b * a.C + c] = v;
      a.s[(size_t)b * a.C + c] = hsigmoid(v);
    }
  }
}

__global__ __launch_bounds__(256) void se_bwd_kernel(const SeArgs a) {
  extern __shared__ float sm[];  // dq[C], dp[R]
  float* dqs = sm;
  float* dps = sm + a.C;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < a.C; c += 256) {
    const size_t i = (size_t)b * a.C + c;
    const float p1 = a.ps[2 * i], p2 = a.ps[2 * i + 1];
    const float ds = a.scale[c] * p2 + a.shift[c] * p1;      // sum_hw dv * u,  u = scale*y + shift
    const float q = a.q[i];
    const float v = (q > -3.f && q < 3.f) ? ds * (1.f / 6.f) : 0.f;   // h_sigmoid' (relu6 passes strictly inside)
    dqs[c] = v;
    a.dq[i] = v;
  }
  __syncthreads();
  // dh[r] = sum_c dq[c] * W2[c][r]   (thread r walks a column; W2 rows are R floats apart)
  for (int r = tid; r < a.R; r += 256) {
    float acc = 0.f;
    for (int c = 0; c < a.C; ++c) acc = fmaf(dqs[c], a.w2[(size_t)c * a.R + r], acc);
    const float v = a.h[(size_t)b * a.R + r] > 0.f ? acc : 0.f;
    dps[r] = v;
    a.dp[(size_t)b * a.R + r] = v;
  }
  __syncthreads();
  // dm[c] = sum_r dp[r] * W1[r][c]  (coalesced over c)
  const float inv = 1.f / (float)a.HW;
  for (int c = tid; c < a.C; c += 256) {
    float acc = 0.f;
    for (int r = 0; r < a.R; ++r) acc = fmaf(dps[r], a.w1[(size_t)r * a.C + c], acc);
    const size_t i = (size_t)b * a.C + c;
    const float gu = acc * inv;   // m = mean_hw(u): every pixel of u receives dL/dm / HW
    a.g[i] = gu;
    const float s = a.s[i], p1 = a.ps[2 * i], p2 = a.ps[2 * i + 1];
    atomicAdd(a.stats + c, (double)(s * p1 + (float)a.HW * gu));
    atomicAdd(a.stats + a.C + c, (double)(s * p2 + gu * a.gap[i]));
  }
}

// weight gradients: grid.x over output elements, each thread one element, loop over the batch (deterministic)
__global__ __launch_bounds__(256) void se_wgrad_kernel(const SeArgs a) {
  const int n2 = a.C * a.R;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n2) {                       // dW2[c][r] = sum_b dq[b][c] * h[b][r]
    const int c = i / a.R, r = i % a.R;
    float acc = 0.f;
    for (int b = 0; b < a.B; ++b) acc = fmaf(a.dq[(size_t)b * a.C + c], a.h[(size_t)b * a.R + r], acc);
    a.dw2[i] = acc;
  } else if (i < 2 * n2) {            // dW1[r][c] = sum_b dp[b][r] * m[b][c]
    const int j = i - n2, r = j / a.C, c = j % a.C;
    float acc = 0.f;
    for (int b = 0; b < a.B; ++b) acc = fmaf(a.dp[(size_t)b * a.R + r], a.m[(size_t)b * a.C + c], acc);
    a.dw1[j] = acc;
  } else if (i < 2 * n2 + a.C) {
    const int c = i - 2 * n2;
    float acc = 0.f;
    for (int b = 0; b < a.B; ++b) acc += a.dq[(size_t)b * a.C + c];
    a.db2[c] = acc;
  } else if (i < 2 * n2 + a.C + a.R) {
    const int r = i - 2 * n2 - a.C;
    float acc = 0.f;
    for (int b = 0; b < a.B; ++b) acc += a.dp[(size_t)b * a.R + r];
    a.db1[r] = acc;
  }
}

}  // namespace

extern "C" int t3d_se_fwd(const float* gap_sum, const float* scale, const float* shift, const float* w1,
                          const float* b1, const float* w2, const float* b2, float* m, float* h, float* q, float* s,
                          int B, int C, int R, int HW, void* stream) {
  if (!gap_sum || !scale || !shift || !w1 || !b1 || !w2 || !b2 || !m || !h || !q || !s || B <= 0 || C <= 0 || R <= 0 ||
      HW <= 0)
    return T3D_ERR_ARG;
  SeArgs a{};
  a.gap = gap_sum; a.scale = scale; a.shift = shift; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
  a.m = m; a.h = h; a.q = q; a.s = s; a.B = B; a.C = C; a.R = R; a.HW = HW;
  hipLaunchKernelGGL(se_fwd_kernel, dim3(B), dim3(256), (size_t)(C + R) * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_se_bwd(const float* ps_stats, const float* gap_sum, const float* scale, const float* shift,
                          const float* w1, const float* w2, const float* m, const float* h, const float* q,
                          const float* s, float* g, float* dq, float* dp, double* stats, float* dw1, float* db1,
                          float* dw2, float* db2, int B, int C, int R, int HW, void* stream) {
  if (!ps_stats || !gap_sum || !scale || !shift || !w1 || !w2 || !m || !h || !q || !s || !g || !dq || !dp || !stats ||
      !dw1 || !db1 || !dw2 || !db2 || B <= 0 || C <= 0 || R <= 0 || HW <= 0)
    return T3D_ERR_ARG;
  SeArgs a{};
  a.ps = ps_stats; a.gap = gap_sum; a.scale = scale; a.shift = shift; a.w1 = w1; a.w2 = w2;
  a.m = const_cast<float*>(m); a.h = const_cast<float*>(h); a.q = const_cast<float*>(q); a.s = const_cast<float*>(s);
  a.g = g; a.dq = dq; a.dp = dp; a.stats = stats; a.dw1 = dw1; a.db1 = db1; a.dw2 = dw2; a.db2 = db2;
  a.B = B; a.C = C; a.R = R; a.HW = HW;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(se_bwd_kernel, dim3(B), dim3(256), (size_t)(C + R) * sizeof(float), st, a);
  const int n = 2 * C * R + C + R;
  hipLaunchKernelGGL(se_wgrad_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
