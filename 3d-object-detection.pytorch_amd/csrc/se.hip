// Squeeze-excite gate of MobileNetV3 (SELayer, torchdet3d/models/mobilenetv3.py:92-107), fp32:
//   m = mean_hw BN(y)  ->  h = relu(W1 m + b1)  ->  q = W2 h + b2  ->  s = h_sigmoid(q),   x * s
// The spatial mean never touches the feature map again: the depthwise kernel already emitted
// sum_hw(y) per (sample, channel), and BN is affine, so m = scale * sum/HW + shift.
// The two FCs run as small tiled products (64 samples x 16 outputs per workgroup, fp32 FMA).
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

// ---- small dense layers as tiled products (one workgroup per sample re-read both weight matrices, 1.8 MB, 256 times:
// 320 us forward / 210 us backward per launch; the tiles below read each weight once per 64-sample tile)
// out[b][o] = sum_i in(b, i) * W(o, i) for a 64-sample x OT-output tile; thread = (sample bl, OT/4 outputs og*PT..).
// WT: W is stored [I][O] (the transposed products of the backward), else [O][I].
// OT = 4: the layers are tiny (B = 256 samples), so what matters is enough workgroups to hide the load latency.
constexpr int OT = 4, PT = OT / 4;
// IT: the input is read transposed (rows of the tile are its fast axis in memory) -- the weight-gradient products.
template <bool WT, bool IT = false, typename InF>
__device__ __forceinline__ void fc_tile(InF in, const float* __restrict__ W, int I, int O, int b0, int o0, int B,
                                        float acc[PT], float (*lin)[65], float (*lw)[65], int ibeg = 0, int iend = 1 << 30) {
  const int t = threadIdx.x, bl = t & 63, og = t >> 6;
#pragma unroll
  for (int j = 0; j < PT; ++j) acc[j] = 0.f;
  iend = min(iend, I);
  for (int i0 = ibeg; i0 < iend; i0 += 64) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int idx = t + 256 * k, row = IT ? (idx & 63) : (idx >> 6), col = IT ? (idx >> 6) : (idx & 63);
      const int b = b0 + row, i = i0 + col;
      lin[row][col] = (b < B && i < iend) ? in(b, i) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < OT / 4; ++k) {
      const int idx = t + 256 * k;
      int row, col;
      if (WT) { row = idx % OT; col = idx / OT; } else { row = idx >> 6; col = idx & 63; }
      const int o = o0 + row, i = i0 + col;
      float v = 0.f;
      if (o < O && i < iend) v = WT ? W[(size_t)i * O + o] : W[(size_t)o * I + i];
      lw[row][col] = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int ii = 0; ii < 64; ++ii) {
      const float x = lin[bl][ii];
#pragma unroll
      for (int j = 0; j < PT; ++j) acc[j] = fmaf(x, lw[og * PT + j][ii], acc[j]);
    }
  }
}

// The contraction over C (<= 960: 15 serial 64-chunks, each a global-load round trip) is split over blockIdx.z; the
// partial products are added into a zeroed h, bias + ReLU follow in se_relu_bias_kernel.
// grid (ceil(R/OT), ceil(B/64), KSPLIT): h_raw += W1[:, slice] m[slice], m = scale*gap/HW + shift (written by the
// first column of blocks of split 0)
constexpr int KSPLIT = 4;
__global__ __launch_bounds__(256) void se_fc1_kernel(const SeArgs a) {
  __shared__ float lin[64][65], lw[OT][65];
  const int o0 = blockIdx.x * OT, b0 = blockIdx.y * 64;
  const float inv = 1.f / (float)a.HW;
  auto in = [&](int b, int c) { return a.scale[c] * (a.gap[(size_t)b * a.C + c] * inv) + a.shift[c]; };
  const int per = ((a.C + KSPLIT - 1) / KSPLIT + 63) / 64 * 64;
  float acc[PT];
  fc_tile<false>(in, a.w1, a.C, a.R, b0, o0, a.B, acc, lin, lw, blockIdx.z * per, (blockIdx.z + 1) * per);
  const int b = b0 + (threadIdx.x & 63), og = threadIdx.x >> 6;
  if (b < a.B) {
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const int r = o0 + og * PT + j;
      if (r < a.R) unsafeAtomicAdd(a.h + (size_t)b * a.R + r, acc[j]);
    }
  }
  if (blockIdx.x == 0 && blockIdx.z == 0) {
    for (int i = threadIdx.x; i < 64 * a.C; i += 256) {
      const int bb = b0 + i / a.C, c = i % a.C;
      if (bb < a.B) a.m[(size_t)bb * a.C + c] = in(bb, c);
    }
  }
}

// h = relu(h_raw + b1)
__global__ __launch_bounds__(256) void se_relu_bias_kernel(const SeArgs a) {
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i < (size_t)a.B * a.R) a.h[i] = fmaxf(a.h[i] + a.b1[i % a.R], 0.f);
}

// grid (ceil(C/OT), ceil(B/64)): q = W2 h + b2, s = h_sigmoid(q)
__global__ __launch_bounds__(256) void se_fc2_kernel(const SeArgs a) {
  __shared__ float lin[64][65], lw[OT][65];
  const int o0 = blockIdx.x * OT, b0 = blockIdx.y * 64;
  auto in = [&](int b, int r) { return a.h[(size_t)b * a.R + r]; };
  float acc[PT];
  fc_tile<false>(in, a.w2, a.R, a.C, b0, o0, a.B, acc, lin, lw);
  const int b = b0 + (threadIdx.x & 63), og = threadIdx.x >> 6;
  if (b < a.B) {
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const int c = o0 + og * PT + j;
      if (c < a.C) {
        const float v = acc[j] + a.b2[c];
        a.q[(size_t)b * a.C + c] = v;
        a.s[(size_t)b * a.C + c] = hsigmoid(v);
      }
    }
  }
}

// dq = h_sigmoid'(q) * ds,  ds = sum_hw dv*u = scale*P2 + shift*P1
__global__ __launch_bounds__(256) void se_dq_kernel(const SeArgs a) {
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i >= (size_t)a.B * a.C) return;
  const int c = (int)(i % a.C);
  const float ds = a.scale[c] * a.ps[2 * i + 1] + a.shift[c] * a.ps[2 * i];
  const float q = a.q[i];
  a.dq[i] = (q > -3.f && q < 3.f) ? ds * (1.f / 6.f) : 0.f;     // relu6 passes strictly inside
}

// grid (ceil(R/OT), ceil(B/64), KSPLIT): dp_raw += dq[:, slice] W2[slice, :]; relu'(h) follows in se_relu_mask_kernel
__global__ __launch_bounds__(256) void se_dh_kernel(const SeArgs a) {
  __shared__ float lin[64][65], lw[OT][65];
  const int o0 = blockIdx.x * OT, b0 = blockIdx.y * 64;
  auto in = [&](int b, int c) { return a.dq[(size_t)b * a.C + c]; };
  const int per = ((a.C + KSPLIT - 1) / KSPLIT + 63) / 64 * 64;
  float acc[PT];
  fc_tile<true>(in, a.w2, a.C, a.R, b0, o0, a.B, acc, lin, lw, blockIdx.z * per, (blockIdx.z + 1) * per);   // W2: [C][R] = [I][O]
  const int b = b0 + (threadIdx.x & 63), og = threadIdx.x >> 6;
  if (b < a.B) {
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const int r = o0 + og * PT + j;
      if (r < a.R) unsafeAtomicAdd(a.dp + (size_t)b * a.R + r, acc[j]);
    }
  }
}

// dp = relu'(h) * dp_raw
__global__ __launch_bounds__(256) void se_relu_mask_kernel(const SeArgs a) {
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i < (size_t)a.B * a.R) a.dp[i] = a.h[i] > 0.f ? a.dp[i] : 0.f;
}

// grid (ceil(C/OT), ceil(B/64)): g = (dp W1) / HW, and the depthwise BatchNorm's backward sums of du = s*dv + g
__global__ __launch_bounds__(256) void se_dm_kernel(const SeArgs a) {
  __shared__ float lin[64][65], lw[OT][65];
  const int o0 = blockIdx.x * OT, b0 = blockIdx.y * 64;
  auto in = [&](int b, int r) { return a.dp[(size_t)b * a.R + r]; };
  float acc[PT];
  fc_tile<true>(in, a.w1, a.R, a.C, b0, o0, a.B, acc, lin, lw);      // W1 is [R][C] = [I][O]
  const int b = b0 + (threadIdx.x & 63), og = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float inv = 1.f / (float)a.HW;
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int c = o0 + og * PT + j;       // wave-uniform
    float v1 = 0.f, v2 = 0.f;
    if (b < a.B && c < a.C) {
      const size_t i = (size_t)b * a.C + c;
      const float gu = acc[j] * inv;     // m = mean_hw(u): every pixel of u receives dL/dm / HW
      a.g[i] = gu;
      const float s = a.s[i], p1 = a.ps[2 * i], p2 = a.ps[2 * i + 1];
      v1 = s * p1 + (float)a.HW * gu;
      v2 = s * p2 + gu * a.gap[i];
    }
    v1 = wave_sum(v1);
    v2 = wave_sum(v2);
    if (lane == 0 && c < a.C) {
      atomicAdd(a.stats + c, (double)v1);
      atomicAdd(a.stats + a.C + c, (double)v2);
    }
  }
}

// weight gradients as the same tiled products, contraction over the batch:
//   dW2[c][r] = sum_b dq[b][c] h[b][r]   (rows c, outputs r)      dW1[r][c] = sum_b dp[b][r] m[b][c]   (rows r, outputs c)
// which = 0: dW2, grid (ceil(R/OT), ceil(C/64));  which = 1: dW1, grid (ceil(C/OT), ceil(R/64))
__global__ __launch_bounds__(256) void se_wgrad_tile_kernel(const SeArgs a, int which) {
  __shared__ float lin[64][65], lw[OT][65];
  const int o0 = blockIdx.x * OT, r0 = blockIdx.y * 64;
  const int rows = which == 0 ? a.C : a.R, outs = which == 0 ? a.R : a.C;
  const float* src = which == 0 ? a.dq : a.dp;        // [B][rows]
  const float* wsrc = which == 0 ? a.h : a.m;         // [B][outs] = [I][O]
  auto in = [&](int row, int b) { return src[(size_t)b * rows + row]; };
  float acc[PT];
  fc_tile<true, true>(in, wsrc, a.B, outs, r0, o0, rows, acc, lin, lw);
  const int row = r0 + (threadIdx.x & 63), og = threadIdx.x >> 6;
  float* dst = which == 0 ? a.dw2 : a.dw1;
  if (row < rows) {
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const int o = o0 + og * PT + j;
      if (o < outs) dst[(size_t)row * outs + o] = acc[j];
    }
  }
}

// bias gradients: db2[c] = sum_b dq[b][c], db1[r] = sum_b dp[b][r]; block = 64 channels x 4 batch slices
__global__ __launch_bounds__(256) void se_bias_kernel(const SeArgs a) {
  __shared__ float red[4][64];
  const int ch = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  const bool second = ch >= a.C;                       // channels [0, C): db2, [C, C+R): db1
  const int c = second ? ch - a.C : ch, n = second ? a.R : a.C;
  const float* src = second ? a.dp : a.dq;
  float acc = 0.f;
  if (c < n)
    for (int b = sl; b < a.B; b += 4) acc += src[(size_t)b * n + c];
  red[sl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sl == 0 && c < n) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    (second ? a.db1 : a.db2)[c] = v;
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
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(h, 0, (size_t)B * R * sizeof(float), st) != hipSuccess) return T3D_ERR_LAUNCH;
  hipLaunchKernelGGL(se_fc1_kernel, dim3(cdiv(R, OT), cdiv(B, 64), KSPLIT), dim3(256), 0, st, a);
  hipLaunchKernelGGL(se_relu_bias_kernel, dim3(cdiv(B * R, 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(se_fc2_kernel, dim3(cdiv(C, OT), cdiv(B, 64)), dim3(256), 0, st, a);
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
  hipLaunchKernelGGL(se_dq_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, st, a);
  if (hipMemsetAsync(dp, 0, (size_t)B * R * sizeof(float), st) != hipSuccess) return T3D_ERR_LAUNCH;
  hipLaunchKernelGGL(se_dh_kernel, dim3(cdiv(R, OT), cdiv(B, 64), KSPLIT), dim3(256), 0, st, a);
  hipLaunchKernelGGL(se_relu_mask_kernel, dim3(cdiv(B * R, 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(se_dm_kernel, dim3(cdiv(C, OT), cdiv(B, 64)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(se_wgrad_tile_kernel, dim3(cdiv(R, OT), cdiv(C, 64)), dim3(256), 0, st, a, 0);
  hipLaunchKernelGGL(se_wgrad_tile_kernel, dim3(cdiv(C, OT), cdiv(R, 64)), dim3(256), 0, st, a, 1);
  hipLaunchKernelGGL(se_bias_kernel, dim3(cdiv(C + R, 64)), dim3(256), 0, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
