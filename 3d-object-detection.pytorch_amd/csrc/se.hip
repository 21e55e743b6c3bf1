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
  int gapq;          // `gap` holds int64 fixed point (t3d_set_exact_pool; common.h: t3d_pool_get)
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
  auto in = [&](int b, int c) { return a.scale[c] * (t3d_pool_get(a.gap, (size_t)b * a.C + c, a.gapq) * inv) + a.shift[c]; };
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
      v2 = s * p2 + gu * t3d_pool_get(a.gap, i, a.gapq);
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


// ------------------------------------------------------------------ one workgroup per group of samples
// The tiled kernels above cost 4 launches forward and 8 backward per gate, each a short serial chain of chunk rounds:
// 70 us + 96 us per gate on the step's critical stream for ~0.1 GFLOP (rocprofv3: 16 % of MobileNetV3-large's kernel
// time together with their memsets).  Here ONE launch does a direction: a workgroup owns SPG samples and walks both
// products for them back to back, thread = output, the contraction split over the thread groups that the output width
// leaves free; weights are read coalesced along the output axis ([I][O] layouts: the backward's natural
// ones, transposed copies for the forward), inputs broadcast from LDS.  Every workgroup streams both weight matrices
// (<= 1.8 MB fp32) out of L2.
constexpr int SPG = 4;          // samples per workgroup
constexpr int SE_T = 512;       // (1024-thread blocks of 8 samples ran 37 us alone but 114 us inside the step: a block that
                                // needs a whole CU's wave slots waits for the weight-gradient stream's blocks to drain)

// out[s][o] = sum_i in[i][s] * Wt[i*O + o]   for s < SPG;  in: LDS [I][SPG];  partials through `red` [SE_T][SPG].
// fin(o, acc) runs once per output o with the finished sums.  Outputs are walked in blocks of OP <= SE_T; within a block
// the contraction is split over the SE_T / OP thread groups.
template <typename Fin>
__device__ __forceinline__ void se_product(const float* __restrict__ Wt, int I, int O, const float* in, float* red, Fin fin) {
  const int t = threadIdx.x;
  const int OP = min((O + 63) & ~63, SE_T);      // whole waves per group
  const int ngrp = SE_T / OP, grp = t / OP, ol = t - grp * OP;
  const int per = (I + ngrp - 1) / ngrp;
  const int i0 = min(grp * per, I), i1 = min(i0 + per, I);
  for (int ob = 0; ob < O; ob += OP) {
    const int o = ob + ol;
    const bool live = grp < ngrp && o < O;
    float acc[SPG];
#pragma unroll
    for (int s = 0; s < SPG; ++s) acc[s] = 0.f;
    if (live) {
      int i = i0;
      for (; i + 16 <= i1; i += 16) {            // sixteen independent weight loads in flight (the walk is latency-bound)
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = Wt[(size_t)(i + u) * O + o];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float4 x0 = *reinterpret_cast<const float4*>(in + (i + u) * SPG);
          acc[0] = fmaf(x0.x, w[u], acc[0]); acc[1] = fmaf(x0.y, w[u], acc[1]);
          acc[2] = fmaf(x0.z, w[u], acc[2]); acc[3] = fmaf(x0.w, w[u], acc[3]);
        }
      }
      for (; i < i1; ++i) {
        const float w = Wt[(size_t)i * O + o];
#pragma unroll
        for (int s = 0; s < SPG; ++s) acc[s] = fmaf(in[i * SPG + s], w, acc[s]);
      }
    }
    if (ngrp > 1) {
      __syncthreads();                             // `red` may still be read from the previous block / product
      if (live && grp > 0) {
#pragma unroll
        for (int s = 0; s < SPG; ++s) red[(size_t)t * SPG + s] = acc[s];
      }
      __syncthreads();
      if (live && grp == 0) {
        for (int g = 1; g < ngrp; ++g) {
#pragma unroll
          for (int s = 0; s < SPG; ++s) acc[s] += red[(size_t)(g * OP + ol) * SPG + s];
        }
      }
    }
    if (live && grp == 0) fin(o, acc);
  }
}

// LDS: in0 [C][SPG] | in1 [R][SPG] | red [SE_T][SPG]
__global__ __launch_bounds__(SE_T) void se_fwd_group_kernel(const SeArgs a, const float* __restrict__ w1t,
                                                            const float* __restrict__ w2t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ms = lds;
  float* hs = ms + (size_t)a.C * SPG;
  float* red = hs + (size_t)a.R * SPG;
  const int b0 = blockIdx.x * SPG, t = threadIdx.x;
  const float inv = 1.f / (float)a.HW;
  for (int i = t; i < a.C * SPG; i += SE_T) {
    const int c = i / SPG, s = i % SPG, b = b0 + s;
    float v = 0.f;
    if (b < a.B) {
      v = a.scale[c] * (t3d_pool_get(a.gap, (size_t)b * a.C + c, a.gapq) * inv) + a.shift[c];
      a.m[(size_t)b * a.C + c] = v;
    }
    ms[i] = v;
  }
  __syncthreads();
  se_product(w1t, a.C, a.R, ms, red, [&](int o, const float* acc) {          // h = relu(W1 m + b1)
    const float b1 = a.b1[o];
#pragma unroll
    for (int s = 0; s < SPG; ++s) {
      const float h = fmaxf(acc[s] + b1, 0.f);
      hs[o * SPG + s] = h;
      if (b0 + s < a.B) a.h[(size_t)(b0 + s) * a.R + o] = h;
    }
  });
  __syncthreads();
  se_product(w2t, a.R, a.C, hs, red, [&](int o, const float* acc) {          // q = W2 h + b2, s = h_sigmoid(q)
    const float b2 = a.b2[o];
#pragma unroll
    for (int s = 0; s < SPG; ++s) {
      if (b0 + s < a.B) {
        const float q = acc[s] + b2;
        a.q[(size_t)(b0 + s) * a.C + o] = q;
        a.s[(size_t)(b0 + s) * a.C + o] = hsigmoid(q);
      }
    }
  });
}

// data part of the backward: dq, dp (kept for the weight gradients), g and the BatchNorm-backward sums
__global__ __launch_bounds__(SE_T) void se_bwd_group_kernel(const SeArgs a, const int nrep, const long long rstride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dqs = lds;
  float* dps = dqs + (size_t)a.C * SPG;
  float* red = dps + (size_t)a.R * SPG;
  const int b0 = blockIdx.x * SPG, t = threadIdx.x;
  for (int i = t; i < a.C * SPG; i += SE_T) {
    const int c = i / SPG, s = i % SPG, b = b0 + s;
    float v = 0.f;
    if (b < a.B) {
      const size_t k = (size_t)b * a.C + c;
      const float ds = a.scale[c] * a.ps[2 * k + 1] + a.shift[c] * a.ps[2 * k];
      const float q = a.q[k];
      v = (q > -3.f && q < 3.f) ? ds * (1.f / 6.f) : 0.f;     // relu6 passes strictly inside
      a.dq[k] = v;
    }
    dqs[i] = v;
  }
  __syncthreads();
  se_product(a.w2, a.C, a.R, dqs, red, [&](int o, const float* acc) {        // dp = relu'(h) * (dq W2);  W2 is [C][R] = [I][O]
#pragma unroll
    for (int s = 0; s < SPG; ++s) {
      float v = 0.f;
      if (b0 + s < a.B) {
        v = a.h[(size_t)(b0 + s) * a.R + o] > 0.f ? acc[s] : 0.f;
        a.dp[(size_t)(b0 + s) * a.R + o] = v;
      }
      dps[o * SPG + s] = v;
    }
  });
  __syncthreads();
  const float inv = 1.f / (float)a.HW;
  se_product(a.w1, a.R, a.C, dps, red, [&](int o, const float* acc) {        // g = (dp W1) / HW;  W1 is [R][C] = [I][O]
    float v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int s = 0; s < SPG; ++s) {
      if (b0 + s < a.B) {
        const size_t k = (size_t)(b0 + s) * a.C + o;
        const float gu = acc[s] * inv;                          // every pixel of u receives dL/dm / HW
        a.g[k] = gu;
        const float sg = a.s[k], p1 = a.ps[2 * k], p2 = a.ps[2 * k + 1];
        v1 += sg * p1 + (float)a.HW * gu;
        v2 += sg * p2 + gu * t3d_pool_get(a.gap, k, a.gapq);
      }
    }
    // one add per (workgroup, channel): spread over the reduction replicas (64 blocks on one address made this launch 3x
    // longer inside the step than alone)
    if (a.stats) {
      double* st = a.stats + (size_t)(blockIdx.x % nrep) * rstride;
      atomicAdd(st + o, (double)v1);
      atomicAdd(st + a.C + o, (double)v2);
    }
  });
}


// ------------------------------------------------------------------ round 6: one launch per PRODUCT, outputs sliced over workgroups
// The group kernels above walk both products of a direction inside 64 workgroups: ~60 dependent rounds of sixteen L2 loads
// per thread -- 27 / 37 us alone, but 33 / 105 us inside the step (rocprofv3, MobileNetV3-large: 8 gates = 0.84 ms of the
// backward's critical stream for ~0.1 GFLOP; the rounds get longer when the weight-gradient stream is busy in L2).  Here a
// workgroup owns SPG samples x 64 OUTPUTS of ONE product and its four waves split the contraction: 960 -> 240 is 15 rounds
// instead of 30, 240 -> 960 four instead of 30, and 256 ... 960 small workgroups fill the chip instead of 64 large ones.
// The price is a second launch per direction (a dependent kernel boundary, ~3 us).  Summation order: four contiguous
// slices of the contraction, added in slice order -- deterministic, different in the last bits from the group kernels.
constexpr int SL_OB = 64, SL_T = 512, SL_G = SL_T / SL_OB;      // eight waves split the contraction (256 threads: 67 us in the backward)
// MODE 0: h = relu(W1 m + b1), m from the pooled sums   1: q = W2 h + b2, s = h_sigmoid(q)
//      2: dp = relu'(h) (dq W2), dq from the per-sample sums   3: g = (dp W1) / HW + the BatchNorm-backward sums
template <int MODE>
__global__ __launch_bounds__(SL_T) void se_slice_kernel(const SeArgs a, const float* __restrict__ Wt, const int nrep, const long long rstride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool FROM_C = MODE == 0 || MODE == 2;      // contraction over the C channels (else over the R hidden units)
  const int I = FROM_C ? a.C : a.R, O = FROM_C ? a.R : a.C;
  float* in = lds;                                     // [I][SPG]
  float* red = in + (size_t)I * SPG;                   // [SL_T][SPG]
  const int b0 = blockIdx.x * SPG, t = threadIdx.x;
  const bool first = blockIdx.y == 0;                  // the slice that also stores the staged input (m / dq)
  const float inv = 1.f / (float)a.HW;
  for (int i = t; i < I * SPG; i += SL_T) {
    const int c = i / SPG, s = i % SPG, b = b0 + s;
    float v = 0.f;
    if (b < a.B) {
      const size_t k = (size_t)b * I + c;
      if (MODE == 0) {
        v = a.scale[c] * (t3d_pool_get(a.gap, k, a.gapq) * inv) + a.shift[c];
        if (first) a.m[k] = v;
      } else if (MODE == 1) {
        v = a.h[k];
      } else if (MODE == 2) {
        const float ds = a.scale[c] * a.ps[2 * k + 1] + a.shift[c] * a.ps[2 * k];
        const float q = a.q[k];
        v = (q > -3.f && q < 3.f) ? ds * (1.f / 6.f) : 0.f;     // relu6 passes strictly inside
        if (first) a.dq[k] = v;
      } else {
        v = a.dp[k];
      }
    }
    in[i] = v;
  }
  __syncthreads();
  const int grp = t / SL_OB, ol = t % SL_OB, o = blockIdx.y * SL_OB + ol;
  const bool live = o < O;
  const int per = (I + SL_G - 1) / SL_G, i0 = min(grp * per, I), i1 = min(i0 + per, I);
  float acc[SPG];
#pragma unroll
  for (int s = 0; s < SPG; ++s) acc[s] = 0.f;
  if (live) {
    // 32 weight loads in flight per thread (the walk is a chain of L2 round trips, ~4x longer beside the weight-gradient stream
    // than alone); the 64 sample groups that read one output slice start at DIFFERENT rows of their slice of the contraction, so
    // that they do not all ask L2 for the same line at the same time.  The order of a thread's adds is fixed by (workgroup,
    // thread) alone: deterministic, as before.
    const int n = i1 - i0, rot = n > 0 ? (int)((blockIdx.x * 37u) % (unsigned)n) : 0;
    int done = 0;
    for (; done + 32 <= n; done += 32) {
      float w[32];
      int ii[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        int r = rot + done + u;
        if (r >= n) r -= n;
        ii[u] = i0 + r;
        w[u] = Wt[(size_t)ii[u] * O + o];
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const float4 x0 = *reinterpret_cast<const float4*>(in + ii[u] * SPG);
        acc[0] = fmaf(x0.x, w[u], acc[0]); acc[1] = fmaf(x0.y, w[u], acc[1]);
        acc[2] = fmaf(x0.z, w[u], acc[2]); acc[3] = fmaf(x0.w, w[u], acc[3]);
      }
    }
    for (; done + 8 <= n; done += 8) {         // (the hidden-unit contractions: 24 ... 240 inputs over eight waves)
      float w[8];
      int ii[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int r = rot + done + u;
        if (r >= n) r -= n;
        ii[u] = i0 + r;
        w[u] = Wt[(size_t)ii[u] * O + o];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 x0 = *reinterpret_cast<const float4*>(in + ii[u] * SPG);
        acc[0] = fmaf(x0.x, w[u], acc[0]); acc[1] = fmaf(x0.y, w[u], acc[1]);
        acc[2] = fmaf(x0.z, w[u], acc[2]); acc[3] = fmaf(x0.w, w[u], acc[3]);
      }
    }
    for (; done < n; ++done) {
      int r = rot + done;
      if (r >= n) r -= n;
      const float w = Wt[(size_t)(i0 + r) * O + o];
#pragma unroll
      for (int s = 0; s < SPG; ++s) acc[s] = fmaf(in[(i0 + r) * SPG + s], w, acc[s]);
    }
  }
  if (grp > 0) {
#pragma unroll
    for (int s = 0; s < SPG; ++s) red[(size_t)t * SPG + s] = acc[s];
  }
  __syncthreads();
  if (grp != 0 || !live) return;
#pragma unroll
  for (int g = 1; g < SL_G; ++g)
#pragma unroll
    for (int s = 0; s < SPG; ++s) acc[s] += red[(size_t)(g * SL_OB + ol) * SPG + s];
  if (MODE == 0) {
    const float b1 = a.b1[o];
#pragma unroll
    for (int s = 0; s < SPG; ++s)
      if (b0 + s < a.B) a.h[(size_t)(b0 + s) * a.R + o] = fmaxf(acc[s] + b1, 0.f);
  } else if (MODE == 1) {
    const float b2 = a.b2[o];
#pragma unroll
    for (int s = 0; s < SPG; ++s) {
      if (b0 + s < a.B) {
        const float q = acc[s] + b2;
        a.q[(size_t)(b0 + s) * a.C + o] = q;
        a.s[(size_t)(b0 + s) * a.C + o] = hsigmoid(q);
      }
    }
  } else if (MODE == 2) {
#pragma unroll
    for (int s = 0; s < SPG; ++s)
      if (b0 + s < a.B) a.dp[(size_t)(b0 + s) * a.R + o] = a.h[(size_t)(b0 + s) * a.R + o] > 0.f ? acc[s] : 0.f;
  } else {
    float v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int s = 0; s < SPG; ++s) {
      if (b0 + s < a.B) {
        const size_t k = (size_t)(b0 + s) * a.C + o;
        const float gu = acc[s] * inv;                          // every pixel of u receives dL/dm / HW
        a.g[k] = gu;
        const float sg = a.s[k], p1 = a.ps[2 * k], p2 = a.ps[2 * k + 1];
        v1 += sg * p1 + (float)a.HW * gu;
        v2 += sg * p2 + gu * t3d_pool_get(a.gap, k, a.gapq);
      }
    }
    if (a.stats) {      // one add per (sample group, channel), spread over the reduction replicas
      double* st = a.stats + (size_t)(blockIdx.x % nrep) * rstride;
      atomicAdd(st + o, (double)v1);
      atomicAdd(st + a.C + o, (double)v2);
    }
  }
}

// (Round 6b, measured and removed: the same products on v_mfma_f32_16x16x4_f32 -- a workgroup per 16 samples x 64 outputs, the 16 input
// rows staged in LDS, exact fp32 products: correct (the suite passed on it) and 0.3 ms SLOWER per MobileNetV3-large step than the
// slices (7.93 against 7.63 ms, two same-box pairs): staging 16 x 960 transformed inputs per workgroup is a 60-round chain of global
// loads where the slice kernel's 4 x 960 over 512 threads is eight, and 64 workgroups do not fill the chip.)
template <int MODE>
static void se_slice_launch(const SeArgs& a, const float* Wt, hipStream_t st) {
  const int I = (MODE == 0 || MODE == 2) ? a.C : a.R, O = (MODE == 0 || MODE == 2) ? a.R : a.C;
  const size_t lds = ((size_t)I * SPG + (size_t)SL_T * SPG) * sizeof(float);
  T3D_LAUNCH(se_slice_kernel<MODE>, dim3(cdiv(a.B, SPG), cdiv(O, SL_OB)), dim3(SL_T), lds, st, a, Wt, g_t3d_reduce.nrep < 1 ? 1 : g_t3d_reduce.nrep,
             g_t3d_reduce.nrep < 1 ? 0 : g_t3d_reduce.stats_stride);
}
static bool se_sliced() { return !T3D_ENV_SET("T3D_SE_GROUP"); }      // (A/B: the one-launch group kernels)

}  // namespace

extern "C" int t3d_se_fwd(const float* gap_sum, const float* scale, const float* shift, const float* w1,
                          const float* b1, const float* w2, const float* b2, float* m, float* h, float* q, float* s,
                          int B, int C, int R, int HW, void* stream) {
  if (!gap_sum || !scale || !shift || !w1 || !b1 || !w2 || !b2 || !m || !h || !q || !s || B <= 0 || C <= 0 || R <= 0 ||
      HW <= 0)
    return T3D_ERR_ARG;
  SeArgs a{};
  a.gap = gap_sum; a.gapq = g_t3d_reduce.pool_exact; a.scale = scale; a.shift = shift; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
  a.m = m; a.h = h; a.q = q; a.s = s; a.B = B; a.C = C; a.R = R; a.HW = HW;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(h, 0, (size_t)B * R * sizeof(float), st) != hipSuccess) return T3D_ERR_LAUNCH;
  T3D_LAUNCH(se_fc1_kernel, dim3(cdiv(R, OT), cdiv(B, 64), KSPLIT), dim3(256), 0, st, a);
  T3D_LAUNCH(se_relu_bias_kernel, dim3(cdiv(B * R, 256)), dim3(256), 0, st, a);
  T3D_LAUNCH(se_fc2_kernel, dim3(cdiv(C, OT), cdiv(B, 64)), dim3(256), 0, st, a);
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
  a.ps = ps_stats; a.gap = gap_sum; a.gapq = g_t3d_reduce.pool_exact; a.scale = scale; a.shift = shift; a.w1 = w1; a.w2 = w2;
  a.m = const_cast<float*>(m); a.h = const_cast<float*>(h); a.q = const_cast<float*>(q); a.s = const_cast<float*>(s);
  a.g = g; a.dq = dq; a.dp = dp; a.stats = stats; a.dw1 = dw1; a.db1 = db1; a.dw2 = dw2; a.db2 = db2;
  a.B = B; a.C = C; a.R = R; a.HW = HW;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  T3D_LAUNCH(se_dq_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, st, a);
  if (hipMemsetAsync(dp, 0, (size_t)B * R * sizeof(float), st) != hipSuccess) return T3D_ERR_LAUNCH;
  T3D_LAUNCH(se_dh_kernel, dim3(cdiv(R, OT), cdiv(B, 64), KSPLIT), dim3(256), 0, st, a);
  T3D_LAUNCH(se_relu_mask_kernel, dim3(cdiv(B * R, 256)), dim3(256), 0, st, a);
  T3D_LAUNCH(se_dm_kernel, dim3(cdiv(C, OT), cdiv(B, 64)), dim3(256), 0, st, a);
  T3D_LAUNCH(se_wgrad_tile_kernel, dim3(cdiv(R, OT), cdiv(C, 64)), dim3(256), 0, st, a, 0);
  T3D_LAUNCH(se_wgrad_tile_kernel, dim3(cdiv(C, OT), cdiv(R, 64)), dim3(256), 0, st, a, 1);
  T3D_LAUNCH(se_bias_kernel, dim3(cdiv(C + R, 64)), dim3(256), 0, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

static size_t se_group_lds(int C, int R) { return ((size_t)(C + R) * SPG + (size_t)SE_T * SPG) * sizeof(float); }

// One launch per direction (kernels above).  w1t [C][R], w2t [R][C]: transposed fp32 copies of fc.0 / fc.2 (the caller
// keeps them current, e.g. t3d_pack_weights_batched).  C, R <= 1024.
extern "C" int t3d_se_fwd_fused(const float* gap_sum, const float* scale, const float* shift, const float* w1t,
                                const float* b1, const float* w2t, const float* b2, float* m, float* h, float* q, float* s,
                                int B, int C, int R, int HW, void* stream) {
  if (!gap_sum || !scale || !shift || !w1t || !b1 || !w2t || !b2 || !m || !h || !q || !s || B <= 0 || C <= 0 || R <= 0 ||
      HW <= 0)
    return T3D_ERR_ARG;
  SeArgs a{};
  a.gap = gap_sum; a.gapq = g_t3d_reduce.pool_exact; a.scale = scale; a.shift = shift; a.b1 = b1; a.b2 = b2;
  a.m = m; a.h = h; a.q = q; a.s = s; a.B = B; a.C = C; a.R = R; a.HW = HW;
  if (se_sliced()) {
    se_slice_launch<0>(a, w1t, reinterpret_cast<hipStream_t>(stream));
    se_slice_launch<1>(a, w2t, reinterpret_cast<hipStream_t>(stream));
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const size_t lds = se_group_lds(C, R);
  (void)t3d_max_lds((const void*)se_fwd_group_kernel, 96 * 1024);      // (cached per device and kernel, misc.hip)
  T3D_LAUNCH(se_fwd_group_kernel, dim3(cdiv(B, SPG)), dim3(SE_T), lds, reinterpret_cast<hipStream_t>(stream), a, w1t, w2t);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_se_bwd_data(const float* ps_stats, const float* gap_sum, const float* scale, const float* shift,
                               const float* w1, const float* w2, const float* h, const float* q, const float* s, float* g,
                               float* dq, float* dp, double* stats, int B, int C, int R, int HW, void* stream) {
  if (!ps_stats || !gap_sum || !scale || !shift || !w1 || !w2 || !h || !q || !s || !g || !dq || !dp || B <= 0 || C <= 0 ||
      R <= 0 || HW <= 0)
    return T3D_ERR_ARG;
  SeArgs a{};
  a.ps = ps_stats; a.gap = gap_sum; a.gapq = g_t3d_reduce.pool_exact; a.scale = scale; a.shift = shift; a.w1 = w1; a.w2 = w2;
  a.h = const_cast<float*>(h); a.q = const_cast<float*>(q); a.s = const_cast<float*>(s);
  a.g = g; a.dq = dq; a.dp = dp; a.stats = stats; a.B = B; a.C = C; a.R = R; a.HW = HW;
  if (se_sliced()) {
    se_slice_launch<2>(a, w2, reinterpret_cast<hipStream_t>(stream));      // W2 is [C][R] = [I][O]
    se_slice_launch<3>(a, w1, reinterpret_cast<hipStream_t>(stream));      // W1 is [R][C] = [I][O]
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  (void)t3d_max_lds((const void*)se_bwd_group_kernel, 96 * 1024);
  T3D_LAUNCH(se_bwd_group_kernel, dim3(cdiv(B, SPG)), dim3(SE_T), se_group_lds(C, R), reinterpret_cast<hipStream_t>(stream), a,
                     g_t3d_reduce.nrep, g_t3d_reduce.stats_stride);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// weight / bias gradients of the two FCs from the dq, dp that t3d_se_bwd_data left (leaves of the backward graph: the
// host side issues them on its weight-gradient stream)
extern "C" int t3d_se_bwd_weights(const float* m, const float* h, const float* dq, const float* dp, float* dw1, float* db1,
                                  float* dw2, float* db2, int B, int C, int R, void* stream) {
  if (!m || !h || !dq || !dp || !dw1 || !db1 || !dw2 || !db2 || B <= 0 || C <= 0 || R <= 0) return T3D_ERR_ARG;
  SeArgs a{};
  a.m = const_cast<float*>(m); a.h = const_cast<float*>(h); a.dq = const_cast<float*>(dq); a.dp = const_cast<float*>(dp);
  a.dw1 = dw1; a.db1 = db1; a.dw2 = dw2; a.db2 = db2; a.B = B; a.C = C; a.R = R; a.HW = 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  T3D_LAUNCH(se_wgrad_tile_kernel, dim3(cdiv(R, OT), cdiv(C, 64)), dim3(256), 0, st, a, 0);
  T3D_LAUNCH(se_wgrad_tile_kernel, dim3(cdiv(C, OT), cdiv(R, 64)), dim3(256), 0, st, a, 1);
  T3D_LAUNCH(se_bias_kernel, dim3(cdiv(C + R, 64)), dim3(256), 0, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
