// Per-class 9-keypoint regression heads + class head (fp32; ModelWrapper.forward,
// torchdet3d/builders/model_builder.py:126-146) as wavefront reductions.
//
// The reference runs a Python loop of B GEMVs (`regressors[c](sample)`, :137).  Here one
// workgroup per sample stages the (activated) feature vector in LDS once, and each wave
// produces rows of the class-selected 18 x F matrix (and of the class head) as a 64-lane dot
// product + wave reduction.  The head weights (9*18*F + ncls*F floats, <1 MB) stay L2-resident.
// Backward: data gradient per sample (thread = feature column, coalesced weight reads);
// weight gradient as a deterministic segmented reduction over the batch (thread = feature
// column, block = (class, column chunk)), no atomics.
#include "common.h"

namespace {

constexpr int NKP = 18;  // 9 keypoints x (x, y)

struct HeadArgs {
  const float* f;
  const float *scale, *shift;
  int act;
  const int64_t* cats;
  const float *wreg, *breg, *wcls, *bcls, *mask;
  float *kp, *logits;
  const float *kpin, *dkp, *dlogits;
  float *dpre, *df;
  double* stats;
  float *dwreg, *dbreg, *dwcls, *dbcls;
  int B, F, ncls;
  int all_heads;   // export mode (forward_to_onnx): every one of the 9 regressors for every sample, kp [9,B,18]
};

__device__ __forceinline__ float feat(const HeadArgs& a, int b, int j) {
  const float v = a.f[(size_t)b * a.F + j];
  return a.scale ? act_apply(v * a.scale[j] + a.shift[j], a.act) : v;
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadArgs a) {
  extern __shared__ float fs[];  // [F] activated features, [F] masked features
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* fm = fs + a.F;
  for (int j = tid; j < a.F; j += 256) {
    const float v = feat(a, b, j);
    fs[j] = v;
    fm[j] = a.mask ? v * a.mask[(size_t)b * a.F + j] : v;
  }
  __syncthreads();
  int cb = a.all_heads ? 0 : (int)a.cats[b];
  cb = cb < 0 ? 0 : (cb > 8 ? 8 : cb);
  const int nreg = a.all_heads ? 9 * NKP : NKP;
  const int nrows = nreg + (a.logits ? a.ncls : 0);
  for (int r = wave; r < nrows; r += 4) {
    const bool reg = r < nreg;
    const int c = a.all_heads ? r / NKP : cb, rr = a.all_heads ? r % NKP : r;      // (class, row) of a regressor row
    const float* w = reg ? a.wreg + ((size_t)c * NKP + rr) * a.F : a.wcls + (size_t)(r - nreg) * a.F;
    const float* x = reg ? fs : fm;
    // eight weight loads in flight per lane (the plain loop issued one load per iteration: 20 dependent round trips per
    // row, 7 rows per wave -- 53 us for 9 MFLOP at the end of the forward, where nothing else runs)
    float s = 0.f;
    int j = lane;
    for (; j + 7 * 64 < a.F; j += 8 * 64) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = w[j + 64 * u];
#pragma unroll
      for (int u = 0; u < 8; ++u) s = fmaf(wv[u], x[j + 64 * u], s);
    }
    for (; j < a.F; j += 64) s = fmaf(w[j], x[j], s);
    s = wave_sum(s);
    if (lane == 0) {
      if (reg) {
        s += a.breg[c * NKP + rr];
        const size_t o = a.all_heads ? ((size_t)c * a.B + b) * NKP + rr : (size_t)b * NKP + rr;
        a.kp[o] = 1.f / (1.f + expf(-s));
      } else {
        a.logits[(size_t)b * a.ncls + (r - nreg)] = s + a.bcls[r - nreg];
      }
    }
  }
}

// data gradient: one block per sample
__global__ __launch_bounds__(256) void head_bwd_data_kernel(const HeadArgs a) {
  __shared__ float dp[NKP];
  __shared__ float dl[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < NKP) {
    const float k = a.kpin[(size_t)b * NKP + tid];
    const float v = a.dkp[(size_t)b * NKP + tid] * k * (1.f - k);  // sigmoid'
    dp[tid] = v;
    a.dpre[(size_t)b * NKP + tid] = v;
  }
  if (tid >= 64 && tid < 64 + a.ncls && a.dlogits) dl[tid - 64] = a.dlogits[(size_t)b * a.ncls + tid - 64];
  __syncthreads();
  int c = (int)a.cats[b];
  c = c < 0 ? 0 : (c > 8 ? 8 : c);
  const float* wr = a.wreg + (size_t)c * NKP * a.F;
  for (int j = tid; j < a.F; j += 256) {
    float g = 0.f;
#pragma unroll
    for (int r = 0; r < NKP; ++r) g = fmaf(dp[r], wr[(size_t)r * a.F + j], g);
    if (a.dlogits) {
      float gc = 0.f;
      for (int q = 0; q < a.ncls; ++q) gc = fmaf(dl[q], a.wcls[(size_t)q * a.F + j], gc);
      g += a.mask ? gc * a.mask[(size_t)b * a.F + j] : gc;
    }
    if (a.scale) {
      const float raw = a.f[(size_t)b * a.F + j];
      g *= act_grad(raw * a.scale[j] + a.shift[j], a.act);  // gradient at the BatchNorm1d output
      if (a.stats) {
        atomicAdd(a.stats + j, (double)g);
        atomicAdd(a.stats + a.F + j, (double)g * (double)raw);
      }
    }
    a.df[(size_t)b * a.F + j] = g;
  }
}

// weight gradient: grid (10, ceil(F/256), S); blockIdx.x < 9: regressor of that class, == 9: class head; blockIdx.z
// takes a contiguous slice of the batch and ADDS its part (the caller's gradient buffer is zeroed once per step).
// (One workgroup per (class, 256 features) walks the whole batch: the 8-way batch split of rounds 1-3a added its partial sums
// with fp32 atomics in arrival order -- the last place where two runs of one training step could differ.)
// The samples a block needs are first compacted (in ascending order: the sum stays deterministic) into LDS, then
// consumed eight at a time so that eight feature loads are in flight per thread (the serial one-load-per-sample loop
// took ~1 us per sample).
__global__ __launch_bounds__(256) void head_bwd_weight_kernel(const HeadArgs a) {
  constexpr int CHUNK = 512, U = 8;
  __shared__ int lst[CHUNK];
  __shared__ int nsel, wcnt[4];
  __shared__ __attribute__((aligned(16))) float ldp[CHUNK * NKP];      // staged gradient rows (class blocks) / logit gradients (classifier block)
  const int cls = blockIdx.x, j = blockIdx.y * 256 + threadIdx.x;
  const bool jon = j < a.F;
  if (cls < 9) {
    float acc[NKP], bacc = 0.f;
#pragma unroll
    for (int r = 0; r < NKP; ++r) acc[r] = 0.f;
    const bool bias_thread = (blockIdx.y == 0 && threadIdx.x < NKP);
    const int per = (a.B + gridDim.z - 1) / gridDim.z, zb0 = blockIdx.z * per, zb1 = min(a.B, zb0 + per);
    for (int b0 = zb0; b0 < zb1; b0 += CHUNK) {
      __syncthreads();
      // compaction in ascending sample order by the whole workgroup (ballot + prefix counts; one thread walking the chunk
      // was 256 dependent-latency loads: 100 us of this launch)
      {
        const int b1 = min(zb1, b0 + CHUNK);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int n = 0;
        for (int s0 = b0; s0 < b1; s0 += 256) {
          const int b = s0 + threadIdx.x;
          bool hit = false;
          if (b < b1) {
            int c = (int)a.cats[b];
            c = c < 0 ? 0 : (c > 8 ? 8 : c);
            hit = c == cls;
          }
          const unsigned long long m = __ballot(hit);
          if (lane == 0) wcnt[wv] = __popcll(m);
          __syncthreads();
          int off = n;
          for (int w2 = 0; w2 < wv; ++w2) off += wcnt[w2];
          if (hit) lst[off + __popcll(m & ((1ull << lane) - 1ull))] = b;
          n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
          __syncthreads();
        }
        if (threadIdx.x == 0) nsel = n;
      }
      __syncthreads();
      const int n = nsel;
      // gradient rows of the selected samples -> LDS (coalesced), read back as broadcasts (as global loads they were 18
      // same-address vector loads per sample and thread: most of this launch's 180 us)
      for (int i = threadIdx.x; i < n * NKP; i += 256) ldp[i] = a.dpre[(size_t)lst[i / NKP] * NKP + i % NKP];
      __syncthreads();
      for (int i0 = 0; i0 < n; i0 += U) {
        float x[U];
        int bb[U];
        const int jc = min(j, a.F - 1);
#pragma unroll
        for (int u = 0; u < U; ++u) {       // all eight loads first, no control flow in between
          bb[u] = lst[min(i0 + u, n - 1)];
          x[u] = a.f[(size_t)bb[u] * a.F + jc];
        }
        if (a.scale) {
          const float sc = a.scale[jc], sh = a.shift[jc];
#pragma unroll
          for (int u = 0; u < U; ++u) x[u] = act_apply(x[u] * sc + sh, a.act);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = (jon && i0 + u < n) ? x[u] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (i0 + u < n) {       // block-uniform
#pragma unroll
            for (int r = 0; r < NKP; ++r) acc[r] = fmaf(ldp[(i0 + u) * NKP + r], x[u], acc[r]);
            if (bias_thread) bacc += ldp[(i0 + u) * NKP + threadIdx.x];
          }
        }
      }
    }
    if (jon) {
#pragma unroll
      for (int r = 0; r < NKP; ++r) unsafeAtomicAdd(a.dwreg + ((size_t)cls * NKP + r) * a.F + j, acc[r]);
    }
    if (bias_thread) unsafeAtomicAdd(a.dbreg + cls * NKP + threadIdx.x, bacc);
  } else if (a.dlogits && a.dwcls) {
    for (int q0 = 0; q0 < a.ncls; q0 += 16) {
      float acc[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const int per = (a.B + gridDim.z - 1) / gridDim.z, zb0 = blockIdx.z * per, zb1 = min(a.B, zb0 + per);
      constexpr int SB = 128;                  // samples whose logit gradients are staged in LDS at a time ([SB][16])
      for (int s0 = zb0; s0 < zb1; s0 += SB) {
        const int s1 = min(zb1, s0 + SB);
        __syncthreads();
        for (int i = threadIdx.x; i < (s1 - s0) * 16; i += 256) {
          const int b = s0 + (i >> 4), q = q0 + (i & 15);
          ldp[i] = q < a.ncls ? a.dlogits[(size_t)b * a.ncls + q] : 0.f;
        }
        __syncthreads();
        for (int b0 = s0; b0 < s1; b0 += U) {
          float x[U], mk[U];
          const int jc = min(j, a.F - 1);
#pragma unroll
          for (int u = 0; u < U; ++u) x[u] = a.f[(size_t)min(b0 + u, a.B - 1) * a.F + jc];
          if (a.mask) {
#pragma unroll
            for (int u = 0; u < U; ++u) mk[u] = a.mask[(size_t)min(b0 + u, a.B - 1) * a.F + jc];
          } else {
#pragma unroll
            for (int u = 0; u < U; ++u) mk[u] = 1.f;
          }
          if (a.scale) {
            const float sc = a.scale[jc], sh = a.shift[jc];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = act_apply(x[u] * sc + sh, a.act);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) x[u] = (jon && b0 + u < s1) ? x[u] * mk[u] : 0.f;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (b0 + u < s1) {
              const float4* dl = reinterpret_cast<const float4*>(ldp + (b0 + u - s0) * 16);
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const float4 d = dl[q4];
                acc[4 * q4] = fmaf(d.x, x[u], acc[4 * q4]);
                acc[4 * q4 + 1] = fmaf(d.y, x[u], acc[4 * q4 + 1]);
                acc[4 * q4 + 2] = fmaf(d.z, x[u], acc[4 * q4 + 2]);
                acc[4 * q4 + 3] = fmaf(d.w, x[u], acc[4 * q4 + 3]);
              }
            }
          }
        }
      }
      if (jon) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (q0 + q < a.ncls) unsafeAtomicAdd(a.dwcls + (size_t)(q0 + q) * a.F + j, acc[q]);
      }
    }
    if (blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < a.ncls) {
      float s = 0.f;
      for (int b = 0; b < a.B; ++b) s += a.dlogits[(size_t)b * a.ncls + threadIdx.x];
      unsafeAtomicAdd(a.dbcls + threadIdx.x, s);
    }
  }
}

// y[m][n] = sum_k x[m][k] w[n][k] + bias[n]: one wave per output element (any N, K; the heads' N = 18 / num_classes
// are not multiples of 8, which the MFMA 1x1 kernels require)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int M,
                                                         int K, int Nn) {
  const int lane = threadIdx.x & 63;
  const long long total = (long long)M * Nn;
  for (long long o = blockIdx.x * 4ll + (threadIdx.x >> 6); o < total; o += gridDim.x * 4ll) {
    const int m = (int)(o / Nn), n = (int)(o % Nn);
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(x[(size_t)m * K + k], w[(size_t)n * K + k], s);
    s = wave_sum(s);
    if (lane == 0) y[o] = s + (bias ? bias[n] : 0.f);
  }
}

inline void fill(HeadArgs& a, const t3d_prologue* pro) {
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
}

}  // namespace

extern "C" int t3d_head_fwd(const float* f, const t3d_prologue* pro, const int64_t* cats, const float* wreg,
                            const float* breg, const float* wcls, const float* bcls, const float* mask, float* kp,
                            float* logits, int B, int F, int ncls, void* stream) {
  if (!f || !cats || !wreg || !breg || !kp || B <= 0 || F <= 0) return T3D_ERR_ARG;
  if (logits && (!wcls || !bcls || ncls <= 0 || ncls > 64)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  HeadArgs a{};
  a.f = f; a.cats = cats; a.wreg = wreg; a.breg = breg; a.wcls = wcls; a.bcls = bcls; a.mask = mask;
  a.kp = kp; a.logits = logits; a.B = B; a.F = F; a.ncls = ncls;
  fill(a, pro);
  T3D_LAUNCH(head_fwd_kernel, dim3(B), dim3(256), (size_t)2 * F * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                              void* stream) {
  if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return T3D_ERR_ARG;
  const long long total = (long long)M * N;
  const int grid = (int)((total + 3) / 4 < 4096 ? (total + 3) / 4 : 4096);
  T3D_LAUNCH(linear_fwd_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, w, bias, y,
                     M, K, N);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_head_fwd_all(const float* f, const t3d_prologue* pro, const float* wreg, const float* breg,
                                const float* wcls, const float* bcls, float* kp_all, float* logits, int B, int F,
                                int ncls, void* stream) {
  if (!f || !wreg || !breg || !kp_all || B <= 0 || F <= 0) return T3D_ERR_ARG;
  if (logits && (!wcls || !bcls || ncls <= 0 || ncls > 64)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  HeadArgs a{};
  a.f = f; a.wreg = wreg; a.breg = breg; a.wcls = wcls; a.bcls = bcls;
  a.kp = kp_all; a.logits = logits; a.B = B; a.F = F; a.ncls = ncls; a.all_heads = 1;
  fill(a, pro);
  T3D_LAUNCH(head_fwd_kernel, dim3(B), dim3(256), (size_t)2 * F * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_head_bwd(const float* f, const t3d_prologue* pro, const int64_t* cats, const float* wreg,
                            const float* wcls, const float* mask, const float* kp, const float* dkp,
                            const float* dlogits, float* dpre, float* df, double* stats, float* dwreg,
                            float* dbreg, float* dwcls, float* dbcls, int B, int F, int ncls, void* stream) {
  if (!f || !cats || !wreg || !kp || !dkp || !dpre || !df || (dwreg && !dbreg) || B <= 0 || F <= 0)
    return T3D_ERR_ARG;
  if (dlogits && (!wcls || (dwreg && (!dwcls || !dbcls)) || ncls <= 0 || ncls > 64)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  HeadArgs a{};
  a.f = f; a.cats = cats; a.wreg = wreg; a.wcls = wcls; a.mask = mask; a.kpin = kp; a.dkp = dkp;
  a.dlogits = dlogits; a.dpre = dpre; a.df = df; a.stats = stats; a.dwreg = dwreg; a.dbreg = dbreg;
  a.dwcls = dwcls; a.dbcls = dbcls; a.B = B; a.F = F; a.ncls = ncls;
  fill(a, pro);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  T3D_LAUNCH(head_bwd_data_kernel, dim3(B), dim3(256), 0, st, a);
  if (dwreg)        // NULL: data gradient only, the weight gradients follow through t3d_head_bwd_weights
    T3D_LAUNCH(head_bwd_weight_kernel, dim3(10, cdiv(F, 256), 1), dim3(256), 0, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// The heads' weight / bias gradients on their own (after t3d_head_bwd(..., dwreg = NULL, ...) left dpre): leaves of the
// backward graph, so the host side issues them on its weight-gradient stream.
extern "C" int t3d_head_bwd_weights(const float* f, const t3d_prologue* pro, const int64_t* cats, const float* mask,
                                    const float* dpre, const float* dlogits, float* dwreg, float* dbreg, float* dwcls,
                                    float* dbcls, int B, int F, int ncls, void* stream) {
  if (!f || !cats || !dpre || !dwreg || !dbreg || B <= 0 || F <= 0) return T3D_ERR_ARG;
  if (dlogits && (!dwcls || !dbcls || ncls <= 0 || ncls > 64)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;
  HeadArgs a{};
  a.f = f; a.cats = cats; a.mask = mask; a.dlogits = dlogits; a.dpre = const_cast<float*>(dpre);
  a.dwreg = dwreg; a.dbreg = dbreg; a.dwcls = dwcls; a.dbcls = dbcls; a.B = B; a.F = F; a.ncls = ncls;
  fill(a, pro);
  T3D_LAUNCH(head_bwd_weight_kernel, dim3(10, cdiv(F, 256), 1), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
