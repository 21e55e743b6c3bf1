// "y-free" backward of a pointwise (1x1) convolution whose input is a finished (materialised) tensor -- the expand
// layer of an inverted-residual block.  bf16 storage only.
//
// The BatchNorm backward makes the gradient at the conv output an affine function of the stored gradient dz and of
// the conv output y itself:  dy = alpha*dz + beta*y + gamma  (per output channel n).  Both consumers normally read
// the WIDE tensors dz and y (M x N).  But y = x W^T with x the NARROW block input (M x K, N = 6K), so
//
//   data gradient    dx = dy W = dz (alpha.W) + x (W^T diag(beta) W) + gamma^T W
//                       = [dz | x] . Wcat^T + c          -- one plain GEMM over two raw bf16 tensors
//   weight gradient  dW = dy^T x = alpha.(dz^T x) + beta.(W (x^T x)) + gamma (1^T x)
//                       -- dz^T x, the K x K Gram matrix x^T x and the column sums of x come out of ONE raw GEMM
//                          [dz | x | 1]^T x; a tiny kernel combines them.
//
// Neither kernel reads y, and neither applies a per-element transform: the expand layer's backward drops from
// 4 to 2 passes over M x N elements (SURVEY.md section 8d counts 4: the figures reported against that budget are
// therefore "fused" fractions).  Reference semantics: autograd of nn.Conv2d(K, N, 1) + nn.BatchNorm2d(N)
// (models/mobilenetv3.py:148-149) -- same sums, reassociated.
#include "pwconv_common.h"

int t3d_pw_wgrad_tr_yfree(const void* dz, const void* x, float* tmp, int M, int HW, int K, int N, hipStream_t st);
// fused y-free backward (pwconv_wgrad_tr.hip)
size_t t3d_pw_bwd_yfree_scratch(int M, int K, int N);
int t3d_pw_bwd_yfree_launch(const void* dz, const void* x, const void* wd, const void* wt, const t3d_bnbwd* bb, const void* x_raw,
                            const void* residual, void* dx, double* stats, void* scratch, int M, int HW, int K, int N, hipStream_t st);
int t3d_pw_bwd_yfree_reduce(void* scratch, float** tmp_out, int M, int K, int N, hipStream_t st);

namespace {

// Wcat [K][NP + KP] (bf16) = [ alpha_n W[n][k]  (n < N, zero padded to NP) | Q[k][k2] = sum_n W[n][k] beta_n W[n][k2]  (zero
// padded to KP) ],  c[k] = sum_n gamma_n W[n][k].  The launch sits on the critical path between the depthwise backward (whose
// sums give beta, gamma) and the expand layer's data gradient, 13 times per step -- round 3's version (one workgroup per
// (k, 16 columns): up to 576 workgroups, EACH deriving all N BatchNorm coefficients from the replica sums, scalar 2-byte
// weight loads) took 14 us per launch in the step.  Now: one workgroup per 16 x 16 tile of Q on the matrix cores, from the
// TRANSPOSED weight copy wt [K][N] (rows contiguous in n, so both MFMA operands are 16-byte row reads): the four waves split
// the contraction over n, meet in LDS; (K/16)^2 workgroups derive the coefficients instead of K * KP/16.
__global__ __launch_bounds__(256) void yfree_prep_kernel(const bf16_t* __restrict__ wt, const float* __restrict__ alpha_g,
                                                         const float* __restrict__ beta_g, const float* __restrict__ gamma_g,
                                                         bf16_t* __restrict__ wcat, float* __restrict__ cvec, int K, int N,
                                                         int NP, int KP, const T3dFold* __restrict__ fold,
                                                         bf16_t* __restrict__ wd, int PBw) {
  // wd (optional, for the fused backward t3d_pwconv_bwd_yfree): the same numbers in the column order of the staged
  // [dz | x | 1] rows -- row k = [alpha_n W[n][k] (N) | Q[k][.] (K) | c[k] | zeros to PBw]; the zero parts are never written
  // (the caller clears the buffer once)
  extern __shared__ float fco[];      // [3][N] (alpha, beta, gamma): derived here when a finalize request rides on this launch
  __shared__ f32x4 red[3][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lg = lane >> 4, lc = lane & 15;
  if (fold) {
    t3d_fold_block(fold, 0, N, fco, N, blockIdx.x == 0 && blockIdx.y == 0);     // (ends with a barrier)
  } else {
    for (int i = t; i < N; i += 256) { fco[i] = alpha_g[i]; fco[N + i] = beta_g[i]; fco[2 * N + i] = gamma_g[i]; }
    __syncthreads();
  }
  const float *alpha = fco, *beta = fco + N, *gamma = fco + 2 * N;
  const int tot = NP + KP;
  const int k0 = blockIdx.x * 16, q0 = blockIdx.y * 16;        // tile rows k0.. (data-gradient output channels), columns q0..
  // ---- Q tile: D[m = k][n = k2] = sum_n A[m][n] B[n][k2],  A = wt rows, B = beta * wt rows (bf16 operands, fp32 accumulate)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nsteps = (N + 31) / 32;
  const int ka = min(k0 + lc, K - 1), kb = min(q0 + lc, K - 1);
  const bf16_t* ra = wt + (size_t)ka * N;
  const bf16_t* rb = wt + (size_t)kb * N;
  constexpr int U = 4;                  // k-steps in flight per wave
  for (int s0 = wave; s0 < nsteps; s0 += 4 * U) {
    bf16x8 fa[U], fb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = min((s0 + 4 * u) * 32 + 8 * lg, N - 8);      // (steps past N re-read a valid address; their beta is zeroed)
      fa[u] = *reinterpret_cast<const bf16x8*>(ra + n);
      fb[u] = *reinterpret_cast<const bf16x8*>(rb + n);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int st = s0 + 4 * u, n = st * 32 + 8 * lg;
      const bool live = st < nsteps && n < N;                     // (N % 8 == 0: whole 8-groups are in or out)
      bf16x8 b;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = (bf16_t)(live ? beta[min(n, N - 8) + j] * (float)fb[u][j] : 0.f);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[u], b, acc, 0, 0, 0);
    }
  }
  if (wave) red[wave - 1][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    acc += red[0][lane];
    acc += red[1][lane];
    acc += red[2][lane];
    // D[row = 4*lg + r -> k][col = lc -> k2]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + 4 * lg + r, k2 = q0 + lc;
      if (k < K) wcat[(size_t)k * tot + NP + k2] = (bf16_t)((k2 < K) ? acc[r] : 0.f);     // k2 in [K, KP): zero padding
      if (wd && k < K && k2 < K) wd[(size_t)k * PBw + N + k2] = (bf16_t)acc[r];
    }
  }
  // ---- alpha . W part and c: the workgroups of tile column 0 own rows k0 .. k0+15 (row k = wt row k scaled by alpha)
  if (blockIdx.y == 0) {
    for (int i = t; i < 16 * (NP / 8); i += 256) {
      const int k = k0 + i / (NP / 8), n = (i % (NP / 8)) * 8;
      if (k >= K) continue;
      bf16x8 o;
      if (n < N) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(wt + (size_t)k * N + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(alpha[n + j] * (float)v[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
      }
      *reinterpret_cast<bf16x8*>(wcat + (size_t)k * tot + n) = o;
      if (wd && n < N) *reinterpret_cast<bf16x8*>(wd + (size_t)k * PBw + n) = o;
    }
    // c[k] = sum_n gamma_n W[n][k]: 16 lanes per row
    const int row = t >> 4, sub = t & 15, k = k0 + row;
    float s2 = 0.f;
    if (k < K)
      for (int n = sub * 8; n < N; n += 128) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(wt + (size_t)k * N + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) s2 = fmaf(gamma[n + j], (float)v[j], s2);
      }
    s2 = row16_sum(s2);
    if (sub == 0 && k < K) {
      cvec[k] = s2;
      if (wd) wd[(size_t)k * PBw + N + K] = (bf16_t)s2;
    }
  }
}

// dw[n][k] += alpha_n P[n][k] + beta_n sum_k2 W[n][k2] G[k2][k] + gamma_n s[k];   tmp = [P (N rows) | G (K rows) | s]
__global__ __launch_bounds__(256) void yfree_combine_kernel(const float* __restrict__ tmp, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta,
                                                            const float* __restrict__ gamma, float* __restrict__ dw, int K,
                                                            int N) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * K) return;
  const int n = e / K, k = e % K;
  const float* G = tmp + (size_t)N * K;
  // (8 weights per 16-byte load and 8 independent Gram loads per round: the one-load-per-multiply loop was K dependent
  //  round trips, 24 us for K = 96)
  float wg = 0.f;
  for (int k2 = 0; k2 < K; k2 += 8) {
    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + (size_t)n * K + k2);
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = G[(size_t)(k2 + j) * K + k];
#pragma unroll
    for (int j = 0; j < 8; ++j) wg = fmaf((float)wv[j], g[j], wg);
  }
  dw[e] += alpha[n] * tmp[e] + beta[n] * wg + gamma[n] * tmp[(size_t)(N + K) * K + k];
}

}  // namespace

static inline int rup32(int v) { return (v + 31) / 32 * 32; }

// include/t3d.h
static int yfree_prep_impl(const void* wt, const t3d_bnbwd* bb, void* wcat, float* cvec, int K, int N, void* wd, int PBw,
                           void* stream);

extern "C" int t3d_pwconv_yfree_prep(const void* wt, const t3d_bnbwd* bb, void* wcat, float* cvec, int K, int N,
                                     void* stream) {
  return yfree_prep_impl(wt, bb, wcat, cvec, K, N, nullptr, 0, stream);
}

// prep + the fused backward's weight layout wd [16 * ceil(K / 16)][PBw] (PBw = 64 * ceil((N + K + 8) / 64), cleared once by
// the caller: only the non-zero entries are written)
extern "C" int t3d_pwconv_yfree_prep2(const void* wt, const t3d_bnbwd* bb, void* wcat, float* cvec, void* wd, int K, int N,
                                      void* stream) {
  if (!wd) return T3D_ERR_ARG;
  return yfree_prep_impl(wt, bb, wcat, cvec, K, N, wd, (N + K + 8 + 63) / 64 * 64, stream);
}

extern "C" int t3d_pwconv_bwd_yfree_scratch(int M, int K, int N) { return (int)t3d_pw_bwd_yfree_scratch(M, K, N); }

// dx and the partial products of the weight gradient in ONE pass over [dz | x] (caller's MAIN stream); the weight gradient
// is finished by t3d_pwconv_wgrad_yfree_finish (any stream ordered behind this launch)
extern "C" int t3d_pwconv_bwd_yfree(const void* dz, const void* x, const void* wd, const void* x_raw, const t3d_prologue* pro_in,
                                    const void* residual, void* dx, double* stats, void* scratch, long long scratch_bytes,
                                    int M, int HW, int K, int N, void* stream) {
  if (!dz || !x || !wd || !dx || !scratch || M <= 0 || HW <= 0 || K <= 0 || N <= 0 || (K % 8) || (N % 8)) return T3D_ERR_ARG;
  if (pro_in && (pro_in->se || pro_in->act != T3D_ACT_NONE || pro_in->scale)) return T3D_ERR_UNSUPPORTED;   // (a linear block output)
  const size_t need = t3d_pw_bwd_yfree_scratch(M, K, N);
  if (!need || (size_t)scratch_bytes < need) return T3D_ERR_UNSUPPORTED;
  return t3d_pw_bwd_yfree_launch(dz, x, wd, nullptr, nullptr, x_raw, residual, dx, stats, scratch, M, HW, K, N,
                                 reinterpret_cast<hipStream_t>(stream));
}

// the same launch with the data gradient's weight rows built in its own prologue: no t3d_pwconv_yfree_prep2 launch in front
extern "C" int t3d_pwconv_bwd_yfree_w(const void* dz, const void* x, const void* wt, const t3d_bnbwd* bb, const void* x_raw,
                                      const t3d_prologue* pro_in, const void* residual, void* dx, double* stats, void* scratch,
                                      long long scratch_bytes, int M, int HW, int K, int N, void* stream) {
  if (!dz || !x || !wt || !bb || !bb->alpha || !bb->beta || !bb->gamma || !dx || !scratch || M <= 0 || HW <= 0 || K <= 0 || N <= 0 ||
      (K % 8) || (N % 8))
    return T3D_ERR_ARG;
  if (bb->per_sample || (pro_in && (pro_in->se || pro_in->act != T3D_ACT_NONE || pro_in->scale))) return T3D_ERR_UNSUPPORTED;
  const size_t need = t3d_pw_bwd_yfree_scratch(M, K, N);
  if (!need || (size_t)scratch_bytes < need) return T3D_ERR_UNSUPPORTED;
  return t3d_pw_bwd_yfree_launch(dz, x, nullptr, wt, bb, x_raw, residual, dx, stats, scratch, M, HW, K, N,
                                 reinterpret_cast<hipStream_t>(stream));
}

extern "C" int t3d_pwconv_wgrad_yfree_finish(void* scratch, const t3d_bnbwd* bb, const void* w, float* dw, int M, int K, int N,
                                             void* stream) {
  if (!scratch || !bb || !bb->alpha || !bb->beta || !bb->gamma || !w || !dw) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* tmp = nullptr;
  if (const int rc = t3d_pw_bwd_yfree_reduce(scratch, &tmp, M, K, N, st)) return rc;
  T3D_LAUNCH(yfree_combine_kernel, dim3(cdiv(N * K, 256)), dim3(256), 0, st, tmp, reinterpret_cast<const bf16_t*>(w),
                     bb->alpha, bb->beta, bb->gamma, dw, K, N);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

static int yfree_prep_impl(const void* wt, const t3d_bnbwd* bb, void* wcat, float* cvec, int K, int N, void* wd, int PBw,
                           void* stream) {
  const void* w = wt;
  if (!w || !bb || !bb->alpha || !bb->beta || !bb->gamma || !wcat || !cvec || K <= 0 || N <= 0 || (K % 8) || (N % 8))
    return T3D_ERR_ARG;
  if (bb->per_sample) return T3D_ERR_UNSUPPORTED;
  const bool derive = true;
  const T3dFold* fold = nullptr;
  if (derive) fold = t3d_take_fold(bb->alpha);
  else if (const int rc = t3d_fold_fallback(bb->alpha, reinterpret_cast<hipStream_t>(stream))) return rc;
  T3D_LAUNCH(yfree_prep_kernel, dim3(rup32(K) / 16, rup32(K) / 16), dim3(256), (size_t)3 * N * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(w), bb->alpha, bb->beta, bb->gamma,
                     reinterpret_cast<bf16_t*>(wcat), cvec, K, N, rup32(N), rup32(K), fold, reinterpret_cast<bf16_t*>(wd), PBw);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pwconv_dgrad_yfree(const void* dz, const void* x, const void* wcat, const float* cvec,
                                      const void* x_raw, const t3d_prologue* pro_in, const void* residual, void* dx,
                                      double* stats, int M, int HW, int K, int N, void* stream) {
  if (!dz || !x || !wcat || !cvec || !dx || M <= 0 || HW <= 0 || K <= 0 || N <= 0 || (K % 8) || (N % 8)) return T3D_ERR_ARG;
  if (pro_in && pro_in->se) return T3D_ERR_UNSUPPORTED;
  t3d_pw::GemmArgs a{};
  a.dgrad = 1;
  a.a0 = dz; a.row0 = N;
  a.a2 = x; a.Kin2 = K; a.ks1 = rup32(N) / 32;
  a.w = wcat; a.bias = cvec; a.e_res = residual; a.out = dx; a.stats = stats;
  if (x_raw) {
    a.e_y = x_raw;
    if (pro_in) { a.e_scale = pro_in->scale; a.e_shift = pro_in->shift; a.e_act = pro_in->act; }
  }
  a.M = M; a.HW = HW; a.Kin = rup32(N) + rup32(K); a.Nout = K;
  return t3d_pw::stream_launch(a, reinterpret_cast<hipStream_t>(stream));   // T3D_ERR_UNSUPPORTED if the shape does not fit
}

extern "C" int t3d_pwconv_wgrad_yfree(const void* dz, const void* x, const t3d_bnbwd* bb, const void* w, float* dw,
                                      int M, int HW, int K, int N, void* stream) {
  if (!dz || !x || !bb || !bb->alpha || !bb->beta || !bb->gamma || !w || !dw || M <= 0 || HW <= 0 || K <= 0 || N <= 0 ||
      (K % 8) || (N % 8))
    return T3D_ERR_ARG;
  if (bb->per_sample) return T3D_ERR_UNSUPPORTED;
  // products live at the tail of the caller's workspace (t3d_set_workspace); the partial tiles use its head
  const size_t tmp_bytes = (size_t)(N + K + 8) * K * sizeof(float);
  if (!g_t3d_ws.ptr || (size_t)g_t3d_ws.bytes < 2 * tmp_bytes + (1 << 20)) return T3D_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* tmp = reinterpret_cast<float*>(reinterpret_cast<char*>(g_t3d_ws.ptr) + ((g_t3d_ws.bytes - tmp_bytes) & ~(size_t)255));
  // (tmp is written, not accumulated into: t3d_pw_wgrad_tr_yfree's reduction assigns every entry, or clears it first)
  const T3dWorkspace saved = g_t3d_ws;
  g_t3d_ws.bytes = reinterpret_cast<char*>(tmp) - reinterpret_cast<char*>(g_t3d_ws.ptr);   // keep the tiles off the tail
  const int rc = t3d_pw_wgrad_tr_yfree(dz, x, tmp, M, HW, K, N, st);
  g_t3d_ws = saved;
  if (rc != T3D_OK) return rc;
  T3D_LAUNCH(yfree_combine_kernel, dim3(cdiv(N * K, 256)), dim3(256), 0, st, tmp, reinterpret_cast<const bf16_t*>(w),
                     bb->alpha, bb->beta, bb->gamma, dw, K, N);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
