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

namespace {

// grid (K, KP/16): output channel k of the data gradient x a 16-wide slice of the concatenated contraction axis's Q
// part; 256 threads.  The K x K block Q = W^T diag(beta) W is the only part with real work (N multiply-adds per entry):
// thread (k2, g) sums the n = g (mod 16) terms, the 16 partial sums meet in LDS.  (Small blocks on purpose: the kernel
// runs while the side stream keeps the CUs busy, and a 1024-thread block waits for a whole free CU.)
__global__ __launch_bounds__(256) void yfree_prep_kernel(const bf16_t* __restrict__ w, const float* __restrict__ alpha_g,
                                                         const float* __restrict__ beta_g, const float* __restrict__ gamma_g,
                                                         bf16_t* __restrict__ wcat, float* __restrict__ cvec, int K, int N,
                                                         int NP, int KP, const T3dFold* __restrict__ fold) {
  __shared__ float part[256];
  extern __shared__ float fco[];      // [3][N] derived (alpha, beta, gamma) when a finalize request rides on this launch
  const float *alpha = alpha_g, *beta = beta_g, *gamma = gamma_g;
  if (fold) {
    // every workgroup needs all N coefficients: each derives them from the replica sums (common.h), workgroup (0, 0)
    // publishes -- 3 us in every workgroup's prologue against a 5-us finalize launch and the dispatch gap behind it
    t3d_fold_block(fold, 0, N, fco, N, blockIdx.x == 0 && blockIdx.y == 0);
    __syncthreads();
    alpha = fco; beta = fco + N; gamma = fco + 2 * N;
  }
  constexpr int G = 16;
  const int k = blockIdx.x, c = blockIdx.y, nc = gridDim.y, tot = NP + KP, t = threadIdx.x;
  for (int j = c * 256 + t; j < NP; j += nc * 256)
    wcat[(size_t)k * tot + j] = (bf16_t)(j < N ? alpha[j] * (float)w[(size_t)j * K + k] : 0.f);
  const int k2 = c * 16 + (t & 15), g = t >> 4;
  float acc = 0.f;
  if (k2 < K) {
    int n = g;
    for (; n + 3 * G < N; n += 4 * G) {      // four independent loads in flight
      const float b0 = beta[n], b1 = beta[n + G], b2 = beta[n + 2 * G], b3 = beta[n + 3 * G];
      const float u0 = (float)w[(size_t)n * K + k2], u1 = (float)w[(size_t)(n + G) * K + k2];
      const float u2 = (float)w[(size_t)(n + 2 * G) * K + k2], u3 = (float)w[(size_t)(n + 3 * G) * K + k2];
      const float v0 = (float)w[(size_t)n * K + k], v1 = (float)w[(size_t)(n + G) * K + k];
      const float v2 = (float)w[(size_t)(n + 2 * G) * K + k], v3 = (float)w[(size_t)(n + 3 * G) * K + k];
      acc = fmaf(b0 * u0, v0, acc); acc = fmaf(b1 * u1, v1, acc); acc = fmaf(b2 * u2, v2, acc); acc = fmaf(b3 * u3, v3, acc);
    }
    for (; n < N; n += G) acc = fmaf(beta[n] * (float)w[(size_t)n * K + k2], (float)w[(size_t)n * K + k], acc);
  }
  part[t] = acc;
  __syncthreads();
  if (t < 16) {
    float v = 0.f;
#pragma unroll
    for (int gg = 0; gg < G; ++gg) v += part[gg * 16 + t];
    wcat[(size_t)k * tot + NP + c * 16 + t] = (bf16_t)v;     // k2 >= K: zero padding (acc stayed 0)
  }
  if (c == 0 && t >= 192) {     // last wave of the first slice: c[k] = sum_n gamma_n W[n][k]
    float s2 = 0.f;
    for (int n = t - 192; n < N; n += 64) s2 = fmaf(gamma[n], (float)w[(size_t)n * K + k], s2);
    s2 = wave_sum(s2);
    if (t == 192) cvec[k] = s2;
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
extern "C" int t3d_pwconv_yfree_prep(const void* w, const t3d_bnbwd* bb, void* wcat, float* cvec, int K, int N,
                                     void* stream) {
  if (!w || !bb || !bb->alpha || !bb->beta || !bb->gamma || !wcat || !cvec || K <= 0 || N <= 0 || (K % 8) || (N % 8))
    return T3D_ERR_ARG;
  if (bb->per_sample) return T3D_ERR_UNSUPPORTED;
  static const bool derive = !getenv("T3D_YFREE_PREP_NO_DERIVE");
  const T3dFold* fold = nullptr;
  if (derive) fold = t3d_take_fold(bb->alpha);
  else if (const int rc = t3d_fold_fallback(bb->alpha, reinterpret_cast<hipStream_t>(stream))) return rc;
  hipLaunchKernelGGL(yfree_prep_kernel, dim3(K, rup32(K) / 16), dim3(256), fold ? (size_t)3 * N * sizeof(float) : 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(w), bb->alpha, bb->beta, bb->gamma,
                     reinterpret_cast<bf16_t*>(wcat), cvec, K, N, rup32(N), rup32(K), fold);
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
  if (hipMemsetAsync(tmp, 0, tmp_bytes, st) != hipSuccess) return T3D_ERR_LAUNCH;
  const T3dWorkspace saved = g_t3d_ws;
  g_t3d_ws.bytes = reinterpret_cast<char*>(tmp) - reinterpret_cast<char*>(g_t3d_ws.ptr);   // keep the tiles off the tail
  const int rc = t3d_pw_wgrad_tr_yfree(dz, x, tmp, M, HW, K, N, st);
  g_t3d_ws = saved;
  if (rc != T3D_OK) return rc;
  hipLaunchKernelGGL(yfree_combine_kernel, dim3(cdiv(N * K, 256)), dim3(256), 0, st, tmp, reinterpret_cast<const bf16_t*>(w),
                     bb->alpha, bb->beta, bb->gamma, dw, K, N);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
