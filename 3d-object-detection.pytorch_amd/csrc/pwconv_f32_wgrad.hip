// Pointwise (1x1) convolution weight gradient in fp32 STORAGE, gfx950 (round 5) -- the parity mode's counterpart of
// pwconv_f32_reg.hip:
//
//   dW[n][k] += sum_m  dy[m][n] * a[m][k],      dy = alpha*dz + beta*y + gamma   (BatchNorm backward)
//                                               a  = act(scale*x + shift)        (recomputed, never stored)
//
// The LDS-tiled kernel of round 1 (pwconv_wgrad.hip) stages 64 pixels of both operands through LDS behind two barriers and
// gives each wave a 16 x 64 strip: 9.4 ms of the fp32 training step (0.84 TB/s).  Here, as in the forward, nothing goes
// through LDS: a wave owns a 64 x 64 block of dW over a range of pixels; per step of FOUR pixels lane (lc, lg) loads one float4
// of dz, y (channels n0 + 4 lc .. + 3 of pixel m0 + lg) and x (channels k0 + 4 lc .. + 3): element i of the transformed dy
// float4 is the A operand of tile row-block i (MFMA row lc <-> channel n0 + 4 lc + i, contraction index lg <-> pixel m0 + lg),
// element j of the activated x float4 the B operand of tile column-block j -- 16 v_mfma_f32_16x16x4_f32 per three 16-byte
// loads, the coefficients of a lane's eight channels in registers for the whole walk, a ring of DEPTH steps of loads in flight.
// The pixel ranges leave as partial 64 x 64 tiles in the caller's workspace (plain stores) and are added in a fixed order
// (t3d_pw_wgrad_reduce): bit-reproducible.  Blocks past the matrix edge (N or K not a multiple of 64) load clamped addresses
// and multiply by zero coefficients.  No squeeze-excite gates, no per-sample coefficients: those stay with pwconv_wgrad.hip.
#include <cstdlib>
#include "pwconv_common.h"

namespace {

struct Wg32Args {
  const float *dz, *y, *x;
  const float *alpha, *beta, *gamma, *scale, *shift;
  float* ws;
  int M, K, N, rows_per_split, tk;
  float lo, hi;
};

constexpr int DEPTH = 6;      // steps (of 4 pixels) of loads in flight per wave: 18 float4

template <bool HS>
__global__ __launch_bounds__(256) void pw_wgrad_f32_reg_kernel(const Wg32Args a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lg = lane >> 4;
  // blockIdx.x = (n block, k block); blockIdx.y = group of 4 pixel splits, one per wave
  const int nb = blockIdx.x / a.tk, kb = blockIdx.x % a.tk;
  const int split = blockIdx.y * 4 + wave;
  const int mbeg = split * a.rows_per_split, mend = min(a.M, mbeg + a.rows_per_split);
  if (mbeg >= mend) return;
  const int nch = nb * 64 + 4 * lc, kch = kb * 64 + 4 * lc;
  const bool nok = nch < a.N, kok = kch < a.K;                 // (N % 8 == 0, K % 8 == 0: a lane's 4 channels are in or out)
  f32x4 al, be, ga, sc, sh;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    al[j] = nok ? a.alpha[nch + j] : 0.f;
    be[j] = nok ? a.beta[nch + j] : 0.f;
    ga[j] = nok ? a.gamma[nch + j] : 0.f;
    sc[j] = kok ? (a.scale ? a.scale[kch + j] : 1.f) : 0.f;
    sh[j] = (kok && a.scale) ? a.shift[kch + j] : 0.f;
  }
  const int nc = min(nch, a.N - 4), kc = min(kch, a.K - 4);     // clamped: valid addresses, zero coefficients
  const float* __restrict__ pz = a.dz + nc;
  const float* __restrict__ py = a.y + nc;
  const float* __restrict__ px = a.x + kc;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 rz[DEPTH], ry[DEPTH], rx[DEPTH];
  const int nsteps = (mend - mbeg + 3) >> 2;
  auto issue = [&](int s, int slot) {
    const size_t m = (size_t)min(mbeg + 4 * min(s, nsteps - 1) + lg, a.M - 1);
    rz[slot] = *reinterpret_cast<const f32x4*>(pz + m * a.N);
    ry[slot] = *reinterpret_cast<const f32x4*>(py + m * a.N);
    rx[slot] = *reinterpret_cast<const f32x4*>(px + m * a.K);
  };
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) issue(u, u);
  for (int s0 = 0; s0 < nsteps; s0 += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int s = s0 + u;
      if (s < nsteps) {                                          // wave-uniform
        const bool mok = mbeg + 4 * s + lg < mend;
        f32x4 dy, av;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = al[j] * rz[u][j] + be[j] * ry[u][j] + ga[j];      // (the tiled kernel's expression)
          dy[j] = mok ? d : 0.f;
          const float v = rx[u][j] * sc[j] + sh[j];
          av[j] = HS ? v * (__builtin_amdgcn_fmed3f(v + 3.f, 0.f, 6.f) * T3D_SIXTH) : __builtin_amdgcn_fmed3f(v, a.lo, a.hi);
        }
        issue(s + DEPTH, u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[i], av[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // acc[i][j][reg] = dW[n0 + 4 (4 lg + reg) + i][k0 + 4 lc + j]: one float4 (j = 0..3) per (i, reg)
  float* wsb = a.ws + ((size_t)split * gridDim.x + blockIdx.x) * (64 * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(wsb + (16 * lg + 4 * r + i) * 64 + 4 * lc) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
}

}  // namespace

int t3d_pw_wgrad_reduce(const float* ws, float* dw, int N, int K, int PB, int QB, int qtiles, int tiles, int S, hipStream_t st);

// fp32 storage, per-channel coefficients, no gates; T3D_ERR_UNSUPPORTED = "not a launch for this kernel" (pwconv_wgrad.hip takes it)
int t3d_pw_wgrad_f32_reg(const float* dz, const float* y, const t3d_bnbwd* bb, const float* x, const t3d_prologue* pro, float* dw,
                         int M, int K, int N, hipStream_t st) {
  if (bb->per_sample || (pro && pro->se) || M < 1024 || !bb->alpha || !bb->gamma) return T3D_ERR_UNSUPPORTED;
  Wg32Args a{};
  a.dz = dz; a.y = y; a.x = x;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma;
  const int act = pro ? pro->act : T3D_ACT_NONE;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; }
  const float inf = __builtin_inff();
  a.lo = (act == T3D_ACT_RELU || act == T3D_ACT_RELU6) ? 0.f : -inf;
  a.hi = act == T3D_ACT_RELU6 ? 6.f : inf;
  a.M = M; a.K = K; a.N = N;
  const int tn = cdiv(N, 64), tk = cdiv(K, 64);
  a.tk = tk;
  // pixel splits: whole rounds of workgroups -- 256 k workgroups of four splits each over the (tn tk) blocks, the largest k <= 3 that
  // leaves a split >= 512 pixels (128 steps); measured per layer (tools/time_pw_f32_bwd.py): 28x28 x 256, 3 blocks: 392 splits
  // = 294 workgroups 97 us, 340 splits = 255 workgroups 67 us; <= 48 MB of partial tiles
  int SGsel = 0;
  for (int k = 3; k >= 1 && !SGsel; --k) {
    const int sg = (256 * k) / (tn * tk);
    if (sg >= 1 && (M / (4 * sg) >= 512 || k == 1)) SGsel = sg;
  }
  int S = 4 * (SGsel > 0 ? SGsel : 1);
  static const int waves_env = getenv("T3D_WG32_WAVES") ? atoi(getenv("T3D_WG32_WAVES")) : 0;      // (sweep knob, read once)
  if (waves_env) S = waves_env / (tn * tk);
  const int maxs = cdiv(M, 512);
  if (S > maxs) S = maxs;
  if (S < 4) S = 4;
  S = (S + 3) & ~3;
  a.rows_per_split = cdiv(cdiv(M, S), 4 * DEPTH) * 4 * DEPTH;
  S = cdiv(M, a.rows_per_split);
  const int SG = cdiv(S, 4);                                     // workgroups along the pixels (4 splits each)
  const size_t need = (size_t)SG * 4 * tn * tk * 64 * 64 * sizeof(float);
  if (!g_t3d_ws.ptr || (size_t)g_t3d_ws.bytes < need) return T3D_ERR_UNSUPPORTED;
  a.ws = reinterpret_cast<float*>(g_t3d_ws.ptr);
  // (splits past S inside the last workgroup return at once: their tiles must read as zeros)
  if (SG * 4 != S && hipMemsetAsync(a.ws + (size_t)S * tn * tk * 4096, 0, (size_t)(SG * 4 - S) * tn * tk * 4096 * sizeof(float), st) != hipSuccess)
    return T3D_ERR_LAUNCH;
  if (act == T3D_ACT_HSWISH)
    T3D_LAUNCH_TIMED((pw_wgrad_f32_reg_kernel<true>), dim3(tn * tk, SG), dim3(256), 0, st, a);
  else
    T3D_LAUNCH_TIMED((pw_wgrad_f32_reg_kernel<false>), dim3(tn * tk, SG), dim3(256), 0, st, a);
  T3D_CHECK_LAUNCH();
  return t3d_pw_wgrad_reduce(a.ws, dw, N, K, 64, 64, tk, tn * tk, SG * 4, st);
}
