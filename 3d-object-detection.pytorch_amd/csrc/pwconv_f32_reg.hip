// Pointwise (1x1) convolution forward in fp32 STORAGE, inference (no BatchNorm statistics), gfx950 (round 5).
//
//   y[m][n] = sum_k act(scale[k] * x[m][k] + shift[k]) * w[n][k]  (+ bias[n])          x, w, y fp32; exact fp32 products
//
// The LDS-tiled kernel of round 1 (pwconv.hip) is what `model.eval()` forwards of a bf16-trained model run through by default
// (fp32 storage: keypoints 1e-4 / arg-max exact, the north-star's bounds as written) -- 62 % of that forward's 6.9 ms.  It
// stages both operands through LDS behind two barriers per 32-deep step; on `v_mfma_f32_16x16x4_f32` the many-pixel layers
// are bound by HBM and the 14x14 / 7x7 ones by the fp32 MFMA rate (157 TFLOP/s nominal = 1/16 of bf16), and neither needs LDS:
//   * a wave takes R pixel groups (16 pixels each) x NT output tiles (16 channels each) for the WHOLE contraction;
//   * per step of 16 contraction indices lane (lc, lg) loads one float4 of each of its R pixel rows and NT weight rows straight
//     from global memory (k = 16 g + 4 lg .. + 3); the x rows come from HBM / L2, the weight rows from L1 / L2 -- the four waves
//     of a workgroup read the same ones.  Element j of the float4 is the operand of MFMA j: the contraction index of lane
//     group lg in that MFMA is 16 g + 4 lg + j for both operands, and a sum does not care how its terms are grouped.  (Two
//     float4 per operand and step -- whole 128-byte lines, twice the MFMAs behind each load -- measured no faster on any layer);
//   * 4 R NT MFMAs per step on R NT independent accumulators, operands double-buffered in registers; the prologue
//     coefficients are the only LDS use, there is no barrier and no branch in the loop;
//   * epilogue: a lane holds 4 consecutive channels of one pixel per tile: one 16-byte store.
// The output chunks of one block of pixel groups sit on one XCD (they read the same rows).  An LDS-resident-weights streaming
// form (the bf16 kernel's structure, pwconv_stream.hip) was built first and measured slower on 18 of MobileNetV2's 19 layer
// shapes (3.5 against 2.6 ms per batch-256 forward; DESIGN.md finding 47).  No squeeze-excite gate, no statistics, no data
// gradient: those stay with pwconv.hip (the training forward / backward of the parity mode, MobileNetV3's gated layers).
#include "pwconv_common.h"
#include <cstdlib>

namespace t3d_pw {
namespace {

// The operand transform is branch-free: act(u) = clamp(u, lo, hi) with (lo, hi) = (0, 6) ReLU6, (0, inf) ReLU, (-inf, inf) none
// -- v_med3_f32, bit-identical to the switch of common.h: act_affine_vec -- and scale = 1, shift = 0 without a prologue; hard-
// swish is the HS instantiation.  Coefficients past K are scale = shift = 0: those lanes' (clamped, valid) loads become zeros.
// Variant V: 0 the clamp form; 1 hard-swish; 2 one step (K <= 16: one operand buffer -- 40 registers less, a third wave per SIMD
// at 4 x 6); 3 / 4 the MATERIALISING forward (t3d_pwconv_fwd_mat): the operand is the finished block output z = scale x + shift
// (3) + residual (4: one more float4 per pixel row and step), and the workgroups of output chunk 0 also store it to z_out;
// SF (any of 0, 1, 3, 4) the TRAINING forward: BatchNorm sums of the output (column sums of y and y^2 over the valid pixels) -- the workgroups are
// persistent over the pixel blocks, the column sums of every 16-pixel group (fp32, DPP) go to fp64 LDS accumulators and leave
// as one fp64 atomic per channel and workgroup into a reduction replica.  fp32 values added in fp64: exact, so the sums do not
// depend on the order (run-to-run bit-identical like the tiled kernel's);
// 6 the DATA GRADIENT (t3d_pwconv_dgrad without gates / per-sample coefficients): the operand is the BatchNorm-backward affine of
// two tensors, alpha dz + beta y + gamma (one more float4 per pixel row and step); epilogue: x the activation derivative at the
// differentiated conv's input (e_y, e_scale, e_shift, e_act), + the residual gradient, and the sums of dx and dx . e_y (or dx^2)
// for the producer's BatchNorm backward -- persistent like the SF variants;
// 7 / 8 the clamp form / hard-swish with a squeeze-excite gate on the operand (p2 [B][K], multiplied in before or after the activation).
template <int R, int NT, int V, bool SF>
__global__ __launch_bounds__(256) void pw_f32_reg_kernel(const GemmArgs a, const int KG, const int nchunks, const float lo,
                                                         const float hi, const int nrep, const long long rstride) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  constexpr bool GT = V == 7 || V == 8;           // squeeze-excite gate of the operand: p2 [B][K], before or after the activation
  constexpr bool HS = V == 1 || V == 8, ONE = V == 2, ZM = V == 3 || V == 4, ZR = V == 4, DG = V == 6, ST = SF || DG;
  constexpr int KS = 16;                                              // contraction indices per step
  float* coef = smem_f;                                               // [2][KG * KS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lg = lane >> 4;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int chunk = jj % nchunks, wgp = (jj / nchunks) * 8 + xcd;      // output chunk; workgroup index along the pixels
  const int npw = gridDim.x / nchunks;
  const int n0 = chunk * 16 * NT;
  const float* __restrict__ x = reinterpret_cast<const float*>(a.a0);
  const float* __restrict__ w = reinterpret_cast<const float*>(a.w);
  float* __restrict__ y = reinterpret_cast<float*>(a.out);
  const int K = a.Kin, N = a.Nout, kpad = KG * KS;
  double* lstat = reinterpret_cast<double*>(smem_f + 3 * kpad);      // [NT * 16][2] (ST)
  for (int i = tid; i < kpad; i += 256) {
    if (DG) {                                                         // alpha | beta | gamma (zeros past K)
      coef[i] = i < K ? a.p0[i] : 0.f;
      coef[kpad + i] = i < K ? a.p1[i] : 0.f;
      coef[2 * kpad + i] = i < K ? a.p2[i] : 0.f;
    } else {
      coef[i] = i < K ? (a.p0 ? a.p0[i] : 1.f) : 0.f;
      coef[kpad + i] = (i < K && a.p0) ? a.p1[i] : 0.f;
    }
  }
  if (ST)
    for (int i = tid; i < NT * 32; i += 256) lstat[i] = 0.0;
  __syncthreads();
  // a wave takes the block of R pixel groups (wgp + it * npw) * 4 + wave; every variant but ST is launched with one block per wave
  for (int it = 0;; ++it) {
  const int m0 = ((wgp + it * npw) * 4 + wave) * R * 16;
  if (m0 >= a.M) break;

  const float* xp[R];
  const float* wp[NT];
#pragma unroll
  for (int r = 0; r < R; ++r) xp[r] = x + (size_t)min(m0 + 16 * r + lc, a.M - 1) * K;
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = w + (size_t)min(n0 + 16 * t + lc, N - 1) * K;
  f32x4 xa[ONE ? 1 : 2][R], wa[ONE ? 1 : 2][NT], ra[ZR || DG || GT ? 2 : 1][ZR || DG || GT ? R : 1];
  const float* __restrict__ res = GT ? a.p2 : reinterpret_cast<const float*>(DG ? a.a1 : a.z_res);      // second operand tensor / gates
  float* __restrict__ zo = reinterpret_cast<float*>(a.z_out);
  size_t roff[R];                                          // (ZM: residual / z_out rows = the operand's rows)
#pragma unroll
  for (int r = 0; r < R; ++r) roff[r] = (size_t)(GT ? min(m0 + 16 * r + lc, a.M - 1) / a.HW : min(m0 + 16 * r + lc, a.M - 1)) * K;     // (GT: the sample's gate row)
  auto load = [&](int g, int b) {
    const int k = min(KS * g + 4 * lg, K - 4);            // (past K: a valid address; the coefficients there are zeros)
#pragma unroll
    for (int r = 0; r < R; ++r) xa[b][r] = *reinterpret_cast<const f32x4*>(xp[r] + k);
    if constexpr (ZR || DG || GT) {
#pragma unroll
      for (int r = 0; r < R; ++r) ra[b][r] = *reinterpret_cast<const f32x4*>(res + roff[r] + k);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) wa[b][t] = *reinterpret_cast<const f32x4*>(wp[t] + k);
  };
  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto step = [&](int g, int b) {
    const int k = KS * g + 4 * lg;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(coef + k);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(coef + kpad + k);
    if constexpr (DG) {
      // BatchNorm backward through the differentiated conv's output: alpha dz + beta y + gamma (the tiled kernel's expression)
      const f32x4 ga = *reinterpret_cast<const f32x4*>(coef + 2 * kpad + k);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[b][r][j] = sc[j] * xa[b][r][j] + sh[j] * ra[b][r][j] + ga[j];
    } else if constexpr (ZM) {
      // z = scale x + shift (+ residual): the fp32 value t3d_bn_apply would have written (fma, then the add); stored by chunk 0
      const bool kin = k < K;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f32x4 z;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          z[j] = fmaf(xa[b][r][j], sc[j], sh[j]);
          if constexpr (ZR) z[j] += ra[b][r][j];
        }
        if (chunk == 0 && kin && m0 + 16 * r + lc < a.M) *reinterpret_cast<f32x4*>(zo + roff[r] + k) = z;
        xa[b][r] = kin ? z : f32x4{0.f, 0.f, 0.f, 0.f};       // (with a residual the zero coefficients past K are not enough)
      }
    } else {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = fmaf(xa[b][r][j], sc[j], sh[j]);
        if constexpr (GT) u *= a.se_after ? 1.f : ra[b][r][j];
        u = HS ? u * (__builtin_amdgcn_fmed3f(u + 3.f, 0.f, 6.f) * T3D_SIXTH) : __builtin_amdgcn_fmed3f(u, lo, hi);
        if constexpr (GT) u *= a.se_after ? ra[b][r][j] : 1.f;
        xa[b][r][j] = u;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][t][j], xa[b][r][j], acc[r][t], 0, 0, 0);
  };
  // (sched_barrier: the loads of step g + 1 are ISSUED before the MFMAs of step g -- left alone, the scheduler sinks them behind
  // the MFMAs into one register buffer and waits for them at the top of the next step)
  load(0, 0);
  if (ONE) step(0, 0);
  for (int g = 0; !ONE && g < KG; g += 2) {
    load(min(g + 1, KG - 1), ONE ? 0 : 1);
    __builtin_amdgcn_sched_barrier(0);
    step(g, 0);
    __builtin_amdgcn_sched_barrier(0);
    load(min(g + 2, KG - 1), 0);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < KG) step(g + 1, ONE ? 0 : 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4 sq[DG ? R : 1][DG ? NT : 1];                      // (DG: the second sum's summands)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int m = m0 + 16 * r + lc;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + 4 * lg;
      if (DG) {
        if (m < a.M && n < N) {
          f32x4 o = acc[r][t], yv = f32x4{0.f, 0.f, 0.f, 0.f};
          if (a.e_y) {
            yv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.e_y) + (size_t)m * N + n);
            const f32x4 es = a.e_scale ? *reinterpret_cast<const f32x4*>(a.e_scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
            const f32x4 eh = a.e_scale ? *reinterpret_cast<const f32x4*>(a.e_shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] *= act_grad(yv[j] * es[j] + eh[j], a.e_act);
          }
          if (a.e_res) o += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.e_res) + (size_t)m * N + n);
          *reinterpret_cast<f32x4*>(y + (size_t)m * N + n) = o;
          acc[r][t] = o;
          if (a.e_y) sq[r][t] = o * yv;                       // second sum: dx . e_y
          else sq[r][t] = o * o;
        } else {
          acc[r][t] = sq[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      } else if (m < a.M && n < N) {
        f32x4 o = acc[r][t];
        if (a.bias) {
          o += *reinterpret_cast<const f32x4*>(a.bias + n);
          if (ST) acc[r][t] = o;                         // (the sums below are those of the STORED values)
        }
        *reinterpret_cast<f32x4*>(y + (size_t)m * N + n) = o;
      }
    }
  }
  if (ST && a.stats) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int nl = 16 * t + 4 * lg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // fp32 only over the 16 pixels of a group (as the tiled kernel: pwconv.hip), fp64 from there on
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float v = m0 + 16 * r + lc < a.M ? acc[r][t][j] : 0.f;      // (rows past M: clamped repeats of the last row)
          s1 += (double)row16_sum(v);
          if constexpr (DG) s2 += (double)row16_sum(sq[r][t][j]);
          else s2 += (double)row16_sum(v * v);
        }
        if (lc == 0 && n0 + nl < N) {
          atomicAdd(lstat + (nl + j) * 2, s1);
          atomicAdd(lstat + (nl + j) * 2 + 1, s2);
        }
      }
    }
  }
  if (!ST) break;
  }   // blocks of this wave
  if (ST && a.stats) {
    __syncthreads();
    double* st = a.stats + (size_t)(blockIdx.x % nrep) * rstride;
    for (int i = tid; i < NT * 32; i += 256) {
      const int n = n0 + (i >> 1);
      if (n < N) atomicAdd(st + (size_t)(i & 1) * N + n, lstat[i]);
    }
  }
}

template <int R, int NT>
int launch_reg(GemmArgs& a, hipStream_t st) {
  const int KG = cdiv(a.Kin, 16), nchunks = cdiv(cdiv(a.Nout, 16), NT);
  const int npb4 = cdiv(cdiv(cdiv(a.M, 16), R), 4);          // workgroups along the pixels: 4 waves, R pixel groups each
  long long npw = (long long)cdiv(npb4, 8) * 8;
  if constexpr (R == 4 && NT > 4) {
    if (a.dgrad) return launch_reg<2, NT>(a, st);            // (three operand streams: 308 / 364 registers at 4 x 5 / 4 x 6)
    if (a.stats && (NT == 6 || a.z_res)) return launch_reg<2, NT>(a, st); // (260 registers with the sums' epilogue at 4 x 6)
    if (a.p2 && !a.dgrad) return launch_reg<2, NT>(a, st);                // (gates: one more float4 per pixel row and step)
  }
  if (a.stats || a.dgrad) {                                  // persistent: two workgroups per CU, whole XCD lanes
    const long long cap = (512 / nchunks) / 8 * 8;
    if (npw > (cap < 8 ? 8 : cap)) npw = cap < 8 ? 8 : cap;
  }
  const long long grid = npw * nchunks;
  if (grid >= (1ll << 31)) return T3D_ERR_UNSUPPORTED;
  const size_t lds = (size_t)3 * KG * 16 * 4 + (size_t)NT * 32 * 8;
  const int nrep = g_t3d_reduce.nrep > 0 ? g_t3d_reduce.nrep : 1;
  const long long rstride = g_t3d_reduce.stats_stride;
  const float inf = __builtin_inff();
  const float lo = (a.act == T3D_ACT_RELU || a.act == T3D_ACT_RELU6) ? 0.f : -inf, hi = a.act == T3D_ACT_RELU6 ? 6.f : inf;
#define T3D_REG_LAUNCH(VV, SS) \
  T3D_LAUNCH_TIMED((pw_f32_reg_kernel<R, NT, VV, SS>), dim3((unsigned)grid), dim3(256), lds, st, a, KG, nchunks, lo, hi, nrep, rstride)
  if (a.dgrad) T3D_REG_LAUNCH(6, false);
  else if (a.p2) {
    if (a.stats) { if (a.act == T3D_ACT_HSWISH) T3D_REG_LAUNCH(8, true); else T3D_REG_LAUNCH(7, true); }
    else { if (a.act == T3D_ACT_HSWISH) T3D_REG_LAUNCH(8, false); else T3D_REG_LAUNCH(7, false); }
  }
  else if (a.stats) {
    if (a.z_out && a.z_res) T3D_REG_LAUNCH(4, true);
    else if (a.z_out) T3D_REG_LAUNCH(3, true);
    else if (a.act == T3D_ACT_HSWISH) T3D_REG_LAUNCH(1, true);
    else T3D_REG_LAUNCH(0, true);
  }
  else if (a.z_out && a.z_res) T3D_REG_LAUNCH(4, false);
  else if (a.z_out) T3D_REG_LAUNCH(3, false);
  else if (a.act == T3D_ACT_HSWISH) T3D_REG_LAUNCH(1, false);
  else if (KG == 1) T3D_REG_LAUNCH(2, false);
  else T3D_REG_LAUNCH(0, false);
#undef T3D_REG_LAUNCH
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <int R>
int launch_reg_nt(GemmArgs& a, int NT, hipStream_t st) {
  switch (NT) {
    case 2: return launch_reg<R, 2>(a, st);
    case 3: return launch_reg<R, 3>(a, st);
    case 4: return launch_reg<R, 4>(a, st);
    case 5: return launch_reg<R, 5>(a, st);
    default: return launch_reg<R, 6>(a, st);
  }
}

}  // namespace

// fp32 storage, inference forward; T3D_ERR_UNSUPPORTED = "not a launch for this kernel" (pwconv.hip takes it)
int f32_reg_launch(GemmArgs& a, hipStream_t st) {
  if (a.dgrad) {
    if (!a.a1 || !a.p0 || !a.p1 || !a.p2 || a.per_sample || a.ps_stats || a.e_se || a.a2 || a.cv.mode || a.fold || a.kz > 1 ||
        a.wfrag || a.bias || a.z_out || a.z_res || !a.out)
      return T3D_ERR_UNSUPPORTED;
  } else
  if (a.ps_stats || (a.p2 && a.z_out) || a.a1 || a.a2 || a.cv.mode || a.fold || a.per_sample || a.e_se || a.kz > 1 ||
      a.wfrag || (a.z_res && !a.z_out) || (a.z_out && (a.act != T3D_ACT_NONE || a.bias)) ||
      (a.stats && !a.out))
    return T3D_ERR_UNSUPPORTED;
  if ((a.Kin % 8) || (a.Nout % 8) || a.M < 1024) return T3D_ERR_UNSUPPORTED;      // (few-pixel layers: the split-contraction path)
  // Task shape (tools/time_pw_f32.py --sweep): output tiles per wave = the count that pads the layer's tiles least, 5 and 4
  // before 6 (4 x 6 takes 224 registers) before 3; 4 pixel groups per wave unless that leaves fewer than 128 workgroups (7x7 x
  // 256 = 784 groups x 10 tiles as 4 x 5 would be 98), then 2
  const int tiles = cdiv(a.Nout, 16), G = cdiv(a.M, 16);
  int NT = 2;
  if (tiles > 2) {
    int pad = 1 << 30;
    for (int c : {5, 4, 6, 3})
      if (cdiv(tiles, c) * c < pad) { pad = cdiv(tiles, c) * c; NT = c; }
  }
  int R = (long long)cdiv(cdiv(G, 4), 4) * cdiv(tiles, NT) >= 128 ? 4 : 2;
  if (const char* e = getenv("T3D_F32_SHAPE")) { R = e[0] - '0'; NT = e[1] - '0'; }    // (sweep knob)
  return R == 4 ? launch_reg_nt<4>(a, NT, st) : launch_reg_nt<2>(a, NT, st);
}

}  // namespace t3d_pw
