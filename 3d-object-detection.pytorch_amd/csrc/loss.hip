// Keypoint / class losses and training metrics as wavefront reductions (fp32).
//
// One workgroup: thread t owns samples t, t+256, ...; every term of
//   torchdet3d/losses/regression_losses.py (DiagLoss :8-20, ADD_loss :22-26, WingLoss :28-49),
//   torchdet3d/builders/loss_builder.py:13-20 (CrossEntropy, SmoothL1, L1, MSE),
// combined as LossManager.parse_losses (:79-95), is evaluated per sample together with its
// analytic gradient, and the metrics of torchdet3d/evaluation/metrics.py:10-37 (ADD, symmetric
// ADD with the strict-< running minimum, arg-max accuracy) ride along.  The 9 scalar sums meet
// through a fixed-order wave + LDS reduction (deterministic), so one launch replaces ~250 tiny
// ATen kernels and 6 host syncs per iteration of the reference loop (trainer/train.py:48-62).
#include "common.h"

namespace {

constexpr int NS = 9;  // reg, ce, add_m, sadd_m, acc

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void loss_kernel(const t3d_loss_cfg cfg, const float* __restrict__ kp,
                                                   const float* __restrict__ gt, const float* __restrict__ logits,
                                                   const int64_t* __restrict__ cats, float* out, float* dkp,
                                                   float* dlogits, int B, int ncls) {
  __shared__ float red[4][NS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float invB = 1.f / (float)B, invN = 1.f / (float)(B * 18);
  float s_reg = 0.f, s_ce = 0.f, s_add = 0.f, s_sadd = 0.f, s_acc = 0.f;
  const float wing_c = cfg.wing_w - cfg.wing_w * logf(1.f + cfg.wing_w / cfg.wing_eps);

  for (int b = tid; b < B; b += 256) {
    float p[18], t[18], g[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      p[i] = kp[(size_t)b * 18 + i];
      t[i] = gt[(size_t)b * 18 + i];
      g[i] = 0.f;
    }
    float lreg = 0.f;
    // ---- elementwise terms (mean over B*18)
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const float d = p[i] - t[i], ad = fabsf(d);
      if (cfg.c_l1 != 0.f) {
        lreg += cfg.c_l1 * ad * invN;
        g[i] += cfg.c_l1 * sgn(d) * invN;
      }
      if (cfg.c_mse != 0.f) {
        lreg += cfg.c_mse * d * d * invN;
        g[i] += cfg.c_mse * 2.f * d * invN;
      }
      if (cfg.c_smoothl1 != 0.f) {
        const float be = cfg.smoothl1_beta;
        if (ad < be) {
          lreg += cfg.c_smoothl1 * 0.5f * d * d / be * invN;
          g[i] += cfg.c_smoothl1 * d / be * invN;
        } else {
          lreg += cfg.c_smoothl1 * (ad - 0.5f * be) * invN;
          g[i] += cfg.c_smoothl1 * sgn(d) * invN;
        }
      }
      if (cfg.c_wing != 0.f) {
        // regression_losses.py:36-38: two sequential in-place masked updates; the second mask
        // sees the already-updated values.
        float v = ad, dv = sgn(d);
        if (ad < cfg.wing_w) {
          v = cfg.wing_w * logf(1.f + ad / cfg.wing_eps);
          dv = sgn(d) * cfg.wing_w / (cfg.wing_eps + ad);
        }
        if (v >= cfg.wing_w) v -= wing_c;
        lreg += cfg.c_wing * v * invN;
        g[i] += cfg.c_wing * dv * invN;
      }
    }
    // ---- ADD loss: mean_b sum_k ||p_k - t_k||   (+ the ADD metric: mean over b and k)
    float addsum = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float dx = p[2 * k] - t[2 * k], dy = p[2 * k + 1] - t[2 * k + 1];
      const float n = sqrtf(dx * dx + dy * dy);
      addsum += n;
      if (cfg.c_add != 0.f && n > 0.f) {
        g[2 * k] += cfg.c_add * dx / n * invB;
        g[2 * k + 1] += cfg.c_add * dy / n * invB;
      }
    }
    if (cfg.c_add != 0.f) lreg += cfg.c_add * addsum * invB;
    s_add += addsum;
    // ---- symmetric ADD metric (metrics.py:13-21)
    float sadd = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      float dx = p[2 * i] - t[2 * i], dy = p[2 * i + 1] - t[2 * i + 1];
      float best = sqrtf(dx * dx + dy * dy);
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        dx = p[2 * i] - t[2 * j];
        dy = p[2 * i + 1] - t[2 * j + 1];
        const float d = sqrtf(dx * dx + dy * dy);
        best = d < best ? d : best;
      }
      sadd += best;
    }
    s_sadd += sadd;
    // ---- diagonal loss: SmoothL1(beta .4) between bounding-box diagonals (mean over B)
    if (cfg.c_diag != 0.f) {
      int ix0 = 0, ix1 = 0, iy0 = 0, iy1 = 0;
      float px0 = p[0], px1 = p[0], py0 = p[1], py1 = p[1];
      float tx0 = t[0], tx1 = t[0], ty0 = t[1], ty1 = t[1];
#pragma unroll
      for (int k = 1; k < 9; ++k) {  // first extremum wins, values tracked beside the indices (no dynamic indexing)
        if (p[2 * k] < px0) { px0 = p[2 * k]; ix0 = k; }
        if (p[2 * k] > px1) { px1 = p[2 * k]; ix1 = k; }
        if (p[2 * k + 1] < py0) { py0 = p[2 * k + 1]; iy0 = k; }
        if (p[2 * k + 1] > py1) { py1 = p[2 * k + 1]; iy1 = k; }
        tx0 = fminf(tx0, t[2 * k]); tx1 = fmaxf(tx1, t[2 * k]);
        ty0 = fminf(ty0, t[2 * k + 1]); ty1 = fmaxf(ty1, t[2 * k + 1]);
      }
      const float wx = px1 - px0, wy = py1 - py0;
      const float dgp = sqrtf(wx * wx + wy * wy);
      const float dgt = sqrtf((tx1 - tx0) * (tx1 - tx0) + (ty1 - ty0) * (ty1 - ty0));
      const float e = dgp - dgt, ae = fabsf(e);
      float de;
      if (ae < 0.4f) {
        lreg += cfg.c_diag * 0.5f * e * e / 0.4f * invB;
        de = e / 0.4f;
      } else {
        lreg += cfg.c_diag * (ae - 0.2f) * invB;
        de = sgn(e);
      }
      if (dgp > 0.f) {
        const float k = cfg.c_diag * de * invB / dgp;
#pragma unroll
        for (int q = 0; q < 9; ++q) {  // static indexing keeps g[] in registers
          if (q == ix1) g[2 * q] += k * wx;
          if (q == ix0) g[2 * q] -= k * wx;
          if (q == iy1) g[2 * q + 1] += k * wy;
          if (q == iy0) g[2 * q + 1] -= k * wy;
        }
      }
    }
    s_reg += lreg;
    if (dkp) {
#pragma unroll
      for (int i = 0; i < 18; ++i) dkp[(size_t)b * 18 + i] = cfg.lam_reg * g[i];
    }
    // ---- class head: cross entropy (mean over B) + arg-max accuracy (first maximum, as torch.argmax)
    const int c = (int)cats[b];
    if (logits) {
      const float* lg = logits + (size_t)b * ncls;
      float mx = lg[0];
      int am = 0;
      for (int q = 1; q < ncls; ++q)
        if (lg[q] > mx) { mx = lg[q]; am = q; }
      s_acc += (am == c) ? 1.f : 0.f;
      if (cfg.c_ce != 0.f) {
        float se = 0.f;
        for (int q = 0; q < ncls; ++q) se += expf(lg[q] - mx);
        const float lse = mx + logf(se);
        s_ce += cfg.c_ce * (lse - lg[c]) * invB;
        if (dlogits) {
          for (int q = 0; q < ncls; ++q) {
            const float sm = expf(lg[q] - lse);
            dlogits[(size_t)b * ncls + q] = cfg.lam_cls * cfg.c_ce * (sm - (q == c ? 1.f : 0.f)) * invB;
          }
        }
      } else if (dlogits) {
        for (int q = 0; q < ncls; ++q) dlogits[(size_t)b * ncls + q] = 0.f;
      }
    } else {
      s_acc += (c == 0) ? 1.f : 0.f;  // width-1 "targets": arg-max is always 0 (metrics.py:33)
    }
  }

  float v[5] = {s_reg, s_ce, s_add, s_sadd, s_acc};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    v[i] = wave_sum(v[i]);
    if (lane == 0) red[wave][i] = v[i];
  }
  __syncthreads();
  if (tid == 0) {
    float r[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) r[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    out[0] = cfg.lam_reg * r[0] + cfg.lam_cls * r[1];
    out[1] = r[0];
    out[2] = r[1];
    out[3] = r[2] * invN * 2.f;      // ADD, mean over B*9
    out[4] = r[3] * invB / 9.f;      // SADD, mean over B of (sum / 9)
    out[5] = r[4] * invB;            // accuracy
    out[6] = r[2] / 9.f;             // reduce_mean=False forms (metrics.py:27-28,37)
    out[7] = r[3] / 9.f;
    out[8] = r[4];
  }
}

// Per-SAMPLE metric terms for the validation loop's per-class aggregation (metrics.py:39-68): out [B][3] =
// sum_k ||p_k - t_k|| / 9, symmetric form / 9, arg-max hit -- the reduce_mean=False summands of metrics.py:27-28,37; the
// caller adds them per class (one read-back per batch instead of two launches + two syncs per class present).
__global__ __launch_bounds__(256) void metrics_per_sample_kernel(const float* __restrict__ kp, const float* __restrict__ gt,
                                                                 const float* __restrict__ logits,
                                                                 const int64_t* __restrict__ cats, float* __restrict__ out,
                                                                 int B, int ncls) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float p[18], t[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    p[i] = kp[(size_t)b * 18 + i];
    t[i] = gt[(size_t)b * 18 + i];
  }
  float addsum = 0.f, sadd = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    float dx = p[2 * i] - t[2 * i], dy = p[2 * i + 1] - t[2 * i + 1];
    float best = sqrtf(dx * dx + dy * dy);
    addsum += best;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      dx = p[2 * i] - t[2 * j];
      dy = p[2 * i + 1] - t[2 * j + 1];
      const float d = sqrtf(dx * dx + dy * dy);
      best = d < best ? d : best;
    }
    sadd += best;
  }
  const int c = (int)cats[b];
  float hit;
  if (logits) {
    const float* lg = logits + (size_t)b * ncls;
    float mx = lg[0];
    int am = 0;
    for (int q = 1; q < ncls; ++q)
      if (lg[q] > mx) { mx = lg[q]; am = q; }
    hit = (am == c) ? 1.f : 0.f;
  } else {
    hit = (c == 0) ? 1.f : 0.f;
  }
  out[(size_t)b * 3 + 0] = addsum / 9.f;
  out[(size_t)b * 3 + 1] = sadd / 9.f;
  out[(size_t)b * 3 + 2] = hit;
}

}  // namespace

extern "C" int t3d_metrics_per_sample(const float* kp, const float* gt_kp, const float* logits, const int64_t* cats,
                                      float* out, int B, int ncls, void* stream) {
  if (!kp || !gt_kp || !cats || !out || B <= 0) return T3D_ERR_ARG;
  if (logits && (ncls <= 0 || ncls > 64)) return T3D_ERR_ARG;
  T3D_LAUNCH(metrics_per_sample_kernel, dim3(cdiv(B, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     kp, gt_kp, logits, cats, out, B, ncls);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_loss_fwd_bwd(const t3d_loss_cfg* cfg, const float* kp, const float* gt_kp, const float* logits,
                                const int64_t* cats, float* out, float* dkp, float* dlogits, int B, int ncls,
                                void* stream) {
  if (!cfg || !kp || !gt_kp || !cats || !out || B <= 0) return T3D_ERR_ARG;
  if (logits && (ncls <= 0 || ncls > 64)) return T3D_ERR_ARG;
  T3D_LAUNCH(loss_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *cfg, kp, gt_kp,
                     logits, cats, out, dkp, dlogits, B, ncls);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
