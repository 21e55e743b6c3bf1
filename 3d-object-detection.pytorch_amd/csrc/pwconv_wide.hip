// Materialising pointwise (1x1) forward for SHALLOW contractions with WIDE outputs on the small planes (the expansion convs of the
// 14x14 / 7x7 stages: 64 -> 384, 96 -> 576, 160 -> 960, 320 -> 1280), bf16 storage, weights in fragment order (T3D_W_FRAG), gfx950.
//
//   z = storage(act(scale * y_in + shift) + residual)   (stored: the finished block input, what t3d_bn_apply would write)
//   y[m][n] = sum_k z[m][k] * W[n][k],  sum(y), sum(y^2) per channel
//
// pwconv_stream.hip keeps a <= 96-channel weight chunk in LDS and streams the pixels past it: a 960-wide output is ten chunks, and
// EVERY chunk reads and transforms the whole operand again (and on these planes a wave gets one 32-pixel group: the launch is a
// latency chain of weight staging, one load round, the multiplies, the statistics -- 160 -> 960 @7x7 28-31 us for 28 MB).
// pwconv_deep.hip swaps the roles for long contractions but is built around K phases (tried on these shapes: slower, DESIGN finding
// 66).  Here:
//   * a workgroup (8 waves) owns 64 pixels and ALL output channels.  The operand (64 x K, K <= 320) is read, transformed and
//     stored ONCE -- to z and, in MFMA-fragment order, to LDS ([row tile][k-step][lane] x 16 B, 20 KB at K = 160);
//   * the output is walked in chunks of 256 channels; a wave owns one 32-channel tile PAIR per chunk (a lane then holds 8 consecutive
//     channels of a pixel: 16-byte stores) and streams its weight fragments from L2 straight into registers, one (chunk, phase)
//     ahead of the multiplies; no barrier after the staging -- the eight waves run through their chunks independently;
//   * statistics: per chunk the 16 pixels of a row tile meet by DPP, the four row tiles in registers, and the one lane that owns a
//     channel in the block adds the snapped fp64 partial into the replicas (as many atomics per block as the other kernels).
// No squeeze-excite gate, no bias, forward only.  OPT-IN (T3D_PW_WIDE=1; see wide_shape below for the measurements and the reason).
#include <cstdlib>
#include <type_traits>
#include "pwconv_common.h"

namespace t3d_pw {
namespace {

constexpr int WRT = 4;        // row tiles (16 pixels) per block

// KSP: k-steps per phase (the weight fragments of one phase are in registers at a time, the next phase's in flight)
// WNW: waves per block (8: one block per CU, the 7x7 planes; 4: two or three resident blocks, more pixels than one round of blocks)
template <int KSP, int WNW>
__global__ __launch_bounds__(64 * WNW, WNW == 4 ? 3 : 2) void pw_wide_kernel(const GemmArgs a, const int KS, const int nrep, const long long rstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* Af = reinterpret_cast<bf16x8*>(smem);                              // [WRT][KS][64]
  const int kpad = KS * 32;
  float* coef = reinterpret_cast<float*>(smem + (size_t)WRT * KS * 1024);    // [2][kpad] (+ [kpad] scratch for the derive)
  double* dstat = reinterpret_cast<double*>(coef + 3 * kpad);                // [2][Nout]: one owner lane per channel in the block, plain stores
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, lc = lane & 15;
  const int m0 = blockIdx.x * (16 * WRT);
  const bf16_t* __restrict__ A0 = reinterpret_cast<const bf16_t*>(a.a0);
  const bf16_t* __restrict__ ZR = reinterpret_cast<const bf16_t*>(a.z_res);
  bf16_t* __restrict__ ZO = reinterpret_cast<bf16_t*>(a.z_out);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
  const bf16x8* __restrict__ Wf = reinterpret_cast<const bf16x8*>(a.w);

  // ---- chunks of 16 tiles (256 channels); this wave: tile pair 8 c + wave of chunk c.  The first (chunk, phase)'s weight fragments are
  // requested before anything else (L2, ~2 us away)
  const int npairs = (a.Nout + 31) / 32, nch = (npairs + WNW - 1) / WNW, nph = (KS + KSP - 1) / KSP;
  const int total = nch * nph;
  bf16x8 wf[2][KSP][2];
  auto w_issue = [&](auto slot_tag, const int idx) {       // weight fragments of (chunk, phase) number idx
    constexpr int SL = decltype(slot_tag)::value;
    const int c = min(idx, total - 1) / nph, ph = min(idx, total - 1) % nph;
    const int pair = min(c * WNW + wave, npairs - 1);
#pragma unroll
    for (int u = 0; u < KSP; ++u) {
      const int ks = min(ph * KSP + u, KS - 1);
#pragma unroll
      for (int t = 0; t < 2; ++t) wf[SL][u][t] = Wf[((size_t)(2 * pair + t) * KS + ks) * 64 + lane];
    }
  };
  // ---- this wave's staging items (16 rows x 32 k): item i = wave + 8 j -> row tile i % WRT, k-step i / WRT; loads first
  constexpr int MAXIT = 5;                       // K <= 320: 4 x 10 items over 8 waves (four waves: K <= 160)
  const int nitems = WRT * KS;
  bf16x8 pa[MAXIT], pz[MAXIT];
  auto a_issue = [&]() {
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      const int i = min(wave + WNW * j, nitems - 1);
      const int r = i % WRT, ks = i / WRT;
      const int mrow = min(m0 + r * 16 + lc, a.M - 1);
      const size_t o = (size_t)mrow * a.Kin + min(ks * 32 + lg * 8, a.Kin - 8);
      pa[j] = *reinterpret_cast<const bf16x8*>(A0 + o);
      if (ZR) pz[j] = *reinterpret_cast<const bf16x8*>(ZR + o);
    }
  };
  // ---- coefficients of the operand transform (derived from the replica sums when a finalize request rides on this launch)
  auto coefs = [&]() {
    if (a.fold) {
      for (int i = a.Kin + tid; i < kpad; i += 64 * WNW) { coef[i] = 1.f; coef[kpad + i] = 0.f; }
      t3d_fold_block(a.fold, 0, a.Kin, coef, kpad, blockIdx.x == 0);
    } else {
      for (int i = tid; i < kpad; i += 64 * WNW) {
        const bool v = i < a.Kin;
        coef[i] = (v && a.p0) ? a.p0[i] : 1.f;
        coef[kpad + i] = (v && a.p0) ? a.p1[i] : 0.f;
      }
      __syncthreads();
    }
  };
  // eight waves (one block per CU): every load in flight while the coefficients are derived -- the derive's ~120 registers on top of
  // the loads' are free there; four waves (three blocks per CU at <= 168 registers): the derive first, its registers dead before the
  // loads are issued -- the other resident blocks cover the round trip
  if constexpr (WNW == 8) {
    w_issue(std::integral_constant<int, 0>{}, 0);
    a_issue();
    coefs();
  } else {
    coefs();
    __builtin_amdgcn_sched_barrier(0);
    w_issue(std::integral_constant<int, 0>{}, 0);
    a_issue();
  }
  // ---- transform, store z, stage
#pragma unroll
  for (int j = 0; j < MAXIT; ++j) {
    const int i = wave + WNW * j;
    if (i < nitems) {                           // wave-uniform
      const int r = i % WRT, ks = i / WRT;
      const int m = m0 + r * 16 + lc, k = ks * 32 + lg * 8;
      const bool ok = m < a.M && k < a.Kin;
      const int kc = min(k, kpad - 8);
      const float4 c0a = *reinterpret_cast<const float4*>(coef + kc), c0b = *reinterpret_cast<const float4*>(coef + kc + 4);
      const float4 c1a = *reinterpret_cast<const float4*>(coef + kpad + kc), c1b = *reinterpret_cast<const float4*>(coef + kpad + kc + 4);
      const float c0[8] = {c0a.x, c0a.y, c0a.z, c0a.w, c0b.x, c0b.y, c0b.z, c0b.w};
      const float c1[8] = {c1a.x, c1a.y, c1a.z, c1a.w, c1b.x, c1b.y, c1b.z, c1b.w};
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (float)pa[j][e];
      act_affine_vec<8>(x, c0, c1, a.act);
      if (ZR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += (float)pz[j][e];
      }
      bf16x8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = ok ? (bf16_t)x[e] : (bf16_t)0.f;
      if (ok && ZO) *reinterpret_cast<bf16x8*>(ZO + (size_t)m * a.Kin + k) = b;
      Af[(r * KS + ks) * 64 + lane] = b;
    }
  }
  __syncthreads();

  f32x4 acc[WRT][2];
  const bool keep_stats = a.stats != nullptr;
  auto step = [&](auto slot_tag, const int idx) {
    constexpr int SL = decltype(slot_tag)::value;
    const int c = idx / nph, ph = idx % nph;
    if (ph == 0) {
#pragma unroll
      for (int r = 0; r < WRT; ++r) acc[r][0] = acc[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    w_issue(std::integral_constant<int, 1 - SL>{}, idx + 1);       // (past the end: clamped re-reads, never used)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KSP; ++u) {
      const int ks = ph * KSP + u;
      if (ks < KS) {                             // wave-uniform (the last phase of K = 96: 3 of KSP k-steps)
        bf16x8 b[WRT];
#pragma unroll
        for (int r = 0; r < WRT; ++r) b[r] = Af[(r * KS + ks) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < WRT; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[SL][u][t], b[r], acc[r][t], 0, 0, 0);
      }
    }
    if (ph != nph - 1) return;
    // ---- epilogue of chunk c: lane holds channels 32 pair + 8 lg .. + 7 of pixel m0 + 16 r + lc (tile 2 pair: + 0..3, 2 pair + 1: + 4..7)
    const int pair = c * WNW + wave;
    if (pair >= npairs) return;                  // wave-uniform
    const int n = pair * 32 + lg * 8;
    const bool nok = n < a.Nout;                 // (whole 8-channel groups are in or out: Nout % 8 == 0)
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int r = 0; r < WRT; ++r) {
      const int m = m0 + r * 16 + lc;
      const bool ok = m < a.M && nok;
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = (bf16_t)acc[r][j >> 2][j & 3];
        const float v = ok ? (float)o[j] : 0.f;
        s1[j] += v;
        s2[j] = fmaf(v, v, s2[j]);
      }
      if (ok) *reinterpret_cast<bf16x8*>(out + (size_t)m * a.Nout + n) = o;
    }
    if (keep_stats) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t1 = row16_sum(s1[j]), t2 = row16_sum(s2[j]);
        if (lc == 0 && nok) {                    // one owner per channel in the block: plain LDS stores
          dstat[n + j] = t3d_snap(t1, a.quant, false);
          dstat[a.Nout + n + j] = t3d_snap(t2, a.quant, true);
        }
      }
    }
  };
  for (int idx = 0; idx < total; idx += 2) {
    step(std::integral_constant<int, 0>{}, idx);
    if (idx + 1 < total) step(std::integral_constant<int, 1>{}, idx + 1);
  }
  if (keep_stats) {        // whole waves of atomics (4 lanes per instruction from the chunk epilogues: 16x as many instructions, 45 us)
    __syncthreads();
    double* st = a.stats + (size_t)(blockIdx.x % nrep) * rstride;
    for (int i = tid; i < 2 * a.Nout; i += 64 * WNW) atomicAdd(st + i, dstat[i]);
  }
}

template <int KSP, int WNW>
int launch_wide(GemmArgs& a, int KS, hipStream_t st) {
  const size_t lds = (size_t)WRT * KS * 1024 + (size_t)3 * KS * 32 * 4 + (size_t)2 * a.Nout * sizeof(double);
  const void* fn = (const void*)pw_wide_kernel<KSP, WNW>;
  if (lds > 64 * 1024) (void)t3d_max_lds(fn, (int)lds);
  a.quant = (a.stats && !T3D_ENV_SET("T3D_NO_SNAP")) ? t3d_quant_for(a.M) : T3dQuant{0.0, 0.0};
  a.fold = t3d_take_fold(a.p0);
  T3D_LAUNCH_TIMED((pw_wide_kernel<KSP, WNW>), dim3(cdiv(a.M, 16 * WRT)), dim3(64 * WNW), lds, st, a, KS, g_t3d_reduce.nrep < 1 ? 1 : g_t3d_reduce.nrep,
                   g_t3d_reduce.nrep < 1 ? 0 : g_t3d_reduce.stats_stride);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

// the layers whose fragment-order weight copy this kernel wants (t3d_pwconv_wants_frag): shallow contraction, output >= 3x as wide
// OPT-IN (T3D_PW_WIDE=1).  Measured (round 6, B = 256): alone 160 -> 960 @7x7 27.6 -> 20.6 us, 320 -> 1280 46.6 -> 31.7; in the step
// 6.755-6.777 -> 6.737-6.742 ms (MobileNetV2), 6.970 -> 6.935-6.947 (MobileNetV3-large): 0.4 %, the launches' derive prologue and the
// next launch's latency chain take most of what the isolated numbers promise.  It is not the default because it regroups the
// BatchNorm partial sums (64 pixels per block instead of 32 per wave): the snapped sums differ in the last bits, the 20-step
// trajectory of tests/test_gpu_bf16_gate.py is chaotic, and its loss-difference bound (2e-2) was met at 2.16e-2 with this kernel --
// not worth re-baselining a gate for 0.03 ms.
bool wide_enabled() { return getenv("T3D_PW_WIDE") != nullptr; }      // (read per call, like T3D_DW3_TILE_MAX: the tests switch it on)
bool wide_shape(int Kin, int Nout) {
  return wide_enabled() && Kin >= 16 && Kin <= 320 && (Kin % 8) == 0 && (Nout % 8) == 0 && Nout >= 3 * Kin && Nout >= 96;
}

// bf16 materialising forward, a.w in fragment order; T3D_ERR_UNSUPPORTED = "not a launch for this kernel"
int wide_launch(GemmArgs& a, hipStream_t st) {
  if (!a.wfrag || !a.z_out || a.dgrad || a.a1 || a.a2 || a.p2 || a.per_sample || a.ps_stats || a.e_se || a.bias || a.cv.mode) return T3D_ERR_UNSUPPORTED;
  if (a.row0 && a.row0 != a.Kin) return T3D_ERR_UNSUPPORTED;
  if (!wide_shape(a.Kin, a.Nout)) return T3D_ERR_UNSUPPORTED;
  static const int max_m_env = getenv("T3D_PW_WIDE_MAX_M") ? atoi(getenv("T3D_PW_WIDE_MAX_M")) : 0;      // (sweep knob)
  const int KS = cdiv(a.Kin, 32);
  // eight waves, one workgroup per CU (220 registers x 512 threads): pays while the blocks fit the chip in ONE round -- 7x7 at B = 256
  // is 196 blocks (160 -> 960 27.6 -> 20.6 us, 320 -> 1280 46.6 -> 32.0 alone); four waves (two or three resident blocks) beyond that
  const bool one_round = cdiv(a.M, 16 * WRT) <= 256;
  // (the four-wave form on the 14x14 planes: 96 -> 576 33.9 -> 31.4 us, 64 -> 384 23.8 -> 22.4 alone, nothing in the step; 28x28 and
  // up: slower than the streaming kernel -- opt-in through T3D_PW_WIDE_MAX_M)
  if (!one_round && (KS > 5 || a.M > max_m_env)) return T3D_ERR_UNSUPPORTED;
#define T3D_WIDE(KSPV) return one_round ? launch_wide<KSPV, 8>(a, KS, st) : launch_wide<KSPV, 4>(a, KS, st)
  switch (KS) {
    case 1: T3D_WIDE(1);
    case 2: T3D_WIDE(2);
    case 3: T3D_WIDE(3);
    case 4: T3D_WIDE(4);
    default: T3D_WIDE(5);     // 5 ... 10 k-steps: phases of five
  }
#undef T3D_WIDE
}

}  // namespace t3d_pw
