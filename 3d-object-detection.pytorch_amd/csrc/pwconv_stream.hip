// Pointwise (1x1) convolution forward / data-gradient, bf16 storage: barrier-free streaming GEMM for gfx950.
//
//   out[m][n] = epilogue( sum_k  pro(A)[m][k] * W[n][k] )        m = NHWC pixel (B*H*W rows), HBM-bound
//
// The layers are HBM-bound with tiny weight matrices (<= 110 KB except at 7x7), so the kernel is built
// around ONE pass over the activations with no LDS round trip and no barrier in the main loop:
//   * the block's weight chunk W[n0 : n0+16*NT][:] is staged ONCE into LDS, already in MFMA-fragment
//     order ([tile][k-step][lane] x 16 B), so every A-operand read is one conflict-free ds_read_b128;
//   * each wave streams its own 16*R-pixel groups: the activation fragment of v_mfma_f32_16x16x32_bf16's
//     B operand (pixel = lane&15, k = 8*(lane>>4)..+8) is exactly a 16-B global load per lane, so the
//     producer's BatchNorm affine + activation (forward) or the BatchNorm-backward affine
//     dy = alpha*dz + beta*y + gamma (data gradient) is applied in registers on the way to the MFMA;
//   * the product is computed transposed (D = W * A^T) with permuted weight rows, so every lane ends up with 8
//     consecutive output channels of one pixel per 32-channel block and the four lane groups of a row write 64
//     contiguous bytes per store instruction: 16-B stores, no accumulator transpose;
//   * BatchNorm sums (sum y, sum y^2 | sum dx, sum dx*x) live in per-lane registers across the wave's
//     whole persistent loop (2 VALU ops per output element) and are reduced across lanes / waves once
//     per block: one fp64 atomic per channel per block.
// Latency is hidden by wave-level parallelism (8-16 resident waves per CU, each with its own loads in
// flight), not by a block-wide pipeline.
#include <cstdlib>
#include <type_traits>
#include "pwconv_common.h"

// Storage type of this translation unit.  The file is compiled twice: as it stands for bf16 (training + inference, every
// variant), and through pwconv_stream_f16.hip for fp16 INFERENCE (forward variants only): three more mantissa bits at every
// MFMA operand put MobileNetV2's 16-bit inference inside the 1e-3 3-D-IoU bound (tests/test_gpu_bf16_gate.py).
#ifndef T3D_PW_F16
typedef bf16_t ST;
typedef bf16x8 ST8;
#define T3D_PW_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define T3D_PW_LAUNCH stream_launch
#else
typedef f16_t ST;
typedef f16x8 ST8;
#define T3D_PW_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#define T3D_PW_LAUNCH stream_launch_f16
#endif

namespace t3d_pw {
// relu6(s x + t) = 6 clamp01((s/6) x + t/6): one v_pk_fma_f32 with the clamp modifier per channel pair (dwconv3_stream.hip has
// the note); the 6 is applied to the accumulators in the epilogue (the product is linear in the operand)
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

namespace {


// DG: data-gradient variant (two input tensors, BN-backward affine, act' epilogue); GEN: squeeze-excite / per-sample
// coefficients present (MobileNetV3 only) -- compiled out of the common variants to keep registers down.
// YF (with DG): y-free data gradient -- the main loop is the plain forward loop over two raw tensors ([dz | x], no
// transform), the epilogue is the data gradient's (activation derivative, residual, BatchNorm-backward sums).
#ifdef T3D_PW_TRACE
// debug build only (tools/pw_trace.sh): wall-clock stamps (10 ns units) of the first and the last block's wave 0
__device__ unsigned long long g_pw_trace[16];
#define PW_STAMP(i)                                                                                       \
  do {                                                                                                    \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                            \
      g_pw_trace[(blockIdx.x == 0 ? 0 : 8) + (i)] = wall_clock64();                                       \
  } while (0)
#else
#define PW_STAMP(i)
#endif

// ZM (forward only): materialising operand -- z = bf16(affine(a0) + residual) is formed on load, used as the operand and
// written out once (what a t3d_bn_apply launch in front of this kernel would have done: one launch and one pass less)
template <int NT, int R, bool DG, bool GEN, bool YF = false, int KU = 2, bool ZM = false>
__global__ __launch_bounds__(512, KU > 2 ? 1 : 2) void pw_stream_kernel(const GemmArgs a, const int nchunks, const int KS, const int nrep, const long long rstride) {
  constexpr int BN = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ST8* Wf = reinterpret_cast<ST8*>(smem);                                   // [NT][KS][64]
  const int kpad = KS * 32;
  float* coef = reinterpret_cast<float*>(smem + (size_t)NT * KS * 1024);          // [3][kpad]
  float* lstat = coef + 3 * kpad;                                                 // [BN][2]
  float* ecoef = lstat + BN * 2;                                                  // [2][BN] epilogue scale / shift
  double* dstat = reinterpret_cast<double*>(ecoef + 2 * BN);                      // [BN][2] block sums, exact adds (common.h)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, lc = lane & 15;
  // block -> (pixel-block lane xb, output chunk).  The chunks of one lane read the SAME activation rows, and consecutive
  // workgroup ids go round-robin over the 8 XCDs (each with its own L2): with chunk = id % nchunks the nchunks readers of a
  // row sat on different XCDs and every one of them fetched it from HBM (K = 960 -> 160 in three chunks: 72 MB read for a
  // 24 MB tensor, the whole launch).  When the grid is a multiple of 8 the lanes are dealt out per XCD instead, so the
  // chunk blocks of a lane share an L2.
  const int nxb = gridDim.x / nchunks;
  int chunk, xb;
  if (nchunks > 1 && (nxb & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    chunk = j % nchunks;
    xb = (j / nchunks) * 8 + xcd;
  } else {
    chunk = blockIdx.x % nchunks;
    xb = blockIdx.x / nchunks;
  }
  const int n0 = chunk * BN;
  const ST* __restrict__ A0 = reinterpret_cast<const ST*>(a.a0);
  const ST* __restrict__ A1 = reinterpret_cast<const ST*>(a.a1);
  const ST* __restrict__ A2 = reinterpret_cast<const ST*>(a.a2);
  const int ks1 = A2 ? a.ks1 : (1 << 30);   // k-steps >= ks1 read the second segment
  const ST* __restrict__ Wg = reinterpret_cast<const ST*>(a.w);
  ST* __restrict__ out = reinterpret_cast<ST*>(a.out);
  PW_STAMP(0);

  const int nthr = blockDim.x, WAVES = nthr >> 6;
  auto stage_weights = [&]() {
  // ---- stage the weight chunk in fragment order: MFMA row lc of tile t <-> n_local = (t>>1)*32 + (lc>>2)*8 + (t&1)*4 + (lc&3)
  // (eight independent loads in flight per thread, branch-free: a block of the 7x7 stage stages up to 120 KB before it
  // can start, and with one dependent load -> store round per 8 KB that prologue WAS most of those launches)
  if (a.wfrag) {
    // fragment-order weights (include/t3d.h: T3D_W_FRAG): the chunk is NT * KS KB contiguous, in exactly the order of Wf -- a
    // linear 16-B copy, fully coalesced (tiles past the matrix's last pair: zero).  Gathering the fragments out of the
    // row-major matrix (below) runs at ~20 KB/us per CU: 2-6 us of every launch, 0.13 ms of the step (DESIGN finding 34)
    constexpr int SU = 8;
    const int total = NT * KS * 64, tiles = (a.Nout + 31) / 32 * 2, t0 = n0 / 16;
    const ST8* __restrict__ P = reinterpret_cast<const ST8*>(a.w) + (size_t)t0 * KS * 64;
    const int valid = max(0, min(NT, tiles - t0)) * KS * 64;
    for (int i0 = tid; i0 < total; i0 += nthr * SU) {
      ST8 v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) v[u] = P[min(i0 + u * nthr, valid - 1)];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + u * nthr;
        if (i >= valid) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] = (ST)0.f;
        }
        if (i < total) Wf[i] = v[u];
      }
    }
  } else {
    constexpr int SU = 8;
    const int total = NT * KS * 64;
    for (int i0 = tid; i0 < total; i0 += nthr * SU) {
      ST8 v[SU];
      bool ok[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = min(i0 + u * nthr, total - 1);
        const int l = i & 63, ks = (i >> 6) % KS, t = (i >> 6) / KS;
        const int n = n0 + (t >> 1) * 32 + ((l & 15) >> 2) * 8 + (t & 1) * 4 + (l & 3), k = ks * 32 + (l >> 4) * 8;
        ok[u] = n < a.Nout && k < a.Kin;
        v[u] = *reinterpret_cast<const ST8*>(Wg + (size_t)min(n, a.Nout - 1) * a.Kin + min(k, a.Kin - 8));
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (!ok[u]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] = (ST)0.f;
        }
        if (i0 + u * nthr < total) Wf[i0 + u * nthr] = v[u];
      }
    }
  }
  for (int i = tid; i < BN * 2; i += nthr) { lstat[i] = 0.f; dstat[i] = 0.0; }
  for (int i = tid; i < BN; i += nthr) {
    const int n = n0 + i;
    const bool v = DG && a.e_scale && n < a.Nout;
    ecoef[i] = v ? a.e_scale[n] : 1.f;
    ecoef[BN + i] = v ? a.e_shift[n] : 0.f;
  }
  };
  constexpr bool DGL = DG && !YF;     // two-tensor main loop with the BatchNorm-backward affine
  if (!YF && a.fold) {
    // the BatchNorm finalize of the operand's coefficients happens HERE (common.h): scale / shift (forward) or alpha /
    // beta / gamma (data gradient) of all Kin channels from the replica sums; workgroup 0 publishes them
    for (int i = a.Kin + tid; i < kpad; i += nthr) {
      coef[i] = DG ? 0.f : 1.f; coef[kpad + i] = 0.f; coef[2 * kpad + i] = 0.f;
    }
    // (the weight staging runs between the issue of the sums' loads and their use: one round trip instead of two)
    t3d_fold_block(a.fold, 0, a.Kin, coef, kpad, blockIdx.x == 0, stage_weights);
  } else {
  stage_weights();
  for (int i = tid; i < kpad; i += nthr) {
    const bool v = i < a.Kin;
    if (YF) {
      coef[i] = 1.f; coef[kpad + i] = 0.f; coef[2 * kpad + i] = 0.f;
    } else if (!DG) {
      coef[i] = (v && a.p0) ? a.p0[i] : 1.f;
      coef[kpad + i] = (v && a.p0) ? a.p1[i] : 0.f;
    } else {
      coef[i] = (v && !a.per_sample) ? a.p0[i] : 0.f;
      coef[kpad + i] = v ? a.p1[i] : 0.f;
      coef[2 * kpad + i] = (v && !a.per_sample) ? a.p2[i] : 0.f;
    }
  }
  }
  __syncthreads();
  // forward with a plain BatchNorm + ReLU6 prologue (every projection layer): the clamp form -- the staged (scale, shift) are
  // divided by 6 (AFTER a derived finalize has published the true values) and the accumulators multiplied by 6 in the epilogue
  const bool c6f = !DG && !ZM && !YF && a.act == T3D_ACT_RELU6 && !(GEN && a.p2);
  if (c6f) {
    for (int i = tid; i < 2 * kpad; i += nthr) coef[i] *= T3D_SIXTH;
    __syncthreads();
  }
  PW_STAMP(1);

  const bool plainA = !ZM && (YF || (!DG && !a.p0 && !a.p2 && a.act == T3D_ACT_NONE));
  const bool keep_stats = a.stats != nullptr;
  float st1[NT / 2][8], st2[NT / 2][8];
#pragma unroll
  for (int q = 0; q < NT / 2; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) st1[q][j] = st2[q][j] = 0.f;

  const int ngroups = (a.M + 16 * R - 1) / (16 * R);
  const int nb = n0 + lg * 8;     // lane group lg owns channels nb + 32*q .. +7 of every 32-channel block q
  // KU = k-steps whose loads are issued together.  2 for the wide, shallow layers (K <= 160: the whole contraction is
  // one or two rounds anyway); 4 / 6 for the deep contractions of the 14x14 and 7x7 stages (K = 384 .. 960): those
  // launches have fewer pixel groups than the chip has wave slots, so a wave's own serial load -> wait -> multiply
  // rounds ARE the kernel's duration (K = 960: 15 rounds of ~1.5 us with KU = 2).

  // per-sample sums (squeeze-excite blocks): the WORKGROUP owns whole samples (its waves share a sample's pixel groups),
  // so a sample's sums are complete inside the block: reduced through `lstat` and written with plain stores.  (A wave per
  // contiguous group range + float atomics per sample change was 25-60 us of a 60-170 us launch: up to 1.5 M device-scope
  // atomics.)
  const bool ps_mode = GEN && a.ps_stats != nullptr;
  const int wv = xb * WAVES + wave, nwv = nxb * WAVES;
  const int gps = (a.HW + 16 * R - 1) / (16 * R);          // pixel groups per sample
  // ... or, on the small planes (<= 16 pixel groups per sample), the WAVE owns whole samples (round 6): its running sums are
  // complete when it has walked the sample's groups -- no block-level exchange, no barrier per sample (at 7x7 a workgroup's four
  // waves met twice per sample for ONE pixel group each: 672 <- 112 @14x14 69 us against 49 for the plain data gradient)
  const bool wos = ps_mode && a.ps_wave;
  const int s_end = ps_mode ? a.M / a.HW : 1, s_step = wos ? nwv : (ps_mode ? nxb : 1);
  for (int sb = wos ? wv : (ps_mode ? xb : 0); sb < s_end; sb += s_step) {
  const int g_begin = wos ? 0 : (ps_mode ? wave : wv), g_end = ps_mode ? gps : ngroups, g_step = wos ? 1 : (ps_mode ? WAVES : nwv);
  const int mbase = ps_mode ? sb * a.HW : 0, mlim = ps_mode ? min(a.M, (sb + 1) * a.HW) : a.M;
  for (int g = g_begin; g < g_end; g += g_step) {
    const int m0 = mbase + g * 16 * R;
    f32x4 acc[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int mrow[R], mld[R];
    bool mok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mrow[r] = m0 + r * 16 + lc;
      mok[r] = mrow[r] < mlim;            // (ps_mode: rows of the next sample belong to another group)
      mld[r] = min(mrow[r], a.M - 1);     // rows past the end re-read a valid row; they are never stored or summed
    }

    // HOIST (the deep-round data-gradient variants, KU > 2): the epilogue's activation-input tensor is loaded here, ahead
    // of the k rounds -- those launches walk several pixel groups per wave with two resident waves per SIMD, and every
    // dependent load round (operands | epilogue tensor) was ~1.5 us of exposed latency per group.  Dispatch guarantees
    // e_y != nullptr and e_res == nullptr for them.
    constexpr bool HOIST = DG && !YF && !GEN && KU > 2;
    ST8 eyh[R][NT / 2];
    if constexpr (HOIST) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < NT / 2; ++q)
          eyh[r][q] = *reinterpret_cast<const ST8*>(reinterpret_cast<const ST*>(a.e_y) + (size_t)mld[r] * a.Nout +
                                                       min(nb + 32 * q, a.Nout - 8));
    }
    // one round = KUr k-steps: all their loads, then their multiplies.  GUARD: the round may run past KS (tail rounds);
    // the unguarded full rounds of the deep variants are straight-line code -- with the wave-uniform guards the compiler
    // places s_waitcnt inside the load sequence at every CFG join and the round's loads no longer overlap
    auto k_round = [&](auto ku_tag, auto guard_tag, const int ks0) {
      constexpr int KUr = decltype(ku_tag)::value;
      constexpr bool GUARD = decltype(guard_tag)::value;
      ST8 fa[KUr][R], fb[KUr][R];
      ST8 fz[ZM ? KUr : 1][R];      // ZM: residual fragments, fetched with the operand
      // squeeze-excite gates of the forward operand (per sample x input channel, fp32): fetched WITH the operand -- loaded
      // inside the transform they were one more dependent global round trip per k-step
      float4 gs[GEN && !DGL ? KUr : 1][R][2];
      const bool gated = GEN && !DGL && a.p2 != nullptr;
      // branch-free loads: k-steps / channels past Kin read a clamped (valid) address -- their weights are zero
#pragma unroll
      for (int u = 0; u < KUr; ++u) {
        const int ks = ks0 + u;
        if (GUARD && ks >= KS) continue;          // wave-uniform (K <= 32: one k-step, no second load)
        if (YF && ks >= ks1) {         // wave-uniform: second segment (y-free data gradient)
          const int k = min((ks - ks1) * 32 + lg * 8, a.Kin2 - 8);
#pragma unroll
          for (int r = 0; r < R; ++r) fa[u][r] = *reinterpret_cast<const ST8*>(A2 + (size_t)mld[r] * a.Kin2 + k);
        } else {
          const int k = min(ks * 32 + lg * 8, a.row0 - 8);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            {
              fa[u][r] = *reinterpret_cast<const ST8*>(A0 + (size_t)mld[r] * a.row0 + k);
              if (DGL) fb[u][r] = *reinterpret_cast<const ST8*>(A1 + (size_t)mld[r] * a.row0 + k);
              if constexpr (ZM) {
                if (a.z_res)
                  fz[u][r] = *reinterpret_cast<const ST8*>(reinterpret_cast<const ST*>(a.z_res) + (size_t)mld[r] * a.row0 + k);
              }
              if constexpr (GEN && !DGL) {
                if (gated) {
                  const float* gp = a.p2 + (size_t)(mld[r] / a.HW) * a.Kin + min(ks * 32 + lg * 8, a.Kin - 8);
                  gs[u][r][0] = *reinterpret_cast<const float4*>(gp);
                  gs[u][r][1] = *reinterpret_cast<const float4*>(gp + 4);
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < KUr; ++u) {
        const int ks = ks0 + u;
        if (!GUARD || ks < KS) {
          const int k = ks * 32 + lg * 8;
          ST8 b[R];
          if (plainA) {
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = fa[u][r];
          } else {
            const float4 c0a = *reinterpret_cast<const float4*>(coef + k), c0b = *reinterpret_cast<const float4*>(coef + k + 4);
            const float4 c1a = *reinterpret_cast<const float4*>(coef + kpad + k),
                         c1b = *reinterpret_cast<const float4*>(coef + kpad + k + 4);
            const float c0[8] = {c0a.x, c0a.y, c0a.z, c0a.w, c0b.x, c0b.y, c0b.z, c0b.w};
            const float c1[8] = {c1a.x, c1a.y, c1a.z, c1a.w, c1b.x, c1b.y, c1b.z, c1b.w};
            if (!DGL) {
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const bool ok = mok[r] && (k < a.Kin);
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = (float)fa[u][r][j];
                if (GEN && a.p2) {
                  const float4 g0 = gs[GEN && !DGL ? u : 0][r][0], g1 = gs[GEN && !DGL ? u : 0][r][1];
                  const float sg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                  float sv[8];
#pragma unroll
                  for (int j = 0; j < 8; ++j) sv[j] = ok ? sg[j] : 1.f;
                  // one activation switch per 8 elements (act_affine_vec), not one per element: gate before the
                  // activation = act(sv * (c0 x + c1)) folds into the affine, gate after it multiplies the result
                  if (!a.se_after) {
                    float c0s[8], c1s[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { c0s[j] = c0[j] * sv[j]; c1s[j] = c1[j] * sv[j]; }
                    act_affine_vec<8>(x, c0s, c1s, a.act);
                  } else {
                    act_affine_vec<8>(x, c0, c1, a.act);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] *= sv[j];
                  }
                } else if (c6f) {
#pragma unroll
                  for (int j = 0; j < 8; j += 2) {
                    const f32x2 t = pk_fma_clamp01(f32x2{x[j], x[j + 1]}, f32x2{c0[j], c0[j + 1]}, f32x2{c1[j], c1[j + 1]});
                    x[j] = t[0];
                    x[j + 1] = t[1];
                  }
                } else {
                  act_affine_vec<8>(x, c0, c1, a.act);
                }
                if constexpr (ZM) {
                  if (a.z_res) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] += (float)fz[u][r][j];
                  }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) b[r][j] = (ST)x[j];
                if constexpr (ZM) {
                  if (chunk == 0 && ok)
                    *reinterpret_cast<ST8*>(reinterpret_cast<ST*>(a.z_out) + (size_t)mrow[r] * a.row0 + k) = b[r];
                }
              }
            } else {
              const float4 c2a = *reinterpret_cast<const float4*>(coef + 2 * kpad + k),
                           c2b = *reinterpret_cast<const float4*>(coef + 2 * kpad + k + 4);
              const float c2[8] = {c2a.x, c2a.y, c2a.z, c2a.w, c2b.x, c2b.y, c2b.z, c2b.w};
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const bool ok = mok[r] && (k < a.Kin);
                const size_t pb = (GEN && a.per_sample && ok) ? (size_t)(mrow[r] / a.HW) * a.Kin + k : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const float al = (GEN && a.per_sample) ? (ok ? a.p0[pb + j] : 0.f) : c0[j];
                  const float ga = (GEN && a.per_sample) ? (ok ? a.p2[pb + j] : 0.f) : c2[j];
                  b[r][j] = (ST)(al * (float)fa[u][r][j] + c1[j] * (float)fb[u][r][j] + ga);
                }
              }
            }
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const ST8 wf = Wf[(t * KS + ks) * 64 + lane];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][t] = T3D_PW_MFMA(wf, b[r], acc[r][t], 0, 0, 0);
          }
        }
      }
    };
    int ks0 = 0;
    if constexpr (KU > 2)
      for (; ks0 + KU <= KS; ks0 += KU) k_round(std::integral_constant<int, KU>{}, std::false_type{}, ks0);
    for (; ks0 < KS; ks0 += 2) k_round(std::integral_constant<int, 2>{}, std::true_type{}, ks0);


    if (g == g_begin) PW_STAMP(2);
    // ---------------- epilogue: lane holds channels nb .. nb+4*NT-1 of pixel mrow[r] ----------
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = mrow[r];
      const bool ok = mok[r];
      const int mrow0 = m0 + r * 16, mlast = mrow0 + 15;
      const int bidx = ok ? m / a.HW : 0;
      // all epilogue loads of the row first: the stores below then issue back to back and the 16-B pieces of a line
      // meet in L2 (interleaved with load waits they were written back separately: 1.84x HBM write traffic, PMC)
      ST8 eyr[NT / 2], err[NT / 2];
      float4 esg[GEN ? NT / 2 : 1][2];       // squeeze-excite gates of the epilogue tensor (per sample x channel)
      if constexpr (GEN && DG) {
        if (a.e_se) {
#pragma unroll
          for (int q = 0; q < NT / 2; ++q) {
            const float* gp = a.e_se + (size_t)(mld[r] / a.HW) * a.Nout + min(nb + 32 * q, a.Nout - 8);
            esg[q][0] = *reinterpret_cast<const float4*>(gp);
            esg[q][1] = *reinterpret_cast<const float4*>(gp + 4);
          }
        }
      }
      if constexpr (HOIST) {
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) eyr[q] = eyh[r][q];
      } else if (DG) {
        if (a.e_y) {
#pragma unroll
          for (int q = 0; q < NT / 2; ++q)
            eyr[q] = *reinterpret_cast<const ST8*>(reinterpret_cast<const ST*>(a.e_y) + (size_t)mld[r] * a.Nout +
                                                      min(nb + 32 * q, a.Nout - 8));
        }
        if (a.e_res) {
#pragma unroll
          for (int q = 0; q < NT / 2; ++q)
            err[q] = *reinterpret_cast<const ST8*>(reinterpret_cast<const ST*>(a.e_res) + (size_t)mld[r] * a.Nout +
                                                      min(nb + 32 * q, a.Nout - 8));
        }
      }
#pragma unroll
      for (int q = 0; q < NT / 2; ++q) {
        const int n = nb + 32 * q;
        if (n >= a.Nout) continue;  // whole 8-channel groups are in or out (Nout % 8 == 0)
        float v[8], yv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = acc[r][2 * q + (j >> 2)][j & 3];
        if (c6f) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= 6.f;
        }
        if ((!DG || YF) && a.bias) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += a.bias[n + j];
        }
        if (DG && a.e_y) {
#pragma unroll
          for (int j = 0; j < 8; ++j) yv[j] = (float)eyr[q][j];
          if (GEN && a.e_se) {
            const float4 s0 = *reinterpret_cast<const float4*>(ecoef + (n - n0)), s1 = *reinterpret_cast<const float4*>(ecoef + (n - n0) + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(ecoef + BN + (n - n0)),
                         h1 = *reinterpret_cast<const float4*>(ecoef + BN + (n - n0) + 4);
            float es[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            float eh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            const float4 g0 = esg[GEN ? q : 0][0], g1 = esg[GEN ? q : 0][1];
            const float sv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            // one derivative switch per 8 elements: gate before the activation folds into the affine (act'(sv (es y + eh))),
            // gate after it scales the result
            if (!a.e_se_after) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { es[j] *= sv[j]; eh[j] *= sv[j]; }
              act_grad_affine_vec<8>(v, yv, es, eh, a.e_act);
            } else {
              act_grad_affine_vec<8>(v, yv, es, eh, a.e_act);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= sv[j];
            }
          } else if (a.e_act != T3D_ACT_NONE) {
            const float4 s0 = *reinterpret_cast<const float4*>(ecoef + (n - n0)), s1 = *reinterpret_cast<const float4*>(ecoef + (n - n0) + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(ecoef + BN + (n - n0)),
                         h1 = *reinterpret_cast<const float4*>(ecoef + BN + (n - n0) + 4);
            const float es[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float eh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            act_grad_affine_vec<8>(v, yv, es, eh, a.e_act);
          }
        }
        if (a.e_res) {
          if constexpr (DG) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)err[q][j];
          } else {
            float rr[8];
            Vec8<ST>::load(reinterpret_cast<const ST*>(a.e_res) + (size_t)mld[r] * a.Nout + n, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rr[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ok ? Vec8<ST>::round(v[j]) : 0.f;
        if (ok && out) Vec8<ST>::store(out + (size_t)m * a.Nout + n, v);   // (out == null: statistics-only pass)
        if (keep_stats) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            st1[q][j] += v[j];
            st2[q][j] = fmaf(v[j], (DG && a.e_y) ? yv[j] : v[j], st2[q][j]);
          }
        } else if (GEN && a.ps_stats) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {                    // v is 0 for rows outside the sample
            st1[q][j] += v[j];
            st2[q][j] = fmaf(v[j], yv[j], st2[q][j]);
          }
        }
      }
    }
  }
  if (wos) {          // sample sb is complete in this wave: lanes of a channel meet by DPP, plain stores
#pragma unroll
    for (int q = 0; q < NT / 2; ++q) {
      const int n = nb + 32 * q;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float s1 = row16_sum(st1[q][j]), s2 = row16_sum(st2[q][j]);
        st1[q][j] = st2[q][j] = 0.f;
        if (lc == 0 && n < a.Nout) {
          a.ps_stats[((size_t)sb * a.Nout + n + j) * 2] = s1;
          a.ps_stats[((size_t)sb * a.Nout + n + j) * 2 + 1] = s2;
        }
      }
    }
  } else if (ps_mode) {      // sample sb is complete in this block
#pragma unroll
    for (int q = 0; q < NT / 2; ++q) {
      const int n = nb + 32 * q;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float s1 = row16_sum(st1[q][j]), s2 = row16_sum(st2[q][j]);
        st1[q][j] = st2[q][j] = 0.f;
        if (lc == 0 && n < a.Nout && g_begin < g_end) {      // (fp64: the waves' fp32 partials add exactly, in any order)
          atomicAdd(dstat + (n - n0 + j) * 2, (double)s1);
          atomicAdd(dstat + (n - n0 + j) * 2 + 1, (double)s2);
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < BN * 2; i += nthr) {
      const int n = n0 + (i >> 1);
      if (n < a.Nout) a.ps_stats[((size_t)sb * a.Nout + n) * 2 + (i & 1)] = (float)dstat[i];
      dstat[i] = 0.0;
    }
    __syncthreads();
  }
  }   // sample loop (one pass when there are no per-sample sums)
  PW_STAMP(3);

  if (keep_stats) {
#pragma unroll
    for (int q = 0; q < NT / 2; ++q) {
      const int n = nb + 32 * q;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float s1 = row16_sum(st1[q][j]), s2 = row16_sum(st2[q][j]);
        if (lc == 0 && n < a.Nout) {
          atomicAdd(dstat + (n - n0 + j) * 2, t3d_snap(s1, a.quant, false));
          atomicAdd(dstat + (n - n0 + j) * 2 + 1, t3d_snap(s2, a.quant, !DG));
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < BN * 2; i += nthr) {
      const int n = n0 + (i >> 1);
      if (n < a.Nout) atomicAdd(a.stats + (size_t)(xb % nrep) * rstride + (size_t)(i & 1) * a.Nout + n, dstat[i]);
    }
  }
  PW_STAMP(4);
}

template <int NT, int R, bool DG, bool GEN, bool YF = false, int KU = 2, bool ZM = false>
int launch_v(GemmArgs& a, int KS, hipStream_t st) {
  constexpr int BN = NT * 16;
  const int kpad = KS * 32;
  const size_t lds = (size_t)NT * KS * 1024 + (size_t)3 * kpad * 4 + BN * 4 * 4 + BN * 2 * 8;
  if (lds > 150 * 1024) return T3D_ERR_UNSUPPORTED;
  const int nchunks = cdiv(a.Nout, BN);
  const int ngroups = cdiv(a.M, 16 * R);
  // small weight chunks: 4-wave blocks, as many per CU as registers / LDS admit (each wave hides its own
  // load latency, so resident waves per CU are what matters); big chunks: one 8-wave block shares the copy
  const bool ps = GEN && a.ps_stats != nullptr;          // per-sample sums: a block owns whole samples (kernel)
  const int gps = cdiv(a.HW, 16 * R);
  const int threads = (lds <= 48 * 1024 || (ps && gps <= 4)) ? 256 : 512;
  const void* fn = (const void*)pw_stream_kernel<NT, R, DG, GEN, YF, KU, ZM>;
  if (lds > 64 * 1024) (void)t3d_max_lds(fn, (int)lds);
  static int occ_cache[2] = {0, 0};   // per instantiation (function-local static of the template), per block size
  int& occ = occ_cache[threads == 512];
  if (occ == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, 0) != hipSuccess || n < 1) n = 1;
    occ = n;
  }
  int per_cu = occ;
  const int by_lds = (int)(160 * 1024 / (lds + 512));
  if (per_cu > by_lds) per_cu = by_lds;
  if (per_cu < 1) per_cu = 1;
  int nxb = (256 * per_cu) / nchunks;
  a.ps_wave = (ps && gps <= 16 && !T3D_ENV_SET("T3D_PW_PS_BLOCK")) ? 1 : 0;
  const int need = ps ? (a.ps_wave ? cdiv(a.M / a.HW, threads / 64) : a.M / a.HW) : cdiv(ngroups, threads / 64);
  if (nxb > need) nxb = need;
  if (nxb < 1) nxb = 1;
  if (nchunks > 1 && nxb >= 8) {              // whole lanes per XCD (see the kernel's block mapping)
    const int up = (nxb + 7) & ~7;            // round up while the launch still fits the chip in one wave of blocks
    nxb = (up * nchunks <= 256 * per_cu) ? up : (nxb & ~7);
  }
  // a pending BatchNorm-finalize request belongs to this launch when it names the coefficients of its operand
  // (per-sample coefficients and the y-free variant have none to derive)
  a.quant = (!DG && a.stats && !T3D_ENV_SET("T3D_NO_SNAP")) ? t3d_quant_for(a.M) : T3dQuant{0.0, 0.0};
  if (YF || a.per_sample) {
    if (const int rc = t3d_fold_fallback(a.p0, st)) return rc;
    a.fold = nullptr;
  } else {
    a.fold = t3d_take_fold(a.p0);
  }
  T3D_LAUNCH_TIMED((pw_stream_kernel<NT, R, DG, GEN, YF, KU, ZM>), dim3(nxb * nchunks), dim3(threads), lds, st, a, nchunks, KS,
                     g_t3d_reduce.nrep, g_t3d_reduce.stats_stride);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <int NT, int R>
int launch_nt(GemmArgs& a, int KS, hipStream_t st, int deep_ku = 0) {
#ifdef T3D_PW_F16
  // fp16 storage: the plain inference forward only (BatchNorm + activation prologue, no gates, no materialising operand)
  if (a.a2 || a.dgrad || a.z_out || a.per_sample || a.ps_stats || a.e_se || a.p2 || a.stats) return T3D_ERR_UNSUPPORTED;
  return launch_v<NT, R, false, false>(a, KS, st);
#else
  if (a.a2) return launch_v<NT, R, true, false, true>(a, KS, st);   // y-free data gradient
  const bool gen = a.per_sample || a.ps_stats || a.e_se || (!a.dgrad && a.p2);
  // (KU = 4 / 6 / 8 variants for the deep contractions of the 14x14 / 7x7 stages were measured, the last with straight-line
  // rounds (16 loads in flight, vmcnt counting down): K = 960 -> 160 k-loop 21 -> 18 us of a 33 us launch.  The in-kernel
  // timeline (tools/pw_trace.sh) shows why: per round the operand transform (unpack, BatchNorm affine, activation, pack: ~4.5
  // VALU ops per element, redone per output chunk) costs as much as the load latency, and with 2 waves per SIMD the two do
  // not overlap; staging 120 KB of weights (5.7 us) and the statistics tail (3 us) are the rest.)
  if (a.dgrad && !gen && deep_ku == 3) return launch_v<NT, R, true, false, false, 3>(a, KS, st);
  if (a.dgrad && !gen && deep_ku == 5) return launch_v<NT, R, true, false, false, 5>(a, KS, st);
  if (a.dgrad) return gen ? launch_v<NT, R, true, true>(a, KS, st) : launch_v<NT, R, true, false>(a, KS, st);
  if (a.z_out) return gen ? T3D_ERR_UNSUPPORTED : launch_v<NT, R, false, false, false, 2, true>(a, KS, st);
  return gen ? launch_v<NT, R, false, true>(a, KS, st) : launch_v<NT, R, false, false>(a, KS, st);
#endif
}

}  // namespace

#if defined(T3D_PW_TRACE) && !defined(T3D_PW_F16)
extern "C" int t3d_debug_pw_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pw_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

int T3D_PW_LAUNCH(GemmArgs& a, hipStream_t st) {
  if (!a.row0) a.row0 = a.Kin;

  if (a.ps_stats && (a.stats || a.M % a.HW)) return T3D_ERR_UNSUPPORTED;   // the block-level per-sample reduction uses the statistics scratch
  const int KS = cdiv(a.Kin, 32);
  // widest chunk whose weights fit ~120 KB of LDS, at most 10 tiles (register budget: 8*NT stat + 4*NT*R acc)
  int nt_cap = (120 * 1024 / 1024) / KS;
  if (nt_cap > 10) nt_cap = 10;
  // narrow contraction (K <= 64): re-reading the activation per output chunk costs almost nothing, so trade chunks for
  // registers / occupancy
  const int small_k_cap = 10;
  if (a.Kin <= 64 && nt_cap > small_k_cap) nt_cap = small_k_cap;
  // squeeze-excite data gradient (per-sample sums / gates in registers): wider tiles spill (NT = 8: 216 B, 10: 412 B)
  const int gen_cap = 6;
  if (a.dgrad && (a.per_sample || a.ps_stats || a.e_se) && nt_cap > gen_cap) nt_cap = gen_cap;
  // small-stage data gradients of the projection convs (contraction over 96 / 160 / 320 bottleneck channels, <= 28x28
  // pixels): all k-steps of a round in flight + hoisted epilogue loads (kernel: HOIST), at a tile width that leaves the
  // registers for it.  OPT-IN (T3D_PW_DEEP_DG=1): the launches themselves get 3-8 % faster alone (576 <- 96 @14x14: 46.2 ->
  // 45.0 us, 960 <- 160 @7x7: 31.7 -> 29.1), but the STEP gets slower (8.46 -> 8.54 ms, three A/B pairs): it is bound by the
  // two streams' combined HBM traffic, and what these launches stop waiting for, the depthwise backward beside them loses
  const int deep_dg = 0;
  int deep_ku = 0;
  if (deep_dg && a.dgrad && !a.a2 && !(a.per_sample || a.ps_stats || a.e_se) && a.e_y && !a.e_res && a.M <= 256 * 28 * 28 &&
      (KS == 3 || KS % 5 == 0) && a.Nout >= 96) {
    deep_ku = KS == 3 ? 3 : 5;
    if (nt_cap > 6) nt_cap = 6;
  }
  nt_cap &= ~1;
  if (nt_cap < 2) return T3D_ERR_UNSUPPORTED;
  int NT = 2;
  {
    // fewest chunks first, then least padding
    int best_chunks = 1 << 30, best_pad = 1 << 30;
    for (int nt = 2; nt <= nt_cap; nt += 2) {
      const int ch = cdiv(a.Nout, nt * 16), pad = ch * nt * 16 - a.Nout;
      if (ch < best_chunks || (ch == best_chunks && pad < best_pad)) { best_chunks = ch; best_pad = pad; NT = nt; }
    }
  }
  // wide outputs that are a whole number of 96-channel tiles (384, 576, 960 of the 14x14 / 7x7 stages) and need several chunks
  // anyway: the 96-channel tile with two pixel groups per iteration beats fewer, wider chunks with one (round 4: 96 -> 576 31.6 ->
  // 28.7 us, 160 -> 960 23.4 -> 21.6, 64 -> 384 22.4 -> 20.8; 144 channels in ONE 160-wide chunk stays: 72 vs 79 us)
  if (NT > 6 && nt_cap >= 6 && cdiv(a.Nout, NT * 16) > 1 && a.Nout % 96 == 0 &&
      !(a.per_sample || a.ps_stats || a.e_se || (!a.dgrad && a.p2)))
    NT = 6;
  switch (NT) {
    case 2: return launch_nt<2, 2>(a, KS, st, deep_ku);
    case 4: return launch_nt<4, 2>(a, KS, st, deep_ku);     // (R = 4 for NT = 2 / 4: 5-40 % slower, measured)
    case 6:
      // two 16-pixel groups per iteration (round 4): twice the loads in flight per wave and two MFMAs per weight fragment read --
      // 16 -> 96 @112x112 216 -> 209 us, 96 <- 24 390 -> 366, 576 -> 96 @14x14 30.0 -> 26.7, step 6.997 -> 6.904 ms (three A/B
      // pairs).  Not for the squeeze-excite / per-sample variants: their extra registers spill at R = 2 (72 VGPRs to scratch)
      if (a.per_sample || a.ps_stats || a.e_se || (!a.dgrad && a.p2)) return launch_nt<6, 1>(a, KS, st, deep_ku);
      return launch_nt<6, 2>(a, KS, st, deep_ku);
    case 8: return launch_nt<8, 1>(a, KS, st, deep_ku);     // (R = 2: forward 64 -> 384 @14x14 -8 %, data gradient spills 32 VGPRs)
    default: return launch_nt<10, 1>(a, KS, st, deep_ku);
  }
}

}  // namespace t3d_pw
