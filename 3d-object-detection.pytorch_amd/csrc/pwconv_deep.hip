// Pointwise (1x1) convolution forward / data-gradient for DEEP contractions (K >= 512 with an output wider than the streaming
// kernel's LDS weight chunk), bf16 storage, weights in FRAGMENT ORDER (t3d_pwconv_pack_frag, T3D_W_FRAG), gfx950.
//
//   out[m][n] = epilogue( sum_k  pro(A)[m][k] * W[n][k] )
//
// pwconv_stream.hip keeps a weight chunk W[n0 : n0+16 NT][all K] in LDS and streams the activations past it.  At K = 960 /
// 1280 (the 7x7 stage: projection forward, expansion data gradient, the last conv's data gradient) 120 KB of LDS hold only
// 64 / 32 output channels, so the launch splits into 3 - 10 output chunks and every chunk reads AND TRANSFORMS the whole
// operand again (BatchNorm affine + activation, or the two-tensor BatchNorm-backward affine): 320 <- 1280 at 7x7 ran 126 us
// for 10 GFLOP.  Here the roles are swapped:
//   * a workgroup (8 waves, two per SIMD -- round 5: the weight stream of ONE wave per SIMD with the whole register file, a phase
//     of 8 k-steps x 3 tiles ahead, ran at ~25 GB/s per CU; eight waves with 4 k-steps x 2 tiles ahead each: 960 -> 160 @7x7
//     22.6 -> 18.3 us, 320 <- 1280 83.8 -> 59.4, ResNet-50's 1024 <- 2048 @7x7 111 -> 45) owns 64 pixels and all output channels
//     of its chunk (<= 256); the operand is read and transformed ONCE, in phases of 4 k-steps, into a double-buffered LDS tile in
//     MFMA-fragment order ([row tile][k-step][lane] x 16 B: linear, conflict-free writes and reads);
//   * the WEIGHTS are streamed from L2 straight into registers, a whole phase ahead.  They have to be in fragment order in
//     memory ([16-row tile][k-step][lane] x 16 B, rows in the streaming kernel's pair permutation, zero padded: include/t3d.h):
//     a wave's fragment is then 1 KB contiguous.  Loading
//     fragments out of the row-major matrix (16 rows x 64 B per instruction, one 16-B request per lane) ran at ~20 KB/us per
//     CU -- the same rate the streaming kernel's weight staging shows -- and WAS the kernel (k-loop 22 us of 36);
//   * loads return IN ORDER per wave (vmcnt), so a weight fragment (L2 hit) issued behind an operand row (HBM) is not usable
//     before that row has arrived.  Issue order per phase p: the weights of the WHOLE phase p + 1, then the operand rows of
//     phase p + 2 -- a multiply never waits on anything younger than its own weights, and both kinds of loads have a full
//     phase / two phases (64 - 128 KB per workgroup in flight) to arrive;
//   * everything inside the phase loop is branch-free (k-steps / tiles past the end are zero fragments or repeats against
//     clamped loads), so the compiler counts vmcnt down instead of draining it at control-flow joins;
//   * epilogue and statistics as in the streaming kernel (a lane holds channels 32 (T >> 1) + 8 lg + 4 (T & 1) .. +3 of one
//     pixel for tile T; sums per
//     lane in registers, one owner per channel in the block, one fp64 atomic per channel per block).
// No squeeze-excite / per-sample coefficients, no bias, no materialising operand: those shapes stay with the streaming kernel.
#include <cstdlib>
#include <type_traits>
#include "pwconv_common.h"

namespace t3d_pw {
namespace {

__device__ __forceinline__ f32x2 deep_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

#ifdef T3D_PW_TRACE
// debug build only (tools/pw_trace.sh): wall-clock stamps (10 ns units) of the first and the last block's wave 0
__device__ unsigned long long g_deep_trace[16];
#define DEEP_STAMP(i)                                                                                     \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))         \
      g_deep_trace[(blockIdx.x == 0 ? 0 : 8) + (i)] = wall_clock64();                                     \
  } while (0)
#else
#define DEEP_STAMP(i)
#endif

constexpr int RT = 4;        // row tiles (16 pixels) per block
constexpr int NW = 8;        // waves per block: two per SIMD (round 5; one per SIMD with the whole register file before)

// CV: implicit 3x3 convolution (GemmArgs::cv) -- a compile-time variant: the plain kernel's phase loop stays branch-free
template <bool DG, bool CV, int NTW, int KSP>
__global__ __launch_bounds__(64 * NW) void pw_deep_kernel(const GemmArgs a, const int KS, const int ntiles, const int nrep,
                                                             const long long rstride) {
  // NTW 16-row tiles per wave (tiles wave, wave + 8, .. of the chunk: <= 8 NTW tiles), KSP k-steps per phase (one LDS buffer = RT KSP KB);
  // IT staging items (16 rows x 32 k) per wave per phase: row tile wave % RT, k-steps (wave / RT) * IT .. + IT - 1
  constexpr int IT = RT * KSP / NW;
  static_assert(IT >= 1 && IT * NW == RT * KSP, "staging items must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* Af = reinterpret_cast<bf16x8*>(smem);                                   // [2][RT][KSP][64]
  const int kpad = KS * 32;
  float* coef = reinterpret_cast<float*>(smem + (size_t)2 * RT * KSP * 1024);     // [3][kpad]
  const int BN = ntiles * 16;
  float* ecoef = coef + 3 * kpad;                                                 // [2][BN]
  double* dstat = reinterpret_cast<double*>(ecoef + 2 * BN);                      // [BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, lc = lane & 15;
  const int m0 = blockIdx.x * (16 * RT), n0 = blockIdx.y * BN;
  const bf16_t* __restrict__ A0 = reinterpret_cast<const bf16_t*>(a.a0);
  const bf16_t* __restrict__ A1 = reinterpret_cast<const bf16_t*>(a.a1);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
  const int nph = (KS + KSP - 1) / KSP;
  DEEP_STAMP(0);

  // the wave's tiles; a slot past the chunk's (or the layer's) last tile repeats it (same instructions in every wave, the
  // repeats' results are dropped)
  int tix[NTW];
  bool tok[NTW];
  const int tlast = min(ntiles, (a.Nout - n0 + 31) / 32 * 2) - 1;      // (whole pairs: a layer's last pair may be half padding)
  const bf16x8* wt_[NTW];        // fragment-order weights of the tile: [k-step][lane]
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int t = wave + NW * i;
    tok[i] = t <= tlast;
    tix[i] = min(t, tlast);
    wt_[i] = reinterpret_cast<const bf16x8*>(a.w) + ((size_t)(n0 / 16 + tix[i]) * KS) * 64 + lane;
  }
  bf16x8 wf[2][KSP][NTW];
  auto w_issue = [&](auto slot_tag, const int ph) {        // weight fragments of phase ph
    constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
    for (int u = 0; u < KSP; ++u) {
      const int ks = min(ph * KSP + u, KS - 1);
#pragma unroll
      for (int t = 0; t < NTW; ++t) wf[SL][u][t] = wt_[t][(size_t)ks * 64];
    }
  };
  // operand rows of this wave's staging items
  const int srt = wave % RT, sks = (wave / RT) * IT;      // this wave's staging items: row tile srt, k-steps sks .. sks + IT - 1 of a phase
  const int mrow = min(m0 + srt * 16 + lc, a.M - 1);
  const size_t arow = (size_t)mrow * a.Kin + lg * 8;
  const bool rok = m0 + srt * 16 + lc < a.M;
  // implicit 3x3 convolution (pwconv_common.h: Conv3): this lane's destination pixel, once
  constexpr bool cv = CV;
  int cvb = 0, cvy = 0, cvx = 0;
  if constexpr (cv) {
    cvx = mrow % a.cv.Dw;
    const int t = mrow / a.cv.Dw;
    cvy = t % a.cv.Dh;
    cvb = t / a.cv.Dh;
  }
  // k-step ks (wave-uniform) -> element offset of the lane's 8 operand channels in the source tensor, or -1: the tap falls
  // outside the source (zero after the transform)
  auto cv_off = [&](const int ks) -> long long {
    const int k0 = min(ks * 32, a.Kin - 32);
    const int tap = k0 >> a.cv.lgCs, c = k0 & (a.cv.Cs - 1);
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;             // tap / 3, tap % 3 for tap <= 8
    int sy, sx;
    bool ok;
    if (a.cv.mode == 1) {
      sy = cvy * a.cv.stride - 1 + ky;
      sx = cvx * a.cv.stride - 1 + kx;
      ok = true;
    } else {
      const int ty = cvy + 1 - ky, tx = cvx + 1 - kx, sm = a.cv.stride - 1;       // stride 1 or 2
      ok = ((ty | tx) & sm) == 0;
      sy = ty >> sm;
      sx = tx >> sm;
    }
    ok = ok && (unsigned)sy < (unsigned)a.cv.Sh && (unsigned)sx < (unsigned)a.cv.Sw;
    return ok ? ((long long)(cvb * a.cv.Sh + sy) * a.cv.Sw + sx) * a.cv.Cs + c + lg * 8 : -1;
  };
  bf16x8 pa[2][IT], pb[DG ? 2 : 1][DG ? IT : 1];
  auto a_issue = [&](auto slot_tag, const int ph) {
    constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      size_t o;
      if constexpr (cv) {
        const long long g = cv_off(ph * KSP + sks + j);
        o = g < 0 ? (size_t)lg * 8 : (size_t)g;                 // (an out-of-range tap reads a valid address; zeroed later)
      } else {
        o = arow + min((ph * KSP + sks + j) * 32, a.Kin - 8 - lg * 8);
      }
      pa[SL][j] = *reinterpret_cast<const bf16x8*>(A0 + o);
      if (DG) pb[DG ? SL : 0][DG ? j : 0] = *reinterpret_cast<const bf16x8*>(A1 + o);
    }
  };
  w_issue(std::integral_constant<int, 0>{}, 0);
  a_issue(std::integral_constant<int, 0>{}, 0);
  a_issue(std::integral_constant<int, 1>{}, 1);

  // ---- coefficients of the operand transform (derived from the replica sums when a finalize request rides on this launch)
  for (int i = tid; i < BN * 2; i += 64 * NW) dstat[i] = 0.0;
  for (int i = tid; i < BN; i += 64 * NW) {
    const int n = n0 + i;
    const bool v = DG && a.e_scale && n < a.Nout;
    ecoef[i] = v ? a.e_scale[n] : 1.f;
    ecoef[BN + i] = v ? a.e_shift[n] : 0.f;
  }
  // (implicit 3x3 convolution: the coefficients are per SOURCE channel, Cs of them, repeated for each of the nine taps)
  const int ncoef = cv ? a.cv.Cs : a.Kin;
  if (a.fold) {
    for (int i = a.Kin + tid; i < kpad; i += 64 * NW) {
      coef[i] = DG ? 0.f : 1.f; coef[kpad + i] = 0.f; coef[2 * kpad + i] = 0.f;
    }
    t3d_fold_block(a.fold, 0, ncoef, coef, kpad, blockIdx.x == 0 && blockIdx.y == 0);
    if constexpr (cv) {
      for (int i = ncoef + tid; i < a.Kin; i += 64 * NW) {
        const int c = i & (ncoef - 1);
        coef[i] = coef[c]; coef[kpad + i] = coef[kpad + c];
        if (DG) coef[2 * kpad + i] = coef[2 * kpad + c];
      }
      __syncthreads();
    }
  } else {
    for (int i = tid; i < kpad; i += 64 * NW) {
      const bool v = i < a.Kin;
      const int c = cv ? (i & (ncoef - 1)) : i;
      if (!DG) {
        coef[i] = (v && a.p0) ? a.p0[c] : 1.f;
        coef[kpad + i] = (v && a.p0) ? a.p1[c] : 0.f;
      } else {
        coef[i] = v ? a.p0[c] : 0.f;
        coef[kpad + i] = v ? a.p1[c] : 0.f;
        coef[2 * kpad + i] = v ? a.p2[c] : 0.f;
      }
    }
    __syncthreads();
  }
  // forward, BatchNorm + ReLU6 prologue: relu6(s x + t) = 6 clamp01((s/6) x + t/6) (pwconv_stream.hip), the 6 in the epilogue
  const bool c6f = !DG && a.p0 && a.act == T3D_ACT_RELU6;
  if (c6f) {
    for (int i = tid; i < 2 * kpad; i += 64 * NW) coef[i] *= T3D_SIXTH;
    __syncthreads();
  }
  const int fact = c6f ? -1 : a.act;      // forward transform: -1 = clamp form
  DEEP_STAMP(1);

  f32x4 acc[RT][NTW];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one phase; SL = ph & 1 selects the LDS buffer, the weight slot in use and the operand-row slot
  auto phase = [&](auto slot_tag, const int ph) {
    constexpr int SL = decltype(slot_tag)::value;
    bf16x8* Ab = Af + (size_t)SL * RT * KSP * 64;
    // ---- transform + store this phase's operand rows (issued two phases ago)
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int k = (ph * KSP + sks + j) * 32 + lg * 8;
      bool ok = rok && (k < a.Kin);
      if constexpr (cv) ok = ok && cv_off(ph * KSP + sks + j) >= 0;
      const int kc = min(k, kpad - 8);
      const float4 c0a = *reinterpret_cast<const float4*>(coef + kc), c0b = *reinterpret_cast<const float4*>(coef + kc + 4);
      const float4 c1a = *reinterpret_cast<const float4*>(coef + kpad + kc),
                   c1b = *reinterpret_cast<const float4*>(coef + kpad + kc + 4);
      const float c0[8] = {c0a.x, c0a.y, c0a.z, c0a.w, c0b.x, c0b.y, c0b.z, c0b.w};
      const float c1[8] = {c1a.x, c1a.y, c1a.z, c1a.w, c1b.x, c1b.y, c1b.z, c1b.w};
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (float)pa[SL][j][e];
      if (!DG) {
        if (fact == -1) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2 t = deep_fma_clamp01(f32x2{x[e], x[e + 1]}, f32x2{c0[e], c0[e + 1]}, f32x2{c1[e], c1[e + 1]});
            x[e] = t[0];
            x[e + 1] = t[1];
          }
        } else {
          act_affine_vec<8>(x, c0, c1, fact);      // (no prologue: scale 1, shift 0, no activation)
        }
      } else {
        const float4 c2a = *reinterpret_cast<const float4*>(coef + 2 * kpad + kc),
                     c2b = *reinterpret_cast<const float4*>(coef + 2 * kpad + kc + 4);
        const float c2[8] = {c2a.x, c2a.y, c2a.z, c2a.w, c2b.x, c2b.y, c2b.z, c2b.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = c0[e] * x[e] + c1[e] * (float)pb[DG ? SL : 0][DG ? j : 0][e] + c2[e];
      }
      bf16x8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = ok ? (bf16_t)x[e] : (bf16_t)0.f;
      Ab[(srt * KSP + sks + j) * 64 + lane] = b;
    }
    __syncthreads();
    if (ph == 0) DEEP_STAMP(2);
    // (sched_barrier: the machine scheduler otherwise sinks the prefetch loads down between the MFMAs, to where their
    //  registers are free)
    __builtin_amdgcn_sched_barrier(0);
    w_issue(std::integral_constant<int, 1 - SL>{}, ph + 1);       // (past the last phase: clamped re-reads, never used)
    a_issue(std::integral_constant<int, SL>{}, ph + 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KSP; ++u) {
      bf16x8 b[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) b[r] = Ab[(r * KSP + u) * 64 + lane];
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[SL][u][t], b[r], acc[r][t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int ph = 0; ph < nph; ph += 2) {
    phase(std::integral_constant<int, 0>{}, ph);
    if (ph + 1 < nph) phase(std::integral_constant<int, 1>{}, ph + 1);
  }

  DEEP_STAMP(3);
  // ---------------- epilogue: lane holds channels n0 + 32 (T >> 1) + 8 lg + 4 (T & 1) .. +3 of pixel m0 + 16 r + lc ----------
  const bool keep_stats = a.stats != nullptr;
  float st1[NTW][4], st2[NTW][4];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) st1[i][j] = st2[i][j] = 0.f;
  // all epilogue loads first (the data gradient's activation input and skip gradient: RT x NTW 8-byte loads each)
  bf16x4 eyr[DG ? RT : 1][NTW], err[DG ? RT : 1][NTW];
  if (DG) {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const size_t mo = (size_t)min(m0 + r * 16 + lc, a.M - 1) * a.Nout;
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int n = min(n0 + (tix[i] >> 1) * 32 + lg * 8 + (tix[i] & 1) * 4, a.Nout - 4);
        if (a.e_y) eyr[DG ? r : 0][i] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(a.e_y) + mo + n);
        if (a.e_res) err[DG ? r : 0][i] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(a.e_res) + mo + n);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int m = m0 + r * 16 + lc;
    const bool ok = m < a.M;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int nl = (tix[i] >> 1) * 32 + lg * 8 + (tix[i] & 1) * 4, n = n0 + nl;      // (n0 is a multiple of 32)
      if (!tok[i] || n >= a.Nout) continue;     // (whole 4-channel groups are in or out: Nout % 8 == 0)
      float v[4], yv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[r][i][j];
      if (c6f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= 6.f;
      }
      if (DG && a.e_y) {
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = (float)eyr[DG ? r : 0][i][j];
        if (a.e_act != T3D_ACT_NONE) {
          const float4 s0 = *reinterpret_cast<const float4*>(ecoef + nl), h0 = *reinterpret_cast<const float4*>(ecoef + BN + nl);
          const float es[4] = {s0.x, s0.y, s0.z, s0.w}, eh[4] = {h0.x, h0.y, h0.z, h0.w};
          act_grad_affine_vec<4>(v, yv, es, eh, a.e_act);
        }
      }
      if (DG && a.e_res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (float)err[DG ? r : 0][i][j];
      }
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (bf16_t)v[j];
        v[j] = ok ? (float)o[j] : 0.f;
      }
      if (ok) *reinterpret_cast<bf16x4*>(out + (size_t)m * a.Nout + n) = o;
      if (keep_stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st1[i][j] += v[j];
          st2[i][j] = fmaf(v[j], (DG && a.e_y) ? yv[j] : v[j], st2[i][j]);
        }
      }
    }
  }
  DEEP_STAMP(4);
  if (keep_stats) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int nl = (tix[i] >> 1) * 32 + lg * 8 + (tix[i] & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s1 = row16_sum(st1[i][j]), s2 = row16_sum(st2[i][j]);
        if (lc == 0 && tok[i] && n0 + nl < a.Nout) {
          dstat[(nl + j) * 2] = t3d_snap(s1, a.quant, false);           // one owner per channel in the block: plain stores
          dstat[(nl + j) * 2 + 1] = t3d_snap(s2, a.quant, !DG);
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < BN * 2; i += 64 * NW) {
      const int n = n0 + (i >> 1);
      if (n < a.Nout) atomicAdd(a.stats + (size_t)(blockIdx.x % nrep) * rstride + (size_t)(i & 1) * a.Nout + n, dstat[i]);
    }
  }
  DEEP_STAMP(5);
}

template <bool DG, bool CV, int NTW, int KSP>
int launch_deep(GemmArgs& a, int KS, int ntiles, int nchunks, hipStream_t st) {
  const size_t lds = (size_t)2 * RT * KSP * 1024 + (size_t)3 * KS * 32 * 4 + (size_t)ntiles * 16 * (2 * 4 + 2 * 8);
  if (lds > 150 * 1024) return T3D_ERR_UNSUPPORTED;
  const void* fn = (const void*)pw_deep_kernel<DG, CV, NTW, KSP>;
  if (lds > 64 * 1024) (void)t3d_max_lds(fn, (int)lds);
  a.quant = (!DG && a.stats && !T3D_ENV_SET("T3D_NO_SNAP")) ? t3d_quant_for(a.M) : T3dQuant{0.0, 0.0};
  a.fold = t3d_take_fold(a.p0);
  T3D_LAUNCH_TIMED((pw_deep_kernel<DG, CV, NTW, KSP>), dim3(cdiv(a.M, 16 * RT), nchunks), dim3(64 * NW), lds, st, a, KS, ntiles,
                     g_t3d_reduce.nrep, g_t3d_reduce.stats_stride);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// fragment order (include/t3d.h): out[((T * KS + ks) * 64 + lg * 16 + lc) * 8 + j] = w[row(T, lc)][ks * 32 + lg * 8 + j],
// row(T, lc) = (T >> 1) * 32 + (lc >> 2) * 8 + (T & 1) * 4 + (lc & 3), zero past rows / cols; any 16-bit storage type
__global__ void pack_frag_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ out, int rows, int cols, int KS,
                                 size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lc = (i >> 3) & 15, lg = (i >> 7) & 3;
    const size_t tk = i >> 9;
    const int ks = (int)(tk % KS), T = (int)(tk / KS);
    const int n = (T >> 1) * 32 + (lc >> 2) * 8 + (T & 1) * 4 + (lc & 3), k = ks * 32 + lg * 8 + j;
    out[i] = (n < rows && k < cols) ? w[(size_t)n * cols + k] : (unsigned short)0;
  }
}

}  // namespace

#ifdef T3D_PW_TRACE
extern "C" int t3d_debug_deep_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_deep_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

// is (contraction Kin -> Nout) a shape for this kernel?  Only where the streaming kernel needs more than one output chunk
// (its chunk: <= 120 KB of weights, <= 10 tiles)
bool deep_shape(int Kin, int Nout) {
  if (Kin < 512 || (Kin % 8) || (Nout % 8)) return false;
  const int KS = cdiv(Kin, 32);
  int nt_cap = 120 / KS;
  if (nt_cap > 10) nt_cap = 10;
  nt_cap &= ~1;
  return !(nt_cap >= 2 && Nout <= nt_cap * 16);
}

// bf16, a.w in fragment order; T3D_ERR_UNSUPPORTED = "not a launch for this kernel"
int deep_launch(GemmArgs& a, hipStream_t st) {
  if (a.a2 || a.z_out || a.per_sample || a.ps_stats || a.e_se || a.bias || (!a.dgrad && a.p2)) return T3D_ERR_UNSUPPORTED;
  if (a.row0 && a.row0 != a.Kin) return T3D_ERR_UNSUPPORTED;
  if (a.dgrad && (!a.a1 || !a.p0 || !a.p1 || !a.p2)) return T3D_ERR_UNSUPPORTED;
  if (!a.cv.mode && !deep_shape(a.Kin, a.Nout)) return T3D_ERR_UNSUPPORTED;
  if (a.cv.mode && (a.Kin != 9 * a.cv.Cs || (a.cv.Cs & (a.cv.Cs - 1)) || a.cv.Cs < 32 || (a.cv.stride != 1 && a.cv.stride != 2)))
    return T3D_ERR_UNSUPPORTED;
  const int KS = cdiv(a.Kin, 32);
  const int pairs = cdiv(a.Nout, 32);
  // Two shapes of a wave's share (measured, tools/scratch/deep_rot_ab.sh): 2 tiles x 4 k-steps of weights ahead (chunks of <= 16
  // tiles = 256 channels), or 3 tiles x 2 k-steps (<= 24 tiles = 384 channels) where that makes ONE chunk of two -- every chunk stages AND
  // transforms the operand again: 960 -> 320 @7x7 35.2 us as two chunks of 10 tiles, 23.5 as one of 20; 320 <- 1280 59.4 / 39.1;
  // 960 -> 160 (one chunk either way) 18.3 / 20.9.  Chunks evenly sized.
  // OPT-IN (T3D_DEEP_WIDE=1): alone the two launches it changes get 12 and 20 us faster, the STEP does not (6.718 / 6.694 / 6.700
  // ms with against 6.698 / 6.669 / 6.680 without, three same-box pairs) -- finding 20's lesson once more
  const bool wide = pairs > 8 && pairs <= 12 && T3D_ENV_SET("T3D_DEEP_WIDE");        // (one chunk instead of two; 32 pairs as 3 x 22 tiles ran 53 us, as 4 x 16 45)
  const int per = wide ? 12 : 8;
  const int nchunks = cdiv(pairs, per), ntiles = 2 * cdiv(pairs, nchunks);
  if (wide) {
    if (a.cv.mode) return a.dgrad ? launch_deep<true, true, 3, 2>(a, KS, ntiles, nchunks, st) : launch_deep<false, true, 3, 2>(a, KS, ntiles, nchunks, st);
    return a.dgrad ? launch_deep<true, false, 3, 2>(a, KS, ntiles, nchunks, st) : launch_deep<false, false, 3, 2>(a, KS, ntiles, nchunks, st);
  }
  if (a.cv.mode) return a.dgrad ? launch_deep<true, true, 2, 4>(a, KS, ntiles, nchunks, st) : launch_deep<false, true, 2, 4>(a, KS, ntiles, nchunks, st);
  return a.dgrad ? launch_deep<true, false, 2, 4>(a, KS, ntiles, nchunks, st) : launch_deep<false, false, 2, 4>(a, KS, ntiles, nchunks, st);
}

}  // namespace t3d_pw

// include/t3d.h
extern "C" int t3d_pwconv_frag_bytes(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return 2 * cdiv(rows, 32) * cdiv(cols, 32) * 1024;
}

// where the fragment-order copy PAYS: the deep-contraction kernel's shapes.  (The streaming kernel takes the layout for every
// shape of its own -- its weight staging becomes one linear copy -- but that is worth ~1 us per launch alone and nothing in
// the step, less than packing a second copy of every layer costs: DESIGN finding 34.)
extern "C" int t3d_pwconv_wants_frag(int K, int N) { return (t3d_pw::deep_shape(K, N) || t3d_pw::wide_shape(K, N)) ? 1 : 0; }

extern "C" int t3d_pwconv_pack_frag(const void* w, void* out, int rows, int cols, void* stream) {
  if (!w || !out || rows <= 0 || cols <= 0) return T3D_ERR_ARG;
  const size_t total = (size_t)t3d_pwconv_frag_bytes(rows, cols) / 2;
  T3D_LAUNCH(t3d_pw::pack_frag_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const unsigned short*>(w), reinterpret_cast<unsigned short*>(out), rows,
                     cols, cdiv(cols, 32), total);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
