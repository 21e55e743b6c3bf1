// Pointwise (1x1) convolution weight gradient, bf16 storage, gfx950:
//
//   dW[n][k] += sum_m dy[m][n] * a[m][k],     dy = alpha*dz + beta*y + gamma   (BatchNorm backward, on load)
//                                             a  = act(scale*x + shift [, se])  (recomputed, never stored)
//
// The contraction runs over PIXELS, i.e. along the slow axis of both (pixel-major) operands, so both MFMA
// fragments are transposes of what memory holds.  gfx950's ds_read_b64_tr_b16 delivers exactly that
// transpose for free (16 lanes read a 4-row x 16-column block column-major), so staging stays trivial:
//   * per 32-pixel step the block converts its slices of dz/y -> dy and x -> a in registers and stores them
//     ROW-major into LDS with 16-B writes (double-buffered: one barrier per step, the next step's global
//     loads are in flight during the MFMAs);
//   * wave w owns rows {w, w+4, ...}*16 of the block's dW tile ("P" side, split over waves) and ALL its
//     columns ("Q" side, shared): per step it reads the Q fragments once, then per own row tile one P
//     fragment + NTQ MFMAs (v_mfma_f32_16x16x32_bf16).  Accumulators live in registers for the block's
//     whole pixel range;
//   * the block tile is as wide as the register file allows (up to 384 x 64 / 192 x 160), so for most
//     layers every operand byte is read exactly once; wider dW are tiled over P (the tensor on the P side
//     is then still read once, the Q side once per P tile);
//   * the larger channel count always takes the P role (`swap` transposes the roles of dy and a).
// Grid = (P tiles * Q tiles) x pixel splits; partial dW leave as fp32 atomics (few, large, spread over the
// whole dW -- no contention problem here, unlike per-channel sums).
#include <cstdlib>
#include "pwconv_common.h"

namespace {

// relu6(s x + t) = 6 clamp01((s/6) x + t/6): one v_pk_fma_f32 with the clamp modifier per channel pair (dwconv3_stream.hip has the
// note); the weight gradient is linear in the activated operand, so the 6 is applied once, to the accumulators at the flush
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

typedef __attribute__((ext_vector_type(4))) short s16x4;

// act_apply (common.h) on eight values, the switch outside the loop
__device__ __forceinline__ void act_vec8(float* v, int act) {
  switch (act) {
    case T3D_ACT_RELU:
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      break;
    case T3D_ACT_RELU6:
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], 0.f, 6.f);
      break;
    case T3D_ACT_HSWISH:
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * (__builtin_amdgcn_fmed3f(v[j] + 3.f, 0.f, 6.f) * T3D_SIXTH);
      break;
    default: break;
  }
}

struct WgtArgs {
  const void *dz, *y, *x;
  const float *alpha, *beta, *gamma;
  int per_sample;
  const float *scale, *shift, *se;
  int act, se_after;
  float* dw;
  int M, HW, K, N;
  int yfree, Nz;     // y-free mode: dy side = [dz (Nz channels) | x (K channels) | 1, 0 x 7], N = Nz + K + 8 virtual channels
  int swap;          // 0: P = N (dy side), Q = K (a side); 1: P = K, Q = N
  int ptiles, qtiles, rows_per_split, nsplit;
  float* ws;         // partial tiles [split][tile][PB][QB] (plain stores) or null (atomics into dw)
  const T3dFold* fold;    // requested BatchNorm-backward finalize: (alpha, beta, gamma) derived per block, NOT published
                          // (the data-gradient kernel of the main stream publishes; common.h)
  // fused y-free backward (DGF): the data gradient of the same layer from the staged [dz | x | 1] rows
  const bf16_t* wd;       // [QB][PB]: row k = [alpha_n W[n][k] (n < Nz) | Q[k][.] (K) | c[k] | 0 ...], the P tile's column order
  // ... or, with wd == null, built in the kernel's prologue from the transposed weights wtr [K][Nz] and the expansion's
  // BatchNorm-backward coefficients (alpha / beta / gamma, or derived from `fold`): what t3d_pwconv_yfree_prep2 computes as a
  // launch of its own on the critical stream (13 us x 6 per step), same arithmetic in the same order
  const bf16_t* wtr;
  const void *e_res, *e_y;   // skip-connection gradient [M][K] or null; raw tensor of x's producer [M][K] or null (its sums)
  void* dx;               // [M][K] bf16
  double* stats;          // [2][K] replicas or null
  int nrep;
  long long rstride;
  // implicit 3x3 convolution (conv3x3.hip): the a side [M][K = 9 C] is not a patch matrix in memory -- column k = tap * C + c
  // of output pixel m reads channel c of input pixel (oy * stride - 1 + ky, ox * stride - 1 + kx) of x [B][H][W][C] (raw, the
  // BatchNorm + activation prologue per channel c), zero outside the image
  struct { int on, H, W, Ho, Wo, C, lgC, stride; unsigned mulW, mulH; } cv;
  int assign;             // dw is WRITTEN, not accumulated into (the y-free product matrix: no clear needed ahead of the launch)
  // squeeze-excite gate of the a side staged in LDS (GEN, `se` set): gs_ns = samples a block's pixel range can touch (0: the
  // gates are read from global memory per staged vector, round 3's path), mulHW = ceil(2^32 / HW) for (m - first sample's
  // first pixel) / HW -- exact for the small differences a block sees (launcher)
  int gs_ns;
  unsigned mulHW;
};

// one 16x16 tile row (transposed) fragment: pixels 8*lg .. 8*lg+7 of the step, channels ch0..ch0+15
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int rs, int ch0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const bf16_t* a0 = tile + (8 * g + q) * rs + ch0 + 4 * p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 4 * rs));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// NTPW: 16-row tiles per wave on the P side (block: 64*NTPW rows); NTQ: 16-column tiles on the Q side
// G: independent 4-wave pipelines per block, taking alternate 32-pixel steps (own LDS buffers, own accumulators);
// they are summed through LDS before the block's single flush -- twice the per-block throughput for one flush.
// SK: 32-pixel contraction sub-steps per barrier-separated step.  The narrow tiles of the wide, shallow layers (16 ... 144
// channels on each side) stage only 4-12 KB per 32 pixels -- a quarter of the threads has a vector to move and the barrier
// comes every few hundred bytes per thread; SK = 2 / 4 stages 64 / 128 pixels per step instead.
// DGF (y-free only, one output tile): ALSO the layer's data gradient  dx = [dz | x | 1] Wd^T (+ skip gradient), formed from
// the SAME staged rows -- the pair t3d_pwconv_dgrad_yfree (main stream) + t3d_pwconv_wgrad_yfree (second stream) read the
// wide gradient tensor twice, at the same time (each at half the bandwidth); here it is read once.  A step's pixels are
// dealt to the pipeline's four waves as (16-pixel tile, 16-channel tile) units: A = Wd rows from LDS, B = the staged rows
// read as they lie (pixel-major, 8 consecutive virtual channels per lane), D[k][pixel] -> a lane holds 4 consecutive
// channels of one pixel: 8-byte stores; the skip gradient and the raw tensor of the producer (BatchNorm-backward sums of x's
// producer, exactly as in the streaming kernel's epilogue) are fetched with the step's operands.
// CV: implicit 3x3 convolution on the a side (WgtArgs::cv) -- a compile-time variant, the plain kernels' staging is untouched
template <int NTPW, int NTQ, bool SWAP, int G, int D, int GEN, bool YF, int SK = 1, bool DGF = false, bool CV = false>
__global__ __launch_bounds__(256 * G) void pw_wgrad_tr_kernel(const WgtArgs a) {
  static_assert(!DGF || (YF && !SWAP), "the fused data gradient exists for the y-free layout only");
  constexpr int PB = 64 * NTPW, QB = 16 * NTQ;
  constexpr int RSP = PB + 8, RSQ = QB + 8;            // LDS row strides (elements): +16 B against bank conflicts
  constexpr int STEP = 32 * SK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BUFE = STEP * (RSP + RSQ);              // elements per buffer: P tile then Q tile
  const int grp = threadIdx.x >> 8;
  bf16_t* const tiles = reinterpret_cast<bf16_t*>(smem) + grp * 2 * BUFE;  // this pipeline's two buffers
  float* coef = reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(smem) + G * 2 * BUFE);   // dy side [3][nd] | a side [2][na]
  constexpr int RSW = PB + 8;                           // Wd row stride in LDS (elements)
  constexpr int NCOEF = 3 * (SWAP ? QB : PB) + 2 * (SWAP ? PB : QB);
  bf16_t* const wdl = reinterpret_cast<bf16_t*>(coef + ((NCOEF + 3) & ~3));                  // DGF: [QB][RSW]
  double* const dstat = reinterpret_cast<double*>(wdl + (DGF ? QB * RSW : 0));               // DGF: [QB][2]

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int P = SWAP ? a.K : a.N, Q = SWAP ? a.N : a.K;
  // block -> (output tile, pixel split).  All tiles of one split read the same pixel rows (dy and a columns of their own
  // tile only), and consecutive workgroup ids go round-robin over the 8 XCDs: with tile = id % ntiles every tile of a split
  // fetched those rows into a different L2.  Splits are dealt out per XCD instead (whole groups of 8 splits; the rest
  // keeps the plain order), so a split's tiles share an L2.
  const int ntiles = a.ptiles * a.qtiles, s8 = a.nsplit & ~7;
  int tile, split;
  {
    const int L = blockIdx.x;
    if (ntiles > 1 && L < ntiles * s8) {
      const int j = L >> 3;
      tile = j % ntiles;
      split = (j / ntiles) * 8 + (L & 7);
    } else {
      const int Lr = ntiles > 1 ? L - ntiles * s8 : L;
      tile = Lr % ntiles;
      split = (ntiles > 1 ? s8 : 0) + Lr / ntiles;
    }
  }
  const int pt = tile / a.qtiles, qt = tile % a.qtiles;
  const int p0 = pt * PB, q0 = qt * QB;
  const int mbeg = split * a.rows_per_split, mend = min(a.M, mbeg + a.rows_per_split);
  // which side carries dy (N channels) / a (K channels)
  constexpr int dyB = SWAP ? QB : PB, aB = SWAP ? PB : QB;
  const int dy0 = SWAP ? q0 : p0, a0c = SWAP ? p0 : q0;
  float* cdy = coef;                 // [3][dyB]
  float* ca = coef + 3 * dyB;        // [2][aB]
  if (!YF && a.fold) {
    const int nv = min(dyB, a.N - dy0);
    for (int i = nv + threadIdx.x; i < dyB; i += 256 * G) cdy[i] = cdy[dyB + i] = cdy[2 * dyB + i] = 0.f;
    t3d_fold_block(a.fold, dy0, nv, cdy, dyB, false);
  } else {
  for (int i = threadIdx.x; i < dyB && !YF; i += 256 * G) {
    const int n = dy0 + i;
    const bool v = n < a.N;
    cdy[i] = (v && !a.per_sample) ? a.alpha[n] : 0.f;
    cdy[dyB + i] = v ? a.beta[n] : 0.f;
    cdy[2 * dyB + i] = (v && !a.per_sample) ? a.gamma[n] : 0.f;
  }
  }
  // ReLU6 on a plain BatchNorm affine (every projection layer's operand): the clamp form
  const bool c6 = !GEN && !YF && a.act == T3D_ACT_RELU6 && !a.se;
  for (int i = threadIdx.x; i < aB; i += 256 * G) {
    const int k = a0c + i;
    const bool v = k < a.K;
    const int kc = CV ? (k & (a.cv.C - 1)) : k;
    ca[i] = ((v && a.scale) ? a.scale[kc] : 1.f) * (c6 ? T3D_SIXTH : 1.f);
    ca[aB + i] = ((v && a.scale) ? a.shift[kc] : 0.f) * (c6 ? T3D_SIXTH : 1.f);
  }
  float* const gsl = coef + ((NCOEF + 3) & ~3);           // GEN: [gs_ns][aB] gate slice of this block's samples and channels
  const int gs_b0 = GEN ? mbeg / a.HW : 0;
  if constexpr (GEN == 2) {
    {
      const int nb = a.M / a.HW;
      for (int i = threadIdx.x; i < a.gs_ns * aB; i += 256 * G) {
        const int sb = i / aB, c = i % aB, b = min(gs_b0 + sb, nb - 1), k = a0c + c;
        gsl[i] = k < a.K ? a.se[(size_t)b * a.K + k] : 0.f;
      }
    }
  }
  if constexpr (DGF) {
    if (a.wd) {
      for (int i = threadIdx.x; i < QB * (PB / 8); i += 256 * G) {
        const int r = i / (PB / 8), c = (i % (PB / 8)) * 8;
        *reinterpret_cast<bf16x8*>(wdl + r * RSW + c) = *reinterpret_cast<const bf16x8*>(a.wd + (size_t)r * PB + c);
      }
    } else {
      // ---- the data gradient's weight rows built here (yfree_prep_kernel's arithmetic, pwconv_yfree.hip, step for step: the
      // contraction of Q split over four waves by step mod 4 and summed in wave order, c by 16-lane rows, alpha . W rounded
      // the same way -- the two ways of getting wd agree bit for bit, tests/test_gpu_pwconv.py)
      const int Nz = a.Nz, K = a.K;
      float* fco = cdy;                              // [3][dyB] (free in the y-free layout): alpha, beta, gamma of the expansion's BN
      if (a.fold) {
        for (int i = Nz + threadIdx.x; i < dyB; i += 256 * G) fco[i] = fco[dyB + i] = fco[2 * dyB + i] = 0.f;
        t3d_fold_block(a.fold, 0, Nz, fco, dyB, blockIdx.x == 0);
      } else {
        for (int i = threadIdx.x; i < dyB; i += 256 * G) {
          const bool v = i < Nz;
          fco[i] = v ? a.alpha[i] : 0.f; fco[dyB + i] = v ? a.beta[i] : 0.f; fco[2 * dyB + i] = v ? a.gamma[i] : 0.f;
        }
      }
      for (int i = threadIdx.x; i < QB * (RSW / 8); i += 256 * G) *reinterpret_cast<bf16x8*>(wdl + i * 8) = bf16x8{};
      __syncthreads();
      const float *alpha = fco, *beta = fco + dyB, *gamma = fco + 2 * dyB;
      const bf16_t* __restrict__ wtr = a.wtr;
      // alpha . W: row k = wt row k scaled by alpha
      for (int i = threadIdx.x; i < QB * (Nz / 8); i += 256 * G) {
        const int k = i / (Nz / 8), n = (i % (Nz / 8)) * 8;
        if (k < K) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(wtr + (size_t)k * Nz + n);
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(alpha[n + j] * (float)v[j]);
          *reinterpret_cast<bf16x8*>(wdl + k * RSW + n) = o;
        }
      }
      // Q[k][k2] = sum_n W[n][k] beta_n W[n][k2]: one 16 x 16 tile at a time, pipeline 0's four waves split the contraction
      f32x4* qred = reinterpret_cast<f32x4*>(tiles);          // [3][64] exchange (the staging buffers are not in use yet)
      const int lgq_ = lane >> 4, lcq_ = lane & 15;
      const int nst = (Nz + 31) / 32;
      for (int kt = 0; kt < QB / 16; ++kt)
        for (int qt2 = 0; qt2 < QB / 16; ++qt2) {
          f32x4 qa = {0.f, 0.f, 0.f, 0.f};
          if (grp == 0) {
            const int ka = min(kt * 16 + lcq_, K - 1), kb = min(qt2 * 16 + lcq_, K - 1);
            const bf16_t* ra = wtr + (size_t)ka * Nz;
            const bf16_t* rb = wtr + (size_t)kb * Nz;
            for (int st = wave; st < nst; st += 4) {
              const int n = st * 32 + 8 * lgq_;
              const bool live = n < Nz;
              const int nc = min(n, Nz - 8);
              const bf16x8 fa = *reinterpret_cast<const bf16x8*>(ra + nc);
              const bf16x8 fb = *reinterpret_cast<const bf16x8*>(rb + nc);
              bf16x8 b;
#pragma unroll
              for (int j = 0; j < 8; ++j) b[j] = (bf16_t)(live ? beta[nc + j] * (float)fb[j] : 0.f);
              qa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, b, qa, 0, 0, 0);
            }
            if (wave) qred[(wave - 1) * 64 + lane] = qa;
          }
          __syncthreads();
          if (grp == 0 && wave == 0) {
            qa += qred[lane];
            qa += qred[64 + lane];
            qa += qred[128 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = kt * 16 + 4 * lgq_ + r, k2 = qt2 * 16 + lcq_;
              if (k < K && k2 < K) wdl[k * RSW + Nz + k2] = (bf16_t)qa[r];
            }
          }
          __syncthreads();
        }
      // c[k] = sum_n gamma_n W[n][k]: 16 lanes per row
      if (grp == 0) {
        for (int kb = 0; kb < QB; kb += 16) {
          const int row = tid >> 4, sub = tid & 15, k = kb + row;
          float s2 = 0.f;
          if (k < K)
            for (int n = sub * 8; n < Nz; n += 128) {
              const bf16x8 v = *reinterpret_cast<const bf16x8*>(wtr + (size_t)k * Nz + n);
#pragma unroll
              for (int j = 0; j < 8; ++j) s2 = fmaf(gamma[n + j], (float)v[j], s2);
            }
          s2 = row16_sum(s2);
          if (sub == 0 && k < K) wdl[k * RSW + Nz + K] = (bf16_t)s2;
        }
      }
    }
    for (int i = threadIdx.x; i < 2 * QB; i += 256 * G) dstat[i] = 0.0;
  }
  const bf16_t* __restrict__ dz = reinterpret_cast<const bf16_t*>(a.dz);
  const bf16_t* __restrict__ yy = reinterpret_cast<const bf16_t*>(a.y);
  const bf16_t* __restrict__ xx = reinterpret_cast<const bf16_t*>(a.x);
  const bool plain_a = !a.scale && !a.se && a.act == T3D_ACT_NONE;

  // staging map: vector v (8 channels x 1 pixel) of the dy slice / the a slice, v = tid + 256*i
  constexpr int dyV = dyB / 8, aV = aB / 8;                  // vectors per pixel row
  constexpr int ndv = STEP * dyV, nav = STEP * aV;
  constexpr int VDY = (ndv + 255) / 256, VA = (nav + 255) / 256;   // vectors per thread and step
  // D register sets: the global loads of step s+D are issued while step s is being multiplied, so a load has D-1
  // whole iterations to land (one iteration ~= one MFMA burst, far shorter than the HBM round trip)
  // DGF units of a step: (16-pixel tile, 16-channel tile) pairs, dealt round-robin to the pipeline's four waves
  constexpr int NU = DGF ? (STEP / 16) * NTQ : 1, UPW = (NU + 3) / 4;
  struct Epi { bf16x4 res[UPW], xr[UPW]; };
  struct NoMask {};
  // okm (implicit 3x3 only): bit i = vector i's tap is inside the image
  struct Regs { bf16x8 rz[VDY], ry[VDY], rx[VA]; Epi e; typename std::conditional<CV, unsigned, NoMask>::type okm; };
  Regs rr[D];
  Epi ecur;                     // epilogue operands of the step that is in LDS now
  const int lgq = lane >> 4, lcq = lane & 15;
  const bf16_t* __restrict__ eres = reinterpret_cast<const bf16_t*>(a.e_res);
  const bf16_t* __restrict__ eyr = reinterpret_cast<const bf16_t*>(a.e_y);

  auto gload = [&](Regs& R, int m0) {
    // branch-free: out-of-range vectors read a clamped (valid) address and are zeroed when they are staged
#pragma unroll
    for (int i = 0; i < VDY; ++i) {
      const int v = min(tid + 256 * i, ndv - 1);
      const int row = v / dyV, m = min(m0 + row, a.M - 1);
      if constexpr (YF) {
        // virtual channel n: < Nz -> dz, < Nz + K -> the conv input itself (Gram rows), beyond -> constants (lstore)
        const int n = dy0 + (v % dyV) * 8;
        const bool fromz = n < a.Nz;
        const bf16_t* src = fromz ? dz + (size_t)m * a.Nz + n : xx + (size_t)m * a.K + min(max(n - a.Nz, 0), a.K - 8);
        R.rz[i] = *reinterpret_cast<const bf16x8*>(src);
      } else {
        const int n = min(dy0 + (v % dyV) * 8, a.N - 8);
        R.rz[i] = *reinterpret_cast<const bf16x8*>(dz + (size_t)m * a.N + n);
        R.ry[i] = *reinterpret_cast<const bf16x8*>(yy + (size_t)m * a.N + n);
      }
    }
    if constexpr (CV) R.okm = ~0u;
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      const int v = min(tid + 256 * i, nav - 1);
      const int row = v / aV, k = min(a0c + (v % aV) * 8, a.K - 8), m = min(m0 + row, a.M - 1);
      if constexpr (CV) {
        const int tap = k >> a.cv.lgC, c = k & (a.cv.C - 1);
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const int t = (int)__umulhi((unsigned)m, a.cv.mulW), ox = m - t * a.cv.Wo;       // m / Wo, m % Wo (exact: launcher)
        const int b = (int)__umulhi((unsigned)t, a.cv.mulH), oy = t - b * a.cv.Ho;
        const int iy = oy * a.cv.stride - 1 + ky, ix = ox * a.cv.stride - 1 + kx;
        const bool ok = (unsigned)iy < (unsigned)a.cv.H && (unsigned)ix < (unsigned)a.cv.W;
        const size_t o = ok ? ((size_t)(b * a.cv.H + iy) * a.cv.W + ix) * a.cv.C + c : 0;
        R.rx[i] = *reinterpret_cast<const bf16x8*>(xx + o);
        if (!ok) R.okm &= ~(1u << i);
      } else {
        R.rx[i] = *reinterpret_cast<const bf16x8*>(xx + (size_t)m * a.K + k);
      }
    }
    if constexpr (DGF) {
#pragma unroll
      for (int i = 0; i < UPW; ++i) {
        const int un = min(wave + 4 * i, NU - 1), pt = un / NTQ, kt = un % NTQ;
        const size_t o = (size_t)min(m0 + pt * 16 + lcq, a.M - 1) * a.K + min(kt * 16 + 4 * lgq, a.K - 4);
        if (eres) R.e.res[i] = *reinterpret_cast<const bf16x4*>(eres + o);
        if (eyr) R.e.xr[i] = *reinterpret_cast<const bf16x4*>(eyr + o);
      }
    }
  };
  auto ld8 = [](const float* p, float* o) {     // 8 consecutive LDS floats as two 16-B reads
    const float4 q0 = *reinterpret_cast<const float4*>(p), q1 = *reinterpret_cast<const float4*>(p + 4);
    o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q0.w; o[4] = q1.x; o[5] = q1.y; o[6] = q1.z; o[7] = q1.w;
  };
  bf16x8 zero8;
#pragma unroll
  for (int j = 0; j < 8; ++j) zero8[j] = (bf16_t)0.f;
  auto lstore = [&](const Regs& R, int m0, int buf) {
    const auto& rz = R.rz; const auto& ry = R.ry; const auto& rx = R.rx;
    bf16_t* const pb_ = tiles + buf * BUFE;
    bf16_t* const qb_ = pb_ + STEP * RSP;
    bf16_t* dyt = SWAP ? qb_ : pb_;
    bf16_t* at = SWAP ? pb_ : qb_;
    constexpr int rsd = SWAP ? RSQ : RSP, rsa = SWAP ? RSP : RSQ;
#pragma unroll
    for (int i = 0; i < VDY; ++i) {
      const int v = tid + 256 * i;
      if (v < ndv) {
        const int row = v / dyV, cl = (v % dyV) * 8, m = m0 + row;
        const bool ok = m < mend && dy0 + cl < a.N;
        bf16x8 o;
        if constexpr (YF) {
          const int n = dy0 + cl;
          o = rz[i];
          if (n >= a.Nz + a.K) {       // the ones column (sum of the conv input) and its zero padding
            o = zero8;
            if (n == a.Nz + a.K) o[0] = (bf16_t)1.f;
          }
          *reinterpret_cast<bf16x8*>(dyt + row * rsd + cl) = ok ? o : zero8;
          continue;
        }
        float al[8], be[8], ga[8];
        ld8(cdy + dyB + cl, be);
        if constexpr (GEN == 1) {
          if (a.per_sample) {
            const size_t pb = (size_t)(min(m, a.M - 1) / a.HW) * a.N + min(dy0 + cl, a.N - 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { al[j] = a.alpha[pb + j]; ga[j] = a.gamma[pb + j]; }
          } else {
            ld8(cdy + cl, al);
            ld8(cdy + 2 * dyB + cl, ga);
          }
        } else {
          ld8(cdy + cl, al);
          ld8(cdy + 2 * dyB + cl, ga);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)fmaf(al[j], (float)rz[i][j], fmaf(be[j], (float)ry[i][j], ga[j]));
        *reinterpret_cast<bf16x8*>(dyt + row * rsd + cl) = ok ? o : zero8;
      }
    }
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      const int v = tid + 256 * i;
      if (v < nav) {
        const int row = v / aV, cl = (v % aV) * 8, m = m0 + row;
        bool ok = m < mend && a0c + cl < a.K;
        if constexpr (CV) ok = ok && ((R.okm >> i) & 1u);
        bf16x8 o = rx[i];
        if (!plain_a) {
          float u[8], sc[8], sh[8];
          ld8(ca + cl, sc);
          ld8(ca + aB + cl, sh);
#pragma unroll
          for (int j = 0; j < 8; ++j) u[j] = (float)rx[i][j];
          bool done = false;
          if constexpr (GEN == 2) {
            // gates from LDS; the activation switch outside the element loops
            float svv[8];
            const int rel = min(m, mend - 1) - gs_b0 * a.HW;
            ld8(gsl + (int)__umulhi((unsigned)rel, a.mulHW) * aB + cl, svv);
            if (a.se_after) {
              act_affine_vec<8>(u, sc, sh, a.act);
#pragma unroll
              for (int j = 0; j < 8; ++j) u[j] *= svv[j];
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) u[j] = fmaf(u[j], sc[j], sh[j]) * svv[j];
              act_vec8(u, a.act);
            }
            done = true;
          }
          if constexpr (GEN == 1) {
            if (a.se) {
              float svv[8];
              {
                const float* se = a.se + (size_t)(min(m, a.M - 1) / a.HW) * a.K + min(a0c + cl, a.K - 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) svv[j] = se[j];
              }
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                float t = fmaf(u[j], sc[j], sh[j]);
                const float sv = svv[j];
                if (!a.se_after) t *= sv;
                t = act_apply(t, a.act);
                if (a.se_after) t *= sv;
                u[j] = t;
              }
              done = true;
            }
          }
          if (c6) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const f32x2 t = pk_fma_clamp01(f32x2{u[j], u[j + 1]}, f32x2{sc[j], sc[j + 1]}, f32x2{sh[j], sh[j + 1]});
              u[j] = t[0];
              u[j + 1] = t[1];
            }
          } else if (!done) act_affine_vec<8>(u, sc, sh, a.act);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)u[j];
        }
        *reinterpret_cast<bf16x8*>(at + row * rsa + cl) = ok ? o : zero8;
      }
    }
  };

  f32x4 acc[NTPW][NTQ];
#pragma unroll
  for (int i = 0; i < NTPW; ++i)
#pragma unroll
    for (int j = 0; j < NTQ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float st1[UPW][4], st2[UPW][4];
#pragma unroll
  for (int i = 0; i < UPW; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) st1[i][r] = st2[i][r] = 0.f;
  // data gradient of the step whose rows sit in `pcur` (first pixel m0)
  auto dgrad_step = [&](const bf16_t* pcur, int m0) {
    bf16_t* __restrict__ dxo = reinterpret_cast<bf16_t*>(a.dx);
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const int un = wave + 4 * i;
      if (un < NU) {                       // wave-uniform
        const int pt = un / NTQ, kt = un % NTQ;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int js = 0; js < PB / 32; ++js) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(wdl + (kt * 16 + lcq) * RSW + js * 32 + 8 * lgq);
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(pcur + (pt * 16 + lcq) * RSP + js * 32 + 8 * lgq);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, d, 0, 0, 0);
        }
        // lane: channels kt*16 + 4*lg .. +3 of pixel m0 + pt*16 + lc
        const int m = m0 + pt * 16 + lcq, k = kt * 16 + 4 * lgq;
        const bool ok = m < mend && k < a.K;
        float v[4] = {d[0], d[1], d[2], d[3]}, yv[4] = {0.f, 0.f, 0.f, 0.f};
        if (eres) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)ecur.res[i][r];
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (bf16_t)v[r];
          v[r] = ok ? (float)o[r] : 0.f;
        }
        if (ok) *reinterpret_cast<bf16x4*>(dxo + (size_t)m * a.K + k) = o;
        if (a.stats) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (eyr) yv[r] = (float)ecur.xr[i][r];
            st1[i][r] += v[r];
            st2[i][r] = fmaf(v[r], eyr ? yv[r] : v[r], st2[i][r]);
          }
        }
      }
    }
  };

  __syncthreads();   // coefficients visible
  const int nsteps = (mend - mbeg + STEP - 1) / STEP;
  const int niter = (nsteps + G - 1) / G;          // same trip count for every pipeline (barriers are block-wide);
                                                   // a step past the range loads zeros and adds nothing
  auto step_m = [&](int it) { return mbeg + (it * G + grp) * STEP; };
  // loads and LDS stores are unconditional (rows past the range read as zeros): a branch around them would make the
  // compiler wait for ALL outstanding loads (vmcnt(0)) instead of just the oldest set
#pragma unroll
  for (int u = 0; u < D; ++u) gload(rr[u], step_m(u));
  lstore(rr[0], step_m(0), 0);
  if constexpr (DGF) ecur = rr[0].e;
  __syncthreads();
  int buf = 0;
  for (int it0 = 0; it0 < niter; it0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int it = it0 + u;
      if (it < niter) {       // block-uniform
        gload(rr[u], step_m(it + D));
        const bf16_t* pcur = tiles + buf * BUFE;
        const bf16_t* qcur = pcur + STEP * RSP;
#pragma unroll
        for (int kk = 0; kk < SK; ++kk) {
          bf16x8 qf[NTQ];
#pragma unroll
          for (int j = 0; j < NTQ; ++j) qf[j] = tr_frag(qcur + kk * 32 * RSQ, RSQ, j * 16, lane);
#pragma unroll
          for (int i = 0; i < NTPW; ++i) {
            const bf16x8 pf = tr_frag(pcur + kk * 32 * RSP, RSP, (wave + 4 * i) * 16, lane);
#pragma unroll
            for (int j = 0; j < NTQ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, qf[j], acc[i][j], 0, 0, 0);
          }
        }
        if constexpr (DGF) {
          dgrad_step(pcur, step_m(it));
          ecur = rr[(u + 1) % D].e;
        }
        lstore(rr[(u + 1) % D], step_m(it + 1), buf ^ 1);
        __syncthreads();
        buf ^= 1;
      }
    }
  }

  if constexpr (DGF) {
    if (a.stats) {
      // BatchNorm-backward sums of x's producer: lanes of one channel group meet by DPP, waves in fp64 LDS, blocks by fp64
      // atomics into the replicas (exact adds in any order, as everywhere on the data-gradient path)
#pragma unroll
      for (int i = 0; i < UPW; ++i) {
        const int un = wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = row16_sum(st1[i][r]), s2 = row16_sum(st2[i][r]);
          const int k = (un % NTQ) * 16 + 4 * lgq + r;
          if (un < NU && lcq == 0 && k < a.K) {
            atomicAdd(dstat + 2 * k, (double)s1);
            atomicAdd(dstat + 2 * k + 1, (double)s2);
          }
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < 2 * a.K; i += 256 * G)
        atomicAdd(a.stats + (size_t)(blockIdx.x % a.nrep) * a.rstride + (size_t)(i & 1) * a.K + (i >> 1), dstat[i]);
      __syncthreads();
    }
  }
  if constexpr (G == 2) {
    // pipeline 1 hands its accumulators to pipeline 0 through LDS, one row-tile group at a time (lane-private slots)
    f32x4* ex = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int i = 0; i < NTPW; ++i) {
      if (grp == 1) {
#pragma unroll
        for (int j = 0; j < NTQ; ++j) ex[(tid * NTQ) + j] = acc[i][j];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int j = 0; j < NTQ; ++j) acc[i][j] += ex[(tid * NTQ) + j];
      }
      __syncthreads();
    }
    if (grp == 1) return;
  }

  if (c6) {
#pragma unroll
    for (int i = 0; i < NTPW; ++i)
#pragma unroll
      for (int j = 0; j < NTQ; ++j) acc[i][j] = acc[i][j] * 6.f;
  }
  // D[row = 4*(lane>>4) + reg -> p][col = lane&15 -> q]
  const int lg = lane >> 4, lc = lane & 15;
  if (a.ws) {
    float* wsb = a.ws + ((size_t)split * ntiles + tile) * (PB * QB);
#pragma unroll
    for (int i = 0; i < NTPW; ++i)
#pragma unroll
      for (int j = 0; j < NTQ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          wsb[((wave + 4 * i) * 16 + lg * 4 + r) * QB + j * 16 + lc] = acc[i][j][r];
    return;
  }
#pragma unroll
  for (int i = 0; i < NTPW; ++i)
#pragma unroll
    for (int j = 0; j < NTQ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = p0 + (wave + 4 * i) * 16 + lg * 4 + r, q = q0 + j * 16 + lc;
        if (p < P && q < Q) {
          const int n = SWAP ? q : p, k = SWAP ? p : q;
          unsafeAtomicAdd(a.dw + (size_t)n * a.K + k, acc[i][j][r]);
        }
      }
}

// dw[n][k] += sum over the S pixel splits of the partial tiles, in a FIXED order (bit-reproducible weight gradients): the
// split range is divided over SP thread groups of one workgroup (not over workgroups adding atomically, as rounds 1-3a did:
// up to 16 fp32 atomics per element in arrival order), each thread sums its partials i = sp, sp + SP, ... and the SP sums
// meet in LDS in index order.  Every element has one owner, so the final add into dW is a plain read-modify-write.
template <int SP>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int N, int K,
                                                           int PB, int QB, int qtiles, int tiles, int S, int assign) {
  // straight orientation (P = N, Q = K): threads run along q = k, the contiguous axis of both the partial tiles and dW
  constexpr int EL = 256 / SP;
  __shared__ float red[SP][EL];
  const int Qpad = qtiles * QB;
  const int el = threadIdx.x % EL, sp = threadIdx.x / EL;
  const int e = blockIdx.x * EL + el;
  const int p = e / Qpad, q = e % Qpad;
  const bool live = p < N && q < K;
  float s = 0.f;
  if (live) {
    const float* src = ws + (size_t)((p / PB) * qtiles + q / QB) * (PB * QB) + (size_t)(p % PB) * QB + q % QB;
    const size_t stride = (size_t)tiles * PB * QB;
#pragma unroll 8
    for (int i = sp; i < S; i += SP) s += src[(size_t)i * stride];
  }
  if (SP > 1) {
    red[sp][el] = s;
    __syncthreads();
    if (sp == 0) {
#pragma unroll
      for (int j = 1; j < SP; ++j) s += red[j][el];
    }
  }
  if (sp == 0 && live) dw[(size_t)p * K + q] = assign ? s : dw[(size_t)p * K + q] + s;
}

// the transposed orientation (dW [N][K] with the partial tiles' contiguous axis q = n): a 32 x 32 patch per workgroup goes
// through LDS, so the partial reads stay coalesced along q AND the adds into dW run along k -- lanes along q added at a
// stride of K floats (64 cache lines per wave instruction; 1.2 M such atomics for the 320 x 960 layer of the 7x7 stage:
// 47 us, now a few).  1024 threads = 32 (q) x 32 / SP (p rows, SP of them per thread) x SP split groups.
template <int SG>
__global__ __launch_bounds__(256) void wgrad_reduce_tr_kernel(const float* __restrict__ ws, float* __restrict__ dw, int N, int K,
                                                               int PB, int QB, int qtiles, int tiles, int S, int assign) {
  // 256 threads = 32 (q) x PT (p rows) x SG (split groups), PT * SG = 8: one row and S / SG partials per thread.  Many
  // splits mean a small dW (a 112x112 layer's 96 x 24): narrow patches (PT = 1 for SG = 8) give it enough workgroups --
  // with 32 x 32 patches three workgroups walked 256 partials per thread, 83 us of pure load latency.  (256 threads, not
  // 1024: a 1024-thread workgroup waits for a whole free CU beside the main stream's persistent kernels.)
  constexpr int PT = 8 / SG;
  __shared__ float part[8][33];           // [sg * PT + pl][q]
  const int Qpad = qtiles * QB, nq = (Qpad + 31) / 32;
  const int p0 = (blockIdx.x / nq) * PT, q0 = (blockIdx.x % nq) * 32;
  const int tx = threadIdx.x & 31, rest = threadIdx.x >> 5, sg = rest / PT, pl = rest % PT;
  const size_t stride = (size_t)tiles * PB * QB;
  const int q = q0 + tx, p = p0 + pl;
  float s = 0.f;
  if (p < K && q < N) {                    // (swap: P = K, Q = N)
    const float* src = ws + (size_t)((p / PB) * qtiles + q / QB) * (PB * QB) + (size_t)(p % PB) * QB + q % QB;
#pragma unroll 8
    for (int i = sg; i < S; i += SG) s += src[(size_t)i * stride];
  }
  part[rest][tx] = s;
  __syncthreads();
  if (threadIdx.x < 32 * PT) {             // one element per thread, lanes along k (p) first
    const int pl2 = threadIdx.x % PT, ql = threadIdx.x / PT;
    const int k = p0 + pl2, n = q0 + ql;
    if (k < K && n < N) {
      float t = part[pl2][ql];
#pragma unroll
      for (int g = 1; g < SG; ++g) t += part[g * PT + pl2][ql];
      dw[(size_t)n * K + k] = assign ? t : dw[(size_t)n * K + k] + t;
    }
  }
}

template <int NTPW, int NTQ, bool SWAP, int D, int GEN, bool YF = false, int SK = 1, bool CV = false>
int launch_d(WgtArgs& a, hipStream_t st) {
  constexpr int G = 2;
  constexpr int STEP = 32 * SK;
  constexpr int PB = 64 * NTPW, QB = 16 * NTQ;
  const int P = a.swap ? a.K : a.N, Q = a.swap ? a.N : a.K;
  a.ptiles = cdiv(P, PB);
  a.qtiles = cdiv(Q, QB);
  const int tiles = a.ptiles * a.qtiles;
  const int dyB = a.swap ? QB : PB, aB = a.swap ? PB : QB;
  size_t lds = (size_t)G * 2 * STEP * ((PB + 8) + (QB + 8)) * 2 + (size_t)(3 * dyB + 2 * aB) * 4;
  // pixel splits: fill the chip (2 blocks per CU), but keep the partial-dW flush (S * N*K atomics) below ~8 MB
  static const int tgt_env = getenv("T3D_WG_TGT_BLOCKS") ? atoi(getenv("T3D_WG_TGT_BLOCKS")) : 0;      // (sweep knob)
  // (256 = one 512-thread workgroup per CU.  Round 6 sweeps, same box: 128 / 192 / 384 / 512 for every layer 7.08 / 6.93 / 6.98 / 7.07
  // against 6.80-6.91 ms per step; 64 / 128 / 192 for the <= 14x14 layers only 7.11 / 7.02 / 6.93 against 6.89-6.90)
  const int tgt_blocks = tgt_env ? tgt_env : 256;
  const long long cap_mb = 8;
  int S = (tgt_blocks + tiles - 1) / tiles;
  const long long tile_bytes = (long long)tiles * PB * QB * 4;
  const bool use_ws = g_t3d_ws.ptr && g_t3d_ws.bytes >= tile_bytes && !T3D_ENV_SET("T3D_WG_ATOMIC");
  if (use_ws) {
    const long long fit = g_t3d_ws.bytes / tile_bytes;   // partial sets the workspace holds
    if (S > fit) S = (int)fit;
  } else {
    const long long flush_cap = (cap_mb << 20) / ((long long)a.N * a.K * 4 + 1);
    if (S > flush_cap) S = (int)(flush_cap < 1 ? 1 : flush_cap);
  }
  const int maxs = cdiv(a.M, STEP * 4 * G);
  if (S > maxs) S = maxs;
  if (S < 1) S = 1;
  a.rows_per_split = cdiv(cdiv(a.M, S), STEP) * STEP;
  S = cdiv(a.M, a.rows_per_split);
  a.gs_ns = 0;
  if constexpr (GEN == 2) {
    // the gate slice [samples of a block's pixel range][aB] has to fit LDS, the in-block sample index a 32-bit multiply-high
    const long long ns = a.rows_per_split / a.HW + 2, span = (long long)a.rows_per_split + a.HW;
    // (HW = 1: ceil(2^32 / HW) does not fit 32 bits)
    if (!a.se || a.per_sample || a.HW < 2 || ns * aB * 4 > 32 * 1024 || span * a.HW >= (1ll << 32)) return launch_d<NTPW, NTQ, SWAP, 1, 1>(a, st);
    a.gs_ns = (int)ns;
    a.mulHW = (unsigned)(((1ull << 32) + a.HW - 1) / a.HW);
    lds = ((lds + 15) & ~(size_t)15) + (size_t)ns * aB * 4;
  }
  a.ws = use_ws ? reinterpret_cast<float*>(g_t3d_ws.ptr) : nullptr;
  // (atomics into dw: an assigned-to dw is cleared first; with partial tiles the reduction writes it)
  if (a.assign && !use_ws && hipMemsetAsync(a.dw, 0, (size_t)a.N * a.K * sizeof(float), st) != hipSuccess) return T3D_ERR_LAUNCH;
  if (lds > 64 * 1024)
    (void)t3d_max_lds((const void*)pw_wgrad_tr_kernel<NTPW, NTQ, SWAP, G, D, GEN, YF, SK, false, CV>, (int)lds);
  a.nsplit = S;
  T3D_LAUNCH_TIMED((pw_wgrad_tr_kernel<NTPW, NTQ, SWAP, G, D, GEN, YF, SK, false, CV>), dim3(tiles * S), dim3(256 * G), lds, st, a);
  if (use_ws) {
    // split groups inside the workgroup: enough parallelism for a small dW with hundreds of splits
#define T3D_WGR(SPV)                                                                                                              \
  do {                                                                                                                            \
    if (SWAP)                                                                                                                     \
      T3D_LAUNCH(wgrad_reduce_tr_kernel<(SPV == 16 ? 8 : SPV)>, dim3(cdiv(a.K, 8 / (SPV == 16 ? 8 : SPV)) * cdiv(a.qtiles * QB, 32)), dim3(256), 0, st, \
                         a.ws, a.dw, a.N, a.K, PB, QB, a.qtiles, tiles, S, a.assign);                                             \
    else                                                                                                                          \
      T3D_LAUNCH(wgrad_reduce_kernel<SPV>, dim3(cdiv(a.N * a.qtiles * QB, 256 / SPV)), dim3(256), 0, st, a.ws, a.dw, a.N, \
                         a.K, PB, QB, a.qtiles, tiles, S, a.assign);                                                              \
  } while (0)
    if (S >= 64) T3D_WGR(16);
    else if (S >= 8) T3D_WGR(4);
    else T3D_WGR(1);
#undef T3D_WGR
  }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <int NTPW, int NTQ, bool SWAP>
int launch_sw(WgtArgs& a, hipStream_t st) {
  const int depth = 2;   // 2 measured best (1: -12 %, 3: -2 %)
  if (a.cv.on) {          // implicit 3x3 convolution: K = 9 C > N, i.e. the swapped orientation, wide tiles only
    if constexpr (SWAP && NTQ >= 4) return launch_d<NTPW, NTQ, true, 2, false, false, 1, true>(a, st);
    else return T3D_ERR_UNSUPPORTED;
  }
  if (a.yfree) {
    if constexpr (!SWAP) {
      const int sk_env = 0;
      constexpr int width = 64 * NTPW + 16 * NTQ;
      if constexpr (width <= 288) {
        const int sk = sk_env ? sk_env : (a.M >= (1 << 20) ? 2 : 1);
        if (sk >= 2) return launch_d<NTPW, NTQ, false, 2, false, true, 2>(a, st);
      }
      return launch_d<NTPW, NTQ, false, 2, false, true>(a, st);
    }
    else return T3D_ERR_UNSUPPORTED;
  }
  // SE layers: per-sample coefficients / gates.  Gates staged in LDS (round 6): the loads of the step after next in flight, as in the
  // plain kernel (the per-vector global gate reads of round 3 forced depth 1: two exposed round trips per 32-pixel step)
  // (GEN = 1, round 3's path, keeps them as global reads inside the staging step -- and every conditional global load there makes
  // the compiler drain ALL outstanding loads, the next steps' operands included: 2x the plain kernel's duration, isolated)
  if (a.per_sample || (a.se && T3D_ENV_SET("T3D_WG_GATE_GLOBAL"))) return launch_d<NTPW, NTQ, SWAP, 1, 1>(a, st);
  if (a.se) return launch_d<NTPW, NTQ, SWAP, 2, 2>(a, st);
  if (depth == 1) return launch_d<NTPW, NTQ, SWAP, 1, false>(a, st);
  // pixels per step (see the kernel): wider steps for the narrow tiles of the layers with many pixels per workgroup
  const int sk_env = 0;
  constexpr int width = 64 * NTPW + 16 * NTQ;
  // (isolated, B = 256: 112x112 32 -> 16: 181 -> 141 us, 32 -> 32: 174 -> 138 us; the 56x56 / 28x28 layers do not move or lose)
  if constexpr (width <= 288) {
    const int sk = sk_env ? sk_env : (a.M >= (1 << 20) ? 2 : 1);
    if (sk >= 4 && width <= 128) return launch_d<NTPW, NTQ, SWAP, 2, false, false, 4>(a, st);
    if (sk >= 2) return launch_d<NTPW, NTQ, SWAP, 2, false, false, 2>(a, st);
  }
  return launch_d<NTPW, NTQ, SWAP, 2, false>(a, st);
}

template <int NTPW, int NTQ>
int launch_cfg(WgtArgs& a, hipStream_t st) {
  return a.swap ? launch_sw<NTPW, NTQ, true>(a, st) : launch_sw<NTPW, NTQ, false>(a, st);
}



static int choose_and_launch(WgtArgs& a, hipStream_t st) {
  const int M = a.M, K = a.K, N = a.N;
  a.swap = K > N;
  const int P = a.swap ? K : N, Q = a.swap ? N : K;
  // Q side: all of Q when it fits 10 tiles, else split.  P side: the widest block tile (64*NTPW rows) that still leaves
  // every pipeline >= ~12 steps of 32 pixels when the chip is filled -- a wide tile reads each operand once but, on the
  // small-pixel-count layers, degenerates into a handful of steps followed by a large partial flush; a narrow tile
  // re-reads the (small) Q-side operand through L2 instead.
  static const int min_steps_env = getenv("T3D_WG_MIN_STEPS") ? atoi(getenv("T3D_WG_MIN_STEPS")) : 0;      // (sweep knob)
  const int min_steps = min_steps_env ? min_steps_env : 12;
  auto steps_with = [&](int ntpw, int qb) {
    const int tiles = cdiv(P, 64 * ntpw) * cdiv(Q, qb);
    int S = cdiv(256, tiles);
    if (S < 1) S = 1;
    return M / (S * 2 * 32);
  };
  if (Q <= 16) return (P <= 64 || steps_with(2, 16) < min_steps) ? launch_cfg<1, 1>(a, st) : launch_cfg<2, 1>(a, st);
  if (Q <= 32) {
    if (P <= 64 || steps_with(3, 32) < min_steps) return launch_cfg<1, 2>(a, st);
    // 96 channels on the wide side (56x56, 96 -> 24): the 192-row tile staged and multiplied a half-empty P side
    // (isolated 63 us for 231 MB; the 144-channel layer of the same stage moves 308 MB in 69)
    return P <= 128 ? launch_cfg<2, 2>(a, st) : launch_cfg<3, 2>(a, st);
  }
  if (Q <= 64) {
    // (six row tiles per wave spill outside the y-free layout -- 76-344 B of scratch at 256 registers: ResNet-50's 256 -> 64
    // bottleneck entries 223 / 118 us per launch, 12.0 -> 11.4 ms per step without it; T3D_WG_6X4=1 restores it)
    if (P > 192 && steps_with(6, 64) >= min_steps && (a.yfree || T3D_ENV_SET("T3D_WG_6X4"))) return launch_cfg<6, 4>(a, st);
    if (P > 64 && steps_with(3, 64) >= min_steps) return launch_cfg<3, 4>(a, st);
    return launch_cfg<1, 4>(a, st);
  }
  // (round 6 sweep, isolated, B = 256 at 14x14: 576 -> 96 55 -> 34 us and 384 -> 96 39 -> 30 us with the wide tile from 5 steps on;
  // 480 -> 112 40 -> 36 us from 8; the 64- and 160-column tiles keep 12: 384 -> 64 and 320 -> 1280 lose below it)
  if (Q <= 96) return (P > 64 && steps_with(3, 96) >= (min_steps_env ? min_steps : 5)) ? launch_cfg<3, 6>(a, st) : launch_cfg<1, 6>(a, st);
  // 112 channels on the narrow side (MobileNetV3-large's 14x14 stage, 480 / 672 -> 112): seven column tiles -- the ten-tile kernel
  // multiplied 30 % padding and, at three row tiles per wave, spilled (256 registers + 200-460 B of scratch)
  if (Q <= 112) return (P > 64 && steps_with(3, 112) >= (min_steps_env ? min_steps : 8)) ? launch_cfg<3, 7>(a, st) : launch_cfg<1, 7>(a, st);
  // 128 / 256 / 512 ... channels on the narrow side (ResNet-50's bottlenecks): eight column tiles, no padding -- ten multiplied
  // 25 % zeros there and spilled at three row tiles per wave
  if (Q % 128 == 0) return (P > 64 && steps_with(3, 128) >= min_steps) ? launch_cfg<3, 8>(a, st) : launch_cfg<1, 8>(a, st);
  return (P > 64 && steps_with(3, 160) >= min_steps) ? launch_cfg<3, 10>(a, st) : launch_cfg<1, 10>(a, st);
}

}  // namespace

// fixed-order sum of partial tiles (straight orientation) for the fp32 parity kernel of pwconv_wgrad.hip
int t3d_pw_wgrad_reduce(const float* ws, float* dw, int N, int K, int PB, int QB, int qtiles, int tiles, int S, hipStream_t st) {
  const int blocks = [&](int sp) { return cdiv(N * qtiles * QB, 256 / sp); }(S >= 64 ? 16 : (S >= 8 ? 4 : 1));
  if (S >= 64) T3D_LAUNCH(wgrad_reduce_kernel<16>, dim3(blocks), dim3(256), 0, st, ws, dw, N, K, PB, QB, qtiles, tiles, S, 0);
  else if (S >= 8) T3D_LAUNCH(wgrad_reduce_kernel<4>, dim3(blocks), dim3(256), 0, st, ws, dw, N, K, PB, QB, qtiles, tiles, S, 0);
  else T3D_LAUNCH(wgrad_reduce_kernel<1>, dim3(blocks), dim3(256), 0, st, ws, dw, N, K, PB, QB, qtiles, tiles, S, 0);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// bf16 path of t3d_pwconv_wgrad (pwconv_wgrad.hip keeps the fp32 parity kernel)
int t3d_pw_wgrad_tr_entry(const void* dz, const void* y, const t3d_bnbwd* bb, const void* x, const t3d_prologue* pro,
                          float* dw, int M, int HW, int K, int N, hipStream_t st) {
  WgtArgs a{};
  a.dz = dz; a.y = y; a.x = x;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma; a.per_sample = bb->per_sample;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.se = pro->se; a.act = pro->act; a.se_after = pro->se_after_act; }
  a.dw = dw; a.M = M; a.HW = HW; a.K = K; a.N = N;
  if (a.per_sample) {
    if (const int rc = t3d_fold_fallback(a.alpha, st)) return rc;
  } else {
    a.fold = t3d_take_fold(a.alpha);
  }
  return choose_and_launch(a, st);
}

// implicit 3x3 convolution weight gradient (conv3x3.hip): dw [N][9 C] (patch-column order) += dy^T * gathered act(x)
int t3d_pw_wgrad_tr_conv3(const void* dz, const void* y, const t3d_bnbwd* bb, const void* x, const t3d_prologue* pro, float* dw,
                          int B, int H, int W, int C, int N, int stride, hipStream_t st) {
  WgtArgs a{};
  a.dz = dz; a.y = y; a.x = x;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  a.dw = dw; a.M = B * Ho * Wo; a.HW = Ho * Wo; a.K = 9 * C; a.N = N;
  int lg = 0;
  while ((1 << lg) < C) ++lg;
  // m / Wo and (m / Wo) / Ho as __umulhi(n, ceil(2^32 / d)): exact while n < 2^32 / d
  if ((unsigned long long)a.M * (unsigned)(Wo > Ho ? Wo : Ho) >= (1ull << 32)) return T3D_ERR_UNSUPPORTED;
  a.cv.on = 1; a.cv.H = H; a.cv.W = W; a.cv.Ho = Ho; a.cv.Wo = Wo; a.cv.C = C; a.cv.lgC = lg; a.cv.stride = stride;
  a.cv.mulW = (unsigned)(((1ull << 32) + Wo - 1) / Wo);
  a.cv.mulH = (unsigned)(((1ull << 32) + Ho - 1) / Ho);
  a.fold = t3d_take_fold(a.alpha);
  a.assign = 1;          // dw_packed is written (by the partial-tile reduction, or cleared first): the caller need not zero it
  return choose_and_launch(a, st);
}

// ---- fused y-free backward (pwconv_yfree.hip: t3d_pwconv_bwd_yfree / _finish) ------------------------------------------
// one configuration per layer shape, shared by the launch and by the later reduction of its partial tiles
struct YfCfg { int ntpw, ntq, sk, S, rows_per_split, PB, QB; };
static bool yf_cfg(int M, int K, int N, YfCfg& c) {
  const int P = N + K + 8;
  // P <= 64: t3d_pwconv_yfree_prep2 lays wd out with 64-column rows there, the kernel's narrowest tile (ntpw = 2) stages
  // 128-column rows -- such shapes take the two-launch pair
  if (K > 32 || P > 256 || P <= 64) return false;
  c.ntq = K <= 16 ? 1 : 2;
  c.ntpw = P <= 128 ? 2 : (P <= 192 ? 3 : 4);
  static const int sk_m_env = getenv("T3D_YF_SK_M") ? atoi(getenv("T3D_YF_SK_M")) : 0, s_env = getenv("T3D_YF_S") ? atoi(getenv("T3D_YF_S")) : 0;   // (sweep knobs)
  c.sk = M >= (sk_m_env ? sk_m_env : (1 << 20)) ? 2 : 1;
  c.PB = 64 * c.ntpw; c.QB = 16 * c.ntq;
  const int step = 32 * c.sk;
  int S = s_env ? s_env : 256;
  const int maxs = cdiv(M, step * 4 * 2);
  if (S > maxs) S = maxs;
  if (S < 1) S = 1;
  c.rows_per_split = cdiv(cdiv(M, S), step) * step;
  c.S = cdiv(M, c.rows_per_split);
  return true;
}

template <int NTPW, int NTQ, int SK>
static int launch_fused(WgtArgs& a, const YfCfg& c, hipStream_t st) {
  constexpr int G = 2, STEP = 32 * SK, PB = 64 * NTPW, QB = 16 * NTQ;
  a.ptiles = a.qtiles = 1;
  a.rows_per_split = c.rows_per_split;
  a.nsplit = c.S;
  const int ncoef = 3 * PB + 2 * QB;
  const size_t lds = (size_t)G * 2 * STEP * ((PB + 8) + (QB + 8)) * 2 + (size_t)((ncoef + 3) & ~3) * 4 + (size_t)QB * (PB + 8) * 2 +
                     (size_t)2 * QB * sizeof(double);
  const void* fn = (const void*)pw_wgrad_tr_kernel<NTPW, NTQ, false, G, 2, false, true, SK, true>;
  if (lds > 64 * 1024) (void)t3d_max_lds(fn, (int)lds);
  T3D_LAUNCH_TIMED((pw_wgrad_tr_kernel<NTPW, NTQ, false, G, 2, false, true, SK, true>), dim3(c.S), dim3(256 * G), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// bytes of scratch one fused launch needs: partial tiles [S][PB][QB] fp32 + the product matrix [(N + K + 8)][K] fp32
size_t t3d_pw_bwd_yfree_scratch(int M, int K, int N) {
  YfCfg c;
  if (!yf_cfg(M, K, N, c)) return 0;
  return (size_t)c.S * c.PB * c.QB * 4 + (((size_t)(N + K + 8) * K * 4 + 255) & ~(size_t)255);
}

// dx = [dz | x | 1] Wd^T (+ residual) AND the partial tiles of [dz | x | 1]^T x into `scratch` (main stream)
int t3d_pw_bwd_yfree_launch(const void* dz, const void* x, const void* wd, const void* wt, const t3d_bnbwd* bb, const void* x_raw,
                            const void* residual, void* dx, double* stats, void* scratch, int M, int HW, int K, int N, hipStream_t st) {
  YfCfg c;
  if (!yf_cfg(M, K, N, c)) return T3D_ERR_UNSUPPORTED;
  WgtArgs a{};
  a.dz = dz; a.x = x; a.y = dz;
  if (!wd) {              // weight rows built in the prologue (from the transposed weights + BatchNorm-backward coefficients)
    a.wtr = reinterpret_cast<const bf16_t*>(wt);
    a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma;
    a.fold = t3d_take_fold(bb->alpha);
  }
  a.M = M; a.HW = HW; a.K = K;
  a.yfree = 1; a.Nz = N; a.N = N + K + 8;
  a.ws = reinterpret_cast<float*>(scratch);
  a.wd = reinterpret_cast<const bf16_t*>(wd);
  a.e_res = residual; a.e_y = x_raw; a.dx = dx; a.stats = stats;
  a.nrep = g_t3d_reduce.nrep; a.rstride = g_t3d_reduce.stats_stride;
  if (a.stats && a.nrep < 1) { a.nrep = 1; a.rstride = 0; }
#define T3D_YFL(P_, Q_, S_) return launch_fused<P_, Q_, S_>(a, c, st)
  if (c.ntpw == 2 && c.ntq == 1) { if (c.sk == 2) T3D_YFL(2, 1, 2); else T3D_YFL(2, 1, 1); }
  if (c.ntpw == 2 && c.ntq == 2) { if (c.sk == 2) T3D_YFL(2, 2, 2); else T3D_YFL(2, 2, 1); }
  if (c.ntpw == 3 && c.ntq == 1) { if (c.sk == 2) T3D_YFL(3, 1, 2); else T3D_YFL(3, 1, 1); }
  if (c.ntpw == 3 && c.ntq == 2) { if (c.sk == 2) T3D_YFL(3, 2, 2); else T3D_YFL(3, 2, 1); }
  if (c.ntpw == 4 && c.ntq == 2) { if (c.sk == 2) T3D_YFL(4, 2, 2); else T3D_YFL(4, 2, 1); }
  if (c.ntpw == 4 && c.ntq == 1) { if (c.sk == 2) T3D_YFL(4, 1, 2); else T3D_YFL(4, 1, 1); }
#undef T3D_YFL
  return T3D_ERR_UNSUPPORTED;
}

// partial tiles of a fused launch -> tmp [(N + K + 8)][K] (fixed-order sum; tmp lives behind the tiles in `scratch`)
int t3d_pw_bwd_yfree_reduce(void* scratch, float** tmp_out, int M, int K, int N, hipStream_t st) {
  YfCfg c;
  if (!yf_cfg(M, K, N, c)) return T3D_ERR_UNSUPPORTED;
  float* ws = reinterpret_cast<float*>(scratch);
  float* tmp = ws + (size_t)c.S * c.PB * c.QB;
  const int rows = N + K + 8;
  // (every entry of tmp has one owner thread that WRITES it: no clear ahead of the launch)
#define T3D_YFR(SPV) T3D_LAUNCH(wgrad_reduce_kernel<SPV>, dim3(cdiv(rows * c.QB, 256 / SPV)), dim3(256), 0, st, ws, tmp, rows, K, c.PB, c.QB, 1, 1, c.S, 1)
  if (c.S >= 64) T3D_YFR(16);
  else if (c.S >= 8) T3D_YFR(4);
  else T3D_YFR(1);
#undef T3D_YFR
  T3D_CHECK_LAUNCH();
  *tmp_out = tmp;
  return T3D_OK;
}

// y-free weight-gradient products (pwconv_yfree.hip):  tmp[(N + K + 8)][K] += [dz | x | 1]^T x   -- rows 0..N-1 = dz^T x,
// rows N..N+K-1 = the Gram matrix x^T x, row N+K = the column sums of x.  Raw bf16 operands, no BatchNorm transform.
int t3d_pw_wgrad_tr_yfree(const void* dz, const void* x, float* tmp, int M, int HW, int K, int N, hipStream_t st) {
  WgtArgs a{};
  a.dz = dz; a.x = x; a.y = dz;
  a.dw = tmp; a.M = M; a.HW = HW; a.K = K;
  a.yfree = 1; a.Nz = N; a.N = N + K + 8;
  a.assign = 1;
  return choose_and_launch(a, st);
}
