// Depthwise 3x3 convolution backward (stride 1; stride 2 at the end of the file) -- data gradient, weight gradient and the producer's
// BatchNorm-backward sums in ONE pass, barrier-free streaming kernel for gfx950 (NHWC).
//
// Reads dz, y (gradient at / raw input of the following BatchNorm) and x (raw input of the conv) once,
// writes dx once: 2*(in + out) elements = the algorithmic traffic.  Same skeleton as the forward
// (dwconv3_stream.hip): a thread owns CH channels of one column and walks down a chunk of rows with
// a prefetch ring; per row it loads the three column taps of dz, y and x (9 coalesced vector loads;
// the x-1 / x+1 overlap is served by L1/L2) and
//   dy   = alpha*dz + beta*y + gamma                       (BatchNorm backward, on load)
//   a    = act(scale*x + shift)                            (forward input, recomputed)
//   dgrad: row r of dy is scattered into three rotating accumulators (dx rows r-1, r, r+1) through the
//          flipped stencil; a finished row is multiplied by act'(.), gets the skip gradient, is stored,
//          and feeds sum(dx), sum(dx*x);
//   wgrad: dw[ky][kx] += dy[r][x] * a[r+ky-1][x+kx-1] using the previous row's activations and centre
//          gradient kept in registers (36 accumulators per thread for the thread's whole life).
// Zero padding: out-of-image columns are cancelled by zeroed stencil weights / activations, rows by
// wave-uniform skips.  Block-level reduction of dw (9*C) and the sums (2*C) through fp64 LDS accumulators; the sums leave
// as fp64 atomics once per block, the weight gradient as a plain store into the block's own slot (common.h:
// t3d_dw_flush -- bit-reproducible; fp32 atomics into replicas when the caller provides no slots).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

struct Dw3BArgs {
  const void *dz, *y, *x, *res;
  void* dx;
  const float* w;
  const float *alpha, *beta, *gamma;
  int per_sample;
  const float *scale, *shift;
  int act;
  double* stats;
  float* dw;
  int B, H, W, C;
  int rows_per_chunk, nchunks, slab, nitems;
  int nrep;
  long long rstride;
  int dw_slots;  // > 0: the weight gradient goes to one slot per workgroup (t3d_set_dw_slots; common.h: t3d_dw_flush)
  int* dw_used;
  int noflush;   // profiling ablation only (T3D_DEBUG_NOFLUSH): skip the end-of-block reduction
  const T3dFold* fold;  // requested BatchNorm-backward finalize of (alpha, beta, gamma), derived in the prologue (common.h)
};

template <typename T, int CH> using rawvec = T __attribute__((ext_vector_type(CH)));

template <typename T, int CH, int PF>
__global__ __launch_bounds__(256) void dw3_bwd_s1_kernel(const Dw3BArgs a) {
  extern __shared__ float lred[];  // end of kernel: [11][C] fp64 accumulators: dw taps 0..8, sum(dx), sum(dx*x)
  using RV = rawvec<T, CH>;
  const int CG = a.C / CH;
  int cg, ox_fixed = 0, q0, qstride;
  bool on;
  if (!a.slab) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    on = j < a.W * CG;
    cg = on ? j % CG : 0;
    ox_fixed = on ? j / CG : 0;
    q0 = blockIdx.y;
    qstride = gridDim.y;
  } else {
    cg = blockIdx.y * 64 + (threadIdx.x & 63);
    on = cg < CG;
    if (!on) cg = 0;
    q0 = blockIdx.x * 4 + (threadIdx.x >> 6);
    qstride = gridDim.x * 4;
  }
  const int c0 = cg * CH;
  const bool affine = a.scale != nullptr || a.act != T3D_ACT_NONE;
  const bool need_x = affine || a.stats != nullptr || a.dw != nullptr;

  float wt[9][CH], sc[CH], sh[CH], al[CH], be[CH], ga[CH], psum[CH], psq[CH], wacc[9][CH];
  {
    float wb[CH * 9];
    const float4* wp = reinterpret_cast<const float4*>(a.w + (size_t)c0 * 9);
#pragma unroll
    for (int i = 0; i < CH * 9 / 4; ++i) {
      const float4 q = wp[i];
      wb[4 * i] = q.x; wb[4 * i + 1] = q.y; wb[4 * i + 2] = q.z; wb[4 * i + 3] = q.w;
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      sc[i] = a.scale ? a.scale[c0 + i] : 1.f;
      sh[i] = a.scale ? a.shift[c0 + i] : 0.f;
      be[i] = a.beta[c0 + i];
      al[i] = a.per_sample ? 0.f : a.alpha[c0 + i];
      ga[i] = a.per_sample ? 0.f : a.gamma[c0 + i];
      psum[i] = psq[i] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        wt[t][i] = wb[i * 9 + t];
        wacc[t][i] = 0.f;
      }
    }
  }

  for (int q = q0; q < a.nitems && on; q += qstride) {
    int ox, rest;
    if (!a.slab) { ox = ox_fixed; rest = q; } else { ox = q % a.W; rest = q / a.W; }
    const int chunk = rest % a.nchunks, b = rest / a.nchunks;
    const size_t img = (size_t)b * a.H * a.W * a.C + c0;
    const T* __restrict__ zg = reinterpret_cast<const T*>(a.dz) + img;
    const T* __restrict__ yg = reinterpret_cast<const T*>(a.y) + img;
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x) + img;
    const T* __restrict__ rg = a.res ? reinterpret_cast<const T*>(a.res) + img : nullptr;
    T* __restrict__ dxg = reinterpret_cast<T*>(a.dx) + img;
    if (a.per_sample) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        al[i] = a.alpha[(size_t)b * a.C + c0 + i];
        ga[i] = a.gamma[(size_t)b * a.C + c0 + i];
      }
    }
    const int r0 = chunk * a.rows_per_chunk, r1 = min(a.H, r0 + a.rows_per_chunk);  // rows owned by this item
    const int ix0 = ox - 1;
    const bool cok[3] = {ix0 >= 0, true, ix0 + 2 < a.W};
    // dgrad stencil: dy column c (= x-1+c) reaches dx column x through tap kx = 2-c
    float wd[9][CH];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < CH; ++i) wd[ky * 3 + c][i] = cok[c] ? wt[ky * 3 + (2 - c)][i] : 0.f;
    int coff[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) coff[c] = min(max(ix0 + c, 0), a.W - 1) * a.C;

    RV rz[PF][3], ry[PF][3], rx[PF][3];
    auto fetch = [&](int r, int slot) {
      const size_t ro = (size_t)min(max(r, 0), a.H - 1) * a.W * a.C;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        rz[slot][c] = *reinterpret_cast<const RV*>(zg + ro + coff[c]);
        ry[slot][c] = *reinterpret_cast<const RV*>(yg + ro + coff[c]);
        if (need_x) rx[slot][c] = *reinterpret_cast<const RV*>(xg + ro + coff[c]);
      }
    };
    const int rf = r0 - 1, rl = r1;  // rows walked (inclusive): one halo row on each side
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(rf + u, u);

    static_assert(PF == 3, "accumulator roles come from the unroll index");
    float acc[3][CH];          // dx rows r-1, r, r+1 = acc[u%3], acc[(u+1)%3], acc[(u+2)%3]
    float a_prev[3][CH];       // activations of row r-1 (three columns)
    float dyc_prev[CH];        // centre gradient of row r-1 (zero when that row is not owned / outside)
    float xc_prev[CH];         // raw centre input of row r-1 (for act' and the sums)
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      acc[0][i] = acc[1][i] = acc[2][i] = 0.f;
      a_prev[0][i] = a_prev[1][i] = a_prev[2][i] = 0.f;
      dyc_prev[i] = xc_prev[i] = 0.f;
    }
    for (int base = rf; base <= rl; base += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int r = base + u;
        if (r <= rl) {
          const bool rok = r >= 0 && r < a.H;
          float dy[3][CH], av[3][CH], xc[CH];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
              // rounded to the storage precision, as the tiled kernel stages it (bit-compatible sums are not required,
              // but the two kernels should agree to rounding)
              dy[c][i] = rok ? fmaf(al[i], (float)rz[u][c][i], fmaf(be[i], (float)ry[u][c][i], ga[i])) : 0.f;
              av[c][i] = need_x ? (float)rx[u][c][i] : 0.f;
            }
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) xc[i] = av[1][i];
          if (affine) {
#pragma unroll
            for (int c = 0; c < 3; ++c) act_affine_vec<CH>(av[c], sc, sh, a.act);
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) {   // zero padding of the activated input
            av[0][i] = (rok && cok[0]) ? av[0][i] : 0.f;
            av[1][i] = rok ? av[1][i] : 0.f;
            av[2][i] = (rok && cok[2]) ? av[2][i] : 0.f;
          }
          fetch(r + PF, u);
          float* accA = acc[u % 3];
          float* accB = acc[(u + 1) % 3];
          float* accC = acc[(u + 2) % 3];
          // ---- data gradient: dy row r -> dx rows r-1 (ky=0), r (ky=1), r+1 (ky=2)
          if (rok) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
              for (int i = 0; i < CH; ++i) {
                accA[i] = fmaf(dy[c][i], wd[c][i], accA[i]);
                accB[i] = fmaf(dy[c][i], wd[3 + c][i], accB[i]);
                accC[i] = fmaf(dy[c][i], wd[6 + c][i], accC[i]);
              }
          }
          // ---- weight gradient: owned output rows only (the halo rows belong to the neighbouring chunk)
          if (a.dw) {
            const bool own = r >= r0 && r < r1;
            float dyc[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) dyc[i] = own ? dy[1][i] : 0.f;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int i = 0; i < CH; ++i) {
                wacc[kx][i] = fmaf(dyc[i], a_prev[kx][i], wacc[kx][i]);          // ky = 0: a row r-1
                wacc[3 + kx][i] = fmaf(dyc[i], av[kx][i], wacc[3 + kx][i]);      // ky = 1: a row r
                wacc[6 + kx][i] = fmaf(dyc_prev[i], av[kx][i], wacc[6 + kx][i]);  // ky = 2: dy row r-1, a row r
              }
#pragma unroll
            for (int i = 0; i < CH; ++i) dyc_prev[i] = dyc[i];
          }
          // ---- dx row r-1 is complete
          const int iy = r - 1;
          if (iy >= r0 && iy < r1) {
            float g[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) g[i] = accA[i];
            if (affine) act_grad_affine_vec<CH>(g, xc_prev, sc, sh, a.act);
            const size_t off = ((size_t)iy * a.W + ox) * a.C;
            if (rg) {
              const RV rr = *reinterpret_cast<const RV*>(rg + off);
#pragma unroll
              for (int i = 0; i < CH; ++i) g[i] += (float)rr[i];
            }
            RV o;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
              o[i] = (T)g[i];
              const float v = (float)o[i];
              psum[i] += v;
              psq[i] = fmaf(v, xc_prev[i], psq[i]);
            }
            *reinterpret_cast<RV*>(dxg + off) = o;
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            accA[i] = 0.f;
            xc_prev[i] = xc[i];
            a_prev[0][i] = av[0][i];
            a_prev[1][i] = av[1][i];
            a_prev[2][i] = av[2][i];
          }
        }
      }
    }
  }  // item loop

  // ---- block-level reduction of the weight gradient and the BatchNorm-backward sums
  const int nred = (a.dw ? 9 : 0) + (a.stats ? 2 : 0);
  if (nred && !a.noflush) {
    double* lacc = reinterpret_cast<double*>(lred);       // [9 + 2][a.C] fp64 accumulators (common.h: t3d_dw_flush)
    for (int i = threadIdx.x; i < 11 * a.C; i += 256) lacc[i] = 0.0;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (a.dw) {
#pragma unroll
          for (int t = 0; t < 9; ++t) atomicAdd(lacc + t * a.C + c0 + i, (double)wacc[t][i]);
        }
        if (a.stats) {
          atomicAdd(lacc + 9 * a.C + c0 + i, (double)psum[i]);
          atomicAdd(lacc + 10 * a.C + c0 + i, (double)psq[i]);
        }
      }
    }
    __syncthreads();
    t3d_dw_flush<9, 256>(lacc, a.C, 0, a.C, a.dw, a.stats, a.nrep, a.rstride, a.dw_slots, (int)(blockIdx.y * gridDim.x + blockIdx.x), a.dw_used);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Two columns per thread + packed fp32 math (the production stride-1 backward).  The one-column kernel above needs
// ~310 VALU instructions per output vector and is VALU-bound at ~1.2 TB/s; here four column loads of dz / y / x feed
// two outputs (dy and the activations are formed twice per element instead of three times), every multiply-add is a
// v_pk_fma_f32, and the 9 x CH stencil weights are read from LDS per use (saves 36 registers for the two
// accumulator sets).  Padding: 0/1 masks per out-of-image column, wave-uniform row skips.
// raw buffer load of one register vector: scalar descriptor + scalar byte offset + per-lane byte offset
// relu6(s x + t) = 6 clamp01((s/6) x + t/6) in one packed instruction (v_pk_fma_f32 with the clamp modifier; bf16 storage
// only, dwconv3_stream.hip has the note): the activated operand a' = a/6 feeds the weight gradient (the 6 is applied once,
// when the accumulators are flushed) and IS the derivative mask of the own columns (0 < a' < 1)
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <typename RV>
__device__ __forceinline__ RV bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (sizeof(RV) == 4) {
    return __builtin_bit_cast(RV, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  } else if constexpr (sizeof(RV) == 8) {
    return __builtin_bit_cast(RV, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  } else {
    return __builtin_bit_cast(RV, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  }
}

template <typename RV>
__device__ __forceinline__ void bufstore(const RV& v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (sizeof(RV) == 4) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
  } else if constexpr (sizeof(RV) == 8) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 0);
  } else {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
  }
}

// ACT: the input's activation as a compile-time constant (a runtime switch in the row loop costs a scalar branch chain
// and a register-merge of all variants per row: ~10 % of the instructions)
// RES: a skip-connection gradient is added to dx (compile time: its two loads per row ride in the prefetch ring -- loaded on
// demand in the epilogue, as rounds 2-3 did, each one sat behind `s_waitcnt vmcnt(0)`, i.e. drained the whole ring twice per
// row in 10 of the 13 stride-1 launches of MobileNetV2; found in the ISA, round 4)
#ifdef T3D_DWB_WAVES
#define T3D_DWB_OCC __attribute__((amdgpu_waves_per_eu(T3D_DWB_WAVES, T3D_DWB_WAVES)))
#else
#define T3D_DWB_OCC
#endif
template <typename T, int PF, int NTH, int CH, int ACT, bool RES>
__global__ __launch_bounds__(NTH) T3D_DWB_OCC void dw3_bwd2_kernel(const Dw3BArgs a) {
  constexpr int H2 = CH / 2;
  constexpr bool C6 = std::is_same<T, bf16_t>::value && ACT == T3D_ACT_RELU6;
  extern __shared__ float lred[];       // [9][Cb] weights by tap, [3][Cb] derived coefficients; end of kernel: [11][Cb] fp64 accumulators
  using RV = rawvec<T, CH>;
  const int CG = a.C / CH, Wp = (a.W + 1) / 2;
  int cg, xp_fixed = 0, q0, qstride;
  bool on;
  if (!a.slab) {
    const int j = blockIdx.x * NTH + threadIdx.x;
    on = j < Wp * CG;
    cg = on ? j % CG : 0;
    xp_fixed = on ? j / CG : 0;
    q0 = blockIdx.y;
    qstride = gridDim.y;
  } else {
    cg = blockIdx.y * 64 + (threadIdx.x & 63);
    on = cg < CG;
    if (!on) cg = 0;
    q0 = blockIdx.x * (NTH / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar item decode
    qstride = gridDim.x * (NTH / 64);
  }
  const int c0 = cg * CH;
  const bool affine = a.scale != nullptr || ACT != T3D_ACT_NONE;
  // the block's own channel range: the whole tensor (flattened mapping) or one 64-group slab -- LDS staging and the
  // final flush touch only these channels
  const int cbase = a.slab ? blockIdx.y * 64 * CH : 0;
  const int Cb = a.slab ? min(64 * CH, a.C - cbase) : a.C;
  // stencil weights: the 2-channel variant keeps its 9 x 2 in registers for the thread's whole life (18 VGPRs; read
  // from LDS per tap they cost 9 ds_read + ~8 lgkmcnt waits per row, 7 % of the row loop's instructions); the
  // 4-channel variant has no registers to spare and reads [tap][Cb] from LDS, one ds_read_b128 per tap
  f32x2 wreg[9];
  const float* wl = lred + (c0 - cbase);
  if constexpr (CH == 2) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wreg[t] = f32x2{a.w[(size_t)c0 * 9 + t], a.w[(size_t)(c0 + 1) * 9 + t]};
  } else {
    for (int i = threadIdx.x; i < 9 * Cb; i += NTH) lred[i] = a.w[(size_t)(cbase + i % Cb) * 9 + i / Cb];
    __syncthreads();
  }

  // requested BatchNorm-backward finalize of the gradient's coefficients (alpha, beta, gamma), derived here (common.h):
  // the block's own channel range into the scratch behind the weights; one block per channel range publishes
  float* fco = lred + 9 * Cb;           // [3][Cb]
  if (a.fold) t3d_fold_block(a.fold, cbase, Cb, fco, Cb, a.slab ? blockIdx.x == 0 : (blockIdx.x == 0 && blockIdx.y == 0));
  f32x2 sc2[H2], sh2[H2], al2[H2], be2[H2], ga2[H2];
  f32x2 wacc[9][H2];
  float psum[CH], psq[CH];
#pragma unroll
  for (int h = 0; h < H2; ++h) {
    const int c = c0 + 2 * h;
    sc2[h] = f32x2{a.scale ? a.scale[c] : 1.f, a.scale ? a.scale[c + 1] : 1.f};
    sh2[h] = f32x2{a.scale ? a.shift[c] : 0.f, a.scale ? a.shift[c + 1] : 0.f};
    if (a.fold) {
      const float* fc = fco + (c - cbase);
      al2[h] = f32x2{fc[0], fc[1]};
      be2[h] = f32x2{fc[Cb], fc[Cb + 1]};
      ga2[h] = f32x2{fc[2 * Cb], fc[2 * Cb + 1]};
    } else {
      be2[h] = f32x2{a.beta[c], a.beta[c + 1]};
      al2[h] = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.alpha[c], a.alpha[c + 1]};
      ga2[h] = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.gamma[c], a.gamma[c + 1]};
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) wacc[t][h] = f32x2{0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) psum[i] = psq[i] = 0.f;
  if (a.fold) __syncthreads();          // the scratch is cleared again for the end-of-kernel reduction

  // Addressing: every row base is wave-uniform (scalar registers / SALU), the lane part (column, channel group) is a
  // 32-bit element offset computed once per item -- no 64-bit vector address arithmetic in the row loop.
  const T* __restrict__ zb = reinterpret_cast<const T*>(a.dz);
  const T* __restrict__ yb = reinterpret_cast<const T*>(a.y);
  const T* __restrict__ xb = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ rb = reinterpret_cast<const T*>(a.res);
  T* __restrict__ dxb = reinterpret_cast<T*>(a.dx);
  const size_t tbytes = (size_t)a.B * a.H * a.W * a.C * sizeof(T);
  const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(zb), 0, (int)tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(yb), 0, (int)tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, (int)tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(dxb, 0, (int)tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(rb ? rb : zb), 0, (int)tbytes, 0x00020000);
  for (int q = q0; q < a.nitems; q += qstride) {
    int xp, rest;
    if (!a.slab) { xp = xp_fixed; rest = q; } else { xp = q % Wp; rest = q / Wp; }
    const int chunk = rest % a.nchunks, b = rest / a.nchunks;          // wave-uniform
    if (a.per_sample) {
#pragma unroll
      for (int h = 0; h < H2; ++h) {
        const size_t o = (size_t)b * a.C + c0 + 2 * h;
        al2[h] = f32x2{a.alpha[o], a.alpha[o + 1]};
        ga2[h] = f32x2{a.gamma[o], a.gamma[o + 1]};
      }
    }
    const int r0 = chunk * a.rows_per_chunk, r1 = min(a.H, r0 + a.rows_per_chunk);
    const int x0 = 2 * xp;                 // columns x0 (always inside) and x0+1
    const bool validB = x0 + 1 < a.W;
    const float m[4] = {x0 - 1 >= 0 ? 1.f : 0.f, 1.f, validB ? 1.f : 0.f, x0 + 2 < a.W ? 1.f : 0.f};
    // zero padding folded into the coefficients: a column outside the image gets alpha = beta = gamma = 0 (dy = 0) and
    // scale = shift = 0 (act(0) = 0 for every supported activation) -- no mask multiplies in the row loop
    f32x2 alm[4][H2], bem[4][H2], gam[4][H2], scm[4][H2], shm[4][H2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int h = 0; h < H2; ++h) {
        const f32x2 mm = {m[c], m[c]};
        alm[c][h] = al2[h] * mm; bem[c][h] = be2[h] * mm; gam[c][h] = ga2[h] * mm;
        const f32x2 m6 = C6 ? mm * f32x2{T3D_SIXTH, T3D_SIXTH} : mm;
        scm[c][h] = sc2[h] * m6; shm[c][h] = sh2[h] * m6;
      }
    unsigned voff[4];                      // lane part of the load addresses (BYTES, unsigned: scalar base + 32-bit offset)
#pragma unroll
    for (int c = 0; c < 4; ++c) voff[c] = (unsigned)(min(max(x0 - 1 + c, 0), a.W - 1) * a.C + c0) * (unsigned)sizeof(T);
    const unsigned vst = (unsigned)(x0 * a.C + c0) * (unsigned)sizeof(T);   // store address, column x0 (x0+1: + C elements)
    const size_t imgrow = (size_t)b * a.H; // rows before this image

    RV rz[PF][4], ry[PF][4], rx[PF][4];
    RV rres[PF][RES ? 2 : 1];              // skip-connection gradient of dx row r-1 (the row that completes at dy row r), own columns
    const unsigned OOB = 0x80000000u;                        // beyond num_records of every tensor here (< 2 GB)
    const unsigned stA = on ? vst : OOB, stB = (on && validB) ? vst + (unsigned)(a.C * sizeof(T)) : OOB;
    // buffer loads: descriptor (scalar) + 32-bit scalar row offset + 32-bit lane offset -- no vector address arithmetic
    auto fetch = [&](int r, int slot) {
      const unsigned so = (unsigned)((imgrow + min(max(r, 0), a.H - 1)) * a.W * a.C * sizeof(T));   // scalar; tensors < 4 GB
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        rz[slot][c] = bufload<RV>(rsz, voff[c], so);
        ry[slot][c] = bufload<RV>(rsy, voff[c], so);
        rx[slot][c] = bufload<RV>(rsx, voff[c], so);
      }
      if constexpr (RES) {
        // (out-of-tile lanes / the missing odd column carry the dropped-lane offset of the store: they read zeros and can
        // never touch memory past the tensor)
        const unsigned sr = (unsigned)((imgrow + min(max(r - 1, 0), a.H - 1)) * a.W * a.C * sizeof(T));
        rres[slot][0] = bufload<RV>(rsr, stA, sr);
        rres[slot][1] = bufload<RV>(rsr, stB, sr);
      }
    };
    const int rf = r0 - 1, rl = r1;
#pragma unroll
    for (int u = 0; u < PF; ++u) {      // ring fill in slot order, pinned (see the note at the row loop)
      __builtin_amdgcn_sched_barrier(0);
      fetch(rf + u, u);
    }
    __builtin_amdgcn_sched_barrier(0);

    static_assert(PF % 3 == 0, "accumulator roles come from the unroll index (mod 3)");
    f32x2 accA[3][H2], accB[3][H2];     // dx rows r-1, r, r+1 of columns x0 / x0+1
    f32x2 a_prev[4][H2], dyc_prev[2][H2], xr_prev[2][H2];
#pragma unroll
    for (int h = 0; h < H2; ++h) {
#pragma unroll
      for (int r = 0; r < 3; ++r) accA[r][h] = accB[r][h] = f32x2{0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) a_prev[c][h] = f32x2{0.f, 0.f};
      dyc_prev[0][h] = dyc_prev[1][h] = xr_prev[0][h] = xr_prev[1][h] = f32x2{0.f, 0.f};
    }

    // Keeping the prefetch ring really PF rows deep takes three things the compiler does not do by itself (all found
    // in the ISA: the ring was drained by `s_waitcnt vmcnt(0..8)` once per PF rows):
    //  * the walk is padded to whole PF-row groups and the unrolled body carries no `if (r <= rl)` guard -- with
    //    several paths to the loop latch the register allocator reconciles the ring registers by v_mov copies there,
    //    i.e. copies of registers whose loads are still in flight, each behind a wait;
    //  * every read of a slot's registers is pinned ABOVE its refill (empty volatile asm + scheduling barriers):
    //    otherwise the conversions of the two edge columns are sunk below the loads, the new loads get fresh
    //    registers and are copied back later;
    //  * the ring is filled in slot order before the loop: the wait at the loop header is the more conservative of
    //    the two ways into the loop, and a reordered prologue made it a full drain.
    // Rows past rl contribute nothing (rok = false).  Worth 4-6 % on the 112x112 / 56x56 layers (isolated launches).
    // Two versions of a PF-row group (generic lambda, FAST = compile-time): the general one carries every row / halo /
    // store predicate; the FAST one is for groups whose rows are all owned, inside the image and past the two warm-up
    // rows -- no predicates, no exec-mask branches (out-of-tile lanes are steered to an out-of-range buffer offset,
    // which the hardware drops).  The predicates were a third of the row loop's instructions (SALU compares / selects
    // / branches); an item walks one general group, then FAST groups, then one or two general groups.
    const float mA = on ? 1.f : 0.f, mB = (on && validB) ? 1.f : 0.f;
    auto group = [&](auto fast_tag, const int base) {
      constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int r = base + u;
        {
          const bool rok = FAST || (r >= 0 && r < a.H && r <= rl);
          f32x2 dy[4][H2], av[4][H2], xr[2][H2];
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < H2; ++h) {
              const f32x2 z = {(float)rz[u][c][2 * h], (float)rz[u][c][2 * h + 1]};
              const f32x2 yy = {(float)ry[u][c][2 * h], (float)ry[u][c][2 * h + 1]};
              av[c][h] = f32x2{(float)rx[u][c][2 * h], (float)rx[u][c][2 * h + 1]};
              dy[c][h] = pk_fma(alm[c][h], z, pk_fma(bem[c][h], yy, gam[c][h]));
            }
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < H2; ++h) asm volatile("" : "+v"(dy[c][h]), "+v"(av[c][h]));
          f32x2 resf[2][H2];
          if constexpr (RES) {
#pragma unroll
            for (int col = 0; col < 2; ++col)
#pragma unroll
              for (int h = 0; h < H2; ++h) {
                resf[col][h] = f32x2{(float)rres[u][col][2 * h], (float)rres[u][col][2 * h + 1]};
                asm volatile("" : "+v"(resf[col][h]));
              }
          }
          __builtin_amdgcn_sched_barrier(0);
          fetch(r + PF, u);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int h = 0; h < H2; ++h) {
            xr[0][h] = av[1][h];
            xr[1][h] = av[2][h];
          }
          if (C6) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int h = 0; h < H2; ++h) av[c][h] = pk_fma_clamp01(av[c][h], scm[c][h], shm[c][h]);
          } else if (affine) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int h = 0; h < H2; ++h) av[c][h] = pk_fma(av[c][h], scm[c][h], shm[c][h]);
            switch (ACT) {
              case T3D_ACT_RELU:
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                  for (int h = 0; h < H2; ++h) av[c][h] = f32x2{fmaxf(av[c][h][0], 0.f), fmaxf(av[c][h][1], 0.f)};
                break;
              case T3D_ACT_RELU6:
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                  for (int h = 0; h < H2; ++h)
                    av[c][h] = f32x2{__builtin_amdgcn_fmed3f(av[c][h][0], 0.f, 6.f), __builtin_amdgcn_fmed3f(av[c][h][1], 0.f, 6.f)};
                break;
              case T3D_ACT_HSWISH:
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                  for (int h = 0; h < H2; ++h) {
                    const f32x2 t = av[c][h];
                    av[c][h] = f32x2{t[0] * (__builtin_amdgcn_fmed3f(t[0] + 3.f, 0.f, 6.f) * T3D_SIXTH),
                                     t[1] * (__builtin_amdgcn_fmed3f(t[1] + 3.f, 0.f, 6.f) * T3D_SIXTH)};
                  }
                break;
              default: break;
            }
          }
          if (!affine) {                       // no prologue: the raw input itself is the operand and needs its column masks
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int h = 0; h < H2; ++h) av[c][h] = av[c][h] * f32x2{m[c], m[c]};
          }
          if (!rok) {                          // out-of-image / padded row (at most 3 per item): contributes nothing
            asm volatile("" ::: "memory");     // (a real wave-uniform branch: if-converted it is 16 selects per row)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int h = 0; h < H2; ++h) dy[c][h] = av[c][h] = f32x2{0.f, 0.f};
          }
          f32x2* aA = accA[u % 3];
          f32x2* bA = accA[(u + 1) % 3];
          f32x2* cA = accA[(u + 2) % 3];
          f32x2* aB = accB[u % 3];
          f32x2* bB = accB[(u + 1) % 3];
          f32x2* cB = accB[(u + 2) % 3];
          const bool own = FAST || (r >= r0 && r < r1);
          f32x2 dycA[H2], dycB[H2];
#pragma unroll
          for (int h = 0; h < H2; ++h) {
            dycA[h] = dy[1][h];
            dycB[h] = dy[2][h];
          }
          if (!own) {                          // the two halo rows of a chunk: their weight-gradient terms belong to the neighbour
            asm volatile("" ::: "memory");
#pragma unroll
            for (int h = 0; h < H2; ++h) dycA[h] = dycB[h] = f32x2{0.f, 0.f};
          }
          // stencil taps
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              f32x2 wv[H2];
              if constexpr (CH == 4) {
                const float4 wq = *reinterpret_cast<const float4*>(wl + (ky * 3 + kx) * Cb);
                wv[0] = f32x2{wq.x, wq.y}; wv[1] = f32x2{wq.z, wq.w};
              } else {
                wv[0] = wreg[ky * 3 + kx];
              }
              // data gradient: dy row r reaches dx row r-1+ky; dy column (x + 1 - kx): local index 2-kx for A, 3-kx for B
              f32x2* dA = ky == 0 ? aA : (ky == 1 ? bA : cA);
              f32x2* dB = ky == 0 ? aB : (ky == 1 ? bB : cB);
#pragma unroll
              for (int h = 0; h < H2; ++h) {
                dA[h] = pk_fma(dy[2 - kx][h], wv[h], dA[h]);
                dB[h] = pk_fma(dy[3 - kx][h], wv[h], dB[h]);
              }
            }
          // weight gradient: dw[ky][kx] += dy[oy][x] * a[oy+ky-1][x+kx-1]; a column of A: local kx, of B: local kx+1
          if (a.dw) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int h = 0; h < H2; ++h) {
                wacc[kx][h] = pk_fma(dycA[h], a_prev[kx][h], pk_fma(dycB[h], a_prev[kx + 1][h], wacc[kx][h]));          // ky=0
                wacc[3 + kx][h] = pk_fma(dycA[h], av[kx][h], pk_fma(dycB[h], av[kx + 1][h], wacc[3 + kx][h]));          // ky=1
                wacc[6 + kx][h] = pk_fma(dyc_prev[0][h], av[kx][h], pk_fma(dyc_prev[1][h], av[kx + 1][h], wacc[6 + kx][h]));  // ky=2
              }
          }
          // dx row r-1 complete
          const int iy = r - 1;
          if (FAST || (iy >= r0 && iy < r1)) {
#pragma unroll
            for (int col = 0; col < 2; ++col) {
              const f32x2* acc = col == 0 ? aA : aB;
              float g[CH], xv[CH];
#pragma unroll
              for (int h = 0; h < H2; ++h) {
                g[2 * h] = acc[h][0]; g[2 * h + 1] = acc[h][1];
                xv[2 * h] = xr_prev[col][h][0]; xv[2 * h + 1] = xr_prev[col][h][1];
              }
              if (C6) {
                // the activated value of dx row r-1, own column (kept for the ky = 0 weight-gradient taps) is its own mask
#pragma unroll
                for (int h = 0; h < H2; ++h) {
                  const f32x2 ap = a_prev[col + 1][h];
                  g[2 * h] = (ap[0] > 0.f && ap[0] < 1.f) ? g[2 * h] : 0.f;
                  g[2 * h + 1] = (ap[1] > 0.f && ap[1] < 1.f) ? g[2 * h + 1] : 0.f;
                }
              } else if (affine) {
                float scf[CH], shf[CH];
#pragma unroll
                for (int h = 0; h < H2; ++h) {
                  scf[2 * h] = sc2[h][0]; scf[2 * h + 1] = sc2[h][1];
                  shf[2 * h] = sh2[h][0]; shf[2 * h + 1] = sh2[h][1];
                }
                act_grad_affine_vec<CH>(g, xv, scf, shf, ACT);
              }
              const size_t rowo = (imgrow + iy) * a.W * a.C;     // scalar
              const unsigned lo = col == 0 ? stA : stB;           // out-of-tile lanes / the missing odd column: dropped
              if constexpr (RES) {
#pragma unroll
                for (int h = 0; h < H2; ++h) {
                  g[2 * h] += resf[col][h][0];
                  g[2 * h + 1] += resf[col][h][1];
                }
              }
              const float mcol = col == 0 ? mA : mB;
              RV o;
#pragma unroll
              for (int i = 0; i < CH; ++i) {
                o[i] = (T)g[i];
                const float v = (float)o[i] * mcol;               // a lane without this column adds nothing to the sums
                psum[i] += v;
                psq[i] = fmaf(v, xv[i], psq[i]);
              }
              bufstore<RV>(o, rsd, lo, (unsigned)(rowo * sizeof(T)));
            }
          }
#pragma unroll
          for (int h = 0; h < H2; ++h) {
            aA[h] = aB[h] = f32x2{0.f, 0.f};
            xr_prev[0][h] = xr[0][h];
            xr_prev[1][h] = xr[1][h];
            dyc_prev[0][h] = dycA[h];
            dyc_prev[1][h] = dycB[h];
#pragma unroll
            for (int c = 0; c < 4; ++c) a_prev[c][h] = av[c][h];
          }
        }
      }
    };
    const int rend = rf + (rl - rf + PF) / PF * PF;
    int base = rf;
    group(std::false_type{}, base);                      // rows r0-1, r0, r0+1: halo row, nothing to store yet
    base += PF;
    // FAST groups: every row in [r0+2, r1-1] (owned, stored, inside the image), and the refill rows (r + PF) need no
    // lower clamp
    for (; base + PF - 1 <= r1 - 1 && base + PF < rend; base += PF) group(std::true_type{}, base);
    for (; base < rend; base += PF) group(std::false_type{}, base);
  }  // item loop

  // ---- block-level reduction (the weights in LDS are dead now)
  __syncthreads();
  const int nred = (a.dw ? 9 : 0) + (a.stats ? 2 : 0);
  if (nred && !a.noflush) {
    double* lacc = reinterpret_cast<double*>(lred);       // [9 + 2][Cb] fp64 accumulators (common.h: t3d_dw_flush)
    for (int i = threadIdx.x; i < 11 * Cb; i += NTH) lacc[i] = 0.0;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int h = 0; h < H2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = c0 - cbase + 2 * h + e;
          if (a.dw) {
#pragma unroll
            for (int t = 0; t < 9; ++t) atomicAdd(lacc + t * Cb + c, (double)wacc[t][h][e] * (C6 ? 6.0 : 1.0));
          }
          if (a.stats) {
            atomicAdd(lacc + 9 * Cb + c, (double)psum[2 * h + e]);
            atomicAdd(lacc + 10 * Cb + c, (double)psq[2 * h + e]);
          }
        }
    }
    __syncthreads();
    t3d_dw_flush<9, NTH>(lacc, Cb, cbase, a.C, a.dw, a.stats, a.nrep, a.rstride, a.dw_slots, a.slab ? (int)blockIdx.x : (int)(blockIdx.y * gridDim.x + blockIdx.x), a.dw_used);
  }
}

template <typename T, int CH>
int launch_s1c(Dw3BArgs& a, hipStream_t st) {
  constexpr int PF = 3;
  const int CG = a.C / CH;
  const bool two_col = true;
  const int Wcols = two_col ? (a.W + 1) / 2 : a.W;
  const long long per_row_chunk = (long long)a.B * Wcols * CG;
  int nchunks = (int)((256LL * 64 * 24 + per_row_chunk - 1) / per_row_chunk);
  int max_chunks = a.H / 8;
  if (max_chunks < 1) max_chunks = 1;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  a.rows_per_chunk = cdiv(a.H, nchunks);
  a.nchunks = cdiv(a.H, a.rows_per_chunk);
  dim3 grid;
  // tools/sweep_dwb.sh: the 4-channel variant needs AGPR spill space (1 wave/SIMD) and is best with one block per CU;
  // the 2-channel variant fits 2 waves/SIMD and is best with two (every extra block is one more flush)
  static const int tb_env = getenv("T3D_DW_TB") ? atoi(getenv("T3D_DW_TB")) : 0;      // (sweep knob, tools/scratch/sweep_tb.sh)
  const int target_blocks = tb_env ? tb_env : (CH == 2 && two_col ? 512 : 256);
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  const int nth = 256;   // 512-thread blocks measured 4-5x slower (register budget)
  // slab mapping (a wave = 64 consecutive channel groups of one column) only when it wastes < 12 % of the lanes
  // (12 %: 14x14x576 -- 288 channel pairs, 5 slabs -- is 5 % faster in slabs than flattened, where every workgroup flushes
  // sums of up to 512 channels instead of 128)
  const int waste_pct = 12;
  const bool flat = CG < 64 || (cdiv(CG, 64) * 64 - CG) * 100 > waste_pct * cdiv(CG, 64) * 64;
  if (flat) {
    a.slab = 0;
    a.nitems = a.B * a.nchunks;
    const int jb = cdiv(Wcols * CG, nth);
    int gy = target_blocks / jb;
    if (gy > a.nitems) gy = a.nitems;
    if (gy < 1) gy = 1;
    grid = dim3(jb, gy);
  } else {
    a.slab = 1;
    a.nitems = Wcols * a.B * a.nchunks;
    const int ns = cdiv(CG, 64);
    int gx = target_blocks / ns;
    if (gx > cdiv(a.nitems, nth / 64)) gx = cdiv(a.nitems, nth / 64);
    if (gx < 1) gx = 1;
    grid = dim3(gx, ns);
  }
  // 32-bit buffer offsets, and the dropped-lane sentinel 0x80000000 has to stay out of range.  This return sits in front
  // of t3d_take_fold below: a refused launch must leave a pending finalize request for the tiled fallback to honour.
  if (two_col && (size_t)a.B * a.H * a.W * a.C * sizeof(T) >= (1ull << 31)) return T3D_ERR_UNSUPPORTED;
  {   // depthwise weight gradient: one slot per workgroup when the caller provides enough of them (t3d_set_dw_slots)
    const int needed = (two_col && a.slab) ? (int)grid.x : (int)(grid.x * grid.y);
    a.dw_slots = (a.dw && g_t3d_reduce.dw_slots >= needed) ? needed : 0;
    a.dw_used = a.dw ? g_t3d_reduce.dw_used : nullptr;
  }
  const size_t lds = (size_t)22 * ((two_col && a.slab) ? 64 * CH : a.C) * sizeof(float);   // [11][Cb] fp64 reduction scratch (before it: [9][Cb] weights, [3][Cb] derived coefficients)
  a.noflush = T3D_ENV_SET("T3D_DEBUG_NOFLUSH") ? 1 : 0;
  // a pending BatchNorm-backward finalize of this launch's gradient coefficients is derived in the two-column kernel
  if (two_col && !a.per_sample) {
    a.fold = t3d_take_fold(a.alpha);
  } else {
    if (const int rc = t3d_fold_fallback(a.alpha, st)) return rc;
    a.fold = nullptr;
  }
  // (forcing 3-4 waves/SIMD through launch bounds spills to scratch: 3-7x slower)
  // (a 6-row prefetch ring needs AGPR spill space -> 1 wave/SIMD: 40 % slower; PMC: VALU busy 46 %, memory unit stalled
  //  0.1 % -- the kernel is bound by the latency two resident waves per SIMD can hide)
#define T3D_DWB2(ACTV)                                                                                              \
  do {                                                                                                              \
    if (a.res) T3D_LAUNCH_TIMED((dw3_bwd2_kernel<T, PF, 256, CH, ACTV, true>), grid, dim3(256), lds, st, a);        \
    else T3D_LAUNCH_TIMED((dw3_bwd2_kernel<T, PF, 256, CH, ACTV, false>), grid, dim3(256), lds, st, a);             \
  } while (0)
  if (two_col) {
    switch (a.act) {
      case T3D_ACT_RELU: T3D_DWB2(T3D_ACT_RELU); break;
      case T3D_ACT_RELU6: T3D_DWB2(T3D_ACT_RELU6); break;
      case T3D_ACT_HSWISH: T3D_DWB2(T3D_ACT_HSWISH); break;
      default: T3D_DWB2(T3D_ACT_NONE); break;
    }
  }
  else T3D_LAUNCH_TIMED((dw3_bwd_s1_kernel<T, CH, PF>), grid, dim3(256), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <typename T>
int launch_s1(Dw3BArgs& a, hipStream_t st) {
  // 2 channels per thread: half the live registers (no AGPR spills, twice the resident waves) beats the wider
  // loads of 4 channels per thread by 5-30 % on every layer
  const int ch = 2;
  return (ch == 2 && a.C % 2 == 0) ? launch_s1c<T, 2>(a, st) : launch_s1c<T, 4>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Stride 2.  A thread owns one OUTPUT column ow of a chunk of output rows and with it the 2x2 input pixels
// (2oh..2oh+1, 2ow..2ow+1) of every step: with pad 1 each (input pixel, tap) pair then belongs to exactly one
// thread and touches only the gradients at (oh, ow), (oh, ow+1), (oh+1, ow), (oh+1, ow+1):
//   dx[2oh  ][2ow  ] = w11 T0                       dw11 += a00 T0
//   dx[2oh  ][2ow+1] = w12 T0 + w10 T1              dw12 += a01 T0, dw10 += a01 T1
//   dx[2oh+1][2ow  ] = w21 T0 + w01 N0              dw21 += a10 T0, dw01 += a10 N0
//   dx[2oh+1][2ow+1] = w22 T0 + w20 T1 + w02 N0 + w00 N1    (and the four matching dw terms)
// (T = gradient row oh, N = row oh+1; 0/1 = column ow / ow+1.)  9 + 9 packed FMAs per four input pixels, the full-
// resolution x is read once and dx written once, the quarter-resolution dz / y are read twice (neighbour column; L2).
// ACT: compile-time activation of the input (round 4: the runtime switch sat in the row loop four times per step, once per
// input pixel of the 2x2 block); C6: ReLU6 through the clamp modifier (pk_fma_clamp01)
template <typename T, int PF, int NTH, int ACT>
__global__ __launch_bounds__(NTH) void dw3_bwd_s2_kernel(const Dw3BArgs a) {
  constexpr bool C6 = std::is_same<T, bf16_t>::value && ACT == T3D_ACT_RELU6;
  // ACT < 0: the activation stays a runtime value (h-swish: its compile-time form needs 276 VGPRs -> one wave per SIMD)
  const int act_rt = ACT >= 0 ? ACT : a.act;
  constexpr int CH = 4, H2 = 2;
  extern __shared__ float lred[];       // end of kernel: [11][Cb] fp64 accumulators
  using RV = rawvec<T, CH>;
  const int CG = a.C / CH, Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
  int cg, ow_fixed = 0, q0, qstride;
  bool on;
  if (!a.slab) {
    const int j = blockIdx.x * NTH + threadIdx.x;
    on = j < Wo * CG;
    cg = on ? j % CG : 0;
    ow_fixed = on ? j / CG : 0;
    q0 = blockIdx.y;
    qstride = gridDim.y;
  } else {
    cg = blockIdx.y * 64 + (threadIdx.x & 63);
    on = cg < CG;
    if (!on) cg = 0;
    q0 = blockIdx.x * (NTH / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar item decode
    qstride = gridDim.x * (NTH / 64);
  }
  const int c0 = cg * CH;
  const bool affine = a.scale != nullptr || act_rt != T3D_ACT_NONE;
  const int cbase = a.slab ? blockIdx.y * 64 * CH : 0;
  const int Cb = a.slab ? min(64 * CH, a.C - cbase) : a.C;
  // requested BatchNorm-backward finalize of the gradient's coefficients, derived here (see dw3_bwd2_kernel)
  if (a.fold) t3d_fold_block(a.fold, cbase, Cb, lred, Cb, a.slab ? blockIdx.x == 0 : (blockIdx.x == 0 && blockIdx.y == 0));

  f32x2 sc2[H2], sh2[H2], al2[H2], be2[H2], ga2[H2];
  f32x2 wt[9][H2], wacc[9][H2];
  float psum[CH], psq[CH];
  {
    float wb[CH * 9];
    const float4* wp = reinterpret_cast<const float4*>(a.w + (size_t)c0 * 9);
#pragma unroll
    for (int i = 0; i < CH * 9 / 4; ++i) {
      const float4 q = wp[i];
      wb[4 * i] = q.x; wb[4 * i + 1] = q.y; wb[4 * i + 2] = q.z; wb[4 * i + 3] = q.w;
    }
#pragma unroll
    for (int h = 0; h < H2; ++h) {
      const int c = c0 + 2 * h;
      sc2[h] = f32x2{a.scale ? a.scale[c] : 1.f, a.scale ? a.scale[c + 1] : 1.f};
      sh2[h] = f32x2{a.scale ? a.shift[c] : 0.f, a.scale ? a.shift[c + 1] : 0.f};
      if (a.fold) {
        const float* fc = lred + (c - cbase);
        al2[h] = f32x2{fc[0], fc[1]};
        be2[h] = f32x2{fc[Cb], fc[Cb + 1]};
        ga2[h] = f32x2{fc[2 * Cb], fc[2 * Cb + 1]};
      } else {
        be2[h] = f32x2{a.beta[c], a.beta[c + 1]};
        al2[h] = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.alpha[c], a.alpha[c + 1]};
        ga2[h] = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.gamma[c], a.gamma[c + 1]};
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        wt[t][h] = f32x2{wb[(2 * h) * 9 + t], wb[(2 * h + 1) * 9 + t]};
        wacc[t][h] = f32x2{0.f, 0.f};
      }
    }
  }
  if (a.fold) __syncthreads();          // the scratch is cleared again for the end-of-kernel reduction
#pragma unroll
  for (int i = 0; i < CH; ++i) psum[i] = psq[i] = 0.f;
  float scf[CH] = {sc2[0][0], sc2[0][1], sc2[1][0], sc2[1][1]}, shf[CH] = {sh2[0][0], sh2[0][1], sh2[1][0], sh2[1][1]};
  f32x2 sc6[H2], sh6[H2];
#pragma unroll
  for (int h = 0; h < H2; ++h) { sc6[h] = sc2[h] * f32x2{T3D_SIXTH, T3D_SIXTH}; sh6[h] = sh2[h] * f32x2{T3D_SIXTH, T3D_SIXTH}; }

  const size_t obytes = (size_t)a.B * Ho * Wo * a.C * sizeof(T), ibytes = (size_t)a.B * a.H * a.W * a.C * sizeof(T);
  const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dz), 0, (int)obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.y), 0, (int)obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)ibytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, (int)ibytes, 0x00020000);
  for (int q = q0; q < a.nitems && on; q += qstride) {
    int ow, rest;
    if (!a.slab) { ow = ow_fixed; rest = q; } else { ow = q % Wo; rest = q / Wo; }
    const int chunk = rest % a.nchunks, b = rest / a.nchunks;
    const size_t oimg = (size_t)b * Ho * Wo * a.C + c0, iimg = (size_t)b * a.H * a.W * a.C + c0;
    const T* __restrict__ zg = reinterpret_cast<const T*>(a.dz) + oimg;
    const T* __restrict__ yg = reinterpret_cast<const T*>(a.y) + oimg;
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x) + iimg;
    const T* __restrict__ rg = a.res ? reinterpret_cast<const T*>(a.res) + iimg : nullptr;
    T* __restrict__ dxg = reinterpret_cast<T*>(a.dx) + iimg;
    if (a.per_sample) {
#pragma unroll
      for (int h = 0; h < H2; ++h) {
        const size_t o = (size_t)b * a.C + c0 + 2 * h;
        al2[h] = f32x2{a.alpha[o], a.alpha[o + 1]};
        ga2[h] = f32x2{a.gamma[o], a.gamma[o + 1]};
      }
    }
    const int o0 = chunk * a.rows_per_chunk, o1 = min(Ho, o0 + a.rows_per_chunk);   // output rows owned
    const int ix = 2 * ow;
    const bool colB = ix + 1 < a.W;
    const float mo1 = ow + 1 < Wo ? 1.f : 0.f, mxB = colB ? 1.f : 0.f;
    const int ocoff[2] = {ow * a.C, min(ow + 1, Wo - 1) * a.C};
    const int icoff[2] = {ix * a.C, min(ix + 1, a.W - 1) * a.C};
    // raw buffer addressing: scalar row offsets + 32-bit lane offsets (the launcher refuses tensors of 2 GB and more)
    const unsigned ob[2] = {(unsigned)(ocoff[0] + c0) * (unsigned)sizeof(T), (unsigned)(ocoff[1] + c0) * (unsigned)sizeof(T)};
    const unsigned ib[2] = {(unsigned)(icoff[0] + c0) * (unsigned)sizeof(T), (unsigned)(icoff[1] + c0) * (unsigned)sizeof(T)};
    const size_t orow0 = (size_t)b * Ho, irow0 = (size_t)b * a.H;

    auto form_dy = [&](const RV& z, const RV& yy, float mask, f32x2* out) {
#pragma unroll
      for (int h = 0; h < H2; ++h) {
        const f32x2 zf = {(float)z[2 * h], (float)z[2 * h + 1]};
        const f32x2 yf = {(float)yy[2 * h], (float)yy[2 * h + 1]};
        out[h] = pk_fma(al2[h], zf, pk_fma(be2[h], yf, ga2[h])) * f32x2{mask, mask};
      }
    };
    f32x2 T0[H2], T1[H2];
    {
      const size_t ro = (size_t)o0 * Wo * a.C;
      const RV z0 = *reinterpret_cast<const RV*>(zg + ro + ocoff[0]), z1 = *reinterpret_cast<const RV*>(zg + ro + ocoff[1]);
      const RV y0 = *reinterpret_cast<const RV*>(yg + ro + ocoff[0]), y1 = *reinterpret_cast<const RV*>(yg + ro + ocoff[1]);
      form_dy(z0, y0, 1.f, T0);
      form_dy(z1, y1, mo1, T1);
    }
    RV rz[PF][2], ry[PF][2], rx[PF][4];
    auto fetch = [&](int o, int slot) {   // everything step `o` consumes: gradient row o+1, input rows 2o, 2o+1
      const unsigned ro = (unsigned)((orow0 + min(o + 1, Ho - 1)) * Wo * a.C * sizeof(T));           // scalar row offsets
      const unsigned ra = (unsigned)((irow0 + min(2 * o, a.H - 1)) * a.W * a.C * sizeof(T));
      const unsigned rb = (unsigned)((irow0 + min(2 * o + 1, a.H - 1)) * a.W * a.C * sizeof(T));
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        rz[slot][c] = bufload<RV>(rsz, ob[c], ro);
        ry[slot][c] = bufload<RV>(rsy, ob[c], ro);
        rx[slot][c] = bufload<RV>(rsx, ib[c], ra);
        rx[slot][2 + c] = bufload<RV>(rsx, ib[c], rb);
      }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(o0 + u, u);

    for (int base = o0; base < o1; base += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int o = base + u;
        if (o < o1) {
          const float mrN = o + 1 < Ho ? 1.f : 0.f;
          const bool rowB = 2 * o + 1 < a.H;
          const float mrB = rowB ? 1.f : 0.f;
          f32x2 N0[H2], N1[H2], xr[4][H2], av[4][H2];
          form_dy(rz[u][0], ry[u][0], mrN, N0);
          form_dy(rz[u][1], ry[u][1], mrN * mo1, N1);
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int h = 0; h < H2; ++h) xr[p][h] = f32x2{(float)rx[u][p][2 * h], (float)rx[u][p][2 * h + 1]};
          fetch(o + PF, u);
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int h = 0; h < H2; ++h) av[p][h] = xr[p][h];
          if (C6) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
              for (int h = 0; h < H2; ++h) av[p][h] = pk_fma_clamp01(av[p][h], sc6[h], sh6[h]);
          } else if (affine) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
              for (int h = 0; h < H2; ++h) av[p][h] = pk_fma(av[p][h], sc2[h], sh2[h]);
            switch (act_rt) {
              case T3D_ACT_RELU:
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                  for (int h = 0; h < H2; ++h) av[p][h] = f32x2{fmaxf(av[p][h][0], 0.f), fmaxf(av[p][h][1], 0.f)};
                break;
              case T3D_ACT_RELU6:
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                  for (int h = 0; h < H2; ++h)
                    av[p][h] = f32x2{__builtin_amdgcn_fmed3f(av[p][h][0], 0.f, 6.f), __builtin_amdgcn_fmed3f(av[p][h][1], 0.f, 6.f)};
                break;
              case T3D_ACT_HSWISH:
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                  for (int h = 0; h < H2; ++h) {
                    const f32x2 t = av[p][h];
                    av[p][h] = f32x2{t[0] * (__builtin_amdgcn_fmed3f(t[0] + 3.f, 0.f, 6.f) * T3D_SIXTH),
                                     t[1] * (__builtin_amdgcn_fmed3f(t[1] + 3.f, 0.f, 6.f) * T3D_SIXTH)};
                  }
                break;
              default: break;
            }
          }
#pragma unroll
          for (int h = 0; h < H2; ++h) {      // pixels outside an odd-sized image
            av[1][h] = av[1][h] * f32x2{mxB, mxB};
            av[2][h] = av[2][h] * f32x2{mrB, mrB};
            av[3][h] = av[3][h] * f32x2{mxB * mrB, mxB * mrB};
          }
          f32x2 g[4][H2];
#pragma unroll
          for (int h = 0; h < H2; ++h) {
            g[0][h] = wt[4][h] * T0[h];
            g[1][h] = pk_fma(wt[5][h], T0[h], wt[3][h] * T1[h]);
            g[2][h] = pk_fma(wt[7][h], T0[h], wt[1][h] * N0[h]);
            g[3][h] = pk_fma(wt[8][h], T0[h], pk_fma(wt[6][h], T1[h], pk_fma(wt[2][h], N0[h], wt[0][h] * N1[h])));
          }
          if (a.dw) {
#pragma unroll
            for (int h = 0; h < H2; ++h) {
              wacc[4][h] = pk_fma(av[0][h], T0[h], wacc[4][h]);
              wacc[5][h] = pk_fma(av[1][h], T0[h], wacc[5][h]);
              wacc[3][h] = pk_fma(av[1][h], T1[h], wacc[3][h]);
              wacc[7][h] = pk_fma(av[2][h], T0[h], wacc[7][h]);
              wacc[1][h] = pk_fma(av[2][h], N0[h], wacc[1][h]);
              wacc[8][h] = pk_fma(av[3][h], T0[h], wacc[8][h]);
              wacc[6][h] = pk_fma(av[3][h], T1[h], wacc[6][h]);
              wacc[2][h] = pk_fma(av[3][h], N0[h], wacc[2][h]);
              wacc[0][h] = pk_fma(av[3][h], N1[h], wacc[0][h]);
            }
          }
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            if (((p & 1) && !colB) || ((p & 2) && !rowB)) continue;
            float gv[CH], xv[CH];
#pragma unroll
            for (int h = 0; h < H2; ++h) {
              gv[2 * h] = g[p][h][0]; gv[2 * h + 1] = g[p][h][1];
              xv[2 * h] = xr[p][h][0]; xv[2 * h + 1] = xr[p][h][1];
            }
            if (C6) {
#pragma unroll
              for (int h = 0; h < H2; ++h) {
                const f32x2 ap = av[p][h];       // (masked to 0 outside an odd-sized image: the gradient there is dropped anyway)
                gv[2 * h] = (ap[0] > 0.f && ap[0] < 1.f) ? gv[2 * h] : 0.f;
                gv[2 * h + 1] = (ap[1] > 0.f && ap[1] < 1.f) ? gv[2 * h + 1] : 0.f;
              }
            } else if (affine) {
              act_grad_affine_vec<CH>(gv, xv, scf, shf, act_rt);
            }
            const size_t off = ((size_t)(2 * o + (p >> 1)) * a.W + ix + (p & 1)) * a.C;
            if (rg) {
              const RV rr = *reinterpret_cast<const RV*>(rg + off);
#pragma unroll
              for (int i = 0; i < CH; ++i) gv[i] += (float)rr[i];
            }
            RV ov;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
              ov[i] = (T)gv[i];
              const float v = (float)ov[i];
              psum[i] += v;
              psq[i] = fmaf(v, xv[i], psq[i]);
            }
            bufstore<RV>(ov, rsd, ib[p & 1], (unsigned)((irow0 + 2 * o + (p >> 1)) * a.W * a.C * sizeof(T)));
          }
#pragma unroll
          for (int h = 0; h < H2; ++h) {
            T0[h] = N0[h];
            T1[h] = N1[h];
          }
        }
      }
    }
  }  // item loop

  const int nred = (a.dw ? 9 : 0) + (a.stats ? 2 : 0);
  if (nred && !a.noflush) {
    double* lacc = reinterpret_cast<double*>(lred);       // [9 + 2][Cb] fp64 accumulators (common.h: t3d_dw_flush)
    for (int i = threadIdx.x; i < 11 * Cb; i += NTH) lacc[i] = 0.0;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int h = 0; h < H2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = c0 - cbase + 2 * h + e;
          if (a.dw) {
#pragma unroll
            for (int t = 0; t < 9; ++t) atomicAdd(lacc + t * Cb + c, (double)wacc[t][h][e] * (C6 ? 6.0 : 1.0));
          }
          if (a.stats) {
            atomicAdd(lacc + 9 * Cb + c, (double)psum[2 * h + e]);
            atomicAdd(lacc + 10 * Cb + c, (double)psq[2 * h + e]);
          }
        }
    }
    __syncthreads();
    t3d_dw_flush<9, NTH>(lacc, Cb, cbase, a.C, a.dw, a.stats, a.nrep, a.rstride, a.dw_slots, a.slab ? (int)blockIdx.x : (int)(blockIdx.y * gridDim.x + blockIdx.x), a.dw_used);
  }
}

template <typename T>
int launch_s2(Dw3BArgs& a, hipStream_t st) {
  constexpr int CH = 4, PF = 3;
  const int CG = a.C / CH, Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
  const long long per_row_chunk = (long long)a.B * Wo * CG;
  int nchunks = (int)((256LL * 64 * 24 + per_row_chunk - 1) / per_row_chunk);
  int max_chunks = Ho / 4;
  if (max_chunks < 1) max_chunks = 1;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  a.rows_per_chunk = cdiv(Ho, nchunks);
  a.nchunks = cdiv(Ho, a.rows_per_chunk);
  dim3 grid;
  static const int tb_env = getenv("T3D_DW_TB") ? atoi(getenv("T3D_DW_TB")) : 0;      // (sweep knob, tools/scratch/sweep_tb.sh)
  const int target_blocks = tb_env ? tb_env : 512;   // (round 4 sweep, tools/scratch/sweep_tb.sh: 14x14x576 46.6 -> 41.4 us, 28x28x192 54.3 -> 51.5 against 384)
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  const int nth = 256;
  if (CG < 64) {
    a.slab = 0;
    a.nitems = a.B * a.nchunks;
    const int jb = cdiv(Wo * CG, nth);
    int gy = target_blocks / jb;
    if (gy > a.nitems) gy = a.nitems;
    if (gy < 1) gy = 1;
    grid = dim3(jb, gy);
  } else {
    a.slab = 1;
    a.nitems = Wo * a.B * a.nchunks;
    const int ns = cdiv(CG, 64);
    int gx = target_blocks / ns;
    if (gx > cdiv(a.nitems, nth / 64)) gx = cdiv(a.nitems, nth / 64);
    if (gx < 1) gx = 1;
    grid = dim3(gx, ns);
  }
  {   // depthwise weight gradient: one slot per workgroup when the caller provides enough of them (t3d_set_dw_slots)
    const int needed = a.slab ? (int)grid.x : (int)(grid.x * grid.y);
    a.dw_slots = (a.dw && g_t3d_reduce.dw_slots >= needed) ? needed : 0;
    a.dw_used = a.dw ? g_t3d_reduce.dw_used : nullptr;
  }
  const size_t lds = (size_t)22 * (a.slab ? 64 * CH : a.C) * sizeof(float);   // [11][Cb] fp64
  a.noflush = T3D_ENV_SET("T3D_DEBUG_NOFLUSH") ? 1 : 0;
  // (in front of t3d_take_fold: a refused launch leaves the pending finalize request to the tiled fallback)
  if ((size_t)a.B * a.H * a.W * a.C * sizeof(T) >= (1ull << 31)) return T3D_ERR_UNSUPPORTED;     // 32-bit buffer offsets
  if (!a.per_sample) {
    a.fold = t3d_take_fold(a.alpha);
  } else {
    if (const int rc = t3d_fold_fallback(a.alpha, st)) return rc;
    a.fold = nullptr;
  }
  switch (a.act) {
    case T3D_ACT_RELU: T3D_LAUNCH_TIMED((dw3_bwd_s2_kernel<T, PF, 256, T3D_ACT_RELU>), grid, dim3(256), lds, st, a); break;
    case T3D_ACT_RELU6: T3D_LAUNCH_TIMED((dw3_bwd_s2_kernel<T, PF, 256, T3D_ACT_RELU6>), grid, dim3(256), lds, st, a); break;
    case T3D_ACT_HSWISH: T3D_LAUNCH_TIMED((dw3_bwd_s2_kernel<T, PF, 256, -1>), grid, dim3(256), lds, st, a); break;
    default: T3D_LAUNCH_TIMED((dw3_bwd_s2_kernel<T, PF, 256, T3D_ACT_NONE>), grid, dim3(256), lds, st, a); break;
  }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

// Called by t3d_dwconv_bwd for k == 3, stride 1 or 2.
int t3d_dw3_bwd_stream(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                       const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int H,
                       int W, int C, int stride, hipStream_t st) {
  if (stride != 1 && stride != 2) return T3D_ERR_UNSUPPORTED;
  Dw3BArgs a{};
  a.dz = dz; a.y = y; a.x = x; a.res = residual; a.dx = dx; a.w = w;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma; a.per_sample = bb->per_sample;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
  a.stats = stats; a.dw = dw;
  a.B = B; a.H = H; a.W = W; a.C = C;
  if (stride == 2) {
    if (dtype == T3D_F32) return launch_s2<float>(a, st);
    if (dtype == T3D_BF16) return launch_s2<bf16_t>(a, st);
    return T3D_ERR_ARG;
  }
  if (dtype == T3D_F32) return launch_s1<float>(a, st);
  if (dtype == T3D_BF16) return launch_s1<bf16_t>(a, st);
  return T3D_ERR_ARG;
}
