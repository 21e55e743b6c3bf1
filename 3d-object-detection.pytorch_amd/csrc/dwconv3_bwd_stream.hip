// Depthwise 3x3 convolution backward, stride 1 -- data gradient, weight gradient and the producer's
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
// wave-uniform skips.  Block-level reduction of dw (9*C) and the sums (2*C) through LDS, then fp32 / fp64
// atomics once per block.
#include <cstdlib>
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
  int noflush;   // profiling ablation only (T3D_DEBUG_NOFLUSH): skip the end-of-block reduction
};

template <typename T, int CH> using rawvec = T __attribute__((ext_vector_type(CH)));

template <typename T, int CH, int PF>
__global__ __launch_bounds__(256) void dw3_bwd_s1_kernel(const Dw3BArgs a) {
  extern __shared__ float lred[];  // [11][C]: dw taps 0..8, sum(dx), sum(dx*x)
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
    for (int i = threadIdx.x; i < 11 * a.C; i += 256) lred[i] = 0.f;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (a.dw) {
#pragma unroll
          for (int t = 0; t < 9; ++t) atomicAdd(lred + t * a.C + c0 + i, wacc[t][i]);
        }
        if (a.stats) {
          atomicAdd(lred + 9 * a.C + c0 + i, psum[i]);
          atomicAdd(lred + 10 * a.C + c0 + i, psq[i]);
        }
      }
    }
    __syncthreads();
    const int rep = (blockIdx.x + blockIdx.y) % a.nrep;
    if (a.dw) {
      for (int i = threadIdx.x; i < 9 * a.C; i += 256) {
        const float v = lred[i];
        if (v != 0.f) unsafeAtomicAdd(a.dw + (size_t)rep * a.C * 9 + (size_t)(i % a.C) * 9 + i / a.C, v);
      }
    }
    if (a.stats) {
      for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
        const float v = lred[9 * a.C + i];
        if (v != 0.f) atomicAdd(a.stats + (size_t)rep * a.rstride + i, (double)v);
      }
    }
  }
}

template <typename T>
int launch_s1(Dw3BArgs& a, hipStream_t st) {
  constexpr int CH = 4, PF = 3;
  const int CG = a.C / CH;
  const long long per_row_chunk = (long long)a.B * a.W * CG;
  int nchunks = (int)((256LL * 64 * 24 + per_row_chunk - 1) / per_row_chunk);
  int max_chunks = a.H / 8;
  if (max_chunks < 1) max_chunks = 1;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  a.rows_per_chunk = cdiv(a.H, nchunks);
  a.nchunks = cdiv(a.H, a.rows_per_chunk);
  dim3 grid;
  const int target_blocks = 256 * 2;   // 2 resident blocks per CU at this register budget; every extra block is one more flush
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  if (CG < 64) {
    a.slab = 0;
    a.nitems = a.B * a.nchunks;
    const int jb = cdiv(a.W * CG, 256);
    int gy = target_blocks / jb;
    if (gy > a.nitems) gy = a.nitems;
    if (gy < 1) gy = 1;
    grid = dim3(jb, gy);
  } else {
    a.slab = 1;
    a.nitems = a.W * a.B * a.nchunks;
    const int ns = cdiv(CG, 64);
    int gx = target_blocks / ns;
    if (gx > cdiv(a.nitems, 4)) gx = cdiv(a.nitems, 4);
    if (gx < 1) gx = 1;
    grid = dim3(gx, ns);
  }
  const size_t lds = (size_t)11 * a.C * sizeof(float);
  a.noflush = getenv("T3D_DEBUG_NOFLUSH") ? 1 : 0;
  hipLaunchKernelGGL((dw3_bwd_s1_kernel<T, CH, PF>), grid, dim3(256), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

// Called by t3d_dwconv_bwd for k == 3, stride 1.
int t3d_dw3_bwd_stream(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                       const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int H,
                       int W, int C, int stride, hipStream_t st) {
  if (stride != 1) return T3D_ERR_UNSUPPORTED;
  Dw3BArgs a{};
  a.dz = dz; a.y = y; a.x = x; a.res = residual; a.dx = dx; a.w = w;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma; a.per_sample = bb->per_sample;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
  a.stats = stats; a.dw = dw;
  a.B = B; a.H = H; a.W = W; a.C = C;
  if (dtype == T3D_F32) return launch_s1<float>(a, st);
  if (dtype == T3D_BF16) return launch_s1<bf16_t>(a, st);
  return T3D_ERR_ARG;
}
