// Depthwise 5x5 convolution backward (MobileNetV3), streaming kernels in the style of dwconv3_bwd_stream.hip: data
// gradient, weight gradient and the producer's BatchNorm-backward sums in one pass, no LDS in the walk.
//
// Stride 2 (pad 2): a thread owns 2 channels of one OUTPUT column ow and with it the 2x2 input pixels
// (2oh..2oh+1, 2ow..2ow+1) of every step.  Input pixel (iy, ix) is reached by tap (ky, kx) from output
// ((iy+2-ky)/2, (ix+2-kx)/2) whenever iy-ky and ix-kx are even, i.e. ky has the parity of iy and kx that of ix:
//   even row 2oh   : ky = 0, 2, 4  <-  gradient rows oh+1, oh, oh-1          odd row 2oh+1 : ky = 1, 3  <-  oh+1, oh
// (same for columns), so the 25 (input pixel, tap) pairs of a 2x2 block touch only the 3x3 gradient neighbourhood of
// (oh, ow) and every pair belongs to exactly one thread: 25 + 25 packed FMAs per four input pixels, the full-resolution
// x read once, dx written once, the quarter-resolution dz / y read three times (neighbour columns; L1 / L2).
#include <cstdlib>
#include "common.h"

namespace {

struct Dw5BArgs {
  const void *dz, *y, *x, *res;
  void* dx;
  const float* w;   // [C][25]
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
};

template <typename T, int CH> using rawvec = T __attribute__((ext_vector_type(CH)));

template <typename T, int PF>
__global__ __launch_bounds__(256) void dw5_bwd_s2_kernel(const Dw5BArgs a) {
  constexpr int CH = 2, K = 5;
  extern __shared__ float lred[];       // end of kernel: [27][Cb] fp64 accumulators (common.h: t3d_dw_flush)
  using RV = rawvec<T, CH>;
  const int CG = a.C / CH, Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
  int cg, ow_fixed = 0, q0, qstride;
  bool on;
  if (!a.slab) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    on = j < Wo * CG;
    cg = on ? j % CG : 0;
    ow_fixed = on ? j / CG : 0;
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
  const int cbase = a.slab ? blockIdx.y * 64 * CH : 0;
  const int Cb = a.slab ? min(64 * CH, a.C - cbase) : a.C;

  f32x2 wt[25], wacc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    wt[t] = f32x2{a.w[(size_t)c0 * 25 + t], a.w[(size_t)(c0 + 1) * 25 + t]};
    wacc[t] = f32x2{0.f, 0.f};
  }
  const f32x2 sc2 = {a.scale ? a.scale[c0] : 1.f, a.scale ? a.scale[c0 + 1] : 1.f};
  const f32x2 sh2 = {a.scale ? a.shift[c0] : 0.f, a.scale ? a.shift[c0 + 1] : 0.f};
  const f32x2 be2 = {a.beta[c0], a.beta[c0 + 1]};
  f32x2 al2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.alpha[c0], a.alpha[c0 + 1]};
  f32x2 ga2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.gamma[c0], a.gamma[c0 + 1]};
  float scf[CH] = {sc2[0], sc2[1]}, shf[CH] = {sh2[0], sh2[1]};
  float psum[CH] = {0.f, 0.f}, psq[CH] = {0.f, 0.f};

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
      const size_t o = (size_t)b * a.C + c0;
      al2 = f32x2{a.alpha[o], a.alpha[o + 1]};
      ga2 = f32x2{a.gamma[o], a.gamma[o + 1]};
    }
    const int o0 = chunk * a.rows_per_chunk, o1 = min(Ho, o0 + a.rows_per_chunk);   // output rows owned
    const int ix = 2 * ow;
    const bool colB = ix + 1 < a.W;
    const float mxB = colB ? 1.f : 0.f;
    float mc[3];          // gradient columns ow-1, ow, ow+1 inside the image?
    int ocoff[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int oc = ow - 1 + c;
      mc[c] = (oc >= 0 && oc < Wo) ? 1.f : 0.f;
      ocoff[c] = min(max(oc, 0), Wo - 1) * a.C;
    }
    const int icoff[2] = {ix * a.C, min(ix + 1, a.W - 1) * a.C};

    auto form_dy = [&](const RV& z, const RV& yy, float mask) {
      const f32x2 zf = {(float)z[0], (float)z[1]}, yf = {(float)yy[0], (float)yy[1]};
      return pk_fma(al2, zf, pk_fma(be2, yf, ga2)) * f32x2{mask, mask};
    };
    auto load_row = [&](int orow, f32x2* out) {       // one gradient row (three columns), zero outside the image
      const float mr = (orow >= 0 && orow < Ho) ? 1.f : 0.f;
      const size_t ro = (size_t)min(max(orow, 0), Ho - 1) * Wo * a.C;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        out[c] = form_dy(*reinterpret_cast<const RV*>(zg + ro + ocoff[c]), *reinterpret_cast<const RV*>(yg + ro + ocoff[c]),
                         mr * mc[c]);
    };
    f32x2 D[3][3];         // gradient rows o-1, o, o+1 x columns ow-1, ow, ow+1
    load_row(o0 - 1, D[0]);
    load_row(o0, D[1]);

    RV rz[PF][3], ry[PF][3], rx[PF][4];
    auto fetch = [&](int o, int slot) {   // everything step `o` consumes: gradient row o+1, input rows 2o, 2o+1
      const size_t ro = (size_t)min(o + 1, Ho - 1) * Wo * a.C;
      const size_t ra = (size_t)min(2 * o, a.H - 1) * a.W * a.C, rb = (size_t)min(2 * o + 1, a.H - 1) * a.W * a.C;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        rz[slot][c] = *reinterpret_cast<const RV*>(zg + ro + ocoff[c]);
        ry[slot][c] = *reinterpret_cast<const RV*>(yg + ro + ocoff[c]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        rx[slot][c] = *reinterpret_cast<const RV*>(xg + ra + icoff[c]);
        rx[slot][2 + c] = *reinterpret_cast<const RV*>(xg + rb + icoff[c]);
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
#pragma unroll
          for (int c = 0; c < 3; ++c) D[2][c] = form_dy(rz[u][c], ry[u][c], mrN * mc[c]);
          f32x2 xr[4], av[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) xr[p] = f32x2{(float)rx[u][p][0], (float)rx[u][p][1]};
          fetch(o + PF, u);
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            float t[CH] = {xr[p][0], xr[p][1]};
            if (affine) act_affine_vec<CH>(t, scf, shf, a.act);
            av[p] = f32x2{t[0], t[1]};
          }
          av[1] = av[1] * f32x2{mxB, mxB};           // pixels outside an odd-sized image
          av[2] = av[2] * f32x2{mrB, mrB};
          av[3] = av[3] * f32x2{mxB * mrB, mxB * mrB};
          f32x2 g[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int pr = p >> 1, pc = p & 1;
            g[p] = f32x2{0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
              if ((ky & 1) != pr) continue;                     // compile time
              const int dr = (pr - ky + 2) / 2;                 // gradient row offset: +1, 0, -1
#pragma unroll
              for (int kx = 0; kx < K; ++kx) {
                if ((kx & 1) != pc) continue;
                const int dc = (pc - kx + 2) / 2;
                const f32x2 d = D[dr + 1][dc + 1];
                g[p] = pk_fma(wt[ky * K + kx], d, g[p]);
                if (a.dw) wacc[ky * K + kx] = pk_fma(av[p], d, wacc[ky * K + kx]);
              }
            }
          }
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            if (((p & 1) && !colB) || ((p & 2) && !rowB)) continue;
            float gv[CH] = {g[p][0], g[p][1]}, xv[CH] = {xr[p][0], xr[p][1]};
            if (affine) act_grad_affine_vec<CH>(gv, xv, scf, shf, a.act);
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
            *reinterpret_cast<RV*>(dxg + off) = ov;
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            D[0][c] = D[1][c];
            D[1][c] = D[2][c];
          }
        }
      }
    }
  }  // item loop

  const int nred = (a.dw ? 25 : 0) + (a.stats ? 2 : 0);
  if (nred) {
    double* lacc = reinterpret_cast<double*>(lred);       // [25 + 2][Cb] fp64 accumulators (common.h: t3d_dw_flush)
    for (int i = threadIdx.x; i < 27 * Cb; i += 256) lacc[i] = 0.0;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = c0 - cbase + e;
        if (a.dw) {
#pragma unroll
          for (int t = 0; t < 25; ++t) atomicAdd(lacc + t * Cb + c, (double)wacc[t][e]);
        }
        if (a.stats) {
          atomicAdd(lacc + 25 * Cb + c, (double)psum[e]);
          atomicAdd(lacc + 26 * Cb + c, (double)psq[e]);
        }
      }
    }
    __syncthreads();
    t3d_dw_flush<25, 256>(lacc, Cb, cbase, a.C, a.dw, a.stats, a.nrep, a.rstride, a.dw_slots, a.slab ? (int)blockIdx.x : (int)(blockIdx.y * gridDim.x + blockIdx.x), a.dw_used);
  }
}

// Stride 1 (pad 2): a thread owns 2 channels of one column ix and walks the rows of its chunk.  dx(iy, ix) gathers the
// 5x5 gradient window around it,  dx = sum_{ky,kx} w[ky][kx] * dy(iy+2-ky, ix+2-kx),  and the weight gradient uses the same
// window with the thread's own activated input pixel:  dw[ky][kx] += a(iy, ix) * dy(iy+2-ky, ix+2-kx)  -- 25 + 25 packed
// FMAs per pixel, no LDS in the walk.  The window lives in registers as 5 row slots whose ROLES rotate with the unroll
// index (no register moves); one new gradient row (5 columns of dz and y; the neighbours' columns come from L1) and one
// input pixel are fetched per step, two steps ahead.  Replaces the LDS-tiled kernel of dwconv_bwd.hip for k = 5, s = 1
// (MobileNetV3-large 28x28x120: 370 us at 0.52 TB/s there).
template <typename T>
__global__ __launch_bounds__(256) void dw5_bwd_s1_kernel(const Dw5BArgs a) {
  constexpr int CH = 2, K = 5, U = 5, AHEAD = 2;
  extern __shared__ float lred[];       // end of kernel: [27][Cb] fp64 accumulators (common.h: t3d_dw_flush)
  using RV = rawvec<T, CH>;
  const int CG = a.C / CH;
  int cg, ix_fixed = 0, q0, qstride;
  bool on;
  if (!a.slab) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    on = j < a.W * CG;
    cg = on ? j % CG : 0;
    ix_fixed = on ? j / CG : 0;
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
  const int cbase = a.slab ? blockIdx.y * 64 * CH : 0;
  const int Cb = a.slab ? min(64 * CH, a.C - cbase) : a.C;

  f32x2 wt[25], wacc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    wt[t] = f32x2{a.w[(size_t)c0 * 25 + t], a.w[(size_t)(c0 + 1) * 25 + t]};
    wacc[t] = f32x2{0.f, 0.f};
  }
  const f32x2 sc2 = {a.scale ? a.scale[c0] : 1.f, a.scale ? a.scale[c0 + 1] : 1.f};
  const f32x2 sh2 = {a.scale ? a.shift[c0] : 0.f, a.scale ? a.shift[c0 + 1] : 0.f};
  const f32x2 be2 = {a.beta[c0], a.beta[c0 + 1]};
  f32x2 al2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.alpha[c0], a.alpha[c0 + 1]};
  f32x2 ga2 = a.per_sample ? f32x2{0.f, 0.f} : f32x2{a.gamma[c0], a.gamma[c0 + 1]};
  float scf[CH] = {sc2[0], sc2[1]}, shf[CH] = {sh2[0], sh2[1]};
  float psum[CH] = {0.f, 0.f}, psq[CH] = {0.f, 0.f};

  for (int q = q0; q < a.nitems && on; q += qstride) {
    int ix, rest;
    if (!a.slab) { ix = ix_fixed; rest = q; } else { ix = q % a.W; rest = q / a.W; }
    const int chunk = rest % a.nchunks, b = rest / a.nchunks;
    const size_t img = (size_t)b * a.H * a.W * a.C + c0;
    const T* __restrict__ zg = reinterpret_cast<const T*>(a.dz) + img;
    const T* __restrict__ yg = reinterpret_cast<const T*>(a.y) + img;
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x) + img;
    const T* __restrict__ rg = a.res ? reinterpret_cast<const T*>(a.res) + img : nullptr;
    T* __restrict__ dxg = reinterpret_cast<T*>(a.dx) + img;
    if (a.per_sample) {
      const size_t o = (size_t)b * a.C + c0;
      al2 = f32x2{a.alpha[o], a.alpha[o + 1]};
      ga2 = f32x2{a.gamma[o], a.gamma[o + 1]};
    }
    const int r0 = chunk * a.rows_per_chunk, r1 = min(a.H, r0 + a.rows_per_chunk);   // dx rows owned
    float mc[5];          // gradient columns ix-2 .. ix+2 inside the image?
    int coff[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int oc = ix - 2 + c;
      mc[c] = (oc >= 0 && oc < a.W) ? 1.f : 0.f;
      coff[c] = min(max(oc, 0), a.W - 1) * a.C;
    }
    auto form_dy = [&](const RV& z, const RV& yy, float mask) {
      const f32x2 zf = {(float)z[0], (float)z[1]}, yf = {(float)yy[0], (float)yy[1]};
      return pk_fma(al2, zf, pk_fma(be2, yf, ga2)) * f32x2{mask, mask};
    };
    f32x2 D[5][5];        // row SLOTS: at unrolled step u the gradient row iy-2+dr sits in slot (u + dr) % 5
    auto load_row = [&](int row, f32x2* out) {       // one gradient row (five columns), zero outside the image
      const float mr = (row >= 0 && row < a.H) ? 1.f : 0.f;
      const size_t ro = (size_t)min(max(row, 0), a.H - 1) * a.W * a.C;
#pragma unroll
      for (int c = 0; c < 5; ++c)
        out[c] = form_dy(*reinterpret_cast<const RV*>(zg + ro + coff[c]), *reinterpret_cast<const RV*>(yg + ro + coff[c]),
                         mr * mc[c]);
    };
#pragma unroll
    for (int dr = 0; dr < 4; ++dr) load_row(r0 - 2 + dr, D[dr]);      // step u = 0 starts with rows iy-2 .. iy+1 in slots 0..3

    RV rz[U][5], ry[U][5], rx[U];
    auto fetch = [&](int iy, int slot) {   // what step iy consumes: gradient row iy+2, input pixel (iy, ix)
      const size_t ro = (size_t)min(iy + 2, a.H - 1) * a.W * a.C;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        rz[slot][c] = *reinterpret_cast<const RV*>(zg + ro + coff[c]);
        ry[slot][c] = *reinterpret_cast<const RV*>(yg + ro + coff[c]);
      }
      rx[slot] = *reinterpret_cast<const RV*>(xg + ((size_t)min(iy, a.H - 1) * a.W + ix) * a.C);
    };
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) fetch(r0 + u, u);

    for (int base = r0; base < r1; base += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int iy = base + u;
        if (iy < r1) {
          const float mrN = iy + 2 < a.H ? 1.f : 0.f;
          f32x2* Dn = D[(u + 4) % 5];
#pragma unroll
          for (int c = 0; c < 5; ++c) Dn[c] = form_dy(rz[u][c], ry[u][c], mrN * mc[c]);
          const f32x2 xr = {(float)rx[u][0], (float)rx[u][1]};
          fetch(iy + AHEAD, (u + AHEAD) % U);
          float t[CH] = {xr[0], xr[1]};
          if (affine) act_affine_vec<CH>(t, scf, shf, a.act);
          const f32x2 av = {t[0], t[1]};
          f32x2 g = {0.f, 0.f};
#pragma unroll
          for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              const f32x2 d = D[(u + 4 - ky) % 5][4 - kx];
              g = pk_fma(wt[ky * K + kx], d, g);
              if (a.dw) wacc[ky * K + kx] = pk_fma(av, d, wacc[ky * K + kx]);
            }
          float gv[CH] = {g[0], g[1]}, xv[CH] = {xr[0], xr[1]};
          if (affine) act_grad_affine_vec<CH>(gv, xv, scf, shf, a.act);
          const size_t off = ((size_t)iy * a.W + ix) * a.C;
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
          *reinterpret_cast<RV*>(dxg + off) = ov;
        }
      }
      // the next group of U steps starts at u = 0 again: its rows iy-2 .. iy+1 must sit in slots 0..3.  After step U-1 the
      // newest row (iy+2 of step U-1 = iy+1 of the next step) is in slot (U-1+4) % 5 = 3 and the older ones in 2, 1, 0: U = 5
      // closes the rotation exactly, nothing to move.
    }
  }  // item loop

  const int nred = (a.dw ? 25 : 0) + (a.stats ? 2 : 0);
  if (nred) {
    double* lacc = reinterpret_cast<double*>(lred);       // [25 + 2][Cb] fp64 accumulators (common.h: t3d_dw_flush)
    for (int i = threadIdx.x; i < 27 * Cb; i += 256) lacc[i] = 0.0;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = c0 - cbase + e;
        if (a.dw) {
#pragma unroll
          for (int t = 0; t < 25; ++t) atomicAdd(lacc + t * Cb + c, (double)wacc[t][e]);
        }
        if (a.stats) {
          atomicAdd(lacc + 25 * Cb + c, (double)psum[e]);
          atomicAdd(lacc + 26 * Cb + c, (double)psq[e]);
        }
      }
    }
    __syncthreads();
    t3d_dw_flush<25, 256>(lacc, Cb, cbase, a.C, a.dw, a.stats, a.nrep, a.rstride, a.dw_slots, a.slab ? (int)blockIdx.x : (int)(blockIdx.y * gridDim.x + blockIdx.x), a.dw_used);
  }
}

template <typename T>
int launch5_s1(Dw5BArgs& a, hipStream_t st) {
  constexpr int CH = 2;
  const int CG = a.C / CH;
  const long long per_row_chunk = (long long)a.B * a.W * CG;
  int nchunks = (int)((256LL * 64 * 24 + per_row_chunk - 1) / per_row_chunk);
  int max_chunks = a.H / 5;                   // every chunk re-reads 4 halo rows of the gradient
  if (max_chunks < 1) max_chunks = 1;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  static const int ch_env = getenv("T3D_DW5_CHUNKS") ? atoi(getenv("T3D_DW5_CHUNKS")) : 0;      // (sweep knobs)
  if (ch_env) nchunks = ch_env;
  a.rows_per_chunk = cdiv(a.H, nchunks);
  a.nchunks = cdiv(a.H, a.rows_per_chunk);
  static const int tb_env = getenv("T3D_DW5_TB") ? atoi(getenv("T3D_DW5_TB")) : 0;
  const int target_blocks = tb_env ? tb_env : 512;
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  dim3 grid;
  bool flat = CG < 64 || (cdiv(CG, 64) * 64 - CG) * 100 > 8 * cdiv(CG, 64) * 64;
  if (flat && CG >= 64 && (size_t)54 * a.C * sizeof(float) > 64 * 1024) flat = false;   // the reduction scratch must fit LDS
  if (flat) {
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
  {   // depthwise weight gradient: one slot per workgroup when the caller provides enough of them (t3d_set_dw_slots)
    const int needed = a.slab ? (int)grid.x : (int)(grid.x * grid.y);
    a.dw_slots = (a.dw && g_t3d_reduce.dw_slots >= needed) ? needed : 0;
    a.dw_used = a.dw ? g_t3d_reduce.dw_used : nullptr;
  }
  const size_t lds = (size_t)54 * (a.slab ? 64 * CH : a.C) * sizeof(float);   // [27][Cb] fp64
  if (lds > 64 * 1024) return T3D_ERR_UNSUPPORTED;
  T3D_LAUNCH_TIMED((dw5_bwd_s1_kernel<T>), grid, dim3(256), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <typename T>
int launch5_s2(Dw5BArgs& a, hipStream_t st) {
  constexpr int CH = 2, PF = 3;
  const int CG = a.C / CH, Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
  const long long per_row_chunk = (long long)a.B * Wo * CG;
  int nchunks = (int)((256LL * 64 * 24 + per_row_chunk - 1) / per_row_chunk);
  int max_chunks = Ho / 4;
  if (max_chunks < 1) max_chunks = 1;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  static const int ch_env = getenv("T3D_DW5_CHUNKS") ? atoi(getenv("T3D_DW5_CHUNKS")) : 0;
  if (ch_env) nchunks = ch_env;
  a.rows_per_chunk = cdiv(Ho, nchunks);
  a.nchunks = cdiv(Ho, a.rows_per_chunk);
  static const int tb_env = getenv("T3D_DW5_TB") ? atoi(getenv("T3D_DW5_TB")) : 0;
  const int target_blocks = tb_env ? tb_env : 512;
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  dim3 grid;
  bool flat = CG < 64 || (cdiv(CG, 64) * 64 - CG) * 100 > 8 * cdiv(CG, 64) * 64;
  if (flat && CG >= 64 && (size_t)54 * a.C * sizeof(float) > 64 * 1024) flat = false;   // the reduction scratch must fit LDS
  if (flat) {
    a.slab = 0;
    a.nitems = a.B * a.nchunks;
    const int jb = cdiv(Wo * CG, 256);
    int gy = target_blocks / jb;
    if (gy > a.nitems) gy = a.nitems;
    if (gy < 1) gy = 1;
    grid = dim3(jb, gy);
  } else {
    a.slab = 1;
    a.nitems = Wo * a.B * a.nchunks;
    const int ns = cdiv(CG, 64);
    int gx = target_blocks / ns;
    if (gx > cdiv(a.nitems, 4)) gx = cdiv(a.nitems, 4);
    if (gx < 1) gx = 1;
    grid = dim3(gx, ns);
  }
  {   // depthwise weight gradient: one slot per workgroup when the caller provides enough of them (t3d_set_dw_slots)
    const int needed = a.slab ? (int)grid.x : (int)(grid.x * grid.y);
    a.dw_slots = (a.dw && g_t3d_reduce.dw_slots >= needed) ? needed : 0;
    a.dw_used = a.dw ? g_t3d_reduce.dw_used : nullptr;
  }
  const size_t lds = (size_t)54 * (a.slab ? 64 * CH : a.C) * sizeof(float);   // [27][Cb] fp64
  if (lds > 64 * 1024) return T3D_ERR_UNSUPPORTED;
  T3D_LAUNCH_TIMED((dw5_bwd_s2_kernel<T, PF>), grid, dim3(256), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

// Called by t3d_dwconv_bwd for k == 5 (both strides; T3D_DW_TILED=1 keeps the LDS-tiled kernel of dwconv_bwd.hip).
int t3d_dw5_bwd_stream(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                       const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int H,
                       int W, int C, int stride, hipStream_t st) {
  if (stride != 1 && stride != 2) return T3D_ERR_UNSUPPORTED;
  if (C % 2) return T3D_ERR_UNSUPPORTED;
  if (const int rc = t3d_fold_fallback(bb->alpha, st)) return rc;     // no derive prologue in the 5x5 kernels
  Dw5BArgs a{};
  a.dz = dz; a.y = y; a.x = x; a.res = residual; a.dx = dx; a.w = w;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma; a.per_sample = bb->per_sample;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
  a.stats = stats; a.dw = dw;
  a.B = B; a.H = H; a.W = W; a.C = C;
  if (dtype == T3D_F32) return stride == 2 ? launch5_s2<float>(a, st) : launch5_s1<float>(a, st);
  if (dtype == T3D_BF16) return stride == 2 ? launch5_s2<bf16_t>(a, st) : launch5_s1<bf16_t>(a, st);
  return T3D_ERR_ARG;
}
