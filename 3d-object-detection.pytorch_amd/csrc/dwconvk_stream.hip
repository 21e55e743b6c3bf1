// Depthwise K x K convolution forward (K = 3 or 5, stride 1 or 2), the streaming skeleton of dwconv3_stream.hip made
// generic in the stencil size, plus the per-sample channel sums squeeze-excite needs (MobileNetV3: the 5x5 layers and
// every SE block went through the LDS-tiled kernel of dwconv_fwd.hip at 0.2-0.3 TB/s).
//
// A thread owns 2 channels of NC ADJACENT output columns (round 5; one column before) and walks down a chunk of rows; no LDS, no
// barriers in the walk.  The K + (NC - 1) S input columns it loads per row feed all NC outputs: 8 loads and activations per four
// outputs of a 5x5 stride-1 layer instead of 20.  Worth 8-20 % on the stride-1 layers, nothing at stride 2 (launch_ks): the walk
// is bound by its 25 packed FMAs per output pair and by registers (222 at K = 5, NC = 4) more than by the tap loads.
// Input row `rel` (counted from the first row the chunk needs) feeds output row (rel - ky) / S for every tap row ky
// with (rel - ky) % S == 0, so at most NA = ceil(K / S) output rows are open at once.  The walk is unrolled over
// U = NA * S input rows: inside the unrolled body the open rows' accumulator slots ((rel - ky) / S mod NA), the row that
// completes ((rel - K + 1) / S) and the prefetch-ring slot are all compile-time constants.
// Zero padding applies to the ACTIVATED tensor: out-of-image columns are masked per item, out-of-image rows skipped
// (wave-uniform); loads are unconditional from clamped addresses.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

struct DwkArgs {
  const void* x;
  void* y;
  const float* w;  // [C][K*K]
  const float *scale, *shift;
  int act;
  double* stats;
  float* gap;      // [B][C] per-sample sums of the (rounded) output, or null (int64 fixed point when gapq: common.h)
  int gapq;
  int B, H, W, C, Ho, Wo;
  int rows_per_chunk, nchunks, slab, nitems;
  int Wb;          // column blocks per row: ceil(Wo / NC)
  int nrep;
  long long rstride;
  T3dQuant quant;  // BatchNorm sums snapped onto a fixed grid: order-independent (common.h)
};

template <typename T, int CH> using rawvec = T __attribute__((ext_vector_type(CH)));

constexpr int floormod(int a, int n) { return ((a % n) + n) % n; }
constexpr int floordiv(int a, int n) { return (a - floormod(a, n)) / n; }

template <typename T, int K, int S, int NC>
__global__ __launch_bounds__(256) void dwk_fwd_kernel(const DwkArgs a) {
  constexpr int CH = 2, PAD = (K - 1) / 2, NA = (K + S - 1) / S, U = NA * S, KC = K + (NC - 1) * S;
  extern __shared__ double lstat[];  // [2][Cb] fp64: exact adds of the snapped partial sums (common.h)
  using RV = rawvec<T, CH>;
  const int CG = a.C / CH;
  int cg, ox_fixed = 0, q0, qstride;
  bool on;
  if (!a.slab) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    on = j < a.Wb * CG;
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

  f32x2 wt[K * K];     // the two channels' weights, packed
  float sc[CH], sh[CH], psum[CH], psq[CH];
#pragma unroll
  for (int t = 0; t < K * K; ++t) wt[t] = f32x2{a.w[(size_t)c0 * (K * K) + t], a.w[(size_t)(c0 + 1) * (K * K) + t]};
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    sc[i] = a.scale ? a.scale[c0 + i] : 1.f;
    sh[i] = a.scale ? a.shift[c0 + i] : 0.f;
    psum[i] = psq[i] = 0.f;
  }

  for (int q = q0; q < a.nitems && on; q += qstride) {
    int ox, rest;
    if (!a.slab) { ox = ox_fixed; rest = q; } else { ox = q % a.Wb; rest = q / a.Wb; }
    ox *= NC;                                   // first output column of this thread's block
    const int chunk = rest % a.nchunks, b = rest / a.nchunks;
    const T* __restrict__ x = reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.C + c0;
    T* __restrict__ y = reinterpret_cast<T*>(a.y) + (size_t)b * a.Ho * a.Wo * a.C + c0;
    const int oy0 = chunk * a.rows_per_chunk, oy1 = min(a.Ho, oy0 + a.rows_per_chunk);
    const int ix0 = ox * S - PAD;
    float cm[KC];       // 0/1: column inside the image
    int coff[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      cm[c] = (ix0 + c >= 0 && ix0 + c < a.W) ? 1.f : 0.f;
      coff[c] = min(max(ix0 + c, 0), a.W - 1) * a.C;
    }
    const int iy_first = oy0 * S - PAD, nrel = (oy1 - 1 - oy0) * S + K;   // input rows walked: rel = 0 .. nrel-1
    RV ring[U][KC];
    auto fetch = [&](int rel, RV* dst) {
      const T* rp = x + (size_t)min(max(iy_first + rel, 0), a.H - 1) * a.W * a.C;
#pragma unroll
      for (int c = 0; c < KC; ++c) dst[c] = *reinterpret_cast<const RV*>(rp + coff[c]);
    };
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(u, ring[u]);
    f32x2 acc[NA][NC];
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
      for (int oc = 0; oc < NC; ++oc) acc[r][oc] = f32x2{0.f, 0.f};
    float gs[CH] = {0.f, 0.f};

    for (int base = 0; base < nrel; base += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rel = base + u;
        if (rel < nrel) {          // wave-uniform in the slab mapping, near-uniform otherwise
          const int iy = iy_first + rel;
          const float rm = (iy >= 0 && iy < a.H) ? 1.f : 0.f;
          f32x2 v[KC];
#pragma unroll
          for (int c = 0; c < KC; ++c) {
            float t[CH] = {(float)ring[u][c][0], (float)ring[u][c][1]};
            if (affine) act_affine_vec<CH>(t, sc, sh, a.act);
            const float m = cm[c] * rm;
            v[c] = f32x2{t[0] * m, t[1] * m};
          }
          fetch(rel + U, ring[u]);   // refill this slot: U rows ahead
          // scatter the row into the open output rows
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            if (floormod(u - ky, S) == 0) {                       // compile time
              const int slot = floormod(floordiv(u - ky, S), NA); // compile time (base is a multiple of U = NA*S)
              if (rel >= ky) {                                    // output row (rel-ky)/S exists (scalar test)
#pragma unroll
                for (int oc = 0; oc < NC; ++oc)
#pragma unroll
                  for (int c = 0; c < K; ++c) acc[slot][oc] = pk_fma(v[oc * S + c], wt[ky * K + c], acc[slot][oc]);
              }
            }
          }
          // the output row whose last tap row this was
          if (floormod(u - (K - 1), S) == 0) {                    // compile time
            const int slot = floormod(floordiv(u - (K - 1), S), NA);
            if (rel >= K - 1) {
              const int oy = oy0 + (rel - (K - 1)) / S;
#pragma unroll
              for (int oc = 0; oc < NC; ++oc) {
                if (NC == 1 || ox + oc < a.Wo) {            // (the last block of a row may hold fewer columns)
                  RV o;
                  o[0] = (T)acc[slot][oc][0];
                  o[1] = (T)acc[slot][oc][1];
#pragma unroll
                  for (int i = 0; i < CH; ++i) {
                    const float r = (float)o[i];
                    psum[i] += r;
                    psq[i] = fmaf(r, r, psq[i]);
                    gs[i] += r;
                  }
                  *reinterpret_cast<RV*>(y + ((size_t)oy * a.Wo + ox + oc) * a.C) = o;
                }
              }
            }
#pragma unroll
            for (int oc = 0; oc < NC; ++oc) acc[slot][oc] = f32x2{0.f, 0.f};
          }
        }
      }
    }
    if (a.gap) {
#pragma unroll
      for (int i = 0; i < CH; ++i) t3d_pool_add(a.gap, (size_t)b * a.C + c0 + i, gs[i], a.gapq);
    }
  }  // item loop

  if (a.stats) {
    const int cbase = a.slab ? blockIdx.y * 64 * CH : 0;
    const int Cb = a.slab ? min(64 * CH, a.C - cbase) : a.C;
    for (int i = threadIdx.x; i < 2 * Cb; i += 256) lstat[i] = 0.0;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        atomicAdd(lstat + c0 - cbase + i, t3d_snap(psum[i], a.quant, false));
        atomicAdd(lstat + Cb + c0 - cbase + i, t3d_snap(psq[i], a.quant, true));
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cb; i += 256)
      if (lstat[i] != 0.0)
        atomicAdd(a.stats + (size_t)((blockIdx.x + blockIdx.y) % a.nrep) * a.rstride + (size_t)(i / Cb) * a.C + cbase + i % Cb,
                  lstat[i]);
  }
}

template <typename T, int K, int S, int NC>
int launch_nc(DwkArgs& a, hipStream_t st) {
  constexpr int CH = 2;
  const int CG = a.C / CH;
  a.Wb = cdiv(a.Wo, NC);
  const long long per_row_chunk = (long long)a.B * a.Wb * CG;
  // row chunks per image: enough (thread, item) pairs to fill the chip once (~1600 waves) and no more -- every chunk re-reads
  // K - 1 halo rows and pays the ring fill again.  (Round 6, isolated at B = 256: 28x28x120 5x5 s1 with four columns per
  // thread ran 79 us in three chunks, 58 in one; the 40-fold target dates from the one-column kernel.)
  const long long want = (K == 5 && S == 1 && NC > 1) ? 256LL * 64 * 6 : 256LL * 64 * 40;
  int nchunks = (int)((want + per_row_chunk - 1) / per_row_chunk);
  int max_chunks = a.Ho / 8;
  if (max_chunks < 1) max_chunks = 1;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  static const int ch_env = getenv("T3D_DWK_CHUNKS") ? atoi(getenv("T3D_DWK_CHUNKS")) : 0;      // (sweep knobs)
  if (ch_env) nchunks = ch_env;
  a.rows_per_chunk = cdiv(a.Ho, nchunks);
  a.nchunks = cdiv(a.Ho, a.rows_per_chunk);
  static const int tb_env = getenv("T3D_DWK_TB") ? atoi(getenv("T3D_DWK_TB")) : 0;
  const int target_blocks = tb_env ? tb_env : ((K == 5 && S == 1) ? 1024 : 768);
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  dim3 grid;
  const bool flat = CG < 64 || (cdiv(CG, 64) * 64 - CG) * 100 > 8 * cdiv(CG, 64) * 64;
  if (flat) {
    a.slab = 0;
    a.nitems = a.B * a.nchunks;
    const int jb = cdiv(a.Wb * CG, 256);
    int gy = target_blocks / jb;
    if (gy > a.nitems) gy = a.nitems;
    if (gy < 1) gy = 1;
    grid = dim3(jb, gy);
  } else {
    a.slab = 1;
    a.nitems = a.Wb * a.B * a.nchunks;
    const int ns = cdiv(CG, 64);
    int gx = target_blocks / ns;
    if (gx > cdiv(a.nitems, 4)) gx = cdiv(a.nitems, 4);
    if (gx < 1) gx = 1;
    grid = dim3(gx, ns);
  }
  const size_t lds = (size_t)2 * (a.slab ? 64 * CH : a.C) * sizeof(double);
  // (throughput mode only, as in dwconv3_stream.hip)
  a.quant = (a.stats && std::is_same<T, bf16_t>::value && !T3D_ENV_SET("T3D_NO_SNAP")) ? t3d_quant_for((long long)a.B * a.Ho * a.Wo)
                                                                                  : T3dQuant{0.0, 0.0};
  T3D_LAUNCH_TIMED((dwk_fwd_kernel<T, K, S, NC>), grid, dim3(256), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <typename T, int K, int S>
int launch_ks(DwkArgs& a, hipStream_t st) {
  // output columns per thread, measured per layer of mobilenetv3_large at B = 256 (tools/scratch/time_dwk.py, us at NC = 1 / 2 / 4):
  //   3x3 s1 14^2 x 480 / 672 (squeeze-excite): 62.6 / 50.1 / 49.2 and 77.9 / 61.4 / 57.3;  5x5 s1 28^2 x 120: 93.7 / 105.7 / 77.8,
  //   7^2 x 960: 66.3 / 78.6 / 62.8;  5x5 s2 56^2 x 72: 93.1 / 95.3 (/ 93.8 at NC = 2), 14^2 x 672: 69.2 / 84.9 -- four columns for
  //   stride 1, one for stride 2 (T3D_DWK_NC: sweep knob)
  static const int env = getenv("T3D_DWK_NC") ? atoi(getenv("T3D_DWK_NC")) : 0;
  const int nc = env ? env : (S == 1 ? 4 : 1);
  if (nc == 1 || a.Wo < 4) return launch_nc<T, K, S, 1>(a, st);
  if (nc == 4 && S == 1) return launch_nc<T, K, S, (S == 1 ? 4 : 2)>(a, st);
  return launch_nc<T, K, S, 2>(a, st);
}

template <typename T>
int launch_t(DwkArgs& a, int k, int s, hipStream_t st) {
  if (k == 5 && s == 1) return launch_ks<T, 5, 1>(a, st);
  if (k == 5 && s == 2) return launch_ks<T, 5, 2>(a, st);
  if (k == 3 && s == 1) return launch_ks<T, 3, 1>(a, st);
  if (k == 3 && s == 2) return launch_ks<T, 3, 2>(a, st);
  return T3D_ERR_UNSUPPORTED;
}

}  // namespace

// Called by t3d_dwconv_fwd for k = 5, and for k = 3 with squeeze-excite pooled sums (no SE gate on the INPUT).
int t3d_dwk_fwd_stream(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats,
                       float* gap_sum, int B, int H, int W, int C, int k, int stride, hipStream_t st) {
  DwkArgs a{};
  a.x = x; a.y = y; a.w = w; a.stats = stats; a.gap = gap_sum; a.gapq = g_t3d_reduce.pool_exact;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
  a.B = B; a.H = H; a.W = W; a.C = C;
  const int pad = (k - 1) / 2;
  a.Ho = (H + 2 * pad - k) / stride + 1;
  a.Wo = (W + 2 * pad - k) / stride + 1;
  if (dtype == T3D_F32) return launch_t<float>(a, k, stride, st);
  if (dtype == T3D_BF16) return launch_t<bf16_t>(a, k, stride, st);
  return T3D_ERR_ARG;
}
