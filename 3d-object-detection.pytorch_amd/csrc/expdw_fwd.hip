// Fused expand (1x1) + BatchNorm + activation + depthwise 3x3 forward of an inverted-residual block, bf16 storage, gfx950
// (round 5, second version; the first one -- tools/scratch/pruned_r4/expdw_fwd.hip -- lost to the two-launch pair).
//
//   y2[b, oy, ox, c] = sum_{ky,kx} w_dw[c][ky][kx] * a[b, S*oy-1+ky, S*ox-1+kx, c],   a = act(scale1[c] * bf16((W1 z)[c]) + shift1[c])
//
// (models/mobilenetv3.py:146-153: conv 1x1 -> BN -> act -> depthwise conv -> BN; the expanded tensor is 6x wider than the
// block input z and is the largest tensor of the network: 616 MB at 112x112x96 for a batch of 256.)  Layer by layer it costs a
// write and a read of HBM in the forward; here the depthwise stencil reads it out of LDS, and the raw expansion is stored
// only when a backward is going to read it (y1 != NULL: training; inference stores nothing).
//
// Work item = (image, tile of TH output rows, slab of CS = 32 expanded channels); a workgroup keeps ONE slab for its whole
// life (expansion weights, BatchNorm coefficients, stencil weights and the partial sums of the next BatchNorm stay in
// registers) and the slab workgroups of one tile sit on one XCD (they read the same narrow rows: its L2 serves all but one).
//   phase 1  the waves expand the tile's input rows (+ halo rows) on the matrix cores -- v_mfma_f32_16x16x32_bf16, A = W1
//            rows permuted so that a lane ends up with 8 CONSECUTIVE channels of one pixel (two 16-channel tiles), B = z
//            fragments fetched from HBM during the PREVIOUS item's stencil phase -- round to bf16 (the stored value: what
//            the backward will recompute the activation from), store 16 B of y1, apply BatchNorm + ReLU6 as
//            6 * clamp01((s/6) y + t/6) (one packed instruction per channel pair, the 6 folded into the stencil weights:
//            the form the depthwise kernels use, DESIGN.md finding 30) and park the fp32 values in the LDS tile
//            [row][column + zero frame][CS + 4]  (~2.5 vector instructions per element; the first version spent ~10);
//   phase 2  a thread owns (4 channels, one output column) and walks DOWN the tile's input rows: three 16-B LDS reads per
//            row feed up to three open output rows (rotating accumulators, packed fp32 FMAs), a finished row is rounded,
//            stored (8 B; eight threads write a pixel's 64 contiguous bytes) and added to the BatchNorm sums in registers.
// The batch statistics of the expansion's own BatchNorm must exist BEFORE this kernel (they are global over the batch): the
// caller runs the 1x1 conv as a statistics-only pass first (t3d_pwconv_fwd[_mat] with y = NULL: reads the narrow tensor,
// stores nothing) and finalizes them.
#include <cstdlib>
#include "common.h"

namespace {

struct EdArgs {
  const void *z, *w1;
  const float *sc1, *sh1;
  const float* wdw;
  void *y1, *y2;
  double* stats;
  int B, H, W, K, C, Ho, Wo;
  int TH, tiles_per_img, nslab, nitems;
  int nrep;
  long long rstride;
  T3dQuant quant;
};

__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// storage types: bf16 (training + inference) and fp16 (inference: three more mantissa bits at every stored layer)
template <typename T> struct St;
template <> struct St<bf16_t> {
  typedef bf16x8 V8; typedef bf16x4 V4;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct St<f16_t> {
  typedef f16x8 V8; typedef _Float16 __attribute__((ext_vector_type(4))) V4;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int CS = 32;           // expanded channels per slab (two MFMA tiles)
constexpr int PS = CS + 4;       // LDS pixel stride in floats: 144 B -- the 16 lanes of a b128 access hit 64 distinct banks

// S: stride; NTH: threads; ACT: T3D_ACT_RELU6 (clamp form) or T3D_ACT_RELU; GMAX: 16-pixel groups per wave and item
template <typename T, int S, int NTH, int ACT, int GMAX>
__global__ __launch_bounds__(NTH) void expdw_fwd_kernel(const EdArgs a) {
  using V8 = typename St<T>::V8;
  using V4 = typename St<T>::V4;
  constexpr int NW = NTH / 64, NSLOT = NTH / 8;
  const T* __restrict__ az = reinterpret_cast<const T*>(a.z);
  const T* __restrict__ aw1 = reinterpret_cast<const T*>(a.w1);
  T* __restrict__ ay1 = reinterpret_cast<T*>(a.y1);
  T* __restrict__ ay2 = reinterpret_cast<T*>(a.y2);
  constexpr bool C6 = ACT == T3D_ACT_RELU6;
  extern __shared__ __attribute__((aligned(16))) float act[];        // [IR][W + 2][PS], then [2][CS] doubles
  const int Wp = a.W + 2;
  const int IRmax = (a.TH - 1) * S + 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lg = lane >> 4;

  // ---- which slab, which tiles: workgroup id -> (xcd, j); the nslab workgroups (xcd, j .. j + nslab - 1) walk the same tiles
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int slab = jj % a.nslab;
  const int lane_id = (jj / a.nslab) * 8 + xcd;                       // this workgroup's position among the tile walkers
  const int nwalk = (gridDim.x / a.nslab);                            // tile walkers (grid is a multiple of 8 * nslab)
  const int ntiles = a.B * a.tiles_per_img;
  const int c0 = slab * CS;

  // ---- per-workgroup constants
  // expansion weights: MFMA row m of tile ct is channel c0 + 8*(m/4) + 4*ct + m%4, so that accumulator register r of lane
  // (lg, lc) in tile ct is channel c0 + 8*lg + 4*ct + r of pixel lc: 8 consecutive channels per lane
  V8 wf[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int crow = c0 + 8 * (lc >> 2) + 4 * ct + (lc & 3);
    const int k = 8 * lg;
    V8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    if (k < a.K && crow < a.C) v = *reinterpret_cast<const V8*>(aw1 + (size_t)crow * a.K + k);
    wf[ct] = v;
  }
  f32x2 sc2[4], sh2[4];                       // BatchNorm affine of the lane's 8 channels (x 1/6 in the clamp form)
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int c = c0 + 8 * lg + 2 * h;
    const float m = C6 ? T3D_SIXTH : 1.f;
    sc2[h] = c < a.C ? f32x2{a.sc1[c] * m, a.sc1[c + 1] * m} : f32x2{0.f, 0.f};
    sh2[h] = c < a.C ? f32x2{a.sh1[c] * m, a.sh1[c + 1] * m} : f32x2{0.f, 0.f};
  }
  const int cg = tid & 7, slot = tid >> 3;
  const int cc = c0 + 4 * cg;                 // stencil phase: this thread's 4 channels
  f32x2 wk[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float m = C6 ? 6.f : 1.f;
      wk[t][h] = cc < a.C ? f32x2{a.wdw[(size_t)(cc + 2 * h) * 9 + t] * m, a.wdw[(size_t)(cc + 2 * h + 1) * 9 + t] * m} : f32x2{0.f, 0.f};
    }
  f32x2 ps[2] = {{0.f, 0.f}, {0.f, 0.f}}, pq[2] = {{0.f, 0.f}, {0.f, 0.f}};

  // zero frame (columns 0 and W + 1 of every tile row): written once, the expansion never touches it
  for (int i = tid; i < IRmax * 2 * (PS / 4); i += NTH) {
    const int r = i / (2 * (PS / 4)), rem = i - r * (2 * (PS / 4));
    const int cx = rem < PS / 4 ? 0 : Wp - 1, q = rem < PS / 4 ? rem : rem - PS / 4;
    *reinterpret_cast<f32x4*>(act + ((size_t)r * Wp + cx) * PS + 4 * q) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // pixel p = (wave + u*NW)*16 + lc of a tile sits at (row pr[u], column px[u]) whatever the item
  int pr[GMAX], loff[GMAX], pxk[GMAX];
#pragma unroll
  for (int u = 0; u < GMAX; ++u) {
    const int p = (wave + u * NW) * 16 + lc;
    pr[u] = p / a.W;
    const int x = p - pr[u] * a.W;
    loff[u] = (pr[u] * Wp + x + 1) * PS + 8 * lg;
    pxk[u] = x * a.K + min(8 * lg, a.K - 8);                  // (k-steps past K read a valid address; their weights are zero)
  }
  V8 zfr[GMAX];
  auto tile_of = [&](int t, int& b, int& oy0, int& oy1) {
    b = t / a.tiles_per_img;
    const int tr = t - b * a.tiles_per_img;
    oy0 = tr * a.TH;
    oy1 = min(a.Ho, oy0 + a.TH);
  };
  auto zfetch = [&](int t) {                                   // issue only: the fragments land while other work runs
    int b, oy0, oy1;
    tile_of(t, b, oy0, oy1);
    const int iy0 = oy0 * S - 1, nrows = (oy1 - oy0 - 1) * S + 3;
    const T* zb = az + (size_t)b * a.H * a.W * a.K;
#pragma unroll
    for (int u = 0; u < GMAX; ++u) {
      const int iy = min(max(iy0 + min(pr[u], nrows - 1), 0), a.H - 1);
      zfr[u] = *reinterpret_cast<const V8*>(zb + (size_t)iy * a.W * a.K + pxk[u]);
    }
  };

  int t = lane_id;
  if (t < ntiles) zfetch(t);
  __syncthreads();
  for (; t < ntiles; t += nwalk) {
    int b, oy0, oy1;
    tile_of(t, b, oy0, oy1);
    const int iy0 = oy0 * S - 1;
    const int nout = oy1 - oy0;
    const int nrows = (nout - 1) * S + 3;
    const int own0 = oy0 * S, own1 = min(a.H, oy1 * S);      // input rows whose raw expansion this item stores
    const int npx = nrows * a.W;
    const bool edge = iy0 < 0 || iy0 + nrows > a.H;            // tile touches the top / bottom of the image (wave-uniform)
    T* y1b = ay1 ? ay1 + ((ptrdiff_t)b * a.H + iy0) * a.W * a.C + c0 + 8 * lg : nullptr;   // (iy0 may be -1)

    // ---- phase 1: expansion -> bf16 rounding (+ y1 store) -> BatchNorm + activation -> LDS tile (fp32)
#pragma unroll
    for (int u = 0; u < GMAX; ++u) {
      if ((wave + u * NW) * 16 < npx) {                        // wave-uniform
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a0 = St<T>::mfma(wf[0], zfr[u], zero4);
        const f32x4 a1 = St<T>::mfma(wf[1], zfr[u], zero4);
        V8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = (T)a0[i]; o[4 + i] = (T)a1[i]; }
        const int r = pr[u];
        const int iy = iy0 + r;
        const bool pv = r < nrows;
        if (ay1 && pv && iy >= own0 && iy < own1 && c0 + 8 * lg < a.C)
          *reinterpret_cast<V8*>(y1b + (size_t)((wave + u * NW) * 16 + lc) * a.C) = o;
        f32x2 v[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const f32x2 y = {(float)o[2 * h], (float)o[2 * h + 1]};
          if constexpr (C6) {
            v[h] = pk_fma_clamp01(y, sc2[h], sh2[h]);
          } else {
            const f32x2 tt = pk_fma(y, sc2[h], sh2[h]);
            v[h] = f32x2{fmaxf(tt[0], 0.f), fmaxf(tt[1], 0.f)};
          }
        }
        if (edge && (iy < 0 || iy >= a.H)) {                   // the depthwise conv pads the ACTIVATED tensor with zeros
#pragma unroll
          for (int h = 0; h < 4; ++h) v[h] = f32x2{0.f, 0.f};
        }
        if (pv) {
          *reinterpret_cast<f32x4*>(act + loff[u]) = f32x4{v[0][0], v[0][1], v[1][0], v[1][1]};
          *reinterpret_cast<f32x4*>(act + loff[u] + 4) = f32x4{v[2][0], v[2][1], v[3][0], v[3][1]};
        }
      }
    }
    __syncthreads();
    // the next item's z fragments: issued now, consumed after the stencil phase
    if (t + nwalk < ntiles) zfetch(t + nwalk);

    // ---- phase 2: depthwise 3x3 out of LDS, walking down the input rows
    if (cc < a.C) {
      T* y2b = ay2 + ((size_t)b * a.Ho + oy0) * a.Wo * a.C + cc;
      for (int ox = slot; ox < a.Wo; ox += NSLOT) {
        const float* col = act + (size_t)(ox * S) * PS + 4 * cg;
        T* yo = y2b + (size_t)ox * a.C;
        f32x2 acc[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i][0] = acc[i][1] = f32x2{0.f, 0.f};
        auto emit = [&](f32x2* ac, int trow) {
          if (trow >= 0 && trow < nout) {
            V4 ov;
            ov[0] = (T)ac[0][0]; ov[1] = (T)ac[0][1]; ov[2] = (T)ac[1][0]; ov[3] = (T)ac[1][1];
            const f32x2 r0 = {(float)ov[0], (float)ov[1]}, r1 = {(float)ov[2], (float)ov[3]};
            ps[0] += r0; ps[1] += r1;
            pq[0] = pk_fma(r0, r0, pq[0]); pq[1] = pk_fma(r1, r1, pq[1]);
            *reinterpret_cast<V4*>(yo + (size_t)trow * a.Wo * a.C) = ov;
          }
          ac[0] = ac[1] = f32x2{0.f, 0.f};
        };
        auto taps = [&](int r, f32x2 (*v)[2]) {
          const float* rp = col + (size_t)r * Wp * PS;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(rp + kx * PS);
            v[kx][0] = f32x2{q[0], q[1]};
            v[kx][1] = f32x2{q[2], q[3]};
          }
        };
        auto mac = [&](f32x2* ac, f32x2 (*v)[2], int ky) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            ac[0] = pk_fma(v[kx][0], wk[ky * 3 + kx][0], ac[0]);
            ac[1] = pk_fma(v[kx][1], wk[ky * 3 + kx][1], ac[1]);
          }
        };
        if constexpr (S == 1) {
          // input row r feeds output rows r (ky 0), r-1 (ky 1), r-2 (ky 2); output row r-2 is complete after row r
          // (reading the taps of three rows ahead of the first multiply-add -- nine LDS reads in flight -- measured no faster)
          for (int r3 = 0; r3 < nrows; r3 += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int r = r3 + u;
              if (r < nrows) {
                f32x2 v[3][2];
                taps(r, v);
                mac(acc[u], v, 0);
                mac(acc[(u + 2) % 3], v, 1);
                mac(acc[(u + 1) % 3], v, 2);
                emit(acc[(u + 1) % 3], r - 2);
              }
            }
          }
        } else {
          // input row r = 2t + ky: even rows close output row t-1 (ky 2) and open row t (ky 0), odd rows are ky 1 of row t
          for (int r4 = 0; r4 < nrows; r4 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int r = r4 + u;
              if (r < nrows) {
                f32x2 v[3][2];
                taps(r, v);
                if (u == 0) { mac(acc[0], v, 0); mac(acc[1], v, 2); emit(acc[1], r / 2 - 1); }
                if (u == 1) mac(acc[0], v, 1);
                if (u == 2) { mac(acc[1], v, 0); mac(acc[0], v, 2); emit(acc[0], r / 2 - 1); }
                if (u == 3) mac(acc[1], v, 1);
              }
            }
          }
        }
      }
    }
    __syncthreads();       // the tile is rewritten by the next item
  }

  // ---- the next BatchNorm's sums of this workgroup's slab: snapped partials meet exactly in fp64 LDS, leave as fp64 atomics
  if (a.stats) {
    double* dstat = reinterpret_cast<double*>(act);          // [2][CS]
    for (int i = tid; i < 2 * CS; i += NTH) dstat[i] = 0.0;
    __syncthreads();
    if (cc < a.C) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        atomicAdd(dstat + 4 * cg + i, t3d_snap(ps[i / 2][i % 2], a.quant, false));
        atomicAdd(dstat + CS + 4 * cg + i, t3d_snap(pq[i / 2][i % 2], a.quant, true));
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * CS; i += NTH) {
      const int c = c0 + i % CS;
      if (c < a.C && dstat[i] != 0.0)
        atomicAdd(a.stats + (size_t)(blockIdx.x % a.nrep) * a.rstride + (size_t)(i / CS) * a.C + c, dstat[i]);
    }
  }
}

// threads per workgroup and 16-pixel fragment groups per wave for an input width (launch_s picks the instantiation from these)
static int expdw_nth(int W) {
  static const int nth_env = getenv("T3D_EXPDW_NTH") ? atoi(getenv("T3D_EXPDW_NTH")) : 0;
  return nth_env ? nth_env : (W > 60 ? 512 : 256);
}
// input rows a workgroup can hold: the LDS budget and the per-wave fragment registers (gmax groups of 16 pixels) bound it;
// fewer than 3 (one output row's stencil) means the shape is not served -- t3d_expdw_supported says so ahead of the launch
static int expdw_rows(int W, int nth) {
  static const int lds_kb_env = getenv("T3D_EXPDW_LDS_KB") ? atoi(getenv("T3D_EXPDW_LDS_KB")) : 0;
  const int lds_kb = lds_kb_env ? lds_kb_env : (nth == 512 ? 150 : 76);
  const int gmax = nth == 512 ? 5 : 7;
  const size_t row_bytes = (size_t)(W + 2) * PS * 4;
  const int ir = (int)(((size_t)lds_kb << 10) / row_bytes), ir_frag = (gmax * (nth / 64) * 16) / W;
  return ir < ir_frag ? ir : ir_frag;
}

template <typename T, int S, int NTH, int ACT, int GMAX>
int launch_g(EdArgs& a, hipStream_t st) {
  static_assert(GMAX == (NTH == 512 ? 5 : 7), "expdw_rows assumes this pairing");
  const size_t row_bytes = (size_t)(a.W + 2) * PS * 4;
  const int ir = expdw_rows(a.W, NTH);
  if (ir < 3) return T3D_ERR_UNSUPPORTED;
  int th = (ir - 3) / S + 1;
  if (th > a.Ho) th = a.Ho;
  a.tiles_per_img = cdiv(a.Ho, th);
  a.TH = cdiv(a.Ho, a.tiles_per_img);        // even tiles
  a.tiles_per_img = cdiv(a.Ho, a.TH);
  a.nslab = cdiv(a.C, CS);
  a.nitems = a.B * a.tiles_per_img * a.nslab;
  const int irmax = (a.TH - 1) * S + 3;
  size_t lds = (size_t)irmax * row_bytes;
  if (lds < 2 * CS * sizeof(double)) lds = 2 * CS * sizeof(double);
  if (lds > 158 * 1024) return T3D_ERR_UNSUPPORTED;
  const void* fn = (const void*)expdw_fwd_kernel<T, S, NTH, ACT, GMAX>;
  if (lds > 64 * 1024 && t3d_max_lds(fn, (int)lds) != hipSuccess) return T3D_ERR_UNSUPPORTED;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NTH, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  // grid: what is resident at once, a multiple of 8 * nslab (the slab workgroups of a tile walker share an XCD)
  const int unit = 8 * a.nslab;
  int grid = 256 * per_cu / unit * unit;
  const int need = cdiv(a.B * a.tiles_per_img, 8) * unit;
  if (grid > need) grid = need;
  if (grid < unit) grid = unit;
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  if (a.stats && a.nrep < 1) { a.nrep = 1; a.rstride = 0; }
  a.quant = (a.stats && std::is_same<T, bf16_t>::value && !T3D_ENV_SET("T3D_NO_SNAP")) ? t3d_quant_for((long long)a.B * a.Ho * a.Wo) : T3dQuant{0.0, 0.0};
  T3D_LAUNCH_TIMED((expdw_fwd_kernel<T, S, NTH, ACT, GMAX>), dim3(grid), dim3(NTH), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <typename T, int S, int ACT>
int launch_s(EdArgs& a, hipStream_t st) {
  const int nth = expdw_nth(a.W);
  if (nth == 512) return launch_g<T, S, 512, ACT, 5>(a, st);
  return launch_g<T, S, 256, ACT, 7>(a, st);
}

template <typename T>
int launch_t(EdArgs& a, int act, int stride, hipStream_t st) {
  if (stride == 1) return act == T3D_ACT_RELU6 ? launch_s<T, 1, T3D_ACT_RELU6>(a, st) : launch_s<T, 1, T3D_ACT_RELU>(a, st);
  return act == T3D_ACT_RELU6 ? launch_s<T, 2, T3D_ACT_RELU6>(a, st) : launch_s<T, 2, T3D_ACT_RELU>(a, st);
}

}  // namespace

// include/t3d.h: would t3d_expdw_fwd take this shape?  (the same tests the launch makes, without launching)
extern "C" int t3d_expdw_supported(int dtype, int act, int B, int H, int W, int K, int C, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || C <= 0) return 0;
  if (dtype != T3D_BF16 && dtype != T3D_F16) return 0;
  if ((K % 8) || (C % 8) || K > 32 || (stride != 1 && stride != 2) || W < 8) return 0;
  if (act != T3D_ACT_RELU6 && act != T3D_ACT_RELU) return 0;
  if ((size_t)B * H * W * C * 2 >= (1ull << 32)) return 0;
  return expdw_rows(W, expdw_nth(W)) >= 3;
}

// include/t3d.h
extern "C" int t3d_expdw_fwd(int dtype, const void* z, const void* w1, const float* scale1, const float* shift1, int act,
                             const float* wdw, void* y1, void* y2, double* stats2, int B, int H, int W, int K, int C,
                             int stride, void* stream) {
  if (!z || !w1 || !scale1 || !shift1 || !wdw || !y2 || B <= 0 || H <= 0 || W <= 0 || K <= 0 || C <= 0) return T3D_ERR_ARG;
  if (dtype != T3D_BF16 && dtype != T3D_F16) return T3D_ERR_UNSUPPORTED;
  if ((K % 8) || (C % 8) || K > 32 || (stride != 1 && stride != 2) || W < 8) return T3D_ERR_UNSUPPORTED;
  if (act != T3D_ACT_RELU6 && act != T3D_ACT_RELU) return T3D_ERR_UNSUPPORTED;
  if ((size_t)B * H * W * C * 2 >= (1ull << 32)) return T3D_ERR_UNSUPPORTED;
  EdArgs a{};
  a.z = z; a.w1 = w1;
  a.sc1 = scale1; a.sh1 = shift1; a.wdw = wdw;
  a.y1 = y1; a.y2 = y2; a.stats = stats2;
  a.B = B; a.H = H; a.W = W; a.K = K; a.C = C;
  a.Ho = (H + 2 - 3) / stride + 1;
  a.Wo = (W + 2 - 3) / stride + 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return dtype == T3D_F16 ? launch_t<f16_t>(a, act, stride, st) : launch_t<bf16_t>(a, act, stride, st);
}
