// Fused expand (1x1) + BatchNorm + activation + depthwise 3x3 forward of an inverted-residual block, TRAINING mode,
// bf16 storage, gfx950:
//
//   y2[b, oy, ox, c] = sum_{ky,kx} w_dw[c][ky][kx] * a[b, S*oy-1+ky, S*ox-1+kx, c],   a = act(scale1[c] * (W1 z)[c] + shift1[c])
//
// (models/mobilenetv3.py:146-153: conv 1x1 -> BN -> act -> depthwise conv -> BN, the expanded tensor being 6x wider than
// the block input z.)  Layer by layer the expanded tensor costs a write and a read of HBM in the forward (the largest
// tensors of the network: 616 MB at 112x112x96 for a batch of 256); here it is produced and consumed in LDS:
//   phase 1  a work item = (image, tile of TH output rows, slab of CS expanded channels).  The waves recompute the
//            expansion of the tile's input rows (+ halo rows) on the matrix cores -- v_mfma_f32_16x16x32_bf16 with
//            A = W1 rows (channels), B = z^T (16 pixels): a lane ends up with 4 consecutive channels of one pixel --
//            apply BatchNorm + activation in registers and store the activated values (bf16) into an LDS tile
//            [row][column + zero padding][channel], optionally also the RAW expansion to HBM (for a backward that
//            does not recompute it);
//   phase 2  a thread owns (4 channels, one output column) and walks down the tile's output rows: 9 LDS reads of
//            8 bytes and 36 multiply-adds per output vector, stores y2 (raw, 8 B) and keeps the per-channel sums of the
//            following BatchNorm in registers; they meet in LDS per item and leave as fp64 atomics once per workgroup.
// The batch statistics of the expansion's own BatchNorm have to exist BEFORE this kernel (they are global over the batch):
// the caller runs the 1x1 conv once as a statistics-only pass (t3d_pwconv_fwd with y = NULL: reads the narrow z, stores
// nothing) and finalizes them.  HBM traffic: z (narrow, re-read per slab through L2) + y2 [+ y1 when stored], against
// z + 2 y1 + y2 layer by layer.
#include <cstdlib>
#include "common.h"

namespace {

struct EdArgs {
  const bf16_t *z, *w1;
  const float *sc1, *sh1;
  int act;
  const float* wdw;
  bf16_t *y1, *y2;
  double* stats;
  int B, H, W, K, C, S, Ho, Wo;
  int TH, tiles_per_img, nslab, nitems;
  int nrep;
  long long rstride;
};

constexpr int KSMAX = 5;     // K <= 160

#ifdef T3D_ED_TRACE
// debug build only (tools/time_expdw.py --trace): accumulated wall-clock ticks (10 ns) per phase of workgroup 0, wave 0
__device__ unsigned long long g_ed_trace[8];
#define ED_T0() unsigned long long t_prev = wall_clock64(), t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define ED_PH(i) do { const unsigned long long t_now = wall_clock64(); t_acc[i] += t_now - t_prev; t_prev = t_now; } while (0)
#define ED_OUT() do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) g_ed_trace[i_] = t_acc[i_]; } while (0)
#else
#define ED_T0()
#define ED_PH(i)
#define ED_OUT()
#endif

template <int CS, int NTH>
__global__ __launch_bounds__(NTH) void expdw_fwd_kernel(const EdArgs a) {
  constexpr int CT = CS / 16;          // 16-channel MFMA tiles per slab
  constexpr int PS = CS + 4;           // LDS pixel stride (elements): +8 B against bank conflicts of the phase-1 writes
  constexpr int CGS = CS / 4, NSLOT = NTH / CGS, NW = NTH / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Wp = a.W + 2;
  const int IRmax = (a.TH - 1) * a.S + 3;
  bf16_t* act = reinterpret_cast<bf16_t*>(smem);                                         // [IR][Wp][PS]
  float* lstat = reinterpret_cast<float*>(smem + (((size_t)IRmax * Wp * PS * 2 + 15) & ~(size_t)15));   // [2][C]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lp = lane & 15, lg = lane >> 4;
  const int KS = (a.K + 31) / 32;
  for (int i = tid; i < 2 * a.C; i += NTH) lstat[i] = 0.f;
  __syncthreads();

  ED_T0();
  for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
    const int slab = item % a.nslab, tile = item / a.nslab;
    const int b = tile / a.tiles_per_img, tr = tile - b * a.tiles_per_img;
    const int oy0 = tr * a.TH, oy1 = min(a.Ho, oy0 + a.TH);
    const int iy0 = oy0 * a.S - 1;
    const int nrows = (oy1 - oy0 - 1) * a.S + 3;
    const int own0 = oy0 * a.S, own1 = min(a.H, oy1 * a.S);     // input rows whose raw expansion this item stores
    const int c0 = slab * CS;

    // ---- phase 1: expansion on the matrix cores -> BatchNorm + activation -> LDS (zero padding explicit)
    for (int i = tid; i < nrows * 2 * (CS / 4); i += NTH) {      // the two padding columns of every row
      const int r = i / (2 * (CS / 4)), rem = i - r * (2 * (CS / 4));
      const int cx = rem < CS / 4 ? 0 : Wp - 1, q = rem < CS / 4 ? rem : rem - CS / 4;
      *reinterpret_cast<uint2*>(act + ((size_t)r * Wp + cx) * PS + 4 * q) = make_uint2(0u, 0u);
    }
    bf16x8 wf[CT][KSMAX];
    float sc[CT][4], sh[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int crow = c0 + ct * 16 + lp;
#pragma unroll
      for (int ks = 0; ks < KSMAX; ++ks) {
        const int k = ks * 32 + 8 * lg;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
        if (ks < KS && k < a.K && crow < a.C) v = *reinterpret_cast<const bf16x8*>(a.w1 + (size_t)crow * a.K + k);
        wf[ct][ks] = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + ct * 16 + 4 * lg + i;
        sc[ct][i] = c < a.C ? a.sc1[c] : 0.f;
        sh[ct][i] = c < a.C ? a.sh1[c] : 0.f;
      }
    }
    ED_PH(0);      // padding columns, weight / coefficient fragments
    const int npx = nrows * a.W, ngroups = (npx + 15) >> 4;
    for (int g = wave; g < ngroups; g += NW) {
      const int p = g * 16 + lp;
      const bool pv = p < npx;
      const int r = pv ? p / a.W : 0, x = pv ? p - r * a.W : 0;
      const int iy = iy0 + r;
      const bool rv = pv && iy >= 0 && iy < a.H;
      const bf16_t* zp = a.z + (((size_t)b * a.H + min(max(iy, 0), a.H - 1)) * a.W + x) * a.K;
      bf16x8 zf[KSMAX];
#pragma unroll
      for (int ks = 0; ks < KSMAX; ++ks) {
        const int k = ks * 32 + 8 * lg;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
        if (ks < KS && k < a.K) v = *reinterpret_cast<const bf16x8*>(zp + k);
        zf[ks] = v;
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks)
          if (ks < KS) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][ks], zf[ks], acc, 0, 0, 0);
        // lane: channels c0 + ct*16 + 4*lg .. +3 of pixel p
        const int cl = ct * 16 + 4 * lg;
        float u[4] = {acc[0], acc[1], acc[2], acc[3]};
        if (a.y1 && rv && iy >= own0 && iy < own1 && c0 + cl < a.C) {
          bf16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (bf16_t)u[i];
          *reinterpret_cast<bf16x4*>(a.y1 + (((size_t)b * a.H + iy) * a.W + x) * a.C + c0 + cl) = o;
#pragma unroll
          for (int i = 0; i < 4; ++i) u[i] = (float)o[i];       // the stored (rounded) value is what gets normalised
        }
        act_affine_vec<4>(u, sc[ct], sh[ct], a.act);
        bf16x4 av;
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = (bf16_t)(rv ? u[i] : 0.f);
        if (pv) *reinterpret_cast<bf16x4*>(act + ((size_t)r * Wp + x + 1) * PS + cl) = av;
      }
    }
    ED_PH(1);      // expansion loop
    __syncthreads();
    ED_PH(2);      // barrier

    // ---- phase 2: depthwise 3x3 out of LDS
    {
      const int cg = tid % CGS, slot = tid / CGS;
      const int cc = c0 + 4 * cg;
      if (cc < a.C) {
        float wk[9][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 9; ++t) wk[t][i] = a.wdw[(size_t)(cc + i) * 9 + t];
        ED_PH(3);  // depthwise weights
        float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ox = slot; ox < a.Wo; ox += NSLOT) {
          const bf16_t* col = act + (size_t)(ox * a.S) * PS + 4 * cg;
          for (int t = 0; t < oy1 - oy0; ++t) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const bf16_t* rowp = col + (size_t)(t * a.S + ky) * Wp * PS;
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) {
                const bf16x4 v = *reinterpret_cast<const bf16x4*>(rowp + kx * PS);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf((float)v[i], wk[ky * 3 + kx][i], o[i]);
              }
            }
            bf16x4 ov;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              ov[i] = (bf16_t)o[i];
              const float rr = (float)ov[i];
              ps[i] += rr;
              pq[i] = fmaf(rr, rr, pq[i]);
            }
            *reinterpret_cast<bf16x4*>(a.y2 + (((size_t)b * a.Ho + oy0 + t) * a.Wo + ox) * a.C + cc) = ov;
          }
        }
        ED_PH(4);  // stencil + stores
        if (a.stats) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            atomicAdd(lstat + cc + i, ps[i]);
            atomicAdd(lstat + a.C + cc + i, pq[i]);
          }
        }
      }
    }
    ED_PH(5);              // statistics into LDS
    __syncthreads();       // the tile is rewritten by the next item
    ED_PH(6);
  }
  ED_OUT();
  if (a.stats) {
    for (int i = tid; i < 2 * a.C; i += NTH) {
      const float v = lstat[i];
      if (v != 0.f) atomicAdd(a.stats + (size_t)(blockIdx.x % a.nrep) * a.rstride + i, (double)v);
    }
  }
}

template <int CS>
int launch(EdArgs& a, hipStream_t st) {
  constexpr int NTH = 256, PS = CS + 4;
  // tile height: the LDS tile of (TH-1)*S + 3 input rows within ~56 KB, so that 2-3 workgroups share a CU and hide each
  // other's phase changes
  static const int lds_kb = getenv("T3D_EXPDW_LDS_KB") ? atoi(getenv("T3D_EXPDW_LDS_KB")) : 56;
  const size_t row_bytes = (size_t)(a.W + 2) * PS * 2;
  int ir = (int)(((size_t)lds_kb << 10) / row_bytes);
  if (ir < 3) ir = 3;
  int th = (ir - 3) / a.S + 1;
  if (th > a.Ho) th = a.Ho;
  a.TH = th;
  a.tiles_per_img = cdiv(a.Ho, th);
  a.TH = cdiv(a.Ho, a.tiles_per_img);        // even tiles
  a.tiles_per_img = cdiv(a.Ho, a.TH);
  a.nslab = cdiv(a.C, CS);
  a.nitems = a.B * a.tiles_per_img * a.nslab;
  const int irmax = (a.TH - 1) * a.S + 3;
  const size_t lds = (((size_t)irmax * row_bytes + 15) & ~(size_t)15) + (size_t)2 * a.C * sizeof(float);
  if (lds > 150 * 1024) return T3D_ERR_UNSUPPORTED;
  const void* fn = (const void*)expdw_fwd_kernel<CS, NTH>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int per_cu = (int)((size_t)(160 * 1024) / (lds + 1024));
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  int grid = 256 * per_cu;
  if (grid > a.nitems) grid = a.nitems;
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  hipLaunchKernelGGL((expdw_fwd_kernel<CS, NTH>), dim3(grid), dim3(NTH), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

// include/t3d.h
extern "C" int t3d_expdw_fwd(const void* z, const void* w1, const float* scale1, const float* shift1, int act,
                             const float* wdw, void* y1, void* y2, double* stats2, int B, int H, int W, int K, int C,
                             int stride, void* stream) {
  if (!z || !w1 || !scale1 || !shift1 || !wdw || !y2 || B <= 0 || H <= 0 || W <= 0 || K <= 0 || C <= 0) return T3D_ERR_ARG;
  if ((K % 8) || (C % 8) || K > 32 * KSMAX || (stride != 1 && stride != 2)) return T3D_ERR_UNSUPPORTED;
  EdArgs a{};
  a.z = reinterpret_cast<const bf16_t*>(z); a.w1 = reinterpret_cast<const bf16_t*>(w1);
  a.sc1 = scale1; a.sh1 = shift1; a.act = act; a.wdw = wdw;
  a.y1 = reinterpret_cast<bf16_t*>(y1); a.y2 = reinterpret_cast<bf16_t*>(y2); a.stats = stats2;
  a.B = B; a.H = H; a.W = W; a.K = K; a.C = C; a.S = stride;
  a.Ho = (H + 2 - 3) / stride + 1;
  a.Wo = (W + 2 - 3) / stride + 1;
  static const int cs_env = getenv("T3D_EXPDW_CS") ? atoi(getenv("T3D_EXPDW_CS")) : 0;
  const int cs = cs_env ? cs_env : ((W > 60 || C % 64) ? 32 : 64);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return cs == 64 ? launch<64>(a, st) : launch<32>(a, st);
}

#ifdef T3D_ED_TRACE
extern "C" int t3d_debug_ed_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ed_trace), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
