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

// KSN: 32-wide contraction steps (compile time: fragment registers follow K)
// ACT: the activation as a compile-time constant (a switch inside the per-tile epilogue is a scalar branch chain per tile)
template <int CS, int NTH, int KSN, int ACT>
__global__ __launch_bounds__(NTH) void expdw_fwd_kernel(const EdArgs a) {
  constexpr int CT = CS / 16;          // 16-channel MFMA tiles per slab
  constexpr int PS = CS + 4;           // LDS pixel stride of the activated tile (elements): +8 B against bank conflicts
  constexpr int CGS = CS / 4, NSLOT = NTH / CGS, NW = NTH / 64;
  // z fragments (the MFMA B operand: 16 B per lane) of the NEXT item's pixel groups, per wave, fetched while the current
  // item's depthwise phase runs: GMAX groups per wave x KSN steps
  constexpr int GMAX = KSN == 1 ? 12 : (KSN == 2 ? 6 : (KSN == 3 ? 4 : 3));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Wp = a.W + 2;
  const int IRmax = (a.TH - 1) * a.S + 3;
  bf16_t* act = reinterpret_cast<bf16_t*>(smem);                                         // [IR][Wp][PS]
  const size_t act_bytes = ((size_t)IRmax * Wp * PS * 2 + 15) & ~(size_t)15;
  float* lstat = reinterpret_cast<float*>(smem + act_bytes);                             // [2][C]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lp = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 2 * a.C; i += NTH) lstat[i] = 0.f;

  // ---- per-workgroup constants: the launcher makes the grid a multiple of the slab count, so a workgroup keeps ONE slab
  const int slab = blockIdx.x % a.nslab;
  const int c0 = slab * CS;
  bf16x8 wf[CT][KSN];
  float sc[CT][4], sh[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int crow = c0 + ct * 16 + lp;
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      const int k = ks * 32 + 8 * lg;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
      if (k < a.K && crow < a.C) v = *reinterpret_cast<const bf16x8*>(a.w1 + (size_t)crow * a.K + k);
      wf[ct][ks] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ct * 16 + 4 * lg + i;
      sc[ct][i] = c < a.C ? a.sc1[c] : 0.f;
      sh[ct][i] = c < a.C ? a.sh1[c] : 0.f;
    }
  }
  const int cg = tid % CGS, slot = tid / CGS;
  const int cc = c0 + 4 * cg;
  f32x2 wk2[9][2];           // depthwise weights of this thread's 4 channels, packed for v_pk_fma_f32
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      wk2[t][h] = cc < a.C ? f32x2{a.wdw[(size_t)(cc + 2 * h) * 9 + t], a.wdw[(size_t)(cc + 2 * h + 1) * 9 + t]} : f32x2{0.f, 0.f};

  // z tile of an item: rows iy0 .. iy0+nrows-1 (clamped into the image: out-of-image rows are zeroed in the expansion)
  auto tile_of = [&](int item, int& b, int& oy0, int& oy1) {
    const int tile = item / a.nslab;
    b = tile / a.tiles_per_img;
    const int tr = tile - b * a.tiles_per_img;
    oy0 = tr * a.TH;
    oy1 = min(a.Ho, oy0 + a.TH);
  };
  // pixel p = (wave + u*NW)*16 + lp of a tile sits at (row pr[u], column px[u]) whatever the item: the divisions happen
  // once per workgroup, not once per group and item (a wave64 VALU instruction is 4 cycles: the first version's ~70
  // instructions per 16 x 16 output tile made the expansion phase the whole kernel)
  int pr[GMAX], pxx[GMAX];
#pragma unroll
  for (int u = 0; u < GMAX; ++u) {
    const int p = (wave + u * NW) * 16 + lp;
    pr[u] = p / a.W;
    pxx[u] = p - pr[u] * a.W;
  }
  bf16x8 zfr[GMAX][KSN];
  auto zfetch = [&](int item) {        // issue only: the fragments land while other work runs
    int b, oy0, oy1;
    tile_of(item, b, oy0, oy1);
    const int iy0 = oy0 * a.S - 1, nrows = (oy1 - oy0 - 1) * a.S + 3;
#pragma unroll
    for (int u = 0; u < GMAX; ++u) {
      const int r = min(pr[u], nrows - 1);
      const int iy = min(max(iy0 + r, 0), a.H - 1);
      const bf16_t* zp = a.z + (((size_t)b * a.H + iy) * a.W + pxx[u]) * a.K;
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks) {
        const int k = min(ks * 32 + 8 * lg, a.K - 8);        // (steps past K read a valid address; their weights are zero)
        zfr[u][ks] = *reinterpret_cast<const bf16x8*>(zp + k);
      }
    }
  };

  int item = blockIdx.x;
  ED_T0();
  if (item < a.nitems) zfetch(item);
  __syncthreads();       // (lstat zeroed)
  for (; item < a.nitems; item += gridDim.x) {
    int b, oy0, oy1;
    tile_of(item, b, oy0, oy1);
    const int iy0 = oy0 * a.S - 1;
    const int nrows = (oy1 - oy0 - 1) * a.S + 3;
    const int own0 = oy0 * a.S, own1 = min(a.H, oy1 * a.S);     // input rows whose raw expansion this item stores
    const int npx = nrows * a.W;

    // ---- phase 1: padding columns; expansion on the matrix cores (z fragments fetched during the previous item's
    // depthwise phase) -> BatchNorm + activation -> activated tile
    for (int i = tid; i < nrows * 2 * (CS / 4); i += NTH) {
      const int r = i / (2 * (CS / 4)), rem = i - r * (2 * (CS / 4));
      const int cx = rem < CS / 4 ? 0 : Wp - 1, q = rem < CS / 4 ? rem : rem - CS / 4;
      *reinterpret_cast<uint2*>(act + ((size_t)r * Wp + cx) * PS + 4 * q) = make_uint2(0u, 0u);
    }
    ED_PH(0);      // padding columns (+ wait for the fragments fetched earlier)
#pragma unroll
    for (int u = 0; u < GMAX; ++u) {
      if ((wave + u * NW) * 16 < npx) {          // wave-uniform
      const int r = pr[u], x = pxx[u];
      const bool pv = r < nrows;
      const int iy = iy0 + r;
      const bool rv = pv && iy >= 0 && iy < a.H;
      bf16_t* dst = act + ((size_t)r * Wp + x + 1) * PS + 4 * lg;
      const bool st1 = a.y1 != nullptr && rv && iy >= own0 && iy < own1;
      bf16_t* y1p = a.y1 + (((size_t)b * a.H + iy) * a.W + x) * a.C + c0 + 4 * lg;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][ks], zfr[u][ks], acc, 0, 0, 0);
        // lane: channels c0 + ct*16 + 4*lg .. +3 of pixel p
        float uv[4] = {acc[0], acc[1], acc[2], acc[3]};
        if (st1 && c0 + ct * 16 + 4 * lg < a.C) {
          bf16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (bf16_t)uv[i];
          *reinterpret_cast<bf16x4*>(y1p + ct * 16) = o;
#pragma unroll
          for (int i = 0; i < 4; ++i) uv[i] = (float)o[i];      // the stored (rounded) value is what gets normalised
        }
        bf16x4 av;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float t = fmaf(uv[i], sc[ct][i], sh[ct][i]);
          t = ACT == T3D_ACT_RELU6 ? __builtin_amdgcn_fmed3f(t, 0.f, 6.f) : fmaxf(t, 0.f);
          av[i] = (bf16_t)(rv ? t : 0.f);
        }
        if (pv) *reinterpret_cast<bf16x4*>(dst + ct * 16) = av;
      }
      }
    }
    ED_PH(1);      // expansion
    __syncthreads();
    ED_PH(2);      // barrier
    // the next item's z fragments: issued now, consumed after the depthwise phase
    if (item + gridDim.x < a.nitems) zfetch(item + gridDim.x);

    // ---- phase 2: depthwise 3x3 out of LDS
    if (cc < a.C) {
      float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};
      for (int ox = slot; ox < a.Wo; ox += NSLOT) {
        const bf16_t* col = act + (size_t)(ox * a.S) * PS + 4 * cg;
        for (int t = 0; t < oy1 - oy0; ++t) {
          f32x2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const bf16_t* rowp = col + (size_t)(t * a.S + ky) * Wp * PS;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              // 4 bf16 = two dwords; bf16 -> fp32 is a shift / a mask
              const uint2 raw = *reinterpret_cast<const uint2*>(rowp + kx * PS);
              const f32x2 v01 = {__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u)};
              const f32x2 v23 = {__uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u)};
              o01 = pk_fma(v01, wk2[ky * 3 + kx][0], o01);
              o23 = pk_fma(v23, wk2[ky * 3 + kx][1], o23);
            }
          }
          const float o[4] = {o01[0], o01[1], o23[0], o23[1]};
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
      ED_PH(4);    // stencil + stores
      if (a.stats) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          atomicAdd(lstat + cc + i, ps[i]);
          atomicAdd(lstat + a.C + cc + i, pq[i]);
        }
      }
    }
    ED_PH(5);
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

template <int CS, int KSN, int ACT>
int launch_a(EdArgs& a, hipStream_t st) {
  constexpr int NTH = 256, PS = CS + 4, NW = NTH / 64;
  constexpr int GMAX = KSN == 1 ? 12 : (KSN == 2 ? 6 : (KSN == 3 ? 4 : 3));
  // tile height: the activated tile of (TH-1)*S + 3 input rows within ~56 KB (two or three workgroups share a CU and
  // overlap each other's phases), and at most GMAX pixel groups per wave (their z fragments wait in registers)
  static const int lds_kb = getenv("T3D_EXPDW_LDS_KB") ? atoi(getenv("T3D_EXPDW_LDS_KB")) : 56;
  const size_t row_bytes = (size_t)(a.W + 2) * PS * 2;
  int ir = (int)(((size_t)lds_kb << 10) / row_bytes);
  const int ir_frag = (GMAX * NW * 16) / a.W;
  if (ir > ir_frag) ir = ir_frag;
  if (ir < 3) return T3D_ERR_UNSUPPORTED;
  int th = (ir - 3) / a.S + 1;
  if (th > a.Ho) th = a.Ho;
  a.tiles_per_img = cdiv(a.Ho, th);
  a.TH = cdiv(a.Ho, a.tiles_per_img);        // even tiles
  a.tiles_per_img = cdiv(a.Ho, a.TH);
  a.nslab = cdiv(a.C, CS);
  a.nitems = a.B * a.tiles_per_img * a.nslab;
  const int irmax = (a.TH - 1) * a.S + 3;
  const size_t lds = (((size_t)irmax * row_bytes + 15) & ~(size_t)15) + (size_t)2 * a.C * sizeof(float);
  if (lds > 150 * 1024) return T3D_ERR_UNSUPPORTED;
  const void* fn = (const void*)expdw_fwd_kernel<CS, NTH, KSN, ACT>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // persistent grid = what is RESIDENT at once (registers allow two workgroups per CU even where LDS would take three: a
  // third one per CU ran as a second round after the first -- 129 us of work per workgroup, 442 us for the launch)
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NTH, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  int grid = 256 * per_cu;
  if (grid > a.nitems) grid = a.nitems;
  if (grid >= a.nslab) grid -= grid % a.nslab;     // a workgroup keeps ONE slab (items stride by the grid)
  else return T3D_ERR_UNSUPPORTED;
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  hipLaunchKernelGGL((expdw_fwd_kernel<CS, NTH, KSN, ACT>), dim3(grid), dim3(NTH), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <int CS, int KSN>
int launch(EdArgs& a, hipStream_t st) {
  if (a.act == T3D_ACT_RELU6) return launch_a<CS, KSN, T3D_ACT_RELU6>(a, st);
  if (a.act == T3D_ACT_RELU) return launch_a<CS, KSN, T3D_ACT_RELU>(a, st);
  return T3D_ERR_UNSUPPORTED;
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
  const int ks = (K + 31) / 32;
  if (cs == 64) return ks == 1 ? launch<64, 1>(a, st) : ks == 2 ? launch<64, 2>(a, st) : ks == 3 ? launch<64, 3>(a, st) : launch<64, 5>(a, st);
  return ks == 1 ? launch<32, 1>(a, st) : ks == 2 ? launch<32, 2>(a, st) : ks == 3 ? launch<32, 3>(a, st) : launch<32, 5>(a, st);
}

#ifdef T3D_ED_TRACE
extern "C" int t3d_debug_ed_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ed_trace), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
