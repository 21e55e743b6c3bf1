// Inverted-residual block, inference mode, in ONE kernel for the 14x14 and 7x7 stages (stride 1, 3x3, no squeeze-excite):
//
//   z = BN3(W2 * act(BN2(dw3x3(act(BN1(W1 * x)))))) [+ x]          (torchdet3d/models/mobilenetv3.py:146-164, eval)
//
// With running statistics every BatchNorm is a per-channel affine, so nothing couples the samples and the block can run
// per image: one workgroup (4 waves) owns one image, keeps its input plane in LDS and walks the expanded channels in slabs
// of 64 -- expand (MFMA) -> BatchNorm + activation -> depthwise 3x3 (from LDS) -> BatchNorm + activation -> project
// (MFMA, accumulated over the slabs in registers) -- so the two expanded tensors (6x the block's input) never leave the
// CU.  HBM traffic of the block: read x, write z (+ the weights, streamed from L2 by every workgroup); the three launches
// + bn_apply it replaces read / write 2 * (1 + 6 + 6 + 1) / (1 + 1) = 14x that.
//
// Rounding points are those of the unfused path (so the two agree to the last bf16 bit except where the depthwise sums
// differ in their last fp32 bit): expand output -> bf16, BN1 + act in fp32 (kept fp32 in LDS), depthwise sum -> bf16,
// BN2 + act -> bf16 (MFMA operand), project output -> bf16, BN3 (+ residual) -> bf16.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

#ifdef T3D_BLK_TRACE
// debug build only: time (10 ns units) block 0 spends in each phase: 0 plane load, 1 weights, 2 expand, 3 depthwise, 4 project, 5 epilogue
__device__ unsigned long long g_blk_trace[8];
#define BLK_T0() unsigned long long t_prev = wall_clock64(), t_acc[6] = {0, 0, 0, 0, 0, 0}
#define BLK_PH(i) do { const unsigned long long t_now = wall_clock64(); t_acc[i] += t_now - t_prev; t_prev = t_now; } while (0)
#define BLK_END() do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i = 0; i < 6; ++i) g_blk_trace[i] = t_acc[i]; } } while (0)
#else
#define BLK_T0()
#define BLK_PH(i)
#define BLK_END()
#endif

struct BlkArgs {
  const bf16_t* x;      // [B][P][Cin] finished input
  const bf16_t* w1;     // [Ce][Cin]
  const bf16_t* w2;     // [Cout][Ce]
  const float* wd;      // [Ce][9]
  const float *s1, *h1, *s2, *h2, *s3, *h3;
  bf16_t* z;            // [B][P][Cout]
  int act1, act2, residual;
  int H, W, Cin, Ce, Cout;
};

constexpr int SL = 64;          // expanded channels per slab
constexpr int SLP = SL + 8;     // bf16 row stride of the slab operand (+16 B against bank conflicts)
constexpr int NTHR = 512, NWAVE = NTHR / 64;
constexpr int SLF = SL + 4;     // fp32 row stride of the activated expand output (16 lanes x 16 B stores: 2-way instead of 16-way conflicts)

// MTW: 16-pixel MFMA tiles per wave (8 waves); NT2: 16-channel output tiles (Cout / 16)
template <int MTW, int NT2>
__global__ __launch_bounds__(512) void ir_block_eval_kernel(const BlkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int P = a.H * a.W, MT = (P + 15) / 16, CinP = a.Cin + 8, KS1 = a.Cin / 32;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                        // [MT*16][CinP]
  float* a1 = reinterpret_cast<float*>(xs + (size_t)MT * 16 * CinP);   // [P][SLF]  activated expand output (fp32)
  bf16_t* dsb = reinterpret_cast<bf16_t*>(a1 + (size_t)P * SLF);       // [MT*16][SLP]
  bf16_t* w1s = dsb + (size_t)MT * 16 * SLP;                           // [SL][CinP]
  bf16_t* w2s = w1s + (size_t)SL * CinP;                               // [Cout][SLP]
  float* wds = reinterpret_cast<float*>(w2s + (size_t)a.Cout * SLP);   // [9][SL] tap-major
  float* afs = wds + SL * 9;                                           // [4][SL]: s1 | h1 | s2 | h2 of the slab

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lg = lane >> 4;
  const size_t img = blockIdx.x;
  const bf16_t* __restrict__ xg = a.x + img * P * a.Cin;
  BLK_T0();

  // ---- input plane -> LDS (rows past P and the operand pad rows of dsb: zero, once)
  {
    const int vpr = a.Cin / 8;                       // 16-B vectors per row
    for (int i = tid; i < MT * 16 * vpr; i += NTHR) {
      const int p = i / vpr, v = i % vpr;
      bf16x8 val;
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = (bf16_t)0.f;
      if (p < P) val = *reinterpret_cast<const bf16x8*>(xg + (size_t)p * a.Cin + v * 8);
      *reinterpret_cast<bf16x8*>(xs + (size_t)p * CinP + v * 8) = val;
    }
    for (int i = tid; i < (MT * 16 - P) * (SL / 8); i += NTHR) {
      const int p = P + i / (SL / 8), v = i % (SL / 8);
      bf16x8 zero;
#pragma unroll
      for (int j = 0; j < 8; ++j) zero[j] = (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(dsb + (size_t)p * SLP + v * 8) = zero;
    }
  }

  f32x4 out[MTW][NT2];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int t = 0; t < NT2; ++t) out[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Weights of slab sl+1 are fetched into registers while slab sl computes and land in LDS after its project phase: staged
  // in place at the top of every slab they were 7-42 us of pure L2 latency per block (15 slabs at 7x7).
  constexpr int W1V = 3, W2V = (NT2 + 3) / 4, WDV = 2;      // 16-B vectors / floats per thread (Cin <= 160, Cout = 16 NT2)
  bf16x8 pw1[W1V], pw2[W2V];
  float pwd[WDV];
  // ... and so do the slab's BatchNorm affines (one float per thread, into LDS with the weights): loaded where they are
  // used, they queued behind the weight prefetch (vmcnt is in order) and the expand phase waited out the whole fetch
  float paf = 0.f;
  const int vpr1 = a.Cin / 8;
  int g1[W1V], l1[W1V];                              // slab-relative offsets of this thread's W1 vectors (runtime divisions: once)
#pragma unroll
  for (int k = 0; k < W1V; ++k) {
    const int i = tid + NTHR * k;
    const bool v = i < SL * vpr1;
    g1[k] = v ? (i / vpr1) * a.Cin + (i % vpr1) * 8 : -1;
    l1[k] = v ? (i / vpr1) * CinP + (i % vpr1) * 8 : 0;
  }
  auto fetch_w = [&](int ce0) {
#pragma unroll
    for (int k = 0; k < W1V; ++k)
      if (g1[k] >= 0) pw1[k] = *reinterpret_cast<const bf16x8*>(a.w1 + (size_t)ce0 * a.Cin + g1[k]);
#pragma unroll
    for (int k = 0; k < W2V; ++k) {
      const int i = tid + NTHR * k;
      if (i < a.Cout * (SL / 8)) pw2[k] = *reinterpret_cast<const bf16x8*>(a.w2 + (size_t)(i / (SL / 8)) * a.Ce + ce0 + (i % (SL / 8)) * 8);
    }
#pragma unroll
    for (int k = 0; k < WDV; ++k) {
      const int i = tid + NTHR * k;
      if (i < SL * 9) pwd[k] = a.wd[(size_t)ce0 * 9 + i];
    }
    if (tid < 4 * SL) {
      const float* src = (tid < SL) ? a.s1 : (tid < 2 * SL) ? a.h1 : (tid < 3 * SL) ? a.s2 : a.h2;
      paf = src[ce0 + (tid & (SL - 1))];
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int k = 0; k < W1V; ++k)
      if (g1[k] >= 0) *reinterpret_cast<bf16x8*>(w1s + l1[k]) = pw1[k];
#pragma unroll
    for (int k = 0; k < W2V; ++k) {
      const int i = tid + NTHR * k;
      if (i < a.Cout * (SL / 8)) *reinterpret_cast<bf16x8*>(w2s + (size_t)(i / (SL / 8)) * SLP + (i % (SL / 8)) * 8) = pw2[k];
    }
#pragma unroll
    for (int k = 0; k < WDV; ++k) {
      const int i = tid + NTHR * k;
      if (i < SL * 9) wds[(i % 9) * SL + i / 9] = pwd[k];      // tap-major [9][SL]
    }
    if (tid < 4 * SL) afs[tid] = paf;
  };
  fetch_w(0);
  store_w();

  const int nslab = a.Ce / SL;
  for (int sl = 0; sl < nslab; ++sl) {
    const int ce0 = sl * SL;
    __syncthreads();                                 // this slab's weights (and, first time, the plane) are in LDS
    BLK_PH(sl == 0 ? 0 : 1);
    if (sl + 1 < nslab) fetch_w(ce0 + SL);

    // ---- expand: e[p][n] = sum_k x[p][k] W1[n][k]  -> bf16 -> BN1 + act -> a1 (fp32)
    {
      float sc[4][4], sh[4][4];                      // this lane's channels: n = 16 t + 4 lg + reg
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 c = *reinterpret_cast<const float4*>(afs + 16 * t + 4 * lg), h = *reinterpret_cast<const float4*>(afs + SL + 16 * t + 4 * lg);
        sc[t][0] = c.x; sc[t][1] = c.y; sc[t][2] = c.z; sc[t][3] = c.w;
        sh[t][0] = h.x; sh[t][1] = h.y; sh[t][2] = h.z; sh[t][3] = h.w;
      }
      for (int mt = wave; mt < MT; mt += NWAVE) {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS1; ++ks) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(xs + (size_t)(mt * 16 + lc) * CinP + ks * 32 + lg * 8);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w1s + (size_t)(t * 16 + lc) * CinP + ks * 32 + lg * 8);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, acc[t], 0, 0, 0);
          }
        }
        const int p = mt * 16 + lc;
        if (p < P) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (float)(bf16_t)acc[t][r];          // the unfused path stores e as bf16
            act_affine_vec<4>(v, sc[t], sh[t], a.act1);
            *reinterpret_cast<float4*>(a1 + (size_t)p * SLF + 16 * t + 4 * lg) = float4{v[0], v[1], v[2], v[3]};
          }
        }
      }
    }
    __syncthreads();
    BLK_PH(2);

    // ---- depthwise 3x3 (pad 1) over the plane, 8 channels per item -> bf16 -> BN2 + act -> bf16 operand.
    // An item's channel group is tid & 7 for every item of the thread (its BN2 affine stays in registers; the 72
    // stencil weights are re-read from LDS: in registers they pushed the kernel into scratch).
    {
      const int cg = tid & 7;
      float sc[8], sh[8];
      {
        const float4 c0 = *reinterpret_cast<const float4*>(afs + 2 * SL + cg * 8), c1 = *reinterpret_cast<const float4*>(afs + 2 * SL + cg * 8 + 4);
        const float4 d0 = *reinterpret_cast<const float4*>(afs + 3 * SL + cg * 8), d1 = *reinterpret_cast<const float4*>(afs + 3 * SL + cg * 8 + 4);
        sc[0] = c0.x; sc[1] = c0.y; sc[2] = c0.z; sc[3] = c0.w; sc[4] = c1.x; sc[5] = c1.y; sc[6] = c1.z; sc[7] = c1.w;
        sh[0] = d0.x; sh[1] = d0.y; sh[2] = d0.z; sh[3] = d0.w; sh[4] = d1.x; sh[5] = d1.y; sh[6] = d1.z; sh[7] = d1.w;
      }
      // item = two horizontally adjacent pixels x 8 channels: the 3x4 input neighbourhood and the 9 weight vectors are
      // read once for both (42 LDS reads per 16 outputs instead of 72)
      const int WP = (a.W + 1) / 2;
      for (int q = tid >> 3; q < a.H * WP; q += NTHR / 8) {
        const int y = q / WP, x = (q - y * WP) * 2;
        const bool two = x + 1 < a.W;
        float s0[8], s1v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s0[j] = s1v[j] = 0.f;
        // (a branch-free form -- clamped addresses + selects -- needs all 12 neighbourhood vectors live at once and
        // pushes the kernel into scratch: 2-3x slower)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = y + ky - 1;
          if (yy < 0 || yy >= a.H) continue;
          float v[4][8];                             // columns x-1 .. x+2 of this row
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int xx = x - 1 + c;
            if (xx >= 0 && xx < a.W) {
              const float* src = a1 + (size_t)(yy * a.W + xx) * SLF + cg * 8;
              const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
              v[c][0] = v0.x; v[c][1] = v0.y; v[c][2] = v0.z; v[c][3] = v0.w; v[c][4] = v1.x; v[c][5] = v1.y; v[c][6] = v1.z; v[c][7] = v1.w;
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
            }
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float4 w0 = *reinterpret_cast<const float4*>(wds + (ky * 3 + kx) * SL + cg * 8),
                         w1v = *reinterpret_cast<const float4*>(wds + (ky * 3 + kx) * SL + cg * 8 + 4);
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1v.x, w1v.y, w1v.z, w1v.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              s0[j] = fmaf(v[kx][j], wv[j], s0[j]);
              s1v[j] = fmaf(v[kx + 1][j], wv[j], s1v[j]);
            }
          }
        }
        auto finish = [&](float* s, int px) {        // always inlined on a named array: no dynamic register indexing
#pragma unroll
          for (int j = 0; j < 8; ++j) s[j] = (float)(bf16_t)s[j];                    // the unfused path stores d as bf16
          act_affine_vec<8>(s, sc, sh, a.act2);
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)s[j];
          *reinterpret_cast<bf16x8*>(dsb + (size_t)px * SLP + cg * 8) = o;
        };
        finish(s0, y * a.W + x);
        if (two) finish(s1v, y * a.W + x + 1);
      }
    }
    __syncthreads();
    BLK_PH(3);

    // ---- project: out[p][n] += sum_k d[p][k] W2[n][ce0 + k]
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const int mt = wave + NWAVE * m;
      if (mt < MT) {
#pragma unroll
        for (int ks = 0; ks < SL / 32; ++ks) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(dsb + (size_t)(mt * 16 + lc) * SLP + ks * 32 + lg * 8);
#pragma unroll
          for (int t = 0; t < NT2; ++t) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w2s + (size_t)(t * 16 + lc) * SLP + ks * 32 + lg * 8);
            out[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, out[m][t], 0, 0, 0);
          }
        }
      }
    }
    BLK_PH(4);
    if (sl + 1 < nslab) {
      __syncthreads();                               // every wave is done with this slab's weights
      store_w();
    }
  }

  BLK_PH(4);
  // ---- epilogue: y3 -> bf16 -> BN3 (+ x) -> bf16
  bf16_t* __restrict__ zg = a.z + img * P * a.Cout;
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int mt = wave + NWAVE * m, p = mt * 16 + lc;
    if (mt < MT && p < P) {
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const int n = 16 * t + 4 * lg;
        const float4 s = *reinterpret_cast<const float4*>(a.s3 + n), h = *reinterpret_cast<const float4*>(a.h3 + n);
        const float sc[4] = {s.x, s.y, s.z, s.w}, sh[4] = {h.x, h.y, h.z, h.w};
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf((float)(bf16_t)out[m][t][r], sc[r], sh[r]);
          if (a.residual) v += (float)xs[(size_t)p * CinP + n + r];
          o[r] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x4*>(zg + (size_t)p * a.Cout + n) = o;
      }
    }
  }
  BLK_PH(5);
  BLK_END();
}

template <int MTW, int NT2>
int launch_blk(const BlkArgs& a, int B, size_t lds, hipStream_t st) {
  const void* fn = (const void*)ir_block_eval_kernel<MTW, NT2>;
  static bool attr = false;
  if (!attr) {
    if (t3d_max_lds(fn, 160 * 1024) != hipSuccess) return T3D_ERR_UNSUPPORTED;
    attr = true;
  }
  T3D_LAUNCH((ir_block_eval_kernel<MTW, NT2>), dim3(B), dim3(NTHR), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

#ifdef T3D_BLK_TRACE
extern "C" int t3d_debug_blk_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk_trace), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int t3d_ir_block_eval(const void* x, const void* w1, const float* scale1, const float* shift1, int act1,
                                 const float* wdw, const float* scale2, const float* shift2, int act2, const void* w2,
                                 const float* scale3, const float* shift3, int residual, void* z, int B, int H, int W, int Cin,
                                 int Ce, int Cout, void* stream) {
  if (!x || !w1 || !scale1 || !shift1 || !wdw || !scale2 || !shift2 || !w2 || !scale3 || !shift3 || !z || B <= 0 || H <= 0 ||
      W <= 0 || Cin <= 0 || Ce <= 0 || Cout <= 0)
    return T3D_ERR_ARG;
  if ((Cin % 32) || (Ce % SL) || (Cout % 16) || (residual && Cin != Cout)) return T3D_ERR_UNSUPPORTED;
  const int P = H * W, MT = (P + 15) / 16;
  const size_t lds = (size_t)MT * 16 * (Cin + 8) * 2 + (size_t)P * SLF * 4 + (size_t)MT * 16 * SLP * 2 + (size_t)SL * (Cin + 8) * 2 +
                     (size_t)Cout * SLP * 2 + (size_t)SL * 9 * 4 + (size_t)4 * SL * 4;
  if (lds > 160 * 1024) return T3D_ERR_UNSUPPORTED;
  BlkArgs a{};
  a.x = (const bf16_t*)x; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.wd = wdw;
  a.s1 = scale1; a.h1 = shift1; a.s2 = scale2; a.h2 = shift2; a.s3 = scale3; a.h3 = shift3;
  a.z = (bf16_t*)z; a.act1 = act1; a.act2 = act2; a.residual = residual;
  a.H = H; a.W = W; a.Cin = Cin; a.Ce = Ce; a.Cout = Cout;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int mtw = (MT + NWAVE - 1) / NWAVE, nt2 = Cout / 16;
  if (mtw <= 1) {
    if (nt2 == 10) return launch_blk<1, 10>(a, B, lds, st);
    if (nt2 == 20) return launch_blk<1, 20>(a, B, lds, st);
  } else if (mtw <= 2) {
    if (nt2 == 4) return launch_blk<2, 4>(a, B, lds, st);
    if (nt2 == 6) return launch_blk<2, 6>(a, B, lds, st);
  }
  return T3D_ERR_UNSUPPORTED;
}
