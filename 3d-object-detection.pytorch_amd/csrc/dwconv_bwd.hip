// Depthwise k x k convolution backward for gfx950: data gradient AND weight gradient in ONE
// pass over HBM (NHWC, wave64).
//
// Per (output tile, 8*cgb channels) a block stages two LDS tiles with 16-B coalesced loads:
//   A = act(scale*x + shift)  over the input halo region   (the forward input, recomputed)
//   D = alpha*dz + beta*y + gamma over the output halo region (BatchNorm backward, on load)
// then
//   dgrad: every thread owns 8 channels x a strip of 4 INPUT pixels, gathers D through the
//          flipped stencil, multiplies by act'(scale*x+shift), adds the skip gradient,
//          writes dx and keeps sum(dx), sum(dx*x) for the producer's BatchNorm backward;
//   wgrad: every thread owns 8 channels x one kernel row (ky) and a share of the tile's
//          OUTPUT pixels; k*8 accumulators live in registers across the block's whole
//          (persistent) tile loop and leave as fp32 atomics once per block.
// Algorithmic traffic: read x, dz, y once, write dx once = 2*(in + out) elements.
#include <cstdlib>
#include "common.h"

namespace {

struct DwBwdArgs {
  const void *dz, *y, *x, *res;
  void* dx;
  const float* w;                      // [C][K*K]
  const float *alpha, *beta, *gamma;   // dy affine
  int per_sample;
  const float *scale, *shift;          // x prologue
  int act;
  double* stats;                       // [2][C]: sum(dx), sum(dx*x)
  float* dw;                           // [C][K*K]
  int B, H, W, C, Ho, Wo;
  int TH, TW, tiles_x, tiles_y, cgb, pix_stride;
  int a_off, d_off;                    // byte offsets of the A and D tiles
  int nrep;                            // reduction replicas (common.h)
  int* dw_used;                        // t3d_set_dw_slots: this kernel adds into the first nrep slots and says so
  long long rstride;
};

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_bwd_kernel(const DwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int P = (K - 1) / 2;
  constexpr int DLO = P / S;                 // output halo before the tile
  constexpr int DHI = (S - 1 + P) / S;       // and after it
  float* wl = reinterpret_cast<float*>(smem);                 // [K*K][CB]
  T* At = reinterpret_cast<T*>(smem + a.a_off);
  T* Dt = reinterpret_cast<T*>(smem + a.d_off);
  float* scratch = reinterpret_cast<float*>(smem + a.a_off);

  const int tid = threadIdx.x, cgb = a.cgb, CB = cgb * 8;
  const int nslots = 256 / cgb;
  const int cg = tid % cgb, slot = tid / cgb;
  const int c0 = (blockIdx.y * cgb + cg) * 8;
  const bool on = (slot < nslots) && (c0 < a.C);
  const int IH = (a.TH - 1) * S + K, IW = (a.TW - 1) * S + K;
  const int DH = a.TH + DLO + DHI, DW = a.TW + DLO + DHI;
  const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ zg = reinterpret_cast<const T*>(a.dz);
  const T* __restrict__ yg = reinterpret_cast<const T*>(a.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(a.res);
  T* __restrict__ dxg = reinterpret_cast<T*>(a.dx);

  for (int i = tid; i < K * K * CB; i += 256) {
    const int tap = i / CB, c = blockIdx.y * CB + i % CB;
    wl[i] = (c < a.C) ? a.w[(size_t)c * (K * K) + tap] : 0.f;
  }
  float sc[8], sh[8], al[8], be[8], ga[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = (a.scale && on) ? a.scale[c0 + i] : 1.f;
    sh[i] = (a.scale && on) ? a.shift[c0 + i] : 0.f;
    be[i] = on ? a.beta[c0 + i] : 0.f;
    al[i] = (on && !a.per_sample) ? a.alpha[c0 + i] : 0.f;
    ga[i] = (on && !a.per_sample) ? a.gamma[c0 + i] : 0.f;
  }
  // wgrad role: (cg, ky, part)
  const int wky = slot % K, wpart = slot / K, nparts = nslots / K;
  const bool won = on && (wpart < nparts);
  float wacc[K][8];
#pragma unroll
  for (int kx = 0; kx < K; ++kx)
#pragma unroll
    for (int i = 0; i < 8; ++i) wacc[kx][i] = 0.f;
  float psum[8], psq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) psum[i] = psq[i] = 0.f;

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.B * tiles_per_img;
  const int RH = a.TH * S, RW = a.TW * S;  // interior input region
  const int spr = RW / 4, nstrips = RH * spr;

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / tiles_per_img, r = t % tiles_per_img;
    const int oy0 = (r / a.tiles_x) * a.TH, ox0 = (r % a.tiles_x) * a.TW;
    const int iy0 = oy0 * S, ix0 = ox0 * S;
    if (a.per_sample) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        al[i] = on ? a.alpha[(size_t)b * a.C + c0 + i] : 0.f;
        ga[i] = on ? a.gamma[(size_t)b * a.C + c0 + i] : 0.f;
      }
    }
    __syncthreads();
    if (on) {
      for (int p = slot; p < IH * IW; p += nslots) {  // A tile: activated forward input, 0 outside
        const int gy = iy0 - P + p / IW, gx = ix0 - P + p % IW;
        float v[8];
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
          Vec8<T>::load(xg + (((size_t)b * a.H + gy) * a.W + gx) * a.C + c0, v);
          act_affine_vec<8>(v, sc, sh, a.act);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
        Vec8<T>::store(At + (size_t)p * a.pix_stride + cg * 8, v);
      }
      for (int p = slot; p < DH * DW; p += nslots) {  // D tile: BN-backward'ed output gradient
        const int oy = oy0 - DLO + p / DW, ox = ox0 - DLO + p % DW;
        float v[8];
        if (oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo) {
          float yv[8];
          const size_t off = (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.C + c0;
          Vec8<T>::load(zg + off, v);
          Vec8<T>::load(yg + off, yv);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = al[i] * v[i] + be[i] * yv[i] + ga[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
        Vec8<T>::store(Dt + (size_t)p * a.pix_stride + cg * 8, v);
      }
    }
    __syncthreads();

    // ---- data gradient over the interior input region ------------------------------
    if (on) {
      for (int s = slot; s < nstrips; s += nslots) {
        const int ry = s / spr, rx = (s % spr) * 4;
        const int iy = iy0 + ry;
        if (iy >= a.H) continue;
        float acc[4][8];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[o][i] = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const int ty = ry + P - ky;              // = (oy - oy0) * S
          if (ty % S != 0 && S > 1) continue;      // (ty may be negative: C++ % keeps the sign, != 0 still right)
          const int dyr = ty / S + DLO;            // row in the D tile (ty >= -P  ->  >= 0 after the offset)
          if (ty < -DLO * S) continue;
          const T* drow = Dt + (size_t)dyr * DW * a.pix_stride + cg * 8;
#pragma unroll
          for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              // rx is a multiple of 4 and the region origin is even, so the parity is static
              if ((o + P - kx) % S != 0) continue;
              const int dxc = (rx + o + P - kx) / S + DLO;  // rx+o+P-kx may be negative only when %S != 0 is false for S=1
              if (rx + o + P - kx < -DLO * S) continue;
              float v[8];
              Vec8<T>::load(drow + (size_t)dxc * a.pix_stride, v);
              const float* wp = wl + (ky * K + kx) * CB + cg * 8;
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[o][i] = fmaf(v[i], wp[i], acc[o][i]);
            }
          }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int ix = ix0 + rx + o;
          if (ix >= a.W) continue;
          const size_t off = (((size_t)b * a.H + iy) * a.W + ix) * a.C + c0;
          float xv[8], o8[8];
          if (a.scale || a.stats) Vec8<T>::load(xg + off, xv);
#pragma unroll
          for (int i = 0; i < 8; ++i) o8[i] = acc[o][i];
          if (a.scale) act_grad_affine_vec<8>(o8, xv, sc, sh, a.act);
          if (rg) {
            float rr[8];
            Vec8<T>::load(rg + off, rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) o8[i] += rr[i];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            o8[i] = Vec8<T>::round(o8[i]);
            if (a.stats) {
              psum[i] += o8[i];
              psq[i] = fmaf(o8[i], xv[i], psq[i]);
            }
          }
          Vec8<T>::store(dxg + off, o8);
        }
      }
    }
    // ---- weight gradient over the tile's output pixels --------------------------------
    if (won && a.dw) {
      for (int p = wpart; p < a.TH * a.TW; p += nparts) {
        const int ty = p / a.TW, tx = p % a.TW;
        float dv[8];
        Vec8<T>::load(Dt + ((size_t)(ty + DLO) * DW + tx + DLO) * a.pix_stride + cg * 8, dv);
        const T* arow = At + ((size_t)(ty * S + wky) * IW + tx * S) * a.pix_stride + cg * 8;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          float av[8];
          Vec8<T>::load(arow + (size_t)kx * a.pix_stride, av);
#pragma unroll
          for (int i = 0; i < 8; ++i) wacc[kx][i] = fmaf(av[i], dv[i], wacc[kx][i]);
        }
      }
    }
  }

  // ---- block-level reductions ------------------------------------------------------------
  if (a.dw) {
    if (a.dw_used && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.dw_used = a.nrep;   // (atomics into the first nrep slots)
    __syncthreads();
    // scratch [slot][cg][kx][8]
    if (slot < nslots) {
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int i = 0; i < 8; ++i) scratch[((slot * cgb + cg) * K + kx) * 8 + i] = won ? wacc[kx][i] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < CB * K * K; i += 256) {
      const int cc = i / (K * K), tap = i % (K * K);
      const int ky = tap / K, kx = tap % K;
      const int c = blockIdx.y * CB + cc;
      if (c < a.C) {
        float s = 0.f;
        for (int q = 0; q < nparts; ++q) {
          const int sl = q * K + ky;
          s += scratch[((sl * cgb + (cc >> 3)) * K + kx) * 8 + (cc & 7)];
        }
        unsafeAtomicAdd(a.dw + (size_t)(blockIdx.x % a.nrep) * a.C * (K * K) + (size_t)c * (K * K) + tap, s);
      }
    }
  }
  if (a.stats) {
    __syncthreads();
    if (slot < nslots) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        scratch[((slot * cgb + cg) * 8 + i) * 2 + 0] = psum[i];
        scratch[((slot * cgb + cg) * 8 + i) * 2 + 1] = psq[i];
      }
    }
    __syncthreads();
    if (tid < 2 * CB) {
      const int cc = tid >> 1, which = tid & 1;
      const int c = blockIdx.y * CB + cc;
      if (c < a.C) {
        double s = 0.0;
        for (int q = 0; q < nslots; ++q) s += (double)scratch[(q * CB + cc) * 2 + which];
        atomicAdd(a.stats + (size_t)(blockIdx.x % a.nrep) * a.rstride + (size_t)which * a.C + c, s);
      }
    }
  }
}

template <typename T>
int launch(const DwBwdArgs& a0, int k, int s, hipStream_t st) {
  DwBwdArgs a = a0;
  a.nrep = g_t3d_reduce.nrep;
  a.dw_used = a.dw ? g_t3d_reduce.dw_used : nullptr;
  a.rstride = g_t3d_reduce.stats_stride;
  const int P = (k - 1) / 2, DLO = P / s, DHI = (s - 1 + P) / s;
  const int CG = a.C / 8;
  const int nchunks = cdiv(CG, 8);
  a.cgb = cdiv(CG, nchunks);
  const int nslots = 256 / a.cgb;
  // interior input region = (TH*s) x (TW*s); aim at ~nslots strips of 4 input pixels
  a.TW = (a.Wo * s >= 16) ? 16 / s : ((a.Wo + 3) / 4) * 4;
  if (a.TW * s % 4) a.TW = ((a.TW + 3) / 4) * 4;
  const int th_mul = 1;
  a.TH = th_mul * (nslots * 4) / (a.TW * s * s);
  if (a.TH < 1) a.TH = 1;
  if (a.TH > a.Ho) a.TH = a.Ho;
  a.pix_stride = a.cgb * 8 + 8;
  const size_t wbytes = (size_t)k * k * a.cgb * 8 * sizeof(float);
  a.a_off = (int)((wbytes + 15) / 16 * 16);
  const size_t scratch = (size_t)256 * (k > 2 ? k : 2) * 8 * sizeof(float);
  size_t abytes, dbytes;
  for (;;) {
    const int IH = (a.TH - 1) * s + k, IW = (a.TW - 1) * s + k;
    const int DH = a.TH + DLO + DHI, DW = a.TW + DLO + DHI;
    abytes = (size_t)IH * IW * a.pix_stride * sizeof(T);
    dbytes = (size_t)DH * DW * a.pix_stride * sizeof(T);
    if (a.a_off + abytes + dbytes <= 64 * 1024 || a.TH == 1) break;
    a.TH = (a.TH + 1) / 2;
  }
  abytes = (abytes + 15) / 16 * 16;
  a.d_off = a.a_off + (int)abytes;
  size_t lds = a.d_off + dbytes;
  if (lds < a.a_off + scratch) lds = a.a_off + scratch;
  if (lds > 160 * 1024) return T3D_ERR_UNSUPPORTED;
  a.tiles_x = cdiv(a.Wo, a.TW);
  a.tiles_y = cdiv(a.Ho, a.TH);
  const long long ntiles = (long long)a.B * a.tiles_x * a.tiles_y;
  int gx = (int)(ntiles < 2048 / nchunks ? ntiles : 2048 / nchunks);
  if (gx < 1) gx = 1;
  dim3 grid(gx, nchunks);
#define T3D_DWB(KK, SS)                                                                             \
  if (k == KK && s == SS) {                                                                         \
    if (lds > 64 * 1024)                                                                            \
      (void)hipFuncSetAttribute((const void*)dw_bwd_kernel<T, KK, SS>,                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    T3D_LAUNCH_TIMED((dw_bwd_kernel<T, KK, SS>), grid, dim3(256), lds, st, a);                    \
  }
  T3D_DWB(3, 1) else T3D_DWB(3, 2) else T3D_DWB(5, 1) else T3D_DWB(5, 2) else return T3D_ERR_UNSUPPORTED;
#undef T3D_DWB
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

int t3d_dw3_bwd_stream(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                       const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int H,
                       int W, int C, int stride, hipStream_t st);   // dwconv3_bwd_stream.hip

int t3d_dw5_bwd_stream(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                       const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int H,
                       int W, int C, int stride, hipStream_t st);   // dwconv5_bwd_stream.hip

int t3d_dw5_plane7_bwd(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                       const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int C,
                       hipStream_t st);   // dwconv5_plane7.hip

int t3d_dw_tile_bwd(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w, const void* x,
                    const t3d_prologue* pro, const void* residual, void* dx, double* stats, float* dw, int B, int H, int W, int C,
                    int k, int stride, hipStream_t st);   // dwconv_tile.hip

extern "C" int t3d_dwconv_bwd(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w,
                              const void* x, const t3d_prologue* pro, const void* residual, void* dx, double* stats,
                              float* dw, int B, int H, int W, int C, int k, int stride, void* stream) {
  if (!dz || !y || !bb || !w || !x || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8)) return T3D_ERR_ARG;
  if (pro && pro->se) return T3D_ERR_UNSUPPORTED;  // no SE gate ever precedes a depthwise conv
  if (k == 3 && !getenv("T3D_DW_TILED")) {   // small planes: register tiles (dwconv_tile.hip)
    const int rc = t3d_dw_tile_bwd(dtype, dz, y, bb, w, x, pro, residual, dx, stats, dw, B, H, W, C, 3, stride, reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  if (k == 3 && (stride == 1 || (stride == 2 && !T3D_ENV_SET("T3D_DW_BWD_S2_TILED"))) && !getenv("T3D_DW_TILED")) {   // streaming kernel (dwconv3_bwd_stream.hip)
    const int rc = t3d_dw3_bwd_stream(dtype, dz, y, bb, w, x, pro, residual, dx, stats, dw, B, H, W, C, stride,
                                      reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  if (k == 5 && stride == 1 && H == 7 && W == 7 && !getenv("T3D_DW_TILED")) {   // 7x7 planes in registers (dwconv5_plane7.hip)
    const int rc = t3d_dw5_plane7_bwd(dtype, dz, y, bb, w, x, pro, residual, dx, stats, dw, B, C, reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  if (k == 5 && !getenv("T3D_DW_TILED")) {   // register tiles (dwconv_tile.hip)
    const int rc = t3d_dw_tile_bwd(dtype, dz, y, bb, w, x, pro, residual, dx, stats, dw, B, H, W, C, 5, stride, reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  if (k == 5 && !getenv("T3D_DW_TILED")) {   // 5x5: streaming kernels (dwconv5_bwd_stream.hip)
    const int rc = t3d_dw5_bwd_stream(dtype, dz, y, bb, w, x, pro, residual, dx, stats, dw, B, H, W, C, stride,
                                      reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  // the tiled kernel reads finished coefficients: a pending derive request for them becomes a launch of its own
  if (const int rc = t3d_fold_fallback(bb->alpha, reinterpret_cast<hipStream_t>(stream))) return rc;
  DwBwdArgs a{};
  a.dz = dz; a.y = y; a.x = x; a.res = residual; a.dx = dx; a.w = w;
  a.alpha = bb->alpha; a.beta = bb->beta; a.gamma = bb->gamma; a.per_sample = bb->per_sample;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
  a.stats = stats; a.dw = dw;
  a.B = B; a.H = H; a.W = W; a.C = C;
  const int pad = (k - 1) / 2;
  a.Ho = (H + 2 * pad - k) / stride + 1;
  a.Wo = (W + 2 * pad - k) / stride + 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == T3D_F32) return launch<float>(a, k, stride, st);
  if (dtype == T3D_BF16) return launch<bf16_t>(a, k, stride, st);
  return T3D_ERR_ARG;
}
