// Depthwise k x k convolution forward for gfx950 (NHWC, wave64).
//
// HBM-bound (3.4-4.9 FLOP/B): one pass over the input, one over the output.
//   * input tile (+halo) is loaded once with 16-B coalesced loads, the producer's
//     BatchNorm affine + activation (+SE) is applied on the fly and the activated
//     tile is parked in LDS (zero padding is applied AFTER the activation, as the
//     reference pads the activated tensor);
//   * each thread owns 8 channels and a strip of 4 output pixels: a sliding window
//     over the LDS tile, weights for its channels held in LDS ([tap][channel]);
//   * per-channel sum / sum-of-squares for the following BatchNorm are kept in
//     registers across the block's (persistent) tile loop and leave as ONE fp64
//     atomic per channel per block.
#include <cstdlib>
#include "common.h"

namespace {

struct DwFwdArgs {
  const void* x;
  void* y;
  const float* w;      // [C][K*K]
  const float* scale;  // prologue
  const float* shift;
  const float* se;
  double* stats;
  float* gap;
  int gapq;     // pooled sums as int64 fixed point (common.h: t3d_pool_add)
  int act, se_after;
  int B, H, W, C, Ho, Wo;
  int TH, TW, tiles_x, tiles_y;  // output tile, tiles per image
  int cgb;                       // channel groups (of 8) per block
  int pix_stride;                // LDS elements per staged pixel
  int tile_off;                  // byte offset of the tile region in LDS
  int nrep;                      // reduction replicas (common.h)
  long long rstride;
};

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const DwFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PAD = (K - 1) / 2;
  constexpr int JW = 3 * S + K;  // input columns feeding a strip of 4 outputs
  float* wl = reinterpret_cast<float*>(smem);                // [K*K][cgb*8]
  T* tile = reinterpret_cast<T*>(smem + a.tile_off);          // [IH*IW][pix_stride]
  float* scratch = reinterpret_cast<float*>(smem + a.tile_off);

  const int tid = threadIdx.x;
  const int cgb = a.cgb, CB = cgb * 8;
  const int nslots = 256 / cgb;
  const int cg = tid % cgb, slot = tid / cgb;
  const int c0 = (blockIdx.y * cgb + cg) * 8;
  const bool on = (slot < nslots) && (c0 < a.C);
  const int IH = (a.TH - 1) * S + K, IW = (a.TW - 1) * S + K;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  T* __restrict__ y = reinterpret_cast<T*>(a.y);

  // weights of this block's channels -> LDS [tap][CB]
  for (int i = tid; i < K * K * CB; i += 256) {
    const int tap = i / CB, cc = i % CB;
    const int c = blockIdx.y * CB + cc;
    wl[i] = (c < a.C) ? a.w[(size_t)c * (K * K) + tap] : 0.f;
  }
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = (a.scale && on) ? a.scale[c0 + i] : 1.f;
    sh[i] = (a.scale && on) ? a.shift[c0 + i] : 0.f;
  }
  float psum[8], psq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) psum[i] = psq[i] = 0.f;

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.B * tiles_per_img;
  const int spr = a.TW / 4;  // strips per tile row
  const int nstrips = a.TH * spr;

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / tiles_per_img, r = t % tiles_per_img;
    const int oy0 = (r / a.tiles_x) * a.TH, ox0 = (r % a.tiles_x) * a.TW;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    float sev[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sev[i] = (a.se && on) ? a.se[(size_t)b * a.C + c0 + i] : 1.f;

    __syncthreads();  // previous tile fully consumed (also orders the weight fill)
    if (on) {
      for (int p = slot; p < IH * IW; p += nslots) {
        const int gy = iy0 + p / IW, gx = ix0 + p % IW;
        float v[8];
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
          Vec8<T>::load(x + (((size_t)b * a.H + gy) * a.W + gx) * a.C + c0, v);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float u = v[i] * sc[i] + sh[i];
            if (!a.se_after) u *= sev[i];
            u = act_apply(u, a.act);
            if (a.se_after) u *= sev[i];
            v[i] = u;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
        Vec8<T>::store(tile + (size_t)p * a.pix_stride + cg * 8, v);
      }
    }
    __syncthreads();

    float gs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) gs[i] = 0.f;
    if (on) {
      for (int s = slot; s < nstrips; s += nslots) {
        const int ty = s / spr, tx = (s % spr) * 4;
        float acc[4][8];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[o][i] = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          float wr[K][8];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const float4 w0 = *reinterpret_cast<const float4*>(wl + (ky * K + kx) * CB + cg * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(wl + (ky * K + kx) * CB + cg * 8 + 4);
            wr[kx][0] = w0.x; wr[kx][1] = w0.y; wr[kx][2] = w0.z; wr[kx][3] = w0.w;
            wr[kx][4] = w1.x; wr[kx][5] = w1.y; wr[kx][6] = w1.z; wr[kx][7] = w1.w;
          }
          const T* rowp = tile + ((size_t)(ty * S + ky) * IW + tx * S) * a.pix_stride + cg * 8;
#pragma unroll
          for (int j = 0; j < JW; ++j) {
            float v[8];
            Vec8<T>::load(rowp + (size_t)j * a.pix_stride, v);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
              const int kx = j - o * S;
              if (kx >= 0 && kx < K) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[o][i] = fmaf(v[i], wr[kx][i], acc[o][i]);
              }
            }
          }
        }
        const int oy = oy0 + ty;
        if (oy < a.Ho) {
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            const int ox = ox0 + tx + o;
            if (ox < a.Wo) {
              float r8[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                r8[i] = Vec8<T>::round(acc[o][i]);
                psum[i] += r8[i];
                psq[i] = fmaf(r8[i], r8[i], psq[i]);
                gs[i] += r8[i];
              }
              Vec8<T>::store(y + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.C + c0, r8);
            }
          }
        }
      }
    }
    if (a.gap) {  // per-sample channel sums for squeeze-excite (block-uniform branch)
      __syncthreads();
      if (slot < nslots) {
#pragma unroll
        for (int i = 0; i < 8; ++i) scratch[(slot * cgb + cg) * 8 + i] = gs[i];
      }
      __syncthreads();
      if (tid < CB) {
        const int c = blockIdx.y * CB + tid;
        if (c < a.C) {
          float s = 0.f;
          for (int q = 0; q < nslots; ++q) s += scratch[q * CB + tid];
          t3d_pool_add(a.gap, (size_t)b * a.C + c, s, a.gapq);
        }
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
int launch(const DwFwdArgs& a0, int k, int s, hipStream_t st) {
  DwFwdArgs a = a0;
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  const int CG = a.C / 8;
  const int nchunks = cdiv(CG, 8);
  a.cgb = cdiv(CG, nchunks);
  const int nslots = 256 / a.cgb;
  a.TW = a.Wo >= 16 ? 16 : ((a.Wo + 3) / 4) * 4;
  a.TH = (nslots * 4) / a.TW;
  if (a.TH < 1) a.TH = 1;
  if (a.TH > a.Ho) a.TH = a.Ho;
  a.pix_stride = a.cgb * 8 + 8;
  const size_t wbytes = (size_t)k * k * a.cgb * 8 * sizeof(float);
  a.tile_off = (int)((wbytes + 15) / 16 * 16);
  const size_t scratch = (size_t)256 * 8 * 2 * sizeof(float);
  size_t tile_bytes;
  for (;;) {
    const int IH = (a.TH - 1) * s + k, IW = (a.TW - 1) * s + k;
    tile_bytes = (size_t)IH * IW * a.pix_stride * sizeof(T);
    if (a.tile_off + tile_bytes <= 64 * 1024 || a.TH == 1) break;
    a.TH = (a.TH + 1) / 2;
  }
  if (a.tile_off + tile_bytes > 160 * 1024) return T3D_ERR_UNSUPPORTED;
  a.tiles_x = cdiv(a.Wo, a.TW);
  a.tiles_y = cdiv(a.Ho, a.TH);
  const size_t lds = a.tile_off + (tile_bytes > scratch ? tile_bytes : scratch);
  const long long ntiles = (long long)a.B * a.tiles_x * a.tiles_y;
  int gx = (int)(ntiles < 4096 / nchunks ? ntiles : 4096 / nchunks);
  if (gx < 1) gx = 1;
  dim3 grid(gx, nchunks);
#define T3D_DW(KK, SS)                                                                          \
  if (k == KK && s == SS) {                                                                     \
    if (lds > 64 * 1024)                                                                        \
      (void)hipFuncSetAttribute((const void*)dw_fwd_kernel<T, KK, SS>,                                \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
    T3D_LAUNCH_TIMED((dw_fwd_kernel<T, KK, SS>), grid, dim3(256), lds, st, a);                \
  }
  T3D_DW(3, 1) else T3D_DW(3, 2) else T3D_DW(5, 1) else T3D_DW(5, 2) else return T3D_ERR_UNSUPPORTED;
#undef T3D_DW
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

}  // namespace

int t3d_dw3_fwd_stream(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats, int B,
                       int H, int W, int C, int stride, hipStream_t st);   // dwconv3_stream.hip

int t3d_dwk_fwd_stream(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats,
                       float* gap_sum, int B, int H, int W, int C, int k, int stride, hipStream_t st);   // dwconvk_stream.hip

int t3d_dw5_plane7_fwd(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats, float* gap_sum,
                       int B, int C, hipStream_t st);   // dwconv5_plane7.hip

int t3d_dw_tile_fwd(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats, float* gap_sum,
                    int B, int H, int W, int C, int k, int stride, hipStream_t st);   // dwconv_tile.hip

extern "C" int t3d_dwconv_fwd(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y,
                              double* stats, float* gap_sum, int B, int H, int W, int C, int k, int stride,
                              void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8)) return T3D_ERR_ARG;
  if (k == 3 && !getenv("T3D_DW_TILED")) {
    // 3x3 on the small planes (14x14 and below by default): register tiles, every load of a tile's window up front (dwconv_tile.hip)
    const int rc = t3d_dw_tile_fwd(dtype, x, pro, w, y, stats, gap_sum, B, H, W, C, 3, stride, reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  if (k == 3 && (stride == 1 || stride == 2) && !gap_sum && !(pro && pro->se) && !getenv("T3D_DW_TILED"))
    return t3d_dw3_fwd_stream(dtype, x, pro, w, y, stats, B, H, W, C, stride, reinterpret_cast<hipStream_t>(stream));
  // the kernels below read finished coefficients: a pending derive request for them becomes a launch of its own
  if (pro)
    if (const int rc = t3d_fold_fallback(pro->scale, reinterpret_cast<hipStream_t>(stream))) return rc;
  if (k == 5 && stride == 1 && H == 7 && W == 7 && !getenv("T3D_DW_TILED")) {
    // 5x5 on 7x7 planes (the 1/32 stage of MobileNetV3): a thread per (image, channel pair) holds the plane (dwconv5_plane7.hip)
    const int rc = t3d_dw5_plane7_fwd(dtype, x, pro, w, y, stats, gap_sum, B, C, reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  if (k == 5 && !getenv("T3D_DW_TILED")) {
    // 5x5 on planes up to 64x64: register tiles, every load of a tile's window up front (dwconv_tile.hip)
    const int rc = t3d_dw_tile_fwd(dtype, x, pro, w, y, stats, gap_sum, B, H, W, C, 5, stride, reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  if ((k == 3 || k == 5) && (stride == 1 || stride == 2) && !(pro && pro->se) && !getenv("T3D_DW_TILED")) {
    // 5x5 layers and the squeeze-excite blocks (per-sample pooled sums): generic streaming kernel
    const int rc = t3d_dwk_fwd_stream(dtype, x, pro, w, y, stats, gap_sum, B, H, W, C, k, stride,
                                      reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  DwFwdArgs a{};
  a.x = x; a.y = y; a.w = w;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.se = pro->se; a.act = pro->act; a.se_after = pro->se_after_act; }
  a.stats = stats; a.gap = gap_sum; a.gapq = g_t3d_reduce.pool_exact;
  a.B = B; a.H = H; a.W = W; a.C = C;
  const int pad = (k - 1) / 2;
  a.Ho = (H + 2 * pad - k) / stride + 1;
  a.Wo = (W + 2 * pad - k) / stride + 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == T3D_F32) return launch<float>(a, k, stride, st);
  if (dtype == T3D_BF16) return launch<bf16_t>(a, k, stride, st);
  return T3D_ERR_ARG;
}
