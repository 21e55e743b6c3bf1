// Depthwise 3x3 / stride 1 forward for the 14x14 and 7x7 stages, one (image, 64-channel slab) PLANE per workgroup.
//
// The streaming kernels (dwconv3_stream.hip) walk the rows of an image with a register prefetch ring; on a 14-row image
// that walk is a latency chain with a prologue as long as the body (14x14x384: 32 us for 77 MB).  A plane of these stages
// fits LDS (196 px x 64 ch x 4 B = 50 KB), so here ALL global loads of the workgroup are issued up front, the BatchNorm
// affine + activation of the producer are applied once on the way into LDS (fp32, as the streaming kernels keep it), and
// the stencil runs from LDS over horizontally adjacent pixel pairs (3x4 neighbourhood and the 9 weight vectors read once
// for both).  Output: the raw convolution in storage precision + this layer's BatchNorm sums (replica atomics), exactly
// what t3d_dwconv_fwd's other kernels produce.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

struct PlArgs {
  const bf16_t* x;
  bf16_t* y;
  const float* w;                 // [C][9]
  const float *scale, *shift;     // producer's BatchNorm affine (may be null: finished input)
  int act;
  double* stats;                  // [2][C] (+ replicas)
  int B, H, W, C, nslab;
  int nrep;
  long long rstride;
};

constexpr int PSL = 64;           // channels per slab
constexpr int PSF = PSL + 4;      // fp32 row stride in LDS

__global__ __launch_bounds__(256) void dw3_plane_fwd_kernel(const PlArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int P = a.H * a.W;
  float* aL = lds;                              // [P][PSF]  activated input plane
  float* wL = aL + (size_t)P * PSF;             // [9][PSL]  tap-major stencil weights
  float* sL = wL + 9 * PSL;                     // [2][PSL]  sum(y), sum(y^2)
  const int tid = threadIdx.x, cg = tid & 7;
  const int slab = blockIdx.x % a.nslab, img = blockIdx.x / a.nslab, c0 = slab * PSL;
  const bf16_t* __restrict__ xg = a.x + (size_t)img * P * a.C + c0;
  bf16_t* __restrict__ yg = a.y + (size_t)img * P * a.C + c0;

  // ---- all of the plane's loads first (a thread's channel group is tid & 7 throughout)
  constexpr int MAXV = 7;                       // ceil(196 * 8 / 256)
  bf16x8 v[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int p = (tid >> 3) + 32 * k;
    if (p < P) v[k] = *reinterpret_cast<const bf16x8*>(xg + (size_t)p * a.C + cg * 8);
  }
  for (int i = tid; i < 9 * PSL; i += 256) wL[(i % 9) * PSL + i / 9] = a.w[(size_t)c0 * 9 + i];
  if (tid < 2 * PSL) sL[tid] = 0.f;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = a.scale ? a.scale[c0 + cg * 8 + j] : 1.f;
    sh[j] = a.scale ? a.shift[c0 + cg * 8 + j] : 0.f;
  }
  const bool affine = a.scale != nullptr || a.act != T3D_ACT_NONE;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int p = (tid >> 3) + 32 * k;
    if (p < P) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = (float)v[k][j];
      if (affine) act_affine_vec<8>(t, sc, sh, a.act);
      *reinterpret_cast<float4*>(aL + (size_t)p * PSF + cg * 8) = float4{t[0], t[1], t[2], t[3]};
      *reinterpret_cast<float4*>(aL + (size_t)p * PSF + cg * 8 + 4) = float4{t[4], t[5], t[6], t[7]};
    }
  }
  __syncthreads();

  // ---- stencil over pixel pairs
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const int WP = (a.W + 1) / 2;
  for (int q = tid >> 3; q < a.H * WP; q += 32) {
    const int y = q / WP, x = (q - y * WP) * 2;
    const bool two = x + 1 < a.W;
    float o0[8], o1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o0[j] = o1[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      if (yy < 0 || yy >= a.H) continue;
      float n[4][8];                             // columns x-1 .. x+2 of this row
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int xx = x - 1 + c;
        if (xx >= 0 && xx < a.W) {
          const float* src = aL + (size_t)(yy * a.W + xx) * PSF + cg * 8;
          const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
          n[c][0] = v0.x; n[c][1] = v0.y; n[c][2] = v0.z; n[c][3] = v0.w; n[c][4] = v1.x; n[c][5] = v1.y; n[c][6] = v1.z; n[c][7] = v1.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) n[c][j] = 0.f;
        }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 w0 = *reinterpret_cast<const float4*>(wL + (ky * 3 + kx) * PSL + cg * 8),
                     w1 = *reinterpret_cast<const float4*>(wL + (ky * 3 + kx) * PSL + cg * 8 + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o0[j] = fmaf(n[kx][j], wv[j], o0[j]);
          o1[j] = fmaf(n[kx + 1][j], wv[j], o1[j]);
        }
      }
    }
    auto finish = [&](float* o, int px) {
      bf16x8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r[j] = (bf16_t)o[j];
        const float f = (float)r[j];               // the sums are those of the STORED values
        s1[j] += f;
        s2[j] = fmaf(f, f, s2[j]);
      }
      *reinterpret_cast<bf16x8*>(yg + (size_t)px * a.C + cg * 8) = r;
    };
    finish(o0, y * a.W + x);
    if (two) finish(o1, y * a.W + x + 1);
  }

  // ---- BatchNorm sums: lanes with the same channel group (tid & 7) -> LDS -> one replica
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float u = s1[j], q2 = s2[j];
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { u += __shfl_xor(u, o, 64); q2 += __shfl_xor(q2, o, 64); }
      if ((tid & 63) < 8) {
        atomicAdd(sL + cg * 8 + j, u);
        atomicAdd(sL + PSL + cg * 8 + j, q2);
      }
    }
    __syncthreads();
    if (tid < 2 * PSL) {
      const int rep = blockIdx.x % a.nrep;
      atomicAdd(a.stats + (size_t)rep * a.rstride + (size_t)(tid / PSL) * a.C + c0 + (tid % PSL), (double)sL[tid]);
    }
  }
}

}  // namespace

// Called by t3d_dwconv_fwd (k = 3, stride 1, bf16, planes of <= 196 pixels, C % 64 == 0); T3D_ERR_UNSUPPORTED otherwise.
int t3d_dw3_plane_fwd(const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats, int B, int H, int W,
                      int C, hipStream_t st) {
  // OPT-IN (T3D_DW_PLANE=1): measured slower than the streaming kernels -- 14x14x384 44 us against 31, 14x14x576 63 against
  // 43, 7x7x960 38 against 22 (the load -> LDS -> barrier -> stencil chain of a 256-thread workgroup is not shorter than the
  // row walk it replaces, and 1536+ workgroups pay it in 2-3 rounds); kept as the measured negative result.
  static const bool on = getenv("T3D_DW_PLANE") != nullptr;
  if (!on || H * W > 196 || (C % PSL) || (pro && pro->se)) return T3D_ERR_UNSUPPORTED;
  PlArgs a{};
  a.x = (const bf16_t*)x; a.y = (bf16_t*)y; a.w = w; a.stats = stats;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
  a.B = B; a.H = H; a.W = W; a.C = C; a.nslab = C / PSL;
  a.nrep = g_t3d_reduce.nrep; a.rstride = g_t3d_reduce.stats_stride;
  const size_t lds = ((size_t)H * W * PSF + 9 * PSL + 2 * PSL) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)dw3_plane_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(dw3_plane_fwd_kernel, dim3(B * a.nslab), dim3(256), lds, st, a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
