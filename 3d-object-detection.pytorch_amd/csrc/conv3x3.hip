// Dense 3x3 convolution (pad 1, stride 1 or 2) as an IMPLICIT GEMM, bf16 storage, gfx950 -- ResNet-50's conv2 layers (BASELINE
// config 4; the reference wraps torchvision / timm backbones the same way, torchdet3d/builders/model_builder.py:73-151).
//
// The first version of these layers (resnet.hip: t3d_im2col + the 1x1 GEMM kernels + t3d_col2im_bwd) wrote a 9x patch matrix
// to HBM in the forward, read it again for the weight gradient, and sent the data gradient through a second 9x matrix: 60 % of
// the model's 15-ms step.  Here the three GEMMs gather their operand rows THEMSELVES: contraction index k = tap * C + c reads
// channel c of the pixel that tap (ky, kx) pairs with the output pixel, straight from the activation tensor (16-byte vectors:
// 8 consecutive channels of one pixel are contiguous in NHWC), BatchNorm + activation -- or the BatchNorm-backward affine --
// applied on the way into the MFMA operand exactly as in the 1x1 layers, out-of-image taps zeroed after that transform.  The
// nine taps of a pixel re-read lines its neighbours' taps have just fetched (L2 / Infinity Cache), so HBM sees each tensor about
// once.  Kernels: the deep-contraction GEMM (pwconv_deep.hip, GemmArgs::cv) for the forward and the data gradient (the same
// gather with the transposed, tap-flipped pairing and (C x 9N) weights), the transposing weight-gradient GEMM
// (pwconv_wgrad_tr.hip, WgtArgs::cv) for dW in patch-column order [N][9C] (t3d_unpack_conv_grad turns it into [N][C][3][3]).
#include "pwconv_common.h"

namespace {

__global__ void pack_dgrad_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int N, int C) {
  // out [C][9 N]: out[c][t * N + n] = w[n][c][t]   (t = ky * 3 + kx; the data gradient contracts over (tap, output channel))
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= C * 9 * N) return;
  const int c = e / (9 * N), r = e - c * 9 * N, t = r / N, n = r - t * N;
  out[e] = (bf16_t)w[((size_t)n * C + c) * 9 + t];
}

inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

int t3d_pw_wgrad_tr_conv3(const void* dz, const void* y, const t3d_bnbwd* bb, const void* x, const t3d_prologue* pro, float* dw,
                          int B, int H, int W, int C, int N, int stride, hipStream_t st);      // pwconv_wgrad_tr.hip

// include/t3d.h
extern "C" int t3d_conv3x3_fwd(int dtype, const void* x, const t3d_prologue* pro, const void* w_frag, void* y, double* stats,
                               int B, int H, int W, int C, int N, int stride, void* stream) {
  if (!x || !w_frag || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return T3D_ERR_ARG;
  if (dtype != (T3D_BF16 | T3D_W_FRAG) || (pro && pro->se) || (N % 8)) return T3D_ERR_UNSUPPORTED;
  t3d_pw::GemmArgs a{};
  a.a0 = x;
  if (pro) { a.p0 = pro->scale; a.p1 = pro->shift; a.act = pro->act; }
  a.w = w_frag; a.wfrag = 1; a.out = y; a.stats = stats;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  a.M = B * Ho * Wo; a.HW = Ho * Wo; a.Kin = 9 * C; a.Nout = N;
  a.cv = {1, Ho, Wo, H, W, C, ilog2(C), stride};
  return t3d_pw::deep_launch(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int t3d_conv3x3_dgrad(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const void* wd_frag,
                                 const void* x_raw, const t3d_prologue* pro_in, void* dx, double* stats, int B, int H, int W, int C,
                                 int N, int stride, void* stream) {
  if (!dz || !y || !bb || !bb->alpha || !bb->beta || !bb->gamma || !wd_frag || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0)
    return T3D_ERR_ARG;
  if (dtype != (T3D_BF16 | T3D_W_FRAG) || bb->per_sample || (pro_in && pro_in->se) || (C % 8)) return T3D_ERR_UNSUPPORTED;
  t3d_pw::GemmArgs a{};
  a.dgrad = 1;
  a.a0 = dz; a.a1 = y;
  a.p0 = bb->alpha; a.p1 = bb->beta; a.p2 = bb->gamma;
  a.w = wd_frag; a.wfrag = 1;
  if (x_raw) {
    a.e_y = x_raw;
    if (pro_in) { a.e_scale = pro_in->scale; a.e_shift = pro_in->shift; a.e_act = pro_in->act; }
  }
  a.out = dx; a.stats = stats;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  a.M = B * H * W; a.HW = H * W; a.Kin = 9 * N; a.Nout = C;     // destination = the conv's INPUT pixels, contraction over (tap, n)
  a.cv = {2, H, W, Ho, Wo, N, ilog2(N), stride};
  return t3d_pw::deep_launch(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int t3d_conv3x3_wgrad(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const void* x, const t3d_prologue* pro,
                                 float* dw_packed, int B, int H, int W, int C, int N, int stride, void* stream) {
  if (!dz || !y || !bb || !bb->alpha || !bb->beta || !bb->gamma || !x || !dw_packed || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0)
    return T3D_ERR_ARG;
  if (dtype != T3D_BF16 || bb->per_sample || (pro && pro->se) || (C & (C - 1)) || C < 8 || (N % 8) || (stride != 1 && stride != 2))
    return T3D_ERR_UNSUPPORTED;
  return t3d_pw_wgrad_tr_conv3(dz, y, bb, x, pro, dw_packed, B, H, W, C, N, stride, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int t3d_pack_conv3x3_dgrad_weight(const float* w, void* out, int N, int C, void* stream) {
  if (!w || !out || N <= 0 || C <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(pack_dgrad_kernel, dim3(cdiv(C * 9 * N, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
             reinterpret_cast<bf16_t*>(out), N, C);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
