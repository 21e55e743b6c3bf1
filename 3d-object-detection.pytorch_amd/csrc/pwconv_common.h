// Shared between the two pointwise-conv GEMM kernels (pwconv.hip: LDS-staged, fp32 + bf16;
// pwconv_stream.hip: barrier-free streaming kernel, bf16).
#pragma once
#include "common.h"

namespace t3d_pw {

struct GemmArgs {
  const void* a0;   // FWD: x (raw or finished); DGRAD: dz
  const void* a1;   // DGRAD: y (raw output of the differentiated conv), else null
  const float *p0, *p1, *p2;  // FWD: scale, shift, se[B][K]; DGRAD: alpha, beta, gamma
  int act, se_after, per_sample, dgrad;
  const void* w;     // [Nout][Kin] storage dtype
  const float* bias; // [Nout] or null
  const void* e_y;   // DGRAD epilogue: raw input tensor of the forward conv [M][Nout]
  const float *e_scale, *e_shift, *e_se;
  int e_act, e_se_after;
  const void* e_res;  // residual gradient to add [M][Nout]
  void* out;
  double* stats;      // [2][Nout]
  float* ps_stats;    // [B][Nout][2] per-sample sums (SE case)
  int ps_wave;        // per-sample sums: a WAVE owns whole samples (small planes; set by the launcher), else the workgroup does
  int M, HW, Kin, Nout, mtiles;
  int kz;             // generic kernel: contraction split over blockIdx.z (fp32 plain products of few-pixel layers),
  float* part;        // ... partial tiles [kz][M][Nout] in the workspace
  // second activation segment (y-free data gradient, streaming kernel only): k-steps >= ks1 read a2 [M][Kin2];
  // `Kin` is then the padded total (ks1*32 + round_up(Kin2, 32)) and indexes the weight rows directly
  const void* a2;
  int Kin2, ks1;
  int row0;   // row stride (= channel count) of a0 / a1; equals Kin without a second segment
  const T3dFold* fold;   // BatchNorm finalize of the operand's coefficients derived in the prologue (streaming kernel only)
  // materialising forward (t3d_pwconv_fwd_mat, streaming kernel only): the operand is z = bf16(p0*a0 + p1 + z_res), i.e.
  // the finished block output that t3d_bn_apply would have written; the blocks of output chunk 0 also STORE it to z_out
  const void* z_res;
  void* z_out;
  // implicit 3x3 convolution (deep-contraction kernel only; resnet.hip: t3d_conv3x3_fwd / _dgrad): the operand row of pixel m
  // is GATHERED -- k = tap * cv.Cs + c reads channel c of the source pixel that tap (ky, kx) pairs with destination pixel m --
  // instead of read from a patch matrix in HBM.  mode 1 (forward): source = destination * stride - 1 + (ky, kx); mode 2 (data
  // gradient): source = (destination + 1 - (ky, kx)) / stride where that divides.  Out-of-range taps contribute zero AFTER
  // the operand transform (the convolution pads the activated tensor; the BatchNorm-backward affine has a constant term).
  struct Conv3 { int mode, Dh, Dw, Sh, Sw, Cs, lgCs, stride; } cv;
  int wfrag;             // `w` is the fragment-order copy (include/t3d.h: T3D_W_FRAG; streaming and deep-contraction kernels)
  T3dQuant quant;        // forward BatchNorm sums snapped onto a fixed grid (order-independent, common.h); q == 0: off
};

template <typename T> __device__ __forceinline__ void ldvec(const T* p, float* v);
template <> __device__ __forceinline__ void ldvec<float>(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void ldvec<bf16_t>(const bf16_t* p, float* v) { Vec8<bf16_t>::load(p, v); }
template <typename T> __device__ __forceinline__ void stvec(T* p, const float* v);
template <> __device__ __forceinline__ void stvec<float>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void stvec<bf16_t>(bf16_t* p, const float* v) { Vec8<bf16_t>::store(p, v); }


// bf16 streaming kernel (pwconv_stream.hip); returns T3D_ERR_UNSUPPORTED when the shape does not fit it
int stream_launch(GemmArgs& a, hipStream_t st);
// bf16 kernel for deep contractions with wide outputs (pwconv_deep.hip: operand staged once, fragment-order weights streamed
// from L2); returns T3D_ERR_UNSUPPORTED for every other shape.  deep_shape: the shapes it takes (t3d_pwconv_wants_frag)
int deep_launch(GemmArgs& a, hipStream_t st);
bool deep_shape(int Kin, int Nout);
// bf16 materialising forward of shallow contractions with wide outputs on the small planes (pwconv_wide.hip: operand staged once,
// all output channels per workgroup); T3D_ERR_UNSUPPORTED for every other launch.  wide_shape: layers whose fragment-order copy it wants
int wide_launch(GemmArgs& a, hipStream_t st);
bool wide_shape(int Kin, int Nout);
// the same kernel in fp16 storage, inference forward only (pwconv_stream_f16.hip)
int stream_launch_f16(GemmArgs& a, hipStream_t st);
// fp32 storage, inference forward of many-pixel layers (pwconv_f32_reg.hip); T3D_ERR_UNSUPPORTED for everything else
int f32_reg_launch(GemmArgs& a, hipStream_t st);

}  // namespace t3d_pw
