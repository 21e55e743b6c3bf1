// Crop + bilinear resize of detections out of a full uint8 frame, on the GPU -- the input side of the two-stage pipeline.
//
// Replaces, for the batched regression stage, the reference's per-detection host loop
//   crop = frame[y0:y1, x0:x1]                      (utils/ie_wrappers.py:154-158, dataloaders/objectron_main.py:98-127)
//   cv.resize(crop, (w, h))                          (utils/ie_wrappers.py:18-21; albumentations Resize in the loaders)
// with one launch over all detections of a frame: frame [H,W,3] uint8 -> crops [n,oh,ow,3] uint8 NHWC, which the stem's
// patch gather (t3d_stem_im2col_u8) normalises and consumes as they are.
//
// Arithmetic: cv::resize's INTER_LINEAR for 8-bit images restated (OpenCV is not in this image: parity with cv2 itself is
// UNPINNED; oracle/crop_resize.py is the same restatement in numpy and the kernel is bit-exact against it):
//   source coordinate  fx = (float)((dx + 0.5) * (sw / ow) - 0.5),  sx = floor(fx),  fx -= sx
//   sx < 0 -> sx = 0, fx = 0;   sx >= sw-1 -> sx = sw-1, fx = 0      (columns);   rows are clipped to [0, sh-1] instead
//   11-bit fixed-point weights  a1 = rint(fx * 2048), a0 = rint((1 - fx) * 2048)   (round half to even)
//   horizontal  D = S[sx] * a0 + S[sx+1] * a1;   vertical  out = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

struct Lin { int i0, i1, w0, w1; };

// column rule (zero the fraction at the borders) or row rule (clip the two taps)
__device__ __forceinline__ Lin lin_coef(int d, int ssize, int dsize, bool column) {
  const double scale = (double)ssize / (double)dsize;
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  Lin r;
  if (column) {
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
    r.i0 = s;
    r.i1 = min(s + 1, ssize - 1);
  } else {
    r.i0 = min(max(s, 0), ssize - 1);
    r.i1 = min(max(s + 1, 0), ssize - 1);
  }
  r.w0 = (int)rintf((1.f - f) * 2048.f);
  r.w1 = (int)rintf(f * 2048.f);
  return r;
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const unsigned char* __restrict__ frame, const int* __restrict__ rects,
                                                          unsigned char* __restrict__ out, int n, int H, int W, int oh, int ow) {
  const size_t total = (size_t)n * oh * ow;
  for (size_t p = blockIdx.x * (size_t)256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
    const int dx = (int)(p % ow), dy = (int)((p / ow) % oh), i = (int)(p / ((size_t)ow * oh));
    // numpy slicing of the reference: frame[y0:y1, x0:x1] with the bounds clamped to the frame
    const int x0 = min(max(rects[4 * i], 0), W), y0 = min(max(rects[4 * i + 1], 0), H);
    const int x1 = min(max(rects[4 * i + 2], 0), W), y1 = min(max(rects[4 * i + 3], 0), H);
    const int sw = x1 - x0, sh = y1 - y0;
    unsigned char* o = out + p * 3;
    if (sw <= 0 || sh <= 0) {      // empty crop (cv.resize would raise): zeros, the caller drops such detections
      o[0] = o[1] = o[2] = 0;
      continue;
    }
    const Lin cx = lin_coef(dx, sw, ow, true), cy = lin_coef(dy, sh, oh, false);
    const unsigned char* r0 = frame + ((size_t)(y0 + cy.i0) * W + x0) * 3;
    const unsigned char* r1 = frame + ((size_t)(y0 + cy.i1) * W + x0) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int d0 = r0[cx.i0 * 3 + c] * cx.w0 + r0[cx.i1 * 3 + c] * cx.w1;
      const int d1 = r1[cx.i0 * 3 + c] * cx.w0 + r1[cx.i1 * 3 + c] * cx.w1;
      const int v = (((cy.w0 * (d0 >> 4)) >> 16) + ((cy.w1 * (d1 >> 4)) >> 16) + 2) >> 2;
      o[c] = (unsigned char)min(max(v, 0), 255);
    }
  }
}

}  // namespace

extern "C" int t3d_crop_resize_u8(const unsigned char* frame, const int* rects, unsigned char* out, int n, int H, int W, int oh,
                                  int ow, void* stream) {
  if (!frame || !rects || !out || n < 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0) return T3D_ERR_ARG;
  if (n == 0) return T3D_OK;
  const size_t total = (size_t)n * oh * ow;
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  T3D_LAUNCH(crop_resize_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), frame, rects, out, n, H,
                     W, oh, ow);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
