// SSD post-processing of the two-stage pipeline's detector (configs/detection/mnv2_ssd_300_2_heads.py:15-39,65-69 of the
// reference: SSDHead on two feature maps, DeltaXYWHBBoxCoder with stds (0.1, 0.1, 0.2, 0.2), softmax scores, per-class
// NMS at IoU 0.45 over the candidates above 0.02) in ONE launch per frame batch: a workgroup owns (class, image), decodes
// every anchor's box and that class's softmax score into LDS and runs greedy NMS there (arg-max of the remaining
// scores, emit, suppress).  The external mmdetection fork that implements those config entries is not part of the
// reference tree; the arithmetic follows the published mmdet definitions (delta2bbox, multiclass_nms) -- parity
// unpinned, restated in oracle/ssd.py.
#include "common.h"

namespace {

struct SsdLevel {
  const void *cls, *reg;   // [B*HW][cls_stride], [B*HW][reg_stride] (storage dtype)
  int HW, A, cls_stride, reg_stride;
};

struct SsdArgs {
  SsdLevel lv[2];
  int nlevels, dtype;
  const float* anchors;    // [Atot][4] x1, y1, x2, y2 (pixels of the network input)
  int Atot, nc, maxk;
  float score_thr, iou_thr, W, H, sx, sy, sw, sh, max_ratio;
  float* out;              // [B][nc][maxk][6]: x1, y1, x2, y2 (pixels), score, label
  int* counts;             // [B][nc]
};

__device__ __forceinline__ float ldval(const void* p, size_t i, int dtype) {
  return dtype == T3D_F32 ? reinterpret_cast<const float*>(p)[i] : (float)reinterpret_cast<const bf16_t*>(p)[i];
}

__global__ __launch_bounds__(256) void ssd_decode_nms_kernel(const SsdArgs a) {
  extern __shared__ float sm[];
  float* score = sm;                 // [Atot]
  float* box = sm + a.Atot;          // [Atot][4]
  __shared__ float rbest[4];
  __shared__ int ribest[4];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  // ---- decode
  for (int i = tid; i < a.Atot; i += 256) {
    int l = 0, j = i;
    if (a.nlevels > 1 && i >= a.lv[0].HW * a.lv[0].A) { l = 1; j = i - a.lv[0].HW * a.lv[0].A; }
    const SsdLevel& L = a.lv[l];
    const int px = j / L.A, an = j - px * L.A;
    const size_t row = (size_t)b * L.HW + px;
    const size_t cb = row * L.cls_stride + (size_t)an * (a.nc + 1);
    float mx = -3.0e38f;
    for (int k = 0; k <= a.nc; ++k) mx = fmaxf(mx, ldval(L.cls, cb + k, a.dtype));
    float den = 0.f;
    for (int k = 0; k <= a.nc; ++k) den += expf(ldval(L.cls, cb + k, a.dtype) - mx);
    const float p = expf(ldval(L.cls, cb + c, a.dtype) - mx) / den;
    score[i] = p > a.score_thr ? p : 0.f;
    const size_t rb = row * L.reg_stride + (size_t)an * 4;
    const float dx = ldval(L.reg, rb, a.dtype) * a.sx, dy = ldval(L.reg, rb + 1, a.dtype) * a.sy;
    float dw = ldval(L.reg, rb + 2, a.dtype) * a.sw, dh = ldval(L.reg, rb + 3, a.dtype) * a.sh;
    dw = fminf(fmaxf(dw, -a.max_ratio), a.max_ratio);
    dh = fminf(fmaxf(dh, -a.max_ratio), a.max_ratio);
    const float ax1 = a.anchors[4 * i], ay1 = a.anchors[4 * i + 1], ax2 = a.anchors[4 * i + 2], ay2 = a.anchors[4 * i + 3];
    const float pw = ax2 - ax1, ph = ay2 - ay1, pxc = 0.5f * (ax1 + ax2), pyc = 0.5f * (ay1 + ay2);
    const float gw = pw * expf(dw), gh = ph * expf(dh), gx = pxc + pw * dx, gy = pyc + ph * dy;
    box[4 * i] = fminf(fmaxf(gx - 0.5f * gw, 0.f), a.W);
    box[4 * i + 1] = fminf(fmaxf(gy - 0.5f * gh, 0.f), a.H);
    box[4 * i + 2] = fminf(fmaxf(gx + 0.5f * gw, 0.f), a.W);
    box[4 * i + 3] = fminf(fmaxf(gy + 0.5f * gh, 0.f), a.H);
  }
  __syncthreads();
  // ---- greedy NMS: highest remaining score first, lowest anchor index on ties
  float* o = a.out + ((size_t)b * a.nc + c) * a.maxk * 6;
  int k = 0;
  for (; k < a.maxk; ++k) {
    float bs = 0.f;
    int bi = 0x7fffffff;
    for (int i = tid; i < a.Atot; i += 256) {
      const float s = score[i];
      if (s > bs) { bs = s; bi = i; }      // ascending i per thread: the first maximum is kept
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(bs, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
    }
    if ((tid & 63) == 0) { rbest[tid >> 6] = bs; ribest[tid >> 6] = bi; }
    __syncthreads();
    bs = rbest[0]; bi = ribest[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (rbest[w] > bs || (rbest[w] == bs && ribest[w] < bi)) { bs = rbest[w]; bi = ribest[w]; }
    __syncthreads();
    if (bs <= 0.f) break;
    const float x1 = box[4 * bi], y1 = box[4 * bi + 1], x2 = box[4 * bi + 2], y2 = box[4 * bi + 3];
    if (tid == 0) {
      o[6 * k] = x1; o[6 * k + 1] = y1; o[6 * k + 2] = x2; o[6 * k + 3] = y2; o[6 * k + 4] = bs; o[6 * k + 5] = (float)c;
    }
    const float ar = (x2 - x1) * (y2 - y1);
    for (int i = tid; i < a.Atot; i += 256) {
      if (score[i] <= 0.f) continue;
      const float ix1 = fmaxf(x1, box[4 * i]), iy1 = fmaxf(y1, box[4 * i + 1]);
      const float ix2 = fminf(x2, box[4 * i + 2]), iy2 = fminf(y2, box[4 * i + 3]);
      const float inter = fmaxf(ix2 - ix1, 0.f) * fmaxf(iy2 - iy1, 0.f);
      const float ai = (box[4 * i + 2] - box[4 * i]) * (box[4 * i + 3] - box[4 * i + 1]);
      const float iou = inter / fmaxf(ar + ai - inter, 1e-6f);
      if (i == bi || iou > a.iou_thr) score[i] = 0.f;
    }
    __syncthreads();
  }
  if (tid == 0) a.counts[b * a.nc + c] = k;
}

}  // namespace

// include/t3d.h
extern "C" int t3d_ssd_decode_nms(int dtype, int nlevels, const void* const* cls, const void* const* reg, const int* hw,
                                  const int* nanchors, const int* cls_stride, const int* reg_stride, const float* anchors,
                                  int B, int num_classes, float score_thr, float iou_thr, int max_per_class, float img_w,
                                  float img_h, const float* stds, float* out, int* counts, void* stream) {
  if (!cls || !reg || !hw || !nanchors || !cls_stride || !reg_stride || !anchors || !stds || !out || !counts) return T3D_ERR_ARG;
  if (nlevels < 1 || nlevels > 2 || B <= 0 || num_classes <= 0 || max_per_class <= 0) return T3D_ERR_ARG;
  if (dtype != T3D_F32 && dtype != T3D_BF16) return T3D_ERR_ARG;
  SsdArgs a{};
  a.nlevels = nlevels; a.dtype = dtype;
  int tot = 0;
  for (int l = 0; l < nlevels; ++l) {
    if (!cls[l] || !reg[l] || hw[l] <= 0 || nanchors[l] <= 0 || cls_stride[l] < nanchors[l] * (num_classes + 1) ||
        reg_stride[l] < nanchors[l] * 4)
      return T3D_ERR_ARG;
    a.lv[l] = SsdLevel{cls[l], reg[l], hw[l], nanchors[l], cls_stride[l], reg_stride[l]};
    tot += hw[l] * nanchors[l];
  }
  a.anchors = anchors; a.Atot = tot; a.nc = num_classes; a.maxk = max_per_class;
  a.score_thr = score_thr; a.iou_thr = iou_thr; a.W = img_w; a.H = img_h;
  a.sx = stds[0]; a.sy = stds[1]; a.sw = stds[2]; a.sh = stds[3];
  a.max_ratio = 4.135166556742356f;       // |log(16 / 1000)|: mmdet's wh_ratio_clip
  a.out = out; a.counts = counts;
  const size_t lds = (size_t)tot * 5 * sizeof(float);
  if (lds > 150 * 1024) return T3D_ERR_UNSUPPORTED;
  if (lds > 64 * 1024)
    (void)t3d_max_lds((const void*)ssd_decode_nms_kernel, (int)lds);
  T3D_LAUNCH(ssd_decode_nms_kernel, dim3(num_classes, B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
