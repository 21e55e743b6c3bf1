"""ORACLE (test infrastructure only -- never imported by the product path): CPU restatement of the two-stage pipeline's
input side and of the reference's per-detection regression loop.

  * `crop(frame, rect)`                 utils/ie_wrappers.py:154-158 (numpy slice; bounds clamped by numpy itself)
  * `resize_linear_u8(img, (w, h))`     cv.resize(img, (w, h)) as called at utils/ie_wrappers.py:18-21 -- INTER_LINEAR on 8-bit
                                        data.  OpenCV is NOT in this image, so this restates its published algorithm
                                        (modules/imgproc/src/resize.cpp: half-pixel centres, 11-bit fixed-point weights
                                        INTER_RESIZE_COEF_BITS = 11, HResizeLinear / VResizeLinear<uchar,int,short>):
                                        **parity with cv2 itself is unpinned**; `resize_linear_float` (textbook bilinear in
                                        fp64) bounds it to one grey level in the tests.
  * `regress_detections(...)`           utils/ie_wrappers.py:128-142 (crop -> forward -> argmax head) over a list of
                                        detections, one crop at a time like the reference.
"""
import numpy as np


def crop(frame, rect):
    x0, y0, x1, y1 = [int(v) for v in rect]
    x0, y0, x1, y1 = max(x0, 0), max(y0, 0), max(x1, 0), max(y1, 0)      # Detector clamps at 0 (ie_wrappers.py:101-104)
    return frame[y0:y1, x0:x1]


def _coef(dsize, ssize, column):
    d = np.arange(dsize, dtype=np.float64)
    scale = float(ssize) / float(dsize)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if column:
        lo = s < 0
        s[lo], f[lo] = 0, 0
        hi = s >= ssize - 1
        s[hi], f[hi] = ssize - 1, 0
        i0, i1 = s, np.minimum(s + 1, ssize - 1)
    else:
        i0, i1 = np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1)
    w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    w1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return i0, i1, w0, w1


def resize_linear_u8(img, size):
    """img [h,w,c] uint8 -> [oh,ow,c] uint8;  size = (ow, oh) like cv.resize."""
    ow, oh = size
    sh, sw = img.shape[:2]
    x0, x1, a0, a1 = _coef(ow, sw, True)
    y0, y1, b0, b1 = _coef(oh, sh, False)
    src = img.astype(np.int64)
    hor = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # [sh, ow, c]
    d0, d1 = hor[y0], hor[y1]
    v = (((b0[:, None, None] * (d0 >> 4)) >> 16) + ((b1[:, None, None] * (d1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_linear_float(img, size):
    """Textbook bilinear (half-pixel centres, replicated border) in fp64, unrounded."""
    ow, oh = size
    sh, sw = img.shape[:2]

    def ax(dsize, ssize):
        f = (np.arange(dsize) + 0.5) * (ssize / dsize) - 0.5
        s = np.floor(f)
        return np.clip(s, 0, ssize - 1).astype(int), np.clip(s + 1, 0, ssize - 1).astype(int), np.where((s < 0) | (s >= ssize - 1), 0.0, f - s)
    x0, x1, fx = ax(ow, sw)
    y0, y1, fy = ax(oh, sh)
    src = img.astype(np.float64)
    hor = src[:, x0] * (1 - fx)[None, :, None] + src[:, x1] * fx[None, :, None]
    # rows: both taps clip to the same row at the border, whatever the fraction
    f = (np.arange(oh) + 0.5) * (sh / oh) - 0.5
    fy = f - np.floor(f)
    return hor[y0] * (1 - fy)[:, None, None] + hor[y1] * fy[:, None, None]


def regress_detections(forward_all_heads, frame, detections, size, mean, std):
    """The reference's loop: for each detection crop, resize, normalise (the exported IR carries mean*255 / std*255,
    scripts/export.py:67-68), run the regressor with all heads, take the head of the argmax class
    (ie_wrappers.py:128-142).  forward_all_heads(x[1,3,h,w] fp32) -> (kp [9,1,9,2], logits [1,C])."""
    out = []
    mean, std = np.asarray(mean, np.float32), np.asarray(std, np.float32)
    for rect in detections:
        img = resize_linear_u8(crop(frame, rect[:4]), size)
        x = ((img.astype(np.float32) * np.float32(1.0 / 255.0) - mean) / std).transpose(2, 0, 1)[None]
        kp, logits = forward_all_heads(x)
        label = int(np.argmax(logits[0]))
        out.append((kp[label], label))
    return out


def objectron_crop(image, keypoints):
    """Objectron.crop (dataloaders/objectron_main.py:98-137): -> (keypoints shifted into the crop, crop, (x0, y0, x1, y1))."""
    real_h, real_w, _ = image.shape
    clamp = lambda x, lo, hi: min(max(x, lo), hi)
    kp = np.asarray(keypoints)
    clipped = np.empty_like(kp)
    clipped[:, 0] = [clamp(x, 3, real_w - 3) for x in kp[:, 0]]
    clipped[:, 1] = [clamp(y, 3, real_h - 3) for y in kp[:, 1]]
    x0 = clamp(min(clipped[:, 0]) - 10, 0, real_w)
    y0 = clamp(min(clipped[:, 1]) - 10, 0, real_h)
    x1 = clamp(max(clipped[:, 0]) + 10, 0, real_w)
    y1 = clamp(max(clipped[:, 1]) + 10, 0, real_h)
    x0, y0, x1, y1 = int(x0), int(y0), int(x1), int(y1)
    return clipped - np.asarray([x0, y0], clipped.dtype), image[y0:y1, x0:x1], (x0, y0, x1, y1)
