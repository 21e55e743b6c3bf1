"""Second stage of the two-stage pipeline on the HIP path: batched keypoint regression over the detections of a frame.

Mirrors the reference's `torchdet3d/utils/ie_wrappers.py` API for the regression stage (`Regressor`, :123-158) so that
`scripts/demo.py:48-90` keeps its call sites -- `regressor.get_detections(prev_frame, detections)`, `regressor.transform_kp`
-- but instead of one OpenVINO inference per detection (crop on the host, `cv.resize`, `net.forward`, :128-133) the frame
goes to the GPU once and ALL its detections are cropped + resized by one kernel (`t3d_crop_resize_u8`), normalised inside the
stem's patch gather, and regressed in one batched forward with every head (`forward_to_onnx` semantics, builders/
model_builder.py:112-124); the head of the arg-max class is selected on the device (:135-139).

The detector stage (`Detector`, :70-120) runs an SSD trained in an external mmdetection fork and deployed as an OpenVINO IR
(configs/detection/mnv2_ssd_300_2_heads.py, README.md:56-57): there is no reference source for its arithmetic; `Detector`
here runs the same architecture built from that config on the product's kernels (models/ssd.py, parity unpinned) -- any other
detector that yields `(left, top, right, bottom, confidence, label)` tuples plugs into `Regressor` as well.
"""
import numpy as np
import torch

from .. import _native as N

__all__ = ['Regressor', 'Detector']


class Regressor:
    """HIP replacement of ie_wrappers.Regressor.  `model`: a `build_model(...)` result on the GPU (its eval-mode engine is
    used; `model.set_input_normalization` / cfg.data.normalization give the mean / std the exported IR carried as
    mean_values / scale_values, scripts/export.py:67-68).  `input_size` = (w, h) of the crops (cfg.data.resize)."""

    def __init__(self, model, input_size=(224, 224), max_detections=64):
        if not next(model.parameters()).is_cuda:
            raise RuntimeError('the HIP path needs the model on the GPU (no CPU fallback)')
        self.model = model
        self.device = next(model.parameters()).device
        self.w, self.h = int(input_size[0]), int(input_size[1])
        self._rects = torch.empty(max_detections, 4, dtype=torch.int32, device=self.device)
        self._rects_host = torch.empty(max_detections, 4, dtype=torch.int32).pin_memory()
        self._crops = torch.empty(max_detections, self.h, self.w, 3, dtype=torch.uint8, device=self.device)

    # ---- device-side API: no host synchronisation ------------------------------------------------------------------
    def crop_resize(self, frame, rects):
        """frame [H,W,3] uint8 (device), rects [n,4] int32 (device, x0,y0,x1,y1) -> crops [n,h,w,3] uint8 (a view into
        the wrapper's buffer, valid until the next call)."""
        n = int(rects.shape[0])
        assert frame.is_cuda and frame.dtype == torch.uint8 and frame.dim() == 3 and frame.shape[2] == 3 and frame.is_contiguous()
        assert rects.is_cuda and rects.dtype == torch.int32 and rects.is_contiguous()
        if n > self._crops.shape[0]:
            self._crops = torch.empty(n, self.h, self.w, 3, dtype=torch.uint8, device=self.device)
        N.call('t3d_crop_resize_u8', N.ptr(frame), N.ptr(rects), N.ptr(self._crops), n, int(frame.shape[0]), int(frame.shape[1]),
               self.h, self.w, N.stream())
        return self._crops[:n]

    @torch.no_grad()
    def regress(self, frame, rects):
        """-> (kp [n,9,2] fp32 of the arg-max class's head, in crop-normalised coordinates; labels [n] int64) on the device."""
        crops = self.crop_resize(frame, rects)
        was_training = self.model.training
        self.model.eval()
        try:
            kp_all, logits = self.model.forward_to_onnx(crops)          # [9,n,9,2], [n,C]
        finally:
            self.model.train(was_training)
        n = crops.shape[0]
        if self.model.num_classes > 1:
            labels = logits.argmax(1)
        else:
            labels = torch.zeros(n, dtype=torch.int64, device=self.device)
        kp = kp_all[labels, torch.arange(n, device=self.device)]
        return kp, labels

    # ---- the reference's host API ------------------------------------------------------------------------------------
    def get_detections(self, frame, detections):
        """Returns [(kp ndarray [1,9,2], label)] for all detections on `frame` (ndarray [H,W,3] uint8 or a device tensor),
        like ie_wrappers.py:128-142; detections with an empty crop are regressed on a black crop (the reference's
        cv.resize would raise on them)."""
        if len(detections) == 0:
            return []
        n = len(detections)
        if n > self._rects.shape[0]:
            self._rects = torch.empty(n, 4, dtype=torch.int32, device=self.device)
            self._rects_host = torch.empty(n, 4, dtype=torch.int32).pin_memory()
        self._rects_host[:n] = torch.as_tensor([[int(v) for v in d[:4]] for d in detections], dtype=torch.int32)
        self._rects[:n].copy_(self._rects_host[:n], non_blocking=True)
        if not torch.is_tensor(frame):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        frame = frame.to(self.device, non_blocking=True).contiguous()
        kp, labels = self.regress(frame, self._rects[:n])
        kp, labels = kp.cpu().numpy(), labels.cpu().numpy()
        return [(kp[i][None], int(labels[i])) for i in range(n)]

    @staticmethod
    def transform_kp(kp: np.array, crop_cords: tuple):
        """ie_wrappers.py:144-152: crop-normalised keypoints -> frame pixels (in place)."""
        x0, y0, x1, y1 = crop_cords
        crop_shape = (x1 - x0, y1 - y0)
        kp[:, 0] = kp[:, 0] * crop_shape[0]
        kp[:, 1] = kp[:, 1] * crop_shape[1]
        kp[:, 0] += x0
        kp[:, 1] += y0
        return kp

    @staticmethod
    def crop(frame, rect):
        """ie_wrappers.py:154-158 (host view; the batched path crops on the device)."""
        x0, y0, x1, y1 = rect
        return frame[y0:y1, x0:x1]


class Detector:
    """HIP replacement of ie_wrappers.Detector (:70-120): the SSD300-MobileNetV2 of configs/detection/mnv2_ssd_300_2_heads.py
    (models/ssd.py -- arithmetic per the published mmdet definitions, parity with the reference's externally trained
    detector unpinned).  Same call surface: `get_detections(frame)`, `run_async(frame)` / `wait_and_grab()`, `confidence`,
    `expand_ratio`; detections are `(left, top, right, bottom, confidence, label)` in frame pixels.  The frame is resized to
    300x300 on the GPU (`t3d_crop_resize_u8`, cv.resize's 8-bit bilinear arithmetic) and normalised inside the stem."""

    def __init__(self, model, conf=.6):
        self.model = model
        self.device = model.device
        self.confidence = conf
        self.expand_ratio = (1., 1.)
        self._pending = None

    def _enqueue(self, frame):
        from ..models.ssd import INPUT_SIZE
        if not torch.is_tensor(frame):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        frame = frame.to(self.device, non_blocking=True).contiguous()
        H, W = int(frame.shape[0]), int(frame.shape[1])
        rect = torch.tensor([[0, 0, W, H]], dtype=torch.int32, device=self.device)
        img = torch.empty(1, INPUT_SIZE, INPUT_SIZE, 3, dtype=torch.uint8, device=self.device)
        N.call('t3d_crop_resize_u8', N.ptr(frame), N.ptr(rect), N.ptr(img), 1, H, W, INPUT_SIZE, INPUT_SIZE, N.stream())
        return img, (H, W)

    def run_async(self, frame):
        self._pending = self._enqueue(frame)
        self.frame_shape = tuple(frame.shape)

    def wait_and_grab(self):
        img, shape = self._pending
        self._pending = None
        return self._decode_detections(self.model.detect(img)[0], shape)

    def get_detections(self, frame):
        """Returns all detections on frame"""
        img, shape = self._enqueue(frame)
        return self._decode_detections(self.model.detect(img)[0], shape)

    def get_detections_batch(self, frames):
        """BASELINE config 5 ("batched on 1 MI355X"): F equally sized frames per launch chain -- `frames` [F,H,W,3] uint8 (a
        device tensor, or anything torch.as_tensor takes) -> list of F detection lists, each as `get_detections` returns it.
        One resize launch for the whole stack (the stack is one tall frame to `t3d_crop_resize_u8`: frame f is the crop
        rows f*H .. (f+1)*H), one pass of the detector over the F x 300 x 300 batch, one decode + NMS launch, one read-back."""
        from ..models.ssd import INPUT_SIZE
        if not torch.is_tensor(frames):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        frames = frames.to(self.device, non_blocking=True).contiguous()
        F, H, W = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        key = (F, H, W)
        if getattr(self, '_batch_key', None) != key:
            ys = torch.arange(F, dtype=torch.int32) * H
            self._batch_rects = torch.stack([torch.zeros_like(ys), ys, torch.full_like(ys, W), ys + H], 1).contiguous().to(self.device)
            self._batch_imgs = torch.empty(F, INPUT_SIZE, INPUT_SIZE, 3, dtype=torch.uint8, device=self.device)
            self._batch_key = key
        N.call('t3d_crop_resize_u8', N.ptr(frames), N.ptr(self._batch_rects), N.ptr(self._batch_imgs), F, F * H, W, INPUT_SIZE,
               INPUT_SIZE, N.stream())
        return [self._decode_detections(rows, (H, W)) for rows in self.model.detect(self._batch_imgs)]

    def _decode_detections(self, rows, frame_shape):
        """ie_wrappers.py:94-120 on rows (x1, y1, x2, y2 normalised, confidence, label)."""
        detections = []
        for x1, y1, x2, y2, confidence, label in rows:
            if confidence > self.confidence:
                left = int(max(x1, 0) * frame_shape[1])
                top = int(max(y1, 0) * frame_shape[0])
                right = int(max(x2, 0) * frame_shape[1])
                bottom = int(max(y2, 0) * frame_shape[0])
                if self.expand_ratio != (1., 1.):
                    w, h = (right - left), (bottom - top)
                    dw, dh = w * (self.expand_ratio[0] - 1.) / 2, h * (self.expand_ratio[1] - 1.) / 2
                    left, right = max(int(left - dw), 0), int(right + dw)
                    top, bottom = max(int(top - dh), 0), int(bottom + dh)
                detections.append((left, top, right, bottom, float(confidence), int(label)))
        if len(detections) > 1:
            detections.sort(key=lambda x: x[1], reverse=True)          # (the reference sorts on element 1, :118-119)
        return detections
