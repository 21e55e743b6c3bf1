"""The input side of the reference's dataset, with the pixels on the GPU.

`Objectron.__getitem__` (`torchdet3d/dataloaders/objectron_main.py:51-96`) reads a frame, cuts the object out of it around
its nine annotated keypoints (`crop`, :98-126: keypoints clipped 3 px inside the frame, the box padded by 10 px and clamped
to the frame), resizes the crop (albumentations `Resize` -> cv.resize) and normalises the keypoints by the final image size
(`utils/transforms.py:112-114`, `utils/utils.py:231-240`).  Here the box arithmetic stays on the host (nine points per
object) and crop + resize run as one launch per frame (`t3d_crop_resize_u8`); the crops stay uint8 NHWC and go straight into
`model(x, cats)` (normalised inside the stem).  Augmentations (flip / colour / blur, `builders/loader_builder.py`) are CPU
data preparation and not rebuilt.
"""
import numpy as np
import torch

from .. import _native as N

__all__ = ['FrameCropper', 'crop_cords_from_keypoints']


def _clamp(x, lo, hi):
    return min(max(x, lo), hi)


def crop_cords_from_keypoints(keypoints, w, h):
    """objectron_main.py:98-137: -> (clipped keypoints [9,2], (x0, y0, x1, y1)).  `keypoints`: unnormalised [9,2] (x, y)."""
    kp = np.asarray(keypoints)
    clipped = np.empty_like(kp)
    clipped[:, 0] = [_clamp(x, 3, w - 3) for x in kp[:, 0]]            # clip_bb (:128-137)
    clipped[:, 1] = [_clamp(y, 3, h - 3) for y in kp[:, 1]]
    x0 = _clamp(min(clipped[:, 0]) - 10, 0, w)
    y0 = _clamp(min(clipped[:, 1]) - 10, 0, h)
    x1 = _clamp(max(clipped[:, 0]) + 10, 0, w)
    y1 = _clamp(max(clipped[:, 1]) + 10, 0, h)
    return clipped, (x0, y0, x1, y1)


class FrameCropper:
    """frames + per-object keypoints -> (uint8 crops [n, h, w, 3] on the device, gt keypoints [n, 9, 2] normalised to the
    crop, crop boxes) -- the tuple `Trainer.train_step(imgs, gt_kp, gt_cats, it)` takes, without the pixels ever being
    cropped or resized on the host."""

    def __init__(self, size=(224, 224), device='cuda', max_objects=256):
        self.w, self.h = int(size[0]), int(size[1])
        self.device = torch.device(device)
        # Pinned staging for the crop boxes.  The upload is asynchronous, so a staging buffer may only be rewritten once the
        # copy that read it has run: NBUF buffers in rotation, each guarded by the event recorded after its last upload
        # (a single reused buffer let frame k be cropped with frame k+1's boxes whenever the stream was running behind).
        self._max = int(max_objects)
        self._slots = [[torch.empty(self._max, 4, dtype=torch.int32).pin_memory(), None] for _ in range(self.NBUF)]
        self._turn = 0

    NBUF = 3

    def _staging(self, n):
        slot = self._slots[self._turn]
        self._turn = (self._turn + 1) % self.NBUF
        if slot[1] is not None:
            slot[1].synchronize()                 # the upload that last used this buffer has completed
        if n > slot[0].shape[0]:
            slot[0] = torch.empty(n, 4, dtype=torch.int32).pin_memory()
        return slot

    def __call__(self, frame, keypoints_per_object):
        """frame: [H, W, 3] uint8 (ndarray or tensor, host or device); keypoints_per_object: list of [9,2] pixel keypoints."""
        if not torch.is_tensor(frame):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        frame = frame.to(self.device, non_blocking=True).contiguous()
        H, W = int(frame.shape[0]), int(frame.shape[1])
        n = len(keypoints_per_object)
        slot = self._staging(n)
        host = slot[0]
        kps, boxes = [], []
        for i, kp in enumerate(keypoints_per_object):
            clipped, (x0, y0, x1, y1) = crop_cords_from_keypoints(kp, W, H)
            x0, y0, x1, y1 = int(x0), int(y0), int(x1), int(y1)                  # A.Crop takes integer pixel bounds
            boxes.append((x0, y0, x1, y1))
            host[i, 0], host[i, 1], host[i, 2], host[i, 3] = x0, y0, x1, y1
            # A.Crop shifts the keypoints, Resize scales them, ToTensor divides by the final size: net (kp - origin) / crop size
            kps.append((np.asarray(clipped, np.float32) - np.float32([x0, y0])) / np.float32([max(x1 - x0, 1), max(y1 - y0, 1)]))
        rects = host[:n].to(self.device, non_blocking=True)
        if self.device.type == 'cuda':
            slot[1] = torch.cuda.Event()
            slot[1].record(torch.cuda.current_stream(self.device))
        crops = torch.empty(n, self.h, self.w, 3, dtype=torch.uint8, device=self.device)
        N.call('t3d_crop_resize_u8', N.ptr(frame), N.ptr(rects), N.ptr(crops), n, H, W, self.h, self.w, N.stream())
        return crops, torch.from_numpy(np.stack(kps)).to(self.device), boxes
