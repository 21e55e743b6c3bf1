"""Oracle of the SSD300-MobileNetV2 detector's forward + post-processing (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference holds only the mmdetection config of this model (configs/detection/mnv2_ssd_300_2_heads.py) and the OpenVINO
wrapper that consumes the exported IR (torchdet3d/utils/ie_wrappers.py:70-120); the implementing fork is external
(README.md:56-57).  **Parity unpinned**: what is restated here are the published mmdet definitions behind the config entries
-- SSDHead with depthwise heads (:15-37: Sequential(depthwise 3x3, BatchNorm, ReLU, 1x1 conv) per level and branch),
DeltaXYWHBBoxCoder.delta2bbox (:32-35; wh_ratio_clip 16/1000), softmax scores with the background class last,
multiclass_nms (:65-69: score_thr 0.02, per-class NMS at IoU 0.45, max 200 per image) -- on top of the oracle's own
MobileNetV2 (oracle/model.py, tapped at features.13 / features.17).  torch-CPU / numpy.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import model as OM
from .specs import arch

MAX_RATIO = abs(np.log(16 / 1000))


def head_outputs(sd, imgs, taps=(13, 17), num_classes=9):
    """imgs fp32 NCHW (normalised) -> per level (cls [B,H,W,A,(nc+1)], reg [B,H,W,A,4])."""
    t = {}
    bb = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    OM.extract_features(bb, arch('mobilenetv2'), imgs, False, t)
    outs = []
    for l, k in enumerate(taps):
        x = t[f'features.{k}']
        res = []
        for br, per in (('cls_convs', num_classes + 1), ('reg_convs', 4)):
            p = f'bbox_head.{br}.{l}'
            y = F.conv2d(x, sd[p + '.0.weight'], None, 1, 1, 1, x.shape[1])
            y = F.batch_norm(y, sd[p + '.1.running_mean'], sd[p + '.1.running_var'], sd[p + '.1.weight'], sd[p + '.1.bias'],
                             False, 0.1, 1e-5)
            y = F.conv2d(F.relu(y), sd[p + '.3.weight'], sd[p + '.3.bias'])
            B, C, H, W = y.shape
            res.append(y.permute(0, 2, 3, 1).reshape(B, H, W, C // per, per))
        outs.append(tuple(res))
    return outs


def delta2bbox(anchors, deltas, stds, max_shape):
    d = deltas * np.asarray(stds, np.float32)
    dw, dh = np.clip(d[:, 2], -MAX_RATIO, MAX_RATIO), np.clip(d[:, 3], -MAX_RATIO, MAX_RATIO)
    pw, ph = anchors[:, 2] - anchors[:, 0], anchors[:, 3] - anchors[:, 1]
    px, py = (anchors[:, 0] + anchors[:, 2]) * 0.5, (anchors[:, 1] + anchors[:, 3]) * 0.5
    gw, gh, gx, gy = pw * np.exp(dw), ph * np.exp(dh), px + pw * d[:, 0], py + ph * d[:, 1]
    b = np.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, max_shape[0])
    b[:, 1::2] = np.clip(b[:, 1::2], 0, max_shape[1])
    return b


def nms(boxes, scores, iou_thr):
    """Greedy NMS, highest score first (lowest index on ties); returns kept indices in order."""
    order = sorted(range(len(scores)), key=lambda i: (-scores[i], i))
    alive = np.ones(len(scores), bool)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep = []
    for i in order:
        if not alive[i]:
            continue
        keep.append(i)
        ix1, iy1 = np.maximum(boxes[i, 0], boxes[:, 0]), np.maximum(boxes[i, 1], boxes[:, 1])
        ix2, iy2 = np.minimum(boxes[i, 2], boxes[:, 2]), np.minimum(boxes[i, 3], boxes[:, 3])
        inter = np.clip(ix2 - ix1, 0, None) * np.clip(iy2 - iy1, 0, None)
        iou = inter / np.maximum(area[i] + area - inter, 1e-6)
        alive &= ~(iou > iou_thr)
        alive[i] = False
    return keep


def postprocess(outs, anchors, stds=(0.1, 0.1, 0.2, 0.2), num_classes=9, score_thr=0.02, iou_thr=0.45, max_per_img=200,
                input_size=300):
    """outs: per level (cls [B,H,W,A,nc+1], reg [B,H,W,A,4]) -> per image [n,6] (x1, y1, x2, y2 normalised, score, label)."""
    B = outs[0][0].shape[0]
    res = []
    for b in range(B):
        cls = np.concatenate([np.asarray(o[0][b]).reshape(-1, num_classes + 1) for o in outs]).astype(np.float32)
        reg = np.concatenate([np.asarray(o[1][b]).reshape(-1, 4) for o in outs]).astype(np.float32)
        e = np.exp(cls - cls.max(1, keepdims=True))
        prob = e / e.sum(1, keepdims=True)
        boxes = delta2bbox(anchors, reg, stds, (input_size, input_size))
        rows = []
        for c in range(num_classes):
            s = np.where(prob[:, c] > score_thr, prob[:, c], 0.0)
            idx = np.nonzero(s > 0)[0]
            for i in [idx[j] for j in nms(boxes[idx], s[idx], iou_thr)][:max_per_img]:
                rows.append([*boxes[i], s[i], c])
        rows = np.asarray(rows, np.float32).reshape(-1, 6)
        rows = rows[np.argsort(-rows[:, 4], kind='stable')[:max_per_img]]
        rows[:, :4] /= input_size
        res.append(rows)
    return res


def detect(sd, imgs, anchors, **kw):
    """-> per image [n,6], like models/ssd.py SSD300.detect."""
    return postprocess(head_outputs(sd, imgs, num_classes=kw.get('num_classes', 9)), anchors, **kw)
