"""Oracle metrics (test infrastructure, see oracle/__init__.py).

Restates torchdet3d/evaluation/metrics.py: ADD / symmetric ADD (:10-29),
accuracy (:31-37), per-class aggregation (:39-68), 2-D based 3-D IoU (:70-89).
"""
import numpy as np
import scipy.spatial
import torch

from .geometry import lift_2d
from .box_iou import Box, IoU


@torch.no_grad()
def average_distance(pred_kp, gt_kp, num_keypoint=9, reduce_mean=True):
    # metrics.py:13-21: running minimum initialised with the same-index distance, strict '<'
    sadd = torch.zeros(pred_kp.shape[0])
    for i in range(num_keypoint):
        dist = torch.linalg.norm(pred_kp[:, i] - gt_kp[:, i], dim=1)
        for j in range(num_keypoint):
            d = torch.linalg.norm(pred_kp[:, i] - gt_kp[:, j], dim=1)
            dist = torch.where(d < dist, d, dist)
        sadd += dist
    per_kp = torch.linalg.norm(pred_kp - gt_kp, dim=2)
    if reduce_mean:                                           # :23-25
        return per_kp.mean().item(), (sadd.mean() / num_keypoint).item()
    return (per_kp.sum() / num_keypoint).item(), (sadd.sum() / num_keypoint).item()  # :27-28


@torch.no_grad()
def accuracy(pred_cats, gt_cats, reduce_mean=True):           # :31-37
    hit = (torch.argmax(pred_cats, dim=1) == gt_cats).float()
    return hit.mean().item() if reduce_mean else hit.sum().item()


def iou_2d_based(pred_kp, gt_kp, reduce_mean=True):           # :70-89
    p = pred_kp.detach().cpu().numpy()
    g = gt_kp.detach().cpu().numpy()
    total = 0.
    for i in range(p.shape[0]):
        k3 = lift_2d([p[i], g[i]], portrait=True)
        try:
            total += IoU(Box(k3[0]), Box(k3[1])).iou()
        except scipy.spatial.QhullError:                      # degenerate hull -> contributes 0
            pass
        except np.linalg.LinAlgError:
            pass
    if reduce_mean:
        return total / p.shape[0] if p.shape[0] else 0
    return total


@torch.no_grad()
def metrics_per_cls(pred_kp, gt_kp, pred_cats, gt_cats, compute_iou=True):   # :39-68
    out = []
    tA = tS = tI = tC = 0
    bs = pred_kp.shape[0]
    for cl in torch.unique(gt_cats):
        m = gt_cats == cl
        A, S = average_distance(pred_kp[m], gt_kp[m], reduce_mean=False)
        I = iou_2d_based(pred_kp[m], gt_kp[m], reduce_mean=False) if compute_iou else 0.
        C = accuracy(pred_cats[m], gt_cats[m], reduce_mean=False)
        n = int(m.sum())
        out.append((int(cl), A / n, S / n, I / n, C / n))
        tA, tS, tI, tC = tA + A, tS + S, tI + I, tC + C
    return out, tA / bs, tS / bs, tI / bs, tC / bs
