"""Training / validation metrics (torchdet3d/evaluation/metrics.py): ADD and symmetric ADD (:10-29), arg-max
accuracy (:31-37), per-class aggregation (:39-68) and the 2-D based 3-D IoU (:70-89).

ADD / SADD / accuracy come out of the single-launch wavefront-reduction kernel `t3d_loss_fwd_bwd`
(csrc/loss.hip) instead of the reference's 9x9 Python loop of tiny kernels; the 3-D IoU (lift_2d of both keypoint
sets + objectron box fit + box-box IoU per sample, a serial numpy / scipy loop with a device-to-host copy in the
reference) is ONE launch of `t3d_iou3d` (csrc/geometry.hip, fp64, one workgroup per sample) and one read-back."""
import torch

from .. import _native as N

_METRIC_CFG = None


def _reduce(pred_kp, gt_kp, pred_cats, gt_cats):
    """One launch -> out[3:9] = ADD, SADD, acc (means), then the reduce_mean=False forms."""
    global _METRIC_CFG
    if _METRIC_CFG is None:
        _METRIC_CFG = N.LossCfg()
        _METRIC_CFG.smoothl1_beta, _METRIC_CFG.wing_w, _METRIC_CFG.wing_eps = 0.2, 1.0, 1.0
    B = pred_kp.shape[0]
    dev = pred_kp.device
    if not pred_kp.is_cuda:
        raise RuntimeError('metrics run on the HIP path only (no CPU fallback)')
    p = pred_kp.detach().reshape(B, 18).float().contiguous()
    t = gt_kp.detach().reshape(B, 18).to(dev).float().contiguous()
    cats = (gt_cats if gt_cats is not None else torch.zeros(B, dtype=torch.int64, device=dev)).to(dev).long().contiguous()
    logits, ncls = None, 1
    if pred_cats is not None and pred_cats.dtype.is_floating_point and pred_cats.dim() == 2:
        logits = pred_cats.detach().float().contiguous()
        ncls = logits.shape[1]
    out = torch.zeros(16, device=dev)
    N.call('t3d_loss_fwd_bwd', _METRIC_CFG, N.ptr(p), N.ptr(t), N.ptr(logits), N.ptr(cats), N.ptr(out), None, None,
           B, ncls, N.stream())
    return out


@torch.no_grad()
def compute_average_distance(pred_kp, gt_kp, num_keypoint=9, reduce_mean=True):
    assert num_keypoint == 9
    if pred_kp.shape[0] == 0:
        return 0., 0.
    o = _reduce(pred_kp, gt_kp, None, None).tolist()
    return (o[3], o[4]) if reduce_mean else (o[6], o[7])


@torch.no_grad()
def compute_accuracy(pred_cats, gt_cats, reduce_mean=True):
    if pred_cats.dtype.is_floating_point and pred_cats.dim() == 2 and pred_cats.shape[1] <= 64 and pred_cats.is_cuda:
        B = pred_cats.shape[0]
        dummy = torch.zeros(B, 18, device=pred_cats.device)
        o = _reduce(dummy, dummy, pred_cats, gt_cats).tolist()
        return o[5] if reduce_mean else o[8]
    # width-1 integer "targets" (num_classes == 1, model_builder.py:144): arg-max is always 0
    hit = (torch.argmax(pred_cats, dim=1) == gt_cats).float()
    return hit.mean().item() if reduce_mean else hit.sum().item()


def iou3d_per_sample(pred_kp, gt_kp, portrait=True, return_lifted=False):
    """Per-sample 2-D based 3-D IoU [B] fp64 on the device (degenerate hulls / singular fits give 0, metrics.py:82-86);
    with return_lifted also the lifted boxes [B,2,9,3] fp64 (= lift_2d of both keypoint sets)."""
    B = pred_kp.shape[0]
    dev = pred_kp.device
    if not pred_kp.is_cuda:
        raise RuntimeError('the 3-D IoU runs on the HIP path only (no CPU fallback)')
    p = pred_kp.detach().reshape(B, 18).float().contiguous()
    g = gt_kp.detach().reshape(B, 18).to(dev).float().contiguous()
    iou = torch.empty(B, device=dev, dtype=torch.float64)
    lifted = torch.empty(B, 2, 9, 3, device=dev, dtype=torch.float64) if return_lifted else None
    if B:
        N.call('t3d_iou3d', N.ptr(p), N.ptr(g), B, int(bool(portrait)), None, N.ptr(iou), None, N.ptr(lifted), N.stream())
    return (iou, lifted) if return_lifted else iou


@torch.no_grad()
def compute_2d_based_iou(pred_kp, gt_kp, reduce_mean=True):
    """metrics.py:70-89."""
    B = pred_kp.shape[0]
    if B == 0:
        return 0
    total = sum(iou3d_per_sample(pred_kp, gt_kp).tolist())
    return total / B if reduce_mean else total


class PendingMetrics:
    """A batch's per-class metrics between their launches and their read-back: `result()` waits for the batch's ONE device ->
    host copy and takes the per-class sums on the host in fp64.  `Evaluator.val` enqueues batch i + 1 before it asks for
    batch i's result, so neither the 3-D IoU kernel nor the copy nor the host arithmetic sits between two forwards."""

    def __init__(self, bs, compute_iou, host=None, event=None, keep=()):
        self.bs, self.compute_iou, self.host, self.event, self.keep = bs, compute_iou, host, event, keep

    def result(self):
        bs, compute_iou = self.bs, self.compute_iou
        if bs == 0:
            return [], 0., 0., 0., 0.
        self.event.synchronize()
        host = self.host.numpy()
        self.keep = ()
        cl_of = host[:, 3].astype('int64')
        out = []
        tA = tS = tI = tC = 0.
        for cl in sorted(set(cl_of.tolist())):
            m = cl_of == cl
            A, S, C = float(host[m, 0].sum()), float(host[m, 1].sum()), float(host[m, 2].sum())
            I = float(host[m, 4].sum()) if compute_iou else 0.
            n = int(m.sum())
            out.append((int(cl), A / n, S / n, I / n, C / n))
            tA, tS, tI, tC = tA + A, tS + S, tI + I, tC + C
        return out, tA / bs, tS / bs, tI / bs, tC / bs


@torch.no_grad()
def enqueue_metrics_per_cls(pred_kp, gt_kp, pred_cats, gt_cats, compute_iou=True):
    """The launches of `compute_metrics_per_cls` on torch's current stream (per-sample ADD / SADD / hit summands, per-sample 3-D
    IoU) and the batch's one device -> host copy into pinned memory, without waiting for any of it -> PendingMetrics."""
    bs = pred_kp.shape[0]
    if bs == 0:
        return PendingMetrics(0, compute_iou)
    if not pred_kp.is_cuda:
        raise RuntimeError('metrics run on the HIP path only (no CPU fallback)')
    dev = pred_kp.device
    p = pred_kp.detach().reshape(bs, 18).float().contiguous()
    t = gt_kp.detach().reshape(bs, 18).to(dev).float().contiguous()
    cats = gt_cats.to(dev).long().contiguous()
    logits, ncls = None, 1
    if pred_cats is not None and pred_cats.dtype.is_floating_point and pred_cats.dim() == 2:
        logits = pred_cats.detach().float().contiguous()
        ncls = logits.shape[1]
    ps = torch.empty(bs, 3, device=dev)
    N.call('t3d_metrics_per_sample', N.ptr(p), N.ptr(t), N.ptr(logits), N.ptr(cats), N.ptr(ps), bs, ncls, N.stream())
    cols = [ps.double(), cats.double()[:, None]]
    if compute_iou:
        cols.append(iou3d_per_sample(pred_kp, gt_kp)[:, None])
    devm = torch.cat(cols, 1)
    host = torch.empty(devm.shape, dtype=devm.dtype, pin_memory=True)
    host.copy_(devm, non_blocking=True)                  # the batch's one device -> host copy
    ev = torch.cuda.Event()
    ev.record()
    return PendingMetrics(bs, compute_iou, host, ev, keep=(p, t, cats, logits, ps, devm, pred_kp, gt_kp, pred_cats, gt_cats))


@torch.no_grad()
def compute_metrics_per_cls(pred_kp, gt_kp, pred_cats, gt_cats, compute_iou=True):
    """metrics.py:39-68: per class present in the batch, sums / class count; totals / batch size.  Two launches (per-sample
    ADD / SADD / hit summands, per-sample 3-D IoU) and ONE read-back per batch; the per-class sums are taken on the host in
    fp64 (the reference: a python loop over the classes with two tiny-kernel chains and three syncs each)."""
    return enqueue_metrics_per_cls(pred_kp, gt_kp, pred_cats, gt_cats, compute_iou).result()
