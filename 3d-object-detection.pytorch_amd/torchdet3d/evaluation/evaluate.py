"""Validation loop (torchdet3d/evaluation/evaluate.py:73-149): eval-mode forward with the ground-truth class
selecting the regression head (:92), per-class meters weighted by the whole batch size (:96-100, as the reference
does), TensorBoard scalars `Val/{ADD,SADD,ACC,IOU}`, a plain-text table in place of PrettyTable; and `visual_test`
(:31-72): per-sample forward + metrics on a few test samples (the jpg drawing needs cv2 + objectron.graphics, which
this image does not have: the predicted / true keypoints are saved as .npy next to where the jpgs would go)."""
import os.path as osp

import numpy as np
import torch

from ..parallel import all_reduce_sums, is_main, world_size
from ..utils import AverageMeter, put_on_device, mkdir_if_missing, OBJECTRON_CLASSES
from .metrics import compute_accuracy, compute_average_distance, compute_metrics_per_cls, enqueue_metrics_per_cls


class Evaluator:
    def __init__(self, model, val_loader, test_loader=None, cfg=None, writer=None, max_epoch=1, device='cuda',
                 num_classes=9, samples='random', num_samples=10, path_to_save_imgs='./testing_images', debug=False,
                 debug_steps=30):
        self.model, self.val_loader, self.test_loader, self.cfg, self.writer = model, val_loader, test_loader, cfg, writer
        self.max_epoch, self.device, self.debug, self.debug_steps = max_epoch, device, debug, debug_steps
        self.path_to_save_imgs, self.samples, self.num_samples = path_to_save_imgs, samples, num_samples
        self.num_classes = cfg.model.num_classes if cfg is not None and cfg.model.num_classes else num_classes

    @torch.no_grad()
    def visual_test(self):
        """evaluate.py:31-72 over the TEST loader's dataset (the reference rebuilds `Objectron(root, mode='test')`,
        which is the dataset behind `test_loader`, loader_builder.py:31-34): forward one sample at a time with the
        ground-truth class selecting the head, print ADD / SADD / accuracy, store the keypoints."""
        ds = getattr(self.test_loader, 'dataset', None)
        if not is_main():          # one process per GPU: rank 0 runs the few visual samples and writes their files
            return []
        if ds is None or len(ds) == 0:
            print('visual_test: no test dataset, nothing to do')
            return []
        mkdir_if_missing(self.path_to_save_imgs)
        if self.samples == 'random':
            indexes = np.random.choice(len(ds), min(self.num_samples, len(ds)), replace=False)
        else:
            assert isinstance(self.samples, list)
            indexes = self.samples
        self.model.eval()
        results = []
        for idx in indexes:
            item = ds[int(idx)]
            crop_cords = None
            if len(item) == 5:                     # test-mode Objectron item (objectron_main.py:93-94)
                _, img, gt_kp, gt_cat, crop_cords = item
            else:
                img, gt_kp, gt_cat = item
            img, gt_kp = put_on_device([torch.as_tensor(img), torch.as_tensor(gt_kp)], self.device)
            cat = torch.as_tensor(gt_cat).view(-1).to(self.device)
            pred_kp, pred_cat = self.model(torch.unsqueeze(img, 0), cat)
            ADD, SADD = compute_average_distance(pred_kp, torch.unsqueeze(gt_kp, 0))
            accuracy = compute_accuracy(pred_cat, cat)
            print(f"\nimage №{idx}.\nComputed metrics:\n"
                  f"ADD ---> {ADD}\n"
                  f"SADD ---> {SADD}\n"
                  f"classification accuracy ---> {accuracy}")
            pk, gk = pred_kp[0].detach().cpu().numpy(), gt_kp.detach().cpu().numpy()
            if crop_cords is not None:
                pk, gk = self.transform_kp(pk, crop_cords), self.transform_kp(gk.copy(), crop_cords)
            label = OBJECTRON_CLASSES[int(torch.argmax(pred_cat, dim=1))] if pred_cat.dtype.is_floating_point else None
            np.save(osp.join(self.path_to_save_imgs, f'tested_image_{idx}_predicted.npy'), pk)
            np.save(osp.join(self.path_to_save_imgs, f'tested_image_{idx}_true.npy'), gk)
            results.append(dict(idx=int(idx), ADD=ADD, SADD=SADD, accuracy=accuracy, label=label))
        return results

    @staticmethod
    def transform_kp(kp, crop_cords):
        """evaluate.py:141-149: crop-normalised keypoints -> pixel coordinates of the original frame."""
        x0, y0, x1, y1 = crop_cords
        kp[:, 0] = kp[:, 0] * (x1 - x0) + x0
        kp[:, 1] = kp[:, 1] * (y1 - y0) + y0
        return kp

    @torch.no_grad()
    def val_step(self, imgs, gt_kp, gt_cats, compute_iou=True):
        """Body of the loop (:90-95) -> (per_class_metrics, ADD, SADD, IOU, ACC)."""
        imgs, gt_kp, gt_cats = put_on_device([imgs, gt_kp, gt_cats], self.device)
        pred_kp, pred_cats = self.model(imgs, gt_cats)
        return compute_metrics_per_cls(pred_kp, gt_kp, pred_cats, gt_cats, compute_iou)

    @torch.no_grad()
    def val_enqueue(self, imgs, gt_kp, gt_cats, compute_iou=True):
        """`val_step` without its wait: the forward on the current stream, the batch's metric kernels (3-D IoU: one workgroup per
        sample, ~0.3 ms at B = 256) and its device -> host copy on a second stream behind it -> PendingMetrics.  `val` calls
        `.result()` on batch i after batch i + 1 is enqueued: the IoU kernel, the copy and the host's per-class sums run beside
        the next forward instead of between two."""
        imgs, gt_kp, gt_cats = put_on_device([imgs, gt_kp, gt_cats], self.device)
        pred_kp, pred_cats = self.model(imgs, gt_cats)
        if not pred_kp.is_cuda:
            raise RuntimeError('metrics run on the HIP path only (no CPU fallback)')
        side = getattr(self, '_metric_stream', None)
        if side is None:
            side = self._metric_stream = torch.cuda.Stream(device=pred_kp.device)
        side.wait_stream(torch.cuda.current_stream(pred_kp.device))
        with torch.cuda.stream(side):
            return enqueue_metrics_per_cls(pred_kp, gt_kp, pred_cats, gt_cats, compute_iou)

    @torch.no_grad()
    def val(self, epoch=None, compute_iou=True):
        meters = [AverageMeter() for _ in range(4)]                      # ADD SADD ACC IOU
        cls_meters = [[AverageMeter() for _ in range(4)] for _ in range(self.num_classes)]
        self.model.eval()
        if world_size() > 1:
            # every rank trains with its own BatchNorm running statistics (per-replica BatchNorm, scripts/main.py:60-61);
            # nn.DataParallel validates -- and save_snap writes -- with replica 0's: take rank 0's buffers on every rank first,
            # so that the all-reduced metrics below describe the model the checkpoint holds
            sync = getattr(self.model, 'grad_sync', None)
            if sync is not None and hasattr(self.model, 'net'):
                sync.broadcast(list(self.model.net.buffers.values()))
        def collect(pending):
            (per_cls, ADD, SADD, IOU, ACC), n = pending[0].result(), pending[1]
            for cl, a, s, i, c in per_cls:
                for m, v in zip(cls_meters[cl], (a, s, c, i)):
                    m.update(v, n)
            for m, v in zip(meters, (ADD, SADD, ACC, IOU)):
                m.update(v, n)

        pending = None                  # one batch in flight: batch i is read back after batch i + 1 is enqueued
        for it, (imgs, gt_kp, gt_cats) in enumerate(self.val_loader):
            cur = (self.val_enqueue(imgs, gt_kp, gt_cats, compute_iou), imgs.size(0))
            if pending is not None:
                collect(pending)
            pending = cur
            if self.debug and it == self.debug_steps:
                break
        if pending is not None:
            collect(pending)
        if world_size() > 1:
            # one process per GPU: every rank validated its own share of the samples (builders/loader_builder.py); the
            # reference's meters (evaluate.py:97-122) are weighted sums / counts, so the global value of each is
            # sum(rank sums) / sum(rank counts) -- ONE small all-reduce of the 8 + 8 * num_classes partial values
            allm = meters + [m for cm in cls_meters for m in cm]
            tot = all_reduce_sums([v for m in allm for v in (m.sum, m.count)])
            for j, m in enumerate(allm):
                m.sum, m.count = tot[2 * j], tot[2 * j + 1]
                m.avg = m.sum / m.count if m.count else 0.0
        res = dict(ADD=meters[0].avg, SADD=meters[1].avg, ACC=meters[2].avg, IOU=meters[3].avg)
        if not is_main():          # rank 0 writes the scalars and prints the table; every rank returns the same numbers
            return res
        if epoch is not None and self.writer is not None:
            self.writer.add_scalar('Val/ADD', meters[0].avg, global_step=epoch)
            self.writer.add_scalar('Val/SADD', meters[1].avg, global_step=epoch)
            self.writer.add_scalar('Val/ACC', meters[2].avg, global_step=epoch)
            if compute_iou:
                self.writer.add_scalar('Val/IOU', meters[3].avg, global_step=epoch)
        hdr = ['category name', 'ADD', 'SADD', 'accuracy'] + (['IOU'] if compute_iou else [])
        rows = [['Average metrics'] + [m.avg for m in meters[:3 + bool(compute_iou)]]]
        for c in range(self.num_classes):
            rows.append([OBJECTRON_CLASSES[c] if c < 9 else str(c)] + [m.avg for m in cls_meters[c][:3 + bool(compute_iou)]])
        print('\nComputed val metrics:' + (f'\nepoch: {epoch}' if epoch is not None else ''))
        print(' | '.join(f'{h:>16s}' for h in hdr))
        for r in rows:
            print(' | '.join([f'{r[0]:>16s}'] + [f'{v:16.4f}' for v in r[1:]]))
        return res

    def run_eval_pipe(self, visual_only=False):
        """evaluate.py:135-139."""
        print('.' * 10, 'Run evaluating protocol', '.' * 10)
        res = None
        if not visual_only:
            res = self.val(compute_iou=True)
        self.visual_test()
        return res
