"""Validation loop (torchdet3d/evaluation/evaluate.py:73-149): eval-mode forward with the ground-truth class
selecting the regression head (:92), per-class meters weighted by the whole batch size (:96-100, as the reference
does), TensorBoard scalars `Val/{ADD,SADD,ACC,IOU}`, a plain-text table in place of PrettyTable."""
import torch

from ..utils import AverageMeter, put_on_device, OBJECTRON_CLASSES
from .metrics import compute_metrics_per_cls


class Evaluator:
    def __init__(self, model, val_loader, test_loader=None, cfg=None, writer=None, max_epoch=1, device='cuda',
                 debug=False, debug_steps=30, path_to_save_imgs='./testing_images', num_classes=9):
        self.model, self.val_loader, self.test_loader, self.cfg, self.writer = model, val_loader, test_loader, cfg, writer
        self.max_epoch, self.device, self.debug, self.debug_steps = max_epoch, device, debug, debug_steps
        self.path_to_save_imgs = path_to_save_imgs
        self.num_classes = cfg.model.num_classes if cfg is not None and cfg.model.num_classes else num_classes

    @torch.no_grad()
    def val_step(self, imgs, gt_kp, gt_cats, compute_iou=True):
        """Body of the loop (:90-95) -> (per_class_metrics, ADD, SADD, IOU, ACC)."""
        imgs, gt_kp, gt_cats = put_on_device([imgs, gt_kp, gt_cats], self.device)
        pred_kp, pred_cats = self.model(imgs, gt_cats)
        return compute_metrics_per_cls(pred_kp, gt_kp, pred_cats, gt_cats, compute_iou)

    @torch.no_grad()
    def val(self, epoch=None, compute_iou=True):
        meters = [AverageMeter() for _ in range(4)]                      # ADD SADD ACC IOU
        cls_meters = [[AverageMeter() for _ in range(4)] for _ in range(self.num_classes)]
        self.model.eval()
        for it, (imgs, gt_kp, gt_cats) in enumerate(self.val_loader):
            per_cls, ADD, SADD, IOU, ACC = self.val_step(imgs, gt_kp, gt_cats, compute_iou)
            n = imgs.size(0)
            for cl, a, s, i, c in per_cls:
                for m, v in zip(cls_meters[cl], (a, s, c, i)):
                    m.update(v, n)
            for m, v in zip(meters, (ADD, SADD, ACC, IOU)):
                m.update(v, n)
            if self.debug and it == self.debug_steps:
                break
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
        return dict(ADD=meters[0].avg, SADD=meters[1].avg, ACC=meters[2].avg, IOU=meters[3].avg)

    def run_eval_pipe(self, visual_only=False):
        if not visual_only:
            return self.val(compute_iou=True)
        raise NotImplementedError('visual_test draws jpgs with cv2 (evaluate.py:31-72): out of scope')
