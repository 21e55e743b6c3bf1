"""Epoch loop of torchdet3d/trainer/train.py:10-114 with the per-iteration body exposed as `train_step`.

Constructor arguments and `.train(epoch, is_last_epoch)` are the reference's (scripts/main.py:68-82,103); its int
field `train_step` (the global TensorBoard step, train.py:26) is accepted as the `train_step=` keyword and kept
as `global_step`, because `train_step` is the method the north-star asks for.  The body differs from the
reference only in where the work happens: one fused loss+metric launch instead of ~250 tiny kernels, metrics
read back in a single device-to-host copy instead of 6 `.item()` syncs."""
import datetime
import time

from ..evaluation.metrics import compute_accuracy, compute_average_distance
from ..utils import AverageMeter, put_on_device, save_snap


class Trainer:
    def __init__(self, model, train_loader, optimizer, scheduler, loss_manager, writer, max_epoch, log_path,
                 device='cuda', save_chkpt=True, debug=False, debug_steps=30, save_freq=10, print_freq=10,
                 train_step=0):
        self.model, self.train_loader, self.optimizer, self.scheduler = model, train_loader, optimizer, scheduler
        self.loss_manager, self.writer, self.max_epoch, self.log_path = loss_manager, writer, max_epoch, log_path
        self.device, self.save_chkpt, self.debug, self.debug_steps = device, save_chkpt, debug, debug_steps
        self.save_freq, self.print_freq, self.global_step = save_freq, print_freq, train_step

    def train_step(self, imgs, gt_kp, gt_cats, it=0):
        """One iteration (train.py:44-55) -> dict(loss, ADD, SADD, acc) of python floats."""
        imgs, gt_kp, gt_cats = put_on_device([imgs, gt_kp, gt_cats], self.device)
        pred_kp, pred_cats = self.model(imgs, gt_cats)
        loss = self.loss_manager.parse_losses(pred_kp, gt_kp, pred_cats, gt_cats, it)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        last = getattr(self.loss_manager, 'last', None)
        if last is not None:                 # the loss launch already reduced the metrics: one read-back
            o = last.tolist()
            return dict(loss=o[0], ADD=o[3], SADD=o[4], acc=o[5])
        ADD, SADD = compute_average_distance(pred_kp, gt_kp)
        return dict(loss=loss.item(), ADD=ADD, SADD=SADD, acc=compute_accuracy(pred_cats, gt_cats))

    def train(self, epoch, is_last_epoch):
        meters = {k: AverageMeter() for k in ('loss', 'ADD', 'SADD', 'acc', 'time')}
        self.model.train()
        self.num_iters = len(self.train_loader)
        start = time.time()
        for it, (imgs, gt_kp, gt_cats) in enumerate(self.train_loader):
            r = self.train_step(imgs, gt_kp, gt_cats, it)
            n = imgs.size(0)
            for k in ('loss', 'ADD', 'SADD', 'acc'):
                meters[k].update(r[k], n)
            if self.writer is not None:
                self.writer.add_scalar('Train/loss', r['loss'], global_step=self.global_step)
                self.writer.add_scalar('Train/ADD', meters['ADD'].avg, global_step=self.global_step)
                self.writer.add_scalar('Train/SADD', meters['SADD'].avg, global_step=self.global_step)
                self.writer.add_scalar('Train/ACC', meters['acc'].avg, global_step=self.global_step)
            self.global_step += 1
            meters['time'].update(time.time() - start)
            left = (self.num_iters - (it + 1)) + (self.max_epoch - (epoch + 1)) * self.num_iters
            if it % self.print_freq == 0 or it == self.num_iters - 1:
                print(f'epoch: [{epoch}/{self.max_epoch}][{it}/{self.num_iters}]\t'
                      f'time {meters["time"].val:.3f} ({meters["time"].avg:.3f})\t'
                      f'eta {datetime.timedelta(seconds=int(meters["time"].avg * left))}\t'
                      f'cls acc {meters["acc"].val:.3f} ({meters["acc"].avg:.3f})\t'
                      f'ADD {meters["ADD"].val:.4f} ({meters["ADD"].avg:.4f})\t'
                      f'SADD {meters["SADD"].val:.4f} ({meters["SADD"].avg:.4f})\t'
                      f'loss {meters["loss"].avg:.5f}\tlr {self.optimizer.param_groups[0]["lr"]:.6f}')
            start = time.time()
            if self.debug and it == self.debug_steps:
                break
        if self.save_chkpt and (epoch % self.save_freq == 0 or is_last_epoch) and not self.debug:
            save_snap(self.model, self.optimizer, self.scheduler, epoch, self.log_path)
        if self.scheduler is not None:
            self.scheduler.step()
        return {k: m.avg for k, m in meters.items()}
