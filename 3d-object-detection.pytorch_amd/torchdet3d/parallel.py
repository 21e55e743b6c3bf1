"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's single-process `torch.nn.DataParallel` (scripts/main.py:60-61): same
semantics -- per-replica BatchNorm statistics, gradients of the full-batch mean loss (equal shards, so
the average of the per-rank mean-loss gradients) -- without its per-iteration parameter broadcast,
scatter and gather.  The flat gradient buffer of the engine is reduced in a few large buckets, each
launched as soon as the backward has finished the parameters it covers (the backward walks the buffer
from its end to its start), so the exchange overlaps the remaining backward kernels.  Works on any
torch.distributed backend ('nccl' = RCCL on ROCm; 'gloo' in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


# ---- who am I: the host-side helpers of the one-process-per-GPU flow (an unchanged scripts/main.py under
# `python -m torch.distributed.run`; the reference's single process did all I/O itself, main.py:36-106) -----------------
def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def launch_rank():
    """Rank as far as it can be known before `build_model` has joined the process group (scripts/main.py:39 builds its
    Logger first): the launcher's RANK when torch.distributed.run started more than one process."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    try:
        return int(os.environ['RANK']) if int(os.environ.get('WORLD_SIZE', '1')) > 1 else 0
    except (KeyError, ValueError):
        return 0


def is_main():
    """Rank 0 writes checkpoints, TensorBoard scalars, the log file and the metric tables; the others compute."""
    return rank() == 0


def barrier():
    if world_size() > 1:
        dist.barrier()


def all_reduce_sums(values, device=None):
    """Element-wise SUM over the ranks of a short list of python floats (metric sums and counts) -> list of floats.
    One small collective; fp64 so that counts stay exact.  RCCL ('nccl') needs device tensors, gloo takes host ones."""
    if world_size() == 1:
        return [float(v) for v in values]
    on_gpu = dist.get_backend() == 'nccl'
    t = torch.tensor([float(v) for v in values], dtype=torch.float64,
                     device=(device if device is not None else torch.device('cuda', torch.cuda.current_device())) if on_gpu else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().tolist()


class GradSync:
    def __init__(self, flat_grad, min_bucket=1 << 20, group=None):
        """flat_grad: 1-D fp32 gradient buffer.  min_bucket: elements per bucket (4 MB default: RCCL over the
        fully connected xGMI mesh is latency-bound below ~1 MB per message)."""
        self.g, self.min_bucket, self.group = flat_grad, min_bucket, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # T3D_FORCE_SYNC=1: run the collectives even on one rank (single-GPU smoke test of the RCCL path)
        self.force = bool(os.environ.get('T3D_FORCE_SYNC')) and dist.is_initialized()
        self.hi = flat_grad.numel()
        self.works = []
        # backend 'gloo' with device tensors (two test ranks sharing one GPU -- RCCL refuses two ranks per device): the
        # collectives run on host copies; 'nccl' (= RCCL) takes the device buffers as they are
        self.staged = dist.is_initialized() and dist.get_backend(group) == 'gloo' and flat_grad.is_cuda

    def broadcast(self, tensors, src=0):
        """One-time parameter / buffer sync from rank `src` (DataParallel replicates from device 0)."""
        if self.world > 1 or self.force:
            for t in tensors:
                if self.staged and t.is_cuda:
                    h = t.cpu()
                    dist.broadcast(h, src, group=self.group)
                    t.copy_(h)
                else:
                    dist.broadcast(t, src, group=self.group)

    def start(self):
        self.hi = self.g.numel()
        self.works = []

    def ready(self, lo):
        """Gradients at offsets >= lo are final."""
        if self.world == 1 and not self.force:
            return
        if lo == 0 or self.hi - lo >= self.min_bucket:
            if self.hi > lo and self.staged:
                h = self.g[lo:self.hi].cpu()           # (synchronises the stream: test configuration only)
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                self.g[lo:self.hi].copy_(h)
            elif self.hi > lo:
                self.works.append(dist.all_reduce(self.g[lo:self.hi], op=dist.ReduceOp.SUM, group=self.group,
                                                  async_op=True))
            self.hi = lo

    def finish(self, scale=True):
        """Wait for the buckets and turn the sums into means (scale=False: the caller's optimizer kernel applies the
        1/world factor while it reads the gradient, `FusedAdamW.grad_scale`)."""
        if self.world == 1 and not self.force:
            return
        if self.hi > 0:
            self.ready(0)
        for w in self.works:
            w.wait()
        self.works = []
        if scale:
            self.g.mul_(1.0 / self.world)
