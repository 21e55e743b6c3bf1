"""Name -> optimizer (torchdet3d/builders/optim_builder.py:3-19; 'adam' builds AdamW, :10-12).  The model exposes
ONE flat parameter (all weights, see models/engine.py), so every optimizer here is a single elementwise update over
~2.4-4.4 M floats instead of ~190 small tensors.  The default ('adam') is the hand-written HIP kernel
`t3d_adamw_step` (csrc/misc.hip) behind the torch.optim.Optimizer interface -- same hyper-parameters, same state-dict
layout (`step`, `exp_avg`, `exp_avg_sq`) and LR-scheduler behaviour as torch.optim.AdamW; the other names keep the
framework optimizers on the flat tensor."""
import torch

from .. import _native as N

AVAILABLE_OPTIMS = ['sgd', 'rmsprop', 'adam', 'adadelta']


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW (decoupled weight decay, no amsgrad / maximize) as one HIP launch per parameter tensor.
    `grad_scale` multiplies the gradient on load (1/world after a summed all-reduce)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.grad_scale = grad_scale
        self._watch = None            # device int64: the first step that met a non-finite gradient (t3d_set_grad_watch)

    def watch_word(self, device):
        if self._watch is None or self._watch.device != torch.device(device):
            self._watch = torch.full((1,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=device)
        return self._watch

    def first_nonfinite_step(self):
        """The 1-based optimizer step whose gradient was the first to hold an inf / NaN, or None (one 8-byte read-back and a
        wait: call it where the loop waits anyway)."""
        if self._watch is None:
            return None
        v = int(self._watch.item())
        return None if v == torch.iinfo(torch.int64).max else v

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.numel() % 4 or not p.is_contiguous():
                    raise RuntimeError('FusedAdamW runs on the HIP path only: contiguous fp32 device parameters with a '
                                       'multiple of 4 elements (the model\'s flat parameter)')
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1                           # a host int (load_state_dict normalises a tensor step)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                N.call('t3d_set_grad_watch', N.ptr(self.watch_word(p.device)))
                N.call('t3d_adamw_step', N.ptr(p), N.ptr(g), N.ptr(st['exp_avg']), N.ptr(st['exp_avg_sq']), p.numel(),
                       float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']),
                       st['step'], float(self.grad_scale), N.stream())
                N.call('t3d_set_grad_watch', None)            # (process-wide pointer: never left pointing at this optimizer's word)
                torch.autograd.graph.increment_version(p)     # written through a raw pointer: tell version-tracking users
        return loss


    # `step` is a host int here; torch.optim.AdamW keeps a float tensor.  Checkpoints travel in torch's form, so that a
    # snapshot written by either optimizer loads into the other (build_optimizer falls back to torch.optim.AdamW for
    # parameters that are not on the GPU), and a loaded device tensor never costs a sync per step.
    def state_dict(self):
        sd = super().state_dict()
        # (the packed state holds the optimizer's own per-parameter dicts: copy before rewriting `step`)
        sd['state'] = {k: ({**st, 'step': torch.tensor(float(st['step']))} if 'step' in st and not torch.is_tensor(st['step'])
                           else st) for k, st in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if 'step' in st:
                st['step'] = int(st['step'].item()) if torch.is_tensor(st['step']) else int(st['step'])


def build_optimizer(cfg, net):
    assert cfg.optim.name in AVAILABLE_OPTIMS
    params = list(net.parameters())
    if cfg.optim.name == 'adadelta':
        return torch.optim.Adadelta(params, lr=cfg.optim.lr, rho=cfg.optim.rho, weight_decay=cfg.optim.wd)
    if cfg.optim.name == 'adam':
        if all(p.is_cuda for p in params):
            return FusedAdamW(params, lr=cfg.optim.lr, betas=tuple(cfg.optim.betas), weight_decay=cfg.optim.wd)
        return torch.optim.AdamW(params, lr=cfg.optim.lr, betas=tuple(cfg.optim.betas), weight_decay=cfg.optim.wd)
    if cfg.optim.name == 'rmsprop':
        return torch.optim.RMSprop(params, lr=cfg.optim.lr, weight_decay=cfg.optim.wd, alpha=cfg.optim.alpha)
    return torch.optim.SGD(params, lr=cfg.optim.lr, weight_decay=cfg.optim.wd, momentum=cfg.optim.momentum,
                           nesterov=cfg.optim.nesterov)
