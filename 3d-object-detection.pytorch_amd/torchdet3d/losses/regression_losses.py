"""Keypoint / class criterions and the multi-loss combiner of torchdet3d/losses/regression_losses.py
(DiagLoss :8-20, ADD_loss :22-26, WingLoss :28-49, LossManager :60-115 incl. the ALWA adaptive class weight)
and of torch.nn.{L1,MSE,SmoothL1,CrossEntropy}Loss as builders/loss_builder.py:13-20 instantiates them.

Every criterion is a thin host object over ONE device kernel (`t3d_loss_fwd_bwd`, csrc/loss.hip) that returns
the value and the analytic gradient in the same launch; `LossManager.parse_losses` folds all configured
criterions into a single launch.  The returned scalar is a regular autograd tensor (`.backward()`, `.item()`),
so the reference's training loop drives it unchanged."""
import torch

from .. import _native as N

_FIELDS = ('c_l1', 'c_mse', 'c_smoothl1', 'c_add', 'c_diag', 'c_wing', 'c_ce')


def _cfg(terms, lam_reg=1.0, lam_cls=1.0):
    """terms: [(criterion, coefficient)] -> LossCfg.  Two criterions of one kind add their coefficients only
    when their parameters agree (one kernel launch evaluates each kind once)."""
    c = N.LossCfg()
    c.smoothl1_beta, c.wing_w, c.wing_eps, c.lam_reg, c.lam_cls = 1.0, 1.0, 1.0, lam_reg, lam_cls
    seen = {}
    for crit, k in terms:
        f = crit.field
        if f in seen and seen[f] != crit.params():
            raise RuntimeError(f'two {type(crit).__name__} criterions with different parameters cannot share a launch')
        seen[f] = crit.params()
        setattr(c, f, getattr(c, f) + float(k))
        for name, v in crit.params().items():
            setattr(c, name, float(v))
    return c


class _Launch(torch.autograd.Function):
    """value = out[idx] of one kernel launch; backward scales the gradients the same launch produced."""

    @staticmethod
    def forward(ctx, kp, logits, gt_kp, cats, cfg, idx):
        if not kp.is_cuda:
            raise RuntimeError('losses run on the HIP path only: move predictions to the GPU (no CPU fallback)')
        B = kp.shape[0]
        dev = kp.device
        p = kp.detach().reshape(B, 18).float().contiguous()
        t = gt_kp.detach().to(dev).reshape(B, 18).float().contiguous()
        c = cats.detach().to(dev).long().contiguous() if cats is not None else torch.zeros(B, dtype=torch.int64, device=dev)
        lg = logits.detach().float().contiguous() if logits is not None else None
        ncls = lg.shape[1] if lg is not None else 1
        out = torch.zeros(16, device=dev)
        dkp = torch.empty(B, 18, device=dev)
        dlg = torch.empty(B, ncls, device=dev) if lg is not None else None
        N.call('t3d_loss_fwd_bwd', cfg, N.ptr(p), N.ptr(t), N.ptr(lg), N.ptr(c), N.ptr(out), N.ptr(dkp), N.ptr(dlg),
               B, ncls, N.stream())
        ctx.save_for_backward(dkp, dlg if dlg is not None else torch.empty(0, device=dev))
        ctx.kp_shape, ctx.has_lg = kp.shape, lg is not None
        ctx.mark_non_differentiable(out)
        return out[idx].clone(), out

    @staticmethod
    def backward(ctx, g, _gout):
        dkp, dlg = ctx.saved_tensors
        return (g * dkp).view(ctx.kp_shape), ((g * dlg) if ctx.has_lg else None), None, None, None, None


class _Criterion(torch.nn.Module):
    field, is_cls = None, False

    def params(self):
        return {}

    def forward(self, pred, target):
        cfg = _cfg([(self, 1.0)])
        if self.is_cls:
            B = pred.shape[0]
            dummy = torch.zeros(B, 18, device=pred.device)
            return _Launch.apply(dummy, pred, dummy, target, cfg, 0)[0]
        return _Launch.apply(pred, None, target, None, cfg, 0)[0]


class L1Loss(_Criterion):            # torch.nn.L1Loss(reduction='mean'), loss_builder.py:17-18
    field = 'c_l1'


class MSELoss(_Criterion):           # torch.nn.MSELoss(), loss_builder.py:19-20
    field = 'c_mse'


class SmoothL1Loss(_Criterion):      # torch.nn.SmoothL1Loss(beta=cfg.loss.smoothl1_beta), loss_builder.py:15-16
    field = 'c_smoothl1'

    def __init__(self, beta=1.0):
        super().__init__()
        self.beta = beta

    def params(self):
        return {'smoothl1_beta': self.beta}


class ADD_loss(_Criterion):          # regression_losses.py:22-26
    field = 'c_add'


class DiagLoss(_Criterion):          # regression_losses.py:8-20 (SmoothL1 beta=.4 between box diagonals)
    field = 'c_diag'


class WingLoss(_Criterion):          # regression_losses.py:28-49
    field = 'c_wing'

    def __init__(self, w=0.05, eps=2):
        super().__init__()
        self.w, self.eps = w, eps

    def params(self):
        return {'wing_w': self.w, 'wing_eps': self.eps}


class CrossEntropyLoss(_Criterion):  # torch.nn.CrossEntropyLoss(), loss_builder.py:13-14
    field, is_cls = 'c_ce', True


def compute_diag(x):
    """regression_losses.py:51-58 (host-side helper, torch ops)."""
    x0, y0 = x[:, :, 0].min(dim=1).values, x[:, :, 1].min(dim=1).values
    x1, y1 = x[:, :, 0].max(dim=1).values, x[:, :, 1].max(dim=1).values
    return torch.sqrt((x1 - x0) ** 2 + (y1 - y0) ** 2)


class LossManager:
    """regression_losses.py:60-115.  `alwa`: object with use / lam_cls / lam_reg / C / compute_std."""

    def __init__(self, criterions, coefficients, alwa):
        self.reg_criterions, self.class_criterions = criterions
        self.reg_coeffs, self.class_coeffs = coefficients
        assert len(self.reg_coeffs) == len(self.reg_criterions)
        assert len(self.class_coeffs) == len(self.class_criterions)
        assert self.reg_criterions
        self.use_alwa = bool(alwa.use)
        if self.use_alwa:
            assert self.class_criterions
            assert self.reg_coeffs[0] == self.class_coeffs[0] == 1.
        self.lam_cls = alwa.lam_cls if self.use_alwa else 1.
        self.lam_reg = alwa.lam_reg if self.use_alwa else 1.
        self.s_cls, self.s_reg = [], []
        self.C = alwa.C
        self.alwa_version = 'ver_1' if alwa.compute_std else 'ver_2'
        self.last = None      # device vector of the last launch: total, reg, cls, ADD, SADD, acc, ...
        self._fused = all(isinstance(c, _Criterion) for c in list(self.reg_criterions) + list(self.class_criterions))

    def loss_cfg(self):
        terms = list(zip(self.reg_criterions, self.reg_coeffs)) + list(zip(self.class_criterions, self.class_coeffs))
        return _cfg(terms, self.lam_reg if self.use_alwa else 1., self.lam_cls if self.use_alwa else 1.)

    def parse_losses(self, pred_kp, gt_kp, pred_cats, gt_cats, iter_):
        if self._fused:
            logits = pred_cats if (self.class_criterions and pred_cats is not None
                                   and pred_cats.dtype.is_floating_point) else None
            total, out = _Launch.apply(pred_kp, logits, gt_kp, gt_cats, self.loss_cfg(), 0)
            if self.use_alwa and self._alwa(out[1], out[2], iter_):
                # the reference combines with the lambda it has just updated (:111-115): one more launch on the
                # (rare, every C-th) update iteration
                total, out = _Launch.apply(pred_kp, logits, gt_kp, gt_cats, self.loss_cfg(), 0)
            self.last = out
            return total if self.class_criterions else total.reshape(1)   # zeros(1) + scalar (:88) has shape [1]
        # foreign callables (e.g. user criterions): plain composition, as the reference does
        class_loss = (sum(cr(pred_cats, gt_cats) * k for k, cr in zip(self.class_coeffs, self.class_criterions))
                      if self.class_criterions else torch.zeros(1, requires_grad=True))
        reg_loss = sum(cr(pred_kp, gt_kp) * k for k, cr in zip(self.reg_coeffs, self.reg_criterions))
        if not self.use_alwa:
            return reg_loss + class_loss
        self._alwa(reg_loss, class_loss, iter_)
        return self.lam_reg * reg_loss + self.lam_cls * class_loss

    def _alwa(self, reg_loss, class_loss, iter_):
        """:96-115.  Returns True when lam_cls changed."""
        self.s_cls.append(self.lam_cls * class_loss.detach())
        self.s_reg.append(self.lam_reg * reg_loss.detach())
        if iter_ % self.C == 0 and iter_ != 0:
            sc, sr = torch.stack([t.reshape(()) for t in self.s_cls]), torch.stack([t.reshape(()) for t in self.s_reg])
            cls, reg = sc.mean(), sr.mean()
            if self.alwa_version == 'ver_1':
                cls, reg = cls + sc.std(), reg + sr.std()
            self.s_cls.clear()
            self.s_reg.clear()
            if cls > reg:
                self.lam_cls = (1 - (cls - reg) / cls).item()
                return True
        return False
