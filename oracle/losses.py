"""Oracle losses + LossManager (test infrastructure, see oracle/__init__.py).

Restates torchdet3d/losses/regression_losses.py and
torchdet3d/builders/loss_builder.py in functional torch-CPU fp32.
"""
import math

import torch
import torch.nn.functional as F


def l1(p, t):                      # loss_builder.py:17-18  torch.nn.L1Loss('mean')
    return (p - t).abs().mean()


def mse(p, t):                     # loss_builder.py:19-20  torch.nn.MSELoss()
    return ((p - t) ** 2).mean()


def smoothl1(p, t, beta):          # loss_builder.py:15-16
    return F.smooth_l1_loss(p, t, reduction='mean', beta=beta)


def add_loss(p, t):                # regression_losses.py:22-26
    return torch.linalg.norm(p - t, dim=2).sum(dim=1).mean()


def compute_diag(x):               # regression_losses.py:51-58
    x0 = x[:, :, 0].min(dim=1).values
    y0 = x[:, :, 1].min(dim=1).values
    x1 = x[:, :, 0].max(dim=1).values
    y1 = x[:, :, 1].max(dim=1).values
    return torch.sqrt((x1 - x0) ** 2 + (y1 - y0) ** 2)


def diag_loss(p, t):               # regression_losses.py:8-20 (SmoothL1 beta=.4)
    return F.smooth_l1_loss(compute_diag(p), compute_diag(t), beta=.4)


def wing(p, t, w=0.05, eps=2):     # regression_losses.py:28-49
    # Literal restatement of the two sequential in-place masked updates
    # (:36-38): the second mask is evaluated on the already-updated values.
    wing_const = w - w * math.log(1. + w / eps)
    loss = (p - t).abs()
    m1 = loss < w
    loss = torch.where(m1, w * torch.log(1. + loss / eps), loss)
    m2 = loss >= w
    loss = torch.where(m2, loss - wing_const, loss)
    return loss.mean()


def cross_entropy(logits, cats):   # loss_builder.py:13-14
    return F.cross_entropy(logits, cats)


def build(names, smoothl1_beta=0.2, w=5.18, eps=1.):
    """loss_builder.py:7-28 -> (regress_criterions, class_criterions)."""
    avail = ['smoothl1', 'l1', 'cross_entropy', 'diag_loss', 'mse', 'add_loss', 'wing']
    reg, cls = [], []
    for n in names:
        assert n in avail
        if n == 'cross_entropy':
            cls.append(cross_entropy)
        elif n == 'smoothl1':
            reg.append(lambda p, t, b=smoothl1_beta: smoothl1(p, t, b))
        elif n == 'l1':
            reg.append(l1)
        elif n == 'mse':
            reg.append(mse)
        elif n == 'wing':
            reg.append(lambda p, t, w_=w, e_=eps: wing(p, t, w_, e_))
        elif n == 'add_loss':
            reg.append(add_loss)
        elif n == 'diag_loss':
            reg.append(diag_loss)
    return reg, cls


class LossManager:
    """regression_losses.py:60-115 (incl. the ALWA adaptive class weight)."""

    def __init__(self, criterions, coefficients, use_alwa=False, lam_cls=1., lam_reg=1.,
                 C=100, compute_std=True):
        self.reg_criterions, self.class_criterions = criterions
        self.reg_coeffs, self.class_coeffs = coefficients
        assert len(self.reg_coeffs) == len(self.reg_criterions)
        assert len(self.class_coeffs) == len(self.class_criterions)
        assert self.reg_criterions
        self.use_alwa = use_alwa
        if use_alwa:
            assert self.class_criterions
            assert self.reg_coeffs[0] == self.class_coeffs[0] == 1.
        self.lam_cls, self.lam_reg, self.C = lam_cls, lam_reg, C
        self.s_cls, self.s_reg = [], []
        self.with_std = compute_std

    def parse_losses(self, pred_kp, gt_kp, pred_cats, gt_cats, iter_):
        if self.class_criterions:
            cls_loss = sum(cr(pred_cats, gt_cats) * k
                           for k, cr in zip(self.class_coeffs, self.class_criterions))
        else:
            cls_loss = torch.zeros(1, requires_grad=True)      # :88 -> result has shape [1]
        reg_loss = sum(cr(pred_kp, gt_kp) * k
                       for k, cr in zip(self.reg_coeffs, self.reg_criterions))
        if not self.use_alwa:
            return reg_loss + cls_loss
        self.s_cls.append(self.lam_cls * cls_loss)
        self.s_reg.append(self.lam_reg * reg_loss)
        if iter_ % self.C == 0 and iter_ != 0:                 # :98
            sc, sr = torch.stack(self.s_cls), torch.stack(self.s_reg)
            cls, reg = sc.mean(), sr.mean()
            if self.with_std:                                  # 'ver_1' :104-106
                cls, reg = cls + sc.std(), reg + sr.std()
            self.s_cls.clear()
            self.s_reg.clear()
            if cls > reg:                                      # :111-113
                self.lam_cls = (1 - (cls - reg) / cls).item()
        return self.lam_reg * reg_loss + self.lam_cls * cls_loss
