"""SSD300-MobileNetV2 detector of the two-stage pipeline (BASELINE config 5), inference, on the HIP path.

What the reference holds of it is the mmdetection CONFIG (`configs/detection/mnv2_ssd_300_2_heads.py`: backbone
`mobilenetv2_w1` tapped at its 96- and 320-channel maps :7-13, `SSDHead` with depthwise heads + ReLU :15-37, clustered
anchors of strides 16 / 32 :19-31, DeltaXYWH coder with stds (0.1, 0.1, 0.2, 0.2) :32-35, test_cfg score 0.02 / NMS 0.45 /
200 per image :65-69) and the OpenVINO wrapper that runs the exported IR (`torchdet3d/utils/ie_wrappers.py:70-120`); the
model code lives in an external mmdetection fork (README.md:56-57).  So the arithmetic here follows the PUBLISHED mmdet
definitions of those config entries and is restated in oracle/ssd.py -- **parity with the reference's detector is
unpinned** (no source, no weights); the layers themselves are the product's own kernels:
  backbone   models.engine.Net('mobilenetv2') in inference mode, tapped after features.13 (96 ch, stride 16) and
             features.17 (320 ch, stride 32)
  heads      per level and branch: depthwise 3x3 (t3d_dwconv_fwd) -> BatchNorm + ReLU applied on load by the 1x1 conv
             with bias (t3d_pwconv_fwd); class branch A*(classes+1) channels (background last), box branch A*4
  post       t3d_ssd_decode_nms: decode + softmax + per-class NMS in one launch; the overall top-`max_per_img` on the host
"""
import ctypes
import math

import numpy as np
import torch

from .. import _native as N
from .engine import Net

INPUT_SIZE = 300
CLASSES = ('bike', 'book', 'bottle', 'camera', 'cereal_box', 'chair', 'cup', 'laptop', 'shoe')     # config :4
STRIDES = (16, 32)
WIDTHS = ([0.2579684384230685, 0.4627705986569778, 0.34682129636083536, 0.641596163690939],
          [0.5420266488537757, 0.430022826081911, 0.7605568897973095, 0.6358004294180672, 0.5529565428117278,
           0.8008912664437589])                                                                     # config :21-25 (x input_size)
HEIGHTS = ([0.2270640055663951, 0.30064816327707244, 0.4627093933691148, 0.33801734483143625],
           [0.47856221526606557, 0.6557960498140745, 0.49101025166070583, 0.6256796503549162, 0.8331586024284066,
            0.7244268959927074])                                                                    # config :26-31
STDS = (0.1, 0.1, 0.2, 0.2)                                                                         # config :34
TAPS = (13, 17)                  # features.13 -> 96 channels @ stride 16, features.17 -> 320 channels @ stride 32
TAP_CHANNELS = (96, 320)
BN_EPS = 1e-5


def make_anchors(input_size=INPUT_SIZE):
    """[sum_l H_l*W_l*A_l, 4] (x1, y1, x2, y2), level-major, then pixel (row-major), then anchor: boxes of the clustered
    widths / heights centred on the cell centres stride*(i + 0.5)."""
    out = []
    for l, s in enumerate(STRIDES):
        fm = math.ceil(input_size / s)
        cy, cx = np.meshgrid((np.arange(fm) + 0.5) * s, (np.arange(fm) + 0.5) * s, indexing='ij')
        w, h = np.asarray(WIDTHS[l]) * input_size, np.asarray(HEIGHTS[l]) * input_size
        a = np.stack([cx[..., None] - w / 2, cy[..., None] - h / 2, cx[..., None] + w / 2, cy[..., None] + h / 2], -1)
        out.append(a.reshape(-1, 4))
    return np.concatenate(out).astype(np.float32)


def head_param_shapes(num_classes=len(CLASSES)):
    """State-dict entries of the SSD head (mmdet's SSDHead with depthwise_heads: Sequential(dw conv, BN, ReLU, 1x1 conv))."""
    out = {}
    for l, C in enumerate(TAP_CHANNELS):
        A = len(WIDTHS[l])
        for br, n in (('cls_convs', A * (num_classes + 1)), ('reg_convs', A * 4)):
            p = f'bbox_head.{br}.{l}'
            out[p + '.0.weight'] = (C, 1, 3, 3)
            for k, shp in (('weight', (C,)), ('bias', (C,)), ('running_mean', (C,)), ('running_var', (C,))):
                out[f'{p}.1.{k}'] = shp
            out[p + '.3.weight'] = (n, C, 1, 1)
            out[p + '.3.bias'] = (n,)
    return out


class SSD300:
    """`detect(frames_u8)` -> per image an [n, 6] array (x1, y1, x2, y2 normalised to [0, 1], score, label)."""

    def __init__(self, device='cuda', dtype=torch.bfloat16, num_classes=len(CLASSES), score_thr=0.02, iou_thr=0.45,
                 max_per_img=200, seed=0):
        self.device, self.dtype, self.nc = torch.device(device), dtype, num_classes
        self.dt = N.F32 if dtype == torch.float32 else N.BF16
        self.score_thr, self.iou_thr, self.max_per_img = score_thr, iou_thr, max_per_img
        self.backbone = Net('mobilenetv2', 9, device, dtype)        # (the regression heads of the engine stay unused)
        self.backbone.reset_parameters(seed=seed)
        self.anchors = torch.from_numpy(make_anchors()).to(self.device)
        g = torch.Generator().manual_seed(seed + 1)
        self.p = {}
        for k, shp in head_param_shapes(num_classes).items():
            if k.endswith('running_var') or k.endswith('.1.weight'):
                v = torch.ones(shp)
            elif k.endswith('.0.weight') or k.endswith('.3.weight'):
                fan = shp[1] * shp[2] * shp[3]
                v = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan)
            else:
                v = torch.zeros(shp)
            self.p[k] = v.to(self.device)
        self._packed = None
        self._stds = (ctypes.c_float * 4)(*STDS)

    def state_dict(self):
        sd = {'backbone.' + k: v for k, v in self.backbone.state_dict().items()
              if not (k.startswith('regressors') or k.startswith('cls_fc'))}
        sd.update({k: v.detach().clone() for k, v in self.p.items()})
        return sd

    def load_state_dict(self, sd):
        bb = self.backbone.state_dict()
        bb.update({k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')})
        self.backbone.load_state_dict(bb)
        for k in self.p:
            self.p[k].copy_(torch.as_tensor(sd[k]).to(self.device).view(self.p[k].shape))
        self._packed = None

    def _pack(self):
        """1x1 weights in the storage dtype, output channels padded to a multiple of 8 (zero rows); BatchNorm folded to the
        per-channel affine the 1x1 conv applies on load."""
        packed = []
        for l, C in enumerate(TAP_CHANNELS):
            lv = {}
            for br in ('cls_convs', 'reg_convs'):
                p = f'bbox_head.{br}.{l}'
                w = self.p[p + '.3.weight'].view(-1, C)
                n = w.shape[0]
                npad = (n + 7) // 8 * 8
                wp = torch.zeros(npad, C, device=self.device)
                wp[:n] = w
                bp = torch.zeros(npad, device=self.device)
                bp[:n] = self.p[p + '.3.bias']
                scale, shift = torch.empty(C, device=self.device), torch.empty(C, device=self.device)
                N.call('t3d_bn_eval_affine', C, N.ptr(self.p[p + '.1.weight']), N.ptr(self.p[p + '.1.bias']),
                       N.ptr(self.p[p + '.1.running_mean']), N.ptr(self.p[p + '.1.running_var']), BN_EPS, N.ptr(scale),
                       N.ptr(shift), N.stream())
                lv[br] = dict(wdw=self.p[p + '.0.weight'].view(C, 9).contiguous(), w=wp.to(self.dtype).contiguous(), b=bp,
                              pro=N.prologue(scale, shift, None, 'relu', False), n=npad)
            packed.append(lv)
        self._packed = packed

    @torch.no_grad()
    def head_outputs(self, imgs):
        """imgs: uint8 NHWC [B,300,300,3] (normalised in the stem) or fp32 NCHW -> per level (cls [B*HW, n_cls], reg
        [B*HW, n_reg], HW)."""
        if self._packed is None:
            self._pack()
        taps = self.backbone.forward_taps(imgs, TAPS)
        st, outs = N.stream(), []
        for l, k in enumerate(TAPS):
            t, B, H, W, C = taps[k]
            lv, res = self._packed[l], []
            for br in ('cls_convs', 'reg_convs'):
                h = lv[br]
                y = torch.empty(B * H * W, C, device=self.device, dtype=self.dtype)
                N.call('t3d_dwconv_fwd', self.dt, N.ptr(t), None, N.ptr(h['wdw']), N.ptr(y), None, None, B, H, W, C, 3, 1, st)
                o = torch.empty(B * H * W, h['n'], device=self.device, dtype=self.dtype)
                N.call('t3d_pwconv_fwd', self.dt, N.ptr(y), h['pro'], N.ptr(h['w']), N.ptr(h['b']), N.ptr(o), None,
                       B * H * W, H * W, C, h['n'], st)
                res.append(o)
            outs.append((res[0], res[1], H * W))
        return outs

    @torch.no_grad()
    def detect(self, imgs):
        outs = self.head_outputs(imgs)
        B = imgs.shape[0]
        nl = len(outs)
        P, I = ctypes.c_void_p * nl, ctypes.c_int * nl
        cls, reg = P(*[o[0].data_ptr() for o in outs]), P(*[o[1].data_ptr() for o in outs])
        hw, na = I(*[o[2] for o in outs]), I(*[len(WIDTHS[l]) for l in range(nl)])
        cs, rs = I(*[o[0].shape[1] for o in outs]), I(*[o[1].shape[1] for o in outs])
        K = self.max_per_img
        out = torch.zeros(B, self.nc, K, 6, device=self.device)
        cnt = torch.zeros(B, self.nc, dtype=torch.int32, device=self.device)
        N.call('t3d_ssd_decode_nms', self.dt, nl, cls, reg, hw, na, cs, rs, N.ptr(self.anchors), B, self.nc,
               float(self.score_thr), float(self.iou_thr), K, float(INPUT_SIZE), float(INPUT_SIZE), self._stds,
               N.ptr(out), N.ptr(cnt), N.stream())
        out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
        res = []
        for b in range(B):
            rows = np.concatenate([out[b, c, :cnt[b, c]] for c in range(self.nc)]) if cnt[b].sum() else np.zeros((0, 6), np.float32)
            # mmdet multiclass_nms: the best `max_per_img` over all classes, by score (stable: class, then NMS order)
            order = np.argsort(-rows[:, 4], kind='stable')[:self.max_per_img]
            rows = rows[order]
            rows[:, :4] /= INPUT_SIZE
            res.append(rows)
        return res
