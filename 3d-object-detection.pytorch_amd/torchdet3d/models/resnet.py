"""ResNet-50 backbone on the HIP path (BASELINE config 4: "ResNet-50 backbone via torchdet3d.builders").

The reference has no ResNet (`torchdet3d/builders/model_builder.py:14-17`): the architecture is the
standard torchvision ResNet-50 (v1.5, stride on the 3x3 conv) wrapped the way the reference wraps its timm / efficientnet
backbones (`model_builder.py:73-151`: global pool -> 9 per-class heads + class head, 2048 features) -- **parity unpinned**,
checked against `oracle/resnet.py`.  It reuses the regression path's machinery (`models/engine.py`): flat parameter /
gradient buffers, raw tensors + BatchNorm sums from the producing kernel, finalize -> affine applied by the consumer, the
1x1 GEMM kernels for every contraction, weight gradients on the second stream.  What it adds (csrc/resnet.hip):
  dense 3x3 / 7x7 conv = patch gather (`t3d_im2col`, producer's BatchNorm + ReLU applied on load) + the 1x1 GEMM kernels
      on the [M, k*k*C] patch matrix; backward = GEMM data gradient into the patch matrix + `t3d_col2im_bwd`
  max-pool (`t3d_maxpool_fwd / _bwd`), bottleneck tail `z = relu(BN3(y3) + shortcut)` (`t3d_res_relu_fwd / _bwd`),
  stride-2 shortcut sampling (`t3d_subsample`).
First version: correct, not tuned (the patch matrices cost 9x the activation traffic; an implicit-GEMM 3x3 kernel is the
obvious next step for this model).
"""
import os

import torch

from .. import _native as N
from .engine import Net, _Src

# 3x3 layers as gather-form implicit GEMMs (csrc/conv3x3.hip): OFF by default -- measured slower than the patch-matrix path at
# config 4's shape (19.6 against 16.7 ms per step, DESIGN.md finding 41: the operand's BatchNorm transform is paid once per tap
# inside the GEMM instead of once in the gather); T3D_IMPLICIT3=1 switches them on
IMPLICIT3 = os.environ.get('T3D_IMPLICIT3', '0') != '0'


class ResNetEngine(Net):
    # ------------------------------------------------------------------ weights
    def _pack_extra(self, st):
        """k x k conv weights -> [N, Kp] patch-column order (storage dtype) + transposed copy for the data gradient."""
        if getattr(self, '_conv_pack', None) is None:
            self._conv_pack, self._conv3 = [], {}
            for k, (s, kind) in self.shapes.items():
                if kind == 'param' and len(s) == 4 and s[2] > 1:
                    n, c, kk = s[0], s[1], s[2]
                    kp = (kk * kk * c + 31) // 32 * 32
                    w32 = self._buf('wc32:' + k, (n, kp), torch.float32)
                    self.w[k] = w32 if self.dt == N.F32 else self._buf('wc:' + k, (n, kp))
                    self.wt[k] = self._buf('wct:' + k, (kp, n))
                    # fragment-order copies for the deep-contraction kernel (engine.py: `_frag`, `_wsel`)
                    fr = frt = None
                    if self.dt in (N.BF16, N.F16) and hasattr(self, '_frag') and self._frag is not None:
                        lib = N.lib()
                        if lib.t3d_pwconv_wants_frag(kp, n):
                            fr = self._buf('wcf:' + k, (lib.t3d_pwconv_frag_bytes(n, kp) // 2,), zero=True)
                            self._frag[self.w[k].data_ptr()] = fr
                        if lib.t3d_pwconv_wants_frag(n, kp):
                            frt = self._buf('wctf:' + k, (lib.t3d_pwconv_frag_bytes(kp, n) // 2,), zero=True)
                            self._frag[self.wt[k].data_ptr()] = frt
                    # implicit-GEMM 3x3 layers (csrc/conv3x3.hip, bf16 storage): the forward streams the fragment-order copy of
                    # the [N][9C] matrix whatever its shape, the data gradient that of the [C][9N] matrix
                    if kk == 3 and self._implicit3(c, n):
                        lib = N.lib()
                        if fr is None:
                            fr = self._buf('wcf:' + k, (lib.t3d_pwconv_frag_bytes(n, kp) // 2,), zero=True)
                        wd = self._buf('wcd:' + k, (c, 9 * n))
                        wdf = self._buf('wcdf:' + k, (lib.t3d_pwconv_frag_bytes(c, 9 * n) // 2,), zero=True)
                        self._conv3[k] = (fr, wd, wdf)
                    self._conv_pack.append((k, n, c, kk, kp, w32, fr, frt))
            # storage-dtype copy, transposed copy and fragment-order copies of every patch-column matrix: ONE launch (round 6: 67 ->
            # 18 launches per step; the 1x1 weights' table in engine.py `_pack` has the same rows)
            rows = [[w32.data_ptr(), 0 if self.dt == N.F32 else self.w[k].data_ptr(), self.wt[k].data_ptr(), n, kp,
                     0 if fr is None else fr.data_ptr(), 0 if frt is None else frt.data_ptr()]
                    for k, n, c, kk, kp, w32, fr, frt in self._conv_pack]
            self._conv_pack_desc = torch.tensor(rows, dtype=torch.int64, device=self.device) if rows else None
        for k, n, c, kk, kp, w32, fr, frt in self._conv_pack:
            N.call('t3d_pack_conv_weight', N.F32, N.ptr(self.p[k]), N.ptr(w32), n, c, kk, kp, st)
        if self._conv_pack_desc is not None:
            N.call('t3d_pack_weights_batched', self.dt, N.ptr(self._conv_pack_desc), self._conv_pack_desc.shape[0], st)
        for k, n, c, kk, kp, w32, fr, frt in self._conv_pack:
            if k in self._conv3:
                _, wd, wdf = self._conv3[k]
                N.call('t3d_pack_conv3x3_dgrad_weight', N.ptr(self.p[k]), N.ptr(wd), n, c, st)
                N.call('t3d_pwconv_pack_frag', N.ptr(wd), N.ptr(wdf), c, 9 * n, st)

    def _implicit3(self, c, n):
        """Dense 3x3 layer as an implicit GEMM (csrc/conv3x3.hip)?  bf16 storage, power-of-two channel counts >= 32; fp32 storage
        (the parity mode) keeps the patch-matrix path."""
        return IMPLICIT3 and self.dt == N.BF16 and c >= 64 and n >= 64 and (c & (c - 1)) == 0 and n % 8 == 0

    def _kp(self, key):
        return self.w[key].shape[1]

    # ------------------------------------------------------------------ helpers
    def _bnf(self, bn, count, act):
        """Sums complete -> finalize now (standalone launch) -> the consumer's prologue."""
        pro = self._bn_fwd(bn, count, act)
        self._settle_f(bn)
        return pro

    def _bnb(self, bn):
        bb = self._bn_bwd(bn)
        self._settle_b(bn)
        return bb

    def _pw(self, x, pro, w, y, bn, M, HW, K, Nn):
        wd, wp = self._wsel(w, pro is None or not pro.se)
        N.call('t3d_pwconv_fwd', wd, N.ptr(x), pro, wp, None, N.ptr(y), self._st(bn), M, HW, K, Nn, N.stream(),
               nbytes=M * (K + Nn) * self.esz)

    # ------------------------------------------------------------------ forward
    def _features(self, imgs, train):
        a, st, dt = self.arch, N.stream(), self.dt
        assert imgs.is_cuda and imgs.dim() == 4 and imgs.dtype == torch.float32 and imgs.shape[1] == 3, \
            'resnet50 takes the normalised fp32 NCHW crops of the input contract'
        imgs = imgs.contiguous()
        self.training = bool(train)
        self._pack()
        B, _, H, W = imgs.shape
        if train:
            self._zero(self._statbuf)
        else:
            self._eval_affines()
        sv = dict(B=B, imgs=imgs, blocks=[])
        # ---- stem: 7x7 / stride 2 conv (patch gather + GEMM) -> BN -> ReLU -> 3x3 / stride 2 max-pool
        H1, W1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        M1 = B * H1 * W1
        kp0 = self._kp('conv1.weight')
        col0 = self._buf('col:stem', (M1, kp0))
        N.call('t3d_im2col_nchw', dt, N.ptr(imgs), N.ptr(col0), B, H, W, 3, 7, 2, 3, kp0, st)
        bn0 = self.bns['bn1']
        y0 = self._buf('y:stem', (M1, 64))
        self._pw(col0, None, self.w['conv1.weight'], y0, bn0, M1, H1 * W1, kp0, 64)
        pro0 = self._bnf(bn0, M1, 'relu')
        H2, W2 = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        p0 = self._buf('pool:stem', (B * H2 * W2, 64))
        idx0 = self._buf('poolidx:stem', (B * H2 * W2, 64), torch.uint8)
        N.call('t3d_maxpool_fwd', dt, N.ptr(y0), pro0, N.ptr(p0), N.ptr(idx0), B, H1, W1, 64, st)
        sv['stem'] = dict(col=col0, y=y0, pro=pro0, bn=bn0, idx=idx0, H=H1, W=W1)
        cur = _Src(p0, None, B, H2, W2, 64)
        for li, (w, n, s) in enumerate(a.layers):
            for i in range(n):
                cur = self._bottleneck_fwd(f'layer{li + 1}.{i}', cur, w, s if i == 0 else 1, i == 0, sv)
        M = cur.B * cur.H * cur.W
        pooled = self._buf('pooled', (B, a.last_c), torch.float32)
        sv.update(last_in=cur, yl=cur.t, prol=None, pooled=pooled, HWl=cur.H * cur.W, bnl=None)
        return sv

    def _bottleneck_fwd(self, p, x, w, s, down, sv):
        st, dt = N.stream(), self.dt
        B, H, W, cin = x.B, x.H, x.W, x.C
        bn1, bn2, bn3 = self.bns[p + '.bn1'], self.bns[p + '.bn2'], self.bns[p + '.bn3']
        M = B * H * W
        y1 = self._buf('y1:' + p, (M, w))
        self._pw(x.t, None, self.w[p + '.conv1.weight'], y1, bn1, M, H * W, cin, w)
        pro1 = self._bnf(bn1, M, 'relu')
        Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        M2 = B * Ho * Wo
        kp = self._kp(p + '.conv2.weight')
        y2 = self._buf('y2:' + p, (M2, w))
        c3 = self._conv3.get(p + '.conv2.weight')
        if c3 is not None:
            # implicit GEMM: the kernel gathers the activated 3x3 neighbourhood itself, no patch matrix in HBM
            col = None
            N.call('t3d_conv3x3_fwd', dt | N.W_FRAG, N.ptr(y1), pro1, N.ptr(c3[0]), N.ptr(y2), self._st(bn2), B, H, W, w, w, s, st,
                   nbytes=(M + M2) * w * self.esz)
        else:
            col = self._buf('col:' + p, (M2, kp))
            N.call('t3d_im2col', dt, N.ptr(y1), pro1, N.ptr(col), B, H, W, w, 3, s, 1, kp, st)
            self._pw(col, None, self.w[p + '.conv2.weight'], y2, bn2, M2, Ho * Wo, kp, w)
        pro2 = self._bnf(bn2, M2, 'relu')
        y3 = self._buf('y3:' + p, (M2, 4 * w))
        self._pw(y2, pro2, self.w[p + '.conv3.weight'], y3, bn3, M2, Ho * Wo, w, 4 * w)
        pro3 = self._bnf(bn3, M2, 'none')
        rec = dict(p=p, x=x, w=w, s=s, down=down, y1=y1, pro1=pro1, col=col, y2=y2, pro2=pro2, y3=y3, Ho=Ho, Wo=Wo)
        z = self._buf('z:' + p, (M2, 4 * w))
        if down:
            bnd = self.bns[p + '.downsample.1']
            xs = x.t
            if s > 1:
                xs = self._buf('xs:' + p, (M2, cin))
                N.call('t3d_subsample', dt, N.ptr(x.t), N.ptr(xs), B, H, W, cin, s, 0, st)
            yd = self._buf('yd:' + p, (M2, 4 * w))
            self._pw(xs, None, self.w[p + '.downsample.0.weight'], yd, bnd, M2, Ho * Wo, cin, 4 * w)
            prod = self._bnf(bnd, M2, 'none')
            N.call('t3d_res_relu_fwd', dt, N.ptr(y3), pro3, N.ptr(yd), prod, N.ptr(z), M2, 4 * w, st)
            rec.update(xs=xs, yd=yd)
        else:
            N.call('t3d_res_relu_fwd', dt, N.ptr(y3), pro3, N.ptr(x.t), None, N.ptr(z), M2, 4 * w, st)
        out = _Src(z, None, B, Ho, Wo, 4 * w)
        rec['out'] = out
        sv['blocks'].append(rec)
        return out

    # ------------------------------------------------------------------ backward
    def _dgrad(self, dz, y, bb, wt, x_raw, gpro, residual, dx, bn_in, M, HW, K, Nn, bn=None):
        """dx [M,K] = (BN-backward of dz through y) W ; with x_raw / gpro: times act'(.) + the producer's backward sums.
        bn: the BatchNorm behind `bb` while its backward finalize is still pending -- the launch derives the coefficients in
        its prologue and publishes them (engine.py `_c`)."""
        wd, wp = self._wsel(wt, not bb.per_sample and not (gpro is not None and gpro.se))
        self._c('t3d_pwconv_dgrad', wd, N.ptr(dz), N.ptr(y), bb, wp, N.ptr(x_raw) if x_raw is not None else None,
                gpro, N.ptr(residual) if residual is not None else None, N.ptr(dx),
                self._bst(bn_in) if bn_in is not None else None, None, M, HW, K, Nn, N.stream(), bwd=bn,
                nbytes=M * (K + Nn) * self.esz)

    def _backward_backbone(self, sv, dpooled, dw32):
        a, st, dt, B = self.arch, N.stream(), self.dt, sv['B']
        x = sv['last_in']
        M, HW = B * sv['HWl'], sv['HWl']
        dz = self._buf('dz:last', (M, a.last_c))
        N.call('t3d_pool_bwd', dt, N.ptr(dpooled), N.ptr(sv['yl']), None, self.pool, N.ptr(sv['pool_argmax']), N.ptr(dz), None,
               B, HW, a.last_c, st)
        for rec in reversed(sv['blocks']):
            dz = self._bottleneck_bwd(rec, dz)
            self._maybe_hook(self.offsets[rec['p'] + '.conv1.weight'][0])
        # ---- stem: max-pool -> BN -> conv 7x7 weight gradient
        s0 = sv['stem']
        bn0 = s0['bn']
        H1, W1 = s0['H'], s0['W']
        M1 = B * H1 * W1
        dy0 = self._buf('dy:stem', (M1, 64))
        N.call('t3d_maxpool_bwd', dt, N.ptr(dz), N.ptr(s0['idx']), N.ptr(s0['y']), s0['pro'], N.ptr(dy0), self._bst(bn0),
               B, H1, W1, 64, st)
        bb0 = self._bnb(bn0)
        self._conv_wgrad('conv1.weight', dy0, s0['y'], bb0, s0['col'], M1, H1 * W1, 64, 3, 7)

    def _conv_wgrad(self, key, dz, y, bb, col, M, HW, Nn, C, k, bn=None):
        """dW of a k x k conv from its patch matrix (second stream), unpacked into the [N,C,k,k] gradient."""
        kp = self._kp(key)
        dwp = self._buf('dwp:' + key, (Nn, kp), torch.float32, zgroup='bwd')     # cleared by the backward's ONE batched launch (17 clears of ~18 us before)
        self._wgrad(self.dt, N.ptr(dz), N.ptr(y), bb, N.ptr(col), None, N.ptr(dwp), M, HW, kp, Nn, ro=bn,
                    nbytes=M * (kp + Nn) * self.esz)
        self._wgrad(N.ptr(dwp), N.ptr(self.g[key]), Nn, C, k, kp, entry='t3d_unpack_conv_grad')

    def _bottleneck_bwd(self, rec, dz):
        st, dt = N.stream(), self.dt
        p, x, w, s, down = rec['p'], rec['x'], rec['w'], rec['s'], rec['down']
        B, H, W, cin = x.B, x.H, x.W, x.C
        Ho, Wo = rec['Ho'], rec['Wo']
        M, M2, HW2 = B * H * W, B * Ho * Wo, Ho * Wo
        bn1, bn2, bn3 = self.bns[p + '.bn1'], self.bns[p + '.bn2'], self.bns[p + '.bn3']
        bnd = self.bns[p + '.downsample.1'] if down else None
        out = rec['out']
        # ---- tail: g = dz * [z > 0] = gradient at BN3's output and at the shortcut
        g = self._buf('g:' + p, (M2, 4 * w))
        N.call('t3d_res_relu_bwd', dt, N.ptr(dz), N.ptr(out.t), N.ptr(rec['y3']), N.ptr(rec['yd']) if down else None, N.ptr(g),
               self._bst(bn3), self._bst(bnd) if down else None, M2, 4 * w, st)
        # BatchNorm-backward coefficients: derived by their first readers (the weight-gradient launch for itself, the
        # data-gradient launch publishes them) -- a finalize launch of its own waits for a compute-unit slot behind the
        # second stream's long weight-gradient workgroups (40 us on average, 200 us at worst, 53 of them per step)
        bb3 = self._bn_bwd(bn3)
        # ---- conv3 (1x1, w -> 4w)
        self._wgrad(dt, N.ptr(g), N.ptr(rec['y3']), bb3, N.ptr(rec['y2']), rec['pro2'], N.ptr(self.g[p + '.conv3.weight']),
                    M2, HW2, w, 4 * w, ro=bn3, nbytes=M2 * 5 * w * self.esz)
        dv2 = self._buf('dv2:' + p, (M2, w))
        self._dgrad(g, rec['y3'], bb3, self.wt[p + '.conv3.weight'], rec['y2'], rec['pro2'], None, dv2, bn2, M2, HW2, w, 4 * w,
                    bn=bn3)
        bb2 = self._bn_bwd(bn2)
        # ---- conv2 (3x3, stride s): GEMM against the patch matrix, gradient back through the gather
        kp = self._kp(p + '.conv2.weight')
        d1 = self._buf('d1:' + p, (M, w))
        c3 = self._conv3.get(p + '.conv2.weight')
        if c3 is not None:
            # implicit GEMMs: dW from the gathered activations (second stream), dx by the transposed gather -- the gradient
            # lands at BN1's output with the ReLU mask and BN1's backward sums, as the 1x1 data gradients do
            key = p + '.conv2.weight'
            dwp = self._buf('dwp:' + key, (w, kp), torch.float32)          # (written by the launch: no clear)
            self._wgrad(dt, N.ptr(dv2), N.ptr(rec['y2']), bb2, N.ptr(rec['y1']), rec['pro1'], N.ptr(dwp), B, H, W, w, w, s,
                        entry='t3d_conv3x3_wgrad', ro=bn2, nbytes=(M + M2) * w * self.esz)
            self._wgrad(N.ptr(dwp), N.ptr(self.g[key]), w, w, 3, kp, entry='t3d_unpack_conv_grad')
            self._c('t3d_conv3x3_dgrad', dt | N.W_FRAG, N.ptr(dv2), N.ptr(rec['y2']), bb2, N.ptr(c3[2]), N.ptr(rec['y1']),
                    rec['pro1'], N.ptr(d1), self._bst(bn1), B, H, W, w, w, s, st, bwd=bn2, nbytes=(M + M2) * w * self.esz)
        else:
            self._conv_wgrad(p + '.conv2.weight', dv2, rec['y2'], bb2, rec['col'], M2, HW2, w, w, 3, bn=bn2)
            dcol = self._buf('dcol:' + p, (M2, kp))
            self._dgrad(dv2, rec['y2'], bb2, self.wt[p + '.conv2.weight'], None, None, None, dcol, None, M2, HW2, kp, w, bn=bn2)
            N.call('t3d_col2im_bwd', dt, N.ptr(dcol), N.ptr(rec['y1']), rec['pro1'], N.ptr(d1), self._bst(bn1), B, H, W, w, 3, s, 1,
                   kp, st)
        bb1 = self._bn_bwd(bn1)
        # ---- shortcut
        if down:
            bbd = self._bn_bwd(bnd)
            self._wgrad(dt, N.ptr(g), N.ptr(rec['yd']), bbd, N.ptr(rec['xs']), None, N.ptr(self.g[p + '.downsample.0.weight']),
                        M2, HW2, cin, 4 * w, ro=bnd, nbytes=M2 * (cin + 4 * w) * self.esz)
            dxs = self._buf('dxs:' + p, (M2, cin))
            self._dgrad(g, rec['yd'], bbd, self.wt[p + '.downsample.0.weight'], None, None, None, dxs, None, M2, HW2, cin, 4 * w,
                        bn=bnd)
            res = dxs
            if s > 1:
                res = self._buf('dxu:' + p, (M, cin))
                N.call('t3d_subsample', dt, N.ptr(dxs), N.ptr(res), B, H, W, cin, s, 1, st)
        else:
            res = g
        # ---- conv1 (1x1, cin -> w); its input is a finished (post-ReLU) tensor: the mask is applied by the block before
        self._wgrad(dt, N.ptr(d1), N.ptr(rec['y1']), bb1, N.ptr(x.t), None, N.ptr(self.g[p + '.conv1.weight']),
                    M, H * W, cin, w, ro=bn1, nbytes=M * (cin + w) * self.esz)
        dx = self._buf('dx:' + p, (M, cin))
        self._dgrad(d1, rec['y1'], bb1, self.wt[p + '.conv1.weight'], None, None, res, dx, None, M, H * W, cin, w, bn=bn1)
        return dx
