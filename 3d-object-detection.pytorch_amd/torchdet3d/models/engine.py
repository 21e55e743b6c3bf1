"""Execution engine of the regression network on MI355X: chains the C-ABI HIP kernels
(include/t3d.h) into the forward and the hand-derived backward of

    crops -> stem -> InvertedResidual x N -> 1x1 conv -> global pool [-> classifier] -> heads

i.e. `MobileNetV3.extract_features` (torchdet3d/models/mobilenetv3.py:199-203) +
`ModelWrapper.forward` (torchdet3d/builders/model_builder.py:126-146) and their autograd.

Data layout in HBM
  * activations are NHWC, stored as 2-D [B*H*W, C] in the storage dtype (fp32 parity mode or
    bf16 throughput mode); every convolution stores its RAW (pre-BatchNorm) output once and
    emits the BatchNorm batch sums from its epilogue; the consumer applies
    `act(scale*y + shift)` (and the squeeze-excite gate) on load, so BatchNorm, activation and
    SE never cost a pass over HBM.  Only the narrow block outputs are materialised
    (`t3d_bn_apply`: BatchNorm + skip connection).
  * backward mirrors this: a gradient-producing kernel stores the gradient at the producer's
    BatchNorm OUTPUT together with sum(dz), sum(dz*y); `t3d_bn_bwd_finalize` turns those into
    the affine  dy = alpha*dz + beta*y + gamma  that the wgrad / dgrad kernels apply on load.
  * all learnable parameters live in ONE flat fp32 buffer (`flat`), their gradients in a second
    (`gflat`): one fused optimizer update, one RCCL all-reduce bucket list; the state-dict
    tensors are views into it.
There is no fallback: every op goes through libt3d_hip.so or raises.
"""
import math
import os

import torch

from .. import _native as N
from .arch import Arch

BN_EPS, BN_MOM = 1e-5, 0.1      # PyTorch defaults the reference relies on
NREP = 16                       # reduction replicas (include/t3d.h: t3d_set_reduction_replicas)
FUSED_EVAL_MIN_B = int(os.environ.get('T3D_FUSED_EVAL_MIN_B', '96'))
DW_SLOTS = int(os.environ.get('T3D_DW_SLOTS', '512'))   # depthwise weight-gradient slots per layer (>= the workgroups of a t3d_dwconv_bwd launch)
PW_FRAG = os.environ.get('T3D_PW_FRAG', '1') != '0'      # fragment-order weights for the 16-bit pointwise kernels (A/B switch)
WORKSPACE_BYTES = 64 << 20      # partial weight-gradient tiles (include/t3d.h: t3d_set_workspace)
MAIN_WORKSPACE_BYTES = 16 << 20
# y-free expand-layer backward (csrc/pwconv_yfree.hip): minimum M*N elements of the expanded tensor; 0 disables
YFREE_MIN_ELEMS = int(os.environ.get('T3D_YFREE_MIN', 8 << 20))
YFREE_MAX_K = int(os.environ.get('T3D_YFREE_MAX_K', 112))     # widest narrow side the y-free backward is used for (the K x K Gram terms); round 6: 96 -> 112 takes MobileNetV3-large's two 112 -> 672 expansions at 14x14 (7.59-7.65 -> 7.43 ms per step; 160, the 7x7 stage: 7.53)
YFREE_FUSED = os.environ.get('T3D_YFREE_FUSED', '1') != '0'     # one-pass expand-layer backward (t3d_pwconv_bwd_yfree)
YFREE_PREP_FUSED = os.environ.get('T3D_YFREE_PREP_FUSED', '1') != '0'   # ... with its weight rows built in its own prologue (A/B switch)
EXPDW_EVAL = os.environ.get('T3D_EXPDW_EVAL', '1') != '0'      # fused expand + depthwise forward in 16-bit inference (A/B switch)
# round 6: the same fused forward in TRAINING, the expansion's BatchNorm statistics taken from the Gram matrix of the narrow block
# input (csrc/gram.hip).  OPT-IN (T3D_GRAM_FWD_MIN_HW = smallest input plane, in pixels per image, it is used at; 12544 = the
# 112x112 block, the only stage where the fused launch with the expansion stored beats the two launches alone: DESIGN.md finding
# 40): measured in the step it is 0.03-0.04 ms SLOWER than the two launches (finding 55), so the default is off.
GRAM_FWD_MIN_HW = int(os.environ.get('T3D_GRAM_FWD_MIN_HW', 0))
SE_FOLD = os.environ.get('T3D_SE_FOLD', '1') != '0'   # gated blocks: the BatchNorm-backward finalizes ride on their first reader (round 6; 0: standalone launches)
HOOK_MIN = 1 << 20              # gradient-exchange granularity (elements): parallel.GradSync's bucket size
HOOK_ON_SIDE = True              # (round 3: the gradient exchange is issued from the second stream; the other order stalled the main one)


class _BN:
    """Per-BatchNorm bookkeeping: parameter views + per-step scratch slices."""
    __slots__ = ('name', 'C', 'gamma', 'beta', 'dgamma', 'dbeta', 'rm', 'rv', 'nbt', 'stats', 'bstats', 'scale',
                 'shift', 'mean', 'invstd', 'alpha', 'bbeta', 'gammac', 'count', 'pro_cache', 'pend', 'desc', 'nrep')


def _nrep_for(C):
    return 16 if C <= 160 else (8 if C <= 640 else 4)


def _concurrent_stream(device, tries=8):
    """A second stream that really runs beside the current one.  HIP maps streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, default 4) round-robin; once RCCL has created its own streams a fresh stream may share the
    main stream's queue, and everything issued on it is then serialized behind the main stream's kernels (measured:
    +1.7 ms per step on the data-parallel path).  Probe candidates with a spin kernel on the main stream and keep the
    first one whose work finishes while the spin is still running."""
    main = torch.cuda.current_stream(device)
    cands = []
    with torch.cuda.device(device):
        probe = torch.zeros(1, device=device)
        for _ in range(tries):
            st = torch.cuda.Stream(device=device, priority=0)
            cands.append(st)
            e0, e_side, e_main = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            torch.cuda.synchronize(device)
            e0.record(main)
            torch.cuda._sleep(4_000_000)            # ~2 ms of spinning on the main stream
            e_main.record(main)
            with torch.cuda.stream(st):
                probe.add_(1.0)
                e_side.record(st)
            torch.cuda.synchronize(device)
            ratio = e0.elapsed_time(e_side) / max(e0.elapsed_time(e_main), 1e-6)
            _concurrent_stream.log.append(round(ratio, 3))
            if ratio < 0.5:
                return st
    return cands[0]


_concurrent_stream.log = []


class _Src:
    """A tensor as a consumer sees it: `t` [M,C] + how to read it (`pro`), and, for the backward,
    the raw producer tensor / BatchNorm its gradient must be reported against."""
    __slots__ = ('t', 'pro', 'B', 'H', 'W', 'C', 'raw', 'bn', 'gpro', 'finished_act', 'dz', 'zres', 'zbuf')

    def __init__(self, t, pro, B, H, W, C, raw=None, bn=None, gpro=None, finished_act=False):
        self.t, self.pro, self.B, self.H, self.W, self.C = t, pro, B, H, W, C
        self.raw, self.bn, self.gpro, self.finished_act = raw, bn, gpro, finished_act
        self.dz = None
        # a block output that has not been materialised yet: t = raw projection output, pro = its BatchNorm affine,
        # zres = the skip connection's tensor (or None), zbuf = where z = BN(t) + zres goes (Net._resolve / _pw_from)
        self.zres = self.zbuf = None


class Net:
    def __init__(self, name, num_classes=9, device='cuda', dtype=torch.float32, pooling_mode='avg', share=None):
        """share: another Net of the same architecture whose master weights / gradient buffer / BatchNorm buffers this
        one aliases (a second storage precision over ONE set of parameters, e.g. fp32 validation of a bf16-trained
        model: builders.model_builder `model.eval_storage_dtype`)."""
        if pooling_mode not in N.POOL:                 # model_builder.py:105-106
            raise ValueError(f'Unknown pooling mode: {pooling_mode}')
        self.pooling_mode, self.pool = pooling_mode, N.POOL[pooling_mode]
        self.arch = Arch(name)
        self.name, self.num_classes = name, num_classes
        self.device = torch.device(device)
        # torch.float16: INFERENCE storage only (the forward kernels exist in fp16, csrc/pwconv_stream_f16.hip etc.; gradients
        # keep bf16's range) -- three more mantissa bits than bf16 at every MFMA operand, same bytes
        assert dtype in (torch.float32, torch.bfloat16, torch.float16)
        self.dtype = dtype
        self.dt = {torch.float32: N.F32, torch.bfloat16: N.BF16, torch.float16: N.F16}[dtype]
        self.esz = 4 if dtype == torch.float32 else 2      # bytes per stored activation element
        self._share = share
        self._layout()
        self._bufs = {}
        self._packed_dirty = True
        self._packed_version = -1
        self.saved = None
        self.generation = 0           # id of the latest train-mode forward (whose activations `saved` holds)
        # called as grad_hook(lo) once every gradient at flat offsets >= lo is final (backward runs from the
        # end of `gflat` towards its start): lets a data-parallel wrapper start the RCCL all-reduce of that
        # tail while the rest of the backward is still being computed
        self.grad_hook = None
        self._cur_pool_exact = False
        self._side = None
        if self.device.type == 'cuda' and not os.environ.get('T3D_NO_SIDE_STREAM'):
            self._side = _concurrent_stream(self.device)
        self._side_busy = False
        self.persistent_outputs = False
        # uint8 input path (include/t3d.h: t3d_stem_fwd, fmt 1): normalisation of configs/default_config.py:9-10
        self.set_input_normalization([0.5931, 0.4690, 0.4229], [0.2471, 0.2214, 0.2157])
        self._fused_eval = not os.environ.get('T3D_NO_FUSED_EVAL')   # 14x14 / 7x7 blocks as one launch in inference mode

    def set_input_normalization(self, mean, std):
        """Per-channel mean / std applied to uint8 crops inside the stem kernel ((u/255 - mean) / std)."""
        self.in_mean = torch.tensor(mean, dtype=torch.float32, device=self.device)
        self.in_istd = 1.0 / torch.tensor(std, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ parameters
    def _layout(self):
        a = self.arch
        shapes = a.param_shapes(self.num_classes)
        self.shapes = shapes
        # flat order: everything in state-dict order, except the 9 regressors which are packed as
        # [9,18,F] weights + [9,18] biases so the head kernel indexes them by class
        order = [k for k, (s, kind) in shapes.items() if kind == 'param' and not k.startswith('regressors.')]
        off, self.offsets = 0, {}
        for k in order:
            n = int(math.prod(shapes[k][0]))
            self.offsets[k] = (off, n)
            off += (n + 3) // 4 * 4          # keep every tensor 16-B aligned
        F = a.feat_c
        self.offsets['__wreg'] = (off, 9 * 18 * F)
        off += 9 * 18 * F
        self.offsets['__breg'] = (off, 9 * 18 + 2)
        off += 164
        for k in range(9):
            self.offsets[f'regressors.{k}.0.weight'] = (self.offsets['__wreg'][0] + k * 18 * F, 18 * F)
            self.offsets[f'regressors.{k}.0.bias'] = (self.offsets['__breg'][0] + k * 18, 18)
        self.nparams = off
        dev = self.device
        sh = self._share
        assert sh is None or (sh.nparams == off and sh.name == self.name and sh.device == dev)
        self.flat = sh.flat if sh is not None else torch.zeros(off, device=dev)
        self.gflat = sh.gflat if sh is not None else torch.zeros(off, device=dev)
        self.p, self.g = {}, {}
        for k, (s, kind) in shapes.items():
            if kind == 'param':
                o, n = self.offsets[k]
                self.p[k] = self.flat[o:o + n].view(s)
                self.g[k] = self.gflat[o:o + n].view(s)
        self.wreg, self.breg = self._view('__wreg', (9, 18, F)), self._view('__breg', (9, 18), 162)
        self.dwreg = self._view('__wreg', (9, 18, F), g=True)
        self.dbreg = self._view('__breg', (9, 18), 162, g=True)
        # buffers
        self.buffers = {}
        self.bns = {}
        for k, (s, kind) in shapes.items():
            if kind == 'buffer':
                self.buffers[k] = sh.buffers[k] if sh is not None else (
                    torch.zeros(s, device=dev, dtype=torch.int64) if k.endswith('tracked')
                    else (torch.ones(s, device=dev) if k.endswith('var') else torch.zeros(s, device=dev)))
        nbn = [k[:-len('.running_mean')] for k in shapes if k.endswith('.running_mean')]
        tot = sum(shapes[k + '.weight'][0][0] for k in nbn)
        self._statbuf = torch.zeros(NREP, 4 * tot, device=dev, dtype=torch.float64)   # replicas x (fwd sums | bwd sums)
        self._stat_stride = 4 * tot
        self._aff = torch.zeros(7 * tot, device=dev)                            # scale shift mean invstd alpha beta gammac
        o = 0
        for k in nbn:
            C = shapes[k + '.weight'][0][0]
            b = _BN()
            b.name, b.C = k, C
            b.gamma, b.beta = self.p[k + '.weight'], self.p[k + '.bias']
            b.dgamma, b.dbeta = self.g[k + '.weight'], self.g[k + '.bias']
            b.rm, b.rv, b.nbt = self.buffers[k + '.running_mean'], self.buffers[k + '.running_var'], \
                self.buffers[k + '.num_batches_tracked']
            b.stats = self._statbuf[0, 2 * o:2 * o + 2 * C]          # replica 0; replica r is _stat_stride doubles further
            b.bstats = self._statbuf[0, 2 * tot + 2 * o:2 * tot + 2 * o + 2 * C]
            sl = [self._aff[i * tot + o:i * tot + o + C] for i in range(7)]
            b.scale, b.shift, b.mean, b.invstd, b.alpha, b.bbeta, b.gammac = sl
            b.pro_cache = {}
            self.bns[k] = b
            o += C
        self._bn_total = tot
        for b in self.bns.values():
            b.pend = [False, False]        # sums complete, finalize not yet run: forward / backward
            b.desc = {}
            # replicas of this BatchNorm's sums: many for the narrow layers (hundreds of workgroups add into a few dozen
            # addresses), few for the wide ones (whoever finalizes reads nrep x 2C doubles)
            b.nrep = _nrep_for(b.C)
        self._cur_nrep = None
        # T3D_NO_LAZY_BN=1: every BatchNorm finalize as a launch of its own (the round-2 behaviour)
        self._lazy = not os.environ.get('T3D_NO_LAZY_BN')
        self._zfuse = True              # block outputs materialised by the consuming 1x1 conv (t3d_pwconv_fwd_mat)
        if sh is None:
            self.reset_parameters()

    def nonfinite(self):
        """True when a BatchNorm of the latest steps saw a non-finite batch sum.  The 16-bit kernels form ReLU6 with the clamp
        modifier (DESIGN.md finding 30), which turns a NaN pre-activation into 0 where the reference's hardtanh propagates it, so
        a diverged step can end in a finite loss; the fp64 batch sums are taken on the RAW convolution outputs and still see
        it, and every coefficient derived from them (scale / shift / mean / invstd, the backward's alpha / beta / gamma) is then
        non-finite.  One small reduction + a host sync: call it where the loop already waits (Trainer.train's drain, bench.py)."""
        return not bool(torch.isfinite(self._aff).all())

    def _view(self, key, shape, n=None, g=False):
        o, m = self.offsets[key]
        return (self.gflat if g else self.flat)[o:o + (n or m)].view(shape)

    @torch.no_grad()
    def reset_parameters(self, seed=None):
        """Initialisation of the reference: mobilenetv3.py:205-218 for the backbone (conv N(0, sqrt(2/(k*k*Cout))),
        BN 1/0, Linear N(0, .01)/0), PyTorch's default Linear init for the heads added afterwards by
        ModelWrapper (model_builder.py:79-85)."""
        gen = torch.Generator(device='cpu')
        if seed is not None:
            gen.manual_seed(seed)
        for k, (s, kind) in self.shapes.items():
            if kind != 'param':
                continue
            if k.startswith('regressors') or k.startswith('cls_fc'):
                bound = 1.0 / math.sqrt(self.arch.feat_c)
                v = (torch.rand(s, generator=gen) * 2 - 1) * bound
            elif len(s) == 4:
                v = torch.randn(s, generator=gen) * math.sqrt(2.0 / (s[2] * s[3] * s[0]))
            elif len(s) == 2:
                v = torch.randn(s, generator=gen) * 0.01
            elif k.endswith('.bias'):
                v = torch.zeros(s)
            else:
                v = torch.ones(s)
            self.p[k].copy_(v)
        self._packed_dirty = True

    def state_dict(self):
        out = {}
        for k, (s, kind) in self.shapes.items():
            out[k] = (self.p[k] if kind == 'param' else self.buffers[k]).detach().clone()
        return out

    @torch.no_grad()
    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.shapes if k not in sd]
        if strict and missing:
            raise RuntimeError(f'missing keys: {missing[:5]}...')
        for k, v in sd.items():
            if k not in self.shapes:
                if strict:
                    raise RuntimeError(f'unexpected key {k}')
                continue
            dst = self.p[k] if self.shapes[k][1] == 'param' else self.buffers[k]
            dst.copy_(torch.as_tensor(v).to(dst.device, dst.dtype).view(dst.shape))
        self._packed_dirty = True

    # ------------------------------------------------------------------ scratch
    def _buf(self, tag, shape, dtype=None, zero=False, zgroup=None):
        """Scratch tensor by (tag, shape, dtype).  zero: cleared here, by a launch of its own; zgroup ('fwd' / 'bwd'): cleared
        by the ONE batched launch at the start of that pass (`_zero_group`) once it is on that launch's list -- the pooled
        sums / per-sample sums of MobileNetV3's squeeze-excite blocks were 18 clears of ~6 us (+ the gap behind each) per step."""
        dtype = dtype or self.dtype
        key = (tag, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(shape, device=self.device, dtype=dtype)
            self._bufs[key] = t
        if zgroup is not None:
            zg = self.__dict__.setdefault('_zgroups', {'fwd': {}, 'bwd': {}})
            cov = self.__dict__.setdefault('_zcovered', {'fwd': frozenset(), 'bwd': frozenset()})
            nb = t.numel() * t.element_size()
            if (t.data_ptr(), nb) not in cov[zgroup]:
                if nb % 16 == 0:
                    zg[zgroup][(t.data_ptr(), nb)] = t          # on the list from the next pass on
                self._zero(t)
        elif zero:
            self._zero(t)
        return t

    def _zero_group(self, group, extra_rows=()):
        """One t3d_zero_batched over `extra_rows` + every buffer registered for `group`; rebuilt when the list grew."""
        zg = self.__dict__.setdefault('_zgroups', {'fwd': {}, 'bwd': {}})
        cov = self.__dict__.setdefault('_zcovered', {'fwd': frozenset(), 'bwd': frozenset()})
        desc = self.__dict__.setdefault('_zgdesc', {})
        keys = frozenset(zg[group])
        dk = (group, len(extra_rows))                 # (train / eval forwards differ in their fixed rows)
        if keys != cov[group]:
            for k in [k for k in desc if k[0] == group]:
                del desc[k]
            cov[group] = keys
        if dk not in desc:
            rows = [list(r) for r in extra_rows] + [[p, nb] for (p, nb) in sorted(keys)]
            desc[dk] = torch.tensor(rows, dtype=torch.int64, device=self.device) if rows else None
        if desc[dk] is not None:
            N.call('t3d_zero_batched', N.ptr(desc[dk]), desc[dk].shape[0], N.stream())

    def _zero(self, t):
        """Clears a device buffer on the current stream with the library's own kernel (`t3d_zero_batched`, one-row descriptor
        kept per buffer): no framework launch inside the step, and a step plan being recorded sees the clear."""
        nb = t.numel() * t.element_size()
        if t.is_cuda and nb % 16 == 0 and t.is_contiguous():
            zd = self.__dict__.setdefault('_zdescs', {})
            d = zd.get((t.data_ptr(), nb))
            if d is None:
                d = zd[(t.data_ptr(), nb)] = torch.tensor([[t.data_ptr(), nb]], dtype=torch.int64, device=self.device)
            N.call('t3d_zero_batched', N.ptr(d), 1, N.stream())
        else:
            if N.recorder is not None:
                N.recorder.broken = 'a framework fill inside the step (buffer size not a multiple of 16 bytes)'
            t.zero_()

    def _pack(self):
        """fp32 master weights -> storage dtype (+ transposed copies for the data-gradient GEMMs).  Re-done whenever
        the master weights may have moved: `flat`, its views `p[...]` and the nn.Parameter the optimizer updates share
        one version counter, which every in-place update bumps (the hand-written optimizer kernel bumps it
        explicitly), so an eval forward right after `optimizer.step()` sees the new weights."""
        if not self._packed_dirty and self._packed_version == self.flat._version:
            return
        st = N.stream()
        if getattr(self, '_pack_desc', None) is None:
            # one descriptor table for every 1x1 weight: a single launch re-packs them all each step
            self.w, self.wt, rows = {}, {}, []
            # fragment-order copies (include/t3d.h: T3D_W_FRAG; the streaming kernel stages them with a linear copy, the
            # deep-contraction kernel streams them), keyed by the plain copy's address: forward (K, N), data gradient (N, K)
            self._frag = {}
            lib = N.lib()
            for k, (s, kind) in self.shapes.items():
                if kind != 'param' or len(s) != 4 or s[2] != 1:
                    continue
                n, kk = s[0], s[1]
                src = self.p[k]
                if self.dt == N.F32:
                    self.w[k], out = src, 0
                else:
                    self.w[k] = self._buf('w:' + k, (n, kk))
                    out = self.w[k].data_ptr()
                self.wt[k] = self._buf('wt:' + k, (kk, n))
                fr = frt = 0
                if self.dt in (N.BF16, N.F16) and PW_FRAG:
                    if lib.t3d_pwconv_wants_frag(kk, n):
                        f = self._buf('wf:' + k, (lib.t3d_pwconv_frag_bytes(n, kk) // 2,), zero=True)
                        self._frag[self.w[k].data_ptr()], fr = f, f.data_ptr()
                    if lib.t3d_pwconv_wants_frag(n, kk):
                        f = self._buf('wtf:' + k, (lib.t3d_pwconv_frag_bytes(kk, n) // 2,), zero=True)
                        self._frag[self.wt[k].data_ptr()], frt = f, f.data_ptr()
                rows.append([src.data_ptr(), out, self.wt[k].data_ptr(), n, kk, fr, frt])
            self._pack_desc = torch.tensor(rows, dtype=torch.int64, device=self.device)
        N.call('t3d_pack_weights_batched', self.dt, N.ptr(self._pack_desc), self._pack_desc.shape[0], st)
        # squeeze-excite FCs (nn.Linear, fp32): transposed copies for the one-launch forward (t3d_se_fwd_fused)
        if getattr(self, '_se_pack_desc', None) is None:
            rows = []
            for k, (s, kind) in self.shapes.items():
                if kind == 'param' and len(s) == 2 and ('.fc.0.weight' in k or '.fc.2.weight' in k):
                    self.wt[k] = self._buf('wt:' + k, (s[1], s[0]), torch.float32)
                    rows.append([self.p[k].data_ptr(), 0, self.wt[k].data_ptr(), s[0], s[1], 0, 0])
            self._se_pack_desc = torch.tensor(rows, dtype=torch.int64, device=self.device) if rows else False
        if self._se_pack_desc is not False:
            N.call('t3d_pack_weights_batched', N.F32, N.ptr(self._se_pack_desc), self._se_pack_desc.shape[0], st)
        self._pack_extra(st)
        if self.arch.kind != 'mobilenet':
            self._packed_dirty = False
            self._packed_version = self.flat._version
            return
        # stem: [C,3,3,3] -> [C,32] patch-row weights (columns 27..31 zero)
        c0 = self.arch.stem_c
        w32 = self._buf('stem32', (c0, 32), torch.float32)
        N.call('t3d_copy_cols', N.ptr(self.p['features.0.0.weight']), N.ptr(w32), c0, 27, 32, st)
        if self.dt == N.F32:
            self.w['stem'] = w32
        else:
            self.w['stem'] = self._buf('w:stem', (c0, 32))
            N.call('t3d_pack_weight', self.dt, N.ptr(w32), N.ptr(self.w['stem']), c0, 32, 0, st)
        if self.arch.classifier:
            wc = self.p['classifier.0.weight']
            self.wt['classifier'] = self._buf('wt:cls', (wc.shape[1], wc.shape[0]), torch.float32)
            N.call('t3d_pack_weight', N.F32, N.ptr(wc), N.ptr(self.wt['classifier']), wc.shape[0], wc.shape[1], 1, st)
        self._packed_dirty = False
        self._packed_version = self.flat._version

    def _pack_extra(self, st):
        """Hook: weight layouts of architectures with more than 1x1 / depthwise / stem convolutions (models/resnet.py)."""

    # ------------------------------------------------------------------ BatchNorm helpers
    def _fold_desc(self, bn, which, count):
        """Device-resident t3d_bn_fold descriptor of one BatchNorm (0: forward, 1: backward finalize), rebuilt only when
        the element count behind the sums changes (a new batch size / resolution)."""
        key = (which, float(count))
        d = bn.desc.get(key)
        if d is None:
            f = N.BnFold()
            f.kind, f.C, f.count = which + 1, bn.C, float(count)
            f.nrep, f.rstride = bn.nrep, self._stat_stride
            f.gamma = N.ptr(bn.gamma)
            if which == 0:
                f.stats, f.beta, f.rm, f.rv, f.nbt = N.ptr(bn.stats), N.ptr(bn.beta), N.ptr(bn.rm), N.ptr(bn.rv), N.ptr(bn.nbt)
                f.momentum, f.eps = BN_MOM, BN_EPS
                f.o0, f.o1, f.o2, f.o3 = N.ptr(bn.scale), N.ptr(bn.shift), N.ptr(bn.mean), N.ptr(bn.invstd)
            else:
                f.stats, f.mean, f.invstd = N.ptr(bn.bstats), N.ptr(bn.mean), N.ptr(bn.invstd)
                f.o0, f.o1, f.o2 = N.ptr(bn.alpha), N.ptr(bn.bbeta), N.ptr(bn.gammac)
                f.o3, f.o4 = N.ptr(bn.dgamma), N.ptr(bn.dbeta)
            host = torch.frombuffer(bytearray(bytes(f)), dtype=torch.uint8)
            d = bn.desc[key] = host.to(self.device)
        return d

    # Deferred BatchNorm finalize (csrc/common.h): `_bn_fwd` / `_bn_bwd` only MARK the sums as complete; the first launch
    # that reads the coefficients either derives them in its own prologue (`_c(..., fwd=bn)` / `bwd=bn`: a request to the
    # entry points that implement it -- they fall back to a finalize launch of their own on paths without the prologue)
    # or is preceded by the standalone finalize (`_settle_f` / `_settle_b`: every other reader).
    # (the depthwise kernels too -- T3D_LAZY_DW=0 | fwd | bwd | 1 --: every one of their 500-700 persistent workgroups then starts
    # with the ~3-us round trip to the sums, but the 5-us finalize launch and the ~6-us dispatch gap behind it go: 7.90 -> 7.83
    # ms per step on one box, 8.00 -> 7.97 on another (round 3b, forward half); with the atomics-free flush of round 3c the
    # backward half costs nothing either (7.64 ms both ways, 18 launches fewer), so both are on)
    _LAZY_DW = '1'
    DERIVING = frozenset(('t3d_pwconv_fwd', 't3d_pwconv_fwd_mat', 't3d_bn_apply', 't3d_pool_fwd', 't3d_pwconv_dgrad', 't3d_pwconv_yfree_prep',
                          't3d_pwconv_yfree_prep2', 't3d_conv3x3_dgrad', 't3d_pwconv_bwd_yfree_w', 't3d_se_bwd_affine')
                         + (('t3d_dwconv_fwd',) if _LAZY_DW in ('1', 'fwd') else ())
                         + (('t3d_dwconv_bwd',) if _LAZY_DW in ('1', 'bwd') else ()))

    def _c(self, entry, *args, fwd=None, bwd=None, **kw):
        """N.call of a coefficient-reading entry point: fwd / bwd = the BatchNorm whose forward / backward coefficients
        the launch reads first."""
        req = False
        if fwd is not None and fwd.pend[0]:
            if self._lazy and entry in self.DERIVING:
                N.call('t3d_fold_request', N.ptr(self._fold_desc(fwd, 0, fwd.count)), N.ptr(fwd.scale))
                fwd.pend[0], req = False, True
            else:
                self._settle_f(fwd)
        if bwd is not None and bwd.pend[1]:
            if self._lazy and entry in self.DERIVING and not req:
                N.call('t3d_fold_request', N.ptr(self._fold_desc(bwd, 1, bwd.count)), N.ptr(bwd.alpha))
                bwd.pend[1], req = False, True
            else:
                self._settle_b(bwd)
        N.call(entry, *args, **kw)
        if req and N.lib().t3d_fold_pending():
            raise RuntimeError(f'{entry} left a BatchNorm finalize request unserved (engine / library mismatch)')

    def _pool_exact(self, on):
        if on != self._cur_pool_exact:
            N.call('t3d_set_exact_pool', int(on))
            self._cur_pool_exact = on

    def _replicas(self, n):
        if n != self._cur_nrep:
            N.call('t3d_set_reduction_replicas', n, self._stat_stride)
            self._cur_nrep = n

    def _settle_f(self, bn):
        if bn.pend[0]:
            keep = self._cur_nrep
            self._replicas(bn.nrep)
            N.call('t3d_bn_finalize', N.ptr(bn.stats), bn.C, bn.count, N.ptr(bn.gamma), N.ptr(bn.beta),
                   N.ptr(bn.rm), N.ptr(bn.rv), N.ptr(bn.nbt), BN_MOM, BN_EPS, N.ptr(bn.scale), N.ptr(bn.shift),
                   N.ptr(bn.mean), N.ptr(bn.invstd), N.stream())
            self._replicas(keep)
            bn.pend[0] = False

    def _settle_b(self, bn):
        if bn.pend[1]:
            keep = self._cur_nrep
            self._replicas(bn.nrep)
            N.call('t3d_bn_bwd_finalize', N.ptr(bn.bstats), bn.C, bn.count, N.ptr(bn.gamma), N.ptr(bn.mean),
                   N.ptr(bn.invstd), N.ptr(bn.alpha), N.ptr(bn.bbeta), N.ptr(bn.gammac), N.ptr(bn.dgamma),
                   N.ptr(bn.dbeta), N.stream())
            self._replicas(keep)
            bn.pend[1] = False

    def _bn_fwd(self, bn, count, act):
        """Batch sums -> consumer affine (train: finalized by / ahead of the first reader) or running estimates -> affine
        (eval: every BatchNorm's affine was folded from the running estimates in one launch at the start of the forward)."""
        if self.training:
            bn.count = float(count)
            bn.pend[0] = True
        return self._pro(bn, act)

    def _eval_affines(self):
        if getattr(self, '_evdesc', None) is None:
            rows = [[N.ptr(b.gamma), N.ptr(b.beta), N.ptr(b.rm), N.ptr(b.rv), N.ptr(b.scale), N.ptr(b.shift), b.C]
                    for b in self.bns.values()]
            self._evdesc = torch.tensor(rows, dtype=torch.int64, device=self.device)
        N.call('t3d_bn_eval_affine_batched', N.ptr(self._evdesc), self._evdesc.shape[0], BN_EPS, N.stream())

    def _const(self, n, v):
        key = f'const:{n}:{v}'
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.full((n,), v, device=self.device, dtype=torch.float32)
        return t

    def _pro(self, bn, act, se=None, se_after=False):
        if se is None:
            p = bn.pro_cache.get(act)
            if p is None:
                p = bn.pro_cache[act] = N.prologue(bn.scale, bn.shift, None, act, False)
            return p
        return N.prologue(bn.scale, bn.shift, se, act, se_after)

    def _bn_bwd(self, bn):
        bn.pend[1] = True
        return N.bnbwd(bn.alpha, bn.bbeta, bn.gammac, False)

    def _st(self, bn):
        """Forward sums of `bn` as the output statistics of the launch being assembled (sets its replica count)."""
        if not self.training:
            return None
        self._replicas(bn.nrep)
        return N.ptr(bn.stats)

    def _bst(self, bn):
        """Backward sums of `bn` as the output statistics of the launch being assembled."""
        self._replicas(bn.nrep)
        return N.ptr(bn.bstats)

    # ------------------------------------------------------------------ forward
    def _main_scratch(self, on):
        # split-contraction scratch of the fp32 classifier products (main stream; include/t3d.h: t3d_set_main_workspace)
        if on and self.arch.classifier:
            ws = self._buf('workspace_main', (MAIN_WORKSPACE_BYTES,), torch.uint8)
            N.call('t3d_set_main_workspace', N.ptr(ws), MAIN_WORKSPACE_BYTES)
        else:
            N.call('t3d_set_main_workspace', None, 0)

    def forward(self, imgs, cats, train=False, dropout_mask=None, all_heads=False):
        self._cur_nrep = None
        self._replicas(NREP)
        self._main_scratch(True)
        try:
            return self._forward(imgs, cats, train, dropout_mask, all_heads)
        finally:
            N.call('t3d_set_reduction_replicas', 1, 0)
            self._pool_exact(False)
            self._cur_nrep = None
            self._main_scratch(False)

    def _forward(self, imgs, cats, train=False, dropout_mask=None, all_heads=False):
        """imgs [B,3,H,W] fp32 NCHW (the reference's input contract), cats int64 [B] ->
        kp [B,9,2] fp32 in (0,1), logits [B,num_classes] fp32 (None when num_classes == 1).
        all_heads (export mode, model_builder.py:112-124): kp [9,B,9,2], every regressor on every sample."""
        a, st = self.arch, N.stream()
        if train and self.dt == N.F16:
            raise RuntimeError('fp16 storage is an inference-only mode (build the model with storage_dtype bf16 or f32 to train)')
        sv = self._features(imgs, train)
        B = sv['B']
        pooled, cur, yl, prol = sv['pooled'], sv['last_in'], sv['yl'], sv['prol']
        if not all_heads:
            cats = cats.to(self.device, torch.int64).contiguous()
        sv['cats'] = cats
        # ---- global pool of the activated last feature map (model_builder.py:96-110)
        amax = None
        if self.pool != N.POOL['avg']:
            amax = self._buf('pool_argmax', (B, a.last_c), torch.int32)
        self._c('t3d_pool_fwd', self.dt, N.ptr(yl), prol, self.pool, N.ptr(pooled), N.ptr(amax), B, cur.H * cur.W,
                a.last_c, st, fwd=sv['bnl'])
        sv['pool_argmax'] = amax

        # ---- classifier Linear + BatchNorm1d + h_swish, MobileNetV3 only (mobilenetv3.py:191-195)
        f, fpro = pooled, None
        if a.classifier:
            bnc = self.bns['classifier.1']
            yc = self._buf('y:cls', (B, a.classifier), torch.float32)
            N.call('t3d_pwconv_fwd', N.F32, N.ptr(pooled), None, N.ptr(self.p['classifier.0.weight']),
                   N.ptr(self.p['classifier.0.bias']), N.ptr(yc), self._st(bnc), B, 1, a.last_c, a.classifier, st)
            fpro = self._bn_fwd(bnc, B, 'hswish')
            self._settle_f(bnc)
            f = yc
        # ---- heads (model_builder.py:137-144)
        ncls = self.num_classes
        keep = self.persistent_outputs and not all_heads
        # (persistent_outputs: the step plan's mode -- kp / logits live in engine buffers whose addresses do not change from
        # step to step; a caller of the plain API gets fresh tensors it may hold on to)
        logits = (self._buf('out:logits', (B, ncls), torch.float32) if keep else torch.empty(B, ncls, device=self.device)) if ncls > 1 else None
        if all_heads:
            kp = torch.empty(9, B, 18, device=self.device)
            N.call('t3d_head_fwd_all', N.ptr(f), fpro, N.ptr(self.wreg), N.ptr(self.breg),
                   N.ptr(self.p['cls_fc.1.weight']) if ncls > 1 else None,
                   N.ptr(self.p['cls_fc.1.bias']) if ncls > 1 else None, N.ptr(kp), N.ptr(logits), B, a.feat_c, ncls, st)
            self.saved = None
            return kp.view(9, B, 9, 2), logits
        mask = None
        if train and ncls > 1:
            mask = dropout_mask
            if mask is None:        # nn.Dropout(0.5): keep with p = .5, scale by 2 (model_builder.py:83)
                mask = self._buf('dropout', (B, a.feat_c), torch.float32)
                self._dropout_calls = getattr(self, '_dropout_calls', 0) + 1
                # (one process per GPU: main.py seeds every rank alike -- the rank term keeps the ranks' masks independent)
                seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * getattr(self, 'seed_rank', 0)) & ((1 << 64) - 1)
                N.call('t3d_dropout_mask', N.ptr(mask), B * a.feat_c, seed, self._dropout_calls, 0.5, st, slots={3: N.SLOT_DROPOUT})
            else:
                mask = mask.to(self.device, torch.float32).contiguous()
        kp = self._buf('out:kp', (B, 18), torch.float32) if keep else torch.empty(B, 18, device=self.device)
        N.call('t3d_head_fwd', N.ptr(f), fpro, N.ptr(cats), N.ptr(self.wreg), N.ptr(self.breg),
               N.ptr(self.p['cls_fc.1.weight']), N.ptr(self.p['cls_fc.1.bias']), N.ptr(mask), N.ptr(kp),
               N.ptr(logits), B, a.feat_c, ncls, st)
        sv.update(f=f, fpro=fpro, mask=mask, kp=kp)
        assert not any(b.pend[0] for b in self.bns.values()), 'a BatchNorm finalize was never run'
        if train:
            self.generation += 1
            sv['generation'] = self.generation
        self.saved = sv if train else None
        return kp.view(B, 9, 2), logits

    def forward_taps(self, imgs, tap_blocks):
        """Inference pass over the backbone up to the last requested block: -> {k: (tensor [B*H*W, C] in the storage dtype,
        B, H, W, C)} with the finished (BatchNorm + skip applied) outputs of `features.k` -- the feature maps an SSD head
        taps (configs/detection/mnv2_ssd_300_2_heads.py:7-18: the 96- and 320-channel maps of mobilenetv2)."""
        self._cur_nrep = None
        self._replicas(NREP)
        self._want_taps = set(tap_blocks)
        try:
            sv = self._features(imgs, False)
        finally:
            self._want_taps = None
            N.call('t3d_set_reduction_replicas', 1, 0)
            self._cur_nrep = None
        return {k: (s.t, s.B, s.H, s.W, s.C) for k, s in sv['taps'].items()}

    def extract_features(self, imgs):
        """`MobileNetV3.extract_features` (mobilenetv3.py:199-203) as a caller sees it: the activated last feature map,
        fp32 NCHW [B, C, H/32, W/32].  Inference only (running BatchNorm statistics); the layout conversion from the
        engine's NHWC storage is a convenience for the API, not part of the hot path."""
        self._cur_nrep = None
        self._replicas(NREP)
        try:
            sv = self._features(imgs, False)
        finally:
            N.call('t3d_set_reduction_replicas', 1, 0)
            self._cur_nrep = None
        cur, a = sv['last_in'], self.arch
        M = sv['B'] * cur.H * cur.W
        z = torch.empty(M, a.last_c, device=self.device, dtype=self.dtype)
        N.call('t3d_bn_apply', self.dt, N.ptr(sv['yl']), sv['prol'], None, N.ptr(z), M, a.last_c, N.stream())
        return z.float().view(sv['B'], cur.H, cur.W, a.last_c).permute(0, 3, 1, 2).contiguous()

    def _features(self, imgs, train):
        """Stem + inverted-residual blocks + last 1x1 conv (`extract_features`): fills and returns the saved-state
        dict with the RAW last feature map `yl` and its BatchNorm/activation prologue `prol`."""
        a, st, dt = self.arch, N.stream(), self.dt
        # crops: fp32 NCHW, normalised (the reference's input contract) -- or uint8 NHWC raw pixels, normalised inside the
        # patch gather with set_input_normalization()'s mean / std
        u8 = imgs.dtype == torch.uint8
        assert imgs.is_cuda and imgs.dim() == 4 and ((u8 and imgs.shape[3] == 3) or (imgs.dtype == torch.float32 and imgs.shape[1] == 3))
        imgs = imgs.contiguous()
        self.training = bool(train)
        self._pack()
        if u8:
            B, H, W, _ = imgs.shape
        else:
            B, _, H, W = imgs.shape
        if train:
            # the BatchNorm sum replicas of the step + the squeeze-excite pooled sums (`zgroup='fwd'` buffers): one launch
            self._zero_group('fwd', [[self._statbuf.data_ptr(), self._statbuf.numel() * self._statbuf.element_size()]])
        else:
            self._zero_group('fwd')
            self._eval_affines()
        sv = dict(B=B, imgs=imgs, blocks=[])
        self.saved_blocks = sv['blocks']

        # ---- stem (mobilenetv3.py:110-115,178): patch gather + GEMM (a direct stem that gathers the patches inside the GEMM
        # kernels was measured slower, DESIGN.md finding 13; tools/scratch/pruned_r4 has what is left of it)
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        M = B * Ho * Wo
        bn0 = self.bns['features.0.1']
        y0 = self._buf('y:stem', (M, a.stem_c))
        col = self._buf('col', (M, 32))
        if u8:
            N.call('t3d_stem_im2col_u8', dt, N.ptr(imgs), N.ptr(self.in_mean), N.ptr(self.in_istd), N.ptr(col), B, H, W, st)
        else:
            N.call('t3d_stem_im2col', dt, N.ptr(imgs), N.ptr(col), B, H, W, st)
        N.call('t3d_pwconv_fwd', dt, N.ptr(col), None, N.ptr(self.w['stem']), None, N.ptr(y0), self._st(bn0),
               M, Ho * Wo, 32, a.stem_c, st, nbytes=M * (32 + a.stem_c) * self.esz)
        pro0 = self._bn_fwd(bn0, M, a.stem_act)
        cur = _Src(y0, pro0, B, Ho, Wo, a.stem_c, raw=y0, bn=bn0, gpro=pro0)
        sv['col'], sv['stem'] = col, cur

        taps = getattr(self, '_want_taps', None)
        for i, blk in enumerate(a.blocks):
            cur = self._block_fwd(i, blk, cur, sv)
            if taps and (i + 1) in taps:        # finished output of features.{i+1} (the detector's feature maps)
                sv.setdefault('taps', {})[i + 1] = self._resolve(cur)

        # ---- last 1x1 conv (mobilenetv3.py:118-123,188) + global average pool (model_builder.py:96-110)
        ln = a.last_name
        bnl = self.bns[ln + '.1']
        M = cur.B * cur.H * cur.W
        yl = self._buf('y:last', (M, a.last_c))
        self._pw_from(cur, self.w[ln + '.0.weight'], yl, bnl, M, cur.H * cur.W, cur.C, a.last_c)
        prol = self._bn_fwd(bnl, M, a.last_act)
        pooled = self._buf('pooled', (B, a.last_c), torch.float32)
        sv.update(last_in=cur, yl=yl, prol=prol, pooled=pooled, HWl=cur.H * cur.W, bnl=bnl)
        return sv

    def _wsel(self, w, plain=True):
        """(dtype argument, weight pointer) of a pointwise launch: the fragment-order copy where the layer has one and the
        launch is a `plain` one (no squeeze-excite gate / per-sample coefficients: the gated variants of a deep layer have no
        fragment-order path in the deep-contraction kernel, csrc/pwconv_deep.hip, and take the row-major copy)."""
        f = self._frag.get(w.data_ptr()) if (plain and self.dt in (N.BF16, N.F16)) else None
        return (self.dt | N.W_FRAG, N.ptr(f)) if f is not None else (self.dt, N.ptr(w))

    def _resolve(self, src):
        """Materialise a pending block output (z = BN(y3) + skip) as a launch of its own -- for the consumers that are not
        a 1x1 conv (a 1x1 conv forms z while it loads its operand: `_pw_from`)."""
        if src.zbuf is not None:
            M = src.B * src.H * src.W
            self._c('t3d_bn_apply', self.dt, N.ptr(src.t), src.pro, N.ptr(src.zres), N.ptr(src.zbuf), M, src.C, N.stream(),
                    fwd=src.bn)
            src.t, src.pro, src.zres, src.zbuf = src.zbuf, None, None, None
        return src

    def _pw_from(self, x, w, y, bn_out, M, HW, K, Nn):
        """1x1 conv forward reading `x`; a pending block output is materialised by the conv itself (t3d_pwconv_fwd_mat)."""
        st, nb = N.stream(), M * (K + Nn) * self.esz
        if x.zbuf is not None:
            wd, wp = self._wsel(w)
            self._c('t3d_pwconv_fwd_mat', wd, N.ptr(x.t), x.pro, N.ptr(x.zres), N.ptr(x.zbuf), wp, N.ptr(y),
                    self._st(bn_out), M, HW, K, Nn, st, nbytes=nb, fwd=x.bn)
            x.t, x.pro, x.zres, x.zbuf = x.zbuf, None, None, None
        else:
            wd, wp = self._wsel(w, x.pro is None or not x.pro.se)
            self._c('t3d_pwconv_fwd', wd, N.ptr(x.t), x.pro, wp, None, N.ptr(y), self._st(bn_out), M, HW, K, Nn,
                    st, nbytes=nb, fwd=x.bn if x.pro is not None else None)

    def _finish(self, src, tag):
        """Materialise act(BN(y)) (needed when a deferred tensor also feeds a skip connection)."""
        M = src.B * src.H * src.W
        z = self._buf(tag, (M, src.C))
        self._c('t3d_bn_apply', self.dt, N.ptr(src.t), src.pro, None, N.ptr(z), M, src.C, N.stream(), fwd=src.bn)
        return _Src(z, None, src.B, src.H, src.W, src.C, raw=src.raw, bn=src.bn, gpro=src.gpro, finished_act=True)

    def _fused_eval_ok(self, blk, x):
        """Inference mode, bf16: may this block run as ONE launch (csrc/block_eval.hip: expanded tensors stay in LDS)?"""
        # one workgroup per image: below ~100 images the launch-per-layer path fills the chip better.  The two paths round at
        # the same points but sum in different orders, so bf16 inference of one sample is not bit-identical across the
        # threshold (last partial validation batch vs full ones); T3D_FUSED_EVAL_MIN_B fixes the choice for a run
        # (0: always fused where the shapes allow, a huge value: never).  fp32 inference (the default) never takes this path.
        if self.training or not self._fused_eval or self.dt != N.BF16 or x.pro is not None or x.B < FUSED_EVAL_MIN_B:
            return False
        if not blk.expand or blk.se or blk.s != 1 or blk.k != 3 or blk.cin % 32 or blk.cexp % 64 or blk.cout % 16:
            return False
        P = x.H * x.W
        mt = (P + 15) // 16
        nt2, mtw = blk.cout // 16, (mt + 7) // 8
        # (built but not used: 7x7 160 -> 960 -> 320, 102 us fused against 97 us layer by layer)
        if not ((mtw == 1 and nt2 == 10) or (mtw == 2 and nt2 in (4, 6))):
            return False
        lds = mt * 16 * (blk.cin + 8) * 2 + P * 68 * 4 + mt * 16 * 72 * 2 + 64 * (blk.cin + 8) * 2 + blk.cout * 72 * 2 + 64 * 9 * 4 + 4 * 64 * 4
        return lds <= 160 * 1024

    def _expdw_ok(self, blk, x):
        """Inference mode, bf16 / fp16 storage: may expand + depthwise of this block run as ONE launch (csrc/expdw_fwd.hip)?  In
        training mode the expansion's BatchNorm needs its batch statistics first, which costs a statistics-only pass of the 1x1
        conv (DESIGN.md finding 40: no gain with the raw expansion still stored for the backward)."""
        if self.training or not EXPDW_EVAL or self.dt not in (N.BF16, N.F16) or x.pro is not None or x.zbuf is not None:
            return False
        if not (blk.expand and not blk.se and blk.k == 3 and isinstance(blk.act, str) and blk.act in ('relu', 'relu6')):
            return False
        # the kernel's own shape tests (channel counts, and an input row no wider than its fragment registers / LDS rows hold:
        # W <= 213 -- a 448 ... 512-pixel crop reaches MobileNetV2's second block wider than that and takes the two launches)
        return bool(N.lib().t3d_expdw_supported(self.dt, N.ACT[blk.act], x.B, x.H, x.W, blk.cin, blk.cexp, blk.s))

    def _gram_fwd_ok(self, blk, x):
        """Training mode, bf16: may expand + depthwise of this block run as ONE launch, its BatchNorm statistics coming from the
        Gram matrix of the block input (one pass over the NARROW tensor, the one that materialises it anyway)?"""
        if not self.training or not GRAM_FWD_MIN_HW or self.dt != N.BF16 or x.H * x.W < GRAM_FWD_MIN_HW:
            return False
        if not (blk.expand and not blk.se and blk.k == 3 and blk.cin in (8, 16) and isinstance(blk.act, str) and blk.act in ('relu', 'relu6')):
            return False
        if x.pro is not None and (x.zbuf is None or x.pro.se is not None):
            return False           # (a deferred activation that is not a pending block output: the two launches)
        return bool(N.lib().t3d_expdw_supported(self.dt, N.ACT[blk.act], x.B, x.H, x.W, blk.cin, blk.cexp, blk.s))

    def _block_fwd(self, i, blk, x, sv):
        st, dt = N.stream(), self.dt
        p = f'features.{i + 1}.conv'
        B, H, W = x.B, x.H, x.W
        if self._fused_eval_ok(blk, x):
            bn1, bn2, bn3 = self.bns[p + '.1'], self.bns[p + '.4'], self.bns[p + '.8']
            z = self._buf(f'z:{i}', (B * H * W, blk.cout))
            act = N.ACT[blk.act] if isinstance(blk.act, str) else blk.act
            N.call('t3d_ir_block_eval', N.ptr(x.t), N.ptr(self.w[p + '.0.weight']), N.ptr(bn1.scale), N.ptr(bn1.shift),
                   act, N.ptr(self.p[p + '.3.weight']), N.ptr(bn2.scale), N.ptr(bn2.shift), act,
                   N.ptr(self.w[p + '.7.weight']), N.ptr(bn3.scale), N.ptr(bn3.shift), int(bool(blk.res)), N.ptr(z),
                   B, H, W, blk.cin, blk.cexp, blk.cout, st,
                   nbytes=B * H * W * (2 * blk.cin + 4 * blk.cexp + 2 * blk.cout) * self.esz)
            return _Src(z, None, B, H, W, blk.cout, raw=None, bn=bn3, gpro=None)
        if self._expdw_ok(blk, x):
            # inference, 16-bit storage, the 112x112 .. 28x28 blocks: expand 1x1 + BatchNorm + activation + depthwise 3x3 in ONE
            # launch, the expanded tensor never leaves LDS (csrc/expdw_fwd.hip; DESIGN.md finding 40)
            bn1, bn2, bn3 = self.bns[p + '.1'], self.bns[p + '.4'], self.bns[p + '.8']
            Ho, Wo = (H + 2 - 3) // blk.s + 1, (W + 2 - 3) // blk.s + 1
            M2 = B * Ho * Wo
            y2 = self._buf(f'y2:{i}', (M2, blk.cexp))
            N.call('t3d_expdw_fwd', dt, N.ptr(x.t), N.ptr(self.w[p + '.0.weight']), N.ptr(bn1.scale), N.ptr(bn1.shift),
                   N.ACT[blk.act] if isinstance(blk.act, str) else blk.act, N.ptr(self.p[p + '.3.weight']), None, N.ptr(y2), None,
                   B, H, W, blk.cin, blk.cexp, blk.s, st, nbytes=(B * H * W * blk.cin + M2 * blk.cexp) * self.esz)
            pro2 = self._bn_fwd(bn2, M2, blk.act)
            y3 = self._buf(f'y3:{i}', (M2, blk.cout))
            wd, wp = self._wsel(self.w[p + '.7.weight'])
            N.call('t3d_pwconv_fwd', wd, N.ptr(y2), pro2, wp, None, N.ptr(y3), None, M2, Ho * Wo, blk.cexp, blk.cout, st,
                   nbytes=M2 * (blk.cexp + blk.cout) * self.esz)
            out = _Src(y3, self._bn_fwd(bn3, M2, 'none'), B, Ho, Wo, blk.cout, raw=y3, bn=bn3, gpro=None)
            out.zres, out.zbuf = (x.t if blk.res else None), self._buf(f'z:{i}', (M2, blk.cout))
            return self._resolve(out)
        if x.zbuf is not None and not blk.expand:
            self._resolve(x)                   # (the depthwise conv of a no-expand block reads the finished tensor)
        if blk.res and x.pro is not None and x.zbuf is None:
            x = self._finish(x, f'z:in{i}')
        rec = dict(x=x)
        src = x
        fused_dw = False
        if blk.expand and self._gram_fwd_ok(blk, x):
            # training, the 112x112 expansion: [materialise the block input + its Gram matrix] -> BatchNorm coefficients of the
            # expansion from the K x K sums -> expand + BatchNorm + activation + depthwise in ONE launch, the raw expansion
            # stored for the backward (csrc/gram.hip, csrc/expdw_fwd.hip; DESIGN.md finding 55)
            bn1, bn2 = self.bns[p + '.1'], self.bns[p + '.4']
            M, K, C = B * H * W, blk.cin, blk.cexp
            gram = self._buf(f'gram:{i}', (16, K * (K + 1) // 2 + K), torch.float64, zgroup='fwd')      # 16 reduction replicas
            if x.zbuf is not None:
                self._settle_f(x.bn)
                N.call('t3d_bn_apply_gram', dt, N.ptr(x.t), x.pro, N.ptr(x.zres), N.ptr(x.zbuf), N.ptr(gram), M, K, st,
                       nbytes=2 * M * K * self.esz)
                x.t, x.pro, x.zres, x.zbuf = x.zbuf, None, None, None
            else:
                N.call('t3d_bn_apply_gram', dt, N.ptr(x.t), None, None, None, N.ptr(gram), M, K, st, nbytes=M * K * self.esz)
            N.call('t3d_gram_bn_finalize', N.ptr(gram), N.ptr(self.w[p + '.0.weight']), C, K, float(M), N.ptr(bn1.gamma),
                   N.ptr(bn1.beta), N.ptr(bn1.rm), N.ptr(bn1.rv), N.ptr(bn1.nbt), BN_MOM, BN_EPS, N.ptr(bn1.scale), N.ptr(bn1.shift),
                   N.ptr(bn1.mean), N.ptr(bn1.invstd), st)
            bn1.count, bn1.pend[0] = float(M), False
            Ho, Wo = (H + 2 - 3) // blk.s + 1, (W + 2 - 3) // blk.s + 1
            M2 = B * Ho * Wo
            y1 = self._buf(f'y1:{i}', (M, C))
            y2 = self._buf(f'y2:{i}', (M2, C))
            N.call('t3d_expdw_fwd', dt, N.ptr(x.t), N.ptr(self.w[p + '.0.weight']), N.ptr(bn1.scale), N.ptr(bn1.shift),
                   N.ACT[blk.act], N.ptr(self.p[p + '.3.weight']), N.ptr(y1), N.ptr(y2), self._st(bn2), B, H, W, K, C, blk.s, st,
                   nbytes=(M * K + 2 * M * C + M2 * C) * self.esz)      # (the two launches' algorithmic bytes, SURVEY 8d: fused > 1 is legitimate)
            pro1 = self._pro(bn1, blk.act)
            src = _Src(y1, pro1, B, H, W, C, raw=y1, bn=bn1, gpro=pro1)
            rec['s1'] = src
            dwn, bnn, pwn, bn3n = p + '.3.weight', p + '.4', p + '.7.weight', p + '.8'
            fused_dw = True
        elif blk.expand:                                                  # mobilenetv3.py:146-150
            bn1 = self.bns[p + '.1']
            M = B * H * W
            y1 = self._buf(f'y1:{i}', (M, blk.cexp))
            self._pw_from(x, self.w[p + '.0.weight'], y1, bn1, M, H * W, blk.cin, blk.cexp)
            pro1 = self._bn_fwd(bn1, M, blk.act)
            src = _Src(y1, pro1, B, H, W, blk.cexp, raw=y1, bn=bn1, gpro=pro1)
            rec['s1'] = src
            dwn, bnn, pwn, bn3n = p + '.3.weight', p + '.4', p + '.7.weight', p + '.8'
        else:                                                             # mobilenetv3.py:133-144
            dwn, bnn, pwn, bn3n = p + '.0.weight', p + '.1', p + '.4.weight', p + '.5'
        se_after = bool(blk.se and not blk.expand)         # gate AFTER the activation (mobilenetv3.py:138-140)
        sen = ((p + '.3') if se_after else (p + '.5')) if blk.se else None
        # depthwise k x k (mobilenetv3.py:136,152)
        pad = (blk.k - 1) // 2
        Ho, Wo = (H + 2 * pad - blk.k) // blk.s + 1, (W + 2 * pad - blk.k) // blk.s + 1
        M2 = B * Ho * Wo
        bn2 = self.bns[bnn]
        y2 = self._buf(f'y2:{i}', (M2, blk.cexp))
        # squeeze-excite pooled sums: int64 fixed point (include/t3d.h: t3d_set_exact_pool) -- the depthwise kernel's work items
        # add integers, so the sums, the gate and everything behind it no longer depend on their arrival order
        gap = self._buf(f'gap:{i}', (B, blk.cexp), torch.int64, zgroup='fwd') if (blk.se and not se_after) else None
        self._pool_exact(gap is not None)
        if not fused_dw:
            self._c('t3d_dwconv_fwd', dt, N.ptr(src.t), src.pro, N.ptr(self.p[dwn]), N.ptr(y2), self._st(bn2), N.ptr(gap),
                    B, H, W, blk.cexp, blk.k, blk.s, st, nbytes=(B * H * W + M2) * blk.cexp * self.esz,
                    fwd=src.bn if src.pro is not None else None)
        pro2 = self._bn_fwd(bn2, M2, blk.act)
        if blk.se:
            self._settle_f(bn2)          # the squeeze-excite kernels read bn2's affine first
        if se_after:
            # the gate sees the ACTIVATED tensor: one pooled pass over it, then the same two FCs (scale 1, shift 0, HW 1
            # make t3d_se_fwd take the pooled mean as it is)
            C, R = blk.cexp, blk.se
            pooled = self._buf(f'se_pool:{i}', (B, C), torch.float32)
            N.call('t3d_gap_fwd', dt, N.ptr(y2), pro2, N.ptr(pooled), B, Ho * Wo, C, st)
            se = dict(gap=pooled, m=self._buf(f'se_m:{i}', (B, C), torch.float32),
                      h=self._buf(f'se_h:{i}', (B, R), torch.float32), q=self._buf(f'se_q:{i}', (B, C), torch.float32),
                      s=self._buf(f'se_s:{i}', (B, C), torch.float32), name=sen, HW=Ho * Wo, after=True, pro2n=pro2)
            ones, zeros = self._const(C, 1.0), self._const(C, 0.0)
            self._pool_exact(False)
            N.call('t3d_se_fwd_fused', N.ptr(pooled), N.ptr(ones), N.ptr(zeros), N.ptr(self.wt[sen + '.fc.0.weight']),
                   N.ptr(self.p[sen + '.fc.0.bias']), N.ptr(self.wt[sen + '.fc.2.weight']),
                   N.ptr(self.p[sen + '.fc.2.bias']), N.ptr(se['m']), N.ptr(se['h']), N.ptr(se['q']), N.ptr(se['s']),
                   B, C, R, 1, st)
            pro2 = self._pro(bn2, blk.act, se['s'], True)
            rec['se'] = se
        elif blk.se:                                                      # mobilenetv3.py:92-107,155
            C, R = blk.cexp, blk.se
            se = dict(gap=gap, m=self._buf(f'se_m:{i}', (B, C), torch.float32),
                      h=self._buf(f'se_h:{i}', (B, R), torch.float32), q=self._buf(f'se_q:{i}', (B, C), torch.float32),
                      s=self._buf(f'se_s:{i}', (B, C), torch.float32), name=sen, HW=Ho * Wo)
            self._pool_exact(True)
            N.call('t3d_se_fwd_fused', N.ptr(gap), N.ptr(bn2.scale), N.ptr(bn2.shift), N.ptr(self.wt[sen + '.fc.0.weight']),
                   N.ptr(self.p[sen + '.fc.0.bias']), N.ptr(self.wt[sen + '.fc.2.weight']),
                   N.ptr(self.p[sen + '.fc.2.bias']), N.ptr(se['m']), N.ptr(se['h']), N.ptr(se['q']), N.ptr(se['s']),
                   B, C, R, Ho * Wo, st)
            pro2 = self._pro(bn2, blk.act, se['s'], False)                # SE before the activation (:155-156)
            rec['se'] = se
        s2 = _Src(y2, pro2, B, Ho, Wo, blk.cexp, raw=y2, bn=bn2, gpro=pro2)
        # linear 1x1 projection (mobilenetv3.py:142-143,158-159)
        bn3 = self.bns[bn3n]
        y3 = self._buf(f'y3:{i}', (M2, blk.cout))
        wd, wp = self._wsel(self.w[pwn], not pro2.se)
        self._c('t3d_pwconv_fwd', wd, N.ptr(y2), pro2, wp, None, N.ptr(y3), self._st(bn3),
                M2, Ho * Wo, blk.cexp, blk.cout, st, nbytes=M2 * (blk.cexp + blk.cout) * self.esz, fwd=bn2)
        pro3 = self._bn_fwd(bn3, M2, 'none')
        z = self._buf(f'z:{i}', (M2, blk.cout))
        # the block output z = BN(y3) (+ x) stays PENDING: the next 1x1 conv forms it on load and writes it out
        # (t3d_pwconv_fwd_mat) -- one launch and one pass over the narrow tensor less per block
        out = _Src(y3, pro3, B, Ho, Wo, blk.cout, raw=y3, bn=bn3, gpro=None)
        out.zres, out.zbuf = (x.t if blk.res else None), z
        # (inference: only the fp32 engine keeps it pending -- its 1x1 kernel materialises on load, csrc/pwconv_f32_reg.hip; the
        # 16-bit inference engines read finished tensors in their fused block / expand + depthwise kernels)
        if not ((self.training or self.dt == N.F32) and self._zfuse):
            self._resolve(out)
        rec.update(src=src, s2=s2, y3=y3, bn3=bn3, out=out, names=(dwn, pwn), blk=blk, idx=i)
        sv['blocks'].append(rec)
        return out

    # ------------------------------------------------------------------ backward
    def _wgrad(self, *args, **kw):
        """Pointwise weight gradients are leaves of the backward graph (only the optimizer reads them), so they are
        issued on a second HIP stream and run concurrently with the data-gradient chain of the main stream; every
        kernel here is latency- rather than bandwidth-bound, so the two streams overlap well.  Ordering: the side
        stream waits for everything enqueued on the main stream so far (inputs, BatchNorm-backward affine, the
        zeroed gradient buffer); `_join_side` makes the main stream wait for the side stream."""
        entry = kw.pop('entry', 't3d_pwconv_wgrad')
        ro = kw.pop('ro', None)
        # ro: BatchNorm whose backward coefficients this launch reads while their finalize is still pending -- the kernel
        # derives them for itself WITHOUT publishing (the data-gradient launch of the main stream, issued next, does)
        # (only where the bf16 transposed kernel will take the request: on the fp32-storage / T3D_WGRAD_TILED paths the entry
        # point would serve it with a PUBLISHING finalize launch on the side stream while the data gradient of the main
        # stream publishes the same values -- there the standalone finalize runs once, on the main stream, ahead of both)
        req = (ro is not None and ro.pend[1] and self._lazy and entry in ('t3d_pwconv_wgrad', 't3d_conv3x3_wgrad') and self.dt == N.BF16
               and not os.environ.get('T3D_WGRAD_TILED'))
        if ro is not None and ro.pend[1] and not req:
            self._settle_b(ro)
        if req:
            N.call('t3d_fold_request', N.ptr(self._fold_desc(ro, 1, ro.count)), N.ptr(ro.alpha))
        try:
            self._wgrad_issue(entry, args, kw)
        finally:
            if req and N.lib().t3d_fold_pending():
                raise RuntimeError(f'{entry} left a BatchNorm finalize request unserved (engine / library mismatch)')

    def _wgrad_issue(self, entry, args, kw):
        if self._side is None:
            N.call(entry, *args, N.stream(), **kw)
            return
        self._fork_side()
        if N.timer is None:
            N.call(entry, *args, self._side.cuda_stream, **kw)     # the ABI takes the stream: no context switch needed
        else:
            with torch.cuda.stream(self._side):                    # timed launches: events on the launch stream
                N.call(entry, *args, N.stream(), **kw)
        self._side_busy = True

    def _se_wgrad(self, se, sen, dq, dp, B, C, R):
        # FC weight / bias gradients of a squeeze-excite gate: off the data-gradient chain, on the weight-gradient stream
        self._wgrad(N.ptr(se['m']), N.ptr(se['h']), N.ptr(dq), N.ptr(dp), N.ptr(self.g[sen + '.fc.0.weight']),
                    N.ptr(self.g[sen + '.fc.0.bias']), N.ptr(self.g[sen + '.fc.2.weight']), N.ptr(self.g[sen + '.fc.2.bias']),
                    B, C, R, entry='t3d_se_bwd_weights')

    def _fork_side(self):
        """The second stream waits for everything enqueued on the main stream so far (an event record + a stream wait; noted in
        the step plan being recorded, if any)."""
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        if N.recorder is not None:
            N.recorder.add_fork(main.cuda_stream, self._side.cuda_stream)

    def _join_side(self):
        if self._side is not None and self._side_busy:
            main = torch.cuda.current_stream()
            main.wait_stream(self._side)
            if N.recorder is not None:
                N.recorder.add_fork(self._side.cuda_stream, main.cuda_stream)
            self._side_busy = False

    def backward(self, dkp, dlogits=None):
        ws = self._buf('workspace', (WORKSPACE_BYTES,), torch.uint8)
        self._cur_nrep = None
        self._replicas(NREP)
        N.call('t3d_set_workspace', N.ptr(ws), WORKSPACE_BYTES)
        self._main_scratch(True)
        try:
            return self._backward(dkp, dlogits)
        finally:
            N.call('t3d_set_reduction_replicas', 1, 0)
            N.call('t3d_set_dw_slots', 0, None)
            self._pool_exact(False)
            self._cur_nrep = None
            N.call('t3d_set_workspace', None, 0)
            self._main_scratch(False)

    def _backward(self, dkp, dlogits=None):
        """Gradient of the last train-mode forward w.r.t. every parameter -> `gflat` (overwritten).
        dkp [B,9,2] / dlogits [B,num_classes]: d loss / d outputs (fp32)."""
        sv = self.saved
        assert sv is not None, 'backward() needs a preceding forward(train=True)'
        a, st, dt, B = self.arch, N.stream(), self.dt, sv['B']
        if getattr(self, '_dwarena', None) is None:
            self._dw_arena_init()
        # one launch clears every accumulate-into buffer of the backward: gradients, depthwise replicas, stem patch-row dW
        dw32 = self._buf('dstem32', (a.stem_c, 32), torch.float32)
        if getattr(self, '_zero_rows', None) is None:
            rows = [[t.data_ptr(), t.numel() * t.element_size()] for t in (self.gflat, dw32) if t.numel()]
            # (of a depthwise layer's slots only the first NREP: what a launch without slot support adds into)
            rows += [[v.data_ptr(), NREP * self.p[k].numel() * 4] for k, v in self._dwviews.items()]
            assert all(r[1] % 16 == 0 for r in rows)
            self._zero_rows = rows
        self._zero_group('bwd', self._zero_rows)      # + the squeeze-excite per-sample sums (`zgroup='bwd'` buffers)
        self._dwpending, self._dwflushed, self._hook_hi = 0, 0, self.gflat.numel()
        dkp = dkp.reshape(B, 18).to(torch.float32).contiguous()
        ncls = self.num_classes
        if ncls > 1:
            assert dlogits is not None
            dlogits = dlogits.to(torch.float32).contiguous()
        else:
            dlogits = None
        F = a.feat_c
        dpre = self._buf('dpre', (B, 18), torch.float32)
        df = self._buf('df', (B, F), torch.float32)
        bnc = self.bns['classifier.1'] if a.classifier else None
        N.call('t3d_head_bwd', N.ptr(sv['f']), sv['fpro'], N.ptr(sv['cats']), N.ptr(self.wreg),
               N.ptr(self.p['cls_fc.1.weight']), N.ptr(sv['mask']), N.ptr(sv['kp']), N.ptr(dkp), N.ptr(dlogits),
               N.ptr(dpre), N.ptr(df), self._bst(bnc) if bnc else None, None, None, None, None, B, F, ncls, st)
        # the heads' weight gradients are leaves: second stream (idle at this point of the step)
        self._wgrad(N.ptr(sv['f']), sv['fpro'], N.ptr(sv['cats']), N.ptr(sv['mask']), N.ptr(dpre), N.ptr(dlogits),
                    N.ptr(self.dwreg), N.ptr(self.dbreg), N.ptr(self.g['cls_fc.1.weight']), N.ptr(self.g['cls_fc.1.bias']),
                    B, F, ncls, entry='t3d_head_bwd_weights')
        dpooled = df
        if a.classifier:
            bb = self._bn_bwd(bnc)
            self._settle_b(bnc)
            yc, pooled = sv['f'], sv['pooled']
            self._wgrad(N.F32, N.ptr(df), N.ptr(yc), bb, N.ptr(pooled), None,
                        N.ptr(self.g['classifier.0.weight']), B, 1, a.last_c, a.classifier)
            # bias gradient = sum_b dy = alpha*sum(dz) + beta*sum(y) + B*gamma (exactly 0 in exact arithmetic)
            N.call('t3d_bn_bias_grad', N.ptr(bnc.stats), self._bst(bnc), a.classifier, float(B), N.ptr(bnc.alpha),
                   N.ptr(bnc.bbeta), N.ptr(bnc.gammac), N.ptr(self.g['classifier.0.bias']), st)
            dpooled = self._buf('dpooled', (B, a.last_c), torch.float32)
            N.call('t3d_pwconv_dgrad', N.F32, N.ptr(df), N.ptr(yc), bb, N.ptr(self.wt['classifier']), None, None,
                   None, N.ptr(dpooled), None, None, B, 1, a.last_c, a.classifier, st)
        self._backward_backbone(sv, dpooled, dw32)
        self._flush_dw()
        self._join_side()
        self._maybe_hook(0, force=True)
        assert not any(b.pend[1] for b in self.bns.values()), 'a BatchNorm backward finalize was never run'
        self.saved = None

    def _backward_backbone(self, sv, dpooled, dw32):
        """From the gradient at the pooled feature vector back to the stem (MobileNet layouts; models/resnet.py overrides)."""
        a, st, dt, B = self.arch, N.stream(), self.dt, sv['B']
        # ---- pool + last conv
        ln = a.last_name
        bnl = self.bns[ln + '.1']
        x = sv['last_in']
        M, HW = B * sv['HWl'], sv['HWl']
        dzl = self._buf('dz:last', (M, a.last_c))
        N.call('t3d_pool_bwd', dt, N.ptr(dpooled), N.ptr(sv['yl']), sv['prol'], self.pool, N.ptr(sv['pool_argmax']),
               N.ptr(dzl), self._bst(bnl), B, HW, a.last_c, st)
        bb = self._bn_bwd(bnl)
        # weight gradient (second stream) ahead of the data gradient: both derive the BatchNorm-backward coefficients for
        # themselves, the data gradient publishes them (issued the other way round the step is 0.27 ms slower)
        def last_wgrad():
            self._wgrad(dt, N.ptr(dzl), N.ptr(sv['yl']), bb, N.ptr(x.t), x.pro,
                        N.ptr(self.g[ln + '.0.weight']), M, HW, x.C, a.last_c, nbytes=M * (x.C + a.last_c) * self.esz,
                        ro=bnl)
        last_wgrad()
        dz = self._pw_dgrad(dzl, sv['yl'], bb, self.wt[ln + '.0.weight'], x, None, M, HW, x.C, a.last_c, 'dz:lastin', bnl)

        self._maybe_hook(self.offsets[ln + '.0.weight'][0])
        for rec in reversed(sv['blocks']):
            dz = self._block_bwd(rec, dz)
            self._maybe_hook(self.offsets[f"features.{rec['idx'] + 1}.conv.0.weight"][0])

        # ---- stem weight gradient
        s0 = sv['stem']
        bn0 = s0.bn
        bb = self._bn_bwd(bn0)
        self._settle_b(bn0)
        M = s0.B * s0.H * s0.W
        self._wgrad(dt, N.ptr(dz), N.ptr(s0.raw), bb, N.ptr(sv['col']), None, N.ptr(dw32),
                    M, s0.H * s0.W, 32, a.stem_c, nbytes=M * (32 + a.stem_c) * self.esz)
        self._flush_dw()
        self._join_side()
        N.call('t3d_copy_cols', N.ptr(dw32), N.ptr(self.g['features.0.0.weight']), a.stem_c, 32, 27, st)

    def _pw_dgrad(self, dz, y, bb, wt, x, residual, M, HW, K, Nn, tag, bn=None):
        """Data gradient of a 1x1 conv into its input `x` (a _Src): returns the gradient at the BatchNorm output
        of x's producer, with that BatchNorm's backward sums accumulated."""
        dx = self._buf(tag, (M, K))
        with_stats = x.bn is not None and not x.finished_act
        wd, wp = self._wsel(wt, not bb.per_sample and not (with_stats and x.gpro is not None and x.gpro.se))
        self._c('t3d_pwconv_dgrad', wd, N.ptr(dz), N.ptr(y), bb, wp,
                N.ptr(x.raw) if with_stats else None, x.gpro if with_stats else None,
                N.ptr(residual) if residual is not None else None, N.ptr(dx),
                self._bst(x.bn) if with_stats else None, None, M, HW, K, Nn, N.stream(),
                nbytes=M * (K + Nn) * self.esz, bwd=bn)
        if x.finished_act:
            dx = self._act_bwd(dx, x, tag + ':a')
        return dx

    # ---- depthwise weight-gradient replicas: one zero fill per step, one batched sum per gradient bucket
    def _dw_arena_init(self):
        """One SLOT per workgroup for every depthwise weight gradient (include/t3d.h: t3d_set_dw_slots): the kernels store
        their partials, `_flush_dw` adds a layer's used slots in index order -- no atomics, bit-reproducible.  Only the first
        NREP slots of a layer are zeroed per step (what a kernel without slot support adds into)."""
        names = [k for k, (sh, kind) in self.shapes.items() if kind == 'param' and len(sh) == 4 and sh[1] == 1 and sh[2] > 1]
        tot = sum(DW_SLOTS * self.p[k].numel() for k in names)
        self._dwarena = torch.zeros(tot, device=self.device, dtype=torch.float32)
        self._dwused = torch.zeros(max(len(names), 1), device=self.device, dtype=torch.int32)
        self._dwviews, self._dwidx, off = {}, {}, 0
        for k in names:
            n = self.p[k].numel()
            self._dwviews[k] = self._dwarena[off:off + DW_SLOTS * n]
            off += DW_SLOTS * n
        # descriptor rows in BACKWARD order (the order the layers finish): a flush is a contiguous row range
        self._dworder = list(reversed(names))
        rows = []
        for i, k in enumerate(self._dworder):
            self._dwidx[k] = i
            rows.append([self._dwviews[k].data_ptr(), self.g[k].data_ptr(), self.p[k].numel(), self._dwused[i:i + 1].data_ptr()])
        self._dwdesc = torch.tensor(rows, dtype=torch.int64, device=self.device) if rows else None

    def _dw_replicas(self, name):
        assert self._dworder[self._dwflushed + self._dwpending] == name
        self._dwpending += 1
        i = self._dwidx[name]
        N.call('t3d_set_dw_slots', DW_SLOTS, N.ptr(self._dwused[i:i + 1]))      # for the t3d_dwconv_bwd launch that follows
        return self._dwviews[name]

    def _flush_dw(self):
        """Slot sums -> gradient buffer for every depthwise layer finished since the last flush (one launch)."""
        if not self._dwpending:
            return
        N.call('t3d_sum_slots_batched', self._dwdesc[self._dwflushed:].data_ptr(), self._dwpending, N.stream())
        self._dwflushed += self._dwpending
        self._dwpending = 0

    def _maybe_hook(self, lo, force=False):
        """Tell the gradient exchange that everything at flat offsets >= lo is final -- in buckets of >= HOOK_MIN
        elements, because each call first has to wait for the side stream (weight gradients) and flush the depthwise
        replica sums."""
        if self.grad_hook is None:
            return
        if force or self._hook_hi - lo >= HOOK_MIN:
            self._flush_dw()
            if self._side is not None and HOOK_ON_SIDE and not force:      # (the last bucket: nothing left to overlap)
                # the exchange has to see the weight gradients of the SECOND stream and the sums / BatchNorm gradients of the
                # main one: issue it with the second stream current, after that stream has been made to wait for the main
                # one -- the collective's own stream then waits for both, and the main stream waits for nobody (joining the
                # streams here, as rounds 1-3a did, stalled the data-gradient chain three or four times per step until the
                # weight-gradient backlog had drained)
                self._fork_side()
                self._hook(lo, True)
                self._side_busy = True
            else:
                self._join_side()
                self._hook(lo, False)
            self._hook_hi = lo

    def _hook(self, lo, on_side):
        """Gradient-exchange callback (host code: an RCCL collective issued through torch.distributed) with the second or
        the main stream current.  A step plan being recorded is cut into segments here: the replay runs a segment, makes
        this call, runs the next."""
        if N.recorder is not None:
            N.recorder.host_break(('hook', lo, on_side))
        if on_side:
            with torch.cuda.stream(self._side):
                self.grad_hook(lo)
        else:
            self.grad_hook(lo)

    def _yfree_ok(self, x, M, K, Nn):
        """Expand layer on a finished bf16 input, wide enough that skipping the two extra passes over the M x N tensors
        pays for the three tiny extra launches (csrc/pwconv_yfree.hip)."""
        return (self.dt == N.BF16 and YFREE_MIN_ELEMS > 0 and x.pro is None and K <= YFREE_MAX_K
                and M * Nn >= YFREE_MIN_ELEMS)

    def _expand_bwd_yfree(self, d1, bb1, bn1, wname, x, res, M, K, Nn, i):
        """Backward of the expand conv without reading its output: dx = [d1 | x] Wcat^T + c on the main stream,
        dW from [d1 | x | 1]^T x on the side stream."""
        st = N.stream()
        tot = (Nn + 31) // 32 * 32 + (K + 31) // 32 * 32
        wcat = self._buf(f'wcat:{i}', (K, tot))
        cvec = self._buf(f'cvec:{i}', (K,), torch.float32)
        HW = x.H * x.W
        with_stats = x.bn is not None and not x.finished_act       # as _pw_dgrad
        # ONE pass over the wide gradient for both products (csrc/pwconv_wgrad_tr.hip, DGF): the data gradient and the partial
        # tiles of the weight-gradient products come out of the same staged rows on the main stream; the second stream only
        # reduces and combines.  (The pair below reads d1 twice, from two streams at the same time.)
        need = N.lib().t3d_pwconv_bwd_yfree_scratch(M, K, Nn) if (YFREE_FUSED and not (with_stats and x.gpro is not None)) else 0
        if need > 0:
            scratch = self._buf(f'yfscr:{i}', (need,), torch.uint8)
            dx = self._buf(f'dzin:{i}', (M, K))
            if YFREE_PREP_FUSED:
                # the data gradient's weight rows are built in the launch's own prologue (and the BatchNorm-backward finalize
                # derived there): no t3d_pwconv_yfree_prep2 launch on the critical stream (round 5: -13 us x 6 per step)
                self._c('t3d_pwconv_bwd_yfree_w', N.ptr(d1), N.ptr(x.t), N.ptr(self.wt[wname]), bb1, N.ptr(x.raw) if with_stats else None,
                        None, N.ptr(res) if res is not None else None, N.ptr(dx), self._bst(x.bn) if with_stats else None,
                        N.ptr(scratch), need, M, HW, K, Nn, st, nbytes=M * (K + Nn) * self.esz, bwd=bn1)
            else:
                key = (f'wd:{i}', ((K + 15) // 16 * 16, (Nn + K + 8 + 63) // 64 * 64), self.dtype)
                wd = self._bufs.get(key) if key in self._bufs else self._buf(key[0], key[1], zero=True)   # cleared ONCE: prep2 writes the non-zero entries
                self._c('t3d_pwconv_yfree_prep2', N.ptr(self.wt[wname]), bb1, N.ptr(wcat), N.ptr(cvec), N.ptr(wd), K, Nn, st, bwd=bn1)
                N.call('t3d_pwconv_bwd_yfree', N.ptr(d1), N.ptr(x.t), N.ptr(wd), N.ptr(x.raw) if with_stats else None, None,
                       N.ptr(res) if res is not None else None, N.ptr(dx), self._bst(x.bn) if with_stats else None, N.ptr(scratch), need,
                       M, HW, K, Nn, st, nbytes=M * (K + Nn) * self.esz)
            self._wgrad(N.ptr(scratch), bb1, N.ptr(self.w[wname]), N.ptr(self.g[wname]), M, K, Nn, entry='t3d_pwconv_wgrad_yfree_finish')
            if x.finished_act:
                dx = self._act_bwd(dx, x, f'dzin:{i}:a')
            return dx
        self._c('t3d_pwconv_yfree_prep', N.ptr(self.wt[wname]), bb1, N.ptr(wcat), N.ptr(cvec), K, Nn, st, bwd=bn1)
        self._wgrad(N.ptr(d1), N.ptr(x.t), bb1, N.ptr(self.w[wname]), N.ptr(self.g[wname]), M, HW, K, Nn,
                    entry='t3d_pwconv_wgrad_yfree', nbytes=M * (K + Nn) * self.esz)
        dx = self._buf(f'dzin:{i}', (M, K))
        N.call('t3d_pwconv_dgrad_yfree', N.ptr(d1), N.ptr(x.t), N.ptr(wcat), N.ptr(cvec),
               N.ptr(x.raw) if with_stats else None, x.gpro if with_stats else None,
               N.ptr(res) if res is not None else None, N.ptr(dx),
               self._bst(x.bn) if with_stats else None, M, HW, K, Nn, st, nbytes=M * (K + Nn) * self.esz)
        if x.finished_act:
            dx = self._act_bwd(dx, x, f'dzin:{i}:a')
        return dx

    def _act_bwd(self, dz, x, tag):
        M = x.B * x.H * x.W
        out = self._buf(tag, (M, x.C))
        N.call('t3d_bn_act_bwd', self.dt, N.ptr(dz), N.ptr(x.raw), x.gpro, N.ptr(out), self._bst(x.bn), M, x.C,
               N.stream())
        return out

    def _block_bwd(self, rec, dz):
        """dz: gradient w.r.t. the block output z (= at the output of the projection BatchNorm)."""
        st, dt = N.stream(), self.dt
        blk, i, x, src, s2 = rec['blk'], rec['idx'], rec['x'], rec['src'], rec['s2']
        dwn, pwn = rec['names']
        B = x.B
        M2, HW2 = s2.B * s2.H * s2.W, s2.H * s2.W
        bb3 = self._bn_bwd(rec['bn3'])
        se = rec.get('se')
        if se is not None and not SE_FOLD:
            self._settle_b(rec['bn3'])

        def proj_wgrad():
            self._wgrad(dt, N.ptr(dz), N.ptr(rec['y3']), bb3, N.ptr(s2.t), s2.pro, N.ptr(self.g[pwn]),
                        M2, HW2, blk.cexp, blk.cout, nbytes=M2 * (blk.cexp + blk.cout) * self.esz, ro=rec['bn3'])
        if se is None:
            proj_wgrad()
            dv2 = self._pw_dgrad(dz, rec['y3'], bb3, self.wt[pwn], s2, None, M2, HW2, blk.cexp, blk.cout, f'dv2:{i}',
                                 rec['bn3'])
            bb2 = self._bn_bwd(s2.bn)
        elif se.get('after'):
            # gate after the activation: dv = gradient at the gated tensor (plain data gradient); the gate's gradient
            # needs sum_hw dv*a, the pooled path adds g to every pixel BEFORE the activation derivative
            C, R, sen = blk.cexp, blk.se, se['name']
            bn2 = s2.bn
            proj_wgrad()
            dv = self._buf(f'dv2g:{i}', (M2, C))
            wd, wp = self._wsel(self.wt[pwn])
            self._c('t3d_pwconv_dgrad', wd, N.ptr(dz), N.ptr(rec['y3']), bb3, wp, None, None, None,
                    N.ptr(dv), None, None, M2, HW2, C, blk.cout, st, nbytes=M2 * (C + blk.cout) * self.esz, bwd=rec['bn3'])
            ps = self._buf(f'se_ps:{i}', (B, C, 2), torch.float32)
            N.call('t3d_se_after_sums', dt, N.ptr(dv), N.ptr(s2.raw), se['pro2n'], N.ptr(ps), B, HW2, C, st)
            g = self._buf(f'se_g:{i}', (B, C), torch.float32)
            dq = self._buf(f'se_dq:{i}', (B, C), torch.float32)
            dp = self._buf(f'se_dp:{i}', (B, R), torch.float32)
            ones, zeros = self._const(C, 1.0), self._const(C, 0.0)
            self._pool_exact(False)
            N.call('t3d_se_bwd_data', N.ptr(ps), N.ptr(se['gap']), N.ptr(zeros), N.ptr(ones),
                   N.ptr(self.p[sen + '.fc.0.weight']), N.ptr(self.p[sen + '.fc.2.weight']),
                   N.ptr(se['h']), N.ptr(se['q']), N.ptr(se['s']), N.ptr(g), N.ptr(dq), N.ptr(dp), None,
                   B, C, R, se['HW'], st)            # (the before-activation sums are not needed here)
            self._se_wgrad(se, sen, dq, dp, B, C, R)
            dv2 = self._buf(f'dv2:{i}', (M2, C))
            N.call('t3d_se_after_apply', dt, N.ptr(dv), N.ptr(s2.raw), se['pro2n'], N.ptr(se['s']), N.ptr(g), N.ptr(dv2),
                   self._bst(bn2), B, HW2, C, st)
            bb2 = self._bn_bwd(bn2)
        else:
            # gated tensor: the data gradient reports per-SAMPLE sums; the gate's backward turns them into the
            # BatchNorm sums and the per-sample affine  dy = (alpha*s) dv + beta y + (gamma + alpha*g)
            C, R, sen = blk.cexp, blk.se, se['name']
            proj_wgrad()
            ps = self._buf(f'se_ps:{i}', (B, C, 2), torch.float32, zgroup='bwd')
            dv2 = self._buf(f'dv2:{i}', (M2, C))
            wd, wp = self._wsel(self.wt[pwn])
            self._c('t3d_pwconv_dgrad', wd, N.ptr(dz), N.ptr(rec['y3']), bb3, wp, N.ptr(s2.raw), s2.gpro,
                    None, N.ptr(dv2), None, N.ptr(ps), M2, HW2, C, blk.cout, st, nbytes=M2 * (C + blk.cout) * self.esz,
                    bwd=rec['bn3'])
            g = self._buf(f'se_g:{i}', (B, C), torch.float32)
            dq = self._buf(f'se_dq:{i}', (B, C), torch.float32)
            dp = self._buf(f'se_dp:{i}', (B, R), torch.float32)
            bn2 = s2.bn
            self._pool_exact(True)
            N.call('t3d_se_bwd_data', N.ptr(ps), N.ptr(se['gap']), N.ptr(bn2.scale), N.ptr(bn2.shift),
                   N.ptr(self.p[sen + '.fc.0.weight']), N.ptr(self.p[sen + '.fc.2.weight']),
                   N.ptr(se['h']), N.ptr(se['q']), N.ptr(se['s']), N.ptr(g), N.ptr(dq), N.ptr(dp), self._bst(bn2),
                   B, C, R, se['HW'], st)
            self._se_wgrad(se, sen, dq, dp, B, C, R)
            self._bn_bwd(bn2)
            if not SE_FOLD:
                self._settle_b(bn2)
            aps = self._buf(f'se_aps:{i}', (B, C), torch.float32)
            gps = self._buf(f'se_gps:{i}', (B, C), torch.float32)
            self._c('t3d_se_bwd_affine', N.ptr(se['s']), N.ptr(g), N.ptr(bn2.alpha), N.ptr(bn2.gammac), N.ptr(aps),
                    N.ptr(gps), B, C, st, bwd=bn2)
            bb2 = N.bnbwd(aps, bn2.bbeta, gps, True)
        M1 = B * x.H * x.W
        res = dz if blk.res else None
        if blk.expand:
            s1 = rec['s1']
            d1 = self._buf(f'dz1:{i}', (M1, blk.cexp))
            dwrep = self._dw_replicas(dwn)
            self._c('t3d_dwconv_bwd', dt, N.ptr(dv2), N.ptr(s2.raw), bb2, N.ptr(self.p[dwn]), N.ptr(s1.t), s1.pro, None,
                    N.ptr(d1), self._bst(s1.bn), N.ptr(dwrep), B, x.H, x.W, blk.cexp, blk.k, blk.s, st,
                    nbytes=2 * (M1 + M2) * blk.cexp * self.esz, bwd=None if bb2.per_sample else s2.bn)
            bb1 = self._bn_bwd(s1.bn)
            p = f'features.{i + 1}.conv'
            if self._yfree_ok(x, M1, blk.cin, blk.cexp):
                return self._expand_bwd_yfree(d1, bb1, s1.bn, p + '.0.weight', x, res, M1, blk.cin, blk.cexp, i)
            def exp_wgrad():
                self._wgrad(dt, N.ptr(d1), N.ptr(s1.raw), bb1, N.ptr(x.t), x.pro,
                            N.ptr(self.g[p + '.0.weight']), M1, x.H * x.W, blk.cin, blk.cexp,
                            nbytes=M1 * (blk.cin + blk.cexp) * self.esz, ro=s1.bn)
            exp_wgrad()
            return self._pw_dgrad(d1, s1.raw, bb1, self.wt[p + '.0.weight'], x, res, M1, x.H * x.W, blk.cin,
                                  blk.cexp, f'dzin:{i}', s1.bn)
        # no-expand layout: the depthwise conv reads the block input directly
        dx = self._buf(f'dzin:{i}', (M1, blk.cexp))
        dwrep = self._dw_replicas(dwn)
        deferred = x.pro is not None           # raw producer tensor read through its prologue
        self._c('t3d_dwconv_bwd', dt, N.ptr(dv2), N.ptr(s2.raw), bb2, N.ptr(self.p[dwn]), N.ptr(x.t), x.pro,
                N.ptr(res) if res is not None else None, N.ptr(dx),
                self._bst(x.bn) if deferred else None, N.ptr(dwrep), B, x.H, x.W, blk.cexp, blk.k, blk.s, st,
                nbytes=2 * (M1 + M2) * blk.cexp * self.esz, bwd=None if bb2.per_sample else s2.bn)
        if not deferred:
            # finished input: the producer's BatchNorm sums have to be taken against its RAW tensor
            dx = self._act_bwd(dx, x, f'dzin:{i}:a')
        return dx
