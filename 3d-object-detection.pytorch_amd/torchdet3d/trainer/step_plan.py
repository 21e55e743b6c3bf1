"""The train iteration of torchdet3d/trainer/train.py:44-55 (forward, losses, `loss.backward()`, `optimizer.step()`) without
the per-launch host loop.

Three forms of the same step, bit-identical to each other (tests/test_gpu_step_plan.py):

  eager    `Trainer.train_step`'s reference-shaped sequence through autograd (`model(...)`, `LossManager.parse_losses`,
           `loss.backward()`, `optimizer.step()`): what runs for anything this module does not cover (user criterions, ALWA,
           a framework optimizer, host-side inputs);
  direct   the same entry points issued from Python in step order WITHOUT the autograd glue: the loss launch hands its
           gradients straight to `engine.backward`, the optimizer kernel reads the engine's gradient buffer -- no framework
           kernel (fill / clone / multiply by the upstream gradient / gradient copy) is left inside the step;
  replay   the direct step recorded once into a `t3d_plan` (csrc/plan.hip) and replayed by ONE call per step
           (`t3d_plan_run`): ~230 launches, the ~35 stream forks of the weight-gradient stream, the metrics read-back and
           its event, with the per-step values (input pointers, dropout counter, optimizer step count, learning rate,
           read-back slot) passed as slots.  Under a multi-rank launch the plan is cut into segments at the gradient
           exchange's callbacks (RCCL collectives are issued by torch.distributed, i.e. from Python).

`StepPlan.run` walks eager-equivalent `direct` steps first (lazily created descriptors, optimizer state), records the
third one and replays from then on; anything that would change the recorded launch list (another batch shape, input dtype,
loss configuration, optimizer hyper-parameter, current stream, seed, a model moved to another device) re-records.
"""
import ctypes
import os

import torch

from .. import _native as N

WARM_STEPS = 2          # direct steps in front of the recorded one
REPLAY = os.environ.get('T3D_STEP_PLAN', '1') != '0'      # 0: stay in the direct form (A/B switch, debugging)


class StepPlan:
    def __init__(self, model, loss_manager, optimizer):
        self.model, self.lm, self.opt = model, loss_manager, optimizer
        self.rec = None            # PlanRecorder of the recorded step
        self.key = None
        self.warm = 0
        self.slots = (ctypes.c_ulonglong * N.NSLOTS)()
        self.replays = 0

    # ------------------------------------------------------------------ applicability
    @staticmethod
    def usable(model, lm, opt):
        from ..builders.model_builder import ModelWrapper
        from ..builders.optim_builder import FusedAdamW
        if not isinstance(model, ModelWrapper) or model.export_mode or model.net.device.type != 'cuda':
            return False
        if lm is None or not getattr(lm, '_fused', False) or lm.use_alwa:
            return False            # (ALWA reads loss values back on the host every C-th iteration: the eager form does that)
        if not isinstance(opt, FusedAdamW) or len(opt.param_groups) != 1:
            return False
        ps = opt.param_groups[0]['params']
        return len(ps) == 1 and ps[0] is model.flat

    def accepts(self, imgs, gt_kp, cats):
        m = self.model
        if not (m.training and torch.is_grad_enabled() and N.timer is None):
            return False
        B = imgs.shape[0]
        return (imgs.is_cuda and imgs.is_contiguous() and imgs.dtype in (torch.float32, torch.uint8) and imgs.dim() == 4
                and gt_kp.is_cuda and gt_kp.dtype == torch.float32 and gt_kp.is_contiguous() and gt_kp.numel() == B * 18
                and cats.is_cuda and cats.dtype == torch.int64 and cats.is_contiguous() and cats.numel() == B
                and imgs.device == m.net.device)

    def _key(self, imgs):
        net, g = self.model.net, self.opt.param_groups[0]
        st = self._state()
        return (id(net), tuple(imgs.shape), imgs.dtype, N.stream(), id(net._side), bytes(self.lm.loss_cfg()),
                tuple(g['betas']), g['eps'], g['weight_decay'], self.opt.grad_scale, torch.initial_seed(),
                id(net.grad_hook), net.num_classes,
                st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr())

    def _state(self):
        """The optimizer's state of the flat parameter, created as FusedAdamW.step creates it (builders/optim_builder.py)."""
        p = self.model.flat
        state = self.opt.state[p]
        if not state:
            state['step'] = 0
            state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return state

    # ------------------------------------------------------------------ the step
    def run(self, imgs, gt_kp, cats, slot):
        """One iteration on device tensors; `slot` = [pinned fp32 [16], torch.cuda.Event (recorded once), owner] of the
        caller's read-back ring.  Returns the device vector of the loss launch (total, reg, cls, ADD, SADD, acc, ...)
        and whether the read-back into `slot` has already been enqueued (replay) or is the caller's to issue."""
        key = self._key(imgs)
        if key != self.key:
            self._drop()
            self.key, self.warm = key, 0
        if self.rec is not None and REPLAY:
            return self._replay(imgs, gt_kp, cats, slot), True
        record = REPLAY and self.warm >= WARM_STEPS
        out = self._direct(imgs, gt_kp, cats, record)
        self.warm += 1
        return out, False

    def _drop(self):
        if self.rec is not None:
            self.rec.close()
            self.rec = None

    def __del__(self):
        try:
            self._drop()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

    def _bufs(self, B):
        net = self.model.net
        ncls = net.num_classes
        out = net._buf('loss:out', (16,), torch.float32)
        dkp = net._buf('loss:dkp', (B, 18), torch.float32)
        dlg = net._buf('loss:dlg', (B, ncls), torch.float32, zero=('loss:dlg', (B, ncls), torch.float32) not in net._bufs) if ncls > 1 else None
        return out, dkp, dlg

    def _direct(self, imgs, gt_kp, cats, record):
        """The step as a plain sequence of entry-point calls.  record: also record it into a plan while it runs."""
        model, lm, opt = self.model, self.lm, self.opt
        net = model.net
        B, st = imgs.shape[0], N.stream()
        rec = None
        if record:
            rec = N.PlanRecorder({imgs.data_ptr(): N.SLOT_IMGS, gt_kp.data_ptr(): N.SLOT_GT, cats.data_ptr(): N.SLOT_CATS})
            if len({imgs.data_ptr(), gt_kp.data_ptr(), cats.data_ptr()}) != 3:
                rec = None                                   # (aliased inputs: nothing to tell apart by address)
        N.recorder = rec
        try:
            net.persistent_outputs = True
            try:
                kp, logits = net.forward(imgs, cats, train=True)
            finally:
                net.persistent_outputs = False
            out, dkp, dlg = self._bufs(B)
            # the class criterions see the logits only when there are any (regression_losses.py:84-88)
            lg = logits if (lm.class_criterions and logits is not None) else None
            ncls = lg.shape[1] if lg is not None else 1
            N.call('t3d_loss_fwd_bwd', lm.loss_cfg(), N.ptr(kp), N.ptr(gt_kp), N.ptr(lg), N.ptr(cats), N.ptr(out), N.ptr(dkp),
                   N.ptr(dlg) if lg is not None else None, B, ncls, st)
            # the metrics' way to the host (replayed steps): copied by the SECOND stream, handed off by the loss kernel itself --
            # the copy and its event packet then ride beside the backward instead of sitting in the main queue between the
            # optimizer kernel and the next step (a ~14-us bubble in front of every step's first kernels)
            rb_side = rec is not None and net._side is not None
            if rb_side:
                rec.add_fork(st, net._side.cuda_stream)
                rec.add_readback(N.SLOT_RB_DST, out, 64, N.SLOT_RB_EVENT, net._side.cuda_stream)
            sync = model.grad_sync
            if sync is not None:
                sync.start()
            net.backward(dkp, dlg)
            if sync is not None:
                if rec is not None:
                    rec.host_break(('finish',))
                sync.finish()
            # optimizer (builders/optim_builder.py: FusedAdamW.step, one parameter)
            g = opt.param_groups[0]
            p = model.flat
            state = self._state()
            state['step'] += 1
            b1, b2 = g['betas']
            N.call('t3d_set_grad_watch', N.ptr(opt.watch_word(p.device)))
            N.call('t3d_adamw_step', N.ptr(p), N.ptr(net.gflat), N.ptr(state['exp_avg']), N.ptr(state['exp_avg_sq']), p.numel(),
                   float(g['lr']), float(b1), float(b2), float(g['eps']), float(g['weight_decay']), state['step'],
                   float(opt.grad_scale), st, slots={5: N.SLOT_LR, 10: N.SLOT_STEP})
            N.call('t3d_set_grad_watch', None)
            torch.autograd.graph.increment_version(p)
            if p.grad is not net.gflat:
                p.grad = net.gflat                           # what `loss.backward()` leaves in the eager form
            if rec is not None:
                if not rb_side:
                    rec.add_readback(N.SLOT_RB_DST, out, 64, N.SLOT_RB_EVENT, st)
                rec.end_segment()
                rec.keep += [out, dkp, dlg, kp, logits, state['exp_avg'], state['exp_avg_sq']]
        except BaseException:
            if rec is not None:
                rec.close()                              # (an exception while recording must not leak the plan)
            raise
        finally:
            N.recorder = None
        lm.last = out
        if rec is not None:
            if rec.broken:
                rec.close()                              # stays in the direct form
                self.warm = -(1 << 30)
            else:
                self.rec = rec
        return out

    def _replay(self, imgs, gt_kp, cats, slot):
        model, opt, rec = self.model, self.opt, self.rec
        net = model.net
        g, p = opt.param_groups[0], model.flat
        state = opt.state[p]
        state['step'] += 1
        net._dropout_calls = getattr(net, '_dropout_calls', 0) + 1
        s = self.slots
        s[N.SLOT_IMGS], s[N.SLOT_GT], s[N.SLOT_CATS] = imgs.data_ptr(), gt_kp.data_ptr(), cats.data_ptr()
        s[N.SLOT_DROPOUT], s[N.SLOT_STEP], s[N.SLOT_LR] = net._dropout_calls, state['step'], N.double_bits(g['lr'])
        s[N.SLOT_RB_DST], s[N.SLOT_RB_EVENT] = slot[0].data_ptr(), slot[1].cuda_event
        lib, plan = N.lib(), rec.plan
        ev = self.timing
        sync = model.grad_sync
        if sync is not None:
            sync.start()
        nseg = len(rec.breaks) + 1
        for i in range(nseg):
            if ev is None:
                rc = lib.t3d_plan_run(plan, i, s, N.NSLOTS, None, 0)
            else:
                rc = ev.run(lib, plan, i, nseg, s)
            if rc < 0:
                code = ctypes.c_int(0)
                op = lib.t3d_plan_failed_op(plan, ctypes.byref(code))
                raise RuntimeError(f't3d_plan_run failed at op {op} with code {code.value}')
            if i < nseg - 1:
                b = rec.breaks[i]
                if b[0] == 'hook':
                    if b[2]:
                        with torch.cuda.stream(net._side):
                            net.grad_hook(b[1])
                    else:
                        net.grad_hook(b[1])
                else:
                    sync.finish()
        torch.autograd.graph.increment_version(p)
        net.saved = None
        self.replays += 1
        out = net._buf('loss:out', (16,), torch.float32)
        self.lm.last = out
        return out

    # optional kernel-exact timing of selected entry points inside replayed steps (bench.py's roofline block)
    timing = None


class PlanTiming:
    """Attaches HIP-event pairs to the launches of selected entry points in replayed steps (include/t3d.h:
    t3d_plan_time_entry / t3d_plan_run's `events`): [(name, int-args signature, ms, algorithmic bytes)] afterwards, like
    _native.KernelTimer.  `entry_sets`: {label: entry-point names}; `select(label)` picks the set the NEXT steps are timed with
    (None: untimed) -- bench.py times the depthwise family on every 4th step and every convolution family on a few."""

    def __init__(self, step_plan, entry_sets, max_events):
        self.sp = step_plan
        if not isinstance(entry_sets, dict):
            entry_sets = {'default': tuple(entry_sets)}
        self.sets = {k: tuple(v) for k, v in entry_sets.items()}
        self.calls = step_plan.rec.calls
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * max_events)]
        for e in self.pool:
            e.record()
        self.used = []             # (label, events) per timed step
        self.on = set()            # entry points currently switched on in the plan
        self.label, self.idx, self.per_step = None, [], 0
        self.cur = None

    @property
    def active(self):
        return self.label is not None

    @active.setter
    def active(self, v):           # (the single-set form: timing.active = True / False)
        self.select(next(iter(self.sets)) if v else None)

    def select(self, label):
        if label == self.label:
            return
        want = set(self.sets[label]) if label is not None else set()
        plan = self.sp.rec.plan
        for n in want - self.on:
            N.lib().t3d_plan_time_entry(plan, n.encode(), 1)
        for n in self.on - want:
            N.lib().t3d_plan_time_entry(plan, n.encode(), 0)
        self.on, self.label = want, label
        self.idx = [i for i, c in enumerate(self.calls) if c[0] in want]
        self.per_step = 2 * len(self.idx)

    def run(self, lib, plan, seg, nseg, slots):
        if seg == 0:
            self.cur = None
            if self.label is not None and self.per_step and len(self.pool) >= self.per_step:
                evs = [self.pool.pop() for _ in range(self.per_step)]
                self.cur = [evs, (ctypes.c_void_p * self.per_step)(*[e.cuda_event for e in evs]), 0, self.label, self.idx]
        if self.cur is None:
            return lib.t3d_plan_run(plan, seg, slots, N.NSLOTS, None, 0)
        evs, arr, off = self.cur[:3]
        tail = ctypes.cast(ctypes.byref(arr, off * ctypes.sizeof(ctypes.c_void_p)), ctypes.POINTER(ctypes.c_void_p))
        rc = lib.t3d_plan_run(plan, seg, slots, N.NSLOTS, tail, len(evs) - off)
        if rc >= 0:
            self.cur[2] = off + rc
            if seg == nseg - 1:
                self.used.append((self.cur[3], self.cur[4], evs[:self.cur[2]]))
        return rc

    def per_launch(self, label=None):
        torch.cuda.synchronize()
        out = []
        for lab, idx, evs in self.used:
            if label is not None and lab != label:
                continue
            for j, ci in enumerate(idx[:len(evs) // 2]):
                name, sig, nb = self.calls[ci]
                out.append((name, sig, evs[2 * j].elapsed_time(evs[2 * j + 1]), nb))
        return out

    def steps(self, label):
        return sum(1 for lab, _, _ in self.used if lab == label)

    def close(self):
        self.select(None)


class ForwardPlan:
    """The eval-mode forward of `ModelWrapper.forward` (validation, serving: model_builder.py:126-146 under `model.eval()`) as ONE
    host call: the engine's ~110 forward launches recorded once per (input shape, dtype, stream) and replayed by
    `t3d_plan_run` with the batch's two device pointers as slots.  The replay writes kp / logits into engine buffers; the
    caller gets fresh tensors (two small device copies), as from the launch-by-launch forward.  Bit-identical to it
    (tests/test_gpu_step_plan.py)."""

    def __init__(self, net):
        self.net = net
        self.rec, self.key, self.warm = None, None, 0
        self.slots = (ctypes.c_ulonglong * N.NSLOTS)()
        self.replays = 0

    def accepts(self, x, cats):
        B = x.shape[0]
        return (REPLAY and N.timer is None and x.is_cuda and x.is_contiguous() and x.dtype in (torch.float32, torch.uint8) and x.dim() == 4
                and cats is not None and cats.is_cuda and cats.dtype == torch.int64 and cats.is_contiguous() and cats.numel() == B
                and x.device == self.net.device and x.data_ptr() != cats.data_ptr())

    def __call__(self, x, cats):
        net = self.net
        key = (tuple(x.shape), x.dtype, N.stream(), id(net._side))
        if key != self.key:
            self.drop()
            self.key, self.warm = key, 0
        net._pack()                                  # (weights moved since the last forward: re-packed outside the plan)
        if self.rec is not None:
            s = self.slots
            s[N.SLOT_IMGS], s[N.SLOT_CATS] = x.data_ptr(), cats.data_ptr()
            rc = N.lib().t3d_plan_run(self.rec.plan, 0, s, N.NSLOTS, None, 0)
            if rc < 0:
                code = ctypes.c_int(0)
                op = N.lib().t3d_plan_failed_op(self.rec.plan, ctypes.byref(code))
                raise RuntimeError(f't3d_plan_run failed at op {op} with code {code.value}')
            self.replays += 1
            kp, logits = self.out
            net.saved = None
            return kp.clone(), (logits.clone() if logits is not None else None)
        rec = None
        if self.warm >= WARM_STEPS:
            rec = N.PlanRecorder({x.data_ptr(): N.SLOT_IMGS, cats.data_ptr(): N.SLOT_CATS})
        N.recorder = rec
        try:
            net.persistent_outputs = True
            try:
                kp, logits = net.forward(x, cats, train=False)
            finally:
                net.persistent_outputs = False
            if rec is not None:
                rec.end_segment()
                rec.keep += [kp, logits]
        except BaseException:
            if rec is not None:
                rec.close()
            raise
        finally:
            N.recorder = None
        self.warm += 1
        if rec is not None:
            if rec.broken or rec.breaks:
                rec.close()
                self.warm = -(1 << 30)
            else:
                self.rec, self.out = rec, (kp, logits)
        return kp.clone(), (logits.clone() if logits is not None else None)

    def drop(self):
        if self.rec is not None:
            self.rec.close()
            self.rec = None

    def __del__(self):
        try:
            self.drop()
        except Exception:       # noqa: BLE001
            pass

