"""ctypes binding of libt3d_hip.so (the C ABI declared in include/t3d.h).

The product path has NO fallback: if the library is missing or a call fails, a
RuntimeError is raised.  Tensors cross the boundary as raw device pointers.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# T3D_LIB: another build of the same library (A/B timing of one kernel file: tools/ab_lib.sh); the product loads the in-tree one
LIB_PATH = os.environ.get('T3D_LIB') or os.path.join(os.path.dirname(_HERE), 'libt3d_hip.so')

W_FRAG = 0x100                  # include/t3d.h: T3D_W_FRAG (weights in fragment order, bf16 pointwise convs)
F32, BF16, F16 = 0, 1, 2        # include/t3d.h: T3D_F32 / T3D_BF16 / T3D_F16 (fp16: inference forward only)
ACT = {'none': 0, 'relu': 1, 'relu6': 2, 'hswish': 3}
POOL = {'avg': 0, 'max': 1, 'avg+max': 2}
_P, _I, _F, _D, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_longlong


class BnBwd(ctypes.Structure):
    _fields_ = [('alpha', _P), ('beta', _P), ('gamma', _P), ('per_sample', _I)]


class Prologue(ctypes.Structure):
    _fields_ = [('scale', _P), ('shift', _P), ('se', _P), ('act', _I), ('se_after_act', _I)]


class LossCfg(ctypes.Structure):
    _fields_ = [(n, _F) for n in ('c_l1', 'c_mse', 'c_smoothl1', 'smoothl1_beta', 'c_add', 'c_diag', 'c_wing',
                                  'wing_w', 'wing_eps', 'c_ce', 'lam_reg', 'lam_cls')]


class BnFold(ctypes.Structure):          # t3d_bn_fold (include/t3d.h); lives in device memory
    _fields_ = [('kind', _I), ('C', _I), ('nrep', _I), ('rstride', _L), ('stats', _P), ('count', _D), ('gamma', _P), ('beta', _P),
                ('rm', _P), ('rv', _P), ('nbt', _P), ('momentum', _F), ('eps', _F), ('o0', _P), ('o1', _P), ('o2', _P),
                ('o3', _P), ('o4', _P), ('mean', _P), ('invstd', _P)]


_PP = ctypes.POINTER(Prologue)
_BP = ctypes.POINTER(BnBwd)
_LP = ctypes.POINTER(LossCfg)

# name -> argtypes (restype is always int); mirrors include/t3d.h one to one
SIGNATURES = {
    't3d_version': [],
    't3d_dwconv_fwd': [_I, _P, _PP, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    't3d_bn_finalize': [_P, _I, _D, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P],
    't3d_bn_eval_affine': [_I, _P, _P, _P, _P, _F, _P, _P, _P],
    't3d_bn_eval_affine_batched': [_P, _I, _F, _P],
    't3d_pwconv_fwd': [_I, _P, _PP, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    't3d_pwconv_dgrad': [_I, _P, _P, _BP, _P, _P, _PP, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    't3d_pack_weight': [_I, _P, _P, _I, _I, _I, _P],
    't3d_sum_replicas_batched': [_P, _I, _I, _P],
    't3d_set_dw_slots': [_I, _P],
    't3d_set_exact_pool': [_I],
    't3d_set_launch_events': [_P, _P],
    't3d_sum_slots_batched': [_P, _I, _P],
    't3d_dwconv_bwd': [_I, _P, _P, _BP, _P, _P, _PP, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    't3d_pwconv_wgrad': [_I, _P, _P, _BP, _P, _PP, _P, _I, _I, _I, _I, _P],
    't3d_pwconv_yfree_prep': [_P, _BP, _P, _P, _I, _I, _P],
    't3d_pwconv_dgrad_yfree': [_P, _P, _P, _P, _P, _PP, _P, _P, _P, _I, _I, _I, _I, _P],
    't3d_pwconv_wgrad_yfree': [_P, _P, _BP, _P, _P, _I, _I, _I, _I, _P],
    't3d_pwconv_yfree_prep2': [_P, _BP, _P, _P, _P, _I, _I, _P],
    't3d_pwconv_bwd_yfree_scratch': [_I, _I, _I],
    't3d_pwconv_bwd_yfree': [_P, _P, _P, _P, _PP, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    't3d_pwconv_bwd_yfree_w': [_P, _P, _P, _BP, _P, _PP, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    't3d_pwconv_wgrad_yfree_finish': [_P, _BP, _P, _P, _I, _I, _I, _P],
    't3d_bn_bwd_finalize': [_P, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    't3d_stem_im2col': [_I, _P, _P, _I, _I, _I, _P],
    't3d_stem_im2col_u8': [_I, _P, _P, _P, _P, _I, _I, _I, _P],
    't3d_crop_resize_u8': [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    't3d_pwconv_fwd_mat': [_I, _P, _PP, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    't3d_ssd_decode_nms': [_I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _F, _F, _P, _P, _P, _P],
    't3d_im2col': [_I, _P, _PP, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    't3d_im2col_nchw': [_I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    't3d_col2im_bwd': [_I, _P, _P, _PP, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    't3d_pack_conv_weight': [_I, _P, _P, _I, _I, _I, _I, _P],
    't3d_unpack_conv_grad': [_P, _P, _I, _I, _I, _I, _P],
    't3d_maxpool_fwd': [_I, _P, _PP, _P, _P, _I, _I, _I, _I, _P],
    't3d_maxpool_bwd': [_I, _P, _P, _P, _PP, _P, _P, _I, _I, _I, _I, _P],
    't3d_res_relu_fwd': [_I, _P, _PP, _P, _PP, _P, _I, _I, _P],
    't3d_res_relu_bwd': [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    't3d_subsample': [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    't3d_ir_block_eval': [_P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    't3d_bn_apply': [_I, _P, _PP, _P, _P, _I, _I, _P],
    't3d_bn_apply_gram': [_I, _P, _PP, _P, _P, _P, _I, _I, _P],
    't3d_gram_bn_finalize': [_P, _P, _I, _I, _D, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P],
    't3d_bn_act_bwd': [_I, _P, _P, _PP, _P, _P, _I, _I, _P],
    't3d_gap_fwd': [_I, _P, _PP, _P, _I, _I, _I, _P],
    't3d_gap_bwd': [_I, _P, _P, _PP, _P, _P, _I, _I, _I, _P],
    't3d_pool_fwd': [_I, _P, _PP, _I, _P, _P, _I, _I, _I, _P],
    't3d_pool_bwd': [_I, _P, _P, _PP, _I, _P, _P, _P, _I, _I, _I, _P],
    't3d_head_fwd': [_P, _PP, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    't3d_linear_fwd': [_P, _P, _P, _P, _I, _I, _I, _P],
    't3d_head_fwd_all': [_P, _PP, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    't3d_head_bwd': [_P, _PP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    't3d_head_bwd_weights': [_P, _PP] + [_P] * 8 + [_I, _I, _I, _P],
    't3d_se_fwd': [_P] * 11 + [_I, _I, _I, _I, _P],
    't3d_se_bwd': [_P] * 18 + [_I, _I, _I, _I, _P],
    't3d_se_fwd_fused': [_P] * 11 + [_I] * 4 + [_P],
    't3d_se_bwd_data': [_P] * 13 + [_I] * 4 + [_P],
    't3d_se_bwd_weights': [_P] * 8 + [_I] * 3 + [_P],
    't3d_se_after_sums': [_I, _P, _P, _PP, _P, _I, _I, _I, _P],
    't3d_se_after_apply': [_I, _P, _P, _PP, _P, _P, _P, _P, _I, _I, _I, _P],
    't3d_set_reduction_replicas': [_I, _L],
    't3d_set_workspace': [_P, _L],
    't3d_set_main_workspace': [_P, _L],
    't3d_fold_request': [_P, _P],
    't3d_fold_pending': [],
    't3d_pack_weights_batched': [_I, _P, _I, _P],
    't3d_pwconv_pack_frag': [_P, _P, _I, _I, _P],
    't3d_pwconv_frag_bytes': [_I, _I],
    't3d_pwconv_wants_frag': [_I, _I],
    't3d_iou3d': [_P, _P, _I, _I, _P, _P, _P, _P, _P],
    't3d_box_iou3d': [_P, _I, _P, _P, _P],
    't3d_adamw_step': [_P, _P, _P, _P, _L, _D, _D, _D, _D, _D, _L, _D, _P],
    't3d_set_grad_watch': [_P],
    't3d_zero_batched': [_P, _I, _P],
    't3d_copy_cols': [_P, _P, _I, _I, _I, _P],
    't3d_bn_bias_grad': [_P, _P, _I, _D, _P, _P, _P, _P, _P],
    't3d_se_bwd_affine': [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    't3d_dropout_mask': [_P, _L, ctypes.c_ulonglong, ctypes.c_ulonglong, _F, _P],
    't3d_loss_fwd_bwd': [_LP, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    't3d_metrics_per_sample': [_P, _P, _P, _P, _P, _I, _I, _P],
    't3d_expdw_fwd': [_I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    't3d_expdw_supported': [_I, _I, _I, _I, _I, _I, _I, _I],
    't3d_conv3x3_fwd': [_I, _P, _PP, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    't3d_conv3x3_dgrad': [_I, _P, _P, _BP, _P, _P, _PP, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    't3d_conv3x3_wgrad': [_I, _P, _P, _BP, _P, _PP, _P, _I, _I, _I, _I, _I, _I, _P],
    't3d_pack_conv3x3_dgrad_weight': [_P, _P, _I, _I, _P],
    # step plans (csrc/plan.hip): record once, replay with one call
    't3d_plan_create': [ctypes.POINTER(_P)],
    't3d_plan_destroy': [_P],
    't3d_plan_add_call': [_P, ctypes.c_char_p, _I, ctypes.POINTER(_I), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(_I)],
    't3d_plan_add_fork': [_P, _P, _P],
    't3d_plan_add_fork_after': [_P, _I, _P, _P],
    't3d_launch_count': [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(_P)],
    't3d_plan_add_copy_d2h': [_P, _I, _P, _L, _P],
    't3d_plan_add_event_record': [_P, _I, _P],
    't3d_plan_end_segment': [_P],
    't3d_plan_num_ops': [_P, _I],
    't3d_plan_time_entry': [_P, ctypes.c_char_p, _I],
    't3d_plan_failed_op': [_P, ctypes.POINTER(_I)],
    't3d_plan_run': [_P, _I, ctypes.POINTER(ctypes.c_ulonglong), _I, ctypes.POINTER(_P), _I],
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} not found: build it with `python __graft_entry__.py` '
                               '(there is no CPU fallback for the HIP path)')
        _lib = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = _I
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'native ops need contiguous device tensors'
    return t.data_ptr()


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise RuntimeError(f'unsupported storage dtype {t.dtype}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """HIP stream handle of torch's current stream on the current device.  (The raw accessor costs ~0.3 us; going
    through torch.cuda.current_stream() was 7 us a call -- 2 ms of host time per training step.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-launch device timing (HIP events on the launch stream), used by bench.py.
    `only`: set of entry-point names to time (None = all).  Events are resolved after a sync."""

    def __init__(self, only=None, prealloc=0, kernel_exact=False):
        self.only, self.rec, self.sig, self.kernel_exact = only, [], [], kernel_exact
        # events created (and recorded once, which is when torch really creates them) ahead of the timed region
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(prealloc)]
        for e in self.pool:
            e.record()

    def event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def per_launch(self):
        """[(name, int-args signature, ms, algorithmic bytes)] in launch order."""
        torch.cuda.synchronize()
        return [(n, sg, e0.elapsed_time(e1), nb) for (n, nb, e0, e1), sg in zip(self.rec, self.sig)]

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, nbytes, e0, e1 in self.rec:
            d = out.setdefault(name, dict(launches=0, ms=0.0, bytes=0))
            d['launches'] += 1
            d['ms'] += e0.elapsed_time(e1)
            d['bytes'] += nbytes or 0
        return out


timer = None    # set to a KernelTimer to time launches
# entry points whose main kernel takes t3d_set_launch_events (T3D_LAUNCH_TIMED in csrc): every convolution of the step
_KERNEL_TIMED = frozenset(('t3d_dwconv_fwd', 't3d_dwconv_bwd', 't3d_expdw_fwd', 't3d_pwconv_fwd', 't3d_pwconv_fwd_mat', 't3d_pwconv_dgrad',
                           't3d_pwconv_wgrad', 't3d_pwconv_dgrad_yfree', 't3d_pwconv_wgrad_yfree', 't3d_pwconv_bwd_yfree',
                           't3d_pwconv_bwd_yfree_w'))
# measurement aid (tools/ablate.sh): entry points whose launches are SKIPPED -- the results are then garbage, only the
# step time means something (an upper bound on what removing / fusing that family of launches can buy)
_ABLATE = frozenset(x for x in os.environ.get('T3D_ABLATE', '').split(',') if x)


# plan slots (values that change from step to step): see trainer/step_plan.py
SLOT_IMGS, SLOT_GT, SLOT_CATS, SLOT_DROPOUT, SLOT_STEP, SLOT_LR, SLOT_RB_DST, SLOT_RB_EVENT, NSLOTS = range(9)
recorder = None    # set to a PlanRecorder while a step is being recorded (torchdet3d/trainer/step_plan.py)


def call(name, *args, nbytes=None, slots=None):
    """Enqueue one C-ABI entry point on the current stream; `nbytes` = algorithmic HBM bytes of the launch
    (bookkeeping for the roofline report only); `slots` = {argument index: plan slot} for the arguments that change
    from step to step (only looked at while a plan is being recorded)."""
    if _ABLATE and name in _ABLATE:
        return
    fn = getattr(lib(), name)
    if recorder is not None:
        recorder.add_call(name, args, nbytes, slots)
        rc = fn(*args)
        recorder.after_call()
        if rc != 0:
            raise RuntimeError(f'{name} failed with code {rc}')
        return
    t = timer
    if t is not None and (t.only is None or name in t.only):
        e0, e1 = t.event(), t.event()
        if name in _KERNEL_TIMED and t.kernel_exact:
            # the events ride on the entry point's main kernel dispatch: begin-to-end of the kernel itself (include/t3d.h)
            lib().t3d_set_launch_events(e0.cuda_event, e1.cuda_event)
            rc = fn(*args)
            lib().t3d_set_launch_events(None, None)
        else:
            e0.record()
            rc = fn(*args)
            e1.record()
        t.rec.append((name, nbytes, e0, e1))
        t.sig.append(tuple(a for a in args[:-1] if isinstance(a, int) and not isinstance(a, bool) and 0 <= a < (1 << 31)))
    else:
        rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed with code {rc}')


def launch_count():
    """Kernels launched by the library so far in this process (t3d_launch_count)."""
    n = ctypes.c_ulonglong(0)
    lib().t3d_launch_count(ctypes.byref(n), None)
    return n.value


def prologue(scale=None, shift=None, se=None, act='none', se_after_act=False):
    """Keeps the referenced tensors alive on the returned object."""
    p = Prologue(ptr(scale), ptr(shift), ptr(se), ACT[act] if isinstance(act, str) else act, int(se_after_act))
    p._keep = (scale, shift, se)
    return p


def bnbwd(alpha, beta, gamma, per_sample=False):
    b = BnBwd(ptr(alpha), ptr(beta), ptr(gamma), int(per_sample))
    b._keep = (alpha, beta, gamma)
    return b


# ---- step plans (include/t3d.h: t3d_plan_*) ---------------------------------------------------------------------------
_U64 = (1 << 64) - 1


def _word(argtype, v):
    """One argument as the 64-bit word t3d_plan_add_call takes (kind 0)."""
    import struct
    if argtype is _F:
        return struct.unpack('<I', struct.pack('<f', float(v)))[0]
    if argtype is _D:
        return struct.unpack('<Q', struct.pack('<d', float(v)))[0]
    if v is None:
        return 0
    return int(v) & _U64


def double_bits(x):
    import struct
    return struct.unpack('<Q', struct.pack('<d', float(x)))[0]


HANDOFF = os.environ.get('T3D_PLAN_HANDOFF', '1') != '0'     # forks as device-side hand-offs (A/B switch)


class PlanRecorder:
    """Builds a t3d_plan while the engine issues a step through `call` (which still executes every call).  `ptr_slots`:
    {device address: slot} -- any pointer argument equal to one of these addresses is bound to the slot (the batch's
    input tensors); scalar slots are named by the call site (`call(..., slots={index: slot})`)."""

    def __init__(self, ptr_slots=None):
        self.plan = _P()
        rc = lib().t3d_plan_create(ctypes.byref(self.plan))
        if rc != 0:
            raise RuntimeError(f't3d_plan_create failed with code {rc}')
        self.ptr_slots = dict(ptr_slots or {})
        self.calls = []          # (name, int-args signature, nbytes) per call op, in order: what a timed replay reports against
        self.keep = []           # objects whose device memory the plan points into
        self.breaks = []         # host callbacks between segment i and i + 1
        n = ctypes.c_ulonglong(0)
        lib().t3d_launch_count(ctypes.byref(n), None)
        self.launches, self.last_on = n.value, {}      # kernels launched so far; stream -> op index of the last call that launched there
        self.broken = None       # why the recorded step cannot be replayed (something in it did not go through `call`)

    def add_call(self, name, args, nbytes, slots):
        types = SIGNATURES[name]
        n = len(types)
        if len(args) != n:
            raise RuntimeError(f'{name}: {len(args)} arguments for a {n}-argument entry point')
        kinds, words, sizes = (_I * n)(), (ctypes.c_ulonglong * n)(), (_I * n)()
        for i, (t, v) in enumerate(zip(types, args)):
            if slots and i in slots:
                kinds[i], words[i] = 2, slots[i]
            elif isinstance(v, ctypes.Structure):
                kinds[i], words[i], sizes[i] = 1, ctypes.addressof(v), ctypes.sizeof(v)
                self.keep.append(getattr(v, '_keep', None))
            elif t is _P and v is not None and int(v) in self.ptr_slots:
                kinds[i], words[i] = 2, self.ptr_slots[int(v)]
            elif t in (_PP, _BP, _LP):
                if v is not None:
                    raise RuntimeError(f'{name}: argument {i} is not a ctypes structure')
                kinds[i], words[i] = 0, 0
            else:
                kinds[i], words[i] = 0, _word(t, v)
        rc = lib().t3d_plan_add_call(self.plan, name.encode(), n, kinds, words, sizes)
        if rc != 0:
            raise RuntimeError(f't3d_plan_add_call({name}) failed with code {rc}')
        self.calls.append((name, tuple(a for a in args[:-1] if isinstance(a, int) and not isinstance(a, bool) and 0 <= a < (1 << 31)),
                           nbytes))

    def after_call(self):
        """Did the call just recorded launch a kernel, and on which stream did its last one go?"""
        n, st = ctypes.c_ulonglong(0), _P()
        lib().t3d_launch_count(ctypes.byref(n), ctypes.byref(st))
        if n.value != self.launches:
            self.launches = n.value
            self.last_on[st.value or 0] = lib().t3d_plan_num_ops(self.plan, -1) - 1

    def add_fork(self, from_stream, to_stream):
        """`to_stream` waits for everything enqueued on `from_stream` so far: for the last kernel a recorded call launched
        there (a hand-off through that kernel's own completion event, nothing enqueued on `from_stream`), or -- when nothing
        of this plan has run there yet -- for an event recorded on it."""
        prod = self.last_on.get(from_stream or 0) if HANDOFF else None
        if prod is not None:
            rc = lib().t3d_plan_add_fork_after(self.plan, prod, from_stream, to_stream)
        else:
            rc = lib().t3d_plan_add_fork(self.plan, from_stream, to_stream)
        if rc != 0:
            raise RuntimeError(f't3d_plan_add_fork failed with code {rc}')

    def add_readback(self, dst_slot, src, nbytes, event_slot, stream_):
        # (a plan that silently lacked the copy or its event would replay and hand back stale pinned memory)
        rc = lib().t3d_plan_add_copy_d2h(self.plan, dst_slot, src.data_ptr(), nbytes, stream_)
        if rc != 0:
            raise RuntimeError(f't3d_plan_add_copy_d2h failed with code {rc}')
        rc = lib().t3d_plan_add_event_record(self.plan, event_slot, stream_)
        if rc != 0:
            raise RuntimeError(f't3d_plan_add_event_record failed with code {rc}')
        self.keep.append(src)

    def end_segment(self):
        rc = lib().t3d_plan_end_segment(self.plan)
        if rc < 0:
            raise RuntimeError(f't3d_plan_end_segment failed with code {rc}')
        return rc

    def close(self):
        """Destroy the plan (a recording that failed or cannot be replayed); idempotent."""
        if self.plan:
            lib().t3d_plan_destroy(self.plan)
            self.plan = _P()

    def host_break(self, what):
        """Host code runs here in the eager step (a gradient-exchange callback): close the segment, remember what to call."""
        self.end_segment()
        self.breaks.append(what)
