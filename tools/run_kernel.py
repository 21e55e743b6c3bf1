"""Profiling aid: launch ONE conv entry point repeatedly on a given shape (for rocprofv3 --pmc / --kernel-trace).
usage: run_kernel.py dwfwd|dwbwd|pwfwd|pwdgrad|pwwgrad  B H W C [k s] | M HW K N   [--reps R] [--f32] [--res] [--gate]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch
from torchdet3d import _native as N

args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith('--') and sys.argv[i] not in ('--reps', '--nrep', '--act')]
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 5
dt = torch.float32 if '--f32' in sys.argv else torch.bfloat16
nrep = int(sys.argv[sys.argv.index('--nrep') + 1]) if '--nrep' in sys.argv else 1
kind, dims = args[0], [int(v) for v in args[1:]]
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
keep = []
if kind in ('dwfwd', 'dwbwd'):
    B, H, W, C = dims[:4]
    k, s = (dims[4], dims[5]) if len(dims) > 5 else (3, 1)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    x = rnd(B * H * W, C).to(dt); w = rnd(C, k * k) * 0.3
    sc, sh = torch.rand(C, device=dev) + 0.5, rnd(C) * 0.2
    act = sys.argv[sys.argv.index('--act') + 1] if '--act' in sys.argv else 'relu6'
    pro = N.prologue(sc, sh, None, act, False)
    gap = torch.zeros(B, C, device=dev, dtype=torch.int64) if '--gap' in sys.argv else None     # squeeze-excite pooled sums (fixed point)
    if gap is not None:
        N.call('t3d_set_exact_pool', 1)
    y = torch.empty(B * Ho * Wo, C, device=dev, dtype=dt)
    stats = torch.zeros(nrep, 2 * C, device=dev, dtype=torch.float64)
    N.call('t3d_set_reduction_replicas', nrep, 2 * C)
    if kind == 'dwfwd':
        fn = lambda: N.call('t3d_dwconv_fwd', N.dtype_code(x), N.ptr(x), pro, N.ptr(w), N.ptr(y), None if '--nostats' in sys.argv else N.ptr(stats), N.ptr(gap),
                            B, H, W, C, k, s, N.stream())
        nbytes = (x.numel() + y.numel()) * x.element_size()
    else:
        dz, yy = rnd(B * Ho * Wo, C).to(dt), rnd(B * Ho * Wo, C).to(dt)
        al, be, ga = torch.rand(C, device=dev) + 0.5, rnd(C) * 0.1, rnd(C) * 0.1
        bb = N.bnbwd(al, be, ga, False)
        dx = torch.empty_like(x); dw = torch.zeros(nrep, C, k * k, device=dev)
        res = rnd(B * H * W, C).to(dt) if '--res' in sys.argv else None      # skip-connection gradient (stride-1 residual blocks)
        fn = lambda: N.call('t3d_dwconv_bwd', N.dtype_code(x), N.ptr(dz), N.ptr(yy), bb, N.ptr(w), N.ptr(x), pro, N.ptr(res),
                            N.ptr(dx), N.ptr(stats), N.ptr(dw), B, H, W, C, k, s, N.stream())
        nbytes = 2 * (x.numel() + y.numel()) * x.element_size()
else:
    M, HW, K, Nn = dims[:4]
    x = rnd(M, K).to(dt); wf = (rnd(Nn, K) / K ** .5)
    wq = wf.to(dt).contiguous(); wt = wf.t().contiguous().to(dt)
    sc, sh = torch.rand(K, device=dev) + 0.5, rnd(K) * 0.2
    gate = torch.rand(M // HW, K, device=dev) if '--gate' in sys.argv else None      # squeeze-excite gate per (sample, channel) on the operand
    pro = N.prologue(sc, sh, gate, sys.argv[sys.argv.index('--act') + 1] if '--act' in sys.argv else 'relu6', False)
    y = torch.empty(M, Nn, device=dev, dtype=dt)
    dz, yy = rnd(M, Nn).to(dt), rnd(M, Nn).to(dt)
    al, be, ga = torch.rand(Nn, device=dev) + 0.5, rnd(Nn) * 0.1, rnd(Nn) * 0.1
    bb = N.bnbwd(al, be, ga, False)
    dx = torch.empty(M, K, device=dev, dtype=dt)
    nbytes = M * (K + Nn) * x.element_size()
    ws = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    N.call('t3d_set_workspace', N.ptr(ws), ws.numel())
    N.call('t3d_set_reduction_replicas', nrep, 2 * max(K, Nn))
    def frag(wm, kk, nn):       # --frag: the fragment-order copy + T3D_W_FRAG where the deep-contraction kernel takes the shape
        if '--frag' not in sys.argv or dt != torch.bfloat16:
            return N.dtype_code(x), wm
        out = torch.zeros(N.lib().t3d_pwconv_frag_bytes(*wm.shape) // 2, device=dev, dtype=dt)
        N.call('t3d_pwconv_pack_frag', N.ptr(wm), N.ptr(out), wm.shape[0], wm.shape[1], N.stream())
        return N.BF16 | N.W_FRAG, out
    if kind == 'pwfwd':
        stats = torch.zeros(nrep, 2 * max(K, Nn), device=dev, dtype=torch.float64)
        code, wv = frag(wq, K, Nn)
        fn = lambda: N.call('t3d_pwconv_fwd', code, N.ptr(x), pro, N.ptr(wv), None, N.ptr(y), N.ptr(stats),
                            M, HW, K, Nn, N.stream())
    elif kind == 'pwdgrad':
        stats = torch.zeros(nrep, 2 * max(K, Nn), device=dev, dtype=torch.float64)
        code, wv = frag(wt, Nn, K)
        fn = lambda: N.call('t3d_pwconv_dgrad', code, N.ptr(dz), N.ptr(yy), bb, N.ptr(wv), N.ptr(x), pro, None,
                            N.ptr(dx), N.ptr(stats), None, M, HW, K, Nn, N.stream())
    elif kind in ('pwdgrad_yf', 'pwwgrad_yf', 'yfprep', 'pwbwd_yf'):
        NP, KP = (Nn + 31) // 32 * 32, (K + 31) // 32 * 32
        wcat = torch.empty(K, NP + KP, device=dev, dtype=dt); cvec = torch.empty(K, device=dev)
        N.call('t3d_pwconv_yfree_prep', N.ptr(wt), bb, N.ptr(wcat), N.ptr(cvec), K, Nn, N.stream())
        stats = torch.zeros(nrep, 2 * max(K, Nn), device=dev, dtype=torch.float64)
        dw = torch.zeros(Nn, K, device=dev)
        if kind == 'pwbwd_yf':      # the one-pass form: data gradient + partial products (main stream) + reduce / combine
            need = N.lib().t3d_pwconv_bwd_yfree_scratch(M, K, Nn)
            assert need > 0, 'shape not supported by the fused kernel'
            wdl = torch.zeros((K + 15) // 16 * 16, (Nn + K + 8 + 63) // 64 * 64, device=dev, dtype=dt)
            N.call('t3d_pwconv_yfree_prep2', N.ptr(wt), bb, N.ptr(wcat), N.ptr(cvec), N.ptr(wdl), K, Nn, N.stream())
            scratch = torch.empty(need, device=dev, dtype=torch.uint8); dw = torch.zeros(Nn, K, device=dev)
            both = '--with-finish' in sys.argv

            def fn():
                N.call('t3d_pwconv_bwd_yfree', N.ptr(dz), N.ptr(x), N.ptr(wdl), N.ptr(x), None, None, N.ptr(dx), N.ptr(stats),
                       N.ptr(scratch), need, M, HW, K, Nn, N.stream())
                if both:
                    N.call('t3d_pwconv_wgrad_yfree_finish', N.ptr(scratch), bb, N.ptr(wq), N.ptr(dw), M, K, Nn, N.stream())
        elif kind == 'yfprep':
            fn = lambda: N.call('t3d_pwconv_yfree_prep', N.ptr(wt), bb, N.ptr(wcat), N.ptr(cvec), K, Nn, N.stream())
        elif kind == 'pwdgrad_yf':
            fn = lambda: N.call('t3d_pwconv_dgrad_yfree', N.ptr(dz), N.ptr(x), N.ptr(wcat), N.ptr(cvec), N.ptr(x), None,
                                None, N.ptr(dx), N.ptr(stats), M, HW, K, Nn, N.stream())
        else:
            fn = lambda: N.call('t3d_pwconv_wgrad_yfree', N.ptr(dz), N.ptr(x), bb, N.ptr(wq), N.ptr(dw), M, HW, K, Nn, N.stream())
    else:
        dw = torch.zeros(Nn, K, device=dev)
        fn = lambda: N.call('t3d_pwconv_wgrad', N.dtype_code(x), N.ptr(dz), N.ptr(yy), bb, N.ptr(x), pro, N.ptr(dw),
                            M, HW, K, Nn, N.stream())
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print(f'{kind} {dims}: {us:.1f} us  {nbytes / us / 1e6:.2f} TB/s algorithmic')
if os.environ.get('T3D_TRACE'):      # library built with -DT3D_PW_TRACE (tools/pw_trace.sh): in-kernel wall-clock stamps
    import ctypes
    lib = ctypes.CDLL(N.LIB_PATH)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    if os.environ['T3D_TRACE'] == 'deep':      # pwconv_deep.hip
        assert lib.t3d_debug_deep_trace(buf) == 0
        t = [v * 0.01 for v in buf]
        base = min(t[0], t[8])
        for nm, o in (('first block', 0), ('last block', 8)):
            print(f'  {nm}: start {t[o] - base:.2f} | coefficients {t[o + 1] - base:.2f} | phase 0 staged {t[o + 2] - base:.2f} | '
                  f'k-loop done {t[o + 3] - base:.2f} | stored {t[o + 4] - base:.2f} | stats done {t[o + 5] - base:.2f}')
        sys.exit(0)
    assert lib.t3d_debug_pw_trace(buf) == 0
    t = [v * 0.01 for v in buf]      # 100 MHz -> us
    base = min(t[0], t[8])
    print(f'  single launch {e0.elapsed_time(e1) * 1e3:.1f} us (events); stamps in us from the earlier block start:')
    for nm, o in (('first block', 0), ('last block', 8)):
        print(f'  {nm}: start {t[o] - base:.2f} | weights staged {t[o + 1] - base:.2f} | first k-loop done {t[o + 2] - base:.2f} | '
              f'groups done {t[o + 3] - base:.2f} | stats done {t[o + 4] - base:.2f}')
