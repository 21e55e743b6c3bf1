#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_stem.py -x -q 2>&1 | tail -2
python - <<'PY'
import sys, os, torch
sys.path[:0] = ['.', '3d-object-detection.pytorch_amd']
from torchdet3d import _native as N
B, H, W = 256, 224, 224
x = torch.randn(B, 3, H, W, device='cuda'); col = torch.empty(B * 112 * 112, 32, device='cuda', dtype=torch.bfloat16)
u = torch.randint(0, 255, (B, H, W, 3), device='cuda', dtype=torch.uint8); m = torch.zeros(3, device='cuda'); s = torch.ones(3, device='cuda')
for nm, fn in (('f32->bf16', lambda: N.call('t3d_stem_im2col', N.BF16, N.ptr(x), N.ptr(col), B, H, W, N.stream())),
               ('u8->bf16', lambda: N.call('t3d_stem_im2col_u8', N.BF16, N.ptr(u), N.ptr(m), N.ptr(s), N.ptr(col), B, H, W, N.stream()))):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(nm, e0.elapsed_time(e1) * 50, 'us')
PY
