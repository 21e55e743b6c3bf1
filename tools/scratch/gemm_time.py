import torch
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
bf = torch.bfloat16
for M, K, N in [(12544, 960, 320), (12544, 960, 160), (12544, 160, 960), (12544, 320, 1280), (12544, 576, 160), (50176, 384, 64), (50176, 64, 384), (50176, 576, 96), (50176, 96, 576), (50176, 192, 64), (200704, 192, 32), (200704, 32, 192)]:
    x = torch.randn(M, K, device='cuda', dtype=bf); w = torch.randn(N, K, device='cuda', dtype=bf)
    y = torch.empty(M, N, device='cuda', dtype=bf)
    t = timeit(lambda: torch.matmul(x, w.t(), out=y))
    # wgrad shape: [N, M] x [M, K]
    dz = torch.randn(M, N, device='cuda', dtype=bf)
    dw = torch.empty(N, K, device='cuda', dtype=bf)
    t2 = timeit(lambda: torch.matmul(dz.t(), x, out=dw))
    print(f'{M:7d} {K:5d}->{N:5d}: fwd gemm {t:6.1f} us ({2e-6*M*K*N/t/1e3:.2f} PF)   wgrad gemm {t2:6.1f} us')
