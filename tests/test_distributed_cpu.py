"""World-size-2 `gloo` test of the data-parallel gradient exchange (torchdet3d/parallel.py): bucketed
all-reduce launched tail-first while "backward" is still filling the buffer, then averaged; parameter broadcast
from rank 0.  Runs on CPU; on the GPU box the same code runs over RCCL ('nccl')."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, PKG]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from torchdet3d.parallel import GradSync
    n = 10000
    g = torch.zeros(n)
    sync = GradSync(g, min_bucket=3000)
    p = torch.full((7,), float(rank + 5))
    sync.broadcast([p])
    sync.start()
    launched, prev_lo = [], [n]
    for lo in (9000, 6500, 6000, 2000, 0):          # "backward" finalises the tail first
        hi = sync.hi if sync.hi < n or not launched else sync.hi
        prev = prev_lo[0]
        g[lo:prev] = torch.arange(lo, prev, dtype=torch.float32) * (rank + 1)   # fill the newly final part
        prev_lo[0] = lo
        sync.ready(lo)
        launched.append(len(sync.works))
    sync.finish()
    expect = torch.arange(0, n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    q.put((rank, bool(torch.allclose(g, expect)), p.tolist(), launched))
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, pvals, launched in res:
        assert ok, f'rank {rank}: averaged gradient wrong'
        assert pvals == [5.0] * 7, 'parameters must come from rank 0'
        assert launched == [0, 1, 1, 2, 3], launched     # buckets of >= 3000 elements, tail first, remainder at 0


def test_bench_refuses_a_rank_count_it_cannot_start():
    """`bench.py --gpus N` run plainly starts its N ranks itself (child `torch.distributed.run`); here there is no GPU,
    so every rank must stop with the explicit message and the parent must pass the failure on -- never a silent
    single-GPU number."""
    import subprocess
    # no device visible to the child whatever the host has: on a multi-GPU box the command would otherwise run a real
    # 2-rank benchmark and exit 0
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
                        '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert 'GPU(s) visible' in (r.stdout + r.stderr)
