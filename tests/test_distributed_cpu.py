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


def _host_worker(rank, world, port, q, out):
    sys.path[:0] = [ROOT, PKG]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from torchdet3d.utils import Logger
    lg = Logger(os.path.join(out, 'train.log'))          # main.py:39 -- BEFORE the process group exists
    has_file = lg.file is not None
    lg.close()
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from torchdet3d.builders import build_loader
    from torchdet3d.parallel import all_reduce_sums, is_main
    from torchdet3d.utils import save_snap
    from torchdet3d.utils.utils import AttrDict
    cfg = AttrDict(dict(data=dict(root='synthetic', resize=(16, 16), train_batch_size=8, val_batch_size=8, synthetic_len=30),
                        model=dict(num_classes=9), utils=dict(random_seeds=5)))
    train, val, _ = build_loader(cfg)
    train.sampler.set_epoch(0)
    e0 = [b[0].double().sum(dim=(1, 2, 3)).tolist() for b in train]
    train.sampler.set_epoch(1)
    e1 = [b[0].double().sum(dim=(1, 2, 3)).tolist() for b in train]
    v = [x for b in val for x in b[0].double().sum(dim=(1, 2, 3)).tolist()]

    class _Obj:
        def state_dict(self):
            return {'w': torch.ones(3) * rank}
    save_snap(_Obj(), _Obj(), None, 7, os.path.join(out, 'ck'))
    seen_after_barrier = os.path.exists(os.path.join(out, 'ck', 'snap_7.pth'))      # the barrier orders the write before this
    sums = all_reduce_sums([rank + 1.0, 10.0])
    bad = None
    try:
        cfg.data.train_batch_size = 7
        build_loader(cfg)
    except ValueError as e:
        bad = str(e)
    q.put((rank, has_file, e0, e1, v, seen_after_barrier, sums, is_main(), bad))
    dist.destroy_process_group()


def test_api_path_on_two_ranks_shards_batches_and_keeps_io_on_rank_0(tmp_path):
    """Host side of the one-process-per-GPU flow (scripts/main.py:60-61 scattered ONE batch over the replicas): the train
    loader hands every rank its own half of each global batch (same shuffle, disjoint indices, reshuffled per epoch), the
    validation loader the samples rank, rank + world, ... without padding; `save_snap` and the Logger's file belong to rank
    0; `all_reduce_sums` is what `Evaluator.val` / `Trainer.train` reduce their meters with.  (The model side of the same
    flow needs a GPU: tests/test_gpu_data_parallel.py::test_main_py_flow_on_two_ranks_...)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_host_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, f0, a0, a1, v0, s0, sums0, m0, bad0), (_, f1, b0, b1, v1, s1, sums1, m1, bad1) = res
    assert f0 and not f1 and m0 and not m1
    assert len(a0) == len(b0) == 3 and all(len(x) == 4 for x in a0 + b0)        # 30 crops -> 3 global batches of 8, 4 per rank
    flat = lambda e: [x for b in e for x in b]
    assert not set(flat(a0)) & set(flat(b0)) and not set(flat(a1)) & set(flat(b1))      # disjoint shares
    assert set(flat(a0)) != set(flat(a1))                                               # reshuffled by set_epoch
    assert len(v0) == len(v1) == 15 and not set(v0) & set(v1)                            # whole validation set, no padding
    assert s0 and s1 and os.listdir(tmp_path / 'ck') == ['snap_7.pth']
    assert torch.load(tmp_path / 'ck' / 'snap_7.pth')['state_dict']['w'].tolist() == [0.0, 0.0, 0.0]   # rank 0's
    assert sums0 == sums1 == [3.0, 20.0]
    assert bad0 and bad1 and 'GLOBAL batch' in bad0


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
