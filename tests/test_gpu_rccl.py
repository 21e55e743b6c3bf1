"""The data-parallel path on hardware: `torch.distributed.run` + backend 'nccl' (RCCL) on one MI355X with
T3D_FORCE_SYNC=1 -- every collective, the overlap hooks of the engine's backward and the second-stream joins run exactly
as on 8 GPUs (tests/_rccl_worker.py).  The ranks are CHILD processes; this process never touches the GPU before it
starts them."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_one_rank_rccl_gradient_exchange_matches_plain_step():
    env = dict(os.environ, T3D_FORCE_SYNC='1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', '_rccl_worker.py')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'RCCL_OK world=1' in r.stdout, r.stdout[-2000:]


def test_step_plan_replay_under_rccl():
    """The recorded step plan under a launcher: segments cut at the gradient exchange's callbacks, RCCL collectives issued from
    Python between two `t3d_plan_run` calls -- bit-identical to the launch-by-launch step (tests/_rccl_plan_worker.py)."""
    env = dict(os.environ, T3D_FORCE_SYNC='1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', '_rccl_plan_worker.py')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'RCCL_PLAN_OK' in r.stdout, r.stdout[-2000:]


def test_bench_spawns_its_own_ranks_and_reports_them():
    """`python bench.py --gpus 1` under the launcher (what the driver does for N > 1) reports rccl_ranks = 1; run
    plainly with --gpus 2 on a one-GPU box it must fail loudly instead of silently benchmarking one GPU."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '2',
           '--batch', '32', '--size', '96', '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['config']['rccl_ranks'] == 1 and line['value'] > 0
    import torch
    if torch.cuda.device_count() < 2:
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                             '--batch', '16', '--size', '96', '--no-cpu-baseline'], env=env, capture_output=True, text=True,
                            timeout=900)
        assert r2.returncode != 0
        assert 'GPU(s) visible' in (r2.stdout + r2.stderr)


def test_rccl_path_costs_under_three_percent_at_the_headline_shape():
    """VERDICT r5 #8 (multi-GPU readiness without the node): `bench.py --gpus 1` under the launcher with T3D_FORCE_SYNC=1 -- the
    segmented step plan, the bucketed RCCL all-reduces issued from the backward's hooks, the second-stream joins -- at BASELINE
    config 2's shape (MobileNetV2, B = 256 @224^2, bf16) against the plain one-process run on the same box: the exchange
    machinery must not cost more than 3 % of the step.  (One rank: the collectives move no bytes over xGMI; what this holds
    is everything around them.  The measured pair is recorded in DESIGN.md section 5.)"""
    import json
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    args = ['--gpus', '1', '--steps', '40', '--warmup', '10', '--no-cpu-baseline']

    def run(cmd, env):
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])

    plain = run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, base)
    launcher = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
                '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py')] + args
    forced = run(launcher, dict(base, T3D_FORCE_SYNC='1'))
    assert plain['config']['rccl_ranks'] == 0 and forced['config']['rccl_ranks'] == 1
    # medians of the per-step times: one slow step (a clock dip on a fresh box) must not decide a 3 % comparison
    a, b = plain['config']['step_ms_min_med_max'][1], forced['config']['step_ms_min_med_max'][1]
    print(f'[rccl readiness] plain {plain["ms_per_step"]} ms/step (median {a}), under RCCL with forced sync '
          f'{forced["ms_per_step"]} ms/step (median {b}): {100 * (b / a - 1):+.2f} %')
    assert b <= 1.03 * a, (a, b)
    assert abs(forced['config']['final_loss'] - plain['config']['final_loss']) < 5e-3
