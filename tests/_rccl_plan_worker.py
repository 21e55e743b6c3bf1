"""Worker of tests/test_gpu_rccl.py::test_step_plan_replay_under_rccl: one rank of `torch.distributed.run` (backend 'nccl' = RCCL,
T3D_FORCE_SYNC=1) trains through `build_model` / `Trainer.train_step` -- the recorded step plan is cut into segments at the
gradient exchange's callbacks (trainer/step_plan.py) -- and must end with the weights of the same steps issued launch by
launch (T3D_STEP_PLAN semantics: the direct form), bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from test_host_logic import _cfg
    from torchdet3d.builders import build_loss, build_model, build_optimizer
    from torchdet3d.losses import LossManager
    from torchdet3d.trainer import Trainer, step_plan
    results = {}
    for replay in (True, False):
        step_plan.REPLAY = replay
        cfg = _cfg('mobilenetv2')
        cfg.model.storage_dtype = 'bf16'
        torch.manual_seed(3)
        model = build_model(cfg).to('cuda')          # joins the RCCL group, attaches the gradient exchange
        assert model.grad_sync is not None and dist.is_initialized() and dist.get_backend() == 'nccl'
        model.grad_sync.min_bucket = 1 << 18          # several buckets per backward
        model.net.reset_parameters(seed=3)
        opt = build_optimizer(cfg, model)
        lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
        tr = Trainer(model, None, opt, None, lm, None, 1, '', device='cuda', save_chkpt=False)
        model.train()
        g = torch.Generator(device='cuda').manual_seed(11)
        B, S = 16, 96
        imgs = [torch.randn(B, 3, S, S, device='cuda', generator=g) for _ in range(3)]
        gts = [torch.rand(B, 9, 2, device='cuda', generator=g) for _ in range(3)]
        cats = [torch.randint(0, 9, (B,), device='cuda', generator=g) for _ in range(3)]
        res = [dict(tr.train_step(imgs[i % 3], gts[i % 3], cats[i % 3], i)) for i in range(7)]
        torch.cuda.synchronize()
        sp = tr._sp
        results[replay] = (model.net.flat.clone(), res, sp.replays, len(sp.rec.breaks) if sp.rec is not None else 0)
    (w1, r1, n1, b1), (w0, r0, n0, b0) = results[True], results[False]
    assert n1 == 4 and n0 == 0, (n1, n0)
    assert b1 >= 3, b1                               # >= 2 bucket callbacks + the final one + finish
    assert r1 == r0, (r1[-1], r0[-1])
    assert torch.equal(w1, w0), (w1 - w0).abs().max().item()
    if dist.get_rank() == 0:
        print(f'RCCL_PLAN_OK segments={b1 + 1} replays={n1}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
