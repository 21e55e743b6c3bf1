"""Worker of tests/test_gpu_rccl.py: one rank of `torch.distributed.run`, backend 'nccl' (= RCCL on ROCm).  With
T3D_FORCE_SYNC=1 the data-parallel machinery runs even at world size 1: parameter broadcast, gradient buckets launched
from the engine's `grad_hook` while the backward is still running (engine.py `_maybe_hook`), the second-stream join in
front of every bucket, the 1/world scaling in the optimizer's gradient load.  At world size 1 the all-reduced gradient
must equal the plain one bit for bit; at world size > 1 every rank must end up with identical weights."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d import _native as N
    from torchdet3d.builders.optim_builder import FusedAdamW
    from torchdet3d.models.engine import Net
    from torchdet3d.parallel import GradSync
    B, HW, nc = 16, 96, 9
    imgs, gt_kp, cats = make_inputs(B, HW, HW, nc, seed=rank)
    sd = make_state_dict('mobilenetv2', nc, seed=rank)        # ranks start DIFFERENT: the broadcast must fix that
    mask = torch.full((B, 1280), 2.0, device=dev)
    cfg = N.LossCfg()
    cfg.c_l1, cfg.c_add, cfg.c_ce, cfg.lam_reg, cfg.lam_cls = 1.0, 0.1, 0.2, 1.0, 1.0
    cfg.smoothl1_beta, cfg.wing_w, cfg.wing_eps = 0.2, 5.18, 1.0
    gtd, cd, im = gt_kp.to(dev).view(B, 18).contiguous(), cats.to(dev), imgs.to(dev)

    def run(with_sync):
        net = Net('mobilenetv2', nc, dev, torch.float32)
        net.load_state_dict(sd)
        sync = GradSync(net.gflat, min_bucket=1 << 18)         # small buckets: several all-reduces per backward
        calls = []
        if with_sync:
            assert sync.force or world > 1
            sync.broadcast([net.flat] + list(net.buffers.values()))
            net.grad_hook = lambda lo: (calls.append(lo), sync.ready(lo))
        flat = torch.nn.Parameter(net.flat)
        flat.grad = net.gflat
        opt = FusedAdamW([flat], lr=1e-3, weight_decay=1e-4, grad_scale=1.0 / world if with_sync else 1.0)
        out = torch.zeros(16, device=dev)
        dkp, dlg = torch.empty(B, 18, device=dev), torch.empty(B, nc, device=dev)
        kp, lg = net.forward(im, cd, train=True, dropout_mask=mask)
        N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp), N.ptr(gtd), N.ptr(lg), N.ptr(cd), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc,
               N.stream())
        sync.start()
        net.backward(dkp, dlg)
        sync.finish(scale=False)
        g = net.gflat.clone()
        opt.step()
        torch.cuda.synchronize()
        return g, net.flat.clone(), calls

    g1, w1, calls = run(True)
    assert len(calls) >= 3 and calls[-1] == 0 and calls == sorted(calls, reverse=True), calls
    if world == 1:
        g0, w0, _ = run(False)
        # same kernels, same inputs (fp32 storage: two bf16 train steps of this randomly initialised network are not
        # comparable, see tests/test_gpu_bf16_gate.py); the only run-to-run difference is the order of the
        # BatchNorm-sum atomics, amplified by the network
        rel = ((g1 - g0).norm() / g0.norm()).item()
        assert rel < 1e-2, rel
        assert (w1 - w0).abs().max().item() < 2.5e-3          # AdamW's first step is +-lr per element
    gathered = [torch.empty_like(w1) for _ in range(world)]
    dist.all_gather(gathered, w1)
    assert all(torch.equal(gathered[0][:1000], t[:1000]) for t in gathered) or world == 1
    if rank == 0:
        print(f'RCCL_OK world={world} buckets={len(calls)}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
