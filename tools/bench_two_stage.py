"""Regression stage of the two-stage pipeline (BASELINE config 5's second stage): detections of a 1080x1920 uint8 frame ->
crop + resize (t3d_crop_resize_u8) -> batched all-heads regression -> arg-max head, per frame.
usage: python tools/bench_two_stage.py [--model mobilenetv2] [--dets 16] [--frames 200] [--dtype bf16]
Prints one JSON line: frames/s and crops/s with the frame resident in HBM, the same with the 6.2 MB H2D copy of every frame
inside the timed region, and the oracle's host crop+resize loop (numpy, 1 core) for the same detections."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='mobilenetv2')
ap.add_argument('--dets', type=int, default=16)
ap.add_argument('--frames', type=int, default=200)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--batch-frames', type=int, nargs='*', default=[8, 32], help='with --detector: frames per launch chain of the '
                'batched detector stage (Detector.get_detections_batch; BASELINE config 5 says "batched on 1 MI355X")')
ap.add_argument('--detector', action='store_true', help='time the whole pipeline: SSD300-MobileNetV2 detector (models/ssd.py) '
                'on the frame, then the regression stage on its detections (scripts/demo.py:48-90)')
args = ap.parse_args()

from torchdet3d.builders import build_model
from torchdet3d.utils import AttrDict, Regressor

cfg = AttrDict(dict(model=dict(name=args.model, num_classes=9, pretrained=False, storage_dtype=args.dtype,
                               eval_storage_dtype=args.dtype)))      # inference in the SAME storage precision (opt-in for bf16)
model = build_model(cfg, export_mode=True).to('cuda')
model.eval()
H, W, n = 1080, 1920, args.dets
rng = np.random.default_rng(0)
frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
x0 = rng.integers(0, W - 400, n); y0 = rng.integers(0, H - 400, n)
rects = np.stack([x0, y0, x0 + rng.integers(60, 400, n), y0 + rng.integers(60, 400, n)], 1).astype(np.int32)
reg = Regressor(model, (224, 224), max_detections=n)
fd, rd = torch.from_numpy(frame).cuda(), torch.from_numpy(rects).cuda()
fh = torch.from_numpy(frame).pin_memory()


def run(frames, upload):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(frames):
        f = fd
        if upload:
            fd.copy_(fh, non_blocking=True)
        kp, labels = reg.regress(f, rd)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / frames


run(20, False)
t_res, t_up = run(args.frames, False), run(args.frames, True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    reg.crop_resize(fd, rd)
e1.record(); torch.cuda.synchronize()
t_crop = e0.elapsed_time(e1) / 50 * 1e-3
from oracle.crop_resize import crop, resize_linear_u8      # cpu_baseline leg only
t = time.perf_counter()
for r in rects:
    resize_linear_u8(crop(frame, r), (224, 224))
t_cpu = time.perf_counter() - t
extra = {}
if args.detector:
    from torchdet3d.models.ssd import SSD300
    from torchdet3d.utils import Detector
    det = Detector(SSD300('cuda', torch.bfloat16 if args.dtype == 'bf16' else torch.float32), conf=0.3)
    gd = torch.Generator().manual_seed(0)
    sd = det.model.state_dict()
    for k in sd:
        if k.startswith('bbox_head.cls_convs') and k.endswith('.3.bias'):
            sd[k] = torch.randn(sd[k].shape, generator=gd) * 2.0
    det.model.load_state_dict(sd)

    def pipeline(frames):
        torch.cuda.synchronize()
        t = time.perf_counter()
        nd = 0
        for _ in range(frames):
            dets = det.get_detections(fd)[:n]
            nd += len(dets)
            if dets:
                reg.get_detections(fd, dets)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / frames, nd / frames

    def detector_only(frames):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(frames):
            det.get_detections(fd)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / frames
    def detector_batched(fb, reps):
        stack = fd.unsqueeze(0).repeat(fb, 1, 1, 1).contiguous()
        det.get_detections_batch(stack)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            out = det.get_detections_batch(stack)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / (reps * fb), out

    def pipeline_batched(fb, reps):
        # detector over fb frames in one chain, then the regression stage frame by frame on its detections
        stack = fd.unsqueeze(0).repeat(fb, 1, 1, 1).contiguous()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            for f, dets in enumerate(det.get_detections_batch(stack)):
                if dets:
                    reg.get_detections(stack[f], dets[:n])
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / (reps * fb)
    pipeline(5)
    tp, nd = pipeline(max(20, args.frames // 4))
    td = detector_only(max(20, args.frames // 4))
    batched = {}
    one = det.get_detections(fd)
    for fb in args.batch_frames:
        tb, outb = detector_batched(fb, max(5, args.frames // (4 * fb)))
        assert all(o == one for o in outb), 'batched detections differ from the one-frame path'
        tpb = pipeline_batched(fb, max(3, args.frames // (8 * fb)))
        batched[str(fb)] = {'detector_ms_per_frame': round(tb * 1e3, 3), 'detector_frames_per_s': round(1 / tb, 1),
                            'pipeline_ms_per_frame': round(tpb * 1e3, 3), 'pipeline_frames_per_s': round(1 / tpb, 1)}
    extra = {'pipeline_ms_per_frame': round(tp * 1e3, 3), 'pipeline_frames_per_s': round(1 / tp, 1), 'detections_regressed_per_frame': round(nd, 1),
             'detector_ms_per_frame': round(td * 1e3, 3), 'batched_frames_per_launch_chain': batched, 'detector': 'SSD300-MobileNetV2 (random weights), one frame per launch chain, host read-back of the detections'}
print(json.dumps({**extra, 'metric': f'two-stage regression stage, {n} detections per 1080x1920 frame, {args.model}', 'frames_per_s': round(1 / t_res, 1),
                  'crops_per_s': round(n / t_res, 1), 'ms_per_frame': round(t_res * 1e3, 3), 'frames_per_s_with_h2d': round(1 / t_up, 1),
                  'crop_resize_us': round(t_crop * 1e6, 1), 'dtype': args.dtype,
                  'cpu_baseline': {'what': 'oracle crop + 8-bit bilinear resize loop (numpy), 1 core', 'ms_per_frame': round(t_cpu * 1e3, 2)}}))
