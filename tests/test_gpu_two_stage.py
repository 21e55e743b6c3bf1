"""GPU: the two-stage pipeline's regression stage.  `t3d_crop_resize_u8` bit-exact against the oracle's crop + 8-bit bilinear
resize on random and edge-case detections out of a 1080x1920 frame; `Regressor.get_detections` against the reference's
per-detection loop (ie_wrappers.py:128-142) through the oracle model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rects(rng, n, H, W):
    x0 = rng.integers(0, W - 8, n); y0 = rng.integers(0, H - 8, n)
    x1 = np.minimum(x0 + rng.integers(2, 700, n), W + 40); y1 = np.minimum(y0 + rng.integers(2, 700, n), H + 40)
    return np.stack([x0, y0, x1, y1], 1).astype(np.int32)


@pytest.mark.parametrize('H,W,size', [(1080, 1920, (224, 224)), (97, 131, (224, 224)), (480, 640, (96, 128))])
def test_crop_resize_kernel_is_bit_exact_against_the_oracle(H, W, size):
    from oracle.crop_resize import crop, resize_linear_u8
    from torchdet3d import _native as N
    rng = np.random.default_rng(H)
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    edge = np.array([[0, 0, W, H], [0, 0, 1, 1], [W - 1, H - 1, W, H], [5, 7, 6, 300 if H > 300 else H], [3, 3, 3 + size[0], 3 + size[1]],
                     [0, 0, 2 * size[0], 2 * size[1]], [W - 9, 2, W + 50, 40], [10, 10, 10, 30], [-5, -7, 40, 50]], np.int32)
    rects = np.concatenate([edge, _rects(rng, 40, H, W)])
    fd, rd = torch.from_numpy(frame).cuda(), torch.from_numpy(rects).cuda()
    out = torch.full((len(rects), size[1], size[0], 3), 77, dtype=torch.uint8, device='cuda')
    N.call('t3d_crop_resize_u8', N.ptr(fd), N.ptr(rd), N.ptr(out), len(rects), H, W, size[1], size[0], N.stream())
    got = out.cpu().numpy()
    for i, r in enumerate(rects):
        c = crop(frame, r)
        ref = resize_linear_u8(c, size) if c.size else np.zeros((size[1], size[0], 3), np.uint8)
        assert np.array_equal(got[i], ref), (i, r, np.abs(got[i].astype(int) - ref.astype(int)).max())


def test_regressor_matches_the_reference_per_detection_loop():
    from oracle import model as om
    from oracle.crop_resize import regress_detections
    from oracle.weights import make_state_dict
    from test_host_logic import _cfg
    from torchdet3d.builders import build_model
    from torchdet3d.utils import Regressor
    name = 'mobilenetv3_large'
    cfg = _cfg(name)
    cfg.model.storage_dtype = 'f32'
    model = build_model(cfg, export_mode=True).to('cuda')
    sd = make_state_dict(name, 9)
    model.load_state_dict(sd)
    model.eval()
    rng = np.random.default_rng(3)
    H, W = 540, 960
    # smooth frame (random images are all high frequency: every crop would look alike to the network)
    yy, xx = np.mgrid[0:H, 0:W]
    frame = np.stack([(127 + 120 * np.sin(xx / 37.0 + c) * np.cos(yy / 53.0 - c)) for c in range(3)], -1)
    frame = np.clip(frame + rng.normal(0, 6, frame.shape), 0, 255).astype(np.uint8)
    dets = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), 0.9, 1) for r in _rects(rng, 12, H, W)]
    reg = Regressor(model, (224, 224), max_detections=4)                      # grows its buffers
    got = reg.get_detections(frame, dets)
    mean, std = [0.5931, 0.4690, 0.4229], [0.2471, 0.2214, 0.2157]          # configs/default_config.py:9-10 (the model's default)
    sdt = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}

    def fwd(x):
        kp, lg = om.forward_to_onnx(sdt, name, torch.from_numpy(x).double(), 9)
        return kp.numpy(), lg.numpy()
    ref = regress_detections(fwd, frame, dets, (224, 224), mean, std)
    assert len(got) == len(ref) == 12
    for (kpg, lg), (kpr, lr) in zip(got, ref):
        assert lg == lr                                                        # arg-max head: bit-exact
        assert kpg.shape == (1, 9, 2)
        np.testing.assert_allclose(kpg, kpr, atol=1e-4, rtol=0)
    assert reg.get_detections(frame, []) == []
    # transform_kp: crop-normalised -> frame pixels, like ie_wrappers.py:144-152
    kp = Regressor.transform_kp(got[0][0][0].copy(), dets[0][:4])
    x0, y0, x1, y1 = dets[0][:4]
    np.testing.assert_allclose(kp[:, 0], got[0][0][0][:, 0] * (x1 - x0) + x0, rtol=1e-6)


def test_frame_cropper_feeds_the_trainer_tuple():
    """FrameCropper: crops == oracle Objectron.crop + 8-bit bilinear resize, keypoints == (kp - origin) / crop size
    (A.Crop + Resize + ToTensor's normalisation, objectron_main.py:98-126, utils/transforms.py:112-114)."""
    from oracle.crop_resize import objectron_crop, resize_linear_u8
    from torchdet3d.dataloaders import FrameCropper
    rng = np.random.default_rng(9)
    H, W = 720, 1280
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    objs = []
    for _ in range(7):
        c = rng.integers(100, [W - 100, H - 100])
        objs.append((c + rng.integers(-90, 90, (9, 2))).astype(np.int64))
    objs.append(rng.integers(-50, 1400, (9, 2)).astype(np.int64))        # partly outside the frame
    crops, kps, boxes = FrameCropper((224, 224))(frame, objs)
    assert crops.shape == (8, 224, 224, 3) and crops.dtype == torch.uint8 and kps.shape == (8, 9, 2)
    for i, kp in enumerate(objs):
        skp, crop, box = objectron_crop(frame, kp)
        assert boxes[i] == box
        assert np.array_equal(crops[i].cpu().numpy(), resize_linear_u8(crop, (224, 224)))
        ref = skp.astype(np.float32) / np.float32([box[2] - box[0], box[3] - box[1]])
        np.testing.assert_allclose(kps[i].cpu().numpy(), ref, rtol=1e-6, atol=1e-7)
        assert kps[i].min().item() >= 0 and kps[i].max().item() <= 1


def test_frame_cropper_on_a_busy_stream_keeps_each_frames_boxes():
    """the crop boxes go up through pinned staging with an asynchronous copy.  With the stream running
    behind, a second call used to overwrite the staging buffer before the first call's copy had run, so frame k was
    cropped with frame k+1's boxes.  Queue ~50 ms of spinning, issue five frames back to back, compare each with the
    oracle."""
    from oracle.crop_resize import objectron_crop, resize_linear_u8
    from torchdet3d.dataloaders import FrameCropper
    rng = np.random.default_rng(21)
    H, W = 360, 640
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    dframe = torch.from_numpy(frame).cuda()
    fc = FrameCropper((96, 96))
    sets = [[(rng.integers(60, [W - 60, H - 60]) + rng.integers(-50, 50, (9, 2))).astype(np.int64) for _ in range(3)]
            for _ in range(5)]
    torch.cuda.synchronize()
    torch.cuda._sleep(100_000_000)
    outs = [fc(dframe, objs) for objs in sets]
    torch.cuda.synchronize()
    for objs, (crops, kps, boxes) in zip(sets, outs):
        for i, kp in enumerate(objs):
            _, crop, box = objectron_crop(frame, kp)
            assert boxes[i] == box
            assert np.array_equal(crops[i].cpu().numpy(), resize_linear_u8(crop, (96, 96)))


def _ssd_pair(dtype, seed=4):
    from oracle.ssd import detect as odetect
    from torchdet3d.models.ssd import SSD300, make_anchors
    m = SSD300('cuda', dtype, seed=seed)
    g = torch.Generator().manual_seed(seed)
    sd = m.state_dict()
    for k in sd:                      # trained-looking BatchNorm buffers and class biases: the scores must discriminate
        if k.startswith('bbox_head') and k.endswith('running_mean'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        elif k.startswith('bbox_head') and k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        elif k.startswith('bbox_head.cls_convs') and k.endswith('.3.bias'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 2.0
        elif k.startswith('bbox_head.reg_convs') and k.endswith('.3.bias'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.5
    m.load_state_dict(sd)
    return m, {k: v.cpu() for k, v in sd.items()}, odetect, make_anchors()


def test_ssd_detector_forward_decode_nms_matches_the_oracle():
    """BASELINE config 5, detector stage: SSD300-MobileNetV2 (fp32 storage) -- head outputs within 1e-4 of the oracle,
    then boxes / scores / labels after decode + softmax + per-class NMS + top-200 (parity with the reference's externally
    trained detector is unpinned; this pins the product to its own restatement of the config)."""
    from oracle.ssd import head_outputs as ohead
    m, sd, odetect, anchors = _ssd_pair(torch.float32)
    g = torch.Generator().manual_seed(1)
    imgs = torch.randn(2, 3, 300, 300, generator=g)
    outs = m.head_outputs(imgs.cuda())
    ref = ohead(sd, imgs)
    for (c, r, hw), (oc, orr) in zip(outs, ref):
        oc2, or2 = oc.reshape(2 * hw, -1), orr.reshape(2 * hw, -1)
        np.testing.assert_allclose(c.cpu().numpy()[:, :oc2.shape[1]], oc2.numpy(), atol=2e-4)
        np.testing.assert_allclose(r.cpu().numpy()[:, :or2.shape[1]], or2.numpy(), atol=2e-4)
        assert (c.cpu().numpy()[:, oc2.shape[1]:] == 0).all()                # padded output channels
    # decode + softmax + NMS + top-200 on RANDOM head outputs (no ties; the synthetic network's feature maps are nearly
    # constant, and among equal scores an ulp between expf and np.exp decides the order): the post-processing kernel against
    # the oracle's on identical inputs ...
    from oracle.ssd import postprocess
    g2 = torch.Generator().manual_seed(7)
    rnd = [(torch.randn(2, hw, c.shape[1], generator=g2) * 3, torch.randn(2, hw, r.shape[1], generator=g2)) for c, r, hw in outs]
    ro = [(cc[:, :, :len(W_) * 10].reshape(2, hw, len(W_), 10).numpy(), rr[:, :, :len(W_) * 4].reshape(2, hw, len(W_), 4).numpy())
          for (cc, rr), (_, _, hw), W_ in zip(rnd, outs, __import__('torchdet3d.models.ssd', fromlist=['WIDTHS']).WIDTHS)]
    want = postprocess(ro, anchors)
    real_heads = m.head_outputs
    m.head_outputs = lambda _: [(cc.reshape(2 * hw, -1).cuda().contiguous(), rr.reshape(2 * hw, -1).cuda().contiguous(), hw)
                                for (cc, rr), (_, _, hw) in zip(rnd, outs)]
    got = m.detect(imgs.cuda())
    m.head_outputs = real_heads
    for a, b in zip(got, want):
        assert a.shape == b.shape and a.shape[0] == 200
        assert (a[:, 5] == b[:, 5]).all()
        np.testing.assert_allclose(a[:, :5], b[:, :5], atol=1e-4)
    want = odetect(sd, imgs, anchors)
    # ... and end to end the product's own detections are the oracle's up to those flips: every oracle detection above 0.3
    # has a product detection of the same class within 2e-3
    for a, b in zip(m.detect(imgs.cuda()), want):
        hit = 0
        strong = b[b[:, 4] > 0.3]
        for row in strong:
            cand = a[a[:, 5] == row[5]]
            hit += bool(len(cand)) and np.abs(cand[:, :5] - row[:5]).max(1).min() < 2e-3
        assert len(strong) == 0 or hit >= 0.8 * len(strong), (hit, len(strong))     # (ties at the top-200 cut, see above)


def test_detector_then_regressor_is_the_whole_two_stage_pipeline():
    """scripts/demo.py:48-90's flow on one 1080x1920 frame: Detector.get_detections -> Regressor.get_detections ->
    transform_kp, everything on the GPU (bf16 storage, the throughput mode)."""
    from test_host_logic import _cfg
    from torchdet3d.builders import build_model
    from torchdet3d.utils import Detector, Regressor
    m, _, _, _ = _ssd_pair(torch.bfloat16)
    det = Detector(m, conf=0.3)
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    dets = det.get_detections(frame)
    det.run_async(frame)
    assert det.wait_and_grab() == dets
    assert len(dets) > 0
    # BASELINE config 5, "batched": several frames through ONE launch chain give each frame's own detections
    frame2 = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    both = det.get_detections_batch(np.stack([frame, frame2, frame]))
    assert both[0] == dets and both[2] == dets and both[1] == det.get_detections(frame2) and both[1] != dets
    for left, top, right, bottom, conf, label in dets:
        assert 0 <= left <= right <= 1920 and 0 <= top <= bottom <= 1080 and 0.3 < conf <= 1 and 0 <= label < 9
    cfg = _cfg('mobilenetv2')
    cfg.model.storage_dtype = 'bf16'
    reg = Regressor(build_model(cfg, export_mode=True).to('cuda'))
    res = reg.get_detections(frame, dets[:16])
    assert len(res) == min(len(dets), 16)
    for (kp, label), d in zip(res, dets):
        assert kp.shape == (1, 9, 2) and 0 <= label < 9
        pix = Regressor.transform_kp(kp[0].copy(), d[:4])
        assert np.isfinite(pix).all()
