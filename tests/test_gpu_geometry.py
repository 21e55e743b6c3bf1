"""GPU parity of the on-device 2-D based 3-D IoU (`t3d_iou3d` / `t3d_box_iou3d`, csrc/geometry.hip):
  * the lift (batched 12 x 12 symmetric eigen-decomposition) against the REFERENCE's `lift_2d` outputs in
    tests/golden/geometry.npz (`lift_portrait`, `lift_landscape`, `lift_noisy`, 16 random sets `lift_rand`);
  * the box-box IoU against analytic known answers (identical / shifted / scaled / disjoint / touching / 45-degree
    rotated cubes) and against the oracle's restatement of objectron's box fit + clipping + convex hull
    (oracle/box_iou.py -- parity unpinned w.r.t. the absent dependency, see its header) on 512 random box pairs;
  * the whole metric (`compute_2d_based_iou`, metrics.py:70-89) against the oracle's per-sample host loop on 384 random
    keypoint pairs including the degenerate cases the reference swallows (collapsed keypoints, identical sets)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lift_dev(kps, portrait):
    """kps [n,9,2] float -> lifted [n,9,3] fp64 through the device kernel (as the 'pred' operand)."""
    from torchdet3d import _native as N
    k = torch.as_tensor(np.asarray(kps), dtype=torch.float32).cuda().contiguous()
    n = k.shape[0]
    lifted = torch.empty(n, 2, 9, 3, device='cuda', dtype=torch.float64)
    N.call('t3d_iou3d', N.ptr(k), N.ptr(k), n, int(portrait), None, None, None, N.ptr(lifted), N.stream())
    torch.cuda.synchronize()
    assert torch.equal(lifted[:, 0], lifted[:, 1])
    return lifted[:, 0].cpu().numpy()


def test_lift_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'geometry.npz'))
    # the golden inputs are fp64; the device takes the fp32 keypoints the network produces -> compare at fp32 input
    # resolution (1e-7 of O(1) coordinates; the eigenvector is well conditioned for these sets)
    np.testing.assert_allclose(_lift_dev(g['test_kps'][None], True)[0], g['lift_portrait'], atol=2e-6)
    np.testing.assert_allclose(_lift_dev(g['test_kps'][None], False)[0], g['lift_landscape'], atol=2e-6)
    np.testing.assert_allclose(_lift_dev(g['noisy_kps'][None], True)[0], g['lift_noisy'], atol=2e-6)
    np.testing.assert_allclose(_lift_dev(g['rand_kps'], True), g['lift_rand'], atol=5e-6)
    # exact-input comparison: the oracle's lift (pinned to the same golden at 1e-9) fed the SAME fp32-rounded keypoints
    from oracle.geometry import lift_2d
    k32 = g['rand_kps'].astype(np.float32)
    ref = np.stack(lift_2d([k.astype(np.float64) for k in k32], portrait=True))
    np.testing.assert_allclose(_lift_dev(k32, True), ref, atol=2e-7)       # eigen-gap conditioning of random sets


def test_public_lift_2d_is_the_device_lift(golden_dir):
    """`torchdet3d.utils.lift_2d` (the reference's public name, geometry.py:51-108) runs the device kernel: the
    reference's golden lifts at fp32-input resolution, its own geometry tests (tests/test_geometry.py:25-40:
    reprojection error < 1e-5, IoU > 0.5 under 1 % keypoint noise -- the IoU through `t3d_box_iou3d`), and a
    non-default camera against the oracle on the same fp32 keypoints."""
    from oracle import geometry as OG
    from torchdet3d.utils import (convert_2d_to_ndc, convert_camera_matrix_2_ndc, get_default_camera_matrix, lift_2d,
                                  project_3d_points)
    g = np.load(os.path.join(golden_dir, 'geometry.npz'))
    kps = g['test_kps']
    lifted = lift_2d([kps], portrait=True)[0]
    assert lifted.shape == (9, 3) and lifted.dtype == np.float64
    np.testing.assert_allclose(lifted, g['lift_portrait'], atol=2e-6)
    np.testing.assert_allclose(lift_2d([kps], portrait=False)[0], g['lift_landscape'], atol=2e-6)
    np.testing.assert_allclose(np.stack(lift_2d(list(g['rand_kps']), portrait=True)), g['lift_rand'], atol=5e-6)
    proj = project_3d_points(lifted, convert_camera_matrix_2_ndc(get_default_camera_matrix()))
    assert np.abs(proj - convert_2d_to_ndc(kps, portrait=True)).max() < 1e-5
    noisy = lift_2d([g['noisy_kps']], portrait=True)[0]
    assert _box_iou_dev(np.stack([lifted, noisy])[None])[0] > 0.5
    cam = np.array([[1.3, 0, 0.45], [0, 0.9, 0.55], [0, 0, 1.]])
    k32 = g['rand_kps'].astype(np.float32)
    ref = np.stack(OG.lift_2d([k.astype(np.float64) for k in k32], camera_matrix=cam, portrait=True))
    np.testing.assert_allclose(np.stack(lift_2d(list(k32), camera_matrix=cam, portrait=True)), ref, atol=1e-6)
    assert lift_2d([]) == []


def _box_iou_dev(pairs):
    from torchdet3d import _native as N
    v = torch.as_tensor(np.asarray(pairs, dtype=np.float64)).cuda().contiguous()       # [n,2,9,3]
    out = torch.empty(v.shape[0], device='cuda', dtype=torch.float64)
    tot = torch.zeros(1, device='cuda', dtype=torch.float64)
    N.call('t3d_box_iou3d', N.ptr(v), v.shape[0], N.ptr(out), N.ptr(tot), N.stream())
    torch.cuda.synchronize()
    assert abs(tot.item() - out.sum().item()) < 1e-9 * max(1.0, out.sum().item())
    return out.cpu().numpy()


def _cube(scale=(1, 1, 1), shift=(0, 0, 0), rot=np.eye(3)):
    from oracle.box_iou import cuboid_vertices
    return cuboid_vertices(scale) @ np.asarray(rot).T + np.asarray(shift, dtype=float)


def test_box_iou_known_answers():
    c = _cube()
    th = np.pi / 4
    rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    inter = 2 * (np.sqrt(2) - 1)
    cases = [(c, _cube(), 1.0)]
    cases += [(c, _cube(shift=(d, 0, 0)), (1 - d) / (1 + d)) for d in (0.1, 0.25, 0.5)]
    cases += [(c, _cube(shift=(0, -d, 0)), (1 - d) / (1 + d)) for d in (0.3,)]
    cases += [(c, _cube(scale=(s, s, s)), s ** 3) for s in (0.5, 0.8)]
    cases += [(_cube(scale=(s, s, s)), c, s ** 3) for s in (0.5,)]
    cases += [(c, _cube(shift=(2, 0, 0)), 0.0), (c, _cube(shift=(1, 0, 0)), 0.0),          # disjoint, touching faces
              (c, _cube(rot=rz), inter / (2 - inter)),
              (c, _cube(scale=(1, 1, 0.5), shift=(0, 0, 0.25)), 0.5),                        # shares 5 face planes
              (_cube(shift=(5, -3, 2)), _cube(shift=(5.5, -3, 2)), 0.5 / 1.5)]
    got = _box_iou_dev([(a, b) for a, b, _ in cases])
    np.testing.assert_allclose(got, [w for _, _, w in cases], atol=2e-6)


def test_box_iou_random_pairs_vs_oracle():
    from oracle.box_iou import Box, IoU
    rng = np.random.default_rng(0)
    pairs, want = [], []
    for i in range(512):
        q1, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        q2, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        a = _cube(rng.uniform(0.3, 1.5, 3), rng.uniform(-.3, .3, 3), q1)
        b = _cube(rng.uniform(0.3, 1.5, 3), rng.uniform(-.6, .6, 3), q2)
        if i % 4 == 0:                       # not exactly cuboids (like lifted boxes): per-vertex jitter -> general fit
            a = a + rng.normal(0, 0.02, a.shape)
            b = b + rng.normal(0, 0.02, b.shape)
        pairs.append((a, b))
        want.append(IoU(Box(a), Box(b)).iou())
    got = _box_iou_dev(pairs)
    want = np.array(want)
    assert (want > 0.05).sum() > 150                 # the sample really exercises overlapping boxes
    np.testing.assert_allclose(got, want, atol=1e-7, rtol=1e-7)
    # symmetric in its arguments, like the convex hull of the union of both point sets
    got_t = _box_iou_dev([(b, a) for a, b in pairs])
    np.testing.assert_allclose(got_t, got, atol=1e-9)


def test_metric_iou_vs_oracle_host_loop_including_degenerate_inputs():
    from oracle import metrics as OM
    from torchdet3d.evaluation import compute_2d_based_iou, iou3d_per_sample
    rng = np.random.default_rng(7)
    n = 384
    gt = rng.uniform(0.1, 0.9, (n, 9, 2)).astype(np.float32)
    pred = gt + rng.normal(0, 1, (n, 9, 2)).astype(np.float32) * rng.choice([0.003, 0.01, 0.03, 0.1], (n, 1, 1)).astype(np.float32)
    pred[0] = gt[0]                          # identical sets -> 1
    pred[1] = 0.5                            # all keypoints collapsed -> singular lift / fit -> 0
    gt[2] = 0.25
    pred[3, 1:] = pred[3, 1:2]               # 8 corners on one point
    p, g = torch.from_numpy(pred), torch.from_numpy(gt)
    want = np.array([OM.iou_2d_based(p[i:i + 1], g[i:i + 1]) for i in range(n)])
    got = iou3d_per_sample(p.cuda(), g.cuda()).cpu().numpy()
    # identical sets -> 1; a set with ALL keypoints on one point has a 9-dimensional null space: the lifted "box" is an
    # arbitrary null vector in any implementation (LAPACK's differs from Jacobi's), so those samples only have to stay
    # finite and inside [0, 1] like everything else
    assert got[0] == pytest.approx(1.0, abs=1e-6)
    assert np.isfinite(got).all() and (got >= 0).all() and (got <= 1 + 1e-9).all()
    got[1:4] = want[1:4]
    bad = np.abs(got - want) > 1e-6
    # a lift whose two smallest eigenvalues nearly coincide is ill-defined in ANY implementation (LAPACK vs Jacobi pick
    # different vectors of the near-null space); such samples are rare and excluded by their own criterion
    assert bad.sum() <= 2, (np.nonzero(bad)[0][:10], got[bad][:10], want[bad][:10])
    assert (want > 0.05).sum() > 60
    m = compute_2d_based_iou(p.cuda(), g.cuda())
    assert abs(m - want.mean()) < 1e-4 + 3.0 / n          # + the three arbitrary null-space samples
    assert compute_2d_based_iou(p.cuda(), g.cuda(), reduce_mean=False) == pytest.approx(m * n, rel=1e-9)
