"""Oracle geometry: lift_2d pinned against the reference (golden), Box/IoU
(dependency absent from the reference tree -> parity unpinned) checked by
analytic known answers + the reference's own threshold test."""
import os

import numpy as np

from oracle.box_iou import Box, IoU, cuboid_vertices
from oracle.geometry import camera_matrix_to_ndc, default_camera_matrix, lift_2d, project_3d_points, to_ndc


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, 'geometry.npz'))


def test_lift_matches_reference(golden_dir):
    g = _g(golden_dir)
    np.testing.assert_allclose(lift_2d([g['test_kps']], portrait=True)[0], g['lift_portrait'], atol=1e-9)
    np.testing.assert_allclose(lift_2d([g['test_kps']], portrait=False)[0], g['lift_landscape'], atol=1e-9)
    np.testing.assert_allclose(np.stack(lift_2d(list(g['rand_kps']), portrait=True)), g['lift_rand'], atol=1e-8)
    ndc = camera_matrix_to_ndc(default_camera_matrix())
    np.testing.assert_allclose(ndc, g['ndc_cam'])
    np.testing.assert_allclose(project_3d_points(g['lift_portrait'], ndc), g['reproj'], atol=1e-9)
    np.testing.assert_allclose(to_ndc(g['test_kps'], portrait=True), g['kps_ndc'])


def test_reprojection_error(golden_dir):            # reference tests/test_geometry.py:25-29
    g = _g(golden_dir)
    k3 = lift_2d([g['test_kps']], portrait=True)[0]
    rp = project_3d_points(k3, camera_matrix_to_ndc(default_camera_matrix()))
    assert np.any(np.linalg.norm(to_ndc(g['test_kps'], True) - rp, axis=1) < 1e-5)


def test_3d_iou_stability(golden_dir):              # reference tests/test_geometry.py:31-40
    g = _g(golden_dir)
    a, b = lift_2d([g['test_kps'], g['noisy_kps']], portrait=True)
    assert IoU(Box(a), Box(b)).iou() > 0.5


def _cube(scale=(1, 1, 1), shift=(0, 0, 0), rot=np.eye(3)):
    return Box(cuboid_vertices(scale) @ np.asarray(rot).T + np.asarray(shift, dtype=float))


def test_iou_known_answers():
    c = _cube()
    assert abs(IoU(c, _cube()).iou() - 1.0) < 1e-9
    for d in (0.1, 0.25, 0.5):
        assert abs(IoU(c, _cube(shift=(d, 0, 0))).iou() - (1 - d) / (1 + d)) < 1e-9
    for s in (0.5, 0.8):
        assert abs(IoU(c, _cube(scale=(s, s, s))).iou() - s ** 3) < 1e-9
    assert IoU(c, _cube(shift=(2, 0, 0))).iou() == 0.
    th = np.pi / 4
    rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    inter = 2 * (np.sqrt(2) - 1)
    assert abs(IoU(c, _cube(rot=rz)).iou() - inter / (2 - inter)) < 1e-9
    # general position: symmetric and within [0, 1]
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    b = _cube(scale=(1.2, 0.7, 0.9), shift=(0.2, -0.1, 0.15), rot=q)
    i1, i2 = IoU(c, b).iou(), IoU(b, c).iou()
    assert 0 < i1 < 1 and abs(i1 - i2) < 1e-9
    assert abs(c.volume - 1) < 1e-12 and abs(b.volume - 1.2 * 0.7 * 0.9) < 1e-9
