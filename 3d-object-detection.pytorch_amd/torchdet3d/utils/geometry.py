"""Geometry helpers of the reference's `torchdet3d.utils` (torchdet3d/utils/geometry.py:16-108; callers: demo / drawing
code and `compute_2d_based_iou`, torchdet3d/evaluation/metrics.py:70-89).

`lift_2d` is the DEVICE lift: the batched 12x12 eigen-decomposition of `t3d_iou3d` (csrc/geometry.hip), the same launch
the validation metric uses, asked for its `lifted` output only.  There is no host copy of the algorithm in the product
(the numpy restatement lives in `oracle/geometry.py`, test infrastructure); the camera helpers below are a few scalar
operations each and stay on the host."""
import numpy as np
import torch

from .. import _native as N


def get_default_camera_matrix():
    """Unit focal length, principal point in the image centre (geometry.py:16-19)."""
    cam = np.eye(3)
    cam[:2, 2] = 0.5
    return cam


def convert_camera_matrix_2_ndc(matrix, img_shape=(1, 1)):
    """Pixel-space intrinsics -> normalised device coordinates (geometry.py:29-37)."""
    sx, sy = 2.0 / img_shape[0], 2.0 / img_shape[1]
    ndc = np.array(matrix, dtype=np.float64)
    ndc[0, 0], ndc[1, 1] = ndc[0, 0] * sx, ndc[1, 1] * sy
    ndc[0, 2], ndc[1, 2] = 1.0 - ndc[0, 2] * sx, 1.0 - ndc[1, 2] * sy
    return ndc


def project_3d_points(points, camera_matrix):
    """Pinhole projection with the camera looking down -z (geometry.py:22-26)."""
    homog = np.asarray(points) @ np.asarray(camera_matrix).T
    return homog[:, :2] / -homog[:, 2:3]


def convert_2d_to_ndc(points, portrait=False):
    """[0, 1]^2 image coordinates -> NDC; portrait frames swap the axes (geometry.py:40-48)."""
    pts = np.asarray(points)
    if portrait:
        return pts[:, ::-1] * 2 - 1
    return np.stack([pts[:, 0] * 2 - 1, 1 - pts[:, 1] * 2], axis=1)


def lift_2d(keypoint_sets, camera_matrix=None, portrait=False):
    """9 normalised 2-D keypoints per box -> 9x3 vertices per box, fp64 numpy (geometry.py:51-108), computed by the
    device kernel on the keypoints rounded to fp32 (what the network and the dataset hand over).  Needs a GPU: like
    every other arithmetic entry of the product it raises instead of falling back."""
    sets = [np.asarray(k) for k in keypoint_sets]
    for k in sets:
        assert len(k) == 9
    if not sets:
        return []
    if not torch.cuda.is_available():
        raise RuntimeError('lift_2d runs on the HIP path only (no CPU fallback)')
    kp = torch.from_numpy(np.stack(sets).astype(np.float32)).reshape(len(sets), 18).cuda().contiguous()
    cam = None
    if camera_matrix is not None:
        ndc = convert_camera_matrix_2_ndc(camera_matrix)
        cam = (N.ctypes.c_double * 4)(ndc[0, 0], ndc[1, 1], ndc[0, 2], ndc[1, 2])
    lifted = torch.empty(len(sets), 2, 9, 3, device='cuda', dtype=torch.float64)
    N.call('t3d_iou3d', N.ptr(kp), N.ptr(kp), len(sets), int(bool(portrait)), cam, None, None, N.ptr(lifted), N.stream())
    return list(lifted[:, 0].cpu().numpy())

