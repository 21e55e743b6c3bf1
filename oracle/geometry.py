"""Oracle geometry: EPnP-style 2-D -> 3-D box lifting (test infrastructure).

numpy fp64 restatement of torchdet3d/utils/geometry.py:6-108.
"""
import numpy as np

# geometry.py:6-13 -- barycentric weights of the 8 box corners w.r.t. 4 control points
EPNP_ALPHA = np.array([[4, -1, -1, -1], [2, -1, -1, 1], [2, -1, 1, -1], [0, -1, 1, 1],
                       [2, 1, -1, -1], [0, 1, -1, 1], [0, 1, 1, -1], [-2, 1, 1, 1]], dtype=np.float64)


def default_camera_matrix():                                   # geometry.py:16-19
    return np.array([[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1.]])


def camera_matrix_to_ndc(m, img_shape=(1, 1)):                 # geometry.py:29-37
    n = np.array(m, dtype=np.float64)
    n[0, 0] *= 2.0 / img_shape[0]
    n[1, 1] *= 2.0 / img_shape[1]
    n[0, 2] = -n[0, 2] * 2.0 / img_shape[0] + 1.0
    n[1, 2] = -n[1, 2] * 2.0 / img_shape[1] + 1.0
    return n


def project_3d_points(points, camera_matrix):                  # geometry.py:22-26
    pr = (camera_matrix @ points.T).T
    pr = pr / -pr[:, 2:3]
    return pr[:, :2]


def to_ndc(points, portrait=False):                            # geometry.py:40-48
    out = np.zeros_like(points)
    if portrait:
        out[:, 0] = points[:, 1] * 2 - 1
        out[:, 1] = points[:, 0] * 2 - 1
    else:
        out[:, 0] = points[:, 0] * 2 - 1
        out[:, 1] = 1 - points[:, 1] * 2
    return out


def lift_system(kp_set, portrait, fx, fy, cx, cy):
    """The 16x12 matrix of geometry.py:65-88 (centre keypoint kp_set[0] unused)."""
    m = np.zeros((16, 12))
    # NDC in the keypoints' OWN dtype, like the reference's element-wise `kp[1] * 2 - 1` on float32 rows (:73-78)
    uv = to_ndc(np.asarray(kp_set)[1:], portrait).astype(np.float64)
    for i in range(8):
        u, v = uv[i]
        for j in range(4):
            a = EPNP_ALPHA[i, j]
            m[2 * i, 3 * j] = fx * a
            m[2 * i, 3 * j + 2] = (cx + u) * a
            m[2 * i + 1, 3 * j + 1] = fy * a
            m[2 * i + 1, 3 * j + 2] = (cy + v) * a
    return m


def lift_2d(keypoint_sets, camera_matrix=None, portrait=False):   # geometry.py:51-108
    cam = camera_matrix_to_ndc(default_camera_matrix() if camera_matrix is None else camera_matrix)
    fx, fy, cx, cy = cam[0, 0], cam[1, 1], cam[0, 2], cam[1, 2]
    out = []
    for kp_set in keypoint_sets:
        assert len(kp_set) == 9
        m = lift_system(kp_set, portrait, fx, fy, cx, cy)
        w, v = np.linalg.eigh(m.T @ m)                         # :90-91, ascending eigenvalues
        ctrl = v[:, 0].reshape(4, 3)
        if ctrl[0, 2] > 0:                                     # :95-96 all points in front (z<0)
            ctrl = -ctrl
        out.append(np.vstack([ctrl[0:1], EPNP_ALPHA @ ctrl]))  # :98-105
    return out
