"""3-D box + exact IoU on the host (numpy / scipy), standing in for `objectron.dataset.box.Box` and
`objectron.dataset.iou.IoU`, which the reference imports from an un-vendored, un-pinned git submodule
(torchdet3d/evaluation/metrics.py:5-6,79-82; `.gitmodules:1-3`; the directory is empty in the reference tree).
PARITY UNPINNED against that dependency: the published algorithm is restated (least-squares box fit,
Sutherland-Hodgman clipping of the faces against the axis-aligned source box, convex-hull volume) and checked
with analytic known answers (tests/test_oracle_geometry.py, tests/test_host_logic.py).
"""
import numpy as np
import scipy.spatial

# vertex 0 = centre; 1..8 = corners, (x,y,z) sign pattern of the unit box
UNIT = 0.5 * np.array([[0, 0, 0], [-1, -1, -1], [-1, -1, 1], [-1, 1, -1], [-1, 1, 1],
                       [1, -1, -1], [1, -1, 1], [1, 1, -1], [1, 1, 1]], dtype=np.float64)
# four parallel edges per axis
EDGES = ((1, 5), (2, 6), (3, 7), (4, 8), (1, 3), (5, 7), (2, 4), (6, 8), (1, 2), (3, 4), (5, 6), (7, 8))
# quads, +x, -x, +y, -y, +z, -z
FACES = ((5, 6, 8, 7), (1, 3, 4, 2), (3, 7, 8, 4), (1, 2, 6, 5), (2, 4, 8, 6), (1, 5, 7, 3))
PLANE_EPS = 1e-6


def cuboid_vertices(scale):
    return UNIT * np.asarray(scale, dtype=np.float64)[None, :]


class Box:
    """9-vertex oriented box; rotation/translation/scale fitted lazily."""

    def __init__(self, vertices):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self._fit = None

    @classmethod
    def from_rts(cls, rotation, translation, scale):
        return cls(cuboid_vertices(scale) @ np.asarray(rotation).T + np.asarray(translation).reshape(1, 3))

    def _fitted(self):
        if self._fit is None:
            v = self.vertices
            scale = np.array([np.mean([np.linalg.norm(v[a] - v[b]) for a, b in EDGES[4 * ax:4 * ax + 4]])
                              for ax in range(3)])
            system = np.concatenate([cuboid_vertices(scale), np.ones((9, 1))], axis=1)
            sol = np.linalg.lstsq(system, v, rcond=None)[0]
            # general (possibly non-orthonormal) 3x3 kept as fitted
            self._fit = (sol[:3, :3].T, sol[3, :3], scale)
        return self._fit

    rotation = property(lambda s: s._fitted()[0])
    translation = property(lambda s: s._fitted()[1])
    scale = property(lambda s: s._fitted()[2])

    @property
    def transformation(self):
        t = np.identity(4)
        t[:3, :3] = self.rotation
        t[:3, 3] = self.translation
        return t

    @property
    def volume(self):
        v = self.vertices
        return abs(np.linalg.det(np.array([v[2] - v[1], v[3] - v[1], v[5] - v[1]])))

    def transformed(self, t):
        """New box regenerated from (T.R * R, T.R * t + T.t, scale)."""
        return Box.from_rts(t[:3, :3] @ self.rotation, t[:3, :3] @ self.translation + t[:3, 3], self.scale)

    def inside(self, p):
        inv = np.linalg.inv(self.transformation)
        q = inv[:3, :3] @ p + inv[:3, 3]
        return bool(np.all(np.abs(q) <= self.scale / 2.))


def _side(p, plane, normal, axis):
    d = normal * (p[axis] - plane[axis])
    return 1 if d > PLANE_EPS else (-1 if d < -PLANE_EPS else 0)


def _clip(poly, plane, normal, axis):
    """Sutherland-Hodgman step against one axis-aligned plane (keeps the 'front' side)."""
    if len(poly) <= 1:
        return []
    out = []
    all_on_plane = True
    for i, cur in enumerate(poly):
        prev = poly[i - 1]
        d1, d2 = _side(prev, plane, normal, axis), _side(cur, plane, normal, axis)
        if d2 == 0:
            if d1 != 0:
                out.append(cur)
            continue
        all_on_plane = False
        if d1 == -d2:                           # edge crosses the plane
            a = (cur[axis] - plane[axis]) / (cur[axis] - prev[axis])
            out.append(a * prev + (1.0 - a) * cur)
        elif d1 == 0 and (not out or not np.array_equal(out[-1], prev)):
            out.append(prev)
        if d2 > 0:
            out.append(cur)
    return poly if all_on_plane else out


def _intersection_points(src, tmpl, acc):
    inv = np.linalg.inv(src.transformation)     # LinAlgError if the fit is singular
    src_aa = src.transformed(inv)
    tm = tmpl.transformed(inv)
    R, t = src.rotation, src.translation
    for face in FACES:
        poly = [tm.vertices[i] for i in face]
        for axis in range(3):
            poly = _clip(poly, src_aa.vertices[1], 1.0, axis)
            poly = _clip(poly, src_aa.vertices[8], -1.0, axis)
        acc.extend(R @ p + t for p in poly)
    for v in tm.vertices:                       # all 9, centre included
        if src_aa.inside(v):
            acc.append(R @ v + t)


class IoU:
    def __init__(self, box1, box2):
        self.b1, self.b2 = box1, box2

    def iou(self):
        pts = []
        _intersection_points(self.b1, self.b2, pts)
        _intersection_points(self.b2, self.b1, pts)
        if not pts:
            return 0.
        inter = scipy.spatial.ConvexHull(np.array(pts)).volume   # QhullError if degenerate
        return inter / (self.b1.volume + self.b2.volume - inter)
