// On-device 2-D based 3-D IoU (torchdet3d/evaluation/metrics.py:70-89): per sample, both keypoint sets are lifted to
// 3-D boxes (lift_2d, torchdet3d/utils/geometry.py:51-108) and the IoU of the two boxes is computed the way
// objectron.dataset.{box,iou} does it: box fit by least squares, intersection of the two
// FITTED boxes, volumes of the RAW vertex sets.  fp64 throughout, one 128-thread workgroup per sample:
//
//   wave w (0: predicted, 1: ground truth)
//     M^T M (12 x 12) of the 16 x 12 EPnP system, built entry-wise from the 8 corner keypoints          (:65-89)
//     cyclic Jacobi eigen-decomposition, matrix + eigenvectors in LDS, lanes = rows                     (:90-91, np.linalg.eigh)
//     eigenvector of the smallest eigenvalue -> 4 control points, z < 0 sign rule -> 9 vertices         (:92-105)
//   wave 0, after the barrier
//     fit (scale = mean edge length per axis; because the scaled unit box is symmetric about its centre the least-
//       squares system [X 1] S = V decouples exactly: R = (1 / 4 s_a) sum_v sign_va V_v, t = mean_v V_v)
//     both boxes expressed in box 1's frame: box 1 = axis-aligned cuboid, box 2 = parallelepiped
//     intersection volume by the divergence theorem over the 12 clipped face polygons (lane f clips face f with
//       Sutherland-Hodgman against the other box's 6 half-spaces, thickness eps = 1e-6 like the reference's clipper):
//       V = 1/3 sum_f dist_f * area_f -- equal to the convex-hull volume of the reference's intersection point set,
//       because that set is exactly the vertex set of this convex polytope.  A face of box 2 that lies ON a face of
//       box 1 with the same outward direction would be counted twice and is dropped.
//     IoU = V / (vol1 + vol2 - V); singular fits and empty / flat intersections give 0 (the reference's
//       LinAlgError / QhullError / no-points cases, metrics.py:82-86).
#include "common.h"

namespace {

constexpr int LD = 13;        // padded leading dimension of the 12 x 12 matrices in LDS
constexpr int MAXV = 12;      // a quad clipped by 6 planes has at most 10 vertices
constexpr double PLANE_EPS = 1e-6;
constexpr int JACOBI_SWEEPS = 12;

__constant__ double c_alpha[8][4] = {{4, -1, -1, -1}, {2, -1, -1, 1}, {2, -1, 1, -1}, {0, -1, 1, 1},
                                     {2, 1, -1, -1},  {0, 1, -1, 1},  {0, 1, 1, -1},  {-2, 1, 1, 1}};
// unit-box corner signs of vertices 1..8 and the quads +x -x +y -y +z -z (vertex numbers 1..8)
__constant__ double c_sign[8][3] = {{-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1},
                                    {1, -1, -1},  {1, -1, 1},  {1, 1, -1},  {1, 1, 1}};
__constant__ int c_faces[6][4] = {{5, 6, 8, 7}, {1, 3, 4, 2}, {3, 7, 8, 4}, {1, 2, 6, 5}, {2, 4, 8, 6}, {1, 5, 7, 3}};
__constant__ int c_edges[12][2] = {{1, 5}, {2, 6}, {3, 7}, {4, 8}, {1, 3}, {5, 7}, {2, 4}, {6, 8}, {1, 2}, {3, 4}, {5, 6}, {7, 8}};

struct Smem {
  double A[2][12 * LD];
  double V[2][12 * LD];
  double vert[2][9][3];
  double poly[12][2][MAXV][3];
};

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// ---- lift_2d of one keypoint set by one wave
__device__ void lift_wave(const float* __restrict__ kp, int portrait, double fx, double fy, double cx, double cy,
                          double* A, double* V, double (*vert)[3], int lane) {
  // NDC coordinates in the keypoints' own precision (the reference converts float32 arrays element-wise, :73-78)
  double rx[8][3], ry[8][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float k0 = kp[(i + 1) * 2], k1 = kp[(i + 1) * 2 + 1];
    const float u = portrait ? k1 * 2.f - 1.f : k0 * 2.f - 1.f;
    const float v = portrait ? k0 * 2.f - 1.f : 1.f - k1 * 2.f;
    rx[i][0] = fx; rx[i][1] = 0.0; rx[i][2] = cx + (double)u;
    ry[i][0] = 0.0; ry[i][1] = fy; ry[i][2] = cy + (double)v;
  }
  for (int e = lane; e < 144; e += 64) {
    const int r = e / 12, c = e % 12, j = r / 3, a = r % 3, j2 = c / 3, a2 = c % 3;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c_alpha[i][j] * c_alpha[i][j2] * (rx[i][a] * rx[i][a2] + ry[i][a] * ry[i][a2]);
    A[r * LD + c] = s;
    V[r * LD + c] = (r == c) ? 1.0 : 0.0;
  }
  wave_sync();
  // cyclic Jacobi: A <- J^T A J, V <- V J
  for (int sweep = 0; sweep < JACOBI_SWEEPS; ++sweep) {
    for (int p = 0; p < 11; ++p) {
      for (int q = p + 1; q < 12; ++q) {
        const double apq = A[p * LD + q], app = A[p * LD + p], aqq = A[q * LD + q];
        double c = 1.0, s = 0.0;
        const bool rot = fabs(apq) > 1e-300 && fabs(apq) > 1e-22 * (fabs(app) + fabs(aqq));   // wave-uniform
        if (rot) {
          const double theta = (aqq - app) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
        }
        wave_sync();
        if (rot) {
          if (lane < 12) {              // columns p, q of A
            const double x = A[lane * LD + p], y = A[lane * LD + q];
            A[lane * LD + p] = c * x - s * y;
            A[lane * LD + q] = s * x + c * y;
          } else if (lane >= 16 && lane < 28) {   // columns p, q of V
            const int k = lane - 16;
            const double x = V[k * LD + p], y = V[k * LD + q];
            V[k * LD + p] = c * x - s * y;
            V[k * LD + q] = s * x + c * y;
          }
        }
        wave_sync();
        if (rot && lane < 12) {         // rows p, q of A
          const double x = A[p * LD + lane], y = A[q * LD + lane];
          A[p * LD + lane] = c * x - s * y;
          A[q * LD + lane] = s * x + c * y;
        }
        wave_sync();
        if (rot && lane == 0) { A[p * LD + q] = 0.0; A[q * LD + p] = 0.0; }
        wave_sync();
      }
    }
  }
  int idx = 0;
  double best = A[0];
  for (int k = 1; k < 12; ++k) {
    const double d = A[k * LD + k];
    if (d < best) { best = d; idx = k; }
  }
  const double sgn = V[2 * LD + idx] > 0.0 ? -1.0 : 1.0;        // all points in front of the camera: z < 0 (:95-96)
  if (lane < 27) {
    const int vi = lane / 3, a = lane % 3;
    double s;
    if (vi == 0) {
      s = V[a * LD + idx];
    } else {
      s = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += c_alpha[vi - 1][j] * V[(3 * j + a) * LD + idx];
    }
    vert[vi][a] = sgn * s;
  }
}

struct Fit { double s[3], R[3][3], t[3], vol; };

__device__ void fit_box(const double (*v)[3], Fit& f) {
  for (int a = 0; a < 3; ++a) {
    double len = 0.0;
    for (int e = 0; e < 4; ++e) {
      const int i = c_edges[a * 4 + e][0], j = c_edges[a * 4 + e][1];
      const double dx = v[i][0] - v[j][0], dy = v[i][1] - v[j][1], dz = v[i][2] - v[j][2];
      len += sqrt(dx * dx + dy * dy + dz * dz);
    }
    f.s[a] = len * 0.25;
  }
  for (int i = 0; i < 3; ++i) {
    double m = 0.0;
    for (int k = 0; k < 9; ++k) m += v[k][i];
    f.t[i] = m / 9.0;
    for (int a = 0; a < 3; ++a) {
      double acc = 0.0;
      for (int k = 0; k < 8; ++k) acc += c_sign[k][a] * v[k + 1][i];
      f.R[i][a] = acc / (4.0 * f.s[a]);          // s = 0 -> inf/nan -> the singular-fit path below
    }
  }
  double e1[3], e2[3], e3[3];
  for (int i = 0; i < 3; ++i) { e1[i] = v[2][i] - v[1][i]; e2[i] = v[3][i] - v[1][i]; e3[i] = v[5][i] - v[1][i]; }
  f.vol = fabs(e1[0] * (e2[1] * e3[2] - e2[2] * e3[1]) - e1[1] * (e2[0] * e3[2] - e2[2] * e3[0]) +
               e1[2] * (e2[0] * e3[1] - e2[1] * e3[0]));
}

__device__ __forceinline__ double det3(const double m[3][3]) {
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
         m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// Sutherland-Hodgman against the half-space n.x - d <= eps; returns the new vertex count
__device__ int clip_poly(const double (*in)[3], int n, double (*out)[3], const double nn[3], double d) {
  if (n < 3) return 0;
  int m = 0;
  double gp = nn[0] * in[n - 1][0] + nn[1] * in[n - 1][1] + nn[2] * in[n - 1][2] - d;
  for (int i = 0; i < n; ++i) {
    const int ip = (i == 0) ? n - 1 : i - 1;
    const double gc = nn[0] * in[i][0] + nn[1] * in[i][1] + nn[2] * in[i][2] - d;
    const bool pin = gp <= PLANE_EPS, cin = gc <= PLANE_EPS;
    if (pin != cin) {
      double t = gp / (gp - gc);
      t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
      if (m < MAXV) {
        for (int a = 0; a < 3; ++a) out[m][a] = in[ip][a] + t * (in[i][a] - in[ip][a]);
        ++m;
      }
    }
    if (cin && m < MAXV) {
      for (int a = 0; a < 3; ++a) out[m][a] = in[i][a];
      ++m;
    }
    gp = gc;
  }
  return m;
}

__global__ __launch_bounds__(128) void iou3d_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                    int portrait, double fx, double fy, double cx, double cy,
                                                    double* __restrict__ iou, double* __restrict__ total,
                                                    double* __restrict__ lifted, const double* __restrict__ verts) {
  __shared__ Smem sm;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (verts) {               // boxes given directly as 9 x 3 vertex sets (t3d_box_iou3d)
    if (tid < 54) sm.vert[tid / 27][(tid % 27) / 3][tid % 3] = verts[(size_t)b * 54 + tid];
  } else {
    const float* kp = (wave == 0 ? pred : gt) + (size_t)b * 18;
    lift_wave(kp, portrait, fx, fy, cx, cy, sm.A[wave], sm.V[wave], sm.vert[wave], lane);
  }
  __syncthreads();
  if (lifted && tid < 54) lifted[(size_t)b * 54 + tid] = sm.vert[tid / 27][(tid % 27) / 3][tid % 3];
  if (!iou || wave != 0) return;

  Fit f1, f2;
  fit_box(sm.vert[0], f1);
  fit_box(sm.vert[1], f2);
  const double d1 = det3(f1.R), d2 = det3(f2.R);
  const bool ok = isfinite(d1) && isfinite(d2) && d1 != 0.0 && d2 != 0.0 && isfinite(f1.vol) && isfinite(f2.vol);
  double contrib = 0.0;
  if (ok && lane < 12) {
    // inverse of R1 (adjugate / det)
    double inv[3][3];
    const double (*R)[3] = f1.R;
    inv[0][0] = (R[1][1] * R[2][2] - R[1][2] * R[2][1]) / d1;
    inv[0][1] = (R[0][2] * R[2][1] - R[0][1] * R[2][2]) / d1;
    inv[0][2] = (R[0][1] * R[1][2] - R[0][2] * R[1][1]) / d1;
    inv[1][0] = (R[1][2] * R[2][0] - R[1][0] * R[2][2]) / d1;
    inv[1][1] = (R[0][0] * R[2][2] - R[0][2] * R[2][0]) / d1;
    inv[1][2] = (R[0][2] * R[1][0] - R[0][0] * R[1][2]) / d1;
    inv[2][0] = (R[1][0] * R[2][1] - R[1][1] * R[2][0]) / d1;
    inv[2][1] = (R[0][1] * R[2][0] - R[0][0] * R[2][1]) / d1;
    inv[2][2] = (R[0][0] * R[1][1] - R[0][1] * R[1][0]) / d1;
    // box 2 in box 1's frame: centre c, half-edge vectors e[a] (columns of inv(R1) R2 scaled by s2/2)
    double c[3], e[3][3], h[3];
    for (int i = 0; i < 3; ++i) {
      h[i] = 0.5 * f1.s[i];
      c[i] = inv[i][0] * (f2.t[0] - f1.t[0]) + inv[i][1] * (f2.t[1] - f1.t[1]) + inv[i][2] * (f2.t[2] - f1.t[2]);
      for (int a = 0; a < 3; ++a)
        e[a][i] = (inv[i][0] * f2.R[0][a] + inv[i][1] * f2.R[1][a] + inv[i][2] * f2.R[2][a]) * (0.5 * f2.s[a]);
    }
    // unit outward normals of box 2's '+a' faces and their half-widths
    double n2[3][3], w2[3];
    for (int a = 0; a < 3; ++a) {
      const int b1 = (a + 1) % 3, b2 = (a + 2) % 3;
      double nx = e[b1][1] * e[b2][2] - e[b1][2] * e[b2][1], ny = e[b1][2] * e[b2][0] - e[b1][0] * e[b2][2],
             nz = e[b1][0] * e[b2][1] - e[b1][1] * e[b2][0];
      const double nl = sqrt(nx * nx + ny * ny + nz * nz);
      nx /= nl; ny /= nl; nz /= nl;
      double w = nx * e[a][0] + ny * e[a][1] + nz * e[a][2];
      if (w < 0.0) { nx = -nx; ny = -ny; nz = -nz; w = -w; }
      n2[a][0] = nx; n2[a][1] = ny; n2[a][2] = nz;
      w2[a] = w;
    }
    double (*pa)[3] = sm.poly[lane][0];
    double (*pb)[3] = sm.poly[lane][1];
    int n = 4;
    double fn[3], fd;       // outward unit normal and plane offset of this lane's face
    bool dropped = false;
    if (lane < 6) {
      // face `lane` of box 2 (quad of its vertices), clipped by the 6 planes of the axis-aligned box 1
      const int a = lane >> 1;
      const double sg = (lane & 1) ? -1.0 : 1.0;
      for (int i = 0; i < 3; ++i) fn[i] = sg * n2[a][i];
      fd = fn[0] * (c[0] + sg * e[a][0]) + fn[1] * (c[1] + sg * e[a][1]) + fn[2] * (c[2] + sg * e[a][2]);
      for (int k = 0; k < 4; ++k) {
        const int vi = c_faces[lane][k] - 1;
        for (int i = 0; i < 3; ++i)
          pa[k][i] = c[i] + c_sign[vi][0] * e[0][i] + c_sign[vi][1] * e[1][i] + c_sign[vi][2] * e[2][i];
      }
      // coincident with a face of box 1 with the same outward direction: that face already carries this area
      for (int ax = 0; ax < 3 && !dropped; ++ax) {
        for (int sgi = 0; sgi < 2; ++sgi) {
          const double s1 = sgi ? -1.0 : 1.0;
          bool on = true;
          for (int k = 0; k < 4; ++k) on = on && fabs(s1 * pa[k][ax] - h[ax]) <= PLANE_EPS;
          if (on && s1 * fn[ax] > 0.0) dropped = true;
        }
      }
      for (int ax = 0; ax < 3 && !dropped; ++ax) {
        double nn[3] = {0.0, 0.0, 0.0};
        nn[ax] = 1.0;
        n = clip_poly(pa, n, pb, nn, h[ax]);
        nn[ax] = -1.0;
        n = clip_poly(pb, n, pa, nn, h[ax]);
      }
    } else {
      // face (lane - 6) of box 1, clipped by the 6 half-spaces of box 2
      const int f = lane - 6, a = f >> 1;
      const double sg = (f & 1) ? -1.0 : 1.0;
      fn[0] = fn[1] = fn[2] = 0.0;
      fn[a] = sg;
      fd = h[a];
      for (int k = 0; k < 4; ++k) {
        const int vi = c_faces[f][k] - 1;
        for (int i = 0; i < 3; ++i) pa[k][i] = c_sign[vi][i] * h[i];
      }
      for (int ax = 0; ax < 3; ++ax) {
        double nn[3] = {n2[ax][0], n2[ax][1], n2[ax][2]};
        const double off = nn[0] * c[0] + nn[1] * c[1] + nn[2] * c[2];
        n = clip_poly(pa, n, pb, nn, off + w2[ax]);
        nn[0] = -nn[0]; nn[1] = -nn[1]; nn[2] = -nn[2];
        n = clip_poly(pb, n, pa, nn, -off + w2[ax]);
      }
    }
    if (!dropped && n >= 3) {
      double ax_ = 0.0, ay_ = 0.0, az_ = 0.0;
      for (int k = 1; k + 1 < n; ++k) {
        const double ux = pa[k][0] - pa[0][0], uy = pa[k][1] - pa[0][1], uz = pa[k][2] - pa[0][2];
        const double vx = pa[k + 1][0] - pa[0][0], vy = pa[k + 1][1] - pa[0][1], vz = pa[k + 1][2] - pa[0][2];
        ax_ += uy * vz - uz * vy;
        ay_ += uz * vx - ux * vz;
        az_ += ux * vy - uy * vx;
      }
      contrib = fd * 0.5 * sqrt(ax_ * ax_ + ay_ * ay_ + az_ * az_) / 3.0;
    }
  }
  const double vframe = wave_sum_d(contrib);
  if (lane == 0) {
    double r = 0.0;
    if (ok) {
      const double inter = fabs(d1) * vframe;
      if (inter > 0.0 && isfinite(inter)) r = inter / (f1.vol + f2.vol - inter);
      if (!isfinite(r)) r = 0.0;
    }
    iou[b] = r;
    if (total) atomicAdd(total, r);
  }
}

}  // namespace

extern "C" int t3d_iou3d(const float* pred_kp, const float* gt_kp, int B, int portrait, const double* camera_ndc,
                         double* iou, double* total, double* lifted, void* stream) {
  if (!pred_kp || !gt_kp || B <= 0 || (!iou && !lifted)) return T3D_ERR_ARG;
  // NDC camera of geometry.py:29-37 applied to the default matrix (:16-19): fx = fy = 2, cx = cy = 0
  double fx = 2.0, fy = 2.0, cx = 0.0, cy = 0.0;
  if (camera_ndc) { fx = camera_ndc[0]; fy = camera_ndc[1]; cx = camera_ndc[2]; cy = camera_ndc[3]; }
  T3D_LAUNCH(iou3d_kernel, dim3(B), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), pred_kp, gt_kp, portrait,
                     fx, fy, cx, cy, iou, total, lifted, nullptr);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_box_iou3d(const double* verts, int B, double* iou, double* total, void* stream) {
  if (!verts || !iou || B <= 0) return T3D_ERR_ARG;
  T3D_LAUNCH(iou3d_kernel, dim3(B), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), nullptr, nullptr, 0, 2.0,
                     2.0, 0.0, 0.0, iou, total, nullptr, verts);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
