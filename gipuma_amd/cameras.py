"""OpenCV-free camera front-end: what the reference does between reading the calibration and
calling runcuda().

Restates, with numpy instead of OpenCV:
  * readPFileStrechaPmvs                      reference fileIoUtils.h:83-110
  * getCameraParameters                       reference cameraGeometryUtils.h:174-353
    (decomposeProjectionMatrix, re-centring on the reference camera, Camera_cu filling)
  * selectViews                               reference main.cpp:430-499
  * depth range -> disparity range            reference main.cpp:898-906

OpenCV's decomposeProjectionMatrix / Mat::inv are not available here and are not pinned by any
reference test; the decomposition below follows the documented algorithm (RQ of P[:, :3] with a
positive diagonal, camera centre = null vector of P) in float64 and rounds to float32 at the end.
"""
import ctypes as C
import math

import numpy as np

from . import abi


def read_p_file(path):
    """3x4 projection matrix in Strecha/PMVS/DTU text form (fileIoUtils.h:83-110): 3 lines of 4
    numbers, an optional leading 'CONTOUR' line is skipped."""
    rows = []
    with open(path) as f:
        for line in f:
            if "CONTOUR" in line:
                continue
            vals = line.split()
            if len(vals) >= 4:
                rows.append([float(v) for v in vals[:4]])
            if len(rows) == 3:
                break
    if len(rows) != 3:
        raise ValueError("%s: not a 3x4 projection matrix" % path)
    return np.array(rows, dtype=np.float64)


def rq3(M):
    """RQ decomposition M = K @ R, K upper triangular with positive diagonal entries 0 and 1
    (the convention of OpenCV's RQDecomp3x3 used by decomposeProjectionMatrix,
    cameraGeometryUtils.h:252), R orthogonal."""
    # QR of the row-reversed transpose gives RQ
    P = np.flipud(np.eye(3))
    q, r = np.linalg.qr((P @ M).T)
    K = P @ r.T @ P
    R = P @ q.T
    # fix signs: diag(K)[0], [1] > 0 (and [2] > 0 when possible)
    D = np.diag(np.where(np.diag(K) < 0, -1.0, 1.0))
    K = K @ D
    R = D @ R
    return K, R


def decompose_projection(P):
    """K, R, C (camera centre, 3-vector) of P = K [R | -R C]."""
    P = np.asarray(P, dtype=np.float64)
    # a projection matrix is defined up to scale, including its sign: with det(M) < 0 the RQ step would have to
    # return a reflection or a negative K[2][2] (what OpenCV's decomposeProjectionMatrix does,
    # cameraGeometryUtils.h:252: it fixes two diagonal signs only); both front-ends decompose -P instead and
    # get the camera of +P (the same deliberate deviation as gipuma_host.cpp decompose())
    if np.linalg.det(P[:, :3]) < 0:
        P = -P
    K, R = rq3(P[:, :3])
    # centre: right null vector of P (cameraGeometryUtils.h:259-261 divides T by T[3])
    _, _, vt = np.linalg.svd(P)
    Ch = vt[-1]
    Cc = Ch[:3] / Ch[3]
    return K, R, Cc


def scale_k(K, s):
    """scaleK, cameraGeometryUtils.h:136-147"""
    K = K.copy()
    K[0, 0] /= s
    K[1, 1] /= s
    K[0, 2] /= s
    K[1, 2] /= s
    return K


def camera_centre(P):
    """getCameraCenter, cameraGeometryUtils.h:22-49 (signed 3x3 minors), normalised."""
    c = np.empty(4)
    c[0] = np.linalg.det(P[:, [1, 2, 3]])
    c[1] = -np.linalg.det(P[:, [0, 2, 3]])
    c[2] = np.linalg.det(P[:, [0, 1, 3]])
    c[3] = -np.linalg.det(P[:, [0, 1, 2]])
    return c[:3] / c[3]


class CameraSet:
    """Result of get_camera_parameters: float64 matrices per view (index 0 = reference) plus the
    filled ctypes array for the C-ABI."""

    def __init__(self, n):
        self.n = n
        self.K = [None] * n       # per-view intrinsics (scaled)
        self.R = [None] * n       # relative rotation (reference = I)
        self.t = [None] * n       # relative translation (reference = 0)
        self.P = [None] * n       # K0 [R|t]  (K of camera 0 for all: cameraGeometryUtils.h:125,298)
        self.C = [None] * n       # centre of P
        self.R_orig = [None] * n  # original world->camera rotation
        self.C_orig = [None] * n  # original centre (world)
        self.f = 0.0
        self.c_array = (abi.Camera * n)()

    def view_vector(self, i, x, y):
        """getViewVector, cameraGeometryUtils.h:68-76"""
        M_inv = np.linalg.inv(self.P[i][:, :3])
        X = M_inv @ (np.array([x, y, 1.0]) - self.P[i][:, 3])
        v = X - self.C[i]
        return v / np.linalg.norm(v)


def get_camera_parameters(P_list, cam_scale=1.0):
    """getCameraParameters, cameraGeometryUtils.h:174-353, for a list of 3x4 matrices whose
    first entry is the reference view."""
    n = len(P_list)
    cs = CameraSet(n)
    Ks, Rs, Cs, ts = [], [], [], []
    for P in P_list:
        K, R, Cc = decompose_projection(P)
        Ks.append(K)
        Rs.append(R)
        Cs.append(Cc)
        ts.append(-R @ Cc)                                   # :264
    T0 = np.eye(4)
    T0[:3, :3] = Rs[0]
    T0[:3, 3] = ts[0]
    transform = np.linalg.inv(T0)                            # :109-115, :270-271
    K0 = scale_k(Ks[0], cam_scale)                           # :281
    cs.f = float(np.float32(K0[0, 0]))
    for i in range(n):
        Ki = scale_k(Ks[i], cam_scale)                       # :285
        Ti = np.eye(4)
        Ti[:3, :3] = Rs[i]
        Ti[:3, 3] = ts[i]
        Tt = Ti @ transform                                  # transformCamera, :117-128
        Pn = K0 @ Tt[:3, :4]
        cs.K[i] = Ki
        cs.R[i] = Tt[:3, :3]
        cs.t[i] = Tt[:3, 3]
        cs.P[i] = Pn
        cs.C[i] = camera_centre(Pn)                          # :130-133
        cs.R_orig[i] = Rs[i]
        cs.C_orig[i] = Cs[i]
        cam = cs.c_array[i]
        _put(cam.K, Ki)
        _put(cam.K_inv, np.linalg.inv(Ki))                   # :286
        _put(cam.R, cs.R[i])
        _put(cam.t, cs.t[i])
        _put(cam.M_inv, np.linalg.inv(Pn[:, :3]))            # :301
        _put(cam.P_col34, Pn[:, 3])                          # :339-345
        _put(cam.C, cs.C[i])
        _put(cam.R_orig_inv, np.linalg.inv(Rs[i]))           # :297
        cam.fx = K0[0, 0]                                    # :314-323
        cam.fy = K0[1, 1]
        cam.f = K0[0, 0]
        cam.alpha = float(np.float32(K0[0, 0]) / np.float32(K0[1, 1]))
        cam.baseline = 0.54                                  # :305
        cam.depth_min = 2.0                                  # camera.h:34
        cam.depth_max = 20.0                                 # camera.h:38
    return cs


def _put(dst, src):
    flat = np.asarray(src, dtype=np.float64).reshape(-1)
    for k in range(len(flat)):
        dst[k] = float(np.float32(flat[k]))


def select_views(cs, cols, rows, min_angle=5.0, max_angle=45.0, max_views=9,
                 depth_min=-1.0, depth_max=-1.0, view_selection=True, rng=None):
    """selectViews, main.cpp:430-499.  Returns (subset, depth_min, depth_max).

    The reference shuffles with srand(time(0)) when more than max_views survive (:491-496); here
    the shuffle uses the caller's numpy Generator (default: keep the first max_views, which is
    deterministic)."""
    x, y = cols // 2, rows // 2
    v_ref = cs.view_vector(0, x, y)
    lo = min_angle * math.pi / 180.0
    hi = max_angle * math.pi / 180.0
    subset = []
    dmin, dmax = 9999.0, 0.0
    for i in range(1, cs.n):
        v = cs.view_vector(i, x, y)
        baseline = float(np.linalg.norm(cs.C[0] - cs.C[i]))
        angle = math.acos(max(-1.0, min(1.0, float(v_ref @ v))))   # getAngle, mathUtils.h:16-24
        if lo < angle < hi:
            if view_selection:
                subset.append(i)
            dmin = min((baseline / 2.0) / math.sin(hi / 2.0), dmin)
            dmax = max((baseline / 2.0) / math.sin(lo / 2.0), dmax)
    if depth_min == -1:
        depth_min = dmin
    if depth_max == -1:
        depth_max = dmax
    if not view_selection:
        return list(range(1, cs.n)), depth_min, depth_max
    if len(subset) >= max_views:
        if rng is not None:
            rng.shuffle(subset)
        subset = subset[:max_views]
    return subset, depth_min, depth_max


def disparity_range(f, baseline, depth_min, depth_max):
    """main.cpp:905-906 with disparityDepthConversion (cameraGeometryUtils.h:103-107), fp32."""
    f32 = np.float32
    mind = f32(f) * f32(baseline) / f32(depth_max)
    maxd = f32(f) * f32(baseline) / f32(depth_min)
    return float(mind), float(maxd)
