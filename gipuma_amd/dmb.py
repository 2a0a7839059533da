"""The reference's `.dmb` dumps (fileIoUtils.h:247-368): int32 {type=1, h, w, nb} followed by
h*w*nb float32, row-major.  `disp.dmb` holds norm4.w (depth), `normals.dmb` the world normals
(main.cpp:1001-1015); these are the "CPU-readable dumps" external tools such as fusibile read."""
import numpy as np


def write_dmb(path, arr):
    a = np.ascontiguousarray(arr, dtype=np.float32)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, nb = a.shape
    with open(path, "wb") as f:
        np.array([1, h, w, nb], dtype=np.int32).tofile(f)
        a.tofile(f)


def read_dmb(path):
    with open(path, "rb") as f:
        hdr = np.fromfile(f, dtype=np.int32, count=4)
        if hdr[0] != 1:
            raise ValueError("%s: only float dmb (type 1) is defined" % path)
        h, w, nb = int(hdr[1]), int(hdr[2]), int(hdr[3])
        data = np.fromfile(f, dtype=np.float32, count=h * w * nb)
    out = data.reshape(h, w, nb)
    return out[:, :, 0] if nb == 1 else out


_PLY_VERTEX = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
                        ("red", "u1"), ("green", "u1"), ("blue", "u1")])
_PLY_HEADER = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n"
               "property float x\nproperty float y\nproperty float z\n"
               "property float nx\nproperty float ny\nproperty float nz\n"
               "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")


def ply_points(depth, M_inv, P_col34):
    """get3Dpoint (cameraGeometryUtils.h:51-61) for every pixel, float32 like the reference:
    M_inv * (depth * (x, y, 1) - P.col(3)).  Returns (rows, cols, 3)."""
    rows, cols = depth.shape
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float32)
    d = depth.astype(np.float32)
    p4 = np.asarray(P_col34, dtype=np.float32)
    M = np.asarray(M_inv, dtype=np.float32).reshape(3, 3)
    v = np.stack([d * x - p4[0], d * y - p4[1], d - p4[2]], axis=-1)
    with np.errstate(invalid="ignore", over="ignore"):  # non-finite depths become (0, 0, 0) below
        X = (M[None, None, :, 0] * v[..., 0:1] + M[None, None, :, 1] * v[..., 1:2]) + M[None, None, :, 2] * v[..., 2:3]
    X = X.astype(np.float32)
    X[~np.isfinite(X).all(axis=-1)] = 0.0
    return X


def write_ply_binary(path, depth, normals, gray, M_inv, P_col34):
    """storePlyFileBinary (displayUtils.h:78-159): one vertex per pixel -- world point, normal and
    the gray value three times -- in the reference's loop order (x outer, y inner).  M_inv / P_col34
    belong to the NOT re-centred camera (getCameraParameters(..., false), main.cpp:1021)."""
    rows, cols = depth.shape
    v = np.zeros((cols, rows), dtype=_PLY_VERTEX)
    X = ply_points(depth, M_inv, P_col34)
    n = np.asarray(normals, dtype=np.float32)
    for k, name in enumerate(("x", "y", "z")):
        v[name] = X[:, :, k].T
    for k, name in enumerate(("nx", "ny", "nz")):
        v[name] = n[:, :, k].T
    g = np.asarray(gray, dtype=np.float32).astype(np.uint8).T
    v["red"] = v["green"] = v["blue"] = g
    with open(path, "wb") as f:
        f.write((_PLY_HEADER % (rows * cols)).encode())
        v.tofile(f)


def read_ply_binary(path):
    """-> structured array of the vertices, in file order"""
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: no end_header" % path)
            if line.startswith(b"element vertex"):
                n = int(line.split()[2])
            if line.strip() == b"end_header":
                break
        return np.fromfile(f, dtype=_PLY_VERTEX, count=n)
