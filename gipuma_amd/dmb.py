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
