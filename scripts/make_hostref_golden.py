#!/usr/bin/env python3
"""tests/golden/hostref_dtu.json: what the reference's OWN camera front-end (getCameraParameters,
cameraGeometryUtils.h:174-353, and selectViews, main.cpp:430-499, compiled from /root/reference against the
functional mini OpenCV by oracle/ref_shim/hostref/build_hostref.sh) produces for the 64 DTU cameras of
/root/reference/data/dtu/calib, reference view 15 first -- every Camera_cu field the hot path reads, the selected
subset and the automatic depth range.  The fixture keeps SURVEY 8f row N1 pinned where /root/reference is absent.
    python scripts/make_hostref_golden.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CALIB = "/root/reference/data/dtu/calib/"
EXE = os.path.join(ROOT, "oracle", "_ref", "hostref")


def run(folder, names, cam_scale=1.0, cols=1600, rows=1200, min_angle=10, max_angle=30, max_views=100, dmin=-1, dmax=-1):
    out = subprocess.run([EXE, folder, repr(float(cam_scale)), str(cols), str(rows), str(min_angle), str(max_angle),
                          str(max_views), str(dmin), str(dmax)] + list(names), capture_output=True, text=True, check=True)
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def dtu_order(ref=15):
    names = ["rect_%03d_3_r5000.png" % i for i in range(1, 65)]
    return [names[ref - 1]] + [n for i, n in enumerate(names) if i != ref - 1], [ref] + [i for i in range(1, 65) if i != ref]


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    names, ids = dtu_order()
    out = {"what": __doc__.split("\n    python")[0], "view_ids": ids,
           "scale_1": run(CALIB, names), "scale_4": run(CALIB, names, cam_scale=4.0, cols=400, rows=300)}
    path = os.path.join(ROOT, "tests", "golden", "hostref_dtu.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print("wrote %s (%d bytes)" % (path, os.path.getsize(path)))
