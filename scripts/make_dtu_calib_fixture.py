#!/usr/bin/env python3
"""Collect the DTU projection matrices that ship with the reference (data/dtu/calib/*.P, public
DTU calibration, mm units) into one small JSON fixture, so tests and bench.py can run where
/root/reference does not exist (the GPU box).

    python scripts/make_dtu_calib_fixture.py [/root/reference]

Only the `rect_0XX_3_r5000` variant is kept (the three variants per position are identical,
SURVEY.md F3).  Values are stored as the decimal strings of the source files, parsed as float.
"""
import json
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
calib = os.path.join(ref, "data", "dtu", "calib")
out = {}
for k in range(1, 65):
    name = "rect_%03d_3_r5000.png.P" % k
    rows = []
    with open(os.path.join(calib, name)) as f:
        for line in f:
            v = line.split()
            if len(v) >= 4 and "CONTOUR" not in line:
                rows.append([float(x) for x in v[:4]])
    assert len(rows) == 3, name
    out["%d" % k] = rows
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gipuma_amd", "data",
                   "dtu_calib_r5000.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
with open(dst, "w") as f:
    json.dump({"source": "kysucix/gipuma data/dtu/calib/rect_0XX_3_r5000.png.P",
               "units": "mm", "image_size": [1600, 1200], "P": out}, f)
print("wrote", os.path.normpath(dst), len(out), "cameras")
