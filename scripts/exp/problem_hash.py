#!/usr/bin/env python3
"""SHA-256 of the inputs of a synthetic problem (images, cameras as the C-ABI sees them, parameters): are the problems the
tests build identical on two machines?  usage: problem_hash.py CFG [cols rows]"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gipuma_amd import synth  # noqa: E402


def problem_hash(gs):
    h = hashlib.sha256()
    hi = hashlib.sha256()
    for im in gs.images:
        hi.update(np.ascontiguousarray(im).tobytes())
    hc = hashlib.sha256()
    for k in range(gs.desc.n_images):
        hc.update(bytes(C.string_at(C.addressof(gs.desc.cameras[k]), C.sizeof(gs.desc.cameras[k]))))
    hp = hashlib.sha256(bytes(C.string_at(C.addressof(gs.desc.params), C.sizeof(gs.desc.params))))
    return hi.hexdigest()[:16], hc.hexdigest()[:16], hp.hexdigest()[:16]


if __name__ == "__main__":
    for spec in (sys.argv[1:] or ["B:640:480", "C:320:256", "C:1600:1216", "D:800:608", "A:320:256", "B:320:256"]):
        cfg, cols, rows = spec.split(":")
        gs, _ = synth.build_problem(cfg, cols=int(cols), rows=int(rows))
        print(spec, "images %s cameras %s params %s" % problem_hash(gs), flush=True)
