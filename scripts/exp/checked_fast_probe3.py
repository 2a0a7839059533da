#!/usr/bin/env python3
"""free-running (no host wait between the half-sweeps): fused against two launches in the FAST flavour of GIPUMA_HIP_LIB after n
half-sweeps, n = 5, 6; where the differing pixels are"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GIPUMA_HIP_EXPERIMENTS", "1")
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

mode = dict(fast=True) if (sys.argv[1:] or ["fast"])[0] == "fast" else {}
gs, _ = synth.build_problem("C")
for n_hs in (5, 6):
    out = []
    for fused in ("1", "0", "1"):
        os.environ["GIPUMA_HIP_GROUP_FUSED"] = fused
        with Session(gs, **mode) as s:
            s.init_planes()
            for hs in range(n_hs):
                s.sweep(hs // 2, hs % 2)
            out.append(s.get_state())
    for name, (na, ca), (nb, cb) in (("fused vs two launches", out[0], out[1]), ("fused vs fused again", out[0], out[2])):
        dp = ~(na.view(np.uint32) == nb.view(np.uint32)).all(-1)
        dc = ca.view(np.uint32) != cb.view(np.uint32)
        bad = dp | dc
        line = "after %d half-sweeps, %s: planes differ at %d pixels, costs at %d" % (n_hs, name, int(dp.sum()), int(dc.sum()))
        if bad.any():
            ys, xs = np.nonzero(bad)
            tiles = set(zip((ys // 16).tolist(), (xs // 32).tolist()))
            ly, lx = ys % 16, xs % 32
            tid = ly * 16 + (lx >> 1)
            line += "; %d tiles of %d; by wavefront %s; by tile row %s; colour of the pixels %s; first (y, x): %s" % (
                len(tiles), (gs.rows // 16) * (gs.cols // 32), np.bincount(tid >> 6, minlength=4).tolist(), np.bincount(ly, minlength=16).tolist(),
                np.bincount((ys + xs) & 1, minlength=2).tolist(), list(zip(ys[:4].tolist(), xs[:4].tolist())))
            yy, xx = int(ys[0]), int(xs[0])
            line += "\n    first pixel: fused plane %s cost %r | other plane %s cost %r" % (na[yy, xx], float(ca[yy, xx]), nb[yy, xx], float(cb[yy, xx]))
        print(line, flush=True)
