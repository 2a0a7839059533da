#!/usr/bin/env python3
"""where the fused kernel of a library's FAST flavour (GIPUMA_HIP_LIB) first leaves the two-launch schedule on config C:
both sessions step through the same half-sweeps; after each one their states are compared and the fused session is put
back on the two-launch session's state, so every half-sweep is judged by itself.  Prints where the differing pixels are."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GIPUMA_HIP_EXPERIMENTS", "1")
from gipuma_amd import abi, synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

mode = dict(fast=True) if (sys.argv[1:] or ["fast"])[0] == "fast" else {}
gs, _ = synth.build_problem("C")
os.environ["GIPUMA_HIP_GROUP_FUSED"] = "1"
a = Session(gs, **mode)
os.environ["GIPUMA_HIP_GROUP_FUSED"] = "0"
b = Session(gs, **mode)
del os.environ["GIPUMA_HIP_GROUP_FUSED"]
print("schedules:", a.schedule(), b.schedule())
a.init_planes()
b.init_planes()
for hs in range(10):
    it, colour = hs // 2, hs % 2
    a.sweep(it, colour)
    b.sweep(it, colour)
    na, ca = a.get_state()
    nb, cb = b.get_state()
    dp = ~(na.view(np.uint32) == nb.view(np.uint32)).all(-1)
    dc = na.view(np.uint32)[..., 0] * 0 != 0
    dc = ca.view(np.uint32) != cb.view(np.uint32)
    bad = dp | dc
    line = "half-sweep %d (colour %d): planes differ at %d pixels, costs at %d" % (hs, colour, int(dp.sum()), int(dc.sum()))
    if bad.any():
        ys, xs = np.nonzero(bad)
        ty, tx = ys // 16, xs // 32
        tiles = set(zip(ty.tolist(), tx.tolist()))
        ly, lx = ys % 16, xs % 32
        tid = ly * 16 + (lx >> 1)
        waves = np.bincount(tid >> 6, minlength=4)
        line += "; %d tiles touched of %d; by wavefront of the tile %s; rows of the tile %s; first (y, x): %s" % (
            len(tiles), (gs.rows // 16) * (gs.cols // 32), waves.tolist(), np.bincount(ly, minlength=16).tolist(),
            list(zip(ys[:5].tolist(), xs[:5].tolist())))
        # a second fused run of the same half-sweep from the same state: the same pixels?
    print(line, flush=True)
    a.set_state(nb, cb)  # (continue from the two-launch session's state)
    if bad.any() and hs >= 6:
        break
a.close()
b.close()
