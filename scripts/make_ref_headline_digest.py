#!/usr/bin/env python3
"""Digest fixtures of the reference's OWN device code (oracle/_ref, CPU, fp32 filter weights) at real frame sizes, so that the
parity claims made at those sizes are tests the driver runs (tests/test_headline_parity.py), not prose:

    config C   1600x1216 (its 1600x1200 padded to the reference's 32-pixel tiles), 10 views, box 15, 8 iterations
    config B   640x480, 6 views, box 11, 8 iterations
    config D   800x608, 20 views, box 25, 8 iterations (its parameters on a quarter-size frame: the reference needs 32 min
               of 6 cores for it here; 1600x1216 would be 2 hours)

For each: tests/golden/ref_headline_<cfg>.npz with
    band_sha256   SHA-256 of the final norm4 (world normal, depth) and of the final cost of every band of 64 rows
                  (the loop whose result this is: /root/reference/gipuma.cu:1911-1944)
    sample_idx, sample_norm4, sample_cost   every 97th pixel, raw
    meta          the problem (so that the test rebuilds exactly it) and the wall time of the reference
The full dump goes to scratch_big/ref_config<cfg>_<cols>x<rows>.npz (not committed; it travels to the GPU box with the snapshot
while it exists) for the per-pixel statistics of scripts/gpu_r06_headline.sh.

    python scripts/make_ref_headline_digest.py B C D     (OMP_NUM_THREADS limits the reference's block loop)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gipuma_amd import synth  # noqa: E402

SIZES = {"C": (1600, 1216), "B": (640, 480), "D": (800, 608)}
BAND = 64
STRIDE = 97


def band_digests(n4, c):
    out = []
    for y0 in range(0, n4.shape[0], BAND):
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(n4[y0:y0 + BAND]).tobytes())
        h.update(np.ascontiguousarray(c[y0:y0 + BAND]).tobytes())
        out.append(h.hexdigest())
    return out


def main(cfgs):
    from tests import ref_lib
    os.makedirs(os.path.join(ROOT, "scratch_big"), exist_ok=True)
    for cfg in cfgs:
        cols, rows = SIZES[cfg]
        gs, _ = synth.build_problem(cfg, cols=cols, rows=rows)
        t0 = time.time()
        rn, rc = ref_lib.RefState(gs, tex_mode=0).run()
        dt = time.time() - t0
        np.savez_compressed(os.path.join(ROOT, "scratch_big", "ref_config%s_%dx%d.npz" % (cfg, cols, rows)), norm4=rn, cost=rc)
        idx = np.arange(0, rows * cols, STRIDE)
        meta = dict(cfg=cfg, cols=cols, rows=rows, band=BAND, stride=STRIDE, seconds=round(dt, 1),
                    iterations=int(gs.params.iterations), n_views=len(gs.selected),
                    what="final maps of /root/reference/gipuma.cu compiled for the CPU (oracle/ref_shim/build_ref.sh), fp32 filter weights")
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_headline_%s.npz" % cfg),
                            band_sha256=np.array(band_digests(rn, rc)),
                            sample_idx=idx.astype(np.int64),
                            sample_norm4=rn.reshape(-1, 4)[idx], sample_cost=rc.reshape(-1)[idx],
                            meta=np.array(json.dumps(meta)))
        print("config %s %dx%d: reference %.0f s, %d bands, %d samples" % (cfg, cols, rows, dt, (rows + BAND - 1) // BAND, len(idx)), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["B", "C", "D"])
