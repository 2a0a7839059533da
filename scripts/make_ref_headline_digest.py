#!/usr/bin/env python3
"""Digest fixtures of the reference's OWN device code (oracle/_ref, CPU, fp32 filter weights) at real frame sizes, so that the
parity claims made at those sizes are tests the driver runs (tests/test_headline_parity.py), not prose:

    config C   1600x1216 (its 1600x1200 padded to the reference's 32-pixel tiles), 10 views, box 15, 8 iterations
    config B   640x480, 6 views, box 11, 8 iterations
    config D   800x608, 20 views, box 25, 8 iterations (its parameters on a quarter-size frame: the reference needs 32 min
               of 6 cores for it here; 1600x1216 would be 2 hours)
    C4         -color_processing (T = float4) with config C's parameters on 320x256

For each: tests/golden/ref_headline_<cfg>.npz with
    band_sha256   SHA-256 of the final norm4 (world normal, depth) and of the final cost of every band of 64 rows
                  (the loop whose result this is: /root/reference/gipuma.cu:1911-1944)
    sample_idx, sample_norm4, sample_cost   every 97th pixel, raw
    meta          the problem (so that the test rebuilds exactly it) and the wall time of the reference
The full dump goes to scratch_big/ref_config<cfg>_<cols>x<rows>.npz (not committed; it travels to the GPU box with the snapshot
while it exists) for the per-pixel statistics of scripts/gpu_r06_headline.sh.

    meta also carries the SHA-256 of the input images and cameras: the test refuses to compare when the problem this tree builds
    on the test machine is not the one the reference solved; config B's images are embedded (images_u8) because they are not
    machine-independent.

    python scripts/make_ref_headline_digest.py [--reuse] B C D C4    (OMP_NUM_THREADS limits the reference's block loop;
                                                                      --reuse: digest the dumps in scratch_big/ again)
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

SIZES = {"C": (1600, 1216), "B": (640, 480), "D": (800, 608), "C4": (320, 256)}
COLOUR = {"C4": "C"}  # -color_processing (T = float4, /root/reference/gipuma.cu:1965-1968) with config C's parameters
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


EMBED_IMAGES = {"B"}  # configurations whose synthetic images are not bit-identical on every machine (see below)


def problem_hashes(gs):
    import ctypes as C
    hi, hc = hashlib.sha256(), hashlib.sha256()
    for im in gs.images:
        hi.update(np.ascontiguousarray(im).tobytes())
    for k in range(gs.desc.n_images):
        hc.update(bytes(C.string_at(C.addressof(gs.desc.cameras[k]), C.sizeof(gs.desc.cameras[k]))))
    return hi.hexdigest(), hc.hexdigest()


def main(cfgs, reuse):
    from tests import ref_lib
    os.makedirs(os.path.join(ROOT, "scratch_big"), exist_ok=True)
    for cfg in cfgs:
        cols, rows = SIZES[cfg]
        gs, _ = synth.build_problem(COLOUR.get(cfg, cfg), cols=cols, rows=rows, colour=cfg in COLOUR)
        dump = os.path.join(ROOT, "scratch_big", "ref_config%s_%dx%d.npz" % (cfg, cols, rows))
        if reuse and os.path.exists(dump):  # (--reuse: digest an existing dump of THIS machine's problem again)
            z = np.load(dump)
            # (the dumps of round 6 were written without their wall time: 4 threads of this container, other work running)
            known = {"B": 164.8, "C": 2536.0, "D": 3205.0, "C4": 719.0}
            rn, rc, dt = z["norm4"], z["cost"], float(z["seconds"]) if "seconds" in z else known.get(cfg, -1.0)
        else:
            t0 = time.time()
            rn, rc = ref_lib.RefState(gs, tex_mode=0).run()
            dt = time.time() - t0
            np.savez_compressed(dump, norm4=rn, cost=rc, seconds=dt)
        idx = np.arange(0, rows * cols, STRIDE)
        h_img, h_cam = problem_hashes(gs)
        meta = dict(cfg=COLOUR.get(cfg, cfg), colour=cfg in COLOUR, cols=cols, rows=rows, band=BAND, stride=STRIDE, seconds=round(dt, 1),
                    iterations=int(gs.params.iterations), n_views=len(gs.selected), images_sha256=h_img, cameras_sha256=h_cam,
                    what="final maps of /root/reference/gipuma.cu compiled for the CPU (oracle/ref_shim/build_ref.sh), fp32 filter weights")
        extra = {}
        if cfg in EMBED_IMAGES:
            # The ring scenes' outermost views contain a few hundred pixels whose rays graze the surface: the renderer's Newton
            # iteration is chaotic there, and two machines (different libm / BLAS code paths) render them differently.  A digest
            # of the reference's OUTPUT is only meaningful for bit-identical INPUT, so the 8-bit images travel with it.
            extra["images_u8"] = np.stack([np.asarray(im).astype(np.uint8) for im in gs.images])
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_headline_%s.npz" % cfg),
                            band_sha256=np.array(band_digests(rn, rc)),
                            sample_idx=idx.astype(np.int64),
                            sample_norm4=rn.reshape(-1, 4)[idx], sample_cost=rc.reshape(-1)[idx],
                            meta=np.array(json.dumps(meta)), **extra)
        print("config %s %dx%d: reference %.0f s, %d bands, %d samples" % (cfg, cols, rows, dt, (rows + BAND - 1) // BAND, len(idx)), flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--reuse"]
    main(args or ["B", "C", "D", "C4"], "--reuse" in sys.argv[1:])
