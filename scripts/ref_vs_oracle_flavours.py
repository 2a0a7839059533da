#!/usr/bin/env python3
"""Which rounding choice of the numerical model (M1 shared tap fraction, M2 x*(1/z), M3 fmaf placement)
separates the oracle from the reference's OWN code (oracle/_ref, CPU) on a free-running solve?  Runs the reference's
code once and the oracle once per flavour (oracle/gipuma_oracle.c: gipuma_oracle_set_flavour; 0 = the model the kernels
implement, 7 = the literal operation order of the reference's source) and prints, per flavour, the fraction of pixels
inside the north_star tolerance (depth 1e-4 relative, unit normals 1e-3), bit-identical planes and bit-identical costs.
CPU only.
    python scripts/ref_vs_oracle_flavours.py B                  # BASELINE config B in full (640x480)
    python scripts/ref_vs_oracle_flavours.py C 320 256 0,7      # config C's parameters on 320x256, flavours 0 and 7"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipuma_amd import synth  # noqa: E402
from tests import oracle_lib, ref_lib  # noqa: E402
from tests.oracle_lib import OracleState  # noqa: E402

NAMES = {0: "model M1+M2+M3 (what the kernels compute)", 1: "literal taps (M1 off)", 2: "literal division (M2 off)",
         4: "literal unfused multiply-adds (M3 off)", 3: "literal taps + division", 5: "literal taps + unfused",
         6: "literal division + unfused", 7: "all literal = the reference's source order"}


def compare(rn, rc, on, oc):
    d_rel = np.abs(rn[..., 3] - on[..., 3]) / np.maximum(np.abs(rn[..., 3]), 1e-30)
    n_err = np.abs(rn[..., :3] - on[..., :3]).max(-1)
    ok = (d_rel < 1e-4) & (n_err < 1e-3)
    same = (rn.view(np.uint32) == on.view(np.uint32)).all(-1)
    csame = rc.view(np.uint32) == oc.view(np.uint32)
    return ok.mean(), same.mean(), csame.mean()


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
    over = {}
    if len(sys.argv) > 3:
        over = dict(cols=int(sys.argv[2]), rows=int(sys.argv[3]))
    flavours = [int(f) for f in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1, 2, 4, 7]
    gs, info = synth.build_problem(cfg, **over)
    t0 = time.time()
    rn, rc = ref_lib.RefState(gs, tex_mode=0).run()
    t1 = time.time()
    print("config %s %dx%d, %d source views, box %d, %d iterations, disparity range %.4g .. %.4g; reference (own code, "
          "CPU, fp32 filter weights) %.0f s" % (cfg, gs.cols, gs.rows, len(gs.selected), gs.params.box_hsize,
                                                 gs.params.iterations, gs.params.min_disparity, gs.params.max_disparity,
                                                 t1 - t0))
    print("%-52s %12s %14s %14s %8s" % ("oracle flavour", "in tolerance", "planes ==", "costs ==", "s"))
    L = oracle_lib.lib()
    try:
        for f in flavours:
            L.gipuma_oracle_set_flavour(f)
            t2 = time.time()
            on, oc = OracleState(gs).run()
            ok, same, csame = compare(rn, rc, on, oc)
            print("%d %-50s %11.4f%% %13.4f%% %13.4f%% %8.0f" % (f, NAMES[f], 100 * ok, 100 * same, 100 * csame,
                                                              time.time() - t2), flush=True)
    finally:
        L.gipuma_oracle_set_flavour(-1)


if __name__ == "__main__":
    main()
