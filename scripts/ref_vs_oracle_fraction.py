#!/usr/bin/env python3
"""Measured agreement of the oracle restatement with the reference's OWN code (oracle/_ref, CPU) on
free-running solves: the fraction of pixels inside the north_star tolerance (depth 1e-4 relative,
unit normals 1e-3) and the fraction that is bit-identical.  CPU only; minutes.
    python scripts/ref_vs_oracle_fraction.py A            # BASELINE config A in full
    python scripts/ref_vs_oracle_fraction.py B 320 240    # config B's parameters on a 320x240 frame"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipuma_amd import synth  # noqa: E402
from tests import ref_lib  # noqa: E402
from tests.oracle_lib import OracleState  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
over = {}
if len(sys.argv) > 3:
    over = dict(cols=int(sys.argv[2]), rows=int(sys.argv[3]))
# optional 4th argument: the texture-filter model the reference's code runs with -- 0: fp32 lerp weights (M1, the
# model the oracle and the kernels use); 1: weights rounded to 8 fractional bits, the 1.8 fixed-point weights the
# CUDA programming guide documents for cudaFilterModeLinear (main.cpp:644-648)
tex_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
gs, info = synth.build_problem(cfg, **over)
print("reference texture filter model: %s" % ("8-bit fixed-point weights (CUDA's documented filter)" if tex_mode else "fp32 weights (M1)"))
t0 = time.time()
rn, rc = ref_lib.RefState(gs, tex_mode=tex_mode).run()
ref_lib.lib().ref_set_tex_mode(0)
t1 = time.time()
on, oc = OracleState(gs).run()
t2 = time.time()
d_rel = np.abs(rn[..., 3] - on[..., 3]) / np.maximum(np.abs(rn[..., 3]), 1e-30)
n_err = np.abs(rn[..., :3] - on[..., :3]).max(-1)
ok = (d_rel < 1e-4) & (n_err < 1e-3)
same = (rn.view(np.uint32) == on.view(np.uint32)).all(-1)
gt = info["gt_depth"]
print("config %s %dx%d, %d source views, box %d, %d iterations: reference (own code, CPU) %.0f s, oracle %.0f s"
      % (cfg, gs.cols, gs.rows, len(gs.selected), gs.params.box_hsize, gs.params.iterations, t1 - t0, t2 - t1))
print("  pixels within tolerance (depth 1e-4 rel, normal 1e-3): %.4f %%" % (100 * ok.mean()))
print("  pixels bit-identical (norm4):                          %.4f %%" % (100 * same.mean()))
print("  cost plane identical where the planes are:             %.4f %%"
      % (100 * (rc.view(np.uint32) == oc.view(np.uint32))[same].mean()))
print("  within 1 %% of the analytic ground truth: reference %.4f, oracle %.4f"
      % ((np.abs(rn[..., 3] - gt) / gt < 0.01).mean(), (np.abs(on[..., 3] - gt) / gt < 0.01).mean()))
