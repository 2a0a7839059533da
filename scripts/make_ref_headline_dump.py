#!/usr/bin/env python3
"""The reference's OWN device code (oracle/_ref, CPU) on the headline configuration at 1600x1216 (config C's 1600x1200 padded
to the reference's 32-pixel tiles): final maps and costs to scratch_big/ref_configC_1600x1216.npz (40 MB, not committed; it
travels to the GPU box with the snapshot), for scripts/gpu_r05_literal_headline.sh.  About 25 minutes on 8 cores."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipuma_amd import synth  # noqa: E402
from tests import ref_lib  # noqa: E402

gs, _ = synth.build_problem("C", cols=1600, rows=1216)
t0 = time.time()
rn, rc = ref_lib.RefState(gs, tex_mode=0).run()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scratch_big", "ref_configC_1600x1216.npz")
np.savez_compressed(out, norm4=rn, cost=rc)
print("reference's own code, config C 1600x1216, 10 views, 8 iterations: %.0f s -> %s (%.1f MB)" % (time.time() - t0, out, os.path.getsize(out) / 1e6))
