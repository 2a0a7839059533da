#!/usr/bin/env python3
"""half-sweep times of a config-C solve in the tolerance-judged flavour (single-path kernels), for occupancy experiments on
pm::push_kernel with variant libraries (GIPUMA_HIP_LIB under GIPUMA_HIP_EXPERIMENTS=1).  Needs a GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

fast = "--exact" not in sys.argv
gs, _ = synth.build_problem("C")
with Session(gs, fast=fast) as s:
    s.solve(timing=True)
    for _ in range(3):
        t = s.solve(timing=True)
        ms, _ = s.launch_times()
        print("%s total %.2f init %.2f half-sweeps %s" % ("fast" if fast else "exact", t.ms_total, t.ms_init, " ".join("%.2f" % x for x in ms)), flush=True)
