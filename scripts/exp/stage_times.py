#!/usr/bin/env python3
"""EXPERIMENT (GPU): wall time of propagation-only / refinement-only / fused half-sweeps of config C
at a given iteration, for the kernel variant selected by GIPUMA_HIP_TUNE (history rule off so that
partial-stage launches do not change the amount of work)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
warm_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
gs, _ = synth.build_problem(cfg, device="cuda:0", keep_on_device=True)


def timed(s, it, colour, stages):
    s.sync()
    t0 = time.perf_counter()
    s.sweep(it, colour, stages)
    s.sync()
    return (time.perf_counter() - t0) * 1e3


with Session(gs) as s:
    s.init_planes()
    for it in range(warm_iters):
        s.sweep(it, 0)
        s.sweep(it, 1)
    out = []
    it = warm_iters
    out.append(("it%d black prop" % it, timed(s, it, 0, 3)))
    out.append(("it%d black refine" % it, timed(s, it, 0, 4)))
    out.append(("it%d red fused" % it, timed(s, it, 1, 7)))
    it += 1
    out.append(("it%d black fused" % it, timed(s, it, 0, 7)))
    out.append(("it%d red prop" % it, timed(s, it, 1, 3)))
    out.append(("it%d red refine" % it, timed(s, it, 1, 4)))
    print("TUNE=%s: " % os.environ.get("GIPUMA_HIP_TUNE", "0") + "  ".join("%s %.2f" % o for o in out))
