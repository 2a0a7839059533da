#!/usr/bin/env python3
"""time per view of one workload in the default mode (best of three solves) with the half-sweep series:
   python scripts/gpu_r06_time.py <name> ...     names: C, B, D, A, box19, box11, colour, C@fast, C@literal, ...
GIPUMA_HIP_LIB selects a variant library (scripts/build_variant.sh); the A/B switches of DESIGN.md 5 apply."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GIPUMA_HIP_EXPERIMENTS", "1")
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

for name in sys.argv[1:] or ["C"]:
    wl, _, mode = name.partition("@")
    kw, cfg = {}, wl
    if wl.startswith("box"):
        cfg, kw = "C", dict(blocksize=int(wl[3:]))
    elif wl == "colour":
        cfg, kw = "C", dict(colour=True)
    gs, _ = synth.build_problem(cfg, **kw)
    skw = {"fast": dict(fast=True), "literal": dict(literal=True)}.get(mode, {})
    with Session(gs, **skw) as s:
        s.solve(timing=True)
        best = None
        for _ in range(3):
            t = s.solve(timing=True)
            ms, _n = s.launch_times()
            if best is None or t.ms_total < best[0]:
                best = (t.ms_total, t.ms_init, list(ms))
    print("%-14s %-24s %8.2f ms per view = %6.2f Mpix/s  (init %.2f; half-sweeps %s)"
          % (name, os.path.basename(os.environ.get("GIPUMA_HIP_LIB", "default")), best[0], gs.rows * gs.cols / best[0] / 1e3,
             best[1], " ".join("%.2f" % m for m in best[2])), flush=True)
