#!/usr/bin/env python3
"""box 19 (the reference's default window) on config C's frame: time per view for a few schedules of the round-6 kernel
families (pushed half-sweeps, column-per-lane half-sweeps, first plane-keyed half-sweep)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GIPUMA_HIP_EXPERIMENTS"] = "1"
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

box = int(sys.argv[1]) if len(sys.argv) > 1 else 19
gs, _ = synth.build_problem("C", blocksize=box)
for push, cols, grp in ((3, 3, 3), (4, 4, 4), (4, 3, 4), (2, 2, 2), (3, 0, 3), (4, 2, 4), (5, 4, 5)):
    os.environ.update(GIPUMA_HIP_PUSH_LAUNCHES=str(push), GIPUMA_HIP_COLS_LAUNCHES=str(cols), GIPUMA_HIP_GROUP_FROM=str(grp))
    with Session(gs) as s:
        s.solve(timing=True)
        best = min((s.solve(timing=True).ms_total, [round(x, 2) for x in s.launch_times()[0]]) for _ in range(2))
    print("box %d push %d cols %d group_from %d: %.1f ms = %.2f Mpix/s  %s" % (box, push, cols, grp, best[0], gs.rows * gs.cols / best[0] / 1e3, best[1]), flush=True)
