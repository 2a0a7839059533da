#!/usr/bin/env python3
"""time per view of one workload under a list of schedules (pushed half-sweeps, column-per-lane half-sweeps, first plane-keyed
half-sweep):  python scripts/gpu_r06_sched.py <C|D|colour|box19|...> "push,cols,group_from" ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GIPUMA_HIP_EXPERIMENTS"] = "1"
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

wl = sys.argv[1]
kw, cfg = {}, wl
if wl.startswith("box"):
    cfg, kw = "C", dict(blocksize=int(wl[3:]))
elif wl == "colour":
    cfg, kw = "C", dict(colour=True)
gs, _ = synth.build_problem(cfg, **kw)
for spec in sys.argv[2:] or ["-"]:
    if spec != "-":
        push, cols, grp = spec.split(",")
        os.environ.update(GIPUMA_HIP_PUSH_LAUNCHES=push, GIPUMA_HIP_COLS_LAUNCHES=cols, GIPUMA_HIP_GROUP_FROM=grp)
    with Session(gs) as s:
        s.solve(timing=True)
        best = min((s.solve(timing=True).ms_total, [round(x, 2) for x in s.launch_times()[0]]) for _ in range(2))
    print("%-8s %-22s schedule %-8s: %.1f ms = %.2f Mpix/s  %s" % (wl, os.path.basename(os.environ.get("GIPUMA_HIP_LIB", "default")), spec, best[0],
                                                                gs.rows * gs.cols / best[0] / 1e3, best[1]), flush=True)
