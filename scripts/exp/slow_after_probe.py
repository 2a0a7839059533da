#!/usr/bin/env python3
"""why are config D / box 19 8x slower inside bench.py's other-configurations leg?  (device-resident frames, no experiments env)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

dev = "cuda:0"
for label, cfg, kw in (("D device-resident", "D", dict(device=dev, keep_on_device=True)), ("D host", "D", {}),
                       ("box19 device-resident", "C", dict(blocksize=19, device=dev, keep_on_device=True)),
                       ("C device-resident", "C", dict(device=dev, keep_on_device=True))):
    for rv in (14, 15):
        g, _ = synth.build_problem(cfg, ref_view=rv, **kw)
        g.desc.device_id = 0
        with Session(g) as s:
            s.solve(timing=True)
            t = min(s.solve(timing=True).ms_total for _ in range(2))
            print("%-24s ref view %d: %.1f ms  schedule %s  experiments env %r" % (label, rv, t, s.schedule(), os.environ.get("GIPUMA_HIP_EXPERIMENTS")), flush=True)
        del g
