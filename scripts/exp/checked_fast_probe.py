#!/usr/bin/env python3
"""the FAST flavour of a library (GIPUMA_HIP_LIB) on config C: agreement with its own default mode under a few schedules, twice
(is a disagreement deterministic?)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GIPUMA_HIP_EXPERIMENTS", "1")
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import runcuda  # noqa: E402

gs, _ = synth.build_problem("C")
ref = runcuda(gs)


def frac(a, b):
    d_rel = np.abs(a[0][..., 3] - b[0][..., 3]) / np.maximum(np.abs(b[0][..., 3]), 1e-30)
    n_err = np.abs(a[0][..., :3] - b[0][..., :3]).max(-1)
    return float(((d_rel < 1e-4) & (n_err < 1e-3)).mean())


for env in ({}, {"GIPUMA_HIP_GROUP_FUSED": "0"}, {"GIPUMA_HIP_TUNE": str(1 << 23)}, {"GIPUMA_HIP_LB_K": "-1"}, {"GIPUMA_HIP_LB_K": "8"},
            {"GIPUMA_HIP_TUNE": str(1 << 25)}, {"GIPUMA_HIP_TUNE": str(1 << 19)}, {"GIPUMA_HIP_GROUP_FROM": "6"},
            {"GIPUMA_HIP_ET_FORCE": "2"}):
    os.environ.update(env)
    outs = [runcuda(gs, fast=True) for _ in range(2)]
    for k in env:
        del os.environ[k]
    same = bool((outs[0][0].view(np.uint32) == outs[1][0].view(np.uint32)).all())
    print("%-60s fast vs default: %.6f %.6f   two runs identical: %s" % (env, frac(outs[0], ref), frac(outs[1], ref), same), flush=True)
