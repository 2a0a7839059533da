#!/usr/bin/env python3
"""GIPUMA_HIP_FLAG_FAST against the exact mode (and, where oracle/_ref is present, against the reference's own code on
the CPU): fraction of pixels inside the north_star tolerance (depth 1e-4 relative, unit normals 1e-3), fraction with
bit-identical planes, device time of both modes.  Needs a GPU.
    python scripts/fast_mode_report.py [--ref] CFG[:COLSxROWS] ...      e.g.  C:320x256 A:320x256 B C"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipuma_amd import synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402


def frac(a, b):
    d_rel = np.abs(a[..., 3] - b[..., 3]) / np.maximum(np.abs(b[..., 3]), 1e-30)
    n_err = np.abs(a[..., :3] - b[..., :3]).max(-1)
    ok = (d_rel < 1e-4) & (n_err < 1e-3)
    same = (a.view(np.uint32) == b.view(np.uint32)).all(-1)
    return float(ok.mean()), float(same.mean())


def solve(gs, fast):
    with Session(gs, fast=fast) as s:
        s.solve(timing=True)
        t = s.solve(timing=True)
        n4, c = s.get_state()
    return n4, c, t.ms_total


def main():
    args = sys.argv[1:]
    with_ref = "--ref" in args
    args = [a for a in args if a != "--ref"]
    print("%-14s %10s %10s | fast vs exact: %9s %9s | %s" % ("config", "exact ms", "fast ms", "in tol", "planes ==",
                                                            "vs reference's own code (in tol): exact, fast" if with_ref else ""))
    for a in args:
        cfg, _, size = a.partition(":")
        over = {}
        if size:
            c, r = size.split("x")
            over = dict(cols=int(c), rows=int(r))
        gs, info = synth.build_problem(cfg, **over)
        ne, ce, te = solve(gs, False)
        nf, cf, tf = solve(gs, True)
        ok, same = frac(nf, ne)
        gt = info["gt_depth"]
        on_surface = np.abs(ne[..., 3] - gt) / gt < 0.01  # where the exact mode reconstructs the surface
        ok_surf = frac(nf[on_surface][None], ne[on_surface][None])[0]
        sat = (cf == ce) & ~(nf.view(np.uint32) == ne.view(np.uint32)).all(-1)  # different plane, bitwise equal cost
        extra = ""
        if with_ref:
            from tests import ref_lib
            if ref_lib.available() and gs.rows % 32 == 0 and gs.cols % 32 == 0:
                t0 = time.time()
                rn, rc = ref_lib.RefState(gs, tex_mode=0).run()
                extra = "%.4f %.4f (reference %.0f s)" % (frac(ne, rn)[0], frac(nf, rn)[0], time.time() - t0)
            else:
                extra = "n/a"
        print("%-14s %10.2f %10.2f | %23.4f %9.4f | %s   [within 1%% of GT: exact %.4f fast %.4f; in tol where the exact "
              "mode is on the surface: %.4f; different plane with equal cost: %.4f]"
              % (a, te, tf, ok, same, extra, (np.abs(ne[..., 3] - gt) / gt < 0.01).mean(),
                 (np.abs(nf[..., 3] - gt) / gt < 0.01).mean(), ok_surf, sat.mean()), flush=True)


if __name__ == "__main__":
    main()
