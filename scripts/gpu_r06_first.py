#!/usr/bin/env python3
"""Round 6, first GPU pass over the three numerical models running through ALL kernel families (pm_sample.h):
   1. parity on small problems: default mode == oracle flavour 6; GIPUMA_HIP_FLAG_LITERAL == the reference's own code
      (oracle/_ref) and == oracle flavour 7; colour likewise; the model-0 build variant == oracle flavour 0
   2. time per config-C view of the modes (and of the model-0 variant, the round-5 default)
   3. the proof by exhaustion of the Markstein quotient (gipuma_hip_selftest_quotient), whole range
usage: python scripts/gpu_r06_first.py [parity] [time] [quotient]      (GIPUMA_HIP_LIB selects a variant library)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GIPUMA_HIP_EXPERIMENTS", "1")
from gipuma_amd import abi, synth  # noqa: E402
from gipuma_amd.problem import Session, runcuda  # noqa: E402

what = sys.argv[1:] or ["parity", "time", "quotient"]
lib = abi.load_library()
print("library:", os.environ.get("GIPUMA_HIP_LIB", "default"), flush=True)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def tol(a, b):
    d_rel = np.abs(a[..., 3] - b[..., 3]) / np.maximum(np.abs(b[..., 3]), 1e-30)
    n_err = np.abs(a[..., :3] - b[..., :3]).max(-1)
    return float(((d_rel < 1e-4) & (n_err < 1e-3)).mean())


def report(name, n4, c, rn, rc):
    same_p = float((bits(n4) == bits(rn)).all(-1).mean())
    same_c = float((bits(c) == bits(rc)).mean())
    print("%-64s planes identical %.6f  costs identical %.6f  in tolerance %.6f  %s"
          % (name, same_p, same_c, tol(n4, rn), "PASS" if same_p == 1.0 and same_c == 1.0 else "DIFFERS"), flush=True)


if "parity" in what:
    from tests import oracle_lib, ref_lib
    from tests.oracle_lib import OracleState
    L = oracle_lib.lib()
    flav = int(os.environ.get("ORACLE_FLAVOUR", "6"))
    cases = [("tiny box 7", synth.tiny_config(), {}),
             ("ragged 150x100 box 11", synth.tiny_config(cols=150, rows=100, n_src=4, blocksize=11, iterations=2, n_best=3), {}),
             ("box 15 160x112", synth.tiny_config(cols=160, rows=112, n_src=4, blocksize=15, iterations=3, n_best=3), {}),
             ("box 25 160x112", synth.tiny_config(cols=160, rows=112, n_src=3, blocksize=25, iterations=2, n_best=3), {}),
             ("box 19 160x112", synth.tiny_config(cols=160, rows=112, n_src=3, blocksize=19, iterations=2, n_best=3), {}),
             ("colour box 15 128x96", synth.tiny_config(cols=128, rows=96, n_src=4, blocksize=15, iterations=3, n_best=3), dict(colour=True)),
             ("colour box 9 96x64", synth.tiny_config(cols=96, rows=64, n_src=3, blocksize=9, iterations=2, n_best=2), dict(colour=True)),
             ("B 320x256", "B", dict(cols=320, rows=256)),
             ("C 320x256", "C", dict(cols=320, rows=256)),
             ("C 832x640 (push + cols + fused plane-keyed kernels)", "C", dict(cols=832, rows=640, iterations=4)),
             ("D 832x640 2 it", "D", dict(cols=832, rows=640, iterations=2)),
             ("colour C 832x640 3 it", "C", dict(cols=832, rows=640, iterations=3, colour=True))]
    for name, cfg, kw in cases:
        gs, _ = synth.build_problem(cfg, **kw)
        t0 = time.time()
        L.gipuma_oracle_set_flavour(flav)
        on, oc = OracleState(gs).run()
        n4, c = runcuda(gs)
        report("default == oracle flavour %d: %s" % (flav, name), n4, c, on.copy(), oc.copy())
        if flav == 6:
            L.gipuma_oracle_set_flavour(7)
            on, oc = OracleState(gs).run()
            n4, c = runcuda(gs, literal=True)
            report("literal == oracle flavour 7: %s" % name, n4, c, on.copy(), oc.copy())
            if ref_lib.available() and gs.rows % 32 == 0 and gs.cols % 32 == 0 and gs.rows * gs.cols <= 400 * 300:
                rn, rc = ref_lib.RefState(gs, tex_mode=0).run()
                report("literal == the reference's own code: %s" % name, n4, c, rn, rc)
        L.gipuma_oracle_set_flavour(-1)
        print("   (%.0f s)" % (time.time() - t0), flush=True)

if "time" in what:
    gs, _ = synth.build_problem("C", keep_on_device=True, device="cuda") if False else synth.build_problem("C")
    for mode, kw in (("default", {}), ("fast", dict(fast=True)), ("literal", dict(literal=True))):
        with Session(gs, **kw) as s:
            s.solve(timing=True)
            best = None
            for _ in range(3):
                t = s.solve(timing=True)
                ms, npush = s.launch_times()
                if best is None or t.ms_total < best[0]:
                    best = (t.ms_total, t.ms_init, list(ms))
            print("config C %-8s %.2f ms per view = %.2f Mpix/s  (init %.2f; half-sweeps %s)"
                  % (mode, best[0], gs.rows * gs.cols / best[0] / 1e3, best[1], " ".join("%.2f" % m for m in best[2])), flush=True)

if "quotient" in what:
    n = C.c_ulonglong(0)
    t0 = time.time()
    total = 0
    step = 1 << 18
    for z0 in range(0, 1 << 23, step):
        abi.check(lib, lib.gipuma_hip_selftest_quotient(0, z0, step, C.byref(n)), "selftest_quotient")
        total += n.value
        if z0 % (1 << 21) == 0:
            print("  denominators %d .. : mismatches so far %d (%.0f s)" % (z0, total, time.time() - t0), flush=True)
    print("gipuma_hip_selftest_quotient: all 2^23 x 2^23 significand pairs, q' = RN(q + RN(x - q z) r) vs x / z: %d mismatches, %.0f s"
          % (total, time.time() - t0), flush=True)
