#!/usr/bin/env python3
"""Experiment (needs a GPU): how many propagation candidates of config C could be skipped EXACTLY?

Rule A: a candidate bitwise equal to the lane's current plane returns the stored cost.
Rule B: a neighbour whose plane did not change during its own last sweep was already evaluated
        (and rejected) by this pixel against a cost that has only decreased since.
Prints, per half-sweep, the mean number of candidates that still need an evaluation and the
number of 256-lane rounds a workgroup would need if (pixel, candidate) tasks were compacted
across lanes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipuma_amd import abi, synth  # noqa: E402
from gipuma_amd.problem import Session  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
gs, info = synth.build_problem(cfg, device="cuda:0", keep_on_device=True)
R, Cc = gs.rows, gs.cols
ys, xs = np.mgrid[0:R, 0:Cc]
changed = np.ones((R, Cc), bool)
tot_need = tot_full = 0.0
with Session(gs) as s:
    s.init_planes()
    n4, _ = s.get_state()
    for it in range(gs.params.iterations):
        for colour in (0, 1):
            act = ((xs + ys) & 1) == colour
            b = n4.view(np.uint32)
            need = np.zeros((R, Cc), np.int32)
            needB = np.zeros((R, Cc), np.int32)
            # stateless variant: distinct planes among the 8 candidates that differ from the own plane
            cands = []
            for dist in (1, 5):
                for dy, dx in ((-dist, 0), (dist, 0), (0, -dist), (0, dist)):
                    ok = (ys + dy >= 0) & (ys + dy < R) & (xs + dx >= 0) & (xs + dx < Cc)
                    cands.append((np.roll(b, (-dy, -dx), axis=(0, 1)), ok))
            needAC = np.zeros((R, Cc), np.int32)
            needACB = np.zeros((R, Cc), np.int32)  # stateless rules AND the history rule B
            k = 0
            for dist in (1, 5):
                for dy, dx in ((-dist, 0), (dist, 0), (0, -dist), (0, dist)):
                    ck, okk = cands[k]
                    fresh = okk & ~(ck == b).all(-1)
                    for j in range(k):
                        fresh &= ~(cands[j][1] & (cands[j][0] == ck).all(-1))
                    needAC += fresh
                    needACB += fresh & np.roll(changed, (-dy, -dx), axis=(0, 1))
                    k += 1
            for dist in (1, 5):
                for dy, dx in ((-dist, 0), (dist, 0), (0, -dist), (0, dist)):
                    ok = (ys + dy >= 0) & (ys + dy < R) & (xs + dx >= 0) & (xs + dx < Cc)
                    same = (np.roll(b, (-dy, -dx), axis=(0, 1)) == b).all(-1)
                    ch = np.roll(changed, (-dy, -dx), axis=(0, 1))
                    need += (ok & ~same & ch)
                    needB += (ok & ch)
            Rt, Ct = R // 16 * 16, Cc // 32 * 32
            blk = np.where(act, need, 0)[:Rt, :Ct].reshape(Rt // 16, 16, Ct // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 512).sum(1)
            rounds = np.ceil(blk / 256.0).mean()
            before = n4.copy()
            s.sweep(it, colour)
            n4, _ = s.get_state()
            chg = (before.view(np.uint32) != n4.view(np.uint32)).any(-1)
            changed[act] = chg[act]
            blkAC = np.where(act, needAC, 0)[:Rt, :Ct].reshape(Rt // 16, 16, Ct // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 512).sum(1)
            wave = np.where(act, needAC, 0)[:Rt // 4 * 4, :Ct].reshape(Rt // 4, 4, Ct // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 128).sum(1)
            tot_ac = globals().get("tot_ac", 0.0) + np.ceil(wave / 64.0).mean() + 3
            globals()["tot_ac"] = tot_ac
            print("   stateless A+dedupe: %.2f / 8 per pixel; with history rule B as well: %.2f; wave-compacted rounds %.2f"
                  % (needAC[act].mean(), needACB[act].mean(), np.ceil(wave / 64.0).mean()))
            tot_need += rounds + 3
            tot_full += 11
            print("it %d colour %d: need eval (A+B) %.2f / 8 per pixel (B only %.2f); compacted rounds %.2f / 8; changed %.3f"
                  % (it, colour, need[act].mean(), needB[act].mean(), rounds, chg[act].mean()), flush=True)
print("hypothesis evaluations with compaction: %.1f vs %.1f  -> x%.2f" % (tot_need, tot_full, tot_full / tot_need))
print("stateless (A + dedupe), wave-level compaction: %.1f vs %.1f -> x%.2f" % (tot_ac, tot_full, tot_full / tot_ac))
