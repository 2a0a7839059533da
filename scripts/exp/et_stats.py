#!/usr/bin/env python3
"""EXPERIMENT driver for et_stats.c (CPU only): early-termination and dis-sharing potential on a
window of a config-C frame.  usage: et_stats.py [cfg] [x0 y0 x1 y1] [n_launch]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gipuma_amd import abi, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SO = "/tmp/libet_stats.so"
subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-fopenmp", "-ffp-contract=off", "-mavx2", "-mfma",
                       "-fno-math-errno", "-shared", "-o", SO, os.path.join(HERE, "et_stats.c"), "-lm"])
L = C.CDLL(SO)
cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
win = [int(v) for v in sys.argv[2:6]] if len(sys.argv) > 5 else [640, 480, 896, 608]
nl = int(sys.argv[6]) if len(sys.argv) > 6 else 16
scene = sys.argv[7] if len(sys.argv) > 7 else None
kw = {}
if scene:
    kw["scene"] = scene
gs, info = synth.build_problem(cfg, **kw)
nd = L.et_stats_sizeof() // 8
out = np.zeros((nl, nd), np.float64)
L.et_stats_run.argtypes = [C.POINTER(abi.Desc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
L.et_stats_run(C.byref(gs.desc), win[0], win[1], win[2], win[3], nl, out.ctypes.data, 1)

# field offsets (doubles), in declaration order
names = [("prop_tasks", 1), ("ref_tasks", 4), ("cols_full_prop", 1), ("cols_full_ref", 4),
         ("cols_lane_prop", 4), ("cols_lane_ref", 16), ("cols_wave_ref", 16), ("cols_wave_prop", 4),
         ("ambiguous_prop", 4), ("ambiguous_ref", 16), ("wrong", 4),
         ("jobs", 1), ("job_union_cols", 1), ("job_target_cols", 1), ("job_hist", 5),
         ("accepted_prop", 1), ("accepted_ref", 4), ("ratio_hist_ref", 32), ("ratio_hist_prop", 8), ("plane_groups", 1), ("plane_union_samples", 1), ("plane_task_samples", 1), ("plane_bbox_samples", 1), ("plane_maxgroup", 1), ("cols_w8_ref", 16), ("seen4", 1), ("seen8", 1), ("seen32", 1), ("seen_own8", 1), ("seen_own16", 1), ("xrun", 4), ("yrun", 4), ("cols_sorted_ref", 16), ("cols_pool_ref", 48), ("cols_wave_nr", 16), ("cols_two_phase", 32), ("items_alive", 16), ("prop_item", 4), ("lb_dead", 24), ("lb_fixed", 8), ("lb_items", 4), ("lb_cand_dead", 24), ("lb_cands", 4), ("lb_need", 4), ("lb_pair_dead", 24), ("lb_quad_dead", 24)]
off = {}
o = 0
for n, k in names:
    off[n] = (o, k)
    o += k
assert o == nd, (o, nd)


def f(row, n):
    a, k = off[n]
    return row[a:a + k]


npx = (win[2] - win[0]) * (win[3] - win[1]) / 2
print("window %s, %d px per colour" % (win, npx))
pol = ["3B", "th1.0", "th1.5", "th2.0/1.0nk"]  # refinement: the 4th policy is theta 1.0 without the k-th-smallest rule
for li in range(nl):
    r = out[li]
    pt = f(r, "prop_tasks")[0]
    rt = f(r, "ref_tasks")
    cfp = f(r, "cols_full_prop")[0]
    cfr = f(r, "cols_full_ref")
    print("launch %2d: prop %.2f/px (acc %.3f)  refine acc %s" %
          (li, pt / npx, f(r, "accepted_prop")[0] / max(pt, 1), np.round(f(r, "accepted_ref")[:3] / np.maximum(rt[:3], 1), 3)))
    print("    F/B hist prop  %s" % np.round(f(r, "ratio_hist_prop") / max(pt, 1), 3))
    rh = f(r, "ratio_hist_ref").reshape(4, 8)
    for s in range(3):
        print("    F/B hist ref%d  %s" % (s, np.round(rh[s] / max(rt[s], 1), 3)))
    clp = f(r, "cols_lane_prop")
    cwp = f(r, "cols_wave_prop")
    clr = f(r, "cols_lane_ref").reshape(4, 4)
    cwr = f(r, "cols_wave_ref").reshape(4, 4)
    amb_r = f(r, "ambiguous_ref").reshape(4, 4)
    for p in range(4):
        print("    %-13s prop lane %.3f wave %.3f (amb %.4f) | ref lane %s wave %s amb %s wrong %d" %
              (pol[p], clp[p] / max(cfp, 1), cwp[p] / max(cfp, 1), f(r, "ambiguous_prop")[p] / max(pt, 1),
               np.round(clr[p][:3] / np.maximum(cfr[:3], 1), 3), np.round(cwr[p][:3] / np.maximum(cfr[:3], 1), 3),
               np.round(amb_r[p][:3] / np.maximum(rt[:3], 1), 4), f(r, "wrong")[p]))
    c8 = f(r, "cols_w8_ref").reshape(4, 4)
    for p in range(4):
        print("    %-13s 8-task wavefronts: ref %s" % (pol[p], np.round(c8[p][:3] / np.maximum(cfr[:3], 1), 3)))
    cs_ = f(r, "cols_sorted_ref").reshape(4, 4)
    for p in range(4):
        print("    %-13s waves of lanes sorted by predicted stop: ref %s" % (pol[p], np.round(cs_[p][:3] / np.maximum(cfr[:3], 1), 3)))
    wn_ = f(r, "cols_wave_nr").reshape(4, 4)
    for p in range(4):
        print("    %-13s wave-level, redo not counted: ref %s  amb %s lane %s" % (pol[p], np.round(wn_[p][:4] / np.maximum(cfr[:4], 1), 3), np.round(amb_r[p][:4] / np.maximum(rt[:4], 1), 4), np.round(clr[p][:4] / np.maximum(cfr[:4], 1), 3)))
    tp_ = f(r, "cols_two_phase").reshape(4, 4, 2)
    ia_ = f(r, "items_alive").reshape(4, 4)
    for st_ in range(3):
        print("    two-phase step %d: G0=1..4 half-views %s  all-views %s  alive items %s" % (st_, np.round(tp_[st_, :, 0] / max(cfr[st_], 1), 3), np.round(tp_[st_, :, 1] / max(cfr[st_], 1), 3), np.round(ia_[st_] * 8 / max(cfr[st_], 1), 3)))
    cp_ = f(r, "cols_pool_ref").reshape(4, 4, 3)
    for p in range(4):
        print("    %-13s workgroup pool, compaction every 1/2/4 columns: ref %s" % (pol[p], " | ".join(str(np.round(cp_[p][:4, g] / np.maximum(cfr[:4], 1), 3)) for g in range(3))))
    pi_ = f(r, "prop_item")
    print("    propagation bounded item-wise vs the cost at the start (open tasks in full): thr only %.3f (open %.3f), with k-th rule %.3f (open %.3f)" % (pi_[0] / max(cfp, 1), pi_[2] / max(pt, 1), pi_[1] / max(cfp, 1), pi_[3] / max(pt, 1)))
    print("    seen-before fraction of needed prop tasks: K=4 %.3f  K=8 %.3f  K=32 %.3f" %
          (f(r, "seen4")[0] / max(pt, 1), f(r, "seen8")[0] / max(pt, 1), f(r, "seen32")[0] / max(pt, 1)))
    print("    ... with the pixel's own refinement-accepted planes in the ring too: K=8 %.3f  K=16 %.3f" %
          (f(r, "seen_own8")[0] / max(pt, 1), f(r, "seen_own16")[0] / max(pt, 1)))
    lbd = f(r, "lb_dead").reshape(4, 6); lbi = np.maximum(f(r, "lb_items"), 1); lbf = f(r, "lb_fixed").reshape(4, 2)
    lcd = f(r, "lb_cand_dead").reshape(4, 6); lbc = np.maximum(f(r, "lb_cands"), 1)
    for st_ in range(3):
        print("    LB prefilter step %d: items dead after the K=4,8,12,16,24,32 heaviest samples %s | centre 4x4 %.3f 2x2 %.3f | candidates rejected outright %s | mean samples to reach thr %.1f" %
              (st_, np.round(lbd[st_] / lbi[st_], 3), lbf[st_][0] / lbi[st_], lbf[st_][1] / lbi[st_], np.round(lcd[st_] / lbc[st_], 3), f(r, "lb_need")[st_] / lbi[st_]))
    xr, yr = f(r, "xrun"), f(r, "yrun")
    print("    runs of equal planes: along x cols %.3f of the tasks' (%.2f tasks per run, %.2f of the runs single) | along y %.3f (%.2f, %.2f)" %
          (xr[0] / max(xr[1], 1), xr[1] / 8 / max(xr[2], 1), xr[3] / max(xr[2], 1), yr[0] / max(yr[1], 1), yr[1] / 8 / max(yr[2], 1), yr[3] / max(yr[2], 1)))
    lpd = f(r, "lb_pair_dead").reshape(4, 6); lqd = f(r, "lb_quad_dead").reshape(4, 6)
    for st_ in range(3):
        print("    LB prefilter step %d, K=4,8,12,16,24,32: singles %s | pairs %s | quads %s" % (st_, np.round(lbd[st_] / lbi[st_], 3), np.round(lpd[st_] / lbi[st_], 3), np.round(lqd[st_] / lbi[st_], 3)))
    jobs = f(r, "jobs")[0]
    print("    sharing: jobs %.2f/px, targets/job hist %s, union cols / target cols = %.3f" %
          (jobs / npx, np.round(f(r, "job_hist") / max(jobs, 1), 3),
           f(r, "job_union_cols")[0] / max(f(r, "job_target_cols")[0], 1)))
    print("    plane-keyed: groups %.2f/px, union samples / task samples = %.3f, bbox / task samples = %.3f" %
          (f(r, "plane_groups")[0] / npx, f(r, "plane_union_samples")[0] / max(f(r, "plane_task_samples")[0], 1),
           f(r, "plane_bbox_samples")[0] / max(f(r, "plane_task_samples")[0], 1)))
