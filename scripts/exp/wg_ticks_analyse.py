#!/usr/bin/env python3
"""per-workgroup clocks of the fused launches (a -DPM_WG_TICKS build run with GIPUMA_HIP_WG_TICKS=<file>): how long a
workgroup lives, how far its wavefronts' ends are apart, how well the slots of the GPU are filled over a launch, and what
list scheduling in other orders would give (longest first; the heaviest tenth first, the rest in place)"""
import heapq
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
pos, launches = 0, []
while pos < raw.size:
    gx, gy, phase, tune = (int(v) for v in raw[pos:pos + 4])
    n = gx * gy
    launches.append((gx, gy, phase, raw[pos + 4:pos + 4 + 4 * n].reshape(n, 4).astype(np.int64)))
    pos += 4 + 4 * n


def makespan(d, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for x in d:
        heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)


prev = {}
for gx, gy, phase, t in launches[-12:]:  # the last solve's fused launches
    start, end, first_end = t[:, 0], t[:, 1], t[:, 2]
    dur = (end - start) * 10e-3  # us
    span = (end.max() - start.min()) * 10e-3
    spread = (end - first_end) * 10e-3
    n = len(dur)
    busy = dur.sum() / (768 * span)
    # per XCD (workgroup b runs on XCD b % 8): when its last workgroup ends
    xcd_end = [(end[k::8].max() - start.min()) * 10e-3 for k in range(8)]
    line = ("phase %2d: %d workgroups, launch %.0f us; workgroup %.0f us mean, cv %.2f, max %.0f; wavefront ends %.0f us apart (mean); "
            "slots busy %.3f; XCDs end at %s" % (phase, n, span, dur.mean(), dur.std() / dur.mean(), dur.max(), spread.mean(), busy,
                                               " ".join("%.0f" % e for e in xcd_end)))
    # list scheduling of the measured durations, 96 slots per XCD, in the order run / longest first / heaviest tenth first
    sims = []
    for order in ("as run", "longest first", "heaviest tenth first"):
        ms = 0.0
        for k in range(8):
            d = dur[k::8]
            if order == "longest first":
                d = np.sort(d)[::-1]
            elif order == "heaviest tenth first":
                idx = np.argsort(d)[::-1]
                top = set(idx[:len(d) // 10].tolist())
                d = np.concatenate([d[sorted(top)], d[[i for i in range(len(d)) if i not in top]]])
            ms = max(ms, makespan(d, 96))
        sims.append("%s %.0f" % (order, ms))
    line += "; list scheduling: " + ", ".join(sims)
    key = phase & 1
    if key in prev and len(prev[key]) == n:
        line += "; correlation with the same colour's previous launch %.2f" % np.corrcoef(prev[key], dur)[0, 1]
    prev[key] = dur
    print(line)
    if phase in (8, 15):  # where the long workgroups are: mean duration per tile row / per tile column (default tile_of mapping)
        ty, tx = np.zeros(n, int), np.zeros(n, int)
        q, r = n >> 3, n & 7
        bh = (gy + 7) >> 3
        for b in range(n):
            xcd, local = b & 7, b >> 3
            t = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + local
            band = t // (bh * gx)
            h = min(bh, gy - band * bh)
            rem = t - band * bh * gx
            tx[b], ty[b] = rem // h, band * bh + rem % h
        rows = [dur[ty == y].mean() for y in range(gy)]
        cols = [dur[tx == x].mean() for x in range(gx)]
        print("   per tile row   :", " ".join("%.0f" % v for v in rows))
        print("   per tile column:", " ".join("%.0f" % v for v in cols))
        order_in_xcd = np.arange(n) >> 3
        for k in (0, 3, 7):
            d = dur[k::8]
            print("   XCD %d, duration by dispatch order (means of 10 chunks): %s" % (k, " ".join("%.0f" % c.mean() for c in np.array_split(d, 10))))
