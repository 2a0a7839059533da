#!/usr/bin/env python3
"""Summarise rocprofv3 output databases (rocpd sqlite) into the small text/JSON files kept under
profiles/.

    python scripts/rocprof_summary.py stats  <results.db>            # per-kernel time table
    python scripts/rocprof_summary.py pmc    <dir-with-pmc*_results.db> [kernel-substring]
    python scripts/rocprof_summary.py pmcunits <dir> <n_units> <kernel-substring> [...]   # sums / n_units

`pmc` prints, for kernels whose name contains the substring (default "sweep_kernel"), the mean
per-dispatch value of every collected counter plus the derived figures used in DESIGN.md.
"""
import glob
import json
import os
import sqlite3
import sys


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    out = []
    for name, calls, total, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        if len(short) > 70:
            short = short[:67] + "..."
        out.append((short, calls, total / 1e3, avg / 1e3, pct))
    print("%-72s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "%"))
    for r in out[:15]:
        print("%-72s %8d %12.1f %12.1f %7.2f" % r)


def pmc(d, sub="sweep_kernel"):
    vals = {}
    dur = []
    for db in sorted(glob.glob(os.path.join(d, "*_results.db"))):
        cur = sqlite3.connect(db).cursor()
        try:
            q = cur.execute("select counter_name, avg(value), count(*) from counters_collection "
                            "where kernel_name like ? group by counter_name", ("%" + sub + "%",))
            for name, v, n in q:
                vals[name] = (v, n)
            for (a,) in cur.execute("select avg(duration) from kernels where name like ?", ("%" + sub + "%",)):
                if a:
                    dur.append(a / 1e6)
        except sqlite3.Error as e:
            print("skip", db, e)
    res = {k: v[0] for k, v in vals.items()}
    res["_dispatches_per_counter"] = max([v[1] for v in vals.values()] or [0])
    res["_kernel_ms_profiled_mean"] = sum(dur) / len(dur) if dur else None
    g = res.get
    if g("FETCH_SIZE") is not None:
        # MI355X_MICROARCH.md HBM section: FETCH_SIZE (KiB) counts 128-B requests at 64 B on gfx950
        # for wide coalesced streams -> doubled; WRITE_SIZE taken as reported (uncalibrated).
        res["hbm_read_bytes_per_launch_x2corr"] = 2.0 * g("FETCH_SIZE") * 1024.0
    if g("WRITE_SIZE") is not None:
        res["hbm_write_bytes_per_launch"] = g("WRITE_SIZE") * 1024.0
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        res["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        res["l1_hit_rate_est"] = 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU") is not None:
        res["valu_active_frac_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
        res["wait_any_frac_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    print(json.dumps(res, indent=1, sort_keys=True))


def pmc_units(d, n_units, subs):
    """per-UNIT counter values: the sum over every dispatch whose kernel name contains one of `subs`,
    divided by n_units -- e.g. the 16 half-sweeps of a view, each of which is one sweep kernel plus,
    where the propagation costs are pushed, one pm::push_kernel (17 + 4 dispatches)"""
    vals, dur, ndisp = {}, [], 0
    where = " or ".join(["kernel_name like ?"] * len(subs))
    where_k = " or ".join(["name like ?"] * len(subs))
    args = ["%" + x + "%" for x in subs]
    for db in sorted(glob.glob(os.path.join(d, "*_results.db"))):
        cur = sqlite3.connect(db).cursor()
        try:
            for name, v, n in cur.execute("select counter_name, sum(value), count(*) from counters_collection "
                                          "where %s group by counter_name" % where, args):
                vals[name] = v / n_units
                ndisp = max(ndisp, n)
            for (a,) in cur.execute("select sum(duration) from kernels where %s" % where_k, args):
                if a:
                    dur.append(a / 1e6 / n_units)
        except sqlite3.Error as e:
            print("skip", db, e)
    res = dict(vals)
    res["_units"] = n_units
    res["_dispatches_per_counter"] = ndisp
    res["_kernels"] = subs
    res["_kernel_ms_profiled_mean"] = sum(dur) / len(dur) if dur else None
    g = res.get
    if g("FETCH_SIZE") is not None:
        res["hbm_read_bytes_per_launch_x2corr"] = 2.0 * g("FETCH_SIZE") * 1024.0
    if g("WRITE_SIZE") is not None:
        res["hbm_write_bytes_per_launch"] = g("WRITE_SIZE") * 1024.0
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        res["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        res["l1_hit_rate_est"] = 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU") is not None:
        res["valu_active_frac_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
        res["wait_any_frac_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    print(json.dumps(res, indent=1, sort_keys=True))


def series(db, sub="sweep_kernel"):
    """durations of consecutive dispatches of one kernel, in launch order"""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select start, duration from kernels where name like ? order by start", ("%" + sub + "%",)))
    print("# %d dispatches of *%s*: index, duration_ms" % (len(rows), sub))
    for i, (_, d) in enumerate(rows):
        print("%3d %8.3f" % (i, d / 1e6))


def pmcseries(db, sub="sweep_kernel"):
    """per-dispatch counter values (launch order) next to the dispatch duration"""
    cur = sqlite3.connect(db).cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    key = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else None)
    if key is None:
        sys.exit("counters_collection columns: %s" % cols)
    rows = list(cur.execute("select %s, counter_name, sum(value) from counters_collection where kernel_name like ? "
                            "group by %s, counter_name order by %s" % (key, key, key), ("%" + sub + "%",)))
    durs = [d for (d,) in cur.execute("select duration from kernels where name like ? order by start", ("%" + sub + "%",))]
    names = sorted({r[1] for r in rows})
    table = {}
    for k, n, v in rows:
        table.setdefault(k, {})[n] = v
    print("# idx duration_ms " + " ".join(names))
    for i, k in enumerate(sorted(table)):
        d = durs[i] / 1e6 if i < len(durs) else float("nan")
        print("%3d %8.3f " % (i, d) + " ".join("%.4g" % table[k].get(n, float("nan")) for n in names))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "pmcseries":
        pmcseries(sys.argv[2], *(sys.argv[3:4]))
    elif sys.argv[1] == "pmcunits":  # pmcunits <dir> <n_units> <substring> [...]
        pmc_units(sys.argv[2], int(sys.argv[3]), sys.argv[4:])
    elif sys.argv[1] == "series":
        series(sys.argv[2], *(sys.argv[3:4]))
    else:
        pmc(sys.argv[2], *(sys.argv[3:4]))
