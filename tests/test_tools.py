"""The profiling helper that turns rocprofv3 databases into the figures under profiles/: per-UNIT sums
(scripts/rocprof_summary.py pmcunits) on a synthetic database -- a half-sweep of a view is one sweep
kernel plus, where the propagation costs are pushed, a pm::push_kernel dispatch."""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_units_sums_all_matching_dispatches(tmp_path):
    db = tmp_path / "pmc1_results.db"
    con = sqlite3.connect(str(db))
    con.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
    con.execute("create table kernels (name text, start integer, duration integer)")
    rows = [("void pm::sweep_kernel<15>(...)", 100.0, 5_000_000)] * 3 + \
           [("void pm::push_kernel<15>(...)", 40.0, 4_000_000)] * 2 + [("void pm::init_kernel(...)", 7.0, 1_000_000)]
    for i, (name, v, d) in enumerate(rows):
        con.execute("insert into counters_collection values (?, 'SQ_INSTS_VALU', ?)", (name, v))
        con.execute("insert into counters_collection values (?, 'TCC_HIT_sum', ?)", (name, 9.0))
        con.execute("insert into counters_collection values (?, 'TCC_MISS_sum', ?)", (name, 1.0))
        con.execute("insert into kernels values (?, ?, ?)", (name, i, d))
    con.commit()
    con.close()
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "rocprof_summary.py"), "pmcunits",
                                   str(tmp_path), "3", "sweep_", "push_kernel"])
    res = json.loads(out)
    assert res["_units"] == 3 and res["_dispatches_per_counter"] == 5
    assert abs(res["SQ_INSTS_VALU"] - (3 * 100.0 + 2 * 40.0) / 3) < 1e-9
    assert abs(res["_kernel_ms_profiled_mean"] - (3 * 5.0 + 2 * 4.0) / 3) < 1e-9
    assert abs(res["l2_hit_rate"] - 0.9) < 1e-12
    # the per-kernel mean of the older mode is untouched by the other kernels
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "rocprof_summary.py"), "pmc",
                                   str(tmp_path), "push_kernel"])
    res = json.loads(out)
    assert res["SQ_INSTS_VALU"] == 40.0 and res["_dispatches_per_counter"] == 2


def test_gray_only_scenes_refuse_colour():
    """the stepped and patchy scenes are rendered in gray; asking for them in colour must fail on the host instead of
    describing (rows, cols) planes as float4 texels to the device (bench.py --colour found this the hard way)"""
    import pytest
    from gipuma_amd import synth
    for scene in ("steps", "patchy"):
        with pytest.raises(ValueError):
            synth.build_problem(synth.tiny_config(), colour=True, scene=scene)
