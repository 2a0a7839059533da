"""GIPUMA_HIP_FLAG_FAST (include/gipuma_hip.h): the tolerance-judged flavour of the kernels -- since round 6 the numerical model
of rounds 1-5 (x * (1/z) for x / z, fused multiply-adds in the sample loop: pm_sample.h PM_MODEL 0) plus the shortcuts of
--use_fast_math's kind (pm_core.h PM_APPROX).  It is judged by the fraction of pixels inside the north_star tolerance (depth
1e-4 relative, unit normals 1e-3) on free-running solves: against the default mode here (committed floors = the measured
agreement less a margin: it is as far from the default mode as the default mode of round 5 was from the reference's own code,
which is where its rounding choices come from), and against the reference's own code at real frame sizes in
tests/test_headline_parity.py.  The default mode itself is bit-identical to the oracle (tests/test_parity_gpu.py), so
"against the default mode" is "against the oracle"."""
import numpy as np
import pytest

from gipuma_amd import abi, synth
from gipuma_amd.problem import Session, runcuda
from tests import ref_lib
from tests.oracle_lib import OracleState

pytestmark = pytest.mark.gpu


def in_tolerance(a, b):
    d_rel = np.abs(a[..., 3] - b[..., 3]) / np.maximum(np.abs(b[..., 3]), 1e-30)
    n_err = np.abs(a[..., :3] - b[..., :3]).max(-1)
    return float(((d_rel < 1e-4) & (n_err < 1e-3)).mean())


# (configuration, frame, floor of fast vs default, measured on an MI355X in round 6)
CASES = [
    ("A", dict(cols=320, rows=256), 0.99, 0.9952),
    ("B", dict(cols=320, rows=256), 0.975, 0.9828),
    ("B", {}, 0.93, 0.9421),                          # 640x480 in full
    ("C", dict(cols=320, rows=256), 0.999, 0.9996),
    ("C", {}, 0.998, 0.99880),                        # the headline frame, 1600x1200
]


@pytest.mark.parametrize("cfg,over,floor,measured", CASES)
def test_fast_mode_agrees_with_the_default_mode(hip, cfg, over, floor, measured):
    assert floor <= measured
    gs, info = synth.build_problem(cfg, **over)
    exact, _ = runcuda(gs)
    fast, cf = runcuda(gs, fast=True)
    got = in_tolerance(fast, exact)
    print("fast vs default, config %s %dx%d: %.4f of the pixels inside the tolerance" % (cfg, gs.cols, gs.rows, got))
    assert got >= floor
    gt = info["gt_depth"]
    q_e, q_f = (np.abs(exact[..., 3] - gt) / gt < 0.01).mean(), (np.abs(fast[..., 3] - gt) / gt < 0.01).mean()
    assert abs(q_e - q_f) < 2e-3  # both reconstruct the same share of the surface
    # size-independent properties of the finalized maps
    valid = cf != abs(abi.MAXCOST)
    assert np.allclose(np.linalg.norm(fast[..., :3], axis=-1)[valid], 1.0, atol=1e-5)
    assert (cf >= 0).all() and (cf <= abi.MAXCOST).all()


def test_fast_mode_against_the_reference_s_own_code(hip):
    """config C's parameters on 320x256 through the reference's own device code on the CPU (oracle/_ref, travels with the
    snapshot): the default mode (measured 0.99995) and the fast mode (0.9996) against it"""
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    gs, _ = synth.build_problem("C", cols=320, rows=256)
    rn, _ = ref_lib.RefState(gs, tex_mode=0).run()
    exact, _ = runcuda(gs)
    fast, _ = runcuda(gs, fast=True)
    assert in_tolerance(exact, rn) >= 0.9995
    assert in_tolerance(fast, rn) >= 0.999
    assert in_tolerance(exact, rn) >= in_tolerance(fast, rn)


def test_fast_session_serves_every_entry_point(hip, tiny_problem):
    """a fast session through the launch-by-launch API; its costs are those of its planes within rounding, and its first
    launch (random planes + their costs) equals the default mode's planes bit for bit (same random numbers) and costs to
    1e-4 relative"""
    gs, _ = tiny_problem
    with Session(gs, fast=True) as s, Session(gs) as e:
        for x in (s, e):
            x.init_planes()
        n_f, c_f = s.get_state()
        n_e, c_e = e.get_state()
        assert np.array_equal(n_f.view(np.uint32), n_e.view(np.uint32))
        assert np.allclose(c_f, c_e, rtol=1e-4, atol=1e-4)
        assert np.allclose(s.eval_cost(n_f), c_f, rtol=1e-5, atol=1e-4)
        assert s.schedule() == e.schedule()
        for it in range(2):
            for colour in (abi.BLACK, abi.RED):
                s.sweep(it, colour)
        n1, c1 = s.get_state()
        assert np.allclose(s.eval_cost(n1), c1, rtol=1e-5, atol=1e-4)  # stored cost == cost of the stored plane
        s.set_state(n_e, c_e)
        n2, _ = s.get_state()
        assert np.array_equal(n2.view(np.uint32), n_e.view(np.uint32))
        s.finalize()
        t = s.solve(timing=True)
        ms, _ = s.launch_times()
        assert t.ms_total > 0 and len(ms) == 2 * gs.params.iterations
    # and the oracle is within the tolerance of it on this tiny frame too
    on, _ = OracleState(gs).run()
    fast, _ = runcuda(gs, fast=True)
    assert in_tolerance(fast, on) >= 0.97


def test_fast_flag_does_not_leak_into_exact_sessions(hip, tiny_problem):
    """an exact session created after (and while) a fast one lives is still bit-identical to the oracle"""
    gs, _ = tiny_problem
    on, oc = OracleState(gs).run()
    with Session(gs, fast=True) as s:
        s.solve(timing=False)
        n4, c = runcuda(gs)
    assert np.array_equal(n4.view(np.uint32), on.view(np.uint32))
    assert np.array_equal(c.view(np.uint32), oc.view(np.uint32))
    assert gs.desc.flags & abi.FLAG_FAST == 0


@pytest.mark.parametrize("what", ["colour", "float_images", "runtime_window", "box25", "generic_combiner"])
def test_fast_mode_serves_every_kernel_family(hip, what):
    """the tolerance-judged flavour of the kernels the headline configurations do not reach -- colour (T = float4),
    images that are not 8-bit (float planes, exp evaluated per sample), a runtime-sized window (box 9), box 25, the
    generic view combiner -- on small frames: they run, and agree with the exact mode like the rest (most pixels bit for bit)"""
    cfg = dict(cols=160, rows=112, n_src=4, blocksize=15, iterations=3, n_best=3)
    kw = {}
    if what == "colour":
        kw["colour"] = True
    elif what == "runtime_window":
        cfg["blocksize"] = 9
    elif what == "box25":
        cfg.update(blocksize=25, n_src=3, iterations=2)
    elif what == "generic_combiner":
        kw["cost_comb"] = abi.COMB_GOOD
    gs, _ = synth.build_problem(synth.tiny_config(**cfg), **kw)
    if what == "float_images":
        from gipuma_amd.problem import GlobalState
        imgs = [im + np.float32(0.25) for im in gs.images]
        gs = GlobalState(imgs, gs.cameras, gs.selected, gs.params, seed=gs.desc.seed)
    exact, ce = runcuda(gs)
    fast, cf = runcuda(gs, fast=True)
    assert np.isfinite(fast).all() and np.isfinite(cf).all()
    assert in_tolerance(fast, exact) >= 0.97, what
