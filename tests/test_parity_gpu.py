"""Parity of the HIP path (through the C-ABI) with the CPU oracle on the same seeded inputs.

The bar is bit-exact: the numerical model (oracle/gipuma_oracle.c header) fixes every
rounding, so norm4 and cost planes must be identical as uint32.  That is stronger than the
north_star's 1e-4 relative depth / 1e-3 normal tolerance, and it is what makes a parity statement
through PatchMatch's argmin chain meaningful (an ulp flips a near-tie).
"""
import ctypes as C
import os

import numpy as np
import pytest

from gipuma_amd import abi, synth
from gipuma_amd.problem import Session, runcuda
from tests.oracle_lib import OracleState

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same(a, b, what):
    a, b = bits(a), bits(b)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        fa = a.view(np.float32)[tuple(bad[0])]
        fb = b.view(np.float32)[tuple(bad[0])]
        raise AssertionError("%s: %d of %d values differ, first at %s: %r vs %r"
                             % (what, len(bad), a.size, tuple(bad[0]), fa, fb))


def random_planes(gs, seed=7):
    """plausible random planes: unit normals facing the camera, d from a depth inside the range"""
    rng = np.random.default_rng(seed)
    n = rng.normal(size=(gs.rows, gs.cols, 3)).astype(np.float32)
    n[..., 2] = -np.abs(n[..., 2]) - 0.5
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    depth = rng.uniform(gs.params.depthMin * 1.2, gs.params.depthMax * 0.8,
                        size=(gs.rows, gs.cols)).astype(np.float32)
    pl = np.empty((gs.rows, gs.cols, 4), dtype=np.float32)
    pl[..., :3] = n
    # d = -n.X with X ~ depth * K^-1 (x, y, 1): close enough to getD_cu for a test input
    fx = gs.cameras.c_array[0].fx
    cx, cy = gs.cameras.c_array[0].K[2], gs.cameras.c_array[0].K[5]
    xs, ys = np.meshgrid(np.arange(gs.cols, dtype=np.float32), np.arange(gs.rows, dtype=np.float32))
    X = np.stack([(xs - cx) / fx * depth, (ys - cy) / fx * depth, depth], axis=-1)
    pl[..., 3] = -(n * X).sum(-1)
    return pl


def test_eval_cost_bit_exact(hip, tiny_problem):
    """kernel-level parity: cost of a GIVEN plane field (no argmin chain), pmCostMultiview_cu"""
    gs, _ = tiny_problem
    planes = random_planes(gs)
    with Session(gs) as s:
        got = s.eval_cost(planes)
    want = OracleState(gs).eval_cost(planes)
    assert_same(got, want, "eval_cost")
    assert (want < abi.MAXCOST).mean() > 0.9


def test_init_bit_exact(hip, tiny_problem):
    gs, _ = tiny_problem
    with Session(gs) as s:
        s.init_planes()
        n4, c = s.get_state()
    o = OracleState(gs)
    o.init_planes()
    assert_same(n4, o.norm4, "init norm4")
    assert_same(c, o.cost, "init cost")


def test_stage_by_stage_bit_exact(hip, tiny_problem):
    """every launch of the reference's schedule (gipuma.cu:1915-1935), compared after each one"""
    gs, _ = tiny_problem
    o = OracleState(gs)
    o.init_planes()
    with Session(gs) as s:
        s.set_state(o.norm4, o.cost)
        for it in range(2):
            for colour in (abi.BLACK, abi.RED):
                for stage in (abi.STAGE_CLOSE, abi.STAGE_FAR, abi.STAGE_REFINE):
                    s.sweep(it, colour, stage)
                    o.sweep(it, colour, stage)
                    n4, c = s.get_state()
                    assert_same(n4, o.norm4, "it %d colour %d stage %d norm4" % (it, colour, stage))
                    assert_same(c, o.cost, "it %d colour %d stage %d cost" % (it, colour, stage))
        s.finalize()
        o.finalize()
        n4, c = s.get_state()
        assert_same(n4, o.norm4, "finalize norm4")


def test_fused_sweep_equals_three_launches(hip, tiny_problem):
    gs, _ = tiny_problem
    o = OracleState(gs)
    o.init_planes()
    with Session(gs) as s:
        s.set_state(o.norm4, o.cost)
        s.sweep(0, abi.BLACK, abi.STAGE_ALL)
        fused = s.get_state()
        s.set_state(o.norm4, o.cost)
        for stage in (abi.STAGE_CLOSE, abi.STAGE_FAR, abi.STAGE_REFINE):
            s.sweep(0, abi.BLACK, stage)
        split = s.get_state()
    assert_same(fused[0], split[0], "fused vs split norm4")
    assert_same(fused[1], split[1], "fused vs split cost")


@pytest.mark.parametrize("cfg", [
    dict(),                                             # tiny default: box 7 (generic window), best-2
    dict(blocksize=11, n_best=3),                       # specialised box 11, best-3
    dict(cols=80, rows=56, blocksize=15, n_src=4, n_best=3, iterations=1),   # box 15
    dict(cols=70, rows=50, blocksize=9, n_src=2, n_best=1),  # cols/rows not multiples of the tile
    dict(cols=96, rows=40, blocksize=25, n_src=2, iterations=1),             # box 25
    dict(cols=90, rows=50, blocksize=19, n_src=3, n_best=2, iterations=2),   # box 19, the reference's default window
])
def test_full_run_bit_exact(hip, cfg):
    gs, _ = synth.build_problem(synth.tiny_config(**cfg))
    n4, c = runcuda(gs)
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "run norm4 %r" % cfg)
    assert_same(c, o_c, "run cost %r" % cfg)


@pytest.mark.parametrize("comb", [abi.COMB_ALL, abi.COMB_GOOD, abi.COMB_ANGLE])
def test_cost_combinations(hip, comb):
    gs, _ = synth.build_problem(synth.tiny_config(n_src=4, iterations=1), cost_comb=comb)
    n4, c = runcuda(gs)
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "comb %d norm4" % comb)
    assert_same(c, o_c, "comb %d cost" % comb)


def test_best_n_larger_than_four_uses_generic_combiner(hip):
    gs, _ = synth.build_problem(synth.tiny_config(n_src=6, n_best=5, iterations=1))
    n4, c = runcuda(gs)
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "n_best 5 norm4")
    assert_same(c, o_c, "n_best 5 cost")


def test_non_integer_image_takes_the_exp_path(hip):
    """float-valued reference image: no weight table, exp_model per sample"""
    gs, _ = synth.build_problem(synth.tiny_config(iterations=1))
    imgs = [im + np.float32(0.25) * (i + 1) for i, im in enumerate(gs.images)]
    from gipuma_amd.problem import GlobalState
    gs2 = GlobalState(imgs, gs.cameras, gs.selected, gs.params, seed=3)
    n4, c = runcuda(gs2)
    o_n4, o_c = OracleState(gs2).run()
    assert_same(n4, o_n4, "float image norm4")
    assert_same(c, o_c, "float image cost")


def test_no_selected_views_gives_maxcost(hip):
    """numConsidered == 0 -> MAXCOST everywhere (gipuma.cu:798-803), depth 0 after finalize"""
    gs, _ = synth.build_problem(synth.tiny_config(iterations=1))
    from gipuma_amd.problem import GlobalState
    gs0 = GlobalState(gs.images, gs.cameras, [], gs.params, seed=1)
    n4, c = runcuda(gs0)
    assert np.all(c == abi.MAXCOST)
    assert np.all(n4[..., 3] == 0.0)
    o_n4, o_c = OracleState(gs0).run()
    assert_same(n4, o_n4, "no views norm4")


def test_seed_changes_result_and_is_reproducible(hip, tiny_problem):
    gs, _ = tiny_problem
    a = runcuda(gs)
    b = runcuda(gs)
    assert_same(a[0], b[0], "same seed twice")
    gs.desc.seed = 99
    try:
        c = runcuda(gs)
    finally:
        gs.desc.seed = 1
    assert not np.array_equal(bits(a[0]), bits(c[0]))


def test_unfused_flag_matches(hip, tiny_problem):
    gs, _ = tiny_problem
    a = runcuda(gs)
    gs.desc.flags |= abi.FLAG_UNFUSED
    try:
        b = runcuda(gs)
    finally:
        gs.desc.flags &= ~abi.FLAG_UNFUSED
    assert_same(a[0], b[0], "unfused flag norm4")
    assert_same(a[1], b[1], "unfused flag cost")


def test_device_resident_images_with_pitch(hip):
    """images already in HBM (torch tensors), row pitch larger than cols"""
    import torch
    gs, _ = synth.build_problem(synth.tiny_config(iterations=1))
    pitch = gs.cols + 5
    dev = []
    for im in gs.images:
        t = torch.zeros((gs.rows, pitch), dtype=torch.float32, device="cuda:0")
        t[:, :gs.cols] = torch.from_numpy(im)
        dev.append(t)
    torch.cuda.synchronize()
    from gipuma_amd.problem import GlobalState
    gsd = GlobalState(dev, gs.cameras, gs.selected, gs.params, seed=1,
                      device_ptrs=[t.data_ptr() for t in dev], rows=gs.rows, cols=gs.cols, pitch=pitch)
    n4, c = runcuda(gsd)
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "device images norm4")
    assert_same(c, o_c, "device images cost")


def test_image_cache_is_pinned_by_live_sessions(hip):
    """GIPUMA_HIP_FLAG_CACHE_IMAGES: sessions share the packed copies of resident images; the cache refuses
    to free them while a session that samples them is alive, frees them afterwards, and a session created
    after the clear packs again and still gives the oracle's bits"""
    import torch
    gs, _ = synth.build_problem(synth.tiny_config(n_src=3, iterations=2))
    dev = [torch.from_numpy(np.ascontiguousarray(im)).to("cuda:0") for im in gs.images]
    torch.cuda.synchronize()
    from gipuma_amd.problem import GlobalState
    gsd = GlobalState(dev, gs.cameras, gs.selected, gs.params, seed=1, device_ptrs=[t.data_ptr() for t in dev],
                      rows=gs.rows, cols=gs.cols, flags=abi.FLAG_CACHE_IMAGES)
    lib = abi.load_library()
    o = OracleState(gs).run()
    with Session(gsd) as s1:
        with Session(gsd) as s2:
            assert lib.gipuma_hip_cache_clear() == abi.ERR_ARG  # two users
            s2.solve()
        assert lib.gipuma_hip_cache_clear() == abi.ERR_ARG      # one user left
        s1.solve()
        n4, c = s1.get_state()
    assert_same(n4, o[0], "cached images norm4")
    assert lib.gipuma_hip_cache_clear() == 0
    a = runcuda(gsd)
    assert_same(a[0], o[0], "after the clear norm4")
    assert lib.gipuma_hip_cache_clear() == 0


@pytest.mark.parametrize("tune", [1 << 30, 1 << 29, 1 << 28, (1 << 30) | (1 << 28), 1 << 27, 1 << 26,
                                  (1 << 26) | (1 << 29), 1 << 23, 1 << 25, (1 << 25) | (1 << 27),
                                  (1 << 27) | (1 << 25), 1 << 19, (1 << 19) | (1 << 27), 1 << 20, (1 << 20) | (1 << 26)])
def test_kernel_variants_are_bit_identical(hip, tune):
    """the performance-only choices of the sweep kernels -- float-encoded window offsets + the
    hand-pipelined loop (off: bit 30), task order owner-major (bit 29) / source-major (bit 28) in
    every iteration instead of switching after iteration 1, the column-per-lane kernel never (bit
    27) / in every half-sweep (bit 26) instead of the first four, the history skip rule off (bit
    23), early termination of view costs off (bit 25), refinement
    bounded per wavefront instead of per (candidate, view) item (bit 19: pm::refine_two_phase off), the tasks of the
    push / column-per-lane kernels in lane order instead of by disparity bucket (bit 20, round 6) -- must
    not change a single bit.
    Box 15 (the pipelined and column-per-lane instantiations), 4 iterations so that the default run
    uses both kernels and both task orders."""
    gs, _ = synth.build_problem(synth.tiny_config(cols=160, rows=112, n_src=4, blocksize=15, iterations=4,
                                                  n_best=3))
    os.environ["GIPUMA_HIP_ET_FORCE"] = "1"  # early termination also on this small frame
    try:
        a = runcuda(gs)
        os.environ["GIPUMA_HIP_TUNE"] = str(tune)
        b = runcuda(gs)
    finally:
        os.environ.pop("GIPUMA_HIP_TUNE", None)
        del os.environ["GIPUMA_HIP_ET_FORCE"]
    assert_same(a[0], b[0], "variant %d norm4" % tune)
    assert_same(a[1], b[1], "variant %d cost" % tune)
    if tune == 1 << 30:
        o = OracleState(gs).run()
        assert_same(a[0], o[0], "default vs oracle norm4")
        assert_same(a[1], o[1], "default vs oracle cost")


@pytest.mark.parametrize("theta", ["0.01,0.01,0.01", "0.5,0.7,0.9", "1,1,1", "1.2,2,3", "1e9,1e9,1e9"])
@pytest.mark.parametrize("cfg", [dict(cols=160, rows=112, n_src=4, blocksize=15, iterations=4, n_best=3),
                                 dict(cols=96, rows=80, n_src=5, blocksize=11, iterations=3, n_best=1),
                                 dict(cols=96, rows=80, n_src=2, blocksize=15, iterations=3, n_best=4)])
def test_early_termination_is_exact_for_any_bound(hip, theta, cfg):
    """refinement evaluations are cut off once every view's partial cost has reached
    theta * (cost to beat) and redone when that leaves the outcome open (pm::multiview_cost):
    exact for ANY theta.  Tiny theta forces the redo path on almost every wavefront, huge theta
    leaves only the value-exact rule; the column-per-lane kernel is switched off (bit 27) so that
    every half-sweep runs the pixel-per-lane kernel that has the cut-off."""
    gs, _ = synth.build_problem(synth.tiny_config(**cfg))
    o = OracleState(gs).run()
    os.environ["GIPUMA_HIP_ET_THETA"] = theta
    os.environ["GIPUMA_HIP_ET_FORCE"] = "1"  # (the library bounds evaluations only on frames of >= 1024 tiles)
    try:
        # one lane per pixel with the two-phase refinement (default) / bounded per wavefront
        for tune in (1 << 27, (1 << 27) | (1 << 19)):
            os.environ["GIPUMA_HIP_TUNE"] = str(tune)
            a = runcuda(gs)
            assert_same(a[0], o[0], "theta %s tune %d norm4" % (theta, tune))
            assert_same(a[1], o[1], "theta %s tune %d cost" % (theta, tune))
    finally:
        del os.environ["GIPUMA_HIP_ET_THETA"]
        del os.environ["GIPUMA_HIP_ET_FORCE"]
        os.environ.pop("GIPUMA_HIP_TUNE", None)


def test_two_phase_refinement_ragged_frame_many_views(hip):
    """the same on a frame that is not a multiple of the 32x16 tile (border tiles take the unbounded
    path, interior ones the item lists) with 12 source views (item groups of 5 + 5 + 2) and best-2"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=150, rows=100, n_src=12, blocksize=15, iterations=2, n_best=2))
    o = OracleState(gs).run()
    os.environ["GIPUMA_HIP_ET_FORCE"] = "2"
    try:
        a = runcuda(gs)
    finally:
        del os.environ["GIPUMA_HIP_ET_FORCE"]
    assert_same(a[0], o[0], "ragged norm4")
    assert_same(a[1], o[1], "ragged cost")


@pytest.mark.parametrize("g0", [1, 2, 5, 8])
@pytest.mark.parametrize("theta", ["0.05,0.05,0.05", "1,1,1"])
def test_two_phase_refinement_is_exact_for_any_split(hip, g0, theta):
    """pm::refine_two_phase: the first g0 window columns by the pixel's own lane, the rest by
    whichever lane picks the (candidate, view) item up; open candidates redone item by item (tiny
    theta: nearly all of them).  Any g0 (8 = the whole window in phase 1), any theta: the oracle's bits."""
    gs, _ = synth.build_problem(synth.tiny_config(cols=160, rows=112, n_src=7, blocksize=15, iterations=3, n_best=3))
    o = OracleState(gs).run()
    os.environ["GIPUMA_HIP_ET_THETA"] = theta
    os.environ["GIPUMA_HIP_ET_FORCE"] = "2"  # small frame, and every workgroup bounds every step
    os.environ["GIPUMA_HIP_TP_G0"] = str(g0)
    os.environ["GIPUMA_HIP_LB_K"] = "-1"  # (phase 1 = window columns, not the lower-bound prefilter)
    os.environ["GIPUMA_HIP_TUNE"] = str(1 << 27)
    try:
        a = runcuda(gs)
    finally:
        for k in ("GIPUMA_HIP_ET_THETA", "GIPUMA_HIP_ET_FORCE", "GIPUMA_HIP_TP_G0", "GIPUMA_HIP_TUNE", "GIPUMA_HIP_LB_K"):
            del os.environ[k]
    assert_same(a[0], o[0], "g0 %d theta %s norm4" % (g0, theta))
    assert_same(a[1], o[1], "g0 %d theta %s cost" % (g0, theta))


@pytest.mark.parametrize("lbk", [0, 2, 8, 12, 16, 32])
@pytest.mark.parametrize("theta", ["0.05,0.05,0.05", "1,1,1", "3,2,1.5"])
def test_lower_bound_prefilter_is_exact_for_any_length(hip, lbk, theta):
    """pm::lb_item: a refinement candidate's view costs are first bounded from below by the sum over the
    pixel's lbk heaviest window samples (pm::weight_order_kernel); an item whose bound reaches
    theta * (cost to beat) is decided, the others run the exact chain.  Any length (0 = chosen by the
    probe workgroups), any theta: the oracle's bits."""
    gs, _ = synth.build_problem(synth.tiny_config(cols=160, rows=112, n_src=7, blocksize=15, iterations=3, n_best=3))
    o = OracleState(gs).run()
    os.environ["GIPUMA_HIP_ET_THETA"] = theta
    os.environ["GIPUMA_HIP_ET_FORCE"] = "2"  # small frame, and every workgroup bounds every step
    os.environ["GIPUMA_HIP_LB_K"] = str(lbk)
    os.environ["GIPUMA_HIP_TUNE"] = str(1 << 27)
    try:
        a = runcuda(gs)
    finally:
        for k in ("GIPUMA_HIP_ET_THETA", "GIPUMA_HIP_ET_FORCE", "GIPUMA_HIP_LB_K", "GIPUMA_HIP_TUNE"):
            del os.environ[k]
    assert_same(a[0], o[0], "lbk %d theta %s norm4" % (lbk, theta))
    assert_same(a[1], o[1], "lbk %d theta %s cost" % (lbk, theta))


@pytest.mark.parametrize("lbk", [0, 6, 32])
@pytest.mark.parametrize("theta", ["0.05,0.05,0.05", "1,1,1.5"])
def test_lower_bound_prefilter_colour(hip, lbk, theta):
    """the same in the colour kernels (pm::lb_item_c4: three window loads per sample, |dB|+|dG|+|dR| weights)"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=96, rows=80, n_src=6, blocksize=15, iterations=3, n_best=3),
                                colour=True)
    o = OracleState(gs).run()
    a = _with_env({"GIPUMA_HIP_ET_THETA": theta, "GIPUMA_HIP_ET_FORCE": 2, "GIPUMA_HIP_LB_K": lbk}, lambda: runcuda(gs))
    assert_same(a[0], o[0], "colour lbk %d theta %s norm4" % (lbk, theta))
    assert_same(a[1], o[1], "colour lbk %d theta %s cost" % (lbk, theta))


@pytest.mark.parametrize("cfg", [dict(cols=96, rows=80, n_src=5, blocksize=11, iterations=3, n_best=1),
                                 dict(cols=96, rows=64, n_src=3, blocksize=25, iterations=2, n_best=2),
                                 dict(cols=96, rows=64, n_src=4, blocksize=19, iterations=3, n_best=2),
                                 dict(cols=150, rows=100, n_src=12, blocksize=15, iterations=2, n_best=2)])
def test_lower_bound_prefilter_other_boxes(hip, cfg):
    """the prefilter with the default (probe-chosen) length on boxes 11 and 25 and on a ragged frame with
    12 views (border tiles, three groups of views)"""
    gs, _ = synth.build_problem(synth.tiny_config(**cfg))
    o = OracleState(gs).run()
    os.environ["GIPUMA_HIP_ET_FORCE"] = "2"
    os.environ["GIPUMA_HIP_TUNE"] = str(1 << 27)
    try:
        a = runcuda(gs)
    finally:
        for k in ("GIPUMA_HIP_ET_FORCE", "GIPUMA_HIP_TUNE"):
            del os.environ[k]
    assert_same(a[0], o[0], "box %d norm4" % cfg["blocksize"])
    assert_same(a[1], o[1], "box %d cost" % cfg["blocksize"])


@pytest.mark.parametrize("theta", ["0.05,0.05,0.05", "1,1,1.5"])
@pytest.mark.parametrize("tune", [0, 1 << 19])
def test_early_termination_colour(hip, theta, tune):
    """the same cut-offs in the colour kernels (T = float4) -- two-phase by (candidate, view) items,
    or per wavefront (bit 19) --: exact for any bound"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=96, rows=80, n_src=6, blocksize=15, iterations=3, n_best=3),
                                colour=True)
    o = OracleState(gs).run()
    os.environ["GIPUMA_HIP_ET_THETA"] = theta
    os.environ["GIPUMA_HIP_ET_FORCE"] = "2"
    os.environ["GIPUMA_HIP_TUNE"] = str(tune)
    try:
        a = runcuda(gs)
    finally:
        del os.environ["GIPUMA_HIP_ET_THETA"]
        del os.environ["GIPUMA_HIP_ET_FORCE"]
        del os.environ["GIPUMA_HIP_TUNE"]
    assert_same(a[0], o[0], "colour theta %s norm4" % theta)
    assert_same(a[1], o[1], "colour theta %s cost" % theta)


def _with_env(env, fn):
    """run fn() with the variables of `env` set (a value of None: unset), then restore the environment"""
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("push", [0, 1, 2, 3, 100])
def test_push_propagation_is_bit_identical(hip, push):
    """pm_push.h: after a half-sweep every plane of its colour is evaluated once, by a group of 8 lanes,
    on the 208-point stencil its eight consumers share; the next half-sweep reads those costs instead
    of evaluating them.  GIPUMA_HIP_PUSH_LAUNCHES = how many leading half-sweeps do that (default 4):
    never, an odd number (the last push is consumed by the pixel-per-lane kernel's replay), every
    half-sweep (rule (H) decides who offers) -- not a bit may change, and the default equals the oracle."""
    gs, _ = synth.build_problem(synth.tiny_config(cols=160, rows=112, n_src=4, blocksize=15, iterations=4,
                                                  n_best=3))
    a = runcuda(gs)
    b = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": push, "GIPUMA_HIP_ET_FORCE": 1}, lambda: runcuda(gs))
    assert_same(a[0], b[0], "push %d norm4" % push)
    assert_same(a[1], b[1], "push %d cost" % push)
    if push == 0:
        o = OracleState(gs).run()
        assert_same(a[0], o[0], "default vs oracle norm4")
        assert_same(a[1], o[1], "default vs oracle cost")


@pytest.mark.parametrize("cfg,push", [
    (dict(cols=112, rows=80, n_src=4, blocksize=15, iterations=4, n_best=3), 100),
    (dict(cols=112, rows=80, n_src=4, blocksize=15, iterations=4, n_best=3), 3),
    (dict(cols=45, rows=37, n_src=9, blocksize=15, iterations=2, n_best=2), 100),   # ragged, two homography blocks
    (dict(cols=70, rows=21, n_src=1, blocksize=15, iterations=3, n_best=1), 100),
])
def test_push_propagation_colour(hip, cfg, push):
    """-color_processing (T = float4): pm::push_kernel_c4 -- three window loads and tap sets per stencil
    point, l1_norm reductions, the 766-entry weight table -- against the oracle's colour path"""
    gs, _ = synth.build_problem(synth.tiny_config(**cfg), colour=True)
    n4, c = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": push, "GIPUMA_HIP_ET_FORCE": 1}, lambda: runcuda(gs))
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "colour push norm4")
    assert_same(c, o_c, "colour push cost")


@pytest.mark.parametrize("cfg", [
    dict(cols=96, rows=80, n_src=3, blocksize=11, iterations=4, n_best=2),    # rows of 6 points: two wraps per step
    dict(cols=130, rows=70, n_src=4, blocksize=25, iterations=3, n_best=3),   # one family at a time in LDS
    dict(cols=45, rows=37, n_src=9, blocksize=25, iterations=2, n_best=4),    # ragged, two homography blocks
    dict(cols=40, rows=33, n_src=2, blocksize=11, iterations=3, n_best=1),
    dict(cols=130, rows=70, n_src=4, blocksize=19, iterations=3, n_best=3),   # box 19 (round 6): rows of 10 points, one family at a time
    dict(cols=45, rows=37, n_src=9, blocksize=19, iterations=2, n_best=2),
])
def test_push_propagation_other_boxes(hip, cfg):
    """boxes 11 and 25 run the push kernel's generic stencil loop (PushEval::family; box 25 with one
    family of the stencil in LDS at a time): every half-sweep pushed, against the oracle"""
    gs, _ = synth.build_problem(synth.tiny_config(**cfg))
    n4, c = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": 100}, lambda: runcuda(gs))
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "push box %d norm4" % cfg["blocksize"])
    assert_same(c, o_c, "push box %d cost" % cfg["blocksize"])


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("push,group_from", [(0, 0), (2, 2), (4, 5), (3, -1)])
@pytest.mark.parametrize("cfg", [dict(cols=160, rows=112, n_src=4, blocksize=15, iterations=4, n_best=3),
                                 dict(cols=150, rows=100, n_src=7, blocksize=11, iterations=3, n_best=2),
                                 dict(cols=64, rows=48, n_src=2, blocksize=15, iterations=3, n_best=4),
                                 dict(cols=150, rows=100, n_src=5, blocksize=25, iterations=3, n_best=3),
                                 dict(cols=150, rows=100, n_src=4, blocksize=19, iterations=3, n_best=3)])
def test_plane_keyed_propagation_is_bit_identical(hip, cfg, push, group_from, fused):
    """pm::sweep_group_kernel / pm::group_kernel (pm_group.h): the candidates of a tile grouped by plane bits, dis
    evaluated once per group on the bounding box of its windows, the reference's chain per task; from half-sweep
    `group_from` on (after `push` pushed ones; -1: never), fused with the sweep (one launch per half-sweep) or as a
    launch of its own in front of it -- always the oracle's bits, on frames that are not multiples of the tile too;
    boxes 11, 15 (chain weights in registers) and 25 (their table indices in registers)"""
    gs, _ = synth.build_problem(synth.tiny_config(**cfg))
    o = OracleState(gs).run()
    a = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": push, "GIPUMA_HIP_GROUP_FROM": group_from, "GIPUMA_HIP_ET_FORCE": 1,
                   "GIPUMA_HIP_GROUP_FUSED": fused}, lambda: runcuda(gs))
    assert_same(a[0], o[0], "grouped propagation push %d from %d norm4" % (push, group_from))
    assert_same(a[1], o[1], "grouped propagation push %d from %d cost" % (push, group_from))


def test_launch_times_of_a_solve(hip):
    """gipuma_hip_launch_times: one device time per half-sweep of the last timed solve, and how many
    leading half-sweeps read pushed costs (box 15: 4 by default, none when switched off)"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=96, rows=64, n_src=3, blocksize=15, iterations=3, n_best=2))
    with Session(gs) as s:
        assert s.launch_times() == ([], 0)
        t = s.solve(timing=True)
        ms, pushed = s.launch_times()
    assert len(ms) == 6 and pushed == 4 and all(m > 0 for m in ms)
    assert abs(sum(ms) - t.ms_sweeps) < 0.05 * t.ms_sweeps + 0.05

    def off():
        with Session(gs) as s2:
            s2.solve(timing=True)
            return s2.launch_times()
    ms0, pushed0 = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": 0}, off)
    assert len(ms0) == 6 and pushed0 == 0


@pytest.mark.parametrize("cols,rows,sel,over", [
    (33, 17, [1, 2, 3], dict(n_best=2)),                   # ragged, smaller than a tile
    (21, 9, [1, 2], dict(n_best=1)),                       # smaller than the window and the stencil
    (70, 40, [1, 2, 3, 4] * 3, dict(n_best=4)),            # 12 views: two homography exchange blocks
    (96, 64, [3], dict(n_best=3)),                         # one view, n_best > views
    (5, 3, [1, 2], dict(n_best=2)),                        # consumers at distance 5 fall outside
])
def test_push_propagation_edge_cases(hip, cols, rows, sel, over):
    """the push kernel in every half-sweep on ragged / tiny frames, many / one view(s), against the oracle"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=96, rows=64, n_src=4, blocksize=15, iterations=3))
    imgs = [np.ascontiguousarray(im[:rows, :cols]) for im in gs.images]
    g = _problem_from(gs, imgs, sel, **over)
    n4, c = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": 100}, lambda: runcuda(g))
    o_n4, o_c = OracleState(g).run()
    assert_same(n4, o_n4, "push %dx%d norm4" % (cols, rows))
    assert_same(c, o_c, "push %dx%d cost" % (cols, rows))


@pytest.mark.parametrize("seq", [
    # (iteration, colour, stages) or "set" = write the state back through set_state
    [(0, 0, 7), (0, 1, 7), (1, 0, 7), (1, 0, 7), (1, 1, 7), (1, 1, 7), (2, 0, 7), (2, 1, 7), (3, 0, 7), (3, 1, 7)],
    [(0, 0, 7), (0, 1, 7), (1, 0, 7), (1, 1, 7), (2, 1, 1), (2, 0, 7), (2, 1, 7), (3, 0, 7), (3, 1, 7), (4, 0, 7)],
    [(0, 0, 7), (0, 1, 7), (1, 0, 7), (1, 1, 7), (2, 0, 7), "set", (2, 1, 7), (3, 0, 7), (3, 1, 7), (4, 0, 7)],
    [(0, 1, 7), (0, 1, 7), (0, 0, 7), (1, 1, 7), (1, 0, 7), (2, 1, 7), (2, 0, 7), (3, 1, 7)],
])
@pytest.mark.parametrize("push", [4, 100])
def test_history_rule_survives_any_launch_sequence(hip, seq, push):
    """rule (H) -- skip a neighbour whose plane did not change in its last half-sweep -- is only
    valid inside a strictly alternating sequence of full half-sweeps; the session must notice
    repeated colours, partial stages and rewritten states by itself.  Each sequence is replayed on
    the oracle launch by launch.  push: the leading half-sweeps that read pushed costs (pm_push.h) --
    the session must also notice by itself when nobody has offered the costs a half-sweep wants."""
    gs, _ = synth.build_problem(synth.tiny_config(cols=128, rows=96, n_src=4, blocksize=15, iterations=5,
                                                  n_best=3))
    o = OracleState(gs)
    o.init_planes()

    def run():
        with Session(gs) as s:
            s.init_planes()
            for step in seq:
                if step == "set":
                    n4, c = s.get_state()
                    s.set_state(n4, c)
                    continue
                it, colour, stages = step
                s.sweep(it, colour, stages)
                o.sweep(it, colour, stages)
            return s.get_state()
    n4, c = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": push}, run)
    assert_same(n4, o.norm4, "sequence norm4")
    assert_same(c, o.cost, "sequence cost")


@pytest.mark.parametrize("cfg,colour", [
    (dict(cols=96, rows=80, n_src=3, blocksize=11, iterations=8, n_best=2), False),   # box 11, long run
    (dict(cols=72, rows=56, n_src=3, blocksize=7, iterations=6, n_best=2), False),    # runtime-sized window
    (dict(cols=80, rows=64, n_src=3, blocksize=11, iterations=5, n_best=2), True),    # colour kernels
])
def test_history_rule_on_other_kernels(hip, cfg, colour):
    """rule (H) lives in the setup shared by every sweep instantiation: long runs on the box-11,
    runtime-window and colour kernels with and without it, and against the oracle"""
    gs, _ = synth.build_problem(synth.tiny_config(**cfg), colour=colour)
    a = runcuda(gs)
    os.environ["GIPUMA_HIP_TUNE"] = str(1 << 23)
    try:
        b = runcuda(gs)
    finally:
        del os.environ["GIPUMA_HIP_TUNE"]
    assert_same(a[0], b[0], "history on/off norm4")
    assert_same(a[1], b[1], "history on/off cost")
    o = OracleState(gs).run()
    assert_same(a[0], o[0], "vs oracle norm4")
    assert_same(a[1], o[1], "vs oracle cost")


def test_packed_plane_beyond_the_float_offset_range(hip):
    """a packed plane of more than 2^21 entries cannot use float-encoded offsets: the session must
    fall back to integer addressing by itself.  1800x1200 is too large for the oracle to run whole,
    so: stored cost == fresh evaluation, and spot checks against the oracle's single-pixel cost,
    biased towards the far corner where the offsets are largest."""
    gs, _ = synth.build_problem("C", cols=1800, rows=1200, n_src=3, iterations=1, n_best=2)
    assert (gs.rows + 3) * (gs.cols + 8) > (1 << 21)
    with Session(gs) as s:
        s.init_planes()
        s.sweep(0, abi.BLACK)
        s.sweep(0, abi.RED)
        n4, c = s.get_state()
        assert_same(c, s.eval_cost(n4), "stored cost == cost of stored plane")
    from tests.oracle_lib import lib, fptr
    rng = np.random.default_rng(1)
    pts = [(gs.cols - 1 - int(rng.integers(0, 40)), gs.rows - 1 - int(rng.integers(0, 40))) for _ in range(24)]
    pts += [(int(rng.integers(0, gs.cols)), int(rng.integers(0, gs.rows))) for _ in range(24)]
    for x, y in pts:
        pl = np.ascontiguousarray(n4[y, x])
        want = lib().gipuma_oracle_multiview_cost(C.byref(gs.desc), x, y, fptr(pl))
        assert np.float32(want).view(np.uint32) == c[y, x].view(np.uint32), (x, y, want, c[y, x])


def test_config_a_full_size(hip):
    """BASELINE.json configs[0]: 320x240, 2 source views, 4 iterations -- the CPU-runnable case"""
    gs, info = synth.build_problem("A")
    n4, c, t = runcuda(gs, timing=True)
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "config A norm4")
    assert_same(c, o_c, "config A cost")
    err = np.abs(n4[..., 3] - info["gt_depth"]) / info["gt_depth"]
    assert (err < 0.01).mean() > 0.9  # and it actually reconstructs the surface


def test_properties_at_full_size(hip):
    """config C size (1600x1200, 10 views), where the oracle is too slow: size-independent
    properties instead -- determinism, plane/depth consistency, hemisphere, cost bounds, and the
    cost plane equals a fresh evaluation of the final planes."""
    gs, info = full_problem("C", iterations=1)
    with Session(gs) as s:
        s.init_planes()
        s.sweep(0, abi.BLACK)
        s.sweep(0, abi.RED)
        n4, c = s.get_state()
        again = s.eval_cost(n4)
        assert_same(c, again, "stored cost == cost of stored plane")
        nn = np.linalg.norm(n4[..., :3], axis=-1)
        assert np.abs(nn - 1).max() < 1e-4
        assert c.min() >= 0 and c.max() <= abi.MAXCOST
        s.finalize()
        f4, _ = s.get_state()
    valid = c != abi.MAXCOST
    d = f4[..., 3][valid]
    assert d.min() > 0
    # spot-check 64 pixels against the oracle's single-pixel cost
    from tests.oracle_lib import lib, fptr
    rng = np.random.default_rng(0)
    for _ in range(64):
        x, y = int(rng.integers(0, gs.cols)), int(rng.integers(0, gs.rows))
        pl = np.ascontiguousarray(n4[y, x])
        want = lib().gipuma_oracle_multiview_cost(C.byref(gs.desc), x, y, fptr(pl))
        assert np.float32(want).view(np.uint32) == c[y, x].view(np.uint32), (x, y, want, c[y, x])


_FULL = {}


def full_problem(cfg, colour=False, iterations=None):
    """the BASELINE-size synthetic problems are rendered once per test session (7-25 s each) and shared; a test that
    wants fewer iterations gets a GlobalState over the same frames and cameras with its own parameter block"""
    key = (cfg, colour)
    if key not in _FULL:
        _FULL[key] = synth.build_problem(cfg, colour=colour)
    gs, info = _FULL[key]
    if iterations is None or iterations == gs.params.iterations:
        return gs, info
    from gipuma_amd.problem import AlgorithmParameters, GlobalState
    ap = AlgorithmParameters(**{k: getattr(gs.params, k) for k in vars(gs.params)})
    ap.iterations = iterations
    return GlobalState(gs.images, gs.cameras, gs.selected, ap, seed=gs.desc.seed), info


def _teacher_forced_bands(gs, bands, what):
    """Every half-sweep launch of a full run at full frame size, checked exactly: the device state
    before launch k is handed to the oracle, which sweeps the rows of `bands` (a pixel's update
    depends only on the state before the launch), and those rows must equal the device state after
    launch k bit for bit.  Also the initial planes on the bands and the final conversion."""
    o = OracleState(gs)
    checked = 0
    with Session(gs) as s:
        s.init_planes()
        n4, c = s.get_state()
        o.norm4[:] = 0
        o.cost[:] = 0
        from tests.oracle_lib import lib, fptr
        for (y0, y1) in bands:  # init: compare the stored cost with the oracle's cost of the stored plane
            for y in range(y0, y1, 7):
                for x in range(3, gs.cols, 97):
                    pl = np.ascontiguousarray(n4[y, x])
                    want = lib().gipuma_oracle_multiview_cost(C.byref(gs.desc), x, y, fptr(pl))
                    assert np.float32(want).view(np.uint32) == c[y, x].view(np.uint32), (what, "init", x, y)
        for it in range(gs.params.iterations):
            for colour in (abi.BLACK, abi.RED):
                o.norm4[...] = n4
                o.cost[...] = c
                s.sweep(it, colour)
                n4, c = s.get_state()
                for (y0, y1) in bands:
                    o.sweep_band(it, colour, y0, y1)
                    assert_same(n4[y0:y1], o.norm4[y0:y1], "%s it %d colour %d rows %d..%d norm4" % (what, it, colour, y0, y1))
                    assert_same(c[y0:y1], o.cost[y0:y1], "%s it %d colour %d rows %d..%d cost" % (what, it, colour, y0, y1))
                    checked += (y1 - y0) * gs.cols // 2
        o.norm4[...] = n4
        o.cost[...] = c
        s.finalize()
        f4, _ = s.get_state()
    o.finalize()
    assert_same(f4, o.norm4, what + " final conversion")
    return checked


def test_config_b_full_run(hip):
    """BASELINE config B (640x480, 6 source views, box 11, 8 iterations, best-3) in full against the
    oracle's own free-running solve (reference loop gipuma.cu:1911-1941 at the sizes of
    scripts/templeRing.sh), bit for bit."""
    gs, info = synth.build_problem("B")
    n4, c = runcuda(gs)
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "config B norm4")
    assert_same(c, o_c, "config B cost")


def test_config_c_every_launch_at_full_size(hip):
    """BASELINE config C (the headline: 1600x1200, 10 source views, box 15, 8 iterations, best-3):
    all 16 half-sweep launches of the shipped schedule (column-per-lane kernel, fused kernel with
    early termination, history rule) checked exactly on three bands of rows -- top border, interior
    across a tile-band boundary, bottom border (scripts/dtu_fast.sh:9-21)."""
    gs, info = full_problem("C")
    n = _teacher_forced_bands(gs, [(0, 12), (592, 612), (1190, 1200)], "config C")
    print("config C: %d pixel updates compared exactly" % n)


EXHAUSTIVE_TUNE = 64 | (1 << 23) | (1 << 25)  # no skip rules (A)/(D), no history rule (H), no bounded evaluation


def _default_equals_exhaustive(gs, what):
    """the shipped schedule (push propagation, skip rules, prefilter + bounded refinement) against the
    exhaustive one -- every candidate of every pixel evaluated in full by the plain kernel -- on the
    WHOLE frame, final maps and costs bit for bit"""
    a = runcuda(gs)
    b = _with_env({"GIPUMA_HIP_TUNE": EXHAUSTIVE_TUNE}, lambda: runcuda(gs))
    assert_same(a[0], b[0], what + ": default vs exhaustive schedule, norm4 (whole frame)")
    assert_same(a[1], b[1], what + ": default vs exhaustive schedule, cost (whole frame)")
    return a


def test_config_c_whole_frame_against_the_oracle(hip):
    """BASELINE config C, the headline (1600x1200, 10 source views, box 15, 8 iterations, best-3;
    scripts/dtu_fast.sh:9-21): the WHOLE free-running solve -- the reference's loop gipuma.cu:1911-1941,
    30.7 M pixel updates -- against the oracle's own free-running solve, every pixel of the final maps
    and costs bit for bit; and the shipped schedule against the exhaustive one on the same frame."""
    import time
    gs, info = full_problem("C")
    a = _default_equals_exhaustive(gs, "config C")
    t0 = time.time()
    o_n4, o_c = OracleState(gs).run()
    print("config C: oracle free-running solve %.1f s" % (time.time() - t0))
    assert_same(a[0], o_n4, "config C whole frame norm4")
    assert_same(a[1], o_c, "config C whole frame cost")


def test_config_d_and_colour_default_equals_exhaustive(hip):
    """config D (20 views, box 25) and the colour variant of config C's geometry: the shipped schedule
    against the exhaustive one on the whole frame (the exhaustive kernel itself is teacher-forced against
    the oracle in test_exhaustive_schedule_every_launch_at_full_size)"""
    gs, info = full_problem("D", iterations=3)
    _default_equals_exhaustive(gs, "config D")
    gs, info = full_problem("C", colour=True, iterations=3)
    _default_equals_exhaustive(gs, "colour config C")


def test_stepped_noisy_scene_whole_run_against_the_oracle(hip):
    """the scene bench.py reports as value_scene_steps -- depth steps of +-30 mm, an occluding disc, sigma = 2 sensor
    noise: candidates are accepted far more often, bounds hold less often, whole regions lose their best views --
    with config C's cameras and parameters on an 832x640 frame (1040 tiles: bounded evaluation and the plane-keyed
    propagation kernel run as on the full frame): the whole free-running solve against the oracle's, and the shipped
    schedule against the exhaustive one"""
    gs, info = synth.build_problem("C", cols=832, rows=640, scene="steps")
    a = _default_equals_exhaustive(gs, "stepped scene")
    o_n4, o_c = OracleState(gs).run()
    assert_same(a[0], o_n4, "stepped scene norm4")
    assert_same(a[1], o_c, "stepped scene cost")
    gt = info["gt_depth"]
    assert (np.abs(a[0][..., 3] - gt) / gt < 0.01).mean() > 0.9  # (and it reconstructs the scene)


def test_patchy_scene_whole_run_against_the_oracle(hip):
    """the scene bench.py reports as value_scene_patchy -- 30 % of the surface with a flat albedo (patch costs tie:
    more near-ties for the argmin chain, fewer bounds that hold, more groups of one plane), a band with a periodic
    texture, sigma = 1 sensor noise -- with config C's cameras and parameters on an 832x640 frame: the whole
    free-running solve against the oracle's, and the shipped schedule against the exhaustive one"""
    gs, info = synth.build_problem("C", cols=832, rows=640, scene="patchy")
    ref = np.asarray(gs.images[0])
    flat_share = float((np.abs(ref - np.round(20.0 + 215.0 * 0.45)) <= 3.0).mean())
    assert 0.15 < flat_share < 0.5, flat_share   # the scenario bites: a large flat share was rendered
    a = _default_equals_exhaustive(gs, "patchy scene")
    o_n4, o_c = OracleState(gs).run()
    assert_same(a[0], o_n4, "patchy scene norm4")
    assert_same(a[1], o_c, "patchy scene cost")


@pytest.mark.parametrize("cfg,colour", [("D", False), ("C", True)])
def test_config_d_and_colour_all_iterations_against_the_oracle(hip, cfg, colour):
    """config D's parameters (20 views, box 25, 8 iterations; scripts/dtu_accurate.sh) and the colour variant of
    config C's (10 views, box 15, 8 iterations, T = float4), free-running through ALL their iterations -- every push
    launch, every column-per-lane launch, every fused launch with the prefilter, bounded items and their redo
    paths -- on an 832x640 frame of the DTU cameras (1040 tiles: the smallest 800x600-class frame that keeps the
    schedule of the full frame -- bounded evaluation needs >= 1024 tiles): final maps and costs of every pixel
    against the oracle's, bit for bit, and the shipped schedule against the exhaustive one.  (At 1600x1200 the same
    configurations are covered launch by launch on bands and by default == exhaustive on whole frames; the oracle's
    free-running solve of a full config-D frame takes ten minutes.)"""
    import time
    gs, info = synth.build_problem(cfg, cols=832, rows=640, colour=colour)
    assert gs.params.iterations == 8
    a = _default_equals_exhaustive(gs, "config %s%s 832x640" % (cfg, " colour" if colour else ""))
    t0 = time.time()
    o_n4, o_c = OracleState(gs).run()
    print("config %s%s, 832x640, 8 iterations: oracle free-running solve %.1f s" % (cfg, " colour" if colour else "", time.time() - t0))
    assert_same(a[0], o_n4, "config %s colour=%r 832x640 norm4" % (cfg, colour))
    assert_same(a[1], o_c, "config %s colour=%r 832x640 cost" % (cfg, colour))


@pytest.mark.parametrize("cfg,colour", [("D", False), ("C", True)])
def test_config_d_and_colour_whole_frame_two_iterations_against_the_oracle(hip, cfg, colour):
    """BASELINE config D (1600x1200, 20 source views, box 25; scripts/dtu_accurate.sh) and the colour variant of config C
    at the FULL frame size, free-running through the first two iterations -- the push launches, the column-per-lane
    launches and the first plane-keyed launch at 3750 tiles -- against the oracle's solve of the same frame: every pixel
    of the final maps and costs, bit for bit.  (All eight iterations: the 832x640 test above; every launch of all eight at
    this size on bands: test_config_d_every_launch_at_full_size, test_colour_every_launch_at_full_size.)"""
    import time
    gs, info = full_problem(cfg, colour=colour, iterations=2)
    n4, c = runcuda(gs)
    t0 = time.time()
    o_n4, o_c = OracleState(gs).run()
    print("config %s%s, %dx%d, 2 iterations: oracle free-running solve %.1f s"
          % (cfg, " colour" if colour else "", gs.cols, gs.rows, time.time() - t0))
    assert_same(n4, o_n4, "config %s colour=%r full frame norm4" % (cfg, colour))
    assert_same(c, o_c, "config %s colour=%r full frame cost" % (cfg, colour))


@pytest.mark.parametrize("cfg,kw,bands", [("C", {}, [(0, 6), (604, 612)]), ("D", dict(iterations=2), [(0, 3), (606, 610)]),
                                          ("C", dict(colour=True, iterations=2), [(0, 3), (606, 610)])])
def test_exhaustive_schedule_every_launch_at_full_size(hip, cfg, kw, bands):
    """the exhaustive schedule (what the whole-frame comparisons above are made against) teacher-forced
    against the oracle on bands of rows, every launch"""
    gs, info = full_problem(cfg, **kw)
    n = _with_env({"GIPUMA_HIP_TUNE": EXHAUSTIVE_TUNE}, lambda: _teacher_forced_bands(gs, bands, "exhaustive " + cfg))
    print("exhaustive %s %r: %d pixel updates compared exactly" % (cfg, kw, n))


def test_config_d_every_launch_at_full_size(hip):
    """BASELINE config D (1600x1200, 20 source views, box 25, 8 iterations; scripts/dtu_accurate.sh):
    every launch checked exactly on two bands (the oracle costs 5x config C per pixel here)."""
    gs, info = full_problem("D")
    n = _teacher_forced_bands(gs, [(0, 6), (156, 164), (602, 610), (1194, 1200)], "config D")
    print("config D: %d pixel updates compared exactly" % n)


# ------------------------------------------------------------------------------------------------
# -color_processing (T = float4)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tune", [1 << 27, 1 << 26, (1 << 26) | (1 << 23)])
def test_colour_column_per_lane_kernels(hip, tune):
    """colour, box 15: init and the first half-sweeps evaluate column-per-lane (view_cost_cols_c4: lane c
    = window column c, three window loads per sample, the relay keeps the summation order) -- never
    (bit 27) / in every half-sweep (bit 26) must give the bits of the default schedule and of the oracle;
    12 views: two homography exchange blocks"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=112, rows=72, n_src=4, blocksize=15, iterations=3, n_best=3),
                                colour=True)
    imgs = gs.images
    g = _problem_from(gs, imgs, [1, 2, 3, 4] * 3, n_best=3)
    a = runcuda(g)
    b = _with_env({"GIPUMA_HIP_TUNE": tune, "GIPUMA_HIP_PUSH_LAUNCHES": 2}, lambda: runcuda(g))
    assert_same(a[0], b[0], "colour cols variant %d norm4" % tune)
    assert_same(a[1], b[1], "colour cols variant %d cost" % tune)
    if tune == 1 << 27:
        o = OracleState(g).run()
        assert_same(a[0], o[0], "colour default vs oracle norm4")
        assert_same(a[1], o[1], "colour default vs oracle cost")


def test_colour_every_launch_at_full_size(hip):
    """the colour variant of config C's geometry (1600x1200, 10 source views, box 15, best-3) as bench.py
    --colour runs it, four iterations: every one of the 8 half-sweep launches -- six of them fed by
    pm::push_kernel_c4, then the plain colour sweep kernel with the history rule -- checked exactly on
    two bands of rows (top border, interior across a tile boundary)"""
    gs, info = full_problem("C", colour=True, iterations=4)
    n = _teacher_forced_bands(gs, [(0, 4), (606, 612)], "colour config C")
    print("colour config C: %d pixel updates compared exactly" % n)



@pytest.mark.parametrize("cfg", [
    dict(blocksize=7, n_src=3, n_best=2),              # runtime-sized window
    dict(cols=80, rows=56, blocksize=15, n_src=4, n_best=3, iterations=1),
    dict(cols=70, rows=50, blocksize=11, n_src=2, n_best=1),
])
def test_colour_full_run_bit_exact(hip, cfg):
    gs, _ = synth.build_problem(synth.tiny_config(**cfg), colour=True)
    assert gs.channels == 4
    n4, c = runcuda(gs)
    o_n4, o_c = OracleState(gs).run()
    assert_same(n4, o_n4, "colour run norm4 %r" % cfg)
    assert_same(c, o_c, "colour run cost %r" % cfg)


def test_colour_stage_by_stage_and_generic_combiner(hip):
    gs, _ = synth.build_problem(synth.tiny_config(n_src=4, iterations=1), colour=True, cost_comb=abi.COMB_GOOD)
    o = OracleState(gs)
    o.init_planes()
    with Session(gs) as s:
        s.init_planes()
        n4, c = s.get_state()
        assert_same(n4, o.norm4, "colour init norm4")
        assert_same(c, o.cost, "colour init cost")
        for colour in (abi.BLACK, abi.RED):
            for stage in (abi.STAGE_CLOSE, abi.STAGE_FAR, abi.STAGE_REFINE):
                s.sweep(0, colour, stage)
                o.sweep(0, colour, stage)
                n4, c = s.get_state()
                assert_same(n4, o.norm4, "colour %d stage %d norm4" % (colour, stage))
                assert_same(c, o.cost, "colour %d stage %d cost" % (colour, stage))


def test_colour_float_valued_images_and_ignored_alpha(hip):
    """non-integer colour images take the float4 gather path; the alpha channel is never read"""
    from gipuma_amd.problem import GlobalState
    gs, _ = synth.build_problem(synth.tiny_config(iterations=1), colour=True)
    imgs = []
    for i, im in enumerate(gs.images):
        im = im.copy()
        im[..., :3] += np.float32(0.125) * (i + 1)
        im[..., 3] = np.float32(1e30) if i % 2 else np.float32("nan")     # must not matter
        imgs.append(im)
    gs2 = GlobalState(imgs, gs.cameras, gs.selected, gs.params, seed=5)
    n4, c = runcuda(gs2)
    o_n4, o_c = OracleState(gs2).run()
    assert_same(n4, o_n4, "float colour norm4")
    assert_same(c, o_c, "float colour cost")
    assert np.isfinite(c).all()


def test_fast_reciprocal_is_exact_on_this_device(hip):
    """the kernels replace the IEEE divide 1.0f/z by v_rcp_f32 + one Newton step wherever the
    whole window's |z| is inside [2^-100, 2^100]; exhaustive check over every float with biased
    exponent 1..252 that the two agree bit for bit on this GPU"""
    n = C.c_ulonglong(123)
    assert hip.gipuma_hip_selftest_reciprocal(0, C.byref(n)) == 0
    assert n.value == 0


def test_markstein_quotient_is_the_ieee_quotient(hip):
    """x / z of the warped point (getCorrespondingPoint_cu, gipuma.cu:207-217; vecdiv4, config.h:44-47) in the default and the
    reference-order flavour: r = RN(1/z), q = RN(x r), q' = RN(q + RN(x - q z) r).  The proof is by exhaustion over all
    2^23 x 2^23 significand pairs (31 s on an MI355X: profiles/r06_selftest_quotient.txt); here three slices of the
    denominators -- the first, one across the middle, the last 2^13 -- against all 2^23 numerators each: 2 * 10^11 pairs"""
    n = C.c_ulonglong(1)
    total = 0
    for z0 in (0, (1 << 22) - (1 << 12), (1 << 23) - (1 << 13)):
        assert hip.gipuma_hip_selftest_quotient(0, z0, 1 << 13, C.byref(n)) == 0
        total += n.value
    assert total == 0
    assert hip.gipuma_hip_selftest_quotient(0, 1 << 23, 1, C.byref(n)) == abi.ERR_ARG  # out of the significand range



def test_degenerate_planes_take_the_safe_divide_path(hip, tiny_problem):
    """planes through / near the source camera centres make some warped denominators tiny, zero or
    negative: the window guard must fall back to the IEEE divide and still match the oracle"""
    gs, _ = tiny_problem
    rng = np.random.default_rng(3)
    planes = random_planes(gs, seed=11)
    planes[..., 3] *= rng.choice([1e-6, 1e-3, -1.0, 1.0, 1e3, 0.0], size=planes.shape[:2]).astype(np.float32)
    planes[0, 0, 3] = np.float32("nan")
    with Session(gs) as s:
        got = s.eval_cost(planes)
    want = OracleState(gs).eval_cost(planes)
    assert_same(got, want, "degenerate eval_cost")


# ------------------------------------------------------------------------------------------------
# edge cases: ragged / tiny images, maximum view count, largest window, zero iterations
# ------------------------------------------------------------------------------------------------
def _problem_from(gs, images, selected, **ap_over):
    from gipuma_amd.problem import AlgorithmParameters, GlobalState
    ap = AlgorithmParameters(**{k: getattr(gs.params, k) for k in vars(gs.params)})
    for k, v in ap_over.items():
        setattr(ap, k, v)
    return GlobalState(images, gs.cameras, selected, ap, seed=7)


@pytest.mark.parametrize("cols,rows", [(33, 17), (5, 3), (1, 1), (64, 2), (3, 40)])
def test_ragged_and_tiny_images(hip, cols, rows):
    """sizes that are not tile multiples, smaller than the window, smaller than the far-neighbour
    distance, down to a single pixel"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=64, rows=48, n_src=2, blocksize=7, iterations=2))
    imgs = [np.ascontiguousarray(im[:rows, :cols]) for im in gs.images]
    g = _problem_from(gs, imgs, gs.selected)
    n4, c = runcuda(g)
    o_n4, o_c = OracleState(g).run()
    assert_same(n4, o_n4, "ragged %dx%d norm4" % (cols, rows))
    assert_same(c, o_c, "ragged %dx%d cost" % (cols, rows))


@pytest.mark.parametrize("cols,rows,sel,over", [
    (21, 9, [1, 2, 3], dict(cost_comb=abi.COMB_ALL)),            # image smaller than the window
    (70, 40, [1, 2, 3, 4] * 3, dict(n_best=5)),                  # 12 views: two exchange blocks; LDS combiner
    (37, 33, [1, 2, 3, 4, 1, 2, 3, 4, 1], dict(cost_comb=abi.COMB_GOOD)),  # 9 views, ragged, GOOD
])
def test_column_per_lane_kernel_edge_cases(hip, cols, rows, sel, over):
    """box 15 on packed gray planes runs its first three half-sweeps (and init) column-per-lane:
    ragged / tiny images, more views than one group exchanges at once, every combiner"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=96, rows=64, n_src=4, blocksize=15, iterations=2))
    imgs = [np.ascontiguousarray(im[:rows, :cols]) for im in gs.images]
    g = _problem_from(gs, imgs, sel, **over)
    n4, c = runcuda(g)
    o_n4, o_c = OracleState(g).run()
    assert_same(n4, o_n4, "cols %dx%d norm4" % (cols, rows))
    assert_same(c, o_c, "cols %dx%d cost" % (cols, rows))


def test_maximum_number_of_views(hip):
    """32 selected views = the reference's costVector[32] limit (gipuma.cu:736), ALL combiner"""
    from gipuma_amd.cameras import CameraSet
    gs, _ = synth.build_problem(synth.tiny_config(cols=48, rows=32, n_src=4, blocksize=7, iterations=1))
    n = 33
    cs = CameraSet(n)
    imgs = []
    for i in range(n):
        src = 0 if i == 0 else 1 + (i - 1) % 4
        C.memmove(C.byref(cs.c_array[i]), C.byref(gs.cameras.c_array[src]), C.sizeof(abi.Camera))
        imgs.append(gs.images[src])
    cs.f = gs.cameras.f
    from gipuma_amd.problem import AlgorithmParameters, GlobalState
    for comb in (abi.COMB_BEST_N, abi.COMB_ALL):
        ap = AlgorithmParameters(iterations=1, n_best=3, cost_comb=comb, depthMin=300.0, depthMax=800.0)
        ap.set_blocksize(7)
        g = GlobalState(imgs, cs, list(range(1, n)), ap, seed=2)
        n4, c = runcuda(g)
        o_n4, o_c = OracleState(g).run()
        assert_same(n4, o_n4, "32 views comb %d norm4" % comb)
        assert_same(c, o_c, "32 views comb %d cost" % comb)
    lib = abi.load_library()
    g.desc.n_selected = 33
    h = C.c_void_p()
    assert lib.gipuma_hip_create(C.byref(g.desc), C.byref(h)) == -1      # one too many
    g.desc.n_selected = 32


@pytest.mark.parametrize("box,colour", [(15, False), (25, False), (19, False), (15, True)])
def test_maximum_number_of_views_pushed(hip, box, colour):
    """32 selected views with the push kernels in every half-sweep: four homography exchange blocks per
    plane (gray unrolled loop, generic loop with one stencil family in LDS at a time, colour)"""
    from gipuma_amd.cameras import CameraSet
    from gipuma_amd.problem import AlgorithmParameters, GlobalState
    gs, _ = synth.build_problem(synth.tiny_config(cols=72, rows=40, n_src=4, blocksize=box, iterations=2),
                                colour=colour)
    n = 33
    cs = CameraSet(n)
    imgs = []
    for i in range(n):
        src = 0 if i == 0 else 1 + (i - 1) % 4
        C.memmove(C.byref(cs.c_array[i]), C.byref(gs.cameras.c_array[src]), C.sizeof(abi.Camera))
        imgs.append(gs.images[src])
    cs.f = gs.cameras.f
    ap = AlgorithmParameters(iterations=2, n_best=4, depthMin=300.0, depthMax=800.0)
    ap.set_blocksize(box)
    g = GlobalState(imgs, cs, list(range(1, n)), ap, seed=3)
    n4, c = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": 100}, lambda: runcuda(g))
    o_n4, o_c = OracleState(g).run()
    assert_same(n4, o_n4, "32 views pushed, box %d norm4" % box)
    assert_same(c, o_c, "32 views pushed, box %d cost" % box)


def test_largest_window_and_rectangular_window(hip):
    gs, _ = synth.build_problem(synth.tiny_config(cols=64, rows=48, n_src=2, blocksize=7, iterations=1))
    for bh, bv in ((49, 49), (11, 5), (3, 25), (1, 1)):
        g = _problem_from(gs, gs.images, gs.selected, box_hsize=bh, box_vsize=bv)
        n4, c = runcuda(g)
        o_n4, o_c = OracleState(g).run()
        assert_same(n4, o_n4, "box %dx%d norm4" % (bh, bv))
        assert_same(c, o_c, "box %dx%d cost" % (bh, bv))


def test_zero_iterations_is_init_plus_final_conversion(hip, tiny_problem):
    gs, _ = tiny_problem
    g = _problem_from(gs, gs.images, gs.selected, iterations=0)
    n4, c = runcuda(g)
    o = OracleState(g)
    o.init_planes()
    o.finalize()
    assert_same(n4, o.norm4, "0 iterations norm4")
    assert_same(c, o.cost, "0 iterations cost")


def test_n_best_larger_than_number_of_views(hip, tiny_problem):
    gs, _ = tiny_problem
    g = _problem_from(gs, gs.images, gs.selected[:1], n_best=4)      # one view, best-4
    n4, c = runcuda(g)
    o_n4, o_c = OracleState(g).run()
    assert_same(n4, o_n4, "n_best > views norm4")
    assert_same(c, o_c, "n_best > views cost")


# ------------------------------------------------------------------------------------------------
# exact skipping of propagation candidates (sweep_kernel): must never change a result
# ------------------------------------------------------------------------------------------------
def test_skipping_matches_exhaustive_evaluation(hip):
    """same run with every candidate evaluated (GIPUMA_HIP_TUNE bit 64) and with the skip rules:
    identical bits, on a converging problem where most candidates end up skipped"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=128, rows=96, n_src=4, blocksize=9, iterations=6,
                                                  n_best=3))
    a = runcuda(gs)
    os.environ["GIPUMA_HIP_TUNE"] = "64"
    try:
        b = runcuda(gs)
    finally:
        del os.environ["GIPUMA_HIP_TUNE"]
    assert_same(a[0], b[0], "skip vs exhaustive norm4")
    assert_same(a[1], b[1], "skip vs exhaustive cost")
    o = OracleState(gs).run()
    assert_same(a[0], o[0], "skip vs oracle norm4")


def test_inconsistent_installed_costs_are_not_trusted(hip, tiny_problem):
    """gipuma_hip_set_state may install costs that are NOT the cost of the installed planes; the
    reference would then re-evaluate a neighbour plane equal to the pixel's own and adopt its
    (lower) true cost.  Rule (A) must be off for such a session: compare with the oracle, which
    always evaluates."""
    gs, _ = tiny_problem
    o = OracleState(gs)
    o.init_planes()
    # make many neighbours share planes, and inflate every stored cost
    o.norm4[1::2, :] = o.norm4[0:-1:2, :]
    o.norm4[:, 1::2] = o.norm4[:, 0:-1:2]
    o.cost[:] = o.eval_cost(o.norm4) * np.float32(1.5) + np.float32(1.0)
    with Session(gs) as s:
        s.set_state(o.norm4, o.cost)
        for it in range(2):
            for colour in (abi.BLACK, abi.RED):
                before_n4, before_c = o.norm4.copy(), o.cost.copy()
                s.sweep(it, colour, abi.STAGE_CLOSE)
                o.sweep(it, colour, abi.STAGE_CLOSE)
                n4, c = s.get_state()
                assert_same(n4, o.norm4, "untrusted it %d colour %d norm4" % (it, colour))
                assert_same(c, o.cost, "untrusted it %d colour %d cost" % (it, colour))
                if it == 0 and colour == abi.BLACK:
                    # the scenario bites: pixels that kept their plane but got a new (true) cost --
                    # exactly what rule (A) would have missed had it trusted the installed costs
                    kept = (bits(before_n4) == bits(o.norm4)).all(-1)
                    assert (kept & (before_c != o.cost)).sum() > 50


@pytest.mark.parametrize("fused", [0, 1])  # (colour sessions always run unfused; the switch must not change that)
@pytest.mark.parametrize("push,group_from", [(0, 0), (2, 3), (6, 6)])
def test_plane_keyed_propagation_colour(hip, push, group_from, fused):
    """pm::group_kernel<15, 4> + the sweep kernel: the plane-keyed propagation for -color_processing (three window loads
    and tap sets per sample, l1_norm(float4) reductions, weights from the 766-entry table), ragged frame, rule (S) on --
    the oracle's bits"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=150, rows=100, n_src=4, blocksize=15, iterations=4, n_best=3),
                                colour=True)
    o = OracleState(gs).run()
    a = _with_env({"GIPUMA_HIP_PUSH_LAUNCHES": push, "GIPUMA_HIP_GROUP_FROM": group_from, "GIPUMA_HIP_ET_FORCE": 1,
                   "GIPUMA_HIP_GROUP_FUSED": fused}, lambda: runcuda(gs))
    assert_same(a[0], o[0], "colour grouped propagation push %d from %d norm4" % (push, group_from))
    assert_same(a[1], o[1], "colour grouped propagation push %d from %d cost" % (push, group_from))


def test_seen_rule_is_reset_when_planes_are_installed(hip):
    """skip rule (S) (colour sessions) -- a plane a pixel's propagation evaluated before can never be accepted,
    because the pixel's cost only decreases -- must forget its rings when the caller installs state: after two
    iterations every stored cost is RAISED through gipuma_hip_set_state (planes unchanged), so planes
    turned down before are acceptable again; the oracle, which always evaluates, adopts many of them."""
    gs, _ = synth.build_problem(synth.tiny_config(cols=96, rows=64, n_src=3, blocksize=11, iterations=4, n_best=2),
                                colour=True)
    o = OracleState(gs)
    o.init_planes()
    with Session(gs) as s:
        s.init_planes()
        for it in range(2):
            for colour in (abi.BLACK, abi.RED):
                s.sweep(it, colour)
                o.sweep(it, colour)
        n4, c = s.get_state()
        assert_same(n4, o.norm4, "before the install, norm4")
        o.cost[:] = o.cost * np.float32(4.0) + np.float32(1.0)
        s.set_state(o.norm4, o.cost)
        changed = 0
        for it in range(2, 4):
            for colour in (abi.BLACK, abi.RED):
                before = o.norm4.copy()
                s.sweep(it, colour, abi.STAGE_CLOSE | abi.STAGE_FAR)
                o.sweep(it, colour, abi.STAGE_CLOSE | abi.STAGE_FAR)
                n4, c = s.get_state()
                assert_same(n4, o.norm4, "after the install, it %d colour %d norm4" % (it, colour))
                assert_same(c, o.cost, "after the install, it %d colour %d cost" % (it, colour))
                changed += int((bits(before) != bits(o.norm4)).any(-1).sum())
        assert changed > 500  # the scenario bites: propagation alone re-adopts planes


def test_shared_planes_with_consistent_costs(hip, tiny_problem):
    """the opposite corner: after init_planes the invariant holds, force many identical neighbour
    planes through the kernels' own state by running sweeps to convergence and compare each"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=64, rows=48, n_src=2, blocksize=7, iterations=1))
    o = OracleState(gs)
    with Session(gs) as s:
        s.init_planes()
        o.init_planes()
        for it in range(5):
            for colour in (abi.BLACK, abi.RED):
                s.sweep(it, colour, abi.STAGE_CLOSE | abi.STAGE_FAR)      # propagation only: planes spread
                o.sweep(it, colour, abi.STAGE_CLOSE | abi.STAGE_FAR)
        n4, c = s.get_state()
        assert_same(n4, o.norm4, "propagation-only norm4")
        assert_same(c, o.cost, "propagation-only cost")
        same_as_right = (n4[:, 1:].view(np.uint32) == n4[:, :-1].view(np.uint32)).all(-1).mean()
        assert same_as_right > 0.3        # the scenario really has shared planes


# ------------------------------------------------------------------------------------------------
# the HIP path against the committed outputs of the REFERENCE's own device code (tests/golden/)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fixture", ["ref_tiny64", "ref_box15"])
def test_hip_against_reference_golden_fixture(hip, fixture):
    """tests/golden/*.npz hold what /root/reference/gipuma.cu itself (compiled for the CPU,
    oracle/ref_shim) produced on two inputs (box 7: the runtime-sized kernels; box 15: the
    column-per-lane and pipelined kernels): the initial planes must be bit-identical, the costs
    agree to the level two implementations of the same fp32 formulas can (see costs_close), a sweep
    started from the reference's own state reproduces its planes, and the free-running result is
    inside BASELINE.json's tolerance (1e-4 relative depth, 1e-3 normals)."""
    from tests.test_oracle_vs_ref import golden_problem, costs_close, rel
    gs, g = golden_problem(fixture)
    with Session(gs) as s:
        s.init_planes()
        n4, c = s.get_state()
        assert_same(n4, g["init_norm4"], "HIP init planes vs reference")
        assert costs_close(g["init_cost"], c)
        s.set_state(g["init_norm4"], g["init_cost"])
        s.sweep(0, abi.BLACK)
        b4, bc = s.get_state()
        same = (bits(b4) == bits(g["black0_norm4"])).all(-1)
        assert same.mean() > 0.999
        assert costs_close(g["black0_cost"][same], bc[same])
    f4, fc = runcuda(gs)
    d_rel = rel(g["final_norm4"][..., 3], f4[..., 3])
    n_err = np.abs(g["final_norm4"][..., :3] - f4[..., :3]).max(-1)
    ok = (d_rel < 1e-4) & (n_err < 1e-3)
    assert ok.mean() >= 0.9995  # measured: 1.0 on both fixtures (every pixel inside the tolerance)
    assert costs_close(g["final_cost"][ok], fc[ok])


def test_production_process_ignores_the_experiment_switches(hip):
    """Without GIPUMA_HIP_EXPERIMENTS the library reads none of its A/B variables: a session created with
    GIPUMA_HIP_GROUP_FROM=-1, ..._PUSH_LAUNCHES=0, ..._COLS_LAUNCHES=0 and ..._TUNE=64 in the environment has the default
    schedule (a regression of the gate -- a variable read with a bare getenv -- would change it), and its solve is the
    oracle's.  With the switch set the same variables do change the schedule."""
    gs, _ = synth.build_problem(synth.tiny_config(cols=1056, rows=512, n_src=2, blocksize=15, iterations=1, n_best=2))
    knobs = {"GIPUMA_HIP_GROUP_FROM": "-1", "GIPUMA_HIP_PUSH_LAUNCHES": "0", "GIPUMA_HIP_COLS_LAUNCHES": "0",
             "GIPUMA_HIP_TUNE": "64"}

    def schedule():
        with Session(gs) as s:
            return s.schedule()

    default = _with_env({"GIPUMA_HIP_EXPERIMENTS": None}, schedule)
    assert default == dict(push_launches=4, group_from=4, group_fused=True, cols_launches=4)
    production = _with_env(dict(knobs, GIPUMA_HIP_EXPERIMENTS=None), schedule)
    assert production == default
    experiment = _with_env(dict(knobs, GIPUMA_HIP_EXPERIMENTS="1"), schedule)
    assert experiment != default and experiment["group_from"] == -1 and experiment["push_launches"] == 0
    a = _with_env(dict(knobs, GIPUMA_HIP_EXPERIMENTS=None), lambda: runcuda(gs))
    o = OracleState(gs).run()
    assert_same(a[0], o[0], "production configuration norm4")
    assert_same(a[1], o[1], "production configuration cost")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "fast", "literal"])
def test_fused_kernel_equals_two_launches_at_headline_size_run_after_run(hip, mode):
    """The guard of a failure class, not of a feature.  Round 6's bounds-checked build (profiles/r06_checked_build.txt) has a
    fused kernel -- the FAST flavour's pm::sweep_group_kernel<15, 1> -- that returns garbage on config C, differently on every
    run, only free-running, and is correct when the same sources are compiled without scalar spills into vector lanes: what
    the compiler generates for ~100 spilled scalars can break a kernel without a line of its source being wrong (round 4 saw
    the same in its fused colour kernel).  Every flavour of the SHIPPED library must therefore show, on the headline frame
    and free-running: the fused launches == plane-keyed costs in a launch of their own (same arithmetic, other code), and
    the same bits from a second run."""
    gs, _ = full_problem("C", iterations=4)
    kw = dict(fast=True) if mode == "fast" else dict(literal=True) if mode == "literal" else {}
    fused = _with_env({"GIPUMA_HIP_GROUP_FUSED": 1}, lambda: runcuda(gs, **kw))
    again = _with_env({"GIPUMA_HIP_GROUP_FUSED": 1}, lambda: runcuda(gs, **kw))
    split = _with_env({"GIPUMA_HIP_GROUP_FUSED": 0}, lambda: runcuda(gs, **kw))
    assert_same(fused[0], again[0], "%s: fused, second run, norm4" % mode)
    assert_same(fused[1], again[1], "%s: fused, second run, cost" % mode)
    assert_same(fused[0], split[0], "%s: fused vs two launches, norm4" % mode)
    assert_same(fused[1], split[1], "%s: fused vs two launches, cost" % mode)


@pytest.mark.gpu
def test_dispatch_orders_do_not_change_a_result(hip):
    """which workgroup does which tile is performance only: the fused launches' order from the previous durations
    (pm::tile_order_kernel), the plain column order of a band (default: the frame's last column right after the first),
    no XCD-aware mapping at all, other band heights -- always the bits of the default mapping, on a frame with more than
    1024 tiles (the plane-keyed fused kernel runs) that is not a multiple of the tile in either direction"""
    gs, _ = synth.build_problem("C", cols=1050, rows=630, iterations=4)
    base = runcuda(gs)
    for env in ({"GIPUMA_HIP_TILE_ORDER": 1}, {"GIPUMA_HIP_TUNE": 1 << 21}, {"GIPUMA_HIP_TUNE": 4},
                {"GIPUMA_HIP_TILE_ORDER": 1, "GIPUMA_HIP_TUNE": (1 << 21) | (3 << 8)}):
        other = _with_env(env, lambda: runcuda(gs))
        assert_same(base[0], other[0], "%s norm4" % env)
        assert_same(base[1], other[1], "%s cost" % env)
