"""GIPUMA_HIP_FLAG_LITERAL (include/gipuma_hip.h): the reference-order flavour of the kernels.  Its results must equal the
reference's OWN device code -- /root/reference/gipuma.cu compiled for the CPU with fp32 texture-filter weights (oracle/_ref,
which travels with the snapshot; its committed outputs tests/golden/ref_*.npz where it does not) -- in EVERY bit of every plane
and every cost, launch by launch and free-running, and the oracle's literal flavour 7 likewise.  Since round 6 the flavour
runs through the same kernels as the default mode (push, column-per-lane, plane-keyed, bounded refinement): the sizes below
are chosen so that each family is reached; the real frame sizes are in tests/test_headline_parity.py."""
import os
import numpy as np
import pytest

from gipuma_amd import abi, synth
from gipuma_amd.problem import Session, runcuda
from tests import oracle_lib, ref_lib
from tests.oracle_lib import OracleState

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("fixture", ["ref_tiny64", "ref_box15"])
def test_literal_mode_reproduces_the_reference_s_golden_dumps(hip, fixture):
    """tests/golden/*.npz: what the reference's own code produced (scripts/make_ref_golden.py) -- initial planes and costs,
    the state after the first black launches, the final maps: all bit-identical"""
    from tests.test_oracle_vs_ref import golden_problem
    gs, g = golden_problem(fixture)
    with Session(gs, literal=True) as s:
        s.init_planes()
        n4, c = s.get_state()
        assert np.array_equal(bits(n4), bits(g["init_norm4"]))
        assert np.array_equal(bits(c), bits(g["init_cost"]))
        s.sweep(0, abi.BLACK)
        b4, bc = s.get_state()
        assert np.array_equal(bits(b4), bits(g["black0_norm4"]))
        assert np.array_equal(bits(bc), bits(g["black0_cost"]))
    f4, fc = runcuda(gs, literal=True)
    assert np.array_equal(bits(f4), bits(g["final_norm4"]))
    assert np.array_equal(bits(fc), bits(g["final_cost"]))


@pytest.mark.parametrize("cfg,over", [("A", dict(cols=128, rows=96)), ("B", dict(cols=160, rows=128)),
                                      ("C", dict(cols=96, rows=64, iterations=3)),
                                      ("C", dict(cols=320, rows=256))])
def test_literal_mode_equals_the_reference_s_own_code_live(hip, cfg, over):
    """free-running solves with the BASELINE configurations' parameters against oracle/_ref run on the box's CPU: every
    plane and every cost bit for bit (config C's parameters on 320x256: 10 views, box 15, 8 iterations, 81 920 pixels)"""
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    gs, _ = synth.build_problem(cfg, **over)
    rn, rc = ref_lib.RefState(gs, tex_mode=0).run()
    n4, c = runcuda(gs, literal=True)
    assert np.array_equal(bits(n4), bits(rn))
    assert np.array_equal(bits(c), bits(rc))


def test_literal_mode_equals_the_oracle_s_literal_flavour(hip):
    """... and the oracle's flavour 7 (gipuma_oracle_set_flavour), on a ragged frame the reference's 32-pixel tiles cannot
    run, launch by launch: the two restatements of the reference's order agree where the reference itself cannot be asked"""
    gs, _ = synth.build_problem(synth.tiny_config(cols=150, rows=100, n_src=4, blocksize=11, iterations=2, n_best=3))
    L = oracle_lib.lib()
    L.gipuma_oracle_set_flavour(7)
    try:
        o = OracleState(gs)
        with Session(gs, literal=True) as s:
            o.init_planes()
            s.init_planes()
            for it in range(2):
                for colour in (abi.BLACK, abi.RED):
                    o.sweep(it, colour)
                    s.sweep(it, colour)
                    n4, c = s.get_state()
                    assert np.array_equal(bits(n4), bits(o.norm4)), (it, colour)
                    assert np.array_equal(bits(c), bits(o.cost)), (it, colour)
            o.finalize()
            s.finalize()
            n4, c = s.get_state()
            assert np.array_equal(bits(n4), bits(o.norm4))
    finally:
        L.gipuma_oracle_set_flavour(-1)


@pytest.mark.parametrize("cfg", [dict(cols=96, rows=64, n_src=3, blocksize=9, iterations=2, n_best=2),
                                 dict(cols=128, rows=96, n_src=4, blocksize=15, iterations=3, n_best=3)])
def test_literal_mode_colour_equals_the_reference_s_float4_instantiation(hip, cfg):
    """-color_processing: the reference's T = float4 kernels (gipuma.cu:1965-1968; tex2D<float4>, the float4 operators of
    vector_operations.h, l1_norm(float4)) run on the CPU against the literal flavour: every plane and cost bit for bit"""
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    gs, _ = synth.build_problem(synth.tiny_config(**cfg), colour=True)
    rn, rc = ref_lib.RefState(gs, tex_mode=0).run()
    n4, c = runcuda(gs, literal=True)
    assert np.array_equal(bits(n4), bits(rn))
    assert np.array_equal(bits(c), bits(rc))


def test_literal_mode_boundaries(hip, tiny_problem):
    """the combination with the fast flag is refused with an error, not run in another mode; an exact session beside a
    literal one is unaffected"""
    import ctypes as C
    gs, _ = tiny_problem
    lib = abi.load_library()
    h = C.c_void_p()
    keep = gs.desc.flags
    try:
        gs.desc.flags = keep | abi.FLAG_FAST | abi.FLAG_LITERAL
        assert lib.gipuma_hip_create(C.byref(gs.desc), C.byref(h)) == abi.ERR_ARG
    finally:
        gs.desc.flags = keep
    on, oc = OracleState(gs).run()
    with Session(gs, literal=True) as s:
        s.solve(timing=False)
        n4, c = runcuda(gs)
    assert np.array_equal(bits(n4), bits(on)) and np.array_equal(bits(c), bits(oc))


@pytest.mark.parametrize("what", ["gray_box15_all_families", "gray_box25", "gray_box19", "gray_box11_small", "colour_box15", "runtime_window"])
def test_literal_taps_one_by_one_path(hip, what):
    """The five taps of a sample normally come from ONE 4x4 window (their separately rounded coordinates land on the centre
    tap's neighbours); a sample for which they do not -- about two in ten million -- fetches every tap by itself
    (pm_sample.h: taps_gather).  variants/libgipuma_hip_gather.so is the same library with -DPM_LITERAL_FORCE_GATHER: every
    8th sample takes that path.  Both must give the reference's bits: the variant's maps and costs equal the shipped
    library's, and the oracle's flavour 7."""
    path = os.path.join(os.path.dirname(abi.LIB_PATH), "variants", "libgipuma_hip_gather.so")
    assert os.path.exists(path), "built by __graft_entry__.build()"
    kw = {}
    if what == "gray_box15_all_families":  # >= 1024 tiles: push, column-per-lane, fused plane-keyed launches, prefilter
        cfg, kw = "C", dict(cols=832, rows=640, iterations=3)
    elif what == "gray_box25":
        cfg, kw = "D", dict(cols=832, rows=640, iterations=2, n_src=6)
    elif what == "gray_box19":  # the reference's default window through its round-6 kernel families
        cfg, kw = "C", dict(cols=832, rows=640, iterations=2, n_src=6, blocksize=19)
    elif what == "gray_box11_small":
        cfg, kw = "B", dict(cols=320, rows=256, iterations=3)
    elif what == "colour_box15":
        cfg, kw = "C", dict(cols=832, rows=640, iterations=2, n_src=5, colour=True)
    else:
        cfg = synth.tiny_config(cols=150, rows=100, n_src=4, blocksize=9, iterations=2, n_best=3)
    gs, _ = synth.build_problem(cfg, **kw)
    n4, c = runcuda(gs, literal=True)
    forced = abi.load_library(path)
    keep = abi._lib
    abi._lib = forced
    try:
        g4, gc = runcuda(gs, literal=True)
    finally:
        abi._lib = keep
    assert np.array_equal(bits(g4), bits(n4)) and np.array_equal(bits(gc), bits(c))
    if gs.rows * gs.cols <= 320 * 256:
        L = oracle_lib.lib()
        L.gipuma_oracle_set_flavour(7)
        try:
            on, oc = OracleState(gs).run()
        finally:
            L.gipuma_oracle_set_flavour(-1)
        assert np.array_equal(bits(n4), bits(on)) and np.array_equal(bits(c), bits(oc))
