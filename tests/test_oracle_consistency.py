"""Properties of the oracle itself that the HIP design leans on."""
import os

import numpy as np

from gipuma_amd import abi, synth
from tests.oracle_lib import OracleState


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_fused_stage_order_equals_the_reference_six_launch_schedule(tiny_problem):
    """running close, far, refine pixel by pixel (what the fused HIP kernel does) gives the same
    bits as three full-image passes per colour (the reference's launches, gipuma.cu:1915-1935):
    a pixel of one colour only ever reads pixels of the other colour."""
    gs, _ = tiny_problem
    a = OracleState(gs).run(unfused=False)
    b = OracleState(gs).run(unfused=True)
    assert np.array_equal(bits(a[0]), bits(b[0]))
    assert np.array_equal(bits(a[1]), bits(b[1]))


def test_result_does_not_depend_on_thread_count(tiny_problem):
    import subprocess
    import sys
    gs, _ = tiny_problem
    a = OracleState(gs).run()
    code = ("import sys; sys.path.insert(0, %r); import numpy as np;"
            "from gipuma_amd import synth; from tests.oracle_lib import OracleState;"
            "gs,_=synth.build_problem(synth.tiny_config()); n4,c=OracleState(gs).run();"
            "sys.stdout.buffer.write(n4.tobytes()+c.tobytes())" % os.path.dirname(os.path.dirname(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.check_output([sys.executable, "-c", code], env=env)
    assert out == a[0].tobytes() + a[1].tobytes()


def test_solver_reconstructs_the_analytic_surface():
    gs, info = synth.build_problem(synth.tiny_config(cols=96, rows=72, n_src=4, blocksize=9,
                                                     iterations=4))
    n4, c = OracleState(gs).run()
    rel = np.abs(n4[..., 3] - info["gt_depth"]) / info["gt_depth"]
    assert (rel < 0.01).mean() > 0.95
    nn = np.linalg.norm(n4[..., :3], axis=-1)
    assert np.abs(nn - 1).max() < 1e-4


def test_vectorised_sample_loop_equals_the_scalar_one():
    """the oracle evaluates eight window rows at a time (AVX2; gipuma_oracle_set_simd) with the scalar functions' operations
    in their order: whole solves are bit-identical to the scalar path's -- the default flavour and the model of rounds 1-5,
    gray and colour, windows of 8 rows, fewer (box 11) and more (box 25: 13), a ragged frame"""
    from tests import oracle_lib
    L = oracle_lib.lib()
    L.gipuma_oracle_set_simd.argtypes = [__import__("ctypes").c_int]
    if not L.gipuma_oracle_get_simd():
        return  # (built without AVX2: there is only the scalar path)
    cases = [(synth.tiny_config(), {}),
             (synth.tiny_config(cols=70, rows=50, n_src=3, blocksize=25, iterations=1, n_best=3), {}),
             (synth.tiny_config(cols=75, rows=50, n_src=3, blocksize=11, iterations=2, n_best=2), {}),
             (synth.tiny_config(cols=64, rows=48, n_src=3, blocksize=15, iterations=2, n_best=2), dict(colour=True))]
    try:
        for cfg, kw in cases:
            gs, _ = synth.build_problem(cfg, **kw)
            for fl in (-1, 0):
                L.gipuma_oracle_set_flavour(fl)
                L.gipuma_oracle_set_simd(0)
                a = OracleState(gs).run()
                a = (a[0].copy(), a[1].copy())
                L.gipuma_oracle_set_simd(1)
                b = OracleState(gs).run()
                assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1])), (cfg, kw, fl)
    finally:
        L.gipuma_oracle_set_simd(1)
        L.gipuma_oracle_set_flavour(-1)
