"""The drop-in boundary: `int runcuda(GlobalState&)` with the reference's signature, compiled by
gipuma_amd/csrc/adapter/build_adapter.sh against the reference's OWN headers (globalstate.h,
camera.h, ...) through the HIP compat include dir.  The library only exists where the reference
tree was present at build time (this container); it travels to the GPU box as a built .so."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from gipuma_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "gipuma_amd", "csrc", "adapter", "libgipuma_runcuda.so")
have = pytest.mark.skipif(not os.path.exists(SO), reason="adapter not built (no reference tree at build time)")


@have
def test_adapter_exports_the_reference_symbol():
    syms = subprocess.check_output(["nm", "-D", "--defined-only", SO]).decode()
    assert "_Z7runcudaR11GlobalState" in syms          # int runcuda(GlobalState&), gipuma.h:2
    assert "gipuma_adapter_selftest" in syms
    needed = subprocess.check_output(["readelf", "-d", SO]).decode()
    assert "libgipuma_hip.so" in needed and "oracle" not in needed


@have
@pytest.mark.gpu
@pytest.mark.parametrize("colour", [False, True])
def test_runcuda_through_the_reference_structs_matches_the_oracle(hip, colour):
    """fill the reference's real GlobalState like main.cpp does, upload images through the
    cudaMallocArray / cudaCreateTextureObject call sequence -- gray (main.cpp:607-656) or float4
    colour texels with color_processing set (main.cpp:560-605 -> gipuma<float4>, gipuma.cu:1965-1968)
    -- call runcuda(), read gs.lines back"""
    from tests.oracle_lib import OracleState
    lib = C.CDLL(SO, mode=C.RTLD_GLOBAL)
    lib.gipuma_adapter_selftest.argtypes = [C.POINTER(abi.Desc), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    gs, _ = synth.build_problem(synth.tiny_config(cols=96, rows=64, n_src=3, blocksize=11, n_best=2), colour=colour)
    n4 = np.zeros((gs.rows, gs.cols, 4), dtype=np.float32)
    c = np.zeros((gs.rows, gs.cols), dtype=np.float32)
    rc = lib.gipuma_adapter_selftest(C.byref(gs.desc), n4.ctypes.data_as(C.POINTER(C.c_float)),
                                     c.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0                                      # runcuda returns 0 unconditionally
    o_n4, o_c = OracleState(gs).run()
    assert np.array_equal(n4.view(np.uint32), o_n4.view(np.uint32))
    assert np.array_equal(c.view(np.uint32), o_c.view(np.uint32))
