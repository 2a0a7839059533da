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


REF_MAIN = "/root/reference/main.cpp"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree is only present in the build container")
def test_reference_main_cpp_compiles_unchanged_against_the_compat_layer():
    """north_star: "drops into main.cpp unchanged".  The reference's OWN main.cpp -- its includes
    (main.cpp:16-23: cuda_runtime.h, cuda.h, cuda_runtime_api.h, cuda_texture_types.h, vector_types.h,
    helper_cuda.h), its texture upload (:501-690: cudaMallocArray / cudaMemcpyToArray / cudaCreateTextureObject
    on cudaResourceDesc / cudaTextureDesc), cudaSetDevice / cudaMemGetInfo / cudaDeviceReset, the managed
    GlobalState and the call runcuda(*gs) at :973 -- goes through the compiler's front end with only
    adapter/cuda_compat/ in place of the CUDA toolkit.  OpenCV, which the image does not have, is stood in for
    by declarations under tests/stubs/ (test infrastructure, never linked); every other header is the
    reference's.  Any CUDA identifier the compat layer failed to provide would be an error here."""
    here = os.path.join(ROOT, "gipuma_amd", "csrc", "adapter", "cuda_compat")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-fsyntax-only", "-std=c++17", "-w", "-ferror-limit=50",
           "-D__HIP_PLATFORM_AMD__", "-x", "hip", "--offload-arch=gfx950", "--cuda-host-only",
           "-include", os.path.join(here, "gipuma_cuda_compat.h"), "-I" + here,
           "-I" + os.path.join(ROOT, "tests", "stubs"), "-I/root/reference", REF_MAIN]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the stub really is declarations only, and the adapter does not see it
    stub = open(os.path.join(ROOT, "tests", "stubs", "opencv2", "opencv_stub.hpp")).read()
    assert "TEST INFRASTRUCTURE" in stub
    assert "stubs" not in open(os.path.join(ROOT, "gipuma_amd", "csrc", "adapter", "build_adapter.sh")).read()


@have
@pytest.mark.gpu
@pytest.mark.parametrize("colour", [False, True])
def test_runcuda_in_the_reference_order_mode_equals_the_reference_s_own_code(hip, colour):
    """the same call through the reference's structs with GIPUMA_LITERAL=1 in the environment (the adapter's way to select
    GIPUMA_HIP_FLAG_LITERAL: GlobalState has no field for it): gs.lines holds, bit for bit, what the reference's OWN device
    code (oracle/_ref, on the CPU, fp32 filter weights) leaves there -- the reference's signature, the reference's structs,
    the reference's results"""
    from tests import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    lib = C.CDLL(SO, mode=C.RTLD_GLOBAL)
    lib.gipuma_adapter_selftest.argtypes = [C.POINTER(abi.Desc), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    gs, _ = synth.build_problem(synth.tiny_config(cols=96, rows=64, n_src=3, blocksize=11, n_best=2), colour=colour)
    n4 = np.zeros((gs.rows, gs.cols, 4), dtype=np.float32)
    c = np.zeros((gs.rows, gs.cols), dtype=np.float32)
    os.environ["GIPUMA_LITERAL"] = "1"
    try:
        rc = lib.gipuma_adapter_selftest(C.byref(gs.desc), n4.ctypes.data_as(C.POINTER(C.c_float)),
                                         c.ctypes.data_as(C.POINTER(C.c_float)))
    finally:
        del os.environ["GIPUMA_LITERAL"]
    assert rc == 0
    rn, rc_ = ref_lib.RefState(gs, tex_mode=0).run()
    assert np.array_equal(n4.view(np.uint32), rn.view(np.uint32))
    assert np.array_equal(c.view(np.uint32), rc_.view(np.uint32))
