"""ctypes binding of oracle/_ref/libgipuma_ref.so: the reference's OWN device functions
(/root/reference/gipuma.cu) compiled for the CPU by oracle/ref_shim/build_ref.sh.  Test
infrastructure; only exists where the reference tree was present at build time."""
import ctypes as C
import os

import numpy as np

from gipuma_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(_ROOT, "oracle", "_ref", "libgipuma_ref.so")
_FP = C.POINTER(C.c_float)
_lib = None


def available():
    return os.path.exists(REF_SO)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(REF_SO)
        L.ref_create.argtypes = [C.POINTER(abi.Desc)]
        L.ref_sweep.argtypes = [C.c_int, C.c_int, C.c_uint]
        L.ref_get_state.argtypes = [_FP, _FP]
        L.ref_set_state.argtypes = [_FP, _FP]
        L.ref_set_tex_mode.argtypes = [C.c_int]
        L.ref_homography.argtypes = [C.c_int, _FP, C.c_float, _FP]
        L.ref_depth_from_plane.argtypes = [_FP, C.c_int, C.c_int]
        L.ref_depth_from_plane.restype = C.c_float
        L.ref_plane_d.argtypes = [_FP, C.c_int, C.c_int, C.c_float]
        L.ref_plane_d.restype = C.c_float
        L.ref_view_vector.argtypes = [C.c_int, C.c_int, _FP]
        # the blocks of a launch run under OpenMP: as many threads as the box GRANTS (its cgroup quota), not as many as it
        # shows -- the GPU box shows 256 hardware threads and grants the time of 16
        if "OMP_NUM_THREADS" not in os.environ and hasattr(L, "ref_set_threads"):
            from tests.oracle_lib import _granted_cores
            L.ref_set_threads.argtypes = [C.c_int]
            L.ref_set_threads(_granted_cores())
        _lib = L
    return _lib


class RefState:
    """the reference's GlobalState driven launch by launch (same call shapes as OracleState)"""

    def __init__(self, gs, tex_mode=0):
        self.gs = gs
        assert lib().ref_create(C.byref(gs.desc)) == 0, "ref_create (needs rows, cols multiples of 32)"
        lib().ref_set_tex_mode(tex_mode)

    def init_planes(self):
        assert lib().ref_init_planes() == 0

    def sweep(self, iteration, colour, stages=abi.STAGE_ALL):
        assert lib().ref_sweep(iteration, colour, stages) == 0

    def finalize(self):
        assert lib().ref_finalize() == 0

    def initial_cost(self):
        assert lib().ref_initial_cost() == 0

    def get_state(self):
        n4 = np.empty((self.gs.rows, self.gs.cols, 4), dtype=np.float32)
        c = np.empty((self.gs.rows, self.gs.cols), dtype=np.float32)
        lib().ref_get_state(n4.ctypes.data_as(_FP), c.ctypes.data_as(_FP))
        return n4, c

    def set_state(self, n4, c):
        n4 = np.ascontiguousarray(n4, dtype=np.float32)
        c = np.ascontiguousarray(c, dtype=np.float32)
        lib().ref_set_state(n4.ctypes.data_as(_FP), c.ctypes.data_as(_FP))

    def run(self):
        self.init_planes()
        for it in range(self.gs.params.iterations):
            self.sweep(it, abi.BLACK)
            self.sweep(it, abi.RED)
        self.finalize()
        return self.get_state()
