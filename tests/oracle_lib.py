"""ctypes binding of the CPU oracle (oracle/libgipuma_oracle.so).  Test infrastructure: imported
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

from gipuma_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(_ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libgipuma_oracle.so")
_FP = C.POINTER(C.c_float)
_lib = None


def build():
    # one `make` at a time: parallel test workers (pytest-xdist) must not link the same .so concurrently
    import fcntl
    with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libgipuma_oracle.so"], stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _granted_cores():
    """cores this process may really use: the affinity mask cut down by the cgroup's CPU quota (the GPU box shows
    256 hardware threads and grants the time of 16; 256 OpenMP threads on that quota spend it being throttled)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        # always through make (incremental): a library left by an earlier checkout may lack newer symbols
        try:
            build()
        except (OSError, subprocess.CalledProcessError):
            if not os.path.exists(ORACLE_SO):
                raise
        L = C.CDLL(ORACLE_SO)
        if "OMP_NUM_THREADS" not in os.environ:  # (libgomp may be loaded already: set it through the library)
            if hasattr(L, "gipuma_oracle_set_threads"):
                L.gipuma_oracle_set_threads.argtypes = [C.c_int]
                L.gipuma_oracle_set_threads(_granted_cores())
        D = C.POINTER(abi.Desc)
        L.gipuma_oracle_run.argtypes = [D, _FP, _FP, C.c_int]
        L.gipuma_oracle_init_planes.argtypes = [D, _FP, _FP]
        L.gipuma_oracle_sweep.argtypes = [D, _FP, _FP, C.c_int, C.c_int, C.c_uint, C.c_int]
        L.gipuma_oracle_sweep_band.argtypes = [D, _FP, _FP, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int]
        L.gipuma_oracle_finalize.argtypes = [D, _FP, _FP]
        L.gipuma_oracle_eval_cost.argtypes = [D, _FP, _FP]
        L.gipuma_oracle_time.argtypes = [D, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.gipuma_oracle_time_band.argtypes = [D, C.c_int, C.c_int, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double)]
        L.gipuma_oracle_exp.argtypes = [C.c_float]
        L.gipuma_oracle_exp.restype = C.c_float
        L.gipuma_oracle_uniform.argtypes = [C.c_uint32] * 5
        L.gipuma_oracle_uniform.restype = C.c_float
        L.gipuma_oracle_homography.argtypes = [C.POINTER(abi.Camera), C.POINTER(abi.Camera), _FP,
                                               C.c_float, _FP]
        L.gipuma_oracle_sample5.argtypes = [_FP, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _FP]
        L.gipuma_oracle_taps3.argtypes = [_FP, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _FP]
        L.gipuma_oracle_aggregate.argtypes = [_FP, C.c_int, C.c_int, C.c_int, C.c_float]
        L.gipuma_oracle_aggregate.restype = C.c_float
        L.gipuma_oracle_view_cost.argtypes = [D, C.c_int, C.c_int, C.c_int, _FP]
        L.gipuma_oracle_view_cost.restype = C.c_float
        L.gipuma_oracle_multiview_cost.argtypes = [D, C.c_int, C.c_int, _FP]
        L.gipuma_oracle_multiview_cost.restype = C.c_float
        L.gipuma_oracle_push_costs.argtypes = [D, C.c_int, C.c_int, _FP, _FP, C.POINTER(C.c_int)]
        L.gipuma_oracle_depth_from_plane.argtypes = [C.POINTER(abi.Camera), _FP, C.c_int, C.c_int]
        L.gipuma_oracle_depth_from_plane.restype = C.c_float
        L.gipuma_oracle_plane_d.argtypes = [C.POINTER(abi.Camera), _FP, C.c_int, C.c_int, C.c_float]
        L.gipuma_oracle_plane_d.restype = C.c_float
        L.gipuma_oracle_view_vector.argtypes = [C.POINTER(abi.Camera), C.c_int, C.c_int, _FP]
        L.gipuma_oracle_refine_schedule.argtypes = [C.c_float, _FP, _FP, C.c_int]
        L.gipuma_oracle_set_flavour.argtypes = [C.c_int]
        _lib = L
    return _lib


def fptr(a):
    return a.ctypes.data_as(_FP)


def farr(vals):
    return np.ascontiguousarray(vals, dtype=np.float32)


class OracleState:
    """host state planes driven through the oracle, same call shapes as gipuma_amd.Session"""

    def __init__(self, gs):
        self.gs = gs
        self.norm4 = np.zeros((gs.rows, gs.cols, 4), dtype=np.float32)
        self.cost = np.zeros((gs.rows, gs.cols), dtype=np.float32)

    def _d(self):
        return C.byref(self.gs.desc)

    def init_planes(self):
        assert lib().gipuma_oracle_init_planes(self._d(), fptr(self.norm4), fptr(self.cost)) == 0

    def sweep(self, iteration, colour, stages=abi.STAGE_ALL, unfused=False):
        assert lib().gipuma_oracle_sweep(self._d(), fptr(self.norm4), fptr(self.cost), iteration,
                                         colour, stages, int(unfused)) == 0

    def sweep_band(self, iteration, colour, y0, y1, stages=abi.STAGE_ALL):
        assert lib().gipuma_oracle_sweep_band(self._d(), fptr(self.norm4), fptr(self.cost), iteration,
                                              colour, stages, y0, y1) == 0

    def finalize(self):
        assert lib().gipuma_oracle_finalize(self._d(), fptr(self.norm4), fptr(self.cost)) == 0

    def run(self, unfused=False):
        assert lib().gipuma_oracle_run(self._d(), fptr(self.norm4), fptr(self.cost), int(unfused)) == 0
        return self.norm4, self.cost

    def eval_cost(self, planes):
        planes = farr(planes)
        out = np.empty((self.gs.rows, self.gs.cols), dtype=np.float32)
        assert lib().gipuma_oracle_eval_cost(self._d(), fptr(planes), fptr(out)) == 0
        return out
