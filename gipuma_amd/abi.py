"""ctypes mirror of include/gipuma_hip.h and the loader of the HIP library.

The HIP library (gipuma_amd/csrc/libgipuma_hip.so, built by ``__graft_entry__.build()``) is the
only compute path of this package.  There is no CPU fallback: if the library is missing or no
MI355X is visible, calls raise ``GipumaHipError``.
"""
import ctypes as C
import os

ABI_VERSION = 1
MAX_VIEWS = 32
MAXCOST = 1000.0
ERR_ARG, ERR_DEVICE, ERR_NO_DEVICE, ERR_UNSUPPORTED = -1, -2, -3, -4  # gipuma_hip_status

COMB_ALL, COMB_BEST_N, COMB_ANGLE, COMB_GOOD = 0, 1, 2, 3
STAGE_CLOSE, STAGE_FAR, STAGE_REFINE, STAGE_ALL = 1, 2, 4, 7
BLACK, RED = 0, 1
FLAG_IMAGES_ON_DEVICE = 1
FLAG_UNFUSED = 2
FLAG_CACHE_IMAGES = 4
FLAG_FAST = 8  # tolerance-judged flavour of the kernels (include/gipuma_hip.h); default: bit-exact
FLAG_LITERAL = 16  # reference-order flavour: bit-identical to the reference's own code (fp32 filter model); slow

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgipuma_hip.so")


class GipumaHipError(RuntimeError):
    pass


class Camera(C.Structure):
    """gipuma_hip_camera == the used part of Camera_cu (reference camera.h:7-62)."""
    _fields_ = [
        ("K", C.c_float * 9), ("K_inv", C.c_float * 9), ("R", C.c_float * 9),
        ("t", C.c_float * 3), ("M_inv", C.c_float * 9), ("P_col34", C.c_float * 3),
        ("C", C.c_float * 3), ("R_orig_inv", C.c_float * 9),
        ("fx", C.c_float), ("fy", C.c_float), ("f", C.c_float), ("alpha", C.c_float),
        ("baseline", C.c_float), ("depth_min", C.c_float), ("depth_max", C.c_float),
    ]


class Params(C.Structure):
    """gipuma_hip_params == the device-read part of AlgorithmParameters
    (reference algorithmparameters.h:52-84)."""
    _fields_ = [
        ("box_hsize", C.c_int32), ("box_vsize", C.c_int32), ("iterations", C.c_int32),
        ("n_best", C.c_int32), ("cost_comb", C.c_int32),
        ("alpha", C.c_float), ("tau_color", C.c_float), ("tau_gradient", C.c_float),
        ("gamma", C.c_float), ("min_disparity", C.c_float), ("max_disparity", C.c_float),
        ("good_factor", C.c_float),
    ]


class Desc(C.Structure):
    """gipuma_hip_desc == what runcuda() reads out of GlobalState (reference globalstate.h:24-45)."""
    _fields_ = [
        ("abi_version", C.c_uint32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("channels", C.c_int32), ("pitch", C.c_int32), ("n_images", C.c_int32),
        ("images", C.POINTER(C.c_void_p)), ("cameras", C.POINTER(Camera)),
        ("n_selected", C.c_int32), ("selected", C.POINTER(C.c_int32)),
        ("params", Params), ("seed", C.c_uint32), ("device_id", C.c_int32),
        ("stream", C.c_void_p), ("flags", C.c_uint32),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("ms_init", C.c_float), ("ms_sweeps", C.c_float), ("ms_finalize", C.c_float),
        ("ms_total", C.c_float), ("n_sweep_launches", C.c_int32), ("ms_sweep_avg", C.c_float),
    ]


# every symbol include/gipuma_hip.h declares: (name, restype, argtypes)
_FP = C.POINTER(C.c_float)
SYMBOLS = [
    ("gipuma_hip_version", C.c_int, []),
    ("gipuma_hip_last_error", C.c_char_p, []),
    ("gipuma_hip_device_count", C.c_int, []),
    ("gipuma_hip_cache_clear", C.c_int, []),
    ("gipuma_hip_selftest_reciprocal", C.c_int, [C.c_int, C.POINTER(C.c_ulonglong)]),
    ("gipuma_hip_selftest_quotient", C.c_int, [C.c_int, C.c_uint, C.c_uint, C.POINTER(C.c_ulonglong)]),
    ("gipuma_hip_create", C.c_int, [C.POINTER(Desc), C.POINTER(C.c_void_p)]),
    ("gipuma_hip_destroy", C.c_int, [C.c_void_p]),
    ("gipuma_hip_init_planes", C.c_int, [C.c_void_p]),
    ("gipuma_hip_sweep", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint]),
    ("gipuma_hip_finalize", C.c_int, [C.c_void_p]),
    ("gipuma_hip_eval_cost", C.c_int, [C.c_void_p, _FP, _FP]),
    ("gipuma_hip_get_state", C.c_int, [C.c_void_p, _FP, _FP]),
    ("gipuma_hip_set_state", C.c_int, [C.c_void_p, _FP, _FP]),
    ("gipuma_hip_state_device_ptrs", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("gipuma_hip_solve", C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    ("gipuma_hip_launch_times", C.c_int, [C.c_void_p, _FP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("gipuma_hip_group_times", C.c_int, [C.c_void_p, _FP, C.c_int, C.POINTER(C.c_int)]),
    ("gipuma_hip_schedule", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("gipuma_hip_run", C.c_int, [C.POINTER(Desc), _FP, _FP, C.POINTER(Timing)]),
]

_lib = None


def load_library(path=None):
    """dlopen the HIP library and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    exp = os.environ.get("GIPUMA_HIP_EXPERIMENTS", "0") not in ("", "0")
    p = path or (os.environ.get("GIPUMA_HIP_LIB") if exp else None) or LIB_PATH  # A/B: a differently built library
    if not os.path.exists(p):
        raise GipumaHipError(
            "HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
            "g.build()'`); gipuma_amd has no CPU fallback" % p)
    try:
        lib = C.CDLL(p)
    except OSError as e:  # e.g. libamdhip64 not loadable
        raise GipumaHipError("cannot load %s: %s" % (p, e))
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.gipuma_hip_version() != ABI_VERSION:
        raise GipumaHipError("ABI mismatch: library %d, bindings %d"
                             % (lib.gipuma_hip_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def check(lib, rc, what):
    if rc != 0:
        msg = lib.gipuma_hip_last_error()
        raise GipumaHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
