"""Host-side assembly of one reference view's problem: the Python mirror of what runGipuma()
puts into GlobalState before it calls runcuda() (reference main.cpp:829-968), and the
runcuda()-shaped entry point on top of the C-ABI.

Names follow the reference: AlgorithmParameters (algorithmparameters.h:19-85), GlobalState
(globalstate.h:24-45), runcuda (gipuma.h:2).
"""
import ctypes as C

import numpy as np

from . import abi
from .cameras import disparity_range


class AlgorithmParameters:
    """Defaults of reference algorithmparameters.h:21-51 (only the fields the path reads, plus the
    host-side view-selection knobs)."""

    def __init__(self, **kw):
        self.box_hsize = 19
        self.box_vsize = 19
        self.tau_color = 10.0
        self.tau_gradient = 2.0
        self.alpha = 0.9
        self.gamma = 10.0
        self.iterations = 8
        self.good_factor = 1.5
        self.n_best = 2
        self.cost_comb = abi.COMB_BEST_N
        self.depthMin = -1.0
        self.depthMax = -1.0
        self.min_angle = 5.0
        self.max_angle = 45.0
        self.max_views = 9
        self.min_disparity = 0.0
        self.max_disparity = 256.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown AlgorithmParameters field %r" % k)
            setattr(self, k, v)

    def set_blocksize(self, b):
        """--blocksize= must be odd and sets both sizes (main.cpp:269-276)."""
        if b % 2 == 0:
            raise ValueError("blocksize must be odd")
        self.box_hsize = self.box_vsize = b

    def to_c(self):
        p = abi.Params()
        p.box_hsize, p.box_vsize = self.box_hsize, self.box_vsize
        p.iterations, p.n_best, p.cost_comb = self.iterations, self.n_best, self.cost_comb
        p.alpha, p.tau_color, p.tau_gradient = self.alpha, self.tau_color, self.tau_gradient
        p.gamma, p.good_factor = self.gamma, self.good_factor
        p.min_disparity, p.max_disparity = self.min_disparity, self.max_disparity
        return p


class GlobalState:
    """Everything runcuda() reads: cameras, selected views, parameters and the images.

    images: list of 2-D float32 arrays (0..255, main.cpp:941), index 0 = reference view; or, with
    ``device_ptrs``, raw device addresses of such planes already resident in HBM.
    """

    def __init__(self, images, camera_set, selected, params, seed=1, device_ptrs=None,
                 rows=None, cols=None, pitch=None, device_id=0, stream=None, flags=0, channels=1):
        self.params = params
        self.cameras = camera_set
        self.selected = list(selected)
        if len(self.selected) > abi.MAX_VIEWS:
            raise ValueError("at most %d selected views (gipuma.cu:736)" % abi.MAX_VIEWS)
        # gray: (rows, cols) planes; -color_processing: (rows, cols, 4) = B, G, R, unused alpha
        # (main.cpp:943-956), pitch counted in floats
        if device_ptrs is None:
            self.images = [np.ascontiguousarray(im, dtype=np.float32) for im in images]
            self.rows, self.cols = self.images[0].shape[:2]
            self.channels = 4 if self.images[0].ndim == 3 else 1
            if self.channels == 4 and self.images[0].shape[2] != 4:
                raise ValueError("colour images must be (rows, cols, 4)")
            self.pitch = self.cols * self.channels
            ptrs = [im.ctypes.data for im in self.images]
        else:
            self.images = images  # keep whatever owns the device memory alive
            self.channels = channels
            self.rows, self.cols, self.pitch = rows, cols, pitch or cols * channels
            ptrs = list(device_ptrs)
            flags |= abi.FLAG_IMAGES_ON_DEVICE
        n = len(ptrs)
        if n != camera_set.n:
            raise ValueError("need one camera per image")
        # depth range -> camera 0 and the disparity range (main.cpp:898-906)
        cam0 = camera_set.c_array[0]
        cam0.depth_min, cam0.depth_max = params.depthMin, params.depthMax
        params.min_disparity, params.max_disparity = disparity_range(
            camera_set.f, cam0.baseline, params.depthMin, params.depthMax)
        self._img_ptrs = (C.c_void_p * n)(*ptrs)
        self._sel = (C.c_int32 * max(1, len(self.selected)))(*self.selected)
        d = abi.Desc()
        d.abi_version = abi.ABI_VERSION
        d.rows, d.cols, d.channels, d.pitch = self.rows, self.cols, self.channels, self.pitch
        d.n_images = n
        d.images = C.cast(self._img_ptrs, C.POINTER(C.c_void_p))
        d.cameras = C.cast(camera_set.c_array, C.POINTER(abi.Camera))
        d.n_selected = len(self.selected)
        d.selected = C.cast(self._sel, C.POINTER(C.c_int32))
        d.params = params.to_c()
        d.seed = seed
        d.device_id = device_id
        d.stream = stream
        d.flags = flags
        self.desc = d

    @property
    def n_pixels(self):
        return self.rows * self.cols


SNAPSHOT_VERSION = 2  # 2: parameters as JSON text (1: repr() text, no longer read)


def save_problem(path, gs, P_matrices, cam_scale=1.0):
    """Snapshot of one reference view's problem (host images, projection matrices, parameters) as an
    .npz, so that another process can rebuild the identical GlobalState without rendering anything
    (bench.py's CPU-baseline workers)."""
    imgs = [np.asarray(im.cpu().numpy() if hasattr(im, "cpu") else im, dtype=np.float32) for im in gs.images]
    # the image planes go to their own .npy so that readers can map them (many workers, one copy in
    # the page cache)
    np.save(path + ".images.npy", np.stack(imgs))
    np.savez(path, P=np.stack([np.asarray(p, dtype=np.float64) for p in P_matrices]),
             cam_scale=float(cam_scale), selected=np.asarray(gs.selected, dtype=np.int32),
             seed=int(gs.desc.seed), params=np.array([_params_to_json(gs.params)]),
             format_version=np.int32(SNAPSHOT_VERSION))


def _params_to_json(ap):
    """AlgorithmParameters as JSON text; non-finite floats as the strings "inf" / "-inf" / "nan" (plain JSON has
    no spelling for them)"""
    import json
    import math
    out = {}
    for k, v in sorted(ap.__dict__.items()):
        if isinstance(v, (bool, np.bool_)):
            out[k] = bool(v)
        elif isinstance(v, (int, np.integer)):
            out[k] = int(v)
        elif isinstance(v, (float, np.floating)):
            v = float(v)
            out[k] = v if math.isfinite(v) else repr(v)
        elif isinstance(v, np.ndarray):
            out[k] = v.tolist()
        elif isinstance(v, (tuple, list)):
            out[k] = [x.item() if isinstance(x, np.generic) else x for x in v]
        else:
            out[k] = v
    return json.dumps(out)


def _params_from_json(text):
    import json
    out = {}
    for k, v in json.loads(text).items():
        if isinstance(v, str) and v in ("inf", "-inf", "nan"):
            v = float(v)
        out[k] = v
    return out


def load_problem(path):
    from .cameras import get_camera_parameters
    z = np.load(path, allow_pickle=False)
    version = int(z["format_version"]) if "format_version" in z.files else 1
    if version != SNAPSHOT_VERSION:
        raise ValueError("%s is a problem snapshot of format %d, this build reads format %d: re-save it with "
                         "gipuma_amd.problem.save_problem" % (path, version, SNAPSHOT_VERSION))
    cs = get_camera_parameters([p for p in z["P"]], cam_scale=float(z["cam_scale"]))
    ap = AlgorithmParameters()
    for k, v in _params_from_json(str(z["params"][0])).items():
        setattr(ap, k, v)
    stack = np.load(path + ".images.npy", mmap_mode="r")
    gs = GlobalState([stack[i] for i in range(stack.shape[0])], cs, [int(v) for v in z["selected"]], ap,
                     seed=int(z["seed"]))
    gs._mapped = stack  # keep the mapping alive
    return gs


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class _mode:
    """`with _mode(gs, fast, literal)`: the descriptor carries GIPUMA_HIP_FLAG_FAST / _LITERAL for the calls inside (the
    library reads the flags when it creates the session)"""

    def __init__(self, gs, fast, literal=False):
        self.gs, self.fast, self.literal = gs, fast, literal

    def __enter__(self):
        self.keep = self.gs.desc.flags
        self.gs.desc.flags = self.keep | (abi.FLAG_FAST if self.fast else 0) | (abi.FLAG_LITERAL if self.literal else 0)

    def __exit__(self, *a):
        self.gs.desc.flags = self.keep


def runcuda(gs, timing=False, fast=False, literal=False):
    """The reference's ``int runcuda(GlobalState&)`` (gipuma.h:2) on the HIP path.

    Returns (norm4, cost[, Timing]): norm4[y, x] = (n_world.xyz, depth), cost[y, x], as the
    reference leaves them in gs.lines (gipuma.cu:1080-1103, main.cpp:976-985).  `fast`: the tolerance-judged
    flavour of the kernels (GIPUMA_HIP_FLAG_FAST) instead of the bit-exact one; `literal`: the reference-order flavour
    (GIPUMA_HIP_FLAG_LITERAL: bit-identical to the reference's own code, slow)."""
    lib = abi.load_library()
    norm4 = np.empty((gs.rows, gs.cols, 4), dtype=np.float32)
    cost = np.empty((gs.rows, gs.cols), dtype=np.float32)
    t = abi.Timing()
    with _mode(gs, fast, literal):
        rc = lib.gipuma_hip_run(C.byref(gs.desc), _fptr(norm4), _fptr(cost), C.byref(t))
    abi.check(lib, rc, "gipuma_hip_run")
    return (norm4, cost, t) if timing else (norm4, cost)


class Session:
    """One reference view resident on the GPU: the launches of gipuma<T>() (gipuma.cu:1825-1960)
    one call at a time."""

    def __init__(self, gs, fast=False, literal=False):
        self.lib = abi.load_library()
        self.gs = gs
        self.fast, self.literal = fast, literal
        self.h = C.c_void_p()
        with _mode(gs, fast, literal):
            rc = self.lib.gipuma_hip_create(C.byref(gs.desc), C.byref(self.h))
        abi.check(self.lib, rc, "gipuma_hip_create")

    def close(self):
        if self.h:
            self.lib.gipuma_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def init_planes(self):
        abi.check(self.lib, self.lib.gipuma_hip_init_planes(self.h), "gipuma_hip_init_planes")

    def sweep(self, iteration, colour, stages=abi.STAGE_ALL):
        abi.check(self.lib, self.lib.gipuma_hip_sweep(self.h, iteration, colour, stages),
                  "gipuma_hip_sweep")

    def finalize(self):
        abi.check(self.lib, self.lib.gipuma_hip_finalize(self.h), "gipuma_hip_finalize")

    def sync(self):
        """wait for the session's stream (gipuma_hip_get_state with no destination copies nothing)"""
        abi.check(self.lib, self.lib.gipuma_hip_get_state(self.h, None, None), "gipuma_hip_get_state")

    def solve(self, timing=True):
        t = abi.Timing()
        abi.check(self.lib, self.lib.gipuma_hip_solve(self.h, C.byref(t) if timing else None),
                  "gipuma_hip_solve")
        return t

    def launch_times(self):
        """(ms per half-sweep of the last timed solve, number of leading half-sweeps that include a
        pm::push_kernel launch)"""
        n, npush = C.c_int(0), C.c_int(0)
        cap = 2 * max(1, int(self.gs.params.iterations))
        buf = (C.c_float * cap)()
        abi.check(self.lib, self.lib.gipuma_hip_launch_times(self.h, buf, cap, C.byref(n), C.byref(npush)),
                  "gipuma_hip_launch_times")
        return [float(buf[i]) for i in range(min(cap, n.value))], npush.value

    def schedule(self):
        """dict(push_launches, group_from, group_fused, cols_launches): which kernels a solve launches per half-sweep"""
        info = (C.c_int * 4)()
        abi.check(self.lib, self.lib.gipuma_hip_schedule(self.h, info), "gipuma_hip_schedule")
        return dict(push_launches=info[0], group_from=info[1], group_fused=bool(info[2]), cols_launches=info[3])

    def group_times(self):
        """ms of the pm::group_kernel launch of every half-sweep of the last timed solve (0 where it had none)"""
        n = C.c_int(0)
        cap = 2 * max(1, int(self.gs.params.iterations))
        buf = (C.c_float * cap)()
        abi.check(self.lib, self.lib.gipuma_hip_group_times(self.h, buf, cap, C.byref(n)), "gipuma_hip_group_times")
        return [float(buf[i]) for i in range(min(cap, n.value))]

    def eval_cost(self, planes):
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        out = np.empty((self.gs.rows, self.gs.cols), dtype=np.float32)
        abi.check(self.lib, self.lib.gipuma_hip_eval_cost(self.h, _fptr(planes), _fptr(out)),
                  "gipuma_hip_eval_cost")
        return out

    def get_state(self):
        norm4 = np.empty((self.gs.rows, self.gs.cols, 4), dtype=np.float32)
        cost = np.empty((self.gs.rows, self.gs.cols), dtype=np.float32)
        abi.check(self.lib, self.lib.gipuma_hip_get_state(self.h, _fptr(norm4), _fptr(cost)),
                  "gipuma_hip_get_state")
        return norm4, cost

    def set_state(self, norm4, cost):
        norm4 = np.ascontiguousarray(norm4, dtype=np.float32)
        cost = np.ascontiguousarray(cost, dtype=np.float32)
        abi.check(self.lib, self.lib.gipuma_hip_set_state(self.h, _fptr(norm4), _fptr(cost)),
                  "gipuma_hip_set_state")
