"""Synthetic multi-view scenes for tests and bench.py.

The reference ships no images and there is no network, so the workloads of
BASELINE.json are built from an analytic textured surface rendered through real DTU projection
matrices (gipuma_amd/data/dtu_calib_r5000.json, made by scripts/make_dtu_calib_fixture.py) or
through a synthetic Middlebury-like camera ring.  Images are quantised to 8 bit and stored as
float32 0..255, like the reference's imread + convertTo(CV_32F) (main.cpp:739-751, :941).

Everything is a deterministic function of (config, scene_seed); torch is used only as an array
library so the 1600x1200 cases render on the GPU in bench.py and on the CPU in tests.
"""
import json
import math
import os

import numpy as np
import torch

from . import abi
from .cameras import get_camera_parameters, select_views
from .problem import AlgorithmParameters, GlobalState

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def dtu_projection_matrices():
    with open(os.path.join(_DATA, "dtu_calib_r5000.json")) as f:
        P = json.load(f)["P"]
    return {int(k): np.array(v, dtype=np.float64) for k, v in P.items()}


def ring_projection_matrices(n, f, cx, cy, radius, height, target_dist):
    """Middlebury-like ring: n pinhole cameras on a circle of `radius` at `height` above the
    object plane, all looking at the origin from `target_dist` away (units: metres)."""
    Ps = []
    K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1.0]])
    for k in range(n):
        ang = 2.0 * math.pi * k / n
        Cc = np.array([radius * math.cos(ang), radius * math.sin(ang), height])
        Cc = Cc / np.linalg.norm(Cc) * target_dist
        z = -Cc / np.linalg.norm(Cc)                 # optical axis towards the origin
        up = np.array([0.0, 0.0, 1.0])
        x = np.cross(z, up)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])                      # world -> camera
        t = -R @ Cc
        Ps.append(K @ np.concatenate([R, t[:, None]], axis=1))
    return Ps


# ---------------------------------------------------------------------------------------------
# analytic surface z = h(x, y) in REFERENCE-camera coordinates, and its albedo
# ---------------------------------------------------------------------------------------------
class Surface:
    def __init__(self, z0, amp, wavelength, tilt=(0.05, -0.03), pixel_footprint=0.2, seed=1234):
        self.z0, self.amp, self.wl = float(z0), float(amp), float(wavelength)
        self.tilt = tilt
        self.fp = float(pixel_footprint)
        self.seed = int(seed)

    def h(self, x, y):
        k = 2.0 * math.pi / self.wl
        z = self.z0 + self.tilt[0] * x + self.tilt[1] * y
        z = z + self.amp * torch.sin(k * x + 0.3) * torch.cos(0.8 * k * y - 0.5)
        return z

    def grad(self, x, y):
        k = 2.0 * math.pi / self.wl
        hx = self.tilt[0] + self.amp * k * torch.cos(k * x + 0.3) * torch.cos(0.8 * k * y - 0.5)
        hy = self.tilt[1] - self.amp * 0.8 * k * torch.sin(k * x + 0.3) * torch.sin(0.8 * k * y - 0.5)
        return hx, hy

    def albedo(self, x, y):
        """band-limited texture in [0,1]: octaves of smooth value noise, finest cell 3 px"""
        tex = torch.zeros_like(x)
        amp_sum = 0.0
        for o, a in enumerate([1.0, 0.9, 0.8, 0.6, 0.5]):
            cell = self.fp * 3.0 * (2.0 ** o)
            tex = tex + a * _value_noise(x / cell, y / cell, self.seed + 101 * o)
            amp_sum += a
        return tex / amp_sum


class SteppedSurface(Surface):
    """the smooth surface plus depth DISCONTINUITIES: a chequer of +-`step` plateaus and a raised
    disc (an occluder in front of the rest).  A stress scene for the skip rules and the early
    termination, which both feed on agreement between neighbours."""

    def __init__(self, *a, step=30.0, period=(300.0, 260.0), disc=(40.0, -30.0, 80.0, 90.0), **k):
        super().__init__(*a, **k)
        self.step, self.period, self.disc = float(step), period, disc

    def h(self, x, y):
        z = super().h(x, y)
        sx = torch.sign(torch.sin(2.0 * math.pi * x / self.period[0]))
        sy = (torch.sin(2.0 * math.pi * y / self.period[1]) > 0).to(z.dtype)
        z = z + self.step * sx * sy
        cx, cy, rad, lift = self.disc
        inside = ((x - cx) ** 2 + (y - cy) ** 2) < rad * rad
        return torch.where(inside, z - lift, z)


class PatchySurface(Surface):
    """the smooth surface with the texture taken away where real scans lose it: about 30 % of the surface -- smooth
    blobs -- carries a FLAT albedo (after the sensor noise of +-1 grey level every window sample there looks like
    every other, so patch costs tie and bounds hold less often), and a diagonal band carries a PERIODIC stripe
    texture (period about 6 pixels: many planes match equally well).  The geometry is the smooth surface's."""

    def __init__(self, *a, flat_fraction=0.30, band=(0.25, 40.0), **k):
        super().__init__(*a, **k)
        self.flat_fraction = float(flat_fraction)
        self.band = band  # (centre offset as a fraction of the wavelength, half-width in scene units)

    def albedo(self, x, y):
        tex = super().albedo(x, y)
        # blobs: low-frequency noise thresholded so that `flat_fraction` of the area falls below the level
        cell = self.wl * 0.18  # blobs of about a hundred pixels
        blob = 0.6 * _value_noise(x / cell, y / cell, self.seed + 7001) + \
            0.4 * _value_noise(x / (0.5 * cell), y / (0.5 * cell), self.seed + 7002)
        # (value noise is a smoothed uniform field around 0.5: this level cuts off about the wanted share;
        #  build_problem reports the share that was really rendered)
        level = 0.5 + (self.flat_fraction - 0.5) * 0.62
        flat = blob < level
        tex = torch.where(flat, torch.full_like(tex, 0.45), tex)
        # stripes across a diagonal band, period 6 pixel footprints, perpendicular to the band
        d = (x + 0.6 * y) / math.sqrt(1.36) - self.band[0] * self.wl
        stripes = 0.5 + 0.35 * torch.sin(2.0 * math.pi * (x - 0.6 * y) / math.sqrt(1.36) / (6.0 * self.fp))
        return torch.where(torch.abs(d) < self.band[1], stripes, tex)

    def flat_mask(self, x, y):
        cell = self.wl * 0.18  # blobs of about a hundred pixels
        blob = 0.6 * _value_noise(x / cell, y / cell, self.seed + 7001) + \
            0.4 * _value_noise(x / (0.5 * cell), y / (0.5 * cell), self.seed + 7002)
        d = (x + 0.6 * y) / math.sqrt(1.36) - self.band[0] * self.wl
        return (blob < 0.5 + (self.flat_fraction - 0.5) * 0.62) & ~(torch.abs(d) < self.band[1])


def render_march(surface, K, R, t, rows, cols, device="cpu", zspan=(380.0, 760.0), noise_sigma=0.0,
                 noise_seed=0):
    """render() for surfaces with discontinuities: first hit of each ray with z = h(x, y) by
    marching through the depth span in reference coordinates, then bisection (occlusions are
    rendered as such).  Optional additive sensor noise before the 8-bit quantisation."""
    dt = torch.float64
    K = torch.as_tensor(K, dtype=dt, device=device)
    R = torch.as_tensor(R, dtype=dt, device=device)
    t = torch.as_tensor(t, dtype=dt, device=device)
    v, u = torch.meshgrid(torch.arange(rows, dtype=dt, device=device),
                          torch.arange(cols, dtype=dt, device=device), indexing="ij")
    pix = torch.stack([u, v, torch.ones_like(u)], dim=-1)
    dirs = pix @ torch.linalg.inv(K).T @ R
    o = -(R.T @ t)

    def g(lam):
        X = o + lam[..., None] * dirs
        return X[..., 2] - surface.h(X[..., 0], X[..., 1])

    # the ray parameter at which the ray crosses the planes z = zspan[0], zspan[1]
    lam0 = (zspan[0] - o[2]) / dirs[..., 2]
    lam1 = (zspan[1] - o[2]) / dirs[..., 2]
    n_steps = 96
    lo = lam0.clone()
    hi = lam1.clone()
    found = torch.zeros_like(lam0, dtype=torch.bool)
    prev = lam0
    for k in range(1, n_steps + 1):
        cur = lam0 + (lam1 - lam0) * (k / n_steps)
        hit = (g(cur) >= 0) & ~found          # z >= h: the ray is behind the surface now
        lo = torch.where(hit, prev, lo)
        hi = torch.where(hit, cur, hi)
        found |= hit
        prev = cur
    for _ in range(20):
        mid = 0.5 * (lo + hi)
        behind = g(mid) >= 0
        hi = torch.where(behind, mid, hi)
        lo = torch.where(behind, lo, mid)
    lam = hi
    X = o + lam[..., None] * dirs
    depth = (X @ R.T + t)[..., 2]
    img = 20.0 + 215.0 * surface.albedo(X[..., 0], X[..., 1])
    if noise_sigma > 0:
        gen = torch.Generator(device="cpu").manual_seed(int(noise_seed))
        img = img + noise_sigma * torch.randn(img.shape, generator=gen, dtype=dt).to(img.device)
    img = torch.clamp(torch.round(img), 0, 255)
    return img.to(torch.float32), depth.to(torch.float32)


def _hash01(ix, iy, seed):
    h = (ix * 374761393 + iy * 668265263 + seed * 2147483647) & 0xFFFFFFFF
    h = ((h ^ (h >> 13)) * 1274126177) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    return (h & 0xFFFFFF).to(torch.float64) / float(1 << 24)


def _value_noise(u, v, seed):
    iu, iv = torch.floor(u), torch.floor(v)
    tu, tv = u - iu, v - iv
    su, sv = tu * tu * (3 - 2 * tu), tv * tv * (3 - 2 * tv)
    iu, iv = iu.to(torch.int64), iv.to(torch.int64)
    n00 = _hash01(iu, iv, seed)
    n10 = _hash01(iu + 1, iv, seed)
    n01 = _hash01(iu, iv + 1, seed)
    n11 = _hash01(iu + 1, iv + 1, seed)
    return (n00 * (1 - su) + n10 * su) * (1 - sv) + (n01 * (1 - su) + n11 * su) * sv


def render(surface, K, R, t, rows, cols, device="cpu", colour=False):
    """Image (float32 0..255, 8-bit quantised) and depth of the surface seen by the camera
    K [R | t] (pose relative to the reference camera).  colour: (rows, cols, 4) = B, G, R and an
    alpha channel the path must ignore (main.cpp:943-956 leaves it unset)."""
    dt = torch.float64
    K = torch.as_tensor(K, dtype=dt, device=device)
    R = torch.as_tensor(R, dtype=dt, device=device)
    t = torch.as_tensor(t, dtype=dt, device=device)
    v, u = torch.meshgrid(torch.arange(rows, dtype=dt, device=device),
                          torch.arange(cols, dtype=dt, device=device), indexing="ij")
    pix = torch.stack([u, v, torch.ones_like(u)], dim=-1)
    dirs = pix @ torch.linalg.inv(K).T @ R             # R^T K^-1 p  (row-vector form)
    o = -(R.T @ t)
    lam = (surface.z0 - o[2]) / dirs[..., 2]
    for _ in range(12):                                  # Newton on g(lam) = z - h(x, y)
        X = o + lam[..., None] * dirs
        g = X[..., 2] - surface.h(X[..., 0], X[..., 1])
        hx, hy = surface.grad(X[..., 0], X[..., 1])
        gp = dirs[..., 2] - (hx * dirs[..., 0] + hy * dirs[..., 1])
        lam = lam - g / gp
    X = o + lam[..., None] * dirs
    depth = (X @ R.T + t)[..., 2]                        # depth in this camera
    if not colour:
        img = torch.clamp(torch.round(20.0 + 215.0 * surface.albedo(X[..., 0], X[..., 1])), 0, 255)
        return img.to(torch.float32), depth.to(torch.float32)
    chans = []
    seed0 = surface.seed
    for c in range(3):                                   # three differently seeded albedos, correlated
        surface.seed = seed0 + 7919 * c
        a = 0.6 * surface.albedo(X[..., 0], X[..., 1])
        surface.seed = seed0
        a = a + 0.4 * surface.albedo(X[..., 0], X[..., 1])
        chans.append(torch.clamp(torch.round(20.0 + 215.0 * a), 0, 255))
    chans.append(torch.full_like(chans[0], 77.0))        # alpha: present, never read
    return torch.stack(chans, dim=-1).to(torch.float32), depth.to(torch.float32)


# ---------------------------------------------------------------------------------------------
# the BASELINE.json configurations
# ---------------------------------------------------------------------------------------------
# DTU reference views with >= 20 neighbours inside the 10-30 degree cone, and their neighbours
# (computed with select_views on the fixture; kept explicit so the workload never depends on a
# shuffle -- the reference would srand(time(0)), main.cpp:491-496)
DTU_REF_VIEWS = [14, 15, 16, 17, 23, 24, 25, 26]

CONFIGS = {
    # name: (kind, cols, rows, n_src, blocksize, iterations, n_best, depth_min, depth_max)
    "A": dict(kind="ring", cols=320, rows=240, n_src=2, blocksize=11, iterations=4, n_best=2,
              depth_min=0.3, depth_max=0.8, min_angle=5.0, max_angle=45.0),
    "B": dict(kind="ring", cols=640, rows=480, n_src=6, blocksize=11, iterations=8, n_best=3,
              depth_min=0.3, depth_max=0.8, min_angle=5.0, max_angle=45.0),
    "C": dict(kind="dtu", cols=1600, rows=1200, n_src=10, blocksize=15, iterations=8, n_best=3,
              depth_min=300.0, depth_max=800.0, min_angle=10.0, max_angle=30.0),
    "D": dict(kind="dtu", cols=1600, rows=1200, n_src=20, blocksize=25, iterations=8, n_best=3,
              depth_min=300.0, depth_max=800.0, min_angle=10.0, max_angle=30.0),
}


def tiny_config(cols=64, rows=48, n_src=3, blocksize=7, iterations=2, n_best=2):
    """a DTU-geometry problem small enough for CPU-side checking in unit tests: the DTU
    cameras with the image plane scaled down (--cam_scale, cameraGeometryUtils.h:136-147)"""
    return dict(kind="dtu", cols=cols, rows=rows, n_src=n_src, blocksize=blocksize,
                iterations=iterations, n_best=n_best, depth_min=300.0, depth_max=800.0,
                min_angle=10.0, max_angle=30.0, cam_scale=1600.0 / cols)


def build_problem(cfg, ref_view=15, scene_seed=1234, solver_seed=1, device="cpu",
                  keep_on_device=False, gamma=10.0, cost_comb=abi.COMB_BEST_N, colour=False, scene="smooth",
                  **overrides):
    """Returns (GlobalState, info) for one reference view of a configuration.

    info: dict with 'gt_depth' (reference-view depth of the analytic surface, numpy), the chosen
    source view ids and the camera set."""
    if isinstance(cfg, str):
        cfg = dict(CONFIGS[cfg])
    cfg = dict(cfg)
    cfg.update(overrides)
    rows, cols, n_src = cfg["rows"], cfg["cols"], cfg["n_src"]
    cam_scale = cfg.get("cam_scale", 1.0)
    if cfg["kind"] == "dtu":
        allP = dtu_projection_matrices()
        order = [ref_view] + [k for k in sorted(allP) if k != ref_view]
        cs_all = get_camera_parameters([allP[k] for k in order], cam_scale=cam_scale)
        cand, _, _ = select_views(cs_all, cols, rows, cfg["min_angle"], cfg["max_angle"],
                                  max_views=10 ** 6)
        if len(cand) < n_src:
            raise ValueError("reference view %d has only %d neighbours in the cone" %
                             (ref_view, len(cand)))
        # spread the chosen sources over the candidates (deterministic)
        pick = [cand[(i * len(cand)) // n_src] for i in range(n_src)]
        ids = [ref_view] + [order[i] for i in pick]
        Ps = [allP[k] for k in ids]
        z0, amp, wl = 600.0, 25.0, 160.0
        tilt = (0.05, -0.03)
    else:
        # Middlebury-like ring (units: metres).  f scales with the image width so A and B see
        # the same object; 16 cameras on the ring, the reference is camera `ref_view % 16`.
        f = 1520.0 * cols / 640.0
        ring = ring_projection_matrices(16, f, cols / 2.0 - 0.5, rows / 2.0 - 0.5,
                                        radius=0.38, height=0.40, target_dist=0.55)
        r0 = ref_view % 16
        # nearest neighbours on the ring, alternating sides
        offs = []
        k = 1
        while len(offs) < n_src:
            offs.append(k)
            if len(offs) < n_src:
                offs.append(-k)
            k += 1
        ids = [r0] + [(r0 + o) % 16 for o in offs]
        Ps = [ring[k] for k in ids]
        z0, amp, wl = 0.55, 0.02, 0.12
        tilt = (0.04, -0.02)
    cs = get_camera_parameters(Ps, cam_scale=cam_scale)
    footprint = z0 / cs.f
    if colour and scene != "smooth":
        # (the stepped and patchy scenes are rendered by the marching renderer, which makes gray planes only: handing them to
        #  a colour session would describe (rows, cols) planes as (rows, cols, 4) -- out-of-bounds reads on the device)
        raise ValueError("scene %r is gray only; colour problems use the smooth scene" % scene)
    if scene == "steps":
        # depth steps + an occluding disc + sensor noise (gray only), in units of the scene depth
        u = z0 / 600.0
        surf = SteppedSurface(z0, amp, wl, tilt=tilt, pixel_footprint=footprint, seed=scene_seed, step=30.0 * u,
                              period=(300.0 * u, 260.0 * u), disc=(40.0 * u, -30.0 * u, 80.0 * u, 90.0 * u))
    elif scene == "patchy":
        # 30 % flat albedo + a periodic band, sensor noise sigma 1 (gray only)
        u = z0 / 600.0
        surf = PatchySurface(z0, amp, wl, tilt=tilt, pixel_footprint=footprint, seed=scene_seed, band=(0.25, 40.0 * u))
    else:
        surf = Surface(z0, amp, wl, tilt=tilt, pixel_footprint=footprint, seed=scene_seed)
    imgs, gt = [], None
    for i in range(cs.n):
        if scene == "patchy":
            img, depth = render_march(surf, cs.K[i], cs.R[i], cs.t[i], rows, cols, device=device,
                                      zspan=(z0 * 0.63, z0 * 1.27), noise_sigma=1.0, noise_seed=scene_seed + i)
        elif scene == "steps":
            img, depth = render_march(surf, cs.K[i], cs.R[i], cs.t[i], rows, cols, device=device,
                                      zspan=(z0 * 0.63, z0 * 1.27), noise_sigma=2.0, noise_seed=scene_seed + i)
        else:
            img, depth = render(surf, cs.K[i], cs.R[i], cs.t[i], rows, cols, device=device, colour=colour)
        if i == 0:
            gt = depth.cpu().numpy()
        imgs.append(img)
    ap = AlgorithmParameters(iterations=cfg["iterations"], n_best=cfg["n_best"],
                             depthMin=cfg["depth_min"], depthMax=cfg["depth_max"],
                             min_angle=cfg["min_angle"], max_angle=cfg["max_angle"],
                             max_views=n_src + 1, gamma=gamma, cost_comb=cost_comb)
    ap.set_blocksize(cfg["blocksize"])
    # the reference's own view selection must keep all n_src views
    subset, dmin, dmax = select_views(cs, cols, rows, ap.min_angle, ap.max_angle, ap.max_views,
                                      ap.depthMin, ap.depthMax)
    ap.depthMin, ap.depthMax = dmin, dmax
    if keep_on_device:
        imgs = [im.contiguous() for im in imgs]
        # The frames were rendered on torch's current stream; the library works on a stream of its own (non-blocking: it does
        # not wait for the default stream).  Hand them over COMPLETE: a session created while a frame is still being written
        # checks and packs half-written planes -- its 8-bit test can then fail, and the session runs the float kernels, with
        # the same results at an eighth of the speed (found in round 6: the second device-resident problem of a process).
        if imgs and imgs[0].is_cuda:
            torch.cuda.synchronize(imgs[0].device)
        gs = GlobalState(imgs, cs, subset, ap, seed=solver_seed,
                         device_ptrs=[im.data_ptr() for im in imgs], rows=rows, cols=cols,
                         channels=4 if colour else 1)
    else:
        gs = GlobalState([im.cpu().numpy() for im in imgs], cs, subset, ap, seed=solver_seed)
    info = dict(gt_depth=gt, view_ids=ids, cameras=cs, surface=surf, cfg=cfg, P_matrices=Ps, cam_scale=cam_scale)
    return gs, info
