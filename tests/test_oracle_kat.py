"""Known-answer tests that pin the CPU oracle to the reference's formulas.

The reference ships no tests or golden vectors for this path, so each check below is derived from
the cited reference lines and has an answer that can be worked out independently of the oracle
(identity warps, analytic projections in float64, hand-made cost vectors...)."""
import ctypes as C
import math
import struct

import numpy as np
import pytest

from gipuma_amd import abi, synth
from gipuma_amd.cameras import (decompose_projection, get_camera_parameters, select_views)
from gipuma_amd.problem import AlgorithmParameters, GlobalState
from tests.oracle_lib import OracleState, farr, fptr, lib


def f32(x):
    return float(np.float32(x))


# ------------------------------------------------------------------------------------------ K1
def test_k1_identity_view_costs_zero():
    """source image == reference image, R = I, t = 0  =>  H = I for every plane, every sample
    difference is 0 and the view cost is exactly 0 (gipuma.cu:348-352, 267-274).  K is chosen with
    an exactly representable inverse so that K*K^-1 == I in fp32 too."""
    rows, cols = 48, 64
    K = np.array([[128.0, 0, 32], [0, 128.0, 24], [0, 0, 1]])
    P0 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    cs = get_camera_parameters([P0, P0])
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(rows, cols)).astype(np.float32)
    ap = AlgorithmParameters(depthMin=10.0, depthMax=200.0, n_best=1)
    ap.set_blocksize(9)
    gs = GlobalState([img, img], cs, [1], ap)
    H = np.zeros(9, dtype=np.float32)
    lib().gipuma_oracle_homography(C.byref(cs.c_array[0]), C.byref(cs.c_array[1]),
                                   fptr(farr([0.1, -0.2, -0.97])), 61.0, fptr(H))
    assert np.array_equal(H.reshape(3, 3), np.eye(3, dtype=np.float32))
    pl = farr([0.1, -0.2, -0.97, 61.0])
    for (x, y) in [(5, 5), (32, 24), (63, 47), (0, 0)]:
        assert lib().gipuma_oracle_view_cost(C.byref(gs.desc), 1, x, y, fptr(pl)) == 0.0
    # and with DTU intrinsics, where K*K^-1 is only I to rounding, the cost is ~0
    gsd, _ = synth.build_problem(synth.tiny_config(n_src=2, iterations=1))
    cam1, ref = gsd.cameras.c_array[1], gsd.cameras.c_array[0]
    for k in range(9):
        cam1.K[k] = ref.K[k]
        cam1.R[k] = 1.0 if k in (0, 4, 8) else 0.0
    for k in range(3):
        cam1.t[k] = 0.0
    gs2 = GlobalState([gsd.images[0], gsd.images[0], gsd.images[2]], gsd.cameras, gsd.selected,
                      gsd.params)
    c = lib().gipuma_oracle_view_cost(C.byref(gs2.desc), 1, 32, 24, fptr(farr([0.1, -0.2, -0.97, 610.0])))
    assert 0.0 <= c < 1e-3


# ------------------------------------------------------------------------------------------ K2
def test_k2_plane_depth_round_trip():
    """getDepthFromPlane3(getD(n, p, z)) == z (gipuma.cu:96-111, 694-705).  DTU K has a skew of
    -2.5e-4 that the closed form ignores, so the round trip is exact only to ~1e-6 relative."""
    P = synth.dtu_projection_matrices()
    cs = get_camera_parameters([P[1], P[2]])
    cam = cs.c_array[0]
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = rng.normal(size=3)
        n[2] = -abs(n[2]) - 0.3
        n = farr(n / np.linalg.norm(n))
        x, y = int(rng.integers(0, 1600)), int(rng.integers(0, 1200))
        z = f32(rng.uniform(300, 800))
        d = lib().gipuma_oracle_plane_d(C.byref(cam), fptr(n), x, y, z)
        pl = farr([n[0], n[1], n[2], d])
        z2 = lib().gipuma_oracle_depth_from_plane(C.byref(cam), fptr(pl), x, y)
        assert abs(z2 - z) / z < 5e-6
    nan_plane = farr([0, 0, -1, float("nan")])
    assert lib().gipuma_oracle_depth_from_plane(C.byref(cam), fptr(nan_plane), 3, 4) == 1000.0


# ------------------------------------------------------------------------------------------ K3
def test_k3_homography_is_plane_induced_projection():
    """H*(q,1) equals projecting the ray/plane intersection into the source camera, computed
    independently in float64 from the decomposed DTU cameras (gipuma.cu:339-356)."""
    P = synth.dtu_projection_matrices()
    cs = get_camera_parameters([P[15], P[16], P[24]])
    ref = cs.c_array[0]
    rng = np.random.default_rng(2)
    for view in (1, 2):
        to = cs.c_array[view]
        K0, Kv = cs.K[0], cs.K[view]
        R, t = cs.R[view], cs.t[view]
        for _ in range(50):
            n = rng.normal(size=3)
            n[2] = -abs(n[2]) - 0.5
            n /= np.linalg.norm(n)
            qx, qy = rng.uniform(100, 1500), rng.uniform(100, 1100)
            z = rng.uniform(400, 750)
            X0 = z * np.linalg.inv(K0) @ np.array([qx, qy, 1.0])     # point on the plane
            d = -n @ X0
            H = np.zeros(9, dtype=np.float32)
            lib().gipuma_oracle_homography(C.byref(ref), C.byref(to), fptr(farr(n)), f32(d), fptr(H))
            H = H.reshape(3, 3).astype(np.float64)
            for dq in [(0, 0), (7, -5), (-6, 6)]:
                q = np.array([qx + dq[0], qy + dq[1], 1.0])
                ray = np.linalg.inv(K0) @ q
                lam = -d / (n @ ray)                                     # n.(lam ray) + d = 0
                Xs = Kv @ (R @ (lam * ray) + t)
                want = Xs[:2] / Xs[2]
                got = H @ q
                got = got[:2] / got[2]
                assert np.abs(got - want).max() < 0.05                  # fp32 H, coordinates ~1e3


# ------------------------------------------------------------------------------------------ K4
def test_k4_fronto_parallel_translation_is_a_pure_shift():
    """t = (-B,0,0), n = (0,0,-1), depth Z  =>  x' = x - fx*B/Z.  With the source image equal to the
    reference shifted by that integer, the true plane costs 0 and a wrong depth costs > 0."""
    rows, cols, fx = 40, 64, 100.0
    K = np.array([[fx, 0, 31.5], [0, fx, 19.5], [0, 0, 1.0]])
    Z, shift = 50.0, 4
    B = shift * Z / fx
    P0 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = K @ np.hstack([np.eye(3), np.array([[-B], [0], [0]])])
    cs = get_camera_parameters([P0, P1])
    rng = np.random.default_rng(3)
    ref = rng.integers(0, 256, size=(rows, cols + 16)).astype(np.float32)
    img0 = ref[:, 8:8 + cols].copy()
    img1 = ref[:, 8 + shift:8 + shift + cols].copy()      # I1(x - shift) = I0(x)
    ap = AlgorithmParameters(depthMin=10.0, depthMax=200.0, n_best=1, iterations=1)
    ap.set_blocksize(7)
    gs = GlobalState([img0, img1], cs, [1], ap)
    n = farr([0, 0, -1])
    x, y = 30, 20
    d_true = lib().gipuma_oracle_plane_d(C.byref(cs.c_array[0]), fptr(n), x, y, f32(Z))
    c_true = lib().gipuma_oracle_multiview_cost(C.byref(gs.desc), x, y, fptr(farr([0, 0, -1, d_true])))
    assert c_true == 0.0
    d_bad = lib().gipuma_oracle_plane_d(C.byref(cs.c_array[0]), fptr(n), x, y, f32(Z * 1.25))
    c_bad = lib().gipuma_oracle_multiview_cost(C.byref(gs.desc), x, y, fptr(farr([0, 0, -1, d_bad])))
    assert c_bad > 1.0


# ------------------------------------------------------------------------------------------ K5
def agg(costs, comb, n_best=2, good=1.5):
    v = farr(costs)
    return lib().gipuma_oracle_aggregate(fptr(v), len(costs), comb, n_best, f32(good))


def test_k5_cost_combination():
    """gipuma.cu:769-805 on hand-made vectors"""
    nan = float("nan")
    assert agg([3, 1, 2], abi.COMB_ALL) == 2.0
    assert agg([3, 1, 2], abi.COMB_BEST_N, 2) == 1.5
    assert agg([3, 1, 2], abi.COMB_BEST_N, 5) == 2.0                    # numBest = min(valid, n)
    assert agg([3, 1, 2], abi.COMB_ANGLE) == 2.0                        # behaves like ALL
    assert agg([4, 1, 10], abi.COMB_GOOD, good=1.5) == f32((1 + 1.5 + 1.5) / 3)
    assert agg([nan, 1, 3], abi.COMB_BEST_N, 3) == 2.0                  # NaN view is invalid
    assert agg([2000, 1, 3], abi.COMB_ALL) == 2.0                       # >= MAXCOST is invalid
    assert agg([nan, nan], abi.COMB_BEST_N, 2) == abi.MAXCOST           # nothing valid
    assert agg([], abi.COMB_BEST_N, 2) == abi.MAXCOST                   # no views at all
    assert agg([nan, 5], abi.COMB_GOOD, good=1.5) == f32((5 + 7.5) / 2) # GOOD averages all N


# ------------------------------------------------------------------------------------------ K6
@pytest.mark.parametrize("box,S", [(15, 64), (25, 169), (11, 36), (7, 16)])
def test_k6_window_sampling(box, S):
    """offsets -r..r step 2 in both directions (gipuma.cu:28, 633-634): with a source image that is
    a constant 5 grey levels above a constant reference, every sample contributes
    w*(1-alpha)*5 with w = 1, so the view cost counts the samples."""
    rows, cols = 64, 64
    K = np.array([[100.0, 0, 32], [0, 100.0, 32], [0, 0, 1]])
    P0 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    cs = get_camera_parameters([P0, P0])
    ap = AlgorithmParameters(depthMin=10.0, depthMax=200.0, alpha=0.5, n_best=1)
    ap.set_blocksize(box)
    img0 = np.full((rows, cols), 100, dtype=np.float32)
    img1 = np.full((rows, cols), 105, dtype=np.float32)
    gs = GlobalState([img0, img1], cs, [1], ap)
    c = lib().gipuma_oracle_view_cost(C.byref(gs.desc), 1, 32, 32, fptr(farr([0, 0, -1, 50.0])))
    assert c == pytest.approx(S * 0.5 * 5.0, rel=1e-6)


# ------------------------------------------------------------------------------------------ K7
def test_k7_checkerboard_colours():
    """black <=> (x+y) even (gipuma.cu:1730-1734); a black sweep must not change any red pixel,
    and distance-1 / distance-5 neighbours always have the other colour."""
    gs, _ = synth.build_problem(synth.tiny_config(iterations=1))
    o = OracleState(gs)
    o.init_planes()
    before = o.norm4.copy()
    o.sweep(0, abi.BLACK)
    ys, xs = np.mgrid[0:gs.rows, 0:gs.cols]
    red = ((xs + ys) & 1) == 1
    assert np.array_equal(before[red], o.norm4[red])
    assert not np.array_equal(before[~red], o.norm4[~red])
    for dist in (1, 5):
        assert ((xs + ys) & 1 != ((xs + dist + ys) & 1)).all()
        assert ((xs + ys) & 1 != ((xs + ys + dist) & 1)).all()


# ------------------------------------------------------------------------------------------ K8
def test_k8_refinement_schedule():
    """max_disparity 5.206 => deltaZ {2.603, 0.2603, 0.02603}, deltaN {1, 1/4, 1/16}
    (gipuma.cu:958-959, 992)"""
    dz = np.zeros(8, dtype=np.float32)
    dn = np.zeros(8, dtype=np.float32)
    k = lib().gipuma_oracle_refine_schedule(f32(5.206), fptr(dz), fptr(dn), 8)
    assert k == 3
    assert np.allclose(dz[:3], [2.603, 0.2603, 0.02603], rtol=1e-6)
    assert list(dn[:3]) == [1.0, 0.25, 0.0625]
    assert lib().gipuma_oracle_refine_schedule(f32(2737.0), fptr(dz), fptr(dn), 8) == 6   # config B


# ------------------------------------------------------------------------------------------ K9
def test_k9_bilinear_and_clamp_addressing():
    """tex2D(x+0.5, y+0.5) with linear filtering = bilinear blend of floor/floor+1 (config.h:245-248,
    main.cpp:644-648); out-of-range texels clamp to the edge."""
    rows, cols = 6, 8
    img = (np.arange(rows * cols, dtype=np.float32).reshape(rows, cols) * 3 + 1)
    out = np.zeros(5, dtype=np.float32)

    def s5(x, y):
        lib().gipuma_oracle_sample5(fptr(img), rows, cols, cols, f32(x), f32(y), fptr(out))
        return out.copy()

    o = s5(3, 2)                               # integer position: exact texels
    assert list(o) == [img[2, 3], img[2, 4], img[2, 2], img[3, 3], img[1, 3]]
    o = s5(3.25, 2.5)                          # img is affine in (x,y): bilinear is exact
    assert o[0] == pytest.approx(1 + 3 * (2.5 * cols + 3.25), rel=1e-6)
    assert o[1] - o[2] == pytest.approx(6.0, rel=1e-5)        # d/dx over 2 texels
    assert o[3] - o[4] == pytest.approx(6.0 * cols, rel=1e-5)
    assert s5(-3.7, 2)[0] == img[2, 0]         # clamp left
    assert s5(100.2, 2)[0] == img[2, cols - 1]
    assert s5(3, -9)[0] == img[0, 3]
    assert s5(3, 50)[0] == img[rows - 1, 3]
    assert s5(0, 0)[2] == img[0, 0] and s5(0, 0)[4] == img[0, 0]   # -1 taps replicate the border
    assert s5(1e20, 1e20)[0] == img[rows - 1, cols - 1]
    assert math.isnan(s5(float("nan"), 1)[0])


def test_k9b_model_taps_equal_the_five_taps():
    """The cost functions read the taps through gipuma_oracle_taps3 (model M1: columns interpolated along y first, the
    +-1 differences taken on the texels before the interpolation).  Same quantities as the five separate fetches of
    gipuma.cu:251-253: identical on integer positions and on an image that is affine in (x, y); within rounding
    (a few ulp of 255) anywhere; clamp addressing and NaN behaviour unchanged."""
    rows, cols = 9, 11
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(rows, cols)).astype(np.float32)
    o5, o3 = np.zeros(5, dtype=np.float32), np.zeros(3, dtype=np.float32)

    def both(im, x, y):
        lib().gipuma_oracle_sample5(fptr(im), rows, cols, cols, f32(x), f32(y), fptr(o5))
        lib().gipuma_oracle_taps3(fptr(im), rows, cols, cols, f32(x), f32(y), fptr(o3))
        return o5.copy(), o3.copy()

    for x, y in [(3, 2), (0, 0), (10, 8), (-5, 3), (4, 40)]:   # integer positions incl. clamped ones: exact texels
        a, b = both(img, x, y)
        assert list(b) == [a[0], a[1] - a[2], a[3] - a[4]]
    aff = (np.arange(rows * cols, dtype=np.float32).reshape(rows, cols) * 3 + 1)
    a, b = both(aff, 3.25, 2.5)
    assert b[0] == pytest.approx(1 + 3 * (2.5 * cols + 3.25), rel=1e-6)
    assert b[1] == pytest.approx(6.0, rel=1e-6) and b[2] == pytest.approx(6.0 * cols, rel=1e-6)
    for _ in range(500):
        x, y = rng.uniform(-3, cols + 2), rng.uniform(-3, rows + 2)
        a, b = both(img, x, y)
        assert abs(b[0] - a[0]) <= 1e-4 and abs(b[1] - (a[1] - a[2])) <= 2e-4 and abs(b[2] - (a[3] - a[4])) <= 2e-4
    assert math.isnan(both(img, float("nan"), 1)[1][0])


# ------------------------------------------------------------------------------------------ K10
def test_k10_initial_normals_face_the_camera(tiny_problem):
    gs, _ = tiny_problem
    o = OracleState(gs)
    o.init_planes()
    cam = gs.cameras.c_array[0]
    v = np.zeros(3, dtype=np.float32)
    for (x, y) in [(0, 0), (10, 40), (63, 47), (31, 7)]:
        lib().gipuma_oracle_view_vector(C.byref(cam), x, y, fptr(v))
        assert float(np.dot(o.norm4[y, x, :3], v)) <= 0.0
        assert np.linalg.norm(o.norm4[y, x, :3]) == pytest.approx(1.0, abs=1e-5)
    # depth of every initial plane lies in [depth_min, depth_max]
    o2 = o.norm4.copy()
    lib().gipuma_oracle_finalize(C.byref(gs.desc), fptr(o2), fptr(o.cost))
    d = o2[..., 3][o.cost != abi.MAXCOST]
    assert d.min() >= gs.params.depthMin * (1 - 1e-4) and d.max() <= gs.params.depthMax * (1 + 1e-4)


# ------------------------------------------------------------------------------------------ K12
def test_k12_dtu_fixture_decomposition():
    """rect_001_3_r5000.png.P -> fx 2892.33, fy 2883.18, cx 823.21, cy 619.07, |C| 192.4 mm"""
    P = synth.dtu_projection_matrices()[1]
    K, R, Cc = decompose_projection(P)
    assert K[0, 0] == pytest.approx(2892.33, abs=0.01)
    assert K[1, 1] == pytest.approx(2883.18, abs=0.01)
    assert K[0, 2] == pytest.approx(823.21, abs=0.01)
    assert K[1, 2] == pytest.approx(619.07, abs=0.01)
    assert np.linalg.norm(Cc) == pytest.approx(192.4, abs=0.1)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and np.linalg.det(R) == pytest.approx(1.0)
    assert np.allclose(K @ np.hstack([R, (-R @ Cc)[:, None]]), P, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------ model
def test_exp_model_accuracy():
    """M2: exp stand-in within 2 ulp of the true exponential over the range weight_cu uses"""
    xs = -np.linspace(0, 30, 20001)
    got = np.array([lib().gipuma_oracle_exp(f32(x)) for x in xs], dtype=np.float64)
    want = np.exp(xs.astype(np.float32).astype(np.float64))
    ulp = np.spacing(want.astype(np.float32)).astype(np.float64)
    assert (np.abs(got - want) / ulp).max() < 2.0
    assert lib().gipuma_oracle_exp(0.0) == 1.0
    assert lib().gipuma_oracle_exp(f32(-100.0)) == 0.0
    assert lib().gipuma_oracle_exp(float("nan")) == 0.0


def test_rng_model_is_uniform_and_keyed():
    """M4: (0,1], roughly uniform, and every key component matters"""
    u = np.array([lib().gipuma_oracle_uniform(1, 0, x, y, 0) for x in range(64) for y in range(64)])
    assert u.min() > 0.0 and u.max() <= 1.0
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    base = lib().gipuma_oracle_uniform(1, 2, 3, 4, 5)
    assert base == lib().gipuma_oracle_uniform(1, 2, 3, 4, 5)
    for key in [(9, 2, 3, 4, 5), (1, 9, 3, 4, 5), (1, 2, 9, 4, 5), (1, 2, 3, 9, 5), (1, 2, 3, 4, 9),
                (1, 2, 4, 3, 5)]:
        assert lib().gipuma_oracle_uniform(*key) != base


def test_view_selection_matches_survey():
    """selectViews (main.cpp:430-499) on the DTU fixture with the 10-30 degree cone of
    scripts/dtu_fast.sh: reference view 15 has 25 candidates, 24 has 31, 1 has 8."""
    P = synth.dtu_projection_matrices()
    for ref, want in [(15, 25), (24, 31), (1, 8)]:
        order = [ref] + [k for k in sorted(P) if k != ref]
        cs = get_camera_parameters([P[k] for k in order])
        sub, dmin, dmax = select_views(cs, 1600, 1200, 10.0, 30.0, max_views=100)
        assert len(sub) == want
        # reference camera is K[I|0] after re-centring (cameraGeometryUtils.h:268-298)
        assert np.allclose(cs.R[0], np.eye(3), atol=1e-9) and np.allclose(cs.t[0], 0, atol=1e-6)


@pytest.mark.parametrize("box,n_best,comb", [(15, 3, abi.COMB_BEST_N), (11, 2, abi.COMB_BEST_N), (25, 3, abi.COMB_BEST_N),
                                             (7, 1, abi.COMB_ALL), (9, 2, abi.COMB_GOOD)])
def test_push_formulation_is_exact_on_the_cpu(box, n_best, comb):
    """What gipuma_amd/csrc/pm_push.h builds on, checked without a GPU: dis_v(q, plane) does not depend on
    the pixel whose window q belongs to (gipuma.cu:207-274), so the plane of a pixel can be evaluated ONCE per
    view on the stencil that the windows of its eight checkerboard consumers share (n +- 1, n +- 5 in x and
    y, gipuma.cu:1437-1462) -- 2 x (N+5) x N points instead of 8 x N x N -- and every consumer's cost follows
    from the reference's summation over its own window (gipuma.cu:633-676).  gipuma_oracle_push_costs restates
    that data flow; it must give the bits of pmCostMultiview_cu (gipuma_oracle_multiview_cost) at every consumer,
    for planes of every kind, at image borders, for every combiner."""
    gs, _ = synth.build_problem(synth.tiny_config(cols=72, rows=56, n_src=4, blocksize=box, iterations=1,
                                                  n_best=n_best), cost_comb=comb)
    o = OracleState(gs)
    o.init_planes()
    rng = np.random.default_rng(box)
    pts = [(0, 0), (gs.cols - 1, gs.rows - 1), (3, gs.rows - 2), (gs.cols - 4, 1), (5, 5)]
    pts += [(int(rng.integers(0, gs.cols)), int(rng.integers(0, gs.rows))) for _ in range(7)]
    L = lib()
    n_side = (box + 1) // 2
    checked = 0
    for (nx, ny) in pts:
        for plane in (np.ascontiguousarray(o.norm4[ny, nx]),                       # the pixel's own (random) plane
                      np.array([0.0, 0.0, -1.0, 550.0], dtype=np.float32),          # fronto-parallel, mid range
                      np.array([0.3, -0.2, -0.93, 0.0], dtype=np.float32)):         # degenerate: d = 0
            out = (C.c_float * 8)()
            valid = (C.c_int * 8)()
            n_dis = L.gipuma_oracle_push_costs(C.byref(gs.desc), nx, ny, fptr(plane), out, valid)
            assert n_dis == 2 * n_side * (n_side + 5)
            for c in range(8):
                dist = 1 if c < 4 else 5
                dx, dy = [(0, dist), (0, -dist), (dist, 0), (-dist, 0)][c & 3]
                px, py = nx + dx, ny + dy
                inside = 0 <= px < gs.cols and 0 <= py < gs.rows
                assert bool(valid[c]) == inside
                if inside:
                    want = L.gipuma_oracle_multiview_cost(C.byref(gs.desc), px, py, fptr(plane))
                    got = np.float32(out[c])
                    assert np.float32(want).view(np.uint32) == got.view(np.uint32), (box, nx, ny, c, want, got)
                    checked += 1
    assert checked > 200
