"""ground-truth evaluation and the multi-view batch runner."""
import json
import os

import numpy as np
import pytest

from gipuma_amd import abi, dmb, evaluate, synth


def test_compute_error_restates_groundtruthutils():
    gt = np.array([[10.0, 0.0, 5.0], [-1.0, 8.0, 2.0]], dtype=np.float32)     # 0 and -1: no ground truth
    disp = np.array([[10.4, 3.0, 7.0], [9.0, 8.9, 2.0]], dtype=np.float32)
    r = evaluate.compute_error(gt, disp, tol=1.0, tol2=0.5)
    assert r["num_gt"] == 4
    assert r["error"] == pytest.approx(1 / 4)       # only |5-7| >= 1
    assert r["error2"] == pytest.approx(2 / 4)      # |5-7| and |8-8.9| >= 0.5
    occ = np.array([[1, 1, 0], [1, 1, 1]])
    valid = np.array([[1, 1, 1], [0, 0, 1]])
    r = evaluate.compute_error(gt, disp, tol=1.0, occ_mask=occ, valid=valid)
    assert r["error_nocc"] == 0.0                   # the erroneous pixel is occluded
    assert r["error_valid"] == pytest.approx(1 / 3) and r["valid_ratio"] == pytest.approx(3 / 4)
    assert r["error_valid_all"] == pytest.approx((1 + 1) / 4)
    assert evaluate.compute_error(gt * 2, disp, tol=1.0, div_factor=2.0)["error"] == pytest.approx(1 / 4)


def test_compute_normal_error():
    g = np.zeros((2, 2, 3), dtype=np.float32)
    g[0, 0] = (0, 0, 1)
    g[0, 1] = (0, 0, 1)
    g[1, 0] = (1, 0, 0)                              # g[1,1] = 0: no ground truth
    n = np.zeros_like(g)
    n[0, 0] = (0, 0, 1)
    n[0, 1] = (0, np.sin(0.25), np.cos(0.25))
    n[1, 0] = (0, 1, 0)
    e, e2, ang = evaluate.compute_normal_error(n, g, tol=0.2, tol2=0.3)
    assert e == pytest.approx(2 / 3) and e2 == pytest.approx(1 / 3)
    assert ang[0, 1] == pytest.approx(0.25, abs=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_batch_runner_matches_single_view_runs(hip, tmp_path, in_flight):
    """three reference views of a small scan through the batch runner (images resident once,
    per-view selectViews; one at a time, or 2 / 3 views in flight on their own streams) == the same
    views solved one by one"""
    from gipuma_amd import batch
    from gipuma_amd.problem import runcuda, GlobalState
    cfg = synth.tiny_config(cols=96, rows=64, n_src=4, blocksize=9, iterations=2, n_best=2)
    gs, info = synth.build_problem(cfg)
    ids = info["view_ids"]
    img_dir, p_dir, out = tmp_path / "img", tmp_path / "calib", tmp_path / "out"
    img_dir.mkdir()
    p_dir.mkdir()
    P = synth.dtu_projection_matrices()
    names = []
    for im, vid in zip(gs.images, ids):
        name = "rect_%03d.pgm" % vid
        with open(img_dir / name, "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (gs.cols, gs.rows) + im.astype(np.uint8).tobytes())
        with open(p_dir / (name + ".P"), "w") as f:
            for r in P[vid]:
                f.write(" ".join("%.6f" % v for v in r) + "\n")
        names.append(name)
    names.sort()
    refs = names[:3]
    rc = batch.main(["--images-folder", str(img_dir), "--p-folder", str(p_dir), "--output-folder", str(out),
                     "--views", ",".join(refs), "--blocksize=9", "--iterations=2", "--n_best=2",
                     "--depth_min=300", "--depth_max=800", "--min_angle=2", "--max_angle=60",
                     "--max_views=10", "--cam_scale=%.9g" % np.float32(cfg["cam_scale"]),
                     "--in_flight=%d" % in_flight])
    assert rc == 0
    rep = json.load(open(out / "batch_rank0.json"))
    assert [v["ref"] for v in rep["views"]] == refs
    P_all = [batch.read_p_file(str(p_dir / (n + ".P"))) for n in names]
    host = [batch.read_pgm(str(img_dir / n)) for n in names]
    ap = batch.AlgorithmParameters(iterations=2, n_best=2, depthMin=300.0, depthMax=800.0, min_angle=2.0,
                                   max_angle=60.0, max_views=10)
    ap.set_blocksize(9)
    for ref in refs:
        cs, used, apv = batch.plan_views(P_all, names, names.index(ref), gs.cols, gs.rows, ap,
                                         float(np.float32(cfg["cam_scale"])))
        g1 = GlobalState([host[i] for i in used], cs, list(range(1, len(used))), apv, seed=1)
        n4, c = runcuda(g1)
        folder = out / os.path.splitext(ref)[0]
        assert np.array_equal(dmb.read_dmb(str(folder / "disp.dmb")).view(np.uint32), n4[..., 3].view(np.uint32))
        assert np.array_equal(dmb.read_dmb(str(folder / "cost.dmb")).view(np.uint32), c.view(np.uint32))
        if ref == refs[1]:   # and one of them against the oracle (the others share its code path)
            from tests.oracle_lib import OracleState
            o4, oc = OracleState(g1).run()
            assert np.array_equal(o4.view(np.uint32), n4.view(np.uint32))
            assert np.array_equal(oc.view(np.uint32), c.view(np.uint32))
    assert rep["in_flight"] == in_flight and rep["mpix_per_s_batch"] > 0
    if in_flight == 1:
        assert all("wall_ms" in v and v["wall_ms"] >= v["device_ms"] for v in rep["views"])
    else:
        assert all("wall_ms" in v and "device_ms" not in v for v in rep["views"])
