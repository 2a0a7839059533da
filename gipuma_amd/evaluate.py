"""Ground-truth evaluation, restating the reference's groundTruthUtils.h:
`computeError` (:22-95) and `computeNormalError` (:97-135).  Pure numpy, host side: these are
reports about a finished depth/normal map, not part of the device path."""
import numpy as np


def compute_error(gt_disp, disp, tol=1.0, tol2=None, occ_mask=None, valid=None, div_factor=1.0):
    """Fractions of erroneous pixels like computeError: a pixel is an error when
    |gt/div_factor - disp| >= tol; pixels whose ground truth is 0 or -1 are not counted.

    Returns a dict: error, error2 (second tolerance), error_nocc (non-occluded pixels of occ_mask),
    error_valid / error_valid_all (pixels that passed the caller's `valid` check), valid_ratio."""
    d1 = np.asarray(gt_disp, dtype=np.float32) / np.float32(div_factor)
    d2 = np.asarray(disp, dtype=np.float32)
    has_gt = ~((d1 == 0.0) | (d1 == -1.0))
    n_gt = int(has_gt.sum())
    diff = np.abs(d1 - d2)
    err = has_gt & (diff >= tol)
    out = {"num_gt": n_gt, "error": err.sum() / n_gt if n_gt else float("nan")}
    t2 = tol if tol2 is None else tol2
    out["error2"] = (has_gt & (diff >= t2)).sum() / n_gt if n_gt else float("nan")
    if occ_mask is not None:
        nocc = has_gt & (np.asarray(occ_mask) != 0)
        out["error_nocc"] = (err & nocc).sum() / max(1, int(nocc.sum()))
    if valid is not None:
        v = has_gt & (np.asarray(valid) != 0)
        nv = int(v.sum())
        ev = int((err & v).sum())
        out["error_valid"] = ev / max(1, nv)
        out["error_valid_all"] = (ev + (n_gt - nv)) / n_gt if n_gt else float("nan")
        out["valid_ratio"] = nv / n_gt if n_gt else float("nan")
    return out


def compute_normal_error(normals, gt_normals, tol=0.2, tol2=0.3):
    """computeNormalError: angle (acos of the dot product, mathUtils.h:16-24) between estimated and
    ground-truth unit normals; pixels whose gt components sum to < 0.1 carry no ground truth
    (groundTruthUtils.h:116-117).  Returns (error, error2, angle_map)."""
    n = np.asarray(normals, dtype=np.float32)
    g = np.asarray(gt_normals, dtype=np.float32)
    has_gt = g.sum(-1) >= 0.1
    dot = np.clip((n * g).sum(-1), -1.0, 1.0)
    ang = np.arccos(dot)
    ang = np.where(has_gt, ang, 0.0)
    n_gt = int(has_gt.sum())
    if not n_gt:
        return float("nan"), float("nan"), ang
    return float((ang[has_gt] > tol).sum() / n_gt), float((ang[has_gt] > tol2).sum() / n_gt), ang
