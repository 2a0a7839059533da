"""Batch runner: every image of a scan as reference view, sharded over the GPUs of a node
(replaces the per-view process loop of the reference's scripts/dtu_fast.sh:30-55).

    python -m gipuma_amd.batch --images-folder scan9/ --p-folder calib/ --output-folder results/ \\
        --blocksize=15 --iterations=8 --n_best=3 --depth_min=300 --depth_max=800 \\
        --min_angle=10 --max_angle=30 --max_views=10
    python -m torch.distributed.run --nproc-per-node 8 -m gipuma_amd.batch ...      # 8 GPUs

MI355X-first differences from the shell loop:
  * one process per GPU handles MANY reference views; the scan's images are decoded and uploaded
    to HBM once per process (a 49-view DTU scan is 0.4 GB of the 288 GB) and every session binds
    them by device pointer -- no per-view process start, disk read or PCIe upload;
  * reference views are sharded round-robin over the ranks (gipuma_amd.shard); there is no
    inter-GPU communication;
  * several reference views are kept in flight per GPU (--in_flight, default 2): one session and HIP
    stream each, the solves enqueued asynchronously -- launch tails of one view fill with workgroups of
    another (bench.py `value_views_in_flight`: +2 % with 2 / 3 views on config C since the fused launches of
    round 4 -- they leave little to overlap --, far more on frames whose tiles do not fill the GPU, e.g. config B);
  * results land in <output>/<refname>/{disp.dmb, normals.dmb, cost.dmb} -- the dumps the
    reference writes (main.cpp:1001-1015) and fusibile reads.

Images: what the reference's scripts hand to imread (main.cpp:739-751) -- PNG, JPG (through PIL), binary PGM / PPM.
Calibration: <p-folder>/<image name>.P (fileIoUtils.h:83-110).
"""
import argparse
import collections
import ctypes as C
import json
import os
import sys
import time

import numpy as np

from . import abi, dmb
from .cameras import get_camera_parameters, read_p_file, select_views
from .problem import AlgorithmParameters, GlobalState, Session
from .shard import views_for_rank


def read_pgm(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"P5":
        raise ValueError("%s: only binary PGM (P5) is read here" % path)
    tokens, pos = [], 2
    while len(tokens) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tokens.append(int(data[pos:end]))
        pos = end
    cols, rows, maxv = tokens
    if maxv != 255:
        raise ValueError("%s: 8-bit images only" % path)
    img = np.frombuffer(data, dtype=np.uint8, count=rows * cols, offset=pos + 1).reshape(rows, cols)
    return img.astype(np.float32)


IMAGE_EXTENSIONS = (".png", ".jpg", ".jpeg", ".pgm", ".ppm", ".pnm")


def read_image(path):
    """imread(path, IMREAD_GRAYSCALE) as float32 (main.cpp:741, :941).  PGM: the bytes.  Everything else through PIL:
    single-channel files as they are (16-bit: the high byte); colour PNG / PPM by the rule of the C++ front-end
    (gipuma_host.cpp read_image_gray: libpng's 15-bit 9797 / 19234 / 3737 for PNG, OpenCV's 14-bit BGR2GRAY for PPM);
    JPEG decoded to luma by libjpeg itself (PIL draft mode 'L'), which is what OpenCV's JPEG reader does for
    IMREAD_GRAYSCALE.  One known deviation: a 16-bit RGB / RGBA PNG reaches this function as PIL's 8-bit RGB, so its gray
    is formed on the high bytes; libpng (and the C++ front-end, read_png_rgb8) forms it on the 16-bit samples and then
    takes the high byte -- the last bit may differ for such files."""
    with open(path, "rb") as f:
        head = f.read(4)
    if head[:2] == b"P5":
        return read_pgm(path)
    from PIL import Image
    im = Image.open(path)
    if im.format == "JPEG":
        im.draft("L", im.size)
        return np.asarray(im.convert("L"), dtype=np.uint8).astype(np.float32)
    if im.mode in ("I;16", "I;16B", "I;16L", "I"):
        return (np.asarray(im, dtype=np.uint32) >> 8).astype(np.uint8).astype(np.float32)
    if im.mode in ("L", "1", "LA"):
        return np.asarray(im.convert("L"), dtype=np.uint8).astype(np.float32)
    if im.mode == "P" and im.palette is not None and im.palette.mode == "L":
        return np.asarray(im.convert("L"), dtype=np.uint8).astype(np.float32)
    rgb = np.asarray(im.convert("RGB"), dtype=np.int64)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    if im.format == "PPM":
        return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.float32)
    return ((9797 * r + 19234 * g + 3737 * b) >> 15).astype(np.float32)


def plan_views(P_all, names, ref_idx, cols, rows, ap, cam_scale=1.0):
    """the reference's per-view recipe: reference first, every other image as a candidate,
    selectViews keeps those inside the angle cone (main.cpp:430-499); returns the camera set
    restricted to [reference] + selected views and their global indices"""
    order = [ref_idx] + [i for i in range(len(names)) if i != ref_idx]
    cs_all = get_camera_parameters([P_all[i] for i in order], cam_scale=cam_scale)
    ap_view = AlgorithmParameters(**{k: getattr(ap, k) for k in vars(ap)})
    subset, dmin, dmax = select_views(cs_all, cols, rows, ap.min_angle, ap.max_angle, ap.max_views,
                                      ap.depthMin, ap.depthMax)
    used = [order[0]] + [order[i] for i in subset]
    cs = get_camera_parameters([P_all[i] for i in used], cam_scale=cam_scale)
    ap_view.depthMin, ap_view.depthMax = dmin, dmax
    return cs, used, ap_view


def main(argv=None):
    pa = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    pa.add_argument("--images-folder", required=True)
    pa.add_argument("--p-folder", required=True)
    pa.add_argument("--output-folder", required=True)
    pa.add_argument("--views", default="all", help="comma separated image names to use as reference (default all)")
    pa.add_argument("--blocksize", type=int, default=19)
    pa.add_argument("--iterations", type=int, default=8)
    pa.add_argument("--n_best", type=int, default=2)
    pa.add_argument("--cost_gamma", type=float, default=10.0)
    pa.add_argument("--depth_min", type=float, default=-1.0)
    pa.add_argument("--depth_max", type=float, default=-1.0)
    pa.add_argument("--min_angle", type=float, default=5.0)
    pa.add_argument("--max_angle", type=float, default=45.0)
    pa.add_argument("--max_views", type=int, default=9)
    pa.add_argument("--cam_scale", type=float, default=1.0)
    pa.add_argument("--seed", type=int, default=1)
    pa.add_argument("--in_flight", type=int, default=2,
                    help="reference views kept in flight per GPU (1: one at a time, with per-view device times)")
    pa.add_argument("--mode", choices=["exact", "fast", "literal"], default="exact",
                    help="exact: bit-identical to the numerical model (default); fast: tolerance-judged kernels "
                         "(GIPUMA_HIP_FLAG_FAST); literal: the reference's own operation order, bit-identical to the "
                         "reference's code, about 20x slower (GIPUMA_HIP_FLAG_LITERAL)")
    args = pa.parse_args(argv)
    # the reference parses these with sscanf("%f") into float fields (main.cpp:300-360)
    for k in ("cost_gamma", "depth_min", "depth_max", "min_angle", "max_angle", "cam_scale"):
        setattr(args, k, float(np.float32(getattr(args, k))))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise abi.GipumaHipError("gipuma_amd.batch needs a GPU; there is no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)

    names = sorted(n for n in os.listdir(args.images_folder) if n.lower().endswith(IMAGE_EXTENSIONS))
    if len(names) < 2:
        raise SystemExit("need at least 2 images (png / jpg / pgm / ppm) in %s" % args.images_folder)
    P_all = [read_p_file(os.path.join(args.p_folder, n + ".P")) for n in names]
    # the whole scan resident in HBM, once
    t0 = time.perf_counter()
    host = [read_image(os.path.join(args.images_folder, n)) for n in names]
    rows, cols = host[0].shape
    dev = [torch.from_numpy(im).to("cuda:%d" % dev_index) for im in host]
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0

    refs = names if args.views == "all" else [v for v in args.views.split(",") if v]
    mine = views_for_rank(refs, rank, world) if len(refs) >= world else refs[rank:rank + 1]
    ap = AlgorithmParameters(iterations=args.iterations, n_best=args.n_best, gamma=args.cost_gamma,
                             depthMin=args.depth_min, depthMax=args.depth_max, min_angle=args.min_angle,
                             max_angle=args.max_angle, max_views=args.max_views)
    ap.set_blocksize(args.blocksize)
    os.makedirs(args.output_folder, exist_ok=True)
    report = []
    in_flight = max(1, args.in_flight)
    pending = collections.deque()  # (session, reference name, source names, start time, timing or None)

    def retire():
        s, ref_name, sources, tw0, t = pending.popleft()
        try:
            n4, cost = s.get_state()  # waits for the session's stream
        finally:
            s.close()
        wall_ms = (time.perf_counter() - tw0) * 1e3  # session set-up + solve (+ what ran beside it) + download
        folder = os.path.join(args.output_folder, os.path.splitext(ref_name)[0])
        os.makedirs(folder, exist_ok=True)
        dmb.write_dmb(os.path.join(folder, "disp.dmb"), n4[..., 3])
        dmb.write_dmb(os.path.join(folder, "normals.dmb"), n4[..., :3])
        dmb.write_dmb(os.path.join(folder, "cost.dmb"), cost)
        entry = {"ref": ref_name, "sources": sources, "wall_ms": wall_ms,
                 "mpix_per_s_wall": rows * cols / (wall_ms * 1e-3) / 1e6}
        if t is not None:  # one view at a time: the device time is that view's alone
            entry.update({"device_ms": t.ms_total, "mpix_per_s": rows * cols / (t.ms_total * 1e-3) / 1e6})
        report.append(entry)

    t_batch0 = time.perf_counter()
    try:
        for ref_name in mine:
            ref_idx = names.index(ref_name)
            cs, used, ap_view = plan_views(P_all, names, ref_idx, cols, rows, ap, args.cam_scale)
            if len(used) < 2:
                report.append({"ref": ref_name, "skipped": "no source view inside the angle cone"})
                continue
            imgs = [dev[i] for i in used]
            # the scan's planes stay put for the whole batch: what the library derives from them (8-bit
            # check, window-packed copies) is made once per image, not once per reference view that uses it
            gs = GlobalState(imgs, cs, list(range(1, len(used))), ap_view, seed=args.seed,
                             device_ptrs=[t.data_ptr() for t in imgs], rows=rows, cols=cols, device_id=dev_index,
                             flags=abi.FLAG_CACHE_IMAGES)
            tw0 = time.perf_counter()
            s = Session(gs, fast=args.mode == "fast", literal=args.mode == "literal")
            try:
                if in_flight == 1:
                    t = s.solve(timing=True)
                else:
                    t = None
                    s.solve(timing=False)  # asynchronous: returns once the launches are enqueued
            except Exception:
                s.close()
                raise
            pending.append((s, ref_name, [names[i] for i in used[1:]], tw0, t))
            while len(pending) >= in_flight:
                retire()
        while pending:
            retire()
    finally:  # (an error above: do not leave sessions of this batch behind)
        for leftover in pending:
            leftover[0].close()
        pending.clear()
        # ... nor the image cache: its entries are keyed by the device addresses of `dev`'s tensors, which
        # torch hands out again once they are freed (the library refuses while a session still uses them)
        abi.load_library().gipuma_hip_cache_clear()
    t_batch = time.perf_counter() - t_batch0
    with open(os.path.join(args.output_folder, "batch_rank%d.json" % rank), "w") as f:
        n_done = sum(1 for r in report if "skipped" not in r)
        json.dump({"rank": rank, "world": world, "device": dev_index, "load_seconds": t_load,
                   "in_flight": in_flight, "batch_seconds": t_batch,
                   "mpix_per_s_batch": n_done * rows * cols / max(t_batch, 1e-9) / 1e6,
                   "views": report}, f, indent=1)
    print("rank %d/%d: %d reference views on cuda:%d" % (rank, world, len(report), dev_index))
    return 0


if __name__ == "__main__":
    sys.exit(main())
