#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the red-black PatchMatch hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole reference view through the hot path: init + `iterations` x (black, red)
sweeps + finalize (what runcuda() does, reference gipuma.cu:1906-1944), on synthetic 1600x1200
DTU-geometry frames (config C of SURVEY.md 8d: 10 source views, box 15, 8 iterations, best-3)
that are already resident in HBM when the timed region starts.

Multi-GPU: reference views are independent problems (the reference runs one process per view,
scripts/dtu_fast.sh:30-55), so rank r solves its own reference view on GPU r; there is no
data-path collective and RCCL is not used.  torch.distributed (gloo) only carries the barrier and
the max-over-ranks of the wall time.  scaling = "weak": per-GPU work is fixed.

Prints ONE JSON line on rank 0.  `value` is the whole-job aggregate (sum over GPUs); at N=1 it is
the BASELINE.json per-GPU figure.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_sweep_launch(n_pixels, n_views):
    """SURVEY.md 8d compulsory-HBM model, unfused 6-kernel schedule: per iteration and pixel
    152 B of state + 24*(N+1) B of images.  One fused launch of ours (one colour: close + far +
    refine) stands for three of those six kernels = half an iteration."""
    return n_pixels * (152 + 24 * (n_views + 1)) / 2.0


def cpu_baseline(gs, iterations, target_seconds=12.0):
    """the oracle ("port") timed on this host's cores on a bounded band of rows of the SAME
    workload, scaled to the full frame (work per pixel and per iteration is constant)."""
    from tests.oracle_lib import lib
    L = lib()
    ti, ts = C.c_double(), C.c_double()
    rows = gs.rows
    band = min(rows, 4)
    y0 = rows // 2
    L.gipuma_oracle_time_band(C.byref(gs.desc), y0, y0 + band, C.byref(ti), C.byref(ts))
    per_row = (ti.value + ts.value) / band
    band = int(max(band, min(rows - y0, target_seconds / max(per_row, 1e-9))))
    L.gipuma_oracle_time_band(C.byref(gs.desc), y0, y0 + band, C.byref(ti), C.byref(ts))
    full = (ti.value + iterations * ts.value) * rows / band
    return {
        "value": gs.rows * gs.cols / full / 1e6,
        "unit": "Mpix/s",
        "cores": L.gipuma_oracle_num_threads(),
        "kind": "port",
        "sample": "oracle (oracle/gipuma_oracle.c, gcc -O2 -fopenmp) on rows [%d,%d) of the same "
                  "frame: init %.2fs + 1 iteration %.2fs, scaled x%d iterations x rows/%d"
                  % (y0, y0 + band, ti.value, ts.value, iterations, band),
        "est_full_frame_seconds": full,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C", help="workload: A, B, C (default, the metric's config) or D")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--colour", action="store_true", help="-color_processing variant of the workload (T=float4)")
    ap.add_argument("--cols", type=int, default=0, help="experiments only: override the frame width")
    ap.add_argument("--rows", type=int, default=0, help="experiments only: override the frame height")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import torch.distributed as dist

    from gipuma_amd import abi, synth
    from gipuma_amd.problem import Session
    from gipuma_amd.shard import views_for_rank

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo prints a connection banner on stdout; keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # one rank per GPU; modulo only matters when a launch oversubscribes the node (e.g. a 2-rank
    # dry run on a 1-GPU box)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index

    # this rank's shard: one reference view (config E = 8 different views, one per GPU)
    ref_view = views_for_rank(synth.DTU_REF_VIEWS, rank, world)[0]
    over = {}
    if args.cols:
        over["cols"] = args.cols
    if args.rows:
        over["rows"] = args.rows
    gs, info = synth.build_problem(args.config, ref_view=ref_view, device=dev, keep_on_device=True,
                                   colour=args.colour, **over)
    gs.desc.device_id = dev_index
    torch.cuda.synchronize()
    n_pix = gs.rows * gs.cols
    n_views = len(gs.selected)
    iterations = gs.params.iterations

    sess = Session(gs)

    def barrier():
        if world > 1:
            dist.barrier()

    sweep_ms, total_ms, init_ms = [], [], []
    for _ in range(args.warmup):
        sess.solve(timing=True)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t = sess.solve(timing=True)  # returns after the last kernel's HIP event
        sweep_ms.append(t.ms_sweep_avg)
        total_ms.append(t.ms_total)
        init_ms.append(t.ms_init)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])

    # quality vs the analytic ground truth (sanity: the solver reconstructs the surface)
    n4, cost = sess.get_state()
    sess.close()
    gt = info["gt_depth"]
    valid = cost != abi.MAXCOST
    rel = np.abs(n4[..., 3] - gt) / gt
    quality = {"depth_rel_err_median_vs_gt": float(np.median(rel[valid])),
               "frac_within_1pct_of_gt": float((rel < 0.01).mean())}

    if rank == 0:
        value = world * args.steps * n_pix / elapsed / 1e6
        ms_launch = float(np.mean(sweep_ms))
        alg = algorithmic_bytes_per_sweep_launch(n_pix, n_views)
        achieved = alg / (ms_launch * 1e-3) / 1e9
        traffic = None
        # HBM-side bytes per sweep launch from the PMC passes of the SAME command
        # (scripts/pmc_passes.sh -> profiles/pmc_latest.json): FETCH_SIZE x2 (gfx950 correction,
        # MI355X_MICROARCH.md HBM section) + WRITE_SIZE; separate --pmc passes
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and args.config == "C":
            try:
                pj = json.load(open(pmc))
                traffic = (pj["hbm_read_bytes_per_launch_x2corr"] + pj["hbm_write_bytes_per_launch"]) / 1e9
            except Exception:
                traffic = None
        box = gs.params.box_hsize
        S = ((box - 1) // 2 + 1) ** 2
        r_ref = 0
        dz = gs.params.max_disparity / 2.0
        while dz >= 0.01:
            r_ref += 1
            dz /= 10.0
        samples_per_frame = n_pix * (1 + iterations * (8 + r_ref)) * n_views * S
        out = {
            "metric": "Mpixels/sec/GPU (1600x1200, 10 src views, 8 iters)" if args.config == "C" and not args.colour
                      else "Mpixels/sec/GPU (config %s%s)" % (args.config, ", colour" if args.colour else ""),
            "value": value,
            "unit": "Mpix/s",
            "value_per_gpu": value / world,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "config %s: %dx%d, %d source views, box %d, %d iterations, best-%d; "
                                   "analytic textured surface rendered through DTU calibration "
                                   "(ref view %d); one reference view per GPU"
                                   % (args.config, gs.cols, gs.rows, n_views, box, iterations,
                                      gs.params.n_best, ref_view),
                       "parallelism": "independent reference views, %d per step" % world,
                       "device_ms_total": float(np.mean(total_ms)),
                       "device_ms_init": float(np.mean(init_ms))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "GB per launch (L2<->fabric incl. Infinity Cache hits)",
                         "kernel": "one half-sweep launch (one colour: close+far+refine fused); mean over the 16 of a "
                                   "view: pm::sweep_cols_kernel x3 + pm::sweep_kernel x13 on config C",
                         "kernel_ms": ms_launch,
                         "algorithmic_bytes_per_launch": alg,
                         "note": "compute/gather bound by construction (SURVEY F5): "
                                 "%.3g patch samples/s" % (samples_per_frame / (np.mean(total_ms) * 1e-3))},
            "quality": quality,
        }
        # The roof that actually binds: vector-ALU issue.  PMC
        # SQ_INSTS_VALU of the same command (profiles/pmc_latest.json) x 64 lanes / launch time,
        # against 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz for full-rate ops (the mix also holds
        # half-rate cvt/min/floor, so 100 % is not reachable; DESIGN.md gives the mix bound).
        if os.path.exists(pmc) and args.config == "C":
            try:
                pj = json.load(open(pmc))
                lane_ops = pj["SQ_INSTS_VALU"] * 64.0
                peak = 256 * 4 * 32 * 2.4e9
                out["roofline_valu"] = {
                    "bound": "valu", "achieved": lane_ops / (ms_launch * 1e-3) / 1e12, "peak": peak / 1e12,
                    "unit": "T lane-instr/s", "frac": lane_ops / (ms_launch * 1e-3) / peak,
                    # per NOMINAL patch sample (E*N*S of SURVEY 8, before the exact skipping) ...
                    "valu_instr_per_nominal_patch_sample": pj["SQ_INSTS_VALU"] * 64.0 / (
                        samples_per_frame / (1 + iterations * (8 + r_ref)) * (8 + r_ref) / 2.0),
                    # ... and per patch sample actually evaluated (one window load each)
                    "valu_instr_per_window_load": pj["SQ_INSTS_VALU"] / max(1.0, pj.get("SQ_INSTS_VMEM_RD", 0.0)),
                    "source": "profiles/pmc_latest.json (rocprofv3 --pmc SQ_INSTS_VALU, same command)"}
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            # the oracle reads host memory: same frames, copied back from HBM
            from gipuma_amd.problem import GlobalState, runcuda
            gs_host = GlobalState([im.cpu().numpy() for im in gs.images], gs.cameras, gs.selected,
                                  gs.params, seed=gs.desc.seed)  # (rows, cols[, 4]) arrays
            # the boundary as the reference's main.cpp uses it: host images in, host planes out
            # (upload + window packing + solve + download); reported, never `value`
            t1 = time.perf_counter()
            runcuda(gs_host)
            host_s = time.perf_counter() - t1
            out["host_boundary"] = {"ms_per_view_host_in_host_out": host_s * 1e3,
                                    "value_pcie_inclusive": n_pix / host_s / 1e6, "unit": "Mpix/s"}
            out["cpu_baseline"] = cpu_baseline(gs_host, iterations)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
