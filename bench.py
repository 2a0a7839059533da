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
data-path collective and RCCL is not used.  torch.distributed (gloo) only carries the barrier, the
max-over-ranks of the wall time and the device identities.  scaling = "weak": per-GPU work is fixed.
`--gpus N` launches its own N ranks (one process per GPU) when it is not already running under a
launcher (no WORLD_SIZE in the environment).

Prints ONE JSON line on rank 0.  `value` is the whole-job aggregate (sum over GPUs); at N=1 it is
the BASELINE.json per-GPU figure.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _exp_env(name, default):
    """the library (and gipuma_amd.abi) read their A/B switches only when GIPUMA_HIP_EXPERIMENTS is set"""
    if os.environ.get("GIPUMA_HIP_EXPERIMENTS", "0") in ("", "0"):
        return default
    return os.environ.get(name, default)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_sweep_launch(n_pixels, n_views):
    """SURVEY.md 8d compulsory-HBM model, unfused 6-kernel schedule: per iteration and pixel
    152 B of state + 24*(N+1) B of images.  One fused launch of ours (one colour: close + far +
    refine) stands for three of those six kernels = half an iteration."""
    return n_pixels * (152 + 24 * (n_views + 1)) / 2.0


# ------------------------------------------------------------------------------------------------
# CPU baselines (reported, never optimised against): bounded samples of the SAME workload
# ------------------------------------------------------------------------------------------------
def effective_cores():
    """(cores this process may really use, how that was found): the affinity mask, cut down by the cgroup's CPU
    quota where one is set -- a container that SEES 256 hardware threads may be allowed the time of 8"""
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:  # noqa: BLE001
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            quota = None
    n = n_aff if quota is None else max(1, min(n_aff, int(quota + 0.5)))
    return n, {"affinity": n_aff, "cgroup_cpu_quota": quota}


def _ref_worker(problem_file, bx0, bx1, by0, by1, cpu):
    """one process = one core: the reference's OWN device code (oracle/_ref/libgipuma_ref.so =
    /root/reference/gipuma.cu lines 1..1824 compiled for the CPU, oracle/ref_shim/build_ref.sh)
    on a window of 32x32-pixel blocks of the frame: its init kernel + one iteration (6 launches).
    The worker pins itself to one CPU, maps the shared frames, runs the same launches once untimed (first
    touch of its state planes) and then measured."""
    if cpu >= 0 and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, {cpu})
        except OSError:
            pass
    from gipuma_amd.problem import load_problem
    from tests import ref_lib
    gs = load_problem(problem_file)
    L = ref_lib.lib()
    L.ref_time_window.argtypes = [C.POINTER(type(gs.desc)), C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]
    ti, ts = C.c_double(), C.c_double()
    L.ref_time_window_warm.argtypes = [C.c_int]
    L.ref_time_window_warm(0)
    L.ref_time_window(C.byref(gs.desc), bx0, bx0 + 1, by0, by1, C.byref(ti), C.byref(ts))  # untimed: pages, caches
    print("READY", flush=True)
    sys.stdin.readline()  # start together with the other workers
    rc = L.ref_time_window(C.byref(gs.desc), bx0, bx1, by0, by1, C.byref(ti), C.byref(ts))
    print(json.dumps({"rc": rc, "sec_init": ti.value, "sec_iter": ts.value,
                      "pixels": (bx1 - bx0) * (by1 - by0) * 1024}), flush=True)


def _ref_baseline(problem_file, rows, cols, iterations, n_proc, blocks_each=4):
    """the reference's own code on `n_proc` cores at once: one single-threaded, pinned process per core
    (its device code keeps block state in globals), each on its own run of interior blocks of
    the same frame; throughput = pixels of all windows / slowest process, scaled by iterations
    (work per pixel and per iteration is constant)."""
    from tests import ref_lib
    if not ref_lib.available():
        return None
    gx, gy = cols // 32, rows // 32  # interior, fully covered blocks only
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    span = max(1, gx - 2 - blocks_each)
    procs = []
    for i in range(n_proc):
        b = (i * 7919) % (span * (gy - 2))
        bx0, by0 = 1 + b % span, 1 + b // span
        cpu = cpus[(i * max(1, len(cpus) // max(n_proc, 1))) % len(cpus)] if cpus else -1
        code = ("import sys; sys.path.insert(0, %r); import bench; bench._ref_worker(%r, %d, %d, %d, %d, %d)"
                % (ROOT, problem_file, bx0, min(bx0 + blocks_each, gx - 1), by0, by0 + 1, cpu))
        errf = tempfile.TemporaryFile(mode="w+")  # (kept: a worker that dies says why)
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      stderr=errf, text=True, env=dict(os.environ, OMP_NUM_THREADS="1")))
        procs[-1].errf = errf
    for p in procs:  # every worker has mapped the frames and touched its pages
        while True:
            line = p.stdout.readline()
            if not line or line.startswith("READY"):
                break
    t0 = time.perf_counter()
    for p in procs:
        p.stdin.write("go\n")
        p.stdin.flush()
    res = []
    for p in procs:
        out = p.stdout.read()
        p.wait()
        lines = [l for l in out.splitlines() if l.startswith("{")]
        if lines:
            res.append(json.loads(lines[-1]))
        else:
            p.errf.seek(0)
            print("bench: a reference-baseline worker produced no result (rc %s): %s"
                  % (p.returncode, p.errf.read()[-400:].strip()), file=sys.stderr)
        p.errf.close()
    wall = time.perf_counter() - t0
    if not res or any(r["rc"] for r in res):
        return None
    px = sum(r["pixels"] for r in res)
    t_init = max(r["sec_init"] for r in res)
    t_iter = max(r["sec_iter"] for r in res)
    full = (t_init + iterations * t_iter) * rows * cols / px
    return {"value": rows * cols / full / 1e6, "unit": "Mpix/s", "cores": n_proc, "kind": "reference",
            "sample": "the reference's own kernels (gipuma.cu compiled for the CPU, oracle/_ref), %d single-threaded "
                      "pinned processes at once, each on %d interior 32x32 blocks of the same frame after an untimed "
                      "pass over one block: init %.2fs + 1 iteration (6 launches) %.2fs (slowest), scaled by pixels "
                      "and x%d iterations; sample wall %.1fs"
                      % (n_proc, blocks_each, t_init, t_iter, iterations, wall),
            "est_full_frame_seconds": full}


def _port_worker(problem_file, seconds):
    from gipuma_amd.problem import load_problem
    from tests.oracle_lib import lib
    gs = load_problem(problem_file)
    L = lib()
    ti, ts = C.c_double(), C.c_double()
    rows = gs.rows
    band = 2
    y0 = rows // 2
    L.gipuma_oracle_time_band(C.byref(gs.desc), y0, y0 + band, C.byref(ti), C.byref(ts))
    per_row = (ti.value + ts.value) / band
    # (rows are the unit of the oracle's OpenMP loops: at least four per thread, so that the slowest thread does
    #  not decide the figure)
    band = int(max(band, 4 * L.gipuma_oracle_num_threads(), seconds / max(per_row, 1e-9)))
    band = min(band, rows - y0)
    L.gipuma_oracle_time_band(C.byref(gs.desc), y0, y0 + band, C.byref(ti), C.byref(ts))
    it = gs.params.iterations
    full = (ti.value + it * ts.value) * rows / band
    print(json.dumps({"value": gs.rows * gs.cols / full / 1e6, "unit": "Mpix/s",
                      "cores": L.gipuma_oracle_num_threads(), "kind": "port",
                      "sample": "oracle (oracle/gipuma_oracle.c, gcc -O3 -fopenmp) on rows [%d,%d) of the same frame: "
                                "init %.2fs + 1 iteration %.2fs, scaled x%d iterations x rows/%d"
                                % (y0, y0 + band, ti.value, ts.value, it, band),
                      "est_full_frame_seconds": full}), flush=True)


def _port_baseline(problem_file, threads, seconds):
    code = ("import sys; sys.path.insert(0, %r); import bench; bench._port_worker(%r, %f)"
            % (ROOT, problem_file, seconds))
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def cpu_baseline(problem_file, rows, cols, iterations):
    """`cpu_baseline` object of the bench line: the reference's own code on all cores of this host
    (kind "reference") when oracle/_ref exists, else the oracle port; the other figures (one core,
    the port) ride along under "all"."""
    ncpu, how = effective_cores()
    allv = {
        "reference_all_cores": _ref_baseline(problem_file, rows, cols, iterations, ncpu),
        "reference_1_core": _ref_baseline(problem_file, rows, cols, iterations, 1),
        "port_all_cores": _port_baseline(problem_file, ncpu, 5.0),
        "port_1_core": _port_baseline(problem_file, 1, 5.0),
    }
    head = allv["reference_all_cores"] or allv["port_all_cores"] or {"value": None, "kind": "port", "cores": ncpu}
    out = dict(head)
    out["cores_available"] = how  # (`cores` = the workers actually run: the affinity mask cut down by the cgroup quota)
    a, b = allv.get("reference_all_cores"), allv.get("reference_1_core")
    if a and b and a.get("value") and b.get("value"):
        out["speedup_all_cores_over_1"] = a["value"] / b["value"]
    out["all"] = allv
    return out


# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """no launcher around us: start one rank per GPU ourselves (the reference's unit of parallelism
    is one process per reference view, scripts/dtu_fast.sh:30-55, main.cpp:689-690)"""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C", help="workload: A, B, C (default, the metric's config) or D")
    ap.add_argument("--scene", default="smooth", choices=["smooth", "steps", "patchy"],
                    help="synthetic scene: smooth height field (default), depth steps + occluder + sensor noise, or the "
                         "smooth surface with 30 %% flat albedo and a periodic texture band")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (exhaustive schedule, "
                    "second scene, host boundary)")
    ap.add_argument("--colour", action="store_true", help="-color_processing variant of the workload (T=float4)")
    ap.add_argument("--cols", type=int, default=0, help="experiments only: override the frame width")
    ap.add_argument("--rows", type=int, default=0, help="experiments only: override the frame height")
    ap.add_argument("--views", type=int, default=0, help="experiments only: override the number of source views")
    ap.add_argument("--blocksize", type=int, default=0, help="experiments only: override the window size (e.g. 19, the "
                    "reference's default: algorithmparameters.h:25-26)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise the rank plumbing only (CPU tests)")
    ap.add_argument("--oversubscribe", action="store_true", help="allow more ranks than GPUs (tests on a 1-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist

    from gipuma_amd import abi, synth
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

    def barrier():
        if world > 1:
            dist.barrier()

    def gather(obj):
        if world == 1:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    # this rank's shard: one reference view (config E = 8 different views, one per GPU)
    ref_view = views_for_rank(synth.DTU_REF_VIEWS, rank, world)[0]

    if args.dry_run:
        ident = gather({"rank": rank, "device": "dry:%d" % local_rank, "ref_view": ref_view})
        barrier()
        if rank == 0:
            print(json.dumps({"metric": "dry run", "value": 0.0, "unit": "Mpix/s", "n_gpus": world, "steps": 0,
                              "warmup": 0, "dry_run": True, "scaling": "weak", "ranks": ident,
                              "config": {"workload": "none (plumbing test)"}}))
        if world > 1:
            dist.destroy_process_group()
        return

    from gipuma_amd.problem import Session
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    n_dev = torch.cuda.device_count()
    if world > n_dev and not args.oversubscribe:
        sys.exit("bench.py: %d ranks but only %d GPUs visible (use --oversubscribe for a plumbing test)" % (world, n_dev))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index
    props = torch.cuda.get_device_properties(dev_index)
    ident = gather({"rank": rank, "device_index": dev_index, "name": props.name,
                    "uuid": str(getattr(props, "uuid", "")), "ref_view": ref_view})
    if not args.oversubscribe:
        keys = {(i["device_index"], i["uuid"]) for i in ident}
        assert len(keys) == world, "ranks do not sit on %d distinct GPUs: %s" % (world, ident)

    over = {}
    if args.cols:
        over["cols"] = args.cols
    if args.rows:
        over["rows"] = args.rows
    if args.views:
        over["n_src"] = args.views
    if args.blocksize:
        over["blocksize"] = args.blocksize
    gs, info = synth.build_problem(args.config, ref_view=ref_view, device=dev, keep_on_device=True,
                                   colour=args.colour, scene=args.scene, **over)
    gs.desc.device_id = dev_index
    torch.cuda.synchronize()
    n_pix = gs.rows * gs.cols
    n_views = len(gs.selected)
    iterations = gs.params.iterations

    sess = Session(gs)
    sched = sess.schedule()
    sweep_ms, total_ms, init_ms, half_sweeps, hs_pushed, group_ms = [], [], [], [], 0, []
    for _ in range(args.warmup):
        sess.solve(timing=True)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t = sess.solve(timing=True)  # returns after the last kernel's HIP event
        hs_ms, hs_pushed = sess.launch_times()
        half_sweeps.append(hs_ms)
        group_ms.append(sess.group_times())
        sweep_ms.append(t.ms_sweep_avg)
        total_ms.append(t.ms_total)
        init_ms.append(t.ms_init)
    torch.cuda.synchronize()
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    per_rank = gather({"rank": rank, "ref_view": ref_view, "value": args.steps * n_pix / elapsed_local / 1e6})

    def quality_of(session, inf):
        n4_, cost_ = session.get_state()
        gt_ = inf["gt_depth"]
        valid_ = cost_ != abi.MAXCOST
        rel_ = np.abs(n4_[..., 3] - gt_) / gt_
        return {"depth_rel_err_median_vs_gt": float(np.median(rel_[valid_])),
                "frac_within_1pct_of_gt": float((rel_ < 0.01).mean())}

    # quality vs the analytic ground truth (sanity: the solver reconstructs the surface)
    quality = quality_of(sess, info)
    final_default = sess.get_state()  # (finalized maps of the last timed solve: compared with the exhaustive schedule's below)
    sess.close()

    if rank == 0:
        value = world * args.steps * n_pix / elapsed / 1e6
        ms_half_sweep = float(np.mean(sweep_ms))
        alg = algorithmic_bytes_per_sweep_launch(n_pix, n_views)
        # The dominant kernel: the fused pixel-per-lane sweep kernel.  Every half-sweep after the
        # column-per-lane / pushed ones is exactly one launch of it, timed by its own pair of HIP events on
        # the library's stream (gipuma_hip_launch_times); the earlier half-sweeps (other kernels, push
        # launches) only enter the per-view mean reported beside it.
        hs = np.asarray(half_sweeps, dtype=np.float64)  # [steps][2 * iterations]
        first_plain = 0
        if hs.size:
            cols_first = int(sched["cols_launches"])
            first_plain = min(hs.shape[1], max(cols_first, hs_pushed))
            if sched["group_from"] >= 0:  # the dominant kernel's half-sweeps: those of the plane-keyed propagation
                first_plain = min(hs.shape[1], max(first_plain, sched["group_from"]))
        dominant = hs[:, first_plain:] if hs.size and first_plain < hs.shape[1] else None
        ms_launch = float(dominant.mean()) if dominant is not None else ms_half_sweep
        # Where the propagation costs of a half-sweep come from pm::group_kernel (pm_group.h: box 15 on frames of
        # >= 1024 tiles, from the fifth half-sweep on), a half-sweep is TWO launches -- that kernel (the cost
        # evaluations of the reference's close + far kernels) and the fused sweep launch (their accept tests replayed
        # + the refinement kernel) -- and the former is the dominant kernel by time.  It is timed by its own pair of
        # HIP events on the library's stream (gipuma_hip_group_times); its algorithmic bytes are the close + far
        # share of SURVEY 8d's per-launch figure: (28 + 28) B of state and 2 x 4 (N + 1) B of images per active pixel.
        gm = np.asarray(group_ms, dtype=np.float64) if group_ms and len(group_ms[0]) else np.zeros((0, 0))
        dom_is_group = bool(gm.size and (gm > 0).any())
        dom_name, dom_pmc_file = "pm::sweep_kernel", "pmc_latest_sweep_kernel.json"
        dom_is_fused = bool(sched["group_from"] >= 0 and sched["group_fused"])
        if dom_is_fused:  # one launch per half-sweep: propagation per plane + accept replay + refinement
            dom_name, dom_pmc_file = "pm::sweep_group_kernel", "pmc_latest_sweep_group_kernel.json"
        ms_sweep_launch = ms_launch
        if dom_is_group:
            used = gm > 0
            ms_group_launch = float(gm[used].mean())
            ms_sweep_launch = float((hs - gm)[used].mean())  # the fused sweep launch of the same half-sweeps
            alg_half_sweep, ms_pair = alg, ms_launch
            alg = n_pix * (56 + 8 * (n_views + 1)) / 2.0
            alg_sweep = n_pix * (20 + 4 * (n_views + 1)) / 2.0
            ms_launch = ms_group_launch
            dom_name, dom_pmc_file = "pm::group_kernel", "pmc_latest_group_kernel.json"
        achieved = alg / (ms_launch * 1e-3) / 1e9
        # HBM-side bytes per sweep launch and VALU instruction counts are NOT measured in this run: they
        # come from the PMC passes of the same command (scripts/pmc_passes.sh -> profiles/pmc_latest.json),
        # FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, separate --pmc
        # passes; the object says so
        pj = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and args.config == "C" and args.scene == "smooth" and not args.colour:
            try:
                pj = json.load(open(pmc))
            except Exception:  # noqa: BLE001
                pj = None
        # ... the traffic of the dominant kernel from its own summary of the same passes
        pd = None
        pmc_dom = os.path.join(ROOT, "profiles", dom_pmc_file)
        if pj is not None and os.path.exists(pmc_dom):
            try:
                pd = json.load(open(pmc_dom))
            except Exception:  # noqa: BLE001
                pd = None
        # imported counters are flagged when they were not collected with the library this run loads
        # (by the hash of the kernel sources, which survives a rebuild; the binary's hash rides along)
        try:
            import glob
            import hashlib
            lib_now = hashlib.sha256(open(_exp_env("GIPUMA_HIP_LIB", None) or abi.LIB_PATH, "rb").read()).hexdigest()[:16]
            src_now = hashlib.sha256(b"".join(open(f, "rb").read() for f in sorted(
                glob.glob(os.path.join(ROOT, "gipuma_amd", "csrc", "*.h*"))))).hexdigest()[:16]
        except Exception:  # noqa: BLE001
            lib_now = src_now = None

        def stale(j):
            return None if not j else (j.get("_src_sha16") != src_now)
        # the imported traffic figure is only printed when it belongs to this build and this kernel time: counters
        # collected from other kernel sources, or at a launch duration more than 5 % away from the one measured in this
        # run, are refused (traffic = null, the reason in traffic_source.refused)
        traffic, traffic_refused = None, None
        if pd and "hbm_read_bytes_per_launch_x2corr" in pd:
            ms_pmc = pd.get("_kernel_ms_profiled_mean")
            if stale(pd):
                traffic_refused = "collected with other kernel sources (src sha16 %s, this run %s)" % (pd.get("_src_sha16"), src_now)
            elif ms_pmc and abs(ms_pmc - ms_launch) > 0.05 * ms_launch:
                traffic_refused = "collected at %.3f ms per launch, this run measures %.3f ms (> 5 %% apart)" % (ms_pmc, ms_launch)
            else:
                traffic = (pd["hbm_read_bytes_per_launch_x2corr"] + pd["hbm_write_bytes_per_launch"]) / 1e9
        imported_dom = {"measured_in_this_run": False, "file": "profiles/" + dom_pmc_file,
                        "collected_at_kernel_ms": pd.get("_kernel_ms_profiled_mean") if pd else None,
                        "collected_with_lib_sha16": pd.get("_lib_sha16") if pd else None, "commit": pd.get("_commit") if pd else None,
                        "collected_with_src_sha16": pd.get("_src_sha16") if pd else None,
                        "lib_sha16_of_this_run": lib_now, "src_sha16_of_this_run": src_now, "stale": stale(pd),
                        "refused": traffic_refused, "note": pd.get("_note") if pd else None}
        imported = {"measured_in_this_run": False, "file": "profiles/pmc_latest.json",
                    "collected_at_kernel_ms": pj.get("_kernel_ms_profiled_mean") if pj else None,
                    "collected_with_lib_sha16": pj.get("_lib_sha16") if pj else None, "commit": pj.get("_commit") if pj else None,
                    "collected_with_src_sha16": pj.get("_src_sha16") if pj else None,
                    "lib_sha16_of_this_run": lib_now, "src_sha16_of_this_run": src_now, "stale": stale(pj),
                    "note": pj.get("_note") if pj else None}
        box = gs.params.box_hsize
        S = ((box - 1) // 2 + 1) ** 2
        r_ref = 0
        dz = gs.params.max_disparity / 2.0
        while dz >= 0.01:
            r_ref += 1
            dz /= 10.0
        samples_per_frame = n_pix * (1 + iterations * (8 + r_ref)) * n_views * S
        cols_l = int(sched["cols_launches"])
        n_launch = 2 * iterations
        # leading half-sweeps whose propagation costs are pushed by pm::push_kernel (pm_push.h: boxes 11 / 15 / 25, gray, best-N <= 4)
        push_l = int(hs_pushed)  # (what the library reports for the timed solve: gipuma_hip_launch_times)
        push_l = max(0, min(push_l, n_launch))
        out = {
            "metric": "Mpixels/sec/GPU (1600x1200, 10 src views, 8 iters)"
                      if args.config == "C" and not args.colour and args.scene == "smooth"
                      else "Mpixels/sec/GPU (config %s%s%s)" % (args.config, ", colour" if args.colour else "",
                                                                 ", scene " + args.scene if args.scene != "smooth" else ""),
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
                                   "analytic textured surface (%s) rendered through DTU calibration "
                                   "(ref view %d); one reference view per GPU"
                                   % (args.config, gs.cols, gs.rows, n_views, box, iterations,
                                      gs.params.n_best, args.scene, ref_view),
                       "parallelism": "independent reference views, %d per step" % world,
                       "device_ms_total": float(np.mean(total_ms)),
                       "device_ms_init": float(np.mean(init_ms))},
            "ranks": [dict(i, value=p["value"]) for i, p in zip(ident, per_rank)],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "GB per launch (L2<->fabric incl. Infinity Cache hits)",
                         "traffic_source": imported_dom,
                         "kernel": ("pm::group_kernel: the propagation costs (close + far) of a half-sweep, one evaluation "
                                    "per PLANE instead of per (pixel, plane); %d of the %d half-sweeps of a view are this "
                                    "launch + one fused pm::sweep_kernel launch (accept replay + refinement); the %d before "
                                    "them: pm::sweep_cols_kernel x%d and pm::push_kernel x%d"
                                    % (int((gm[0] > 0).sum()), n_launch, first_plain, min(cols_l, n_launch), push_l))
                                   if dom_is_group else
                                   ("pm::sweep_group_kernel: one fused launch per half-sweep (one colour: close+far+refine; the "
                                    "propagation costs evaluated once per PLANE, pm_group.h), %d of the %d half-sweeps of a view "
                                    "(the %d before them: pm::sweep_cols_kernel x%d, and pm::push_kernel x%d, which evaluates "
                                    "the propagation costs of the first %d half-sweeps once per plane)"
                                    % (n_launch - first_plain, n_launch, first_plain, min(cols_l, n_launch), push_l, push_l))
                                   if dom_is_fused else
                                   ("pm::sweep_kernel: one fused launch per half-sweep (one colour: close+far+refine), "
                                    "%d of the %d half-sweeps of a view (the %d before them: pm::sweep_cols_kernel x%d, "
                                    "and pm::push_kernel x%d, which evaluates the propagation costs of the first %d "
                                    "half-sweeps once per plane)"
                                    % (n_launch - first_plain, n_launch, first_plain, min(cols_l, n_launch), push_l, push_l)),
                         "kernel_ms": ms_launch,
                         "kernel_ms_source": ("its own pair of HIP events on the library's stream (gipuma_hip_group_times), "
                                              "mean over the kernel's launches, this run") if dom_is_group else
                                             ("HIP events around each half-sweep on the library's stream "
                                              "(gipuma_hip_launch_times), mean over the kernel's launches, this run"),
                         "half_sweep_ms_mean_all": ms_half_sweep,
                         "half_sweep_ms": [float(x) for x in hs.mean(axis=0)] if hs.size else None,
                         "algorithmic_bytes_per_launch": alg,
                         "second_kernel": ({"kernel": "pm::sweep_kernel (accept replay + refinement of the same half-sweeps)",
                                            "kernel_ms": ms_sweep_launch, "algorithmic_bytes_per_launch": alg_sweep,
                                            "achieved": alg_sweep / (ms_sweep_launch * 1e-3) / 1e9, "unit": "GB/s"}
                                           if dom_is_group else None),
                         "half_sweep_pair": ({"algorithmic_bytes": alg_half_sweep, "ms": ms_pair,
                                              "achieved": alg_half_sweep / (ms_pair * 1e-3) / 1e9, "unit": "GB/s",
                                              "what": "SURVEY 8d's figure for the three reference kernels a half-sweep "
                                                      "replaces over the time of both launches"} if dom_is_group else None),
                         "group_kernel_ms": [float(x) for x in gm.mean(axis=0)] if gm.size else None,
                         "note": "compute/gather bound by construction (SURVEY F5): "
                                 "%.3g patch samples/s" % (samples_per_frame / (np.mean(total_ms) * 1e-3))},
            "quality": quality,
            "schedule": sched,
        }
        # The roof that actually binds: vector-ALU issue.  SQ_INSTS_VALU comes from
        # the committed PMC pass (see `source`), only the launch time is this run's.
        if pj and "SQ_INSTS_VALU" in pj:
            lane_ops = pj["SQ_INSTS_VALU"] * 64.0
            peak = 256 * 4 * 32 * 2.4e9
            out["roofline_valu"] = {
                "bound": "valu", "achieved": lane_ops / (ms_half_sweep * 1e-3) / 1e12, "peak": peak / 1e12,
                "unit": "T lane-instr/s", "frac": lane_ops / (ms_half_sweep * 1e-3) / peak,
                "per": "half-sweep of a view (all its kernels: sums over the dispatches / %d)" % n_launch,
                "valu_instr_per_window_load": pj["SQ_INSTS_VALU"] / max(1.0, pj.get("SQ_INSTS_VMEM_RD", 0.0)),
                "source": dict(imported, counter="rocprofv3 --pmc SQ_INSTS_VALU")}
        # ... and for the dominant kernel alone (its own summary of the same passes, this run's launch time)
        if pd and "SQ_INSTS_VALU" in pd:
            lane_ops = pd["SQ_INSTS_VALU"] * 64.0
            peak = 256 * 4 * 32 * 2.4e9
            out["roofline_valu_kernel"] = {
                "bound": "valu", "kernel": dom_name, "achieved": lane_ops / (ms_launch * 1e-3) / 1e12,
                "peak": peak / 1e12, "unit": "T lane-instr/s", "frac": lane_ops / (ms_launch * 1e-3) / peak,
                "valu_wave_instr_per_launch": pd["SQ_INSTS_VALU"], "kernel_ms": ms_launch,
                "source": dict(imported_dom, counter="rocprofv3 --pmc SQ_INSTS_VALU")}
        if world == 1 and not args.no_extras:
            from gipuma_amd.problem import GlobalState, runcuda
            # (a) the same workload with every exact work-reduction switched off (skip rules A/D/H, early
            #     termination): what a scene that defeats them would cost
            print("bench.py: extra leg -- exhaustive schedule", file=sys.stderr, flush=True)
            exp_before = os.environ.get("GIPUMA_HIP_EXPERIMENTS")
            os.environ["GIPUMA_HIP_EXPERIMENTS"] = "1"  # (the library reads its A/B switches only under this one)
            os.environ["GIPUMA_HIP_TUNE"] = str(64 | (1 << 23) | (1 << 25))
            try:
                with Session(gs) as s2:
                    s2.solve(timing=True)
                    t2 = s2.solve(timing=True)
                    final_ex = s2.get_state()
                # full-frame parity statement of the shipped schedule: its final maps and costs equal the
                # exhaustive schedule's bit for bit on every pixel (the exhaustive kernel is what the tests
                # teacher-force against the oracle at this size; tests/test_parity_gpu.py)
                out["quality"]["default_equals_exhaustive"] = bool(
                    np.array_equal(final_default[0].view(np.uint32), final_ex[0].view(np.uint32)) and
                    np.array_equal(final_default[1].view(np.uint32), final_ex[1].view(np.uint32)))
                del final_ex
                out["value_exhaustive"] = {"value": n_pix / (t2.ms_total * 1e-3) / 1e6, "unit": "Mpix/s",
                                           "ms_per_step": float(t2.ms_total),
                                           "what": "same frames, GIPUMA_HIP_TUNE=64|2^23|2^25: no skip rules, no early "
                                                   "termination (all 11 hypotheses x 640 samples per pixel and half-sweep)"}
            finally:
                del os.environ["GIPUMA_HIP_TUNE"]
                if exp_before is None:
                    del os.environ["GIPUMA_HIP_EXPERIMENTS"]
                else:
                    os.environ["GIPUMA_HIP_EXPERIMENTS"] = exp_before
            print("bench.py: extra leg -- fast mode", file=sys.stderr, flush=True)
            # (a') GIPUMA_HIP_FLAG_FAST: the tolerance-judged flavour of the same kernels (include/gipuma_hip.h) on the same
            #      frames, with its agreement with the exact mode's final maps of the timed solve -- the fraction of pixels
            #      inside the north_star tolerance (depth 1e-4 relative, unit normal 1e-3), the way the reference's own
            #      code is judged against the exact mode.  `value` stays the exact mode.
            with Session(gs, fast=True) as sf:
                sf.solve(timing=True)
                tf = sf.solve(timing=True)
                hs_fast, _ = sf.launch_times()
                qf = quality_of(sf, info)
                n4f, cf = sf.get_state()
            n4e, ce = final_default
            d_rel = np.abs(n4f[..., 3] - n4e[..., 3]) / np.maximum(np.abs(n4e[..., 3]), 1e-30)
            n_err = np.abs(n4f[..., :3] - n4e[..., :3]).max(-1)
            out["value_fast"] = {
                "value": n_pix / (tf.ms_total * 1e-3) / 1e6, "unit": "Mpix/s", "ms_per_step": float(tf.ms_total),
                "device_ms_init": float(tf.ms_init), "half_sweep_ms": [float(x) for x in hs_fast],
                "parity_vs_exact_mode": {
                    "frac_within_tolerance": float(((d_rel < 1e-4) & (n_err < 1e-3)).mean()),
                    "frac_bit_identical_planes": float((n4f.view(np.uint32) == n4e.view(np.uint32)).all(-1).mean()),
                    "tolerance": "depth 1e-4 relative, unit normal 1e-3 (BASELINE.json north_star)"},
                "quality": qf,
                "what": "same frames and schedule, session created with GIPUMA_HIP_FLAG_FAST: the numerical model of rounds 1-5 "
                        "(x * (1/z) for x / z, fused multiply-adds in the sample loop: pm_sample.h PM_MODEL 0) with v_rcp_f32 "
                        "without the Newton step, no proof of the reciprocal's range, and the nine divisions of the homography "
                        "by the plane offset as one reciprocal + a Markstein step each (pm_core.h PM_APPROX); judged by "
                        "tolerance, not bit-exact"}
            print("bench.py: extra leg -- reference-order mode", file=sys.stderr, flush=True)
            # (a'') GIPUMA_HIP_FLAG_LITERAL: the reference-order flavour -- bit-identical to the reference's own code
            #       (tests/test_literal_mode.py), a validation mode -- and how far the exact mode's maps are from it
            n4l = None
            if True:
                with Session(gs, literal=True) as sl:
                    sl.solve(timing=True)
                    tl = sl.solve(timing=True)
                    n4l, _ = sl.get_state()
                d_rel = np.abs(n4e[..., 3] - n4l[..., 3]) / np.maximum(np.abs(n4l[..., 3]), 1e-30)
                n_err = np.abs(n4e[..., :3] - n4l[..., :3]).max(-1)
                out["value_literal"] = {
                    "value": n_pix / (tl.ms_total * 1e-3) / 1e6, "unit": "Mpix/s", "ms_per_step": float(tl.ms_total),
                    "exact_mode_vs_this": {"frac_within_tolerance": float(((d_rel < 1e-4) & (n_err < 1e-3)).mean()),
                                           "frac_bit_identical_planes": float((n4e.view(np.uint32) == n4l.view(np.uint32)).all(-1).mean())},
                    "what": "session created with GIPUMA_HIP_FLAG_LITERAL: the per-sample arithmetic in the reference's own operation "
                            "order (every tap its own bilinear fetch with its own fraction, correctly rounded x / z, unfused "
                            "multiply-adds: pm_sample.h PM_MODEL 7) through the same push / column-per-lane / plane-keyed / bounded "
                            "kernels and schedule as the default mode; equals the reference's own code bit for bit "
                            "(tests/test_headline_parity.py, tests/test_literal_mode.py); `exact_mode_vs_this` is therefore the "
                            "default mode's distance from the reference on this frame"}
            del n4f, cf, n4e, ce, d_rel, n_err, n4l
            print("bench.py: extra leg -- scene with depth steps", file=sys.stderr, flush=True)
            # (b) a scene with depth discontinuities, an occluder and sensor noise
            if args.scene == "smooth" and not args.colour:  # (the stepped scene is rendered in gray only)
                gs3, info3 = synth.build_problem(args.config, ref_view=ref_view, device=dev, keep_on_device=True,
                                                 scene="steps", **over)
                gs3.desc.device_id = dev_index
                with Session(gs3) as s3:
                    s3.solve(timing=True)
                    t3 = s3.solve(timing=True)
                    q3 = quality_of(s3, info3)
                out["value_scene_steps"] = {"value": n_pix / (t3.ms_total * 1e-3) / 1e6, "unit": "Mpix/s",
                                            "ms_per_step": float(t3.ms_total), "quality": q3,
                                            "what": "same cameras and parameters, scene with +-30 mm depth steps, a raised "
                                                    "disc (occlusions) and sigma=2 sensor noise"}
                del gs3, info3
            print("bench.py: extra leg -- patchy scene", file=sys.stderr, flush=True)
            # (c) the smooth geometry with the texture taken away where real scans lose it: 30 % of the surface
            #     with a flat albedo (costs tie, bounds hold less often) and a periodic stripe band
            if args.scene == "smooth" and not args.colour:
                gs4, info4 = synth.build_problem(args.config, ref_view=ref_view, device=dev, keep_on_device=True,
                                                 scene="patchy", **over)
                gs4.desc.device_id = dev_index
                with Session(gs4) as s4:
                    s4.solve(timing=True)
                    t4 = s4.solve(timing=True)
                    q4 = quality_of(s4, info4)
                out["value_scene_patchy"] = {"value": n_pix / (t4.ms_total * 1e-3) / 1e6, "unit": "Mpix/s",
                                             "ms_per_step": float(t4.ms_total), "quality": q4,
                                             "what": "same cameras, parameters and geometry; about 30 % of the surface "
                                                     "with a flat albedo, a diagonal band with a periodic stripe texture "
                                                     "(period 6 px), sigma=1 sensor noise"}
                del gs4, info4
            print("bench.py: extra leg -- views in flight", file=sys.stderr, flush=True)
            # (d) throughput of a batch runner that keeps several reference views in flight on one GPU: one
            #     session (= one HIP stream) per view, whole solves enqueued back to back, one host wait
            #     at the end.  Launch tails of one view fill with workgroups of another, and kernels bound
            #     by different units (the push kernels by the vector L1, the sweep kernels by VALU issue)
            #     overlap.  Reported beside `value`, which stays one view at a time.
            if args.scene == "smooth":
                inflight = {}
                others = [v for v in synth.DTU_REF_VIEWS if v != ref_view]
                extra = []
                for v in others[:2]:
                    g_k, _ = synth.build_problem(args.config, ref_view=v, device=dev, keep_on_device=True,
                                                 colour=args.colour, scene=args.scene, **over)
                    g_k.desc.device_id = dev_index
                    extra.append(g_k)
                for kfl in (2, 3):
                    group = [gs] + extra[:kfl - 1]
                    sessions = [Session(g) for g in group]
                    try:
                        for ss in sessions:
                            ss.solve(timing=True)  # warm-up, one at a time
                        reps = 3
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(reps):
                            for ss in sessions:
                                ss.solve(timing=False)  # asynchronous: nothing waits for the stream
                        for ss in sessions:
                            ss.sync()
                        dt = time.perf_counter() - t1
                    finally:
                        for ss in sessions:
                            ss.close()
                    inflight["%d_views" % kfl] = {"value": kfl * reps * n_pix / dt / 1e6, "unit": "Mpix/s",
                                                  "ms_per_view": dt / (kfl * reps) * 1e3,
                                                  "ref_views": [ref_view] + others[:kfl - 1]}
                out["value_views_in_flight"] = dict(inflight, what="the same workload with 2 / 3 reference views "
                                                    "in flight on this GPU (one session and stream each, solves "
                                                    "enqueued asynchronously); `value` is one view at a time")
                del extra
            print("bench.py: extra leg -- host boundary", file=sys.stderr, flush=True)
            # (c) the boundary as the reference's main.cpp uses it: host images in, host planes out
            #     (upload + window packing + solve + download); reported, never `value`
            gs_host = GlobalState([im.cpu().numpy() for im in gs.images], gs.cameras, gs.selected,
                                  gs.params, seed=gs.desc.seed)
            t1 = time.perf_counter()
            runcuda(gs_host)
            host_s = time.perf_counter() - t1
            out["host_boundary"] = {"ms_per_view_host_in_host_out": host_s * 1e3,
                                    "value_pcie_inclusive": n_pix / host_s / 1e6, "unit": "Mpix/s"}
            # (e) BASELINE.json's other configurations and the two variants of the headline frame (the reference's default
            #     window, box 19: algorithmparameters.h:24-25; -color_processing: gipuma.cu:1965-1968), one view at a time,
            #     default mode: parity-test cases, not bench lines -- reported so that the driver's record holds them
            if args.config == "C" and args.scene == "smooth" and not args.colour and not over:
                print("bench.py: extra leg -- other configurations", file=sys.stderr, flush=True)
                del gs_host
                others_out = {}
                for key, cfg, kw in (("config_A", "A", {}), ("config_B", "B", {}), ("config_D", "D", {}),
                                     ("config_C_box19", "C", dict(blocksize=19)), ("config_C_colour", "C", dict(colour=True))):
                    try:  # (a secondary figure must not cost the run its headline line)
                        g_o, i_o = synth.build_problem(cfg, ref_view=ref_view, device=dev, keep_on_device=True, **kw)
                        g_o.desc.device_id = dev_index
                        with Session(g_o) as s_o:
                            s_o.solve(timing=True)
                            t_o = min(s_o.solve(timing=True).ms_total for _ in range(2))
                            q_o = quality_of(s_o, i_o)
                        others_out[key] = {"value": g_o.rows * g_o.cols / (t_o * 1e-3) / 1e6, "unit": "Mpix/s", "ms_per_view": float(t_o),
                                           "frame": "%dx%d" % (g_o.cols, g_o.rows), "source_views": len(g_o.selected),
                                           "box": int(g_o.params.box_hsize), "iterations": int(g_o.params.iterations),
                                           "frac_within_1pct_of_gt": q_o["frac_within_1pct_of_gt"]}
                        del g_o, i_o
                    except Exception as e:  # noqa: BLE001
                        others_out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                out["value_other_configs"] = dict(others_out, what="BASELINE.json configs A, B, D, and config C's frame with the "
                                                  "reference's default window (box 19) / with -color_processing (T = float4); "
                                                  "one reference view at a time, default mode, best of two solves")
        if world == 1 and not args.no_cpu_baseline:
            import tempfile
            from gipuma_amd.problem import save_problem
            with tempfile.TemporaryDirectory() as td:  # the same frames, copied back from HBM
                pf = os.path.join(td, "problem.npz")
                save_problem(pf, gs, info["P_matrices"], info["cam_scale"])
                out["cpu_baseline"] = cpu_baseline(pf, gs.rows, gs.cols, iterations)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
