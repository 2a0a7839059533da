"""Multi-GPU path: whole reference views per rank, no data-path collective.
Covered here with world_size-2 gloo processes on the CPU: the shard table partitions the view
list, and the control plane bench.py uses (barrier + MAX all-reduce of the wall time) works."""
import os
import subprocess
import sys
import textwrap

from gipuma_amd import synth
from gipuma_amd.shard import shard_table, views_for_rank

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_table_partitions_the_views():
    views = synth.DTU_REF_VIEWS
    for world in (1, 2, 4, 8):
        tab = shard_table(views, world)
        flat = [v for r in range(world) for v in tab[r]]
        assert sorted(flat) == sorted(views)
        assert max(len(v) for v in tab.values()) - min(len(v) for v in tab.values()) <= 1
    assert views_for_rank(views, 9, 16) == [views[1]]   # more ranks than views: wrap, never empty


def test_two_rank_gloo_run_covers_all_views_without_exchange():
    code = textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from gipuma_amd import synth
        from gipuma_amd.shard import views_for_rank
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mine = views_for_rank(synth.DTU_REF_VIEWS, rank, world)
        # each rank builds ITS OWN problem from its own view: nothing is received from a peer
        gs, info = synth.build_problem(synth.tiny_config(cols=48, rows=32, n_src=2), ref_view=mine[0])
        dist.barrier()
        t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py's max-over-ranks timing
        got = [None] * world
        dist.all_gather_object(got, (rank, mine, info["view_ids"][0]))
        if rank == 0:
            views = sorted(v for _, m, _ in got for v in m)
            assert views == sorted(synth.DTU_REF_VIEWS), views
            assert abs(float(t[0]) - 0.1 * world) < 1e-12
            assert all(ref == m[0] for _, m, ref in got)
            print("OK")
        dist.destroy_process_group()
    """ % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                          "29533", "-c", code] if False else
                         [sys.executable, "-c", _launcher(code)], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout


def _launcher(code):
    """spawn two ranks with plain subprocesses (no torchrun dependency on hostname resolution)"""
    return textwrap.dedent("""
        import os, subprocess, sys
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                       MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
            procs.append(subprocess.Popen([sys.executable, "-c", %r], env=env))
        rc = [p.wait() for p in procs]
        sys.exit(max(rc))
    """ % code)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (one process per
    GPU, the reference's unit: scripts/dtu_fast.sh:30-55); --dry-run exercises exactly that plumbing
    -- spawn, gloo rendezvous on 127.0.0.1, shard of reference views, gather, ONE JSON line -- on CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"]
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    assert len({r["ref_view"] for r in d["ranks"]}) == 2      # two different reference views
    assert len({r["device"] for r in d["ranks"]}) == 2        # bound to two different devices


def test_bench_eight_rank_dry_run():
    """the command shape the driver's scaling run has at its widest (`bench.py --gpus 8`, config E of BASELINE.json: eight
    reference views, one per GPU, /root/reference/scripts/dtu_fast.sh:30-55): eight processes, one rendezvous, eight distinct
    reference views = synth.DTU_REF_VIEWS, eight device ids, ONE JSON line"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["dry_run"] and d["scaling"] == "weak"
    assert [r["rank"] for r in d["ranks"]] == list(range(8))
    assert sorted(r["ref_view"] for r in d["ranks"]) == sorted(synth.DTU_REF_VIEWS[:8]) and len(synth.DTU_REF_VIEWS) >= 8
    assert len({r["device"] for r in d["ranks"]}) == 8


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


# ---- on hardware (the driver's GPU box has one MI355X; an 8-GPU node runs the same code with 8 devices) ----
import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_sessions_on_two_devices_from_one_process(hip):
    """config E in miniature: two DIFFERENT reference views solved concurrently by two sessions -- on devices
    0 and min(1, count - 1), i.e. two GPUs where there are two, the same GPU twice on the one-GPU box -- each
    with its own stream, enqueued asynchronously; each result equals the same view solved alone on device 0
    (the path has no cross-device state: the shard is the whole view)."""
    import numpy as np
    from gipuma_amd import abi
    from gipuma_amd.problem import Session, runcuda
    lib = abi.load_library()
    n_dev = lib.gipuma_hip_device_count()
    assert n_dev >= 1
    views = synth.DTU_REF_VIEWS[:2]
    cfg = synth.tiny_config(cols=160, rows=112, n_src=4, blocksize=15, iterations=3, n_best=3)
    problems = [synth.build_problem(cfg, ref_view=v)[0] for v in views]
    alone = [runcuda(g) for g in problems]
    for k, g in enumerate(problems):
        g.desc.device_id = min(k, n_dev - 1)
    sessions = [Session(g) for g in problems]
    try:
        for s in sessions:
            s.solve(timing=False)  # asynchronous: both solves are in flight before either is waited for
        got = [s.get_state() for s in sessions]
    finally:
        for s in sessions:
            s.close()
    for k in range(2):
        assert np.array_equal(got[k][0].view(np.uint32), alone[k][0].view(np.uint32)), "view %d norm4" % views[k]
        assert np.array_equal(got[k][1].view(np.uint32), alone[k][1].view(np.uint32)), "view %d cost" % views[k]
    assert not np.array_equal(got[0][0], got[1][0])  # (two different problems)


@pytest.mark.gpu
def test_bench_two_ranks_on_hardware(hip):
    """the real rank path of `bench.py --gpus 2` on the GPU box: two processes, gloo rendezvous on 127.0.0.1,
    device binding by LOCAL_RANK, one problem per rank from its own reference view, barrier + MAX over ranks,
    ONE JSON line with the whole-job value.  --oversubscribe lets both ranks share the box's single GPU (on an
    8-GPU node the driver runs it without: N ranks on N distinct devices, asserted by bench.py itself)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "1",
                        "--warmup", "0", "--config", "B", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    assert len({r["ref_view"] for r in d["ranks"]}) == 2
    # whole-job value = both ranks' pixels over the slower rank's time
    assert d["value"] <= sum(r["value"] for r in d["ranks"]) * 1.0001
    assert d["quality"]["frac_within_1pct_of_gt"] > 0.5


@pytest.mark.gpu
def test_bench_eight_ranks_on_hardware(hip):
    """the eight-rank path on the one-GPU box before an 8-GPU node ever runs it: eight processes, each with its own
    reference view's config-C problem resident (8 x 325 MB), rendezvous, barrier, MAX over ranks, one JSON line whose value
    is the eight views over the slowest rank's time"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--oversubscribe", "--steps", "1",
                        "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert [r["rank"] for r in d["ranks"]] == list(range(8))
    assert len({r["ref_view"] for r in d["ranks"]}) == 8
    assert d["value"] <= sum(r["value"] for r in d["ranks"]) * 1.0001
