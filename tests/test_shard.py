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


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
