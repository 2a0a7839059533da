"""Multi-GPU sharding of the path: whole reference views, no exchange.

Each reference view is an independent problem (own images, cameras and state); the reference runs
one process per reference view (scripts/dtu_fast.sh:30-55) on a single device (main.cpp:689-690).
Here rank r of `world` takes every world-th view of the job's list, so N GPUs work on N views at
once with no inter-GPU traffic at all -- RCCL/xGMI are not used (BASELINE.json north_star).
"""


def views_for_rank(ref_views, rank, world):
    """round-robin shard of the reference-view list; never empty (wraps around) so that a weak
    scaling run with more ranks than listed views still gives every GPU one view"""
    views = list(ref_views)
    if not views:
        raise ValueError("no reference views")
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    mine = views[rank::world]
    if not mine:
        mine = [views[rank % len(views)]]
    return mine


def shard_table(ref_views, world):
    """{rank: [views]} for logging / tests: a partition of ref_views when len >= world"""
    return {r: views_for_rank(ref_views, r, world) for r in range(world)}
