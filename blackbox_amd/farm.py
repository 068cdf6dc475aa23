"""Frame farming: one process per GPU, frames are independent.

The reference hands one frame to each multiprocessing worker / Slurm job
(blackbox.py:363-379, blackbox_slurm_google.py:302-381) with no communication between
workers.  Here rank r of `world` takes every world-th file; there is no data-path
collective -- torch.distributed is only used by callers that want a barrier or to gather
the list of products.
"""
import os


def rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def shard(items, rank=None, world=None):
    """round-robin shard of a list: item k goes to rank k % world"""
    if rank is None or world is None:
        rank, world, _ = rank_world()
    items = list(items)
    return items[rank::world]


def gather_results(local_results, group=None):
    """all ranks' result lists on every rank (object gather; control plane only)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(local_results)
    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, list(local_results), group=group)
    merged = []
    for k in range(max(len(o) for o in out) if out else 0):
        for r in range(world):
            if k < len(out[r]):
                merged.append(out[r][k])
    return merged
