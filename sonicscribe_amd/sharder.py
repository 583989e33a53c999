"""Multi-GPU sharding of independent segments: one engine replica per MI355X, one process per GPU,
segments split into contiguous ranges; no data-path collective (results are a few hundred int32 per segment)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of rank's share; earlier ranks take the remainder."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_results(local: List[List[int]], rank: int, world: int) -> List[List[int]]:
    """Collect per-segment token id lists on every rank, in global segment order (torch.distributed object
    gather over gloo / RCCL; control-plane only, outside any timed region)."""
    if world == 1:
        return local
    import torch.distributed as dist
    out: List = [None] * world
    dist.all_gather_object(out, local)
    return [ids for part in out for ids in part]
