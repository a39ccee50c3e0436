"""Row sharding of the gridcell batch across ranks and reassembly of the OH field.

Every gridcell is independent, so rank r of W predicts the contiguous
rows [row0, row0 + n) with a replicated booster and no data-path collective; the one
exchange step is an all-gather of the float32 OH shard (RCCL over xGMI on the GPUs,
`gloo` in the CPU tests).  Inside GEOS itself the export stays distributed and no
collective is needed at all."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def row_shard(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """(row0, nrows) of rank's shard: contiguous, sizes differing by at most one row."""
    base, rem = divmod(n_total, world)
    n = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, n


def all_gather_rows(out_full: torch.Tensor, out_local: torch.Tensor, n_total: int, world: int, even: bool) -> None:
    """Every rank ends with the whole field, shards in rank (= row) order."""
    if even:
        dist.all_gather_into_tensor(out_full, out_local)
        return
    # ragged shards: gather fixed-size slots, then compact
    slot = (n_total + world - 1) // world
    padded = out_local.new_zeros(slot)
    padded[: out_local.numel()] = out_local
    slots = out_local.new_empty(world * slot)
    dist.all_gather_into_tensor(slots, padded)
    for r in range(world):
        row0, n = row_shard(n_total, world, r)
        out_full[row0:row0 + n] = slots[r * slot: r * slot + n]
