"""Row-sharded data parallelism helpers (the exchange itself is described in DESIGN.md section 6):
contiguous row blocks per rank, SUM of the batch normalisers W_f once per batch, ONE all-reduce of the flat
[gradients | loss] buffer per step, rank 0 alone adding the row-independent terms (VFM_FLAG_NO_PRIOR_TERMS elsewhere).
`backend="nccl"` is RCCL on ROCm; the CPU tests use gloo."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_rows(lo: int, hi: int, rank: int, world: int):
    """Contiguous row block [a, b) of batch [lo, hi) owned by `rank`."""
    n = hi - lo
    per = (n + world - 1) // world
    a = min(lo + rank * per, hi)
    return a, min(a + per, hi)


def sum_normalisers(W: torch.Tensor, group=None) -> torch.Tensor:
    """W_f summed over the ranks' shards (in place)."""
    if group is not None or (dist.is_initialized() and dist.get_world_size() > 1):
        dist.all_reduce(W, op=dist.ReduceOp.SUM, group=group)
    return W


def allreduce_flat(flat: torch.Tensor, group=None) -> torch.Tensor:
    """The one collective of a step: SUM of [gradients | loss] over ranks (in place)."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def prior_terms_flag(rank: int) -> int:
    """VFM_FLAG_NO_PRIOR_TERMS for every rank but rank 0."""
    return 0 if rank == 0 else 1
