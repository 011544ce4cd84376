"""Row-sharded data parallelism for the VFM step.

Every batch is cut into contiguous row blocks, one per rank; both tables, `nb_occ` and the scalars
are replicated.  Because the ELBO is a sum over rows (plus terms that only depend on the
parameters), three exchanges reproduce the single-process step exactly:

  1. once per batch (parameter-free, cached over epochs): SUM of the column normalisers W_f
     -> `sum_normalisers`  (they sit inside a ratio n_g / W_g, so they must be global BEFORE the
     KL term and its gradient are formed);
  2. once per step: ONE all-reduce (SUM) of the flat fp32 buffer
     [g_entity | g_bias | g_scalars | loss, nll, kl]  -> `allreduce_flat`.
     Rank 0 alone adds the terms that do not depend on rows (KL of the global bias and its
     gradient: VFM_FLAG_NO_PRIOR_TERMS on the other ranks), so the sum counts them once;
  3. identical dense Adam on every rank keeps the replicas bit-identical (no broadcast needed).

eps is keyed on (seed, step, entity id), so every rank draws the same sample for an entity.
`backend="nccl"` is RCCL on ROCm (xGMI inside a node); the CPU tests use gloo.
"""
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
