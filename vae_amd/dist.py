"""Row-sharded data parallelism: contiguous row blocks per rank, SUM of the batch
normalisers W_f once per batch, ONE all-reduce per step, rank 0 alone adding the row-independent terms
(VFM_FLAG_NO_PRIOR_TERMS elsewhere), identical dense Adam on every rank.  Two forms of the step's exchange:

  "stats" (`step_stats` below)  the all-reduce carries the gradient's SUFFICIENT STATISTICS -- per entity
          (sum of grow_r, occurrences, A_e = sum grow_r * sumz_r) -- half the bytes of the gradient, cut in entity
          ranges so that the all-reduce of one range overlaps the kernels of its neighbours;
  "grads" (`VFM._step_unfused` with a process group)  the literal pattern: all-reduce of [gradients | loss], flat Adam.

`backend="nccl"` is RCCL on ROCm; the CPU tests use gloo; tests/thread_ranks.py runs any number of in-process ranks."""
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


def step_stats(model, plan, lr, step, group, eps=None, out_pred=None, mark=lambda name: None):
    """One multi-rank training step of `model` on its row shard `plan`, exchanging the gradient's sufficient statistics
    (per rank the reference's loop body, vfm-torch.py:351-370; across ranks one all-reduce of a flat fp32 buffer
    [T records (sum grow, count, 0, 0 | A_e) | row sums | loss], in `model.exchange_chunks` entity ranges: the
    all-reduce of range k overlaps the statistics kernel of range k+1 and the epilogue + Adam kernel of range k-1).
    Every rank then applies the same dense Adam update from the global statistics: replicas stay bit-identical."""
    from . import ops
    ent, bia, scal = model._views(model._flat)
    loss3 = model._gflat[model._n_flat: model._n_flat + 3]
    sumz, grow, pred = model._step_buffers(plan.B)
    st = ops.elbo_forward(plan, ent, bia, scal, model.inv_occ, eps=eps, seed=model.rng_seed, step=step, train=True,
                          flags=prior_terms_flag(dist.get_rank(group)), out_pred=out_pred if out_pred is not None else pred,
                          out_sumz=sumz, out_grow=grow, out_partials=model._partials)
    mark("fwd")
    model._set_moment_form(model.scaled_moments)
    xacc, xs, xl, bounds = model._xviews()
    ops.elbo_finalize(st, scal, out=xl)               # this shard's loss terms (prior terms: rank 0)
    mark("finalize")
    works = []
    rl = ops.exchange_record_len(model.d)
    for k in range(len(bounds) - 1):
        lo, hi = bounds[k], bounds[k + 1]
        ops.elbo_backward_acc(plan, st, xacc, xs, lo, hi)
        end = model._xflat.numel() if hi == model.T else hi * rl      # the last range carries the row sums + the loss
        works.append(dist.all_reduce(model._xflat[lo * rl: end], group=group, async_op=True))
    mark("bwd_acc")
    model._adam_t += 1
    for k in range(len(bounds) - 1):
        works[k].wait()
        ops.elbo_apply_adam(plan, st, xacc, xs, ent, bia, scal, model.inv_occ, model._views(model._adam_m),
                            model._views(model._adam_v), lr, model._adam_t, e_lo=bounds[k], e_hi=bounds[k + 1],
                            scaled_moments=model._moments_scaled)
    mark("exchange_apply_adam")
    loss3.copy_(xl)
    return loss3, st.pred
