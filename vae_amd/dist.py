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


def global_touched(plan, group, T: int) -> torch.Tensor:
    """Sorted ids (int64, on the device) of the entities that ANY rank's shard of this batch contains.  Plans are
    parameter-free and reused every epoch (the loader does not shuffle, vfm-torch.py:121-122), so the ranks agree on the
    set ONCE per plan: an all-gather of their sorted id lists (padded to the longest), then a sorted unique.  Collective:
    every rank calls it at the same point (the first statistics-exchange step of the plan does)."""
    hit = plan.__dict__.get("_gids")
    if hit is not None and hit[0] is group:
        return hit[1]
    mine = plan.touched_ids()
    dev = mine.device
    n = torch.tensor([mine.numel()], dtype=torch.int64, device=dev)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    nmax = max(int(n.item()), 1)
    pad = torch.full((nmax,), T, dtype=torch.int32, device=dev)        # (T: a sentinel past every id)
    pad[: mine.numel()] = mine
    parts = [torch.empty_like(pad) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, pad, group=group)
    g = torch.unique(torch.cat(parts))
    g = g[g < T].to(torch.int64).contiguous()
    plan.__dict__["_gids"] = (group, g)
    return g


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
    rl = ops.exchange_record_len(model.d)
    T, n = model.T, model.T * rl
    # COMPACT exchange: only the records of entities that some rank's shard contains are non-zero, and the ranks know
    # that set (global_touched) -- so what travels is [U_global records | row sums | loss] gathered out of the dense
    # table, not all T records (cfg4 split 8 ways: 59 % of them; cfg5 at 8 x 2,048 rows: ~41 %).  Still ONE all-reduce
    # per step, in the same ranges; the kernels keep working on the dense table (zeros stay zeros on every rank).
    gids = None
    if model.exchange_compact:
        g = global_touched(plan, group, T)
        if g.numel() <= model.exchange_compact_below * T:
            gids = g
    works = []
    if gids is None:
        for k in range(len(bounds) - 1):
            lo, hi = bounds[k], bounds[k + 1]
            ops.elbo_backward_acc(plan, st, xacc, xs, lo, hi)
            end = model._xflat.numel() if hi == T else hi * rl      # the last range carries the row sums + the loss
            works.append(dist.all_reduce(model._xflat[lo * rl: end], group=group, async_op=True))
        model._exchanged_floats = model._xflat.numel()
    else:
        Ug = gids.numel()
        sb = plan.__dict__.get("_gid_bounds")
        if sb is None or sb[0] != tuple(bounds):        # slot range of every entity range (once per plan)
            sb = plan.__dict__["_gid_bounds"] = (tuple(bounds), torch.searchsorted(
                gids, torch.tensor(bounds, dtype=torch.int64, device=gids.device)).tolist())
        sb = sb[1]
        if model._xcompact is None or model._xcompact.numel() < Ug * rl + 8:
            model._xcompact = torch.empty(n + 8, dtype=torch.float32, device=model.device)
        cbuf, dense = model._xcompact, xacc.view(T, rl)
        ops.elbo_backward_acc(plan, st, xacc, xs, 0, T)                      # the statistics kernels are short: one launch
        torch.index_select(dense, 0, gids, out=cbuf[: Ug * rl].view(Ug, rl))     # ... and one gather of the touched records
        cbuf[Ug * rl: Ug * rl + 8].copy_(model._xflat[n: n + 8])               # (row sums | loss)
        for k in range(len(bounds) - 1):
            end = Ug * rl + 8 if k == len(bounds) - 2 else sb[k + 1] * rl
            works.append(dist.all_reduce(cbuf[sb[k] * rl: end], group=group, async_op=True))
        model._exchanged_floats = Ug * rl + 8
    mark("bwd_acc")
    model._adam_t += 1
    for k in range(len(bounds) - 1):
        works[k].wait()
        if gids is not None:
            if sb[k + 1] > sb[k]:
                dense.index_copy_(0, gids[sb[k]: sb[k + 1]], cbuf[sb[k] * rl: sb[k + 1] * rl].view(-1, rl))
            if k == len(bounds) - 2:
                model._xflat[n: n + 8].copy_(cbuf[Ug * rl: Ug * rl + 8])
        ops.elbo_apply_adam(plan, st, xacc, xs, ent, bia, scal, model.inv_occ, model._views(model._adam_m),
                            model._views(model._adam_v), lr, model._adam_t, e_lo=bounds[k], e_hi=bounds[k + 1],
                            scaled_moments=model._moments_scaled)
    mark("exchange_apply_adam")
    loss3.copy_(xl)
    return loss3, st.pred
