"""Row-sharded data parallelism: contiguous row blocks per rank, SUM of the batch
normalisers W_f once per batch, ONE all-reduce per step, rank 0 alone adding the row-independent terms
(VFM_FLAG_NO_PRIOR_TERMS elsewhere), identical dense Adam on every rank.  Two forms of the step's exchange:

  "stats" (`step_stats` below)  the all-reduce carries the gradient's SUFFICIENT STATISTICS -- per entity
          (sum of grow_r, occurrences, A_e = sum grow_r * sumz_r) -- half the bytes of the gradient, cut in entity
          ranges so that the all-reduce of one range overlaps the kernels of its neighbours;
  "grads" (`VFM._step_unfused` with a process group)  the literal pattern: all-reduce of [gradients | loss], flat Adam.
  "rows"  (`step_rows` below)  the all-reduce carries the per-ROW gradient dloss/dpred_r of the whole batch and the six
          ELBO sums -- B_global + 8 doubles (0.8 MB at cfg4 where the statistics are 50 MB): the tables are replicated and
          eps is keyed on the entity id, so every rank can form every entity's sample itself and needs from the others
          only what their rows contributed to the likelihood.  Two fields, one sample.

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


def step_stats(model, plan, lr, step, group, eps=None, out_pred=None, mark=lambda name: None, wrec=None):
    """One multi-rank training step of `model` on its row shard `plan`, exchanging the gradient's sufficient statistics
    (per rank the reference's loop body, vfm-torch.py:351-370; across ranks one all-reduce of a flat fp32 buffer
    [records (sum grow, count, 0, 0 | A_e) | row sums | loss], in `model.exchange_chunks` entity ranges: the
    all-reduce of range k overlaps the epilogue + Adam kernel of range k-1).  Every rank then applies the same dense
    Adam update from the global statistics: replicas stay bit-identical.

    COMPACT exchange (`model.exchange_compact`): only the records of entities that some rank's shard contains are
    non-zero, and the ranks know that set (`global_touched`) -- what travels is [U_global records | row sums | loss]
    gathered out of the dense table, not all T records (cfg4 split 8 ways: 59 % of them; cfg5 at 8 x 2,048 rows: ~41 %).
    LAZY exact Adam on top of it (`model.exchange_lazy`; scaled moments): every row outside that set has a zero gradient on
    EVERY rank, so -- exactly as in the single-rank row-list form -- it is skipped and its zero-gradient updates are
    replayed (bitwise as the dense kernel applies them) right before a later batch needs it: per step a catch-up pass on
    the set's rows (before the forward reads them), then the apply stage over the same list; the last step of every moment
    period brings all rows up to date and runs the dense apply.  The same lists on every rank: replicas stay identical."""
    from . import ops
    ent, bia, scal = model._views(model._flat)
    loss3 = model._gflat[model._n_flat: model._n_flat + 3]
    sumz, grow, pred = model._step_buffers(plan.B)
    rl = ops.exchange_record_len(model.d)
    T, n = model.T, model.T * rl
    model._set_moment_form(model.scaled_moments)
    gids = gids32 = None
    if model.exchange_compact:
        g = global_touched(plan, group, T)
        if g.numel() <= model.exchange_compact_below * T:
            gids = g
            gids32 = plan.__dict__.get("_gids32")
            if gids32 is None or gids32.numel() != g.numel():
                gids32 = plan.__dict__["_gids32"] = g.to(torch.int32)
    # ---- lazy exact Adam over the exchanged set: the rows of the set are brought up to date BEFORE the forward reads them
    t = model._adam_t + 1
    k = (t - 1) % ops.MOMENT_PERIOD + 1
    if k == 1:
        model._lazy_lr = {}
    model._lazy_lr[k] = float(lr)
    lazy = gids is not None and model.exchange_lazy and model._moments_scaled and eps is None
    mv, vv = model._views(model._adam_m), model._views(model._adam_v)
    if model._lazy_dirty and (not lazy or model._lazy_kind != "list"):
        model.sync_lazy(t - 1)
    listed = False
    if lazy:
        if model._lazy_last is None:
            model._lazy_last = torch.empty(T, dtype=torch.int32, device=model.device)
        if not model._lazy_dirty:
            model._lazy_last.fill_(t - 1)
        if k < ops.MOMENT_PERIOD:
            ops.adam_catchup(ent, bia, mv, vv, model._lazy_last, gids32, model._lazy_lrs(k - 1), upto=t - 1, mark=t, wrec=wrec)
            model._lazy_dirty, model._lazy_kind, listed = True, "list", True
        else:       # the dense kernel rewrites every row's moments at a period end: every row up to date first
            ops.adam_catchup(ent, bia, mv, vv, model._lazy_last, None, model._lazy_lrs(k - 1), upto=t - 1, mark=t, wrec=wrec)
            model._lazy_dirty = False
        mark("catchup")
    st = ops.elbo_forward(plan, ent, bia, scal, model.inv_occ, eps=eps, seed=model.rng_seed, step=step, train=True,
                          flags=prior_terms_flag(dist.get_rank(group)), out_pred=out_pred if out_pred is not None else pred,
                          out_sumz=sumz, out_grow=grow, out_partials=model._partials, wrec=wrec)
    mark("fwd")
    xacc, xs, xl, bounds = model._xviews()
    works = []
    nch = len(bounds) - 1
    if gids is not None:
        Ug = gids.numel()
        sb = plan.__dict__.get("_gid_bounds")
        if sb is None or sb[0] != tuple(bounds):        # slot range of every entity range (once per plan)
            sb = plan.__dict__["_gid_bounds"] = (tuple(bounds), torch.searchsorted(
                gids, torch.tensor(bounds, dtype=torch.int64, device=gids.device)).tolist())
        sb = sb[1]
        if model._xcompact is None:
            model._xcompact = torch.empty(n + 8, dtype=torch.float32, device=model.device)
        cbuf = model._xcompact                              # [U_global records | row sums: 2 | pad: 2 | loss: 3 | pad]
        tail = cbuf[Ug * rl: Ug * rl + 8]
        model._exchanged_floats = Ug * rl + 8
    if listed:
        # compact AND lazy: the statistics are written straight into the compact buffer (record i <-> gids[i]), that is
        # all-reduced, and the apply stage reads it in place -- no dense statistics table in the step at all.  Ranges of
        # SLOTS here, and no more of them than keeps an apply launch at >= ~48 K rows (a 24 K-row launch measured 51 us
        # for 146 MB -- latency, not bandwidth; cfg4 split 8 ways: 2 ranges, cfg5: 4)
        nch = max(1, min(int(model.exchange_chunks), -(-Ug // 49152)))
        sb = [Ug * i // nch for i in range(nch + 1)]
        xs, xl = tail[0:2], tail[4:7]
        ops.elbo_finalize(st, scal, out=xl)           # this shard's loss terms (prior terms: rank 0)
        mark("finalize")
        ops.elbo_backward_acc_rows(plan, st, gids32, cbuf, xs)
        for c in range(nch):
            end = Ug * rl + 8 if c == nch - 1 else sb[c + 1] * rl      # the last range carries the row sums + the loss
            works.append(dist.all_reduce(cbuf[sb[c] * rl: end], group=group, async_op=True))
    else:
        ops.elbo_finalize(st, scal, out=xl)
        mark("finalize")
        if gids is None:
            for c in range(nch):
                lo, hi = bounds[c], bounds[c + 1]
                ops.elbo_backward_acc(plan, st, xacc, xs, lo, hi)
                end = model._xflat.numel() if hi == T else hi * rl
                works.append(dist.all_reduce(model._xflat[lo * rl: end], group=group, async_op=True))
            model._exchanged_floats = model._xflat.numel()
        else:       # compact exchange around the dense kernels (every row updated this step: period ends, exchange_lazy off)
            dense = xacc.view(T, rl)
            ops.elbo_backward_acc(plan, st, xacc, xs, 0, T)
            torch.index_select(dense, 0, gids, out=cbuf[: Ug * rl].view(Ug, rl))
            tail.copy_(model._xflat[n: n + 8])
            for c in range(nch):
                end = Ug * rl + 8 if c == nch - 1 else sb[c + 1] * rl
                works.append(dist.all_reduce(cbuf[sb[c] * rl: end], group=group, async_op=True))
    mark("bwd_acc")
    model._adam_t = t
    for c in range(nch):
        works[c].wait()
        if listed:
            ops.elbo_apply_adam_rows(plan, st, cbuf[sb[c] * rl:], xs, gids32[sb[c]: sb[c + 1]], ent, bia, scal, model.inv_occ,
                                     mv, vv, lr, t, move_scalars=c == nch - 1, compact=True, wrec=wrec)
            continue
        if gids is not None:
            if sb[c + 1] > sb[c]:
                dense.index_copy_(0, gids[sb[c]: sb[c + 1]], cbuf[sb[c] * rl: sb[c + 1] * rl].view(-1, rl))
            if c == nch - 1:
                model._xflat[n: n + 8].copy_(tail)
        ops.elbo_apply_adam(plan, st, xacc, xs, ent, bia, scal, model.inv_occ, mv, vv, lr, t,
                            e_lo=bounds[c], e_hi=bounds[c + 1], scaled_moments=model._moments_scaled)
        model._wrec_ok = False          # (the dense apply stage does not refresh the packed records: rebuilt by the next step)
    mark("exchange_apply_adam")
    loss3.copy_(xl)
    return loss3, st.pred


def gather_shards(x: torch.Tensor, y: torch.Tensor, group):
    """All ranks' row shards ([B_r, F] ids and [B_r] targets; B_r may differ and may be 0) concatenated in rank order on
    every rank.  Returns (X [sum B_r, F], Y [sum B_r], offset of this rank's rows)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = x.device
    n = torch.zeros(world, dtype=torch.int64, device=dev)
    n[rank] = x.shape[0]
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    counts = [int(v) for v in n.tolist()]
    nmax = max(max(counts), 1)
    xp = torch.zeros(nmax, x.shape[1], dtype=x.dtype, device=dev)
    yp = torch.zeros(nmax, dtype=torch.float32, device=dev)
    xp[: x.shape[0]] = x
    yp[: x.shape[0]] = y
    xs = [torch.empty_like(xp) for _ in range(world)]
    ys = [torch.empty_like(yp) for _ in range(world)]
    dist.all_gather(xs, xp, group=group)
    dist.all_gather(ys, yp, group=group)
    X = torch.cat([t[:c] for t, c in zip(xs, counts)]).contiguous()
    Y = torch.cat([t[:c] for t, c in zip(ys, counts)]).contiguous()
    return X, Y, sum(counts[:rank])


def global_plan(plan, group, model):
    """The plan of the WHOLE batch, on every rank: the ranks' row shards (contiguous blocks in rank order, `shard_rows`)
    all-gathered once per plan -- ids and targets, 20 bytes per row -- and indexed locally.  Returns (plan of all rows,
    offset of this rank's rows in it).  Collective: every rank calls it at the same point (the first rows-exchange step of
    the plan does); plans are parameter-free and reused every epoch (vfm-torch.py:121-122: no shuffling)."""
    hit = plan.__dict__.get("_gplan")
    if hit is not None and hit[0] is group:
        return hit[1], hit[2]
    from . import ops
    X, Y, off = gather_shards(plan.x, plan.y, group)
    if X.shape[0] != plan.B_global:
        raise RuntimeError(f"the ranks' shards hold {X.shape[0]} rows, the plan says B_global = {plan.B_global}")
    gp = ops.BatchPlan(plan.spec, X, Y, model.inv_occ, B_global=plan.B_global)       # (no group: W from all rows, locally)
    plan.__dict__["_gplan"] = (group, gp, off)
    return gp, off


def rows_supported(spec) -> bool:
    """The rows exchange runs the two-field record kernels (csrc/vfm_fwd2.hpp EPS_ZREC, k_bwd<PIPE>)."""
    from . import ops
    return ops.pipeline_supported(spec)


def step_rows(model, plan, lr, step, group, mark=lambda name: None, next_plan=None):
    """One multi-rank training step of `model` on its row shard `plan`, exchanging per-ROW gradients (per rank the
    reference's loop body, vfm-torch.py:351-370; across ranks ONE all-reduce of B_global + 8 doubles):
      1. every rank holds the records (w_e, KL share, z_e) of ALL entities of the global batch, made from its replica of
         the tables -- eps is keyed on (entity, step), so the replicas draw the same samples: written by the previous
         step's backward when that step was told this batch would follow (`next_plan`), else sampled now
         (vfm_sample_records_f32);
      2. forward of the rank's OWN rows as a gather of those records: pred_r, g_r = dloss/dpred_r, the ELBO sums of its rows;
      3. all-reduce of [g of all rows (zeros outside the own block) | the six sums]: afterwards every rank holds every row's g;
      4. loss + backward + dense Adam over the WHOLE batch from the records and the gathered g, on every rank
         (vfm_elbo_bwd_adam_pipe_f32) -- with `next_plan` in the look-ahead form (rows in neither this global batch nor
         the next are skipped and replayed later, exactly as in the single-rank pipelined step) and writing the next
         batch's records.  The replicas apply the same update to the same values.
    Against the statistics exchange: 0.8 MB instead of 50 MB per step at cfg4 -- the all-reduce stops being the step --
    for a backward that walks all B_global rows instead of B_global / N."""
    from . import ops
    ent, bia, scal = model._views(model._flat)
    loss3 = model._gflat[model._n_flat: model._n_flat + 3]
    gp, off = global_plan(plan, group, model)
    gnext = global_plan(next_plan, group, model)[0] if (next_plan is not None and next_plan.y is not None) else None
    Bg, B = gp.B, plan.B
    scaled = model.scaled_moments and not model.sparse_adam
    model._set_moment_form(scaled)
    model._adam_t += 1
    t = model._adam_t
    k = (t - 1) % ops.MOMENT_PERIOD + 1
    if k == 1:
        model._lazy_lr = {}
    model._lazy_lr[k] = float(lr)
    la = (gnext is not None and scaled and k < ops.MOMENT_PERIOD and model.pipeline_lookahead
          and model._lookahead_pays(gp, gnext))
    if model._lazy_dirty and not (la and model._lazy_kind == "la"):
        model.sync_lazy(t - 1)            # (the last step of a moment period / another step form before: every row current)
        mark("catchup")
    if model._zrec is None:
        rl = ops.record_len(model.d)
        model._zrec = [torch.zeros(model.T, rl, dtype=torch.float32, device=model.device) for _ in range(2)]
    cur, nxt = model._zrec
    fresh = not model._records_ready(gp, step)
    mv, vv = model._views(model._adam_m), model._views(model._adam_v)
    if la:
        if model._lazy_last is None:
            model._lazy_last = torch.empty(model.T, dtype=torch.int32, device=model.device)
        if not model._lazy_dirty:
            model._lazy_last.fill_(t - 1)
        if model._la_tab is None:
            model._la_tab = torch.zeros(2 * (ops.MOMENT_PERIOD + 1), dtype=torch.float32, device=model.device)
        ready = model._la_ready_for
        if model._lazy_dirty and not (ready is not None and ready[0] is gp and ready[1] == t - 1):
            ops.adam_catchup(ent, bia, mv, vv, model._lazy_last, gp.touched_ids(), model._lazy_lrs(k - 1), upto=t - 1, mark=t - 1)
            mark("catchup")
            fresh = True
    bufs = model.__dict__.setdefault("_rows_bufs", {})
    hit = bufs.get(Bg)
    if hit is None:
        dev = model.device
        hit = bufs[Bg] = (torch.zeros(Bg + 8, dtype=torch.float64, device=dev),            # the exchanged buffer
                          torch.zeros(Bg, dtype=torch.float32, device=dev),                # g of all rows
                          torch.zeros(_PARTIALS_LEN(), dtype=torch.float64, device=dev))   # the global sums as ONE slot
        hit[2][7] = 1.0                                    # (one "workgroup slot": the global sums)
        if len(bufs) > 8:
            bufs.pop(next(iter(bufs)))
    xbuf, g_all, pg = hit
    if fresh:
        ops.sample_records(gp, ent, bia, model.inv_occ, cur, model.rng_seed, step)
        mark("sample_rec")
    _, grow, pred = model._step_buffers(B)
    if B > 0:
        st = ops.elbo_forward_records(plan, cur, scal, model.rng_seed, step, pred, grow, model._partials)
        ops.elbo_finalize(st, scal, out=model.__dict__.setdefault("_rows_loss", torch.zeros(3, dtype=torch.float32, device=model.device)))
    mark("fwd")
    xbuf.zero_()
    if B > 0:
        xbuf[off: off + B] = grow[:B]
        xbuf[Bg: Bg + 6] = model._partials[:6]
    dist.all_reduce(xbuf, op=dist.ReduceOp.SUM, group=group)
    model._exchanged_floats = 2 * int(xbuf.numel())          # (doubles, counted in 4-byte units like the other forms)
    g_all.copy_(xbuf[:Bg])
    pg[8:14] = xbuf[Bg: Bg + 6]
    mark("exchange")
    p = ops._problem(gp.spec, Bg, Bg, gp.id_bits, model.rng_seed, step, 0)
    stg = ops.FwdState(pred, pg, None, g_all, p, None)
    ops.elbo_backward_adam_pipe(gp, stg, cur, nxt if gnext is not None else None, gnext, step + 1, ent, bia, scal, model.inv_occ,
                                mv, vv, lr, t, loss3, scaled_moments=scaled,
                                last_step=model._lazy_last if la else None, step_tab=model._la_tab if la else None,
                                listed=model.lookahead_list, la_next=gnext if la else None)
    mark("bwd_adam")
    if gnext is not None:
        model._zrec = [nxt, cur]
        model._zrec_for = (gnext, step + 1, model._flat._version)
    else:
        model._zrec_for = None
    if la:
        model._lazy_dirty, model._lazy_kind = True, "la"
        model._la_ready_for = (gnext, t)
    return loss3, pred[:B]


def _PARTIALS_LEN():
    from . import _lib
    return _lib.PARTIALS_LEN
