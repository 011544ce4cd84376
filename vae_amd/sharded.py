"""Entity-sharded multi-rank training step (opt-in: `VFM.exchange = "sharded"`).

The row-sharded default (vae_amd/dist.py) replicates the tables and all-reduces [T, d+4] statistics
every step: fine while the table is small next to the batch, hopeless when it is not (Criteo shape:
10^6 x 512 parameters, 2,048 rows per rank).  Here the ENTITIES are partitioned too -- rank o owns
e = o (mod N): only the owner runs Adam on a row, and only the rows a batch touches travel:

  plan time (once per batch; the loader does not shuffle):
      unique ids of my rows, grouped by owner -> slot numbers; my rows' ids rewritten as slots;
      every owner learns which of its entities each rank needs (all-to-all of id lists);
  every step:
      1. owner:   vfm_shard_sample_f32 on the requested ids -> records (w, 0,0,0 | z)          [kernel]
      2. all-to-all #1: records to the requesting ranks                                         [RCCL]
      3. consumer: forward on slots with VFM_FLAG_ZPRE (no RNG / KL: the owner did / does it)   [kernel]
                   vfm_elbo_bwd_acc_f32 over the slots -> statistics records                    [kernel]
      4. all-to-all #2: statistics back to the owners                                           [RCCL]
      5. owner:   vfm_records_add_f32 per source rank into its dense local records,             [kernel]
                  vfm_elbo_apply_adam_f32 (own_mod/own_rank) = epilogue + Adam on OWNED rows    [kernel]
                  + its share of the KL term
      6. one 4-float all-reduce (row sums for the scalar gradients, likelihood and KL terms),
         scalar Adam on every rank (the three scalars stay replicated).

Every rank keeps a full-size parameter buffer with global row numbering, but only the rows it owns
are up to date; `sync_params()` (an all-gather, called before predict / save) refreshes the rest.
Bytes per rank and step: 2 * U_rank * (d+4) * 4  (U_rank = unique entities of its rows) instead of the
[T, d+4] all-reduce, and Adam traffic / N.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops


class ShardedPlan:
    """Parameter-independent state of one batch shard in entity-sharded mode."""

    def __init__(self, spec: ops.Spec, x: torch.Tensor, y: torch.Tensor, inv_occ: torch.Tensor, B_global: int,
                 group, rank: int, world: int):
        if spec.n_samples != 1:
            raise NotImplementedError("entity-sharded exchange supports n_samples == 1 only (use exchange='grads')")
        dev = x.device
        self.rank, self.world, self.group = rank, world, group
        self.spec = spec
        T, N = spec.T, world
        x = x.to(torch.int64)
        if x.numel() and (int(x.min()) < 0 or int(x.max()) >= T):
            raise IndexError(f"entity id out of range [0,{T})")
        # --- batch normalisers with GLOBAL ids (parameter-free), summed over ranks
        base = ops.BatchPlan(spec, x.contiguous(), y, inv_occ, B_global=B_global, build_index=False,
                             validate=False, process_group=group)
        self.W = base.W
        self.y = base.y
        self.B, self.B_global = base.B, base.B_global
        # --- slots: unique ids of my rows, grouped by owner (then by id)
        uniq = torch.unique(x)
        key = (uniq % N) * T + uniq
        order = torch.argsort(key)
        self.slot_ids = uniq[order].contiguous()                     # [U] global id of each slot
        key_sorted = key[order].contiguous()
        self.U = int(self.slot_ids.numel())
        self.x_slots = torch.searchsorted(key_sorted, (x % N) * T + x).to(torch.int32).contiguous()
        need = torch.bincount(self.slot_ids % N, minlength=N)          # records I need from each owner
        # --- tell every owner which of its entities I need
        need_c = need.to(torch.int64)
        got_c = torch.empty_like(need_c)
        dist.all_to_all_single(got_c, need_c, group=group)
        self.need_counts = [int(v) for v in need_c.tolist()]          # per owner
        self.req_counts = [int(v) for v in got_c.tolist()]            # per requesting rank
        req = torch.empty(sum(self.req_counts), dtype=torch.int32, device=dev)
        dist.all_to_all_single(req, self.slot_ids.to(torch.int32), self.req_counts, self.need_counts, group=group)
        self.req_ids = req.contiguous()                               # global ids (all owned by me), by source
        self.req_local = (req // N).to(torch.int32).contiguous()      # local record index at the owner
        # gather index of the owner: for each owned (local) entity, where its records land in the
        # receive buffer of the statistics all-to-all (one per rank that has it in its rows)
        t_own = owned_rows(T, rank, N)
        rl64 = self.req_local.to(torch.int64)
        self.rec_pos = torch.argsort(rl64, stable=True).to(torch.int32).contiguous()
        rp = torch.zeros(t_own + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(rl64, minlength=t_own), 0, out=rp[1:])
        self.rec_ptr = rp.to(torch.int32).contiguous()
        # --- inverted index over slots (what vfm_elbo_bwd_acc_f32 walks)
        sspec = ops.Spec(T=max(self.U, 1), F=spec.F, d=spec.d, group_hi=tuple([max(self.U, 1)] * spec.F),
                         group_n=tuple([1.0] * spec.F), likelihood=spec.likelihood, nb_train=spec.nb_train,
                         link=spec.link)
        self.slot_plan = ops.BatchPlan(sspec, self.x_slots, self.y, None, B_global=B_global,
                                       build_index=True, validate=False)
        self.slot_plan.W = self.W
        rl = ops.exchange_record_len(spec.d)
        self.rl = rl
        self.R = int(self.req_ids.numel())
        # exchange buffers (records of `rl` floats)
        self.zsend = torch.empty(max(self.R, 1) * rl, dtype=torch.float32, device=dev)
        self.zrecv = torch.empty(max(self.U, 1) * rl, dtype=torch.float32, device=dev)
        self.srec = torch.zeros(max(self.U, 1) * rl, dtype=torch.float32, device=dev)
        self.rrecv = torch.empty(max(self.R, 1) * rl, dtype=torch.float32, device=dev)

    def splits(self, counts):
        return [c * self.rl for c in counts]


def owned_rows(T: int, rank: int, world: int) -> int:
    return (T - rank + world - 1) // world


def train_step_sharded(model, plan: ShardedPlan, lr: float, eps=None, out_pred=None, mark=None):
    """One step of vfm-torch.py:351-370 with entity-sharded tables (see the module docstring).
    Returns (loss3 [loss, nll, kl] -- global values, identical on all ranks --, pred of my rows)."""
    from . import _lib
    mark = mark or (lambda name: None)
    model._ensure_opt_state()
    spec, N, r = plan.spec, plan.world, plan.rank
    dev = model.device
    ent, bia, scal = model._views(model._flat)
    step = model.global_step
    model.global_step += 1
    e = eps if eps is not None else (None, None, None)
    mark("start")
    # 1. owners sample what was requested; 2. ship
    if plan.R:
        _lib.ops().shard_sample(plan.req_ids, ent, bia, e[0], e[1], plan.zsend, model.rng_seed & ops._I63, step,
                                spec.link_flag)
    dist.all_to_all_single(plan.zrecv[: plan.U * plan.rl], plan.zsend[: plan.R * plan.rl],
                           plan.splits(plan.need_counts), plan.splits(plan.req_counts), group=plan.group)
    mark("sample_a2a")
    # 3. forward on slots + statistics of my rows
    sumz, grow, pred = model._step_buffers(plan.B)
    if out_pred is not None:
        pred = out_pred
    partials = model._partials
    flags = 0 if r == 0 else ops.FLAG_NO_PRIOR_TERMS
    _lib.ops().elbo_fwd_zpre(plan.x_slots, plan.y, plan.zrecv, scal, e[2], pred, partials, sumz, grow,
                             spec.d, spec.nb_train, plan.B_global, spec.likelihood, flags | spec.link_flag,
                             model.rng_seed & ops._I63, step)
    st = ops.FwdState(pred, partials, sumz, grow, ops._problem(plan.slot_plan.spec, plan.B, plan.B_global, 32,
                                                             model.rng_seed, step, flags), None)
    model._ensure_shard_state(N, r)
    small = model._shard_small      # [sum g, alpha term, nll, KL(q(w0)) (rank 0), my KL share, -, -, - | scratch]
    loss_local = small[8:11]
    ops.elbo_finalize(st, scal, out=loss_local)     # nll of my rows (+ KL(q(w0)) on rank 0)
    mark("fwd")
    ops.elbo_backward_acc(plan.slot_plan, st, plan.srec, small[0:2])
    # 4. statistics back to the owners
    dist.all_to_all_single(plan.rrecv[: plan.R * plan.rl], plan.srec[: plan.U * plan.rl],
                           plan.splits(plan.req_counts), plan.splits(plan.need_counts), group=plan.group)
    mark("acc_a2a")
    # 5. owner: sum the sources' records, epilogue + Adam on owned rows (+ my share of the KL term)
    rec_index = None
    if model.shard_gather:                 # the apply kernel sums the sources' records itself (default)
        acc = plan.rrecv
        rec_index = (plan.rec_ptr, plan.rec_pos)
    else:                                  # dense local table of records, filled by vfm_records_add_f32
        acc = model._shard_acc
        acc.zero_()
        if model.shard_deterministic:      # one launch per source rank, plain adds
            off = 0
            for s in range(N):
                c = plan.req_counts[s]
                if c:
                    _lib.ops().records_add(acc, plan.req_local[off: off + c],
                                           plan.rrecv[off * plan.rl: (off + c) * plan.rl], spec.d, False)
                off += c
        elif plan.R:                       # one launch, float atomics
            _lib.ops().records_add(acc, plan.req_local, plan.rrecv, spec.d, True)
    model._adam_t += 1
    t_own = owned_rows(spec.T, r, N)
    st_own = ops.FwdState(pred, partials, sumz, grow,
                          ops._problem(spec, 0, plan.B_global, 64, model.rng_seed, step, 0), eps)
    own_plan = _OwnPlan(spec, plan.W)
    ops.elbo_apply_adam(own_plan, st_own, acc, small[0:2], ent, bia, scal, model.inv_occ,
                        model._views(model._adam_m), model._views(model._adam_v), lr, model._adam_t,
                        e_lo=0, e_hi=t_own, own_mod=N, own_rank=r, kl_ws=model._shard_klws, rec_index=rec_index,
                        scaled_moments=model._moments_scaled)
    # 6. one tiny all-reduce: [sum g, alpha term, nll, KL(q(w0)) (rank 0 only), my KL share]
    _lib.ops().shard_pack(small, loss_local, model._shard_klws)
    dist.all_reduce(small[0:8], group=plan.group)
    # scalar Adam (replicated, identical on every rank): an empty last chunk of the apply kernel
    if N > 1:
        ops.elbo_apply_adam(own_plan, st_own, acc, small[0:2], ent, bia, scal, model.inv_occ,
                            model._views(model._adam_m), model._views(model._adam_v), lr, model._adam_t,
                            e_lo=spec.T, e_hi=spec.T, own_mod=N, own_rank=r, rec_index=rec_index,
                            scaled_moments=model._moments_scaled)
    mark("apply_adam")
    loss3 = model._gflat[model._n_flat: model._n_flat + 3]
    _lib.ops().shard_loss(small, loss3)
    model._mark_stale(plan.group, "sharded")      # rows owned by other ranks are stale until sync_params()
    return loss3, pred


class _OwnPlan:
    """The two fields elbo_apply_adam reads from a plan."""

    def __init__(self, spec, W):
        self.spec, self.W = spec, W


def sync_params(model, group, rank: int, world: int, moments: bool = True):
    """Refresh the rows this rank does not own (all-gather of the owned rows) -- parameters and, when they
    exist, the Adam moments, so that a checkpoint taken on any rank is complete.  Plumbing only: strided
    copies + collectives."""
    T = model.T
    flats = [model._flat] + ([model._adam_m, model._adam_v] if (moments and model._adam_m is not None) else [])
    for flat in flats:
        ent, bia, _ = model._views(flat)
        for tab in (ent, bia):
            mx = owned_rows(T, 0, world)
            mine = torch.zeros(mx, tab.shape[1], dtype=tab.dtype, device=tab.device)
            own = tab[rank::world]
            mine[: own.shape[0]] = own
            out = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(out, mine, group=group)
            for s in range(world):
                n = owned_rows(T, s, world)
                tab[s::world] = out[s][:n]
