"""Embedding-dimension-sharded multi-rank training step (`VFM.exchange = "dims"`).

The row-sharded modes (vae_amd/dist.py, vae_amd/sharded.py) move gradient statistics of (nearly) every
table row through xGMI each step -- ~50-90 MB per rank at the ML-20M shape, against a 0.2 ms compute step.
This mode cuts the MODEL instead: rank r of N holds the coordinates [r*d/N, (r+1)*d/N) of both halves (mu, s)
of every entity row, sees ALL rows of the batch, and everything per coordinate -- sampling, KL, gradients,
Adam -- is local.  The only quantity that couples coordinates is the FM row value
    sum_f w_f + 1/2 sum_k [ (sum_f z_fk)^2 - sum_f z_fk^2 ]            (vfm-torch.py:244-245)
a sum over k, so the ONE exchange of a step is an all-reduce of B + 4096 floats (each rank's share of every row
value + its forward workgroups' shares of the KL term): 3.2 MB at N = 8, B = 800 K.  Per rank and step:

    1. forward on the local coordinates, VFM_FLAG_PARTIAL_PRED          -> shares, sumz slice, KL share   [k_fwd]
    2. all-reduce of the shares                                                                           [RCCL]
    3. vfm_elbo_lik_f32: predictions, likelihood terms, dloss/dpred                                       [k_lik]
    4. fused backward + Adam on the local coordinates                                                     [k_bwd]

Rank 0 also carries the first-order weights (bias_params); the three scalars are replicated and every rank
applies the same update to them.  The dense Adam traffic -- 80 % of the single-GPU step -- is divided by N
while the global batch grows by N, so the step time per rank FALLS with N (measured per-rank shapes at cfg3:
0.17 / 0.14 / 0.13 ms for N = 2 / 4 / 8 against 0.22 ms on one GPU).  Results equal the single-process step
on the same global batch up to the summation order of the row values.

`sync_params()` (an all-gather; before predict / save / checkpoint) assembles the full tables on every rank.
Needs d % (8 N) == 0 and n_samples == 1.
"""
from __future__ import annotations

import copy
import dataclasses

import torch
import torch.distributed as dist

from . import _lib, ops


def _round4(n):
    return (n + 3) // 4 * 4


def supported(d: int, world: int, n_samples: int = 1) -> bool:
    return world > 1 and n_samples == 1 and d % (8 * world) == 0


class DimsState:
    """This rank's slice of the model: flat buffers [entity slice (T x 2 dl) | bias (T x 2) | scalars]."""

    def __init__(self, model, group):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if not supported(model.d, self.world, model.n_samples):
            raise ValueError(f"exchange='dims' needs d % (8 * world) == 0 and n_samples == 1 "
                             f"(d={model.d}, world={self.world}, n_samples={model.n_samples})")
        self.T, self.d, self.dl = model.T, model.d, model.d // self.world
        self.off = self.rank * self.dl
        self.n_ent = self.T * 2 * self.dl
        self.off_bias = _round4(self.n_ent)
        self.off_scal = self.off_bias + _round4(self.T * 2)
        self.n_flat = self.off_scal + 4
        dev = model.device
        self.flat = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.partials_lik = torch.zeros_like(model._partials)
        self._bufs = {}
        self.scatter(model)

    def views(self, flat):
        return (flat[: self.n_ent].view(self.T, 2 * self.dl),
                flat[self.off_bias: self.off_bias + 2 * self.T].view(self.T, 2),
                flat[self.off_scal: self.off_scal + 3])

    def _cols(self, full):        # [T, 2d] -> this rank's [T, 2dl] = (mu slice | s slice)
        d, dl, o = self.d, self.dl, self.off
        return torch.cat([full[:, o: o + dl], full[:, d + o: d + o + dl]], dim=1)

    def scatter(self, model):
        """Take this rank's slice of the model's full parameter (and moment) buffers."""
        for mine, full in ((self.flat, model._flat), (self.m, model._adam_m), (self.v, model._adam_v)):
            ent, bia, scal = model._views(full)
            e, b, s = self.views(mine)
            e.copy_(self._cols(ent)); b.copy_(bia); s.copy_(scal)

    def gather(self, model, moments: bool = True):
        """Assemble the full tables (parameters and, unless `moments` is False, the Adam moments) on every rank:
        all-gather of the entity slices, first-order weights from rank 0, scalars as they are (replicated)."""
        d, dl = self.d, self.dl
        pairs = [(self.flat, model._flat)] + ([(self.m, model._adam_m), (self.v, model._adam_v)] if moments else [])
        for mine, full in pairs:
            ent, bia, scal = model._views(full)
            e, b, s = self.views(mine)
            parts = [torch.empty_like(e) for _ in range(self.world)]
            dist.all_gather(parts, e.contiguous(), group=self.group)
            for r, p in enumerate(parts):
                ent[:, r * dl: (r + 1) * dl] = p[:, :dl]
                ent[:, d + r * dl: d + (r + 1) * dl] = p[:, dl:]
            bb = b.clone()
            dist.broadcast(bb, src=dist.get_global_rank(self.group, 0), group=self.group)
            bia.copy_(bb)
            b.copy_(bb)
            scal.copy_(s)

    def step_buffers(self, B, dev):
        b = self._bufs.get(B)
        if b is None:
            b = (torch.empty(B, self.dl, dtype=torch.float32, device=dev),       # sumz slice
                 torch.empty(B, dtype=torch.float32, device=dev),                # grow
                 torch.zeros(B + _lib.MAX_FWD_BLOCKS, dtype=torch.float32, device=dev))   # row values + KL shares
            if len(self._bufs) > 4:
                self._bufs.clear()
            self._bufs[B] = b
        return b


def local_plan(plan: ops.BatchPlan, st: DimsState) -> ops.BatchPlan:
    """The plan with the kernels' view of the model: d = d / N coordinates starting at rank * d / N."""
    lp = getattr(plan, "_dims_plan", None)
    if lp is None or lp.spec.coord_off != st.off or lp.spec.d != st.dl:
        lp = copy.copy(plan)
        lp.spec = dataclasses.replace(plan.spec, d=st.dl, coord_off=st.off)
        plan._dims_plan = lp
    return lp


def train_step_dims(model, plan: ops.BatchPlan, lr: float, group, eps=None, mark=None):
    """One step of vfm-torch.py:351-370 with the embedding dimension cut over the ranks (module docstring).
    `plan` holds ALL rows of the batch (the same on every rank).  Returns (loss3 [loss, nll, kl] -- identical
    on all ranks --, pred [B])."""
    mark = mark or (lambda name: None)
    model._ensure_opt_state()
    st = model._dims_state(group)
    r = st.rank
    lp = local_plan(plan, st)
    ent, bia, scal = st.views(st.flat)
    step = model.global_step
    model.global_step += 1
    if eps is not None:                      # eps tables are local like the parameter tables
        eps = (eps[0][:, st.off: st.off + st.dl].contiguous(), eps[1], eps[2])
    sumz, grow, vals = st.step_buffers(plan.B, model.device)
    nobias = 0 if r == 0 else ops.FLAG_NO_BIAS
    mark("start")
    # 1. this rank's share of every row value; its forward workgroups' shares of the entity KL term land behind
    #    them in `vals` (KL(q(w0)) is added once, by every rank identically, when the backward forms the loss)
    ops.elbo_forward(lp, ent, bia, scal, model.inv_occ, eps=eps, seed=model.rng_seed, step=step, train=True,
                     flags=ops.FLAG_PARTIAL_PRED | ops.FLAG_NO_PRIOR_TERMS | nobias, out_pred=vals, out_sumz=sumz,
                     out_grow=grow, out_partials=model._partials)
    mark("fwd")
    # 2. the exchange of the step
    dist.all_reduce(vals, group=st.group)
    mark("allreduce_row_values")
    # 3. predictions, likelihood, dloss/dpred; partials as a full forward would have left them
    bw = ops.FwdState(vals, st.partials_lik, sumz, grow,
                      ops._problem(lp.spec, plan.B, plan.B_global, plan.id_bits, model.rng_seed, step, nobias), eps)
    ops.elbo_lik(bw, lp.y, scal)
    mark("lik")
    # 4. loss + gradients + dense Adam of the local coordinates (rank 0: also the first-order weights); the
    #    three scalars are replicated: every rank applies the same update
    scaled = model.scaled_moments and not model.sparse_adam
    if scaled != model._moments_scaled:
        ops.moments_rescale(st.m, st.v, model._adam_t, to_scaled=scaled)
        model._set_moment_form(scaled)
    model._adam_t += 1
    loss3 = model._gflat[model._n_flat: model._n_flat + 3]
    ops.elbo_backward_adam(lp, bw, ent, bia, scal, model.inv_occ, st.views(st.m), st.views(st.v), lr, model._adam_t,
                           loss_out=loss3, sparse=model.sparse_adam, scaled_moments=scaled)
    mark("bwd_adam")
    model._mark_stale(st.group, "dims")            # the full tables are stale until sync_params()
    return loss3, vals[: plan.B]
