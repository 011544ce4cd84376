"""`VFM` -- the reference's `CF` module + training / evaluation loop (vfm-torch.py:129-422) on
the HIP hot path.

What maps to what
  CF.__init__ (vfm-torch.py:133-164)      -> VFM.__init__  (same parameter names, shapes, init order)
  CF.forward  (:189-324) + loss (:359)    -> VFM.elbo(x, y)   differentiable (autograd.Function)
                                             VFM.forward(x)   inference, likelihood-like result
  loop body   (:351-370)                  -> VFM.train_step(plan)   fwd + bwd + Adam, no autograd
  training loop / eval (:337-422)         -> VFM.fit(...)
  eval block  (:402-417) + save_weights   -> VFM.predict(X)
All arithmetic of forward / backward / Adam runs in libvfm_hip.so; torch provides device memory,
streams, (optionally) autograd and torch.distributed.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .dist import shard_rows, allreduce_flat, prior_terms_flag


class LikelihoodResult:
    """Stand-in for the torch.distributions object `CF.forward` returns (vfm-torch.py:267-270):
    `.mean` and `.log_prob(y)` with batch shape [S, B] (S variational samples)."""

    def __init__(self, pred: torch.Tensor, output: str, alpha: torch.Tensor, link: str = "abs"):
        self.logits = pred.reshape(1, -1) if pred.dim() == 1 else pred
        self.output = output
        self._alpha = alpha
        self._link = link

    @property
    def mean(self):
        return self.logits if self.output == "reg" else torch.sigmoid(self.logits)

    def log_prob(self, y):
        y = y.to(self.logits.dtype).reshape(1, -1)
        if self.output == "reg":
            a = self._alpha.abs() if self._link == "abs" else torch.nn.functional.softplus(self._alpha)
            return -0.5 * a * (y - self.logits) ** 2 + 0.5 * torch.log(a) - 0.5 * math.log(2 * math.pi)
        return y * self.logits - torch.nn.functional.softplus(self.logits)


_CHECK_WREC = bool(int(__import__("os").environ.get("VFM_CHECK_WREC", "0")))


def _round4(n):
    return (n + 3) // 4 * 4


class VFM(nn.Module):
    """Variational Factorization Machine.

    Reference mode: `VFM(N, M, embedding_size, output)` -- two fields (user, item+N), the
    `<= N` group test of vfm-torch.py:316 reproduced (`quirk_le_N=True`).
    General mode: `VFM(field_sizes=[n_0, ..., n_{F-1}], ...)` -- F fields with consecutive id ranges.
    Build it under `torch.manual_seed(s)` to get the initial weights the reference gets under the
    same seed (alpha ~ U(0,1), then the two N(0,1) embedding tables, drawn on the CPU).
    """

    def __init__(self, N: Optional[int] = None, M: Optional[int] = None, embedding_size: int = 20,
                 output: str = "reg", field_sizes: Optional[Sequence[int]] = None,
                 quirk_le_N: bool = True, device="cuda", rng_seed: int = 0, sparse_adam: bool = False,
                 n_samples: int = 1, link: str = "abs"):
        """n_samples = the reference's global N_VARIATIONAL_SAMPLES (vfm-torch.py:19); link = its global
        LINK (:125-126): "abs" is the assignment in effect there, "softplus" the one it overwrites."""
        super().__init__()
        self.n_samples, self.link = int(n_samples), str(link)
        ops.Spec(T=1, F=1, d=1, group_hi=(1,), group_n=(1.0,), likelihood=0, n_samples=self.n_samples,
                 link=self.link)     # validates both
        if field_sizes is None:
            if N is None or M is None:
                raise ValueError("give N and M, or field_sizes")
            field_sizes = [int(N), int(M)]
        self.field_sizes = [int(s) for s in field_sizes]
        self.F = len(self.field_sizes)
        self.T = int(sum(self.field_sizes))
        self.N, self.M = self.field_sizes[0], (self.field_sizes[1] if self.F > 1 else 0)
        self.d = int(embedding_size)
        if output not in ("reg", "class"):
            raise ValueError("output must be 'reg' or 'class'")
        self.output = output
        hi = np.cumsum(self.field_sizes).astype(np.int64)
        if quirk_le_N and self.F == 2:
            hi[0] += 1                      # `uniq_entities <= N` (vfm-torch.py:316)
        self.group_hi = tuple(int(h) for h in hi)
        self.group_n = tuple(float(s) for s in self.field_sizes)
        self.rng_seed = int(rng_seed)
        self.global_step = 0
        self.fuse_adam = True      # single-rank train_step uses the fused backward+Adam kernel
        # OPT-IN, changes results: Adam only on the rows of the batch (no momentum drift of the
        # other rows); the reference's dense Adam (vfm-torch.py:339) moves every row every step
        self.sparse_adam = bool(sparse_adam)
        # fused single-rank step: keep the Adam moments in the scaled form (VFM_FLAG_SCALED_MOMENTS: rows the
        # batch does not touch do not write their moments back; same dense Adam up to fp32 rounding)
        self.scaled_moments = True
        self._moments_scaled = False       # which form _adam_m / _adam_v are in right now
        # Lazy EXACT dense Adam (single rank, fused step, scaled moments): rows a batch does not contain are skipped
        # and their zero-gradient updates replayed -- bitwise as the dense kernel applies them -- right before a
        # later batch (or predict / save) needs them.  Pays when a batch touches a small part of the table
        # (Criteo shape: 6 %); "auto" turns it on per batch below `lazy_threshold` touched rows.  The trajectory
        # is the dense one, bit for bit (tests/test_gpu_lazy_adam.py).
        # Software-pipelined step (single rank, fused, two fields, one sample, Philox eps): when the caller names the
        # NEXT batch (`train_step(plan, next_plan=...)`; fit() and bench.py do), the fused backward writes that
        # batch's sample records while the updated rows are in its registers, and the next forward is a gather of
        # those records -- no sampling, no sumz (include/vfm_hip.h: vfm_elbo_bwd_adam_pipe_f32).
        # It pays when a batch has many more ROWS than distinct entities (the forward's saving grows with the rows, the
        # backward's extra work -- sampling + writing one record per entity of the next batch -- with the entities):
        # ML-100K shape, any real data set at the reference's B = 100,000, B = 1 M rows at ML-20M shape.  With
        # uniform-random ids over 165 K entities and B = 100 K (U ~ B) it is a wash (measured: forward -20 us,
        # backward +46 us), so "auto" turns it on from `pipeline_ratio` rows per entity of the next batch -- and only
        # from `pipeline_min_d` coordinates on: the records save bytes in proportion to d, the extra sampling in the
        # backward does not shrink with it (800 K rows over 165 K entities: d = 16 0.143 vs 0.125 ms plain, d = 32 a
        # tie, d = 128 at B = 1 M 0.56 vs 0.78).
        self.pipeline = "auto"
        self.pipeline_ratio = 2.0
        self.pipeline_min_d = 20
        self.pipeline_min_T = 8192         # small tables: every list is cut in work items and the plain step is faster
                                           # (ML-100K shape: 0.056 vs 0.063 ms)
        # ... and, in its every-row form, only where a batch touches a good part of the table: that record backward visits
        # ALL T rows.  Rows in the data files' order (ML-20M's ratings.csv is sorted by user: 100,000 consecutive ratings are
        # ~900 users x ~10,000 items, 8 % of the table) have B = 8 U and ran 0.219 ms per step that way against 0.147-0.173 in
        # the plain look-ahead form (`bench.py --user-order --zipf 1.1`) ...
        self.pipeline_min_touch = 0.35
        # ... so the pipelined step has a look-ahead form of its own (k_bwd<PIPE, LA>: rows in neither this batch nor the
        # next are skipped and replayed later, bitwise the every-row pipelined step): 0.113-0.131 ms per step on that shape.
        self.pipeline_lookahead = True
        # plans a streamed loop (fit(stream_plans=True), bench.py --plans stream) keeps in hand: the build of batch t + depth
        # is enqueued while step t runs.  The host reads a build's counts when the plan is first named to a step (as
        # `next_plan`): with four, that build was enqueued three steps earlier and the host never waits for it
        self.plan_prefetch_depth = int(__import__("os").environ.get("VFM_PLAN_DEPTH", "4"))
        # ... and how many steps the host may queue ahead of the stream in such a loop (each holds its plan's buffers until the
        # stream gets there; without a bound the host, which no longer waits for anything, runs hundreds of steps ahead.
        # 4 / 16 read the same; 64 and more were measured SLOWER -- 0.34-0.36 instead of 0.23 ms per step: that many plans in
        # flight keep the allocator going to the device inside the loop)
        self.step_lead = int(__import__("os").environ.get("VFM_STEP_LEAD", "16"))
        self._zrec = None                  # two record tables [T, 4 + d]
        self._zrec_for = None              # (plan, philox step, version of _flat) the first table was prepared for
        self.lazy_adam = "auto"            # (row-list form: used when no next batch is named; with one, the look-ahead
                                           #  form below is preferred -- no separate replay pass)
        self.lazy_threshold = 0.35         # "auto": batches touching less than this share of the rows ...
        self.lazy_min_params = 1 << 24     # ... of a table with at least this many parameters (small tables: dense is cheap)
        self._lazy_last = None             # [T] int32: last Adam step applied to each row
        self._lazy_dirty = False           # some rows lag behind _adam_t
        self._lazy_lr = {}                 # {k: learning rate of the k-th step of the current moment period}
        self._lazy_kind = None             # which step form let rows lag: "list" (lazy_adam) or "la" (look-ahead)
        # Look-ahead form of the same idea for mid-range touch fractions (cfg3: 59 % of the rows per batch): when the
        # caller names the next batch, the fused dense step skips the rows that are in neither batch -- (1-f)^2 of the
        # table -- and rows of the next batch replay what they skipped inside that kernel (no separate pass).
        self.lookahead = True
        self.lookahead_list = True         # ... walking a list of those rows made once per pair of plans (else: scan all T)
        self.lookahead_min_skip = 0.05     # ... when at least this share of the rows would be skipped
        # Packed first-order records [T,4] = (mu_w, s_w, 1/occ, 0): a cache of bias_params + inv_occ the fused step's
        # forward reads with ONE 16-byte load per sampling task (vfm_problem_t.wrec); the fused backward forms and the
        # catch-up kernel refresh the rows they update, anything else that writes bias_params drops it (_wrec_ok).
        self.use_wrec = True
        self._wrec, self._wrec_ok = None, False
        self._la_tab = None                # device table of the period's per-step constants (written by the kernels)
        self._la_ready_for = None          # (plan, adam step): rows of that plan are current through that step
        # what the ranks exchange per step when a process group is given: "stats" = sufficient
        # statistics of the gradient ([T,d+2] floats), "grads" = the dense gradient ([T,2d+2] floats),
        # "rows" = every row's dloss/dpred and the six ELBO sums (B_global + 8 doubles; two fields, one sample): each rank
        # samples all the batch's entities from its replica and runs the whole batch's backward itself (dist.step_rows);
        # "auto" = the row-sharded batch + one all-reduce the north star names: "stats" (the all-reduce
        # carries the gradient's sufficient statistics, half the bytes of the gradient), "grads" with
        # n_samples > 1.
        self.exchange = "auto"
        self.exchange_chunks = 4
        # "stats": exchange only the records of entities some rank's shard contains (vae_amd/dist.py: the ranks agree on
        # that set once per plan) when they are at most this share of the table
        self.exchange_compact, self.exchange_compact_below = True, 0.85
        # ... and let the rows outside that set (zero gradient on every rank) wait, replayed bit for bit later: the
        # multi-rank form of the row-list lazy exact Adam (vae_amd/dist.py::step_stats)
        self.exchange_lazy = True
        self._xcompact, self._exchanged_floats = None, 0
        self._xflat = None

        # ---- parameters: same names / shapes / RNG order as CF.__init__ (vfm-torch.py:136-153)
        alpha = torch.rand(1)                                   # nn.init.uniform_ (:145)
        self.prec_global_bias_prior = nn.Parameter(torch.ones(1))    # unused, as in the reference (:139-143)
        self.prec_user_bias_prior = nn.Parameter(torch.ones(1))
        self.prec_item_bias_prior = nn.Parameter(torch.ones(1))
        self.prec_user_entity_prior = nn.Parameter(torch.ones(self.d))
        self.prec_item_entity_prior = nn.Parameter(torch.ones(self.d))
        bias = torch.randn(self.T, 2)                           # nn.Embedding default init (:152)
        entity = torch.randn(self.T, 2 * self.d)                # (:153)

        # one flat fp32 buffer [entity | bias | alpha, m0, s0 | pad]; the nn.Parameters are views
        # (a single fused Adam launch and a single all-reduce cover everything)
        self._n_ent, self._n_bias = self.T * 2 * self.d, self.T * 2
        self._off_bias = _round4(self._n_ent)
        self._off_scal = self._off_bias + _round4(self._n_bias)
        self._n_flat = self._off_scal + 4
        flat = torch.zeros(self._n_flat)
        flat[: self._n_ent] = entity.reshape(-1)
        flat[self._off_bias: self._off_bias + self._n_bias] = bias.reshape(-1)
        flat[self._off_scal: self._off_scal + 3] = torch.tensor([alpha.item(), 0.0, 1.0])
        self.alpha = nn.Parameter(torch.empty(1))
        self.global_bias_mean = nn.Parameter(torch.empty(1))
        self.global_bias_scale = nn.Parameter(torch.empty(1))
        self.bias_params = nn.Embedding(self.T, 2, _weight=torch.empty(self.T, 2))
        self.entity_params = nn.Embedding(self.T, 2 * self.d, _weight=torch.empty(self.T, 2 * self.d))
        self._flat = flat.to(device)
        self._tie()

        # training state set by set_training_data()
        self.nb_train = 1
        self.inv_occ = None
        self.nb_occ = None
        self.lr = None
        self._adam_m = self._adam_v = self._gflat = None
        self._adam_t = 0
        # epoch-averaged posterior means (save_weights, vfm-torch.py:179-185)
        self._n_saved = 0
        self._mean_flat = None
        self._last_flat = None
        self._all_preds_sum = None
        self._all_preds_n = 0

    # ------------------------------------------------------------------ parameter plumbing
    def _tie(self):
        f = self._flat
        self.entity_params.weight.data = f[: self._n_ent].view(self.T, 2 * self.d)
        self.bias_params.weight.data = f[self._off_bias: self._off_bias + self._n_bias].view(self.T, 2)
        self.alpha.data = f[self._off_scal: self._off_scal + 1]
        self.global_bias_mean.data = f[self._off_scal + 1: self._off_scal + 2]
        self.global_bias_scale.data = f[self._off_scal + 2: self._off_scal + 3]

    def _views(self, flat):
        """(entity [T,2d], bias [T,2], scalars [3]) views of a flat buffer; cached per buffer (the step path
        asks for the same few persistent buffers every step)."""
        cache = self.__dict__.setdefault("_view_cache", {})
        hit = cache.get(id(flat))
        if hit is None or hit[0] is not flat:
            if len(cache) > 16:
                cache.clear()
            hit = (flat, (flat[: self._n_ent].view(self.T, 2 * self.d),
                          flat[self._off_bias: self._off_bias + self._n_bias].view(self.T, 2),
                          flat[self._off_scal: self._off_scal + 3]))
            cache[id(flat)] = hit
        return hit[1]

    def _apply(self, fn, *a, **k):
        # keep the parameters tied to the flat buffer across .to()/.cuda(); the kernels are fp32 only
        if fn(torch.zeros(1, dtype=torch.float32, device=self._flat.device)).dtype != torch.float32:
            raise TypeError("VFM computes in fp32 only (like the reference); dtype conversions are not supported")
        if getattr(self, "_lazy_dirty", False):
            self.sync_lazy()
        with torch.no_grad():
            cur = torch.cat([self.entity_params.weight.reshape(-1),
                             torch.zeros(self._off_bias - self._n_ent, device=self._flat.device),
                             self.bias_params.weight.reshape(-1),
                             torch.zeros(self._off_scal - self._off_bias - self._n_bias, device=self._flat.device),
                             self.alpha, self.global_bias_mean, self.global_bias_scale,
                             torch.zeros(1, device=self._flat.device)])
        out = super()._apply(fn, *a, **k)
        self._flat = fn(cur)
        self._tie()
        for name in ("inv_occ", "nb_occ", "_adam_m", "_adam_v", "_gflat", "_mean_flat", "_last_flat", "_lazy_last",
                     "_gout", "_partials"):
            t = getattr(self, name, None)
            if t is not None:
                setattr(self, name, fn(t))
        # buffers derived from the old device's tensors: rebuilt on demand (plans built before the move hold
        # ids on the old device and must be rebuilt by the caller)
        self.__dict__.pop("_view_cache", None)
        self._xflat = None
        self._la_tab = self._la_ready_for = None          # (look-ahead step table: no row lags after sync_lazy above)
        self._wrec, self._wrec_ok = None, False
        self._xcompact = None
        self._zrec = self._zrec_for = None                # (sample records of the pipelined step: re-made on demand)
        if getattr(self, "_state_bufs", None) is not None:
            self._state_bufs = {}
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        """Loaded weights are CURRENT by definition: whatever the lazy / look-ahead step forms still owed the old rows
        (skipped zero-gradient updates) is dropped, not replayed on top of the restored values; derived caches
        (sample records of the pipelined step, the packed first-order records) are re-made on demand."""
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self._tie()
        self._forget_derived_state()
        return out

    def _forget_derived_state(self):
        """The parameters were written from outside the step kernels (load_state_dict, a checkpoint, a direct write the
        caller announces with `params_changed()`): no row lags any more and nothing derived from the old values is valid."""
        self._lazy_dirty, self._lazy_kind, self._la_ready_for, self._zrec_for = False, None, None, None
        self._wrec_ok = False

    def params_changed(self):
        """Call after writing `entity_params.weight` / `bias_params.weight` / the scalars directly (e.g. through `.data`).
        Between steps of the lazy step forms the tables LAG (rows a batch skipped have not had their zero-gradient
        updates yet): read them through `state_dict()` / `sync_lazy()` first, which bring every row up to date."""
        self._forget_derived_state()

    # ------------------------------------------------------------------ checkpoint / resume
    def training_state_dict(self):
        """Everything needed to resume training bit-for-bit: parameters (state_dict), Adam moments and
        step count, the Philox step counter, the epoch-averaged posterior means of save_weights().  (Every rank of a
        multi-rank run holds the full replicated state: a checkpoint taken by one rank alone is complete.)"""
        self.sync_lazy()
        opt = None
        if self._adam_m is not None:
            opt = {"m": self._adam_m.detach().cpu().clone(), "v": self._adam_v.detach().cpu().clone(),
                   "t": int(self._adam_t), "scaled_form": bool(self._moments_scaled)}   # (buffers stored as they are)
        snap = None
        if self._n_saved:
            snap = {"n": int(self._n_saved), "mean": self._mean_flat.detach().cpu().clone(),
                    "last": self._last_flat.detach().cpu().clone()}
        return {"model": {k: v.detach().cpu().clone() for k, v in self.state_dict().items()},
                "adam": opt, "global_step": int(self.global_step), "rng_seed": int(self.rng_seed),
                "lr": self.lr, "snapshots": snap}

    def load_training_state_dict(self, state):
        self.load_state_dict(state["model"])
        self.global_step, self.rng_seed, self.lr = int(state["global_step"]), int(state["rng_seed"]), state["lr"]
        if state.get("adam") is not None:
            self._ensure_opt_state()
            self._adam_m.copy_(state["adam"]["m"].to(self.device))
            self._adam_v.copy_(state["adam"]["v"].to(self.device))
            self._adam_t = int(state["adam"]["t"])
            self._moments_scaled = bool(state["adam"].get("scaled_form", False))
            self._lazy_lr = {}             # (no row lags: the next lazy step stamps every entry of _lazy_last before reading one)
        self._forget_derived_state()          # (with or without optimiser state: the restored rows are current)
        if state.get("snapshots") is not None:
            self._n_saved = int(state["snapshots"]["n"])
            self._mean_flat = state["snapshots"]["mean"].to(self.device).clone()
            self._last_flat = state["snapshots"]["last"].to(self.device).clone()

    @property
    def device(self):
        return self._flat.device

    def spec(self) -> ops.Spec:
        lik = _lib.LIK_NORMAL if self.output == "reg" else _lib.LIK_BERNOULLI
        return ops.Spec(T=self.T, F=self.F, d=self.d, group_hi=self.group_hi, group_n=self.group_n,
                        likelihood=lik, nb_train=int(self.nb_train), n_samples=self.n_samples, link=self.link)

    # ------------------------------------------------------------------ data-dependent state
    def set_training_data(self, X_train: torch.Tensor, nb_train: Optional[int] = None,
                          nb_occ: Optional[torch.Tensor] = None):
        """nb_occ = bincount(X_train.flatten()) (vfm-torch.py:89) and nb_train (:91)."""
        X_train = torch.as_tensor(X_train)
        self.nb_train = int(nb_train if nb_train is not None else X_train.shape[0])
        if nb_occ is None:
            nb_occ = torch.bincount(X_train.reshape(-1).to(self.device).to(torch.int64), minlength=self.T)
        self.nb_occ = torch.as_tensor(nb_occ).to(self.device).to(torch.int64).contiguous()
        if self.nb_occ.numel() != self.T:
            raise ValueError("nb_occ must have one entry per entity")
        self.inv_occ = ops.inv_occ_from_counts(self.nb_occ)
        self._wrec_ok = False

    def plan(self, x, y=None, B_global=None, build_index=True, process_group=None, defer_readback=False, stream=None) -> ops.BatchPlan:
        x = torch.as_tensor(x).to(self.device)
        if x.dtype not in (torch.int32, torch.int64):
            x = x.to(torch.int64)
        x = x.contiguous()
        if y is not None:
            y = torch.as_tensor(y).to(self.device)
            if self.inv_occ is None:
                raise RuntimeError("call set_training_data() before building training plans")
        if process_group is not None and y is not None:
            self._resolve_exchange(process_group)
        return ops.BatchPlan(self.spec(), x, y, self.inv_occ, B_global=B_global,
                             build_index=build_index and y is not None, process_group=process_group,
                             defer_readback=defer_readback, stream=stream)

    def _resolve_exchange(self, process_group):
        """exchange == "auto" -> the pattern for this model / world size (see __init__)."""
        if self.n_samples > 1:
            self.exchange = "grads"      # the statistics / rows exchanges carry one sample
        if self.exchange == "auto":
            self.exchange = "stats"          # row-sharded batch, ONE (chunk-overlapped) all-reduce per step
        return self.exchange

    # ------------------------------------------------------------------ forward surfaces
    def _scalars(self):
        return self._flat[self._off_scal: self._off_scal + 3]

    def elbo(self, x=None, y=None, plan: Optional[ops.BatchPlan] = None, eps=None):
        """Differentiable ELBO loss of one batch: `-log_prob(y).mean()*nb_train + kl`
        (vfm-torch.py:353-359).  Returns (loss[1], pred[B], detail[3] = loss, nll, kl)."""
        self._fresh_params()
        self._wrec_ok = False     # (the caller's optimiser writes the parameters: the packed records go stale)
        if plan is None:
            plan = self.plan(x, y)
        step = self.global_step
        self.global_step += 1
        return ops.ElboFunction.apply(self.entity_params.weight, self.bias_params.weight, self.alpha,
                                      self.global_bias_mean, self.global_bias_scale, plan, self.inv_occ,
                                      eps, self.rng_seed, step)

    @torch.no_grad()
    def forward(self, x, eps=None, sample=True):
        """Inference surface of `CF.forward(x)` (vfm-torch.py:189-324): returns
        (likelihood, last_logits, mean_logits, kl_term).  `likelihood.mean` is the prediction from
        a fresh posterior sample; last/mean logits are the deterministic predictions from the last /
        epoch-averaged posterior means once `save_weights()` has run (None before).  The KL branch
        is not evaluated on inference inputs (kl_term is None)."""
        self._fresh_params()
        plan = self.plan(x, None) if not isinstance(x, ops.BatchPlan) else x
        ent, bia, scal = self._views(self._flat)
        step = self.global_step
        self.global_step += 1
        st = ops.elbo_forward(plan if sample else _single_sample(plan), ent, bia, scal, None, eps=eps,
                              seed=self.rng_seed, step=step, train=False, flags=0 if sample else ops.FLAG_EPS_ZERO)
        lik = LikelihoodResult(st.pred, self.output, self.alpha.detach(), self.link)
        last = mean = None
        if self._n_saved > 0:
            last = self._mean_logits(plan, self._last_flat)
            mean = self._mean_logits(plan, self._mean_flat)
        return lik, last, mean, None

    def _mean_logits(self, plan, flat):
        ent, bia, scal = self._views(flat)
        st = ops.elbo_forward(_single_sample(plan), ent, bia, scal, None, train=False, flags=ops.FLAG_EPS_ZERO)
        return st.pred

    def _fresh_params(self):
        """Bring the rows the lazy Adam forms skipped up to date (called before anything reads the parameters)."""
        self.sync_lazy()

    def sync_lazy(self, upto=None):
        """Lazy Adam modes: replay the skipped updates of every row (all rows are then at step `upto` = _adam_t, exactly
        where dense Adam has them).  A no-op otherwise.  Called before anything reads the parameters."""
        if self._lazy_dirty:
            upto = self._adam_t if upto is None else upto
            ent, bia, _ = self._views(self._flat)
            k = (upto - 1) % ops.MOMENT_PERIOD + 1 if upto > 0 else 0
            ops.adam_catchup(ent, bia, self._views(self._adam_m), self._views(self._adam_v), self._lazy_last, None,
                             self._lazy_lrs(k), upto=upto, mark=upto, wrec=self._wrec if self._wrec_ok else None)
            self._lazy_dirty, self._lazy_kind, self._la_ready_for = False, None, None

    def _lazy_lrs(self, kmax):
        """Learning rates of steps 1..kmax of the current moment period (steps that were not fused steps are never
        replayed -- every row was brought up to date before them -- so their entries are placeholders)."""
        return [self._lazy_lr.get(k, 0.0) for k in range(1, kmax + 1)]

    def state_dict(self, *a, **k):
        self.sync_lazy()
        return super().state_dict(*a, **k)

    def _lookahead_pays(self, plan, next_plan) -> bool:
        if not self.lookahead:
            return False
        return (1.0 - plan.U / self.T) * (1.0 - next_plan.U / self.T) >= self.lookahead_min_skip

    def _will_pipeline(self, plan, next_plan) -> bool:
        """The rule train_step applies for the software-pipelined step form in steady state."""
        if not self.pipeline or self.sparse_adam or not ops.pipeline_supported(plan.spec):
            return False
        if self.pipeline is True:
            return True
        return (self.d >= self.pipeline_min_d and self.T >= self.pipeline_min_T
                and next_plan.B >= self.pipeline_ratio * next_plan.U and plan.B >= self.pipeline_ratio * plan.U
                and (min(plan.U, next_plan.U) >= self.pipeline_min_touch * self.T
                     or (self.pipeline_lookahead and self._lookahead_pays(plan, next_plan))))

    def _use_lazy(self, plan) -> bool:
        if self.lazy_adam is True:
            return True
        if self.lazy_adam != "auto" or plan.B == 0:
            return False
        return self._n_ent >= self.lazy_min_params and plan.U < self.lazy_threshold * self.T

    @torch.no_grad()
    def save_weights(self):
        self._fresh_params()
        self._save_weights()

    @torch.no_grad()
    def _save_weights(self):
        """Snapshot the posterior means and update their running average over epochs
        (vfm-torch.py:179-185; the reference keeps every snapshot and re-averages, the running
        mean is the same quantity)."""
        cur = self._flat.detach().clone()
        self._last_flat = cur
        self._n_saved += 1
        if self._mean_flat is None:
            self._mean_flat = cur.clone()
        else:
            self._mean_flat += (cur - self._mean_flat) / self._n_saved

    # ------------------------------------------------------------------ training
    def _ensure_opt_state(self):
        if self._adam_m is None:
            self._adam_m = torch.zeros_like(self._flat)
            self._adam_v = torch.zeros_like(self._flat)
            self._gflat = torch.zeros(self._n_flat + 4, dtype=torch.float32, device=self.device)
            self._adam_t = 0
            self._gout = torch.ones(1, dtype=torch.float32, device=self.device)
            self._partials = torch.zeros(_lib.PARTIALS_LEN, dtype=torch.float64, device=self.device)
            self._state_bufs = {}

    def _set_moment_form(self, scaled: bool):
        """Bring _adam_m / _adam_v into the scaled / plain form (a no-op unless the form changes)."""
        if self._adam_m is not None and scaled != self._moments_scaled:
            if self._lazy_dirty:          # lagging rows replay with the constants of the form they were skipped in
                self.sync_lazy()
            ops.moments_rescale(self._adam_m, self._adam_v, self._adam_t, to_scaled=scaled)
        self._moments_scaled = bool(scaled)

    def _xviews(self):
        """Flat exchange buffer: T records [sum grow, count, 0, 0 | A_e (d)] then [row sums: 2 | pad: 2 |
        loss: 3 | pad]; plus the entity boundaries of the exchange chunks."""
        rl = ops.exchange_record_len(self.d)
        n = self.T * rl
        if self._xflat is None:
            self._xflat = torch.zeros(n + 8, dtype=torch.float32, device=self.device)
        f = self._xflat
        k = max(1, min(int(self.exchange_chunks), self.T))
        bounds = [self.T * i // k for i in range(k + 1)]
        return f[:n], f[n: n + 2], f[n + 4: n + 7], bounds

    def _step_buffers(self, B):
        """Persistent per-step training state (sumz [S*B,d], grow [B], pred [B] or [S,B]) -- no allocator
        traffic inside the step."""
        b = self._state_bufs.get(B)
        if b is None:
            S = self.n_samples
            b = (torch.empty(S * B, self.d, dtype=torch.float32, device=self.device),
                 torch.empty(B, dtype=torch.float32, device=self.device),
                 torch.empty((B,) if S == 1 else (S, B), dtype=torch.float32, device=self.device))
            if len(self._state_bufs) > 4:
                self._state_bufs.clear()
            self._state_bufs[B] = b
        return b

    def plan_async(self, x, y, pair_with: Optional[ops.BatchPlan] = None, fork: bool = True) -> ops.BatchPlan:
        """The plan of a batch, built on the model's side stream while whatever is already enqueued on the current stream
        runs (`ops.PlanStream`): returns at once; the first `train_step` that uses the plan makes its stream wait for the
        build.  pair_with: the plan of the batch BEFORE this one -- the look-ahead step's row list of the pair is made on
        the side stream too.  fork=False: x and y are long-lived (nothing on the current stream is still writing them), so the
        build need not wait for the steps already enqueued there.  Single rank (a multi-rank plan sums W over the ranks: a
        collective, built in line)."""
        # (the pair's row list only where the caller asks for it: the streamed loop lets the look-ahead kernel classify the
        #  table rows itself -- +4 us of kernel time against 35 us of host time and three more launches per step)
        pair = pair_with if (pair_with is not None and self.lookahead and self.lookahead_list and self.n_samples == 1) else None
        return self.plan_streams().build(lambda st: self.plan(x, y, defer_readback=True, stream=st), pair_with=pair, fork=fork)

    def plan_streams(self) -> "ops.PlanStream":
        """The side streams of this model's plan builds (made on first use)."""
        if getattr(self, "_plan_stream", None) is None or self._plan_stream.device != self.device:
            self._plan_stream = ops.PlanStream(self.device)
        return self._plan_stream

    def train_step(self, plan: ops.BatchPlan, lr: Optional[float] = None, eps=None, out_pred=None,
                   process_group=None, adam: bool = True, mark=None, fused: Optional[bool] = None,
                   next_plan: Optional[ops.BatchPlan] = None, prefetch=None):
        """One iteration of vfm-torch.py:351-370 without autograd: forward, loss, backward, dense
        Adam (betas (0.9, 0.999), eps 1e-8).  Everything is enqueued on the current stream; nothing
        synchronises with the host.  Returns (loss3 device tensor [loss, nll, kl], pred [B]).
        prefetch = (x, y[, fork[, pair_with]]) of a LATER batch: its plan is built on the side stream while this step runs
        (`plan_async`; enqueued after this step's own launches) and waits in `self.prefetched` for the caller to take.
        pair_with: the plan of the batch right before it, or None (default) -- with it the look-ahead row list of that pair is
        made too; without, a look-ahead step on such plans classifies the table rows itself (the scan form: what the
        streamed loop takes).  Prefetch TWO batches beyond `next_plan` (as `fit(stream_plans=True)` does): a step reads
        the size of its next batch's index on the host, so a plan enqueued during the previous step would be waited for."""
        self._ensure_opt_state()
        self._sharing = prefetch is not None        # (plan builds run beside this step: its row kernels leave them slots)
        if prefetch is not None:
            try:
                return self._train_step(plan, lr, eps, out_pred, process_group, adam, mark, fused, next_plan)
            finally:
                # the host stays within `plan_prefetch_depth` steps of this stream: builds with fork=False do not wait for the
                # steps in flight, so nothing else would stop the host from queueing hundreds of steps -- every one of them
                # holding its plan's buffers until the stream gets there (measured: +1.1 MB of reserved memory per step and a
                # device allocation -- a stall -- every other step: 0.33 instead of 0.23 ms per step)
                dq = self.__dict__.setdefault("_steps_in_flight", [])
                ev = torch.cuda.Event()
                ev.record()
                dq.append(ev)
                if len(dq) > self.step_lead:
                    dq.pop(0).synchronize()
                pw = prefetch[3] if len(prefetch) > 3 else None
                self.prefetched = self.plan_async(prefetch[0], prefetch[1], pair_with=pw, fork=prefetch[2] if len(prefetch) > 2 else True)
        return self._train_step(plan, lr, eps, out_pred, process_group, adam, mark, fused, next_plan)

    def _train_step(self, plan, lr, eps, out_pred, process_group, adam, mark, fused, next_plan):
        plan.use_on_current()
        if next_plan is not None:
            next_plan.use_on_current()
        lr = self.lr if lr is None else lr
        if lr is None:
            raise RuntimeError("learning rate not set (fit() uses 1/(1+nb_train//batch_size))")
        stats_step = process_group is not None and adam and self.exchange == "stats" and self.n_samples == 1
        rows_step = process_group is not None and adam and self.exchange == "rows"
        if process_group is not None and not (stats_step or rows_step):
            self.sync_lazy()          # these multi-rank steps update every row they own (the statistics / rows steps keep their own lazy state)
        if process_group is not None and self.exchange not in ("stats", "grads", "rows"):
            raise ValueError(f"exchange = {self.exchange!r}: the multi-rank step forms are 'stats', 'grads', 'rows' ('auto' is "
                             "resolved when the plan is built with the process group)")
        step = self.global_step
        self.global_step += 1
        mark = mark or (lambda name: None)      # bench.py records HIP events at these points
        mark("start")
        if fused is None:
            fused = self.fuse_adam
        fused = fused and adam and process_group is None
        # the packed first-order records: kept coherent by the fused single-rank step and the multi-rank statistics step
        # (every other step form drops them)
        wrec = self._wrec_for_step((fused or stats_step) and eps is None and self.n_samples == 1 and not self.sparse_adam)
        if fused:
            return self._step_fused(plan, next_plan, lr, step, eps, out_pred, mark, wrec)
        if process_group is not None and adam and self.exchange == "stats" and self.n_samples == 1:
            from .dist import step_stats
            return step_stats(self, plan, lr, step, process_group, eps, out_pred, mark, wrec)
        if process_group is not None and adam and self.exchange == "rows":
            from .dist import step_rows, rows_supported
            if not rows_supported(plan.spec) or eps is not None or out_pred is not None or self.sparse_adam:
                raise ValueError('exchange = "rows": two fields, one sample, d % 4 == 0, d <= 512, Philox eps, dense Adam')
            return step_rows(self, plan, lr, step, process_group, mark, next_plan=next_plan)
        if self._lazy_dirty:
            self.sync_lazy()
        return self._step_unfused(plan, lr, step, eps, out_pred, adam, mark, process_group)

    def _step_unfused(self, plan, lr, step, eps, out_pred, adam, mark, process_group=None):
        """forward, loss, backward (the dense gradient written), [ONE all-reduce of the flat [gradients | loss] buffer --
        the literal north-star exchange, `exchange = "grads"`], [flat dense Adam]: the single-rank step with separate
        kernels, and the multi-rank step in its gradient form."""
        ent, bia, scal = self._views(self._flat)
        g_ent, g_bias, g_scal = self._views(self._gflat)
        loss3 = self._gflat[self._n_flat: self._n_flat + 3]
        rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        sumz, grow, pred = self._step_buffers(plan.B)
        st = ops.elbo_forward(plan, ent, bia, scal, self.inv_occ, eps=eps, seed=self.rng_seed, step=step,
                              train=True, flags=prior_terms_flag(rank), out_pred=out_pred if out_pred is not None else pred,
                              out_sumz=sumz, out_grow=grow, out_partials=self._partials)
        mark("fwd")
        self._set_moment_form(False)          # (the flat k_adam: plain moments)
        ops.elbo_finalize(st, scal, out=loss3)
        mark("finalize")
        ops.elbo_backward(plan, st, ent, bia, scal, self.inv_occ, self._gout, g_ent, g_bias, g_scal)
        mark("bwd")
        if process_group is not None:
            # the ONE collective of the step: [g_entity | g_bias | g_scalars | loss] summed over ranks
            allreduce_flat(self._gflat, process_group)
            mark("allreduce")
        if adam:
            self._adam_t += 1
            ops.adam_step(self._flat, self._gflat, self._adam_m, self._adam_v, lr, self._adam_t)
            mark("adam")
        return loss3, st.pred

    def _step_fused(self, plan, next_plan, lr, step, eps, out_pred, mark, wrec):
        """The single-rank step with loss + backward + dense Adam in ONE kernel, in the form the state allows: look-ahead
        (rows of this batch and of the next), row-list lazy (+ catch-up pass), software-pipelined, or plain dense.  All
        forms are the same dense Adam trajectory (the lazy ones bit for bit: tests/test_gpu_lazy_adam.py,
        test_gpu_state_machine.py)."""
        ent, bia, scal = self._views(self._flat)
        loss3 = self._gflat[self._n_flat: self._n_flat + 3]
        sumz, grow, pred = self._step_buffers(plan.B)
        flags = ops.FLAG_SHARE_GPU if getattr(self, "_sharing", False) else 0
        lazy = rows = None
        la = False
        scaled = self.scaled_moments and not self.sparse_adam
        self._set_moment_form(scaled)
        self._adam_t += 1
        k = (self._adam_t - 1) % ops.MOMENT_PERIOD + 1            # position of this step in its moment period
        # look-ahead form: visit only the rows of this batch and of the next (named by the caller).  Preferred to the
        # row-list form below wherever both apply: no separate replay pass (B = 5,000 at cfg3: 0.079 vs 0.099 ms;
        # Criteo shape: 0.361 vs 0.364)
        la = (scaled and k < ops.MOMENT_PERIOD and eps is None and self.n_samples == 1 and self.lazy_adam is not True
              and next_plan is not None and next_plan.y is not None and next_plan.spec.T == self.T
              and self._lookahead_pays(plan, next_plan))
        lazy = scaled and not la and self.n_samples >= 1 and self._use_lazy(plan)
        if k == 1:
            self._lazy_lr = {}
        self._lazy_lr[k] = float(lr)
        pipe_la = la and self.pipeline_lookahead      # the pipelined step in its look-ahead form (rows may go on lagging)
        if (self.pipeline and not lazy and (not self._lazy_dirty or self._lazy_kind == "la") and eps is None
                and not self.sparse_adam
                and out_pred is None and ops.pipeline_supported(plan.spec)
                and (self.pipeline is True or (self.d >= self.pipeline_min_d and self.T >= self.pipeline_min_T))):
            ready = self._records_ready(plan, step)
            nxt = next_plan
            # (a batch that covers a small part of the table pipelines only in the look-ahead form: the every-row record
            #  backward would visit all T rows for it)
            if nxt is not None and self.pipeline == "auto" and (nxt.B < self.pipeline_ratio * nxt.U or (
                    nxt.U < self.pipeline_min_touch * self.T and not pipe_la)):
                nxt = None                # too few rows per entity / too small a part of the table for the records to pay
            if ready or (nxt is not None and (self.pipeline is True or (plan.B >= self.pipeline_ratio * plan.U and (
                    plan.U >= self.pipeline_min_touch * self.T or pipe_la)))):
                if self._lazy_dirty and not pipe_la:      # (the last step of a moment period, or no next batch named: the
                    self.sync_lazy(self._adam_t - 1)      #  every-row form -- all rows up to date first; the records stay valid)
                    mark("catchup")
                return self._train_step_pipelined(plan, nxt, lr, step, scaled, mark, wrec,
                                                  la_next=next_plan if pipe_la else None)
        kind = "list" if lazy else ("la" if la else None)
        if self._lazy_dirty and kind != self._lazy_kind:
            self.sync_lazy(self._adam_t - 1)      # another step form than the one that let rows lag: all rows current first
        if kind is not None:
            if self._lazy_last is None:
                self._lazy_last = torch.empty(self.T, dtype=torch.int32, device=self.device)
            if not self._lazy_dirty:     # no row lags: every row is at the step before this one, however it got there
                self._lazy_last.fill_(self._adam_t - 1)
            mv, vv = self._views(self._adam_m), self._views(self._adam_v)
            if lazy and k < ops.MOMENT_PERIOD:
                # rows of this batch: replay what they skipped, they get step _adam_t below; the others wait
                ops.adam_catchup(ent, bia, mv, vv, self._lazy_last, plan.touched_ids(), self._lazy_lrs(k - 1),
                                 upto=self._adam_t - 1, mark=self._adam_t, wrec=wrec)
                rows, self._lazy_dirty, self._lazy_kind = "touched", True, "list"
                mark("catchup")
            elif la:
                if self._la_tab is None:
                    self._la_tab = torch.zeros(2 * (ops.MOMENT_PERIOD + 1), dtype=torch.float32, device=self.device)
                ready = self._la_ready_for
                if self._lazy_dirty and not (ready is not None and ready[0] is plan and ready[1] == self._adam_t - 1):
                    # this batch was not the one announced to the previous step: bring its rows up to date now
                    ops.adam_catchup(ent, bia, mv, vv, self._lazy_last, plan.touched_ids(), self._lazy_lrs(k - 1),
                                     upto=self._adam_t - 1, mark=self._adam_t - 1, wrec=wrec)
                    mark("catchup")
            else:
                # last step of a moment period (the dense kernel rewrites every row's moments): every row up to
                # date first, then the ordinary dense step
                ops.adam_catchup(ent, bia, mv, vv, self._lazy_last, None, self._lazy_lrs(k - 1),
                                 upto=self._adam_t - 1, mark=self._adam_t, wrec=wrec)
                self._lazy_dirty = False
                mark("catchup")
        elif self._lazy_dirty:
            self.sync_lazy(self._adam_t - 1)

        # the launches of the fused step: forward, then loss + backward + dense Adam in ONE kernel (the gradient rows
        # never reach HBM)
        st = ops.elbo_forward(plan, ent, bia, scal, self.inv_occ, eps=eps, seed=self.rng_seed, step=step,
                              train=True, flags=flags, out_pred=out_pred if out_pred is not None else pred,
                              out_sumz=sumz, out_grow=grow, out_partials=self._partials, wrec=wrec)
        mark("fwd")
        if la:      # ... visiting only the rows of this batch and of the next one (look-ahead lazy exact Adam)
            ops.elbo_backward_adam_lookahead(plan, st, next_plan, ent, bia, scal, self.inv_occ,
                                             self._views(self._adam_m), self._views(self._adam_v), lr, self._adam_t,
                                             loss3, self._lazy_last, self._la_tab,
                                             listed=self.lookahead_list and not getattr(self, "_sharing", False), wrec=wrec)
        else:
            ops.elbo_backward_adam(plan, st, ent, bia, scal, self.inv_occ, self._views(self._adam_m),
                                   self._views(self._adam_v), lr, self._adam_t, loss_out=loss3,
                                   sparse=self.sparse_adam, scaled_moments=scaled, rows=rows, wrec=wrec)
        mark("bwd_adam")
        if la:
            self._lazy_dirty, self._lazy_kind = True, "la"
            self._la_ready_for = (next_plan, self._adam_t)
        return loss3, st.pred

    def _wrec_for_step(self, keep: bool):
        """The packed first-order records for a step that keeps them coherent (None: this step form does not -- they are
        dropped and rebuilt by the next step that does)."""
        if not (keep and self.use_wrec and self.d % 4 == 0 and 16 <= self.d <= 512):
            self._wrec_ok = False
            return None
        _, bia, _ = self._views(self._flat)
        if self._wrec is None:
            self._wrec = torch.empty(self.T, 4, dtype=torch.float32, device=self.device)
            self._wrec_ok = False
        if not self._wrec_ok:
            ops.wrec_build(bia, self.inv_occ, self._wrec)
            self._wrec_ok = True
        elif _CHECK_WREC:          # debug (VFM_CHECK_WREC=1, the state-machine fuzz): the cache IS the two tables
            assert torch.equal(self._wrec[:, :2], bia) and torch.equal(self._wrec[:, 2], self.inv_occ), "stale packed records"
        return self._wrec

    def _records_ready(self, plan, step) -> bool:
        f = self._zrec_for
        return f is not None and f[0] is plan and f[1] == step and f[2] == self._flat._version

    def _train_step_pipelined(self, plan, next_plan, lr, step, scaled, mark, wrec=None, la_next=None):
        """The fused step with the sampling of the NEXT batch moved into this step's backward (see __init__).
        la_next (the next batch's plan): the look-ahead form -- the backward visits only the rows of this batch and of that
        one, replaying the zero-gradient updates a visited row skipped (exactly the bookkeeping of the look-ahead form in
        `_step_fused`: `_lazy_last`, `_la_tab`, `_la_ready_for`)."""
        ent, bia, scal = self._views(self._flat)
        loss3 = self._gflat[self._n_flat: self._n_flat + 3]
        if self._zrec is None:
            rl = ops.record_len(self.d)
            self._zrec = [torch.zeros(self.T, rl, dtype=torch.float32, device=self.device) for _ in range(2)]
        cur, nxt = self._zrec
        fresh = not self._records_ready(plan, step)
        la = la_next is not None
        if la:
            k = (self._adam_t - 1) % ops.MOMENT_PERIOD + 1
            if self._lazy_last is None:
                self._lazy_last = torch.empty(self.T, dtype=torch.int32, device=self.device)
            if not self._lazy_dirty:     # no row lags: every row is at the step before this one
                self._lazy_last.fill_(self._adam_t - 1)
            if self._la_tab is None:
                self._la_tab = torch.zeros(2 * (ops.MOMENT_PERIOD + 1), dtype=torch.float32, device=self.device)
            ready = self._la_ready_for
            if self._lazy_dirty and not (ready is not None and ready[0] is plan and ready[1] == self._adam_t - 1):
                # this batch was not the one announced to the previous step: bring its rows up to date now (the records
                # below are then sampled from current rows; the kernel finds nothing left to replay on them)
                ops.adam_catchup(ent, bia, self._views(self._adam_m), self._views(self._adam_v), self._lazy_last,
                                 plan.touched_ids(), self._lazy_lrs(k - 1), upto=self._adam_t - 1, mark=self._adam_t - 1,
                                 wrec=wrec)
                mark("catchup")
                fresh = True             # (records written for this batch before its rows were caught up would be stale)
        if fresh:                                     # first step of a run (or the tables changed since): from the tables
            ops.sample_records(plan, ent, bia, self.inv_occ, cur, self.rng_seed, step)
            mark("sample_rec")
        _, grow, pred = self._step_buffers(plan.B)
        if next_plan is not None and (next_plan.spec.T != self.T or next_plan.y is None):
            next_plan = None

        st = ops.elbo_forward_records(plan, cur, scal, self.rng_seed, step, pred, grow, self._partials,
                                      flags=ops.FLAG_SHARE_GPU if getattr(self, "_sharing", False) else 0)
        mark("fwd")
        ops.elbo_backward_adam_pipe(plan, st, cur, nxt, next_plan, step + 1, ent, bia, scal, self.inv_occ,
                                    self._views(self._adam_m), self._views(self._adam_v), lr, self._adam_t, loss3,
                                    scaled_moments=scaled, wrec=wrec,
                                    last_step=self._lazy_last if la else None, step_tab=self._la_tab if la else None,
                                    listed=self.lookahead_list and not getattr(self, "_sharing", False), la_next=la_next)
        mark("bwd_adam")
        out = st.pred
        if next_plan is not None:
            self._zrec = [nxt, cur]
            self._zrec_for = (next_plan, step + 1, self._flat._version)
        else:
            self._zrec_for = None
        if la:
            self._lazy_dirty, self._lazy_kind = True, "la"
            self._la_ready_for = (la_next, self._adam_t)
        return loss3, out

    def fit(self, X_train, y_train, n_epochs: int = 50, batch_size: int = 100000, X_test=None,
            y_test=None, display_every: int = 1, lr: Optional[float] = None, verbose: bool = True,
            process_group=None, sort_within_batch: bool = True, stream_plans: bool = False):
        """The training loop of vfm-torch.py:337-422: sequential batches without shuffling
        (:121-122), lr = 1/(1 + nb_train // batch_size) (:92), dense Adam, per-epoch train metrics,
        `save_weights()` each epoch for 'reg' (:380), test metrics every `display_every` epochs.
        With a process group, every batch is split in contiguous row blocks over the ranks.
        `sort_within_batch`: reorder the rows INSIDE each batch by the id of the last column (items)
        once, before training -- the batch composition, the loss and the gradients are unchanged
        (they are sums over the batch's rows), but rows sharing an item become neighbours, so the
        forward gather re-reads item rows from L2 instead of HBM.
        `stream_plans` (single rank): keep NO plan across steps -- every batch's plan (inverted index, normalisers, look-ahead
        row list) is built again each time the batch comes up, two steps ahead on a side stream (`plan_async`), as a caller
        that streams or shuffles its batches has to; same trajectory, bit for bit (tests/test_gpu_model.py).
        Returns a history dict (lists per evaluated epoch)."""
        X_train = torch.as_tensor(X_train)
        y_train = torch.as_tensor(y_train, dtype=torch.float32)
        nb_train = X_train.shape[0]
        self.set_training_data(X_train, nb_train)
        self.lr = lr if lr is not None else 1.0 / (1 + nb_train // batch_size)
        world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        Xd = X_train.to(self.device)
        yd = y_train.to(self.device)
        if sort_within_batch:
            Xd, yd = sort_rows_within_batches(Xd, yd, batch_size)
        plans, spans = [], []
        stream_plans = bool(stream_plans) and world == 1
        for lo in range(0, nb_train, batch_size):
            hi = min(lo + batch_size, nb_train)
            a, b = shard_rows(lo, hi, rank, world)
            spans.append((a, b))
            if not stream_plans:
                plans.append(self.plan(Xd[a:b], yd[a:b], B_global=hi - lo, process_group=process_group,
                                       defer_readback=True))      # (the index builds are enqueued back to back)
        if stream_plans:
            return self._fit_streamed(Xd, yd, spans, n_epochs, X_test, y_test, display_every, verbose)
        if world == 1 and self.lookahead and self.lookahead_list and len(plans) > 1 and self.n_samples == 1:
            for i, plan in enumerate(plans):              # row lists of the look-ahead step, once per pair of batches --
                nxt = plans[(i + 1) % len(plans)]         # only where that step form will run (U: one deferred readback)
                if self._lookahead_pays(plan, nxt) and (not self._will_pipeline(plan, nxt) or self.pipeline_lookahead):
                    plan.prepare_lookahead(nxt)       # (the pipelined step has a look-ahead form too)
        train_pred = torch.zeros(nb_train, dtype=torch.float32, device=self.device)
        hist = {"epoch": [], "elbo": [], "train_rmse": [], "train_auc": [], "test": []}
        losses = torch.zeros(len(plans), dtype=torch.float32, device=self.device)
        for epoch in range(n_epochs):
            for i, (plan, (a, b)) in enumerate(zip(plans, spans)):
                if self.n_samples == 1:
                    # (the pipelined / look-ahead forms, and the multi-rank rows exchange, are told which batch follows)
                    nxt = plans[(i + 1) % len(plans)] if (world == 1 or self.exchange == "rows") else None
                    loss3, pr = self.train_step(plan, process_group=process_group, next_plan=nxt)
                    train_pred[a:b] = pr
                else:       # [S,B] predictions: the train metrics use their mean over the samples
                    loss3, pr = self.train_step(plan, process_group=process_group)
                    train_pred[a:b] = pr.mean(0)
                losses[i] = loss3[0]
            # ---- end of epoch (vfm-torch.py:378-384)
            if epoch == 0 or epoch == n_epochs - 1:
                for plan in plans:            # (did a kernel have to clamp an entry of a plan's index?  vfm_index_t.status)
                    plan.check_status()
            if self.output == "reg":
                self.save_weights()
            if epoch % display_every == 0:
                rec = {"epoch": epoch, "elbo": float(losses.mean())}
                tp = train_pred
                if world > 1:                  # each rank only wrote its own row blocks (the rest stays 0)
                    tp = train_pred.clone()
                    torch.distributed.all_reduce(tp, group=process_group)
                if self.output == "reg":
                    pr = tp.clamp(1, 5)
                    rec["train_rmse"] = float(torch.sqrt(torch.mean((pr - yd) ** 2)))
                else:
                    rec["train_auc"], rec["train_map"] = _auc_map(yd, torch.sigmoid(tp))
                if X_test is not None:
                    rec["test"] = self.evaluate(X_test, y_test)
                hist["epoch"].append(epoch)
                hist["elbo"].append(rec["elbo"])
                hist["train_rmse"].append(rec.get("train_rmse"))
                hist["train_auc"].append(rec.get("train_auc"))
                hist["test"].append(rec.get("test"))
                if verbose and rank == 0:
                    print(f"Epoch {epoch}: Elbo {rec['elbo']:.4f} " +
                          (f"Minibatch train RMSE {rec['train_rmse']:.4f}" if self.output == "reg" else
                           f"Minibatch train AUC {rec['train_auc']:.4f} Minibatch train MAP {rec['train_map']:.4f}"),
                          rec.get("test", ""))
        self.sync_lazy()          # (lazy Adam mode: every row up to date before the caller looks at the tables)
        return hist

    def _fit_streamed(self, Xd, yd, spans, n_epochs, X_test, y_test, display_every, verbose):
        """fit() with every plan built when its batch comes up (two steps ahead, on the side stream): the loop of
        vfm-torch.py:347-422 as a caller that cannot reuse plans runs it."""
        nb, nb_train = len(spans), Xd.shape[0]
        train_pred = torch.zeros(nb_train, dtype=torch.float32, device=self.device)
        hist = {"epoch": [], "elbo": [], "train_rmse": [], "train_auc": [], "test": []}
        losses = torch.zeros(nb, dtype=torch.float32, device=self.device)
        total = n_epochs * nb
        # (fork=False: Xd / yd are resident since before the loop -- the builds need not wait for the steps in flight)
        batch = lambda t: (Xd[spans[t % nb][0]:spans[t % nb][1]], yd[spans[t % nb][0]:spans[t % nb][1]], False)
        # `plan_prefetch_depth` plans in hand: this batch's, the next one's (named to the step), and the ones being built
        D = max(2, int(self.plan_prefetch_depth))
        self.plan_streams().wait_current()       # (Xd / yd were put together on this stream: the fork=False builds come after that)
        q = [self.plan(*batch(0)[:2], defer_readback=True)] + [self.plan_async(*batch(k)[:2]) for k in range(1, min(D, total))]
        for t in range(total):
            epoch, i = divmod(t, nb)
            a, b = spans[i]
            nxt = q[1] if len(q) > 1 else None
            loss3, pr = self.train_step(q[0], next_plan=nxt if self.n_samples == 1 else None,
                                        prefetch=batch(t + D) if t + D < total else None)
            train_pred[a:b] = pr if self.n_samples == 1 else pr.mean(0)
            losses[i] = loss3[0]
            q.pop(0)
            if t + D < total:
                q.append(self.prefetched)
            if i == nb - 1:
                if self.output == "reg":
                    self.save_weights()
                if epoch % display_every == 0:
                    rec = {"epoch": epoch, "elbo": float(losses.mean())}
                    if self.output == "reg":
                        rec["train_rmse"] = float(torch.sqrt(torch.mean((train_pred.clamp(1, 5) - yd) ** 2)))
                    else:
                        rec["train_auc"], rec["train_map"] = _auc_map(yd, torch.sigmoid(train_pred))
                    if X_test is not None:
                        rec["test"] = self.evaluate(X_test, y_test)
                    for k in ("epoch", "elbo"):
                        hist[k].append(rec[k])
                    hist["train_rmse"].append(rec.get("train_rmse"))
                    hist["train_auc"].append(rec.get("train_auc"))
                    hist["test"].append(rec.get("test"))
                    if verbose:
                        print(f"Epoch {epoch}: Elbo {rec['elbo']:.4f}", rec.get("train_rmse", rec.get("train_auc")), rec.get("test", ""))
        self.sync_lazy()
        return hist

    @torch.no_grad()
    def predict(self, X, eps=None):
        """The four predictors of the evaluation block (vfm-torch.py:402-417):
        y_pred (fresh posterior sample), mean_pred (average of y_pred over the calls so far --
        "Test RMSE all"), y_pred_of_last / y_pred_of_mean (deterministic, from the last /
        epoch-averaged posterior means; None until save_weights() has run).  'reg' outputs are
        clipped to [1, 5] like the reference (:405-406); 'class' outputs are probabilities.
        `eps`: eps tables for the sampled prediction (tests replay the reference's draws); default Philox."""
        lik, last, mean, _ = self.forward(X, eps=eps)
        y_pred = lik.mean.mean(0)          # (S = 1: the sample itself; S > 1: mean over the S samples)
        if self.output == "reg":
            y_pred = y_pred.clamp(1, 5)
            mean = mean.clamp(1, 5) if mean is not None else None
        else:
            last = torch.sigmoid(last) if last is not None else None
            mean = torch.sigmoid(mean) if mean is not None else None
        if self._all_preds_sum is None or self._all_preds_sum.shape != y_pred.shape:
            self._all_preds_sum = torch.zeros_like(y_pred)
            self._all_preds_n = 0
        self._all_preds_sum += y_pred
        self._all_preds_n += 1
        return {"y_pred": y_pred, "mean_pred": self._all_preds_sum / self._all_preds_n,
                "y_pred_of_last": last, "y_pred_of_mean": mean}

    @torch.no_grad()
    def predict_samples(self, X, n_samples: int = 10):
        """Posterior-predictive mean and variance of the prediction over `n_samples` fresh posterior
        samples (the quantity the paper's preference-elicitation use case consumes: mean + logit
        variance, cf. vfm.py:1024-1057).  Each sample is one forward launch with its own Philox step.
        Returns dict(mean, var, logits_mean, logits_var); 'reg': mean == logits_mean."""
        self._fresh_params()
        plan = self.plan(X, None)
        ent, bia, scal = self._views(self._flat)
        n = 0
        mean = torch.zeros(plan.B, dtype=torch.float32, device=self.device)
        m2 = torch.zeros_like(mean)
        pmean = torch.zeros_like(mean)
        for _ in range(int(n_samples)):
            step = self.global_step
            self.global_step += 1
            logit = ops.elbo_forward(_single_sample(plan), ent, bia, scal, None, seed=self.rng_seed, step=step,
                                     train=False).pred
            n += 1
            delta = logit - mean
            mean += delta / n
            m2 += delta * (logit - mean)
            pmean += logit if self.output == "reg" else torch.sigmoid(logit)
        var = m2 / max(n - 1, 1)
        pm = pmean / n
        return {"mean": pm, "var": var if self.output == "reg" else None, "logits_mean": mean, "logits_var": var}

    @torch.no_grad()
    def evaluate(self, X_test, y_test):
        """Test metrics of vfm-torch.py:410-422."""
        out = self.predict(X_test)
        y = torch.as_tensor(y_test, dtype=torch.float32).to(self.device)
        if self.output == "reg":
            rm = lambda p: None if p is None else float(torch.sqrt(torch.mean((p - y) ** 2)))
            return {"rmse": rm(out["y_pred"]), "rmse_all": rm(out["mean_pred"]),
                    "rmse_of_last": rm(out["y_pred_of_last"]), "rmse_of_mean": rm(out["y_pred_of_mean"])}
        auc, ap = _auc_map(y, out["y_pred"])
        return {"auc": auc, "map": ap}


def _single_sample(plan):
    """The plan with n_samples = 1 (deterministic predictions / one draw per launch)."""
    if plan.spec.n_samples == 1:
        return plan
    import copy
    import dataclasses
    p = copy.copy(plan)
    p.spec = dataclasses.replace(plan.spec, n_samples=1)
    return p


def sort_rows_within_batches(X, y, batch_size):
    """Stable sort of the rows of every batch window [lo, lo+batch_size) by the last id column."""
    n = X.shape[0]
    key = X[:, -1].to(torch.int64) + (torch.arange(n, device=X.device) // batch_size) * (int(X.max()) + 1)
    order = torch.argsort(key, stable=True)
    return X[order].contiguous(), y[order].contiguous()


def _auc_map(y, p):
    from sklearn.metrics import roc_auc_score, average_precision_score
    yy, pp = y.detach().cpu().numpy(), p.detach().cpu().numpy()
    if len(np.unique(yy)) < 2:
        return float("nan"), float("nan")
    return float(roc_auc_score(yy, pp)), float(average_precision_score(yy, pp))
