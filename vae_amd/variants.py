"""ELBO variants of the reference's sibling scripts on the general HIP kernels of
`csrc/vfm_variants.hip`:

  * closed-form expected log-likelihood   vfm-tomasrch.py:369-451 (+ its loss, :569-588)
  * learnable group priors                vfm-tomasrch.py:206-290
  * sparse features with values != 1      vfm.py:483-509

`variant_forward / variant_backward` are the launch wrappers, `VariantElbo` the autograd.Function,
`VFMClosedForm` the module with the parameter names of vfm-tomasrch.py's `CF`.  All arithmetic runs in the
HIP kernels (no CPU / PyTorch fallback); torch provides memory, autograd plumbing and the optimizer.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from ._lib import check, ptr, current_stream_ptr

OBJECTIVES = {"sampled": _lib.OBJ_SAMPLED, "closed_form": _lib.OBJ_CLOSED_FORM}


def priors_len(G: int, d: int) -> int:
    """Floats of the flat prior vector [mean0, scale0 | mean_w[G] | scale_w[G] | mean_v[G,d] | scale_v[G,d]]."""
    return 2 + 2 * G + 2 * G * d


_index_struct = ops._index_struct


def variant_forward(plan: ops.BatchPlan, objective: str, entity_params, bias_params, scalars, inv_occ, *,
                    priors=None, values=None, eps=None, seed=0, step=0, train=True):
    """Launch vfm_variant_fwd_f32.  Returns dict(pred [B], loss3 [3] or None, state, grow, partials, problem)."""
    spec, dev = plan.spec, plan.x.device
    for t, n in ((entity_params, "entity_params"), (bias_params, "bias_params"), (scalars, "scalars")):
        ops._need_cuda(t, n)
    B, d = plan.B, spec.d
    train = train and plan.y is not None
    ns = 3 if objective == "closed_form" else 1
    pred = torch.empty(B, dtype=torch.float32, device=dev)
    partials = torch.empty(_lib.PARTIALS_LEN, dtype=torch.float64, device=dev)
    state = torch.empty(B, ns * d, dtype=torch.float32, device=dev) if train else None
    grow = torch.empty(B, dtype=torch.float32, device=dev) if train else None
    loss3 = torch.empty(3, dtype=torch.float32, device=dev) if train else None
    if priors is not None and priors.numel() != priors_len(spec.F, d):
        raise ValueError("priors must hold 2 + 2G + 2Gd floats")
    if values is not None:
        values = values.to(torch.float32).contiguous()
        if values.shape != (B, spec.F):
            raise ValueError("values must be [B,F]")
    p = ops._problem(spec, B, plan.B_global, plan.id_bits, seed, step, 0)
    e = eps if eps is not None else (None, None, None)
    check(_lib.load().vfm_variant_fwd_f32(
        C.byref(p), OBJECTIVES[objective], ptr(plan.x), ptr(values), ptr(plan.y if train else None), ptr(entity_params),
        ptr(bias_params), ptr(inv_occ), ptr(scalars), ptr(plan.W if train else None), ptr(priors), ptr(e[0]), ptr(e[1]),
        ptr(e[2]), ptr(pred), ptr(partials), ptr(state), ptr(grow), ptr(loss3), current_stream_ptr(dev)),
        "vfm_variant_fwd_f32")
    return {"pred": pred, "loss3": loss3, "state": state, "grow": grow, "partials": partials, "problem": p,
            "objective": objective, "priors": priors, "values": values, "eps": eps}


def variant_backward(plan: ops.BatchPlan, st, entity_params, bias_params, scalars, inv_occ, grad_out):
    """Launch vfm_variant_bwd_f32: dense gradients of both tables, the scalars and (if given) the priors."""
    spec, dev = plan.spec, plan.x.device
    g_ent, g_bias = torch.empty_like(entity_params), torch.empty_like(bias_params)
    g_sc = torch.empty(3, dtype=torch.float32, device=dev)
    g_pr = torch.empty_like(st["priors"]) if st["priors"] is not None else None
    ws = torch.empty(int(_lib.load().vfm_variant_workspace_elems(plan.B, spec.F, spec.d)), dtype=torch.int32, device=dev)
    ix = _index_struct(plan)
    e = st["eps"] if st["eps"] is not None else (None, None, None)
    check(_lib.load().vfm_variant_bwd_f32(
        C.byref(st["problem"]), OBJECTIVES[st["objective"]], C.byref(ix), ptr(ws), ptr(plan.x), ptr(st["values"]),
        ptr(entity_params), ptr(bias_params), ptr(inv_occ), ptr(scalars), ptr(plan.W), ptr(st["priors"]), ptr(e[0]),
        ptr(e[1]), ptr(e[2]), ptr(st["state"]), ptr(st["grow"]), ptr(st["partials"]), ptr(grad_out), ptr(g_ent),
        ptr(g_bias), ptr(g_sc), ptr(g_pr), current_stream_ptr(dev)), "vfm_variant_bwd_f32")
    return g_ent, g_bias, g_sc, g_pr


class VariantElbo(torch.autograd.Function):
    """loss, pred, loss3 = VariantElbo.apply(entity_params, bias_params, scalars[3], priors_flat | None, plan,
    inv_occ, objective, values, eps, seed, step) -- differentiable in the first four."""

    @staticmethod
    def forward(ctx, entity_params, bias_params, scalars, priors, plan, inv_occ, objective, values, eps, seed, step):
        ent, bia, sc = entity_params.detach().contiguous(), bias_params.detach().contiguous(), scalars.detach().contiguous()
        pr = priors.detach().contiguous() if priors is not None else None
        st = variant_forward(plan, objective, ent, bia, sc, inv_occ, priors=pr, values=values, eps=eps, seed=seed, step=step)
        ctx.plan, ctx.st, ctx.inv_occ, ctx.args = plan, st, inv_occ, (ent, bia, sc)
        ctx.has_priors = priors is not None
        ctx.mark_non_differentiable(st["pred"], st["loss3"])
        return st["loss3"][0:1].clone(), st["pred"], st["loss3"]

    @staticmethod
    def backward(ctx, g_loss, _gp, _g3):
        ent, bia, sc = ctx.args
        gout = g_loss.to(torch.float32).reshape(1).contiguous()
        g_ent, g_bias, g_sc, g_pr = variant_backward(ctx.plan, ctx.st, ent, bia, sc, ctx.inv_occ, gout)
        return (g_ent, g_bias, g_sc, g_pr if ctx.has_priors else None, None, None, None, None, None, None, None)


class VFMClosedForm(nn.Module):
    """The model of vfm-tomasrch.py (`class CF`, :186-453): G id groups, learnable group priors, closed-form
    expected log-likelihood -- same parameter names and shapes, so state_dicts of the two line up.
    `elbo(x, y)` is the loss of :569-588 (differentiable); `fit` its Adam loop (:464-590, lr as given there)."""

    def __init__(self, group_sizes: Sequence[int], embedding_size: int = 2, alpha_0: float = 300.0, device="cuda"):
        super().__init__()
        self.group_sizes = [int(s) for s in group_sizes]
        self.G, self.T, self.d = len(self.group_sizes), int(sum(self.group_sizes)), int(embedding_size)
        G, d, start_scale = self.G, self.d, 0.2
        # same construction order as the reference (RNG: mean_global_bias, bias means per group, entity means per group)
        self.alpha = nn.Parameter(torch.tensor([float(alpha_0)]))
        self.mean_global_bias_prior = nn.Parameter(torch.zeros(1))
        self.scale_global_bias_prior = nn.Parameter(torch.ones(1))
        self.mean_global_bias = nn.Parameter(torch.normal(torch.zeros(1), torch.ones(1)))
        self.scale_global_bias = nn.Parameter(torch.tensor([start_scale]))
        self.mean_group_bias_prior = nn.ParameterList([nn.Parameter(torch.zeros(1)) for _ in range(G)])
        self.scale_group_bias_prior = nn.ParameterList([nn.Parameter(torch.ones(1)) for _ in range(G)])
        self.bias_params = nn.Parameter(torch.cat([
            torch.cat((torch.normal(torch.zeros(n, 1), 1e-1 * torch.ones(n, 1)), start_scale * torch.ones(n, 1)), 1)
            for n in self.group_sizes]))
        self.mean_group_entity_prior = nn.ParameterList([nn.Parameter(torch.zeros(d)) for _ in range(G)])
        self.scale_group_entity_prior = nn.ParameterList([nn.Parameter(torch.ones(d)) for _ in range(G)])
        self.entity_params = nn.Parameter(torch.cat([
            torch.cat((torch.normal(torch.zeros(n, d), 1e-7 * torch.ones(n, d)), start_scale * torch.ones(n, d)), 1)
            for n in self.group_sizes]))
        self.to(device)
        self.nb_train, self.inv_occ = 1, None

    def spec(self) -> ops.Spec:
        return ops.Spec(T=self.T, F=self.G, d=self.d, group_hi=tuple(int(v) for v in np.cumsum(self.group_sizes)),
                        group_n=tuple(float(s) for s in self.group_sizes), likelihood=_lib.LIK_NORMAL,
                        nb_train=int(self.nb_train))

    def set_training_data(self, X_train, nb_train: Optional[int] = None, nb_occ=None):
        """entity_count = bincount(X_train.flatten()) (vfm-tomasrch.py:182)."""
        X_train = torch.as_tensor(X_train)
        self.nb_train = int(nb_train if nb_train is not None else X_train.shape[0])
        dev = self.alpha.device
        if nb_occ is None:
            nb_occ = torch.bincount(X_train.reshape(-1).to(dev).to(torch.int64), minlength=self.T)
        self.inv_occ = ops.inv_occ_from_counts(torch.as_tensor(nb_occ).to(dev).to(torch.int64).contiguous())

    def priors_flat(self) -> torch.Tensor:
        return torch.cat([self.mean_global_bias_prior, self.scale_global_bias_prior,
                          torch.cat(list(self.mean_group_bias_prior)), torch.cat(list(self.scale_group_bias_prior)),
                          torch.cat(list(self.mean_group_entity_prior)), torch.cat(list(self.scale_group_entity_prior))])

    def plan(self, x, y=None) -> ops.BatchPlan:
        dev = self.alpha.device
        x = torch.as_tensor(x).to(dev).contiguous()
        y = torch.as_tensor(y).to(dev) if y is not None else None
        return ops.BatchPlan(self.spec(), x, y, self.inv_occ if y is not None else None)

    def elbo(self, x=None, y=None, plan=None, values=None):
        """(loss[1], y_bar[B], (loss, likelihood term, KL term)) of one batch -- vfm-tomasrch.py:548-588."""
        if plan is None:
            plan = self.plan(x, y)
        scalars = torch.cat([self.alpha, self.mean_global_bias, self.scale_global_bias])
        return VariantElbo.apply(self.entity_params, self.bias_params, scalars, self.priors_flat(), plan, self.inv_occ,
                                 "closed_form", values, None, 0, 0)

    @torch.no_grad()
    def forward(self, x, values=None):
        """y_bar of the rows x (the mean prediction the reference reports, vfm-tomasrch.py:342-347 for two groups)."""
        plan = self.plan(x, None)
        scalars = torch.cat([self.alpha, self.mean_global_bias, self.scale_global_bias]).contiguous()
        return variant_forward(plan, "closed_form", self.entity_params.detach(), self.bias_params.detach(), scalars, None,
                               values=values, train=False)["pred"]

    def fit(self, X_train, y_train, n_epochs=10, batch_size=8000, lr=0.02, verbose=False):
        """The Adam loop of vfm-tomasrch.py:464-590 (sequential batches, no shuffle, :177-178)."""
        X_train, y_train = torch.as_tensor(X_train), torch.as_tensor(y_train, dtype=torch.float32)
        self.set_training_data(X_train)
        opt = torch.optim.Adam(self.parameters(), lr=lr)
        plans = [self.plan(X_train[lo:lo + batch_size], y_train[lo:lo + batch_size])
                 for lo in range(0, len(y_train), batch_size)]
        hist = []
        for epoch in range(n_epochs):
            tot = 0.0
            for plan in plans:
                loss, _, _ = self.elbo(plan=plan)
                opt.zero_grad()
                loss.backward()
                opt.step()
                tot += float(loss.detach())
            hist.append(tot / len(plans))
            if verbose:
                print(f"epoch {epoch}: elbo {hist[-1]:.4f}")
        return hist
