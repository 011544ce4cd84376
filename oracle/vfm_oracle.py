"""CPU oracle for the Variational-FM ELBO step.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product package `vae_amd` never does (its ops fail
loudly when the HIP library is missing).

Two restatements of the per-batch hot path of the reference
(`/root/reference/vfm-torch.py`, class `CF` + loss line):

* `reference_shaped_*` -- torch-CPU, the SAME op graph as the reference:
  `torch.unique` (:190-192) -> embedding lookups of the unique rows (:207-208)
  -> `Normal(loc, |scale|)` samplers (:200-203,212-215,223-226) -> `rsample`
  (:238-241) -> expand to rows with the inverse index, sum / prod over the two
  fields (:244-245) -> likelihood (:264-270) -> `kl_divergence` to N(0,1)
  (:290,295,322) -> occurrence re-weighting with the `<= N` test (:298-317)
  -> loss `-log_prob(y).mean()*nb_train + kl` (:359) -> autograd backward
  (:368-369) -> dense Adam (:339,370).  This is the function `bench.py` times
  on the host cores as `cpu_baseline` (kind "port").

* `rowwise_elbo` -- numpy float64, the row-wise identity the HIP kernels
  implement: every term is a sum over batch rows,
  general number of fields F, general id groups, analytic gradients.

Parity pin: both are checked in `tests/test_oracle.py` against the golden
vectors in `tests/golden/*.npz`, which were produced by running the
reference's own `CF` class (lifted with `ast` by `tools/make_golden.py`) in the
build container.  The reference repo itself has no tests / known answers for
this path, so those vectors are the pin.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import distributions

LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


# --------------------------------------------------------------------------
# (a) reference-shaped torch-CPU restatement
# --------------------------------------------------------------------------
def make_params(T, d, dtype=torch.float32, seed=42, alpha=None):
    """Parameters of `CF.__init__` (vfm-torch.py:133-153): alpha~U(0,1), global
    bias N(0, 1) posterior init (mean 0, scale 1), Embedding tables ~N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    P = {
        "alpha": torch.rand(1, generator=g) if alpha is None else torch.tensor([float(alpha)]),
        "global_bias_mean": torch.zeros(1),
        "global_bias_scale": torch.ones(1),
        "bias_params": torch.randn(T, 2, generator=g),
        "entity_params": torch.randn(T, 2 * d, generator=g),
    }
    return {k: v.to(dtype).requires_grad_(True) for k, v in P.items()}


TORCH_LINKS = {"abs": torch.abs, "softplus": torch.nn.functional.softplus}     # vfm-torch.py:126 / :125


def reference_shaped_forward(P, x, nb_occ, N, M, output="reg", eps=None, n_samples=1, link="abs", dead_gathers=False):
    """One `CF.forward` (vfm-torch.py:189-324) with S = n_samples variational samples
    (the reference's global N_VARIATIONAL_SAMPLES, :19) and `link` = its global LINK (:125-126).

    eps: None -> draw with rsample exactly like the reference (RNG order eps0[S,1],
    eps_w[S,U], eps_v[S,U,d] over the SORTED unique ids); or a tuple
    (eps0[S], eps_w[S,U], eps_v[S,U,d]) (S = 1: the leading axis may be absent) to replay
    recorded draws.  dead_gathers: also perform the reference's two unused per-row lookups (`bias_batch`,
    `entity_batch`, vfm-torch.py:204-205: [B,2,2] and [B,2,2d], never read) -- the timing baseline pays what the
    reference pays.  Returns (likelihood distribution with batch shape [S,B], kl_term[1])."""
    LINK, S = TORCH_LINKS[link], int(n_samples)
    if dead_gathers:
        _bias_batch = torch.nn.functional.embedding(x, P["bias_params"])        # noqa: F841  (as the reference: unused)
        _entity_batch = torch.nn.functional.embedding(x, P["entity_params"])    # noqa: F841
    uniq, pos, cnt = torch.unique(x, return_inverse=True, return_counts=True)
    users, cnt_u = torch.unique(x[:, 0], return_counts=True)
    items, cnt_i = torch.unique(x[:, 1], return_counts=True)
    prior = distributions.Normal(0, 1)

    theta = torch.nn.functional.embedding(uniq, P["bias_params"])
    phi = torch.nn.functional.embedding(uniq, P["entity_params"])
    d = phi.shape[1] // 2
    q0 = distributions.Normal(P["global_bias_mean"], LINK(P["global_bias_scale"]))
    qw = distributions.Normal(theta[:, 0], LINK(theta[:, 1]))
    qv = distributions.Normal(loc=phi[:, :d], scale=LINK(phi[:, d:]))

    if eps is None:
        w0, w, z = q0.rsample((S,)), qw.rsample((S,)), qv.rsample((S,))
    else:
        e0, ew, ev = (torch.as_tensor(e, dtype=phi.dtype) for e in eps)
        w0 = (q0.loc + e0.reshape(S, 1) * q0.scale)
        w = (qw.loc + ew.reshape(S, -1) * qw.scale)
        z = (qv.loc + ev.reshape(S, -1, d) * qv.scale)

    first = w[:, pos].sum(axis=2).mean(axis=0).squeeze()
    second = z[:, pos].prod(axis=2).sum(axis=2).mean(axis=0)
    logits = w0 + first + second
    if output == "reg":
        lik = distributions.Normal(logits, torch.sqrt(1 / LINK(P["alpha"])))
    else:
        lik = distributions.Bernoulli(logits=logits)

    kl_e = distributions.kl_divergence(qw, prior) + distributions.kl_divergence(qv, prior).sum(axis=1)
    occ = nb_occ[uniq]
    w_user = (cnt_u / nb_occ[users]).sum(axis=0)
    w_item = (cnt_i / nb_occ[items]).sum(axis=0)
    kl = (kl_e * (cnt / occ) * ((uniq <= N) * N / w_user + (uniq > N) * M / w_item)).sum(axis=0)
    return lik, distributions.kl_divergence(q0, prior) + kl


def reference_shaped_loss(P, x, y, nb_occ, N, M, nb_train, output="reg", eps=None, n_samples=1, link="abs",
                          dead_gathers=False):
    lik, kl = reference_shaped_forward(P, x, nb_occ, N, M, output, eps, n_samples, link, dead_gathers)
    loss = -lik.log_prob(y.to(kl.dtype)).mean() * nb_train + kl          # vfm-torch.py:359
    return loss, lik, kl


def reference_shaped_step(P, opt, x, y, nb_occ, N, M, nb_train, output="reg", eps=None, n_samples=1,
                          link="abs", dead_gathers=False):
    """forward + loss + backward + optimiser step (vfm-torch.py:353-370)."""
    loss, lik, _ = reference_shaped_loss(P, x, y, nb_occ, N, M, nb_train, output, eps, n_samples, link, dead_gathers)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.detach(), lik.mean.detach()


class SavedWeights:
    """`CF.save_weights` (vfm-torch.py:179-185): snapshots of the posterior MEANS (global bias,
    first-order weights, embeddings) appended once per epoch, and their mean over the snapshots."""

    def __init__(self):
        self.g, self.b, self.e = [], [], []

    def save(self, P):
        d = P["entity_params"].shape[1] // 2
        self.g.append(P["global_bias_mean"].detach().numpy().copy())
        self.b.append(P["bias_params"][:, 0].detach().numpy().copy())
        self.e.append(P["entity_params"][:, :d].detach().numpy().copy())

    def logits(self, x):
        """(last_logits, mean_logits) of rows x [B,2] (vfm-torch.py:248-259): deterministic
        predictions from the last snapshot and from the mean of all snapshots."""
        x = np.asarray(x)
        out = []
        for g, b, e in ((self.g[-1], self.b[-1], self.e[-1]),
                        (np.array(self.g).mean(axis=0), np.array(self.b).mean(axis=0), np.array(self.e).mean(axis=0))):
            out.append(g + b[x].sum(axis=1).squeeze() + e[x].prod(axis=1).sum(axis=1))
        return out[0], out[1]


# --------------------------------------------------------------------------
# (b) row-wise float64 restatement (what the HIP kernels compute)
# --------------------------------------------------------------------------
def group_of(ids, group_hi):
    """Group of an entity id: first g with id < group_hi[g].  The reference's two
    groups are `id <= N` and `id > N` (vfm-torch.py:316) -> group_hi = [N+1, T]."""
    return np.searchsorted(np.asarray(group_hi), ids, side="right")


def batch_norms(x, nb_occ):
    """W_f = sum_r 1/occ(x[r,f]) per COLUMN f (vfm-torch.py:305-306: the
    normalisers are built from the unique values of each column)."""
    return (1.0 / nb_occ[x].astype(np.float64)).sum(axis=0)


def _np_link(link):
    """(sigma(s), dsigma/ds) of the link function (vfm-torch.py:125-126), float64."""
    if link == "abs":
        return np.abs, np.sign
    if link == "softplus":
        return (lambda s: np.logaddexp(0.0, s)), (lambda s: 1.0 / (1.0 + np.exp(-s)))
    raise ValueError(link)


def rowwise_elbo(P, x, y, nb_occ, group_hi, group_n, nb_train, eps0, eps_w, eps_v,
                 output="reg", W=None, B_global=None, want_grads=True, link="abs"):
    """Row-wise ELBO + analytic gradients, float64.

    P: dict of numpy arrays (alpha[1], global_bias_mean[1], global_bias_scale[1],
       bias_params[T,2], entity_params[T,2d]).
    x [B,F] int, y [B], nb_occ [T] int; eps_w [T], eps_v [T,d] are indexed BY ENTITY ID
    (one draw per entity per batch, shared by all rows that contain it).  With S > 1
    variational samples (vfm-torch.py:19,238-245): eps0 [S], eps_w [S,T], eps_v [S,T,d]; the
    entity terms are averaged over the samples before the likelihood, the global bias is not
    (:244-245,265), and the likelihood is averaged over S*B (:359).
    group_hi [G] exclusive id upper bounds, group_n [G] the n_g multipliers (N, M);
    column f's normaliser W_f divides group f's KL sum (G == F).
    W / B_global: batch-global normalisers / row count when `x` is only a shard.
    link: the scale parameters' link function, "abs" (:126) or "softplus" (:125).
    Outputs `pred`, `mean`: [B] for S = 1, else [S,B]."""
    f8 = np.float64
    L, dL = _np_link(link)
    x = np.asarray(x)
    B, F = x.shape
    Bg = B if B_global is None else B_global
    ent = np.asarray(P["entity_params"])
    bia = np.asarray(P["bias_params"])
    d = ent.shape[1] // 2
    alpha, m0, s0 = (float(np.asarray(P[k]).reshape(-1)[0])
                     for k in ("alpha", "global_bias_mean", "global_bias_scale"))
    eps_v = np.asarray(eps_v, f8)
    multi = eps_v.ndim == 3
    eps_v = eps_v if multi else eps_v[None]
    S = eps_v.shape[0]
    eps0 = np.asarray(eps0, f8).reshape(-1)[:S]
    eps_w = np.asarray(eps_w, f8).reshape(S, -1)
    group_n = np.asarray(group_n, f8)
    inv_occ = 1.0 / nb_occ.astype(f8)
    if W is None:
        W = batch_norms(x, nb_occ)
    W = np.asarray(W, f8)
    G = len(group_n)
    assert G == F == len(W)

    bx, ex = bia[x].astype(f8), ent[x].astype(f8)       # gather first, then widen
    mu_w, s_w = bx[..., 0], bx[..., 1]                  # [B,F]
    mu_v, s_v = ex[..., :d], ex[..., d:]                # [B,F,d]
    sg_w, sg_v = L(s_w), L(s_v)
    sg0, a = float(L(np.float64(s0))), float(L(np.float64(alpha)))
    ew, ev = eps_w[:, x], eps_v[:, x]                    # [S,B,F], [S,B,F,d]
    w = mu_w + sg_w * ew
    z = mu_v + sg_v * ev
    w0 = m0 + sg0 * eps0                                 # [S]
    sz = z.sum(axis=2)                                   # [S,B,d]
    q = 0.5 * ((sz * sz).sum(axis=2) - (z * z).sum(axis=(2, 3)))
    m = (w.sum(axis=2) + q).mean(axis=0)                 # [B]  (mean over samples BEFORE the likelihood)
    pred = w0[:, None] + m[None, :]                      # [S,B]

    if output == "reg":
        ll = -0.5 * a * (y - pred) ** 2 + 0.5 * math.log(a) - LOG_SQRT_2PI
        dll = a * (y - pred)
        mean = pred
    else:
        ll = y * pred - np.logaddexp(0.0, pred)
        mean = 0.5 * (1.0 + np.tanh(0.5 * pred))          # sigmoid, without overflow for large |pred|
        dll = y - mean

    kl_w = 0.5 * (sg_w ** 2 + mu_w ** 2 - 1.0) - np.log(sg_w)
    kl_v = (0.5 * (sg_v ** 2 + mu_v ** 2 - 1.0) - np.log(sg_v)).sum(axis=2)
    kl_e = kl_w + kl_v                                   # [B,F] per occurrence
    grp = group_of(x, group_hi)                          # [B,F]
    io = inv_occ[x]
    Sg = np.zeros(G, f8)
    np.add.at(Sg, grp.reshape(-1), (kl_e * io).reshape(-1))
    kl0 = 0.5 * (sg0 * sg0 + m0 * m0 - 1.0) - math.log(sg0)
    cscale = group_n / W                                 # n_g / W_g
    kl = kl0 + (cscale * Sg).sum()
    ll_sum = ll.sum()
    scale = nb_train / (Bg * S)
    loss = -scale * ll_sum + kl
    out = {"pred": pred if multi else pred[0], "mean": mean if multi else mean[0], "ll_sum": ll_sum,
           "S": Sg, "W": W, "kl0": kl0, "kl": kl, "loss": loss}
    if not want_grads:
        return out

    gs_ = -scale * dll                                   # dloss/dpred[s,r]  [S,B]
    g = gs_.sum(axis=0)                                  # dloss/dm_r        [B]
    c = cscale[grp] * io                                 # [B,F]
    g_bias = np.zeros(bia.shape, f8)
    g_ent = np.zeros(ent.shape, f8)
    np.add.at(g_bias[:, 0], x, g[:, None] + c * mu_w)
    np.add.at(g_bias[:, 1], x, dL(s_w) * (g[:, None] * ew.mean(axis=0) + c * (sg_w - 1.0 / sg_w)))
    other = sz[:, :, None, :] - z                        # [S,B,F,d] sum over the other fields
    gm = g[:, None, None] * other.mean(axis=0) + c[:, :, None] * mu_v
    gsd = dL(s_v) * (g[:, None, None] * (other * ev).mean(axis=0) + c[:, :, None] * (sg_v - 1.0 / sg_v))
    np.add.at(g_ent, x.reshape(-1), np.concatenate([gm, gsd], axis=2).reshape(B * F, 2 * d))
    gsum_s = gs_.sum(axis=1)                             # [S]
    out["g_global_bias_mean"] = np.array([gsum_s.sum() + m0])
    out["g_global_bias_scale"] = np.array([float(dL(np.float64(s0))) * ((eps0 * gsum_s).sum() + sg0 - 1.0 / sg0)])
    if output == "reg":
        out["g_alpha"] = np.array([float(dL(np.float64(alpha))) * scale *
                                   (0.5 * (y - pred) ** 2 - 0.5 / a).sum()])
    else:
        out["g_alpha"] = np.zeros(1)
    out["g_bias_params"] = g_bias
    out["g_entity_params"] = g_ent
    out["g_row"] = g
    return out


def variant_elbo(P, x, y, nb_occ, group_hi, group_n, nb_train, objective="sampled", priors=None, values=None,
                 eps=None, output="reg"):
    """The ELBO variants of SURVEY 8(f)4 in one torch-fp64 restatement (general F, row-wise sums, autograd):

    objective "sampled": vfm-torch.py:189-324,359 (eps = (eps0[1], eps_w[T], eps_v[T,d]) by entity id);
    objective "closed_form": the expected log-likelihood of vfm-tomasrch.py:369-451 and the loss of :569-588 --
        y_bar = m0 + sum_f v_f mu_w + sum_{f<g} v_f v_g <mu_f, mu_g>,
        T_n = s0^2 + sum_f v_f^2 s_w^2 + sum_{f<g} v_f^2 v_g^2 sum_k (mu_f^2 s_g^2 + mu_g^2 s_f^2 + s_f^2 s_g^2),
        loss = -nb_train/B sum_n [1/2 log|alpha| - |alpha|/2 ((y_n - y_bar_n)^2 + T_n)] + KL terms;
    priors: None = N(0,1), else dict(global=(mean[1], scale[1]), bias=(mean[G], scale[G]), entity=(mean[G,d], scale[G,d]))
        -- the learnable group priors of vfm-tomasrch.py:206-290 (sigma = |scale|), group of an id = first g with
        id < group_hi[g];
    values: None or [B,F] feature values (vfm.py:483-509: x.w + 1/2 sum((x.V)^2 - x^2.V^2); x^2 is squared here --
        the reference's `x2 = x` carries a FIXME that it only holds for 0/1 values).
    P / priors entries may be tensors with requires_grad: returns dict(loss, pred, kl, nll)."""
    f8 = torch.float64
    T = lambda v: torch.as_tensor(v).to(f8) if not torch.is_tensor(v) else v.to(f8)
    x = torch.as_tensor(np.asarray(x), dtype=torch.int64)
    B, F = x.shape
    y = T(y)
    ent, bia = T(P["entity_params"]), T(P["bias_params"])
    d = ent.shape[1] // 2
    alpha, m0, s0 = T(P["alpha"]).reshape(()), T(P["global_bias_mean"]).reshape(()), T(P["global_bias_scale"]).reshape(())
    a, sg0 = alpha.abs(), s0.abs()
    v = torch.ones(B, F, dtype=f8) if values is None else T(values)
    grp = torch.as_tensor(np.searchsorted(np.asarray(group_hi), x.numpy(), side="right"))     # [B,F]
    G = len(group_n)
    mu_w, s_w = bia[x][..., 0], bia[x][..., 1]
    mu_v, s_v = ent[x][..., :d], ent[x][..., d:]
    if objective == "sampled":
        e0, ew, ev = (T(e) for e in eps)
        w0 = m0 + sg0 * e0.reshape(())
        w = mu_w + s_w.abs() * ew[x]
        z = mu_v + s_v.abs() * ev[x]
    else:
        w0, w, z = m0, mu_w, mu_v
    vz = v[..., None] * z
    pred = w0 + (v * w).sum(1) + 0.5 * ((vz.sum(1) ** 2).sum(1) - (vz ** 2).sum((1, 2)))
    if objective == "sampled":
        if output == "reg":
            ll = -0.5 * a * (y - pred) ** 2 + 0.5 * torch.log(a) - LOG_SQRT_2PI
        else:
            ll = y * pred - torch.nn.functional.softplus(pred)
    else:
        am, bs = (v[..., None] ** 2) * mu_v ** 2, (v[..., None] ** 2) * s_v ** 2
        q = am + bs
        t2 = 0.5 * ((q.sum(1) ** 2 - am.sum(1) ** 2).sum(1) - (q ** 2 - am ** 2).sum((1, 2)))
        Tn = s0 ** 2 + (v ** 2 * s_w ** 2).sum(1) + t2
        ll = 0.5 * torch.log(a) - 0.5 * a * ((y - pred) ** 2 + Tn)
    if priors is None:
        pg = (torch.zeros((), dtype=f8), torch.ones((), dtype=f8))
        pw = (torch.zeros(G, dtype=f8), torch.ones(G, dtype=f8))
        pv = (torch.zeros(G, d, dtype=f8), torch.ones(G, d, dtype=f8))
    else:
        pg = tuple(T(t).reshape(()) for t in priors["global"])
        pw = tuple(T(t).reshape(G) for t in priors["bias"])
        pv = tuple(T(t).reshape(G, d) for t in priors["entity"])

    def kl(mu, sg, pm, ps):
        ps = ps.abs()
        return torch.log(ps / sg) + (sg ** 2 + (mu - pm) ** 2) / (2 * ps ** 2) - 0.5
    kl_e = kl(mu_w, s_w.abs(), pw[0][grp], pw[1][grp]) + kl(mu_v, s_v.abs(), pv[0][grp], pv[1][grp]).sum(2)   # [B,F]
    io = 1.0 / T(np.asarray(nb_occ))[x]
    W = io.sum(0)                                          # per column
    cs = T(np.asarray(group_n)) / W
    klr = (kl_e * io * cs[grp]).sum()
    kl0 = kl(m0, sg0, pg[0], pg[1])
    nll = -(nb_train / B) * ll.sum()
    return {"loss": nll + kl0 + klr, "pred": pred, "nll": nll, "kl": kl0 + klr, "kl0": kl0, "partial_loss": ll.sum()}


def pairwise_second_order(z):
    """Explicit sum_{f<g} <z_f, z_g> for z [B,F,d] -- the identity the FM trick
    1/2((sum z)^2 - sum z^2) must satisfy (vfm.py:491-493, vfm-tomasrch.py:379-393)."""
    B, F, d = z.shape
    acc = np.zeros(B)
    for f in range(F):
        for g in range(f + 1, F):
            acc += (z[:, f] * z[:, g]).sum(axis=1)
    return acc


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (defaults of vfm-torch.py:339), numpy.
    t is the 1-based step count."""
    m[:] = b1 * m + (1 - b1) * g
    v[:] = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    p[:] = p - (lr / bc1) * m / (np.sqrt(v) / math.sqrt(bc2) + eps)


def scatter_eps(T, d, uniq, eps_w_u, eps_v_u, dtype=np.float32):
    """Golden fixtures store eps over the sorted unique ids of the batch; the
    kernels (and `rowwise_elbo`) index eps by entity id."""
    ew = np.zeros(T, dtype)
    ev = np.zeros((T, d), dtype)
    ew[uniq] = eps_w_u
    ev[uniq] = eps_v_u
    return ew, ev
