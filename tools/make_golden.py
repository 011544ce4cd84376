#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's own `CF` class.

Runs ONLY in the build container (needs /root/reference).  Nothing from the
reference is copied into the repo: the class `CF` is located in
`/root/reference/vfm-torch.py` with `ast`, compiled in memory, executed in a
namespace that provides the module globals it reads (vfm-torch.py:18-19,
87-89,125-126), and the *data* it produces (inputs, the epsilon draws, outputs,
gradients, a short Adam trajectory) is written to `tests/golden/*.npz`.

The loss line is vfm-torch.py:359, the optimiser vfm-torch.py:339, the
learning-rate rule vfm-torch.py:92.

Usage:  python tools/make_golden.py [--only name1,name2]     (writes tests/golden/)
(the reference's fp32 sums are not bitwise reproducible from run to run, so regenerate a committed
fixture only on purpose: `--only` limits the run to the named cases)
"""
import ast
import os
import sys

import numpy as np
import torch
from torch import nn, distributions

REF = os.environ.get("VFM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "tests", "golden")


def lift_cf(namespace):
    """Compile the reference's `class CF` in `namespace` (no file is imported)."""
    path = os.path.join(REF, "vfm-torch.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    node = next(n for n in tree.body
                if isinstance(n, ast.ClassDef) and n.name == "CF")
    mod = ast.Module(body=[node], type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return namespace["CF"], (node.lineno, node.end_lineno)


class EpsRecorder:
    """Record every standard-normal draw made by torch.distributions.Normal.rsample
    (torch/distributions/normal.py:83-86 calls `_standard_normal`)."""

    def __init__(self):
        import torch.distributions.normal as tn
        self.tn = tn
        self.orig = tn._standard_normal
        self.draws = []

    def __enter__(self):
        def rec(shape, dtype, device):
            e = self.orig(shape, dtype, device)
            self.draws.append(e.detach().clone())
            return e
        self.tn._standard_normal = rec
        return self

    def __exit__(self, *a):
        self.tn._standard_normal = self.orig


def lift_cf_closed_form(namespace):
    """Compile `class CF` of the reference's vfm-tomasrch.py (G id groups, learnable group priors :206-290,
    closed-form expected log-likelihood :369-451) in `namespace`; its `torchmin` / `rich` imports are outside
    the class and are not needed."""
    path = os.path.join(REF, "vfm-tomasrch.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CF")
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
    return namespace["CF"], (node.lineno, node.end_lineno)


def closed_form_case(name, group_sizes, d, B, nb_train, seed):
    """One step of vfm-tomasrch.py's objective: model(inverse, present, closed_form_loss=True, target) (:548-553)
    and the loss of :569-588 (the branch in effect), with every parameter -- the learnable priors too -- moved
    off its initial value so that all of them matter; outputs + gradients of all parameters."""
    if not wanted(name):
        return
    G, T = len(group_sizes), int(sum(group_sizes))
    ns = {"torch": torch, "nn": nn, "distributions": distributions, "np": np, "N": group_sizes[0], "M": group_sizes[-1]}
    CF, span = lift_cf_closed_form(ns)
    g = np.random.default_rng(seed)
    off = np.concatenate([[0], np.cumsum(group_sizes)[:-1]])
    Xall = np.stack([off[f] + g.integers(0, group_sizes[f], nb_train) for f in range(G)], 1)
    entity_count = torch.bincount(torch.as_tensor(Xall).flatten(), minlength=T).float()      # :182
    x = torch.as_tensor(Xall[:B], dtype=torch.int64)
    y = torch.as_tensor(g.integers(1, 6, B).astype(np.float32))
    torch.manual_seed(seed)
    model = CF(embedding_size=d, n_groups=G, group_sizes=list(group_sizes), alpha_0=1.7, output="reg")
    with torch.no_grad():                      # move everything off the initial point
        tg = torch.Generator().manual_seed(seed + 1)
        rn = lambda *sh: torch.randn(*sh, generator=tg)
        model.bias_params.copy_(torch.cat([0.4 * rn(T, 1), 0.2 + 0.5 * torch.rand(T, 1, generator=tg) * torch.sign(rn(T, 1))], 1))
        model.entity_params.copy_(torch.cat([0.5 * rn(T, d), (0.2 + 0.6 * torch.rand(T, d, generator=tg)) * torch.sign(rn(T, d))], 1))
        model.mean_global_bias.copy_(0.3 * rn(1)); model.scale_global_bias.copy_(torch.tensor([-0.7]))
        model.mean_global_bias_prior.copy_(0.2 * rn(1)); model.scale_global_bias_prior.copy_(torch.tensor([1.3]))
        for i in range(G):
            model.mean_group_bias_prior[i].copy_(0.3 * rn(1)); model.scale_group_bias_prior[i].copy_(0.6 + torch.rand(1, generator=tg))
            model.mean_group_entity_prior[i].copy_(0.3 * rn(d))
            model.scale_group_entity_prior[i].copy_((0.6 + torch.rand(d, generator=tg)) * torch.sign(rn(d)))
    present, inverse, counts = [], [], []
    for i in range(G):                                            # :536-545
        p_, inv, c_ = torch.unique(x[:, i], return_inverse=True, return_counts=True)
        present.append(p_); inverse.append(inv); counts.append(c_)
    outputs, kls, partial_loss = model(inverse, present, closed_form_loss=True, target=y)     # :548-553
    loss = (- nb_train * partial_loss / len(x) + kls[0] + (                                   # :569-588
        (kls[1] + kls[2].sum(axis=1))
        * torch.cat([torch.Tensor(group_sizes[i] / (counts[i] / entity_count[present[i]]).sum()).repeat(len(present[i]))
                     for i in range(G)])
        * torch.concat(counts) / entity_count[torch.concat(present)]).sum())
    model.zero_grad()
    loss.backward()
    rec = {"group_sizes": np.array(group_sizes), "d": d, "nb_train": nb_train, "x": x.numpy(), "y": y.numpy(),
           "nb_occ": entity_count.numpy().astype(np.int64),
           "y_bar": outputs.mean.detach().numpy().reshape(-1), "kl0": kls[0].detach().numpy().reshape(-1),
           "partial_loss": partial_loss.detach().numpy().reshape(-1), "loss": loss.detach().numpy().reshape(-1)}

    def put(key, prm):
        rec["p_" + key] = prm.detach().numpy().copy()
        rec["g_" + key] = (prm.grad.numpy().copy() if prm.grad is not None else np.zeros_like(prm.detach().numpy()))
    put("alpha", model.alpha); put("mean_global_bias", model.mean_global_bias); put("scale_global_bias", model.scale_global_bias)
    put("mean_global_bias_prior", model.mean_global_bias_prior); put("scale_global_bias_prior", model.scale_global_bias_prior)
    put("bias_params", model.bias_params); put("entity_params", model.entity_params)
    for i in range(G):
        put(f"mean_group_bias_prior_{i}", model.mean_group_bias_prior[i]); put(f"scale_group_bias_prior_{i}", model.scale_group_bias_prior[i])
        put(f"mean_group_entity_prior_{i}", model.mean_group_entity_prior[i]); put(f"scale_group_entity_prior_{i}", model.scale_group_entity_prior[i])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: lifted CF of vfm-tomasrch.py lines {span}, G={G} B={B} loss={float(loss)}")


LINKS = {"abs": torch.abs, "softplus": nn.functional.softplus}     # vfm-torch.py:126 / :125
ONLY = None


def wanted(name):
    return ONLY is None or name in ONLY


def make_model(ns, CF, N, M, d, nb_occ, output, dtype, seed, n_samples=1, link="abs"):
    ns.update(N=N, M=M, nb_occ=nb_occ, EMBEDDING_SIZE=d,
              N_VARIATIONAL_SAMPLES=n_samples, LINK=LINKS[link])
    torch.manual_seed(seed)
    model = CF(d, output=output)
    if dtype == torch.float64:
        model = model.double()
    return model


def param_dict(model):
    return {
        "alpha": model.alpha.detach().numpy().copy(),
        "global_bias_mean": model.global_bias_mean.detach().numpy().copy(),
        "global_bias_scale": model.global_bias_scale.detach().numpy().copy(),
        "bias_params": model.bias_params.weight.detach().numpy().copy(),
        "entity_params": model.entity_params.weight.detach().numpy().copy(),
    }


def out_dtype(model):
    return model.alpha.detach().numpy().dtype


def one_step(model, x, y, nb_train):
    """forward + loss (vfm-torch.py:353,359) + backward (:368-369)."""
    with EpsRecorder() as rec:
        lik, _, _, kl = model(x)
    eps0, eps_w, eps_v = rec.draws  # order of vfm-torch.py:238-241
    loss = -lik.log_prob(y.to(kl.dtype)).mean() * nb_train + kl
    model.zero_grad()
    loss.backward()
    uniq = torch.unique(x)
    S = eps0.shape[0]                                       # N_VARIATIONAL_SAMPLES
    out = {
        "uniq": uniq.numpy(),
        "eps0": eps0.numpy().reshape(-1),                   # [S]
        # [U] / [U,d] over sorted uniq; with S > 1 samples: [S,U] / [S,U,d]
        "eps_w": eps_w.numpy().reshape(-1) if S == 1 else eps_w.numpy().reshape(S, -1),
        "eps_v": (eps_v.numpy().reshape(len(uniq), -1) if S == 1
                  else eps_v.numpy().reshape(S, len(uniq), -1)),
        # sigmoid(logit) for 'class'; [B], with S > 1 samples [S,B]
        "pred": (lik.mean.detach().numpy().reshape(-1) if S == 1
                 else lik.mean.detach().numpy().reshape(S, -1)),
        "kl": kl.detach().numpy().reshape(-1),
        "loss": loss.detach().numpy().reshape(-1),
        # alpha takes no part in the Bernoulli likelihood (vfm-torch.py:270): grad None
        "g_alpha": (model.alpha.grad.numpy().copy() if model.alpha.grad is not None
                    else np.zeros(1, dtype=out_dtype(model))),
        "g_global_bias_mean": model.global_bias_mean.grad.numpy().copy(),
        "g_global_bias_scale": model.global_bias_scale.grad.numpy().copy(),
        "g_bias_params": model.bias_params.weight.grad.numpy().copy(),
        "g_entity_params": model.entity_params.weight.grad.numpy().copy(),
    }
    if model.output != "reg":
        out["logits"] = (lik.logits.detach().numpy().reshape(-1) if S == 1
                         else lik.logits.detach().numpy().reshape(S, -1))
    return out


def single_case(ns, CF, name, N, M, d, x, y, nb_train, nb_occ, output,
                seed=42, eps_seed=7, also_f64=True, sparse_rows=False, n_samples=1, link="abs"):
    if not wanted(name):
        return
    x = torch.as_tensor(x, dtype=torch.int64)
    y = torch.as_tensor(y, dtype=torch.float32)
    nb_occ = torch.as_tensor(nb_occ, dtype=torch.int64)
    rec = {"N": N, "M": M, "d": d, "nb_train": nb_train,
           "output": np.array(output), "x": x.numpy(), "y": y.numpy(),
           "nb_occ": nb_occ.numpy()}
    if n_samples != 1 or link != "abs":
        rec["n_samples"], rec["link"] = n_samples, np.array(link)
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        if dtype == torch.float64 and not also_f64:
            continue
        model = make_model(ns, CF, N, M, d, nb_occ, output, dtype, seed, n_samples, link)
        if dtype == torch.float32:
            p = param_dict(model)
        torch.manual_seed(eps_seed)
        out = one_step(model, x, y, nb_train)
        if dtype == torch.float32:
            for k in ("uniq",):
                rec[k] = out[k]
        for k, v in out.items():
            if k == "uniq":
                continue
            rec[f"{tag}_{k}"] = v
    if sparse_rows:
        # big tables: keep only the rows the batch touches (everything else has
        # zero gradient, asserted here, and does not influence any output)
        u = rec["uniq"]
        for tag in ("f32", "f64"):
            for k in ("g_bias_params", "g_entity_params"):
                key = f"{tag}_{k}"
                if key in rec:
                    full = rec[key]
                    mask = np.ones(len(full), bool)
                    mask[u] = False
                    assert not full[mask].any()
                    rec[key] = full[u]
        p["bias_params"] = p["bias_params"][u]
        p["entity_params"] = p["entity_params"][u]
        rec["sparse_rows"] = np.array(1)
    for k, v in p.items():
        rec[f"p_{k}"] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: B={len(y)} U={len(rec['uniq'])} loss_f32={rec['f32_loss']}",
          f"loss_f64={rec.get('f64_loss')}")


def trajectory_case(ns, CF, name, N, M, d, X, Y, batch, output, n_epochs=2,
                    seed=42, eps_seed=11, n_samples=1, link="abs", X_test=None):
    """The loop of vfm-torch.py:347-370 for a few steps (incl. a short last batch).  With `X_test`:
    also the end-of-epoch block (:378-406) -- `save_weights()` for 'reg' (:380, :179-185) and the
    forward over the whole test set (:402), whose draws, sampled prediction and last / mean logits
    (:248-262) are recorded per epoch as e{epoch}_*."""
    if not wanted(name):
        return
    X = torch.as_tensor(X, dtype=torch.int64)
    Y = torch.as_tensor(Y, dtype=torch.float32)
    nb_train = len(Y)
    nb_occ = torch.bincount(X.flatten(), minlength=N + M)   # vfm-torch.py:89
    lr = 1 / (1 + nb_train // batch)                        # vfm-torch.py:92
    model = make_model(ns, CF, N, M, d, nb_occ, output, torch.float32, seed, n_samples, link)
    rec = {"N": N, "M": M, "d": d, "nb_train": nb_train, "batch": batch,
           "lr": lr, "n_epochs": n_epochs, "output": np.array(output),
           "x": X.numpy(), "y": Y.numpy(), "nb_occ": nb_occ.numpy()}
    if n_samples != 1 or link != "abs":
        rec["n_samples"], rec["link"] = n_samples, np.array(link)
    S = n_samples
    for k, v in param_dict(model).items():
        rec[f"p0_{k}"] = v
    opt = torch.optim.Adam(model.parameters(), lr=lr)       # vfm-torch.py:339
    torch.manual_seed(eps_seed)
    step = 0
    losses = []
    for epoch in range(n_epochs):
        for lo in range(0, nb_train, batch):                # DataLoader, no shuffle (:121-122)
            x, y = X[lo:lo + batch], Y[lo:lo + batch]
            with EpsRecorder() as r:
                lik, _, _, kl = model(x)
            loss = -lik.log_prob(y).mean() * nb_train + kl
            opt.zero_grad()
            loss.backward()
            opt.step()
            e0, ew, ev = r.draws
            uniq = torch.unique(x)
            rec[f"s{step}_uniq"] = uniq.numpy()
            rec[f"s{step}_eps0"] = e0.numpy().reshape(-1)
            rec[f"s{step}_eps_w"] = ew.numpy().reshape(-1) if S == 1 else ew.numpy().reshape(S, -1)
            rec[f"s{step}_eps_v"] = (ev.numpy().reshape(len(uniq), -1) if S == 1
                                     else ev.numpy().reshape(S, len(uniq), -1))
            rec[f"s{step}_pred"] = (lik.mean.detach().numpy().reshape(-1) if S == 1
                                    else lik.mean.detach().numpy().reshape(S, -1))
            losses.append(float(loss))
            step += 1
        if X_test is not None:
            if output == "reg":
                model.save_weights()                        # vfm-torch.py:380
            xt = torch.as_tensor(X_test, dtype=torch.int64)
            with EpsRecorder() as r:
                lik, last, mean, _ = model(xt)              # vfm-torch.py:402
            e0, ew, ev = r.draws
            uniq = torch.unique(xt)
            rec[f"e{epoch}_uniq"] = uniq.numpy()
            rec[f"e{epoch}_eps0"] = e0.numpy().reshape(-1)
            rec[f"e{epoch}_eps_w"] = ew.numpy().reshape(-1) if S == 1 else ew.numpy().reshape(S, -1)
            rec[f"e{epoch}_eps_v"] = (ev.numpy().reshape(len(uniq), -1) if S == 1
                                      else ev.numpy().reshape(S, len(uniq), -1))
            rec[f"e{epoch}_pred"] = (lik.mean.detach().numpy().reshape(-1) if S == 1
                                     else lik.mean.detach().numpy().reshape(S, -1))
            if last is not None:                            # numpy arrays (:248-259)
                rec[f"e{epoch}_last_logits"] = np.asarray(last, dtype=np.float32).reshape(-1)
                rec[f"e{epoch}_mean_logits"] = np.asarray(mean, dtype=np.float32).reshape(-1)
    if X_test is not None:
        rec["x_test"] = np.asarray(X_test, dtype=np.int64)
    rec["n_steps"] = step
    rec["losses"] = np.array(losses, dtype=np.float64)
    for k, v in param_dict(model).items():
        rec[f"pT_{k}"] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: steps={step} lr={lr:.4f} losses={losses}")


def long_trajectory_case(ns, CF, name, N, M, d, X, Y, batch, n_steps, checkpoints, seed=42, eps_seed=13):
    """The same loop (vfm-torch.py:347-370) for MANY steps on a small problem -- past a 128-step boundary, where the
    build's scaled-moment Adam re-normalises its buffers -- recording the draws of every step, the losses and the
    weights + Adam moments at a few checkpoints: pins the optimiser form the bench times against torch.optim.Adam
    as the reference runs it."""
    if not wanted(name):
        return
    X = torch.as_tensor(X, dtype=torch.int64)
    Y = torch.as_tensor(Y, dtype=torch.float32)
    nb_train = len(Y)
    nb_occ = torch.bincount(X.flatten(), minlength=N + M)
    lr = 1 / (1 + nb_train // batch)
    model = make_model(ns, CF, N, M, d, nb_occ, "reg", torch.float32, seed)
    rec = {"N": N, "M": M, "d": d, "nb_train": nb_train, "batch": batch, "lr": lr, "n_steps": n_steps,
           "x": X.numpy(), "y": Y.numpy(), "nb_occ": nb_occ.numpy(), "checkpoints": np.array(checkpoints)}
    for k, v in param_dict(model).items():
        rec[f"p0_{k}"] = v
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    torch.manual_seed(eps_seed)
    losses, step = [], 0
    while step < n_steps:
        for lo in range(0, nb_train, batch):
            if step >= n_steps:
                break
            x, y = X[lo:lo + batch], Y[lo:lo + batch]
            with EpsRecorder() as r:
                lik, _, _, kl = model(x)
            loss = -lik.log_prob(y).mean() * nb_train + kl
            opt.zero_grad()
            loss.backward()
            opt.step()
            e0, ew, ev = r.draws
            uniq = torch.unique(x)
            rec[f"s{step}_uniq"] = uniq.numpy().astype(np.int32)
            rec[f"s{step}_eps0"] = e0.numpy().reshape(-1)
            rec[f"s{step}_eps_w"] = ew.numpy().reshape(-1)
            rec[f"s{step}_eps_v"] = ev.numpy().reshape(len(uniq), -1)
            losses.append(float(loss))
            step += 1
            if step in checkpoints:
                for k, v in param_dict(model).items():
                    rec[f"p{step}_{k}"] = v
    rec["losses"] = np.array(losses, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: steps={step} lr={lr:.4f} first/last loss {losses[0]:.2f} {losses[-1]:.2f}")


def main():
    global ONLY
    if "--only" in sys.argv:
        ONLY = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    os.makedirs(OUT, exist_ok=True)
    ns = {"torch": torch, "nn": nn, "distributions": distributions, "np": np}
    CF, span = lift_cf(ns)
    print("lifted CF from vfm-torch.py lines", span)

    # (1) tiny quirk case: N=50, M=30, d=8, B=512; contains item id == N (vfm-torch.py:316)
    g = np.random.default_rng(0)
    N, M, d, B, nb_train = 50, 30, 8, 512, 4000
    Xall = np.stack([g.integers(0, N, nb_train), N + g.integers(0, M, nb_train)], 1)
    Xall[:40, 1] = N                                   # make sure id == N is present
    nb_occ = np.bincount(Xall.reshape(-1), minlength=N + M)
    x = Xall[:B]
    assert (x[:, 1] == N).any() and nb_occ.min() > 0
    y = g.integers(1, 6, B).astype(np.float32)
    single_case(ns, CF, "quirk_reg_d8", N, M, d, x, y, nb_train, nb_occ, "reg")
    # (1b) the reference's two other globals of the path (vfm-torch.py:19,125-126) on the same batch:
    # N_VARIATIONAL_SAMPLES = 3, LINK = softplus, and both
    single_case(ns, CF, "multi_reg_d8_s3", N, M, d, x, y, nb_train, nb_occ, "reg", n_samples=3)
    single_case(ns, CF, "softplus_reg_d8", N, M, d, x, y, nb_train, nb_occ, "reg", link="softplus")
    single_case(ns, CF, "softplus_multi_class_d8_s2", N, M, d, x, (y >= 3).astype(np.float32), nb_train, nb_occ,
                "class", n_samples=2, link="softplus")

    # (2) fraction data set (reference data/fraction/data.csv: user,item,outcome), d=5, Bernoulli
    import pandas as pd
    df = pd.read_csv(os.path.join(REF, "data", "fraction", "data.csv"))
    N, M, d = int(df.user.nunique()), int(df.item.nunique()), 5
    Xall = np.stack([df.user.to_numpy(), df.item.to_numpy() + N], 1)
    Yall = df.outcome.to_numpy().astype(np.float32)
    perm = np.random.default_rng(0).permutation(len(Yall))
    tr = perm[: int(0.8 * len(Yall))]
    nb_occ = np.bincount(Xall[tr].reshape(-1), minlength=N + M)
    keep = tr[(nb_occ[Xall[tr]] > 0).all(1)]
    single_case(ns, CF, "fraction_class_d5", N, M, d, Xall[keep], Yall[keep],
                len(keep), nb_occ, "class")

    # (3) ML-100K-shape, d=20, B=1000 of nb_train=80000
    g = np.random.default_rng(1)
    N, M, d, B, nb_train = 943, 1682, 20, 1000, 80000
    Xall = np.stack([g.integers(0, N, nb_train), N + g.integers(0, M, nb_train)], 1)
    nb_occ = np.bincount(Xall.reshape(-1), minlength=N + M)
    x = Xall[:B]
    y = g.integers(1, 6, B).astype(np.float32)
    single_case(ns, CF, "ml100k_reg_d20", N, M, d, x, y, nb_train, nb_occ, "reg")

    # (3b) same shape, Bernoulli likelihood
    yb = (y >= 4).astype(np.float32)
    single_case(ns, CF, "ml100k_class_d20", N, M, d, x, yb, nb_train, nb_occ, "class",
                also_f64=False)

    # (4) d=128 slice with ML-20M-shape ids, B=256 (only touched rows are stored)
    g = np.random.default_rng(2)
    N, M, d, B, nb_train = 138493, 26744, 128, 256, 16_000_000
    x = np.stack([g.integers(0, N, B), N + g.integers(0, M, B)], 1)
    x[:8, 0] = x[0, 0]                                  # repeated user
    x[8:24, 1] = x[8, 1]                                # repeated item
    nb_occ = g.integers(1, 400, N + M)
    y = (g.integers(1, 11, B) / 2).astype(np.float32)
    single_case(ns, CF, "ml20m_reg_d128", N, M, d, x, y, nb_train, nb_occ, "reg",
                also_f64=False, sparse_rows=True)

    # (4b) few entities, many duplicates (every entity in hundreds of rows): long inverted-index lists
    g = np.random.default_rng(4)
    N, M, d, B, nb_train = 5, 4, 12, 2000, 2000
    x = np.stack([g.integers(0, N, B), N + g.integers(0, M, B)], 1)
    nb_occ = np.bincount(x.reshape(-1), minlength=N + M)
    y = g.integers(1, 6, B).astype(np.float32)
    single_case(ns, CF, "dup_reg_d12", N, M, d, x, y, nb_train, nb_occ, "reg")
    single_case(ns, CF, "dup_class_d12", N, M, d, x, (y >= 3).astype(np.float32), nb_train, nb_occ, "class",
                also_f64=False)
    single_case(ns, CF, "dup_multi_reg_d12_s2", N, M, d, x, y, nb_train, nb_occ, "reg", n_samples=2)

    # (5) 3 batches/epoch (1000,1000,500) x 2 epochs Adam trajectory, short last batch
    g = np.random.default_rng(3)
    N, M, d = 120, 200, 16
    nb = 2500
    X = np.stack([g.integers(0, N, nb), N + g.integers(0, M, nb)], 1)
    Y = g.integers(1, 6, nb).astype(np.float32)
    trajectory_case(ns, CF, "traj_reg_d16", N, M, d, X, Y, 1000, "reg")
    trajectory_case(ns, CF, "traj_softplus_s2_reg_d16", N, M, d, X, Y, 1000, "reg", n_samples=2, link="softplus")

    # (5b) 140 Adam steps on a small problem (4 batches per epoch): across a 128-step boundary
    g = np.random.default_rng(8)
    Nl, Ml, dl, nbl = 40, 60, 8, 1000
    Xl = np.stack([g.integers(0, Nl, nbl), Nl + g.integers(0, Ml, nbl)], 1)
    Yl = g.integers(1, 6, nbl).astype(np.float32)
    long_trajectory_case(ns, CF, "longtraj_reg_d8", Nl, Ml, dl, Xl, Yl, 250, 140, (64, 127, 128, 129, 140))

    # (6) the evaluation block (vfm-torch.py:378-406): 3 epochs x 3 batches + save_weights() + model(X_test)
    g = np.random.default_rng(5)
    nb, nt = 2200, 300
    X = np.stack([g.integers(0, N, nb + nt), N + g.integers(0, M, nb + nt)], 1)
    Y = g.integers(1, 6, nb + nt).astype(np.float32)
    trajectory_case(ns, CF, "eval_reg_d16", N, M, d, X[:nb], Y[:nb], 800, "reg", n_epochs=3, X_test=X[nb:])
    trajectory_case(ns, CF, "eval_class_d16_s2", N, M, d, X[:nb], (Y[:nb] >= 3).astype(np.float32), 800, "class",
                    n_epochs=2, n_samples=2, X_test=X[nb:])

    # (7) the sibling script's objective (vfm-tomasrch.py): closed-form expected log-likelihood + learnable group
    # priors, two groups (user, item) and three (the 'fr_en' layout: format, item, user)
    closed_form_case("cf_reg_d8_g2", [50, 30], 8, 400, 4000, 21)
    closed_form_case("cf_reg_d12_g3", [3, 40, 25], 12, 300, 3000, 22)


if __name__ == "__main__":
    sys.exit(main())
