"""GPU: the ELBO variants of SURVEY 8(f)4 (csrc/vfm_variants.hip, vae_amd/variants.py).

* closed-form expected log-likelihood + learnable group priors: against the reference's own vfm-tomasrch.py `CF`
  (tests/golden/cf_*.npz, tools/make_golden.py) -- loss and the gradient of every parameter, priors included;
* feature values != 1 (vfm.py:483-509: TF1 code that cannot run here -> unpinned by the reference) and every
  other combination: against the fp64 restatement oracle.variant_elbo, which the same goldens pin;
* with unit values, N(0,1) priors and the sampled objective the general kernels must agree with the fused ones."""
import os

import numpy as np
import pytest
import torch

from golden_util import Case, GOLDEN, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cf_reg_d8_g2", "cf_reg_d12_g3"])
def test_closed_form_and_learnable_priors_vs_reference(name):
    from vae_amd.variants import VFMClosedForm
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sizes = [int(v) for v in z["group_sizes"]]
    G = len(sizes)
    m = VFMClosedForm(sizes, int(z["d"]), device="cuda")
    names = ["alpha", "mean_global_bias", "scale_global_bias", "mean_global_bias_prior", "scale_global_bias_prior",
             "bias_params", "entity_params"]
    with torch.no_grad():
        for k in names:
            getattr(m, k).copy_(torch.tensor(z["p_" + k]))
        for i in range(G):
            for k in ("mean_group_bias_prior", "scale_group_bias_prior", "mean_group_entity_prior", "scale_group_entity_prior"):
                getattr(m, k)[i].copy_(torch.tensor(z[f"p_{k}_{i}"]))
    m.set_training_data(torch.tensor(z["x"]), nb_train=int(z["nb_train"]), nb_occ=torch.tensor(z["nb_occ"]))
    loss, y_bar, loss3 = m.elbo(torch.tensor(z["x"]), torch.tensor(z["y"]))
    assert abs(loss.item() - z["loss"][0]) / abs(z["loss"][0]) < 1e-4
    if G == 2:
        assert rel_err(y_bar.cpu().numpy(), z["y_bar"]) < 1e-4
    want_nll = -int(z["nb_train"]) * z["partial_loss"][0] / len(z["y"])
    assert abs(loss3[1].item() - want_nll) / abs(want_nll) < 1e-4
    loss.backward()
    for k in names:
        assert rel_err(getattr(m, k).grad.cpu().numpy(), z["g_" + k]) < 2e-4, k
    for i in range(G):
        for k in ("mean_group_bias_prior", "scale_group_bias_prior", "mean_group_entity_prior", "scale_group_entity_prior"):
            assert rel_err(getattr(m, k)[i].grad.cpu().numpy(), z[f"g_{k}_{i}"]) < 2e-4, (k, i)
    # the prediction-only launch gives the same y_bar
    assert rel_err(m(torch.tensor(z["x"])).cpu().numpy(), y_bar.cpu().numpy()) < 1e-6


def _random_problem(g, F, d, B, output):
    from vae_amd import ops, _lib
    sizes = [int(g.integers(2, 40)) for _ in range(F)]
    T = sum(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    x = np.stack([off[f] + g.integers(0, sizes[f], B) for f in range(F)], 1)
    y = (g.integers(1, 6, B) if output == "reg" else g.integers(0, 2, B)).astype(np.float32)
    nb_occ = np.bincount(x.reshape(-1), minlength=T) + g.integers(1, 4, T)
    P = {"alpha": np.array([g.uniform(0.3, 1.5)], np.float32), "global_bias_mean": np.array([g.normal()], np.float32),
         "global_bias_scale": np.array([g.uniform(0.3, 1.2) * g.choice([-1, 1])], np.float32),
         "bias_params": (0.6 * g.standard_normal((T, 2))).astype(np.float32),
         "entity_params": (0.5 * g.standard_normal((T, 2 * d))).astype(np.float32)}
    hi, gn = np.cumsum(sizes), np.array(sizes, np.float64)
    spec = ops.Spec(T=T, F=F, d=d, group_hi=tuple(int(v) for v in hi), group_n=tuple(gn), nb_train=int(B * 7),
                    likelihood=_lib.LIK_NORMAL if output == "reg" else _lib.LIK_BERNOULLI)
    return sizes, T, x, y, nb_occ, P, hi, gn, spec


@pytest.mark.parametrize("seed", range(12))
def test_variants_vs_oracle(seed):
    """values != 1, learnable priors and both objectives in random combinations (F in 1..5, both likelihoods)
    against oracle.variant_elbo (fp64 autograd): loss, predictions, every gradient.  Seeds 0-5 draw embedding sizes
    that go through the scalar kernels (d % 8 != 0), seeds 6-11 sizes of the lane-group kernels (d % 8 == 0: one
    partly filled and one two-block-per-lane shape among them)."""
    from oracle import vfm_oracle as O
    from vae_amd import ops
    from vae_amd.variants import variant_forward, variant_backward, priors_len
    dev = torch.device("cuda:0")
    g = np.random.default_rng(100 + seed)
    for _ in range(4):
        F, B = int(g.choice([1, 2, 3, 5])), int(g.choice([1, 33, 500]))
        d = int(g.choice([5, 20, 70, 130] if seed < 6 else [8, 16, 64, 136, 520]))
        objective = str(g.choice(["sampled", "closed_form"]))
        output = "reg" if objective == "closed_form" else str(g.choice(["reg", "class"]))
        sizes, T, x, y, nb_occ, P, hi, gn, spec = _random_problem(g, F, d, B, output)
        use_pri, use_val = g.random() < 0.6, g.random() < 0.6
        G = F
        pri_np = None
        if use_pri:
            pri_np = np.concatenate([[g.normal() * 0.3, g.uniform(0.6, 1.5) * g.choice([-1, 1])], 0.3 * g.standard_normal(G),
                                     g.uniform(0.6, 1.5, G), 0.3 * g.standard_normal(G * d),
                                     g.uniform(0.6, 1.5, G * d) * g.choice([-1, 1], G * d)]).astype(np.float32)
            assert pri_np.size == priors_len(G, d)
        vals = g.uniform(0.3, 2.0, (B, F)).astype(np.float32) if use_val else None
        ent, bia = torch.tensor(P["entity_params"], device=dev), torch.tensor(P["bias_params"], device=dev)
        scal = torch.tensor(np.concatenate([P["alpha"], P["global_bias_mean"], P["global_bias_scale"]]), device=dev)
        inv_occ = ops.inv_occ_from_counts(torch.tensor(nb_occ, device=dev))
        plan = ops.BatchPlan(spec, torch.tensor(x, device=dev), torch.tensor(y, device=dev), inv_occ)
        pri = torch.tensor(pri_np, device=dev) if use_pri else None
        v_t = torch.tensor(vals, device=dev) if use_val else None
        seed_, step_ = int(g.integers(0, 2 ** 31)), int(g.integers(0, 10 ** 6))
        st = variant_forward(plan, objective, ent, bia, scal, inv_occ, priors=pri, values=v_t, seed=seed_, step=step_)
        g_ent, g_bias, g_sc, g_pr = variant_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=dev))
        ee, eb, eg = (t.cpu().numpy().astype(np.float64) for t in ops.philox_eps(spec, seed=seed_, step=step_, device=dev))
        leaf = lambda a_: torch.tensor(np.asarray(a_, np.float64), requires_grad=True)
        Pt = {k: leaf(v) for k, v in P.items()}
        prt = None
        if use_pri:
            flat = leaf(pri_np)
            prt = {"global": (flat[0], flat[1]), "bias": (flat[2:2 + G], flat[2 + G:2 + 2 * G]),
                   "entity": (flat[2 + 2 * G:2 + 2 * G + G * d], flat[2 + 2 * G + G * d:])}
        r = O.variant_elbo(Pt, x, y, nb_occ, hi, gn, spec.nb_train, objective, priors=prt, values=vals,
                           eps=(eg, eb, ee), output=output)
        r["loss"].backward()
        cfg = dict(F=F, d=d, B=B, objective=objective, output=output, priors=use_pri, values=use_val)
        assert abs(st["loss3"][0].item() - r["loss"].item()) / abs(r["loss"].item()) < 1e-4, cfg
        assert rel_err(st["pred"].cpu().numpy(), r["pred"].detach().numpy()) < 2e-4, cfg
        tol = 0.2 if F == 1 and objective == "sampled" else 5e-4      # (F = 1 sampled: A - z*sum(g) is pure cancellation)
        assert rel_err(g_ent.cpu().numpy(), Pt["entity_params"].grad.numpy()) < tol, cfg
        assert rel_err(g_bias.cpu().numpy(), Pt["bias_params"].grad.numpy()) < 5e-4, cfg
        want_sc = np.array([0.0 if Pt[k].grad is None else Pt[k].grad.numpy()[0]       # (alpha: unused by Bernoulli)
                            for k in ("alpha", "global_bias_mean", "global_bias_scale")])
        mag = np.abs(r["pred"].detach().numpy()).sum() * spec.nb_train / B + 1.0      # scale of the summed terms
        assert np.all(np.abs(g_sc.cpu().numpy() - want_sc) <= 5e-4 * np.abs(want_sc) + 2e-4 * mag), cfg
        if use_pri:
            assert rel_err(g_pr.cpu().numpy(), flat.grad.numpy()) < 1e-3, cfg


def test_general_kernels_agree_with_the_fused_ones():
    """Sampled objective, unit values, N(0,1) priors: the general pair reproduces the fused kernels' step."""
    from vae_amd import ops, _lib
    from vae_amd.variants import variant_forward, variant_backward
    dev = torch.device("cuda:0")
    c = Case("ml100k_reg_d20")
    spec = ops.Spec(T=c.T, F=2, d=c.d, group_hi=tuple(c.group_hi), group_n=tuple(c.group_n), likelihood=_lib.LIK_NORMAL,
                    nb_train=c.nb_train)
    P = c.params()
    ent, bia = torch.tensor(P["entity_params"], device=dev), torch.tensor(P["bias_params"], device=dev)
    scal = torch.tensor(np.concatenate([P["alpha"], P["global_bias_mean"], P["global_bias_scale"]]), device=dev)
    inv_occ = ops.inv_occ_from_counts(torch.tensor(c.nb_occ, device=dev))
    plan = ops.BatchPlan(spec, torch.tensor(c.x, device=dev), torch.tensor(c.y, device=dev), inv_occ)
    e0, ew, ev = c.eps("f32")
    eps = (torch.tensor(ev, device=dev), torch.tensor(ew, device=dev), torch.tensor(e0, device=dev))
    st = variant_forward(plan, "sampled", ent, bia, scal, inv_occ, eps=eps)
    g_ent, g_bias, g_sc, _ = variant_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=dev))
    assert abs(st["loss3"][0].item() - c.expected("loss")[0]) / abs(c.expected("loss")[0]) < 1e-4
    assert rel_err(g_ent.cpu().numpy(), c.expected("g_entity_params")) < 1e-4
    assert rel_err(g_bias.cpu().numpy(), c.expected("g_bias_params")) < 1e-4
    fs = ops.elbo_forward(plan, ent, bia, scal, inv_occ, eps=eps)
    assert rel_err(st["pred"].cpu().numpy(), fs.pred.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("objective,use_pri,use_val", [("closed_form", True, False), ("sampled", False, True),
                                                       ("closed_form", True, True), ("sampled", True, False)])
def test_lane_group_kernels_agree_with_the_scalar_ones(objective, use_pri, use_val, monkeypatch):
    """d % 8 == 0: the kernels of csrc/vfm_variants8.hpp against the scalar pair of csrc/vfm_variants.hip
    (VFM_VARIANT_SCALAR=1) on a skewed batch (long occurrence lists, three id groups); the prior gradients of the
    lane-group kernels are reproducible bit for bit."""
    from vae_amd import ops, _lib
    from vae_amd.variants import variant_forward, variant_backward, priors_len
    dev = torch.device("cuda:0")
    g = np.random.default_rng(5)
    sizes, d, B, F = [700, 90, 30], 32, 6000, 3
    T = sum(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    x = np.stack([off[f] + np.minimum(g.zipf(1.3, B) - 1, sizes[f] - 1) for f in range(F)], 1)
    y = g.integers(1, 6, B).astype(np.float32)
    nb_occ = np.bincount(x.reshape(-1), minlength=T) + 1
    hi = np.cumsum(sizes)
    spec = ops.Spec(T=T, F=F, d=d, group_hi=tuple(int(v) for v in hi), group_n=tuple(float(v) for v in sizes),
                    nb_train=7 * B, likelihood=_lib.LIK_NORMAL)
    ent = torch.tensor((0.5 * g.standard_normal((T, 2 * d))).astype(np.float32), device=dev)
    bia = torch.tensor((0.6 * g.standard_normal((T, 2))).astype(np.float32), device=dev)
    scal = torch.tensor([0.8, 0.3, -0.7], dtype=torch.float32, device=dev)
    inv_occ = ops.inv_occ_from_counts(torch.tensor(nb_occ, device=dev))
    plan = ops.BatchPlan(spec, torch.tensor(x, device=dev), torch.tensor(y, device=dev), inv_occ)
    pri = None
    if use_pri:
        pri = torch.tensor(np.concatenate([[0.2, -1.1], 0.3 * g.standard_normal(F), g.uniform(0.6, 1.5, F),
                                           0.3 * g.standard_normal(F * d), g.uniform(0.6, 1.5, F * d) * g.choice([-1, 1], F * d)
                                           ]).astype(np.float32), device=dev)
        assert pri.numel() == priors_len(F, d)
    vals = torch.tensor(g.uniform(0.3, 2.0, (B, F)).astype(np.float32), device=dev) if use_val else None
    one = torch.ones(1, device=dev)

    def run():
        st = variant_forward(plan, objective, ent, bia, scal, inv_occ, priors=pri, values=vals, seed=3, step=9)
        return st, variant_backward(plan, st, ent, bia, scal, inv_occ, one)

    st8, g8 = run()
    st8b, g8b = run()
    monkeypatch.setenv("VFM_VARIANT_SCALAR", "1")
    st1, g1 = run()
    assert abs(st8["loss3"][0].item() - st1["loss3"][0].item()) <= 2e-6 * abs(st1["loss3"][0].item())
    assert rel_err(st8["pred"].cpu().numpy(), st1["pred"].cpu().numpy()) < 1e-5
    for a_, b_, name in zip(g8, g1, ("entity", "bias", "scalars", "priors")):
        if a_ is not None:
            assert rel_err(a_.cpu().numpy(), b_.cpu().numpy()) < 2e-5, name
    for a_, b_ in zip(g8, g8b):
        if a_ is not None:
            assert torch.equal(a_, b_)


def test_closed_form_fit_learns():
    """VFMClosedForm.fit (the loop of vfm-tomasrch.py:464-590) lowers its objective on synthetic ratings."""
    from vae_amd.variants import VFMClosedForm
    from vae_amd.data import synthetic_triples
    torch.manual_seed(0)
    X, y = synthetic_triples([60, 40], 4000, seed=2)
    m = VFMClosedForm([60, 40], 4, alpha_0=1.0)
    hist = m.fit(X, y, n_epochs=8, batch_size=1000, lr=0.05)
    assert np.isfinite(hist).all() and hist[-1] < 0.8 * hist[0]
    assert set(dict(m.named_parameters())) >= {"alpha", "mean_global_bias", "scale_global_bias", "bias_params", "entity_params",
                                               "mean_group_entity_prior.0", "scale_group_bias_prior.1"}
