"""CPU: the oracle (oracle/vfm_oracle.py) against the golden vectors produced by the
reference's own CF class (tools/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from golden_util import Case, SINGLE_CASES, F64_CASES, VARIANT_CASES, PARAM_KEYS, rel_err
from oracle import vfm_oracle as O

GRADS = ("g_alpha", "g_global_bias_mean", "g_global_bias_scale", "g_bias_params",
         "g_entity_params")


def _torch_params(P, dtype):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in P.items()}


@pytest.mark.parametrize("name", SINGLE_CASES + VARIANT_CASES)
def test_reference_shaped_f32(name):
    c = Case(name)
    P = _torch_params(c.params(), torch.float32)
    loss, lik, kl = O.reference_shaped_loss(
        P, torch.tensor(c.x), torch.tensor(c.y), torch.tensor(c.nb_occ), c.N, c.M,
        c.nb_train, c.output, eps=c.eps_uniq("f32"), n_samples=c.n_samples, link=c.link)
    loss.backward()
    assert rel_err(loss.detach().numpy(), c.expected("loss")) < 1e-6
    assert rel_err(kl.detach().numpy(), c.expected("kl")) < 1e-6
    assert rel_err(lik.mean.detach().numpy().reshape(-1), c.expected("pred").reshape(-1)) < 1e-6
    for g in GRADS:
        got = P[g[2:]].grad
        got = np.zeros(1, np.float32) if got is None else got.numpy()
        assert rel_err(got, c.expected(g)) < 1e-5, g


@pytest.mark.parametrize("name", F64_CASES + VARIANT_CASES)
def test_reference_shaped_f64(name):
    c = Case(name)
    P = _torch_params(c.params(np.float64), torch.float64)
    loss, lik, kl = O.reference_shaped_loss(
        P, torch.tensor(c.x), torch.tensor(c.y), torch.tensor(c.nb_occ), c.N, c.M,
        c.nb_train, c.output, eps=c.eps_uniq("f64"), n_samples=c.n_samples, link=c.link)
    loss.backward()
    assert rel_err(loss.detach().numpy(), c.expected("loss", "f64")) < 1e-12
    for g in GRADS:
        got = P[g[2:]].grad
        got = np.zeros(1) if got is None else got.numpy()
        assert rel_err(got, c.expected(g, "f64")) < 1e-11, g


@pytest.mark.parametrize("name", F64_CASES + VARIANT_CASES)
def test_rowwise_f64(name):
    """Row-wise identity == the reference's unique-based form.  Residual: the
    reference forms cnt/occ in fp32 even for a double model (int64/int64 -> fp32)."""
    c = Case(name)
    e0, ew, ev = c.eps("f64")
    r = O.rowwise_elbo(c.params(np.float64), c.x, c.y.astype(np.float64), c.nb_occ,
                       c.group_hi, c.group_n, c.nb_train, e0, ew, ev, c.output, link=c.link)
    assert rel_err(r["loss"], c.expected("loss", "f64")) < 1e-7
    assert rel_err(r["kl"], c.expected("kl", "f64")) < 1e-6
    pred = r["pred"] if c.output == "reg" else r["mean"]
    assert rel_err(pred, c.expected("pred", "f64")) < 1e-12
    for g in GRADS:
        assert rel_err(r[g], c.expected(g, "f64")) < 1e-6, g


@pytest.mark.parametrize("name", SINGLE_CASES + VARIANT_CASES)
def test_rowwise_vs_f32_golden(name):
    """fp64 row-wise restatement on the fp32 inputs vs the reference's fp32 outputs:
    bounds the fp32 rounding of the reference itself (tolerance budget for the kernels)."""
    c = Case(name)
    e0, ew, ev = c.eps("f32")
    r = O.rowwise_elbo(c.params(), c.x, c.y.astype(np.float64), c.nb_occ,
                       c.group_hi, c.group_n, c.nb_train, e0, ew, ev, c.output, link=c.link)
    assert rel_err(r["loss"], c.expected("loss")) < 2e-6
    pred = r["pred"] if c.output == "reg" else r["mean"]
    assert rel_err(pred, c.expected("pred")) < 1e-5
    for g in GRADS:
        assert rel_err(r[g], c.expected(g)) < 2e-5, g


def test_quirk_case_contains_boundary_id():
    c = Case("quirk_reg_d8")
    assert (c.x[:, 1] == c.N).any()
    # with the "clean" grouping (id < N) the loss must differ -> the quirk is exercised
    e0, ew, ev = c.eps("f64")
    P = c.params(np.float64)
    a = O.rowwise_elbo(P, c.x, c.y, c.nb_occ, c.group_hi, c.group_n, c.nb_train, e0, ew, ev,
                       want_grads=False)["loss"]
    b = O.rowwise_elbo(P, c.x, c.y, c.nb_occ, np.array([c.N, c.T]), c.group_n, c.nb_train,
                       e0, ew, ev, want_grads=False)["loss"]
    assert abs(a - b) / abs(a) > 1e-9


def test_fm_identity_general_F():
    """1/2((sum_f z)^2 - sum_f z^2) == sum_{f<g} <z_f,z_g>; at F=2 == prod().sum()
    (vfm-torch.py:245)."""
    g = np.random.default_rng(0)
    for F in (2, 3, 5, 32):
        z = g.standard_normal((17, F, 12))
        sz = z.sum(1)
        trick = 0.5 * ((sz * sz).sum(1) - (z * z).sum((1, 2)))
        np.testing.assert_allclose(trick, O.pairwise_second_order(z), rtol=1e-10, atol=1e-10)
    z = g.standard_normal((9, 2, 7))
    np.testing.assert_allclose(O.pairwise_second_order(z), z.prod(1).sum(1), rtol=1e-12)


def test_rowwise_grads_match_autograd_general_F():
    """F=4 fields (no reference oracle for F>2): analytic grads vs torch autograd of the
    same row-wise loss in float64."""
    g = np.random.default_rng(5)
    F, d, B = 4, 6, 64
    sizes = [7, 5, 9, 4]
    hi = np.cumsum(sizes)
    T = int(hi[-1])
    lo = hi - sizes
    x = np.stack([g.integers(lo[f], hi[f], B) for f in range(F)], 1)
    nb_occ = g.integers(1, 50, T)
    y = g.standard_normal(B)
    P = {"alpha": np.array([0.7]), "global_bias_mean": np.array([0.1]),
         "global_bias_scale": np.array([-0.8]),
         "bias_params": g.standard_normal((T, 2)), "entity_params": g.standard_normal((T, 2 * d))}
    e0, ew, ev = g.standard_normal(1), g.standard_normal(T), g.standard_normal((T, d))
    nb_train = 1000
    for output in ("reg", "class"):
        yy = y if output == "reg" else (y > 0).astype(np.float64)
        r = O.rowwise_elbo(P, x, yy, nb_occ, hi, np.array(sizes, float), nb_train, e0, ew, ev, output)
        # autograd restatement
        tp = {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}
        xt = torch.tensor(x)
        mu_w, s_w = tp["bias_params"][xt][..., 0], tp["bias_params"][xt][..., 1]
        E = tp["entity_params"][xt]
        mu_v, s_v = E[..., :d], E[..., d:]
        w = mu_w + s_w.abs() * torch.tensor(ew)[xt]
        z = mu_v + s_v.abs() * torch.tensor(ev)[xt]
        w0 = tp["global_bias_mean"] + tp["global_bias_scale"].abs() * torch.tensor(e0)
        pred = w0 + w.sum(1) + 0.5 * ((z.sum(1) ** 2).sum(1) - (z ** 2).sum((1, 2)))
        yt = torch.tensor(yy)
        if output == "reg":
            ll = torch.distributions.Normal(pred, torch.sqrt(1 / tp["alpha"].abs())).log_prob(yt)
        else:
            ll = torch.distributions.Bernoulli(logits=pred).log_prob(yt)
        kl_e = (0.5 * (s_w ** 2 + mu_w ** 2 - 1) - s_w.abs().log()
                + (0.5 * (s_v ** 2 + mu_v ** 2 - 1) - s_v.abs().log()).sum(2))
        io = 1.0 / torch.tensor(nb_occ, dtype=torch.float64)[xt]
        Wt = io.sum(0)
        kl = (kl_e * io * (torch.tensor(sizes, dtype=torch.float64) / Wt)).sum()
        s0 = tp["global_bias_scale"]
        kl0 = 0.5 * (s0 ** 2 + tp["global_bias_mean"] ** 2 - 1) - s0.abs().log()
        loss = -(nb_train / B) * ll.sum() + kl0 + kl
        loss.backward()
        assert rel_err(r["loss"], loss.item()) < 1e-12
        for k in PARAM_KEYS:
            got = tp[k].grad
            got = np.zeros(1) if got is None else got.numpy()
            assert rel_err(r["g_" + k], got) < 1e-10, (output, k)


def test_adam_step_matches_torch():
    g = np.random.default_rng(1)
    p = g.standard_normal((5, 3))
    pt = torch.tensor(p.copy(), requires_grad=True)
    opt = torch.optim.Adam([pt], lr=0.25)
    m, v = np.zeros_like(p), np.zeros_like(p)
    for t in range(1, 5):
        grad = g.standard_normal((5, 3))
        pt.grad = torch.tensor(grad)
        opt.step()
        O.adam_step(p, grad, m, v, t, 0.25)
        np.testing.assert_allclose(p, pt.detach().numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", ["traj_reg_d16", "traj_softplus_s2_reg_d16"])
def test_trajectory_fixture_reference_shaped(name):
    """The 6-step Adam trajectory (incl. a short last batch) replayed with the
    reference-shaped restatement reproduces the reference's losses and final weights."""
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, M = int(z["N"]), int(z["M"])
    S = int(z["n_samples"]) if "n_samples" in z.files else 1
    link = str(z["link"]) if "link" in z.files else "abs"
    P = {k: torch.tensor(z["p0_" + k], requires_grad=True) for k in PARAM_KEYS}
    opt = torch.optim.Adam(list(P.values()), lr=float(z["lr"]))
    X, Y, occ = torch.tensor(z["x"]), torch.tensor(z["y"]), torch.tensor(z["nb_occ"])
    nb, B = int(z["nb_train"]), int(z["batch"])
    step = 0
    for _ in range(int(z["n_epochs"])):
        for lo in range(0, nb, B):
            eps = (z[f"s{step}_eps0"], z[f"s{step}_eps_w"], z[f"s{step}_eps_v"])
            loss, mean = O.reference_shaped_step(P, opt, X[lo:lo + B], Y[lo:lo + B], occ, N, M,
                                                 nb, "reg", eps, n_samples=S, link=link)
            assert abs(loss.item() - z["losses"][step]) / abs(z["losses"][step]) < 1e-5
            step += 1
    for k in PARAM_KEYS:
        assert rel_err(P[k].detach().numpy(), z["pT_" + k]) < 1e-4, k


def test_long_trajectory_fixture_reference_shaped():
    """The reference's 140-step Adam run (tests/golden/longtraj_reg_d8.npz) replayed with the reference-shaped
    restatement: every loss and the weights at the recorded checkpoints."""
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "longtraj_reg_d8.npz"))
    N, M = int(z["N"]), int(z["M"])
    P = {k: torch.tensor(z["p0_" + k], requires_grad=True) for k in PARAM_KEYS}
    opt = torch.optim.Adam(list(P.values()), lr=float(z["lr"]))
    X, Y, occ = torch.tensor(z["x"]), torch.tensor(z["y"]), torch.tensor(z["nb_occ"])
    nb, B = int(z["nb_train"]), int(z["batch"])
    checkpoints = set(int(c) for c in z["checkpoints"])
    for step in range(int(z["n_steps"])):
        lo = (step * B) % nb
        eps = (z[f"s{step}_eps0"], z[f"s{step}_eps_w"], z[f"s{step}_eps_v"])
        loss, _ = O.reference_shaped_step(P, opt, X[lo:lo + B], Y[lo:lo + B], occ, N, M, nb, "reg", eps)
        assert abs(loss.item() - z["losses"][step]) / abs(z["losses"][step]) < 1e-5, step
        if step + 1 in checkpoints:
            for k in PARAM_KEYS:
                assert rel_err(P[k].detach().numpy(), z[f"p{step + 1}_{k}"]) < 1e-4, (step + 1, k)


@pytest.mark.parametrize("name", ["eval_reg_d16", "eval_class_d16_s2"])
def test_eval_block_fixture_reference_shaped(name):
    """The evaluation block (vfm-torch.py:378-406) replayed with the reference-shaped restatement: after each
    epoch `save_weights` (:179-185, 'reg' only) and the forward over the test rows (:402) reproduce the
    reference's sampled predictions and its last / mean logits (:248-259)."""
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, M, output = int(z["N"]), int(z["M"]), str(z["output"])
    S = int(z["n_samples"]) if "n_samples" in z.files else 1
    P = {k: torch.tensor(z["p0_" + k], requires_grad=True) for k in PARAM_KEYS}
    opt = torch.optim.Adam(list(P.values()), lr=float(z["lr"]))
    X, Y, occ = torch.tensor(z["x"]), torch.tensor(z["y"]), torch.tensor(z["nb_occ"])
    xt = torch.tensor(z["x_test"])
    nb, B = int(z["nb_train"]), int(z["batch"])
    saved = O.SavedWeights()
    step = 0
    for epoch in range(int(z["n_epochs"])):
        for lo in range(0, nb, B):
            eps = (z[f"s{step}_eps0"], z[f"s{step}_eps_w"], z[f"s{step}_eps_v"])
            loss, _ = O.reference_shaped_step(P, opt, X[lo:lo + B], Y[lo:lo + B], occ, N, M, nb, output, eps,
                                              n_samples=S)
            assert abs(loss.item() - z["losses"][step]) / abs(z["losses"][step]) < 1e-5
            step += 1
        if output == "reg":
            saved.save(P)
        eps = (z[f"e{epoch}_eps0"], z[f"e{epoch}_eps_w"], z[f"e{epoch}_eps_v"])
        with torch.no_grad():
            lik, _ = O.reference_shaped_forward(P, xt, occ, N, M, output, eps, n_samples=S)
        want = z[f"e{epoch}_pred"]
        assert rel_err(lik.mean.numpy().reshape(want.shape), want) < 1e-4
        if output == "reg":
            last, mean = saved.logits(z["x_test"])
            assert rel_err(last, z[f"e{epoch}_last_logits"]) < 1e-4
            assert rel_err(mean, z[f"e{epoch}_mean_logits"]) < 1e-4
        else:
            assert f"e{epoch}_last_logits" not in z.files      # the reference saves weights for 'reg' only (:378-380)


def _cf_case(name):
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sizes = [int(v) for v in z["group_sizes"]]
    G, d = len(sizes), int(z["d"])
    leaf = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    P = {"alpha": leaf(z["p_alpha"]), "global_bias_mean": leaf(z["p_mean_global_bias"]),
         "global_bias_scale": leaf(z["p_scale_global_bias"]), "bias_params": leaf(z["p_bias_params"]),
         "entity_params": leaf(z["p_entity_params"])}
    pri = {"global": (leaf(z["p_mean_global_bias_prior"]), leaf(z["p_scale_global_bias_prior"])),
           "bias": (leaf(np.concatenate([z[f"p_mean_group_bias_prior_{i}"] for i in range(G)])),
                    leaf(np.concatenate([z[f"p_scale_group_bias_prior_{i}"] for i in range(G)]))),
           "entity": (leaf(np.stack([z[f"p_mean_group_entity_prior_{i}"] for i in range(G)])),
                      leaf(np.stack([z[f"p_scale_group_entity_prior_{i}"] for i in range(G)])))}
    return z, sizes, G, d, P, pri


@pytest.mark.parametrize("name", ["cf_reg_d8_g2", "cf_reg_d12_g3"])
def test_closed_form_variant_vs_reference(name):
    """The sibling script's objective -- closed-form expected log-likelihood (vfm-tomasrch.py:369-451), learnable
    group priors (:206-290), loss of :569-588 -- restated row-wise (O.variant_elbo) against the reference's own
    class run by tools/make_golden.py: loss, y_bar and the gradient of EVERY parameter, priors included."""
    z, sizes, G, d, P, pri = _cf_case(name)
    r = O.variant_elbo(P, z["x"], z["y"], z["nb_occ"], np.cumsum(sizes), np.array(sizes, np.float64), int(z["nb_train"]),
                       objective="closed_form", priors=pri)
    assert abs(r["loss"].item() - z["loss"][0]) / abs(z["loss"][0]) < 1e-5
    if G == 2:      # (`likelihood.mean` of the reference multiplies ALL groups' embeddings, :342-347: it is the
        assert rel_err(r["pred"].detach().numpy(), z["y_bar"]) < 1e-5     # pairwise y_bar of :369-393 only for G = 2)
    assert abs(r["partial_loss"].item() - z["partial_loss"][0]) / abs(z["partial_loss"][0]) < 1e-5
    assert abs(r["kl0"].item() - z["kl0"][0]) < 1e-5 * max(1.0, abs(z["kl0"][0]))
    r["loss"].backward()
    for key, t in (("alpha", P["alpha"]), ("mean_global_bias", P["global_bias_mean"]),
                   ("scale_global_bias", P["global_bias_scale"]), ("bias_params", P["bias_params"]),
                   ("entity_params", P["entity_params"]), ("mean_global_bias_prior", pri["global"][0]),
                   ("scale_global_bias_prior", pri["global"][1])):
        assert rel_err(t.grad.numpy(), z["g_" + key]) < 2e-4, key
    for i in range(G):
        assert rel_err(pri["bias"][0].grad.numpy()[i:i + 1], z[f"g_mean_group_bias_prior_{i}"]) < 2e-4
        assert rel_err(pri["bias"][1].grad.numpy()[i:i + 1], z[f"g_scale_group_bias_prior_{i}"]) < 2e-4
        assert rel_err(pri["entity"][0].grad.numpy()[i], z[f"g_mean_group_entity_prior_{i}"]) < 2e-4
        assert rel_err(pri["entity"][1].grad.numpy()[i], z[f"g_scale_group_entity_prior_{i}"]) < 2e-4


def test_variant_oracle_reduces_to_the_main_path():
    """variant_elbo with N(0,1) priors, unit values and the sampled objective IS the main path: equal to
    rowwise_elbo (which the vfm-torch.py goldens pin); and feature values: v = 2 on one field equals doubling that
    field's first-order weight and embedding."""
    c = Case("quirk_reg_d8")
    P = c.params(np.float64)
    e0, ew, ev = c.eps("f64")
    want = O.rowwise_elbo(P, c.x, c.y.astype(np.float64), c.nb_occ, c.group_hi, c.group_n, c.nb_train, e0, ew, ev, "reg")
    got = O.variant_elbo(P, c.x, c.y, c.nb_occ, c.group_hi, c.group_n, c.nb_train, "sampled", eps=(e0, ew, ev))
    assert abs(got["loss"].item() - want["loss"]) / abs(want["loss"]) < 1e-10
    assert rel_err(got["pred"].numpy(), want["pred"]) < 1e-10
    vals = np.ones(c.x.shape); vals[:, 1] = 2.0
    P2 = {k: v.copy() for k, v in P.items()}
    items = np.unique(c.x[:, 1])
    P2["bias_params"][items, 0] *= 2.0
    P2["entity_params"][items, :c.d] *= 2.0
    zero = (np.zeros(1), np.zeros(c.T), np.zeros((c.T, c.d)))
    a = O.variant_elbo(P, c.x, c.y, c.nb_occ, c.group_hi, c.group_n, c.nb_train, "sampled", values=vals, eps=zero)
    b = O.variant_elbo(P2, c.x, c.y, c.nb_occ, c.group_hi, c.group_n, c.nb_train, "sampled", eps=zero)
    assert rel_err(a["pred"].numpy(), b["pred"].numpy()) < 1e-12
