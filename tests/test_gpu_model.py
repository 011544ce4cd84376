"""GPU: the VFM module (autograd path, train_step path, fit/predict) vs golden vectors."""
import os

import numpy as np
import pytest
import torch

from golden_util import Case, GOLDEN, PARAM_KEYS, rel_err

pytestmark = pytest.mark.gpu


def _model_from_case(c, dev):
    from vae_amd.model import VFM
    torch.manual_seed(42)
    m = VFM(c.N, c.M, c.d, output=c.output, device=dev)
    P = c.params()
    with torch.no_grad():
        m.entity_params.weight.copy_(torch.tensor(P["entity_params"]))
        m.bias_params.weight.copy_(torch.tensor(P["bias_params"]))
        m.alpha.copy_(torch.tensor(P["alpha"]))
        m.global_bias_mean.copy_(torch.tensor(P["global_bias_mean"]))
        m.global_bias_scale.copy_(torch.tensor(P["global_bias_scale"]))
    m.set_training_data(torch.tensor(c.x), nb_train=c.nb_train, nb_occ=torch.tensor(c.nb_occ))
    return m


def _eps(c, dev):
    e0, ew, ev = c.eps("f32")
    return (torch.tensor(ev, device=dev), torch.tensor(ew, device=dev), torch.tensor(e0, device=dev))


def test_init_matches_reference_seed():
    from vae_amd.model import VFM
    c = Case("quirk_reg_d8")
    torch.manual_seed(42)
    m = VFM(c.N, c.M, c.d, device="cuda")
    P = c.params()
    assert np.array_equal(m.entity_params.weight.detach().cpu().numpy(), P["entity_params"])
    assert np.array_equal(m.bias_params.weight.detach().cpu().numpy(), P["bias_params"])
    assert m.alpha.item() == P["alpha"][0]
    assert set(m.state_dict()) >= {"alpha", "global_bias_mean", "global_bias_scale",
                                   "bias_params.weight", "entity_params.weight"}


@pytest.mark.parametrize("name", ["quirk_reg_d8", "fraction_class_d5", "ml100k_reg_d20"])
def test_autograd_path(name):
    dev = torch.device("cuda:0")
    c = Case(name)
    m = _model_from_case(c, dev)
    loss, pred, detail = m.elbo(torch.tensor(c.x), torch.tensor(c.y), eps=_eps(c, dev))
    loss.backward()
    assert abs(loss.item() - c.expected("loss")[0]) / abs(c.expected("loss")[0]) < 1e-4
    assert rel_err(m.entity_params.weight.grad.cpu().numpy(), c.expected("g_entity_params")) < 1e-4
    assert rel_err(m.bias_params.weight.grad.cpu().numpy(), c.expected("g_bias_params")) < 1e-4
    assert abs(m.global_bias_mean.grad.item() - c.expected("g_global_bias_mean")[0]) <= 1e-4 * abs(
        c.expected("g_global_bias_mean")[0])
    assert m.prec_global_bias_prior.grad is None        # unused params get no grad (reference quirk 7)


# Multi-step checks against the reference's own runs: 5e-4 of the largest entry, the bound the 140-step run
# (test_long_trajectory_vs_reference_adam) holds.  What is compared are fp32 trajectories of two implementations of dense
# Adam that round differently per step (hardware sqrt / rcp in the fused scaled-moment form: 1 ulp each; torch: IEEE), so
# the distance grows with the step count; single-step quantities are held to 1e-4 / 1e-5 elsewhere.
TRAJ_TOL = 5e-4


def test_trajectory_train_step_and_torch_adam():
    """6 Adam steps (3 batches x 2 epochs, short last batch): (a) train_step = HIP fwd/bwd + HIP
    Adam, (b) autograd path + torch.optim.Adam; both must land on the reference's weights."""
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, "traj_reg_d16.npz"))
    from vae_amd.model import VFM
    N, M, d = int(z["N"]), int(z["M"]), int(z["d"])
    nb, B, lr = int(z["nb_train"]), int(z["batch"]), float(z["lr"])
    X, Y = torch.tensor(z["x"]), torch.tensor(z["y"])

    def fresh():
        torch.manual_seed(42)
        m = VFM(N, M, d, device=dev)
        for k in PARAM_KEYS:
            assert np.array_equal(dict(m.state_dict())[k if "params" not in k else k + ".weight"].cpu().numpy(),
                                  z["p0_" + k]), k
        m.set_training_data(X, nb_train=nb)
        assert np.array_equal(m.nb_occ.cpu().numpy(), z["nb_occ"])
        return m

    def eps_of(step):
        T = N + M
        ev = torch.zeros(T, d)
        ew = torch.zeros(T)
        u = torch.tensor(z[f"s{step}_uniq"])
        ev[u] = torch.tensor(z[f"s{step}_eps_v"])
        ew[u] = torch.tensor(z[f"s{step}_eps_w"])
        return ev.to(dev), ew.to(dev), torch.tensor(z[f"s{step}_eps0"]).to(dev)

    for mode in ("hip_fused", "hip_adam", "torch_adam"):
        m = fresh()
        opt = torch.optim.Adam(m.parameters(), lr=lr) if mode == "torch_adam" else None
        step = 0
        for _ in range(int(z["n_epochs"])):
            for lo in range(0, nb, B):
                plan = m.plan(X[lo:lo + B], Y[lo:lo + B])
                if mode.startswith("hip"):
                    loss3, pred = m.train_step(plan, lr=lr, eps=eps_of(step), fused=mode == "hip_fused")
                    loss = loss3[0].item()
                else:
                    l, pred, _ = m.elbo(plan=plan, eps=eps_of(step))
                    opt.zero_grad()
                    l.backward()
                    opt.step()
                    loss = l.item()
                assert abs(loss - z["losses"][step]) / abs(z["losses"][step]) < 1e-4, (mode, step)
                assert rel_err(pred.cpu().numpy(), z[f"s{step}_pred"]) < TRAJ_TOL, (mode, step)
                step += 1
        sd = m.state_dict()
        for k in PARAM_KEYS:
            got = sd[k if "params" not in k else k + ".weight"].cpu().numpy()
            assert rel_err(got, z["pT_" + k]) < TRAJ_TOL, (mode, k)


@pytest.mark.parametrize("mode", ["fused_scaled", "fused_plain", "unfused"])
def test_long_trajectory_vs_reference_adam(mode):
    """140 steps of the reference's loop with torch.optim.Adam (tests/golden/longtraj_reg_d8.npz: the reference's own
    CF class, its recorded draws) -- past the 128-step boundary where the scaled-moment form of the fused kernel
    re-normalises its buffers: every loss, and the weights at steps 64 / 127 / 128 / 129 / 140."""
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, "longtraj_reg_d8.npz"))
    from vae_amd.model import VFM
    N, M, d = int(z["N"]), int(z["M"]), int(z["d"])
    nb, B, lr, T = int(z["nb_train"]), int(z["batch"]), float(z["lr"]), int(z["N"]) + int(z["M"])
    X, Y = torch.tensor(z["x"]), torch.tensor(z["y"])
    torch.manual_seed(42)
    m = VFM(N, M, d, device=dev)
    for k in PARAM_KEYS:
        assert np.array_equal(dict(m.state_dict())[k if "params" not in k else k + ".weight"].cpu().numpy(), z["p0_" + k]), k
    m.set_training_data(X, nb_train=nb)
    m.scaled_moments = mode == "fused_scaled"
    plans = [m.plan(X[lo:lo + B], Y[lo:lo + B]) for lo in range(0, nb, B)]
    checkpoints = set(int(c) for c in z["checkpoints"])
    worst_loss = worst_w = 0.0
    for step in range(int(z["n_steps"])):
        ev, ew = torch.zeros(T, d), torch.zeros(T)
        u = torch.tensor(z[f"s{step}_uniq"]).long()
        ev[u], ew[u] = torch.tensor(z[f"s{step}_eps_v"]), torch.tensor(z[f"s{step}_eps_w"])
        eps = (ev.to(dev), ew.to(dev), torch.tensor(z[f"s{step}_eps0"]).to(dev))
        loss3, _ = m.train_step(plans[step % len(plans)], lr=lr, eps=eps, fused=mode != "unfused")
        worst_loss = max(worst_loss, abs(loss3[0].item() - z["losses"][step]) / abs(z["losses"][step]))
        if step + 1 in checkpoints:
            sd = m.state_dict()
            for k in PARAM_KEYS:
                got = sd[k if "params" not in k else k + ".weight"].cpu().numpy()
                worst_w = max(worst_w, rel_err(got, z[f"p{step + 1}_{k}"]))
    # the trajectory is the reference's to fp32 rounding: at lr = 0.2 a last-bit difference of one step's update
    # is carried, not amplified, by Adam's normalised steps
    print(f"longtraj {mode}: worst relative loss error {worst_loss:.2e}, worst weight error / largest entry {worst_w:.2e}")
    assert worst_loss < 2e-5, worst_loss          # measured 2.2e-6
    assert worst_w < 5e-4, worst_w                # measured 1.0e-4 (scaled moments) / 1.2e-4 of the largest entry


@pytest.mark.parametrize("name", ["eval_reg_d16", "eval_class_d16_s2"])
def test_eval_block_vs_reference(name):
    """The end-of-epoch block of vfm-torch.py:378-417 against the reference's own run (tools/make_golden.py):
    train with the recorded draws, `save_weights()` per epoch ('reg'), then `model(X_test)` -- the sampled
    prediction with the recorded test-time draws, `last_logits` / `mean_logits` (the EPS_ZERO launches on the
    last / epoch-averaged posterior means, :248-259) and the four predictors of `predict()` (:402-417)."""
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    from vae_amd.model import VFM
    N, M, d, output = int(z["N"]), int(z["M"]), int(z["d"]), str(z["output"])
    S = int(z["n_samples"]) if "n_samples" in z.files else 1
    nb, B, lr = int(z["nb_train"]), int(z["batch"]), float(z["lr"])
    X, Y, xt = torch.tensor(z["x"]), torch.tensor(z["y"]), torch.tensor(z["x_test"])
    torch.manual_seed(42)
    m = VFM(N, M, d, output=output, device=dev, n_samples=S)
    for k in PARAM_KEYS:
        assert np.array_equal(dict(m.state_dict())[k if "params" not in k else k + ".weight"].cpu().numpy(), z["p0_" + k])
    m.set_training_data(X, nb_train=nb)
    T = N + M

    def tables(prefix):
        lead = () if S == 1 else (S,)
        ev, ew = torch.zeros(*lead, T, d), torch.zeros(*lead, T)
        u = torch.tensor(z[prefix + "_uniq"])
        ev[..., u, :] = torch.tensor(z[prefix + "_eps_v"])
        ew[..., u] = torch.tensor(z[prefix + "_eps_w"])
        return ev.to(dev), ew.to(dev), torch.tensor(z[prefix + "_eps0"]).to(dev)

    step, all_preds = 0, []
    for epoch in range(int(z["n_epochs"])):
        for lo in range(0, nb, B):
            loss3, _ = m.train_step(m.plan(X[lo:lo + B], Y[lo:lo + B]), lr=lr, eps=tables(f"s{step}"))
            assert abs(loss3[0].item() - z["losses"][step]) / abs(z["losses"][step]) < 1e-4, step
            step += 1
        if output == "reg":
            m.save_weights()                                     # vfm-torch.py:380
        lik, last, mean, kl = m.forward(xt, eps=tables(f"e{epoch}"))
        want = z[f"e{epoch}_pred"]
        assert kl is None
        assert rel_err(lik.mean.cpu().numpy().reshape(want.shape), want) < TRAJ_TOL, epoch
        if output == "reg":
            assert rel_err(last.cpu().numpy(), z[f"e{epoch}_last_logits"]) < TRAJ_TOL
            assert rel_err(mean.cpu().numpy(), z[f"e{epoch}_mean_logits"]) < TRAJ_TOL
            out = m.predict(xt, eps=tables(f"e{epoch}"))         # :402-417
            y_pred = np.clip(want, 1, 5)
            all_preds.append(y_pred)
            assert rel_err(out["y_pred"].cpu().numpy(), y_pred) < TRAJ_TOL
            assert rel_err(out["mean_pred"].cpu().numpy(), np.mean(all_preds, axis=0)) < TRAJ_TOL
            assert rel_err(out["y_pred_of_last"].cpu().numpy(), z[f"e{epoch}_last_logits"]) < TRAJ_TOL   # (not clipped, :402-417)
            assert rel_err(out["y_pred_of_mean"].cpu().numpy(), np.clip(z[f"e{epoch}_mean_logits"], 1, 5)) < TRAJ_TOL
        else:
            assert last is None and mean is None                 # the reference saves weights for 'reg' only
            out = m.predict(xt, eps=tables(f"e{epoch}"))
            assert rel_err(out["y_pred"].cpu().numpy(), want.mean(axis=0)) < TRAJ_TOL


def test_predict_samples_moments_vs_oracle():
    """predict_samples (posterior-predictive mean + logit variance over fresh Philox draws) against the fp64
    oracle fed with the SAME draws (dumped per step by vfm_philox_eps_f32)."""
    from oracle import vfm_oracle as O
    from vae_amd import ops
    dev = torch.device("cuda:0")
    for name, n in (("ml100k_reg_d20", 6), ("fraction_class_d5", 5)):
        c = Case(name)
        m = _model_from_case(c, dev)
        xq = torch.tensor(c.x[:300])
        step0 = m.global_step
        got = m.predict_samples(xq, n_samples=n)
        P = c.params()
        logits = []
        for k in range(n):
            ee, eb, eg = ops.philox_eps(m.spec(), seed=m.rng_seed, step=step0 + k, device=dev)
            r = O.rowwise_elbo(P, c.x[:300], np.zeros(300), c.nb_occ, c.group_hi, c.group_n, c.nb_train,
                               eg.cpu().numpy(), eb.cpu().numpy(), ee.cpu().numpy(), c.output, want_grads=False)
            logits.append(r["pred"])
        L = np.array(logits)
        assert rel_err(got["logits_mean"].cpu().numpy(), L.mean(axis=0)) < 1e-4
        assert rel_err(got["logits_var"].cpu().numpy(), L.var(axis=0, ddof=1)) < 1e-3
        want_mean = L.mean(axis=0) if c.output == "reg" else (1 / (1 + np.exp(-L))).mean(axis=0)
        assert rel_err(got["mean"].cpu().numpy(), want_mean) < 1e-4


def test_fit_predict_fraction_runs_and_learns(tmp_path):
    """cfg 1 plumbing: the shipped toy data (copied fixture of data/fraction/data.csv), d=5, Bernoulli."""
    from vae_amd.model import VFM
    from vae_amd.data import load_fraction
    N, M, Xtr, Xte, ytr, yte = load_fraction(os.path.join(GOLDEN, "fraction"))
    torch.manual_seed(42)
    m = VFM(N, M, 5, output="class", device="cuda")
    hist = m.fit(Xtr, ytr, n_epochs=60, batch_size=100000, X_test=Xte, y_test=yte, display_every=20,
                 verbose=False)
    assert all(np.isfinite(hist["elbo"]))       # lr = 1 here (nb_train < batch, vfm-torch.py:92): noisy
    assert hist["test"][-1]["auc"] > 0.70          # libFM MCMC reaches 0.80 on this set (table.py:21)
    out = m.predict(Xte)
    assert out["y_pred"].shape == (len(yte),) and out["y_pred"].min() >= 0 and out["y_pred"].max() <= 1


def test_predict_reg_four_predictors():
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    torch.manual_seed(0)
    X, y = synthetic_triples([200, 300], 6000, seed=3)
    m = VFM(200, 300, 8, device="cuda")
    m.fit(X[:5000], y[:5000], n_epochs=3, batch_size=2000, verbose=False)
    out = m.predict(X[5000:])
    for k in ("y_pred", "mean_pred", "y_pred_of_last", "y_pred_of_mean"):
        assert out[k].shape == (1000,)
        assert out[k].min() >= 1 - 1e-6 or k == "y_pred_of_last"
    lik, last, mean, kl = m(X[5000:])
    assert lik.mean.shape == (1, 1000) and lik.log_prob(y[5000:].cuda()).shape == (1, 1000)


def test_out_of_range_ids_raise():
    from vae_amd.model import VFM
    m = VFM(10, 10, 4, device="cuda")
    m.set_training_data(torch.tensor([[0, 10]]), nb_train=1)
    with pytest.raises(IndexError):
        m.plan(torch.tensor([[0, 20]]), torch.tensor([1.0]))


def test_philox_mode_matches_table_mode_and_is_standard_normal():
    from vae_amd import ops
    dev = torch.device("cuda:0")
    c = Case("ml100k_reg_d20")
    m = _model_from_case(c, dev)
    spec = m.spec()
    ee, eb, eg = ops.philox_eps(spec, seed=99, step=5, device=dev)
    assert abs(ee.mean().item()) < 0.02 and abs(ee.std().item() - 1) < 0.02
    assert abs((ee ** 4).mean().item() - 3) < 0.15
    ee2, _, _ = ops.philox_eps(spec, seed=99, step=6, device=dev)
    assert abs((ee * ee2).mean().item()) < 0.02            # steps are independent streams
    plan = m.plan(torch.tensor(c.x), torch.tensor(c.y))
    ent, bia, scal = m._views(m._flat)
    a = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, eps=None, seed=99, step=5)
    b = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, eps=(ee, eb, eg))
    la, lb = ops.elbo_finalize(a, scal)[0].item(), ops.elbo_finalize(b, scal)[0].item()
    assert abs(la - lb) / abs(lb) < 1e-6
    ga = ops.elbo_backward(plan, a, ent, bia, scal, m.inv_occ, torch.ones(1, device=dev))
    gb = ops.elbo_backward(plan, b, ent, bia, scal, m.inv_occ, torch.ones(1, device=dev))
    assert rel_err(ga[0].cpu().numpy(), gb[0].cpu().numpy()) < 1e-6


def test_zero_scale_parameter_is_guarded():
    """An Adam update can land a scale parameter on exactly 0.0 (seen at ML-20M shape after ~20
    steps); the reference would fail there.  The kernels keep loss and gradients finite."""
    from vae_amd import ops
    dev = torch.device("cuda:0")
    c = Case("ml100k_reg_d20")
    m = _model_from_case(c, dev)
    e = int(c.x[0, 1])
    with torch.no_grad():
        m.entity_params.weight[e, c.d + 3] = 0.0
        m.bias_params.weight[e, 1] = 0.0
    loss, pred, _ = m.elbo(torch.tensor(c.x), torch.tensor(c.y))
    loss.backward()
    assert torch.isfinite(loss).all()
    for p in (m.entity_params.weight, m.bias_params.weight, m.alpha, m.global_bias_mean, m.global_bias_scale):
        assert torch.isfinite(p.grad).all()
    assert m.entity_params.weight.grad[e, c.d + 3] < 0      # pushed away from 0 (towards +)


def test_sort_within_batch_keeps_batches_and_results():
    from vae_amd.model import VFM, sort_rows_within_batches
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([300, 200], 5000, seed=4, device="cuda")
    Xs, ys = sort_rows_within_batches(X, y, 2000)
    for lo in range(0, 5000, 2000):
        a = torch.cat([X[lo:lo + 2000], y[lo:lo + 2000, None].long()], 1)
        b = torch.cat([Xs[lo:lo + 2000], ys[lo:lo + 2000, None].long()], 1)
        assert torch.equal(a[torch.argsort(a[:, 0] * 1000003 + a[:, 1] * 7 + a[:, 2], stable=True)],
                           b[torch.argsort(b[:, 0] * 1000003 + b[:, 1] * 7 + b[:, 2], stable=True)])
        assert (Xs[lo:lo + 2000, 1][1:] >= Xs[lo:lo + 2000, 1][:-1]).all()
    res = []
    for sort in (False, True):
        torch.manual_seed(1)
        m = VFM(300, 200, 16, device="cuda", rng_seed=5)
        h = m.fit(X, y, n_epochs=2, batch_size=2000, verbose=False, sort_within_batch=sort)
        res.append((h["elbo"], m._flat.clone()))
    assert abs(res[0][0][-1] - res[1][0][-1]) / abs(res[0][0][-1]) < 1e-4
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-3, atol=1e-3)


def test_sparse_adam_opt_in_touches_only_batch_rows():
    dev = torch.device("cuda:0")
    c = Case("ml100k_reg_d20")
    dense, sparse = _model_from_case(c, dev), _model_from_case(c, dev)
    sparse.sparse_adam = True
    plan_d = dense.plan(torch.tensor(c.x), torch.tensor(c.y))
    plan_s = sparse.plan(torch.tensor(c.x), torch.tensor(c.y))
    before = dense.entity_params.weight.detach().clone()
    for _ in range(2):
        dense.train_step(plan_d, lr=0.1, eps=_eps(c, dev))
        sparse.train_step(plan_s, lr=0.1, eps=_eps(c, dev))
    touched = torch.zeros(c.T, dtype=torch.bool, device=dev)
    touched[torch.tensor(c.uniq, device=dev)] = True
    # rows of the batch: identical updates in both modes; other rows: frozen in sparse mode
    assert torch.allclose(dense.entity_params.weight[touched], sparse.entity_params.weight[touched], rtol=1e-4, atol=1e-5)
    assert torch.equal(sparse.entity_params.weight[~touched], before[~touched])
    assert torch.equal(dense.entity_params.weight[~touched], before[~touched])   # zero grads, zero moments: no drift yet


def test_predict_samples_mean_and_variance():
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    torch.manual_seed(0)
    X, y = synthetic_triples([100, 150], 3000, seed=8)
    m = VFM(100, 150, 8, device="cuda", rng_seed=3)
    with torch.no_grad():
        m.entity_params.weight[:, 8:] *= 0.1         # small posterior scales -> small predictive variance
        m.bias_params.weight[:, 1] *= 0.1
        m.global_bias_scale.fill_(0.05)
    out = m.predict_samples(X[:500], n_samples=64)
    det = m(X[:500], sample=False)[0].mean.reshape(-1)
    assert out["logits_var"].min() >= 0 and out["logits_var"].mean() < 1.0
    # the sample mean approaches the deterministic prediction from the posterior means
    assert (out["logits_mean"] - det).abs().mean() < 4 * (out["logits_var"].mean() / 64).sqrt() + 0.05


def test_end_to_end_learning_matches_cpu_reference_port():
    """Train on synthetic low-rank ratings with the HIP path (Philox eps) and with the reference-shaped
    CPU port (torch RNG eps): different noise, same model and loop -> both must learn the data, to a
    similar train/test RMSE (statistical end-to-end parity of fit()/predict())."""
    from vae_amd.model import VFM
    from oracle import vfm_oracle as O
    g = torch.Generator().manual_seed(0)
    N, M, k, n = 150, 200, 4, 24000
    U, V = torch.randn(N, k, generator=g) * 0.8, torch.randn(M, k, generator=g) * 0.8
    bu, bi = torch.randn(N, generator=g) * 0.3, torch.randn(M, generator=g) * 0.3
    u, i = torch.randint(0, N, (n,), generator=g), torch.randint(0, M, (n,), generator=g)
    y = (3.0 + bu[u] + bi[i] + (U[u] * V[i]).sum(1) + 0.1 * torch.randn(n, generator=g)).clamp(1, 5)
    X = torch.stack([u, i + N], 1)
    Xtr, ytr, Xte, yte = X[:20000], y[:20000], X[20000:], y[20000:]
    d, epochs, B = 8, 40, 5000
    torch.manual_seed(42)
    m = VFM(N, M, d, device="cuda", rng_seed=11)
    hist = m.fit(Xtr, ytr, n_epochs=epochs, batch_size=B, X_test=Xte, y_test=yte, display_every=epochs - 1,
                 verbose=False)
    gpu_rmse = hist["test"][-1]["rmse_of_last"]        # deterministic prediction from the final posterior means
    # CPU port, same init distribution, same loop -- for THREE sampler seeds, compared through their median: at this
    # learning rate (1 / (1 + 20000 // 5000) = 0.2) a single 160-step trajectory of the reference's own algorithm is
    # chaotic enough that one seed on one host CPU now and then ends far from the others (seen once in a dozen suite
    # runs on the GPU pool: 1.21 where the same code had read 0.62-0.68 on other hosts)
    nb_occ = torch.bincount(Xtr.flatten(), minlength=N + M)
    cpu = []
    for sampler_seed in (7, 8, 9):
        torch.manual_seed(42)
        P = O.make_params(N + M, d)
        opt = torch.optim.Adam(list(P.values()), lr=1 / (1 + len(ytr) // B))
        torch.manual_seed(sampler_seed)
        for ep in range(epochs):
            for lo in range(0, len(ytr), B):
                O.reference_shaped_step(P, opt, Xtr[lo:lo + B], ytr[lo:lo + B], nb_occ, N, M, len(ytr), "reg")
        with torch.no_grad():
            mu_w, mu_v = P["bias_params"][:, 0], P["entity_params"][:, :d]
            pred = (P["global_bias_mean"] + mu_w[Xte].sum(1) + mu_v[Xte].prod(1).sum(1)).clamp(1, 5)
            cpu.append(float(torch.sqrt(torch.mean((pred - yte) ** 2))))
    cpu_rmse = sorted(cpu)[1]
    base = float(torch.sqrt(torch.mean((ytr.mean() - yte) ** 2)))
    assert gpu_rmse < 0.8 * base and cpu_rmse < 0.8 * base, (gpu_rmse, cpu, base)
    assert abs(gpu_rmse - cpu_rmse) < 0.2 * cpu_rmse, (gpu_rmse, cpu)


def test_philox_stream_statistics():
    """The in-kernel eps stream (Philox4x32-10 + Box-Muller, 16-bit radius / 10-bit angle): moments,
    Kolmogorov distance to N(0,1), no point mass at 0, no correlation between the two normals of a pair,
    between neighbouring entities, or between the embedding and first-order streams."""
    from vae_amd import ops, _lib
    dev = torch.device("cuda:0")
    spec = ops.Spec(T=20000, F=2, d=64, group_hi=(10001, 20000), group_n=(10000.0, 10000.0),
                    likelihood=_lib.LIK_NORMAL, nb_train=1)
    ee, eb, eg = ops.philox_eps(spec, seed=2024, step=3, device=dev)
    x = ee.reshape(-1)
    n = x.numel()
    assert abs(x.mean().item()) < 4 / n ** 0.5 and abs(x.var().item() - 1) < 6 * (2 / n) ** 0.5
    assert abs((x ** 4).mean().item() - 3) < 0.05 and abs((x ** 3).mean().item()) < 0.02
    assert (x == 0).sum().item() == 0
    xs, _ = torch.sort(x.double())
    ks = (torch.special.ndtr(xs) - (torch.arange(n, device=dev, dtype=torch.float64) + 0.5) / n).abs().max().item()
    assert ks < 2.5 / n ** 0.5, ks                       # ~ the 0.1 % critical value of the KS statistic
    assert x.abs().max().item() < 4.9                     # 16-bit radius: sqrt(2 ln 2^17) = 4.85
    assert abs((ee[:, 0::2] * ee[:, 1::2]).mean().item()) < 4 / (n / 2) ** 0.5      # cos / sin partners
    assert abs((ee[:-1] * ee[1:]).mean().item()) < 4 / n ** 0.5                    # neighbouring entities
    assert abs((ee[:, 0] * eb).mean().item()) < 4 / 20000 ** 0.5                   # embedding vs bias stream
    assert abs(eb.mean().item()) < 0.03 and abs(eb.var().item() - 1) < 0.04
    ee2, eb2, _ = ops.philox_eps(spec, seed=2025, step=3, device=dev)              # another seed
    assert abs((ee * ee2).mean().item()) < 4 / n ** 0.5


def test_state_dict_round_trip_and_device_moves():
    """state_dict has the reference's keys; loading it into a fresh model reproduces the predictions;
    .to() keeps the parameters tied to the flat buffer; dtype conversions are refused."""
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([60, 40], 3000, seed=2)
    torch.manual_seed(1)
    a = VFM(60, 40, 12, device="cuda", rng_seed=4)
    a.fit(X, y, n_epochs=2, batch_size=1000, verbose=False)
    sd = {k: v.detach().cpu().clone() for k, v in a.state_dict().items()}
    assert {"alpha", "global_bias_mean", "global_bias_scale", "bias_params.weight", "entity_params.weight",
            "prec_global_bias_prior", "prec_user_entity_prior"} <= set(sd)
    torch.manual_seed(99)
    b = VFM(60, 40, 12, device="cuda", rng_seed=4)
    b.load_state_dict(sd)
    pa = a(X[:500], sample=False)[0].mean
    pb = b(X[:500], sample=False)[0].mean
    assert torch.equal(pa, pb)
    # parameters stay views of ONE flat buffer after a device round trip
    b = b.to("cpu").to("cuda")
    assert b.entity_params.weight.data_ptr() == b._flat.data_ptr()
    assert torch.equal(b(X[:500], sample=False)[0].mean, pa)
    with pytest.raises(TypeError):
        b.double()


def test_checkpoint_resume_is_bit_exact():
    """training_state_dict / load_training_state_dict: stopping after 3 steps and resuming in a fresh
    model continues exactly like the uninterrupted run (parameters, Adam moments, eps stream)."""
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([80, 70], 4000, seed=6)

    def fresh():
        torch.manual_seed(5)
        m = VFM(80, 70, 16, device="cuda", rng_seed=8)
        m.set_training_data(X, nb_train=4000)
        m.lr = 0.05
        return m

    a = fresh()
    plans = [a.plan(X[i:i + 1000], y[i:i + 1000]) for i in range(0, 4000, 1000)]
    for s in range(6):
        a.train_step(plans[s % 4])
    b = fresh()
    for s in range(3):
        b.train_step(plans[s % 4])
    ckpt = b.training_state_dict()
    c = fresh()
    c.load_training_state_dict(ckpt)
    for s in range(3, 6):
        c.train_step(plans[s % 4])
    assert torch.equal(a._flat, c._flat) and torch.equal(a._adam_m, c._adam_m) and torch.equal(a._adam_v, c._adam_v)


def test_checkpoint_resume_with_lagging_rows_is_bit_exact():
    """The same with the look-ahead step form (rows in neither this batch nor the next lag behind until a later batch
    names them): a checkpoint taken while rows lag, resumed in a fresh model, continues bit for bit -- and equals the
    run that never skipped a row."""
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([900, 700], 6000, seed=6)

    def fresh(lookahead=True):
        torch.manual_seed(5)
        m = VFM(900, 700, 16, device="cuda", rng_seed=8)
        m.set_training_data(X, nb_train=6000)
        m.lr, m.lookahead, m.pipeline = 0.05, lookahead, False
        return m

    a = fresh()
    plans = [a.plan(X[i:i + 200], y[i:i + 200]) for i in range(0, 6000, 200)]
    n = len(plans)

    def run(m, lo, hi):
        for s in range(lo, hi):
            m.train_step(plans[s % n], next_plan=plans[(s + 1) % n])

    run(a, 0, 140)                       # (crosses a moment-period boundary)
    assert a._lazy_kind == "la" and a._lazy_dirty
    b = fresh()
    run(b, 0, 70)
    assert b._lazy_dirty
    ckpt = b.training_state_dict()
    c = fresh()
    c.load_training_state_dict(ckpt)
    run(c, 70, 140)
    dense = fresh(lookahead=False)
    run(dense, 0, 140)
    for m in (a, c):
        m.sync_lazy()
    for m in (c, dense):
        assert torch.equal(a._flat, m._flat) and torch.equal(a._adam_m, m._adam_m) and torch.equal(a._adam_v, m._adam_v)


@pytest.mark.parametrize("n_steps", [5, 300])
def test_scaled_moments_equal_plain_dense_adam(n_steps):
    """VFM_FLAG_SCALED_MOMENTS (rows without gradient do not write their moments back) is the same dense
    Adam: parameters AND (converted) moments follow the plain-form run over several periods of 128 steps,
    with batches that leave most rows untouched, and switching forms mid-run is seamless."""
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([300, 200], 2000, seed=3)

    def fresh(scaled):
        torch.manual_seed(9)
        m = VFM(300, 200, 12, device="cuda", rng_seed=4)
        m.set_training_data(X, nb_train=2000)
        m.lr = 0.02
        m.scaled_moments = scaled
        return m

    a, b = fresh(True), fresh(False)
    plans_a = [a.plan(X[i:i + 100], y[i:i + 100]) for i in range(0, 2000, 100)]      # ~150 of 500 rows per batch
    plans_b = [b.plan(X[i:i + 100], y[i:i + 100]) for i in range(0, 2000, 100)]
    for s in range(n_steps):
        la, _ = a.train_step(plans_a[s % 20])
        lb, _ = b.train_step(plans_b[s % 20])
        if s == n_steps // 2:           # a detour through the unfused path converts the form and back
            a.train_step(plans_a[0], fused=False)
            b.train_step(plans_b[0], fused=False)
    assert a._moments_scaled and not b._moments_scaled
    a.sync_lazy()              # (30 % of the rows per batch: "auto" ran model a's steps in the lazy exact-Adam mode)
    a._set_moment_form(False)
    rel = lambda u, v: float((u - v).abs().max() / v.abs().max())
    # the two forms round differently (~1e-7 per step); over hundreds of steps the optimisation
    # trajectory amplifies that like any other ulp-level difference (measured: parameters 8e-6, moments 2e-4
    # after 300 steps; the plain fused vs unfused pair, same arithmetic, sits at 1e-7 / 3e-6)
    tol_p, tol_m = (1e-6, 1e-6) if n_steps <= 5 else (1e-4, 2e-3)
    assert torch.allclose(la, lb, rtol=2e-5)
    assert rel(a._flat, b._flat) < tol_p
    assert rel(a._adam_m, b._adam_m) < tol_m and rel(a._adam_v, b._adam_v) < tol_m


def test_step_is_capturable_in_a_hip_graph():
    """include/vfm_hip.h: launch-only, safe under hipGraph capture.  The forward + fused backward of a skewed batch in a
    large table (the long-list pre-reduction forks to the library's side stream and joins again by events) captured
    once in a graph and replayed twice == the same two calls made eagerly twice, bit for bit."""
    from vae_amd import ops
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    dev = torch.device("cuda:0")
    sizes = [20000, 3000]
    X, y = synthetic_triples(sizes, 30000, seed=2, device=dev, zipf=1.2)

    def fresh():
        torch.manual_seed(1)
        m = VFM(field_sizes=sizes, embedding_size=32, device=dev, rng_seed=5)
        m.set_training_data(X, nb_train=30000)
        m._ensure_opt_state()
        m._set_moment_form(True)
        return m

    def one_step(m, plan, bufs):
        sumz, grow, pred, loss3 = bufs
        ent, bia, scal = m._views(m._flat)
        st = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, seed=5, step=3, train=True, out_pred=pred, out_sumz=sumz,
                              out_grow=grow, out_partials=m._partials)
        ops.elbo_backward_adam(plan, st, ent, bia, scal, m.inv_occ, m._views(m._adam_m), m._views(m._adam_v), 0.01, 1,
                               loss_out=loss3, scaled_moments=True)

    def buffers(m, B):
        return (torch.empty(B, m.d, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev),
                torch.empty(3, device=dev))

    a, b = fresh(), fresh()
    pa, pb = a.plan(X, y), b.plan(X, y)
    assert pa.heavy is not None and pa.heavy[0].numel() * 16 < a.T        # the overlapped path
    ba, bb = buffers(a, 30000), buffers(b, 30000)
    one_step(a, pa, ba)
    one_step(a, pa, ba)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        pb.index_tensors()                      # (nothing but launches inside the capture)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            one_step(b, pb, bb)
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(a._flat, b._flat) and torch.equal(a._adam_m, b._adam_m) and torch.equal(ba[3], bb[3])


@pytest.mark.parametrize("shape", ["lookahead", "pipelined", "dense_small_table"])
def test_streamed_plans_give_the_resident_plans_trajectory(shape):
    """fit(stream_plans=True): no plan is kept -- every batch's index, normalisers and look-ahead row list are built again
    when the batch comes up, two steps ahead on the side stream (plan_async / train_step(prefetch=)) -- and the run is the
    resident-plans run BIT FOR BIT (plans are parameter-free: the same buffers, built on another stream).  The regime of a
    caller that streams or shuffles batches; the reference pays torch.unique x3 inside every step (vfm-torch.py:190-192)."""
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    sizes, d, n, bs = {"lookahead": ([3000, 900], 32, 9000, 1000), "pipelined": ([300, 200], 32, 12000, 3000),
                       "dense_small_table": ([60, 40], 20, 2400, 400)}[shape]
    X, y = synthetic_triples(sizes, n, seed=9, device="cuda")
    Xt, yt = synthetic_triples(sizes, 500, seed=10, device="cuda")
    runs = []
    for stream in (False, True):
        torch.manual_seed(5)
        m = VFM(sizes[0], sizes[1], d, device="cuda", rng_seed=3)
        if shape == "pipelined":
            m.pipeline, m.pipeline_min_T = True, 0
        h = m.fit(X, y, n_epochs=3, batch_size=bs, X_test=Xt, y_test=yt, verbose=False, stream_plans=stream)
        runs.append((m, h))
    (m0, h0), (m1, h1) = runs
    assert h0["elbo"] == h1["elbo"] and h0["train_rmse"] == h1["train_rmse"] and h0["test"] == h1["test"]
    assert torch.equal(m0._flat, m1._flat) and torch.equal(m0._adam_m, m1._adam_m) and torch.equal(m0._mean_flat, m1._mean_flat)
    assert m1._plan_stream is not None and getattr(m0, "_plan_stream", None) is None


@pytest.mark.parametrize("sizes,d,B,kw", [
    ((943, 1682), 20, 8000, dict()),                                   # ML-100K shape (BASELINE configs[1]): ~3 work items per entity
    ((5, 4), 32, 40000, dict()),                                       # nine entities, ~490 work items each: the partial-sum tree
    ((400, 12), 128, 3000, dict()),                                    # d = 128: two lane groups per wave
    ((60, 40), 8, 4000, dict(link="softplus", output="class")),        # d = 8 on the 4-lane shape, softplus link, Bernoulli
    ((300, 200), 64, 500, dict()),                                     # mostly short lists (k_bwd's own walk), some rows not in the batch
    ((943, 1682), 20, 8000, dict(scaled_moments=False)),               # plain moments
    ((60, 40), 256, 4000, dict()),                                     # 64 lanes per entity: k_heavy_sum has FOUR lane groups, fewer than
                                                                       #  VFM_HEAVY_DIRECT -- 5..8 items are k_bwd's in-order sum (fuzz-found)
    ((60, 40), 192, 40000, dict()),                                    # ... and ~45 items each: the four partial sums
])
def test_small_table_step_is_bitwise_the_three_launch_step(sizes, d, B, kw, monkeypatch):
    """k_bwd_small (one launch, a wave per table row, the work items re-derived from the lists) against k_heavy + k_heavy_sum +
    k_bwd<ADAM> on the same plans: losses, parameters and both moments BIT FOR BIT over 140 steps (across a moment-period
    boundary), three batches in turn, with the packed first-order records on.  VFM_BWD_SMALL=0 forces the old path."""
    import vae_amd.model as M
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    monkeypatch.setattr(M, "_CHECK_WREC", True)
    attrs = dict(kw)
    link, output = attrs.pop("link", "abs"), attrs.pop("output", "reg")
    X, y = synthetic_triples(list(sizes), 3 * B, seed=12, device="cuda", output=output, zipf=1.1)
    runs = []
    for small in ("1", "0"):
        monkeypatch.setenv("VFM_BWD_SMALL", small)
        torch.manual_seed(2)
        m = VFM(field_sizes=list(sizes), embedding_size=d, device="cuda", rng_seed=8, link=link, output=output)
        m.lookahead = m.pipeline = False
        m.lazy_adam = False
        for k_, v_ in attrs.items():
            setattr(m, k_, v_)
        m.set_training_data(X, nb_train=3 * B)
        plans = [m.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(3)]
        losses = [m.train_step(plans[s % 3], lr=0.03)[0].clone() for s in range(140)]
        for pl in plans:
            pl.check_status()
        runs.append((m, torch.stack(losses)))
    (a, la), (b, lb) = runs
    assert torch.equal(la, lb)
    assert torch.equal(a._flat, b._flat) and torch.equal(a._adam_m, b._adam_m) and torch.equal(a._adam_v, b._adam_v)
    assert not torch.isnan(a._flat).any()
