"""GPU parity for the reference's two other globals of the hot path (vfm-torch.py:19,125-126):
N_VARIATIONAL_SAMPLES = S > 1 and LINK = softplus.  Same bars as test_gpu_kernels.py: golden vectors of
the reference's own CF class run with those globals, the fp64 row-wise oracle, the recorded Adam
trajectory; plus equivalences between the code paths (eps tables vs Philox, fused vs unfused Adam,
two fields vs the general-F kernel, long occurrence lists)."""
import os

import numpy as np
import pytest
import torch

from golden_util import Case, VARIANT_CASES, GOLDEN, PARAM_KEYS, rel_err
from test_gpu_kernels import _dev, _setup, LOSS_TOL, VEC_TOL

pytestmark = pytest.mark.gpu


def _run(ops, plan, ent, bia, scal, inv_occ, eps, **kw):
    st = ops.elbo_forward(plan, ent, bia, scal, inv_occ, eps=eps, **kw)
    loss3 = ops.elbo_finalize(st, scal)
    g = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=ent.device))
    torch.cuda.synchronize()
    return st, loss3, g


@pytest.mark.parametrize("id_dtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("name", VARIANT_CASES)
def test_variants_vs_golden(name, id_dtype):
    dev = _dev()
    c = Case(name)
    ops, spec, plan, ent, bia, scal, inv_occ, eps = _setup(c, dev, id_dtype)
    st, loss3, (g_ent, g_bias, g_sc) = _run(ops, plan, ent, bia, scal, inv_occ, eps)
    assert st.partials[4].item() == 0
    exp_loss = c.expected("loss")[0]
    assert abs(loss3[0].item() - exp_loss) / abs(exp_loss) < LOSS_TOL
    assert abs(loss3[2].item() - c.expected("kl")[0]) / abs(c.expected("kl")[0]) < LOSS_TOL
    pred = st.pred.cpu().numpy()
    assert pred.shape == c.expected("pred").shape          # [S,B] for S > 1, as the reference's likelihood.mean
    assert rel_err(pred, c.expected("logits" if c.output == "class" else "pred")) < VEC_TOL
    assert rel_err(g_ent.cpu().numpy(), c.expected("g_entity_params")) < VEC_TOL
    assert rel_err(g_bias.cpu().numpy(), c.expected("g_bias_params")) < VEC_TOL
    gs = g_sc.cpu().numpy()
    for i, k in enumerate(("g_alpha", "g_global_bias_mean", "g_global_bias_scale")):
        exp = c.expected(k)[0]
        assert abs(gs[i] - exp) <= VEC_TOL * max(abs(exp), 1e-3), k


@pytest.mark.parametrize("name", VARIANT_CASES)
def test_variants_vs_rowwise_oracle_f64(name):
    from oracle import vfm_oracle as O
    dev = _dev()
    c = Case(name)
    ops, spec, plan, ent, bia, scal, inv_occ, eps = _setup(c, dev)
    st, loss3, (g_ent, g_bias, g_sc) = _run(ops, plan, ent, bia, scal, inv_occ, eps)
    e0, ew, ev = c.eps("f32")
    r = O.rowwise_elbo(c.params(), c.x, c.y.astype(np.float64), c.nb_occ, c.group_hi, c.group_n,
                       c.nb_train, e0, ew, ev, c.output, link=c.link)
    assert abs(loss3[0].item() - r["loss"]) / abs(r["loss"]) < 2e-6
    assert rel_err(st.pred.cpu().numpy(), r["pred"]) < 1e-5
    assert rel_err(st.grow.cpu().numpy(), r["g_row"]) < 1e-5
    assert rel_err(g_ent.cpu().numpy(), r["g_entity_params"]) < 2e-5
    assert rel_err(g_bias.cpu().numpy(), r["g_bias_params"]) < 2e-5
    gs = g_sc.cpu().numpy()
    for i, k in enumerate(("g_alpha", "g_global_bias_mean", "g_global_bias_scale")):
        assert abs(gs[i] - r[k][0]) <= 2e-5 * max(abs(r[k][0]), 1e-3), k


@pytest.mark.parametrize("name", ["multi_reg_d8_s3", "softplus_multi_class_d8_s2", "dup_multi_reg_d12_s2"])
def test_philox_samples_match_table_mode(name):
    """The in-kernel eps stream of S samples, dumped with vfm_philox_eps_f32 and fed back as tables,
    gives the same step; the samples are distinct draws."""
    dev = _dev()
    c = Case(name)
    ops, spec, plan, ent, bia, scal, inv_occ, _ = _setup(c, dev)
    ee, eb, eg = ops.philox_eps(spec, seed=1234, step=5, device=dev)
    assert ee.shape == (c.n_samples, c.T, c.d) and eg.shape == (c.n_samples,)
    assert not torch.equal(ee[0], ee[1]) and eg[0].item() != eg[1].item()
    a = _run(ops, plan, ent, bia, scal, inv_occ, None, seed=1234, step=5)
    b = _run(ops, plan, ent, bia, scal, inv_occ, (ee, eb, eg))
    assert torch.allclose(a[1], b[1], rtol=1e-6)
    assert torch.allclose(a[0].pred, b[0].pred, rtol=1e-5, atol=1e-5)
    for ga, gb in zip(a[2], b[2]):
        assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-5 * float(gb.abs().max()))
    # sample 0 of an S-sample stream is the single-sample stream
    import dataclasses
    e1 = ops.philox_eps(dataclasses.replace(spec, n_samples=1), seed=1234, step=5, device=dev)
    assert torch.equal(e1[0], ee[0]) and torch.equal(e1[1], eb[0]) and e1[2][0].item() == eg[0].item()


@pytest.mark.parametrize("name", ["multi_reg_d8_s3", "softplus_reg_d8", "dup_multi_reg_d12_s2"])
def test_variants_fused_adam_equals_unfused(name):
    dev = _dev()
    c = Case(name)
    ops, spec, plan, ent, bia, scal, inv_occ, eps = _setup(c, dev)
    st, loss3, (g_ent, g_bias, g_sc) = _run(ops, plan, ent, bia, scal, inv_occ, eps)
    # unfused: dense Adam on copies with the gradients above
    p_ref = [ent.clone(), bia.clone(), scal.clone()]
    for p, g in zip(p_ref, (g_ent, g_bias, g_sc)):
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        pf, gf = p.reshape(-1), g.reshape(-1).contiguous()
        n4 = (pf.numel() + 3) // 4 * 4
        buf = [torch.zeros(n4, device=dev) for _ in range(4)]
        buf[0][: pf.numel()] = pf; buf[1][: pf.numel()] = gf
        ops.adam_step(buf[0], buf[1], buf[2], buf[3], 0.01, 1)
        p.copy_(buf[0][: pf.numel()].reshape(p.shape))
    # fused
    e2, b2, s2 = ent.clone(), bia.clone(), scal.clone()
    mv = [(torch.zeros_like(e2), torch.zeros_like(b2), torch.zeros(3, device=dev)) for _ in range(2)]
    st2 = ops.elbo_forward(plan, e2, b2, s2, inv_occ, eps=eps)
    l3 = torch.zeros(3, device=dev)
    ops.elbo_backward_adam(plan, st2, e2, b2, s2, inv_occ, mv[0], mv[1], 0.01, 1, loss_out=l3)
    torch.cuda.synchronize()
    assert torch.allclose(l3, loss3, rtol=1e-6)
    skip_alpha = c.output == "class"
    assert torch.allclose(e2, p_ref[0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(b2, p_ref[1], rtol=1e-5, atol=1e-6)
    assert torch.allclose(s2[1:], p_ref[2][1:], rtol=1e-5, atol=1e-6)
    if not skip_alpha:
        assert torch.allclose(s2[:1], p_ref[2][:1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("link,S", [("softplus", 1), ("abs", 3), ("softplus", 2)])
@pytest.mark.parametrize("d", [6, 20, 128, 256])
def test_variants_general_F_and_shapes_vs_oracle(link, S, d):
    """F = 3 (general-F kernel) and F = 2, several lane-group shapes, both likelihoods, against the
    fp64 row-wise oracle with the kernels' own eps stream."""
    from oracle import vfm_oracle as O
    from vae_amd import ops, _lib
    dev = _dev()
    g = np.random.default_rng(d + S)
    for F, output in ((3, "reg"), (2, "class")):
        sizes = [40, 30, 20][:F]
        T, B, nb_train = sum(sizes), 700, 5000
        off = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        x = np.stack([off[f] + g.integers(0, sizes[f], B) for f in range(F)], 1)
        y = (g.integers(1, 6, B) if output == "reg" else g.integers(0, 2, B)).astype(np.float32)
        nb_occ = np.bincount(x.reshape(-1), minlength=T) + g.integers(1, 5, T)
        hi, gn = tuple(int(v) for v in np.cumsum(sizes)), tuple(float(s) for s in sizes)
        spec = ops.Spec(T=T, F=F, d=d, group_hi=hi, group_n=gn, nb_train=nb_train, n_samples=S, link=link,
                        likelihood=_lib.LIK_NORMAL if output == "reg" else _lib.LIK_BERNOULLI)
        P = {"alpha": np.array([0.7], np.float32), "global_bias_mean": np.array([0.1], np.float32),
             "global_bias_scale": np.array([-0.8], np.float32),
             "bias_params": g.standard_normal((T, 2)).astype(np.float32),
             "entity_params": (0.3 * g.standard_normal((T, 2 * d))).astype(np.float32)}
        ent, bia = torch.tensor(P["entity_params"], device=dev), torch.tensor(P["bias_params"], device=dev)
        scal = torch.tensor([0.7, 0.1, -0.8], device=dev)
        inv_occ = ops.inv_occ_from_counts(torch.tensor(nb_occ, device=dev))
        plan = ops.BatchPlan(spec, torch.tensor(x, device=dev), torch.tensor(y, device=dev), inv_occ)
        st, loss3, (g_ent, g_bias, g_sc) = _run(ops, plan, ent, bia, scal, inv_occ, None, seed=3, step=9)
        ee, eb, eg = (t.cpu().numpy() for t in ops.philox_eps(spec, seed=3, step=9, device=dev))
        r = O.rowwise_elbo(P, x, y.astype(np.float64), nb_occ, np.array(hi), np.array(gn), nb_train,
                           eg, eb, ee, output, link=link)
        assert abs(loss3[0].item() - r["loss"]) / abs(r["loss"]) < 1e-5, (F, output)
        assert rel_err(st.pred.cpu().numpy(), r["pred"]) < 2e-5
        assert rel_err(g_ent.cpu().numpy(), r["g_entity_params"]) < 1e-4
        assert rel_err(g_bias.cpu().numpy(), r["g_bias_params"]) < 1e-4
        gs = g_sc.cpu().numpy()
        for i, k in enumerate(("g_alpha", "g_global_bias_mean", "g_global_bias_scale")):
            assert abs(gs[i] - r[k][0]) <= 1e-4 * max(abs(r[k][0]), 1e-3), (k, F, output)


def test_trajectory_softplus_two_samples():
    """The reference's 6-step Adam trajectory with LINK = softplus and 2 variational samples, replayed
    through VFM.train_step (fused backward + Adam) with the recorded eps."""
    from vae_amd.model import VFM
    dev = _dev()
    z = np.load(os.path.join(GOLDEN, "traj_softplus_s2_reg_d16.npz"))
    N, M, d, S = int(z["N"]), int(z["M"]), int(z["d"]), int(z["n_samples"])
    T = N + M
    m = VFM(N, M, d, output="reg", device=dev, n_samples=S, link=str(z["link"]))
    with torch.no_grad():
        for k in PARAM_KEYS:
            p = getattr(m, k)
            (p.weight if hasattr(p, "weight") else p).copy_(torch.tensor(z["p0_" + k]))
    X, Y = torch.tensor(z["x"]), torch.tensor(z["y"])
    nb, B = int(z["nb_train"]), int(z["batch"])
    m.set_training_data(X, nb, nb_occ=torch.tensor(z["nb_occ"]))
    m.lr = float(z["lr"])
    step = 0
    for _ in range(int(z["n_epochs"])):
        for lo in range(0, nb, B):
            uniq = z[f"s{step}_uniq"]
            ew = np.zeros((S, T), np.float32); ev = np.zeros((S, T, d), np.float32)
            ew[:, uniq] = z[f"s{step}_eps_w"]; ev[:, uniq] = z[f"s{step}_eps_v"]
            eps = (torch.tensor(ev, device=dev), torch.tensor(ew, device=dev),
                   torch.tensor(z[f"s{step}_eps0"], device=dev))
            plan = m.plan(X[lo:lo + B], Y[lo:lo + B])
            loss3, pred = m.train_step(plan, eps=eps)
            exp = z["losses"][step]
            assert abs(loss3[0].item() - exp) / abs(exp) < 1e-4, step
            assert rel_err(pred.cpu().numpy(), z[f"s{step}_pred"]) < 1e-3
            step += 1
    for k in PARAM_KEYS:
        p = getattr(m, k)
        got = (p.weight if hasattr(p, "weight") else p).detach().cpu().numpy()
        assert rel_err(got, z["pT_" + k]) < 2e-3, k


def test_multi_sample_rejections():
    """Entry points that carry one sample refuse S > 1 loudly; S out of range is invalid."""
    from vae_amd import ops, _lib
    dev = _dev()
    with pytest.raises(ValueError):
        ops.Spec(T=4, F=2, d=4, group_hi=(2, 4), group_n=(2.0, 2.0), likelihood=0, n_samples=65)
    c = Case("multi_reg_d8_s3")
    o, spec, plan, ent, bia, scal, inv_occ, eps = _setup(c, dev)
    st = o.elbo_forward(plan, ent, bia, scal, inv_occ, eps=eps)
    p = st.problem
    import ctypes as C
    lib = _lib.load()
    rc = lib.vfm_elbo_bwd_acc_f32(C.byref(p), None, None, None, None, None, None, None)
    assert rc == -2 and b"n_samples" in lib.vfm_last_error()        # VFM_E_UNSUPPORTED
