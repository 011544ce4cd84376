"""GPU parity: the HIP kernels through the C ABI vs the reference's golden vectors and vs the
fp64 row-wise oracle.  Tolerances (fp32 path): ELBO 1e-4 relative (north-star), gradients
1e-4 of the largest entry, predictions 1e-4 of the largest entry."""
import numpy as np
import pytest
import torch

from golden_util import Case, SINGLE_CASES, rel_err

pytestmark = pytest.mark.gpu

LOSS_TOL = 1e-4
VEC_TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _setup(c, dev, id_dtype=torch.int64):
    from vae_amd import ops, _lib
    lik = _lib.LIK_NORMAL if c.output == "reg" else _lib.LIK_BERNOULLI
    spec = ops.Spec(T=c.T, F=2, d=c.d, group_hi=tuple(c.group_hi), group_n=tuple(c.group_n),
                    likelihood=lik, nb_train=c.nb_train, n_samples=c.n_samples, link=c.link)
    P = c.params()
    ent = torch.tensor(P["entity_params"], device=dev)
    bia = torch.tensor(P["bias_params"], device=dev)
    scal = torch.tensor(np.concatenate([P["alpha"], P["global_bias_mean"], P["global_bias_scale"]]),
                        device=dev)
    inv_occ = ops.inv_occ_from_counts(torch.tensor(c.nb_occ, device=dev))
    x = torch.tensor(c.x, device=dev).to(id_dtype).contiguous()
    y = torch.tensor(c.y, device=dev)
    e0, ew, ev = c.eps("f32")
    eps = (torch.tensor(ev, device=dev), torch.tensor(ew, device=dev), torch.tensor(e0, device=dev))
    plan = ops.BatchPlan(spec, x, y, inv_occ)
    return ops, spec, plan, ent, bia, scal, inv_occ, eps


@pytest.mark.parametrize("id_dtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("name", SINGLE_CASES)
def test_forward_backward_vs_golden(name, id_dtype):
    dev = _dev()
    c = Case(name)
    ops, spec, plan, ent, bia, scal, inv_occ, eps = _setup(c, dev, id_dtype)
    st = ops.elbo_forward(plan, ent, bia, scal, inv_occ, eps=eps)
    loss3 = ops.elbo_finalize(st, scal)
    gout = torch.ones(1, device=dev)
    g_ent, g_bias, g_sc = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, gout)
    torch.cuda.synchronize()
    assert st.partials[4].item() == 0
    loss = loss3[0].item()
    assert abs(loss - c.expected("loss")[0]) / abs(c.expected("loss")[0]) < LOSS_TOL
    assert abs(loss3[2].item() - c.expected("kl")[0]) / abs(c.expected("kl")[0]) < LOSS_TOL
    pred = st.pred.cpu().numpy()
    if c.output == "class":
        assert rel_err(pred, c.expected("logits")) < VEC_TOL
    else:
        assert rel_err(pred, c.expected("pred")) < VEC_TOL
    assert rel_err(g_ent.cpu().numpy(), c.expected("g_entity_params")) < VEC_TOL
    assert rel_err(g_bias.cpu().numpy(), c.expected("g_bias_params")) < VEC_TOL
    gs = g_sc.cpu().numpy()
    for i, k in enumerate(("g_alpha", "g_global_bias_mean", "g_global_bias_scale")):
        exp = c.expected(k)[0]
        assert abs(gs[i] - exp) <= VEC_TOL * max(abs(exp), 1e-3), k


@pytest.mark.parametrize("name", ["quirk_reg_d8", "ml100k_reg_d20", "ml20m_reg_d128"])
def test_vs_rowwise_oracle_f64(name):
    """Same inputs through the fp64 row-wise oracle: tighter check of every gradient entry."""
    from oracle import vfm_oracle as O
    dev = _dev()
    c = Case(name)
    ops, spec, plan, ent, bia, scal, inv_occ, eps = _setup(c, dev)
    st = ops.elbo_forward(plan, ent, bia, scal, inv_occ, eps=eps)
    loss3 = ops.elbo_finalize(st, scal)
    g_ent, g_bias, g_sc = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=dev))
    e0, ew, ev = c.eps("f32")
    r = O.rowwise_elbo(c.params(), c.x, c.y.astype(np.float64), c.nb_occ, c.group_hi, c.group_n,
                       c.nb_train, e0, ew, ev, c.output)
    assert abs(loss3[0].item() - r["loss"]) / abs(r["loss"]) < 2e-6
    assert rel_err(st.pred.cpu().numpy(), r["pred"]) < 1e-5
    assert rel_err(st.grow.cpu().numpy(), r["g_row"]) < 1e-5
    assert rel_err(g_ent.cpu().numpy(), r["g_entity_params"]) < 2e-5
    assert rel_err(g_bias.cpu().numpy(), r["g_bias_params"]) < 2e-5
    assert rel_err(plan.W.cpu().numpy(), r["W"]) < 1e-6


def test_grad_out_scaling_and_dense_zero_rows():
    dev = _dev()
    c = Case("ml100k_reg_d20")
    ops, spec, plan, ent, bia, scal, inv_occ, eps = _setup(c, dev)
    st = ops.elbo_forward(plan, ent, bia, scal, inv_occ, eps=eps)
    ops.elbo_finalize(st, scal)
    g1 = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=dev))
    g3 = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, torch.full((1,), 3.0, device=dev),
                           g_entity=torch.full_like(ent, 7.0), g_bias=torch.full_like(bia, 7.0))
    for a, b in zip(g1, g3):
        assert torch.allclose(3.0 * a, b, rtol=1e-6, atol=1e-6)
    untouched = torch.ones(c.T, dtype=torch.bool, device=dev)
    untouched[torch.tensor(c.uniq, device=dev)] = False
    assert untouched.any()
    assert (g3[0][untouched] == 0).all() and (g3[1][untouched] == 0).all()


def test_backward_without_reduced_sums_is_loud():
    """The backward needs the forward's slot sums: skipping vfm_elbo_finalize_f32 (and not asking the fused backward
    to do it) yields NaN scalar gradients / scalars, never silently stale sums."""
    dev = _dev()
    c = Case("quirk_reg_d8")
    ops, spec, plan, ent, bia, scal, inv_occ, eps = _setup(c, dev)
    st = ops.elbo_forward(plan, ent, bia, scal, inv_occ, eps=eps)
    assert st.partials[6].item() == 0.0
    g = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=dev))
    assert torch.isnan(g[2][1:]).all()                         # global_bias_mean / scale gradients
    ops.elbo_finalize(st, scal)
    assert st.partials[6].item() == 1.0
    g = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=dev))
    assert torch.isfinite(g[2]).all()
