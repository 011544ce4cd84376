"""GPU: every kernel shape (lanes per row, chunks per lane, vector width), general number of fields,
id widths, edge cases -- against the fp64 row-wise oracle on seeded random inputs."""
import numpy as np
import pytest
import torch

from golden_util import rel_err
from oracle import vfm_oracle as O

pytestmark = pytest.mark.gpu


def _random_problem(sizes, d, B, output, seed, id_dtype=torch.int64, quirk=False):
    from vae_amd import ops, _lib
    g = np.random.default_rng(seed)
    F = len(sizes)
    hi = np.cumsum(sizes).astype(np.int64)
    lo = hi - np.array(sizes)
    T = int(hi[-1])
    x = np.stack([g.integers(lo[f], hi[f], B) for f in range(F)], 1).astype(np.int64) if B else np.zeros((0, F), np.int64)
    nb_occ = g.integers(1, 60, T).astype(np.int64)
    y = g.standard_normal(B).astype(np.float32) * 2 + 3
    if output == "class":
        y = (y > 3).astype(np.float32)
    P = {"alpha": np.array([0.6], np.float32), "global_bias_mean": np.array([0.2], np.float32),
         "global_bias_scale": np.array([-0.9], np.float32),
         "bias_params": g.standard_normal((T, 2)).astype(np.float32),
         "entity_params": (g.standard_normal((T, 2 * d)) * 0.5).astype(np.float32)}
    P["entity_params"][:, d:] += np.sign(P["entity_params"][:, d:]) * 0.2      # keep |s| away from 0
    P["bias_params"][:, 1] += np.sign(P["bias_params"][:, 1]) * 0.2
    eps = (g.standard_normal(1).astype(np.float32), g.standard_normal(T).astype(np.float32),
           g.standard_normal((T, d)).astype(np.float32))
    group_hi = hi.copy()
    if quirk and F == 2:
        group_hi[0] += 1
    lik = _lib.LIK_NORMAL if output == "reg" else _lib.LIK_BERNOULLI
    spec = ops.Spec(T=T, F=F, d=d, group_hi=tuple(int(h) for h in group_hi), group_n=tuple(float(s) for s in sizes),
                    likelihood=lik, nb_train=5000)
    return spec, P, x, y, nb_occ, eps, group_hi


def _run_gpu(spec, P, x, y, nb_occ, eps, id_dtype=torch.int64, fused_check=True):
    from vae_amd import ops
    dev = torch.device("cuda:0")
    ent = torch.tensor(P["entity_params"], device=dev)
    bia = torch.tensor(P["bias_params"], device=dev)
    scal = torch.tensor(np.concatenate([P["alpha"], P["global_bias_mean"], P["global_bias_scale"]]), device=dev)
    inv_occ = ops.inv_occ_from_counts(torch.tensor(nb_occ, device=dev))
    plan = ops.BatchPlan(spec, torch.tensor(x, device=dev).to(id_dtype).contiguous(), torch.tensor(y, device=dev), inv_occ)
    e = (torch.tensor(eps[2], device=dev), torch.tensor(eps[1], device=dev), torch.tensor(eps[0], device=dev))
    st = ops.elbo_forward(plan, ent, bia, scal, inv_occ, eps=e)
    loss3 = ops.elbo_finalize(st, scal)
    g = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=dev))
    torch.cuda.synchronize()
    return plan, st, loss3, g


def _check(spec, P, x, y, nb_occ, eps, group_hi, output, id_dtype=torch.int64, tol=2e-5):
    plan, st, loss3, (g_ent, g_bias, g_sc) = _run_gpu(spec, P, x, y, nb_occ, eps, id_dtype)
    r = O.rowwise_elbo(P, x, y.astype(np.float64), nb_occ, group_hi, spec.group_n, spec.nb_train,
                       eps[0], eps[1], eps[2], output)
    assert abs(loss3[0].item() - r["loss"]) / abs(r["loss"]) < 1e-5
    assert rel_err(st.pred.cpu().numpy(), r["pred"]) < tol
    assert rel_err(st.sumz.cpu().numpy(), (P["entity_params"][x][..., :spec.d].astype(np.float64)
                                           + np.abs(P["entity_params"][x][..., spec.d:]) * eps[2][x]).sum(1)) < tol
    assert rel_err(g_ent.cpu().numpy(), r["g_entity_params"]) < tol
    assert rel_err(g_bias.cpu().numpy(), r["g_bias_params"]) < tol
    gs = g_sc.cpu().numpy()
    for i, k in enumerate(("g_alpha", "g_global_bias_mean", "g_global_bias_scale")):
        assert abs(gs[i] - r[k][0]) <= 1e-4 * max(abs(r[k][0]), 1e-2), k


# d -> (LPE, CPL, VEC): 4:(1,1,4) 8:(2,1,4) 12:(4,1,4) 20:(8,1,4) 64:(16,1,4) 100:(32,1,4) 128:(32,1,4)
# 256:(64,1,4) 512:(64,2,4) 1024:(64,4,4) 5,7:(8,1,1) 33:(64,1,1) 130:(64,4,1)
@pytest.mark.parametrize("d", [4, 8, 12, 20, 64, 100, 128, 256, 512, 1024, 5, 7, 33, 130])
@pytest.mark.parametrize("output", ["reg", "class"])
def test_all_kernel_shapes_F2(d, output):
    B = 300 if d <= 256 else 96
    args = _random_problem([37, 29], d, B, output, seed=d)
    _check(*args[:6], args[6], output)


@pytest.mark.parametrize("F,d", [(1, 16), (3, 8), (4, 20), (5, 128), (32, 256), (7, 5), (64, 12)])
def test_general_number_of_fields(F, d):
    sizes = [11 + (f % 5) for f in range(F)]
    args = _random_problem(sizes, d, 130, "class" if F % 2 else "reg", seed=100 + F)
    _check(*args[:6], args[6], "class" if F % 2 else "reg")


@pytest.mark.parametrize("id_dtype", [torch.int32, torch.int64])
def test_quirk_group_and_id_width(id_dtype):
    spec, P, x, y, nb_occ, eps, group_hi = _random_problem([40, 30], 16, 257, "reg", seed=9, quirk=True)
    x[:17, 1] = 40                                    # item id == N lands in the user group (:316)
    _check(spec, P, x, y, nb_occ, eps, group_hi, "reg", id_dtype)


@pytest.mark.parametrize("B", [1, 2, 7, 63, 64, 65, 4097])
def test_ragged_batch_sizes(B):
    args = _random_problem([50, 60], 128, B, "reg", seed=B)
    _check(*args[:6], args[6], "reg")


def test_empty_shard():
    """B == 0 (a rank with no rows): zero sums, dense zero gradients, finite loss."""
    spec, P, x, y, nb_occ, eps, group_hi = _random_problem([20, 20], 8, 0, "reg", seed=1)
    from vae_amd import ops
    dev = torch.device("cuda:0")
    ent = torch.tensor(P["entity_params"], device=dev)
    bia = torch.tensor(P["bias_params"], device=dev)
    scal = torch.tensor([0.6, 0.2, -0.9], device=dev)
    inv_occ = ops.inv_occ_from_counts(torch.tensor(nb_occ, device=dev))
    plan = ops.BatchPlan(spec, torch.zeros(0, 2, dtype=torch.int64, device=dev), torch.zeros(0, device=dev),
                         inv_occ, B_global=10)
    plan.W = torch.ones(2, dtype=torch.float64, device=dev)     # global normalisers come from the other ranks
    st = ops.elbo_forward(plan, ent, bia, scal, inv_occ)
    loss3 = ops.elbo_finalize(st, scal)
    g = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, torch.ones(1, device=dev))
    assert torch.isfinite(loss3).all()
    assert (g[0] == 0).all() and (g[1] == 0).all()
    kl0 = 0.5 * (0.81 + 0.04 - 1) - np.log(0.9)
    assert abs(loss3[0].item() - kl0) < 1e-6


def test_large_batch_properties():
    """Full-size batch (cfg3 shape, B=100K): size-independent properties instead of the oracle --
    (i) permuting the rows leaves loss and gradients unchanged (up to fp32 summation order),
    (ii) splitting the batch in two shards with global W / B_global and summing reproduces it,
    (iii) pred is linear in the global bias mean."""
    from vae_amd import ops, _lib
    dev = torch.device("cuda:0")
    sizes, d, B = [138493, 26744], 128, 100000
    T = sum(sizes)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.stack([torch.randint(0, sizes[0], (B,), generator=g, device=dev),
                     sizes[0] + torch.randint(0, sizes[1], (B,), generator=g, device=dev)], 1)
    y = torch.randint(1, 6, (B,), generator=g, device=dev).float()
    ent = torch.randn(T, 2 * d, generator=g, device=dev)
    bia = torch.randn(T, 2, generator=g, device=dev)
    scal = torch.tensor([0.7, 0.1, 0.9], device=dev)
    inv_occ = ops.inv_occ_from_counts(torch.randint(1, 300, (T,), generator=g, device=dev))
    spec = ops.Spec(T=T, F=2, d=d, group_hi=(sizes[0] + 1, T), group_n=(float(sizes[0]), float(sizes[1])),
                    likelihood=_lib.LIK_NORMAL, nb_train=16000000)
    one = torch.ones(1, device=dev)

    def run(xx, yy, W=None, B_global=None, flags=0):
        plan = ops.BatchPlan(spec, xx.contiguous(), yy.contiguous(), inv_occ, B_global=B_global)
        if W is not None:
            plan.W = W
        st = ops.elbo_forward(plan, ent, bia, scal, inv_occ, seed=3, step=11, flags=flags)
        l3 = ops.elbo_finalize(st, scal)
        gr = ops.elbo_backward(plan, st, ent, bia, scal, inv_occ, one)
        return plan, st, l3, gr

    plan, st, l3, gr = run(x, y)
    perm = torch.randperm(B, generator=g, device=dev)
    _, st_p, l3_p, gr_p = run(x[perm], y[perm])
    assert abs(l3_p[0].item() - l3[0].item()) / abs(l3[0].item()) < 1e-6
    assert torch.allclose(st_p.pred, st.pred[perm], rtol=1e-5, atol=1e-4)       # same rows (two loop-body copies: last-bit fma differences)
    assert rel_err(gr_p[0].cpu().numpy(), gr[0].cpu().numpy()) < 1e-5
    h = B // 3
    _, _, la, ga = run(x[:h], y[:h], W=plan.W, B_global=B)
    _, _, lb, gb = run(x[h:], y[h:], W=plan.W, B_global=B, flags=ops.FLAG_NO_PRIOR_TERMS)
    assert abs((la[0] + lb[0]).item() - l3[0].item()) / abs(l3[0].item()) < 1e-6
    assert rel_err((ga[0] + gb[0]).cpu().numpy(), gr[0].cpu().numpy()) < 1e-5
    assert rel_err((ga[2] + gb[2]).cpu().numpy(), gr[2].cpu().numpy()) < 1e-5
    scal2 = scal.clone(); scal2[1] += 2.5
    plan2 = ops.BatchPlan(spec, x, None, None)
    p1 = ops.elbo_forward(plan2, ent, bia, scal, None, seed=3, step=11, train=False).pred
    p2 = ops.elbo_forward(plan2, ent, bia, scal2, None, seed=3, step=11, train=False).pred
    assert torch.allclose(p2 - p1, torch.full_like(p1, 2.5), atol=1e-3)
    assert torch.allclose(p1, st.pred, rtol=1e-5, atol=1e-4)                    # predict mode == train mode pred


@pytest.mark.parametrize("d", [16, 128])
def test_skewed_batch_heavy_lists(d):
    """A few entities own most of the rows (lists far longer than VFM_HEAVY_LIST): the pre-reduced
    heavy path must give the oracle's gradients, through all three backward entry points."""
    from vae_amd import ops
    spec, P, x, y, nb_occ, eps, group_hi = _random_problem([300, 200], d, 6000, "reg", seed=77)
    g = np.random.default_rng(5)
    hot = g.random(6000) < 0.7
    x[hot, 1] = 300 + g.integers(0, 3, hot.sum())          # 3 items own 70 % of the rows
    x[g.random(6000) < 0.3, 0] = 7                         # one user owns 30 %
    plan, st, loss3, (g_ent, g_bias, g_sc) = _run_gpu(spec, P, x, y, nb_occ, eps)
    assert plan.heavy is not None and plan.heavy[0].numel() >= 4
    r = O.rowwise_elbo(P, x, y.astype(np.float64), nb_occ, group_hi, spec.group_n, spec.nb_train,
                       eps[0], eps[1], eps[2], "reg")
    assert abs(loss3[0].item() - r["loss"]) / abs(r["loss"]) < 1e-5
    assert rel_err(g_ent.cpu().numpy(), r["g_entity_params"]) < 5e-5
    assert rel_err(g_bias.cpu().numpy(), r["g_bias_params"]) < 5e-5
    # staged (multi-rank) form on one rank: statistics -> apply == fused backward + Adam
    dev = torch.device("cuda:0")
    ent = torch.tensor(P["entity_params"], device=dev); bia = torch.tensor(P["bias_params"], device=dev)
    scal = torch.tensor(np.concatenate([P["alpha"], P["global_bias_mean"], P["global_bias_scale"]]), device=dev)
    inv_occ = ops.inv_occ_from_counts(torch.tensor(nb_occ, device=dev))
    e = (torch.tensor(eps[2], device=dev), torch.tensor(eps[1], device=dev), torch.tensor(eps[0], device=dev))
    outs = []
    for staged in (False, True):
        pe, pb, ps = ent.clone(), bia.clone(), scal.clone()
        mv = [(torch.zeros_like(pe), torch.zeros_like(pb), torch.zeros(3, device=dev)) for _ in range(2)]
        st2 = ops.elbo_forward(plan, pe, pb, ps, inv_occ, eps=e)
        l3 = ops.elbo_finalize(st2, ps)
        if staged:
            rl = ops.exchange_record_len(d)
            acc = torch.zeros(spec.T * rl, device=dev); sums = torch.zeros(2, device=dev)
            mid = spec.T // 2
            ops.elbo_backward_acc(plan, st2, acc, sums, 0, mid)
            ops.elbo_backward_acc(plan, st2, acc, sums, mid, spec.T)
            ops.elbo_apply_adam(plan, st2, acc, sums, pe, pb, ps, inv_occ, mv[0], mv[1], 0.05, 1, e_lo=0, e_hi=mid)
            ops.elbo_apply_adam(plan, st2, acc, sums, pe, pb, ps, inv_occ, mv[0], mv[1], 0.05, 1, e_lo=mid, e_hi=spec.T)
        else:
            ops.elbo_backward_adam(plan, st2, pe, pb, ps, inv_occ, mv[0], mv[1], 0.05, 1)
        outs.append((pe, pb, ps))
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_randomised_configurations_against_oracle():
    """Seeded random sweep over (F, d, B, likelihood, id width, skew): forward, loss, all gradients and
    the fused backward+Adam step vs the fp64 oracle."""
    from vae_amd import ops
    rng = np.random.default_rng(2024)
    dev = torch.device("cuda:0")
    ds = [4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 160, 256, 3, 6, 10, 50]
    for trial in range(24):
        F = int(rng.choice([1, 2, 2, 2, 3, 4, 6, 9]))
        d = int(rng.choice(ds))
        B = int(rng.integers(1, 700))
        output = "reg" if rng.random() < 0.5 else "class"
        id_dtype = torch.int32 if rng.random() < 0.5 else torch.int64
        sizes = [int(rng.integers(2, 40)) for _ in range(F)]
        spec, P, x, y, nb_occ, eps, group_hi = _random_problem(sizes, d, B, output, seed=1000 + trial)
        if rng.random() < 0.4 and B > 64:                   # skew: one entity per column owns most rows
            for f in range(F):
                x[rng.random(B) < 0.6, f] = x[0, f]
        tag = (trial, F, d, B, output, str(id_dtype))
        plan, st, loss3, (g_ent, g_bias, g_sc) = _run_gpu(spec, P, x, y, nb_occ, eps, id_dtype)
        r = O.rowwise_elbo(P, x, y.astype(np.float64), nb_occ, group_hi, spec.group_n, spec.nb_train,
                           eps[0], eps[1], eps[2], output)
        assert abs(loss3[0].item() - r["loss"]) / abs(r["loss"]) < 2e-5, tag
        assert rel_err(st.pred.cpu().numpy(), r["pred"]) < 5e-5, tag
        # F == 1 is degenerate (no interactions: the true FM gradient is 0 and the kernel's
        # sum_r g_r*sumz_r - z_e*sum_r g_r cancels two equal fp32 numbers): looser bound there
        gtol = 1e-4 if F > 1 else 1e-3
        assert rel_err(g_ent.cpu().numpy(), r["g_entity_params"]) < gtol, tag
        assert rel_err(g_bias.cpu().numpy(), r["g_bias_params"]) < 1e-4, tag
        # fused backward + Adam == oracle gradients through the numpy Adam
        ent = torch.tensor(P["entity_params"], device=dev); bia = torch.tensor(P["bias_params"], device=dev)
        scal = torch.tensor(np.concatenate([P["alpha"], P["global_bias_mean"], P["global_bias_scale"]]), device=dev)
        inv_occ = ops.inv_occ_from_counts(torch.tensor(nb_occ, device=dev))
        e = (torch.tensor(eps[2], device=dev), torch.tensor(eps[1], device=dev), torch.tensor(eps[0], device=dev))
        mv = [(torch.zeros_like(ent), torch.zeros_like(bia), torch.zeros(3, device=dev)) for _ in range(2)]
        st2 = ops.elbo_forward(plan, ent, bia, scal, inv_occ, eps=e)
        l3 = torch.empty(3, device=dev)
        ops.elbo_backward_adam(plan, st2, ent, bia, scal, inv_occ, mv[0], mv[1], 0.01, 1, loss_out=l3)
        pe = P["entity_params"].astype(np.float64).copy()
        O.adam_step(pe, r["g_entity_params"], np.zeros_like(pe), np.zeros_like(pe), 1, 0.01)
        assert rel_err(ent.cpu().numpy(), pe) < (1e-5 if F > 1 else 1e-2), tag   # (Adam normalises: sign noise at F == 1)
        assert abs(l3[0].item() - r["loss"]) / abs(r["loss"]) < 2e-5, tag


def test_tables_beyond_4GiB_index_like_small_ones():
    """A 6.4 GB entity table (12.5 M rows x 128 floats; with gradients / eps ~20 GB of HBM): the same batch
    through the big table and through a compact table holding only the touched rows gives the same loss,
    predictions and gradient rows -- byte offsets beyond 2^32 are exercised on every path (forward, backward,
    fused Adam), untouched rows get exact zeros."""
    from vae_amd import ops, _lib
    dev = torch.device("cuda:0")
    N, M, d, B = 9_000_000, 3_500_000, 64, 40_000
    T = N + M
    g = torch.Generator(device="cpu").manual_seed(5)
    users = torch.randint(0, N, (B,), generator=g)
    users[: B // 2] = torch.randint(N - 200_000, N, (B // 2,), generator=g)      # many rows far beyond 4 GiB
    items = N + 1 + torch.randint(0, M - 1, (B,), generator=g)                   # (id == N, the quirk id, unused)
    x = torch.stack([users, items], 1).to(dev)
    y = torch.randint(1, 6, (B,), generator=g).float().to(dev)
    uniq = torch.unique(x)
    U = int(uniq.numel())
    nu = int((uniq < N).sum())
    xc = torch.searchsorted(uniq, x)
    ent = torch.randn(T, 2 * d, device=dev) * 0.3
    bia = torch.randn(T, 2, device=dev)
    scal = torch.tensor([0.8, 0.1, -0.9], device=dev)
    ee = torch.randn(T, d, device=dev)
    eb = torch.randn(T, device=dev)
    eg = torch.randn(1, device=dev)
    occ = torch.randint(1, 50, (T,), device=dev)
    lik = _lib.LIK_NORMAL
    big = ops.Spec(T=T, F=2, d=d, group_hi=(N + 1, T), group_n=(float(N), float(M)), likelihood=lik, nb_train=10 * B)
    small = ops.Spec(T=U, F=2, d=d, group_hi=(nu, U), group_n=(float(N), float(M)), likelihood=lik, nb_train=10 * B)
    res = []
    for spec, xs, sel in ((big, x, None), (small, xc, uniq)):
        pick = (lambda t: t) if sel is None else (lambda t: t[sel].contiguous())
        e_, b_ = pick(ent), pick(bia)
        inv = ops.inv_occ_from_counts(pick(occ))
        plan = ops.BatchPlan(spec, xs.contiguous(), y, inv)
        eps = (pick(ee), pick(eb), eg)
        st = ops.elbo_forward(plan, e_, b_, scal, inv, eps=eps)
        loss3 = ops.elbo_finalize(st, scal)
        ge, gb, gs = ops.elbo_backward(plan, st, e_, b_, scal, inv, torch.ones(1, device=dev))
        # fused Adam on copies of the touched rows' tables is checked through the parameters after one step
        e2, b2, s2 = e_.clone(), b_.clone(), scal.clone()
        mv = [(torch.zeros_like(e2), torch.zeros_like(b2), torch.zeros(3, device=dev)) for _ in range(2)]
        st2 = ops.elbo_forward(plan, e2, b2, s2, inv, eps=eps)
        ops.elbo_backward_adam(plan, st2, e2, b2, s2, inv, mv[0], mv[1], 0.01, 1, loss_out=torch.zeros(3, device=dev))
        res.append((loss3.clone(), st.pred.clone(), ge, gb, gs.clone(), e2, b2))
        del st, st2, mv
    (lb, pb, geb, gbb, gsb, e2b, b2b), (ls, ps, ges, gbs, gss, e2s, b2s) = res
    assert torch.allclose(lb, ls, rtol=1e-6)
    assert torch.allclose(pb, ps, rtol=1e-5, atol=1e-5)
    assert torch.allclose(geb[uniq], ges, rtol=1e-5, atol=1e-5 * float(ges.abs().max()))
    assert torch.allclose(gbb[uniq], gbs, rtol=1e-5, atol=1e-5 * float(gbs.abs().max()))
    assert torch.allclose(gsb, gss, rtol=1e-5)
    untouched = torch.ones(T, dtype=torch.bool, device=dev)
    untouched[uniq] = False
    assert not geb[untouched].any() and not gbb[untouched].any()
    # (Adam's first step is -lr * sign(g): entries whose gradient is rounding noise may differ -- compare the bulk)
    assert float(((e2b[uniq] - e2s).abs() > 1e-6).float().mean()) < 1e-4
    assert torch.equal(e2b[untouched], ent[untouched])
