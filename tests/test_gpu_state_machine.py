"""GPU: fixed-seed STATE-MACHINE FUZZ of `VFM.train_step`.  The step has eight forms (plain fused, look-ahead with a row
list or a scan, row-list lazy, pipelined in its every-row and its look-ahead form, unfused, with or without the packed
first-order records) steered by hidden state (`_lazy_dirty`, `_lazy_kind`, `_la_ready_for`, `_zrec_for`,
`_moments_scaled`, `_wrec_ok`); every transition has a hand-written test somewhere, this file walks RANDOM sequences of
them.

Reference: a second model that only ever takes the plain dense fused step (and an unfused step where the sequence has
one).  After EVERY step the loss triple and the predictions must equal the reference's bit for bit; at random points, and
at the end, so must every parameter and both Adam moments.  In between: predictions, save_weights, state_dict reads,
checkpoint round trips of the model into itself, learning-rate changes, un-announced batches.  The software-pipelined
step sums in another order (the backward adds the other entity's sample instead of sumz - z), so sequences that contain
it are compared to 1e-4 of the largest entry of every table instead -- the three scalars included: round 3 had to allow
alpha's first moment 1e-2 there, because its gradient was summed as per-row differences (y - pred)^2 / 2 - 1 / (2|alpha|)
in fp32; since ABI 5 the lanes add the positive halves and the constant comes off once in fp64 (VFM_P_ALPHA), and the
third test puts the fp64 oracle's value beside both forward forms.  In those walks a TWIN model takes the same sequence
with every lazy form replaced by its every-row equivalent (look-ahead / row list -> dense, pipelined look-ahead ->
pipelined): the two must agree BIT FOR BIT after every step.  Reference loop: vfm-torch.py:351-370."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_TRACE = bool(__import__("os").environ.get("VFM_SM_TRACE"))
CONFIGS = {"F2_d32": ((300, 200), 32, 64, "reg"), "F3_d16": ((200, 150, 100), 16, 48, "class"),
           "F2_d128_heavy": ((400, 12), 128, 96, "reg")}


def _make(sizes, d, B, nb, output):
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    torch.manual_seed(3)
    m = VFM(field_sizes=list(sizes), embedding_size=d, device="cuda", rng_seed=11, output=output)
    m.lazy_min_params = 0
    m.pipeline_min_T, m.pipeline_min_d = 0, 0
    X, y = synthetic_triples(list(sizes), nb * B, seed=4, device="cuda", output=output)
    m.set_training_data(X, nb_train=nb * B)
    plans = [m.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(nb)]
    return m, plans, X


def _same_state(a, b, exact, where):
    a.sync_lazy(); b.sync_lazy()
    for name in ("_flat", "_adam_m", "_adam_v"):
        ta, tb = getattr(a, name), getattr(b, name)
        if name != "_flat" and a._moments_scaled != b._moments_scaled:
            continue                               # (the two hold the moments in different forms right now)
        if exact:
            assert torch.equal(ta, tb), (where, name)
            continue
        # per table (entity rows, first-order rows, the three scalars): 1e-4 of its largest entry.  The scalars' first moments
        # against 1e-4 of the scale of what they are a DIFFERENCE of: dloss/dalpha = nb_train/B sum_r (y-pred)^2/2 - nb_train/(2|alpha|)
        # -- the two models' predictions differ by the trajectories' 1e-5 after a hundred steps, the halves by as much, their
        # difference by that much of a HALF (each form by itself is within 2e-6 of the halves of the fp64 truth: last test)
        for part, (pa, pb) in enumerate(zip(a._views(ta), b._views(tb))):
            scale = float(pb.abs().max())
            if part == 2 and name == "_adam_m":         # (scaled form: the buffers hold m / beta1^k, k = steps into the moment period)
                scale = max(scale, a.nb_train / (2.0 * abs(float(b.alpha))) * (0.9 ** -(a._adam_t % 128) if a._moments_scaled else 1.0))
            assert float((pa - pb).abs().max()) <= 1e-4 * scale + 1e-30, (where, name, part)
    assert a._adam_t == b._adam_t and a.global_step == b.global_step, where


def _walk(cfg, seed, with_pipeline, n_steps=140, monkeypatch=None, with_twin=False):
    import vae_amd.model as M
    if monkeypatch is not None:
        monkeypatch.setattr(M, "_CHECK_WREC", True)
    sizes, d, B, output = CONFIGS[cfg]
    nb = 5
    ref, plans_r, X = _make(sizes, d, B, nb, output)
    tst, plans_t, _ = _make(sizes, d, B, nb, output)
    ref.lookahead = ref.pipeline = ref.use_wrec = False
    ref.lazy_adam = False
    twin = plans_w = None
    if with_twin:           # the same sequence with every lazy form replaced by its every-row equivalent: bitwise `tst`
        twin, plans_w, _ = _make(sizes, d, B, nb, output)
        twin.lookahead = twin.use_wrec = False
        twin.lazy_adam = False
    g = np.random.default_rng(seed)
    lr = 0.05
    cur = 0
    announced = None
    trace = []
    for s in range(n_steps):
        # ---- something between two steps
        act = g.choice(["none", "none", "none", "predict", "save", "state_dict", "checkpoint", "lr", "reload_own"])
        if act == "predict":
            pa, pb = ref.predict(X[:70]), tst.predict(X[:70])
            assert (torch.equal(pa["y_pred"], pb["y_pred"]) if not with_pipeline else
                    float((pa["y_pred"] - pb["y_pred"]).abs().max()) <= 1e-4 * float(pa["y_pred"].abs().max())), (s, trace[-5:])
            if twin is not None:
                assert torch.equal(twin.predict(X[:70])["y_pred"], pb["y_pred"]), (s, trace[-5:])
        elif act == "save":
            ref.save_weights(); tst.save_weights()
            if twin is not None:
                twin.save_weights()
        elif act == "state_dict":
            tst.state_dict()
        elif act == "checkpoint":
            for mdl in (tst, twin):         # (the twin mirrors what happens between steps: it differs in the step FORMS only)
                if mdl is not None:
                    mdl.load_training_state_dict(mdl.training_state_dict())
        elif act == "reload_own":
            for mdl in (tst, twin):
                if mdl is not None:
                    mdl.load_state_dict(mdl.state_dict())
        elif act == "lr":
            lr = float(g.choice([0.05, 0.02, 0.08]))
        # ---- the step form of the model under test
        # (twin walks: no row-list lazy form -- rows lagging in THAT form keep the next step from pipelining, so the twin,
        #  whose rows never lag, would take another step form: a rule of the step, not a difference to test)
        form = str(g.choice(["dense", "la_list", "la_scan"] + ([] if with_twin else ["lazy_list", "lazy_auto"]) + ["unfused"] +
                            (["pipe", "pipe_la", "pipe_la"] if with_pipeline else [])))
        tst.lookahead = form in ("la_list", "la_scan", "pipe_la")
        tst.lookahead_list = form != "la_scan"
        tst.lazy_adam = {"lazy_list": True, "lazy_auto": "auto"}.get(form, False)
        tst.lazy_threshold = 0.9 if form == "lazy_auto" else 0.35
        tst.pipeline = True if form in ("pipe", "pipe_la") else False
        tst.use_wrec = bool(g.random() < 0.6)
        if twin is not None:
            twin.pipeline = tst.pipeline
        nxt = (cur + 1) % nb
        if g.random() < 0.12:
            nxt = int(g.integers(0, nb))                     # the next batch is not the one in line ...
        name_next = g.random() < 0.85                        # ... and sometimes nobody names one
        fused = form != "unfused"
        trace.append((s, str(act), form, tst.use_wrec, cur, nxt if name_next else None))
        if _TRACE:
            print(trace[-1], "adam_t", tst._adam_t, "dirty", tst._lazy_dirty, tst._lazy_kind, flush=True)
        lt, pt = tst.train_step(plans_t[cur], lr=lr, next_plan=plans_t[nxt] if name_next else None, fused=fused)
        lr_, pr = ref.train_step(plans_r[cur], lr=lr, fused=fused)
        if twin is not None:
            lw, pw = twin.train_step(plans_w[cur], lr=lr, next_plan=plans_w[nxt] if name_next else None, fused=fused)
            assert torch.equal(lt, lw) and torch.equal(pt, pw), ("twin", trace[-6:], lt, lw)
        if with_pipeline:
            assert torch.allclose(lt, lr_, rtol=1e-4), (trace[-6:], lt, lr_)
            assert float((pt - pr).abs().max()) <= 1e-4 * float(pr.abs().max()), trace[-6:]
        else:
            assert torch.equal(lt, lr_), (trace[-6:], lt, lr_)
            assert torch.equal(pt, pr), trace[-6:]
        if g.random() < 0.12:
            _same_state(tst, ref, not with_pipeline, trace[-6:])
            if twin is not None:
                _same_state(tst, twin, True, ("twin", trace[-6:]))
        cur = nxt if g.random() < 0.9 else int(g.integers(0, nb))      # an un-announced batch now and then
    _same_state(tst, ref, not with_pipeline, "end")
    if twin is not None:
        _same_state(tst, twin, True, "end (twin)")
    for pl in plans_t:
        pl.check_status()            # no kernel had to clamp an index entry
    return trace


@pytest.mark.parametrize("cfg", sorted(CONFIGS))
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_step_form_sequences_are_bitwise_the_dense_trajectory(cfg, seed, monkeypatch):
    trace = _walk(cfg, seed, with_pipeline=False, monkeypatch=monkeypatch)
    forms = {t[2] for t in trace}
    assert {"dense", "la_list", "lazy_list", "unfused"} <= forms         # the walk did visit the forms


@pytest.mark.parametrize("seed", [4, 5, 100, 102, 103, 111, 119])     # (100 .. 119: the seeds whose alpha first moment left
def test_random_sequences_with_the_pipelined_step(seed, monkeypatch):  #  1e-4 in round 3, when alpha's gradient was summed in fp32)
    trace = _walk("F2_d32", seed, with_pipeline=True, monkeypatch=monkeypatch)
    assert {"pipe", "pipe_la"} <= {t[2] for t in trace}


@pytest.mark.parametrize("seed", [6, 7, 8, 104])
def test_lazy_forms_are_bitwise_their_every_row_twins_in_walks_with_the_pipelined_step(seed, monkeypatch):
    """the pipelined look-ahead form against the every-row pipelined form INSIDE the random walk -- a twin model
    takes the same sequence (steps, predictions, checkpoints, un-announced batches) with look-ahead off, i.e. every
    look-ahead step as a dense one and every pipelined look-ahead step as an every-row pipelined one: loss triple and
    predictions equal BIT FOR BIT after every step, parameters and moments at random points and at the end."""
    trace = _walk("F2_d32", seed, with_pipeline=True, monkeypatch=monkeypatch, with_twin=True)
    assert {"pipe", "pipe_la", "la_list"} <= {t[2] for t in trace}


def test_alpha_gradient_of_both_forward_forms_against_the_fp64_oracle():
    """dloss/dalpha = sign(alpha) nb_train / B * sum_r [(y - pred)^2 / 2 - 1 / (2|alpha|)] is a CANCELLING sum (zero at the
    optimum of alpha), the least accurate number the step emits when the rows' differences are added in fp32.  Here alpha
    is set where the two halves nearly cancel (the sum is < 1e-3 of either half), and the gradient the library forms --
    through the sampling forward (k_fwd2) and through the record-gather forward of the pipelined step (k_fwd2<ZREC>) --
    is compared with the fp64 oracle on the same draws: both within 2e-3 of the TRUE (cancelled) value, i.e. ~2e-6 of the
    halves.  (Formed per row in fp32 the error was ~1e-4 of the halves: 10 % of the value at this alpha.)"""
    import numpy as np
    from oracle import vfm_oracle as O
    from vae_amd import ops, _lib
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    dev = torch.device("cuda:0")
    sizes, d, B = (300, 200), 32, 4096
    torch.manual_seed(3)
    m = VFM(field_sizes=list(sizes), embedding_size=d, device=dev, rng_seed=11)
    X, y = synthetic_triples(list(sizes), B, seed=4, device=dev)
    m.set_training_data(X, nb_train=7 * B)
    plan = m.plan(X, y)
    ent, bia, scal = m._views(m._flat)
    spec = m.spec()
    ee, eb, eg = (t.cpu().numpy() for t in ops.philox_eps(spec, seed=11, step=5, device=dev))

    def oracle():
        ent_, bia_, scal_ = (t.detach().cpu().numpy() for t in m._views(m._flat))
        P = {"alpha": scal_[0:1], "global_bias_mean": scal_[1:2], "global_bias_scale": scal_[2:3], "bias_params": bia_,
             "entity_params": ent_}
        return O.rowwise_elbo(P, X.cpu().numpy(), y.cpu().numpy().astype(np.float64), m.nb_occ.cpu().numpy(), np.array(m.group_hi),
                              np.array(m.group_n), m.nb_train, eg, eb, ee, "reg")
    r0 = oracle()
    half = float(np.sum(0.5 * (y.cpu().numpy().astype(np.float64) - r0["pred"]) ** 2))      # sum_r (y - pred)^2 / 2
    with torch.no_grad():
        m.alpha.fill_(float(np.float32(B / (2.0 * half) * (1 + 3e-4))))        # the halves now cancel to ~3e-4 of themselves
    want = float(oracle()["g_alpha"][0])
    assert abs(want) < 2e-3 * (m.nb_train / B) * half                           # ... they really do

    def g_alpha(st):
        loss3 = ops.elbo_finalize(st, scal)
        return float(torch.sign(m.alpha)[0]) * (m.nb_train / B) * float(st.partials[_lib.P_ALPHA])
    st = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, seed=11, step=5)
    got_plain = g_alpha(st)
    g_sc = ops.elbo_backward(plan, st, ent, bia, scal, m.inv_occ, torch.ones(1, device=dev))[2]
    zrec = torch.zeros(m.T, ops.record_len(d), device=dev)
    ops.sample_records(plan, ent, bia, m.inv_occ, zrec, 11, 5)
    st2 = ops.elbo_forward_records(plan, zrec, scal, 11, 5, torch.empty(B, device=dev), torch.empty(B, device=dev),
                                   torch.empty(_lib.PARTIALS_LEN, dtype=torch.float64, device=dev))
    got_pipe = g_alpha(st2)
    for got in (got_plain, got_pipe, float(g_sc[0])):
        assert abs(got - want) <= 2e-3 * abs(want) + 2e-6 * (m.nb_train / B) * half, (got, want, half)
