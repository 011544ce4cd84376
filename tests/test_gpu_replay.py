"""GPU: the REPLAYABLE training step (step-dependent constants in device memory, launches captured once per (batch, next
batch, step form) in a HIP graph and replayed -- include/vfm_hip.h: vfm_dev_step_t) and the packed first-order records
(vfm_problem_t.wrec) against the eager step with host-side constants: the trajectory must be the eager one BIT FOR BIT --
parameters, Adam moments, losses -- across moment-period boundaries, learning-rate changes, predictions in between (they
consume Philox steps), an un-announced batch (eager catch-up in the middle of replays) and a checkpoint round trip.
Reference loop: vfm-torch.py:351-370."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(sizes, d, B, nb, output="reg", **attrs):
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    torch.manual_seed(3)
    m = VFM(field_sizes=list(sizes), embedding_size=d, device="cuda", rng_seed=11, output=output)
    for k, v in attrs.items():
        assert hasattr(m, k), k
        setattr(m, k, v)
    X, y = synthetic_triples(list(sizes), nb * B, seed=4, device="cuda", output=output)
    m.set_training_data(X, nb_train=nb * B)
    plans = [m.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(nb)]
    return m, plans, X


FORMS = {
    # name: (field sizes, d, B, batches, attributes, the step form the replayed run must be in)
    "lookahead_F3": ((900, 700, 400), 16, 48, 6, dict(pipeline=False), "la"),
    "lookahead_F2_d128": ((1500, 500), 128, 200, 5, dict(pipeline=False), "la"),
    "lookahead_scan": ((900, 700, 400), 16, 48, 6, dict(pipeline=False, lookahead_list=False), "la"),
    "dense_small_table": ((60, 40), 20, 400, 4, dict(pipeline=False, lookahead=False), "dense"),
    "dense_heavy_lists": ((3000, 40), 32, 600, 4, dict(pipeline=False, lookahead=False), "dense"),
    "pipelined": ((300, 200), 32, 2000, 4, dict(pipeline=True), "pipe"),
    "criteo_like_class": ((500,) * 12, 32, 64, 5, dict(pipeline=False), "la"),
}


@pytest.mark.parametrize("name", sorted(FORMS))
def test_replayed_steps_are_bitwise_the_eager_trajectory(name, monkeypatch):
    import vae_amd.model as M
    monkeypatch.setattr(M, "_CHECK_WREC", True)           # every step also checks the packed records against the tables
    sizes, d, B, nb, attrs, form = FORMS[name]
    out = "class" if "class" in name else "reg"
    eager, plans_e, X = _setup(sizes, d, B, nb, out, replay=False, **attrs)
    rep, plans_r, _ = _setup(sizes, d, B, nb, out, replay=True, **attrs)
    n_steps = 300                                         # two moment-period boundaries (128, 256)
    order = [s % nb for s in range(n_steps + 1)]
    for s in range(n_steps):
        lr = 0.05 if (s // 60) % 2 == 0 else 0.02         # the learning rate changes now and then (table refill)
        i, j = order[s], order[s + 1]
        if s == 99:
            j = (j + 2) % nb                              # step 100 does not take the batch step 99 announced
        le, pe = eager.train_step(plans_e[i], lr=lr, next_plan=plans_e[j])
        lr_, pr = rep.train_step(plans_r[i], lr=lr, next_plan=plans_r[j])
        if s % 25 == 17 or s in (127, 128, 129, 255, 256):
            assert torch.equal(le, lr_), (name, s)
            assert torch.equal(pe, pr), (name, s)
        if s in (140, 141, 200):                          # predictions consume Philox steps: the device counter is re-set
            assert torch.equal(eager.predict(X[:100])["y_pred"], rep.predict(X[:100])["y_pred"])
    captured = [k for k, v in rep._graphs.items() if v is not None]
    assert captured and all(k[0] == form for k in captured), (form, [k[0] for k in rep._graphs])
    assert not rep._step_state.error()
    assert eager._step_state is None and not eager._graphs
    eager.sync_lazy(); rep.sync_lazy()
    assert torch.equal(eager._flat, rep._flat)
    assert torch.equal(eager._adam_m, rep._adam_m) and torch.equal(eager._adam_v, rep._adam_v)
    assert eager._adam_t == rep._adam_t == n_steps and eager.global_step == rep.global_step


def test_packed_first_order_records_do_not_change_the_trajectory(monkeypatch):
    """use_wrec on / off: the forward reads the same (mu_w, s_w, 1/occ) either way -- bitwise the same run; the records
    survive look-ahead steps, catch-up passes and predictions, and are dropped by an unfused step, a load_state_dict and
    a direct write announced with params_changed()."""
    import vae_amd.model as M
    monkeypatch.setattr(M, "_CHECK_WREC", True)
    a, pa, X = _setup((1500, 500), 64, 200, 5, pipeline=False, use_wrec=True)
    b, pb, _ = _setup((1500, 500), 64, 200, 5, pipeline=False, use_wrec=False)
    for s in range(140):
        fused = s not in (40, 41)
        la_, _ = a.train_step(pa[s % 5], lr=0.04, next_plan=pa[(s + 1) % 5], fused=fused)
        lb_, _ = b.train_step(pb[s % 5], lr=0.04, next_plan=pb[(s + 1) % 5], fused=fused)
        assert a._wrec_ok == fused and b._wrec is None
        if s % 20 == 3:
            assert torch.equal(la_, lb_), s
        if s == 77:
            assert torch.equal(a.predict(X[:64])["y_pred"], b.predict(X[:64])["y_pred"])
    sd = a.state_dict()
    a.sync_lazy(); b.sync_lazy()
    assert torch.equal(a._flat, b._flat)
    assert a._wrec_ok
    a.load_state_dict(sd)
    assert not a._wrec_ok and not a._lazy_dirty
    a.train_step(pa[0], lr=0.04, next_plan=pa[1])
    assert a._wrec_ok
    a.bias_params.weight.data[3, 0] = 0.25
    a.params_changed()
    assert not a._wrec_ok
    a.train_step(pa[1], lr=0.04, next_plan=pa[2])           # (rebuilt from the tables: the debug check holds again)
    assert a._wrec_ok


def test_fit_replays_and_matches_the_eager_fit():
    """VFM.fit with replay on (one graph per (batch, next batch) pair, replayed every epoch) == the same fit with
    replay off (the default): history, parameters, the epoch-averaged snapshots, the predictors."""
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    sizes = (943, 1682)
    X, y = synthetic_triples(list(sizes), 6000, seed=9, device="cuda")
    Xt, yt = synthetic_triples(list(sizes), 500, seed=10, device="cuda")
    runs = []
    for replay in (False, True):
        torch.manual_seed(5)
        m = VFM(sizes[0], sizes[1], 20, device="cuda", rng_seed=3)
        m.replay = replay
        h = m.fit(X, y, n_epochs=6, batch_size=1500, X_test=Xt, y_test=yt, verbose=False)
        runs.append((m, h))
    (m0, h0), (m1, h1) = runs
    assert [k for k, v in m1._graphs.items() if v is not None] and not m0._graphs
    assert h0["elbo"] == h1["elbo"] and h0["train_rmse"] == h1["train_rmse"] and h0["test"] == h1["test"]
    assert torch.equal(m0._flat, m1._flat) and torch.equal(m0._mean_flat, m1._mean_flat)
    assert m1.replay is True and m0.replay is False


def test_replay_survives_a_checkpoint_round_trip():
    rep, plans, X = _setup((900, 700, 400), 16, 48, 6, pipeline=False, replay=True)
    ref, plans_f, _ = _setup((900, 700, 400), 16, 48, 6, pipeline=False, replay=False)
    for s in range(30):
        rep.train_step(plans[s % 6], lr=0.03, next_plan=plans[(s + 1) % 6])
        ref.train_step(plans_f[s % 6], lr=0.03, next_plan=plans_f[(s + 1) % 6])
    sd = rep.training_state_dict()
    for s in range(30, 50):
        rep.train_step(plans[s % 6], lr=0.03, next_plan=plans[(s + 1) % 6])
    rep.load_training_state_dict(sd)                       # back to step 30: the device counters must follow
    for s in range(30, 60):
        rep.train_step(plans[s % 6], lr=0.03, next_plan=plans[(s + 1) % 6])
        ref.train_step(plans_f[s % 6], lr=0.03, next_plan=plans_f[(s + 1) % 6])
    rep.sync_lazy(); ref.sync_lazy()
    assert torch.equal(rep._flat, ref._flat) and torch.equal(rep._adam_m, ref._adam_m)
