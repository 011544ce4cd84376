"""GPU: fixed-seed FUZZ of the multi-rank statistics step (`VFM.exchange = "stats"`, vae_amd/dist.py::step_stats) over
in-process thread ranks (tests/thread_ranks.py: the real per-rank code path, the collectives replaced by an in-process
sum in rank order).

The step has three forms -- the dense table of records, the exchange compacted to the globally touched entities, and
that with the other rows' Adam updates deferred (catch-up pass + apply stage over the list) -- chosen per step by
`exchange_compact`, `exchange_compact_below`, `exchange_lazy`, the position in the moment period and what ran before.
All three are the SAME dense Adam trajectory: random configurations (2-8 ranks, 2-3 fields, shards with no row at all,
skewed ids, batches touching 2 %-90 % of the table), random per-step learning rates and thresholds, predictions and
state_dict reads in between, across a moment-period boundary -- parameters, both moments and every recorded loss of the
compact/lazy run must equal the every-row run BIT FOR BIT, and all replicas must agree.  Reference loop:
vfm-torch.py:351-370 (one rank's body); north star: row-sharded batch + one all-reduce per step."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(cfg, lazy, compact, monkeypatch):
    from thread_ranks import run_ranks
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    from vae_amd.dist import shard_rows
    sizes, d, world, B, nb, n_steps, zipf, actions = cfg
    X, y = synthetic_triples(sizes, nb * B, seed=4, device="cuda", zipf=zipf)
    torch.manual_seed(3)
    first = VFM(field_sizes=sizes, embedding_size=d, device="cuda", rng_seed=11)
    first.exchange, first.exchange_lazy, first.exchange_compact = "stats", lazy, compact
    models = [first] + [copy.deepcopy(first) for _ in range(world - 1)]
    for m in models[1:]:
        m._tie(); m.__dict__.pop("_view_cache", None)

    def rank_body(rank, group):
        m = models[rank]
        m.set_training_data(X, nb_train=nb * B)
        plans = []
        for i in range(nb):
            a, b = shard_rows(i * B, (i + 1) * B, rank, world)
            plans.append(m.plan(X[a:b], y[a:b], B_global=B, process_group=group))
        rec, forms = [], set()
        for s in range(n_steps):
            act, lr, below = actions[s]
            if lazy or compact:
                m.exchange_compact_below = below
            if act == "predict":
                rec.append(m.predict(X[:40])["y_pred"].sum().reshape(1).repeat(3))
            elif act == "state_dict":
                rec.append(m.state_dict()["entity_params.weight"].sum().reshape(1).repeat(3).to(torch.float32))
            l3, _ = m.train_step(plans[s % nb], lr=lr, process_group=group)
            forms.add((bool(m._lazy_dirty), int(m._exchanged_floats)))
            if s % 9 == 2 or s in (127, 128):
                rec.append(l3.clone())
        m.sync_lazy()
        return m._flat, m._adam_m, m._adam_v, torch.stack(rec), forms

    out, _ = run_ranks(world, rank_body, monkeypatch)
    for r in range(1, world):
        assert all(torch.equal(out[r][i], out[0][i]) for i in range(4)), ("replicas differ", r)
    return out[0]


def _config(seed):
    g = np.random.default_rng(seed)
    F = int(g.choice([2, 2, 3]))
    d = int(g.choice([16, 32, 128]))
    world = int(g.choice([2, 3, 5, 8]))
    sizes = [int(g.integers(30, 1500 if d < 128 else 400)) for _ in range(F)]
    B = int(g.choice([world - 1, 7, 60, 300, 900]))        # (world - 1 rows: one rank's shard is empty)
    nb = int(g.integers(3, 6))
    n_steps = 150
    zipf = float(g.choice([0.0, 0.0, 1.1])) or None
    actions = [(str(g.choice(["none"] * 8 + ["predict", "state_dict"])), float(g.choice([0.05, 0.02, 0.08])),
                float(g.choice([0.85, 0.85, 0.0, 2.0]))) for _ in range(n_steps)]
    return sizes, d, world, max(B, 1), nb, n_steps, zipf, actions


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 9])
def test_random_multi_rank_statistics_runs_are_bitwise_the_every_row_run(seed, monkeypatch):
    import vae_amd.model as M
    monkeypatch.setattr(M, "_CHECK_WREC", True)
    cfg = _config(seed)
    dense = _run(cfg, lazy=False, compact=False, monkeypatch=monkeypatch)
    assert len(dense[4]) == 1 and not next(iter(dense[4]))[0]            # the every-row run: one form, nothing lags
    full = next(iter(dense[4]))[1]
    for lazy, compact in ((True, True), (False, True)):
        got = _run(cfg, lazy=lazy, compact=compact, monkeypatch=monkeypatch)
        for i, name in enumerate(("parameters", "first moments", "second moments", "losses / predictions")):
            assert torch.equal(got[i], dense[i]), (seed, cfg[:7], lazy, compact, name)
        assert any(n < full for _, n in got[4]) and any(n == full for _, n in got[4])      # compact AND dense exchanges ran
        assert any(lag for lag, _ in got[4]) == lazy                                       # rows lagged iff the lazy form is on


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_switching_between_the_exchange_forms_mid_training(seed, monkeypatch):
    """The three north-star exchange forms ("stats" with its compact / lazy state, "grads", "rows") taken in a random
    order step by step by the same replicas -- every switch has to leave the tables, the moments, the lagging-row
    bookkeeping, the packed first-order records and the sample records in a state the next form can start from.  The
    forms sum in different orders, so the comparison with ONE rank's plain step is to 1e-4 per table (2e-4 on the
    losses); the replicas themselves must stay BIT-identical."""
    from thread_ranks import run_ranks
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    from vae_amd.dist import shard_rows
    import vae_amd.model as M
    monkeypatch.setattr(M, "_CHECK_WREC", True)
    g = np.random.default_rng(seed)
    world = int(g.choice([2, 3, 5]))
    sizes, d, B, nb, n_steps = [int(g.integers(100, 900)), int(g.integers(60, 500))], 16, int(g.choice([48, 200])), 4, 70
    modes = [str(g.choice(["stats", "grads", "rows"])) for _ in range(n_steps)]
    announce = [bool(g.random() < 0.6) for _ in range(n_steps)]       # (a step told which batch follows: the rows form then lets rows wait)
    lrs = [float(g.choice([0.02, 0.01])) for _ in range(n_steps)]
    X, y = synthetic_triples(sizes, nb * B, seed=4, device="cuda")
    torch.manual_seed(3)
    first = VFM(field_sizes=sizes, embedding_size=d, device="cuda", rng_seed=11)
    single = copy.deepcopy(first); single._tie(); single.__dict__.pop("_view_cache", None)
    models = [first] + [copy.deepcopy(first) for _ in range(world - 1)]
    for m in models[1:]:
        m._tie(); m.__dict__.pop("_view_cache", None)

    def rank_body(rank, group):
        m = models[rank]
        m.set_training_data(X, nb_train=nb * B)
        plans = []
        for i in range(nb):
            a, b = shard_rows(i * B, (i + 1) * B, rank, world)
            plans.append(m.plan(X[a:b], y[a:b], B_global=B, process_group=group))
        losses = []
        for s in range(n_steps):
            m.exchange = modes[s]
            losses.append(m.train_step(plans[s % nb], lr=lrs[s], process_group=group,
                                       next_plan=plans[(s + 1) % nb] if announce[s] else None)[0].clone())
            if s % 23 == 7:
                losses.append(m.predict(X[:30])["y_pred"].sum().reshape(1).repeat(3))
        m.sync_lazy()
        return m._flat, m._adam_m, m._adam_v, torch.stack(losses)

    out, _ = run_ranks(world, rank_body, monkeypatch)
    for r in range(1, world):
        assert all(torch.equal(out[r][i], out[0][i]) for i in range(4)), ("replicas differ", r)
    single.set_training_data(X, nb_train=nb * B)
    plans = [single.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(nb)]
    want = []
    for s in range(n_steps):
        want.append(single.train_step(plans[s % nb], lr=lrs[s], fused=False)[0].clone())
        if s % 23 == 7:
            want.append(single.predict(X[:30])["y_pred"].sum().reshape(1).repeat(3))
    want = torch.stack(want)
    errs = {"losses": float(((out[0][3] - want).abs() / want.abs()).max())}
    for name, got in zip(("_flat", "_adam_m", "_adam_v"), out[0][:3]):
        if name != "_flat" and models[0]._moments_scaled != single._moments_scaled:
            continue
        for part, (pa, pb) in enumerate(zip(models[0]._views(got), single._views(getattr(single, name)))):
            errs[(name, part)] = float((pa - pb).abs().max()) / (float(pb.abs().max()) + 1e-30)
    for key, e in errs.items():
        tol = 2e-4 if key == "losses" else (1e-2 if key == ("_adam_m", 2) else 1e-4)
        assert e <= tol, (modes[:12], errs)
