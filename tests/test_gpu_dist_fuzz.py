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
