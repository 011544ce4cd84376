"""GPU, 2 processes sharing cuda:0 (gloo transport): the real HIP step, row-sharded over two ranks
with W all-reduce + one flat all-reduce + NO_PRIOR_TERMS on rank 1, equals the 1-rank step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import Case, rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model(c, dev):
    from vae_amd.model import VFM
    torch.manual_seed(42)
    m = VFM(c.N, c.M, c.d, output=c.output, device=dev, rng_seed=77, n_samples=c.n_samples, link=c.link)
    m.set_training_data(torch.tensor(c.x), nb_train=c.nb_train, nb_occ=torch.tensor(c.nb_occ))
    return m


def _worker(rank, world, port, name, out_dir, exchange="stats"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.dist import shard_rows
    dev = torch.device("cuda:0")
    c = Case(name)
    m = _make_model(c, dev)
    m.exchange = exchange
    B = len(c.y)
    losses = []
    for step in range(3):
        a, b = shard_rows(0, B, rank, world)
        plan = m.plan(torch.tensor(c.x[a:b]), torch.tensor(c.y[a:b]), B_global=B, process_group=dist.group.WORLD)
        loss3, _ = m.train_step(plan, lr=0.05, process_group=dist.group.WORLD)
        losses.append(loss3.cpu().numpy().copy())
    m.sync_lazy()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"params_{rank}.npy"), m._flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"loss_{rank}.npy"), np.array(losses))
    dist.destroy_process_group()


@pytest.mark.parametrize("name,exchange", [("ml100k_reg_d20", "stats"), ("ml100k_reg_d20", "grads"),
                                           ("ml20m_reg_d128", "stats"), ("ml20m_reg_d128", "grads"),   # cfg4's shape (d = 128, ML-20M ids)
                                           ("softplus_reg_d8", "stats"),
                                           ("multi_reg_d8_s3", "auto"),             # S > 1: falls back to "grads"
                                           ("softplus_multi_class_d8_s2", "auto")])
def test_two_ranks_match_one_rank(name, exchange, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path), exchange), nprocs=world, join=True)
    c = Case(name)
    dev = torch.device("cuda:0")
    m = _make_model(c, dev)
    plan = m.plan(torch.tensor(c.x), torch.tensor(c.y))
    ref_losses = []
    for step in range(3):
        loss3, _ = m.train_step(plan, lr=0.05, fused=False)
        ref_losses.append(loss3.cpu().numpy().copy())
    want = m._flat.cpu().numpy()
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    assert np.array_equal(p0, p1)                         # replicas stay bit-identical
    assert rel_err(p0, want) < 1e-5
    l0 = np.load(tmp_path / "loss_0.npy")
    assert rel_err(l0, np.array(ref_losses)) < 1e-5


def _fit_worker(rank, world, port, out_dir, exchange="auto"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([120, 90], 5000, seed=21)
    torch.manual_seed(3)
    m = VFM(120, 90, 16, device="cuda:0", rng_seed=9)
    m.exchange = exchange
    h = m.fit(X, y, n_epochs=2, batch_size=2000, verbose=False, process_group=dist.group.WORLD,
              X_test=X[:500], y_test=y[:500])
    m.sync_lazy()                     # ("rows": fit() names the next batch, so rows outside two global batches may wait)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"fit_{rank}.npy"), m._flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"elbo_{rank}.npy"), np.array(h["elbo"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["auto", "rows"])
def test_fit_two_ranks_matches_single_rank(exchange, tmp_path):
    """VFM.fit with a process group (row shards, W all-reduce, statistics or rows exchange,
    short last batch, per-epoch evaluation) lands on the same weights as the single-rank fit."""
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path), exchange), nprocs=2, join=True)
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([120, 90], 5000, seed=21)
    torch.manual_seed(3)
    m = VFM(120, 90, 16, device="cuda:0", rng_seed=9)
    h = m.fit(X, y, n_epochs=2, batch_size=2000, verbose=False, X_test=X[:500], y_test=y[:500])
    p0, p1 = np.load(tmp_path / "fit_0.npy"), np.load(tmp_path / "fit_1.npy")
    assert np.allclose(p0, p1, rtol=1e-6, atol=1e-7)
    assert rel_err(p0, m._flat.cpu().numpy()) < 1e-4
    assert rel_err(np.load(tmp_path / "elbo_0.npy"), np.array(h["elbo"])) < 1e-4


@pytest.mark.parametrize("F,d", [(3, 16), (2, 32)])
def test_lazy_statistics_step_is_bitwise_the_dense_one(F, d, monkeypatch):
    """The multi-rank statistics step with the exchange compacted to the globally touched entities and the other rows'
    Adam updates deferred (`exchange_lazy`: catch-up pass + apply stage over the list, vfm_elbo_apply_adam_rows_f32)
    against the same step with every row updated every step: 140 steps over 3 in-process ranks -- a moment-period
    boundary, changing learning rates, predictions in between -- parameters and both moments BIT FOR BIT, replicas
    identical."""
    from thread_ranks import run_ranks
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    from vae_amd.dist import shard_rows
    import copy
    import vae_amd.model as M
    monkeypatch.setattr(M, "_CHECK_WREC", True)       # (every step also checks the packed first-order records)
    sizes, world, B, nb, n_steps = [900, 700, 400][:F], 3, 96, 5, 140
    X, y = synthetic_triples(sizes, nb * B, seed=4, device="cuda")
    results = []
    for lazy in (True, False):
        torch.manual_seed(3)
        first = VFM(field_sizes=sizes, embedding_size=d, device="cuda", rng_seed=11)
        first.exchange, first.exchange_lazy = "stats", lazy
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
            losses, lagged = [], False
            for s in range(n_steps):
                l3, _ = m.train_step(plans[s % nb], lr=0.05 if s % 7 else 0.02, process_group=group)
                lagged = lagged or m._lazy_dirty
                if s % 20 == 3 or s in (127, 128):
                    losses.append(l3.clone())
                if s == 60:
                    losses.append(m.predict(X[:50])["y_pred"].sum().reshape(1).repeat(3))
            m.sync_lazy()
            return m._flat, m._adam_m, m._adam_v, torch.stack(losses), lagged, m._exchanged_floats

        out, _ = run_ranks(world, rank_body, monkeypatch)
        for r in range(1, world):
            assert all(torch.equal(out[r][i], out[0][i]) for i in range(4))
        results.append(out[0])
    lz, dn = results
    assert lz[4] and not dn[4]                       # rows did lag in the lazy run
    assert lz[5] == dn[5] and lz[5] < 0.5 * first.T * (d + 4)      # both exchanged the compact buffer
    for i in range(4):
        assert torch.equal(lz[i], dn[i]), i


@pytest.mark.parametrize("world,B,d,output,n_steps,announce", [(3, 96, 32, "reg", 60, False), (3, 96, 32, "reg", 150, True),
                                                              (5, 4, 16, "class", 60, True), (8, 300, 128, "reg", 16, True)])
def test_rows_exchange_follows_the_single_rank_pipelined_step(world, B, d, output, n_steps, announce, monkeypatch):
    """`exchange = "rows"` (vae_amd/dist.py::step_rows): the ranks all-reduce every row's dloss/dpred and the six ELBO sums
    (B_global + 8 doubles), each samples the records of all the batch's entities from its replica and runs the whole
    batch's backward + Adam itself.  Against one rank taking the software-pipelined step on the whole batch (the same
    record kernels): steps with changing learning rates -- the same per-row arithmetic, the sums of the likelihood terms
    taken shard by shard -- parameters and moments within 1e-4 of each table's largest entry, losses within 1e-4, replicas
    BIT-identical; (5 ranks, 4 rows: one rank's shard is empty).  The d = 128 case stops at 20 steps: 400 entities learning
    300 rows at lr = 0.05 is a chaotic trajectory -- the losses of the two runs are EQUAL for 16 steps and a factor 3 per four
    steps apart after that, and the single rank's own dense and pipelined forms part even faster there."""
    from thread_ranks import run_ranks
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    from vae_amd.dist import shard_rows
    import copy
    sizes, nb = [230, 170], 4
    X, y = synthetic_triples(sizes, nb * B, seed=4, device="cuda", output=output)
    torch.manual_seed(3)
    first = VFM(field_sizes=sizes, embedding_size=d, device="cuda", rng_seed=11, output=output)
    single = copy.deepcopy(first); single._tie(); single.__dict__.pop("_view_cache", None)
    first.exchange = "rows"
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
            # announce: the step is told which batch follows -- it then runs in the look-ahead form (rows of neither
            # global batch wait) and writes the next batch's records itself; step 40 meets a batch nobody announced
            nxt = plans[(s + 1) % nb] if (announce and s != 39) else None
            cur = plans[s % nb] if s != 40 else plans[(s + 2) % nb]
            l3, pr = m.train_step(cur, lr=0.05 if s % 7 else 0.02, process_group=group, next_plan=nxt)
            assert pr.numel() == cur.B
            losses.append(l3.clone())
        lagged = bool(m._lazy_dirty)
        m.sync_lazy()
        return m._flat, m._adam_m, m._adam_v, torch.stack(losses), m._exchanged_floats, lagged

    out, sh = run_ranks(world, rank_body, monkeypatch)
    for r in range(1, world):
        assert all(torch.equal(out[r][i], out[0][i]) for i in range(4)), r
    assert out[0][4] == 2 * (B + 8)
    assert out[0][5] == (announce and B * 2 < 0.5 * sum(sizes))          # rows did wait where a batch leaves most of the table alone
    single.set_training_data(X, nb_train=nb * B)
    single.pipeline, single.lookahead, single.lazy_adam = True, False, False
    single.pipeline_min_T = single.pipeline_min_d = 0
    plans = [single.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(nb)]
    want_l = torch.stack([single.train_step(plans[s % nb] if s != 40 else plans[(s + 2) % nb], lr=0.05 if s % 7 else 0.02,
                                            next_plan=plans[(s + 1) % nb] if s != 39 else None)[0].clone()
                          for s in range(n_steps)])
    single.sync_lazy()
    assert single._zrec is not None                                   # (the single rank did take the pipelined step)
    errs = {"loss": float(((out[0][3] - want_l).abs() / want_l.abs()).max())}
    for name, got in zip(("_flat", "_adam_m", "_adam_v"), out[0][:3]):
        for part, (pa, pb) in enumerate(zip(models[0]._views(got), single._views(getattr(single, name)))):
            errs[(name, part)] = float((pa - pb).abs().max()) / (float(pb.abs().max()) + 1e-30)
    for key, e in errs.items():
        # (alpha's first moment: a cancelling sum over all rows, see test_gpu_state_machine)
        tol = 1e-4 if key == "loss" else (1e-2 if key == ("_adam_m", 2) else 1e-4)
        assert e <= tol, errs
