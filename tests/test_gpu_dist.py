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
    if exchange == "sharded":
        m.sync_params(dist.group.WORLD)
    m.sync_lazy()                     # ("rows": fit() names the next batch, so rows outside two global batches may wait)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"fit_{rank}.npy"), m._flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"elbo_{rank}.npy"), np.array(h["elbo"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["auto", "sharded", "rows"])
def test_fit_two_ranks_matches_single_rank(exchange, tmp_path):
    """VFM.fit with a process group (row shards, W all-reduce, statistics exchange or entity-sharded
    tables, short last batch, per-epoch evaluation) lands on the same weights as the single-rank fit."""
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


def _sharded_worker(rank, world, port, name, out_dir, use_tables):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.dist import shard_rows
    dev = torch.device("cuda:0")
    c = Case(name)
    m = _make_model(c, dev)
    m.exchange = "sharded"
    m.shard_gather = use_tables                   # cover the gather form and both dense-table forms
    m.shard_deterministic = (world == 3)
    eps = None
    if use_tables:
        e0, ew, ev = c.eps("f32")
        eps = (torch.tensor(ev, device=dev), torch.tensor(ew, device=dev), torch.tensor(e0, device=dev))
    B = len(c.y)
    losses = []
    a, b = shard_rows(0, B, rank, world)
    plan = m.plan(torch.tensor(c.x[a:b]), torch.tensor(c.y[a:b]), B_global=B, process_group=dist.group.WORLD)
    for step in range(3):
        loss3, _ = m.train_step(plan, lr=0.05, eps=eps, process_group=dist.group.WORLD)
        losses.append(loss3.cpu().numpy().copy())
    m.sync_params(dist.group.WORLD)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"sh_params_{rank}.npy"), m._flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"sh_loss_{rank}.npy"), np.array(losses))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name,use_tables", [("ml100k_reg_d20", True), ("ml100k_reg_d20", False),
                                             ("dup_class_d12", True), ("softplus_reg_d8", False)])
def test_entity_sharded_step_matches_one_rank(name, use_tables, world, tmp_path):
    """Entity-sharded mode (tables partitioned by e mod N, two all-to-alls per step) == the 1-rank step:
    losses, and after sync_params every parameter."""
    mp.spawn(_sharded_worker, args=(world, _free_port(), name, str(tmp_path), use_tables), nprocs=world, join=True)
    c = Case(name)
    dev = torch.device("cuda:0")
    m = _make_model(c, dev)
    eps = None
    if use_tables:
        e0, ew, ev = c.eps("f32")
        eps = (torch.tensor(ev, device=dev), torch.tensor(ew, device=dev), torch.tensor(e0, device=dev))
    plan = m.plan(torch.tensor(c.x), torch.tensor(c.y))
    ref_losses = []
    for step in range(3):
        loss3, _ = m.train_step(plan, lr=0.05, eps=eps, fused=False)
        ref_losses.append(loss3.cpu().numpy().copy())
    want = m._flat.cpu().numpy()
    ps = [np.load(tmp_path / f"sh_params_{r}.npy") for r in range(world)]
    n_tab = m._off_scal                                    # tables (entity + bias): identical after sync_params
    for p in ps[1:]:
        assert np.array_equal(ps[0][:n_tab], p[:n_tab])
        assert np.allclose(ps[0][n_tab:], p[n_tab:], rtol=1e-6)    # replicated scalars
    assert rel_err(ps[0], want) < 2e-5
    l0 = np.load(tmp_path / "sh_loss_0.npy")
    assert rel_err(l0[:, 0], np.array(ref_losses)[:, 0]) < 1e-5


def _dims_worker(rank, world, port, name, out_dir, use_tables):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    c = Case(name)
    m = _make_model(c, dev)
    m.exchange = "dims"
    eps = None
    if use_tables:
        e0, ew, ev = c.eps("f32")
        eps = (torch.tensor(ev, device=dev), torch.tensor(ew, device=dev), torch.tensor(e0, device=dev))
    plan = m.plan(torch.tensor(c.x), torch.tensor(c.y), process_group=dist.group.WORLD)   # ALL rows on every rank
    losses, preds = [], []
    for step in range(3):
        loss3, pred = m.train_step(plan, lr=0.05, eps=eps, process_group=dist.group.WORLD)
        losses.append(loss3.cpu().numpy().copy())
        preds.append(pred.cpu().numpy().copy())
    st = m.training_state_dict()             # collective: gathers parameters and moments
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"dm_params_{rank}.npy"), m._flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"dm_m_{rank}.npy"), st["adam"]["m"].numpy())
    np.save(os.path.join(out_dir, f"dm_loss_{rank}.npy"), np.array(losses))
    np.save(os.path.join(out_dir, f"dm_pred_{rank}.npy"), np.array(preds))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name,use_tables", [(2, "ml20m_reg_d128", True), (4, "ml20m_reg_d128", False),
                                                   (2, "softplus_multi_class_d8_s2", False)])
def test_dimension_sharded_step_matches_one_rank(world, name, use_tables, tmp_path):
    """Embedding-dimension-sharded mode (every rank: all rows, d / N coordinates; one all-reduce of B + 1 floats
    per step) == the 1-rank step on the same batch: losses, predictions, and after the gather every parameter
    and Adam moment."""
    if name == "softplus_multi_class_d8_s2":        # d = 8, two samples: not shardable this way -> loud error
        from vae_amd.dims import supported
        assert not supported(8, world, 2) and not supported(20, 2) and supported(128, 8) and supported(16, 2)
        return
    mp.spawn(_dims_worker, args=(world, _free_port(), name, str(tmp_path), use_tables), nprocs=world, join=True)
    c = Case(name)
    dev = torch.device("cuda:0")
    m = _make_model(c, dev)
    eps = None
    if use_tables:
        e0, ew, ev = c.eps("f32")
        eps = (torch.tensor(ev, device=dev), torch.tensor(ew, device=dev), torch.tensor(e0, device=dev))
    plan = m.plan(torch.tensor(c.x), torch.tensor(c.y))
    ref_losses, ref_preds = [], []
    for step in range(3):
        loss3, pred = m.train_step(plan, lr=0.05, eps=eps)
        ref_losses.append(loss3.cpu().numpy().copy())
        ref_preds.append(pred.cpu().numpy().copy())
    want = m._flat.cpu().numpy()
    m._set_moment_form(False)
    want_m = m._adam_m.cpu().numpy()
    ps = [np.load(tmp_path / f"dm_params_{r}.npy") for r in range(world)]
    for p in ps[1:]:
        assert np.array_equal(ps[0], p)                      # identical full tables after the gather
    assert rel_err(ps[0], want) < 2e-5
    assert rel_err(np.load(tmp_path / "dm_loss_0.npy"), np.array(ref_losses)) < 1e-5
    assert np.array_equal(np.load(tmp_path / "dm_loss_0.npy"), np.load(tmp_path / f"dm_loss_{world - 1}.npy"))
    assert rel_err(np.load(tmp_path / "dm_pred_0.npy"), np.array(ref_preds)) < 2e-5
    got_m = np.load(tmp_path / "dm_m_0.npy")
    if m.scaled_moments:                                      # the checkpoint stores the buffers in their form
        k = 3 % 128
        got_m = got_m * (0.9 ** k)
    assert rel_err(got_m, want_m) < 1e-4


def _dims_fit_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([60, 50], 3000, seed=4)
    torch.manual_seed(1)
    m = VFM(60, 50, 16, device="cuda:0", rng_seed=5)
    m.exchange = "dims"                        # opt-in ("auto" = the row-sharded "stats" exchange)
    h = m.fit(X[:2400], y[:2400], n_epochs=3, batch_size=1000, X_test=X[2400:], y_test=y[2400:], verbose=False,
              process_group=dist.group.WORLD)
    assert m.exchange == "dims"
    np.save(os.path.join(out_dir, f"df_elbo_{rank}.npy"), np.array(h["elbo"]))
    np.save(os.path.join(out_dir, f"df_rmse_{rank}.npy"), np.array([t["rmse_of_mean"] for t in h["test"]]))
    np.save(os.path.join(out_dir, f"df_params_{rank}.npy"), m._flat.cpu().numpy())
    dist.destroy_process_group()


def test_fit_dimension_sharded_matches_single_rank(tmp_path):
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    mp.spawn(_dims_fit_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    X, y = synthetic_triples([60, 50], 3000, seed=4)
    torch.manual_seed(1)
    m = VFM(60, 50, 16, device="cuda:0", rng_seed=5)
    h = m.fit(X[:2400], y[:2400], n_epochs=3, batch_size=1000, X_test=X[2400:], y_test=y[2400:], verbose=False)
    assert rel_err(np.load(tmp_path / "df_elbo_0.npy"), np.array(h["elbo"])) < 1e-4
    assert np.array_equal(np.load(tmp_path / "df_params_0.npy"), np.load(tmp_path / "df_params_1.npy"))
    assert rel_err(np.load(tmp_path / "df_params_0.npy"), m._flat.cpu().numpy()) < 1e-3
    assert rel_err(np.load(tmp_path / "df_rmse_0.npy"), np.array([t["rmse_of_mean"] for t in h["test"]])) < 1e-3


def _dims_general_worker(rank, world, port, out_dir, output):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, X, y = _general_model(output)
    m.exchange = "dims"
    plan = m.plan(X, y, process_group=dist.group.WORLD)
    losses = []
    for step in range(3):
        loss3, _ = m.train_step(plan, lr=0.03, process_group=dist.group.WORLD)
        losses.append(loss3.cpu().numpy().copy())
    m.sync_params(dist.group.WORLD)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"dg_params_{rank}.npy"), m._flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"dg_loss_{rank}.npy"), np.array(losses))
    dist.destroy_process_group()


def _general_model(output):
    """Three fields (the general-F kernels), d = 32, softplus link, skewed ids (heavy lists)."""
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    sizes = [40, 30, 50]
    X, y = synthetic_triples(sizes, 3000, seed=8, output=output, zipf=1.2)
    torch.manual_seed(2)
    m = VFM(field_sizes=sizes, embedding_size=32, output=output, device="cuda:0", rng_seed=6, link="softplus")
    m.set_training_data(X, nb_train=3000)
    return m, X, y


@pytest.mark.parametrize("output", ["reg", "class"])
def test_dimension_sharded_general_fields(output, tmp_path):
    world = 2
    mp.spawn(_dims_general_worker, args=(world, _free_port(), str(tmp_path), output), nprocs=world, join=True)
    m, X, y = _general_model(output)
    plan = m.plan(X, y)
    ref = []
    for step in range(3):
        loss3, _ = m.train_step(plan, lr=0.03)
        ref.append(loss3.cpu().numpy().copy())
    p0, p1 = np.load(tmp_path / "dg_params_0.npy"), np.load(tmp_path / "dg_params_1.npy")
    assert np.array_equal(p0, p1)
    assert rel_err(p0, m._flat.cpu().numpy()) < 5e-5
    assert rel_err(np.load(tmp_path / "dg_loss_0.npy"), np.array(ref)) < 1e-5


def _tiny_model():
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([9, 7], 2, seed=11)
    torch.manual_seed(4)
    m = VFM(9, 7, 8, device="cuda:0", rng_seed=2)
    m.set_training_data(X, nb_train=10)
    return m, X, y


def _tiny_sharded_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.dist import shard_rows
    m, X, y = _tiny_model()
    m.exchange = "sharded"
    a, b = shard_rows(0, 2, rank, world)            # 2 rows over 3 ranks: one rank has no rows, owners without requests
    plan = m.plan(X[a:b], y[a:b], B_global=2, process_group=dist.group.WORLD)
    for _ in range(2):
        loss3, _ = m.train_step(plan, lr=0.03, process_group=dist.group.WORLD)
    m.sync_params(dist.group.WORLD)
    np.save(os.path.join(out_dir, f"ts_params_{rank}.npy"), m._flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"ts_loss_{rank}.npy"), loss3.cpu().numpy())
    dist.destroy_process_group()


def test_entity_sharded_more_ranks_than_rows(tmp_path):
    """Two rows over three ranks (found by tools/fuzz_multirank.py): an empty row shard, owners nobody asks
    anything of -- the entity-sharded step still equals the single-process one."""
    mp.spawn(_tiny_sharded_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    m, X, y = _tiny_model()
    plan = m.plan(X, y)
    for _ in range(2):
        loss3, _ = m.train_step(plan, lr=0.03)
    assert rel_err(np.load(tmp_path / "ts_params_0.npy"), m._flat.cpu().numpy()) < 1e-5
    assert rel_err(np.load(tmp_path / "ts_loss_0.npy"), loss3.cpu().numpy()) < 1e-5
    assert np.array_equal(np.load(tmp_path / "ts_params_0.npy")[: m._off_scal], np.load(tmp_path / "ts_params_2.npy")[: m._off_scal])


def _dims_resume_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    X, y = synthetic_triples([70, 50], 1500, seed=12)
    pg = dist.group.WORLD

    def fresh():
        torch.manual_seed(3)
        m = VFM(70, 50, 32, device="cuda:0", rng_seed=21)
        m.set_training_data(X, nb_train=1500)
        m.exchange, m.lr = "dims", 0.04
        return m

    a = fresh()
    plans = [a.plan(X[i:i + 500], y[i:i + 500], process_group=pg) for i in range(0, 1500, 500)]
    for s in range(5):
        a.train_step(plans[s % 3], process_group=pg)
    a.sync_params(pg)
    b = fresh()
    for s in range(3):
        b.train_step(plans[s % 3], process_group=pg)
    ckpt = b.training_state_dict()                 # collective: gathers the slices
    c = fresh()
    c.load_training_state_dict(ckpt)
    for s in range(3, 5):
        c.train_step(plans[s % 3], process_group=pg)
    c.sync_params(pg)
    torch.cuda.synchronize()
    ok = torch.equal(a._flat, c._flat) and torch.equal(a._adam_m, c._adam_m) and torch.equal(a._adam_v, c._adam_v)
    np.save(os.path.join(out_dir, f"dr_ok_{rank}.npy"), np.array([int(ok), int(c._adam_t), int(c.global_step)]))
    dist.destroy_process_group()


def test_dimension_sharded_checkpoint_resume_is_bit_exact(tmp_path):
    """training_state_dict() / load_training_state_dict() in the dimension-sharded mode: the checkpoint holds the
    gathered full tables and moments; resuming in a fresh model continues bit for bit."""
    mp.spawn(_dims_resume_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, t, gs = np.load(tmp_path / f"dr_ok_{r}.npy")
        assert ok == 1 and t == 5 and gs == 5


def _switch_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    from vae_amd.dist import shard_rows
    X, y = synthetic_triples([70, 50], 600, seed=12)
    pg = dist.group.WORLD
    torch.manual_seed(3)
    m = VFM(70, 50, 32, device="cuda:0", rng_seed=21)
    m.set_training_data(X, nb_train=600)
    m.lr = 0.04
    a, b = shard_rows(0, 600, rank, world)
    for mode in ("sharded", "dims", "stats", "dims", "sharded", "grads"):      # two steps each
        m.exchange = mode
        plan = (m.plan(X, y, process_group=pg) if mode == "dims"
                else m.plan(X[a:b], y[a:b], B_global=600, process_group=pg))
        for _ in range(2):
            loss3, _ = m.train_step(plan, process_group=pg)
    if m._stale_group is not None:
        m.sync_params(pg)
    m._set_moment_form(False)
    np.save(os.path.join(out_dir, f"sw_{rank}.npy"), torch.cat([m._flat, m._adam_m, m._adam_v]).cpu().numpy())
    dist.destroy_process_group()


def test_switching_exchange_modes_mid_training(tmp_path):
    """Changing VFM.exchange between steps (sharded -> dims -> stats -> dims -> sharded -> grads) keeps training the
    same model: parameters AND Adam moments are gathered when the mode that left them stale is left."""
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    mp.spawn(_switch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    X, y = synthetic_triples([70, 50], 600, seed=12)
    torch.manual_seed(3)
    m = VFM(70, 50, 32, device="cuda:0", rng_seed=21)
    m.set_training_data(X, nb_train=600)
    m.lr = 0.04
    plan = m.plan(X, y)
    for _ in range(12):
        m.train_step(plan)
    m._set_moment_form(False)
    want = torch.cat([m._flat, m._adam_m, m._adam_v]).cpu().numpy()
    g0, g1 = np.load(tmp_path / "sw_0.npy"), np.load(tmp_path / "sw_1.npy")
    n = m._n_flat
    assert rel_err(g0[:n], want[:n]) < 1e-4 and rel_err(g1[:n], want[:n]) < 1e-4
    assert rel_err(g0[n:2 * n], want[n:2 * n]) < 1e-3 and rel_err(g0[2 * n:], want[2 * n:]) < 1e-3
    assert np.allclose(g0, g1, rtol=1e-5, atol=1e-6 * np.abs(g0).max())


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
                                                              (5, 4, 16, "class", 60, True), (8, 300, 128, "reg", 20, True)])
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
