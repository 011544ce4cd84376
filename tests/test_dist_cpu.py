"""CPU, world_size 2, gloo: the row-sharded step reproduces the single-process step.

The per-rank arithmetic is done by the fp64 row-wise oracle (test infrastructure); the exchange
logic under test is the product's: vae_amd.dist.shard_rows / sum_normalisers / allreduce_flat /
prior_terms_flag, used exactly as VFM.fit / VFM.train_step use them."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import Case, PARAM_KEYS, rel_err
from oracle import vfm_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.dist import shard_rows, sum_normalisers, allreduce_flat, prior_terms_flag
    c = Case(name)
    P = c.params(np.float64)
    e0, ew, ev = c.eps("f64" if f"f64_eps0" in c.z.files else "f32")
    B = len(c.y)
    a, b = shard_rows(0, B, rank, world)
    x, y = c.x[a:b], c.y[a:b].astype(np.float64)
    # 1. batch normalisers: local partial sums -> SUM over ranks
    W = torch.tensor(O.batch_norms(x, c.nb_occ))
    sum_normalisers(W)
    # 2. local ELBO pieces with global W and B_global
    r = O.rowwise_elbo(P, x, y, c.nb_occ, c.group_hi, c.group_n, c.nb_train, e0, ew, ev, c.output,
                       W=W.numpy(), B_global=B)
    no_prior = prior_terms_flag(rank)
    m0, s0 = float(P["global_bias_mean"][0]), float(P["global_bias_scale"][0])
    loss = r["loss"] - (r["kl0"] if no_prior else 0.0)
    g_m0 = r["g_global_bias_mean"][0] - (m0 if no_prior else 0.0)
    g_s0 = r["g_global_bias_scale"][0] - (np.sign(s0) * (abs(s0) - 1 / abs(s0)) if no_prior else 0.0)
    flat = torch.tensor(np.concatenate([r["g_entity_params"].reshape(-1), r["g_bias_params"].reshape(-1),
                                        r["g_alpha"], [g_m0], [g_s0], [loss]]))
    # 3. the one collective
    allreduce_flat(flat)
    np.save(os.path.join(out_dir, f"flat_{rank}.npy"), flat.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["quirk_reg_d8", "fraction_class_d5"])
def test_two_rank_step_equals_single_process(name, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    c = Case(name)
    tag = "f64"
    e0, ew, ev = c.eps(tag)
    full = O.rowwise_elbo(c.params(np.float64), c.x, c.y.astype(np.float64), c.nb_occ, c.group_hi,
                          c.group_n, c.nb_train, e0, ew, ev, c.output)
    want = np.concatenate([full["g_entity_params"].reshape(-1), full["g_bias_params"].reshape(-1),
                           full["g_alpha"], full["g_global_bias_mean"], full["g_global_bias_scale"],
                           [full["loss"]]])
    got0 = np.load(tmp_path / "flat_0.npy")
    got1 = np.load(tmp_path / "flat_1.npy")
    assert np.array_equal(got0, got1)                       # replicas stay identical
    assert rel_err(got0, want) < 1e-12
    # and both equal the reference's own numbers
    assert abs(got0[-1] - c.expected("loss", tag)[0]) / abs(c.expected("loss", tag)[0]) < 1e-7


def test_shard_rows_partitions_every_batch():
    from vae_amd.dist import shard_rows
    for lo, hi in ((0, 10), (7, 8), (100, 100), (3, 1000003)):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(lo, hi, r, world) for r in range(world)]
            assert spans[0][0] == lo and spans[-1][1] == hi
            for (a, b), (c_, d) in zip(spans[:-1], spans[1:]):
                assert b == c_ and a <= b
            assert sum(b - a for a, b in spans) == hi - lo


class _FakePlan:
    """What vae_amd.dist.global_touched needs of a plan: the sorted ids of the entities its shard contains."""

    def __init__(self, ids):
        self._ids = ids

    def touched_ids(self):
        return self._ids


def _gids_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.dist import global_touched
    T = 1000
    g = np.random.default_rng(100 + rank)
    # shards of different lengths (rank 2's is empty): the id lists are padded to the longest before the all-gather
    n = [37, 120, 0][rank]
    ids = torch.tensor(np.sort(g.choice(T, n, replace=False)).astype(np.int32))
    plan = _FakePlan(ids)
    got = global_touched(plan, dist.group.WORLD, T)
    again = global_touched(plan, dist.group.WORLD, T)          # cached on the plan: no second collective
    assert again is got
    np.save(os.path.join(out_dir, f"gids_{rank}.npy"), got.numpy())
    np.save(os.path.join(out_dir, f"mine_{rank}.npy"), ids.numpy())
    dist.destroy_process_group()


def test_global_touched_set_is_the_sorted_union_on_every_rank(tmp_path):
    """The set the compacted statistics exchange is built on (vae_amd/dist.py::global_touched): every rank ends up with
    the same sorted union of the ranks' touched ids -- ragged list lengths, an empty shard -- over gloo, world size 3."""
    world = 3
    mp.spawn(_gids_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = np.unique(np.concatenate([np.load(tmp_path / f"mine_{r}.npy") for r in range(world)])).astype(np.int64)
    for r in range(world):
        got = np.load(tmp_path / f"gids_{r}.npy")
        assert got.dtype == np.int64 and np.array_equal(got, want), r


def _gather_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_amd.dist import gather_shards, shard_rows
    B, F = 7, 2                                      # 7 rows over 4 ranks: blocks of 2, 2, 2, 1 ... and over 8: some empty
    g = np.random.default_rng(5)
    X = torch.tensor(g.integers(0, 50, (B, F)))
    Y = torch.tensor(g.random(B).astype(np.float32))
    a, b = shard_rows(0, B, rank, world)
    Xg, Yg, off = gather_shards(X[a:b], Y[a:b], dist.group.WORLD)
    assert off == a and torch.equal(Xg, X) and torch.equal(Yg, Y), (rank, off, a)
    open(os.path.join(out_dir, f"ok_{rank}"), "w").write("1")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gather_shards_rebuilds_the_batch_on_every_rank(world, tmp_path):
    """What the rows exchange (vae_amd/dist.py::step_rows) is built on: the ranks' contiguous row blocks -- ragged, the
    last ones possibly short -- all-gathered into the whole batch, in row order, on every rank (gloo)."""
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok_{r}") for r in range(world))
