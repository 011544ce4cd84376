"""GPU: the multi-rank step through the REAL collective library.  This pool leases one GPU, and RCCL (like NCCL) refuses
two ranks on one device, so what can run here is a communicator of world size 1: `init_process_group("nccl")`, the
all-reduce / all-gather launches on RCCL's own stream, `async_op=True` + `wait()` ordering against the step's kernels on
the current stream, the compacted exchange and the lazy apply stage -- everything about the backend except a second
peer.  The step must equal the single-process step (per rank the reference's loop body, vfm-torch.py:351-370)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    from vae_amd.data import synthetic_triples
    return synthetic_triples([3000, 900], 4 * 1500, seed=21, device="cuda:0")


def _model():
    from vae_amd.model import VFM
    torch.manual_seed(3)
    return VFM(3000, 900, 32, device="cuda:0", rng_seed=9)


def _worker(rank, world, port, exchange, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    pg = dist.group.WORLD
    one = torch.ones(1, device="cuda:0")
    dist.all_reduce(one, group=pg)
    assert float(one.item()) == 1.0 and dist.get_backend(pg) == "nccl"
    X, y = _data()
    m = _model()
    m.exchange = exchange
    m.set_training_data(X, nb_train=6000)
    plans = [m.plan(X[i * 1500:(i + 1) * 1500], y[i * 1500:(i + 1) * 1500], B_global=1500, process_group=pg) for i in range(4)]
    losses = []
    for s in range(12):
        l3, _ = m.train_step(plans[s % 4], lr=0.03, process_group=pg)
        losses.append(l3.cpu().numpy().copy())
    lagged = bool(m._lazy_dirty)
    m.sync_lazy()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "p.npy"), m._flat.cpu().numpy())
    np.save(os.path.join(out_dir, "l.npy"), np.array(losses))
    np.save(os.path.join(out_dir, "meta.npy"), np.array([int(lagged), int(m._exchanged_floats)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["stats", "grads", "rows"])
def test_step_through_rccl_world_size_one(exchange, tmp_path):
    mp.spawn(_worker, args=(1, _free_port(), exchange, str(tmp_path)), nprocs=1, join=True)
    X, y = _data()
    m = _model()
    m.set_training_data(X, nb_train=6000)
    plans = [m.plan(X[i * 1500:(i + 1) * 1500], y[i * 1500:(i + 1) * 1500]) for i in range(4)]
    ref = np.array([m.train_step(plans[s % 4], lr=0.03, fused=False)[0].cpu().numpy().copy() for s in range(12)])
    assert rel_err(np.load(tmp_path / "l.npy"), ref) < 1e-5
    assert rel_err(np.load(tmp_path / "p.npy"), m._flat.cpu().numpy()) < 1e-5
    lagged, xfloats = np.load(tmp_path / "meta.npy")
    if exchange == "stats":          # 1,500 rows over 3,900 entities: the compact + lazy form ran through RCCL
        assert lagged == 1 and 0 < xfloats < 0.8 * 3900 * 36
    if exchange == "rows":           # every row's dloss/dpred + the six sums, as doubles (all_gather of the ids + a float64 all-reduce through RCCL)
        assert lagged == 0 and xfloats == 2 * (1500 + 8)
