#!/usr/bin/env python3
"""Whole multi-rank step (Python + launches + collectives on a 1-rank gloo group) at cfg3 on one GPU: shows the
host-side cost of the step next to the kernel time (tools/shard_probe.py).  EXCHANGE=sharded|stats|grads, or
EXCHANGE=dims DIMS_N=8: rank 0's work of the embedding-dimension-sharded step at the N-rank shape (all N*B rows,
d/N coordinates; the all-reduce replaced by the identity)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from vae_amd.model import VFM
from vae_amd.data import synthetic_triples

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
if os.environ.get("LOCAL_COLLECTIVES", "1") == "1":
    # a 1-rank group: every collective is the identity -- replace gloo's host-staged copies by
    # device-side ones so that what remains is the step's own Python + launch + kernel time
    import vae_amd.sharded as _sh
    import vae_amd.dist as _vd

    def _a2a(out, inp, *a, **k):
        out.copy_(inp)

    def _ar(t, *a, **k):
        class _W:
            def wait(self):
                pass
        return _W()

    dist.all_to_all_single = _a2a
    dist.all_reduce = _ar
DIMS_N = int(os.environ.get("DIMS_N", "0"))
if DIMS_N > 1:          # pretend to be rank 0 of N
    dist.get_world_size = lambda group=None: DIMS_N
    dist.get_rank = lambda group=None: 0
dev = torch.device("cuda")
sizes, d, B, nb_train = [138493, 26744], 128, 100000, 16000210
torch.manual_seed(42)
m = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=3)
m.exchange = os.environ.get("EXCHANGE", "sharded")
if DIMS_N > 1:
    B *= DIMS_N
X, y = synthetic_triples(sizes, 4 * B, seed=1, device=dev)
occ = torch.clamp(torch.bincount(X.reshape(-1), minlength=sum(sizes)) * 40, min=1)
m.set_training_data(X, nb_train=nb_train, nb_occ=occ)
m.lr = 0.006
plans = [m.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B], process_group=dist.group.WORLD) for i in range(4)]
for s in range(10):
    m.train_step(plans[s % 4], process_group=dist.group.WORLD)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for s in range(n):
    m.train_step(plans[s % 4], process_group=dist.group.WORLD)
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(m.exchange, "ms/step", round(dt / n * 1e3, 4), "host enqueue ms/step", round(th / n * 1e3, 4))
if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    for s in range(n):
        m.train_step(plans[s % 4], process_group=dist.group.WORLD)
    pr.disable()
    torch.cuda.synchronize()
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(18)
    print(st.getvalue()[:4000])
dist.destroy_process_group()
