"""Splits the streamed loop's extra time into INTERFERENCE (index builds running beside the step's kernels) and
DEPENDENCE (the step waiting for a build, on the host or through a stream event): the same loop three ways at cfg3 --
  resident        plans kept, nothing built inside the loop
  beside          plans kept AND a build enqueued per step on the side streams, its result never used (interference only)
  streamed        the real thing: every step uses the plan built `depth` steps earlier
usage: python tools/stream_split_probe.py [steps]"""
import os
import sys
import time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from vae_amd.model import VFM, sort_rows_within_batches
from vae_amd.data import synthetic_triples
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda")
sizes, d, nb_train, B, NB = [138493, 26744], 128, 16000210, 100000, 16
torch.manual_seed(42)
model = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=1234)
X, y = synthetic_triples(sizes, NB * B, seed=1000, device=dev)
occ = torch.clamp(torch.bincount(X.reshape(-1), minlength=sum(sizes)), min=1)
model.set_training_data(X, nb_train=nb_train, nb_occ=occ)
model.lr = 1.0 / (1 + nb_train // B)
X, y = sort_rows_within_batches(X, y, B)
bt = [(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(NB)]
plans = [model.plan(*b) for b in bt]
D = int(model.plan_prefetch_depth)


def resident(n):
    for s in range(n):
        model.train_step(plans[s % NB], next_plan=plans[(s + 1) % NB])


def beside(n):
    keep = []
    for s in range(n):
        model.train_step(plans[s % NB], next_plan=plans[(s + 1) % NB], prefetch=bt[(s + D) % NB] + (False,))
        keep.append(model.prefetched)          # (never used; kept alive a few steps like the real loop does)
        if len(keep) > D:
            keep.pop(0)


def streamed(n):
    model.plan_streams().wait_current()
    q = [model.plan(*bt[0], defer_readback=True)] + [model.plan_async(*bt[k]) for k in range(1, D)]
    for s in range(n):
        model.train_step(q[0], next_plan=q[1], prefetch=bt[(s + D) % NB] + (False,))
        q.pop(0)
        q.append(model.prefetched)


for rnd in range(2):
    for name, fn in (("resident", resident), ("beside", beside), ("streamed", streamed)):
        fn(200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"{name:10s} {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step (host {th / steps * 1e3:.4f})", flush=True)
