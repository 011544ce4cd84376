"""Where the HOST time of a training step goes (cProfile, cfg3 shape): `resident` = plans reused, `streamed` = every step builds
the plan of the batch two steps ahead (side stream).  usage: python tools/host_profile.py [resident|streamed] [steps]"""
import cProfile
import os
import pstats
import sys
import time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from vae_amd.model import VFM, sort_rows_within_batches
from vae_amd.data import synthetic_triples
mode = sys.argv[1] if len(sys.argv) > 1 else "streamed"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
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


def resident(n, plans):
    for s in range(n):
        model.train_step(plans[s % NB], next_plan=plans[(s + 1) % NB])


def streamed(n):
    cur = model.plan(*bt[0], defer_readback=True)
    nxt = model.plan_async(*bt[1])
    nx2 = model.plan_async(*bt[2])
    for s in range(n):
        model.train_step(cur, next_plan=nxt, prefetch=bt[(s + 3) % NB] + (False,))
        cur, nxt, nx2 = nxt, nx2, model.prefetched


if mode == "resident":
    plans = [model.plan(*b) for b in bt]
    run = lambda n: resident(n, plans)
else:
    run = streamed
run(50)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{mode}: {steps} steps, host enqueue {t_host / steps * 1e3:.4f} ms/step, wall {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
run(steps)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
