"""Where the streamed loop's period goes, measured with HIP events on the main stream and no profiler attached:
per step  top -> [waits on the plans' build events] -> k_fwd2 -> k_bwd -> next top.
usage: python tools/stream_gap_probe.py [steps]"""
import os
import sys
import time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from vae_amd.model import VFM, sort_rows_within_batches
from vae_amd.data import synthetic_triples
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
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


def streamed(n, rec=None):
    cur = model.plan(*bt[0], defer_readback=True)
    nxt = model.plan_async(*bt[1])
    nx2 = model.plan_async(*bt[2])
    for s in range(n):
        mark = None
        if rec is not None:
            ev = {}
            rec.append(ev)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev["top"] = e
            ev["host_top"] = time.perf_counter()

            def mark(name, ev=ev):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev[name] = e
        model.train_step(cur, next_plan=nxt, prefetch=bt[(s + 3) % NB] + (False,), mark=mark)
        cur, nxt, nx2 = nxt, nx2, model.prefetched


streamed(100)
torch.cuda.synchronize()
t0 = time.perf_counter()
streamed(steps)
torch.cuda.synchronize()
print(f"no events: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step")
rec = []
t0 = time.perf_counter()
streamed(steps, rec)
torch.cuda.synchronize()
print(f"with events: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step; marks: {[k for k in rec[5] if k != 'host_top']}")
acc = {"top->fwd (waits + k_fwd2)": 0.0, "fwd->bwd (k_bwd)": 0.0, "bwd->next top (main stream empty: host late)": 0.0, "period": 0.0}
n = 0
for s in range(50, steps - 1):
    a, b = rec[s], rec[s + 1]
    if "fwd" not in a or "bwd_adam" not in a:
        continue
    acc["top->fwd (waits + k_fwd2)"] += a["top"].elapsed_time(a["fwd"]) * 1e3
    acc["fwd->bwd (k_bwd)"] += a["fwd"].elapsed_time(a["bwd_adam"]) * 1e3
    acc["bwd->next top (main stream empty: host late)"] += a["bwd_adam"].elapsed_time(b["top"]) * 1e3
    acc["period"] += a["top"].elapsed_time(b["top"]) * 1e3
    n += 1
for k, v in acc.items():
    print(f"  {k:50s} {v / n:8.2f} us")
