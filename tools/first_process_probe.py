"""Is the FIRST process on a fresh box slower, and where?  Host time of every train_step call of a resident-plans loop at
cfg3 (perf_counter around the call), the spikes listed with their step numbers; region times every 250 steps.
usage (as the first command of a gpurun call, then again): python tools/first_process_probe.py [steps]"""
import os
import resource
import sys
import time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
t_imp = time.perf_counter()
import torch
from vae_amd.model import VFM, sort_rows_within_batches
from vae_amd.data import synthetic_triples
print(f"imports {time.perf_counter() - t_imp:.1f} s", flush=True)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda")
sizes, d, nb_train, B, NB = [138493, 26744], 128, 16000210, 100000, 16
torch.manual_seed(42)
model = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=1234)
X, y = synthetic_triples(sizes, NB * B, seed=1000, device=dev)
occ = torch.clamp(torch.bincount(X.reshape(-1), minlength=sum(sizes)), min=1)
model.set_training_data(X, nb_train=nb_train, nb_occ=occ)
model.lr = 1.0 / (1 + nb_train // B)
X, y = sort_rows_within_batches(X, y, B)
plans = [model.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(NB)]
torch.cuda.synchronize()
host = []
ru0 = resource.getrusage(resource.RUSAGE_SELF)
print("loadavg", open("/proc/loadavg").read().strip(), "| pressure cpu:", open("/proc/pressure/cpu").read().split("\n")[0] if os.path.exists("/proc/pressure/cpu") else "n/a", flush=True)
t_region = time.perf_counter()
evs = []
for s in range(steps):
    t0 = time.perf_counter()
    mark = None
    if s % 5 == 0:                 # HIP events on every 5th step, as bench.py's commanded region records them
        ev = {}
        evs.append(ev)

        def mark(name, ev=ev):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev[name] = e
    model.train_step(plans[s % NB], next_plan=plans[(s + 1) % NB], mark=mark)
    host.append(time.perf_counter() - t0)
    if (s + 1) % 250 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        h = host[-250:]
        bw = [e["fwd"].elapsed_time(e["bwd_adam"]) * 1e3 for e in evs if "fwd" in e and "bwd_adam" in e]
        fw = [e["start"].elapsed_time(e["fwd"]) * 1e3 for e in evs if "fwd" in e and "start" in e]
        evs.clear()
        print(f"[events: start->fwd {sum(fw) / max(len(fw), 1):6.1f} us, fwd->bwd mean {sum(bw) / max(len(bw), 1):6.1f} max {max(bw) if bw else 0:7.1f} us] ", end="")
        print(f"steps {s - 249:5d}..{s:5d}: {(t1 - t_region) / 250 * 1e3:.4f} ms/step  host mean {sum(h) / 250 * 1e6:6.1f} us, max {max(h) * 1e6:8.1f} us, "
              f"> 0.5 ms: {sum(1 for v in h if v > 5e-4)}", end="")
        ru = resource.getrusage(resource.RUSAGE_SELF)
        print(f" | majflt {ru.ru_majflt - ru0.ru_majflt} minflt {ru.ru_minflt - ru0.ru_minflt} invol.ctxsw {ru.ru_nivcsw - ru0.ru_nivcsw} vol.ctxsw {ru.ru_nvcsw - ru0.ru_nvcsw}", flush=True)
        ru0 = ru
        t_region = time.perf_counter()
big = sorted(((v, i) for i, v in enumerate(host)), reverse=True)[:15]
print("largest host step times (us @ step):", [(round(v * 1e6), i) for v, i in big])
