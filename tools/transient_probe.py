"""GPU: per-step HIP-event durations of the forward and the fused backward over the first N steps of a fresh cfg3 run
(look-ahead form), averaged in groups of 20 -- the evidence behind bench.py's settling steps.  At step N either the GPU
is left idle for 0.3 s (RELOAD=idle; at 2N it is kept busy with unrelated copies instead), or the training state is
reloaded (RELOAD=init: the initial values; RELOAD=same: the current ones).  Measured (profiles/r03_transient.txt): the
backward kernel reads 163 us in steps 20-39 and 151-154 us from step 60-260 on; the slow stretch comes back after the
idle period and after either reload (each of which leaves the GPU idle for a few hundred ms), so it is the chip's state
after idleness, not the values.   usage: N=200 RELOAD=idle python tools/transient_probe.py"""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vae_amd.model import VFM
from vae_amd.data import synthetic_triples
dev = torch.device("cuda")
sizes, d, B, nb_train, nbt = [138493, 26744], 128, 100000, 16000210, 16
torch.manual_seed(42)
m = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=1234)
X, y = synthetic_triples(sizes, nbt * B, seed=1000, device=dev)
occ = torch.clamp((torch.bincount(X.reshape(-1), minlength=m.T).double() * (nb_train / (nbt * B))).round().long(), min=1)
m.set_training_data(X, nb_train=nb_train, nb_occ=occ)
m.lr = 1.0 / (1 + nb_train // B)
plans = []
for i in range(nbt):
    xb, yb = X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]
    o = torch.argsort(xb[:, -1], stable=True)
    plans.append(m.plan(xb[o].contiguous(), yb[o].contiguous()))
for i, p in enumerate(plans):
    p.lookahead_rows(plans[(i + 1) % nbt])
torch.cuda.synchronize()
m.train_step(plans[nbt - 1], next_plan=plans[0])
sd0 = m.training_state_dict()
evs = []
N = int(os.environ.get("N", "400"))
import time
for s in range(3 * N):
    if s == N:
        what = os.environ.get("RELOAD", "init")
        if what == "idle":            # the GPU idle for 0.3 s, nothing else changed
            torch.cuda.synchronize(); time.sleep(0.3)
        else:                         # back to the initial parameters and moments ("init"), or the same values reloaded ("same")
            m.load_training_state_dict(sd0 if what == "init" else m.training_state_dict())
    if s == 2 * N and os.environ.get("RELOAD") == "idle":      # the GPU kept busy for 0.3 s with unrelated copies
        a_ = torch.empty(1 << 27, dtype=torch.float32, device=dev); b_ = torch.empty_like(a_)
        torch.cuda.synchronize(); t_ = time.perf_counter()
        while time.perf_counter() - t_ < 0.3:
            for _ in range(10):
                b_.copy_(a_)
            torch.cuda.synchronize()
        del a_, b_
    ev = {}
    def mark(name, ev=ev):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev[name] = e
    m.train_step(plans[s % nbt], next_plan=plans[(s + 1) % nbt], mark=mark)
    evs.append(ev)
torch.cuda.synchronize()
fw = [e["start"].elapsed_time(e["fwd"]) * 1e3 for e in evs]
bw = [e["fwd"].elapsed_time(e["bwd_adam"]) * 1e3 for e in evs]
for lo in range(0, len(evs), 20):
    print(lo, "fwd %.1f" % (sum(fw[lo:lo + 20]) / 20), "bwd %.1f" % (sum(bw[lo:lo + 20]) / 20), "max bwd %.1f" % max(bw[lo:lo + 20]))
