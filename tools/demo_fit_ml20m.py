#!/usr/bin/env python3
"""End-to-end fit()/predict() at ML-20M shape on one MI355X (synthetic low-rank ratings): times whole
epochs of the reference loop (vfm-torch.py:347-422) through VFM.fit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd.model import VFM

dev = torch.device("cuda")
N, M, d, n = 138493, 26744, 128, 20_000_263
g = torch.Generator(device=dev).manual_seed(0)
k = 8
U, V = torch.randn(N, k, generator=g, device=dev) * 0.7, torch.randn(M, k, generator=g, device=dev) * 0.7
u = torch.randint(0, N, (n,), generator=g, device=dev)
i = torch.randint(0, M, (n,), generator=g, device=dev)
y = (3.2 + (U[u] * V[i]).sum(1) + 0.3 * torch.randn(n, generator=g, device=dev)).clamp(1, 5)
X = torch.stack([u, i + N], 1)
if os.environ.get("USER_ORDER"):     # USER_ORDER=1: the TRAINING rows in the data files' order (ratings.csv is sorted by user):
    ntr0 = int(0.8 * n)              # a batch = the consecutive ratings of ~700 users; item popularity Zipf(1.1)
    w = 1.0 / torch.arange(1, M + 1, dtype=torch.float64, device=dev) ** 1.1
    i = torch.multinomial(w, n, replacement=True, generator=g)
    y = (3.2 + (U[u] * V[i]).sum(1) + 0.3 * torch.randn(n, generator=g, device=dev)).clamp(1, 5)
    X = torch.stack([u, i + N], 1)
    o = torch.argsort(X[:ntr0, 0], stable=True)
    X = torch.cat([X[:ntr0][o], X[ntr0:]]).contiguous()
    y = torch.cat([y[:ntr0][o], y[ntr0:]]).contiguous()
ntr = int(0.8 * n)
torch.manual_seed(42)
m = VFM(N, M, d, device=dev, rng_seed=1)
for kv in os.environ.get("VFM_SET", "").split(","):        # e.g. VFM_SET=pipeline=False,pipeline_min_touch=0 (A/B of the step-form rules)
    if "=" in kv:
        key, val = kv.split("=")
        setattr(m, key, {"True": True, "False": False}.get(val, float(val) if val.replace(".", "").isdigit() else val))
t0 = time.perf_counter()
hist = m.fit(X[:ntr], y[:ntr], n_epochs=int(os.environ.get("EPOCHS", "6")), batch_size=100000,
             X_test=X[ntr:], y_test=y[ntr:], display_every=1, verbose=True,
             stream_plans=bool(os.environ.get("STREAM_PLANS")))      # STREAM_PLANS=1: no plan kept, every batch's built inside the loop
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"fit: {len(hist['epoch'])} epochs of {ntr} triples in {dt:.2f} s (incl. plan building, per-epoch "
      f"evaluation on {n - ntr} test triples)")
base = float(torch.sqrt(torch.mean((y[:ntr].mean() - y[ntr:]) ** 2)))
print("test RMSE of the training-mean predictor:", round(base, 4))
