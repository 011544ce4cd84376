"""Times the per-batch plan build (batch normalisers + inverted index + its one readback) at the cfg3 shape;
run under `rocprofv3 --kernel-trace --stats` to see its kernels."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from vae_amd.model import VFM, sort_rows_within_batches
from vae_amd.data import synthetic_triples
dev = torch.device("cuda")
sizes, d, nb_train = [138493, 26744], 128, 16000210
B = int(os.environ.get("PB", "100000"))
NB = 16
model = VFM(field_sizes=sizes, embedding_size=d, device=dev)
X, y = synthetic_triples(sizes, NB * B, seed=1000, device=dev, zipf=float(os.environ.get("PZIPF", "0")) or None)
occ = torch.clamp(torch.bincount(X.reshape(-1), minlength=sum(sizes)), min=1)
model.set_training_data(X, nb_train=nb_train, nb_occ=occ)
X, y = sort_rows_within_batches(X, y, B)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plans = [model.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B], defer_readback=True) for i in range(NB)]   # as fit() does
    for p in plans:
        p.U                    # (collect the 16-byte readbacks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / NB * 1e3
    print(f"rep {rep}: plan build {dt:.3f} ms per batch (B={B}, T={sum(sizes)}, heavy lists {0 if plans[0].heavy is None else plans[0].heavy[0].numel()})", flush=True)
