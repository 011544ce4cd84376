import sys, os, subprocess, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from vae_amd.model import VFM
from vae_amd import ops
from vae_amd.data import synthetic_triples
dev = torch.device("cuda")
sizes, d, nb_train = [138493, 26744], 128, 16000210
B = int(os.environ.get("PB", "100000"))
torch.manual_seed(42)
model = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=1234)
X, y = synthetic_triples(sizes, 4 * B, seed=1000, device=dev)
occ = torch.clamp(torch.bincount(X.reshape(-1), minlength=sum(sizes)), min=1)
model.set_training_data(X, nb_train=nb_train, nb_occ=occ)
if os.environ.get("PSORT", "1") == "1":       # rows of each batch ordered by item id, as VFM.fit / bench.py do
    from vae_amd.model import sort_rows_within_batches
    X, y = sort_rows_within_batches(X, y, B)
plans = [model.plan(X[i*B:(i+1)*B], y[i*B:(i+1)*B]) for i in range(4)]
pplans = [model.plan(X[i*B:(i+1)*B], None) for i in range(4)]
ent, bia, scal = model._views(model._flat)
ee, eb, eg = ops.philox_eps(model.spec(), 1, 1, dev)
def timeit(fn, n=30):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
sumz = torch.empty(B, d, device=dev); grow = torch.empty(B, device=dev); pred = torch.empty(B, device=dev); part = torch.zeros(8*4097, dtype=torch.float64, device=dev)
res = {}
res["predict_zero"] = timeit(lambda i: ops.elbo_forward(pplans[i % 4], ent, bia, scal, None, train=False, flags=ops.FLAG_EPS_ZERO, out_pred=pred, out_partials=part))
res["predict_philox"] = timeit(lambda i: ops.elbo_forward(pplans[i % 4], ent, bia, scal, None, train=False, seed=1, step=i, out_pred=pred, out_partials=part))
res["train_philox"] = timeit(lambda i: ops.elbo_forward(plans[i % 4], ent, bia, scal, model.inv_occ, seed=1, step=i, out_pred=pred, out_partials=part, out_sumz=sumz, out_grow=grow))
res["train_table"] = timeit(lambda i: ops.elbo_forward(plans[i % 4], ent, bia, scal, model.inv_occ, eps=(ee, eb, eg), out_pred=pred, out_partials=part, out_sumz=sumz, out_grow=grow))
print("kernel", os.environ.get("VFM_FWD_KERNEL", "2"), "norng", os.environ.get("VFM_FWD_AB_NORNG", "0"), "sorted", os.environ.get("PSORT", "1"), "B", B, {k: round(v, 1) for k, v in res.items()}, flush=True)
