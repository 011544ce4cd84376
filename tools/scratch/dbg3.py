import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from vae_amd.model import VFM
from vae_amd.data import synthetic_triples
dev = torch.device("cuda")
sizes, d, B, nb_train = [138493, 26744], 128, 100000, 16000210
nbt = 16
mode = sys.argv[1]
X, y = synthetic_triples(sizes, nbt * B, seed=1000, device=dev)
occ = torch.bincount(X.reshape(-1), minlength=sum(sizes))
occ = torch.clamp((occ.double() * (nb_train / float(nbt * B))).round().long(), min=1)
def run(fused, keep, sync_every, steps=110):
    torch.manual_seed(42)
    model = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=1234)
    model.set_training_data(X, nb_train=nb_train, nb_occ=occ)
    model.lr = 1.0 / (1 + nb_train // B)
    plans = [model.plan(X[i*B:(i+1)*B], y[i*B:(i+1)*B]) for i in range(nbt)]
    losses = torch.zeros(steps, device=dev)
    kept = []
    torch.cuda.synchronize()
    for s in range(steps):
        l3, pred = model.train_step(plans[s % nbt], fused=fused)
        losses[s] = l3[0]
        if keep:
            kept.append((l3, pred))
        if sync_every and s % sync_every == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    nan = torch.isnan(losses)
    first = int(nan.nonzero()[0]) if nan.any() else -1
    print(mode, "fused", fused, "keep", keep, "sync_every", sync_every, "first NaN step", first,
          "last loss", losses[-1].item(), "nan params", torch.isnan(model._flat).sum().item(), flush=True)
    if first >= 0:
        print("  losses around", losses[max(0, first-3):first+2].tolist())
run(True, False, 0)
run(True, True, 0)
run(True, False, 10)
run(False, False, 0)
