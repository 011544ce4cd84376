import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from vae_amd.model import VFM
from vae_amd import ops
from vae_amd.data import synthetic_triples
dev = torch.device("cuda")
sizes, d, B, nb_train = [138493, 26744], 128, 100000, 16000210
nbt = 16
torch.manual_seed(42)
model = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=1234)
X, y = synthetic_triples(sizes, nbt * B, seed=1000, device=dev)
occ = torch.bincount(X.reshape(-1), minlength=sum(sizes))
occ = torch.clamp((occ.double() * (nb_train / float(nbt * B))).round().long(), min=1)
model.set_training_data(X, nb_train=nb_train, nb_occ=occ)
model.lr = 1.0 / (1 + nb_train // B)
plans = [model.plan(X[i*B:(i+1)*B], y[i*B:(i+1)*B]) for i in range(nbt)]
seq = list(range(10)) + [s % nbt for s in range(30)]
losses = torch.zeros(len(seq), device=dev)
snaps = []
torch.cuda.synchronize()
for i, pi in enumerate(seq):
    if i >= 16:
        snaps.append((i, pi, model._flat.clone(), model._adam_m.clone(), model._adam_v.clone(), model.global_step, model._adam_t))
    l3, pred = model.train_step(plans[pi], fused=False)
    losses[i] = l3[0]
torch.cuda.synchronize()
nan = torch.isnan(losses)
first = int(nan.nonzero()[0]) if nan.any() else -1
print("first NaN loss idx", first, losses[max(first-2,0):first+2].tolist())
for (i, pi, flat, m, v, gs, at) in snaps:
    pn = torch.isnan(flat).sum().item()
    print("step", i, "plan", pi, "params nan before step:", pn, "m nan", torch.isnan(m).sum().item(), "v nan", torch.isnan(v).sum().item(), "v inf", torch.isinf(v).sum().item(), "m inf", torch.isinf(m).sum().item())
    if pn:
        ent, bia, scal = model._views(flat)
        rows = torch.isnan(ent).any(1).nonzero().reshape(-1)
        print("  nan rows", rows.numel(), rows[:8].tolist(), "bias nan", torch.isnan(bia).sum().item(), "scal", scal.tolist())
        break
    prev = (i, pi, flat, m, v, gs, at)
# replay the step that produced the first NaN params from prev snapshot
(i, pi, flat, m, v, gs, at) = prev
ent, bia, scal = model._views(flat)
st = ops.elbo_forward(plans[pi], ent, bia, scal, model.inv_occ, seed=model.rng_seed, step=gs)
l = ops.elbo_finalize(st, scal)
g = ops.elbo_backward(plans[pi], st, ent, bia, scal, model.inv_occ, torch.ones(1, device=dev))
torch.cuda.synchronize()
print("replay step", i, "loss", l.tolist(), "partials", st.partials.tolist())
print(" grad nan", [torch.isnan(t).sum().item() for t in g], "inf", [torch.isinf(t).sum().item() for t in g], "absmax", [t.abs().max().item() for t in g])
print(" min|s|", ent[:, d:].abs().min().item(), "pred absmax", st.pred.abs().max().item(), "grow absmax", st.grow.abs().max().item())
