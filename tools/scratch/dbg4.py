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
seq = list(range(10)) + [s % nbt for s in range(100)]
losses = torch.zeros(len(seq), device=dev)
snap = None
for i, pi in enumerate(seq):
    if i >= 15:
        snap = (model._flat.clone(), model._adam_m.clone(), model._adam_v.clone(), model.global_step, model._adam_t)
    l3, pred = model.train_step(plans[pi], fused=True)
    losses[i] = l3[0]
    if i >= 15:
        torch.cuda.synchronize()
        if torch.isnan(model._flat).any() or torch.isnan(l3).any():
            print("first NaN at i", i, "plan", pi, "loss", l3.tolist(), "nan params", torch.isnan(model._flat).sum().item())
            flat, m, v, gs, at = snap
            print("prev params nan", torch.isnan(flat).sum().item(), "m nan", torch.isnan(m).sum().item(), torch.isinf(m).sum().item(), "v nan", torch.isnan(v).sum().item(), torch.isinf(v).sum().item())
            ent, bia, scal = model._views(flat)
            print("scal", scal.tolist(), "min|s|", ent[:, d:].abs().min().item(), "min|sw|", bia[:, 1].abs().min().item())
            st = ops.elbo_forward(plans[pi], ent, bia, scal, model.inv_occ, seed=model.rng_seed, step=gs)
            print("partials", st.partials.tolist(), "pred nan", torch.isnan(st.pred).sum().item(), "sumz nan", torch.isnan(st.sumz).sum().item(), "grow nan", torch.isnan(st.grow).sum().item())
            g = ops.elbo_backward(plans[pi], st, ent, bia, scal, model.inv_occ, torch.ones(1, device=dev))
            print("grad nan", [torch.isnan(t).sum().item() for t in g], "inf", [torch.isinf(t).sum().item() for t in g], "max", [t.abs().max().item() for t in g])
            ge = g[0]
            bad = torch.isnan(ge) | torch.isinf(ge)
            rows = bad.any(1).nonzero().reshape(-1)
            print("bad rows", rows[:10].tolist(), "n", rows.numel())
            if rows.numel():
                e = int(rows[0]); print("entity", e, "params s", ent[e, d:][bad[e, d:]][:5].tolist() if bad[e, d:].any() else None, "mu", ent[e,:4].tolist(), "s", ent[e, d:d+4].tolist(), "m", m[:ent.numel()].view_as(ent)[e, d:d+4].tolist(), "v", v[:ent.numel()].view_as(ent)[e, d:d+4].tolist())
                cols = bad[e].nonzero().reshape(-1); print("cols", cols[:8].tolist(), "vals", ent[e][cols[:8]].tolist(), "grad", ge[e][cols[:8]].tolist())
            break
torch.cuda.synchronize()
nan = torch.isnan(losses)
print("first NaN loss idx", int(nan.nonzero()[0]) if nan.any() else -1, losses[95:].tolist())
