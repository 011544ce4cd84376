import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from vae_amd.model import VFM
from vae_amd.data import synthetic_triples
dev = torch.device("cuda")
sizes, d, B, nb_train = [138493, 26744], 128, 100000, 16000210
nbt = 16
torch.manual_seed(42)
model = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=1234)
X, y = synthetic_triples(sizes, nbt * B, seed=1000, device=dev)
occ = torch.bincount(X.reshape(-1), minlength=model.T)
scale = nb_train / float(nbt * B)
occ = torch.clamp((occ.double() * scale).round().long(), min=1)
model.set_training_data(X, nb_train=nb_train, nb_occ=occ)
model.lr = 1.0 / (1 + nb_train // B)
plans = [model.plan(X[i*B:(i+1)*B], y[i*B:(i+1)*B]) for i in range(nbt)]
from vae_amd import ops
for s in range(130):
    prev = model._flat.clone()
    l3, pred = model.train_step(plans[s % nbt], fused=(len(sys.argv) > 1))
    torch.cuda.synchronize()
    bad = torch.isnan(l3).any().item()
    if s % 10 == 0 or bad:
        print(s, l3.tolist(), "pred nan", torch.isnan(pred).sum().item(), "param nan", torch.isnan(model._flat).sum().item(),
              "min|s|", model.entity_params.weight[:, d:].abs().min().item())
    if bad:
        ent, bia, scal = model._views(prev)
        st = ops.elbo_forward(plans[s % nbt], ent, bia, scal, model.inv_occ, seed=model.rng_seed, step=model.global_step - 1)
        print("partials", st.partials.tolist())
        print("scal", scal.tolist(), "prev nan", torch.isnan(prev).sum().item())
        print("pred nan", torch.isnan(st.pred).sum().item(), "sumz nan", torch.isnan(st.sumz).sum().item(), "inf", torch.isinf(st.sumz).sum().item())
        sv = ent[:, d:].abs()
        print("min |s|", sv.min().item(), (sv == 0).sum().item(), "bias min|s|", bia[:,1].abs().min().item())
        break
