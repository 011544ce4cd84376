import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, numpy as np
from vae_amd.model import VFM
from vae_amd import ops
from vae_amd.data import synthetic_triples
torch.manual_seed(0)
for d in (8, 5, 128):
    X, y = synthetic_triples([200, 300], 6000, seed=3)
    m = VFM(200, 300, d, device="cuda")
    m.set_training_data(X, nb_train=6000)
    plan = m.plan(X[:2000], y[:2000])
    ent, bia, scal = m._views(m._flat)
    spec = m.spec()
    ee, eb, eg = ops.philox_eps(spec, seed=99, step=5, device="cuda")
    print("d", d, "eps stats", ee.mean().item(), ee.std().item(), eb.mean().item(), eb.std().item(), eg.item(), torch.isnan(ee).any().item(), torch.isnan(eb).any().item())
    a = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, eps=None, seed=99, step=5)
    b = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, eps=(ee, eb, eg))
    print(" partials philox", a.partials.tolist())
    print(" partials table ", b.partials.tolist())
    print(" pred diff", (a.pred - b.pred).abs().max().item(), "sumz diff", (a.sumz - b.sumz).abs().max().item())
    for fused in (False, True):
        torch.manual_seed(0)
        m = VFM(200, 300, d, device="cuda")
        m.set_training_data(X, nb_train=6000)
        m.lr = 0.25
        plans = [m.plan(X[i:i+2000], y[i:i+2000]) for i in (0, 2000, 4000)]
        ls = []
        for ep in range(4):
            for p in plans:
                l3, _ = m.train_step(p, fused=fused)
                ls.append(round(l3[0].item(), 1))
        print(" fused", fused, ls)
