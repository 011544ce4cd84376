#!/usr/bin/env python3
"""Times the general ELBO-variant kernels (csrc/vfm_variants.hip: vfm_variant_fwd_f32 / vfm_variant_bwd_f32) at the bench
shape (ML-20M ids, d = 128, B = 100,000) for the three objectives SURVEY 8(f)4 lists -- closed-form expected
log-likelihood with learnable group priors (vfm-tomasrch.py), sampled ELBO with feature values (vfm.py) -- next to the
fused two-field kernels of the main path (forward + unfused backward, same batch).  One JSON line.
    python tools/variant_probe.py [d] [B]"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from vae_amd import ops  # noqa: E402
from vae_amd.data import synthetic_triples  # noqa: E402
from vae_amd.model import VFM  # noqa: E402
from vae_amd.variants import priors_len, variant_backward, variant_forward  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
sizes = [138493, 26744]
dev = torch.device("cuda", 0)
X, y = synthetic_triples(sizes, B * 4, seed=3)
m = VFM(field_sizes=sizes, embedding_size=d, device="cuda", rng_seed=1)
m.set_training_data(X, nb_train=16000210)
Xd, yd = X.to(dev), y.to(dev)
Xs = Xd[:B]
order = torch.argsort(Xs[:, 1], stable=True)           # rows ordered by item id, as VFM.fit does
plan = m.plan(Xs[order], yd[:B][order])
ent, bia, scal = m._views(m._flat)
gout = torch.ones(1, device=dev)
priors = torch.zeros(priors_len(2, d), device=dev)
priors[1] = 1.0
priors[2 + 2:2 + 4] = 1.0
priors[2 + 4 + 2 * d:] = 1.0
values = torch.rand(B, 2, device=dev) + 0.5


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 1)


out = {"shape": {"T": sum(sizes), "d": d, "B": B}, "us": {}}
for name, kw in (("closed_form+priors", dict(objective="closed_form", priors=priors)),
                 ("sampled+values", dict(objective="sampled", values=values)),
                 ("sampled (general kernels)", dict(objective="sampled"))):
    obj = kw.pop("objective")
    st = variant_forward(plan, obj, ent, bia, scal, m.inv_occ, seed=1, step=1, **kw)
    out["us"][name] = {"fwd": timed(lambda: variant_forward(plan, obj, ent, bia, scal, m.inv_occ, seed=1, step=1, **kw)),
                       "bwd": timed(lambda: variant_backward(plan, st, ent, bia, scal, m.inv_occ, gout))}
st = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, seed=1, step=1, train=True)
g = [torch.empty_like(t) for t in (ent, bia, scal)]


def main_bwd():
    ops.elbo_finalize(st, scal)
    ops.elbo_backward(plan, st, ent, bia, scal, m.inv_occ, gout, g_entity=g[0], g_bias=g[1], g_scalars=g[2])


out["us"]["main path (k_fwd2 / k_finalize + k_bwd, dense gradient written)"] = {
    "fwd": timed(lambda: ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, seed=1, step=1, train=True)),
    "bwd": timed(main_bwd)}
print(json.dumps(out))
