import torch, sys
sys.path.insert(0, '/root/repo')
from vae_amd.model import VFM
from vae_amd.data import synthetic_triples
X, y = synthetic_triples([300, 200], 2000, seed=3)
def fresh(scaled):
    torch.manual_seed(9)
    m = VFM(300, 200, 12, device="cuda", rng_seed=4)
    m.set_training_data(X, nb_train=2000)
    m.lr = 0.02
    m.scaled_moments = scaled
    return m
a, b, c = fresh(True), fresh(False), fresh(False)
c.fuse_adam = False
pa = [a.plan(X[i:i + 100], y[i:i + 100]) for i in range(0, 2000, 100)]
pb = [b.plan(X[i:i + 100], y[i:i + 100]) for i in range(0, 2000, 100)]
pc = [c.plan(X[i:i + 100], y[i:i + 100]) for i in range(0, 2000, 100)]
def rel(u, v): return float((u - v).abs().max() / v.abs().max())
for s in range(300):
    la, _ = a.train_step(pa[s % 20]); lb, _ = b.train_step(pb[s % 20]); lc, _ = c.train_step(pc[s % 20])
    if s in (0, 1, 4, 20, 100, 127, 128, 129, 200, 299):
        am, av = a._adam_m.clone(), a._adam_v.clone()
        from vae_amd import ops
        if a._moments_scaled:
            ops.moments_rescale(am, av, a._adam_t, to_scaled=False)
        print(s, 'loss', float(la[0]), float(lb[0]), float(lc[0]), 'p', rel(a._flat, b._flat), rel(c._flat, b._flat), 'm', rel(am, b._adam_m), rel(c._adam_m, b._adam_m), 'v', rel(av, b._adam_v), rel(c._adam_v, b._adam_v))
