import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from test_gpu_lazy_adam import _setup
dense, pd, X = _setup(False)
lazy, pl, _ = _setup(True)
for s in range(140):
    lr = 0.05 if s % 7 else 0.02
    ld, _ = dense.train_step(pd[s % 12], lr=lr)
    ll, _ = lazy.train_step(pl[s % 12], lr=lr)
    # compare the rows that are up to date in the lazy model: those stamped with the current step
    cur = (lazy._lazy_last == lazy._adam_t)
    ent_d, ent_l = dense.entity_params.weight.detach(), lazy.entity_params.weight.detach()
    bad_rows = ((ent_d[cur] != ent_l[cur]).any(1)).sum().item()
    m_bad = (dense._adam_m != lazy._adam_m).sum().item()
    if not torch.equal(ld, ll) or bad_rows or m_bad:
        print("step", s, "loss equal", torch.equal(ld, ll), ld.tolist(), ll.tolist(), "current rows differing", bad_rows, "of", int(cur.sum()), "moment entries differing", m_bad, flush=True)
        if bad_rows:
            idx = torch.nonzero(cur)[:, 0][(ent_d[cur] != ent_l[cur]).any(1)][:5]
            print("  rows", idx.tolist(), "touched now?", [int(i) in set(pl[s % 12].touched_ids().tolist()) for i in idx.tolist()])
        break
else:
    print("no divergence in 140 steps")
d = torch.nonzero(dense._adam_m != lazy._adam_m)[:, 0]
print("differing m entries", d.tolist()[:40], "n_ent", lazy._n_ent, "off_bias", lazy._off_bias, "off_scal", lazy._off_scal)
print("dense", dense._adam_m[d][:10].tolist(), "lazy", lazy._adam_m[d][:10].tolist())
t = pl[0].touched_ids()
print("touched", t.tolist()[:20], "n", t.numel())
