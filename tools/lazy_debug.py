"""Step-by-step comparison of a lazy exact Adam form with the dense step (first differing step / rows / moments):
    python tools/lazy_debug.py list|la [B F d]
Rows compared are those the lazy model holds up to date (stamped with the step just taken)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from test_gpu_lazy_adam import _setup
kind = sys.argv[1] if len(sys.argv) > 1 else "list"
B, F, d = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (48, 3, 16)
dense, pd, X = _setup(False, F=F, d=d, B=B)
lazy, pl, _ = _setup(kind == "list", F=F, d=d, B=B)
dense.lookahead = False
lazy.lookahead = kind == "la"
n = len(pl)
for s in range(140):
    lr = 0.05 if s % 7 else 0.02
    ld, _ = dense.train_step(pd[s % n], lr=lr)
    ll, _ = lazy.train_step(pl[s % n], lr=lr, next_plan=pl[(s + 1) % n] if kind == "la" else None)
    cur = (lazy._lazy_last == lazy._adam_t) if lazy._lazy_dirty else torch.ones(lazy.T, dtype=torch.bool, device="cuda")
    ent_d, ent_l = dense.entity_params.weight.detach(), lazy.entity_params.weight.detach()
    bia_d, bia_l = dense.bias_params.weight.detach(), lazy.bias_params.weight.detach()
    bad = (ent_d[cur] != ent_l[cur]).any(1) | (bia_d[cur] != bia_l[cur]).any(1)
    m_bad = (dense._adam_m != lazy._adam_m).sum().item()
    sc_bad = not torch.equal(dense._flat[dense._off_scal:], lazy._flat[lazy._off_scal:])
    if not torch.equal(ld, ll) or bad.any() or m_bad or sc_bad:
        print("step", s, "kind", lazy._lazy_kind, "loss equal", torch.equal(ld, ll), ld.tolist(), ll.tolist(), "current rows differing",
              int(bad.sum()), "of", int(cur.sum()), "moment entries differing", m_bad, "scalars differ", sc_bad, flush=True)
        idx = torch.nonzero(cur)[:, 0][bad][:5]
        now = set(pl[s % n].touched_ids().tolist())
        for i in idx.tolist():
            print("  row", i, "in this batch:", i in now, "\n   dense", ent_d[i].tolist(), bia_d[i].tolist(), "\n   lazy ", ent_l[i].tolist(), bia_l[i].tolist())
        break
else:
    print("no divergence in 140 steps")
