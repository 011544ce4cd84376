"""Pipelined step against the dense step, step by step, on one state-machine configuration: is a difference a bug (there at
step 1-2) or the two trajectories drifting apart (growing with the step count)?  usage: python tools/pipe_vs_dense_probe.py [cfg] [steps] [lr]"""
import os
import sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_gpu_state_machine as SM
cfg = sys.argv[1] if len(sys.argv) > 1 else "F2_d128_heavy"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
sizes, d, B, output = SM.CONFIGS[cfg]
nb = 5
ref, plans_r, X = SM._make(sizes, d, B, nb, output)
for form in ("pipe", "pipe_la"):
    tst, plans_t, _ = SM._make(sizes, d, B, nb, output)
    ref2, plans_r2, _ = SM._make(sizes, d, B, nb, output)
    tst.pipeline = True
    tst.lookahead = form == "pipe_la"
    ref2.pipeline = False
    ref2.lookahead = False
    print(f"== {cfg} {form}: step, rel loss diff, max |param diff| / max |param|, alpha m diff")
    for s in range(steps):
        cur, nxt = s % nb, (s + 1) % nb
        lt, pt = tst.train_step(plans_t[cur], lr=lr, next_plan=plans_t[nxt])
        lr_, pr = ref2.train_step(plans_r2[cur], lr=lr)
        tst.sync_lazy(); ref2.sync_lazy()
        dl = float((lt[0] - lr_[0]).abs() / lr_[0].abs())
        dp = float((tst._flat - ref2._flat).abs().max() / ref2._flat.abs().max())
        dpred = float((pt - pr).abs().max() / pr.abs().max())
        if s < 6 or s % 5 == 4:
            print(f"  {s:3d}  loss {dl:.2e}  pred {dpred:.2e}  params {dp:.2e}", flush=True)
