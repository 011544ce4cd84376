#!/usr/bin/env python3
"""Kernel times of the entity-sharded step at an N-rank shape, on ONE GPU without communication:
what one rank computes per step at cfg3 when N ranks each hold B rows (collectives excluded)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import _lib, ops
from vae_amd.model import VFM
from vae_amd.data import synthetic_triples
from vae_amd.sharded import owned_rows

N = int(os.environ.get("N", "8"))
dev = torch.device("cuda")
sizes, d, B, nb_train = [138493, 26744], 128, 100000, 16000210
T = sum(sizes)
torch.manual_seed(42)
m = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=3)
X, y = synthetic_triples(sizes, B, seed=1, device=dev)
o = torch.argsort(X[:, 1], stable=True)
X, y = X[o].contiguous(), y[o].contiguous()
occ = torch.clamp(torch.bincount(X.reshape(-1), minlength=T) * 160, min=1)
m.set_training_data(X, nb_train=nb_train, nb_occ=occ)
m._ensure_opt_state()
ent, bia, scal = m._views(m._flat)
rl = ops.exchange_record_len(d)
# my rows' slots (as ShardedPlan does, without the exchanges)
uniq = torch.unique(X)
key = (uniq % N) * T + uniq
order = torch.argsort(key)
slot_ids = uniq[order].contiguous()
U = slot_ids.numel()
x_slots = torch.searchsorted(key[order].contiguous(), (X % N) * T + X).to(torch.int32).contiguous()
# what an owner is asked for: with N similar ranks, ~N * U / N records of owned entities
own = torch.arange(0, T, N, device=dev, dtype=torch.int32)            # rank 0's entities
req_ids = own[torch.randint(0, own.numel(), (U,), device=dev)].contiguous()
req_local = (req_ids // N).to(torch.int32).contiguous()
sspec = ops.Spec(T=U, F=2, d=d, group_hi=(U, U), group_n=(1.0, 1.0), likelihood=_lib.LIK_NORMAL, nb_train=nb_train)
base = ops.BatchPlan(m.spec(), X, y, m.inv_occ, build_index=False, validate=False)
splan = ops.BatchPlan(sspec, x_slots, y, None, validate=False)
splan.W = base.W
zsend = torch.empty(U * rl, device=dev); srec = torch.zeros(U * rl, device=dev)
acc = torch.zeros(owned_rows(T, 0, N) * rl, device=dev)
sumz = torch.empty(B, d, device=dev); grow = torch.empty(B, device=dev); pred = torch.empty(B, device=dev)
part = torch.zeros(_lib.PARTIALS_LEN, dtype=torch.float64, device=dev)
small = torch.zeros(16, device=dev); klws = torch.zeros(4097, dtype=torch.float64, device=dev)
o_ = _lib.ops()
GATHER = os.environ.get("GATHER", "1") == "1"
t_own = owned_rows(T, 0, N)
rec_pos = torch.argsort(req_local.to(torch.int64), stable=True).to(torch.int32).contiguous()
rp = torch.zeros(t_own + 1, dtype=torch.int64, device=dev)
torch.cumsum(torch.bincount(req_local.to(torch.int64), minlength=t_own), 0, out=rp[1:])
rec_ptr = rp.to(torch.int32).contiguous()


class P:
    spec, W = m.spec(), base.W


def step(i):
    t = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    ev[0].record()
    o_.shard_sample(req_ids, ent, bia, None, None, zsend, 3, i)
    ev[1].record()
    o_.elbo_fwd_zpre(x_slots, y, zsend, scal, None, pred, part, sumz, grow, d, nb_train, B * N, 0, 0, 3, i)
    st = ops.FwdState(pred, part, sumz, grow, ops._problem(sspec, B, B * N, 32, 3, i, 0), None)
    ops.elbo_finalize(st, scal, out=small[8:11])
    ev[2].record()
    ops.elbo_backward_acc(splan, st, srec, small[0:2])
    ev[3].record()
    if GATHER:
        ev[4].record()
        st2 = ops.FwdState(pred, part, sumz, grow, ops._problem(m.spec(), 0, B * N, 64, 3, i, 0), None)
        ops.elbo_apply_adam(P, st2, srec, small[0:2], ent, bia, scal, m.inv_occ, m._views(m._adam_m),
                            m._views(m._adam_v), 0.006, i + 1, e_lo=0, e_hi=owned_rows(T, 0, N), own_mod=N,
                            own_rank=0, kl_ws=klws, rec_index=(rec_ptr, rec_pos))
    else:
        acc.zero_()
        o_.records_add(acc, req_local, srec, d, True)
        ev[4].record()
        st2 = ops.FwdState(pred, part, sumz, grow, ops._problem(m.spec(), 0, B * N, 64, 3, i, 0), None)
        ops.elbo_apply_adam(P, st2, acc, small[0:2], ent, bia, scal, m.inv_occ, m._views(m._adam_m),
                            m._views(m._adam_v), 0.006, i + 1, e_lo=0, e_hi=owned_rows(T, 0, N), own_mod=N,
                            own_rank=0, kl_ws=klws)
    ev[5].record()
    return ev


for i in range(5):
    step(i)
torch.cuda.synchronize()
tot = [0.0] * 5
n = 30
for i in range(n):
    ev = step(10 + i)
    torch.cuda.synchronize()
    for k in range(5):
        tot[k] += ev[k].elapsed_time(ev[k + 1])
names = ["sample", "fwd_zpre+finalize", "bwd_acc(slots)", "zero+records_add", "apply_adam(owned)"]
print("N =", N, "U =", U, {k: round(v / n * 1e3, 1) for k, v in zip(names, tot)}, "us; sum",
      round(sum(tot) / n * 1e3, 1), "us; records", round(U * rl * 4 / 1e6, 1), "MB per all-to-all")
