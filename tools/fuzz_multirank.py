#!/usr/bin/env python3
"""Randomised check of the multi-rank steps against the single-process step, several ranks sharing one GPU over
gloo:  [EXCHANGE=dims|sharded|stats|grads] python -m torch.distributed.run --nproc-per-node 2
--master-addr 127.0.0.1 tools/fuzz_multirank.py [n_configs] [seed]      (default: the dimension-sharded step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from vae_amd.model import VFM
from vae_amd.data import synthetic_triples
from vae_amd.dist import shard_rows


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)       # same stream on every rank
    worst = 0.0
    for it in range(n):
        F = int(g.choice([2, 2, 3, 5]))
        dl = int(g.choice([8, 16, 24, 32, 40, 64]))
        d = dl * world
        B = int(g.choice([1, 7, 300, 2500]))
        sizes = [int(g.integers(2, 80)) for _ in range(F)]
        output = str(g.choice(["reg", "class"]))
        link = str(g.choice(["abs", "softplus"]))
        zipf = float(g.choice([0.0, 1.3]))
        seed = int(g.integers(0, 1 << 30))
        X, y = synthetic_triples(sizes, B, seed=seed, output=output, zipf=zipf if zipf > 0 else None)

        def fresh():
            torch.manual_seed(seed)
            m = VFM(field_sizes=sizes, embedding_size=d, output=output, device="cuda:0", rng_seed=seed + 1, link=link)
            m.set_training_data(X, nb_train=max(B, 1) * 5)
            return m

        m = fresh()
        m.exchange = os.environ.get("EXCHANGE", "dims")
        if m.exchange == "dims":
            plan = m.plan(X, y, process_group=dist.group.WORLD)          # every rank: all rows
        else:
            a, b = shard_rows(0, B, rank, world)                         # row blocks
            plan = m.plan(X[a:b], y[a:b], B_global=B, process_group=dist.group.WORLD)
        losses = [m.train_step(plan, lr=0.03, process_group=dist.group.WORLD)[0].clone() for _ in range(2)]
        if m.exchange in ("dims", "sharded"):
            m.sync_params(dist.group.WORLD)
        if rank == 0:
            r = fresh()
            rp = r.plan(X, y)
            ref = [r.train_step(rp, lr=0.03, fused=m.exchange != "grads")[0].clone() for _ in range(2)]
            e = max(rel(torch.stack(losses), torch.stack(ref)), rel(m._flat, r._flat))
            worst = max(worst, e)
            if not e < 1e-4:
                print("MISMATCH", dict(F=F, d=d, B=B, sizes=sizes, output=output, link=link, zipf=zipf), e, flush=True)
        dist.barrier()
    if rank == 0:
        print(os.environ.get("EXCHANGE", "dims"), "world", world, "configs", n, "worst relative error", "%.3g" % worst)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
