#!/usr/bin/env python3
"""VERDICT r3 item 9, costed before any code: which part of a batch's rows has BOTH its entities inside one workgroup of
the fused backward?  (Only for those rows could the backward of step t run step t+1's forward from registers / LDS; the
others still need the forward launch.)  Pure numpy on the bench's own synthetic batches (bench.py: uniform ids, ML-20M
shape, B = 100,000 rows per step):

  contiguous  the backward as built: a workgroup owns a CONTIGUOUS range of table rows (entity ids), so a row is local
              when |user id - item id| fits in one range.  vfm-torch.py:77-80 offsets the item ids by N: the two ids of
              a row live in disjoint parts of the table.
  ideal       an upper bound for ANY per-batch assignment of entities to workgroups (a data-dependent partition rebuilt
              with every plan, table rows no longer contiguous per workgroup): an entity has one owner, so of the k rows
              of a user at most one per distinct owner... at most the rows whose user sits with THAT row's item; a user
              with k rows on k different items has 1 local row at best.  bound = (#distinct users in the batch) / B when
              items may hold any number of users; tightened by a cap on the entities a workgroup can hold in LDS.
usage: tools/colocation_estimate.py [--batches 8]"""
import argparse
import json

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--users", type=int, default=138493)
    ap.add_argument("--items", type=int, default=26744)
    ap.add_argument("--batch", type=int, default=100000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--workgroups", type=int, default=1024, help="persistent grid of the fused backward (4 per CU)")
    ap.add_argument("--lds-bytes", type=int, default=160 * 1024 // 4, help="LDS per workgroup at 4 workgroups per CU")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    N, M, B, T = a.users, a.items, a.batch, a.users + a.items
    span = -(-T // a.workgroups)
    rec = 4 * (a.d + 4)                       # one sample record (w, weighted KL, -, - | z[d]) in bytes
    cap = a.lds_bytes // rec                  # records a workgroup can keep in LDS
    out = {"shape": {"users": N, "items": M, "B": B, "d": a.d}, "workgroups": a.workgroups, "table_rows_per_workgroup": span,
           "records_per_workgroup_in_LDS": cap, "per_batch": []}
    for _ in range(a.batches):
        u = rng.integers(0, N, B)
        i = N + rng.integers(0, M, B)
        local_contig = np.mean(u // span == i // span)
        users, first = np.unique(u, return_index=True)
        ideal = len(users) / B
        # tightened: a workgroup holds the item + at most cap-1 of its first-row users (LDS); rows beyond that are remote
        it_first = i[first]
        _, cnt = np.unique(it_first, return_counts=True)
        capped = np.minimum(cnt, cap - 1).sum() / B
        out["per_batch"].append({"contiguous": float(local_contig), "ideal": float(ideal), "ideal_lds_capped": float(capped)})
    for k in ("contiguous", "ideal", "ideal_lds_capped"):
        out[k + "_mean"] = float(np.mean([p[k] for p in out["per_batch"]]))
    del out["per_batch"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
