#!/usr/bin/env python3
"""Timeline of the streamed loop from a rocprofv3 kernel trace (csv): per step of the main stream (k_fwd* .. k_bwd*) the
kernels' durations, the idle gaps between them, and how long the index-build kernels of the side stream ran under them.
usage: tools/stream_timeline.py <dir with *_kernel_trace.csv> [first_step last_step]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
K = re.compile(r"vfm::(?:\(anonymous namespace\)::)?(k_[a-z0-9_]+)")
rows = []
for r in csv.DictReader(open(f)):
    m = K.search(r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else "other:" + r["Kernel_Name"][:40], r.get("Queue_Id"), r.get("Stream_Id")))
rows.sort()
main = [r for r in rows if r[2].startswith("k_fwd") or r[2].startswith("k_bwd")]
# steps = (fwd, bwd) pairs
steps = []
i = 0
while i + 1 < len(main):
    if main[i][2].startswith("k_fwd") and main[i + 1][2].startswith("k_bwd"):
        steps.append((main[i], main[i + 1]))
        i += 2
    else:
        i += 1
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) // 2
hi = int(sys.argv[3]) if len(sys.argv) > 3 else min(len(steps) - 1, lo + 200)
side = [r for r in rows if not (r[2].startswith("k_fwd") or r[2].startswith("k_bwd"))]
acc = defaultdict(float)
n = 0
for s in range(lo, hi):
    (f0, f1, fn, *_), (b0, b1, bn, *_) = steps[s]
    nf0 = steps[s + 1][0][0]
    acc["fwd_us"] += (f1 - f0) / 1e3
    acc["gap_fwd_to_bwd_us"] += (b0 - f1) / 1e3
    acc["bwd_us"] += (b1 - b0) / 1e3
    acc["gap_bwd_to_next_fwd_us"] += (nf0 - b1) / 1e3
    acc["period_us"] += (nf0 - f0) / 1e3
    for (s0, s1, sn, *_) in side:
        if s1 <= f0 or s0 >= nf0:
            continue
        acc["side_busy_us"] += (min(s1, nf0) - max(s0, f0)) / 1e3
        acc["side:" + sn] += (min(s1, nf0) - max(s0, f0)) / 1e3
    n += 1
print(f"{n} steps ({lo}..{hi}) of {len(steps)}; kernel names: {steps[lo][0][2]}, {steps[lo][1][2]}")
for k, v in acc.items():
    print(f"  {k:34s} {v / n:8.2f}")
