#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (kernel trace and --pmc counter collections): per kernel of
this library, launch count, mean duration and mean counter values per launch.
usage: tools/pmc_summary.py [--json OUT.json] DIR [DIR...]

HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads, so
    hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
(the read side is an upper bound where narrower accesses are mixed in)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import re

# kernels of this library: every `k_*` function of namespace vfm, named with its template arguments
_KERNEL = re.compile(r"vfm::(?:\(anonymous namespace\)::)?(k_[a-z0-9_]+)(<[^>]*>)?")


def short(name):
    m = _KERNEL.search(name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")


def main(argv):
    out_json = None
    if argv and argv[0] == "--json":
        out_json, argv = argv[1], argv[2:]
    summary = defaultdict(dict)
    for d in argv:
        for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            dur = defaultdict(list)
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if k:
                        dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            tag = os.path.basename(os.path.normpath(d))
            for k, v in dur.items():
                summary[k].setdefault("avg_us", {})[tag] = round(sum(v) / len(v) / 1e3, 2)
                summary[k]["launches"] = len(v)
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(lambda: defaultdict(list))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if k:
                        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, cs in acc.items():
                for c, v in cs.items():
                    summary[k].setdefault("counters", {})[c] = sum(v) / len(v)
    for k, s in summary.items():
        c = s.get("counters", {})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            s["hbm_bytes_per_launch"] = int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
            s["fetch_bytes_raw"] = int(c["FETCH_SIZE"] * 1024)
            s["write_bytes"] = int(c["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            s["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    for k in sorted(summary):
        s = summary[k]
        print(k)
        for key in ("launches", "avg_us", "hbm_bytes_per_launch", "fetch_bytes_raw", "write_bytes", "l2_hit_rate"):
            if key in s:
                print(f"    {key}: {s[key]}")
        for c, v in sorted(s.get("counters", {}).items()):
            print(f"    {c}: {v:.6g}")
    if out_json:
        with open(out_json, "w") as fh:
            json.dump(summary, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
