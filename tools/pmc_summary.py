#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel launch
(our kernels only), joined with the kernel trace durations when present.
usage: tools/pmc_summary.py DIR [DIR...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("k_fwd", "k_bwd", "k_adam", "k_finalize", "k_norms", "k_zero", "k_inv_occ", "k_philox"):
        if k in name:
            if k in ("k_fwd", "k_bwd"):
                i = name.index(k)
                return name[i:name.index(">", i) + 1] if ">" in name[i:] else k
            return k
    return None


def main(dirs):
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if k:
                        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print("==", f)
            for k, cs in sorted(acc.items()):
                print(" ", k, " ".join(f"{c}={sum(v) / len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            dur = defaultdict(list)
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if k:
                        dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            print("== durations", f)
            for k, v in sorted(dur.items()):
                print(f"  {k}: n={len(v)} avg={sum(v) / len(v) / 1e3:.2f}us min={min(v) / 1e3:.2f}us")


if __name__ == "__main__":
    main(sys.argv[1:])
