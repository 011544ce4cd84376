#!/usr/bin/env python3
"""Assemble the judged evidence under profiles/ from the scratch output of a GPU run (gpurun_out/):
    tools/collect_profiles.py r02 [bench_line.json]
reads gpurun_out/prof_<tag>/ (tools/profile_round.sh <tag>: rocprofv3 kernel trace + the separate PMC passes of the
default bench command) and writes profiles/<tag>_kernel_stats.csv (our kernels' rows of the --stats summary),
profiles/<tag>_pmc_summary.{txt,json}, profiles/latest_traffic.json (HBM bytes per launch of the step's kernels:
what bench.py reports as roofline.traffic) and, if given, profiles/<tag>_bench.json (the bench line)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
if os.path.exists(os.path.join(src, "summary.json")) and not os.path.isdir(os.path.join(src, "trace")):
    # summarised on the GPU box by tools/profile_round.sh (tools/pmc_summary.py there); the per-dispatch CSVs stayed behind
    rows = list(csv.reader(open(os.path.join(src, "kernel_stats.csv"))))
    shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{tag}_pmc_summary.json"))
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, f"{tag}_pmc_summary.txt"))
else:
    # raw passes present (KEEP_RAW=1).  gpurun merges every call's output into gpurun_out/: keep only the newest process' files
    for sub in ("trace", "pmc_fetch", "pmc_write", "pmc_sq"):
        files = glob.glob(os.path.join(src, sub, "**", "*_agent_info.csv"), recursive=True)
        if len(files) > 1:
            newest = max(files, key=os.path.getmtime)
            keep = os.path.basename(newest).split("_")[0]
            for f in glob.glob(os.path.join(os.path.dirname(newest), "*.csv")):
                if not os.path.basename(f).startswith(keep + "_"):
                    os.remove(f)
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.reader(open(stats)))
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "--json",
                          os.path.join(dst, f"{tag}_pmc_summary.json"), os.path.join(src, "trace"), os.path.join(src, "pmc_fetch"),
                          os.path.join(src, "pmc_write"), os.path.join(src, "pmc_sq")], capture_output=True, text=True, check=True).stdout
    open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w").write(txt)
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0])
    for r in rows[1:]:
        if "vfm::" in r[0]:
            w.writerow(r)
summ = json.load(open(os.path.join(dst, f"{tag}_pmc_summary.json")))
traffic = {}
for k, v in summ.items():
    name = "fwd" if k.startswith("k_fwd") else ("bwd_adam" if k.startswith("k_bwd") else None)
    if name and "hbm_bytes_per_launch" in v and v.get("launches", 0) >= 50:
        traffic[name] = {"kernel": k, "hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "fetch_size_bytes_raw": v["fetch_bytes_raw"],
                         "write_size_bytes": v["write_bytes"], "l2_hit_rate": v.get("l2_hit_rate"),
                         "avg_us_under_rocprof": v["avg_us"].get("trace")}
traffic["_source"] = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum passes of `python bench.py --steps 100 "
                      f"--warmup 10 --no-cpu-baseline --no-events` (tools/profile_round.sh {tag}); hbm_bytes = 2*FETCH_SIZE*1024 + "
                      "WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md; Infinity-Cache hits are "
                      "included in the count)")
sys.path.insert(0, ROOT)
from vae_amd.build import sources_digest        # noqa: E402  (hash of csrc/: bench.py drops the figures when the kernels changed)
stamp = os.path.join(src, "csrc_sha1.txt")      # written on the GPU box by tools/profile_round.sh: the sources the passes ran on
traffic["_csrc_sha1"] = open(stamp).read().strip() if os.path.exists(stamp) else sources_digest()
if traffic["_csrc_sha1"] != sources_digest():
    print("WARNING: the profiled sources are not the current ones (bench.py will not attach these figures)")
# only the default bench command's passes feed bench.py's roofline.traffic (tags like r04cfg2 / r04b1m are other workloads:
# their summaries are written as profiles/<round>_<point>_pmc_summary.* and nothing else)
point = tag[3:] if len(tag) > 3 and tag[0] == "r" and tag[1:3].isdigit() else ""
if not point:
    json.dump(traffic, open(os.path.join(dst, "latest_traffic.json"), "w"), indent=1)
else:
    for ext in ("_kernel_stats.csv", "_pmc_summary.txt", "_pmc_summary.json"):
        os.replace(os.path.join(dst, tag + ext), os.path.join(dst, f"{tag[:3]}_{point}{ext}"))
    tag = f"{tag[:3]}_{point}"
line = os.path.join(src, "trace.json")             # the bench line of the kernel-trace pass itself (its alg_bytes belong to these kernels)
if os.path.exists(line) and os.path.getsize(line) > 0:
    shutil.copy(line, os.path.join(dst, f"{tag}_profiled_line.json"))
if len(sys.argv) > 2:
    shutil.copy(sys.argv[2], os.path.join(dst, f"{tag}_bench.json"))
print("wrote", sorted(f for f in os.listdir(dst) if f.startswith(tag) or f == "latest_traffic.json"))
