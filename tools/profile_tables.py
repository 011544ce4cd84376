#!/usr/bin/env python3
"""Markdown tables for profiles/README.md, generated from the committed JSON / CSV files of a round (no number in those
tables is typed by hand):   tools/profile_tables.py r03 > /tmp/tables.md     (or --write to splice them into
profiles/README.md between the `<!-- tables:r03 -->` ... `<!-- /tables:r03 -->` markers)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load_lines(path):
    out = []
    for l in open(path):
        if l.startswith("{"):
            out.append(json.loads(l))
    return out


def bench_table(tag):
    j = load_lines(os.path.join(P, f"{tag}_bench.json"))[0]
    r, k = j["roofline"], j["kernels"]
    rows = ["| figure | value |", "|---|---|",
            f"| `value` (commanded region: {j['steps']} steps, events on every {j.get('kernel_events_on_every_nth_step', 1)}th) | **{j['value'] / 1e6:.1f} M triples/s**, {j['ms_per_step']} ms/step |"]
    if j.get("settle"):
        rows.append(f"| `settle` ({j['settle']['untimed_steps_before_warmup']} untimed steps before the warm-up; the same warm-up + {j['settle']['cold_steps']} steps timed first, without them) | cold: {j['settle']['cold_ms_per_step']} ms/step |")
    if j.get("sustained"):
        rows.append(f"| `sustained` ({j['sustained']['steps']} eager steps, no events) | {j['sustained']['triples_per_s'] / 1e6:.1f} M triples/s, {j['sustained']['ms_per_step']} ms/step |")
    if j.get("streamed"):
        t = j["streamed"]
        rows.append(f"| `streamed` ({t['steps']} steps, every plan built inside the timed region, {t.get('prefetch_depth', 3)} in hand) | {t['triples_per_s'] / 1e6:.1f} M triples/s, **{t['ms_per_step']} ms/step** (host {t['host_enqueue_ms_per_step']} ms/step; one build alone {t.get('plan_build_gpu_us_alone')} µs) |")
    for name in ("fwd", "bwd_adam"):
        if name in k:
            v = k[name]
            t = v.get("hbm_bytes_pmc")
            rows.append(f"| `{name}`: {v['kernel'][:60]}... | {v['avg_us']} µs (HIP events), {v['alg_bytes'] / 1e6:.1f} MB algorithmic → {v['achieved_GBs']:.0f} GB/s = **{v['frac_hbm_peak']:.3f}** of 8 TB/s"
                        + (f"; {t / 1e6:.1f} MB at the fabric (PMC) → {t / v['avg_us'] / 1e3:.0f} GB/s" if t else "") + " |")
    rows.append(f"| `roofline.frac` (dominant kernel) / `frac_fwd_8d` / `frac_K_8d` (SURVEY 8(d)'s own bytes; target {r.get('target_8d')}) | {r['frac']} / {r.get('frac_fwd_8d')} / {r.get('frac_K_8d')} |")
    rows.append(f"| box stream copy | {r['box_stream_copy_GBs']} GB/s (the dominant kernel runs at {r['frac_of_box_stream_copy']} of it on algorithmic bytes) |")
    if j.get("regions"):
        g = j["regions"]
        rows.append(f"| regions F / K / S | {g['F_us']} / {g['K_us']} / {g['S_us']} µs |")
    if j.get("cpu_baseline"):
        c = j["cpu_baseline"]
        rows.append(f"| `cpu_baseline` (kind {c['kind']}, {c['cores']} threads; dead gathers included) | {c['value'] / 1e3:.1f} K triples/s ({c['ms_per_step']} ms/step); sweep {c['thread_sweep_triples_per_s']} |")
    pb = j.get("plan_build_ms_per_batch", {})
    rows.append(f"| plan build per batch (first / rebuilt) | {pb.get('single')} / {j.get('plan_build_rebuilt_ms_per_batch', {}).get('single')} ms |")
    return "\n".join(rows)


def points_table(tag):
    rows = ["| point | ms / step | triples/s | kernels µs (HIP events) | sustained / streamed ms |", "|---|---|---|---|---|"]
    for j in load_lines(os.path.join(P, f"{tag}_points.jsonl")):
        if j.get("error"):
            rows.append(f"| {j['point']} | error | | | |")
            continue
        ks = ", ".join(f"{k} {v['avg_us']}" for k, v in j["kernels"].items())
        s, r = (j.get("sustained") or {}).get("ms_per_step"), j.get("ms_per_step_streamed")
        rows.append(f"| {j['point']} | {j['ms_per_step']} | {j['value'] / 1e6:.1f} M | {ks} | {s if s else '—'} / {r if r else '—'} |")
    return "\n".join(rows)


def kernel_table(tag):
    s = json.load(open(os.path.join(P, f"{tag}_pmc_summary.json")))
    rows = ["| kernel (≥ 50 launches) | launches | avg µs (trace pass) | fabric bytes / launch (PMC) | → GB/s | L2 hit | VALU instr. | SQ_WAIT_ANY / SQ_WAVE_CYCLES |",
            "|---|---|---|---|---|---|---|---|"]
    for k in sorted(s):
        v = s[k]
        if v.get("launches", 0) < 50 or "hbm_bytes_per_launch" not in v:
            continue
        c = v.get("counters", {})
        us = v["avg_us"].get("trace") or list(v["avg_us"].values())[0]
        rows.append(f"| `{k}` | {v['launches']} | {us} | {v['hbm_bytes_per_launch'] / 1e6:.1f} MB | {v['hbm_bytes_per_launch'] / us / 1e3:.0f} | {v.get('l2_hit_rate')} | "
                    f"{c.get('SQ_INSTS_VALU', 0) / 1e6:.1f} M | {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f} |")
    return "\n".join(rows)


def profiled_points_table(tag):
    """Per profiled point (tools/final_measure.sh: the default line, cfg2, cfg5, B = 1,048,576, data-file order, cfg5 row-list
    form): the step's two kernels with SURVEY 8(d)'s / the bench line's algorithmic bytes / rocprofv3's average duration /
    8 TB/s, and the fabric bytes of the PMC passes / the algorithmic ones."""
    rows = ["| point | kernel | launches | avg µs (rocprofv3 trace pass) | algorithmic MB | → frac of 8 TB/s | fabric MB (PMC) | PMC ÷ algorithmic | L2 hit |",
            "|---|---|---|---|---|---|---|---|---|"]
    for point in ("", "cfg2", "cfg5", "b1m", "fileorder", "cfg5list"):
        stem = f"{tag}_{point}" if point else tag
        sj, lj = os.path.join(P, f"{stem}_pmc_summary.json"), os.path.join(P, f"{stem}_profiled_line.json")
        if not (os.path.exists(sj) and os.path.exists(lj)):
            continue
        summ, line = json.load(open(sj)), load_lines(lj)[0]
        label = point or "cfg3 (default line)"
        for name, pref in (("fwd", "k_fwd"), ("bwd_adam", "k_bwd"), ("catchup", "k_adam_catchup")):
            if name not in line["kernels"]:
                continue
            alg = line["kernels"][name]["alg_bytes"]
            cands = [(k, v) for k, v in summ.items() if k.startswith(pref) and "hbm_bytes_per_launch" in v and v.get("launches", 0) >= 20]
            if not cands:
                continue
            k, v = max(cands, key=lambda kv: kv[1]["launches"])
            us = v["avg_us"].get("trace") or list(v["avg_us"].values())[0]
            rows.append(f"| {label} | `{k}` | {v['launches']} | {us} | {alg / 1e6:.1f} | **{alg / us / 1e3 / 8000:.3f}** | "
                        f"{v['hbm_bytes_per_launch'] / 1e6:.1f} | {v['hbm_bytes_per_launch'] / alg:.2f} | {v.get('l2_hit_rate')} |")
        r = line["roofline"]
        rows.append(f"| {label} | SURVEY 8(d): forward {r.get('bytes_fwd_8d', 0) / 1e6:.1f} MB, region K {r.get('bytes_K_8d', 0) / 1e6:.1f} MB | | "
                    f"step {line['ms_per_step']} ms under the profiler | | `frac_fwd_8d` {r.get('frac_fwd_8d')}, `frac_K_8d` {r.get('frac_K_8d')} | | | |")
    return "\n".join(rows)


def gather_table(tag):
    path = os.path.join(P, f"{tag}_gather_bench.jsonl")
    if not os.path.exists(path):
        return ""
    pts = load_lines(path)
    out = []
    for tb in sorted({p["table_MB"] for p in pts}):
        for rb in sorted({p["row_bytes"] for p in pts}, reverse=True):
            sel = [p for p in pts if p["table_MB"] == tb and p["row_bytes"] == rb]
            if not sel:
                continue
            rs = sorted({p["rows_in_flight_per_wave"] for p in sel})
            out.append(f"\n{sel[0]['rows_per_launch']:,} random {rb}-byte rows of a {tb} MB table, each fetched once — GB/s (µs per launch):\n")
            out.append("| waves per CU | " + " | ".join(f"{r} row(s) in flight per wave" for r in rs) + " |")
            out.append("|---|" + "---|" * len(rs))
            for w in sorted({p["waves_per_cu"] for p in sel}):
                cells = []
                for r in rs:
                    q = [p for p in sel if p["waves_per_cu"] == w and p["rows_in_flight_per_wave"] == r]
                    cells.append(f"{q[0]['GBs']:.0f} ({q[0]['us']:.0f})" if q else "")
                out.append(f"| {w} | " + " | ".join(cells) + " |")
    return "\n".join(out)


def stats_rows(tag, names):
    rows = ["| kernel (rocprofv3 --kernel-trace --stats) | calls | avg µs |", "|---|---|---|"]
    for r in csv.DictReader(open(os.path.join(P, f"{tag}_kernel_stats.csv"))):
        n = r["Name"]
        if any(x in n for x in names):
            m = re.search(r"(k_\w+(<[^>]*>)?)", n)
            rows.append(f"| `{m.group(1) if m else n[:60]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} |")
    return "\n".join(rows)


def main():
    tag = sys.argv[1]
    makers = {"bench": lambda: bench_table(tag), "kernels": lambda: kernel_table(tag), "points": lambda: points_table(tag),
              "gather": lambda: gather_table(tag),
              "stats": lambda: stats_rows(tag, ("k_fwd", "k_bwd", "k_adam", "k_finalize", "k_index", "k_radix", "k_heavy")),
              "profiled": lambda: profiled_points_table(tag)}
    parts = {}
    for name, make in makers.items():
        try:
            parts[name] = make()
        except FileNotFoundError as e:           # (a round that has not produced that file yet)
            print(f"[profile_tables] {name}: {e.filename} missing, table skipped", file=sys.stderr)
    if "--write" in sys.argv:
        path = os.path.join(P, "README.md")
        txt = open(path).read()
        for name, body in parts.items():
            a, b = f"<!-- tables:{tag}:{name} -->", f"<!-- /tables:{tag}:{name} -->"
            if a in txt and b in txt:
                txt = txt[:txt.index(a) + len(a)] + "\n" + body + "\n" + txt[txt.index(b):]
        open(path, "w").write(txt)
    else:
        for name, body in parts.items():
            print(f"## {name}\n{body}\n")


if __name__ == "__main__":
    main()
