#!/usr/bin/env python3
"""Instruction-class counts of named kernel instances from the gfx950 ISA hipcc emits (`--offload-device-only -S`):
the evidence behind "hand-written CDNA4, no scratch, counted waits" without asking the reader to recompile.
usage: tools/isa_summary.py > profiles/rNN_isa_summary.txt       (cross-compiles; no GPU needed; ~3 minutes)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vae_amd", "csrc")
# (translation unit, extra flags, substring of the demangled instance, what it is)
WANT = [
    ("vfm_fwd2.hip", [], "k_fwd2<16, true, 0, 1, true, 0, true>", "cfg3 forward (d = 128, Philox, training, int64 ids, |.| link, packed records)"),
    ("vfm_fwd2.hip", [], "k_fwd2<4, false, 0, 1, true, 0, true>", "cfg2 forward (d = 20)"),
    ("vfm_fwdg.hip", [], "k_fwdg<32, true, 0, 1, true, 0, true>", "cfg5 forward (F = 32, d = 256: fields split over lane groups)"),
    ("vfm_bwd.hip", ["-ffp-contract=on", "-DVFM_BWD_PART=1"], "k_bwd<32, 1, 4, 0, 1, 0, 0, false, false, true>", "cfg3 fused backward + dense Adam, look-ahead form"),
    ("vfm_bwd.hip", ["-ffp-contract=on", "-DVFM_BWD_PART=1"], "k_bwd<64, 1, 4, 0, 1, 0, 0, false, false, true>", "cfg5 fused backward + dense Adam, look-ahead form"),
    ("vfm_bwd.hip", ["-ffp-contract=on", "-DVFM_BWD_PART=1"], "k_bwd<32, 1, 4, 0, 1, 0, 0, false, true, true>", "pipelined step's backward in the look-ahead form (data-file-order batches; the rows exchange)"),
    ("vfm_bwd.hip", ["-ffp-contract=on", "-DVFM_BWD_PART=1"], "k_bwd_small<8, 0>", "cfg2 one-launch backward + dense Adam (a wave per table row; the 48 bytes of scratch are the epilogue's eps chunk select, once per row)"),
    ("vfm_bwd.hip", ["-ffp-contract=on", "-DVFM_BWD_PART=1"], "k_bwd<32, 1, 4, 0, 1, 2, 0, false, false, false>", "multi-rank apply stage (epilogue + Adam from the summed statistics)"),
    ("vfm_abi.hip", ["-ffp-contract=on"], "k_heavy<32, 1, 4>", "pre-reduction of the long lists' work items (eight occurrences in flight, the next eight's row numbers fetched under them)"),
    ("vfm_index.hip", [], "k_index_keys", "index build: ids -> keys, first digit counts, batch normalisers"),
    ("vfm_index.hip", [], "k_radix_scatter<true>", "index build: last radix pass (writes occ_rows / occ_other)"),
    ("vfm_index.hip", [], "k_index_count<true>", "index build: occ_ptr by LDS-staged lower bounds + per-chunk counts (256-thread workgroups)"),
    ("vfm_index.hip", [], "k_index_write<false>", "index build: heavy lists, work items, touched list, totals, W"),
]
CLASSES = ["global_load_dwordx4", "global_load_dwordx2", "global_load_dword", "global_store_dwordx4", "global_store_dwordx2",
           "global_store_dword", "ds_read_b128", "ds_write_b128", "ds_read", "ds_write", "s_barrier", "v_pk_fma_f32", "v_pk_mul_f32",
           "v_pk_add_f32", "v_fma_f32", "v_mad_u64_u32", "v_log_f32", "v_sqrt_f32", "v_rcp_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32",
           "v_mov_b32_dpp", "v_add_f32_dpp", "v_permlane", "scratch_", "buffer_", "global_atomic"]


def main():
    hipcc = os.environ.get("HIPCC", "hipcc")
    cache = {}
    for tu, extra, inst, what in WANT:
        if (tu, tuple(extra)) not in cache:
            out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
            subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                            "-DVFM_LINK=0", "--offload-device-only", "-S", os.path.join(CSRC, tu), "-o", out] + extra, check=True,
                           stderr=subprocess.DEVNULL)
            txt = open(out).read()
            os.unlink(out)
            labels = re.findall(r"\n(_Z[^\s:]+):", txt)
            dem = subprocess.run(["c++filt"] + labels, capture_output=True, text=True).stdout.strip().split("\n")
            cache[(tu, tuple(extra))] = (txt, dict(zip(dem, labels)))
        txt, names = cache[(tu, tuple(extra))]
        hit = [(d, l) for d, l in names.items() if inst in d]
        if not hit:
            print(f"## {inst}: not found in {tu}\n")
            continue
        d, label = hit[0]
        body = txt[txt.index("\n" + label + ":"):]
        end = body.index(".Lfunc_end")                    # (a kernel may hold several s_endpgm: early exits)
        meta = body[end: end + 6000]
        body = body[:end]
        ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";", "//", "_Z")) and not l.strip().endswith(":")]
        ops = collections.Counter(i.split()[0] for i in ins)
        print(f"## {inst}\n{what}   ({tu})")
        g = lambda k: (re.search(r"; " + k + r": (\d+)", meta) or [None, "?"])[1]
        print(f"instructions {len(ins)}; vgprs {g('NumVgprs')}, "
              f"scratch {g('ScratchSize')} bytes/lane, LDS {g('LDSByteSize')} bytes, occupancy {g('Occupancy')} waves/SIMD")
        for c in CLASSES:
            n = sum(v for k, v in ops.items() if k.startswith(c)) if c.endswith("_") or c in ("ds_read", "ds_write", "v_permlane", "global_atomic") else \
                sum(v for k, v in ops.items() if k == c or k.startswith(c + "_e"))
            if n:
                print(f"  {c:<22}{n}")
        waits = collections.Counter(i for i in ins if i.startswith("s_waitcnt"))
        vm = {k: v for k, v in waits.items() if "vmcnt" in k}
        print("  s_waitcnt vmcnt:", ", ".join(f"{k.split('vmcnt')[1].split()[0]}x{v}" for k, v in sorted(vm.items())),
              "  (vmcnt(0) = wait for every load in flight)")
        print()


if __name__ == "__main__":
    main()
