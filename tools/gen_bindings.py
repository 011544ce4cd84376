#!/usr/bin/env python3
"""ctypes mirrors of the C ABI's structs and constants, GENERATED from include/vfm_hip.h -- the one place the layouts are
written down.  Two outputs, both committed and both checked by tests/test_abi_cpu.py against this generator AND against
gcc's sizeof / offsetof:

  vae_amd/_abi_gen.py      what the package itself binds with (vae_amd/_lib.py imports it)
  INTEGRATION.md           the block between `<!-- BEGIN GENERATED BINDINGS -->` and `<!-- END GENERATED BINDINGS -->`:
                           the binding a maintainer of the reference would paste (the reference's call being replaced is
                           `model(indices)`, vfm-torch.py:353)

usage: python tools/gen_bindings.py            (rewrites both)
       python tools/gen_bindings.py --check    (exit 1 if either is out of date)
Round 3's vfm_index_t grew by a field while the hand-written mirror in INTEGRATION.md did not: a caller following the
document passed a 72-byte struct and the library read past it.  Since ABI 5 the library refuses a struct whose leading
(struct_size, abi_version) differ from its own, and nobody types a mirror by hand any more."""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vfm_hip.h")
OUT_PY = os.path.join(ROOT, "vae_amd", "_abi_gen.py")
OUT_MD = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED BINDINGS -->", "<!-- END GENERATED BINDINGS -->"

CTYPE = {"uint32_t": "C.c_uint32", "int32_t": "C.c_int32", "int64_t": "C.c_int64", "uint64_t": "C.c_uint64",
         "float": "C.c_float", "double": "C.c_double"}
CLASS = {"vfm_problem_t": "Problem", "vfm_index_t": "Index", "vfm_pipe_t": "Pipe", "vfm_step_consts_t": "StepConsts",
         "vfm_dev_step_t": "DevStep"}
GUARDED = ("vfm_problem_t", "vfm_index_t", "vfm_pipe_t")        # start with (struct_size, abi_version)


def strip_comments(text: str) -> str:
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse(header_text: str):
    """-> (constants {name: int}, structs [(c_name, [(field, ctype expr, c decl)])])"""
    text = strip_comments(header_text)
    consts = {}
    for m in re.finditer(r"^#define\s+(VFM_[A-Z0-9_]+)\s+(\(?-?[0-9A-Za-z_*+() ]+\)?)\s*$", text, flags=re.M):
        name, val = m.group(1), m.group(2)
        try:
            consts[name] = int(eval(val, {"__builtins__": {}}, dict(consts)))
        except Exception:
            pass
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        body, cname = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            dm = re.match(r"^(const\s+)?(\w+)\s*(.*)$", decl)
            base, rest = dm.group(2), dm.group(3)
            for d in rest.split(","):
                d = d.strip()
                ptr = d.startswith("*") or base not in CTYPE
                name = d.lstrip("* ").strip()
                arr = re.match(r"^(\w+)\[(\w+)\]$", name)
                if ptr and "*" not in d:
                    raise ValueError(f"{cname}: cannot parse declarator {decl!r}")
                if ptr:
                    fields.append((name, "C.c_void_p", f"{base}*"))
                elif arr:
                    n = consts.get(arr.group(2), None) if not arr.group(2).isdigit() else int(arr.group(2))
                    fields.append((arr.group(1), f"{CTYPE[base]} * {arr.group(2) if not arr.group(2).isdigit() else n}", f"{base}[{arr.group(2)}]"))
                else:
                    fields.append((name, CTYPE[base], base))
        structs.append((cname, fields))
    return consts, structs


def render(consts, structs) -> str:
    out = ["import ctypes as C", ""]
    for k in sorted(consts):
        out.append(f"{k} = {consts[k]}")
    out.append("")
    for cname, fields in structs:
        cls = CLASS.get(cname, cname)
        out.append("")
        out.append(f"class {cls}(C.Structure):")
        out.append(f'    """Mirror of `{cname}` (include/vfm_hip.h)."""')
        out.append("    _fields_ = [")
        for name, ctype, cdecl in fields:
            out.append(f'        ("{name}", {ctype}),'.ljust(56) + f"# {cdecl}")
        out.append("    ]")
        if cname in GUARDED:
            out.append("")
            out.append("    def __init__(self, *a, **k):")
            out.append("        super().__init__(*a, **k)")
            out.append("        # what VFM_STRUCT_INIT does in C: the library refuses a struct whose size / version differ from its own")
            out.append("        self.struct_size, self.abi_version = C.sizeof(type(self)), VFM_ABI_VERSION")
        out.append("")
    return "\n".join(out).replace("VFM_MAX_FIELDS]", "VFM_MAX_FIELDS]").rstrip() + "\n"


def generated_python(header_text: str) -> str:
    consts, structs = parse(header_text)
    body = render(consts, structs)
    return ('"""GENERATED by tools/gen_bindings.py from include/vfm_hip.h -- do not edit; rerun the script after changing the\n'
            'header (tests/test_abi_cpu.py fails on drift and checks every size / offset against gcc)."""\n' + body)


def generated_markdown(header_text: str) -> str:
    consts, structs = parse(header_text)
    return (BEGIN + "\n(generated by `python tools/gen_bindings.py` from `include/vfm_hip.h`; `tests/test_abi_cpu.py` extracts this "
            "block, executes it and compares every `sizeof` / field offset with gcc's)\n\n```python\n" + render(consts, structs) + "```\n" + END)


def splice_markdown(md: str, block: str) -> str:
    if BEGIN not in md or END not in md:
        raise SystemExit(f"INTEGRATION.md lacks the markers {BEGIN} ... {END}")
    a, b = md.index(BEGIN), md.index(END) + len(END)
    return md[:a] + block + md[b:]


def main():
    check = "--check" in sys.argv
    h = open(HEADER).read()
    py, md = generated_python(h), splice_markdown(open(OUT_MD).read(), generated_markdown(h))
    stale = []
    if not os.path.exists(OUT_PY) or open(OUT_PY).read() != py:
        stale.append(OUT_PY)
        if not check:
            open(OUT_PY, "w").write(py)
    if open(OUT_MD).read() != md:
        stale.append(OUT_MD)
        if not check:
            open(OUT_MD, "w").write(md)
    if check and stale:
        print("out of date:", *stale)
        raise SystemExit(1)
    print("up to date" if not stale else "rewrote: " + ", ".join(stale))


if __name__ == "__main__":
    main()
