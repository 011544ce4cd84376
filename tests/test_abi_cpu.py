"""CPU: the C-ABI library and the torch.ops shim load and export everything include/vfm_hip.h
declares (no compute without a GPU); argument errors are reported, not crashed on."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "vfm_hip.h")).read()
    return sorted(set(re.findall(r"\b(vfm_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported():
    from vae_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.EXPORTS)
    assert lib.vfm_abi_version() == _lib.ABI_VERSION


def _gcc_layouts(tmp_path):
    """{struct: (sizeof, {field: offset})} as gcc lays out include/vfm_hip.h -- every struct, every field."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_bindings as G
    _, structs = G.parse(open(G.HEADER).read())
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include <string.h>', '#include "vfm_hip.h"', 'int main(){']
    for cname, fields in structs:
        lines.append(f'printf("S {cname} %zu\\n", sizeof({cname}));')
        for name, _, _ in fields:
            lines.append(f'printf("F {cname} {name} %zu\\n", offsetof({cname}, {name}));')
    lines.append("return 0;}")
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = {}
    for ln in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        t = ln.split()
        if t[0] == "S":
            out[t[1]] = (int(t[2]), {})
        else:
            out[t[1]][1][t[2]] = int(t[3])
    return out, structs, G


def _check_mirrors(ns, layouts, structs, G):
    for cname, fields in structs:
        cls = ns[G.CLASS[cname]]
        size, offs = layouts[cname]
        assert C.sizeof(cls) == size, cname
        assert [n for n, _ in cls._fields_] == [n for n, _, _ in fields], cname       # every member, in order
        for name, _, _ in fields:
            assert getattr(cls, name).offset == offs[name], (cname, name)


def test_every_struct_mirror_matches_gcc(tmp_path):
    """sizeof and EVERY field offset of the package's ctypes mirrors (vae_amd/_abi_gen.py, generated) == what a C compiler
    makes of include/vfm_hip.h -- vfm_problem_t, vfm_index_t (occ_other, max_items, status included), vfm_pipe_t
    (last_step, step_tab included), vfm_step_consts_t, vfm_dev_step_t."""
    from vae_amd import _lib
    layouts, structs, G = _gcc_layouts(tmp_path)
    ns = {"Problem": _lib.Problem, "Index": _lib.Index, "Pipe": _lib.Pipe, "StepConsts": _lib.StepConsts, "DevStep": _lib.DevStep}
    _check_mirrors(ns, layouts, structs, G)
    assert C.sizeof(_lib.StepConsts) == 64 and C.sizeof(_lib.DevStep) == 64      # (table rows of 16 floats; an int64[8] device tensor)
    p, ix, pp = _lib.Problem(), _lib.Index(), _lib.Pipe()
    assert (p.struct_size, p.abi_version) == (C.sizeof(_lib.Problem), _lib.ABI_VERSION)
    assert (ix.struct_size, pp.struct_size) == (C.sizeof(_lib.Index), C.sizeof(_lib.Pipe))


def test_the_published_binding_in_INTEGRATION_md_matches_gcc(tmp_path):
    """The ctypes block a maintainer would paste from INTEGRATION.md is extracted from the DOCUMENT, executed, and checked
    field by field against gcc -- the round-3 defect was exactly a stale hand-typed mirror there."""
    layouts, structs, G = _gcc_layouts(tmp_path)
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md[md.index(G.BEGIN): md.index(G.END)]
    code = block[block.index("```python") + len("```python"): block.rindex("```")]
    ns = {}
    exec(code, ns)
    _check_mirrors(ns, layouts, structs, G)
    assert ns["VFM_ABI_VERSION"] == 5 and ns["Index"]().struct_size == layouts["vfm_index_t"][0]


def test_generated_bindings_are_up_to_date():
    """vae_amd/_abi_gen.py and the INTEGRATION.md block are what tools/gen_bindings.py makes of the header today."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_a_struct_of_another_layout_is_refused():
    """ABI 5: a caller built against another layout is REFUSED (VFM_E_INVALID), not read past its end.  The 72-byte
    vfm_index_t of the round-3 document (ten fields, no max_items) is the regression case; also a wrong version, a
    zeroed header, a vfm_problem_t and a vfm_pipe_t of the wrong size."""
    from vae_amd import _lib
    lib = _lib.load()

    class OldIndex(C.Structure):              # INTEGRATION.md as of round 3: 72 bytes
        _fields_ = [("occ_ptr", C.c_void_p), ("occ_rows", C.c_void_p), ("heavy_ids", C.c_void_p),
                    ("heavy_items", C.c_void_p), ("heavy_acc", C.c_void_p), ("n_heavy", C.c_int32),
                    ("n_items", C.c_int32), ("touched_ids", C.c_void_p), ("n_touched", C.c_int64),
                    ("occ_other", C.c_void_p)]
    assert C.sizeof(OldIndex) == 72
    p = _lib.Problem()
    p.B, p.B_global, p.T, p.F, p.d, p.id_bits, p.n_samples = 4, 4, 10, 2, 8, 64, 1
    old = OldIndex(0x1000, 0x1000, None, None, None, 0, 0)
    bwd = lib.vfm_elbo_bwd_f32
    keep = bwd.argtypes
    bwd.argtypes = [C.POINTER(_lib.Problem), C.c_void_p] + list(keep[2:])
    try:
        rc = bwd(C.byref(p), C.cast(C.pointer(old), C.c_void_p), *([None] * 16))
    finally:
        bwd.argtypes = keep
    assert rc == -1 and b"vfm_index_t" in lib.vfm_last_error() and b"struct_size" in lib.vfm_last_error()
    ix = _lib.Index()
    ix.occ_ptr = ix.occ_rows = 0x1000
    ix.abi_version = 4
    assert lib.vfm_elbo_bwd_f32(C.byref(p), C.byref(ix), *([None] * 16)) == -1 and b"abi_version" in lib.vfm_last_error()
    ix.abi_version, ix.struct_size = _lib.ABI_VERSION, C.sizeof(_lib.Index) - 8
    assert lib.vfm_elbo_bwd_adam_f32(C.byref(p), C.byref(ix), *([None] * 17), 0.1, 0.9, 0.999, 1e-8, 1, None, None) == -1
    assert b"vfm_index_t" in lib.vfm_last_error()
    ix.struct_size = C.sizeof(_lib.Index)
    assert lib.vfm_elbo_bwd_f32(C.byref(p), C.byref(ix), *([None] * 16)) == -1 and b"NULL pointer" in lib.vfm_last_error()
    p.struct_size = 0
    assert lib.vfm_elbo_fwd_f32(C.byref(p), *([None] * 15)) == -1 and b"vfm_problem_t" in lib.vfm_last_error()
    assert lib.vfm_batch_norms(C.byref(p), None, None, None, None) == -1
    assert lib.vfm_variant_fwd_f32(C.byref(p), 0, *([None] * 18)) == -1 and b"struct_size" in lib.vfm_last_error()
    p.struct_size = C.sizeof(_lib.Problem)
    p.F, p.d, p.flags = 2, 8, 32
    pipe = _lib.Pipe()
    pipe.zrec = 0x1000
    pipe.struct_size -= 16                      # (vfm_pipe_t before last_step / step_tab were added)
    assert lib.vfm_elbo_bwd_adam_pipe_f32(C.byref(p), C.byref(ix), C.byref(pipe), *([None] * 13), 0.1, 0.9, 0.999, 1e-8, 1,
                                          None, None) == -1
    assert b"vfm_pipe_t" in lib.vfm_last_error()


def test_step_consts_host_helper_and_dev_step_rejections():
    """vfm_step_consts is pure host arithmetic (runs here): bias corrections, the scaled-moment factors, the period
    position; entry points that do not read the step from device memory reject a dev_step pointer."""
    from vae_amd import _lib
    lib = _lib.load()
    c = _lib.StepConsts()
    assert lib.vfm_step_consts(0.01, 0.9, 0.999, 1e-8, 128, 1, C.byref(c)) == 0
    assert c.k == 128 and c.store_true == 1 and c.scaled == 1
    b1, b2, lr_ = C.c_float(0.9).value, C.c_float(0.999).value, C.c_float(0.01).value      # (the fp32 values the C side sees)
    assert abs(c.step_size - lr_ / (1 - b1 ** 128)) < 1e-9 and abs(c.bc2_sqrt - (1 - b2 ** 128) ** 0.5) < 1e-7
    assert abs(c.s1 - b1 ** 128) < 1e-12 and abs(c.a1 - c.step_size * b1 ** 128) < 1e-12
    assert lib.vfm_step_consts(0.01, 0.9, 0.999, 1e-8, 129, 0, C.byref(c)) == 0
    assert c.k == 1 and c.store_true == 0 and c.scaled == 0 and abs(c.c1 - (1 - b1)) < 1e-7
    assert lib.vfm_step_consts(0.01, 0.9, 0.999, 1e-8, 0, 1, C.byref(c)) == -1
    p = _lib.Problem()
    p.B, p.B_global, p.T, p.F, p.d, p.id_bits, p.n_samples = 4, 4, 10, 2, 8, 64, 1
    p.dev_step = 0x1000
    assert lib.vfm_elbo_finalize_f32(C.byref(p), None, None, None, None) == -2 and b"dev_step" in lib.vfm_last_error()
    assert lib.vfm_elbo_bwd_f32(C.byref(p), *([None] * 17)) == -2
    p.dev_step = 0x1008
    assert lib.vfm_elbo_fwd_f32(C.byref(p), *([None] * 15)) == -1 and b"16-byte" in lib.vfm_last_error()
    p.dev_step, p.wrec = None, 0x1004
    assert lib.vfm_elbo_fwd_f32(C.byref(p), *([None] * 15)) == -1 and b"wrec" in lib.vfm_last_error()
    # the catch-up call never reads its learning-rate table past n_lr
    lr = (C.c_float * 1)(0.1)
    one = 0x1000
    assert lib.vfm_adam_catchup_f32(one, one, one, one, one, one, one, None, 10, 10, 8, lr, 1, 0.9, 0.999, 1e-8, 200, 200,
                                    None, None) == -1
    assert b"n_lr" in lib.vfm_last_error()


def test_problem_mirror_has_every_header_field_and_rejects_unknown_names():
    """Every member of vfm_problem_t is a ctypes field of the same name (ctypes would otherwise keep an
    assignment as a plain Python attribute and pass 0 to the library), and a flag set through the mirror
    reaches the library."""
    from vae_amd import _lib, ops
    hdr = open(os.path.join(ROOT, "include", "vfm_hip.h")).read()
    body = hdr[hdr.index("typedef struct vfm_problem {"): hdr.index("} vfm_problem_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    members = []
    for decl in re.findall(r"\b(?:u?int(?:32|64)_t\s|double\s|float\s*\*|vfm_dev_step_t\s*\*)\s*([^;]+);", body):
        members += [re.sub(r"\[.*\]", "", m).strip() for m in decl.split(",")]
    assert members == [n for n, _ in _lib.Problem._fields_]
    p = _lib.Problem()
    with pytest.raises(AttributeError):
        p.flagz = 3
    # everything ops._problem sets is a real field, and a non-zero flag is seen by the C side
    spec = ops.Spec(T=10, F=2, d=8, group_hi=(5, 10), group_n=(5.0, 5.0), likelihood=0, n_samples=2)
    q = ops._problem(spec, 4, 4, 64, flags=256)       # (a bit no flag of this ABI uses: VFM_FLAG_PARTIAL_PRED of ABI 4)
    assert q.flags == 256 and q.n_samples == 2
    lib = _lib.load()
    assert lib.vfm_elbo_fwd_f32(C.byref(q), *([None] * 15)) == -1      # rejected BECAUSE of the flag
    assert b"unknown bit" in lib.vfm_last_error()
    q.flags = 0
    assert lib.vfm_elbo_fwd_f32(C.byref(q), *([None] * 15)) == -1      # same call without it: NULL pointers


def test_bad_arguments_return_errors_not_crashes():
    from vae_amd import _lib
    lib = _lib.load()
    p = _lib.Problem()
    assert lib.vfm_elbo_fwd_f32(C.byref(p), *([None] * 15)) != 0
    assert b"bad B or T" in lib.vfm_last_error() or len(lib.vfm_last_error()) > 0
    p.B, p.B_global, p.T, p.F, p.d, p.id_bits, p.n_samples = 4, 4, 10, 2, 8, 64, 1
    assert lib.vfm_elbo_fwd_f32(C.byref(p), *([None] * 15)) == -1          # NULL pointers
    p.n_samples = 65
    assert lib.vfm_elbo_fwd_f32(C.byref(p), *([None] * 15)) == -1          # S outside [1,64]
    p.n_samples = 3
    assert lib.vfm_elbo_bwd_acc_f32(C.byref(p), *([None] * 7)) == -2       # the multi-rank stages carry one sample
    assert b"n_samples" in lib.vfm_last_error()
    p.n_samples, p.d = 1, 1027
    assert lib.vfm_elbo_fwd_f32(C.byref(p), *([None] * 15)) == -2          # unsupported d
    assert lib.vfm_adam_f32(None, None, None, None, 4, 0.1, 0.9, 0.999, 1e-8, 1, None) == -1
    # the heavy lists rebuilt with a lower threshold: VFM_HEAVY_MIN <= threshold <= heavy_list, no NULL tables
    buf = (C.c_int32 * 64)()
    assert lib.vfm_rebuild_heavy(10, buf, buf, 64, 65, buf, 4, buf, 4, buf, None) == -1 and b"threshold" in lib.vfm_last_error()
    assert lib.vfm_rebuild_heavy(10, buf, buf, 64, 4, buf, 4, buf, 4, buf, None) == -1
    assert lib.vfm_rebuild_heavy(10, None, buf, 64, 16, buf, 4, buf, 4, buf, None) == -1
    assert lib.vfm_heavy_threshold(64) == 16 and lib.vfm_heavy_threshold(19) == 8 and lib.vfm_heavy_threshold(8) == 8


def test_torch_ops_shim_registers_schemas():
    from vae_amd import _lib
    ops = _lib.ops()
    assert ops.abi_version() == _lib.ABI_VERSION
    for name in ("elbo_fwd", "elbo_finalize", "elbo_bwd", "elbo_bwd_adam", "adam"):
        assert hasattr(ops, name)
    # CPU tensors are rejected loudly (no fallback)
    x = torch.zeros(4, 2, dtype=torch.int64)
    t = torch.zeros(10, 16)
    with pytest.raises(RuntimeError):
        ops.elbo_fwd(x, None, t, torch.zeros(10, 2), None, torch.zeros(3), None, None, None, None,
                     torch.zeros(4), torch.zeros(8 * 4097, dtype=torch.float64), None, None,
                     [5, 10], [5.0, 5.0], 1, 4, 0, 0, 0, 0)


def test_cpu_model_ops_fail_loudly():
    from vae_amd.model import VFM
    from vae_amd._lib import VfmLibraryError
    m = VFM(5, 5, 4, device="cpu")
    with pytest.raises(VfmLibraryError):
        m.set_training_data(torch.tensor([[0, 5]]), nb_train=1)
