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


def test_problem_struct_layout_matches_header(tmp_path):
    """sizeof / field offsets of the ctypes mirror == what a C compiler makes of include/vfm_hip.h."""
    import subprocess
    from vae_amd._lib import Problem
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vfm_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vfm_problem_t), offsetof(vfm_problem_t, F),'
                   'offsetof(vfm_problem_t, group_hi), offsetof(vfm_problem_t, group_n), offsetof(vfm_problem_t, seed),'
                   'offsetof(vfm_problem_t, e_lo), offsetof(vfm_problem_t, flags), offsetof(vfm_problem_t, n_samples),'
                   'offsetof(vfm_problem_t, coord_off));'
                   'printf("%zu %zu\\n", offsetof(vfm_problem_t, dev_step), offsetof(vfm_problem_t, wrec));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [C.sizeof(Problem), Problem.F.offset, Problem.group_hi.offset, Problem.group_n.offset,
                   Problem.seed.offset, Problem.e_lo.offset, Problem.flags.offset, Problem.n_samples.offset,
                   Problem.coord_off.offset, Problem.dev_step.offset, Problem.wrec.offset]


def test_step_state_struct_layouts_match_header(tmp_path):
    """vfm_step_consts_t (64 bytes, 16 floats per table row) and vfm_dev_step_t (64 bytes = the int64[8] device tensor)."""
    import subprocess
    from vae_amd._lib import StepConsts, DevStep
    src = tmp_path / "st.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vfm_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vfm_step_consts_t), offsetof(vfm_step_consts_t, store_true),'
                   'offsetof(vfm_step_consts_t, k), offsetof(vfm_step_consts_t, lr), sizeof(vfm_dev_step_t),'
                   'offsetof(vfm_dev_step_t, adam_step_bwd), offsetof(vfm_dev_step_t, tab), offsetof(vfm_dev_step_t, error));return 0;}\n')
    exe = tmp_path / "st"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [C.sizeof(StepConsts), StepConsts.store_true.offset, StepConsts.k.offset, StepConsts.lr.offset,
                   C.sizeof(DevStep), DevStep.adam_step_bwd.offset, DevStep.tab.offset, DevStep.error.offset]
    assert C.sizeof(StepConsts) == 64 and C.sizeof(DevStep) == 64


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


def test_index_struct_layout_matches_header(tmp_path):
    import subprocess
    from vae_amd._lib import Index
    src = tmp_path / "ix.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vfm_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(vfm_index_t), offsetof(vfm_index_t, heavy_acc),'
                   'offsetof(vfm_index_t, n_items), offsetof(vfm_index_t, touched_ids), offsetof(vfm_index_t, n_touched));'
                   'return 0;}\n')
    exe = tmp_path / "ix"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [C.sizeof(Index), Index.heavy_acc.offset, Index.n_items.offset, Index.touched_ids.offset,
                   Index.n_touched.offset]


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
    q = ops._problem(spec, 4, 4, 64, flags=ops.FLAG_PARTIAL_PRED)
    assert q.flags == ops.FLAG_PARTIAL_PRED and q.n_samples == 2
    lib = _lib.load()
    assert lib.vfm_elbo_fwd_f32(C.byref(q), *([None] * 15)) == -2      # rejected BECAUSE of the flag (S > 1)
    assert b"VFM_FLAG_PARTIAL_PRED" in lib.vfm_last_error()
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
