"""Loader for the C-ABI library `libvfm_hip.so` (include/vfm_hip.h).

The library is built in-tree by `vae_amd.build.build_all()` (hipcc, gfx950).  There is NO
fallback: if it is missing or does not load, every op of this package raises.
`import torch` must happen before the library is opened so that both resolve the same
HIP runtime (`libamdhip64.so.7`).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

HERE = os.path.dirname(os.path.abspath(__file__))
# VFM_LIB_DIR: another build of the two libraries (the host-sanitizer build of vae_amd.build.build_sanitized)
LIB_DIR = os.environ.get("VFM_LIB_DIR") or HERE
LIB_PATH = os.path.join(LIB_DIR, "libvfm_hip.so")
OPS_PATH = os.path.join(LIB_DIR, "libvfm_torch_ops.so")

from . import _abi_gen as _gen      # GENERATED from include/vfm_hip.h (tools/gen_bindings.py): constants + struct mirrors

ABI_VERSION = _gen.VFM_ABI_VERSION
MAX_FWD_BLOCKS = _gen.VFM_MAX_FWD_BLOCKS
MAX_FIELDS = _gen.VFM_MAX_FIELDS
N_PARTIALS = _gen.VFM_N_PARTIALS
PARTIALS_LEN = _gen.VFM_PARTIALS_LEN
P_LL, P_KL, P_G, P_ALPHA, P_BADID = _gen.VFM_P_LL, _gen.VFM_P_KL, _gen.VFM_P_G, _gen.VFM_P_ALPHA, _gen.VFM_P_BADID
LIK_NORMAL, LIK_BERNOULLI = _gen.VFM_LIK_NORMAL, _gen.VFM_LIK_BERNOULLI
OBJ_SAMPLED, OBJ_CLOSED_FORM = _gen.VFM_OBJ_SAMPLED, _gen.VFM_OBJ_CLOSED_FORM

EXPORTS = (
    "vfm_abi_version", "vfm_last_error", "vfm_inv_occ_f32", "vfm_batch_norms",
    "vfm_elbo_fwd_f32", "vfm_elbo_finalize_f32", "vfm_elbo_bwd_f32", "vfm_philox_eps_f32",
    "vfm_adam_f32", "vfm_elbo_bwd_adam_f32", "vfm_elbo_bwd_acc_f32", "vfm_elbo_apply_adam_f32",
    "vfm_moments_rescale_f32", "vfm_index_workspace_bytes", "vfm_build_index", "vfm_heavy_list_for",
    "vfm_variant_fwd_f32", "vfm_variant_bwd_f32", "vfm_variant_workspace_elems", "vfm_adam_catchup_f32", "vfm_union_rows", "vfm_union_workspace_bytes",
    "vfm_sample_records_f32", "vfm_elbo_bwd_adam_pipe_f32", "vfm_elbo_bwd_adam_lookahead_f32",
    "vfm_step_consts", "vfm_dev_step_set", "vfm_wrec_build_f32", "vfm_elbo_apply_adam_rows_f32",
    "vfm_elbo_bwd_acc_rows_f32", "vfm_heavy_threshold", "vfm_rebuild_heavy",
)


class _Strict:
    """ctypes accepts ANY attribute name and silently keeps it as a Python attribute: a misspelt or missing field
    would leave the struct member 0 without an error."""

    def __setattr__(self, name, value):
        if name not in type(self)._names:
            raise AttributeError(f"{type(self).__name__} has no field {name!r}")
        super().__setattr__(name, value)


class Problem(_Strict, _gen.Problem):
    """`vfm_problem_t` (struct_size / abi_version are set by the generated __init__)."""


class Index(_Strict, _gen.Index):
    """`vfm_index_t`."""


class Pipe(_Strict, _gen.Pipe):
    """`vfm_pipe_t`."""


StepConsts, DevStep = _gen.StepConsts, _gen.DevStep
for _c in (Problem, Index, Pipe):
    _c._names = frozenset(n for n, _ in _c._fields_)


def heavy_list_for(n_occ: int, T: int) -> int:
    """Length above which an occurrence list is cut in work items: the library's rule (vfm_heavy_list_for)."""
    return int(load().vfm_heavy_list_for(int(n_occ), int(T)))


class VfmLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Open libvfm_hip.so (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VfmLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  vae_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    PP = C.POINTER(Problem)
    lib.vfm_abi_version.restype = C.c_int
    lib.vfm_abi_version.argtypes = []
    lib.vfm_last_error.restype = C.c_char_p
    lib.vfm_last_error.argtypes = []
    lib.vfm_inv_occ_f32.argtypes = [vp, vp, i64, vp]
    lib.vfm_batch_norms.argtypes = [PP, vp, vp, vp, vp]
    lib.vfm_elbo_fwd_f32.argtypes = [PP] + [vp] * 15
    lib.vfm_elbo_finalize_f32.argtypes = [PP, vp, vp, vp, vp]
    lib.vfm_elbo_bwd_f32.argtypes = [PP] + [vp] * 17
    lib.vfm_philox_eps_f32.argtypes = [PP, vp, vp, vp, vp]
    lib.vfm_elbo_bwd_adam_f32.argtypes = ([PP] + [vp] * 18 +
                                          [C.c_float, C.c_float, C.c_float, C.c_float, i64, vp, vp])
    lib.vfm_adam_f32.argtypes = [vp, vp, vp, vp, i64, C.c_float, C.c_float, C.c_float, C.c_float,
                                 i64, vp]
    lib.vfm_elbo_bwd_acc_f32.argtypes = [PP] + [vp] * 7
    lib.vfm_elbo_apply_adam_f32.argtypes = ([PP] + [vp] * 16 +
                                            [C.c_float, C.c_float, C.c_float, C.c_float, i64, vp])
    lib.vfm_elbo_bwd_acc_rows_f32.argtypes = [PP, C.POINTER(Index), vp, i64] + [vp] * 6
    lib.vfm_elbo_apply_adam_rows_f32.argtypes = ([PP, vp, vp, vp, i64, i32] + [vp] * 11 +
                                                 [C.c_float, C.c_float, C.c_float, C.c_float, i64, vp])
    lib.vfm_moments_rescale_f32.argtypes = [vp, vp, i64, C.c_float, C.c_float, i64, i32, vp]
    lib.vfm_index_workspace_bytes.argtypes = [i64, i32, i64]
    lib.vfm_heavy_list_for.argtypes = [i64, i64]
    lib.vfm_build_index.argtypes = [i64, i32, i64, i32, vp, vp, vp, vp, i32, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.vfm_rebuild_heavy.argtypes = [i64, vp, vp, i32, i32, vp, i64, vp, i64, vp, vp]
    lib.vfm_adam_catchup_f32.argtypes = [vp] * 8 + [i64, i64, i32, C.POINTER(C.c_float), i64, C.c_float, C.c_float, C.c_float,
                                         i64, i64, vp, vp]
    lib.vfm_step_consts.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, i64, i32, C.POINTER(StepConsts)]
    lib.vfm_dev_step_set.argtypes = [vp, C.c_uint64, i64, vp]
    lib.vfm_wrec_build_f32.argtypes = [vp, vp, i64, vp, vp]
    lib.vfm_sample_records_f32.argtypes = [PP, vp, i64, vp, vp, vp, vp, vp, vp]
    lib.vfm_elbo_bwd_adam_pipe_f32.argtypes = ([PP, C.POINTER(Index), C.POINTER(Pipe)] + [vp] * 13 +
                                               [C.c_float, C.c_float, C.c_float, C.c_float, i64, vp, vp])
    lib.vfm_elbo_bwd_adam_lookahead_f32.argtypes = ([PP, C.POINTER(Index)] + [vp] * 14 +
                                                    [C.c_float, C.c_float, C.c_float, C.c_float, i64, vp, vp, vp, vp, vp])
    lib.vfm_variant_fwd_f32.argtypes = [PP, i32] + [vp] * 18
    lib.vfm_variant_bwd_f32.argtypes = [PP, i32, C.POINTER(Index)] + [vp] * 21
    lib.vfm_union_workspace_bytes.argtypes = [i64]
    lib.vfm_union_workspace_bytes.restype = i64
    lib.vfm_union_rows.argtypes = [i64] + [vp] * 7
    lib.vfm_variant_workspace_elems.argtypes = [i64, i32, i32]
    lib.vfm_variant_workspace_elems.restype = i64
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name != "vfm_last_error":
            fn.restype = C.c_int
    for name in ("vfm_index_workspace_bytes", "vfm_union_workspace_bytes", "vfm_variant_workspace_elems"):
        getattr(lib, name).restype = i64
    if lib.vfm_abi_version() != ABI_VERSION:
        raise VfmLibraryError(f"ABI mismatch: library {lib.vfm_abi_version()}, package {ABI_VERSION}")
    _lib = lib
    return lib


_ops = None


def ops():
    """`torch.ops.vfm_hip` -- the TORCH_LIBRARY shim over the same C ABI (libvfm_torch_ops.so).
    The per-step calls (forward, finalize, backward, Adam) go through it."""
    global _ops
    if _ops is not None:
        return _ops
    load()
    if not os.path.exists(OPS_PATH):
        raise VfmLibraryError(f"{OPS_PATH} not found: run __graft_entry__.build()")
    torch.ops.load_library(OPS_PATH)
    if torch.ops.vfm_hip.abi_version() != ABI_VERSION:
        raise VfmLibraryError("ABI mismatch between libvfm_torch_ops.so and the package")
    _ops = torch.ops.vfm_hip
    return _ops


def check(rc, what):
    if rc != 0:
        msg = load().vfm_last_error().decode(errors="replace")
        raise VfmLibraryError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def raw_stream(device) -> int:
    """The current HIP stream of `device` as an integer handle (torch.cuda.current_stream() builds a Python Stream object
    through three layers of device-index helpers: 4.5 us; a training step asked a dozen times)."""
    idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def current_stream_ptr(device):
    return C.c_void_p(raw_stream(device))
