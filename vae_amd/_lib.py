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

ABI_VERSION = 4
MAX_FWD_BLOCKS = 4096
MAX_FIELDS = 64
N_PARTIALS = 8
MAX_FWD_BLOCKS = 4096
PARTIALS_LEN = N_PARTIALS * (1 + MAX_FWD_BLOCKS)
P_LL, P_KL, P_G, P_ALPHA, P_BADID = 0, 1, 2, 3, 4
LIK_NORMAL, LIK_BERNOULLI = 0, 1
OBJ_SAMPLED, OBJ_CLOSED_FORM = 0, 1

EXPORTS = (
    "vfm_abi_version", "vfm_last_error", "vfm_inv_occ_f32", "vfm_batch_norms",
    "vfm_elbo_fwd_f32", "vfm_elbo_finalize_f32", "vfm_elbo_bwd_f32", "vfm_philox_eps_f32",
    "vfm_adam_f32", "vfm_elbo_bwd_adam_f32", "vfm_elbo_bwd_acc_f32", "vfm_elbo_apply_adam_f32",
    "vfm_shard_sample_f32", "vfm_records_add_f32", "vfm_shard_pack_f32", "vfm_shard_loss_f32",
    "vfm_moments_rescale_f32", "vfm_elbo_lik_f32", "vfm_index_workspace_bytes", "vfm_build_index", "vfm_heavy_list_for",
    "vfm_variant_fwd_f32", "vfm_variant_bwd_f32", "vfm_variant_workspace_elems", "vfm_adam_catchup_f32", "vfm_union_rows", "vfm_union_workspace_bytes",
    "vfm_sample_records_f32", "vfm_elbo_bwd_adam_pipe_f32", "vfm_elbo_bwd_adam_lookahead_f32",
    "vfm_step_consts", "vfm_dev_step_set", "vfm_wrec_build_f32", "vfm_elbo_apply_adam_rows_f32",
    "vfm_elbo_bwd_acc_rows_f32", "vfm_heavy_threshold", "vfm_rebuild_heavy",
)


class Problem(C.Structure):
    """Mirror of `vfm_problem_t`."""
    _fields_ = [
        ("B", C.c_int64), ("B_global", C.c_int64), ("T", C.c_int64), ("nb_train", C.c_int64),
        ("F", C.c_int32), ("d", C.c_int32), ("likelihood", C.c_int32), ("id_bits", C.c_int32),
        ("n_samples", C.c_int32), ("flags", C.c_int32),
        ("group_hi", C.c_int64 * MAX_FIELDS), ("group_n", C.c_double * MAX_FIELDS),
        ("seed", C.c_uint64), ("step", C.c_uint64), ("e_lo", C.c_int64), ("e_hi", C.c_int64),
        ("own_mod", C.c_int32), ("own_rank", C.c_int32),
        ("coord_off", C.c_int32), ("reserved0", C.c_int32),
        ("dev_step", C.c_void_p), ("wrec", C.c_void_p),
    ]

    def __setattr__(self, name, value):
        # ctypes accepts ANY attribute name and silently keeps it as a Python attribute: a misspelt or
        # missing field would leave the struct member 0 without an error
        if name not in type(self)._names:
            raise AttributeError(f"vfm_problem_t has no field {name!r}")
        super().__setattr__(name, value)


Problem._names = frozenset(n for n, _ in Problem._fields_)


class Index(C.Structure):
    """Mirror of `vfm_index_t`."""
    _fields_ = [("occ_ptr", C.c_void_p), ("occ_rows", C.c_void_p), ("heavy_ids", C.c_void_p),
                ("heavy_items", C.c_void_p), ("heavy_acc", C.c_void_p), ("n_heavy", C.c_int32),
                ("n_items", C.c_int32), ("touched_ids", C.c_void_p), ("n_touched", C.c_int64),
                ("occ_other", C.c_void_p), ("max_items", C.c_int32)]


class StepConsts(C.Structure):
    """Mirror of `vfm_step_consts_t` (64 bytes: the constants of one Adam step, made by vfm_step_consts)."""
    _fields_ = [("step_size", C.c_float), ("bc2_sqrt", C.c_float), ("a1", C.c_float), ("q2", C.c_float),
                ("c1", C.c_float), ("c2", C.c_float), ("s1", C.c_float), ("s2", C.c_float),
                ("store_true", C.c_int32), ("k", C.c_int32), ("scaled", C.c_int32), ("reserved", C.c_int32),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


class DevStep(C.Structure):
    """Mirror of `vfm_dev_step_t` (64 bytes, lives in DEVICE memory: an int64[8] tensor)."""
    _fields_ = [("philox_step", C.c_uint64), ("adam_step", C.c_int64), ("philox_step_bwd", C.c_uint64),
                ("adam_step_bwd", C.c_int64), ("tab_first", C.c_int64), ("tab_len", C.c_int64), ("tab", C.c_void_p),
                ("error", C.c_int64)]


class Pipe(C.Structure):
    """Mirror of `vfm_pipe_t`."""
    _fields_ = [("zrec", C.c_void_p), ("zrec_next", C.c_void_p), ("next_occ_ptr", C.c_void_p), ("next_W", C.c_void_p),
                ("next_step", C.c_uint64), ("last_step", C.c_void_p), ("step_tab", C.c_void_p)]


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
                                            [C.c_float, C.c_float, C.c_float, C.c_float, i64, vp, vp, vp, vp])
    lib.vfm_elbo_bwd_acc_rows_f32.argtypes = [PP, C.POINTER(Index), vp, i64] + [vp] * 6
    lib.vfm_elbo_apply_adam_rows_f32.argtypes = ([PP, vp, vp, vp, i64, i32] + [vp] * 11 +
                                                 [C.c_float, C.c_float, C.c_float, C.c_float, i64, vp])
    lib.vfm_shard_sample_f32.argtypes = [PP, vp, i64, vp, vp, vp, vp, vp, vp]
    lib.vfm_records_add_f32.argtypes = [vp, vp, vp, i64, i32, i32, vp]
    lib.vfm_shard_pack_f32.argtypes = [vp, vp, vp, vp]
    lib.vfm_shard_loss_f32.argtypes = [vp, vp, vp]
    lib.vfm_elbo_lik_f32.argtypes = [PP, vp, vp, vp, vp, vp, vp, vp]
    lib.vfm_moments_rescale_f32.argtypes = [vp, vp, i64, C.c_float, C.c_float, i64, i32, vp]
    lib.vfm_index_workspace_bytes.argtypes = [i64, i32, i64]
    lib.vfm_heavy_list_for.argtypes = [i64, i64]
    lib.vfm_build_index.argtypes = [i64, i32, i64, i32, vp, vp, vp, vp, i32, vp, i64, vp, i64, vp, vp, vp, vp]
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
    lib.vfm_union_rows.argtypes = [i64] + [vp] * 6
    lib.vfm_variant_workspace_elems.argtypes = [i64, i32, i32]
    lib.vfm_variant_workspace_elems.restype = i64
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name != "vfm_last_error":
            fn.restype = C.c_int
    lib.vfm_index_workspace_bytes.restype = i64
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


def current_stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
