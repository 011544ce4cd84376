"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip_library(force=False, verbose=False):
    """libvfm_hip.so: kernels + C ABI (include/vfm_hip.h), gfx950 only."""
    src = os.path.join(HERE, "csrc", "vfm_kernels.hip")
    hdr = os.path.join(ROOT, "include", "vfm_hip.h")
    parts = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc")))
             if f.endswith(".hpp")]             # the kernels, included by vfm_kernels.hip (one TU)
    out = os.path.join(HERE, "libvfm_hip.so")
    if not force and not _stale(out, [src, hdr] + parts):
        return out
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-o", out, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_torch_ops(force=False, verbose=False):
    """libvfm_torch_ops.so: TORCH_LIBRARY shim `torch.ops.vfm_hip.*` over the C ABI (host C++ only)."""
    import torch
    src = os.path.join(HERE, "csrc", "vfm_torch_ops.cpp")
    hdr = os.path.join(ROOT, "include", "vfm_hip.h")
    out = os.path.join(HERE, "libvfm_torch_ops.so")
    lib = os.path.join(HERE, "libvfm_hip.so")
    if not force and not _stale(out, [src, hdr, lib]):
        return out
    ti = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ti, "include"),
           "-I" + os.path.join(ti, "include", "torch", "csrc", "api", "include"),
           "-I" + os.path.join(rocm, "include"), "-o", out, src,
           "-L" + os.path.join(ti, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           "-L" + HERE, "-lvfm_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ti, "lib")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_all(force=False, verbose=False):
    return [build_hip_library(force, verbose), build_torch_ops(force, verbose)]


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
