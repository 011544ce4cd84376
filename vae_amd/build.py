"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def sources_digest() -> str:
    """sha1 over the kernel sources (csrc/*.hip, *.hpp, *.cpp + include/vfm_hip.h), in name order: ties measured
    figures (profiles/latest_traffic.json) to the code they were measured on."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".cpp")))
    files.append(os.path.join(os.path.dirname(here), "include", "vfm_hip.h"))
    h = hashlib.sha1()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


# translation units of libvfm_hip.so: (source, object suffix, extra flags).  The row kernels are
# compiled once per link function (vfm-torch.py:125-126: |.| and softplus).
# -ffp-contract=on (a*b+c fuses only inside one source expression) for the units holding the update arithmetic: under
# hipcc's default ("fast": fusion across statements, decided after inlining) two template instances of k_bwd rounded
# the same source differently in the last bit, and the lazy / look-ahead step forms are specified as BITWISE the dense one.
_EXACT = ["-ffp-contract=" + os.environ.get("VFM_FP_CONTRACT", "on")]      # (the variable: A/B builds only)
_UNITS = [("vfm_abi.hip", "", _EXACT), ("vfm_index.hip", "", []), ("vfm_variants.hip", "", []),
          ("vfm_fwd.hip", "_abs", ["-DVFM_LINK=0"]), ("vfm_fwd.hip", "_softplus", ["-DVFM_LINK=1"]),
          ("vfm_fwd2.hip", "_abs", ["-DVFM_LINK=0"]), ("vfm_fwd2.hip", "_softplus", ["-DVFM_LINK=1"]),
          ("vfm_fwd2m.hip", "_abs", ["-DVFM_LINK=0"]), ("vfm_fwd2m.hip", "_softplus", ["-DVFM_LINK=1"]),
          ("vfm_fwdg.hip", "_abs", ["-DVFM_LINK=0"]), ("vfm_fwdg.hip", "_softplus", ["-DVFM_LINK=1"]),
          ("vfm_bwd.hip", "_abs", ["-DVFM_LINK=0"] + _EXACT), ("vfm_bwd.hip", "_softplus", ["-DVFM_LINK=1"] + _EXACT)]


def build_hip_library(force=False, verbose=False):
    """libvfm_hip.so: kernels + C ABI (include/vfm_hip.h), gfx950 only.  The translation units are
    compiled in parallel (hipcc -c) and linked into one shared object."""
    csrc = os.path.join(HERE, "csrc")
    hdr = os.path.join(ROOT, "include", "vfm_hip.h")
    parts = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hpp", ".hip"))]
    out = os.path.join(HERE, "libvfm_hip.so")
    if not force and not _stale(out, [hdr, os.path.abspath(__file__)] + parts):      # (this file holds the compile flags)
        return out
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility-inlines-hidden",
              "-I" + os.path.join(ROOT, "include"), "-I" + csrc]
    jobs, objs = [], []
    for src, suffix, extra in _UNITS:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + suffix + ".o")
        cmd = common + extra + ["-c", os.path.join(csrc, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
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
