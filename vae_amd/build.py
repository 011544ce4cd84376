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
    out = os.path.join(HERE, "libvfm_hip.so")
    if not force and not _stale(out, [src, hdr]):
        return out
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-o", out, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_all(force=False, verbose=False):
    return [build_hip_library(force, verbose)]


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
