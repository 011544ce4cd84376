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
# The softplus link (the reference's own dead assignment, vfm-torch.py:125) is served by the GENERAL kernels only (k_fwd,
# k_bwd); the specialised forward kernels and the record step are built for |.| (:126, the assignment in effect).  vfm_bwd.hip
# is compiled in two parts per link (VFM_BWD_PART: gradient / statistics forms, fused-Adam forms) so that no single unit
# dominates the wall time of a clean build.
_UNITS = [("vfm_abi.hip", "", _EXACT), ("vfm_index.hip", "", []), ("vfm_variants.hip", "", []),
          ("vfm_fwd.hip", "_abs", ["-DVFM_LINK=0"]), ("vfm_fwd.hip", "_softplus", ["-DVFM_LINK=1"]),
          ("vfm_fwd2.hip", "_abs", ["-DVFM_LINK=0"]), ("vfm_fwd2m.hip", "_abs", ["-DVFM_LINK=0"]),
          ("vfm_fwdg.hip", "_abs", ["-DVFM_LINK=0"]),
          ("vfm_bwd.hip", "0_abs", ["-DVFM_LINK=0", "-DVFM_BWD_PART=0"] + _EXACT),
          ("vfm_bwd.hip", "1_abs", ["-DVFM_LINK=0", "-DVFM_BWD_PART=1"] + _EXACT),
          ("vfm_bwd.hip", "0_softplus", ["-DVFM_LINK=1", "-DVFM_BWD_PART=0"] + _EXACT),
          ("vfm_bwd.hip", "1_softplus", ["-DVFM_LINK=1", "-DVFM_BWD_PART=1"] + _EXACT)]


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
    shared = [hdr, os.path.abspath(__file__)] + [f for f in parts if f.endswith(".hpp")]      # what every unit depends on
    for src, suffix, extra in _UNITS:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + suffix + ".o")
        objs.append(obj)
        if not force and not _stale(obj, shared + [os.path.join(csrc, src)]):
            continue                      # (a change to one .hip recompiles that unit only)
        cmd = common + extra + ["-c", os.path.join(csrc, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    n_inst = 0                            # kernel instances in the library (host-side launch stubs of the objects)
    for obj in objs:
        try:
            n_inst += subprocess.run(["nm", "-C", obj], capture_output=True, text=True).stdout.count("__device_stub__")
        except OSError:
            pass
    print(f"libvfm_hip.so: {len(jobs)} of {len(objs)} translation units compiled, {n_inst} kernel instances", file=sys.stderr)
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


def sanitizer_runtime():
    """ROCm clang's shared ASan runtime (LD_PRELOAD it into an uninstrumented python to load the sanitized libraries)."""
    out = subprocess.run([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def build_sanitized(force=False, verbose=False):
    """The HOST side of both libraries under AddressSanitizer + UndefinedBehaviorSanitizer (`-Xarch_host
    -fsanitize=address,undefined`: the device code objects are the ordinary ones -- GPU sanitizers are not available on
    this pool), into vae_amd/build/san/.  What it covers: the argument checks, struct handling, host-side tables and
    launch set-up of csrc/vfm_abi.hip / vfm_index.hip / vfm_variants.hip and the TORCH_LIBRARY shim.  Load it with
    VFM_LIB_DIR=<returned dir> and LD_PRELOAD=sanitizer_runtime() (tests/test_sanitized_host.py does)."""
    import torch
    csrc = os.path.join(HERE, "csrc")
    outdir = os.path.join(HERE, "build", "san")
    os.makedirs(outdir, exist_ok=True)
    stamp = os.path.join(outdir, "digest.txt")
    digest = sources_digest()
    out = os.path.join(outdir, "libvfm_hip.so")
    out_ops = os.path.join(outdir, "libvfm_torch_ops.so")
    if (not force and os.path.exists(out) and os.path.exists(out_ops) and os.path.exists(stamp)
            and open(stamp).read().strip() == digest):
        return outdir
    san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer",
           "-Xarch_host", "-fno-sanitize-recover=undefined", "-Xarch_device", "-O0"]     # (device code: never run from here)
    common = [HIPCC, "-O1", "-g", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility-inlines-hidden",
              "-I" + os.path.join(ROOT, "include"), "-I" + csrc] + san
    jobs, objs = [], []
    for src, suffix, extra in _UNITS:
        obj = os.path.join(outdir, os.path.splitext(src)[0] + suffix + ".o")
        cmd = common + extra + ["-c", os.path.join(csrc, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan",
                    "-o", out] + objs, check=True)
    ti = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    clangxx = os.path.join(rocm, "lib", "llvm", "bin", "clang++")
    cmd = [clangxx, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-shared-libsan",
           "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-fno-sanitize=vptr",
           "-mllvm", "-asan-globals=0",    # (libstdc++'s merged string literals trip the globals check: "not properly aligned")
           "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ti, "include"),
           "-I" + os.path.join(ti, "include", "torch", "csrc", "api", "include"),
           "-I" + os.path.join(rocm, "include"), "-o", out_ops, os.path.join(csrc, "vfm_torch_ops.cpp"),
           "-L" + os.path.join(ti, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           "-L" + outdir, "-lvfm_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ti, "lib")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    # the native example: compiled (not run: it needs a GPU) under the same sanitizers -- its host code is UB-checked
    # by the compiler's diagnostics at least
    subprocess.run([HIPCC, "-O1", "-g", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include")] + san +
                   ["-c", os.path.join(ROOT, "examples", "c_abi_step.cpp"), "-o", os.path.join(outdir, "c_abi_step.o")], check=True)
    open(stamp, "w").write(digest)
    return outdir


def build_all(force=False, verbose=False):
    return [build_hip_library(force, verbose), build_torch_ops(force, verbose)]


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
