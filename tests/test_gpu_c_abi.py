"""GPU: the C ABI driven from native code alone (examples/c_abi_step.cpp: include/vfm_hip.h + the HIP
runtime, no Python / PyTorch in the process) reproduces the Python host path step for step."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_host_program_matches_python_path(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "c_abi_step"
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_step.cpp"), "-L" + os.path.join(ROOT, "vae_amd"), "-lvfm_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "vae_amd"), "-o", str(exe)], check=True)
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    N, M, d, B, n_steps, lr, seed = 300, 200, 16, 4000, 5, 0.05, 77
    X, y = synthetic_triples([N, M], B, seed=3)
    torch.manual_seed(1)
    m = VFM(N, M, d, device="cuda", rng_seed=seed)
    m.set_training_data(X, nb_train=20000)
    # inputs for the native program
    (tmp_path / "meta.txt").write_text(f"{B} {N} {M} {d} 20000 0 {seed}\n")
    X.cpu().numpy().astype(np.int64).tofile(tmp_path / "x.i64")
    y.cpu().numpy().astype(np.float32).tofile(tmp_path / "y.f32")
    m.nb_occ.cpu().numpy().astype(np.int64).tofile(tmp_path / "nb_occ.i64")
    m.entity_params.weight.detach().cpu().numpy().tofile(tmp_path / "entity.f32")
    m.bias_params.weight.detach().cpu().numpy().tofile(tmp_path / "bias.f32")
    m._scalars().detach().cpu().numpy().tofile(tmp_path / "scalars.f32")
    out = subprocess.run([str(exe), str(tmp_path), str(n_steps), str(lr)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    # the same steps through the Python host path
    plan = m.plan(X, y)
    ref = []
    for s in range(n_steps):
        loss3, _ = m.train_step(plan, lr=lr)
        ref.append(loss3.cpu().numpy().copy())
    got = np.fromfile(tmp_path / "losses.f32", dtype=np.float32).reshape(n_steps, 3)
    assert np.array_equal(got, np.array(ref))                      # same kernels, same arguments: same bits
    ent = np.fromfile(tmp_path / "entity_out.f32", dtype=np.float32).reshape(N + M, 2 * d)
    assert np.array_equal(ent, m.entity_params.weight.detach().cpu().numpy())
