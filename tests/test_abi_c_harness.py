"""The C ABI from plain C: tests/abi_smoke.c is compiled with gcc against include/colnde.h alone and linked to libcolnde.so.
On CPU the harness must build (and fail loudly at colnde_create: no device); on the GPU box it runs create -> set_problem ->
forward -> loss_grad on the golden vectors, for both engines."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "climateparameterizations.jl_amd")


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", os.path.join(ROOT, "tests", "abi_smoke.c"),
                           "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lcolnde", "-lm",
                           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def _dump(tmp_path, golden_dir):
    z = np.load(os.path.join(golden_dir, "wind_mixing_mpp.npz"))
    for k in ("x0", "bcs", "weights", "truth", "scalings", "sol", "grad", "total"):
        np.ascontiguousarray(z[k], dtype=np.float32).reshape(-1).tofile(str(tmp_path / (k + ".f32")))
    return str(tmp_path)


def test_c_harness_builds_and_refuses_to_run_without_a_gpu(tmp_path, golden_dir):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the run is covered by the gpu-marked test")
    r = subprocess.run([exe, _dump(tmp_path, golden_dir)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "colnde_create" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("engine", [1, 2])
def test_c_harness_matches_golden_on_gpu(tmp_path, golden_dir, engine):
    exe = _build(tmp_path)
    r = subprocess.run([exe, _dump(tmp_path, golden_dir), str(engine)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "abi_smoke: OK" in r.stdout and ("engine %d" % engine) in r.stdout
