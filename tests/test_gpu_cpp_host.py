"""Builds and runs the C++ host mirror's test program (tests/cpp/test_bellman_host.cpp against
phase2-bn254_amd/host/{bellman,ceremony,prover}.hpp -> libmi355zk.so, checked with the oracle) on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "test_bellman_host")


def _build():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    import torch  # the binary must resolve libamdhip64 the same way the python process does

    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "test_bellman_host.cpp"), "-o", BIN,
           "-L" + os.path.join(ROOT, "phase2-bn254_amd"), "-lmi355zk", "-L" + os.path.join(ROOT, "oracle", "_build"), "-loracle",
           "-Wl,-rpath," + os.path.join(ROOT, "phase2-bn254_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build"), "-lpthread"]
    subprocess.check_call(cmd)


def test_cpp_host_program_compiles_on_cpu():
    _build()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_host_mirror_against_oracle():
    _build()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    for name in ("multiexp_equals_naive", "density_and_source_errors", "multi_device_cells", "evaluation_domain", "ceremony_mirror", "groth16_create_proof"):
        assert "ok " + name in out.stdout
