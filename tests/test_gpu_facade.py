"""Builds the C++ facade test (reference class names over the C ABI) and runs it on the GPU."""
import os
import subprocess

import pytest

from conftest import PKG_DIR, ROOT

pytestmark = pytest.mark.gpu


def test_cpp_facade_matches_oracle(api, oracle, tmp_path):
    exe = str(tmp_path / "facade_test")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-o", exe,
           "-L" + PKG_DIR, "-lmyslam_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
           "-Wl,-rpath," + PKG_DIR, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FACADE TEST OK" in r.stdout, r.stdout + r.stderr


def test_c_abi_gated_handles(tmp_path):
    """INTEGRATION.md's two-handle pipeline through the plain C ABI + HIP runtime API (no Python in the loop)."""
    exe = str(tmp_path / "pipeline_test")
    cmd = ["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "pipeline_test.cpp"), "-o", exe, "-L" + PKG_DIR, "-lmyslam_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + PKG_DIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PIPELINE TEST OK" in r.stdout, r.stdout + r.stderr
