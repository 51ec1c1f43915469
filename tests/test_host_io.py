"""Host-side format helpers (config reader, KITTI listing, trajectory / loop-edge writers): C++ test, CPU only."""
import os
import subprocess

from conftest import ROOT


def test_host_io_formats(tmp_path):
    exe = str(tmp_path / "io_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "io_test.cpp"), "-o", exe])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "IO TEST OK" in r.stdout, r.stdout + r.stderr
