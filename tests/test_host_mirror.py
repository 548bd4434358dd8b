"""Builds (g++, no GPU, no HIP) and runs the C++ checks of the host-side mirror of the reference interface."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd")


def test_cpp_host_mirror(tmp_path):
    exe = str(tmp_path / "test_host_mirror")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(PKG, "host"),
                           os.path.join(ROOT, "tests", "host", "test_host_mirror.cpp"),
                           os.path.join(PKG, "host", "Quadratures", "SphericalDesign.cpp"), "-o", exe])
    out = subprocess.run([exe, os.path.join(PKG, "data", "sph_design")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all host-mirror checks passed" in out.stdout
