"""-m gpu: the C++ table-shaped host mirror (vvenc_amd/csrc/host) vs the oracle, through a C++ test written like the reference's unit test."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_shim_parity():
    exe = os.path.join(ROOT, "tests", "cpp", "test_shim")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert "shim parity OK" in r.stdout
