import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "sim: host-logic tests of shim + binding against the CPU test double of the device library (tests/sim)")
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libvvenc_ref.so (the compiled reference)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


def _reflib(simd):
    from oracle.oracle import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref/libvvenc_ref.so not built (needs /root/reference)")
    return RefLib(simd)


@pytest.fixture(scope="session", params=[0, 1], ids=["ref-scalar", "ref-simd"])
def reflib(request):
    return _reflib(request.param)
