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
    config.addinivalue_line("markers", "long: full-length BASELINE configs[3] / [4] gates, minutes of encoder time each: opt-in with -m \"gpu and long\" or VVHIP_LONG_TESTS=1")


def pytest_collection_modifyitems(config, items):
    """`long` tests are opt-in: the default GPU suite (-m gpu) stays at ~5 minutes"""
    if "long" in (config.getoption("-m") or "") or os.environ.get("VVHIP_LONG_TESTS") == "1":
        return
    skip = pytest.mark.skip(reason="long gate: run with -m \"gpu and long\" or VVHIP_LONG_TESTS=1")
    for it in items:
        if "long" in it.keywords:
            it.add_marker(skip)


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
