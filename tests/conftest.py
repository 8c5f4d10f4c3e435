import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/viterbi_oracle.c).  Test infrastructure: only tests may use it."""
    from oracle import pyoracle
    pyoracle.ensure_built()
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def reflib():
    """The real reference compiled in place (oracle/_ref/libvitref.so); absent => tests that need it skip."""
    from oracle import pyoracle
    pyoracle.ensure_built()
    if not pyoracle.RefLib.available():
        pytest.skip("oracle/_ref/libvitref.so not built (reference tree absent)")
    return pyoracle.RefLib()
