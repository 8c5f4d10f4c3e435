import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    # a clean checkout has no built artefacts (they are git-ignored): build them once, exactly as the driver's build() does
    need = [os.path.join(ROOT, "viterbidecodercpp_amd", "libvit_hip.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


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
