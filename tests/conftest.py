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


def _session_jit_cache():
    """tests/_jit_cache/ (filled by build() from tests/jit_codes.txt) -> a private copy that serves as this session's USER cache
    of run-time compiled kernels (the library only trusts a directory and files that belong to this user: csrc/reg_jit.hpp)"""
    import atexit
    import glob
    import shutil
    import tempfile

    if os.environ.get("VIT_HIP_CACHE_DIR"):
        return
    src = os.path.join(ROOT, "tests", "_jit_cache")
    d = tempfile.mkdtemp(prefix="vit_hip_test_cache-")
    os.chmod(d, 0o700)
    for f in glob.glob(os.path.join(src, "*.hsaco")):
        dst = os.path.join(d, os.path.basename(f))
        shutil.copyfile(f, dst)
        os.chmod(dst, 0o600)
    os.environ["VIT_HIP_CACHE_DIR"] = d
    atexit.register(shutil.rmtree, d, ignore_errors=True)


def pytest_sessionstart(session):
    _session_jit_cache()


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
