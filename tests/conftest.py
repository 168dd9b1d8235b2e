import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the CPU-side libraries exist (the HIP ones are built by
    __graft_entry__.build() and travel to the GPU box prebuilt)."""
    import subprocess
    need = [os.path.join(ROOT, "handbrake_amd", "libhbrt.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "handbrake_amd/libhbrt.so", "oracle"],
                              stdout=subprocess.DEVNULL)
    return True
