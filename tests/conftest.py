import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the scorers split only windows of more than 512 K distinct hits over several workgroups; the test pools stay far below, so the tests
# run with the split at 128 hits (test_deep_windows_many_slices_and_multiplicities asserts that it happens).  Read once per process.
os.environ.setdefault("VDJX_HIT_CHUNK", "128")
# ... and windows are mapped in groups of eight (k_group_pairs) from 4,096 windows up only (below, one by one is faster): the suite's
# pools have tens of windows, so it groups from the first one -- the tests that run the scorers "both ways" set VDJX_WINDOW_GROUP=0 for
# the other way; bench.py and the production-slicing test run with the shipped threshold
os.environ.setdefault("VDJX_GROUP_MIN", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """a fresh checkout has no native libraries (they are built artefacts, not history): build what is missing once"""
    need = [os.path.join(ROOT, "vdjer_amd", "libvdjx.so"), os.path.join(ROOT, "vdjer_amd", "libvdjhost.so"),
            os.path.join(ROOT, "vdjer_amd", "vdjer"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    import subprocess
    for d, target in ((os.path.join("vdjer_amd", "csrc"), None), (os.path.join("vdjer_amd", "csrc", "host"), None), ("oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, d)] + ([target] if target else []))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    return oracle.lib()
