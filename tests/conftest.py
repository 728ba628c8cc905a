import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import ctypes as C
        from refactored_orb_slam2_amd import _lib
        n = C.c_int(0)
        return _lib.lib().orbfe_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: only auto-skip when gpu tests were not asked for
    if "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="needs a GPU (-m gpu)")
    for item in items:
        if "gpu" in item.keywords and not _has_gpu():
            item.add_marker(skip)
