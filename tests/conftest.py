import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # debugging aid: a native backtrace on SIGABRT / SIGSEGV (tools/abrt_trace.c), so that a crash inside a native
    # library says where instead of only "Fatal Python error: Aborted"
    so = os.path.join(ROOT, "tools", "libabrt.so")
    if os.path.exists(so):
        try:
            import ctypes
            ctypes.CDLL(so)
        except OSError:
            pass


@pytest.fixture(scope="session")
def goldens():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "plan_goldens.json")) as f:
        g = json.load(f)
    return {c["name"]: c for c in g["cases"]}
