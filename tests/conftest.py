import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# a process that dies in abort() under pytest's fd capture leaves only the signal behind: the library appends the native backtrace here
# (csrc/ivfadc_hip.hip, abort_trace_handler); gpurun_out/ travels back from the GPU box
_trace_dir = os.path.join(ROOT, "gpurun_out")
try:
    os.makedirs(_trace_dir, exist_ok=True)
    os.environ.setdefault("IVFADC_ABORT_TRACE", os.path.join(_trace_dir, "abort_trace.txt"))
except OSError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native():
    """The HIP library, built in-tree; GPU tests fail loudly if it cannot be loaded."""
    import ivfadc_jl_amd as pkg
    pkg.build_library()
    pkg.load_library()
    return pkg
