import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Every test process (and the subprocesses it starts: the simka driver, the cross-check scripts) runs with the library's fault trace on
# (simka_amd/csrc/simka_trace.h): should a GPU memory access fault end one of them, the registry of device ranges and the last launches
# land in gpurun_out/fault_trace/ (merged back from the GPU box) and scripts/fault_resolve.py names the buffer and the kernels.
os.environ.setdefault("SIMKA_FAULT_TRACE", "1")
_TRACE_DIR = os.path.join(ROOT, "gpurun_out", "fault_trace")
try:
    os.makedirs(_TRACE_DIR, exist_ok=True)
    os.environ.setdefault("SIMKA_FAULT_TRACE_DIR", _TRACE_DIR)
except OSError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure): built from oracle/ with its Makefile, loaded through ctypes."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def simka_lib():
    import simka_amd
    return simka_amd.load_library()


@pytest.fixture(scope="session")
def gpu_required():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu and needs a HIP device (no CPU fallback exists)")
    return torch
