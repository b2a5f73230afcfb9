import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # The CPU oracle (torch fp32 convolutions at batch 4) is the long pole of the parity tests, and torch's default of one thread
    # per core is the wrong setting on the MI355X box's 256-thread host: profiles/r06_cpu_threads.txt -- 16 threads 4.4 s per
    # iteration, 128 threads 25.8 s, 256 threads 330 s.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so it is imported by string)."""
    return importlib.import_module("fast-srgan_amd")


def load_npz(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))


def sd_from(npz, prefix):
    import torch
    return {k[len(prefix):]: torch.from_numpy(npz[k].copy()) for k in npz.files if k.startswith(prefix)}


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The suite needs libfsr_hip.so (ABI checks here, everything with -m gpu): build it in-tree if it is missing
    (hipcc cross-compiles gfx950 without a GPU; about a minute).  The package itself never builds implicitly."""
    lib = os.path.join(ROOT, "fast-srgan_amd", "libfsr_hip.so")
    if not os.path.exists(lib):
        importlib.import_module("fast-srgan_amd.build").build_hip(verbose=False)
    yield
