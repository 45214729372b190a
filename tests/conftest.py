import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _torch_touches_the_gpu_first():
    """PyTorch bundles its own HIP runtime; libxmaps_hip.so links the system one.  Whichever is loaded first serves both
    (same SONAME), and torch only finds the GPU through its own copy: let torch initialise before the first handle is
    created (INTEGRATION.md says the same for hosts that share a process with PyTorch)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


def xm_option(name, value):
    """a variant switch of the library for the current test (removed again by the fixture below); value None removes it"""
    from x_maps_amd import _native as N
    N.debug_option(name, value)


@pytest.fixture(autouse=True)
def _no_library_options_leak_between_tests():
    yield
    try:
        from x_maps_amd import _native as N
        if os.path.exists(N.LIB_PATH):
            N.debug_option(None)
    except Exception:
        pass
