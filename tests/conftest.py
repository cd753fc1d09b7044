import os
import sys

import pytest

# PyTorch-ROCm bundles its own libamdhip64; librwkv_mi355x.so links the system one (same SONAME).  Whichever is
# loaded first serves the whole process, and torch's device initialisation only works on its own copy -- so
# torch is imported before any test can load the engine, whatever subset of files is collected.
try:
    import torch  # noqa: F401
except Exception:  # the CPU suite does not need it everywhere
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """build (or reuse) the HIP engine and the oracle; returns the dict of artefact paths"""
    import build_checkers
    return build_checkers.build_all()


@pytest.fixture(scope="session")
def oracle(built):
    import oracle_lib
    return oracle_lib.Oracle()
