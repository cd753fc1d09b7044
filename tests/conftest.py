import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """build (or reuse) the HIP engine and the oracle; returns the dict of artefact paths"""
    from rwkv_cpp_accelerated_amd import build
    return build.build_all()


@pytest.fixture(scope="session")
def oracle(built):
    import oracle_lib
    return oracle_lib.Oracle()
