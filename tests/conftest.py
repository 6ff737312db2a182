import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    from zksnake_amd import _native as N
    return N.load()


@pytest.fixture(scope="session")
def gpu():
    """the library with a selected HIP device; GPU tests fail loudly when the extension or the GPU is missing"""
    from zksnake_amd import _native as N
    return N.ensure_gpu()
