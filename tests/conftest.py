import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def nttx():
    """The product library, initialised on GPU 0.  Fails (does not skip) without a GPU:
    there is no CPU fallback to fall back to."""
    import ntt_amd
    ntt_amd.init(0)
    yield ntt_amd
    ntt_amd.shutdown()
