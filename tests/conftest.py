import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi
    return oracle_ffi.load()


@pytest.fixture(scope="session")
def engine():
    """One HIP engine on cuda:0 for the whole GPU session (fails loudly if the
    library is not built or no device is visible)."""
    from mosaicmagnifique_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
