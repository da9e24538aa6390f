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


@pytest.fixture(autouse=True, scope="session")
def _suite_engine_options():
    """Engines of the suite run with the events between the phases on and the seeds / early exit at ANY job size
    (defaults: both only above a work threshold, because they cost a small job more than they give): most scenarios
    here are small, and the tests read phase times and work counters to see WHICH path ran.  The defaults themselves
    are exercised by tests/test_gpu_small_jobs.py, the option matrix of tests/test_gpu_fuzz.py and the
    "shipped_defaults" leg of the COMPARE matrix (tests/test_gpu_generator.py)."""
    try:
        from mosaicmagnifique_amd import engine as E
    except Exception:
        yield
        return
    orig = E.Engine.__init__

    def init(self, device=0):
        orig(self, device)
        self.set_option("phase_timing", 1)
        self.set_option("prune_min_kpairs", 0)
        # soak runs: MOSAIC_TEST_OPTS="pretier=1,audit=1" puts every engine of the session on a non-default path
        for kv in filter(None, os.environ.get("MOSAIC_TEST_OPTS", "").split(",")):
            name, value = kv.split("=")
            self.set_option(name.strip(), int(value))
        self.options.clear()

    E.Engine.__init__ = init
    try:
        yield
    finally:
        E.Engine.__init__ = orig
