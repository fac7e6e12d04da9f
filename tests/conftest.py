import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# An index builds its presence filters and position-sorted lists on SECOND use by default (option lazy_aux: they do not
# pay back within one pass).  The suite wants every first call on the filtered / list-assisted paths -- that is where
# the code is -- so it switches the laziness off; test_lazy_filter_and_lists_same_results covers the default.
os.environ.setdefault("ASGART_LAZY_AUX", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_collection_modifyitems(config, items):
    """A test that hangs (a kernel that never returns, a lost rendezvous) ends the run with the stacks of every thread
    instead of stalling it: pytest-timeout, when it is installed, bounds each test ("thread" method: a main thread
    blocked inside a HIP call never returns to the interpreter, so a signal handler would not run)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900, method="thread"))


@pytest.fixture(scope="session")
def hiplib():
    """The product library.  GPU tests must exercise it, never a fallback."""
    import asgart_amd

    return asgart_amd.load_library()
