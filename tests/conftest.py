import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True, scope="session")
def _guard_pages():
    """PACE_GUARD_MODE=over|under: every CPU tensor the host layer allocates ends at / starts after an inaccessible page
    (tests/guard.py), so an out-of-bounds access of an emulated kernel faults.  Used by tests/test_guard_pages.py, which runs
    part of the emulation suite in a child process in both modes."""
    mode = os.environ.get("PACE_GUARD_MODE")
    if not mode:
        yield
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import guard

    with guard.guarded(mode):
        yield
