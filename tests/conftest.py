import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """Without a GPU (the CPU tier: oracle, host logic, emulated kernels -- five six-tile DynamicalCore steps under emulation at
    ~50 s each) the tests are spread over four worker processes (pytest-xdist) unless -n / PACE_TEST_WORKERS says otherwise:
    8 min -> 2.5 min.  On a machine with a GPU nothing changes: the -m gpu tests share one device and run one at a time."""
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return None  # this IS a worker process (it must never start workers of its own)
    try:
        import xdist  # noqa: F401
    except ImportError:
        return None
    if getattr(config.option, "numprocesses", None) is not None or not hasattr(config.option, "numprocesses"):
        return None
    want = os.environ.get("PACE_TEST_WORKERS")
    if want is not None:
        n = int(want)
    else:
        try:
            import torch

            if torch.cuda.device_count() > 0:  # (counting devices does not initialise the GPU)
                return None
        except Exception:  # noqa: BLE001
            pass
        n = max(1, min(4, (os.cpu_count() or 2) // 2))
    if n > 1:
        config.option.numprocesses = n
    return None


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
