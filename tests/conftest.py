import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Every infrastructure retry of the multi-process tests (tests/_util.py::retry_infra) is listed here: a retried test passed on its
    second attempt, but the first failure stays visible in the log the driver keeps."""
    try:
        from _util import RETRIES
    except Exception:      # pragma: no cover
        return
    if RETRIES:
        terminalreporter.section("infrastructure retries (tests/_util.py::retry_infra)")
        for name, reason in RETRIES:
            terminalreporter.write_line(f"RETRIED {name}: {reason}")
    else:
        terminalreporter.write_line("infrastructure retries: none")
