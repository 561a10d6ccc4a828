import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A per-test wall-clock cap (pytest-timeout, where installed): a persistent kernel that waits for a hand-off that never comes must end the
    run with a stack dump, not hold the GPU box until somebody else's limit kills it.  The slowest test takes ~20 s."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(900, method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
