import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a GPU test that also runs minutes of float64 oracle on the host (still part of -m gpu)")
    # property tests must not replay (or depend on) an example database that is not part of the tree
    try:
        from hypothesis import settings
        settings.register_profile("xt", database=None, deadline=None)
        settings.load_profile("xt")
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
